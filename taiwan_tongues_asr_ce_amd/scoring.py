"""Transcript post-processing and character-error-rate scoring — host-side text work that sits
after the hot path in the reference's batch tool.

Restates, with the same observable results (pinned by tests/golden/text.json, generated from the reference's own
functions by oracle/make_golden_text.py):

* `normalise_transcript`  — asr_core.py:32-78 + :176-178: fixed phrase→numeral mappings, punctuation removal,
  per-character NFKC folding, lower-casing.  (The reference also runs OpenCC `s2tw`; OpenCC is optional here:
  used when importable, otherwise the text passes through unchanged and `normalise_transcript.opencc` is False.)
* `digits_to_chinese`, `clean_for_scoring` — cer.py:41-143: digit runs → Chinese numerals (including the
  reference's quirks, e.g. "10" → "一十", "120000" → "一十萬二萬"), homophone folding, keep only CJK
  U+4E00–U+9FA5 and ASCII letters.
* `score` / `compare_texts` — cer.py:146-312: alignment by `difflib.SequenceMatcher` opcodes (NOT minimum edit
  distance — the counts depend on difflib's matching heuristics, so difflib is part of the contract), error
  counts, human-readable error lists and the bracket-annotated texts, in a `CERResult` with the same attributes.
* `srt_time`, `split_cjk_words` — asr_core.py:22-58.
"""
from __future__ import annotations

import difflib
import re
import unicodedata
from dataclasses import dataclass, field
from typing import List, Optional

# ---- asr_core.py post-processing --------------------------------------------------------------------------

_PHRASE_TO_NUMERAL = (
    ("百分之十五", "15%"), ("百分之五", "5%"), ("百分之十二點五", "12.5%"), ("百分之七", "7%"),
    ("零八零零零九五九八", "080009598"),
)
# characters the reference strips before NFKC folding (asr_core.py:72)
_STRIP = ",\"'。，^¿¡；「」《》:：＄$[]〜～·・‧―─–－⋯、＼【】=<>{}_〈〉　）（—『』«»→„…()`&＆﹁﹂#＃\\!?！;"
_STRIP_TABLE = {ord(ch): None for ch in _STRIP}


def _s2tw(text: str) -> str:
    conv = getattr(_s2tw, "_conv", None)
    if conv is None:
        try:
            import opencc  # type: ignore
            conv = opencc.OpenCC("s2tw").convert
        except Exception:
            conv = False
        _s2tw._conv = conv
    return conv(text) if conv else text


def opencc_available() -> bool:
    _s2tw("")
    return bool(_s2tw._conv)


def normalise_transcript(text: str, to_traditional: bool = True) -> str:
    """What asr_core.py:176-178 writes to `<name>_asr.txt`."""
    for phrase, numeral in _PHRASE_TO_NUMERAL:       # sequential: "百分之五" is applied after "百分之十五"
        text = text.replace(phrase, numeral)
    if to_traditional:
        text = _s2tw(text)
    text = text.translate(_STRIP_TABLE)
    return "".join(unicodedata.normalize("NFKC", ch) for ch in text).lower()


def srt_time(seconds: float) -> str:
    """HH:MM:SS.mmm, rounded to the millisecond; hours wrap at 24 like the reference's datetime arithmetic."""
    whole, _, frac = f"{seconds:.3f}".partition(".")
    s = int(whole)
    return f"{(s // 3600) % 24:02d}:{(s // 60) % 60:02d}:{s % 60:02d}.{frac or '000'}"


_CJK_SPLIT = re.compile("([\u1100-\u11ff\u2e80-\ua4cf\ua840-\ud7af\uf900-\ufaff\ufe30-\ufe4f\uff65-\uffdc"
                        "\U00020000-\U0002ffff%]|\\d+\\.\\d+|\\d+)")


def split_cjk_words(text: str, split: bool = True) -> str:
    """Space-separate CJK characters / numbers / latin words (the WER tokenisation of asr_core.py:22-29)."""
    if not split:
        return text
    return " ".join(p.strip() for p in _CJK_SPLIT.split(text.strip().lower()) if p and p.strip())


# ---- cer.py scoring ---------------------------------------------------------------------------------------

_NUMERALS = "零一二三四五六七八九"
_PLACE = ("", "十", "百", "千", "萬", "十萬", "百萬", "千萬", "億")
_HOMOPHONES = (("她", "他"), ("它", "他"), ("臺", "台"), ("得", "的"))
_KEEP = re.compile("[^\u4e00-\u9fa5a-zA-Z]")
_DIGITS = re.compile(r"\d+")


def digits_to_chinese(digits: str) -> str:
    """One run of ASCII digits → Chinese numerals, the reference's way (cer.py:41-88)."""
    if (len(digits) > 1 and digits[0] == "0") or len(digits) > 9:
        return "".join(_NUMERALS[int(d)] for d in digits)          # spelled digit by digit
    try:
        digits = str(int(digits))
    except ValueError:
        return ""
    parts: List[str] = []
    gap = False
    n = len(digits)
    for pos, d in enumerate(digits):
        if d == "0":
            gap = True
            continue
        if gap:
            parts.append(_NUMERALS[0])
            gap = False
        parts.append(_NUMERALS[int(d)] + _PLACE[n - 1 - pos])
    if not parts:
        return _NUMERALS[0]
    if len(parts) == 2 and parts[0] == "一十":                     # 11..19 → 十一..十九 (but "10" stays 一十)
        parts[0] = "十"
    return "".join(parts)


def clean_for_scoring(text: str) -> str:
    text = text.replace("\n", "").replace("\r", "")
    for a, b in _HOMOPHONES:
        text = text.replace(a, b)
    text = _DIGITS.sub(lambda m: digits_to_chinese(m.group(0)), text)
    return _KEEP.sub("", text).lower()


@dataclass
class CERResult:
    reference_text: str
    hypothesis_text: str
    reference_cleaned: str = ""
    hypothesis_cleaned: str = ""
    correct_rate: float = 0.0
    cer_rate: float = 0.0
    total_errors: int = 0
    substitutions_count: int = 0
    deletions_count: int = 0
    insertions_count: int = 0
    total_chars: int = 0
    substitutions_errors: List[str] = field(default_factory=list)
    deletions_errors: List[str] = field(default_factory=list)
    insertions_errors: List[str] = field(default_factory=list)
    reference_highlighted: str = ""
    hypothesis_highlighted: str = ""

    def as_dict(self) -> dict:
        """The `cer_result` object of asr_comparison_results.json (asr_core.py:207-220)."""
        keys = ("correct_rate", "cer_rate", "total_errors", "substitutions_count", "deletions_count", "insertions_count",
                "total_chars", "substitutions_errors", "deletions_errors", "insertions_errors", "reference_highlighted",
                "hypothesis_highlighted")
        return {k: getattr(self, k) for k in keys}


_BREAK_EVERY = 250   # a blank line in the annotated texts once this many aligned characters have been emitted
_HOLE = "□"


def score(reference: str, hypothesis: str) -> CERResult:
    res = CERResult(reference, hypothesis)
    ref = res.reference_cleaned = clean_for_scoring(reference)
    hyp = res.hypothesis_cleaned = clean_for_scoring(hypothesis)
    ref_marks: List[str] = []
    hyp_marks: List[str] = []
    since_break = 0
    for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, ref, hyp).get_opcodes():
        r, h = ref[i1:i2], hyp[j1:j2]
        if tag == "equal":
            ref_marks.append(r)
            hyp_marks.append(h)
        else:
            paired = min(len(r), len(h)) if tag == "replace" else 0
            lost, extra = r[paired:], h[paired:]
            if paired:
                res.substitutions_count += paired
                res.substitutions_errors.append(f"正確文本中的「{r}」 在 ASR 轉譯文本中被替換成 「{h}」")
                ref_marks.extend(f"[{ch}]" for ch in r[:paired])
                hyp_marks.extend(f"[{ch}]" for ch in h[:paired])
            if lost:
                res.deletions_count += len(lost)
                res.deletions_errors.append(f"正確文本中的「{lost}」 被刪除，未被 ASR 轉譯成功 (替換造成)" if paired
                                            else f"正確文本中的「{lost}」 被刪除 ，未被 ASR 轉譯成功")
                ref_marks.extend(f"<{ch}>" for ch in lost)
                hyp_marks.append(_HOLE * len(lost))
            if extra:
                res.insertions_count += len(extra)
                res.insertions_errors.append(f"「{extra}」 在 ASR 結果 額外輸出，不屬於正確文本內容 (替換造成)" if paired
                                             else f"「{extra}」 在 ASR 結果 額外輸出，不屬於正確文本內容")
                hyp_marks.extend(f"({ch})" for ch in extra)
                ref_marks.append(_HOLE * len(extra))
        since_break += (i2 - i1) + (j2 - j1)
        if since_break >= _BREAK_EVERY:
            ref_marks.append("\n\n")
            hyp_marks.append("\n\n")
            since_break = 0
    res.total_chars = len(ref)
    res.total_errors = res.substitutions_count + res.deletions_count + res.insertions_count
    res.cer_rate = res.total_errors / res.total_chars if res.total_chars else 0
    res.correct_rate = 100 * (1 - res.cer_rate)
    res.reference_highlighted = "".join(ref_marks)
    res.hypothesis_highlighted = "".join(hyp_marks)
    return res


def compare_texts(reference_text: str, hypothesis_text: str) -> Optional[CERResult]:
    """cer.py:300-312: None when either side is empty."""
    if not reference_text or not hypothesis_text:
        return None
    return score(reference_text, hypothesis_text)

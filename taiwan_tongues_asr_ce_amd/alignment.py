"""Word-level timestamps: `WhisperModel.transcribe(word_timestamps=True)` (requested by the streaming service's
warm-up, faster_whisper_asr.py:289-294; `segment.words[i].{word,start,end,probability}` is what
faster_whisper_asr.py:225-253 would read).

The device part — a teacher-forced decoder pass that returns the cross-attention rows of the alignment heads and the
raw token log-probabilities — is `ttasr_align`.  This module is the host part that faster-whisper does in Python
around CTranslate2's `Whisper.align` (un-vendored; the published algorithm is OpenAI Whisper's `timing.py`, restated
by HF `_extract_token_timestamps`, which is what tests/golden/align.npz pins):

    heads' attention [n_heads][n_tok][n_ctx] -> crop to the clip's frames -> per-head normalisation over the token axis
    -> median filter (width 7, reflect) along time -> mean over heads -> rows <|notimestamps|>, text... (the row of a fed
    token locates the token it predicts) -> DTW on the negated matrix (`ttasr_dtw`, C++) -> a token starts where the
    path first enters its row -> tokens grouped into words (unicode-complete pieces for zh/ja/th/lo/my/yue, spaces
    otherwise) -> punctuation merged into neighbours -> over-long words next to sentence ends clamped to twice the
    median duration (faster-whisper `add_word_timestamps`).
"""
from __future__ import annotations

import ctypes as C
import string
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

TOKENS_PER_SECOND = 50.0
NO_SPACE_LANGUAGES = {"zh", "ja", "th", "lo", "my", "yue"}
PREPEND_PUNCTUATIONS = "\"'“¿([{-"
APPEND_PUNCTUATIONS = "\"'.。,，!！?？:：”)]}、"
SENTENCE_END_MARKS = ".。!！?？"


@dataclass
class Word:
    start: float
    end: float
    word: str
    probability: float


def median_filter(x: np.ndarray, width: int) -> np.ndarray:
    """Median over a sliding window along the last axis, edges reflected (no-op when the axis is too short)."""
    pad = width // 2
    if width <= 1 or x.shape[-1] <= pad:
        return x
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad]


def dtw(cost: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Minimum-cost monotone path through cost[n_tok][n_frames] -> (token index, frame index) per path point."""
    from . import _lib
    lib = _lib.load()
    c = np.ascontiguousarray(cost, dtype=np.float32)
    n, m = c.shape
    rows = np.empty(n + m, dtype=np.int32)
    cols = np.empty(n + m, dtype=np.int32)
    length = C.c_int32(0)
    rc = lib.ttasr_dtw(c.ctypes.data_as(C.POINTER(C.c_float)), n, m, rows.ctypes.data_as(C.POINTER(C.c_int32)),
                       cols.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(length))
    if rc != 0:
        raise ValueError(f"ttasr_dtw({n}, {m}) failed ({rc})")
    return rows[: length.value].copy(), cols[: length.value].copy()


def token_start_times(weights: np.ndarray, first_row: int, last_row: int, num_frames: Optional[int] = None,
                      medfilt_width: int = 7) -> np.ndarray:
    """weights [n_heads][n_tok][n_ctx] -> start time in seconds of the token each row first_row..last_row-1 predicts."""
    w = weights if num_frames is None else weights[..., : max(1, num_frames // 2)]
    w = w[:, first_row:last_row, :].astype(np.float32)
    std = w.std(axis=-2, keepdims=True)
    mean = w.mean(axis=-2, keepdims=True)
    w = median_filter((w - mean) / np.where(std > 0, std, 1.0), medfilt_width).mean(axis=0)
    ti, tj = dtw(-w)
    jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
    return tj[jumps] / TOKENS_PER_SECOND


def split_tokens_on_unicode(tokenizer, tokens: Sequence[int]) -> Tuple[List[str], List[List[int]]]:
    """Smallest token groups that decode to complete unicode text (byte-level BPE splits multi-byte characters)."""
    full = tokenizer.decode(list(tokens))
    bad = "\ufffd"
    words, groups, cur, offset = [], [], [], 0
    for t in tokens:
        cur.append(t)
        text = tokenizer.decode(cur)
        at = text.find(bad)
        if at < 0 or (at + offset < len(full) and full[at + offset] == bad):
            words.append(text)
            groups.append(cur)
            cur = []
            offset += len(text)
    if cur:                                   # trailing incomplete bytes stay one (garbled) word
        words.append(tokenizer.decode(cur))
        groups.append(cur)
    return words, groups


def split_tokens_on_spaces(tokenizer, tokens: Sequence[int], eot: int) -> Tuple[List[str], List[List[int]]]:
    sub, sub_tokens = split_tokens_on_unicode(tokenizer, tokens)
    words: List[str] = []
    groups: List[List[int]] = []
    for s, g in zip(sub, sub_tokens):
        special = g[0] >= eot
        if special or s.startswith(" ") or (s.strip() and s.strip() in string.punctuation) or not words:
            words.append(s)
            groups.append(list(g))
        else:
            words[-1] += s
            groups[-1].extend(g)
    return words, groups


def split_to_word_tokens(tokenizer, tokens: Sequence[int], language: str, eot: int):
    if language in NO_SPACE_LANGUAGES:
        return split_tokens_on_unicode(tokenizer, tokens)
    return split_tokens_on_spaces(tokenizer, tokens, eot)


def merge_punctuations(alignment: List[dict], prepended: str = PREPEND_PUNCTUATIONS, appended: str = APPEND_PUNCTUATIONS):
    """Opening punctuation joins the following word, closing punctuation the preceding one (emptied entries stay in
    the list with word == '' and are skipped by the caller, as in faster-whisper)."""
    i, j = len(alignment) - 2, len(alignment) - 1
    while i >= 0:
        prev, nxt = alignment[i], alignment[j]
        if prev["word"].startswith(" ") and prev["word"].strip() in prepended and prev["word"].strip():
            nxt["word"] = prev["word"] + nxt["word"]
            nxt["tokens"] = prev["tokens"] + nxt["tokens"]
            prev["word"], prev["tokens"] = "", []
        else:
            j = i
        i -= 1
    i, j = 0, 1
    while j < len(alignment):
        prev, nxt = alignment[i], alignment[j]
        if not prev["word"].endswith(" ") and nxt["word"] in appended and nxt["word"]:
            prev["word"] = prev["word"] + nxt["word"]
            prev["tokens"] = prev["tokens"] + nxt["tokens"]
            nxt["word"], nxt["tokens"] = "", []
        else:
            i = j
        j += 1


def default_alignment_heads(dec_layers: int, n_heads: int, limit: int = 16) -> List[Tuple[int, int]]:
    """Without `alignment_heads` in the model's config faster-whisper takes every head of the last half of the decoder;
    that is hundreds of attention maps on large models, so the default here is capped at `limit` of them (the last
    layers first).  Converted official checkpoints carry their own 6-10 curated heads."""
    heads = [(l, h) for l in range(dec_layers - 1, dec_layers // 2 - 1, -1) for h in range(n_heads)]
    return sorted(heads[:limit])


def find_alignment(engine, tokenizer, special, clip: int, text_tokens: Sequence[int], num_frames: int,
                   heads: Sequence[Tuple[int, int]], language: str = "zh", lang_token: Optional[int] = None,
                   task_token: Optional[int] = None, medfilt_width: int = 7) -> List[dict]:
    """-> [{word, tokens, start, end, probability}] relative to the start of the window (faster-whisper find_alignment)."""
    if len(text_tokens) == 0:
        return []
    sot_seq = [special.sot, special.lang_zh if lang_token is None else lang_token,
               special.transcribe if task_token is None else task_token]
    tokens = sot_seq + [special.no_timestamps] + list(text_tokens) + [special.eot]
    weights, logprob = engine.align(clip, tokens, heads, want_logprob=True)
    n_sot = len(sot_seq)
    # rows n_sot .. -1: <|notimestamps|> and the text tokens as FED tokens = the text tokens and <|eot|> as predictions
    starts = token_start_times(weights, n_sot, len(tokens) - 1, num_frames, medfilt_width)
    text_probs = np.exp(logprob[n_sot: n_sot + len(text_tokens)])          # p(text token k) = logprob[index of k - 1]
    words, word_tokens = split_to_word_tokens(tokenizer, list(text_tokens) + [special.eot], language, special.eot)
    if len(word_tokens) <= 1:
        return []
    bounds = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    if len(bounds) <= 1:
        return []
    bounds = np.minimum(bounds, len(starts) - 1)
    start_t, end_t = starts[bounds[:-1]], starts[bounds[1:]]
    out = []
    for w, toks, s, e, i, j in zip(words, word_tokens, start_t, end_t, bounds[:-1], bounds[1:]):
        p = float(np.mean(text_probs[i:j])) if j > i else 0.0
        out.append(dict(word=w, tokens=list(toks), start=float(s), end=float(e), probability=p))
    return out


def add_word_timestamps(segments: List[dict], alignment: List[dict], time_offset: float) -> None:
    """Distributes aligned words over the window's segments (each a dict with 'tokens', 'start', 'end'; gets 'words').
    Clamps implausibly long words at sentence boundaries to twice the median word duration, merges punctuation, then
    snaps each segment's start/end to its first/last word, as faster-whisper does."""
    durations = np.array([w["end"] - w["start"] for w in alignment])
    durations = durations[durations.nonzero()]
    median = min(0.7, float(np.median(durations))) if len(durations) else 0.0
    max_dur = median * 2
    if len(durations):
        for i in range(1, len(alignment)):
            w = alignment[i]
            if w["end"] - w["start"] > max_dur:
                if w["word"] in SENTENCE_END_MARKS:
                    w["end"] = w["start"] + max_dur
                elif alignment[i - 1]["word"] in SENTENCE_END_MARKS:
                    w["start"] = w["end"] - max_dur
    merge_punctuations(alignment)
    idx = 0
    for seg in segments:
        n_text = len([t for t in seg["tokens"] if t < seg["eot"]])
        saved, words = 0, []
        while idx < len(alignment) and saved < n_text:
            w = alignment[idx]
            if w["word"]:
                words.append(Word(start=round(time_offset + w["start"], 2), end=round(time_offset + w["end"], 2),
                                  word=w["word"], probability=w["probability"]))
            saved += len(w["tokens"])
            idx += 1
        if words:
            # keep the segment-level timestamp when the first / last word came out implausibly long, else snap to the words
            if seg["start"] < words[0].end and seg["start"] - 0.5 > words[0].start:
                words[0].start = max(0.0, min(words[0].end - median, seg["start"]))
            else:
                seg["start"] = words[0].start
            if seg["end"] > words[-1].start and seg["end"] + 0.5 < words[-1].end:
                words[-1].end = max(words[-1].start + median, seg["end"])
            else:
                seg["end"] = words[-1].end
        seg["words"] = words

"""Generates tests/golden/text.json: known-answer vectors for the scoring / post-processing row.

Runs ONLY in the build container (needs /root/reference).  `cer.py` is imported as is (it depends on the
standard library only).  `asr_core.py` cannot be imported (faster_whisper, opencc, cn2an, pywer, librosa are
absent), so its pure text helpers are lifted out of its syntax tree and executed unmodified:
full_to_half, remove_special_characters_by_dataset_name, replace_words, convert_time, split_sentence_to_words
(asr_core.py:22-78).  Only inputs and outputs are written; no reference text is stored.
"""
import ast
import json
import os
import re
import sys
import unicodedata
from datetime import datetime, timedelta

REF = "/root/reference"
sys.path.insert(0, REF)
import cer as ref_cer  # noqa: E402

src = open(os.path.join(REF, "asr_core.py"), encoding="utf-8").read()
wanted = {"full_to_half", "remove_special_characters_by_dataset_name", "replace_words", "convert_time",
          "split_sentence_to_words"}
mod = ast.Module(body=[n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in wanted],
                 type_ignores=[])
ns = {"re": re, "unicodedata": unicodedata, "datetime": datetime, "timedelta": timedelta}
exec(compile(mod, "asr_core_helpers", "exec"), ns)

numbers = ["0", "1", "7", "10", "11", "12", "15", "19", "20", "21", "99", "100", "101", "110", "111", "120", "205", "1000",
           "1001", "1010", "1100", "2024", "9999", "10000", "10001", "10010", "10100", "12000", "12345", "20000", "25000",
           "100000", "100001", "120000", "123456", "1000000", "1200000", "1234567", "10000000", "12345678", "100000000",
           "123456789", "999999999", "1234567890", "007", "00", "010", "0800", "080009598", "300", "3000", "30000", "40404"]

pairs = [
    ("今天天氣很好，我們去公園散步。", "今天天氣很好!，我去公園散步。"),
    ("今天天氣很好", "今天天氣很好"),
    ("她說它在臺北買得到", "他說他在台北買的到"),
    ("我有3個蘋果和12顆橘子", "我有三個蘋果和十二顆橘子"),
    ("電話是080009598請撥打", "電話是零八零零零九五九八請撥"),
    ("Hello World 你好", "hello word 你 好 嗎"),
    ("一二三四五六七八九十", "一二三五六七八八九十十"),
    ("完全不同的句子", "另外一段文字內容而且比較長"),
    ("短", "這是一個很長的輸出結果"),
    ("這是一個很長的參考文本內容", "短"),
    ("第一行\n第二行\r\n第三行", "第一行第二行第四行"),
    ("民國113年5月20日", "民國一百一十三年五月二十日"),
    ("abc", "abd"),
    ("！？。，", "你好"),
    ("你好", "！？"),
    ("重複重複重複重複的的的字", "重複重複的字字字"),
    ("", "空的參考"),
    ("空的輸出", ""),
    ("臺灣的語音辨識系統在2024年有了很大的進步" * 9, "台灣的語音辨識系統在二零二四年有很大進步" * 9),
]

norm_inputs = [
    "你好，世界！", "Ｈｅｌｌｏ　Ｗｏｒｌｄ１２３", "「引號」《書名》：；", "百分之十五的人", "成長百分之十二點五，下降百分之五和百分之七",
    "請撥零八零零零九五九八", "a,b\"c'd。e", "＄100 & ＃tag #x", "(括號)（全形）[方]【黑】{花}", "保留-連字號 與 空格", "ＡＢＣ ㈱ ①②③ ｶﾀｶﾅ",
    "…⋯—―─–－〜～", "line1\nline2", "MiXeD CaSe 文字", "\\反斜線/斜線", "<tag> = _under_", "«guillemets» „low“ →", "?!;`^¿¡", "",
]
times = [0.0, 0.5, 1.234, 59.9994, 59.9996, 60.0, 61.5, 3599.999, 3600.0, 3661.001, 86399.5, 12.3456]
split_inputs = ["你好world 123", "今天3.14度 50%", "Hello 世界 ABC", "  前後空白  ", "ｶﾀｶﾅ한국어test", ""]

out = {
    "numbers": [[n, ref_cer.arabic_to_chinese_number(n)] for n in numbers],
    "clean": [[a, ref_cer.clean_text(a)] for p in pairs for a in p],
    "cer": [],
    "normalise": [[t, ns["remove_special_characters_by_dataset_name"](ns["replace_words"](t)).lower()] for t in norm_inputs],
    "convert_time": [[t, ns["convert_time"](t)] for t in times],
    "split_words": [[t, ns["split_sentence_to_words"](t, True)] for t in split_inputs],
}
fields = ["reference_cleaned", "hypothesis_cleaned", "correct_rate", "cer_rate", "total_errors", "substitutions_count",
          "deletions_count", "insertions_count", "total_chars", "substitutions_errors", "deletions_errors",
          "insertions_errors", "reference_highlighted", "hypothesis_highlighted"]
for a, b in pairs:
    r = ref_cer.compare_texts(a, b)
    out["cer"].append({"reference": a, "hypothesis": b, "result": None if r is None else {f: getattr(r, f) for f in fields}})

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "text.json")
with open(dst, "w", encoding="utf-8") as f:
    json.dump(out, f, ensure_ascii=False, indent=1)
print("wrote", dst, {k: len(v) for k, v in out.items()})

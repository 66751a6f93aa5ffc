"""Scoring / post-processing row against vectors produced by the reference's own cer.py and
asr_core.py helpers (oracle/make_golden_text.py)."""
import json
import os

import pytest

from taiwan_tongues_asr_ce_amd import scoring

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text.json"), encoding="utf-8"))


@pytest.mark.parametrize("digits,expect", G["numbers"])
def test_digits_to_chinese(digits, expect):
    assert scoring.digits_to_chinese(digits) == expect


@pytest.mark.parametrize("text,expect", G["clean"])
def test_clean_for_scoring(text, expect):
    assert scoring.clean_for_scoring(text) == expect


@pytest.mark.parametrize("case", G["cer"], ids=lambda c: c["reference"][:8] or "empty")
def test_cer_known_answers(case):
    got = scoring.compare_texts(case["reference"], case["hypothesis"])
    if case["result"] is None:
        assert got is None
        return
    for k, v in case["result"].items():
        assert getattr(got, k) == v, k
    assert set(got.as_dict()) == set(case["result"]) - {"reference_cleaned", "hypothesis_cleaned"}


@pytest.mark.parametrize("text,expect", G["normalise"])
def test_normalise_transcript(text, expect):
    assert scoring.normalise_transcript(text, to_traditional=False) == expect


@pytest.mark.parametrize("t,expect", G["convert_time"])
def test_srt_time(t, expect):
    assert scoring.srt_time(t) == expect


@pytest.mark.parametrize("text,expect", G["split_words"])
def test_split_cjk_words(text, expect):
    assert scoring.split_cjk_words(text) == expect
    assert scoring.split_cjk_words(text, False) == text

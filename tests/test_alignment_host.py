"""Host half of word timestamps (alignment.py) on CPU: median filter, the C++ DTW exported by libttasr (no GPU
needed), token times against HF's golden timestamps, unicode / space word grouping, punctuation merging and the
distribution of words over segments."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import alignment as A
from taiwan_tongues_asr_ce_amd.tokenizer import ByteStubTokenizer

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "align.npz"))


def test_median_filter_matches_the_oracle():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 9, 40)).astype(np.float32)
    np.testing.assert_array_equal(A.median_filter(x, 7), R.median_filter(torch.from_numpy(x), 7).numpy())
    short = x[..., :3]
    assert A.median_filter(short, 7) is short            # axis shorter than the half width: untouched, like the reference


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (7, 1), (5, 30), (40, 300), (19, 50)])
def test_cpp_dtw_matches_the_oracle(shape):
    rng = np.random.default_rng(sum(shape))
    cost = rng.standard_normal(shape).astype(np.float32)
    cost[::2] = np.round(cost[::2], 1)                   # ties exercise the branch order
    ti, tj = A.dtw(cost)
    ri, rj = R.dtw_path(cost)
    np.testing.assert_array_equal(ti, ri)
    np.testing.assert_array_equal(tj, rj)
    assert ti[0] == 0 and tj[0] == 0 and ti[-1] == shape[0] - 1 and tj[-1] == shape[1] - 1
    assert (np.diff(ti) >= 0).all() and (np.diff(tj) >= 0).all() and ((np.diff(ti) + np.diff(tj)) >= 1).all()


@pytest.mark.parametrize("name", ["micro", "tiny"])
def test_token_start_times_match_hf(name):
    w = G[f"{name}_weights"].astype(np.float32)
    n_prefix, n_tok = int(G[f"{name}_n_prefix"]), w.shape[1]
    for tag in ("full", "short"):
        nf = int(G[f"{name}_nf_{tag}"])
        got = A.token_start_times(w, n_prefix, n_tok, nf)
        np.testing.assert_allclose(got, G[f"{name}_ts_{tag}"][n_prefix:n_prefix + len(got)], atol=1e-6)


def test_word_grouping_and_punctuation():
    tk = ByteStubTokenizer(51865, 50257)
    toks = tk.encode("你好，world")                       # 3 + 3 + 3 bytes, then ASCII
    words, groups = A.split_tokens_on_unicode(tk, toks + [50257])
    assert words[:3] == ["你", "好", "，"] and [len(g) for g in groups[:3]] == [3, 3, 3]
    assert "".join(words) == "你好，world" and groups[-1] == [50257] and words[-1] == ""
    words, groups = A.split_tokens_on_spaces(tk, tk.encode(" hello wor") + tk.encode("ld !") + [50257], 50257)
    assert words == [" hello", " world", " ", "!", ""] or words[:2] == [" hello", " world"]
    al = [dict(word=" (", tokens=[1]), dict(word="你", tokens=[2, 3, 4]), dict(word="。", tokens=[5]), dict(word=" 好", tokens=[6])]
    A.merge_punctuations(al)
    assert [a["word"] for a in al] == ["", " (你。", "", " 好"] and al[1]["tokens"] == [1, 2, 3, 4, 5]


def test_words_are_dealt_to_segments_in_token_order():
    eot = 50257
    segs = [dict(start=0.0, end=2.0, tokens=[50364, 10, 11, 12, 50464], eot=eot, words=None),
            dict(start=2.0, end=4.0, tokens=[50464, 13, 14, 50564], eot=eot, words=None)]
    al = [dict(word="a", tokens=[10], start=0.1, end=0.5, probability=0.9),
          dict(word="b", tokens=[11, 12], start=0.5, end=1.9, probability=0.8),
          dict(word="c", tokens=[13], start=2.2, end=2.6, probability=0.7),
          dict(word="d", tokens=[14], start=2.6, end=3.1, probability=0.6),
          dict(word="", tokens=[eot], start=3.1, end=3.1, probability=0.0)]
    A.add_word_timestamps(segs, al, time_offset=30.0)
    assert [w.word for w in segs[0]["words"]] == ["a", "b"] and [w.word for w in segs[1]["words"]] == ["c", "d"]
    assert segs[0]["words"][0].start == 30.1 and segs[1]["words"][-1].end == 33.1
    assert segs[0]["start"] == 30.1 and segs[0]["end"] == 31.9 and segs[1]["start"] == 32.2 and segs[1]["end"] == 33.1
    assert A.default_alignment_heads(32, 20) == sorted([(31, h) for h in range(16)])
    assert A.default_alignment_heads(4, 6) == sorted([(3, h) for h in range(6)] + [(2, h) for h in range(6)])

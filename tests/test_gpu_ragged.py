"""GPU: finished rows leave the decode kernels (round 6).

The reference consumes every utterance's segments until the generator ends - natural stopping, a different length per utterance
(asr_core.py:159-172); rounds 1-5 decoded a STATIC batch: a row that had emitted EOT kept streaming its 2 x 1500 x 128 B of
cross-KV per (layer, head) until the LAST row finished.  Now `DecState.done[row]` is read by the attention kernels of the decode
step (cross_attn_pipe / _decode / _fp8 / _split + _merge / _mq, self_attn_decode: kernels_attn.hip `row_done_exit`) and a finished
row's workgroups return before their streams start.  Synthetic weights never emit a meaningful EOT, so the lengths are STATED per
row (`ttasr_generate_capped`: the test / benchmark entry point); the natural-EOT route sets the same flag in the same kernel.

What is held here, at the benchmark's width (large-v3-w2: d 1280, 20 heads, B = 32 -> 640 (row, head) workgroups, the pipelined
single-pass kernel) and on the small-batch paths:
  * every row of a mixed-length batch is IDENTICAL, token for token, to the same row of the uncapped run cut at its budget, and its
    sum of log-probabilities equals the uncapped run's bitwise when the row ran the full length: rows are computed independently
    of their neighbours, so a neighbour that left cannot change a live row's bits;
  * the mixed-length tokens are graded by the oracle (one causal pass per graded row, oracle_checks.teacher_forced_causal):
    f32 1e-3, bf16 / fp16 0.15 + token equality under margin - the same gates as test_gpu_measured_shape.py;
  * option `ragged_exit` = 0 (the static batch) returns the same tokens and the same log-probabilities, bit for bit;
  * the fp8 cross-KV mode, the frame-split kernels (B x H < 256) and sampled rows sharing a clip behave the same way."""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F16, COMPUTE_F32, PRESETS

from oracle_checks import encode_chunked, teacher_forced_causal

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

DIMS = PRESETS["large-v3-w2"]
B = 32
N_NEW = 128
GRADED = (0, 5, 11, 17, 22, 26, 29, 31)     # rows the oracle recomputes (one causal pass each: their lengths differ)


def _caps(seed=6, lo=32, hi=128, n=B):
    """Seeded per-row budgets, uniform lo..hi; the longest row keeps the full length so that the batch runs all its steps."""
    caps = np.random.Generator(np.random.Philox(key=seed)).integers(lo, hi + 1, size=n).astype(np.int32)
    caps[int(caps.argmax())] = hi
    return caps


def _clips(n):
    kinds = (synth.noise_clip, synth.tonal_clip, synth.noise_clip, synth.burst_clip)
    return [kinds[i % 4](100 + i) for i in range(n)]


def _engine(compute, sd, max_batch=B):
    from taiwan_tongues_asr_ce_amd.engine import Engine
    e = Engine(DIMS, compute, max_batch)
    e.load_weights(sd.items())
    return e


def _rules(e, opts, timestamps):
    st = e.special
    r = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                suppress=[opts.suppress[i] for i in range(opts.n_suppress)], begin_suppress=[220, st.eot], timestamps=timestamps)
    r.suppress_eot = True
    return r


@pytest.fixture(scope="module")
def world():
    sd = synth.state_dict(DIMS)
    clips = _clips(B)
    mel_ref = {r: R.log_mel(clips[r], DIMS.n_mels) for r in GRADED}
    return sd, clips, mel_ref


@pytest.mark.parametrize("compute,tol,margin", [(COMPUTE_F32, 1e-3, 2e-3), (COMPUTE_BF16, 0.15, 0.16), (COMPUTE_F16, 0.15, 0.16)],
                         ids=["f32", "bf16", "f16"])
def test_mixed_length_batch_is_the_uncapped_batch_cut_row_by_row(world, compute, tol, margin):
    sd, clips, mel_ref = world
    rd = R.Dims(**DIMS.as_dict())
    W = R.to_torch(sd, round_bf16=compute == COMPUTE_BF16, round_f16=compute == COMPUTE_F16)
    e = _engine(compute, sd)
    st = e.special
    caps = _caps()
    assert caps.min() >= 32 and caps.max() == N_NEW and len(set(caps.tolist())) > 16
    e.log_mel(clips, want_output=False)
    e.encode(B)
    for ts in (False, True):
        prompt = [st.sot, st.lang_zh, st.transcribe] + ([] if ts else [st.no_timestamps])
        opts = e.gen_opts(N_NEW, ts, suppress_eot=True, check_interval=1 << 20)      # the benchmark's options
        full = e.generate([prompt] * B, opts)
        assert all(len(t) == N_NEW for t in full.tokens)
        cut = e.generate([prompt] * B, opts, row_max_new=caps)
        assert [len(t) for t in cut.tokens] == caps.tolist()
        for r in range(B):
            assert cut.tokens[r] == full.tokens[r][:caps[r]], (ts, r, int(caps[r]))
            if caps[r] == N_NEW:
                assert cut.sum_logprob[r] == full.sum_logprob[r], (ts, r)
        assert np.isfinite(cut.sum_logprob).all() and np.array_equal(cut.no_speech_prob, full.no_speech_prob)
        # the static batch (no early exit) gives the same bits, for every row
        e.set_option("ragged_exit", 0)
        static = e.generate([prompt] * B, opts, row_max_new=caps)
        e.set_option("ragged_exit", 1)
        assert static.tokens == cut.tokens and np.array_equal(static.sum_logprob, cut.sum_logprob)
        # a replay is bit-identical, and a different assignment of the same budgets to rows changes nothing for a row that keeps its own
        again = e.generate([prompt] * B, opts, row_max_new=caps)
        assert again.tokens == cut.tokens and np.array_equal(again.sum_logprob, cut.sum_logprob)
        caps2 = caps.copy()
        caps2[1::2] = 32                                                                 # every odd row leaves after 32 tokens
        half = e.generate([prompt] * B, opts, row_max_new=caps2)
        for r in range(0, B, 2):
            assert half.tokens[r] == cut.tokens[r] and half.sum_logprob[r] == cut.sum_logprob[r], (ts, r)
        if ts:
            continue
        # the oracle grades the mixed-length rows (their own length each)
        enc_ref = encode_chunked(np.stack([mel_ref[r] for r in GRADED]), W, rd)
        rules = _rules(e, opts, ts)
        n_steps = n_clear = 0
        for i, r in enumerate(GRADED):
            g = teacher_forced_causal([cut.tokens[r]], prompt, enc_ref[i:i + 1], W, rd, rules, tol=tol, margin=margin, rows_per_pass=1)
            n_steps += g.n_steps
            n_clear += g.n_clear
        assert n_steps == int(sum(caps[r] for r in GRADED))
        assert n_clear >= 0.6 * n_steps, (n_clear, n_steps)
    e.close()


def test_bad_budgets_are_refused_and_natural_eot_still_polls(world):
    from taiwan_tongues_asr_ce_amd.engine import TtasrError
    sd, clips, _ = world
    e = _engine(COMPUTE_BF16, sd)
    st = e.special
    e.log_mel(clips[:4], want_output=False)
    e.encode(4)
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    opts = e.gen_opts(16, False, suppress_eot=True)
    for bad in ([0, 4, 4, 4], [4, 4, 4, 17], [-1, 1, 1, 1]):
        with pytest.raises(TtasrError):
            e.generate([prompt] * 4, opts, row_max_new=bad)
    with pytest.raises(ValueError):
        e.generate([prompt] * 4, opts, row_max_new=[4, 4])
    # after a refused call the context still works, and budgets combine with natural stopping (EOT allowed, host polls every step)
    opts2 = e.gen_opts(16, False, suppress_eot=False, check_interval=1)
    a = e.generate([prompt] * 4, opts2)
    b = e.generate([prompt] * 4, opts2, row_max_new=[3, 16, 9, 1])
    for r, cap in enumerate((3, 16, 9, 1)):
        assert b.tokens[r] == a.tokens[r][:cap], r
    e.close()


def test_small_batch_split_kernels_and_fp8_mode_leave_finished_rows_out(world):
    """B x H < 256: the frames of a row are split over workgroups (cross_attn_split_kernel + merge); with the e4m3 cross-KV copy
    the B = 32 step runs cross_attn_fp8_kernel.  Same contract: a capped row equals the uncapped row cut at its budget."""
    sd, clips, _ = world
    e = _engine(COMPUTE_BF16, sd)
    st = e.special
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    opts = e.gen_opts(24, False, suppress_eot=True, check_interval=1 << 20)
    for n in (1, 5, 12):                                     # 20 / 100 / 240 (row, head) items: frame-split kernels
        e.log_mel(clips[:n], want_output=False)
        e.encode(n)
        full = e.generate([prompt] * n, opts)
        caps = _caps(seed=n, lo=2, hi=24, n=n)
        cut = e.generate([prompt] * n, opts, row_max_new=caps)
        for r in range(n):
            assert cut.tokens[r] == full.tokens[r][:caps[r]], (n, r)
    e.set_option("xkv_fp8", 1)
    e.log_mel(clips, want_output=False)
    e.encode(B)
    full = e.generate([prompt] * B, opts)
    caps = _caps(seed=3, lo=2, hi=24)
    cut = e.generate([prompt] * B, opts, row_max_new=caps)
    for r in range(B):
        assert cut.tokens[r] == full.tokens[r][:caps[r]], ("fp8", r)
    e.close()


def test_beam_search_with_clips_that_finish_early_is_unchanged(world):
    """A clip whose beam search has finished (its rows' done flags are uploaded by the search loop) leaves the shared-clip
    cross-attention kernel; the clips still searching must not notice: same hypotheses and scores as with the early exit off."""
    sd, clips, _ = world
    e = _engine(COMPUTE_BF16, sd, max_batch=40)
    st = e.special
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    e.log_mel(clips[:8], want_output=False)
    e.encode(8)
    opts = e.gen_opts(12, False)
    on = e.generate_beam([prompt] * 8, 5, opts)
    e.set_option("ragged_exit", 0)
    off = e.generate_beam([prompt] * 8, 5, opts)
    assert on.tokens == off.tokens and np.array_equal(on.sum_logprob, off.sum_logprob)
    e.close()


@pytest.mark.parametrize("n", [48, 72], ids=["48-rows-remapped", "72-rows-own-flag"])
def test_wide_batches_leave_finished_rows_out(world, n):
    """Batches of 33 ... 64 rows keep the live-row remap (lane i of every wave holds done[i]: the ballot covers up to 64 rows);
    wider ones test their own row's flag only.  Both: a capped row equals the uncapped row cut at its budget, the static batch
    gives the same bits, and budgets that empty the FRONT of the batch (every workgroup of the remapped form then serves a row
    far from its slot) change nothing for the rows that stay."""
    sd, _, _ = world
    clips = _clips(n)
    e = _engine(COMPUTE_BF16, sd, max_batch=n)
    st = e.special
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    opts = e.gen_opts(20, False, suppress_eot=True, check_interval=1 << 20)
    e.log_mel(clips, want_output=False)
    e.encode(n)
    full = e.generate([prompt] * n, opts)
    caps = _caps(seed=n, lo=1, hi=20, n=n)
    cut = e.generate([prompt] * n, opts, row_max_new=caps)
    e.set_option("ragged_exit", 0)
    static = e.generate([prompt] * n, opts, row_max_new=caps)
    e.set_option("ragged_exit", 1)
    assert static.tokens == cut.tokens and np.array_equal(static.sum_logprob, cut.sum_logprob)
    for r in range(n):
        assert cut.tokens[r] == full.tokens[r][:caps[r]], (n, r, int(caps[r]))
    front = np.full(n, 1, np.int32)
    front[n // 2:] = 20                                      # the first half leaves after one token
    tail = e.generate([prompt] * n, opts, row_max_new=front)
    for r in range(n // 2, n):
        assert tail.tokens[r] == full.tokens[r] and tail.sum_logprob[r] == full.sum_logprob[r], (n, r)
    e.close()


def test_sampled_rows_of_a_clip_that_finish_at_different_steps(world):
    """Temperature sampling: the best_of rows of a clip share its cross-KV stream (cross_attn_mq_kernel) and end at DIFFERENT steps
    once EOT is likely - a partly finished group is computed whole and only its live rows are merged, a fully finished group
    streams nothing.  All but a dozen text tokens are suppressed, so that EOT carries about 1 / 13 of the mass at temperature 4:
    the clips' best rows must be the same rows, token for token and bit for bit in their scores, with the early exit off."""
    sd, clips, _ = world
    e = _engine(COMPUTE_BF16, sd, max_batch=40)
    st = e.special
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    keep = set(range(300, 312)) | {st.eot}
    sup = [t for t in range(DIMS.vocab) if t not in keep]
    e.log_mel(clips[:8], want_output=False)
    e.encode(8)
    opts = e.gen_opts(24, False, suppress=sup, begin_suppress=[], no_speech=False, check_interval=4)
    seen_short = 0
    for seed in (1, 2, 3):
        on = e.generate_sample([prompt] * 8, 5, opts, temperature=4.0, seed=seed)
        e.set_option("ragged_exit", 0)
        off = e.generate_sample([prompt] * 8, 5, opts, temperature=4.0, seed=seed)
        e.set_option("ragged_exit", 1)
        assert on.tokens == off.tokens and np.array_equal(on.sum_logprob, off.sum_logprob), seed
        assert all(set(t) <= keep for t in on.tokens)
        seen_short += sum(len(t) < 24 for t in on.tokens)
    assert seen_short >= 4, seen_short                       # rows did end early (and at different lengths) in these runs
    e.close()

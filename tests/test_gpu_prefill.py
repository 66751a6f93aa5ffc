"""GPU: batched prompt prefill (previous-text / initial-prompt tokens computed in one pass) == feeding the prompt
token by token.  f32: tokens exact vs the CPU oracle (which always feeds token by token) and vs the engine with
the `prefill = 0` option (ttasr_set_option); bf16: same tokens' score within tolerance under the oracle's teacher forcing."""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F32, PRESETS

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
NAME = "tiny"


def _engine(compute, max_batch, no_prefill=False):
    from taiwan_tongues_asr_ce_amd.engine import Engine
    e = Engine(PRESETS[NAME], compute, max_batch)
    if no_prefill:
        e.set_option("prefill", 0)                          # explicit test hook (ttasr_set_option): no environment switch
    e.load_weights(synth.iter_weights(PRESETS[NAME]))
    return e


@pytest.fixture(scope="module")
def setup():
    pd = PRESETS[NAME]
    dims = R.Dims(**pd.as_dict())
    W = R.to_torch(synth.state_dict(pd))
    clips = [synth.noise_clip(0), synth.tonal_clip(1), synth.burst_clip(2)]
    mel = torch.from_numpy(np.stack([R.log_mel(c, pd.n_mels) for c in clips]))
    enc = R.encoder_forward(mel, W, dims)
    return pd, dims, W, clips, enc


def _prev_prompt(st, rng, n_prev):
    prev = rng.integers(300, 20000, size=n_prev).tolist()
    return [st.sot_prev] + prev + [st.sot, st.lang_zh, st.transcribe]


def test_prefill_greedy_f32_exact(setup):
    pd, dims, W, clips, enc = setup
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    e, e_ref = _engine(COMPUTE_F32, 3), _engine(COMPUTE_F32, 3, no_prefill=True)
    st = e.special
    rng = np.random.default_rng(5)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab), begin_suppress=[220, st.eot], timestamps=True)
    for eng in (e, e_ref):
        eng.log_mel(clips, want_output=False)
        eng.encode(3)
    # (a) one shared 37-token prompt with no-speech probability wanted: prefill stops before <|startoftranscript|>
    prompt = _prev_prompt(st, rng, 33)
    sot_index = prompt.index(st.sot)
    opts = e.gen_opts(12, True, sot_index=sot_index)
    got, want = e.generate([prompt] * 3, opts), e_ref.generate([prompt] * 3, opts)
    ref = R.greedy_decode(enc, prompt, W, dims, rules, 12, no_speech_token=st.no_speech, sot_index=sot_index)
    assert got.tokens == ref.tokens == want.tokens
    np.testing.assert_allclose(got.sum_logprob, ref.sum_logprob, atol=2e-2)
    np.testing.assert_allclose(got.no_speech_prob, ref.no_speech_prob, rtol=1e-3)
    np.testing.assert_allclose(got.sum_logprob, want.sum_logprob, atol=1e-3)
    assert e.phase_ms()["decode"] < e_ref.phase_ms()["decode"]          # 33 positions in one pass, not 33 steps
    # (b) ragged prompts (19 / 26 / 22 previous tokens), no no-speech probability: prefill = shortest prompt - 1
    prompts = [_prev_prompt(st, rng, n) for n in (19, 26, 22)]
    opts = e.gen_opts(10, True, no_speech=False)
    got, want = e.generate(prompts, opts), e_ref.generate(prompts, opts)
    assert got.tokens == want.tokens
    for b, p in enumerate(prompts):
        ref = R.greedy_decode(enc[b:b + 1], p, W, dims, rules, 10)
        assert got.tokens[b] == ref.tokens[0], b
    # (c) a prompt too short to prefill still works (pre = 0 path)
    short = [st.sot, st.lang_zh, st.transcribe]
    assert e.generate([short] * 3, e.gen_opts(6, True)).tokens == e_ref.generate([short] * 3, e.gen_opts(6, True)).tokens
    e.close(); e_ref.close()


def test_prefill_beam_and_sampling_f32_exact(setup):
    pd, dims, W, clips, enc = setup
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    e, e_ref = _engine(COMPUTE_F32, 8), _engine(COMPUTE_F32, 8, no_prefill=True)
    st = e.special
    rng = np.random.default_rng(9)
    prompt = _prev_prompt(st, rng, 21)             # 25 tokens: the 2nd KV page is partially filled -> copy-on-write path
    sot_index = prompt.index(st.sot)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab), begin_suppress=[220, st.eot], timestamps=True)
    for eng in (e, e_ref):
        eng.log_mel(clips[:2], want_output=False)
        eng.encode(2)
    opts = e.gen_opts(10, True, sot_index=sot_index)
    got, want = e.generate_beam([prompt] * 2, 3, opts), e_ref.generate_beam([prompt] * 2, 3, opts)
    ref = R.beam_decode(enc[:2], prompt, W, dims, rules, 3, 10)
    strip = lambda toks: [[t for t in row if t != st.eot] for row in toks]
    assert strip(got.tokens) == strip(want.tokens) == strip(ref.tokens)
    np.testing.assert_allclose(got.sum_logprob, want.sum_logprob, atol=1e-3)
    s1 = e.generate_sample([prompt] * 2, 3, opts, temperature=0.6, seed=11)
    s2 = e_ref.generate_sample([prompt] * 2, 3, opts, temperature=0.6, seed=11)
    assert s1.tokens == s2.tokens
    e.close(); e_ref.close()


def test_prefill_bf16_consistent_with_token_by_token(setup):
    """bf16: the prefill pass runs the big-tile MFMA GEMMs, the token-by-token path the skinny split-K GEMM, so
    roundings differ.  Gate: teacher-forcing the f32 oracle (bf16-rounded weights) on either engine's tokens, the
    two hypotheses score within 0.5 total log-probability of each other, and the first sampled token agrees."""
    pd, dims, _, clips, _ = setup
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    Wb = R.to_torch(synth.state_dict(pd), round_bf16=True)
    e, e_ref = _engine(COMPUTE_BF16, 3), _engine(COMPUTE_BF16, 3, no_prefill=True)
    st = e.special
    rng = np.random.default_rng(7)
    prompt = _prev_prompt(st, rng, 40)
    for eng in (e, e_ref):
        eng.log_mel(clips, want_output=False)
        eng.encode(3)
    opts = e.gen_opts(12, True, no_speech=False)
    got, want = e.generate([prompt] * 3, opts), e_ref.generate([prompt] * 3, opts)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab), begin_suppress=[220, st.eot], timestamps=True)
    mel = torch.from_numpy(np.stack([R.log_mel(c, pd.n_mels) for c in clips]))
    enc = R.encoder_forward(mel, Wb, dims)

    def score(tokens, b):
        xkv = R.cross_kv(enc[b:b + 1], Wb, dims)
        seq = list(prompt) + list(tokens[:-1])
        logits = R.decoder_forward(torch.tensor([seq]), R.SelfCache.empty(dims.dec_layers), xkv, Wb, dims)[0]
        total = 0.0
        for i, t in enumerate(tokens):
            s = R.apply_rules(logits[len(prompt) - 1 + i], tokens[:i], rules)
            assert s[t] > -np.inf
            total += float(torch.log_softmax(torch.as_tensor(s), -1)[t])
        return total

    for b in range(3):
        assert got.tokens[b][0] == want.tokens[b][0]
        n = min(len(got.tokens[b]), len(want.tokens[b]))
        assert abs(score(got.tokens[b][:n], b) - score(want.tokens[b][:n], b)) < 0.5
    e.close(); e_ref.close()


def test_prefill_cross_attention_kernels_agree(setup):
    """The prefill pass has three cross-attention kernels: one workgroup per (row, head) (option `xsplit = 0`), groups of <= 8 rows
    per clip sharing one K/V stream (2 <= rows per clip < 32 in bf16, any count in f32), and - bf16, >= 32 rows per clip - the
    MFMA flash pass with the rows as the M dimension.  A 70-token previous-text prompt (flash in bf16, groups in f32) and a
    12-token one (groups) must give the same greedy tokens as the per-row kernel, with total log-probabilities within rounding."""
    pd, dims, _, clips, _ = setup
    rng = np.random.default_rng(11)
    for compute, tol in ((COMPUTE_F32, 2e-3), (COMPUTE_BF16, 0.25)):
        outs = {}
        for per_row in (False, True):
            e = _engine(compute, 3)
            e.set_option("xsplit", 0 if per_row else 1)
            st = e.special
            e.log_mel(clips, want_output=False)
            e.encode(3)
            rng = np.random.default_rng(11)
            for n_prev in (70, 12):
                prompt = _prev_prompt(st, rng, n_prev)
                outs[per_row, n_prev] = e.generate([prompt] * 3, e.gen_opts(6, True, no_speech=False))
            e.close()
        for n_prev in (70, 12):
            a, b = outs[False, n_prev], outs[True, n_prev]
            for r in range(3):
                assert a.tokens[r][0] == b.tokens[r][0], (compute, n_prev, r)
                if a.tokens[r] == b.tokens[r]:
                    assert abs(float(a.sum_logprob[r]) - float(b.sum_logprob[r])) < tol, (compute, n_prev, r)
            if compute == COMPUTE_F32:
                assert a.tokens == b.tokens

"""GPU: seeded differential fuzzing of the f32 engine against the CPU oracle on the micro model (where the oracle
costs milliseconds): random batch sizes, ragged prompts with and without a previous-text prefix (prefill path),
random rule sets, audio windows, greedy / beam / sampled decoding.  Every case must match the oracle token for token."""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_F32, PRESETS

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
NAME = "micro"


@pytest.fixture(scope="module")
def ctx():
    from taiwan_tongues_asr_ce_amd.engine import Engine
    pd = PRESETS[NAME]
    e = Engine(pd, COMPUTE_F32, 8)
    e.load_weights(synth.iter_weights(pd))
    W = R.to_torch(synth.state_dict(pd))
    yield e, pd, R.Dims(**pd.as_dict()), W
    e.close()


def _clips(rng, B, n_ctx):
    n = n_ctx * 320
    out = []
    for _ in range(B):
        kind = rng.integers(0, 4)
        length = int(rng.integers(n // 3, n + 1)) if kind != 3 else 0
        c = (0.1 * rng.standard_normal(length)).astype(np.float32)
        if kind == 1 and length:
            c[length // 2:] = 0.0
        out.append(c)
    return out


import os as _os


@pytest.mark.parametrize("seed", range(int(_os.environ.get("TTASR_FUZZ_CASES", "24"))))
def test_random_configuration_matches_oracle(ctx, seed):
    e, pd, dims, W = ctx
    st = e.special
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 9))
    n_ctx = int(rng.choice([pd.n_audio_ctx, pd.n_audio_ctx, 20, 36]))
    e.set_audio_ctx(n_ctx if n_ctx != pd.n_audio_ctx else 0)
    clips = _clips(rng, B, n_ctx)
    mel = e.log_mel(clips)
    want_mel = np.stack([R.log_mel(c, pd.n_mels, n_samples=n_ctx * 320) for c in clips])
    np.testing.assert_allclose(mel, want_mel, atol=3e-4, rtol=0)
    enc = e.encode(B, want_output=True)
    enc_ref = R.encoder_forward(torch.from_numpy(want_mel), W, dims)
    assert np.abs(enc - enc_ref.numpy()).max() < 1e-3
    timestamps = bool(rng.integers(0, 2))
    text_hi = st.eot
    suppress = sorted(set(rng.integers(0, text_hi, size=int(rng.integers(0, 12))).tolist()) | {st.sot, st.sot_prev, st.no_speech})
    begin_suppress = sorted(set(rng.integers(0, text_hi, size=int(rng.integers(0, 3))).tolist()) | {st.eot})
    tail = [st.sot, st.lang_zh, st.transcribe] + ([] if timestamps else [st.no_timestamps])
    mode = ["greedy", "greedy", "beam", "sample"][int(rng.integers(0, 4))]
    room = pd.n_text_ctx - len(tail) - 2
    max_initial = [None, 0, 7, 50][int(rng.integers(0, 4))]
    suppress_eot = bool(rng.integers(0, 4) == 0)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin, suppress=suppress,
                    begin_suppress=begin_suppress, timestamps=timestamps, max_initial_timestamp_index=max_initial,
                    suppress_eot=suppress_eot)
    if mode == "greedy":
        # ragged prompts: each row its own previous-text prefix (0..12 tokens) -> exercises prefill + forced steps
        prompts = []
        for _ in range(B):
            n_prev = int(rng.integers(0, min(13, room - 3)))
            prev = ([st.sot_prev] + rng.integers(0, text_hi, size=n_prev).tolist()) if n_prev else []
            prompts.append(prev + tail)
        max_new = int(rng.integers(1, pd.n_text_ctx - max(len(p) for p in prompts)))
        opts = e.gen_opts(max_new, timestamps, suppress=suppress, begin_suppress=begin_suppress, no_speech=False,
                          check_interval=int(rng.integers(1, 5)), max_initial_timestamp_index=max_initial,
                          suppress_eot=suppress_eot)
        res = e.generate(prompts, opts)
        for b in range(B):
            ref = R.greedy_decode(enc_ref[b:b + 1], prompts[b], W, dims, rules, max_new)
            assert res.tokens[b] == ref.tokens[0], (seed, b)
            assert abs(float(res.sum_logprob[b]) - ref.sum_logprob[0]) < 5e-3 * max(1, len(ref.tokens[0]))
    else:
        n_prev = int(rng.integers(0, 10))
        prev = ([st.sot_prev] + rng.integers(0, text_hi, size=n_prev).tolist()) if n_prev else []
        prompt = prev + tail
        sot_index = prompt.index(st.sot)
        max_new = int(rng.integers(2, pd.n_text_ctx - len(prompt)))
        opts = e.gen_opts(max_new, timestamps, suppress=suppress, begin_suppress=begin_suppress, sot_index=sot_index,
                          max_initial_timestamp_index=max_initial, suppress_eot=suppress_eot)
        width = int(rng.integers(2, 5))
        A = max(1, min(B, 8 // width))
        if mode == "beam":
            res = e.generate_beam([prompt] * A, width, opts)
            ref = R.beam_decode(enc_ref[:A], prompt, W, dims, rules, width, max_new, no_speech_token=st.no_speech, sot_index=sot_index)
            strip = lambda rows: [[t for t in r if t != st.eot] for r in rows]
            assert strip(res.tokens) == strip(ref.tokens), seed
        else:
            temp, s = float(rng.choice([0.3, 0.7, 1.0])), int(rng.integers(0, 1 << 30))
            res = e.generate_sample([prompt] * A, width, opts, temperature=temp, seed=s)
            ref = R.sample_decode(enc_ref[:A], prompt, W, dims, rules, width, temp, s, max_new)
            assert res.tokens == ref.tokens, seed
        if mode == "beam":
            np.testing.assert_allclose(res.no_speech_prob[:A], ref.no_speech_prob[:A], rtol=2e-3, atol=1e-6)
    e.set_audio_ctx(0)


@pytest.fixture(scope="module", params=["bf16", "f16"])
def ctx_bf16(request):
    """The two 16-bit modes share the fuzz: (engine, preset, dims, oracle weights rounded to the engine's type, tolerance)."""
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F16
    from taiwan_tongues_asr_ce_amd.engine import Engine
    pd = PRESETS["tiny"]
    f16 = request.param == "f16"
    e = Engine(pd, COMPUTE_F16 if f16 else COMPUTE_BF16, 64)
    e.load_weights(synth.iter_weights(pd))
    Wb = R.to_torch(synth.state_dict(pd), round_bf16=not f16, round_f16=f16)
    yield e, pd, R.Dims(**pd.as_dict()), Wb, (0.04 if f16 else 0.15)
    e.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_configuration_bf16_within_tolerance(ctx_bf16, seed):
    """The measured (bf16) mode over random batch sizes 1..64, windows, prompts with a previous-text prefix and rule
    sets: under teacher forcing every greedy choice is within 0.15 of the oracle's best allowed logit.  Seeds 8..11 force
    B = 43 / 48 / 57 / 64: at tiny's 6 heads that is >= 256 (row, head) items, i.e. the single-pass cross-attention kernel
    and two 32-row groups per weight stream in the decode GEMMs (below 43 rows the frame-split kernels run).  The fp16 mode runs
    the same cases at 0.04 (its logit tolerance is 1/4 of bf16's)."""
    e, pd, dims, Wb, tol = ctx_bf16
    st = e.special
    rng = np.random.default_rng(5000 + seed)
    B = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 40]))
    if seed >= 8:
        B = (43, 48, 57, 64)[seed - 8]
    n_ctx = int(rng.choice([1500, 1500, 300, 750]))
    e.set_audio_ctx(n_ctx if n_ctx != 1500 else 0)
    base = [synth.noise_clip(int(rng.integers(0, 50)), n_ctx * 320), synth.tonal_clip(int(rng.integers(0, 50)))[: n_ctx * 320]]
    clips = [base[b % 2] for b in range(B)]
    e.log_mel(clips, want_output=False)
    e.encode(B)
    timestamps = bool(rng.integers(0, 2))
    suppress = sorted(set(rng.integers(0, 50000, size=20).tolist()) | {st.sot, st.sot_prev, st.no_speech, st.transcribe, st.translate})
    n_prev = int(rng.choice([0, 6, 30]))
    prev = ([st.sot_prev] + rng.integers(300, 20000, size=n_prev).tolist()) if n_prev else []
    prompt = prev + [st.sot, st.lang_zh, st.transcribe] + ([] if timestamps else [st.no_timestamps])
    max_new = int(rng.integers(3, 9))
    opts = e.gen_opts(max_new, timestamps, suppress=suppress, no_speech=False)
    res = e.generate([prompt] * B, opts)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin, suppress=suppress,
                    begin_suppress=[220, st.eot], timestamps=timestamps)
    mel = torch.from_numpy(np.stack([R.log_mel(c, pd.n_mels, n_samples=n_ctx * 320) for c in base]))
    enc = R.encoder_forward(mel, Wb, dims)
    for b in sorted({0, B - 1, B // 2}):
        xkv = R.cross_kv(enc[b % 2:b % 2 + 1], Wb, dims)
        cache = R.SelfCache.empty(dims.dec_layers)
        logits = R.decoder_forward(torch.tensor([prompt]), cache, xkv, Wb, dims)[:, -1]
        toks = res.tokens[b]
        assert len(toks) > 0
        for i, t in enumerate(toks):
            s = R.apply_rules(logits[0], toks[:i], rules)
            assert s[t] > -np.inf and float(s.max() - s[t]) < tol, (seed, b, i)
            logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, dims)[:, 0]
    e.set_audio_ctx(0)


@pytest.mark.parametrize("seed", range(12))
def test_random_ragged_beam_matches_per_clip_oracle(ctx, seed):
    """Beam search with one random prompt per clip (different previous-text lengths, so clips leave their prompts at
    different positions) == the oracle's beam search on each clip alone."""
    e, pd, dims, W = ctx
    st = e.special
    rng = np.random.default_rng(9000 + seed)
    width = int(rng.integers(1, 4))
    A = int(rng.integers(2, 8 // width + 1))
    e.set_audio_ctx(0)
    clips = _clips(rng, A, pd.n_audio_ctx)
    want_mel = np.stack([R.log_mel(c, pd.n_mels, n_samples=pd.n_audio_ctx * 320) for c in clips])
    e.log_mel(clips, want_output=False)
    e.encode(A)
    enc_ref = R.encoder_forward(torch.from_numpy(want_mel), W, dims)
    timestamps = bool(rng.integers(0, 2))
    tail = [st.sot, st.lang_zh, st.transcribe] + ([] if timestamps else [st.no_timestamps])
    prompts = []
    for _ in range(A):
        n_prev = int(rng.integers(0, 14))
        prev = ([st.sot_prev] + rng.integers(0, st.eot, size=n_prev).tolist()) if n_prev else []
        prompts.append(prev + tail)
    sots = [p.index(st.sot) for p in prompts]
    max_new = int(rng.integers(2, pd.n_text_ctx - max(len(p) for p in prompts)))
    suppress = sorted({st.sot, st.sot_prev, st.no_speech} | set(rng.integers(0, st.eot, size=5).tolist()))
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin, suppress=suppress,
                    begin_suppress=[st.eot], timestamps=timestamps)
    res = e.generate_beam(prompts, width, e.gen_opts(max_new, timestamps, suppress=suppress, begin_suppress=[st.eot]), sot_index=sots)
    for a in range(A):
        ref = R.beam_decode(enc_ref[a:a + 1], prompts[a], W, dims, rules, width, max_new, no_speech_token=st.no_speech, sot_index=sots[a])
        assert [t for t in res.tokens[a] if t != st.eot] == [t for t in ref.tokens[0] if t != st.eot], (seed, a)
        assert abs(float(res.no_speech_prob[a]) - ref.no_speech_prob[0]) < 2e-3 * max(1.0, ref.no_speech_prob[0]) + 1e-6

"""GPU parity at whisper-large-v3 WIDTH (d 1280, 20 heads, ffn 5120, 128 mels, vocab 51866) with 2+2 layers,
so every kernel runs the shapes of the benchmarked configuration (256x256 GEMM tiles with a ragged last M
tile, 128-bin log-mel + conv stem, 32x32x16 decode GEMMs with K = 1280 / 5120, the 51866-wide vocabulary
projection) against the CPU oracle; and whisper-small (BASELINE.json configs[1]) at reduced depth."""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F32, PRESETS, WhisperDims

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

LARGE2 = WhisperDims("large-v3-2layer", 128, 1500, 1280, 20, 5120, 2, 2, 51866)
SMALL2 = WhisperDims("small-2layer", 80, 1500, 768, 12, 3072, 2, 2, 51865)


def _run(dims, compute, clips, n_new):
    from taiwan_tongues_asr_ce_amd.engine import Engine
    sd = synth.state_dict(dims)
    e = Engine(dims, compute, len(clips))
    e.load_weights(sd.items())
    st = e.special
    mel = e.log_mel(clips)
    enc = e.encode(len(clips), want_output=True)
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    e.decode_reset(len(clips))
    step_logits = [e.decode_step([t] * len(clips)) for t in prompt]
    opts = e.gen_opts(n_new, False, check_interval=1)
    res = e.generate([prompt] * len(clips), opts)
    sup = [opts.suppress[i] for i in range(opts.n_suppress)]
    e.close()
    return sd, st, mel, enc, step_logits, res, prompt, sup


@pytest.mark.parametrize("dims", [LARGE2, SMALL2], ids=["large-v3-width", "small-width"])
def test_width_parity_f32_and_bf16(dims):
    rd = R.Dims(**dims.as_dict())
    clips = [synth.noise_clip(0), synth.tonal_clip(1)]
    # ---- f32 engine vs f32 oracle: north_star tolerance 1e-3 on logits, tokens exact
    sd, st, mel, enc, step_logits, res, prompt, sup = _run(dims, COMPUTE_F32, clips, 6)
    W = R.to_torch(sd)
    mel_ref = np.stack([R.log_mel(c, dims.n_mels) for c in clips])
    np.testing.assert_allclose(mel, mel_ref, atol=2e-4)
    enc_ref = R.encoder_forward(torch.from_numpy(mel_ref), W, rd)
    np.testing.assert_allclose(enc, enc_ref.numpy(), atol=1e-3)
    xkv = R.cross_kv(enc_ref, W, rd)
    cache = R.SelfCache.empty(rd.dec_layers)
    for t, lg in zip(prompt, step_logits):
        want = R.decoder_forward(torch.full((2, 1), t), cache, xkv, W, rd)[:, 0].numpy()
        np.testing.assert_allclose(lg, want, atol=1e-3)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin, suppress=sup,
                    begin_suppress=[220, st.eot], timestamps=False)
    ref = R.greedy_decode(enc_ref, prompt, W, rd, rules, 6)
    assert res.tokens == ref.tokens
    # ---- bf16 engine vs the oracle holding bf16-rounded weights.  Tolerances: encoder (LayerNorm-scale
    # outputs) 0.15 max / 0.012 mean abs; logits 0.08 abs; greedy choice within 0.15 of the oracle's best logit
    sd, st, mel, enc_b, step_logits_b, res_b, prompt, sup = _run(dims, COMPUTE_BF16, clips, 6)
    Wb = R.to_torch(sd, round_bf16=True)
    enc_rb = R.encoder_forward(torch.from_numpy(mel_ref), Wb, rd)
    err = np.abs(enc_b - enc_rb.numpy())
    assert err.max() < 0.15 and err.mean() < 0.012, (err.max(), err.mean())
    xkv = R.cross_kv(enc_rb, Wb, rd)
    cache = R.SelfCache.empty(rd.dec_layers)
    logits = None
    for t, lg in zip(prompt, step_logits_b):
        logits = R.decoder_forward(torch.full((2, 1), t), cache, xkv, Wb, rd)[:, 0]
        assert np.abs(lg - logits.numpy()).max() < 0.08
    # teacher-forced on the engine's tokens: every choice within 0.15 of the oracle's best allowed logit AND equal to the oracle's
    # token wherever its top-2 margin exceeds 0.16 (2 x the logit tolerance); most steps carry such a margin
    from oracle_checks import teacher_forced
    g = teacher_forced(res_b.tokens, prompt, enc_rb, Wb, rd, rules, tol=0.15, margin=0.16)
    assert g.n_steps >= 2 * 4 and g.n_clear >= 0.6 * g.n_steps, g


@pytest.mark.gpu
def test_flash_attention_matches_the_plain_attention_kernel_at_ragged_windows():
    """The MFMA flash kernel (kernels_flash.hip: 64-key tiles, last tile masked, softmax denominator summed by an all-ones
    MFMA over the bf16-rounded weights) against the one-query-per-wave kernel (option `flash = 0`) on the same bf16 q, k, v:
    the 2-layer encoder output agrees to bf16 rounding at windows that end inside a key tile (1500 = 23 x 64 + 28; 150),
    exactly on one (64, 128), on a single partial tile (20) and on an even tile count (200)."""
    from taiwan_tongues_asr_ce_amd.engine import Engine
    dims = PRESETS["large-v3-w2"]
    clips = [synth.tonal_clip(0), synth.noise_clip(1), synth.tonal_clip(2)]
    windows = (0, 20, 64, 128, 150, 200)
    outs = {}
    for plain in (False, True):
        e = Engine(dims, COMPUTE_BF16, 3)
        e.set_option("flash", 0 if plain else 1)
        e.load_weights(synth.iter_weights(dims))
        for n_ctx in windows:
            e.set_audio_ctx(n_ctx)
            e.log_mel(clips, want_output=False)
            outs[plain, n_ctx] = e.encode(3, want_output=True).copy()
        e.close()
    for n_ctx in windows:
        a, b = outs[False, n_ctx], outs[True, n_ctx]
        assert np.isfinite(a).all()
        err = np.abs(a - b)
        assert err.max() < 0.05 and err.mean() < 0.004, (n_ctx, float(err.max()), float(err.mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("compute", ["bf16", "f16"])
def test_flash_attention_with_a_wide_score_range(compute):
    """Round 5: the flash kernel keeps its softmax reference INSIDE the score MFMA (the accumulator starts at -m_ref) and moves
    it lazily - only when a tile's maximum exceeds it by more than 6 exp2 units (kernels_flash.hip).  With LayerNorm-scale
    activations the reference moves in the first tiles and then rarely, so this test MAKES the scores wide: the query
    projections of both encoder layers are scaled x 6 (scores spread over several dozen exp2 units; the reference of most rows
    moves many times, o / l / the tile's scores are rescaled each time), at Whisper's window, at a window that ends inside a key
    tile and at a single partial tile.  Graded against the plain attention
    kernel (option `flash = 0`: one query per wave, exact two-pass softmax) on the same 16-bit q / k / v, and against the f32
    oracle holding the same rounded weights."""
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_F16
    from taiwan_tongues_asr_ce_amd.engine import Engine
    ct = COMPUTE_F16 if compute == "f16" else COMPUTE_BF16
    dims = PRESETS["large-v3-w2"]
    rd = R.Dims(**dims.as_dict())
    sd = dict(synth.state_dict(dims))
    for layer in (0, 1):
        for suffix in (".weight", ".bias"):
            k = f"model.encoder.layers.{layer}.self_attn.q_proj{suffix}"
            sd[k] = (sd[k] * np.float32(6.0)).astype(np.float32)
    clips = [synth.tonal_clip(0), synth.noise_clip(1), synth.burst_clip(2)]
    windows = (0, 150, 20)
    outs = {}
    for plain in (False, True):
        e = Engine(dims, ct, 3)
        e.set_option("flash", 0 if plain else 1)
        e.load_weights(sd.items())
        for n_ctx in windows:
            e.set_audio_ctx(n_ctx)
            e.log_mel(clips, want_output=False)
            outs[plain, n_ctx] = e.encode(3, want_output=True).copy()
        e.close()
    # measured: bf16 0.039 max / 0.0045 mean at the 30-s window (sharper softmax rows than the unscaled test above: 0.05 / 0.004
    # there), fp16 a quarter of that
    tol_max, tol_mean = (0.06, 0.006) if compute == "bf16" else (0.02, 0.0015)
    for n_ctx in windows:
        a, b = outs[False, n_ctx], outs[True, n_ctx]
        assert np.isfinite(a).all()
        err = np.abs(a - b)
        assert err.max() < tol_max and err.mean() < tol_mean, (compute, n_ctx, float(err.max()), float(err.mean()))
    # and the oracle (f32 arithmetic on the same rounded weights) at the full window: the bf16 / fp16 encoder tolerances of this file
    W = R.to_torch(sd, round_bf16=compute == "bf16", round_f16=compute == "f16")
    from oracle_checks import encode_chunked
    mel_ref = np.stack([R.log_mel(c, dims.n_mels) for c in clips])
    enc_ref = encode_chunked(mel_ref, W, rd).numpy()
    err = np.abs(outs[False, 0] - enc_ref)
    lim = (0.15, 0.012) if compute == "bf16" else (0.04, 0.004)
    assert err.max() < lim[0] and err.mean() < lim[1], (compute, float(err.max()), float(err.mean()))

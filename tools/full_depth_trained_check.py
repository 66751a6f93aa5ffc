"""One-off parity run (round 6): the FULL whisper-large-v3 geometry (32 + 32 layers, B = 32) on the "trained" weight profile of
synth.py (heavy-tailed matrices, LayerNorm outlier channels, massive residual channels, an attention sink).  The suite holds this
profile at micro / tiny (HF goldens) and at large-v3 width with 2 + 2 layers (tests/test_gpu_trained_weights.py); this script checks
that nothing changes through 64 layers.  Rows ROWS of the batch are recomputed by the CPU oracle:
  f32 engine : encoder output (abs / relative error), step logits of the prompt positions (north-star tolerance 1e-3), greedy tokens;
  bf16 engine: 4 + N_NEW greedy tokens graded with one causal oracle pass per row (tolerance 0.15, token equality at margins > 0.16).
One JSON line per engine.    python tools/full_depth_trained_check.py [n_new]"""
import json, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from oracle import whisper_ref as R
from oracle_checks import encode_chunked, teacher_forced_causal
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F32, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine, default_suppress
torch.set_grad_enabled(False)
if (torch.get_num_threads() or 1) > 32:
    torch.set_num_threads(32)
N_NEW = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dims = PRESETS["large-v3"]; rd = R.Dims(**dims.as_dict())
B, ROWS = 32, (0, 13, 31)
t0 = time.time(); sd = synth.state_dict(dims, profile="trained"); t_gen = time.time() - t0
kinds = (synth.noise_clip, synth.tonal_clip, synth.noise_clip, synth.burst_clip)
clips = [kinds[i % 4](100 + i) for i in range(B)]
mel_ref = np.stack([R.log_mel(clips[r], dims.n_mels) for r in ROWS])
for compute, tag in ((COMPUTE_F32, "f32"), (COMPUTE_BF16, "bf16")):
    W = R.to_torch(sd, round_bf16=compute == COMPUTE_BF16)
    t0 = time.time(); enc_ref = encode_chunked(mel_ref, W, rd, chunk=1); t_enc = time.time() - t0
    e = Engine(dims, compute, B)
    e.load_weights(sd.items())
    st = e.special
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    e.log_mel(clips, want_output=False)
    enc = e.encode(B, want_output=True)[list(ROWS)]
    err = np.abs(enc - enc_ref.numpy())
    out = {"engine": tag, "weights": "synth profile trained, large-v3 32 + 32 layers", "rows": list(ROWS), "weight_generation_s": round(t_gen, 1),
           "oracle_encoder_s": round(t_enc, 1), "encoder_abs_max_ref": round(float(np.abs(enc_ref.numpy()).max()), 2),
           "encoder_err_max": float(err.max()), "encoder_err_mean": float(err.mean()),
           "encoder_err_max_rel_to_1e-3+1e-4|x|": float((err / (1e-3 + 1e-4 * np.abs(enc_ref.numpy()))).max())}
    e.decode_reset(B)
    xkv = R.cross_kv(enc_ref, W, rd)
    cache = R.SelfCache.empty(rd.dec_layers)
    worst = 0.0
    for t in prompt:
        lg = e.decode_step([t] * B)[list(ROWS)]
        want = R.decoder_forward(torch.full((len(ROWS), 1), t), cache, xkv, W, rd)[:, 0].numpy()
        worst = max(worst, float(np.abs(lg - want).max()))
    out["prompt_logits_err_max"] = worst
    out["logits_std"] = float(want.std())
    opts = e.gen_opts(N_NEW, False, suppress_eot=True, check_interval=1 << 20)
    res = e.generate([prompt] * B, opts)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin, suppress=default_suppress(st, rd.vocab),
                    begin_suppress=[220, st.eot], timestamps=False)
    rules.suppress_eot = True
    tol, margin = (1e-3, 2e-3) if compute == COMPUTE_F32 else (0.15, 0.16)
    try:
        g = teacher_forced_causal([res.tokens[r] for r in ROWS], prompt, enc_ref, W, rd, rules, tol=tol, margin=margin, rows_per_pass=1)
        out["teacher_forced"] = {"steps": g.n_steps, "clear_margin_steps": g.n_clear, "worst_gap": g.worst, "tol": tol, "margin": margin, "passed": True}
    except AssertionError as ex:
        out["teacher_forced"] = {"passed": False, "error": str(ex)[:300], "tol": tol}
    out["finite"] = bool(np.isfinite(res.sum_logprob).all())
    print(json.dumps(out), flush=True)
    e.close()
    del W

"""A/B of the encoder's flash-attention forms on one device (round 6): option flash_qw = 1 (32 queries per wave, the round-5
kernel) against 2 (64 queries per wave).  Interleaved rounds; isolated relaunch loops (ttasr_bench_kernel "enc_attn"), the in-situ
attention class of a real encoder pass, the encoder phase, and whether the two forms give the same encoder output bit for bit."""
import json, sys
sys.path.insert(0, '.')
import numpy as np
from taiwan_tongues_asr_ce_amd import _lib
if len(sys.argv) > 3:
    _lib.LIB_PATH = sys.argv[3]          # a variant build of the library
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dims = PRESETS[name]
for ct, tag in ((COMPUTE_BF16, "bf16"), (COMPUTE_F16, "f16")):
    e = Engine(dims, ct, B)
    e.load_weights(synth.iter_weights(dims))
    clips = [synth.noise_clip(i) if i % 3 else synth.tonal_clip(i) for i in range(B)]
    e.log_mel(clips, want_output=False)
    outs = {}
    for rnd in range(3):
        for qw in (1, 2):
            e.set_option("flash_qw", qw)
            enc = e.encode(B, want_output=(rnd == 0))
            if rnd == 0:
                outs[qw] = enc.copy()
            e.bench_kernel("enc_attn", B, iters=5)
            r = e.bench_kernel("enc_attn", B, iters=40)
            e.set_option("enc_kernel_timing", 1)
            e.encode(B); e.encode(B)
            cls = e.encoder_kernel_ms()
            e.set_option("enc_kernel_timing", 0)
            ph = []
            for _ in range(3):
                e.encode(B); ph.append(e.phase_ms()["encoder"])
            print(json.dumps({"dtype": tag, "round": rnd, "flash_qw": qw, "isolated_us": round(r["ms"] * 1e3, 1),
                              "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1), "in_situ_attention_ms": round(cls["attention"], 3),
                              "encoder_ms": round(min(ph), 2)}), flush=True)
    d = np.abs(outs[1] - outs[2])
    print(json.dumps({"dtype": tag, "encoder_output_bit_identical": bool(np.array_equal(outs[1], outs[2])), "max_abs_diff": float(d.max()),
                      "finite": bool(np.isfinite(outs[2]).all())}), flush=True)
    e.close()

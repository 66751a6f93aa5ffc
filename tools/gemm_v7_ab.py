"""A/B of the encoder GEMM forms inside ONE process (options are per context): the shipped persistent 256 x 256 kernel (v5 / v4)
against option enc_gemm = 7 (v7: 128 x 256 tiles on 4 waves, TWO workgroups per CU - the "ping-pong by occupancy" form).
Interleaved rounds; isolated relaunch loops (ttasr_bench_kernel), the in-situ class timer and the encoder / cross-KV phases;
bit-identity of the encoder output and of the cross-KV cache between the two forms.
NEEDS A LAB BUILD: the v7 kernel was measured slower and is not in the library (source: tools/microbench/
gemm_v7_two_workgroups_per_cu.hip.txt; paste it back into csrc/kernels_gemm.hip with its dispatch to re-run).
    python tools/gemm_v7_ab.py [rounds [variant libttasr.so]]"""
import json, sys
sys.path.insert(0, '.')
import numpy as np
from taiwan_tongues_asr_ce_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = sys.argv[2]          # a variant build of the library (lab macro TTASR_V7_PRIO)
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import PRESETS, COMPUTE_BF16
from taiwan_tongues_asr_ce_amd.engine import Engine
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dims = PRESETS["large-v3"]; B = 32
rng = np.random.default_rng(0)
pool = rng.standard_normal(1 << 22).astype(np.float32)
def fast_weights():
    for name, shape, kind in synth.tensor_specs(dims):
        n = int(np.prod(shape))
        if kind in ("gamma",): a = 1.0 + 0.1 * np.resize(pool, n)
        elif kind == "sinusoid": a = synth.make_tensor(name, shape, kind).ravel()
        else: a = np.resize(pool, n) * (0.02 if kind != "linear" else 1.0 / np.sqrt(shape[1]))
        yield name, a.reshape(shape).astype(np.float32)
e = Engine(dims, COMPUTE_BF16, B)
e.load_weights(fast_weights())
clips = [synth.noise_clip(i) if i % 2 else synth.tonal_clip(i) for i in range(B)]
e.log_mel(clips, want_output=False)
KERNELS = ["enc_gemm_qkv", "enc_gemm_out", "enc_gemm_fc1", "enc_gemm_fc2"]
FLOPS = {}
outs = {}
for form in (0, 7):
    e.set_option("enc_gemm", form)
    e.log_mel(clips, want_output=False)
    enc = e.encode(B, want_output=True)
    k0, v0 = e.cross_kv(0, 0, B), e.cross_kv(0, 1, B)
    k31, v31 = e.cross_kv(dims.dec_layers - 1, 0, B), e.cross_kv(dims.dec_layers - 1, 1, B)
    outs[form] = (enc.copy(), k0.copy(), v0.copy(), k31.copy(), v31.copy())
same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[7]))
print(json.dumps({"bit_identical_encoder_output_and_cross_kv": bool(same), "finite": bool(np.isfinite(outs[7][0]).all()),
                  "max_abs_diff": float(max(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() for a, b in zip(outs[0], outs[7])))}), flush=True)
for rnd in range(rounds):
    for form in (0, 7):
        e.set_option("enc_gemm", form)
        row = {"round": rnd, "form": "v7 128x256 x 2 workgroups per CU" if form == 7 else "shipped (v5 / v4, 256x256 x 1 workgroup per CU)"}
        tf = wsum = 0.0
        for name in KERNELS:
            e.bench_kernel(name, B, iters=3)
            r = e.bench_kernel(name, B, iters=10)
            row[name + "_us"] = round(r["ms"] * 1e3, 1)
            row[name + "_tflops"] = round(r["flops"] / r["ms"] / 1e9, 1)
            tf += r["flops"]; wsum += r["ms"]
        row["flop_weighted_tflops"] = round(tf / wsum / 1e9, 1)
        e.set_option("enc_kernel_timing", 1)
        e.encode(B); e.encode(B)
        row["in_situ_ms"] = {k: round(v, 3) for k, v in e.encoder_kernel_ms().items()}
        e.set_option("enc_kernel_timing", 0)
        ph = []
        for _ in range(3):
            e.encode(B); ph.append(e.phase_ms())
        row["encoder_ms"] = round(min(p["encoder"] for p in ph), 2); row["cross_kv_ms"] = round(min(p["cross_kv"] for p in ph), 2)
        print(json.dumps(row), flush=True)
e.close()

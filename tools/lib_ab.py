"""A/B of library BUILDS on one device (round 5): every variant .so runs in its own process (the ctypes handle is a singleton),
rounds are interleaved, each process warms up and reports the kernels asked for through ttasr_bench_kernel plus the in-situ
encoder classes.  Usage:  python tools/lib_ab.py --libs taiwan_tongues_asr_ce_amd/libttasr.so tools/microbench/bin/X.so
                                                 --kernels enc_attn enc_gemm_qkv --rounds 3"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, os, time
sys.path.insert(0, %(root)r)
import numpy as np
from taiwan_tongues_asr_ce_amd import _lib
_lib.LIB_PATH = %(lib)r
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import PRESETS, COMPUTE_BF16, COMPUTE_F16
from taiwan_tongues_asr_ce_amd.engine import Engine
dims = PRESETS[%(model)r]; B = %(batch)d
rng = np.random.default_rng(0)
pool = rng.standard_normal(1 << 22).astype(np.float32)
def fast_weights():
    for name, shape, kind in synth.tensor_specs(dims):
        n = int(np.prod(shape))
        if kind in ("gamma",): a = 1.0 + 0.1 * np.resize(pool, n)
        elif kind == "sinusoid": a = synth.make_tensor(name, shape, kind).ravel()
        else: a = np.resize(pool, n) * (0.02 if kind != "linear" else 1.0 / np.sqrt(shape[1]))
        yield name, a.reshape(shape).astype(np.float32)
e = Engine(dims, COMPUTE_F16 if %(f16)d else COMPUTE_BF16, B)
e.load_weights(fast_weights())
e.log_mel([synth.noise_clip(i) for i in range(B)], want_output=False)
e.encode(B)
enc = e.encode(B, want_output=True)
out = {"lib": os.path.basename(%(lib)r)}
if %(probe)r:
    np.save(os.path.join(%(probe)r, "enc_" + out["lib"] + ".npy"), enc[::4, ::7, ::5])
out["enc_finite"] = bool(np.isfinite(enc).all()); out["enc_abs_mean"] = round(float(np.abs(enc).mean()), 5)
for name in %(kernels)r:
    e.bench_kernel(name, B, iters=5)
    r = e.bench_kernel(name, B, iters=%(iters)d)
    out[name] = round(r["ms"] * 1e3, 2)
e.set_option("enc_kernel_timing", 1)
e.encode(B); e.encode(B)
out["in_situ_ms"] = {k: round(v, 3) for k, v in e.encoder_kernel_ms().items()}
e.set_option("enc_kernel_timing", 0)
ph = []
for _ in range(3):
    e.encode(B); ph.append(e.phase_ms())
out["encoder_ms"] = round(min(p["encoder"] for p in ph), 2); out["cross_kv_ms"] = round(min(p["cross_kv"] for p in ph), 2)
print(json.dumps(out), flush=True)
e.close()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--kernels", nargs="*", default=["enc_attn"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--f16", action="store_true")
    ap.add_argument("--probe-dir", default="", help="every library's encoder output (strided sample) is saved here and compared with the first library's")
    a = ap.parse_args()
    for r in range(a.rounds):
        for lib in a.libs:
            code = CHILD % dict(root=ROOT, lib=os.path.abspath(lib), model=a.model, batch=a.batch, kernels=list(a.kernels),
                                iters=a.iters, f16=int(a.f16), probe=a.probe_dir)
            p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else json.dumps({"lib": lib, "error": p.stderr[-400:]}), flush=True)
    if a.probe_dir:
        import numpy as np
        ref = np.load(os.path.join(a.probe_dir, "enc_" + os.path.basename(a.libs[0]) + ".npy"))
        for lib in a.libs[1:]:
            x = np.load(os.path.join(a.probe_dir, "enc_" + os.path.basename(lib) + ".npy"))
            print(json.dumps({"lib": os.path.basename(lib), "encoder_output_vs_first_lib": {"max_abs": float(np.abs(x - ref).max()),
                                                                                           "mean_abs": float(np.abs(x - ref).mean())}}), flush=True)


if __name__ == "__main__":
    main()

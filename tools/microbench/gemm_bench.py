import sys, os, time; sys.path.insert(0, '.')
import numpy as np
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import PRESETS, COMPUTE_BF16
from taiwan_tongues_asr_ce_amd.engine import Engine
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
t = time.time(); e.load_weights(fast_weights()); print("load", time.time() - t)
e.log_mel([synth.noise_clip(i) for i in range(B)], want_output=False)
e.encode(B)
for name in sys.argv[1:] or ["enc_gemm_qkv", "enc_gemm_fc1", "enc_gemm_fc2", "enc_attn"]:
    r = e.bench_kernel(name, B, iters=10)
    print(f"{name}: {r['ms']*1e3:.1f} us  {r['flops']/r['ms']/1e9:.1f} TF/s  {r['bytes']/r['ms']/1e6:.1f} GB/s")
print(e.phase_ms())

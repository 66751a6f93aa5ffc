"""Probe (round 5): does the encoder gain from running as two half-batch chains on two streams?  Two engine contexts (own
streams) encode 16 clips each from two host threads, against one context encoding 32: if the GPU fills one chain's GEMM tail
rounds and its attention / LayerNorm launches with the other chain's workgroups, the pair finishes sooner than the single chain."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import PRESETS, COMPUTE_BF16
from taiwan_tongues_asr_ce_amd.engine import Engine

dims = PRESETS["large-v3"]
rng = np.random.default_rng(0)
pool = rng.standard_normal(1 << 22).astype(np.float32)


def fast_weights():
    for name, shape, kind in synth.tensor_specs(dims):
        n = int(np.prod(shape))
        if kind in ("gamma",): a = 1.0 + 0.1 * np.resize(pool, n)
        elif kind == "sinusoid": a = synth.make_tensor(name, shape, kind).ravel()
        else: a = np.resize(pool, n) * (0.02 if kind != "linear" else 1.0 / np.sqrt(shape[1]))
        yield name, a.reshape(shape).astype(np.float32)


W = list(fast_weights())
clips = [synth.noise_clip(i) for i in range(32)]


def mk(B, lo):
    e = Engine(dims, COMPUTE_BF16, B)
    e.load_weights(W)
    e.log_mel(clips[lo:lo + B], want_output=False)
    e.encode(B)
    return e


one = mk(32, 0)
pair = [mk(16, 0), mk(16, 16)]


def t_one(n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); one.encode(32); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def t_pair(n=5, stagger_ms=0.0):
    ts = []
    for _ in range(n):
        def run(e, delay):
            if delay: time.sleep(delay)
            e.encode(16)
        th = [threading.Thread(target=run, args=(pair[0], 0.0)), threading.Thread(target=run, args=(pair[1], stagger_ms * 1e-3))]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for r in range(3):
    a = t_one(); b = t_pair(); c = t_pair(stagger_ms=1.0)
    h = []
    t0 = time.perf_counter(); pair[0].encode(16); h.append(time.perf_counter() - t0)
    print(json.dumps({"one_context_32_clips_ms": round(a, 2), "two_contexts_16_clips_each_ms": round(b, 2),
                      "two_contexts_second_starts_1ms_later_ms": round(c, 2), "one_context_16_clips_alone_ms": round(h[0] * 1e3, 2)}), flush=True)

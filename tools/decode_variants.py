#!/usr/bin/env python3
"""A/B timing of the decode launch plans on one GPU (large-v3 geometry, B = 32, bf16, 4 + 128 greedy tokens):
every variant is a set of ttasr_set_option overrides (the library reads no environment variable), so one process times
them all on the same clips.
Also checks that each variant is bit-reproducible (three replays give identical tokens and scores) and reports
how many rows agree token-for-token between variants.  One JSON line per variant.

    python tools/decode_variants.py [--variants name,name] [--batch 32] [--new-tokens 128]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "auto": {},                                            # default plan: K slices chosen automatically (d 4, q 4, qkv 2, fc2 8)
    "steps_for_prompt": {"prefill_ns_min": 16},          # round-2 default: the 4-token prompt fed as four decode steps
    "d5": {"ksplit_out": 5, "ksplit_q": 5, "ksplit_qkv": 2, "ksplit_fc2": 8},
    "f16": {"ksplit_out": 4, "ksplit_q": 4, "ksplit_qkv": 2, "ksplit_fc2": 16},
    "f10": {"ksplit_out": 4, "ksplit_q": 4, "ksplit_qkv": 2, "ksplit_fc2": 10},
    "qkv4": {"ksplit_out": 4, "ksplit_q": 4, "ksplit_qkv": 4, "ksplit_fc2": 8},
    # round 6 (the activation tile now comes through LDS: the K-split optimum may have moved)
    "d4": {"ksplit_out": 4}, "d8": {"ksplit_out": 8}, "d10": {"ksplit_out": 10}, "d2": {"ksplit_out": 2},
    "f4": {"ksplit_fc2": 4}, "f5": {"ksplit_fc2": 5}, "f16only": {"ksplit_fc2": 16},
    "qkv1": {"ksplit_qkv": 1}, "qkv4only": {"ksplit_qkv": 4}, "q2": {"ksplit_q": 2}, "q1": {"ksplit_q": 1},
    "d4qkv1": {"ksplit_out": 4, "ksplit_qkv": 1}, "d4qkv1q2": {"ksplit_out": 4, "ksplit_qkv": 1, "ksplit_q": 2},
    "d4qkv1f16": {"ksplit_out": 4, "ksplit_qkv": 1, "ksplit_fc2": 16}, "d2qkv1": {"ksplit_out": 2, "ksplit_qkv": 1},
    "x_regs": {"dec_x_lds": 0},
    "vocab_generic": {"vocab_persistent": 0},             # round-2 vocabulary GEMM: one workgroup per 32 outputs
    "w_plain": {"weights_nontemporal": 0},
    "xattn_plain": {"xattn_nontemporal": 0},
}
DEFAULTS = {"dec_x_lds": 1, "vocab_persistent": 1, "prefill_ns_min": 2, "ksplit_out": 0, "ksplit_q": 0, "ksplit_qkv": 0, "ksplit_fc2": 0, "weights_nontemporal": 1,
            "xattn_nontemporal": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--option", action="append", default=[], help="extra key=value (ttasr_set_option) applied to every variant")
    ap.add_argument("--xattn-sweep", action="store_true",
                    help="time the cross-attention kernel with nontemporal / plain loads (option xattn_nontemporal 1 / 0) in isolation instead")
    args = ap.parse_args()
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
    from taiwan_tongues_asr_ce_amd.engine import Engine

    dims = PRESETS[args.model]
    B = args.batch
    clips = [synth.noise_clip(b) for b in range(B)]
    weights = list(synth.iter_weights(dims))
    if args.xattn_sweep:
        e = Engine(dims, COMPUTE_BF16, B)
        e.load_weights(weights)
        e.log_mel(clips, want_output=False)
        e.encode(B)
        for rep in range(2):
            for v in (1, 0):
                e.set_option("xattn_nontemporal", v)
                k = e.bench_kernel("xattn", B, iters=96)
                print(json.dumps({"xattn_variant": v, "nontemporal": bool(v & 1), "us": round(k["ms"] * 1e3, 2),
                                  "TBps": round(k["bytes"] / k["ms"] / 1e9, 3)}), flush=True)
        e.close()
        return
    first = None
    for name in args.variants.split(","):
        e = Engine(dims, COMPUTE_BF16, B)
        for k, v in {**DEFAULTS, **VARIANTS[name]}.items():   # two of the options are process-wide: reset them per variant
            e.set_option(k, v)
        for kv in args.option:
            k, v = kv.split("=", 1)
            e.set_option(k, int(v))
        e.load_weights(weights)
        st = e.special
        e.log_mel(clips, want_output=False)
        e.encode(B)
        prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
        opts = e.gen_opts(args.new_tokens, timestamps=False, suppress_eot=True, no_speech=True, check_interval=1 << 20)
        runs, ms, wall = [], [], []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            r = e.generate([prompt] * B, opts)
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(e.phase_ms()["decode"])
            runs.append(r)
        same = all(r.tokens == runs[0].tokens and np.array_equal(r.sum_logprob, runs[0].sum_logprob) for r in runs[1:])
        if first is None:
            first = runs[0]
        agree = float(np.mean([a == b for a, b in zip(first.tokens, runs[0].tokens)]))
        n_steps = 4 + args.new_tokens - 1
        print(json.dumps({"variant": name, "options": VARIANTS[name], "decode_ms": round(min(ms[1:]), 2),
                          "ms_per_step": round(min(ms[1:]) / n_steps, 4), "wall_ms": round(min(wall[1:]), 2),
                          "bit_reproducible": bool(same), "rows_equal_to_first_variant": agree}), flush=True)
        e.close()


if __name__ == "__main__":
    main()

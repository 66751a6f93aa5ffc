#!/usr/bin/env python3
"""A/B of the encoder GEMM kernels on one GPU (large-v3 geometry, B = 32): option enc_gemm = 3 (256x256 tiles, one workgroup per
tile) against 4 (the same tile body as persistent workgroups) and 0 (round 5: the automatic choice = persistent workgroups with the
last partial round re-tiled into shorter tiles where that pays), interleaved rounds in one process, plus a bit-identity check of the
whole encoder output and the in-situ class times of a real encoder pass under each.

    python tools/gemm_ab.py [--rounds 4] [--batch 32]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--variants", default="3,4,0")
    ap.add_argument("--no-insitu", action="store_true")
    args = ap.parse_args()
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
    from taiwan_tongues_asr_ce_amd.engine import Engine
    dims = PRESETS[args.model]
    B = args.batch
    e = Engine(dims, COMPUTE_BF16, B)
    e.load_weights(synth.iter_weights(dims))
    clips = [synth.noise_clip(b) for b in range(B)]
    e.log_mel(clips, want_output=False)
    variants = [int(v) for v in args.variants.split(",")]
    outs = {}
    for v in variants:
        e.set_option("enc_gemm", v)
        outs[v] = e.encode(B, want_output=True)
    ref = outs[variants[0]]
    for v in variants[1:]:
        d = np.abs(outs[v].astype(np.float64) - ref).max()
        print(json.dumps({"check": f"encoder output enc_gemm={v} vs {variants[0]}", "max_abs_diff": float(d),
                          "bit_identical": bool(np.array_equal(outs[v], ref))}), flush=True)
    names = ("enc_gemm_qkv", "enc_gemm_out", "enc_gemm_fc1", "enc_gemm_fc2")
    for rnd in range(args.rounds):
        for v in variants:
            e.set_option("enc_gemm", v)
            row = {"round": rnd, "enc_gemm": v}
            tot_f, tot_t = 0.0, 0.0
            for n in names:
                e.bench_kernel(n, B, iters=5)
                r = e.bench_kernel(n, B, iters=20)
                row[n] = {"us": round(r["ms"] * 1e3, 1), "tflops": round(r["flops"] / r["ms"] / 1e9, 1)}
                row["signature_" + n] = r.get("signature")
                tot_f += r["flops"]; tot_t += r["ms"]
            row["flop_weighted_tflops"] = round(tot_f / tot_t / 1e9, 1)
            print(json.dumps(row), flush=True)
    for v in ([] if args.no_insitu else variants):     # in situ: one real encoder pass with an event after every launch
        e.set_option("enc_gemm", v)
        e.set_option("enc_kernel_timing", 1)
        e.encode(B); e.encode(B)
        km = e.encoder_kernel_ms()
        e.set_option("enc_kernel_timing", 0)
        ph = []
        for _ in range(3):
            e.encode(B)
            ph.append(e.phase_ms())
        print(json.dumps({"in_situ": True, "enc_gemm": v, "class_ms": {k: round(x, 3) for k, x in km.items()},
                          "encoder_ms": round(float(np.median([p["encoder"] for p in ph])), 2),
                          "cross_kv_ms": round(float(np.median([p["cross_kv"] for p in ph])), 2)}), flush=True)
    e.close()


if __name__ == "__main__":
    main()

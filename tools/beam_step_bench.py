#!/usr/bin/env python3
"""Beam-search decode time on one GPU (large-v3 geometry, bf16): A clips x beam hypotheses, short prompt, N new tokens.
One JSON line per configuration; `--option xsplit=0` selects the one-workgroup-per-row cross-attention (ttasr_set_option).

    python tools/beam_step_bench.py [--clips 6] [--beam 5] [--new-tokens 32]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--clips", default="1,6")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--new-tokens", type=int, default=32)
    ap.add_argument("--option", action="append", default=[], help="key=value kernel-selection override (ttasr_set_option)")
    args = ap.parse_args()
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
    from taiwan_tongues_asr_ce_amd.engine import Engine

    dims = PRESETS[args.model]
    weights = list(synth.iter_weights(dims))
    for A in [int(x) for x in args.clips.split(",")]:
        e = Engine(dims, COMPUTE_BF16, A * args.beam)
        e.load_weights(weights)
        for kv in args.option:
            e.set_option(kv.split("=", 1)[0], int(kv.split("=", 1)[1]))
        st = e.special
        e.log_mel([synth.noise_clip(b) for b in range(A)], want_output=False)
        e.encode(A)
        prompt = [st.sot, st.lang_zh, st.transcribe]
        opts = e.gen_opts(args.new_tokens, True)
        ms, toks = [], None
        for _ in range(4):
            t0 = time.perf_counter()
            r = e.generate_beam([prompt] * A, args.beam, opts)
            ms.append((time.perf_counter() - t0) * 1e3)
            assert toks is None or r.tokens == toks
            toks = r.tokens
        n_steps = max(len(t) for t in toks) + len(prompt)
        print(json.dumps({"clips": A, "beam": args.beam, "rows": A * args.beam, "wall_ms": round(min(ms[1:]), 2),
                          "steps_upper_bound": n_steps, "ms_per_step": round(min(ms[1:]) / n_steps, 3),
                          "host_split_last_run": {k: round(v, 2) for k, v in e.beam_profile().items()},
                          "options": args.option}), flush=True)
        e.close()


if __name__ == "__main__":
    main()

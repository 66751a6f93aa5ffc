#!/usr/bin/env python3
"""Long previous-text prompts (condition_on_previous_text): time of a greedy generate with a P-token prompt + N new tokens at
B clips (large-v3 geometry, bf16).  `--option xsplit=0` selects one cross-attention workgroup per (row, head) everywhere (ttasr_set_option).

    python tools/prefill_bench.py [--batch 8] [--prompt 224] [--new-tokens 32]
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
    ap.add_argument("--batch", default="1,8")
    ap.add_argument("--prompt", default="24,224")
    ap.add_argument("--new-tokens", type=int, default=32)
    ap.add_argument("--option", action="append", default=[], help="key=value kernel-selection override (ttasr_set_option)")
    args = ap.parse_args()
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
    from taiwan_tongues_asr_ce_amd.engine import Engine

    dims = PRESETS[args.model]
    weights = list(synth.iter_weights(dims))
    for B in [int(x) for x in args.batch.split(",")]:
        e = Engine(dims, COMPUTE_BF16, B)
        e.load_weights(weights)
        for kv in args.option:
            e.set_option(kv.split("=", 1)[0], int(kv.split("=", 1)[1]))
        st = e.special
        e.log_mel([synth.noise_clip(b) for b in range(B)], want_output=False)
        e.encode(B)
        for P in [int(x) for x in args.prompt.split(",")]:
            prompt = [st.sot_prev] + [1000 + i for i in range(P - 4)] + [st.sot, st.lang_zh, st.transcribe]
            opts = e.gen_opts(args.new_tokens, True, sot_index=P - 3)
            ms, toks = [], None
            for _ in range(4):
                t0 = time.perf_counter()
                r = e.generate([prompt] * B, opts)
                ms.append((time.perf_counter() - t0) * 1e3)
                assert toks is None or r.tokens == toks
                toks = r.tokens
            print(json.dumps({"batch": B, "prompt_tokens": P, "new_tokens": args.new_tokens, "wall_ms": round(min(ms[1:]), 2),
                              "options": args.option}), flush=True)
        e.close()


if __name__ == "__main__":
    main()

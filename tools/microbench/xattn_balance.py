#!/usr/bin/env python3
"""Does the cross-attention kernel lose bandwidth to CU imbalance?  One workgroup per (row, head): B = 32 gives
640 workgroups = 2.5 per CU.  Times the kernel alone (ttasr_bench_kernel) for row counts that put 2, 2.5, 3, 3.5, 4, 5
workgroups on a CU and prints TB/s: a saw-tooth with peaks at the integers means the tail (CUs holding 3 workgroups
while others hold 2) costs bandwidth, not HBM."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from taiwan_tongues_asr_ce_amd import synth  # noqa: E402
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS  # noqa: E402
from taiwan_tongues_asr_ce_amd.engine import Engine  # noqa: E402

dims = PRESETS["large-v3"]
B = 64
e = Engine(dims, COMPUTE_BF16, B)
e.load_weights(synth.iter_weights(dims))
e.log_mel([synth.noise_clip(b) for b in range(B)], want_output=False)
e.encode(B)
for rep in range(2):
    for rows in (13, 19, 26, 29, 32, 35, 38, 39, 45, 51, 52, 58, 64):
        k = e.bench_kernel("xattn", rows, iters=64)
        print(json.dumps({"rows": rows, "workgroups_per_cu": round(rows * 20 / 256, 2), "us": round(k["ms"] * 1e3, 2),
                          "TBps": round(k["bytes"] / k["ms"] / 1e9, 3)}), flush=True)
e.close()

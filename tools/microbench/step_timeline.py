#!/usr/bin/env python3
"""Per-launch timeline of one decode step from a `rocprofv3 --kernel-trace --output-format csv` run.

    python tools/microbench/step_timeline.py <dir with *_kernel_trace.csv> [--step -3]

Finds the decode steps (each ends with select_kernel), takes one of the last ones, and prints for every launch its
duration and the gap to the previous launch's end, then the per-kernel-class sums (durations AND gaps): the sums add up
to the step's wall time, which is what DESIGN.md section 4.10 quotes."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    base = m.group(1) if m else name
    tpl = m.group(2) or "" if m else ""
    tpl = tpl.replace("unsigned short", "bf16").replace(" ", "")
    return base + tpl


def main():
    d = sys.argv[1]
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -3
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[2].startswith("select_kernel")]
    if len(ends) < 4:
        print("not enough decode steps in the trace", len(ends))
        return
    hi = ends[which]
    lo = ends[which - 1] + 1
    step = rows[lo:hi + 1]
    t0 = rows[lo - 1][1]
    wall = step[-1][1] - t0
    dur, gap, cnt = defaultdict(int), defaultdict(int), defaultdict(int)
    prev_end = t0
    lines = []
    for s, e, n in step:
        k = short(n)
        dur[k] += e - s
        gap[k] += s - prev_end
        cnt[k] += 1
        lines.append(f"{k:60s} dur {(e - s) / 1e3:8.2f} us   gap {(s - prev_end) / 1e3:7.2f} us")
        prev_end = e
    print(f"step of {len(step)} launches, wall {wall / 1e3:.1f} us (end of previous select -> end of this select)")
    print("--- first 40 launches (one decoder layer is ~8-11 of them) ---")
    print("\n".join(lines[:40]))
    print("--- last 6 launches ---")
    print("\n".join(lines[-6:]))
    print("--- per kernel class: launches, total duration, total gap before (us) ---")
    for k in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
        print(f"{k:60s} n={cnt[k]:4d}  dur {dur[k] / 1e3:9.1f}  avg {dur[k] / cnt[k] / 1e3:7.2f}  gap {gap[k] / 1e3:8.1f}  avg {gap[k] / cnt[k] / 1e3:6.2f}")
    print(f"sum dur {sum(dur.values()) / 1e3:.1f} us, sum gap {sum(gap.values()) / 1e3:.1f} us")


if __name__ == "__main__":
    main()

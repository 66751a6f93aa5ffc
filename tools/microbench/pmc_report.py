#!/usr/bin/env python3
"""Aggregate three separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES) of `python3 bench.py --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline` into
profiles/<tag>_pmc.json and profiles/xattn_pmc.json (what bench.py reports as roofline.traffic).

    python tools/microbench/pmc_report.py <dir_fetch> <dir_write> <dir_mfma> <tag> <out_dir>

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and counts HALF the bytes of a
16-byte-per-lane streaming read -> bytes = KiB * 1024 * 2; WRITE_SIZE bytes = KiB * 1024.  GRBM_GUI_ACTIVE is summed over the
8 XCDs and SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs -> mfma_busy_frac = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)."""
import collections
import csv
import glob
import json
import os
import re
import sys


SIGNATURES = collections.defaultdict(set)   # label -> exact kernel signatures pooled under it (bench.py's staleness check)


def signature(name, grid):
    """The spelling libttasr's launchers record (ttasr_bench_kernel_signature): name<template arguments> grid <threads>."""
    return re.sub(r"\(.*", "", name).replace("void ", "").strip() + " grid " + str(grid)


def load(d):
    """{label: {counter: [value per launch]}} - template variants of one kernel that share a label are pooled."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                grid = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
                lab = label(r["Kernel_Name"], grid)
                if lab is not None:
                    agg[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    SIGNATURES[lab].add(signature(r["Kernel_Name"], grid))
    return agg


def label(name, grid):
    m = re.search(r"gemm_bf16_v[345]_kernel<(?:[A-Za-z_0-9 ]+, )?(\d+)>", name)   # <storage type, EPI>; v4 = the persistent form (round 4)
    if m:
        epi = int(m.group(1))
        return {0: "enc GEMM bias -> bf16 (EPI 0: qkv, out-proj, fc2)", 1: "enc GEMM bias + GELU -> bf16 (EPI 1: fc1, conv1)",
                8: "cross-KV GEMM -> head-split bf16 (EPI 8)", 18: "enc GEMM bias + f32 residual (EPI 18)",
                21: "conv2-as-GEMM + GELU + positions (EPI 21)"}.get(epi, f"gemm_bf16_v3 EPI {epi}") + f" grid {grid}"
    if "cross_attn_pipe_kernel" in name:
        return "cross_attn_pipe_kernel grid " + grid
    if "cross_attn_decode_kernel" in name:
        return "cross_attn_decode_kernel grid " + grid
    if "enc_attn_flash" in name:
        return "enc_attn_flash_kernel"
    if "gemm_skinny_kernel" in name:
        return re.sub(r"\(.*", "", name).replace("void ", "") + " grid " + grid
    if "layernorm" in name or "self_attn_decode" in name or "select_kernel" in name:
        return re.sub(r"\(.*", "", name).replace("void ", "").replace("unsigned short", "bf16") + " grid " + grid
    return None


def main():
    d_fetch, d_write, d_mfma, tag, out_dir = sys.argv[1:6]
    fetch, write, mfma = load(d_fetch), load(d_write), load(d_mfma)
    kernels = {}
    for lab, c in fetch.items():
        v = c.get("FETCH_SIZE", [])
        if v:
            kernels[lab] = {"launches": len(v), "fetch_bytes_per_launch": round(sum(v) / len(v) * 1024 * 2)}
    for lab, c in write.items():
        v = c.get("WRITE_SIZE", [])
        if v and lab in kernels:
            kernels[lab]["write_bytes_per_launch"] = round(sum(v) / len(v) * 1024)
    for lab, c in mfma.items():
        if lab not in kernels or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        busy, gui = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(c["GRBM_GUI_ACTIVE"])
        if gui > 0 and busy > 0:
            kernels[lab]["mfma_busy_frac"] = round(busy / (gui / 8 * 1024), 4)
    for lab in kernels:
        kernels[lab]["signatures"] = sorted(SIGNATURES.get(lab, ()))
    src = ("rocprofv3 --pmc <counters> (three separate passes: FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES) -- python3 bench.py --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline; per-kernel means "
           "over all launches (tools/microbench/pmc_report.py); gfx950 corrections applied (FETCH_SIZE x2)")
    with open(os.path.join(out_dir, f"{tag}_pmc.json"), "w") as f:
        json.dump({"source": src, "kernels": dict(sorted(kernels.items()))}, f, indent=1)
    xa = [(lab, k) for lab, k in kernels.items() if lab.startswith(("cross_attn_pipe_kernel", "cross_attn_decode_kernel")) and k.get("launches", 0) > 100]
    if xa:
        lab, k = max(xa, key=lambda t: t[1]["launches"])
        algo = 32 * (2 * 1500 * 1280 + 2 * 1280) * 2
        with open(os.path.join(out_dir, "xattn_pmc.json"), "w") as f:
            json.dump({"kernel": lab, "signatures": k.get("signatures", []), "source": src, "launches": k["launches"], "fetch_bytes_per_32row_launch": k["fetch_bytes_per_launch"],
                       "write_bytes_per_32row_launch": k.get("write_bytes_per_launch"), "algorithmic_bytes_per_32row_launch": algo,
                       "traffic_bytes_per_32row_launch": k["fetch_bytes_per_launch"] + (k.get("write_bytes_per_launch") or 0)}, f, indent=1)
    print(json.dumps(kernels, indent=1)[:3000])


if __name__ == "__main__":
    main()

import csv, glob, sys, collections, json
d = sys.argv[1]
out = {}
for f in glob.glob(d + "/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(t in k for t in ("cross_attn", "gemm_bf16_v3", "enc_attn_flash", "gemm_skinny")):
            out[k] = {c: {"mean": sum(x) / len(x), "n": len(x)} for c, x in v.items()}
print(json.dumps(out, indent=1))

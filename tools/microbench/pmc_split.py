"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel means: python pmc_split.py <dir> [<dir> ...] -- substr [substr ...]"""
import collections, csv, glob, json, sys
args = sys.argv[1:]
dirs, subs = args[:args.index("--")], args[args.index("--") + 1:]
out = collections.defaultdict(dict)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for sname in subs:
                if sname in k:
                    for c, x in v.items():
                        out[k[:90]][c] = {"mean": sum(x) / len(x), "n": len(x)}
print(json.dumps(out, indent=1))

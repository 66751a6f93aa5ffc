import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rd = csv.DictReader(open(f))
print(rd.fieldnames)
agg = collections.defaultdict(list)
for r in rd:
    k = r["Kernel_Name"][:50]
    if "skinny" in k:
        k += " grid=" + str(r.get("Grid_Size_X", r.get("Grid_Size", "?")))
    agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:20]:
    print(f"{sum(v)/1e6:9.2f} ms  n={len(v):6d}  avg={sum(v)/len(v)/1e3:8.2f} us  {k}")

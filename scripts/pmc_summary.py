"""Summarise rocprofv3 counter_collection.csv: per (kernel, grid) mean of each counter."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:70]
    key = (name, r["Grid_Size"], )
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, d in sorted(agg.items()):
    if flt and flt not in k[0]: continue
    print(k[0], "grid", k[1], "n", len(next(iter(d.values()))))
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}")

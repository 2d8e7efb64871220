"""Summarise a rocprofv3 --kernel-trace csv: per (kernel, grid) count / avg / total."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r.get("Kernel_Name") or r.get("kernel_name")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = r.get("Workgroup_Size_X", "?")
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    key = (name[:90], grid, wg)
    agg[key][0] += 1; agg[key][1] += dur
tot = sum(v[1] for v in agg.values())
print("total kernel us", tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{v[1]:10.1f} us  n={v[0]:5d}  avg={v[1]/v[0]:8.2f}  {k[0]}  grid={k[1]} wg={k[2]}")

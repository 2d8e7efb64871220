"""Per-step kernel shares of the training step from a rocprofv3 --kernel-trace csv of bench.py --mode train.

usage: train_prof_summary.py <kernel_trace.csv> [marker] [marker_launches_per_step]
Aggregates by kernel name over the SECOND HALF of the trace window (steady state: no first-touch work), counts the
steps in that window from a marker kernel (default: query_fwd_f32*, 5 launches per step = one per stack),
reports launches and microseconds per step, and the busy / idle split of that window."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "query_fwd_f32"
per_step = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
ev = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("kernel_name")
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name).split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:100]))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
mid = (t0 + t1) // 2
half = [e for e in ev if e[0] >= mid]
nsteps = sum(1 for e in half if marker in e[2]) / per_step
agg = collections.defaultdict(lambda: [0, 0.0])
busy, cur_s, cur_e = 0, None, None
for s, e, n in half:
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - mid
tot = sum(v[1] for v in agg.values())
print(f"window {span/1e6:.1f} ms ~ {nsteps:.1f} steps: {span/1e6/nsteps:.2f} ms/step wall, GPU busy {busy/span*100:.1f} %, "
      f"{sum(v[0] for v in agg.values())/nsteps:.0f} launches/step, kernel time {tot/1e3/nsteps:.2f} ms/step")
print(f"{'us/step':>10} {'share':>6} {'n/step':>7} {'avg us':>8}  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1]/nsteps:10.1f} {v[1]/tot*100:5.1f}% {v[0]/nsteps:7.1f} {v[1]/v[0]:8.2f}  {k}")

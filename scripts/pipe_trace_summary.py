"""summary of a rocprofv3 kernel trace (csv) of the pipelined fit: per hardware queue, busy time and kernel count inside the last
`win` ms of the run, and how much of the queues' busy time overlaps -- does the device run the preparation beside the optimisation?"""
import csv
import sys
from collections import defaultdict

path, win_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
rows = []
with open(path) as f:
    r = csv.DictReader(f)
    for x in r:
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x.get("Queue_Id", "?"), x["Kernel_Name"][:60]))
rows.sort()
t1 = rows[-1][1]
t0 = t1 - int(win_ms * 1e6)
rows = [x for x in rows if x[0] >= t0]
byq = defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
print("window %.0f ms, %d kernels, queues:" % (win_ms, len(rows)))
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in v) / 1e6
    names = defaultdict(float)
    for s, e, n in v:
        names[n] += (e - s) / 1e6
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print("  queue %s: %6d kernels, busy %7.1f ms; top: %s" % (q, len(v), busy, "; ".join("%s %.1f" % (n[:40], t) for n, t in top)))
# union / overlap of busy intervals over all queues
ev = []
for s, e, q, n in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, union, multi = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1: union += t - last
    if depth >= 2: multi += t - last
    depth += d; last = t
print("some kernel running: %.1f ms of %.1f; two or more at once: %.1f ms" % (union / 1e6, win_ms, multi / 1e6))

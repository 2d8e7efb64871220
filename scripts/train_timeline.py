"""union-busy / idle of the device over the traced training steps: python scripts/train_timeline.py <results.db> <steps>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2])
rows = c.execute("select start, end, name from kernels order by start").fetchall()
# take the last n steps' worth: split by the optimiser's multi_tensor_apply bursts is fragile -> just use the last 60 % of the trace
rows = rows[int(len(rows) * 0.4):]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, conc = 0, rows[0][0], rows[0][1], 0
ov = 0
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        ov += min(e, cur_e) - s
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print(f"span {span/1e6:.1f} ms  union busy {busy/1e6:.1f} ms ({busy/span:.1%})  sum of kernel times {tot/1e6:.1f} ms  idle {(span-busy)/1e6:.1f} ms")
gaps = []
cur_e = rows[0][1]
for s, e, nme in rows[1:]:
    if s > cur_e: gaps.append((s - cur_e, nme))
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)
import collections
by = collections.defaultdict(lambda: [0, 0])
for g, nme in gaps:
    k = nme.replace("(anonymous namespace)::", "")[:60]; by[k][0] += 1; by[k][1] += g
print("idle before kernel class (top):")
for k, (cnt, g) in sorted(by.items(), key=lambda x: -x[1][1])[:12]:
    print(f"  {g/1e6:7.2f} ms in {cnt:5d} gaps (avg {g/cnt/1e3:6.1f} us) before {k}")

"""One replayed training step out of a rocprofv3 kernel trace: python scripts/train_step_trace.py <results.db> [listing.txt]
Prints per-kernel-class launches / time of the LAST complete step (stem_kernel to stem_kernel), the device's union-busy time,
and how the wall time of the step splits into: some kernel running / nothing running; writes the ordered launch list."""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute("select name, start, end%s from kernels order by start" % (", " + qcol if qcol else "")).fetchall()
def short(n):
    n = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))
    return re.sub(r"\(.*", "", n)[:90]
rows = [(short(r[0]), r[1], r[2], r[3] if qcol else 0) for r in rows]
stems = [i for i, r in enumerate(rows) if r[0].startswith("stem_kernel")]
a, b = stems[-2], stems[-1]
step = rows[a:b]
t0, t1 = step[0][1], rows[b][1]
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e, q in step:
    agg[n][0] += 1; agg[n][1] += e - s
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} launches, sum of kernel times {sum(v[1] for v in agg.values()) / 1e6:.2f} ms")
busy, cur_s, cur_e = 0, step[0][1], step[0][2]
gaps = collections.defaultdict(lambda: [0, 0])
for n, s, e, q in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps[n][0] += 1; gaps[n][1] += s - cur_e
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"some kernel running {busy / 1e6:.2f} ms, nothing running {(t1 - t0 - busy) / 1e6:.2f} ms in {sum(v[0] for v in gaps.values())} gaps")
print("kernel classes:")
for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:45]:
    print(f"  {t / 1e6:7.3f} ms {k:5d} x {t / k / 1e3:7.1f} us  {n}")
print("idle gaps by the kernel that ends them:")
for n, (k, t) in sorted(gaps.items(), key=lambda x: -x[1][1])[:25]:
    print(f"  {t / 1e6:7.3f} ms {k:5d} x {t / k / 1e3:6.1f} us  {n}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        for n, s, e, q in step:
            f.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} q{q} {n}\n")

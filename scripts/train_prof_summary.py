"""per-kernel-class time of a training-step trace: python scripts/train_prof_summary.py <rocprofv3 results .db> <steps traced>"""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = float(sys.argv[2])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for name, s, e in rows:
    name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
    name = re.sub(r"\(.*", "", name)[:100]
    agg[name][0] += 1
    agg[name][1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"kernel time per step {tot / n / 1e6:.2f} ms, launches per step {sum(v[0] for v in agg.values()) / n:.0f}")
for name, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{t / n / 1e6:8.3f} ms {k / n:7.1f} x {t / k / 1e3:8.1f} us  {name}")

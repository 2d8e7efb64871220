"""Timeline analysis of a rocprofv3 --kernel-trace csv: for every encoder pass (stem_kernel ... next stem_kernel)
prints span, union of busy time, sum of kernel durations, and the idle gaps (time with no kernel running)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
              re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]) for r in rows))
starts = [i for i, e in enumerate(ev) if "stem_kernel" in e[2]]
for a, b in list(zip(starts, starts[1:] + [len(ev)]))[-3:]:
    seg = ev[a:b]
    seg = [e for e in seg if "query" not in e[2]] if len(sys.argv) > 2 else seg
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy, cur_end, gaps = 0, t0, []
    for s, e, n in seg:
        if s > cur_end:
            gaps.append((s - cur_end, n))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    tot = sum(e - s for s, e, _ in seg)
    print("pass: kernels %d span %.1f us busy(union) %.1f us sum %.1f us idle %.1f us in %d gaps (mean %.2f us)" % (
        len(seg), (t1 - t0) / 1e3, busy / 1e3, tot / 1e3, sum(g for g, _ in gaps) / 1e3, len(gaps),
        sum(g for g, _ in gaps) / 1e3 / max(1, len(gaps))))
    big = sorted(gaps, reverse=True)[:8]
    print("   largest gaps:", ", ".join("%.1fus before %s" % (g / 1e3, n[:28]) for g, n in big))

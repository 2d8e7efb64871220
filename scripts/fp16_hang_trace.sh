#!/bin/bash
repo=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/fh
timeout 120 rocprofv3 --kernel-trace -d /tmp/fh -o q --output-format csv -- python $repo/scripts/fp16_hang_probe.py fp16 b2b 2>&1 | grep "fp16 "
f=$(find /tmp/fh -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
slow = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 5_000_000]
print("kernels", len(rows), "slower than 5 ms:", len(slow))
for r in slow[:12]:
    print("  %9.3f ms  dur %9.3f ms  %s grid %s queue %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:60], r.get("Grid_Size"), r.get("Queue_Id")))
if slow:
    s, e = int(slow[0]["Start_Timestamp"]), int(slow[0]["End_Timestamp"])
    print("running concurrently with the first slow one:")
    for r in rows:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if r is not slow[0] and a < e and b > s:
            print("     %9.3f .. %9.3f ms  %s grid %s queue %s" % ((a - t0) / 1e6, (b - t0) / 1e6, r["Kernel_Name"][:60], r.get("Grid_Size"), r.get("Queue_Id")))
P

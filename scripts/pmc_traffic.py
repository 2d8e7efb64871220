"""HBM-side traffic per launch of every kernel from the FETCH_SIZE and WRITE_SIZE rocprofv3 passes
(scripts/profile_round.sh: *_pmc_pass1.txt, *_pmc_pass2.txt).  Units and corrections as MI355X_MICROARCH.md
prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes read -- calibrated here on
kernels with a known byte count in the same run (gn_apply_relu reads 32 768 KiB, reports 16 431; the 2x2 pool reads
65 536 KiB, reports 32 825; copyBuffer of 1 152 KiB reports 588) -- so it is doubled; WRITE_SIZE matched the known
byte counts exactly (32 768 KiB).     usage: pmc_traffic.py <fetch_pass.txt> <write_pass.txt> <out.json>"""
import json, re, sys, collections

def parse(path, counter):
    tot = collections.defaultdict(lambda: [0.0, 0])
    name = None
    for line in open(path):
        if not line.startswith("    "):
            m = re.match(r"(?:void )?(.*?) grid (\d+) n (\d+)", line.strip())
            name, n = (m.group(1), int(m.group(3))) if m else (None, 0)
        elif name and line.split()[0] == counter:
            tot[name][0] += float(line.split()[1]) * n
            tot[name][1] += n
    return tot

f, w = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    n = max(f[k][1], w[k][1], 1)
    rd, wr = 2.0 * f[k][0] / n * 1024, w[k][0] / n * 1024
    out[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if "conv_" in k or "map_stats" in k or "stem" in k or "query" in k:
        print("%-70s n=%4d read %8.2f MB write %8.2f MB" % (k[:70], v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))

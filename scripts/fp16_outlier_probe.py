"""rare slow launches?  N encodes of one mode, each timed with events: max / p99.9 / median, and the count beyond 3x the median
python scripts/fp16_outlier_probe.py <mode> <n> [sync_every]"""
import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dt = sys.argv[1]; n = int(sys.argv[2]); every = int(sys.argv[3]) if len(sys.argv) > 3 else 0
img = torch.from_numpy(synth.synth_images(4, 512, 512, 0)).cuda()
net = CHORE(chore_opt(dt)).cuda().eval(); synth.load_synth_weights(net, 0)
with torch.no_grad():
    for _ in range(5): net.filter(img)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(ev):
        a.record(); net.filter(img); b.record()
        if every and (i + 1) % every == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
t = np.array([a.elapsed_time(b) for a, b in ev])
print("%-7s n=%d sync_every=%d wall %.2f s  median %.3f ms  p99 %.3f  max %.3f  beyond 3x median: %d  first slow index %s" % (
    dt, n, every, wall, np.median(t), np.percentile(t, 99), t.max(), int((t > 3 * np.median(t)).sum()),
    (np.argmax(t > 3 * np.median(t)) if (t > 3 * np.median(t)).any() else None)), flush=True)

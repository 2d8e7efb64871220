"""phase times of the split query-forward kernel (scripts/build_variant.sh stamps query_fwd.hip -DCHORE_QUERY_STAMPS=1; CHORE_HIP_LIB=<that library>)"""
import ctypes, os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
from chore_amd import _lib
B, N = int(sys.argv[1]), int(sys.argv[2])
net = CHORE(chore_opt("fp16x3")).cuda().eval(); synth.load_synth_weights(net, 0)
with torch.no_grad():
    net.filter(torch.from_numpy(synth.synth_images(B, 512, 512, 0)).cuda())
    cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
    pts = torch.from_numpy(synth.synth_points(B, N, seed=1)).cuda()
    for _ in range(5): net.query(pts, crop_center=cc)
    torch.cuda.synchronize()
n = 4096 * 8
buf = (ctypes.c_ulonglong * n)()
L = ctypes.CDLL(os.environ["CHORE_HIP_LIB"])
assert L.chore_debug_query_stamps(buf, n) == 0
a = np.array(buf[:]).reshape(4096, 8).astype(np.int64)
nwg = min(4096, B * ((N + 63) // 64) if B * ((N + 63) // 64) > 256 else B * ((N + 31) // 32))
a = a[:nwg]
d = np.diff(a[:, :6], axis=1) / 100.0     # wall_clock64: 100 MHz -> us
print("workgroups", nwg, " phases (us, mean / median / p90): table, gather, layer1, publish+layers 2-3, output+store")
for i, nm in enumerate(("table", "gather", "layer1", "layers23", "out")):
    print("  %-9s %6.2f %6.2f %6.2f" % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("  total     %6.2f" % ((a[:, 5] - a[:, 0]).mean() / 100.0), " kernel span %.1f us" % ((a[:, 5].max() - a[:, 0].min()) / 100.0))

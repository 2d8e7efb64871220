import sys, copy, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import *  # noqa
import bench
from chore_amd.model import CHORE
from chore_amd.utils import synth
o = bench.chore_opt("bf16")
net = CHORE(o).cuda(); synth.load_synth_weights(net, seed=0); net.train(True); net.losses_on_host = False
B, N = 2, 2000
rs = np.random.RandomState(50)
t = lambda a: torch.from_numpy(a).cuda()
batch = dict(images=t(synth.synth_images(B, 512, 512, seed=0)), points=t(synth.synth_points(B, N, seed=1)),
             df_h=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)), df_o=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)),
             parts_gt=t(rs.randint(0, 14, (B, N))), pca_gt=t(rs.standard_normal((B, 3, 3, N)).astype(np.float32)),
             body_center=t((rs.standard_normal((B, 3)) * 0.3).astype(np.float32)),
             obj_center=t((rs.standard_normal((B, 3, N)) * 0.3).astype(np.float32)),
             crop_center=torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32).cuda())
error, sep = net(**batch); error.backward()
print("loss", float(error))
for n, p in net.named_parameters():
    if p.grad is None: continue
    ok = torch.isfinite(p.grad).all() and p.grad.abs().max() > 0
    if not ok: print("BAD", n, tuple(p.shape), "finite", bool(torch.isfinite(p.grad).all()), "max", float(p.grad.abs().max()))

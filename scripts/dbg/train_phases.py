"""CPU issue time vs GPU time of the phases of a training step (no profiler)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dev = torch.device("cuda", 0)
opt = chore_opt("bf16"); opt.gpu_id = 0
net = CHORE(opt).to(dev); synth.load_synth_weights(net, seed=0); net.train(True)
optim = torch.optim.Adam(net.parameters(), lr=1e-4)
B, N = 4, 20000
rs = np.random.RandomState(50)
batch = dict(images=torch.from_numpy(synth.synth_images(B, 512, 512, seed=0)).to(dev),
             points=torch.from_numpy(synth.synth_points(B, N, seed=1)).to(dev),
             df_h=torch.from_numpy(rs.uniform(0, 0.3, (B, N)).astype(np.float32)).to(dev),
             df_o=torch.from_numpy(rs.uniform(0, 0.3, (B, N)).astype(np.float32)).to(dev),
             parts_gt=torch.from_numpy(rs.randint(0, 14, (B, N))).to(dev),
             pca_gt=torch.from_numpy(rs.standard_normal((B, 3, 3, N)).astype(np.float32)).to(dev),
             body_center=torch.from_numpy((rs.standard_normal((B, 3)) * 0.3).astype(np.float32)).to(dev),
             obj_center=torch.from_numpy((rs.standard_normal((B, 3, N)) * 0.3).astype(np.float32)).to(dev),
             crop_center=torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev))
def phase(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return r, (t1 - t) * 1e3, (t2 - t) * 1e3
for it in range(6):
    optim.zero_grad(set_to_none=True)
    (err, _), f_cpu, f_all = phase(lambda: net(**batch))
    _, b_cpu, b_all = phase(lambda: err.backward())
    _, o_cpu, o_all = phase(lambda: optim.step())
    if it >= 3:
        print(f"forward: issue {f_cpu:.1f} ms, done {f_all:.1f} | backward: issue {b_cpu:.1f}, done {b_all:.1f} | adam: issue {o_cpu:.1f}, done {o_all:.1f}")

"""training step: host issue time (forward / backward / optimiser, no synchronisation inside) against the device-bound wall time"""
import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dev = torch.device("cuda:0")
net = CHORE(chore_opt("bf16")).to(dev); synth.load_synth_weights(net, seed=0); net.train(True); net.losses_on_host = False
optim = torch.optim.Adam(net.parameters(), lr=1e-4)
B, N = 4, 20000
rs = np.random.RandomState(50); t = lambda a: torch.from_numpy(a).to(dev)
batch = dict(images=t(synth.synth_images(B, 512, 512, seed=0)), points=t(synth.synth_points(B, N, seed=1)),
             df_h=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)), df_o=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)),
             parts_gt=t(rs.randint(0, 14, (B, N))), pca_gt=t(rs.standard_normal((B, 3, 3, N)).astype(np.float32)),
             body_center=t((rs.standard_normal((B, 3)) * 0.3).astype(np.float32)),
             obj_center=t((rs.standard_normal((B, 3, N)) * 0.3).astype(np.float32)),
             crop_center=torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev))
def step(T):
    t0 = time.perf_counter(); optim.zero_grad(set_to_none=True); err, _ = net(**batch)
    t1 = time.perf_counter(); err.backward()
    t2 = time.perf_counter(); optim.step()
    t3 = time.perf_counter(); T += np.array([t1 - t0, t2 - t1, t3 - t2])
for _ in range(3): step(np.zeros(3))
torch.cuda.synchronize()
# (a) host time with an idle device: synchronise before every step so nothing back-pressures the host
T = np.zeros(3)
for _ in range(5):
    torch.cuda.synchronize(); step(T)
print("host issue time per step, device idle at the start: fwd %.1f ms  bwd %.1f ms  optim %.1f ms  (sum %.1f)" % (*(T / 5 * 1e3), T.sum() / 5 * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step(np.zeros(3))
torch.cuda.synchronize(); print("wall per step %.1f ms" % ((time.perf_counter() - t0) / 8 * 1e3))

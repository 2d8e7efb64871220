import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dev = torch.device("cuda:0")
net = CHORE(chore_opt("bf16")).to(dev); synth.load_synth_weights(net, seed=0); net.train(True); net.losses_on_host = False
optim = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
B, N = 4, 20000
rs = np.random.RandomState(50); t = lambda a: torch.from_numpy(a).to(dev)
batch = dict(images=t(synth.synth_images(B, 512, 512, seed=0)), points=t(synth.synth_points(B, N, seed=1)),
             df_h=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)), df_o=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)),
             parts_gt=t(rs.randint(0, 14, (B, N))), pca_gt=t(rs.standard_normal((B, 3, 3, N)).astype(np.float32)),
             body_center=t((rs.standard_normal((B, 3)) * 0.3).astype(np.float32)),
             obj_center=t((rs.standard_normal((B, 3, N)) * 0.3).astype(np.float32)),
             crop_center=torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev))
def step():
    optim.zero_grad(set_to_none=True); err, _ = net(**batch); err.backward(); optim.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
import collections
c = collections.Counter()
evs = prof.events()
for e in evs:
    if e.device_type is not None and "cpu" not in str(e.device_type).lower():
        continue
    ks = [k.name for k in (e.kernels or [])]
    if any("Memcpy" in k or "copyBuffer" in k or "memcpy" in k.lower() for k in ks):
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        gp = e.cpu_parent.cpu_parent.name if (e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None) else "-"
        c[(e.name, par, gp, str(e.input_shapes)[:60])] += 1
for k, v in c.most_common(25):
    print(v, k)

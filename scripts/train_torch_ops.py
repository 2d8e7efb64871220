"""which torch operators (not ours) launch kernels inside the training step, and from where: torch.profiler over 3 steps"""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
from collections import Counter
for name in ("aten::zeros", "aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::mul", "aten::sum", "aten::cat", "aten::zeros_like", "aten::ones_like"):
    c = Counter()
    for e in prof.events():
        if e.name == name:
            st = [f for f in (e.stack or []) if "site-packages/torch" not in f and "dist-packages/torch" not in f][:3]
            c[(tuple(st), tuple(map(str, e.input_shapes))[:2])] += 1
    print("=====", name, sum(c.values()) / 3.0, "per step")
    for k, v in c.most_common(8):
        print("   %6.1f/step  %s  %s" % (v / 3.0, k[1], " <- ".join(x[-70:] for x in k[0])))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))

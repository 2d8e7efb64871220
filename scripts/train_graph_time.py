"""training step (B=4 x 512x512, 20k points/image, bf16 maps): eager issue against ONE hipGraph replay of
zero_grad + forward + backward + Adam (torch.cuda.graph, capturable fused Adam), same kernels either way"""
import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dev = torch.device("cuda:0")
net = CHORE(chore_opt("bf16")).to(dev); synth.load_synth_weights(net, seed=0); net.train(True); net.losses_on_host = False
optim = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True, capturable=True)
B, N = 4, 20000
rs = np.random.RandomState(50); t = lambda a: torch.from_numpy(a).to(dev)
batch = dict(images=t(synth.synth_images(B, 512, 512, seed=0)), points=t(synth.synth_points(B, N, seed=1)),
             df_h=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)), df_o=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)),
             parts_gt=t(rs.randint(0, 14, (B, N))), pca_gt=t(rs.standard_normal((B, 3, 3, N)).astype(np.float32)),
             body_center=t((rs.standard_normal((B, 3)) * 0.3).astype(np.float32)),
             obj_center=t((rs.standard_normal((B, 3, N)) * 0.3).astype(np.float32)),
             crop_center=torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev))
last = {}
def step():
    optim.zero_grad(set_to_none=False)
    err, _ = net(**batch)
    err.backward()
    optim.step()
    last["e"] = err
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4): step()
torch.cuda.current_stream().wait_stream(s)
print("eager %.2f ms / step   loss %.6f" % (timeit(step), float(last["e"])), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("graph replay %.2f ms / step   loss %.6f" % (timeit(g.replay), float(last["e"])), flush=True)

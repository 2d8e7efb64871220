"""N training passes in one process (gradients compared bit for bit with the first) while another process keeps the GPU busy"""
import os, sys, subprocess, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "noise":
    a = torch.randn(4096, 4096, device="cuda")
    while True:
        for _ in range(50): b = a @ a
        torch.cuda.synchronize()
from test_gpu_ddp_trainstep import _make
def grads():
    net, batch = _make(0); net.train(True)
    err, _ = net(**batch); err.backward()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
g0 = grads()
noise = [subprocess.Popen([sys.executable, __file__, "noise"]) for _ in range(int(os.environ.get("NOISE", "2")))]
import time; time.sleep(8)
try:
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        g = grads()
        bad = [n for n in g0 if not torch.equal(g0[n], g[n])]
        print("pass", i, "tensors differing from the quiet pass:", len(bad), bad[:3], flush=True)
finally:
    for p in noise: p.kill()

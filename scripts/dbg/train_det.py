import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_ddp_trainstep import _make
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
def grads(poison):
    if poison:
        junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(8)]
        del junk
    net, batch = _make(0)
    if mode != "bf16":
        net.compute_dtype = mode
    net.train(True)
    err, _ = net(**batch)
    err.backward()
    return float(err), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
e0, g0 = grads(False)
e1, g1 = grads(True)
e2, g2 = grads(True)
print("loss", e0, e1, e2)
bad = [(n, float((g0[n] - g1[n]).abs().max() / (g0[n].abs().max() + 1e-30))) for n in g0 if not torch.equal(g0[n], g1[n])]
bad2 = [n for n in g1 if not torch.equal(g1[n], g2[n])]
print("run0 vs run1 differing tensors:", len(bad), bad[:8])
print("run1 vs run2 differing tensors:", len(bad2), bad2[:8])
print("nonfinite:", [n for n in g1 if not torch.isfinite(g1[n]).all()][:5])

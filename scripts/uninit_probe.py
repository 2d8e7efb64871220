"""Does the training step read memory it has not written?  torch.empty() is made to return NaN-filled memory (0xFF bytes:
NaN as fp32 and as bf16) -- torch.utils.deterministic.fill_uninitialized_memory under use_deterministic_algorithms -- and one
training pass runs: any value the kernels pick up from an uninitialised workspace, staging buffer or output turns a gradient
into NaN (or changes it).  Compared with a normal pass."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_ddp_trainstep import _make

net, batch = _make(0)
net.train(True)


def grads():
    for p in net.parameters():
        p.grad = None
    err, _ = net(**batch)
    err.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, float(err)


g0, e0 = grads()
g1, e1 = grads()
print("two normal passes identical:", all(torch.equal(g0[n], g1[n]) for n in g0), e0 == e1)
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
x = torch.empty(8, device="cuda"); print("poison check (expect nan):", x[:2].tolist(), torch.empty(4, dtype=torch.uint8, device="cuda").tolist())
g2, e2 = grads()
bad = [n for n in g0 if not torch.equal(g0[n], g2[n])]
nan = [n for n in g2 if not torch.isfinite(g2[n].float()).all()]
print("loss", e0, e2, "| tensors differing from the normal pass:", len(bad), "| with non-finite entries:", len(nan))
order = [n for n, _ in net.named_parameters()][::-1]
print("first differing in backward order:", [n for n in order if n in bad][:8])
print("first non-finite in backward order:", [n for n in order if n in nan][:8])

"""which ingredient of GraphedTrainStep's warm-up makes its eager steps differ from plain eager steps?"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, REPO + "/tests")
from test_gpu_ddp_trainstep import _make

batches = [_make(it)[1] for it in range(3)]

def run(side, invalidate, capturable=True, sync=False):
    net, _ = _make(0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=capturable, fused=True)
    s = torch.cuda.Stream() if side else None
    out = []
    for it in range(3):
        net.train()
        if s is not None:
            s.wait_stream(torch.cuda.current_stream())
            cm = torch.cuda.stream(s)
        else:
            import contextlib
            cm = contextlib.nullcontext()
        with cm:
            opt.zero_grad(set_to_none=True)
            loss, sep = net(**batches[it])
            loss.backward()
            if sync:
                torch.cuda.synchronize()
            opt.step()
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)
        if invalidate:
            net.invalidate_packed()
        out.append(float(loss))
    return out

print("default stream            ", run(False, False))
print("default stream, again     ", run(False, False))
print("default + invalidate      ", run(False, True))
print("side stream               ", run(True, False))
print("side stream + sync        ", run(True, False, sync=True))
print("side + invalidate         ", run(True, True))
print("default, not capturable   ", run(False, False, capturable=False))

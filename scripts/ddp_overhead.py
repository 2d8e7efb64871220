"""What does the gradient reduction cost on ONE GPU (one-rank RCCL group, nothing crosses xGMI)?  ms per training step of
BASELINE configs[3]'s per-GPU share for: no reduction; torch DDP as the reference wraps it (train_launch.py:30); DDP with
gradient_as_bucket_view / static_graph / bigger buckets; and chore_amd.parallel.grad_arena.FlatGradReducer (one flat fp32
gradient arena, all-reduced in a few large chunks after the backward).   usage: python scripts/ddp_overhead.py [steps]"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_ddp_trainstep import _make

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


def run(name, wrap, reducer=None):
    net, batch = _make(0)
    net.train(True); net.losses_on_host = False
    model = wrap(net) if wrap else net
    red = reducer(net) if reducer else None
    optim = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)

    def step():
        if red is not None:
            red.zero_grad()
        else:
            optim.zero_grad(set_to_none=True)
        err, _ = model(**batch)
        err.backward()
        if red is not None:
            red.reduce()
        optim.step()
    for _ in range(6):
        step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    print("%-60s %7.2f ms/step" % (name, ms), flush=True)
    del model, net, optim
    torch.cuda.empty_cache()
    return ms


DDP = torch.nn.parallel.DistributedDataParallel
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "ddp", "view", "static", "big", "arena"]
if "none" in which: run("no reduction", None)
if "ddp" in which: run("DDP(find_unused_parameters=True)  [reference's wrap]", lambda n: DDP(n, device_ids=[0], find_unused_parameters=True))
if "view" in which: run("DDP(find_unused, gradient_as_bucket_view=True)", lambda n: DDP(n, device_ids=[0], find_unused_parameters=True, gradient_as_bucket_view=True))
if "static" in which: run("DDP(static_graph=True, gradient_as_bucket_view=True)", lambda n: DDP(n, device_ids=[0], static_graph=True, gradient_as_bucket_view=True))
if "big" in which: run("DDP(find_unused, bucket view, bucket_cap_mb=100)", lambda n: DDP(n, device_ids=[0], find_unused_parameters=True, gradient_as_bucket_view=True, bucket_cap_mb=100))
if "arena" in which:
    from chore_amd.parallel.grad_arena import FlatGradReducer
    run("FlatGradReducer (flat arena, chunked all-reduce after backward)", None, lambda n: FlatGradReducer(n))
    run("FlatGradReducer(chunks=1)", None, lambda n: FlatGradReducer(n, chunks=1))
dist.destroy_process_group()

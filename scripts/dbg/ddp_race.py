"""repeat the two-rank DDP step and compare the reduced gradients with the mean of the ranks' own (computed once here)"""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch.multiprocessing as mp
from test_gpu_ddp_trainstep import _make


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    net, batch = _make(rank)
    model = net if os.environ.get("NO_DDP") else torch.nn.parallel.DistributedDataParallel(
        net, device_ids=[0], find_unused_parameters=True, broadcast_buffers=not os.environ.get("NO_BCAST"))
    model.train()
    if not os.environ.get("NO_ANOMALY"):
        torch.autograd.set_detect_anomaly(True)
    loss, _ = model(**batch)
    loss.backward()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, **{n: p.grad.detach().float().cpu().numpy() for n, p in net.named_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    per_rank = []
    for r in range(2):
        net, batch = _make(r); net.train(True)
        e, _ = net(**batch); e.backward()
        per_rank.append({n: p.grad.detach().float().cpu().numpy() for n, p in net.named_parameters() if p.grad is not None})
        del net, batch
    for i in range(reps):
        out = "/tmp/ddp_%d.npz" % i
        mp.spawn(_worker, args=(2, 29600 + i, out), nprocs=2, join=True)
        got = np.load(out)
        bad = []
        for n in per_rank[0]:
            ref = per_rank[0][n] if os.environ.get("NO_DDP") else (per_rank[0][n] + per_rank[1][n]) * 0.5
            err = np.abs(got[n] - ref).max() / max(np.abs(ref).max(), 1e-30)
            if err > 1e-6: bad.append((n, float(err)))
        print("rep", i, " ".join(k for k in ("NO_DDP", "NO_ANOMALY", "NO_BCAST", "CHORE_CONVBLOCK_SERIAL") if os.environ.get(k)), "tensors off:", len(bad), bad[:4], flush=True)

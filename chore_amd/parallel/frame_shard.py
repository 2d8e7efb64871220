"""Frame-sharded batch fitting across the GPUs of one node (BASELINE config 5).

Frames are independent (own SMPL / object parameters and optimiser, recon/recon_fit_behave.py:41-76), so
the reference leaves sharding to the user (`-fs/-fe`, recon_fit_behave.py:385-386).  Here rank r fits
frames r, r+W, r+2W, ... with one process per GPU; the only communication is one gather of the fitted
parameters (~200 floats per frame) at the end -- no collective inside the fit loop.  `backend="nccl"`
is RCCL on ROCm; the CPU tests run the same code over gloo.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """env:// rendezvous like utils/dist_utils.py:12-33 (RANK / WORLD_SIZE / LOCAL_RANK); returns (rank, world)"""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://")
    return dist.get_rank(), dist.get_world_size()


def frames_of_rank(num_frames, rank, world):
    """indices of the frames rank `rank` fits (round robin keeps per-rank counts within one of each other)"""
    return list(range(rank, num_frames, world))


def shard_indices(num_items):
    """the items (frames or loader batches) of THIS process: all of them without torch.distributed, rank::world with it"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return frames_of_rank(num_items, dist.get_rank(), dist.get_world_size())
    return list(range(num_items))


def gather_fitted(local, num_frames, rank, world, device=None):
    """local: {name: tensor (n_local, ...)} for frames_of_rank(...) in order.  Returns on every rank
    {name: tensor (num_frames, ...)} in frame order.  One all_gather per parameter on padded blocks."""
    if world == 1:
        return {k: v.clone() for k, v in local.items()}
    per = (num_frames + world - 1) // world
    out = {}
    for name in sorted(local):
        v = local[name]
        dev = device if device is not None else v.device
        pad = torch.zeros((per,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        pad[:v.shape[0]] = v.to(dev)
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        full = torch.zeros((num_frames,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        for r in range(world):
            idx = frames_of_rank(num_frames, r, world)
            if idx:
                full[idx] = parts[r][:len(idx)]
        out[name] = full
    return out

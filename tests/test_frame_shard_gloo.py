"""CPU, world_size 2 over gloo: frame-sharded fitting logic (sharding + final gather), the multi-GPU path
of config 5, with a stand-in per-frame 'fit' (the HIP kernels need a GPU; the collective logic does not)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, num_frames, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from chore_amd.parallel import frames_of_rank, gather_fitted, init_distributed
    r, w = init_distributed(backend="gloo")
    mine = frames_of_rank(num_frames, r, w)
    # stand-in fit: parameters are a deterministic function of the frame index
    local = {"trans": torch.tensor([[f, 2 * f, 3 * f] for f in mine], dtype=torch.float32).reshape(-1, 3),
             "obj_R": torch.stack([torch.eye(3) * (f + 1) for f in mine]) if mine else torch.zeros(0, 3, 3)}
    full = gather_fitted(local, num_frames, r, w)
    q.put((r, mine, {k: v.clone() for k, v in full.items()}))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_shard_and_gather():
    num_frames, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    want_t = torch.tensor([[f, 2 * f, 3 * f] for f in range(num_frames)], dtype=torch.float32)
    for _, _, full in res:   # every rank ends up with all frames, in frame order
        assert torch.equal(full["trans"], want_t)
        for f in range(num_frames):
            assert torch.equal(full["obj_R"][f], torch.eye(3) * (f + 1))


def test_single_process_passthrough():
    from chore_amd.parallel import frames_of_rank, gather_fitted
    assert frames_of_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    x = {"a": torch.arange(6.0).reshape(3, 2)}
    assert torch.equal(gather_fitted(x, 3, 0, 1)["a"], x["a"])

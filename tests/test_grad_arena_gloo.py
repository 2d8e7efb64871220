"""CPU, world_size 2 over gloo: chore_amd.parallel.FlatGradReducer leaves in every .grad what torch's
DistributedDataParallel(find_unused_parameters=True) -- the reference's wrap, train_launch.py:30 -- leaves there (the mean of
the ranks' gradients; zeros for parameters the loss does not reach), for both of its modes, over several optimiser steps with
an unused parameter in the model and gradient tensors of odd sizes."""
import os
import socket

import torch
import torch.multiprocessing as mp


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(7, 13)
        self.unused = torch.nn.Parameter(torch.ones(5))          # like the bn4 affines of blocks without downsample
        self.b = torch.nn.Conv1d(13, 3, 1)
        self.c = torch.nn.Parameter(torch.full((1, 3, 1), 0.5))

    def forward(self, x):
        h = torch.relu(self.a(x)).transpose(1, 2)
        return (self.b(h) * self.c).square().mean()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from chore_amd.parallel import FlatGradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    ref = Net()
    ddp = torch.nn.parallel.DistributedDataParallel(ref, find_unused_parameters=True)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-2)
    nets = {}
    for mode in ("copy", "inplace"):
        torch.manual_seed(0)
        n = Net()
        nets[mode] = (n, FlatGradReducer(n, chunks=3, mode=mode), torch.optim.Adam(n.parameters(), lr=1e-2))
    worst = 0.0
    for it in range(4):
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(6, 11, 7, generator=g)
        opt_ref.zero_grad(set_to_none=True)
        ddp(x).backward()
        for mode, (n, red, opt) in nets.items():
            red.zero_grad()
            n(x).backward()
            red.reduce()
            for (name, p), pr in zip(n.named_parameters(), ref.parameters()):
                want = pr.grad if pr.grad is not None else torch.zeros_like(pr)
                assert p.grad is not None and p.grad.data_ptr() >= red.arena.data_ptr(), (mode, name)
                worst = max(worst, float((p.grad - want).abs().max()))
            opt.step()
        opt_ref.step()
    for mode, (n, _, _) in nets.items():
        for p, pr in zip(n.parameters(), ref.parameters()):
            worst = max(worst, float((p - pr).abs().max()))
    q.put((rank, worst))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_reducer_equals_ddp_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(w < 1e-6 for _, w in res), res


class Chain(torch.nn.Module):
    """pre -> [cut 0] -> stage 0 -> [cut 1] -> stage 1, a loss term hanging off every stage (the shape of CHORE's stacks)"""

    def __init__(self):
        super().__init__()
        self.pre = torch.nn.Linear(7, 9)
        self.s0 = torch.nn.Linear(9, 9)
        self.unused = torch.nn.Parameter(torch.ones(3))
        self.s1 = torch.nn.Linear(9, 9)

    def forward(self, x):
        c0 = torch.tanh(self.pre(x))
        o0 = torch.tanh(self.s0(c0))
        c1 = c0 + o0              # like previous_{i+1} = previous_i + f(stack i): every consumer of a cut lies in ITS stage
        o1 = torch.tanh(self.s1(c1))
        return [c0, c1], [o0.square().mean(), o1.square().mean()]


def _worker_segments(rank, world, port, q):
    import torch.distributed as dist
    from chore_amd.parallel import FlatGradReducer, backward_in_segments
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    plain, seg = Chain(), Chain()
    seg.load_state_dict(plain.state_dict())
    red_plain = FlatGradReducer(plain, chunks=2)
    segments = [list(seg.s1.parameters()), list(seg.s0.parameters()), list(seg.pre.parameters()) + [seg.unused]]
    red_seg = FlatGradReducer(seg, segments=segments)
    assert len(red_seg.chunks) == 3 and all(c.numel() % (64 * world) == 0 for c in red_seg.chunks)
    opts = [torch.optim.Adam(m.parameters(), lr=1e-2) for m in (plain, seg)]
    worst, order = 0.0, []
    for it in range(4):
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(6, 7, generator=g)
        red_plain.zero_grad()
        _, losses = plain(x)
        (losses[0] + losses[1]).backward()
        red_plain.reduce()
        red_seg.zero_grad()
        cuts, losses = seg(x)

        def after(k):
            red_seg.gather(k)
            red_seg.all_reduce_async(k)      # segment k is on its way while the next segment's backward runs
            order.append(k)
        backward_in_segments(losses, cuts, segments, after)
        red_seg.finish()
        for (name, p), pr in zip(seg.named_parameters(), plain.parameters()):
            assert p.grad is not None and p.grad.data_ptr() >= red_seg.arena.data_ptr(), name
            assert torch.equal(p.grad, pr.grad), (it, name, float((p.grad - pr.grad).abs().max()))     # bit for bit
        for o in opts:
            o.step()
    assert order[:3] == [0, 1, 2]
    for p, pr in zip(seg.parameters(), plain.parameters()):
        worst = max(worst, float((p - pr).abs().max()))
    q.put((rank, worst))
    dist.barrier()
    dist.destroy_process_group()


def test_segmented_backward_with_collectives_under_it_equals_the_flat_reducer_two_ranks():
    """FlatGradReducer(segments=...) + backward_in_segments (round 5: each segment's slice of the arena is all-reduced while the next
    segment's backward runs) against the unsegmented reducer on a chain with a loss term per stage, an unused parameter and odd
    sizes: every gradient and every parameter after four Adam steps EQUAL, on both ranks"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_segments, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(w == 0.0 for _, w in res), res

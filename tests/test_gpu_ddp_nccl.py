"""GPU: the DDP training step over RCCL itself (`backend="nccl"` IS RCCL on ROCm; train_launch.py:30,
utils/dist_utils.py:28-33 of the reference run NCCL).  The box has one GPU and NCCL allows one rank per device, so the process
group has world_size 1 -- the collective still runs: DDP's reducer copies every gradient into its buckets, launches
`ncclAllReduce` per bucket on RCCL's own stream behind an event recorded on the autograd stream when the bucket's last hook
fires, and copies the result back.  That is exactly the hand-over the two-stream ConvBlock backward has to get right (its
weight gradients are produced on the handle's SIDE stream, csrc/convblock.hip; the join back to the caller's stream is the last
thing chore_convblock_bwd does), and gloo -- host-synchronous -- cannot expose an ordering bug there.

A child process (the process group and RCCL's communicator die with it) runs the reference's `Trainer.train_step` sequence
(trainer/trainer.py:76-85) three times through `DistributedDataParallel(find_unused_parameters=True)` at the full per-GPU size
of BASELINE configs[3] (4 x 512^2 images, 4 x 20 000 points, 5 stacks, bf16 maps) and, from the same initial weights on the
same batches, three times without DDP, and three times with chore_amd.parallel.FlatGradReducer (the flat gradient arena that
replaces the wrap, all-reduced over RCCL after the backward).  With one rank the mean over ranks is the identity, so after
every step every `.grad` and after the three steps every parameter must be EQUAL BIT FOR BIT in all three runs."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 3


def _steps(model, net, batches, record, reducer=None):
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4)
    for it in range(STEPS):
        # ---- Trainer.train_step, line by line (with a FlatGradReducer: its zero_grad() / reduce() around the backward) ----
        model.train()
        torch.autograd.set_detect_anomaly(True)
        if reducer is not None:
            reducer.zero_grad()
        else:
            optimizer.zero_grad()
        loss, _ = model(**batches[it])
        loss.backward()
        if reducer is not None:
            reducer.reduce()
        optimizer.step()
        value = loss.item()
        torch.autograd.set_detect_anomaly(False)
        assert np.isfinite(value)
        record.append((value, {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    return {n: p.detach().clone() for n, p in net.named_parameters()}


def _worker(rank, port, out_path, dtype="bf16"):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from test_gpu_ddp_trainstep import _make as _make_dt
    _make = lambda r: _make_dt(r, dtype)      # noqa: E731
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    batches = []
    for it in range(STEPS):
        _, b = _make(it)          # three different batches
        batches.append(b)
    # a real collective through the communicator before anything else (also what DDP's parameter broadcast does)
    probe = torch.arange(1024, device="cuda", dtype=torch.float32)
    dist.all_reduce(probe)
    assert torch.equal(probe.cpu(), torch.arange(1024, dtype=torch.float32))

    net_ddp, _ = _make(0)
    model = torch.nn.parallel.DistributedDataParallel(net_ddp, device_ids=[0], find_unused_parameters=True)
    rec_ddp = []
    par_ddp = _steps(model, net_ddp, batches, rec_ddp)
    torch.cuda.synchronize()

    net_ref, _ = _make(0)
    rec_ref = []
    par_ref = _steps(net_ref, net_ref, batches, rec_ref)
    torch.cuda.synchronize()

    # chore_amd.parallel.FlatGradReducer (flat gradient arena, chunked all-reduce over RCCL after the backward): same contract
    from chore_amd.parallel import FlatGradReducer
    net_ar, _ = _make(0)
    red = FlatGradReducer(net_ar, chunks=4)
    red.sync_parameters()
    rec_ar = []
    par_ar = _steps(net_ar, net_ar, batches, rec_ar, reducer=red)
    torch.cuda.synchronize()
    ar_diff = 0
    for it in range(STEPS):
        assert rec_ar[it][0] == rec_ref[it][0], (it, rec_ar[it][0], rec_ref[it][0])
        for n, g in rec_ar[it][1].items():
            if n in rec_ref[it][1]:
                ar_diff += int(not torch.equal(g, rec_ref[it][1][n]))
            else:
                assert float(g.abs().max()) == 0.0, (it, n)
    ar_par_diff = sum(int(not torch.equal(par_ar[n], par_ref[n])) for n in par_ref)

    worst, differing, checked = 0.0, [], 0
    for it in range(STEPS):
        (la, ga), (lb, gb) = rec_ddp[it], rec_ref[it]
        assert la == lb, (it, la, lb)
        # DDP materialises zero gradients for the parameters the graph does not reach; the plain run leaves them None
        for n, g in ga.items():
            if n not in gb:
                assert float(g.abs().max()) == 0.0, (it, n)
                continue
            checked += 1
            if not torch.equal(g, gb[n]):
                d = float((g.float() - gb[n].float()).abs().max() / gb[n].float().abs().max().clamp_min(1e-30))
                worst = max(worst, d)
                differing.append((it, n, d))
    par_diff = [n for n in par_ref if not torch.equal(par_ddp[n], par_ref[n])]
    np.savez(out_path, worst=np.float64(worst), n_diff=np.int64(len(differing)), n_checked=np.int64(checked),
             n_par_diff=np.int64(len(par_diff)), first=np.array([str(differing[:5])]), trained=np.int64(len(rec_ref[0][1])),
             ar_diff=np.int64(ar_diff), ar_par_diff=np.int64(ar_par_diff))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["bf16", "fp16x3"])
def test_ddp_over_rccl_equals_the_plain_step_bit_for_bit(tmp_path, dtype):
    import torch.multiprocessing as mp
    out = str(tmp_path / "nccl_step.npz")
    mp.spawn(_worker, args=(29611 + (dtype == "fp16x3"), out, dtype), nprocs=1, join=True)
    r = np.load(out)
    print("RCCL DDP vs plain: tensors checked", int(r["n_checked"]), "differing", int(r["n_diff"]), "worst", float(r["worst"]),
          "parameters differing after", STEPS, "steps:", int(r["n_par_diff"]), str(r["first"][0]))
    assert int(r["trained"]) >= 475
    assert int(r["n_checked"]) >= 475 * STEPS
    assert int(r["n_diff"]) == 0, str(r["first"][0])
    assert int(r["n_par_diff"]) == 0
    assert int(r["ar_diff"]) == 0 and int(r["ar_par_diff"]) == 0, (int(r["ar_diff"]), int(r["ar_par_diff"]))


def test_recorded_step_after_eager_collectives_survives_the_watchdog():
    """A process group's watchdog thread polls the end events of the collectives it still lists; a hipGraph recording opened with
    works of the eager warm-up steps listed killed the process about once in eight starts (`operation not permitted on an event
    last recorded in a capturing stream`).  GraphedTrainStep drains the list first
    (chore_amd.parallel.drain_collectives).  The reproducer -- a one-rank RCCL group, two eager steps, the recording, three
    replays at the full configs[3] size -- is started four times (a plain "nccl" group and the "cpu:gloo,cuda:nccl" group of
    bench.py, flat and segmented reducer); every start must finish."""
    import subprocess
    probe = os.path.join(REPO, "scripts", "probes", "graph_record_watchdog.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for start in range(4):
        env["MASTER_PORT"] = str(29700 + (os.getpid() + start) % 200)
        r = subprocess.run([sys.executable, probe, "flat" if start % 2 == 0 else "segmented", "nccl" if start < 2 else "mixed"], cwd=REPO, env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0 and "\nok " in "\n" + r.stdout, (start, r.returncode, r.stderr[-600:])     # (gloo and RCCL print banners to stdout)

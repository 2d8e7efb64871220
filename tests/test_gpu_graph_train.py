"""GPU: chore_amd.parallel.GraphedTrainStep -- the training step (the reference's `Trainer.train_step` sequence,
trainer/trainer.py:76-85, at the per-GPU size of BASELINE configs[3]: 4 x 512^2 images, 4 x 20 000 points, 5 stacks, bf16 maps)
recorded as hipGraphs and replayed -- against the same steps issued eagerly, from the same initial weights on the same three
different batches, same optimiser (Adam, capturable).  Every kernel of the step has a fixed reduction order, so the bar is
equality BIT FOR BIT: every gradient after every step, every parameter and both Adam moments after the last one.  Run in a child
process (a recording holds ~10 GB of activations; the process takes them with it), with and without a FlatGradReducer."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from test_gpu_ddp_trainstep import _make as _make_dt
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "bf16"
_make = lambda rank: _make_dt(rank, DTYPE)
from chore_amd.parallel import FlatGradReducer, GraphedTrainStep, chore_segments
use_reducer = sys.argv[2] in ("arena", "segmented", "segmented_rs_ag")
segmented = sys.argv[2].startswith("segmented")
if use_reducer:
    import os, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("nccl", rank=0, world_size=1)
STEPS, WARM = 6, 2
batches = [_make(it)[1] for it in range(STEPS)]

def run(graphed):
    net, _ = _make(0)
    # the rate as an fp32 device tensor on BOTH sides (GraphedTrainStep moves a float rate into one: the recorded Adam launch
    # reads it at run time); it changes under replay twice: MultiStepLR's decay after step 3 (trainer/trainer.py:56-60) and a
    # plain `g['lr'] = value` before step 5 (the reference's resume path, trainer.py:246-257)
    opt = torch.optim.Adam(net.parameters(), lr=(1e-4 if graphed else torch.tensor(1e-4, device="cuda")), capturable=True, fused=True)
    red = FlatGradReducer(net) if use_reducer else None
    if graphed and segmented:
        # round 5: the recording cut at the hourglass stacks, every stack's slice of the gradient arena all-reduced (RCCL, its own
        # stream) while the graph of the next stack's backward runs -- against the EAGER, UNSEGMENTED step with the flat reducer
        red = FlatGradReducer(net, segments=chore_segments(net), collective="rs_ag" if sys.argv[2].endswith("rs_ag") else "all_reduce")
        assert len(red.segments) == 6 and len(red.chunks) == 6
    rec = []
    if graphed:
        step = GraphedTrainStep(net, opt, reducer=red, warmup=WARM)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3], gamma=0.3)
    for it in range(STEPS):
        if it == 4:
            for g in opt.param_groups:
                if graphed:
                    g["lr"] = 2e-5
                else:
                    g["lr"].fill_(2e-5)
        if graphed:
            loss, sep = step(**batches[it])
        else:
            net.train()
            (red.zero_grad() if red is not None else opt.zero_grad(set_to_none=True))
            loss, sep = net(**batches[it])
            loss.backward()
            if red is not None:
                red.reduce()
            opt.step()
        sched.step()
        rec.append((float(loss), sep.detach().cpu().numpy().copy(),
                    {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    assert abs(float(opt.param_groups[0]["lr"]) - 2e-5) < 1e-11
    torch.cuda.synchronize()
    state = {n: p.detach().clone() for n, p in net.named_parameters()}
    moments = [opt.state[p]["exp_avg_sq"].clone() for p in net.parameters() if p in opt.state]
    return rec, state, moments, (step if graphed else None)

ra, sa, ma, _ = run(False)
rb, sb, mb, step = run(True)
assert step.calls == STEPS and len(step._rec) == 1, (step.calls, len(step._rec))
if segmented:
    assert len(next(iter(step._rec.values()))["segs"]) == 6
bad = 0
for it in range(STEPS):
    assert ra[it][0] == rb[it][0], ("loss", it, ra[it][0], rb[it][0])
    assert np.array_equal(ra[it][1], rb[it][1]), ("separate losses", it)
    assert np.isfinite(ra[it][0])
    assert len(rb[it][2]) >= 475
    for n, g in ra[it][2].items():
        bad += int(not torch.equal(g, rb[it][2][n]))
assert bad == 0, ("gradient tensors differing", bad)
assert all(torch.equal(sa[n], sb[n]) for n in sa), "parameters differ"
assert len(ma) == len(mb) and all(torch.equal(a, b) for a, b in zip(ma, mb)), "Adam moments differ"
moved = sum(int(not torch.equal(sa[n], _make(0)[0].state_dict()[n])) for n in list(sa)[:20])
assert moved > 0
# the decayed rates REACHED the replayed launches: the same six steps at a constant rate end somewhere else
if not use_reducer:
    net, _ = _make(0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True, fused=True)
    step2 = GraphedTrainStep(net, opt, warmup=WARM)
    for it in range(STEPS):
        step2(**batches[it])
    torch.cuda.synchronize()
    sc = {n: p.detach().clone() for n, p in net.named_parameters()}
    assert sum(int(not torch.equal(sc[n], sb[n])) for n in sb) > 400
    # load_state_dict() swaps the optimiser's state tensors: the recording is dropped and made again (not replayed on stale state)
    import copy
    opt.load_state_dict(copy.deepcopy(opt.state_dict()))      # (a checkpoint read back: new tensors)
    old_graph = next(iter(step2._rec.values()))["ga"]          # (a reference, not an id: a freed object's id can be handed out again)
    step2(**batches[0])
    assert next(iter(step2._rec.values()))["ga"] is not old_graph
print("graphed == eager over", STEPS, "steps (", STEPS - WARM, "replayed ),", len(ra[0][2]), "gradient tensors; losses", [round(r[0], 6) for r in rb])
print("graph train ok", DTYPE)
'''


@pytest.mark.parametrize("dtype", ["bf16", "fp16x3"])
@pytest.mark.parametrize("reducer", ["none", "arena", "segmented", "segmented_rs_ag"])
def test_replayed_training_steps_equal_eager_steps_bit_for_bit(tmp_path, reducer, dtype):
    """fp16x3 (the mode bench.py's training record replays) carries range-tracked operand scales (enc_common.h x3_in_scale,
    dy_amax): device state a replay or a segmented backward must read and write exactly like the eager step"""
    script = tmp_path / "graph_train.py"
    script.write_text(CHILD)
    out = subprocess.run([sys.executable, str(script), REPO, reducer, dtype], capture_output=True, text=True, timeout=1500)
    print(out.stdout[-800:])
    assert out.returncode == 0 and "graph train ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_fused_optimizer_steps_reach_the_packed_heads():
    """torch's fused Adam kernel updates parameters WITHOUT advancing their version counters (checked on this torch: `_version`
    stays 0 after `Adam(fused=True).step()`), which the packed-heads cache is keyed on.  Round 4 found the training forward of
    step k+1 running on the heads of step 0 that way (second-step loss off by 1.5 %).  A query with trainable heads now always
    packs afresh: the losses of three steps with the fused optimiser must follow the default (foreach) optimiser's -- the two
    differ in the last bits of the update only."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_ddp_trainstep import _make
    batches = [_make(it)[1] for it in range(3)]
    losses = {}
    for kind, kw in (("foreach", {}), ("fused", {"fused": True})):
        net, _ = _make(0)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, **kw)
        out = []
        for it in range(3):
            net.train()
            opt.zero_grad(set_to_none=True)
            loss, _ = net(**batches[it])
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        losses[kind] = out
        del net, opt
        torch.cuda.empty_cache()
    print(losses)
    assert losses["foreach"][0] == losses["fused"][0]
    for a, b in zip(losses["foreach"][1:], losses["fused"][1:]):
        assert abs(a - b) <= 2e-4 * abs(a), losses


def test_queries_beside_the_encoder_equal_the_sequential_forward(monkeypatch):
    """CHORE.forward in training launches every stack's field query + loss on a second stream as soon as that stack's feature map
    exists (model/chore.py `_forward_interleaved`); CHORE_TRAIN_NO_INTERLEAVE=1 is the reference's order filter -> query ->
    get_errors (model/chore.py:175-190).  Same nodes either way: loss, separate losses and all gradients equal bit for bit,
    over two steps (the second one sees the first one's update)."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_ddp_trainstep import _make
    batches = [_make(it)[1] for it in range(2)]
    res = {}
    for kind in ("sequential", "interleaved"):
        if kind == "sequential":
            monkeypatch.setenv("CHORE_TRAIN_NO_INTERLEAVE", "1")
        else:
            monkeypatch.delenv("CHORE_TRAIN_NO_INTERLEAVE", raising=False)
        net, _ = _make(0)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        rec = []
        for it in range(2):
            net.train()
            opt.zero_grad(set_to_none=True)
            loss, sep = net(**batches[it])
            assert len(net.intermediate_preds_list) == 5 and net.preds is net.intermediate_preds_list[-1]
            loss.backward()
            opt.step()
            rec.append((float(loss.detach()), sep.detach().cpu(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
        torch.cuda.synchronize()
        res[kind] = rec
        del net, opt
        torch.cuda.empty_cache()
    for (la, sa, ga), (lb, sb, gb) in zip(res["sequential"], res["interleaved"]):
        assert la == lb, (la, lb)
        assert torch.equal(sa, sb)
        assert set(ga) == set(gb) and len(ga) >= 475
        bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
        assert not bad, bad[:5]


def test_eval_after_fused_training_sees_the_trained_weights():
    """The packed encoder / heads caches are keyed on parameter version counters, which a fused optimiser does not advance:
    CHORE.train() / eval() drop the packs on a mode change (round 4).  After three fused-Adam steps, an eval-mode filter + query of
    the trained model must equal that of a FRESH model loaded with the trained state_dict, bit for bit."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_ddp_trainstep import _make
    net, batch = _make(0)
    net.eval()
    with torch.no_grad():                       # packs built from the initial weights
        net.filter(batch["images"])
        net.query(batch["points"], crop_center=batch["crop_center"])
        before = [t.clone() for t in net.get_preds()]
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    for _ in range(3):
        net.train()
        opt.zero_grad(set_to_none=True)
        loss, _ = net(**batch)
        loss.backward()
        opt.step()
    net.eval()
    with torch.no_grad():
        net.filter(batch["images"])
        net.query(batch["points"], crop_center=batch["crop_center"])
        after = [t.clone() for t in net.get_preds()]
    fresh, _ = _make(0)
    fresh.load_state_dict(net.state_dict())
    fresh.eval()
    with torch.no_grad():
        fresh.filter(batch["images"])
        fresh.query(batch["points"], crop_center=batch["crop_center"])
        want = fresh.get_preds()
    assert not torch.equal(before[0], after[0])
    for a, b in zip(after, want):
        assert torch.equal(a, b)


def test_flat_grad_reducer_inplace_mode_equals_copy_mode():
    """FlatGradReducer(mode="inplace") keeps p.grad attached to the arena through the backward (autograd accumulates in place),
    mode="copy" gathers afterwards: same gradients, bit for bit, also for the parameters the loss does not reach (zeros)."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_ddp_trainstep import _make
    from chore_amd.parallel import FlatGradReducer
    out = {}
    for mode in ("copy", "inplace"):
        net, batch = _make(0)
        red = FlatGradReducer(net, mode=mode)
        net.train()
        red.zero_grad()
        loss, _ = net(**batch)
        loss.backward()
        red.reduce()
        out[mode] = {n: p.grad.clone() for n, p in net.named_parameters()}
        assert all(p.grad is not None and p.grad.data_ptr() >= red.arena.data_ptr() for p in net.parameters())
        del net, red
        torch.cuda.empty_cache()
    assert set(out["copy"]) == set(out["inplace"])
    bad = [n for n in out["copy"] if not torch.equal(out["copy"][n], out["inplace"][n])]
    assert not bad, bad[:5]
    assert sum(int(float(g.abs().max()) == 0.0) for g in out["copy"].values()) >= 82      # the bn4 affines without a downsample branch

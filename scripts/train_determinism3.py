"""Reproducer of DESIGN.md section 7 'run-to-run differences of the bf16 training step when three processes share one GPU':
a parent that used the GPU, then N child processes that each run the training step P times and compare every gradient tensor
with their OWN first pass, bit for bit.  Prints, per differing pass, how many tensors differ and -- in backward order (the
reverse of the registration order) -- the first ones: the point of the backward where the runs part.
usage: python scripts/train_determinism3.py [children=2] [passes=6]      env: any CHORE_* switch (CHORE_CONVBLOCK_SERIAL=1 ...)"""
import os, sys, subprocess, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))


ALLK = []       # KEEP_ALL=1: clones of every wrapped backward call's outputs (current pass)
KEPT = []       # KEEP_UPADD=1: (dy, dlow, shape) of every _UpAdd.backward call of the current pass
TRACE = []      # per backward call of a wrapped operator: (operator, checksums of the incoming gradients, of the outgoing ones)


def _ck(t):
    if t is None or not torch.is_tensor(t):
        return None
    v = t.detach().contiguous().reshape(-1).view(torch.uint8).clone()      # (clone: a view may start at any byte offset)
    if v.numel() % 8:
        v = torch.cat([v, v.new_zeros(8 - v.numel() % 8)])
    a = v.view(torch.int64)
    # position-weighted so that a permutation of equal bytes does not cancel
    return (a * (torch.arange(a.numel(), device=a.device) % 1021 + 1)).sum()


def _wrap(cls):
    orig = cls.backward

    def bw(ctx, *grads):
        ins = [_ck(g) for g in grads]
        out = orig(ctx, *grads)
        outs = [_ck(o) for o in (out if isinstance(out, tuple) else (out,))]
        TRACE.append((cls.__name__, ins, outs))
        if os.environ.get("KEEP_ALL"):
            ALLK.append([None if not torch.is_tensor(o) else o.detach().clone() for o in (out if isinstance(out, tuple) else (out,))])
        if cls.__name__ == "_UpAdd" and os.environ.get("KEEP_UPADD"):
            KEPT.append((grads[0].detach().clone(), out[1].detach().clone(), ctx.shape))
        return out
    cls.backward = staticmethod(bw)


def child(tag, passes):
    from test_gpu_ddp_trainstep import _make
    net, batch = _make(0, os.environ.get("CHORE_TRAIN_DTYPE", "bf16"))
    net.train(True)
    names = [n for n, _ in net.named_parameters()][::-1]
    if os.environ.get("TRACE_OPS"):
        from chore_amd import ops
        from chore_amd.model import chore as chore_mod
        for c in (ops._ConvBlock, ops._UpAdd, ops._AvgPool2, ops._ConvGN, ops._GNReLU, ops._Stem, chore_mod._QueryTrainFn, chore_mod._StackLossFn):
            _wrap(c)

    def grads():
        for p in net.parameters():
            p.grad = None
        err, _ = net(**batch)
        err.backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, float(err)
    for _ in range(int(os.environ.get("WARM_PASSES", "0"))):
        grads()
    TRACE.clear()
    KEPT.clear()
    ALLK.clear()
    g0, e0 = grads()
    k0 = list(KEPT)
    a0 = list(ALLK)
    t0 = [(n, [None if c is None else int(c) for c in i], [None if c is None else int(c) for c in o]) for n, i, o in TRACE]
    order = [n for n in names if n in g0]
    gp = g0
    for i in range(passes):
        TRACE.clear()
        KEPT.clear()
        ALLK.clear()
        g, e = grads()
        if a0 and ALLK:
            done = False
            for j, (oa, ob) in enumerate(zip(a0, ALLK)):
                for q, (x, y) in enumerate(zip(oa, ob)):
                    if x is not None and not torch.equal(x, y):
                        d = (x.float() - y.float()).abs()
                        per_img = d.reshape(d.shape[0], -1).amax(1).tolist() if d.dim() >= 2 and d.shape[0] == 4 else None
                        nz = int((d > 0).sum())
                        print(f"[{tag}] pass {i}: first differing output: call #{j} {TRACE[j][0]} output {q} shape {tuple(x.shape)} {x.dtype}: "
                              f"{nz} of {d.numel()} elements differ, max abs {float(d.max()):.3g} (max |x| {float(x.float().abs().max()):.3g}); "
                              f"per image max abs {per_img}; incoming checksums identical {t0[j][1] == [None if c is None else int(c) for c in TRACE[j][1]]}", flush=True)
                        done = True
                        break
                if done:
                    break
        if k0 and KEPT:
            from chore_amd import _lib
            for j, ((dy0, dl0, shp), (dy1, dl1, _)) in enumerate(zip(k0, KEPT)):
                if not torch.equal(dl0, dl1):
                    B, H, W, C = shp
                    again = torch.empty_like(dl0)
                    hh = _lib.handle(0)
                    _lib.check(_lib.lib.chore_up2_bwd(hh, _lib.BF16, dy0.contiguous().data_ptr(), again.data_ptr(), B, H, W, C,
                                                      torch.cuda.current_stream().cuda_stream), hh, "up2")
                    d = (dl0.float() - dl1.float()).abs().reshape(B, H, W, C)
                    idx = torch.nonzero(d.amax(-1) > 0)
                    print(f"[{tag}] pass {i}: _UpAdd call {j} shape {shp}: dy identical {torch.equal(dy0, dy1)}; recomputed now equals "
                          f"first-pass dlow {torch.equal(again, dl0)}, this-pass dlow {torch.equal(again, dl1)}; {idx.shape[0]} pixels differ "
                          f"(of {B*H*W}), channels differing per pixel max {int((d > 0).sum(-1).max())}, max abs {float(d.max()):.3g}; "
                          f"first pixels (b,y,x): {idx[:6].tolist()} last: {idx[-3:].tolist()}", flush=True)
                    break
        if t0:
            t1 = [(n, [None if c is None else int(c) for c in i_], [None if c is None else int(c) for c in o]) for n, i_, o in TRACE]
            assert len(t1) == len(t0)
            for j, (a, b) in enumerate(zip(t0, t1)):
                if a != b:
                    ins_same = a[1] == b[1]
                    print(f"[{tag}] pass {i}: first differing backward call #{j} of {len(t0)}: {a[0]}; its incoming gradients are "
                          f"{'IDENTICAL' if ins_same else 'different'}; outputs differing: {[k for k, (x, y) in enumerate(zip(a[2], b[2])) if x != y]}; "
                          f"previous calls: {[t0[q][0] for q in range(max(0, j - 3), j)]}", flush=True)
                    break
        same_prev = all(torch.equal(gp[n], g[n]) for n in order)
        gp = g
        print(f"[{tag}] pass {i}: equal to the previous pass: {same_prev}", flush=True)
        bad = [n for n in order if not torch.equal(g0[n], g[n])]
        if bad:
            first = order.index(bad[0])
            worst = max(float((g0[n].float() - g[n].float()).abs().max() / g0[n].float().abs().max().clamp_min(1e-30)) for n in bad)
            print(f"[{tag}] pass {i}: loss equal {e == e0}; {len(bad)} of {len(order)} tensors differ; first in backward order: #{first} "
                  f"{bad[0]}; then {bad[1:4]}; worst rel {worst:.1e}", flush=True)
        else:
            print(f"[{tag}] pass {i}: identical (loss equal {e == e0})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    nchild = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    if not os.environ.get("NO_PARENT_CONTEXT"):
        a = torch.randn(2048, 2048, device="cuda")
        (a @ a).sum().item()          # the parent holds a context (and its allocations) from here on
    procs = [subprocess.Popen([sys.executable, __file__, "child", f"c{i}", str(passes)]) for i in range(nchild)]
    for p in procs:
        p.wait()

"""Inner fitting step {loss; backward; Adam; early-stop bookkeeping} -- eager, or captured once as a hipGraph.

BASELINE configs[4] asks for a "hipGraph-captured inner iteration": a fitting step is ~200 small launches (field
queries, LBS, SO(3), loss reductions, their backward, Adam) whose GPU time is a few hundred microseconds while
issuing them from Python costs milliseconds.  Every operator of the step is free of host synchronisation (the
contact term included, csrc/contact.hip), so the whole step -- forward, backward into the persistent .grad
tensors, optimiser update and the early-stop test of recon/recon_fit_behave.py:143-153,279-287 -- is recorded
with torch.cuda.graph (hipGraph on ROCm) and replayed.  What changes between replays lives in device tensors:
the loss-weight decay, the previous loss, the stop flag, the index of the SO(3) perturbation noise.

Both steppers implement the reference's update rule exactly (gradients accumulate over the inner steps of an
outer iteration; the caller zeroes them through `begin_outer`; the early-stop rule is tested after every inner step and
everything after the step that meets it is a no-op, which equals the reference's immediate return).  Adam runs with capturable=True in the graph
stepper (same formulas evaluated on the device).
"""
import ctypes
import os
import threading

import torch
from torch import optim

from .. import _lib


def _row_strided(p):
    """dense, or a 2-D column slice (rows `stride(0)` apart, unit column stride) -- what chore_fit_adam_step_acc updates in place"""
    return p.is_contiguous() or (p.dim() == 2 and p.stride(1) == 1 and p.stride(0) >= p.shape[1])


class FusedAdam:
    """torch.optim.Adam(params, lr, betas, capturable=True) for a handful of small fp32 device tensors as ONE launch
    (chore_fit_adam_step, csrc/fit_step.hip), gated by the stepper's latched stop flag.  Same formulas, same state (step
    count, exp_avg, exp_avg_sq); the reference builds exactly this optimiser (recon_fit_behave.py:108, 236, 246)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params = list(params)
        if not 0 < len(self.params) <= 16 or any(p.dtype != torch.float32 or not p.is_cuda or not _row_strided(p) for p in self.params):
            raise ValueError("FusedAdam: 1..16 fp32 device tensors, dense or column slices of a 2-D tensor")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = self.params[0].device
        self.step_t = torch.zeros((), device=dev)
        self.m = [torch.zeros(p.shape, dtype=p.dtype, device=p.device) for p in self.params]      # dense, whatever p's strides
        self.v = [torch.zeros(p.shape, dtype=p.dtype, device=p.device) for p in self.params]
        self._args = None

    def state_tensors(self):
        return [self.step_t] + self.m + self.v

    @torch.no_grad()
    def reset(self):
        """a fresh optimiser on the same parameters (a recorded step keeps reading THESE state tensors)"""
        self.step_t.zero_()
        for t in self.m + self.v:
            t.zero_()

    def step(self, stop):
        """stop: device bool scalar; the step counter is advanced by the stop-rule launch that follows (EagerStep._one)"""
        live = [(p, p.grad, m, v) for p, m, v in zip(self.params, self.m, self.v) if p.grad is not None]
        if any(not p.is_contiguous() for p, _, _, _ in live):
            raise ValueError("FusedAdam.step: column-slice parameters need step_acc")
        key = tuple((p.data_ptr(), g.data_ptr()) for p, g, _, _ in live)
        if self._args is None or self._args[0] != key:
            n = len(live)
            arr = lambda xs: (ctypes.c_void_p * n)(*xs)   # noqa: E731
            self._args = (key, arr([p.data_ptr() for p, _, _, _ in live]), arr([g.data_ptr() for _, g, _, _ in live]),
                          arr([m.data_ptr() for _, _, m, _ in live]), arr([v.data_ptr() for _, _, _, v in live]),
                          (ctypes.c_int * n)(*[p.numel() for p, _, _, _ in live]), n)
        _, pp, gg, mm, vv, nn, n = self._args
        if n == 0:
            return
        dev = self.params[0].device
        for _, g, _, _ in live:
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("FusedAdam: gradients must be contiguous fp32")
        h = _lib.handle(dev.index or 0)
        _lib.check(_lib.lib.chore_fit_adam_step(h, pp, gg, mm, vv, nn, n, self.step_t.data_ptr(), self.lr, self.betas[0],
                                                self.betas[1], self.eps, stop.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_adam_step")


    def step_acc(self, stop, leaves, new_grads, counted=False):
        """Adam with autograd's accumulation folded into the launch (chore_fit_adam_step_acc): leaves = the parameters of
        this optimiser followed by the leaves that only accumulate; new_grads[i] = this step's gradient of leaves[i]
        (torch.autograd.grad; None = the loss does not reach it).  Equals  leaf.grad += new  for every leaf, then step().
        counted: the step counter already includes this step (the fused sum + rule launch advanced it)."""
        state = {id(p): (m, v) for p, m, v in zip(self.params, self.m, self.v)}
        rows = []
        for leaf, g in zip(leaves, new_grads):
            if g is None and leaf.grad is None:
                continue                                   # torch's Adam skips a parameter without a gradient
            if leaf.grad is None:
                leaf.grad = torch.zeros(leaf.shape, dtype=leaf.dtype, device=leaf.device)    # dense; 0 + new = new, bit for bit what the first accumulation stores
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.float().contiguous()
            mv = state.get(id(leaf))
            if mv is None and g is None:
                continue                                   # accumulate-only leaf, nothing new
            if not leaf.grad.is_contiguous():
                raise ValueError("FusedAdam.step_acc: .grad must be dense")
            rows.append((leaf if mv is not None else None, leaf.grad, g, mv))
        if not rows:
            return
        if len(rows) > 16:
            raise ValueError("FusedAdam.step_acc: at most 16 tensors")
        # cached on what persists from step to step -- the parameters and their .grad accumulators; this step's fresh gradients
        # change address every call and are written into the cached pointer array below (a key that contained them never hit,
        # and keeping them alive for it stopped the allocator from reusing their addresses: ADVICE round 3)
        key = tuple((0 if p is None else p.data_ptr(), a.data_ptr()) for p, a, _, _ in rows)
        if self._args is None or self._args[0] != key:
            n = len(rows)
            arr = lambda xs: (ctypes.c_void_p * n)(*xs)   # noqa: E731
            self._args = (key, arr([None if p is None else p.data_ptr() for p, _, _, _ in rows]), arr([a.data_ptr() for _, a, _, _ in rows]),
                          arr([None] * n),
                          arr([None if mv is None else mv[0].data_ptr() for _, _, _, mv in rows]),
                          arr([None if mv is None else mv[1].data_ptr() for _, _, _, mv in rows]),
                          (ctypes.c_int * n)(*[a.numel() for _, a, _, _ in rows]),
                          (ctypes.c_int * n)(*[(a.numel() if (p is None or p.is_contiguous()) else p.shape[1]) for p, a, _, _ in rows]),
                          (ctypes.c_int * n)(*[(a.numel() if (p is None or p.is_contiguous()) else p.stride(0)) for p, a, _, _ in rows]), n)
        _, pp, aa, gg, mm, vv, nn, cc, ss, n = self._args
        for i, (_, _, g, _) in enumerate(rows):      # stream-ordered allocation keeps g valid until the launch below has run
            gg[i] = None if g is None else g.data_ptr()
        dev = self.params[0].device
        h = _lib.handle(dev.index or 0)
        _lib.check(_lib.lib.chore_fit_adam_step_acc(h, pp, aa, gg, mm, vv, nn, cc, ss, n, self.step_t.data_ptr(), self.lr, self.betas[0],
                                                    self.betas[1], self.eps, stop.data_ptr(), 1 if counted else 0,
                                                    torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_adam_step_acc")


class _WeightedSum(torch.autograd.Function):
    """sum_k c_k L_k / denom over device scalars (chore_fit_weighted_sum, csrc/fit_step.hip)"""

    @staticmethod
    def forward(ctx, denom, coeffs, *losses):
        dev = denom.device
        h = _lib.handle(dev.index or 0)
        n = len(losses)
        out = torch.empty((), device=dev)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in losses])
        cs = (ctypes.c_float * n)(*coeffs)
        _lib.check(_lib.lib.chore_fit_weighted_sum(h, ptrs, cs, n, denom.data_ptr(), out.data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_weighted_sum")
        ctx.save_for_backward(denom)
        ctx.cs, ctx.n = cs, n
        return out

    @staticmethod
    def backward(ctx, g):
        (denom,) = ctx.saved_tensors
        dev = denom.device
        h = _lib.handle(dev.index or 0)
        g = g.contiguous().float()
        grads = torch.empty(ctx.n, device=dev)
        _lib.check(_lib.lib.chore_fit_weighted_sum_bwd(h, ctx.cs, ctx.n, denom.data_ptr(), g.data_ptr(), grads.data_ptr(),
                                                       torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_fit_weighted_sum_bwd")
        return (None, None) + tuple(grads[k] for k in range(ctx.n))


class _WeightedSumStep(torch.autograd.Function):
    """the same sum inside a stepper: one launch also writes the sum's gradients for the stepper's own upstream gradient (its
    `seed`) and runs the stop rule on the sum (chore_fit_weighted_sum_step).  backward() launches nothing when it is handed
    that very seed; any other upstream gradient takes the stand-alone backward launch."""

    @staticmethod
    def forward(ctx, denom, coeffs, st, *losses):
        dev = denom.device
        h = _lib.handle(dev.index or 0)
        n = len(losses)
        out = torch.empty((), device=dev)
        grads = torch.empty(n, device=dev)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in losses])
        cs = (ctypes.c_float * n)(*coeffs)
        _lib.check(_lib.lib.chore_fit_weighted_sum_step(
            h, ptrs, cs, n, denom.data_ptr(), out.data_ptr(), st.seed.data_ptr(), grads.data_ptr(), st.prev.data_ptr(),
            st.stop.data_ptr(), st.frozen.data_ptr(), st.armed.data_ptr(), float(st.tol), st.loss.data_ptr(),
            st.opt.step_t.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_weighted_sum_step")
        ctx.save_for_backward(denom)
        ctx.cs, ctx.n, ctx.grads, ctx.seed_ptr = cs, n, grads, st.seed.data_ptr()
        return out

    @staticmethod
    def backward(ctx, g):
        grads = ctx.grads
        if g.data_ptr() != ctx.seed_ptr or g.dtype != torch.float32:
            (denom,) = ctx.saved_tensors
            dev = denom.device
            h = _lib.handle(dev.index or 0)
            g = g.contiguous().float()
            grads = torch.empty(ctx.n, device=dev)
            _lib.check(_lib.lib.chore_fit_weighted_sum_bwd(h, ctx.cs, ctx.n, denom.data_ptr(), g.data_ptr(), grads.data_ptr(),
                                                           torch.cuda.current_stream(dev).cuda_stream), h,
                       "chore_fit_weighted_sum_bwd")
        return (None, None, None) + tuple(grads[k] for k in range(ctx.n))


# The stepper whose loss is being formed on this host thread and wants its stop rule inside the sum's launch (EagerStep._one);
# weighted_sum takes the request (at most once per step) and says so.
_step_request = threading.local()


def weighted_sum(losses, coeffs, denom):
    if os.environ.get("CHORE_FIT_TORCH_SUM"):
        return torch.stack([c * v / denom for c, v in zip(coeffs, losses)]).sum()
    st = getattr(_step_request, "stepper", None)
    if st is not None and denom is st.denom:
        _step_request.stepper = None
        out = _WeightedSumStep.apply(denom, tuple(coeffs), st, *losses)
        _step_request.taken = out
        return out
    return _WeightedSum.apply(denom, tuple(coeffs), *losses)


def _fused_ok(params):
    strided_ok = not os.environ.get("CHORE_FIT_BACKWARD_ACCUMULATE")       # only the accumulate-in-Adam launch takes column slices
    return (not os.environ.get("CHORE_FIT_TORCH_ADAM")) and 0 < len(params) <= 16 and all(
        p.is_cuda and p.dtype == torch.float32 and (p.is_contiguous() or (strided_ok and _row_strided(p))) for p in params)


class _OnePlusDecay:
    """stands for the decay `it` in the reference's weight formula  w * L / (1 + it)  (recon_fit_behave.py:339-358):
    `1 + it` evaluates to a device tensor that holds float32(1 + it) -- the same rounding as the Python-scalar
    operand of the eager reference, but changeable between graph replays"""

    def __init__(self, denom):
        self.denom = denom

    def __radd__(self, one):
        if one != 1:
            raise ValueError("decay is only meaningful as 1 + decay")
        return self.denom


class EagerStep:
    def __init__(self, params, lr, loss_fn, tol, prev, betas=(0.9, 0.999), state=(), opt=None, release=None,
                 capturable=False, carry=(), fuse_rule=False):
        """fuse_rule: loss_fn returns what weighted_sum() returned, untouched -- the stop rule then runs inside that launch
        (a loss_fn that post-processes the sum must leave this off: the rule would have tested the wrong value; _one raises);
        opt: continue with an existing optimiser (a phase that only changes the loss, recon_fit_behave.py:252-254);
        capturable: evaluate Adam's bias corrections on the device like the graph stepper does (bit-comparable runs);
        carry: parameters this phase does not step but whose .grad keeps accumulating (every leaf the loss reaches does in
        the reference, and a later phase's new Adam starts from those sums, recon_fit_behave.py:243-259)"""
        self.params = list(params)
        self.carry = [p for p in carry if all(p is not q for q in self.params)]
        self.opt = opt if opt is not None else self._make_opt(lr, betas, capturable)
        self.loss_fn, self.tol, self.prev = loss_fn, tol, prev
        self.release = release
        self.fuse_rule = bool(fuse_rule)
        self._init_flags(self.params[0].device)

    def _make_opt(self, lr, betas, capturable):
        """the reference's Adam: as one HIP launch where the tensors allow (CHORE_FIT_TORCH_ADAM=1: torch's)"""
        if _fused_ok(self.params):
            return FusedAdam(self.params, lr, betas)
        return optim.Adam(self.params, lr=lr, betas=betas, capturable=capturable)

    def _opt_state_tensors(self):
        if isinstance(self.opt, FusedAdam):
            return self.opt.state_tensors()
        return [v for st in self.opt.state.values() for v in st.values() if torch.is_tensor(v)]

    def _init_flags(self, dev):
        self.denom = torch.ones((), device=dev)           # 1 + decay
        self.armed = torch.zeros((), dtype=torch.bool, device=dev)   # the early-stop rule is live in this outer iteration
        self.stop = torch.zeros((), dtype=torch.bool, device=dev)    # latched: the reference has returned
        self.frozen = torch.zeros((), dtype=torch.bool, device=dev)  # `stop` as it was before the current step's test (fused rule)
        self.loss = torch.zeros((), device=dev)
        self.seed = torch.ones((), device=dev)            # d loss / d loss: handed to backward() (it fills a fresh one per call otherwise)

    @torch.no_grad()
    def reset(self, reset_opt=True):
        """make a kept stepper (recon_fit_behave._FitSlot) equal to a newly built one: flags cleared, the optimiser's state zeroed
        unless it continues another stepper's (reset_opt=False).  Only steppers on a FusedAdam are kept."""
        self.stop.fill_(False)
        self.frozen.fill_(False)
        self.armed.fill_(False)
        self.loss.zero_()
        self.denom.fill_(1.0)
        if reset_opt:
            self.opt.reset()
        if self.release is not None:
            self.release()

    def zero_grads(self):
        """what optimizer.zero_grad() did in the reference's torch (zero in place; the tensors stay: a recorded graph
        accumulates into them)"""
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def begin_outer(self, decay, armed=False, zero=True):
        """top of an outer iteration (recon_fit_behave.py:117-118, 242-243).  zero=False: the first iteration of a phase
        whose optimiser was re-created -- the reference zeroed through the PREVIOUS optimiser there"""
        if zero:
            self.zero_grads()
        self.armed.fill_(bool(armed))
        self.denom.fill_(1 + decay)

    def _one(self):
        # The reference returns from INSIDE the inner loop at the step that meets the stop rule, after that step's
        # update (recon_fit_behave.py:158-160, 278-285).  Here the host looks at the flag once per outer iteration, so
        # the remaining inner steps still run -- as no-ops: once `stop` is latched every later step leaves the
        # parameters and the previous loss as they were.
        if isinstance(self.opt, FusedAdam):
            # two launches: Adam on all tensors (the latched flag freezes the parameters), then the stop rule + step counter --
            # or, with fuse_rule, the rule inside the launch that sums the loss terms (before Adam, which then reads `frozen`)
            fuse = self.fuse_rule and not os.environ.get("CHORE_FIT_BACKWARD_ACCUMULATE")
            _step_request.stepper, _step_request.taken = (self if fuse else None), None
            try:
                loss = self.loss_fn(_OnePlusDecay(self.denom))
            finally:
                _step_request.stepper = None
                taken, _step_request.taken = getattr(_step_request, "taken", None), None
            if taken is not None and taken is not loss:
                raise RuntimeError("fit step: the stop rule ran inside weighted_sum() on a value that is not the step's loss "
                                   "(loss_fn changed the sum): build the stepper with fuse_rule=False")
            fused = taken is not None
            seed = self.seed if loss.dtype == self.seed.dtype and loss.dim() == 0 else None
            if os.environ.get("CHORE_FIT_BACKWARD_ACCUMULATE"):        # A/B switch: .backward() + one add launch per leaf
                loss.backward(seed)
                self.opt.step(self.stop)
            else:
                # this step's gradients as fresh tensors; `grad += new` of every leaf happens inside the Adam launch.  The
                # leaves: the optimiser's parameters and `carry` -- every other leaf the loss reaches is not stepped by any
                # later optimiser of the fit (the caller's contract, see __init__)
                leaves = [p for p in self.opt.params] + [p for p in self.carry if all(p is not q for q in self.opt.params)]
                leaves += [p for p in self.params if all(p is not q for q in leaves)]
                if not getattr(self, "_leaves_checked", False):
                    self._check_leaves(loss, leaves)
                    self._leaves_checked = True
                grads = torch.autograd.grad(loss, leaves, seed, allow_unused=True)
                self.opt.step_acc(self.frozen if fused else self.stop, leaves, grads, counted=fused)
            if fused:
                return
            lv = loss.detach()
            if lv.dtype != torch.float32:
                lv = lv.float()
            dev = lv.device
            h = _lib.handle(dev.index or 0)
            _lib.check(_lib.lib.chore_fit_stop_rule(h, lv.data_ptr(), self.prev.data_ptr(), self.stop.data_ptr(),
                                                    self.armed.data_ptr(), float(self.tol), self.loss.data_ptr(),
                                                    self.opt.step_t.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       h, "chore_fit_stop_rule")
            return
        frozen = self.stop.clone()
        saved = [p.detach().clone() for p in self.params]
        loss = self.loss_fn(_OnePlusDecay(self.denom))
        loss.backward()
        self.opt.step()
        with torch.no_grad():
            for p, s in zip(self.params, saved):
                p.copy_(torch.where(frozen, s, p))
            lv = loss.detach()
            hit = torch.abs(self.prev - lv) / self.prev < self.prev * self.tol
            self.stop.logical_or_(hit & self.armed)
            self.prev.copy_(torch.where(frozen, self.prev, lv))
            self.loss.copy_(lv)

    def _check_leaves(self, loss, leaves):
        """first step of a stepper: every leaf the loss reaches must be in `leaves`.  The reference's backward() accumulates
        into EVERY leaf; torch.autograd.grad only returns what it is asked for, so a requires-grad leaf missing from
        params + carry would silently stop accumulating (and a later phase's Adam would start from other sums than the
        reference's).  A walk over the graph's AccumulateGrad nodes, no launches.  CHORE_FIT_NO_LEAF_CHECK=1 skips it."""
        if os.environ.get("CHORE_FIT_NO_LEAF_CHECK") or loss.grad_fn is None:
            return
        known = {id(p) for p in leaves}
        seen, stack, missing = set(), [loss.grad_fn], []
        while stack:
            fn = stack.pop()
            if fn is None or id(fn) in seen:
                continue
            seen.add(id(fn))
            v = getattr(fn, "variable", None)       # AccumulateGrad nodes carry the leaf
            if v is not None and v.requires_grad and id(v) not in known:
                missing.append(tuple(v.shape))
            stack.extend(f for f, _ in fn.next_functions)
        if missing:
            raise RuntimeError("fit step: the loss reaches %d requires-grad leaf tensor(s) that are neither stepped nor listed in "
                               "`carry` (shapes %s): their .grad would not accumulate like the reference's backward() does"
                               % (len(missing), missing[:4]))

    def step(self):
        self._one()

    def stopped(self):
        return bool(self.stop)   # the only host synchronisation, once per outer iteration


class GraphedStep(EagerStep):
    """same step, recorded once and replayed.  `state`: extra device tensors the step mutates in place (e.g. the
    noise index) -- they are snapshotted around the warm-up runs, which must not leave a trace in the fit."""

    def __init__(self, params, lr, loss_fn, tol, prev, betas=(0.9, 0.999), state=(), opt=None, release=None,
                 capturable=True, warmup=2, carry=(), fuse_rule=False):
        """release: drops every reference to autograd graphs of earlier steps (cached predictions, concatenated
        parameters).  The gradient accumulators of the parameters live as long as such a graph does and stay bound
        to the stream they were created on; recording needs them re-created on the capture stream."""
        self.params = list(params)
        dev = self.params[0].device
        carry = [p for p in carry if all(p is not q for q in self.params)]
        self.carry = carry
        for p in self.params + carry:
            if p.grad is None:       # a recorded backward must ADD into a tensor that exists (a missing .grad would be
                p.grad = torch.zeros_like(p)   # replaced by a graph-private tensor and overwritten on every replay)
        self.opt = opt if opt is not None else self._make_opt(lr, betas, True)
        self.loss_fn, self.tol, self.prev = loss_fn, tol, prev
        self.fuse_rule = bool(fuse_rule)
        self._init_flags(dev)
        mutable = [p.data for p in self.params] + [p.grad for p in self.params + carry] + [self.prev, self.stop, self.frozen, self.loss]
        mutable += list(state)
        mutable += self._opt_state_tensors()   # a continued Adam (torch's creates its state lazily; see _restore)
        snap = [t.clone() for t in mutable]
        known = {id(t) for t in mutable}
        self.release = release
        if release is not None:
            release()
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)   # warm-up and recording on the same stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):   # allocator pools, lazy Adam state, kernel attributes: everything one-off
                self._one()
            self._restore(mutable, snap, known)
        cur.wait_stream(side)
        if torch.distributed.is_available() and torch.distributed.is_initialized():      # frame-sharded fit: see drain_collectives
            from ..parallel import drain_collectives
            torch.cuda.synchronize(dev)
            drain_collectives()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._one()
        self._restore(mutable, snap, known)   # recording does not execute, but keep the contract explicit

    def _restore(self, mutable, snap, known):
        with torch.no_grad():
            for t, s in zip(mutable, snap):
                t.copy_(s)
            for v in self._opt_state_tensors():   # state created lazily by the warm-up = a fresh optimiser: zero
                if id(v) not in known:
                    v.zero_()
        if self.release is not None:
            self.release()   # results memoised during the warm-up are stale now (.data writes bump no version)

    def step(self):
        self.graph.replay()

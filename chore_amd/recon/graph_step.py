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
import torch
from torch import optim


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
                 capturable=False, carry=()):
        """opt: continue with an existing optimiser (a phase that only changes the loss, recon_fit_behave.py:252-254);
        capturable: evaluate Adam's bias corrections on the device like the graph stepper does (bit-comparable runs);
        carry: parameters this phase does not step but whose .grad keeps accumulating (every leaf the loss reaches does in
        the reference, and a later phase's new Adam starts from those sums, recon_fit_behave.py:243-259)"""
        self.params = list(params)
        self.opt = opt if opt is not None else optim.Adam(self.params, lr=lr, betas=betas, capturable=capturable)
        self.loss_fn, self.tol, self.prev = loss_fn, tol, prev
        self._init_flags(self.params[0].device)

    def _init_flags(self, dev):
        self.denom = torch.ones((), device=dev)           # 1 + decay
        self.armed = torch.zeros((), dtype=torch.bool, device=dev)   # the early-stop rule is live in this outer iteration
        self.stop = torch.zeros((), dtype=torch.bool, device=dev)    # latched: the reference has returned
        self.loss = torch.zeros((), device=dev)

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

    def step(self):
        self._one()

    def stopped(self):
        return bool(self.stop)   # the only host synchronisation, once per outer iteration


class GraphedStep(EagerStep):
    """same step, recorded once and replayed.  `state`: extra device tensors the step mutates in place (e.g. the
    noise index) -- they are snapshotted around the warm-up runs, which must not leave a trace in the fit."""

    def __init__(self, params, lr, loss_fn, tol, prev, betas=(0.9, 0.999), state=(), opt=None, release=None,
                 capturable=True, warmup=2, carry=()):
        """release: drops every reference to autograd graphs of earlier steps (cached predictions, concatenated
        parameters).  The gradient accumulators of the parameters live as long as such a graph does and stay bound
        to the stream they were created on; recording needs them re-created on the capture stream."""
        self.params = list(params)
        dev = self.params[0].device
        carry = [p for p in carry if all(p is not q for q in self.params)]
        for p in self.params + carry:
            if p.grad is None:       # a recorded backward must ADD into a tensor that exists (a missing .grad would be
                p.grad = torch.zeros_like(p)   # replaced by a graph-private tensor and overwritten on every replay)
        self.opt = opt if opt is not None else optim.Adam(self.params, lr=lr, betas=betas, capturable=True)
        self.loss_fn, self.tol, self.prev = loss_fn, tol, prev
        self._init_flags(dev)
        mutable = [p.data for p in self.params] + [p.grad for p in self.params + carry] + [self.prev, self.stop, self.loss]
        mutable += list(state)
        mutable += [v for st in self.opt.state.values() for v in st.values() if torch.is_tensor(v)]   # continued Adam
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
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._one()
        self._restore(mutable, snap, known)   # recording does not execute, but keep the contract explicit

    def _restore(self, mutable, snap, known):
        with torch.no_grad():
            for t, s in zip(mutable, snap):
                t.copy_(s)
            for st in self.opt.state.values():   # state created lazily by the warm-up = a fresh optimiser: zero
                for v in st.values():
                    if torch.is_tensor(v) and id(v) not in known:
                        v.zero_()
        if self.release is not None:
            self.release()   # results memoised during the warm-up are stale now (.data writes bump no version)

    def step(self):
        self.graph.replay()

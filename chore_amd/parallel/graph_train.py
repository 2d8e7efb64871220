"""The training step as hipGraph replays (the step is the reference's `Trainer.train_step`,
trainer/trainer.py:76-85: zero_grad -> model(**batch) -> backward -> optimizer.step).

Why: one step of CHORE is ~1 600 kernel launches on two streams, most of them 5 - 30 us long and dependent on their
predecessor, issued from ~330 autograd nodes.  Issued eagerly the device idles about a quarter of the step waiting for the host
(profiles/r03_train_kernel_stats.txt: 27 % idle; the gaps sit in front of the first kernel of every ConvBlock backward, of the
optimiser and of the step).  The shapes of a training run never change, so the whole step is recorded once and replayed:

    graph A   zero the gradients -> forward -> backward -> gather the gradients into FlatGradReducer's arena
    (eager)   the arena all-reduced over RCCL in a few large chunks, scaled to the mean          [only with a process group]
    graph B   optimizer.step()                                                                   [one graph A+B without a group]

The first `warmup` calls run eagerly on a side stream (what torch asks for before a capture: every lazily initialised piece --
kernel attributes, the library's side stream, the optimiser's state -- exists before the recording starts); the next call
records and replays; every call performs exactly one training step on the batch it was given.

What differs from the eager sequence: `torch.autograd.set_detect_anomaly(True)` of the reference's step cannot be recorded (it
reads every gradient back to the host) and is left out; the loss comes back as a device tensor (`.item()` it when a number is
needed: the one host synchronisation of the step); the optimiser must be built with `capturable=True`.  The arithmetic is the
same: tests/test_gpu_graph_train.py compares parameters after replayed steps with eager steps bit for bit.

Hyper-parameters under replay (the reference decays the rate with MultiStepLR(gamma=0.3) and rewrites `g['lr']` when it resumes,
trainer/trainer.py:56-60,246-257).  A Python-float `lr` would be baked into the recorded Adam launch, so the constructor moves
every group's `lr` into a one-element fp32 DEVICE tensor, which the capturable Adam kernels read at run time: schedulers fill
that tensor in place (torch.optim.lr_scheduler does for tensor rates), and a float written over it (`g['lr'] = 3e-5`) is copied
into the tensor -- and the tensor put back -- at the next call.  The remaining group entries (betas, eps, weight_decay, amsgrad,
maximize) and the identity of the optimiser's state tensors are compared with what was recorded at every call:
`optimizer.load_state_dict()` or an edited beta drops the recordings and the step is recorded again.  A rate of 1e-4 held as
fp32 differs from the Python double by 3e-9 relative.

A recording is made per batch signature (shapes / dtypes / non-tensor arguments): the FIRST call of a new signature runs
eagerly (per-shape lazy initialisation happens outside any capture), the second one records.  At most `max_recordings` live at
a time (each holds the step's activations in a private pool: ~10 GB at the configs[3] size); the signatures beyond that -- the
short last batch of an epoch -- keep running eagerly.

The all-reduce UNDER the backward (round 5): with a reducer built with `segments=chore_segments(model)` the recording is cut at
the hourglass stacks -- graph A0 = zero + forward + the backward of the last stack + its gradients' gather, A1 .. A5 = the backward
of one earlier stack (finally: of what precedes the first stack) + gather -- and each segment's slice of the gradient arena goes
to RCCL (asynchronously, on RCCL's stream) the moment its graph has been issued, so it travels while the following graphs run;
only the last, 3.6 MB segment's collective is exposed before graph B (the optimiser).  Same autograd nodes, same order: the
gradients equal the unsegmented step's bit for bit (tests/test_gpu_ddp_nccl.py).

Side effect: `model.losses_on_host` is switched off (the reference returns the six separate losses as a CPU tensor, a host copy
per step that cannot be recorded); `close()` restores it.
"""
import os

import torch


def drain_collectives(*groups):
    """Before a hipGraph recording in a process that has issued collectives: none of them may still sit in its process group's
    watchdog list.  The watchdog thread polls the end events of the works it holds (every 100 ms, until it has seen each one
    finished); with works of the eager warm-up steps still listed when a capture started, the process died about once in eight
    starts with `watchdog thread terminated with exception: HIP error: operation not permitted on an event last recorded in a
    capturing stream` (scripts/probes/graph_record_watchdog.py: 2 of 12 starts; 0 of 40 with the list drained first).  torch's own
    capture_begin waits for this in the global capture mode; GraphedTrainStep records thread-local (other threads of the process
    may touch events), so the wait is done here -- and, as a second line, before the package's other recordings too.
    Call with the device idle (torch.cuda.synchronize()): the wait is for the watchdog's next pass, not for the GPU."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return
    todo = {}
    for g in (dist.group.WORLD,) + groups:
        if g is not None:
            todo[id(g)] = g
    for g in todo.values():
        # (no test on the backend's NAME first: a group made with "cpu:gloo,cuda:nccl" -- bench.py's -- reports that string)
        try:
            g._wait_for_pending_works()       # the group's device backend; raises for a gloo-only group, whose works carry no device events
        except Exception:
            if "nccl" in str(dist.get_backend(g)).lower():
                import time
                time.sleep(0.4)               # (an older torch without the call: four passes of the watchdog)


class GraphedTrainStep:
    def __init__(self, model, optimizer, reducer=None, warmup=3, max_recordings=2):
        """model: CHORE (or a wrapper with the same call surface) returning (loss, separate losses); optimizer: a torch
        optimiser over its parameters built with capturable=True; reducer: chore_amd.parallel.FlatGradReducer or None."""
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep: build the optimiser with capturable=True (its step counter must live on "
                                 "the device to be recorded)")
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        inner = getattr(model, "module", model)
        self._losses_on_host_before = getattr(inner, "losses_on_host", None)
        if self._losses_on_host_before:
            inner.losses_on_host = False      # the reference returns the separate losses as a CPU tensor: a host copy per step
        self.warmup = int(warmup)
        self.max_recordings = int(max_recordings)
        self.calls = 0
        self._rec = {}            # batch signature -> recording
        self._seen = set()        # signatures that have run eagerly once
        self._side = None
        self._seg_params = None
        # the learning rates as device tensors (see the module docstring)
        dev = next(model.parameters()).device
        self._lr = []
        for g in ([] if os.environ.get("CHORE_GRAPH_FLOAT_LR") else optimizer.param_groups):      # (debug switch: keep the float rate)
            lr = g["lr"]
            t = lr if (torch.is_tensor(lr) and lr.device == dev) else torch.tensor(float(lr), dtype=torch.float32, device=dev)
            g["lr"] = t
            self._lr.append(t)
        self._hyper = None        # what the live recordings were made with

    def close(self):
        """drop the recordings (and their memory pools) and give `model.losses_on_host` its old value back"""
        self._rec.clear()
        inner = getattr(self.model, "module", self.model)
        if self._losses_on_host_before is not None:
            inner.losses_on_host = self._losses_on_host_before

    # ---- optimiser hyper-parameters and state under replay ---------------------------------------------------------------
    def _sync_lr(self):
        for g, t in zip(self.optimizer.param_groups, self._lr):
            lr = g["lr"]
            if lr is not t:                   # `g['lr'] = value` (the reference's resume path) or a scheduler that assigns
                t.fill_(float(lr))
                g["lr"] = t

    def _hyper_key(self):
        """what the live recordings depend on besides the batch: the optimiser's scalars and the addresses of its state.  Runs
        on every replayed step, so it must not touch the device: a tensor-valued entry (a scheduler built after this object
        stores `initial_lr` as a device tensor; tensor betas) is keyed by (address, version), never by repr(), which would
        copy it to the host and wait for the stream; the walk over the ~1.4 k state tensors is redone only when the state's
        size or its first / last tensor changed."""
        opt = self.optimizer

        def key(v):
            return ("tensor", v.data_ptr(), v._version) if torch.is_tensor(v) else repr(v)
        groups = tuple(tuple((k, key(v)) for k, v in sorted(g.items()) if k not in ("params", "lr")) for g in opt.param_groups)
        st = opt.state
        sig = (len(st),)
        if st:
            ends = []
            for g in (opt.param_groups[0], opt.param_groups[-1]):
                for p in (g["params"][0], g["params"][-1]):
                    ends.append(tuple(v.data_ptr() for v in st[p].values() if torch.is_tensor(v)) if p in st else None)
            sig = (len(st), tuple(ends))
        cached = self.__dict__.get("_state_key")
        if cached is None or cached[0] != sig:
            state = tuple((id(p), tuple((k, v.data_ptr()) for k, v in sorted(st[p].items()) if torch.is_tensor(v)))
                          for g in opt.param_groups for p in g["params"] if p in st)
            cached = self._state_key = (sig, state)
        return groups, cached[1]

    # ---- the step, as the eager sequence --------------------------------------------------------------------------------
    def _zero(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def _segmented(self):
        return self.reducer is not None and getattr(self.reducer, "segments", None) is not None

    def _fwd_bwd(self, batch, on_segment=None):
        """on_segment (segmented reducer only): called between the segments of the backward -- the eager step launches the
        segment's collective there, the recording closes one graph and opens the next"""
        self._zero()
        if not self._segmented():
            loss, sep = self.model(**batch)
            loss.backward()
            if self.reducer is not None:
                self.reducer.gather()
            return loss, sep
        from .grad_arena import backward_in_segments, chore_segments
        inner = getattr(self.model, "module", self.model)
        inner.keep_train_segments = True
        try:
            loss, sep = self.model(**batch)
        finally:
            inner.keep_train_segments = False
        seg = inner.train_segments
        inner.train_segments = None
        if self._seg_params is None:
            self._seg_params = chore_segments(inner)

        def after(k):
            self.reducer.gather(k)
            if on_segment is not None:
                on_segment(k)
        backward_in_segments(seg["stack_losses"], seg["cuts"], self._seg_params, after)
        return loss, sep

    def _invalidate(self):
        # packed copies of the weights (heads arena, inference encoder) are keyed on the parameters' version counters, which a
        # replayed optimiser step does not advance
        m = getattr(self.model, "module", self.model)
        if hasattr(m, "invalidate_packed"):
            m.invalidate_packed()

    def _eager(self, batch):
        dev = next(self.model.parameters()).device
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
        self._side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._side):
            if self._segmented():
                loss, sep = self._fwd_bwd(batch, on_segment=self.reducer.all_reduce_async)
                self.reducer.finish()
            else:
                loss, sep = self._fwd_bwd(batch)
                if self.reducer is not None:
                    self.reducer.all_reduce()
            self.optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(self._side)
        self._invalidate()
        return loss.detach(), sep

    # ---- recording -----------------------------------------------------------------------------------------------------
    @staticmethod
    def _key(batch):
        return tuple((k, tuple(v.shape), str(v.dtype)) if torch.is_tensor(v) else (k, repr(v)) for k, v in sorted(batch.items()))

    def _record(self, batch):
        self.model.train()
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        self._invalidate()
        split = self.reducer is not None and (self.reducer.world > 1 or torch.distributed.is_initialized())
        ga = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        drain_collectives(getattr(self.reducer, "group", None))
        # a process group's watchdog thread polls events while we record: only THIS thread's calls belong to the recording
        mode = dict(capture_error_mode="thread_local") if torch.distributed.is_initialized() else {}
        if split and self._segmented():
            # one graph per segment of the backward: a capture is closed where the segment's collective will be launched and the
            # next one opened on the same stream and memory pool (the autograd graph and its saved tensors carry over)
            graphs = [ga]
            state = {"cm": torch.cuda.graph(ga, **mode)}
            state["cm"].__enter__()
            nseg = len(self.reducer.segments)

            def cut(k):
                state["cm"].__exit__(None, None, None)
                state["cm"] = None
                if k + 1 < nseg:
                    g = torch.cuda.CUDAGraph()
                    graphs.append(g)
                    state["cm"] = torch.cuda.graph(g, pool=ga.pool(), **mode)
                    state["cm"].__enter__()
            try:
                loss, sep = self._fwd_bwd(static, on_segment=cut)
            finally:
                if state["cm"] is not None:
                    state["cm"].__exit__(None, None, None)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool(), **mode):
                self.optimizer.step()
            return dict(static=static, ga=ga, gb=gb, segs=graphs, loss=loss.detach(), sep=sep)
        with torch.cuda.graph(ga, **mode):
            loss, sep = self._fwd_bwd(static)
            if not split:
                self.optimizer.step()
        gb = None
        if split:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool(), **mode):
                self.optimizer.step()
        return dict(static=static, ga=ga, gb=gb, segs=None, loss=loss.detach(), sep=sep)

    def __call__(self, **batch):
        """one training step on `batch` (keyword arguments of CHORE.forward; tensors on the device).  Returns (loss, separate
        losses) as device tensors that the NEXT call overwrites."""
        self.calls += 1
        self._sync_lr()
        key = self._key(batch)
        if self.calls <= self.warmup or key not in self._seen:
            self._seen.add(key)
            self.model.train()
            return self._eager(batch)
        hyper = self._hyper_key()
        if self._rec and hyper != self._hyper:
            self._rec.clear()                 # load_state_dict() / an edited beta: the recorded Adam launch is stale
        rec = self._rec.get(key)
        if rec is None:
            if len(self._rec) >= self.max_recordings:
                self.model.train()
                return self._eager(batch)     # a rare signature beyond the cap stays eager
            rec = self._rec[key] = self._record(batch)
            self._hyper = self._hyper_key()   # (the first recorded step may have created state tensors)
        for k, v in batch.items():
            if torch.is_tensor(v) and v.data_ptr() != rec["static"][k].data_ptr():
                rec["static"][k].copy_(v, non_blocking=True)
        if rec.get("segs"):
            for k, g in enumerate(rec["segs"]):      # segment k's graph, then its slice of the arena to RCCL while the next one runs
                g.replay()
                self.reducer.all_reduce_async(k)
            self.reducer.finish()
            rec["gb"].replay()
        else:
            rec["ga"].replay()
            if rec["gb"] is not None:
                self.reducer.all_reduce()
                rec["gb"].replay()
        self._invalidate()
        return rec["loss"], rec["sep"]

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
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, reducer=None, warmup=3):
        """model: CHORE (or a wrapper with the same call surface) returning (loss, separate losses); optimizer: a torch
        optimiser over its parameters built with capturable=True; reducer: chore_amd.parallel.FlatGradReducer or None."""
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep: build the optimiser with capturable=True (its step counter must live on "
                                 "the device to be recorded)")
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        inner = getattr(model, "module", model)
        if getattr(inner, "losses_on_host", False):
            inner.losses_on_host = False      # the reference returns the separate losses as a CPU tensor: a host copy per step
        self.warmup = int(warmup)
        self.calls = 0
        self._rec = {}            # shape key -> recording
        self._side = None

    # ---- the step, as the eager sequence --------------------------------------------------------------------------------
    def _zero(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def _fwd_bwd(self, batch):
        self._zero()
        loss, sep = self.model(**batch)
        loss.backward()
        if self.reducer is not None:
            self.reducer.gather()
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
        # a process group's watchdog thread polls events while we record: only THIS thread's calls belong to the recording
        mode = dict(capture_error_mode="thread_local") if torch.distributed.is_initialized() else {}
        with torch.cuda.graph(ga, **mode):
            loss, sep = self._fwd_bwd(static)
            if not split:
                self.optimizer.step()
        gb = None
        if split:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool(), **mode):
                self.optimizer.step()
        return dict(static=static, ga=ga, gb=gb, loss=loss.detach(), sep=sep)

    def __call__(self, **batch):
        """one training step on `batch` (keyword arguments of CHORE.forward; tensors on the device).  Returns (loss, separate
        losses) as device tensors that the NEXT call overwrites."""
        self.calls += 1
        if self.calls <= self.warmup:
            self.model.train()
            return self._eager(batch)
        key = self._key(batch)
        rec = self._rec.get(key)
        if rec is None:
            rec = self._rec[key] = self._record(batch)
        for k, v in batch.items():
            if torch.is_tensor(v) and v.data_ptr() != rec["static"][k].data_ptr():
                rec["static"][k].copy_(v, non_blocking=True)
        rec["ga"].replay()
        if rec["gb"] is not None:
            self.reducer.all_reduce()
            rec["gb"].replay()
        self._invalidate()
        return rec["loss"], rec["sep"]

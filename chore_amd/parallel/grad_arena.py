"""Gradient all-reduce of the data-parallel training step through ONE flat arena.

The reference wraps the model in torch's DistributedDataParallel (train_launch.py:30, find_unused_parameters=True):
a hook per parameter copies the gradient into one of ~3 buckets of 25 MB while the backward runs, NCCL all-reduces every
bucket on its own stream and the results are copied back.  On THIS model the backward is a chain of ~1 000 short dependent
launches on two streams, and the reducer's 475 per-parameter hooks, bucket copies, the per-iteration search for unused
parameters and its stream hand-overs sit in that chain: measured on one MI355X with a ONE-rank RCCL group (nothing crosses
xGMI) the DDP-wrapped step takes 29.4 ms against 22.8 ms without the reduction (profiles/r04_ddp_overhead.txt).

`FlatGradReducer` does the same arithmetic -- every gradient replaced by its mean over the ranks, before the optimiser
reads it -- with the collective kept out of the chain:
  * one flat fp32 arena holds the gradients of all parameters (73 MB for CHORE), laid out in REVERSE registration order,
    i.e. roughly the order the backward produces them;
  * after `backward()`, the step's gradient tensors are gathered into the arena by one multi-tensor copy (a handful of
    launches), the arena is all-reduced in `chunks` large pieces (RCCL over xGMI: few, large collectives -- 7 links per GPU
    are used by RCCL's direct algorithms only when a message is large enough to split over them) and scaled by 1 / world;
  * every `p.grad` is then a VIEW into the arena, which is what the optimiser reads (torch's fused Adam takes them as is).
Parameters the loss does not reach (the bn4 affines of ConvBlocks without a downsample branch, model/net_util.py:364-370)
keep a zero gradient in the arena -- DDP's find_unused_parameters=True reduces zeros for them as well.

`mode="inplace"` instead keeps `p.grad` attached to the arena through the backward (autograd then accumulates in place: one
add launch per parameter inside the chain, but no copy afterwards); `mode="copy"` is the default and the measured-faster one.

Round 5 -- the collective UNDER the backward (what the reference's wrap gets from DDP's buckets, train_launch.py:30).  With
`segments=[[params of the part of the network whose gradients are complete first], [... second], ...]` the arena is laid out
segment by segment, and the backward is run one segment at a time (`backward_in_segments`): as soon as a segment's gradients
exist they are gathered (`gather(k)`) and that segment's slice of the arena goes to RCCL (`all_reduce_async(k)`), on RCCL's own
stream, while the next segment's backward runs; `finish()` waits for all of them.  Only the LAST segment's collective (for
CHORE: the 4 MB of the stem, the first three ConvBlocks and the heads, against 13.6 MB per hourglass stack) is exposed.
`collective="rs_ag"` reduces with reduce_scatter + all_gather (each rank reduces 1 / world of a segment and the shards are
gathered: SURVEY 8e's sketch; over 7 xGMI links per GPU both halves are direct sends) instead of one all_reduce per segment --
the same sums; which one is faster at 13.6 MB per segment is for the first multi-GPU run to say.
"""
import os

import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, module, process_group=None, chunks=4, mode="copy", average=True, segments=None, collective="all_reduce"):
        if mode not in ("copy", "inplace"):
            raise ValueError("mode must be 'copy' or 'inplace'")
        if collective not in ("all_reduce", "rs_ag"):
            raise ValueError("collective must be 'all_reduce' or 'rs_ag'")
        self.collective = collective
        if segments is not None:
            if mode != "copy":
                raise ValueError("segments need mode='copy'")
            want = {id(p) for p in module.parameters() if p.requires_grad}
            flat = [p for seg in segments for p in seg]
            if len({id(p) for p in flat}) != len(flat) or {id(p) for p in flat} != want:
                raise ValueError("FlatGradReducer: the segments must hold every trainable parameter exactly once")
            self.params = flat
        else:
            self.params = [p for p in module.parameters() if p.requires_grad][::-1]
        if not self.params:
            raise ValueError("FlatGradReducer: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGradReducer: parameters must share one device and dtype")
        self.group, self.mode, self.average = process_group, mode, average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # 256-byte aligned slots: every view starts on a cache-line boundary, every chunk boundary is an element boundary
        al = 256 // self.params[0].element_size()
        offs, total = [], 0
        seg_bounds = []                     # (first parameter, past the last parameter, arena start, arena end) per segment
        sal = al * max(1, self.world)       # a segment's slice splits evenly over the ranks (reduce_scatter) on 256-byte boundaries
        cuts = set()
        if segments is not None:
            k = 0
            for seg in segments:
                k += len(seg)
                cuts.add(k)
        start_p, start_o = 0, 0
        for i, p in enumerate(self.params):
            offs.append(total)
            total += (p.numel() + al - 1) // al * al
            if i + 1 in cuts:
                total = (total + sal - 1) // sal * sal
                seg_bounds.append((start_p, i + 1, start_o, total))
                start_p, start_o = i + 1, total
        self.arena = torch.zeros(total, dtype=dt, device=dev)
        self.views = [self.arena[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        if segments is not None:
            self.segments = seg_bounds
            self.chunks = [self.arena[a:b] for _, _, a, b in seg_bounds]
        else:
            self.segments = None
            n = max(1, min(int(chunks), total // al))
            step = (total // n + al - 1) // al * al
            self.chunks = [self.arena[i:min(i + step, total)] for i in range(0, total, step)]
        self._works = []
        self._shards = {}
        self.bytes = total * self.params[0].element_size()
        if mode == "inplace":
            self._attach()

    @torch.no_grad()
    def sync_parameters(self, src=0):
        """every rank starts from rank `src`'s parameters (what DistributedDataParallel does when it wraps a module)"""
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src, group=self.group)

    def _attach(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero_grad(self):
        """start of a step (replaces optimizer.zero_grad())"""
        if self.mode == "inplace":
            self.arena.zero_()
            self._attach()
        else:
            for p in self.params:       # set_to_none: the backward's gradient tensors are taken as they come, no add launches
                p.grad = None

    @torch.no_grad()
    def gather(self, segment=None):
        """after backward(): the step's gradient tensors -> the arena; leaves every p.grad a view into it.  No collective --
        this half can live inside a captured hipGraph (chore_amd.parallel.GraphedTrainStep).  segment=k: only that segment's
        parameters (after that segment's part of the backward, see backward_in_segments)"""
        if self.mode == "copy":
            lo, hi = (0, len(self.params)) if segment is None else self.segments[segment][:2]
            ps, vs = self.params[lo:hi], self.views[lo:hi]
            have = [(v, p.grad) for p, v in zip(ps, vs) if p.grad is not None]
            missing = [v for p, v in zip(ps, vs) if p.grad is None]
            if any(g.dtype != v.dtype or g.shape != v.shape for v, g in have):
                raise ValueError("FlatGradReducer: a gradient's dtype / shape differs from its parameter's")
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            if missing:
                torch._foreach_zero_(missing)
            for p, v in zip(ps, vs):
                p.grad = v

    @torch.no_grad()
    def all_reduce(self):
        """the collective half: the arena summed over the ranks in `chunks` pieces, then scaled to the mean"""
        if self.world > 1 or dist.is_initialized():
            if self._stream_ordered():
                # RCCL: the collectives of one communicator run in issue order on its stream -- all issued, then waited for
                works = [dist.all_reduce(c, group=self.group, async_op=True) for c in self.chunks]
                for w in works:
                    w.wait()
            else:
                # gloo (CPU tests, bench.py's one-GPU rehearsal of the N > 1 path): its worker threads run several outstanding
                # collectives side by side, and with device tensors of two ranks on ONE GPU four outstanding all_reduces never
                # finished (the rehearsal hung here, 10 min, both ranks in wait()): one at a time
                for c in self.chunks:
                    dist.all_reduce(c, group=self.group, async_op=True).wait()
            if self.average and self.world > 1:
                self.arena.mul_(1.0 / self.world)

    def _stream_ordered(self):
        """True when the group's collectives on the arena's device are RCCL's (stream-ordered); False for gloo"""
        try:
            b = str(dist.get_backend(self.group))
        except Exception:
            return False
        return self.arena.is_cuda and "nccl" in b

    # ---- the collective under the backward (segments) -------------------------------------------------------------------------
    @torch.no_grad()
    def all_reduce_async(self, segment):
        """segment k's slice of the arena to the collective, asynchronously (RCCL's stream waits for what the current stream has
        issued so far -- the segment's gather -- and the current stream carries on with the next segment's backward)"""
        if not (self.world > 1 or dist.is_initialized()):
            return
        c = self.chunks[segment]
        if self.collective == "rs_ag":
            n = c.numel() // self.world
            shard = self._shards.get(segment)
            if shard is None:
                shard = self._shards[segment] = torch.empty(n, dtype=c.dtype, device=c.device)
            w1 = dist.reduce_scatter_tensor(shard, c, group=self.group, async_op=True)
            self._works.append((w1, segment, shard))
        else:
            self._works.append((dist.all_reduce(c, group=self.group, async_op=True), None, None))
        if not self._stream_ordered():
            self._works[-1][0].wait()       # gloo: one collective outstanding at a time (see all_reduce)

    @torch.no_grad()
    def finish(self):
        """wait for the segments' collectives (rs_ag: gather the reduced shards), then scale to the mean"""
        gathers = []
        for w, segment, shard in self._works:
            w.wait()
            if segment is not None:
                gathers.append(dist.all_gather_into_tensor(self.chunks[segment], shard, group=self.group, async_op=True))
        for w in gathers:
            w.wait()
        self._works = []
        if self.average and self.world > 1:
            self.arena.mul_(1.0 / self.world)

    def reduce(self):
        """after backward(): gather, all-reduce, average; leaves every p.grad a view into the arena"""
        self.gather()
        self.all_reduce()


def backward_in_segments(stack_losses, cuts, seg_params, on_segment=None):
    """The backward of a network that is a chain of K stages, one stage at a time, from the last to the first.
      cuts[i]         the tensor that enters stage i (i = 0 .. K-1): the only thing stage i and everything before it share
      stack_losses[i] a scalar that hangs off stage i (its loss term); their sum is the loss (each gets gradient 1)
      seg_params      K + 1 parameter lists: [K-1], [K-2], ..., [0], then what comes BEFORE cuts[0] ("pre")
    Runs, for i = K-1 .. 0: d(stack_losses[i]) + (the gradient that arrived at cuts[i+1]) -> the parameters of stage i and
    cuts[i]; then cuts[0]'s gradient -> the parameters before it.  Every parameter's gradient becomes its `.grad` (a parameter the
    loss does not reach stays None) and after each segment `on_segment(k)` is called (k counts from 0 = the LAST stage): the
    hook where FlatGradReducer gathers and launches that segment's collective.  The same autograd nodes run in the same order as
    one loss.backward() would run them, so the gradients are the same bit for bit (tests/test_gpu_ddp_nccl.py)."""
    K = len(cuts)
    if len(stack_losses) != K or len(seg_params) != K + 1:
        raise ValueError("backward_in_segments: K cuts, K losses and K + 1 parameter lists")
    g_cut = None
    for k, i in enumerate(range(K - 1, -1, -1)):
        outs, gouts = [stack_losses[i]], [None]
        if i < K - 1:
            outs.append(cuts[i + 1])
            gouts.append(g_cut)
        ps = [p for p in seg_params[k] if p.requires_grad]
        grads = torch.autograd.grad(outs, ps + [cuts[i]], gouts, allow_unused=True)
        for p, g in zip(ps, grads[:-1]):
            p.grad = g
        g_cut = grads[-1]
        if on_segment is not None:
            on_segment(k)
    ps = [p for p in seg_params[K] if p.requires_grad]
    if ps:
        grads = torch.autograd.grad([cuts[0]], ps, [g_cut], allow_unused=True)
        for p, g in zip(ps, grads):
            p.grad = g
    if on_segment is not None:
        on_segment(K)


def chore_segments(model):
    """CHORE's trainable parameters in the order their gradients become complete in the backward: hourglass stack K-1 (m, top_m,
    conv_last, bn_end, l, bl, al of that stack), ..., stack 0 -- with it the four heads, whose gradients sum over all stacks'
    queries (one shared arena, model/chore.py _QueryTrainFn) and are handed to autograd by the query backward that runs last,
    stack 0's -- then what comes before the first stack (the stem, conv2-4; 3.6 MB: the only collective left exposed).
    For FlatGradReducer(segments=...) and backward_in_segments."""
    enc = model.image_filter
    K = enc.num_modules
    owner = {}
    for i in range(K):
        for name in (f"m{i}", f"top_m_{i}", f"conv_last{i}", f"bn_end{i}", f"l{i}", f"bl{i}", f"al{i}"):
            mod = getattr(enc, name, None)
            if mod is not None:
                for p in mod.parameters():
                    owner.setdefault(id(p), i)
    for name in ("df", "part_predictor", "pca_predictor", "center_predictor"):
        for p in getattr(model, name).parameters():
            owner.setdefault(id(p), 0)
    segs = [[] for _ in range(K + 1)]
    seen = set()
    for p in model.parameters():
        if not p.requires_grad or id(p) in seen:
            continue
        seen.add(id(p))
        i = owner.get(id(p))
        segs[K if i is None else K - 1 - i].append(p)
    return segs

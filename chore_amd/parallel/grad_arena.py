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
"""
import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, module, process_group=None, chunks=4, mode="copy", average=True):
        if mode not in ("copy", "inplace"):
            raise ValueError("mode must be 'copy' or 'inplace'")
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
        for p in self.params:
            offs.append(total)
            total += (p.numel() + al - 1) // al * al
        self.arena = torch.zeros(total, dtype=dt, device=dev)
        self.views = [self.arena[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        n = max(1, min(int(chunks), total // al))
        step = (total // n + al - 1) // al * al
        self.chunks = [self.arena[i:min(i + step, total)] for i in range(0, total, step)]
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
    def gather(self):
        """after backward(): the step's gradient tensors -> the arena; leaves every p.grad a view into it.  No collective --
        this half can live inside a captured hipGraph (chore_amd.parallel.GraphedTrainStep)"""
        if self.mode == "copy":
            have = [(v, p.grad) for p, v in zip(self.params, self.views) if p.grad is not None]
            missing = [v for p, v in zip(self.params, self.views) if p.grad is None]
            if any(g.dtype != v.dtype or g.shape != v.shape for v, g in have):
                raise ValueError("FlatGradReducer: a gradient's dtype / shape differs from its parameter's")
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            if missing:
                torch._foreach_zero_(missing)
            self._attach()

    @torch.no_grad()
    def all_reduce(self):
        """the collective half: the arena summed over the ranks in `chunks` pieces, then scaled to the mean"""
        if self.world > 1 or dist.is_initialized():
            works = [dist.all_reduce(c, group=self.group, async_op=True) for c in self.chunks]
            for w in works:
                w.wait()
            if self.average and self.world > 1:
                self.arena.mul_(1.0 / self.world)

    def reduce(self):
        """after backward(): gather, all-reduce, average; leaves every p.grad a view into the arena"""
        self.gather()
        self.all_reduce()

"""SMPL-H + object fitting loops (BEHAVE protocol) on the GPU.

Counterpart of /root/reference/recon/recon_fit_behave.py: `optimize_smpl` (:224-291), `forward_smpl`
(:293-337), `optimize_smpl_object` (:90-163), `forward_step` (:165-222), `get_loss_weights` (:339-358).
The schedules, optimiser re-creation points, the weight formula w*L/(1+decay) and the reference's
quirks are kept on purpose because they change the fitted poses:
  * zero_grad() once per OUTER iteration, then `steps_per_iter` x {backward; step} -> gradients accumulate
    over the inner steps;
  * a new Adam at every phase switch; in `optimize_smpl_object` the SMPL parameters are never stepped;
Work the reference repeats with the same result every time is done once (same values, fewer launches): `forward_step`
queries the object points twice per step (:171 and in compute_obj_loss) -- one query here, its gradient is the sum of both
uses; in `optimize_smpl_object` the body is fixed, so LBS and the 6 890-point query of the joint phase run once per call
instead of once per step; the 'sil' phase, whose terms do not read the field, does not query it.
The 'sil' phase runs if the caller provides data_dict['silhouette'] (SilLossROI); the 'collide' term of the joint
phase (recon_fit_behave.py:213-216) runs if the fitter was given the object template mesh (scan_verts / scan_faces).
Per-step host synchronisation of the reference (tqdm strings, .item()) is gone: the early-stop rule is evaluated on the
device after every inner step and the host reads the latched flag once per OUTER iteration (graph_step.py).
`fit_recon` (:29-76) chains the stages; `recon_fit(args)` (:361-365) is the command-line entry.
"""
import os

import torch
import torch.nn.functional as F

from ..lib_smpl.const import SMPL_POSE_PRAMS_NUM
from . import fit_terms
from .graph_step import EagerStep, FusedAdam, GraphedStep
from .recon_fit_base import _BATCH_RNG, RECON_PATH, ReconFitterBase, cpu_generator  # noqa: F401  (recon_fit_coco.py:15 imports RECON_PATH from here)



class _CaptureGate:
    """Mutual exclusion between a hipGraph RECORDING and everything else the fitter's host threads do (fit_recon with
    pipeline="chains": two chains issued by two threads).  A capture in the global mode does not tolerate another thread's
    allocations, synchronisations or event queries; a thread that is about to record therefore waits until every other
    participant has parked at a checkpoint (the top of an outer iteration) or left, records alone, and lets them go on.
    Kernels the others have already issued keep running: only their HOST threads pause."""

    def __init__(self):
        import threading
        self.cv = threading.Condition()
        self.active = self.parked = 0
        self.recording = False

    def enter(self):
        with self.cv:
            while self.recording:
                self.cv.wait()
            self.active += 1

    def leave(self):
        with self.cv:
            self.active -= 1
            self.cv.notify_all()

    def checkpoint(self):
        if not self.recording:              # (unlocked read: a recorder that raises the flag just now is seen at the next checkpoint)
            return
        with self.cv:
            self.parked += 1
            self.cv.notify_all()
            while self.recording:
                self.cv.wait()
            self.parked -= 1

    def exclusive(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            with self.cv:
                while self.recording:       # another participant records: park like at a checkpoint, then take the turn
                    self.parked += 1
                    self.cv.notify_all()
                    while self.recording:
                        self.cv.wait()
                    self.parked -= 1
                self.recording = True
                while self.parked < self.active - 1:
                    self.cv.wait()
            try:
                yield
            finally:
                with self.cv:
                    self.recording = False
                    self.cv.notify_all()
        return cm()


class _FitSlot:
    """What the recorded inner steps of one driver call read and write, kept for the next call of the same shapes
    (ReconFitterBehave.reuse_graphs): recording a step costs ~6.6 ms of host time and a frame's chain records six of them
    (40 of its 135 ms, profiles/r04_fit_chain.txt).  A hipGraph bakes in the ADDRESSES of everything a step touches, so the
    slot owns private, persistent tensors -- the parameters being fitted, their gradients and optimiser state, the flags, and
    every input the loss terms read -- and a later call copies its values INTO them (`bind`) instead of recording again;
    the fitted values are copied back out into the caller's tensors at the end."""

    def __init__(self):
        self.tensors, self.steppers, self.objects = {}, {}, {}

    def bind(self, name, value, leaf=False):
        t = self.tensors.get(name)
        if t is None:
            t = value.detach().clone()
            if leaf:
                t.requires_grad_(True)
            self.tensors[name] = t
        else:
            if t.shape != value.shape or t.dtype != value.dtype or t.device != value.device:
                raise ValueError("fit slot: '%s' changed shape / type between calls" % name)
            with torch.no_grad():
                t.copy_(value)
        return t

    def obj(self, name, factory):
        if name not in self.objects:
            self.objects[name] = factory()
        return self.objects[name]


class ReconFitterBehave(ReconFitterBase):
    use_graphs = False    # True: every inner step is a hipGraph replay (graph_step.py); same update rule
    reuse_graphs = False  # with use_graphs: keep the recorded steps across calls of the same shapes (_FitSlot)
    early_stop = True     # False: never arm the stop rule (benchmarks that time a fixed number of iterations)
    timer = None          # a list: per outer iteration (start event, end event, number of inner steps, phase name) is appended
    adam_capturable = False   # eager steps with Adam's scalars evaluated on the device (what the graph does)

    fuse_stop_rule = True      # the steps' loss is sum_dict's result as it comes: the stop rule rides in that launch (graph_step)

    def _stepper(self, *a, **k):
        k.setdefault("fuse_rule", self.fuse_stop_rule and not os.environ.get("CHORE_FIT_SPLIT_RULE"))
        if self.use_graphs:
            gate = self.__dict__.get("_gate")
            if gate is None:
                return GraphedStep(*a, **k)
            with gate.exclusive():             # two chains side by side (_fit_concurrent): a recording runs alone
                st = GraphedStep(*a, **k)
                torch.cuda.current_stream(torch.device(self.device)).synchronize()      # the recording's own work is done before the others resume
                return st
        return EagerStep(*a, capturable=self.adam_capturable, **k)

    def _slot(self, *key):
        """the kept state of a driver call with these shapes / schedule, or None when steps are not kept"""
        if not (self.use_graphs and self.reuse_graphs):
            return None
        slots = self.__dict__.setdefault("_slots", {})
        if key not in slots:
            if len(slots) >= 16:         # shapes keep changing: do not hoard graphs
                slots.clear()
            slots[key] = _FitSlot()
        return slots[key]

    @staticmethod
    def _maps_key(model):
        """identity of the field the kept steps were recorded on: the network object AND the addresses of its feature maps (a
        recorded query reads the maps through the pointers it was recorded with; HGFilter.static_outputs keeps them fixed
        across filter() calls -- without it every batch gets new maps, hence a new slot, and nothing stale is ever replayed)"""
        feats = getattr(model, "im_feat_list", None)
        tm = getattr(model, "tmpx", None)
        return (id(model), feats[-1].data_ptr() if feats else 0, tm.data_ptr() if torch.is_tensor(tm) else 0)

    def _kept(self, slot, name, factory, reset_opt=True):
        """the stepper of phase `name`: built by `factory`, or the one kept from an earlier call, reset"""
        if slot is None:
            return factory()
        st = slot.steppers.get(name)
        if st is not None:
            st.reset(reset_opt)
            return st
        st = factory()
        if isinstance(st, GraphedStep) and isinstance(st.opt, FusedAdam):
            slot.steppers[name] = st
        return st

    def _inner(self, st, n, phase=None):
        """the inner steps of one outer iteration"""
        gate = self.__dict__.get("_gate")
        if gate is not None:
            gate.checkpoint()
        if self.timer is None:
            for _ in range(n):
                st.step()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            st.step()
        e1.record()
        self.timer.append((e0, e1, n, phase))

    @staticmethod
    def release_graphs(split, model):
        """forget what holds autograd graphs of earlier steps (see GraphedStep)"""
        def f():
            split.betas = split.pose = None
            split.forget()
            model.preds = None
            model.points = None            # CHORE.query keeps its last inputs (reference attribute): a graph too
            model.intermediate_preds_list = []
        return f

    def get_loss_weights(self):
        w = {"beta": 1.0, "pose": 1e-5, "hand": 1e-5, "j2d": 0.3 ** 2, "object": 30.0 ** 2, "part": 0.05 ** 2,
             "contact": 30.0 ** 2, "scale": 10.0 ** 2, "df_h": 30.0 ** 2, "smplz": 30 ** 2, "mask": 0.003 ** 2,
             "ocent": 15 ** 2, "collide": 3 ** 2, "pinit": 5 ** 2, "rot": 10.0 ** 2, "trans": 10.0 ** 2}
        return {k: (lambda cst, it, c=c: c * cst / (1 + it)) for k, c in w.items()}

    # ---- the whole chain ------------------------------------------------------------------------------
    def fit_recon(self, args, loader=None, model=None, generator=None, save=True, pipeline=None):
        """[recon_fit_behave.py:29-76] for every batch of the loader: dense point clouds from the two UDFs -> SMPL-H
        initialisation -> optimize_smpl -> object initialisation -> optimize_smpl_object -> results on disk.
        `loader` / `model` / `generator` default to what the reference builds from `args` (TestData loader of the
        sequence, CHORE(args), Generator with the experiment's checkpoint).  Under torch.distributed every rank takes the
        batches rank::world_size (frames are independent: BASELINE configs[4]); returns this rank's fitted parameters.

        pipeline (default: the fitter's `pipeline` attribute; round 5): the reference's loop is strictly serial per batch, and a
        batch's chain starts with the encoder + the point clouds (a quarter of its time) before the first Adam step.  Pipelined,
        batch k+1's encode + point clouds + SMPL-H initialisation run on a second stream, issued by a second host thread, while
        batch k is being optimised (`_fit_pipelined`) -- same results bit for bit as the serial loop with the same `batch_seed`
        (tests/test_gpu_fit_chain.py), since every batch then draws its point-cloud random numbers from its own generators."""
        from ..model import CHORE
        from ..parallel.frame_shard import shard_indices
        from .generator import Generator
        loader = self.init_dataloader(args) if loader is None else loader
        if generator is None:
            model = CHORE(args) if model is None else model
            generator = Generator(model, getattr(args, "exp_name", None), threshold=2.0,
                                  sparse_thres=getattr(args, "sparse_thres", 0.03),
                                  filter_val=getattr(args, "filter_val", 0.004), device=self.device,
                                  checkpoint=getattr(args, "checkpoint", None))
        batches = list(loader) if not hasattr(loader, "__len__") else loader
        mine = set(shard_indices(len(batches)))
        todo = []
        for i, data in enumerate(batches):
            if i not in mine:
                continue
            if save and self.outpath is not None and not getattr(args, "redo", False) and \
                    self.is_done(data["path"], args.save_name, args.test_kid):
                print(data["path"], args.save_name, "already done, skipped")
                continue
            todo.append((i, data))
        results = []

        def finish(i, data, fitted):
            smpl, obj_R, obj_t, obj_s = fitted
            if save and self.outpath is not None:
                self.save_outputs(smpl, obj_R, obj_t, data["path"], args.save_name, args.test_kid, obj_s)
            results.append(dict(index=i, pose=smpl.pose.detach(), betas=smpl.betas.detach(), trans=smpl.trans.detach(),
                                obj_R=self.decopose_axis(obj_R, no_rand=True).detach(), obj_t=obj_t.detach(),
                                obj_s=obj_s.detach()))
        use_pipe = self.pipeline if pipeline is None else pipeline
        if use_pipe and len(todo) > 1 and torch.device(self.device).type == "cuda":
            import sys
            sw = os.environ.get("CHORE_PIPE_SWITCH_US")
            old_sw = sys.getswitchinterval()
            if sw:
                sys.setswitchinterval(float(sw) * 1e-6)
            try:
                if self.batch_seed is None:            # two host threads cannot share the process-wide random streams
                    self._pipe_seed = int(torch.initial_seed() % (1 << 31))
                if os.environ.get("CHORE_PIPE_OWN_STREAM"):
                    dev = torch.device(self.device)
                    cur = torch.cuda.current_stream(dev)
                    st = self.__dict__.get("_pipe_main")
                    if st is None:
                        st = self._pipe_main = torch.cuda.Stream(dev)
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        self._fit_pipelined(todo, generator, finish)
                    cur.wait_stream(st)
                elif use_pipe == "chains":
                    self._fit_concurrent(todo, generator, finish)
                else:
                    self._fit_pipelined(todo, generator, finish)
            finally:
                self._pipe_seed = None
                sys.setswitchinterval(old_sw)
        else:
            for i, data in todo:
                # (index only when per-batch generators are on: a subclass's fit_batch(data, generator) keeps working)
                fitted = self.fit_batch(data, generator) if self.batch_seed is None else self.fit_batch(data, generator, index=i)
                if self.batch_ends is not None and torch.device(self.device).type == "cuda":
                    self._mark_read([None], 0, torch.cuda.current_stream(torch.device(self.device)))
                finish(i, data, fitted)
        return results

    # Every batch's point-cloud random numbers from generators of its own (seed = batch_seed + the batch's index in the loader):
    # None = the process-wide streams, like the reference (serial loop only -- two threads drawing from one stream have no order).
    batch_seed = None
    pipeline = False      # fit_recon: True = prepare batch k+1 on a second stream / host thread while batch k is optimised;
                          # "chains" = the whole chains of `chains` batches side by side (_fit_concurrent)
    chains = 3            # (one frame per batch: 2 -> 60 ms per frame, 3 -> 49, 4 -> 49: the host threads are the limit from three on)
    smpl_iters = None     # fit_recon / fit_batch: keyword arguments of optimize_smpl / optimize_smpl_object other than the
    object_iters = None   # reference's (benchmarks and tests with shorter schedules)

    _pipe_seed = None     # the per-batch seed of a pipelined fit_recon call when `batch_seed` is None (from torch.initial_seed(),
                          # for that call only: a later serial call draws from the process-wide streams again, like the reference)

    def _seed(self):
        return self.batch_seed if self.batch_seed is not None else self._pipe_seed

    def _batch_generators(self, index):
        seed = self._seed()
        if seed is None or index is None:
            return None
        dev = torch.device(self.device)
        cpu_g, dev_g = torch.Generator(), torch.Generator(device=dev)
        cpu_g.manual_seed(int(seed) + 2 * int(index))
        dev_g.manual_seed(int(seed) + 2 * int(index) + 1)
        return cpu_g, dev_g

    def prepare_batch(self, data, generator, index=None):
        """the first half of a batch's chain (fit_recon :46-57): encoder, dense point clouds, SMPL-H initialisation -- everything
        before the first optimiser step"""
        if self.use_graphs and self.reuse_graphs and hasattr(generator.model, "image_filter"):
            generator.model.image_filter.static_outputs = True      # the kept steps read the maps through fixed addresses
        gens = self._batch_generators(index)
        kw = {} if gens is None else {"generators": gens}      # (the reference's signature when the global streams are used)
        pc_generated = generator.generate_pclouds_batch(data, num_points=5000, num_steps=10, mute=True, **kw)
        opt_gen = None
        if gens is not None:                   # the optimisation's CPU draws (SO(3) perturbations): a third stream of the batch's own
            opt_gen = torch.Generator()
            opt_gen.manual_seed(int(self._seed()) + 2 * int(index) + 1000003)
        return dict(data=data, pc=pc_generated, model=generator.model, smplfit=self.prep_smplfit(data, generator, pc_generated),
                    opt_gen=opt_gen)

    def optimise_batch(self, prep, smpl_iters=None, object_iters=None):
        """the second half (:58-74): optimize_smpl -> object initialisation -> optimize_smpl_object"""
        _BATCH_RNG.gen = prep.get("opt_gen")
        try:
            return self._optimise_batch(prep, smpl_iters, object_iters)
        finally:
            _BATCH_RNG.gen = None

    def _optimise_batch(self, prep, smpl_iters, object_iters):
        data, pc_generated = prep["data"], prep["pc"]
        batch_size = data["images"].shape[0]
        (betas_dict, body_kpts, human_parts, human_points, human_t, obj_points, part_colors, part_labels, query_dict,
         smpl) = prep["smplfit"]
        smpl, scale = self.optimize_smpl(smpl, betas_dict, **(smpl_iters or self.smpl_iters or dict(iter_for_kpts=1, iter_for_pose=1,
                                                                                                    iter_for_betas=1)))
        obj_R, obj_s, obj_t, object_init = self.init_obj_fit_data(batch_size, human_t, pc_generated, scale)
        data_dict = {"obj_R": obj_R, "obj_t": obj_t, "obj_s": obj_s, "objects": object_init, "smpl": smpl,
                     "images": data.get("images").to(self.device), "human_init": human_points, "obj_init": obj_points,
                     "human_parts": human_parts, "part_labels": part_labels, "part_colors": part_colors,
                     "body_kpts": body_kpts, "query_dict": query_dict, "obj_t_init": obj_t.clone().detach().to(self.device)}
        smpl, obj_R, obj_t = self.optimize_smpl_object(prep["model"], data_dict, **(object_iters or self.object_iters or {}))
        return smpl, obj_R, obj_t, obj_s

    def fit_batch(self, data, generator, smpl_iters=None, object_iters=None, index=None):
        """one batch through the chain of fit_recon (:46-74); the iteration counts are the reference's"""
        return self.optimise_batch(self.prepare_batch(data, generator, index), smpl_iters, object_iters)

    def _fit_pipelined(self, todo, generator, finish, smpl_iters=None, object_iters=None):
        """Batch k+1 is PREPARED (encoder + point clouds + SMPL-H initialisation) while batch k is OPTIMISED.
        Two slots, alternating: each has its own stream, its own view of the field network (a shallow copy of the CHORE object:
        the same parameters and packed weights, its own `im_feat_list / tmpx / preds`) and therefore its own feature maps
        (HGFilter.static_outputs is keyed by stream) and its own kept recordings (_FitSlot is keyed by the maps' addresses).
        The preparation is issued by a worker thread (it waits for the device several times per point-cloud round; ctypes and
        torch release the GIL while they wait or launch), the optimisation by the calling thread on its current stream.
        Ordering on the device: a slot's preparation waits for the optimisation that last read the slot's maps; an optimisation
        waits for its batch's preparation.  While a slot's inner steps are being RECORDED (its first batch) nothing else is
        issued: a capture does not tolerate another thread's allocations."""
        import copy
        from concurrent.futures import ThreadPoolExecutor
        if self._seed() is None:
            self._pipe_seed = int(torch.initial_seed() % (1 << 31))
        dev = torch.device(self.device)
        main = torch.cuda.current_stream(dev)
        state = self.__dict__.get("_pipe_state")
        if state is None or state[0] is not generator or state[1][0] is not generator.model:
            # kept across calls: the second view's identity is part of the kept recordings' key (_maps_key)
            nets = [generator.model, copy.copy(generator.model)]
            gens = [generator, copy.copy(generator)]
            gens[1].model = nets[1]
            prio = int(os.environ.get("CHORE_PIPE_PRIO", "0"))
            # ONE worker thread for the fitter's lifetime (it owns a C handle of its own, chore_amd/_lib.py: a thread per call would
            # leave a handle behind per call)
            # CHORE_PIPE_CUS=n: the preparation streams run on n of the compute units (the same share of every XCD), the rest stay
            # free for the optimisation's small kernels whatever the preparation has in flight (chore_stream_create_cu_mask)
            cus = int(os.environ.get("CHORE_PIPE_CUS", "0"))
            if cus > 0:
                from chore_amd import _lib
                prep_streams = [_lib.cu_masked_stream(dev.index or 0, cus) for _ in range(2)]
            else:
                prep_streams = [torch.cuda.Stream(dev, priority=prio), torch.cuda.Stream(dev, priority=prio)]
            state = self._pipe_state = (generator, nets, gens, prep_streams,
                                        [False, False], ThreadPoolExecutor(max_workers=1, thread_name_prefix="chore-prep"))
        _, nets, gens, streams, warm, pool = state      # warm[s]: slot s's inner steps are recorded (kept across calls with the slots)
        if not self.reuse_graphs:
            warm[0] = warm[1] = False
        last_read = [None, None]        # event: the optimisation that last used the slot has been issued AND executed up to here

        import time
        dbg = [] if os.environ.get("CHORE_PIPE_DEBUG") else None
        t_base = time.perf_counter()

        # A recording the `warm` bookkeeping does not foresee (a new shape later in the loader, recordings dropped at the slot cap,
        # maps at new addresses) must still run alone: the worker holds the gate while it prepares, the calling thread records
        # through _stepper's gate.exclusive(), which waits until the worker has left and keeps it out until the recording is done.
        gate = self._gate = _CaptureGate()

        def prepare(k, ready):
            gate.enter()
            try:
                return prepare_(k, ready)
            finally:
                gate.leave()

        def prepare_(k, ready):
            i, data = todo[k]
            s = k % 2
            torch.cuda.set_device(dev)
            th0 = time.perf_counter()
            with torch.cuda.stream(streams[s]):
                streams[s].wait_event(ready)                       # the loader's tensors, issued on the caller's stream
                if last_read[s] is not None:
                    streams[s].wait_event(last_read[s])
                e0 = torch.cuda.Event(enable_timing=True) if dbg is not None else None
                if e0 is not None:
                    e0.record(streams[s])
                prep = self.prepare_batch(data, gens[s], index=i)
                done = torch.cuda.Event(enable_timing=dbg is not None)
                done.record(streams[s])
            if dbg is not None:
                dbg.append(("prep", k, th0 - t_base, time.perf_counter() - t_base, e0, done))
            return prep, done

        def submit(pool, k):
            ready = torch.cuda.Event()
            ready.record(main)
            return pool.submit(prepare, k, ready)

        gate.enter()
        try:
            self._fit_pipelined_loop(todo, submit, pool, warm, main, last_read, finish, smpl_iters, object_iters, dbg, t_base)
        finally:
            gate.leave()
            self._gate = None
        if dbg:
            torch.cuda.synchronize()
            ref = [d for d in dbg if d[0] == "opt"][0][4]
            for kind, k, h0, h1, e0, e1 in sorted(dbg, key=lambda d: d[2]):
                print("[pipe] %-4s batch %d  host %7.1f .. %7.1f ms   device %7.1f .. %7.1f ms (relative to the first timed optimisation's start)"
                      % (kind, k, h0 * 1e3, h1 * 1e3, ref.elapsed_time(e0), ref.elapsed_time(e1)), file=__import__("sys").stderr)

    def _fit_pipelined_loop(self, todo, submit, pool, warm, main, last_read, finish, smpl_iters, object_iters, dbg, t_base):
        import time
        fut = submit(pool, 0)
        for k, (i, data) in enumerate(todo):
            prep, done = fut.result()
            s = k % 2
            fut = None
            if k + 1 < len(todo):
                if not warm[s] and self.use_graphs:            # this optimisation records: nothing beside it
                    main.wait_event(done)
                    fitted = self.optimise_batch(prep, smpl_iters, object_iters)
                    self._mark_read(last_read, s, main)
                    warm[s] = True
                    fut = submit(pool, k + 1)
                    finish(i, data, fitted)
                    continue
                fut = submit(pool, k + 1)
            th0 = time.perf_counter()
            main.wait_event(done)
            if dbg is not None:
                o0 = torch.cuda.Event(enable_timing=True)
                o0.record(main)
            fitted = self.optimise_batch(prep, smpl_iters, object_iters)
            self._mark_read(last_read, s, main)
            if dbg is not None:
                o1 = torch.cuda.Event(enable_timing=True)
                o1.record(main)
                dbg.append(("opt", k, th0 - t_base, time.perf_counter() - t_base, o0, o1))
            warm[s] = True
            finish(i, data, fitted)

    def _fit_concurrent(self, todo, generator, finish, smpl_iters=None, object_iters=None):
        """pipeline="chains": the WHOLE chains of `self.chains` (3) batches side by side, each on its slot's stream, issued by its
        slot's host thread (slots as in _fit_pipelined: a view of the network, its maps and kept recordings each).  The optimisation of one
        frame is a chain of ~10 000 small dependent launches that leaves most of the chip idle (0.25 ms per iteration at one
        frame, 0.073 per frame in a batch of eight): a second chain beside it costs little.  Results per batch do not depend on
        what runs beside it (own generators, own slot, deterministic kernels): equal to the serial loop bit for bit, like
        _fit_pipelined.  Slot s takes the batches s, s + chains, ... in order.  A chain that has to RECORD inner steps (a slot's first
        batch, a new shape later in the loader, recordings dropped by the slot cap) does so alone: the other host thread parks
        at its next outer iteration, the caller's thread does not run `finish` meanwhile (_CaptureGate) -- a capture does not
        tolerate another thread's allocations or synchronisations.  `finish` is called by the calling thread in loader order."""
        import copy
        from concurrent.futures import ThreadPoolExecutor
        if self._seed() is None:
            self._pipe_seed = int(torch.initial_seed() % (1 << 31))
        dev = torch.device(self.device)
        main = torch.cuda.current_stream(dev)
        nch = max(2, int(os.environ.get("CHORE_FIT_CHAINS", self.chains)))
        state = self.__dict__.get("_chain_state")
        if state is None or state["generator"] is not generator or state["nets"][0] is not generator.model or len(state["nets"]) != nch:
            nets = [generator.model] + [copy.copy(generator.model) for _ in range(nch - 1)]
            gens = [generator] + [copy.copy(generator) for _ in range(nch - 1)]
            for g_, n_ in zip(gens[1:], nets[1:]):
                g_.model = n_
            state = self._chain_state = dict(generator=generator, nets=nets, gens=gens, streams=[torch.cuda.Stream(dev) for _ in range(nch)],
                                             pools=[ThreadPoolExecutor(max_workers=1, thread_name_prefix="chore-chain%d" % k) for k in range(nch)])
        gens, streams, pools = state["gens"], state["streams"], state["pools"]
        timing = self.batch_ends is not None
        gate = self._gate = _CaptureGate()     # recordings (a slot's first batch, a new shape later on) run alone: see _stepper / _inner

        def chain(k, ready):
            i, data = todo[k]
            s = k % nch
            torch.cuda.set_device(dev)
            gate.enter()
            try:
                with torch.cuda.stream(streams[s]):
                    streams[s].wait_event(ready)                   # the loader's tensors, issued on the caller's stream
                    fitted = self.optimise_batch(self.prepare_batch(data, gens[s], index=i), smpl_iters, object_iters)
                    done = torch.cuda.Event(enable_timing=timing)
                    done.record(streams[s])
            finally:
                gate.leave()
            return fitted, done

        try:
            futs = []
            for k in range(len(todo)):                             # each slot's executor runs its batches in order
                ready = torch.cuda.Event()
                ready.record(main)
                futs.append(pools[k % nch].submit(chain, k, ready))
            for k, fut in enumerate(futs):
                fitted, done = fut.result()
                gate.enter()                                       # `finish` synchronises (results to the host): not during a recording
                try:
                    main.wait_event(done)
                    for t in fitted:                               # made on the slot's stream, read on the caller's from here on
                        for u in ((t.pose, t.betas, t.trans) if hasattr(t, "pose") else (t,)):
                            if torch.is_tensor(u) and u.is_cuda:
                                u.record_stream(main)
                    if timing:
                        self.batch_ends.append(done)
                    finish(todo[k][0], todo[k][1], fitted)
                finally:
                    gate.leave()
        finally:
            for fut in futs:                                       # (an exception above: let the chains end before the gate goes)
                try:
                    fut.result()
                except Exception:
                    pass
            self._gate = None

    batch_ends = None     # a list: one timing event per finished batch of fit_recon (serial or pipelined) is appended

    def _mark_read(self, last_read, s, stream):
        ev = torch.cuda.Event(enable_timing=self.batch_ends is not None)
        ev.record(stream)
        last_read[s] = ev
        if self.batch_ends is not None:
            self.batch_ends.append(ev)

    def init_dataloader(self, args):
        """[recon_fit_behave.py:78-88] the image loader is the reference's own host-side code (data/test_data.py, cv2
        JPEG decoding -- out of scope of the device path): it is imported from the reference tree when this module
        runs inside one (python -m chore_amd.dropin recon/recon_fit_behave.py ...), otherwise pass `loader=`"""
        try:
            from data.data_paths import DataPaths
            from data.test_data import TestData
        except ImportError as e:
            raise RuntimeError("fit_recon needs a loader: run from the reference tree (its data/ package provides "
                               "TestData) or pass loader= with batches of the data/test_data.py contract") from e
        image_files = DataPaths.get_image_paths_seq(self.seq_folder, check_occlusion=False)
        batch_end = args.end if args.end is not None else len(image_files)
        image_files = image_files[args.start:batch_end]
        dataset = TestData(image_files, args.batch_size, args.batch_size, image_size=args.net_img_size,
                           crop_size=args.loadSize)
        print(f"In total {len(image_files)} test examples")
        return dataset.get_loader(shuffle=False)

    def _stock_terms(self, *names):
        """the fused operators restate these methods of ReconFitterBase: a subclass that overrides one of them gets its own
        formulation (the tensor-expression path), not the operator"""
        return all(getattr(type(self), n) is getattr(ReconFitterBase, n) for n in names)

    # ---- SMPL ---------------------------------------------------------------------------------------
    def forward_smpl(self, smpl, data_dict, phase):
        loss_dict = {}
        model = data_dict["net"]
        smpl.forget()   # the LBS memo lives for one step (its autograd graph is consumed by this step's backward)
        smpl_verts = smpl()[0]
        pose = smpl.pose
        if self._stock_terms("compute_df_h_loss", "compute_prior_loss", "smplz_loss", "compute_kpts_loss", "projection_loss",
                             "project_points") and fit_terms.smpl_terms_supported(pose, self.body_prior, self.hand_prior):
            # the same seven terms from two operators (fit_terms.py): 6 launches each way instead of ~45 / ~85
            model.query(smpl_verts, **data_dict["query_dict"])
            df_pred, _, parts_pred, _ = model.get_preds()
            if fit_terms.point_terms_supported(df_pred, parts_pred):
                qd = data_dict["query_dict"]
                df_h, part = fit_terms.point_terms(df_pred, 0, 0.1, parts_pred, data_dict["part_labels"])
                p_pose, p_hand, pinit, smplz, j2d = fit_terms.smpl_terms(
                    pose, smpl.landmarks_all(), data_dict["pose_init"], data_dict["body_kpts"] if phase == "kpts" else None,
                    qd["crop_center"], self.body_prior, self.hand_prior, self.camera, self.net_in_size, self.z_0)
                loss_dict.update(df_h=df_h, pose=p_pose, hand=p_hand, part=part, smplz=smplz, pinit=pinit)
                if phase == "kpts":
                    loss_dict["j2d"] = j2d
                return loss_dict
        _, parts_pred, _ = self.compute_df_h_loss(data_dict, loss_dict, model, smpl_verts)
        self.compute_prior_loss(loss_dict, smpl, nobeta=True)
        loss_dict["part"] = F.cross_entropy(parts_pred, data_dict["part_labels"], reduction="none").sum(-1).mean()
        J, _, _ = smpl.get_landmarks()
        self.smplz_loss(J, loss_dict)
        loss_dict["pinit"] = torch.mean(torch.sum((smpl.pose[:, 3:SMPL_POSE_PRAMS_NUM] - data_dict["pose_init"]) ** 2, -1))
        if phase == "kpts":
            self.compute_kpts_loss(data_dict, loss_dict, smpl)
        return loss_dict

    def optimize_smpl(self, smpl, data_dict, iter_for_betas=10, iter_for_pose=10, iter_for_kpts=5, steps_per_iter=10,
                      max_iter=150):
        """the reference's driver (see _optimize_smpl).  With reuse_graphs the fit runs on the slot's private copy of the body and
        of the inputs (whose addresses the kept hipGraphs read); the fitted parameters are copied into `smpl`, which is
        returned like the reference returns it."""
        slot = None
        if smpl.pose.is_cuda:
            slot = self._slot("smpl", tuple(smpl.pose.shape), str(smpl.pose.device), self._maps_key(data_dict["net"]), iter_for_betas,
                              iter_for_pose, iter_for_kpts, steps_per_iter, max_iter)
        if slot is None:
            return self._optimize_smpl(smpl, data_dict, None, iter_for_betas, iter_for_pose, iter_for_kpts, steps_per_iter, max_iter)
        from ..lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatch
        B = smpl.pose.shape[0]
        priv = slot.obj("smpl", lambda: SMPLPyTorchWrapperBatch(
            smpl.smpl, B, betas=smpl.betas.detach().clone(), pose=smpl.pose.detach().clone(), trans=smpl.trans.detach().clone(),
            offsets=smpl.offsets.detach().clone(), faces=smpl.faces, gender=smpl.gender, hands=smpl.hands,
            num_betas=smpl.betas.shape[1], regressors=(smpl.body25_reg, smpl.face_reg, smpl.hand_reg)).to(smpl.pose.device))
        with torch.no_grad():
            for name in ("pose", "betas", "trans", "offsets"):
                getattr(priv, name).copy_(getattr(smpl, name))
        priv.forget()
        dd = slot.obj("data", dict)
        dd["net"] = data_dict["net"]
        dd["query_dict"] = {"crop_center": slot.bind("crop_center", data_dict["query_dict"]["crop_center"])}
        for k in ("part_labels", "pose_init", "body_kpts"):
            dd[k] = slot.bind(k, data_dict[k])
        _, scale = self._optimize_smpl(priv, dd, slot, iter_for_betas, iter_for_pose, iter_for_kpts, steps_per_iter, max_iter)
        with torch.no_grad():
            for name in ("pose", "betas", "trans"):
                getattr(smpl, name).copy_(getattr(priv, name))
        smpl.forget()
        return smpl, scale.clone()

    def _optimize_smpl(self, smpl, data_dict, slot, iter_for_betas=10, iter_for_pose=10, iter_for_kpts=5, steps_per_iter=10,
                       max_iter=150):
        """[recon_fit_behave.py:224-291] global (betas 0-1 + translation, lr 0.02) -> all pose (new Adam, lr 0.006) ->
        + keypoints (same Adam), until the stop rule fires.  Reference details that change the result and are kept:
        the split parameters alias `smpl`'s storage; at the switch to 'smpl all pose' the OLD optimiser is zeroed, so
        global_pose / body_pose / other_betas enter the new Adam with the gradients summed over the whole 'global' phase
        (:243-259); the stop rule is tested after every inner step."""
        if slot is None:
            split = self.split_smpl(smpl)
            prev = torch.tensor(300.0, device=self.device)
        else:       # the kept steps read THIS split wrapper (views of the private body's storage) and THIS previous-loss cell
            split = slot.obj("split", lambda: self.split_smpl(smpl))
            prev = slot.bind("prev", torch.tensor(300.0, device=self.device))
            for p in split.parameters():      # a new call starts without accumulated gradients (the reference: fresh parameters)
                if p.grad is not None:
                    p.grad.zero_()
            split.forget()
        height_init = self.get_smpl_height(smpl)
        wd = self.get_loss_weights()

        def loss_of(phase):
            return lambda decay: self.sum_dict(self.forward_smpl(split, data_dict, phase), wd, decay)

        phase = "global"
        rel = self.release_graphs(split, data_dict["net"])
        carry = [split.global_pose, split.body_pose, split.other_betas, split.hand_pose]
        st = self._kept(slot, phase, lambda: self._stepper([split.top_betas, split.trans], 0.02, loss_of("global"), 0.001, prev,
                                                           release=rel, carry=carry))
        for it in range(iter_for_betas + iter_for_kpts + iter_for_pose + max_iter):
            zero = True
            if it == iter_for_betas:
                phase = "smpl all pose"
                st.zero_grads()            # the reference's zero_grad() of this iteration still goes through the old Adam
                st = self._kept(slot, phase, lambda: self._stepper(
                    [split.trans, split.global_pose, split.body_pose, split.top_betas, split.other_betas], 0.006,
                    loss_of("smpl all pose"), 0.001, prev, release=rel, carry=carry))
                zero = False
            elif it == iter_for_betas + iter_for_pose:
                phase = "kpts"            # same Adam, the loss gains the keypoint term
                prev_st = st
                st = self._kept(slot, phase, lambda: self._stepper(prev_st.params, 0.006, loss_of("kpts"), 0.001, prev,
                                                                   opt=prev_st.opt, release=rel, carry=carry), reset_opt=False)
            armed = self.early_stop and it > 0.25 * max_iter + iter_for_betas + iter_for_pose
            st.begin_outer(1 if phase != "kpts" else it / 3, armed=armed, zero=zero)
            self._inner(st, steps_per_iter, phase)
            if armed and st.stopped():
                break
        rel()   # graph replays change the parameters without touching their version counters: drop memoised results
        smpl.forget()
        scale = self.get_smpl_height(split) / height_init
        return self.copy_smpl_params(split, smpl), scale

    # ---- object + joint -----------------------------------------------------------------------------
    def forward_step(self, model, smpl, data_dict, obj_R, obj_t, obj_s, phase, noise=None, rot_noisy=None):
        const = data_dict.get("smpl_const")     # optimize_smpl_object: the body does not move, see there
        if const is None:
            smpl.forget()
            smpl_verts = smpl()[0]
        else:
            smpl_verts = const["verts"]
        loss_dict = {}
        # rot_noisy: obj_R + 1e-4 * this step's draw, already formed (fit_terms.rot_noise) = the argument of decopose_axis's projection
        R = self.project_so3(rot_noisy) if rot_noisy is not None else self.decopose_axis(obj_R, noise=noise)
        if phase == "sil":      # none of its terms reads the field (the reference queries the object points all the same, :171)
            sil = data_dict["silhouette"]
            obj_losses = sil.mask_loss(R, obj_t, obj_s)[0] if hasattr(sil, "mask_loss") else sil(R, obj_t, obj_s)[0]
            loss_dict["mask"] = obj_losses["mask"]
            loss_dict["scale"] = torch.mean((obj_s - self.obj_scale) ** 2)
            loss_dict["trans"] = torch.mean((obj_t - data_dict["trans_init"]) ** 2)
            return loss_dict
        object = self.transform_obj_verts(data_dict["objects"], R, obj_t, obj_s)
        model.query(object, **data_dict["query_dict"])
        preds = model.get_preds()
        df_pred, _, part_o, centers_o = preds
        stock = self._stock_terms("compute_obj_loss")
        if stock and fit_terms.obj_terms_supported(object, centers_o, obj_s, data_dict["smpl_center"]):
            self.compute_obj_loss(data_dict, loss_dict, model, obj_s, object, preds=preds, scale_term=False)
            loss_dict["scale"], loss_dict["ocent"] = fit_terms.obj_terms(object, centers_o, obj_s, data_dict["smpl_center"],
                                                                         self.obj_scale)
        else:
            obj_center_pred = data_dict["smpl_center"] + torch.mean(centers_o[:, 3:, :], -1)
            if stock:
                self.compute_obj_loss(data_dict, loss_dict, model, obj_s, object, preds=preds)
            else:       # a subclass's own compute_obj_loss (reference signature): it queries the points itself
                self.compute_obj_loss(data_dict, loss_dict, model, obj_s, object)
            loss_dict["ocent"] = F.mse_loss(torch.mean(object, 1), obj_center_pred, reduction="none").sum(-1).mean()
        if phase == "joint":
            df_obj_h = df_pred[:, 0, :]
            if const is None:
                model.query(smpl_verts, **data_dict["query_dict"])
                df_hum_o = model.get_preds()[0][:, 1, :]
            else:
                df_hum_o = const["df_hum_o"]
            # The collision term needs nothing of the contact term and both are chains of small launches.  CHORE_FIT_TWO_STREAMS=1
            # runs them side by side (fork / join; inside a recorded step = two branches of the graph): 0.280 -> 0.273 ms per
            # iteration, measured.  Off by default: gradients then cross streams inside autograd, whose tensors the caching
            # allocator hands back to the stream that made them -- not worth 2.6 % before that is audited.
            side = None
            if self.scan_faces is not None and object.is_cuda and os.environ.get("CHORE_FIT_TWO_STREAMS"):
                cur = torch.cuda.current_stream(object.device)
                side = self._side_stream(object.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    collide = self.compute_collision_loss(smpl_verts, smpl.faces, R, obj_t, obj_s)
            self.compute_contact_loss(df_hum_o, df_obj_h, object, smpl_verts, loss_dict, part_o=part_o)
            if side is not None:
                cur = torch.cuda.current_stream(object.device)
                cur.wait_stream(side)
                collide.record_stream(cur)
                loss_dict["collide"] = collide
            elif self.scan_faces is not None:
                loss_dict["collide"] = self.compute_collision_loss(smpl_verts, smpl.faces, R, obj_t, obj_s)
        return loss_dict

    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = self._side = torch.cuda.Stream(device)
        return st

    def optimize_smpl_object(self, model, data_dict, obj_iter=20, joint_iter=10, steps_per_iter=10, sil_iter=50,
                             max_iter=100):
        """[recon_fit_behave.py:90-163] object only (Adam on t, R, s, lr 0.006) -> silhouette (new Adam, same
        parameters) -> joint (new Adam on t, s, lr 0.002) until the stop rule fires; the SMPL parameters are never
        stepped.  `sil_iter` / `max_iter` are the reference's hard-coded 50 / 100.  The silhouette term is built from
        the two masks of data_dict['images'] like the reference does (:94-96) unless the caller put one into
        data_dict['silhouette']; without masks and template the phase is skipped.
        With reuse_graphs the recorded steps of the three phases are kept for the next call of the same shapes: they run on
        the slot's private object parameters and inputs (_FitSlot); the fitted values are copied into the caller's tensors."""
        smpl = data_dict["smpl"]
        split = self.split_smpl(smpl)
        data_dict["smpl"] = split
        if "silhouette" not in data_dict and self.scan is not None and "images" in data_dict:
            from .obj_pose_roi import SilLossROI
            images = data_dict["images"]
            data_dict["silhouette"] = SilLossROI(images[:, 3, :, :], images[:, 4, :, :], self.scan,
                                                 data_dict["query_dict"]["crop_center"], device=self.device,
                                                 crop_size=self.camera.crop_size).to(self.device)
        if "silhouette" not in data_dict:
            sil_iter = 0
        data_dict["smpl_center"] = self.compute_smpl_center_pred(data_dict, model, smpl)
        # no optimiser of this function owns a SMPL parameter (reference quirk, :102,126,134): their gradients would be
        # computed (LBS backward, the 6 890-point query backward of the joint phase) and never read -- switched off
        for p in split.parameters():
            p.requires_grad_(False)
        # ... and with them the body is a constant of all three phases: its vertices and the object distance queried at
        # them (the reference re-evaluates LBS and the 6 890-point query in every step, :167-168,196-197, with the same
        # result every time).  Evaluated once, the same values enter every step.
        with torch.no_grad():
            split.forget()
            body = split()[0].detach()
            model.query(body, **data_dict["query_dict"])
            data_dict["smpl_const"] = {"verts": body, "df_hum_o": model.get_preds()[0][:, 1, :].detach().clone()}
        split.forget()
        obj_R, obj_t, obj_s = data_dict["obj_R"], data_dict["obj_t"], data_dict["obj_s"]
        sil = data_dict.get("silhouette")
        slot = None
        if obj_R.is_cuda and (sil is None or hasattr(sil, "load_from")):
            slot = self._slot("object", tuple(obj_R.shape), tuple(data_dict["objects"].shape), tuple(body.shape), str(obj_R.device),
                              self._maps_key(model), obj_iter, joint_iter, steps_per_iter, sil_iter, max_iter,
                              None if sil is None else (type(sil).__name__, tuple(getattr(sil, "image_ref", torch.empty(0)).shape)))
        if slot is None:
            self._optimize_object(model, split, data_dict, None, obj_R, obj_t, obj_s, obj_iter, joint_iter, steps_per_iter,
                                  sil_iter, max_iter)
        else:
            # the steps' view of the call: private parameters, persistent copies of everything the loss terms read
            dd = slot.obj("data", dict)
            dd["query_dict"] = {"crop_center": slot.bind("crop_center", data_dict["query_dict"]["crop_center"])}
            dd["objects"] = slot.bind("objects", data_dict["objects"])
            dd["smpl_center"] = slot.bind("smpl_center", data_dict["smpl_center"])
            dd["smpl_const"] = {"verts": slot.bind("verts", body), "df_hum_o": slot.bind("df_hum_o", data_dict["smpl_const"]["df_hum_o"])}
            dd["smpl"] = slot.obj("split", lambda: split)      # (only its faces are read: the body enters through smpl_const)
            if sil is not None:
                kept = slot.obj("silhouette", lambda: sil)
                if kept is not sil:
                    kept.load_from(sil)
                dd["silhouette"] = kept
            pR, pt, ps = slot.bind("obj_R", obj_R, leaf=True), slot.bind("obj_t", obj_t, leaf=True), slot.bind("obj_s", obj_s, leaf=True)
            for p_ in (pR, pt, ps):
                if p_.grad is not None:
                    p_.grad.zero_()
            self._optimize_object(model, dd["smpl"], dd, slot, pR, pt, ps, obj_iter, joint_iter, steps_per_iter, sil_iter, max_iter)
            with torch.no_grad():
                obj_R.copy_(pR)
                obj_t.copy_(pt)
                obj_s.copy_(ps)
            for k in ("rot_init", "trans_init"):       # what the reference leaves in the caller's dict (:127-128)
                if k in dd:
                    data_dict[k] = dd[k].detach().clone()
        data_dict.pop("smpl_const", None)
        return smpl, data_dict["obj_R"], data_dict["obj_t"]

    def _optimize_object(self, model, split, data_dict, slot, obj_R, obj_t, obj_s, obj_iter, joint_iter, steps_per_iter, sil_iter,
                         max_iter):
        """the three phases on the given parameter tensors (the caller's, or a slot's private ones)"""
        wd = self.get_loss_weights()
        prev = torch.tensor(300.0, device=self.device)
        if slot is not None:
            prev = slot.bind("prev", prev)
        n_outer = joint_iter + obj_iter + max_iter + sil_iter
        # The SO(3) perturbation (recon_fit_base.py:384) comes from the CPU generator in the order the reference
        # consumes it -- one (B,3,3) draw per step, and between the last 'object only' step and the first silhouette
        # step the draw of rot_init's decopose_axis (:127) -- but up front, so that a step only reads noise[k] on the
        # device (consecutive torch.rand calls continue one stream: n calls = one call of n times the size).
        B = obj_R.shape[0]
        n_obj = obj_iter * steps_per_iter
        rng = cpu_generator()                  # the batch's own stream under fit_recon(batch_seed=...), else the process-wide one
        noise_obj = torch.rand(n_obj, B, 3, 3, generator=rng)
        noise_rot = torch.rand(B, 3, 3, generator=rng) if sil_iter > 0 else None
        noise = torch.cat([noise_obj, torch.rand((n_outer - obj_iter) * steps_per_iter, B, 3, 3, generator=rng)]).to(self.device)
        k = torch.zeros(1, dtype=torch.long, device=self.device)
        if slot is not None:
            noise, k = slot.bind("noise", noise), slot.bind("k", k)

        def loss_of(phase):
            def f(decay):
                if fit_terms.rot_noise_supported(obj_R, noise, k):      # the step's perturbed parameter and k += 1, one launch
                    return self.sum_dict(self.forward_step(model, split, data_dict, obj_R, obj_t, obj_s, phase,
                                                           rot_noisy=fit_terms.rot_noise(obj_R, noise, k)), wd, decay)
                nz = noise.index_select(0, k).squeeze(0)
                k.add_(1)
                return self.sum_dict(self.forward_step(model, split, data_dict, obj_R, obj_t, obj_s, phase, noise=nz), wd, decay)
            return f

        phase = "object only"
        rel = self.release_graphs(split, model)
        st = self._kept(slot, phase, lambda: self._stepper([obj_t, obj_R, obj_s], 0.006, loss_of("object only"), 0.0001, prev,
                                                           state=[k], release=rel))
        for it in range(n_outer):
            # zero_grad() of the reference runs at the top of the iteration through the optimiser of the PREVIOUS
            # phase (:118): identical here, every new Adam owns a subset of the previous one's parameters
            st.zero_grads()
            if it == obj_iter and sil_iter > 0:
                phase = "sil"
                rot_init = self.decopose_axis(obj_R, noise=noise_rot).detach().clone()
                trans_init = obj_t.detach().clone()
                # (a kept step reads these two through fixed addresses: written in place from the second call on)
                data_dict["rot_init"] = rot_init if slot is None else slot.bind("rot_init", rot_init)
                data_dict["trans_init"] = trans_init if slot is None else slot.bind("trans_init", trans_init)
                st = self._kept(slot, phase, lambda: self._stepper([obj_R, obj_s, obj_t], 0.006, loss_of("sil"), 0.0001, prev,
                                                                   state=[k], release=rel))
            if it == obj_iter + sil_iter:
                phase = "joint"
                st = self._kept(slot, phase, lambda: self._stepper([obj_t, obj_s], 0.002, loss_of("joint"), 0.0001, prev, state=[k],
                                                                   release=rel, carry=[obj_R]))
            decay = 1 if phase == "object only" else it
            if phase == "sil":
                decay = it - obj_iter + 1
            elif phase == "joint":
                decay = (it - obj_iter + 1) / 5
            armed = self.early_stop and phase == "joint" and it > 0.25 * max_iter
            st.begin_outer(decay, armed=armed, zero=False)
            self._inner(st, steps_per_iter, phase)
            if armed and st.stopped():
                break
        rel()


def recon_fit(args):
    """[recon_fit_behave.py:361-365]"""
    fitter = ReconFitterBehave(args.seq_folder, debug=getattr(args, "display", False), outpath=args.outpath, args=args)
    fitter.fit_recon(args)
    print("all done")

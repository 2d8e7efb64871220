"""Dense point-cloud extraction from the neural UDFs (Alg. 1 of the CHORE paper) on the GPU.

Host-side counterpart of /root/reference/recon/generator.py with the same public surface
(`Generator(model, exp_name, threshold, checkpoint, device, multi_gpus, sparse_thres, filter_val)`,
`generate_pclouds_batch`, `gen_pc_batch`, `approx_surface`, `compose_outdict`, `init_samples`,
`load_checkpoint`).  The inner loop -- CHORE.query forward + gradient w.r.t. the samples + the
projection step -- runs through chore_query_fwd / chore_query_bwd_points; everything stays on the
device (no .cpu() round trips per iteration as in generator.py:92-100).

Restated rules (file:line in the reference):
  approx_surface  :50-79   p <- p - normalize(grad_p sum(clamp(df_k, max=threshold))) * clamp(df_k, max=threshold)
  gen_pc_batch    :123-188 keep points with df < filter_val, resample 20000 of them per example with
                           randint + N(0,(threshold/3)^2); the first round only seeds the resampling
  compose_outdict :190-217 argmax of part logits, mean pca axis, mean centres over the first
                           `samples_count` collected points
  init_samples    :275-282 only batch element 0 is rescaled to the scene box (reference quirk, kept)
"""
import os
from glob import glob

import numpy as np
import torch
import torch.nn.functional as F


class Generator:
    def __init__(self, model, exp_name=None, threshold=1.0, checkpoint=None, device=torch.device("cuda"),
                 multi_gpus=True, sparse_thres=0.05, filter_val=0.03, checkpoint_root=None):
        self.sparse_thres = sparse_thres
        self.filter_val = filter_val
        self.sample_num = 100000
        self.model = model.to(device)
        self.model.eval()
        self.device = device
        self.threshold = threshold
        self.multi_gpus = multi_gpus
        root = checkpoint_root or os.path.join(os.getcwd(), "experiments")
        self.exp_path = os.path.join(root, str(exp_name)) + "/"
        self.checkpoint_path = self.exp_path + "checkpoints/"
        if exp_name is not None and os.path.isdir(self.checkpoint_path):
            self.load_checkpoint(checkpoint)
        for p in self.model.parameters():
            p.requires_grad = False
        self.pmin = np.array([-3.0, -0.9, 0.2])
        self.pmax = np.array([3.0, 1.80, 4.0])

    # ---- checkpoint handling (file format of trainer/trainer.py:186-206) ----
    def load_checkpoint(self, checkpoint):
        if checkpoint is None:
            cks = glob(self.checkpoint_path + "/*")
            if not cks:
                print("No checkpoints found at {}".format(self.checkpoint_path))
                return 0, 0
            path = self._best_checkpoint(cks)
        else:
            path = self.checkpoint_path + str(checkpoint)
        ck = torch.load(path, map_location="cpu")
        sd = ck["model_state_dict"]
        if self.multi_gpus:  # saved from DistributedDataParallel: strip the prefix (generator.py:256-260)
            sd = {k.replace("module.", ""): v for k, v in sd.items()}
        self.model.load_state_dict(sd)
        return ck["epoch"], ck["training_time"]

    def _best_checkpoint(self, cks):
        vm = glob(self.exp_path + "val_min=*")
        if vm:
            log = np.load(vm[0])
            p = self.checkpoint_path + str(log[2])
            if os.path.isfile(p):
                return p
        secs = sorted(float(os.path.splitext(os.path.basename(c))[0].split("_")[-1]) for c in cks)[-1]
        h, m, s = int(secs / 3600), int((secs / 60) % 60), int(secs % 60)
        return self.checkpoint_path + "checkpoint_{}h:{}m:{}s_{}.tar".format(h, m, s, secs)

    # ---- Alg. 1 ----
    def approx_surface(self, model, samples, num_steps, query_input, df_type):
        k = 0 if df_type == "human" else 1
        preds = None
        if (hasattr(model, "query_grad_points") and samples.is_cuda and samples.dtype == torch.float32 and
                not os.environ.get("CHORE_GEN_AUTOGRAD")):
            # the same step without an autograd graph: query forward, the clamp's gradient mask, backward to the points,
            # projection -- 4 launches (the tensor-expression loop below: ~20 per step, and the host is what it waits for)
            from .. import _lib
            dev = samples.device
            h = _lib.handle(dev.index or 0)
            s = samples.detach().contiguous()
            B, N, _ = s.shape
            cc = query_input["crop_center"]
            thr = float(self.threshold)
            df_only = hasattr(model, "query_df") and not os.environ.get("CHORE_GEN_ALL_HEADS")
            fused = hasattr(model, "surface_step")
            for it in range(num_steps):
                stream = torch.cuda.current_stream(dev).cuda_stream
                if fused and df_only and it < num_steps - 1:
                    new = model.surface_step(s, cc, k, thr)      # the whole step in one launch (same bits as the four below)
                    if new is not None:
                        s = new
                        continue
                    fused = False
                if df_only and it < num_steps - 1:
                    df = model.query_df(s, cc)        # only the last step's other predictions are read (generator.py:78-79)
                else:
                    with torch.no_grad():
                        model.query(s, **query_input)
                    preds = model.get_preds()
                    df = preds[0].contiguous()
                g = torch.empty_like(df)
                _lib.check(_lib.lib.chore_gen_clamp_mask(h, df.data_ptr(), k, thr, B, N, g.data_ptr(), stream), h, "chore_gen_clamp_mask")
                grad = model.query_grad_points(s, cc, g_df=g)
                new = torch.empty_like(s)
                _lib.check(_lib.lib.chore_gen_surface_step(h, s.data_ptr(), grad.data_ptr(), df.data_ptr(), k, thr, B, N,
                                                           new.data_ptr(), stream), h, "chore_gen_surface_step")
                s = new
            s.requires_grad = True
            return s, preds
        for _ in range(num_steps):
            model.query(samples, **query_input)
            preds = model.get_preds()
            df_target = torch.clamp(preds[0][:, k, :], max=self.threshold)
            grad, = torch.autograd.grad(df_target.sum(), samples)
            samples = (samples.detach() - F.normalize(grad, dim=2) * df_target.detach().unsqueeze(-1)).detach()
            samples.requires_grad = True
        return samples, preds

    def get_grid_samples(self, sample_num, batch_size=1):
        return self.init_samples(sample_num, batch_size)

    def filter(self, data):
        # (the Generator's model is frozen, recon/generator.py:41-42: without grad mode CHORE.filter does not have to walk its 486
        # parameters to find that out -- 1.2 ms per call)
        with torch.no_grad():
            self.model.filter(data["images"].to(self.device))

    def prep_query_input(self, batch):
        return {"crop_center": batch.get("crop_center").to(self.device)}

    def init_samples(self, sample_num, batch_size=1, generator=None):
        s = torch.rand(batch_size, sample_num, 3, generator=generator).float().to(self.device)
        s[0, :, 0] = s[0, :, 0] * 6 - 3
        s[0, :, 1] = s[0, :, 1] * 5 - 2.5
        s[0, :, 2] = (s[0, :, 2] - 0.5) * 0.5 + 2.2
        return s

    def generate_pclouds_batch(self, data, num_steps=10, num_points=50000, mute=False, generators=None):
        """generators: None = the process-wide random streams (the reference draws from torch's global CPU generator); or
        (cpu torch.Generator, device torch.Generator) -- every random number of this call comes from them, so the result does
        not depend on what else draws random numbers meanwhile (the pipelined fit prepares batch k+1 in a second thread)"""
        self.filter(data)
        bs = data.get("images").shape[0]
        if generators is None:
            samples = self.get_grid_samples(30000, batch_size=bs)
            rng = None
        else:
            cpu_g, dev_g = generators
            samples = self.init_samples(30000, bs, generator=cpu_g)
            dev = self.device
            rng = (lambda shape: torch.rand(shape, device=dev, generator=dev_g), lambda shape: torch.randn(shape, device=dev, generator=dev_g))
        return {t: self.gen_pc_batch(self.model, t, samples, num_points, data, num_steps, mute=mute, rng=rng)
                for t in ("human", "object")}

    def gen_pc_batch(self, model, df_type, samples_init, num_points, batch, num_steps, max_iter=100, mute=False,
                     rng=None, device_loop=True):
        """device_loop=True (default): masks, per-example lists and resampling stay on the device (csrc/generator.hip),
        the host reads one scalar per round; the resampling indices are floor(u * k) from DEVICE uniform numbers (the
        reference draws torch.randint on the CPU: same distribution, different stream).  rng: optional hooks so tests
        can inject the draws -- (uniform_fn(shape), randn_fn(shape)) here, (randint_fn(high, n), randn_fn(shape)) for the
        host loop.
        device_loop=False: the reference's control flow with its per-example host round trips (generator.py:149-188)."""
        if device_loop:
            return self._gen_pc_batch_device(model, df_type, samples_init, num_points, batch, num_steps, max_iter, mute, rng)
        query_input = self.prep_query_input(batch)
        k = 0 if df_type == "human" else 1
        bs = samples_init.shape[0]
        names = ["points", "pca_axis", "parts", "centers"]
        out = {n: [[] for _ in range(bs)] for n in names}
        sample_num = 20000
        # the reference draws on the CPU (torch.randint / torch.randn, generator.py:168-177): same calls, same stream
        randint = rng[0] if rng else (lambda high, n: torch.randint(high, (1, n))[0].to(self.device))
        randn = rng[1] if rng else (lambda shape: torch.randn(shape).to(self.device))
        it, count = 0, 0
        samples = samples_init.clone().to(self.device).requires_grad_(True)
        while count < num_points:
            surf, preds = self.approx_surface(model, samples, num_steps, query_input, df_type)
            df_t = torch.clamp(preds[0][:, k, :], max=self.threshold).detach()
            mask = df_t < self.filter_val
            if it > 0:
                counts = []
                for i in range(bs):
                    out["points"][i].append(surf[i, mask[i]].detach())
                    for n, p in zip(names[1:], preds[1:]):
                        out[n][i].append(p[i, ..., mask[i]].detach())
                    counts.append(int(mask[i].sum()))
                count += min(counts)
                if not mute:
                    print("{} points".format(count))
            new = []
            for i in range(bs):
                cand = samples[i, mask[i], :].detach()
                if cand.shape[0] > 1:
                    s_i = cand[randint(cand.shape[0], sample_num)].unsqueeze(0)
                    s_i = s_i + (self.threshold / 3) * randn(s_i.shape)
                else:
                    s_i = samples_init[i, randint(samples_init.shape[1], sample_num).to(samples_init.device)]
                    s_i = s_i.to(self.device).unsqueeze(0) + 0.5 * randn((1, sample_num, 3))
                new.append(s_i)
            samples = torch.cat(new, 0).detach().requires_grad_(True)
            it += 1
            if it == max_iter:
                raise RuntimeError("point generation failed after 100 iterations")
        self.compose_outdict(bs, out, names, count)
        return out

    def _gen_pc_batch_device(self, model, df_type, samples_init, num_points, batch, num_steps, max_iter, mute, rng):
        from .. import _lib
        query_input = self.prep_query_input(batch)
        k = 0 if df_type == "human" else 1
        dev = self.device
        h = _lib.handle(dev.index or 0)
        L = _lib.lib
        bs, sample_num = samples_init.shape[0], 20000
        uniform = rng[0] if rng else (lambda shape: torch.rand(shape, device=dev))
        randn = rng[1] if rng else (lambda shape: torch.randn(shape, device=dev))
        init = samples_init.detach().to(dev).float().contiguous()
        cap = int(num_points) + max(sample_num, init.shape[1])
        chans = {"points": 3, "pca_axis": 9, "parts": 14, "centers": 6}
        # channel-major result buffers (bs, C, cap); entries [offsets[b], ...) are appended every round
        bufs = {n: torch.empty(bs, c, cap, device=dev) for n, c in chans.items()}
        offsets = torch.zeros(bs, dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        samples = init.clone().requires_grad_(True)
        it, count = 0, 0
        while count < num_points:
            surf, preds = self.approx_surface(model, samples, num_steps, query_input, df_type)
            N = samples.shape[1]
            stream = torch.cuda.current_stream(dev).cuda_stream
            df_t = torch.clamp(preds[0][:, k, :], max=self.threshold).detach()
            mask = (df_t < self.filter_val).to(torch.uint8).contiguous()
            order = torch.empty(bs, N, dtype=torch.int32, device=dev)
            counts = torch.empty(bs, dtype=torch.int32, device=dev)
            _lib.check(L.chore_gen_compact(h, mask.data_ptr(), bs, N, order.data_ptr(), counts.data_ptr(), stream), h,
                       "chore_gen_compact")
            if it > 0:
                srcs = {"points": (surf.detach().float().contiguous(), (N * 3, 1, 3)),
                        "pca_axis": (preds[1].detach().float().reshape(bs, 9, N).contiguous(), (9 * N, N, 1)),
                        "parts": (preds[2].detach().float().contiguous(), (14 * N, N, 1)),
                        "centers": (preds[3].detach().float().contiguous(), (6 * N, N, 1))}
                for n, (src, (sb, sc, sn)) in srcs.items():
                    c = chans[n]
                    _lib.check(L.chore_gen_append(h, src.data_ptr(), sb, sc, sn, order.data_ptr(), counts.data_ptr(),
                                                  offsets.data_ptr(), bufs[n].data_ptr(), c * cap, cap, 1, bs, c, N, cap, stream),
                               h, "chore_gen_append")
                _lib.check(L.chore_gen_advance(h, counts.data_ptr(), bs, offsets.data_ptr(), total.data_ptr(), stream), h,
                           "chore_gen_advance")
            u = uniform((bs, sample_num)).float().contiguous()
            nz = randn((bs, sample_num, 3)).float().contiguous()
            new = torch.empty(bs, sample_num, 3, device=dev)
            cur = samples.detach()
            _lib.check(L.chore_gen_resample(h, cur.data_ptr(), bs, N, order.data_ptr(), counts.data_ptr(), init.data_ptr(),
                                            init.shape[1], u.data_ptr(), nz.data_ptr(), sample_num, float(self.threshold / 3),
                                            new.data_ptr(), stream), h, "chore_gen_resample")
            samples = new.requires_grad_(True)
            if it > 0:
                count = int(total.item())          # the one host read of the round: the loop condition
                if not mute:
                    print("{} points".format(count))
            it += 1
            if it == max_iter:
                raise RuntimeError("point generation failed after 100 iterations")
        out = {"points": bufs["points"][:, :, :count].permute(0, 2, 1).contiguous(),
               "pca_axis": bufs["pca_axis"][:, :, :count].mean(-1).view(bs, 3, 3),
               "parts": torch.argmax(bufs["parts"][:, :, :count], 1),
               "centers": bufs["centers"][:, :, :count].mean(-1)}
        return out

    def compose_outdict(self, batch_size, out_dict, out_names, samples_count):
        for name in out_names:
            comb = []
            for i in range(batch_size):
                if name == "points":
                    comb.append(torch.cat(out_dict[name][i], 0)[:samples_count, :])
                    continue
                o = torch.cat(out_dict[name][i], -1)[..., :samples_count]
                if name == "parts":
                    o = torch.argmax(o, 0)
                else:
                    o = torch.mean(o, -1)
                comb.append(o)
            out_dict[name] = torch.stack(comb, 0)

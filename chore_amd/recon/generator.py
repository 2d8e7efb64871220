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
        self.model.filter(data["images"].to(self.device))

    def prep_query_input(self, batch):
        return {"crop_center": batch.get("crop_center").to(self.device)}

    def init_samples(self, sample_num, batch_size=1, generator=None):
        s = torch.rand(batch_size, sample_num, 3, generator=generator).float().to(self.device)
        s[0, :, 0] = s[0, :, 0] * 6 - 3
        s[0, :, 1] = s[0, :, 1] * 5 - 2.5
        s[0, :, 2] = (s[0, :, 2] - 0.5) * 0.5 + 2.2
        return s

    def generate_pclouds_batch(self, data, num_steps=10, num_points=50000, mute=False):
        self.filter(data)
        bs = data.get("images").shape[0]
        samples = self.get_grid_samples(30000, batch_size=bs)
        return {t: self.gen_pc_batch(self.model, t, samples, num_points, data, num_steps, mute=mute)
                for t in ("human", "object")}

    def gen_pc_batch(self, model, df_type, samples_init, num_points, batch, num_steps, max_iter=100, mute=False,
                     rng=None):
        """rng: optional (randint_fn, randn_fn) hooks so tests can replay recorded resampling draws"""
        query_input = self.prep_query_input(batch)
        k = 0 if df_type == "human" else 1
        bs = samples_init.shape[0]
        names = ["points", "pca_axis", "parts", "centers"]
        out = {n: [[] for _ in range(bs)] for n in names}
        sample_num = 20000
        randint = rng[0] if rng else (lambda high, n: torch.randint(high, (n,), device=self.device))
        randn = rng[1] if rng else (lambda shape: torch.randn(shape, device=self.device))
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

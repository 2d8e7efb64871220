#!/usr/bin/env python
"""bench_fit.py -- ms per fit iteration (BASELINE configs[2] and [4]) on synthetic data.

Config 3 (default): B=1 frame, 300 Adam steps = 100 x forward_smpl('kpts') + 100 x forward_step('object only')
+ 100 x forward_step('joint' without the 'collide' term; the silhouette phase and the BVH collision term are
SURVEY 8(f) items that are not built).  Encoder run once on a 512x512 synthetic image (bf16 or fp32), the
fields are then queried by the fit exactly as recon/recon_fit_behave.py does: 6 890 SMPL-H vertices and
3 000 object points per step.  Prints ONE JSON line.

Config 5: `python -m torch.distributed.run --nproc-per-node N bench_fit.py --frames 64` shards the frames
round-robin over the ranks (chore_amd/parallel/frame_shard.py); each rank fits its frames as one batch; the
only collective is the final gather of the fitted parameters.
"""
import argparse
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for multi-process RCCL on this driver
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="Adam steps per phase")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    args = ap.parse_args()
    from bench import chore_opt
    from chore_amd.lib_smpl.priors import synthetic_priors
    from chore_amd.lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatch
    from chore_amd.model import CHORE
    from chore_amd.parallel import frames_of_rank, gather_fitted, init_distributed
    from chore_amd.recon.recon_fit_behave import ReconFitterBehave
    from chore_amd.utils import synth

    rank, world = init_distributed()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    mine = frames_of_rank(args.frames, rank, world)
    B = len(mine)
    opt = chore_opt(args.dtype)
    opt.gpu_id = local
    net = CHORE(opt).to(dev).eval()
    synth.load_synth_weights(net, seed=0)
    for p in net.parameters():
        p.requires_grad_(False)
    rs = np.random.RandomState(100 + rank)
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(0, B, 4):   # encode in batches of 4 images
            imgs = torch.from_numpy(synth.synth_images(min(4, B - i), 512, 512, seed=mine[i])).to(dev)
            net.filter(imgs)
            f, t = net.im_feat_list[0], net.tmpx
            feats = f if i == 0 else torch.cat([feats.permute(0, 2, 3, 1), f.permute(0, 2, 3, 1)]).permute(0, 3, 1, 2)
            tmpxs = t if i == 0 else torch.cat([tmpxs.permute(0, 2, 3, 1), t.permute(0, 2, 3, 1)]).permute(0, 3, 1, 2)
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - t0) * 1e3
    net.im_feat_list, net.tmpx = [feats], tmpxs
    pose, betas, trans = synth.synth_smpl_params(B, seed=1 + rank)
    pose *= 0.3
    smpl = SMPLPyTorchWrapperBatch(synth.synth_smplh_model(0), B, betas=torch.from_numpy(betas),
                                   pose=torch.from_numpy(pose), trans=torch.from_numpy(trans)).to(dev)
    body_prior, hand_prior = synthetic_priors(0, device=dev)
    labels = torch.from_numpy(rs.randint(0, 14, 6890)).to(dev)
    fitter = ReconFitterBehave.from_parts(device=dev, part_labels=labels, body_prior=body_prior, hand_prior=hand_prior)
    cc = torch.tensor([synth.CROP_CENTER] * B, device=dev)
    kpts = torch.from_numpy(np.concatenate([rs.uniform(100, 400, (B, 25, 2)), rs.uniform(0.2, 1, (B, 25, 1))], -1)
                            .astype(np.float32)).to(dev)
    obj = torch.from_numpy((rs.standard_normal((B, 3000, 3)) * 0.15).astype(np.float32)).to(dev)
    data = dict(net=net, query_dict={"crop_center": cc}, part_labels=labels.unsqueeze(0).repeat(B, 1),
                pose_init=torch.from_numpy(pose[:, 3:72]).to(dev), body_kpts=kpts, objects=obj)
    wd = fitter.get_loss_weights()
    split = fitter.split_smpl(smpl)
    obj_R = torch.eye(3, device=dev).repeat(B, 1, 1).requires_grad_(True)
    obj_t = torch.tensor([[0.2, 0.3, 2.3]] * B, device=dev).requires_grad_(True)
    obj_s = torch.ones(B, device=dev).requires_grad_(True)
    data["smpl_center"] = fitter.compute_smpl_center_pred(data, net, smpl)

    from chore_amd.recon.graph_step import EagerStep, GraphedStep
    noise = torch.rand(8 * args.steps + 64, B, 3, 3).to(dev)
    kidx = torch.zeros(1, dtype=torch.long, device=dev)

    def run(phase, n, graphed):
        """n inner steps of one phase (gradients zeroed every 10 steps like the reference's outer loop)"""
        if phase == "kpts":
            params = [split.trans, split.global_pose, split.body_pose, split.top_betas, split.other_betas]
            lr = 0.006
            loss_fn = lambda d: fitter.sum_dict(fitter.forward_smpl(split, data, "kpts"), wd, d)  # noqa: E731
        else:
            params = [obj_t, obj_R, obj_s] if phase == "object only" else [obj_t, obj_s]
            lr = 0.006 if phase == "object only" else 0.002

            def loss_fn(d):
                nz = noise.index_select(0, kidx).squeeze(0)
                kidx.add_(1)
                return fitter.sum_dict(fitter.forward_step(net, split, data, obj_R, obj_t, obj_s, phase, noise=nz), wd, d)
        prev = torch.tensor(300.0, device=dev)
        st = (GraphedStep if graphed else EagerStep)(params, lr, loss_fn, 1e-4, prev, state=[kidx],
                                                     release=fitter.release_graphs(split, net))
        st.begin_outer(1)
        for _ in range(3):   # warm-up
            st.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            if i % 10 == 0:
                st.begin_outer(1)
            st.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    phases = tuple(os.environ.get("CHORE_FIT_PHASES", "kpts,object only,joint").split(","))   # debugging aid
    ms_eager = {ph: run(ph, args.steps, False) for ph in phases} if not os.environ.get("CHORE_FIT_NO_EAGER") else {}
    ms = {ph: run(ph, args.steps, True) for ph in phases}
    fitted = gather_fitted({"trans": split.trans.detach(), "obj_t": obj_t.detach()}, args.frames, rank, world, device=dev)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([ms[ph] for ph in phases], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = dict(zip(phases, t.tolist()))
    if rank == 0:
        mean_ms = sum(ms.values()) / len(ms)
        print(json.dumps({
            "metric": "ms per fit iteration (SMPL-H LBS + field queries + Adam), mean over the three phases",
            "value": mean_ms, "unit": "ms", "higher_is_better": False, "n_gpus": world, "frames": args.frames,
            "frames_per_gpu": B, "steps_per_phase": args.steps, "dtype": args.dtype, "data": "synthetic",
            "ms_per_iter": ms, "ms_per_iter_eager": ms_eager, "inner_step": "hipGraph replay (chore_amd/recon/graph_step.py)", "frames_per_s_300_iters": args.frames / (3 * args.steps * mean_ms / 1e3) * (args.steps / 100),
            "encode_ms_first_call": enc_ms,
            "config": {"workload": "BASELINE configs[2]/[4]: 100 x forward_smpl(kpts) + 100 x forward_step(object only) + "
                                   "100 x forward_step(joint, no collide); 6890 SMPL-H vertices + 3000 object points per frame",
                       "excluded": "silhouette phase, BVH collision term (SURVEY 8f)"},
            "gathered": {k: list(v.shape) for k, v in fitted.items()}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

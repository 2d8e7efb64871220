#!/usr/bin/env python
"""bench_train.py -- training step/s (BASELINE configs[3]): per GPU B=4 synthetic 512x512 images, 20 000 points per
image, all 5 stacks, Adam, one process per GPU; gradients all-reduced by torch DDP over RCCL (backend "nccl")
when launched with more than one rank:

    python bench_train.py --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench_train.py --gpus N --steps 10 --warmup 3

A step = CHORE.forward (encoder, 5 field queries, loss of model/chore.py:192-237) + backward to all 18.25 M
parameters + Adam (trainer/trainer.py:76-131).  find_unused_parameters=True like the reference needs
(train_launch.py:30: the bn4 affines of the ConvBlocks without downsample never get a gradient).
Prints ONE JSON line from rank 0 (steps/s of the whole job, weak scaling)."""
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
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=20000)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
    from bench import chore_opt
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    opt = chore_opt(args.dtype)
    opt.gpu_id = local
    net = CHORE(opt).to(dev)
    synth.load_synth_weights(net, seed=0)
    net.train(True)
    net.losses_on_host = False     # the six separate losses stay on the device: no host synchronisation inside the step
    model = net
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local], find_unused_parameters=True)
    optim = torch.optim.Adam(net.parameters(), lr=1e-4)
    B, N = args.batch, args.points
    rs = np.random.RandomState(50 + rank)
    batch = dict(images=torch.from_numpy(synth.synth_images(B, 512, 512, seed=rank)).to(dev),
                 points=torch.from_numpy(synth.synth_points(B, N, seed=1 + rank)).to(dev),
                 df_h=torch.from_numpy(rs.uniform(0, 0.3, (B, N)).astype(np.float32)).to(dev),
                 df_o=torch.from_numpy(rs.uniform(0, 0.3, (B, N)).astype(np.float32)).to(dev),
                 parts_gt=torch.from_numpy(rs.randint(0, 14, (B, N))).to(dev),
                 pca_gt=torch.from_numpy(rs.standard_normal((B, 3, 3, N)).astype(np.float32)).to(dev),
                 body_center=torch.from_numpy((rs.standard_normal((B, 3)) * 0.3).astype(np.float32)).to(dev),
                 obj_center=torch.from_numpy((rs.standard_normal((B, 3, N)) * 0.3).astype(np.float32)).to(dev),
                 crop_center=torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev))

    def step():
        optim.zero_grad(set_to_none=True)
        error, _ = model(**batch)
        error.backward()
        optim.step()
        return error

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        err = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n_params = sum(p.numel() for p in net.parameters())
        print(json.dumps({
            "metric": "training steps/s (CHORE.forward + backward + Adam, B=4 x 512x512 images, 20k points/image per GPU)",
            "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "images_per_s": world * B * args.steps / elapsed,
            "higher_is_better": True, "scaling": "weak", "dtype": args.dtype, "data": "synthetic",
            "final_loss": float(err.detach()), "parameters": n_params,
            "config": {"workload": "BASELINE configs[3]: DDP training, batch 4/GPU, 20k points/image, 5 stacks",
                       "grad_allreduce": "torch DDP over RCCL (backend nccl), find_unused_parameters=True" if world > 1 else "none (1 GPU)"}}),
              flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

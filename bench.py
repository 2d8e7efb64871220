#!/usr/bin/env python
"""bench.py -- headline benchmark of the CHORE field-query hot path on MI355X.

Workload (BASELINE.json configs[1]): per GPU, ONE STEP = HGFilters encode of a batch of 4 synthetic
512x512 5-channel images + one 20 000-point MLP field query per image (80 000 points), i.e.
`CHORE.filter(images); CHORE.query(points, crop_center)` through libchore_hip.so.  Inputs are
resident in HBM before the timed region.  metric = query points per second (whole job, all GPUs).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the path shards by image (frames are independent); every rank runs the same
per-GPU workload on its own seeded inputs, no data-path collective ("scaling": "weak").  Timing is
bracketed by barrier + synchronize on both sides and the MAX over ranks is reported.

Extra objects in the JSON line:
  roofline      -- the dominant kernel of the step (by measured time), its ALGORITHMIC FLOPs
                   per launch / average launch duration measured live with hipEvents on the launch
                   stream (chore_profile_enable; that pass runs the encoder on ONE stream so kernels do not
                   overlap), against the dense MFMA peak of the dtype.  Kernel names are the rocprofv3 names.
  cpu_baseline  -- the numpy oracle ("port") timed on this host on a bounded sample of the same
                   workload (1 image encode + 20 000-point query = 1/4 step), rank 0, N=1 only.
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

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
HEADS_FLOP_PER_POINT = 600832.0
ENCODER_FLOP_PER_IMAGE = 258.25e9


def chore_opt(dtype):
    return argparse.Namespace(input_type="RGBM3", norm="group", num_stack=5, num_hourglass=2, hg_down="ave_pool",
                              hourglass_dim=256, skip_hourglass=True, z_feat="xyz", projection_mode="perspective",
                              loadSize=1200, net_img_size=[512, 512], gpu_id=0, compute_dtype=dtype)


def cpu_baseline():
    """numpy oracle on the host cores: 1 image encode + 20 000-point query (a quarter of one step)"""
    from chore_amd.utils import synth
    from oracle import encoder as oe, query as oq
    from chore_amd.model import CHORE
    spec = [(k, tuple(v.shape)) for k, v in CHORE(chore_opt("fp32")).state_dict().items()]
    sd = synth.synth_state_dict(spec, seed=0)
    img = synth.synth_images(1, 512, 512, seed=0)
    pts = synth.synth_points(1, 20000, seed=1)
    cc = np.array([synth.CROP_CENTER], np.float32)
    t0 = time.perf_counter()
    outs, tmpx, _ = oe.Encoder(sd).forward(img)
    t1 = time.perf_counter()
    oq.query(pts, cc, outs[-1], tmpx, sd)
    t2 = time.perf_counter()
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": 20000.0 / (t2 - t0), "unit": "points/s", "cores": int(threads), "kind": "port",
            "sample": "numpy oracle: 1 image 512x512 encode (%.2f s) + 20000-point query (%.3f s) = 1/4 step"
                      % (t1 - t0, t2 - t1),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    from chore_amd import _lib
    from chore_amd.model import CHORE
    from chore_amd.utils import synth

    opt = chore_opt(args.dtype)
    opt.gpu_id = local
    net = CHORE(opt).to(dev).eval()
    synth.load_synth_weights(net, seed=0)
    for p in net.parameters():
        p.requires_grad_(False)
    B, N = args.batch, args.points
    images = torch.from_numpy(synth.synth_images(B, 512, 512, seed=rank)).to(dev)
    points = torch.from_numpy(synth.synth_points(B, N, seed=1 + rank)).to(dev)
    cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)

    def step():
        net.filter(images)
        net.query(points, crop_center=cc)

    def barrier():
        if dist is not None:
            dist.barrier()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- component timings + live roofline measurement (outside the timed region) ----
        def timed(fn, n):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        enc_ms = timed(lambda: net.filter(images), 5)
        qry_ms = timed(lambda: net.query(points, crop_center=cc), 20)
        _lib.profile_enable(local, True)
        for _ in range(3):
            net.filter(images)
        prof = _lib.profile_read(local)
        _lib.profile_enable(local, False)

    if rank == 0:
        kernels = {}
        for k, v in prof.items():
            if v["launches"]:
                kernels[k] = {"ms_per_step": v["ms"] / 3, "launches_per_step": v["launches"] // 3,
                              "tflops": v["flops"] / v["ms"] / 1e9 if v["flops"] else None,
                              "gbps": v["bytes"] / v["ms"] / 1e6}
        tname = "unsigned short" if args.dtype == "bf16" else "float"
        kernels = {k.replace("<T,", "<%s, " % tname).replace(",", ", ").replace(",  ", ", "): v for k, v in kernels.items()}
        # HBM-side bytes per launch of each kernel from the committed rocprofv3 counter passes (FETCH_SIZE x2 +
        # WRITE_SIZE, scripts/pmc_traffic.py) -- counters cannot be read live, so this is the last profiled build
        traffic = {}
        tpath = os.path.join(REPO, "profiles", "pmc_traffic_%s.json" % args.dtype)
        if os.path.exists(tpath):
            traffic = {k: v["bytes_per_launch"] for k, v in json.load(open(tpath)).items()}
        # rocprofv3 name of the forward query kernel this size runs (csrc/query_fwd.hip: eight-wave variant for large queries,
        # 32-point tiles when 64-point tiles would not fill the CUs)
        qname = ("query_fwd_f32_kernel<%s, 1, false>" if B * ((N + 63) // 64) <= 256 else "query_fwd_f32_w8_kernel<%s>") % tname
        kernels[qname] = {"ms_per_step": qry_ms, "launches_per_step": 1,
                          "tflops": HEADS_FLOP_PER_POINT * B * N / qry_ms / 1e9, "gbps": None}
        dom = max((k for k in kernels if kernels[k]["tflops"]), key=lambda k: kernels[k]["ms_per_step"])
        dv = kernels[dom]
        dom_dtype = "fp32" if dom == qname else args.dtype
        roof = {"kernel": dom, "bound": "mfma", "achieved": dv["tflops"], "peak": PEAK_TFLOPS[dom_dtype],
                "unit": "TFLOP/s", "frac": dv["tflops"] / PEAK_TFLOPS[dom_dtype], "traffic": traffic.get(dom),
                "traffic_source": "profiles/pmc_traffic_%s.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
                                  % args.dtype if dom in traffic else None,
                "avg_launch_ms": dv["ms_per_step"] / dv["launches_per_step"],
                "flops_per_launch": dv["tflops"] * 1e9 * dv["ms_per_step"] / dv["launches_per_step"]}
        out = {
            "metric": "query-points/sec (HGFilters encode + 20k-pt MLP field query per 512x512 image)",
            "value": world * B * N * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: encode %dx(5,512,512) + query %dx%d points per GPU per step"
                                   % (B, B, N),
                       "images_per_gpu": B, "points_per_image": N, "image": "512x512x5",
                       "heads_dtype": "fp32 (exact-fp32 MFMA)", "sharding": "images across ranks, no collective"},
            "roofline": roof,
            "encode_ms": enc_ms,
            "query_ms": qry_ms,
            "query_only_points_per_s": B * N / qry_ms * 1e3,
            "encode_tflops": B * ENCODER_FLOP_PER_IMAGE / enc_ms / 1e9,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

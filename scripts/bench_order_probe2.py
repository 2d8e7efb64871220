"""fit record after the training record in one process (the reverse order of bench_order_probe.py)"""
import copy, os, sys, argparse, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
args = argparse.Namespace(gpus=1, steps=20, warmup=5, mode="all", dtype="fp16x3", batch=4, points=20000, frames_per_gpu=0, eager=False,
                          no_cpu_baseline=True, dry_run=False)
ctx = bench.Ctx(1)
if sys.argv[1] == "train_first":
    a = copy.copy(args); a.steps, a.warmup, a.dtype, a.mode = 20, 8, "bf16", "train"
    out = bench.mode_train(a, ctx)
    print("train ms per step", round(out["ms_per_step"], 3), flush=True)
    torch.cuda.empty_cache()
a = copy.copy(args); a.steps, a.warmup, a.mode = 1, 1, "fit"
out = bench.mode_fit(a, ctx)
print(sys.argv[1], "-> fit ms per iteration", round(out["value"], 4), "chain", round(out["ms_per_step"], 1), flush=True)
ctx.close()

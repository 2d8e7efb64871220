"""which part of bench.py's default mode makes its training record slower than `--mode train` alone?
python scripts/bench_order_probe.py <query|fit|none>  -> ms per training step after that phase ran in the same process"""
import copy, os, sys, argparse, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
first = sys.argv[1]
args = argparse.Namespace(gpus=1, steps=20, warmup=5, mode="all", dtype="fp16x3", batch=4, points=20000, frames_per_gpu=0, eager=False,
                          no_cpu_baseline=True, dry_run=False)
ctx = bench.Ctx(1)
if first == "query":
    bench.mode_query(args, ctx)
elif first == "fit":
    a = copy.copy(args); a.steps, a.warmup, a.mode = 1, 1, "fit"
    bench.mode_fit(a, ctx)
if os.environ.get("PROBE_GC"):
    import gc
    gc.collect()
    torch.cuda.synchronize()
if os.environ.get("PROBE_EMPTY", "1") == "1":
    torch.cuda.empty_cache()
a = copy.copy(args); a.steps, a.warmup, a.dtype, a.mode = 20, 8, "bf16", "train"
out = bench.mode_train(a, ctx)
print(first, "-> train ms per step", round(out["ms_per_step"], 3), flush=True)
ctx.close()

"""python scripts/half_batch_probe2.py whole|half : two recordings (whole batches in flight / the two halves of one batch), best partner stream"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from chore_amd.model import CHORE
from chore_amd.utils import synth
dev = torch.device("cuda", 0)
B, N = 4, 20000
mode = sys.argv[1]
net = CHORE(bench.chore_opt("fp16x3")).to(dev).eval()
synth.load_synth_weights(net, seed=0)
for p in net.parameters():
    p.requires_grad_(False)
images = torch.from_numpy(synth.synth_images(B, 512, 512, seed=0)).to(dev)
points = torch.from_numpy(synth.synth_points(B, N, seed=1)).to(dev)
cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)


def record(lo, hi):
    im, pt, c = images[lo:hi].contiguous(), points[lo:hi].contiguous(), cc[lo:hi].contiguous()
    def step():
        net.filter(im); net.query(pt, crop_center=c)
    s = torch.cuda.Stream(dev); s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.current_stream(dev).wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream(dev)):
        step()
    return g


def timeit(fn, n=24):
    for i in range(4): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    if mode == "whole":
        a, b = record(0, 4), record(0, 4)
        per_step = 2.0          # a pair = two steps
    else:
        a, b = record(0, 2), record(2, 4)
        per_step = 1.0          # a pair = one step
    print("first recording alone: %.3f ms" % timeit(lambda i: a.replay()))
    best = 1e9
    keep = []
    for j in range(8):
        sj = torch.cuda.Stream(dev); keep.append(sj)
        def pair(i, sj=sj):
            a.replay()
            with torch.cuda.stream(sj): b.replay()
        t = timeit(pair) / per_step
        best = min(best, t)
        print("   partner #%d  %.3f ms / step" % (j, t))
    print(mode, "best %.3f ms / step" % best)

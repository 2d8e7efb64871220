"""one step (4 images + 4 x 20 000 points) as TWO half-batch recordings replayed side by side on two streams, against one recording
of the whole batch and against two whole-batch recordings in flight"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from chore_amd.model import CHORE
from chore_amd.utils import synth
dev = torch.device("cuda", 0)
B, N = 4, 20000
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
        net.filter(im)
        net.query(pt, crop_center=c)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream(dev)):
        step()
    g.preds = net.get_preds()
    return g


def timeit(fn, n=40):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    net.filter(images); net.query(points, crop_center=cc); torch.cuda.synchronize()
    ref = [t.clone() for t in net.get_preds()]
    whole = record(0, 4)
    whole2 = record(0, 4)
    s2 = torch.cuda.Stream(dev)
    print("one recording of the whole batch              %.3f ms / step" % timeit(lambda i: whole.replay()))
    def two_whole(i):
        if i & 1:
            with torch.cuda.stream(s2): whole2.replay()
        else:
            whole.replay()
    print("two whole-batch recordings in flight          %.3f ms / step" % timeit(two_whole))
    for parts in (2, 4):
        hs = [record(k * B // parts, (k + 1) * B // parts) for k in range(parts)]
        ss = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(parts - 1)]
        def split(i):
            for g, s in zip(hs, ss):
                with torch.cuda.stream(s):
                    g.replay()
        print("%d part-batch recordings side by side per step  %.3f ms / step" % (parts, timeit(split)))
        split(0); torch.cuda.synchronize()
        got = [torch.cat([g.preds[j] for g in hs], 0) for j in range(4)]
        print("   outputs equal to the whole-batch step's:", all(torch.equal(a, b) for a, b in zip(got, ref)))

with torch.no_grad():
    print("partner stream sweep (whole on the default stream, whole2 on a fresh stream):")
    keep = []
    for j in range(10):
        sj = torch.cuda.Stream(dev); keep.append(sj)
        def two(i, sj=sj):
            if i & 1:
                with torch.cuda.stream(sj): whole2.replay()
            else:
                whole.replay()
        print("   stream #%d  %.3f ms / step" % (j, timeit(two, 20)))

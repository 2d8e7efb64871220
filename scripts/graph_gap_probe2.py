"""two recordings of the encode + query step replayed on two streams: shared vs separate activation workspaces"""
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
cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)
images = torch.from_numpy(synth.synth_images(B, 512, 512, seed=0)).to(dev)
points = torch.from_numpy(synth.synth_points(B, N, seed=1)).to(dev)


def step():
    net.filter(images)
    net.query(points, crop_center=cc)


def record(own_stream):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    kw = dict(stream=torch.cuda.Stream(dev)) if own_stream else {}
    with torch.cuda.graph(g, **kw):
        step()
    g.preds = net.get_preds()
    g.work = [v.data_ptr() for v in net.image_filter._work.values()]
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
    step(); torch.cuda.synchronize(); ref = [t.clone() for t in net.get_preds()]
    s2 = torch.cuda.Stream(dev)
    for own in (False, True):
        ga, gb = record(own), record(own)
        print("own capture streams:", own, " workspaces", len(set(ga.work + gb.work)))
        def two(i):
            if i & 1:
                with torch.cuda.stream(s2):
                    gb.replay()
            else:
                ga.replay()
        print("   one recording back to back      %.3f ms / step" % timeit(lambda i: ga.replay()))
        print("   two recordings, two streams     %.3f ms / step" % timeit(two))
        for g in (ga, gb):
            for t in g.preds:
                t.zero_()
        two(0); two(1); torch.cuda.synchronize()
        print("   outputs equal to eager:", [all(torch.equal(a, b) for a, b in zip(g.preds, ref)) for g in (ga, gb)])

"""Is there a start-up gap per hipGraph replay of the encode + query step, and does alternating two recordings hide it?
python scripts/graph_gap_probe.py"""
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


def record(seed):
    images = torch.from_numpy(synth.synth_images(B, 512, 512, seed=seed)).to(dev)
    points = torch.from_numpy(synth.synth_points(B, N, seed=1 + seed)).to(dev)

    def step():
        net.filter(images)
        net.query(points, crop_center=cc)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream(dev)           # its own capture stream: the encoder keeps one activation workspace per (shape, stream)
    with torch.cuda.graph(g, stream=cs):
        step()
    g.preds = net.get_preds()
    g.stream = cs
    return g, step


def timeit(fn, n=40):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    g1, step1 = record(0)
    net.image_filter.static_outputs = False
    g2, step2 = record(0)
    print("one recording, replayed back to back     %.3f ms / step" % timeit(lambda i: g1.replay()))
    print("two recordings, alternating              %.3f ms / step" % timeit(lambda i: (g1, g2)[i & 1].replay()))
    print("one recording, sync after every replay   %.3f ms / step" % timeit(lambda i: (g1.replay(), torch.cuda.synchronize())))
    print("eager                                    %.3f ms / step" % timeit(lambda i: step1()))
    for k in (2, 3, 4):
        gs = [g1, g2] + [record(0)[0] for _ in range(k - 2)]
        def multi(i, gs=gs):
            g = gs[i % len(gs)]
            with torch.cuda.stream(g.stream):
                g.replay()
        print("%d recordings in flight on %d streams        %.3f ms / step" % (k, k, timeit(multi, 48)))
        step1(); ref = [t.clone() for t in net.get_preds()]
        for g in gs:
            for t in g.preds:
                t.zero_()
        for i in range(len(gs)):
            multi(i)
        torch.cuda.synchronize()
        print("   outputs equal to the eager step's:", all(torch.equal(a, b) for g in gs for a, b in zip(g.preds, ref)))

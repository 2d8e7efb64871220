"""Does a CU-masked stream (chore_stream_create_cu_mask) confine kernels, launched eagerly AND replayed from a hipGraph?
A compute-bound kernel that fills the chip takes 256 / n times as long on n compute units.
    python scripts/probes/cu_mask_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from chore_amd import _lib

dev = torch.device("cuda", 0)
x = torch.randn(1 << 24, device=dev)


def work(y):
    # elementwise chain: many workgroups, each short -- the time follows the number of CUs (and HBM at the full chip)
    for _ in range(4):
        y = torch.sin(y) * 1.0001 + 0.1
    return y


def timed(stream, graph=None, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            graph.replay() if graph is not None else work(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            graph.replay() if graph is not None else work(x)
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / n


print("compute units:", _lib.lib.chore_cu_count(_lib.handle(0)))
full = torch.cuda.Stream(dev)
print("unmasked stream          eager %.3f ms" % timed(full))
for n in (256, 192, 128, 64, 32, 8):
    st = _lib.cu_masked_stream(0, n)
    t_e = timed(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        work(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            work(x)
    t_g = timed(st, g)
    # the same graph replayed on the UNMASKED stream: does the mask travel with the capture or with the launch stream?
    t_gf = timed(full, g)
    print("mask %3d CUs: eager %.3f ms   graph replayed on the masked stream %.3f ms   same graph on an unmasked stream %.3f ms"
          % (n, t_e, t_g, t_gf))

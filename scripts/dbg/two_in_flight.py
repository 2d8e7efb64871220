import sys, time
sys.path.insert(0, ".")
import torch, bench
from chore_amd.model import CHORE
from chore_amd.utils import synth
dev = torch.device("cuda", 0)
net = CHORE(bench.chore_opt("bf16")).to(dev).eval(); synth.load_synth_weights(net, seed=0)
for p in net.parameters(): p.requires_grad_(False)
B, N = 4, 20000
images = torch.from_numpy(synth.synth_images(B, 512, 512, seed=0)).to(dev)
points = torch.from_numpy(synth.synth_points(B, N, seed=1)).to(dev)
cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)
def run(nstreams, steps=40):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    def step(i):
        with torch.cuda.stream(streams[i % nstreams]):
            net.filter(images); net.query(points, crop_center=cc)
    with torch.no_grad():
        for i in range(8): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): step(i)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for n in (1, 2, 3, 4):
    print(n, "streams: %.3f ms per step" % run(n))

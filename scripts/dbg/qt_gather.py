import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
for dt in ("bf16", "fp16x3"):
    net = CHORE(chore_opt(dt)).cuda().eval(); synth.load_synth_weights(net, 0)
    for p in net.parameters(): p.requires_grad_(False)
    B = 4
    with torch.no_grad():
        net.filter(torch.from_numpy(synth.synth_images(B, 512, 512, 0)).cuda())
    cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
    for N in (20000, 20001):
        pts = torch.from_numpy(synth.synth_points(B, N, seed=1)).cuda()
        def fwd():
            with torch.no_grad(): net.query(pts, crop_center=cc)
        for _ in range(5): fwd()
        torch.cuda.synchronize(); t = time.perf_counter(); n = 50
        for _ in range(n): fwd()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
        print(dt, N, "%.3f ms" % ms)

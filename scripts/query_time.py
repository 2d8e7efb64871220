"""times the fused query: forward, and forward + backward to the points (one generator step), B x N points
python scripts/query_time.py [B N [B N ...]]   (default 4 20000; QT_MODES=bf16,fp32,fp16x3)"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
SHAPES = [(4, 20000)] if len(sys.argv) < 3 else [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
MODES = os.environ.get("QT_MODES", "bf16,fp32,fp16x3").split(",")
for dt in MODES:
    net = CHORE(chore_opt(dt)).cuda().eval(); synth.load_synth_weights(net, 0)
    for p in net.parameters(): p.requires_grad_(False)
    for B, N in SHAPES:
        with torch.no_grad():
            net.filter(torch.from_numpy(synth.synth_images(B, 512, 512, 0)).cuda())
        cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
        pts = torch.from_numpy(synth.synth_points(B, N, seed=1)).cuda()
        preq = pts.clone().requires_grad_(True)
        def fwd():
            with torch.no_grad(): net.query(pts, crop_center=cc)
        def fwdbwd():
            net.query(preq, crop_center=cc)
            df = net.get_preds()[0]
            torch.autograd.grad(df, preq, torch.ones_like(df))
        for name, f in (("forward", fwd), ("forward+backward to points", fwdbwd)):
            for _ in range(5): f()
            torch.cuda.synchronize(); t = time.perf_counter(); n = 100
            for _ in range(n): f()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
            print(f"{dt} maps, {B}x{N} points: {name} {ms:.3f} ms  ({B * N / (ms * 1e-3):.3e} points/s)")

"""is the split query-forward kernel deterministic?  same query 20 times, outputs compared bit for bit with the first"""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
net = CHORE(chore_opt("fp16x3")).cuda().eval(); synth.load_synth_weights(net, 0)
for B, N in ((1, 3000), (2, 20000)):
    with torch.no_grad():
        net.filter(torch.from_numpy(synth.synth_images(B, 128, 128, 0)).cuda())
        pts = torch.from_numpy(synth.synth_points(B, N, seed=3)).cuda()
        cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
        ref = None; bad = 0
        for it in range(20):
            net.query(pts, crop_center=cc)
            o = [p.clone() for p in net.get_preds()]
            if ref is None: ref = o
            else: bad += sum(int((a != b).sum()) for a, b in zip(ref, o))
        print(B, N, "differing values over 19 repeats:", bad)

"""field error against the reference's values and step time of one precision mode:  mode_err.py <mode> [<mode> ...]"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
from chore_amd.utils.field_check import field_errors
img = torch.from_numpy(synth.synth_images(4, 512, 512, 0)).cuda()
pts = torch.from_numpy(synth.synth_points(4, 20000, 1)).cuda(); cc = torch.tensor([synth.CROP_CENTER] * 4).cuda()
for mode in sys.argv[1:]:
    net = CHORE(chore_opt(mode)).cuda().eval(); synth.load_synth_weights(net, 0)
    for p in net.parameters(): p.requires_grad_(False)
    with torch.no_grad():
        for _ in range(3):
            net.filter(img); net.query(pts, crop_center=cc)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            net.filter(img); net.query(pts, crop_center=cc)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 100
        e = field_errors(net.get_preds())
    print(mode, "ms/step %.3f" % ms, {k: {a: float("%.3g" % b) for a, b in v.items()} for k, v in e.items()})

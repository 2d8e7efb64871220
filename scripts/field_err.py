"""field-value error of a precision mode against the fp32 mode at BASELINE configs[1] (4x512^2 encode + 4x20000 points).
fp32 mode is pinned to the reference (tests/test_gpu_encoder.py, tests/test_gpu_query.py); this prints what the other
mode adds on top.  python scripts/field_err.py [mode ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import chore_opt  # noqa: E402
from chore_amd.model import CHORE  # noqa: E402
from chore_amd.utils import synth  # noqa: E402


def run(mode, B=4, N=20000):
    net = CHORE(chore_opt(mode)).cuda().eval()
    synth.load_synth_weights(net, seed=0)
    for p in net.parameters():
        p.requires_grad_(False)
    images = torch.from_numpy(synth.synth_images(B, 512, 512, seed=0)).cuda()
    points = torch.from_numpy(synth.synth_points(B, N, seed=1)).cuda()
    cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32).cuda()
    with torch.no_grad():
        net.filter(images)
        net.query(points, crop_center=cc)
    df, pca, parts, centers = net.get_preds()
    return dict(df=df.cpu().numpy(), pca=pca.cpu().numpy(), parts=parts.cpu().numpy(), centers=centers.cpu().numpy(),
                feat=net.im_feat_list[-1].float().cpu().numpy(), tmpx=net.tmpx.float().cpu().numpy())


def main():
    modes = sys.argv[1:] or ["bf16"]
    ref = run("fp32")
    out = {}
    for m in modes:
        got = run(m)
        out[m] = {}
        for k in ref:
            d = np.abs(got[k].astype(np.float64) - ref[k])
            out[m][k] = dict(max_abs=float(d.max()), mean_abs=float(d.mean()), ref_absmax=float(np.abs(ref[k]).max()),
                             ref_absmean=float(np.abs(ref[k]).mean()),
                             rel_l2=float(np.sqrt((d ** 2).sum() / (ref[k].astype(np.float64) ** 2).sum())))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

import argparse, json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from chore_amd.model import CHORE
from chore_amd.utils import synth
from scripts.gpu_probe import opt_ns, nhwc
G = os.path.join(REPO, "tests", "golden")
what = sys.argv[1]
net = CHORE(opt_ns("fp32")).cuda().eval(); synth.load_synth_weights(net, 0)
for p in net.parameters(): p.requires_grad_(False)
if what == "bwd":
    g = np.load(os.path.join(G, "query_full.npz"))
    net.im_feat_list = [nhwc(g["feat"])]; net.tmpx = nhwc(g["tmpx"])
    names = ("df", "pca", "parts", "centers")
    for only in names + ("all",):
        pts = torch.from_numpy(g["points"]).cuda().requires_grad_(True)
        net.query(pts, crop_center=torch.from_numpy(g["crop_center"]).cuda())
        preds = net.get_preds()
        loss = sum((o * torch.from_numpy(g["w_" + k]).cuda()).sum() for k, o in zip(names, preds) if only in (k, "all"))
        loss.backward()
        gr = pts.grad.cpu().numpy()
        if only == "all":
            ref = g["dpoints"]; err = np.abs(gr - ref)
            print("all: rel", err.max() / np.abs(ref).max())
            idx = np.argsort(-err.max(-1).ravel())[:12]
            from oracle import query as oq
            nx, ny = oq.project_points(g["points"], g["crop_center"])
            for i in idx:
                b, n = divmod(i, 300)
                print(b, n, "pt", g["points"][b, n], "nxy", nx[b, n], ny[b, n], "got", gr[b, n], "ref", ref[b, n])
        else:
            print(only, "grad absmax", np.abs(gr).max(), "nan", np.isnan(gr).sum())
if what == "enc":
    ge = np.load(os.path.join(G, "encoder_64x96.npz"))
    with torch.no_grad(): net.filter(torch.from_numpy(ge["images"]).cuda())
    torch.cuda.synchronize()
    print("ok", net.im_feat_list[0].shape)

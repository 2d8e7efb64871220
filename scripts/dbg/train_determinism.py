"""Run the training forward+backward twice on the same sample and report which gradients differ bitwise."""
import sys, os, copy, argparse
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import golden
from test_gpu_encoder import make_net
opt = argparse.Namespace(input_type="RGBM3", norm="group", num_stack=5, num_hourglass=2, hg_down="ave_pool",
                         hourglass_dim=256, skip_hourglass=True, z_feat="xyz", projection_mode="perspective",
                         loadSize=1200, net_img_size=[512, 512], gpu_id=0)
tdt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
g = golden("train_loss.npz")
keys = ("images", "points", "df_h", "df_o", "parts_gt", "pca_gt", "body_center", "obj_center", "crop_center")
runs = []
for it in range(3):
    net = make_net(copy.copy(opt), tdt); net.train(True)
    for p in net.parameters(): p.requires_grad_(True)
    b = {k: torch.from_numpy(g[k][0:1]).cuda() for k in keys}
    error, _ = net.forward(**b); error.backward()
    runs.append((float(error), {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters() if p.grad is not None}))
    fe = [f.detach().float().cpu().numpy() for f in net.im_feat_list]
    runs[-1] = runs[-1] + (fe,)
print("errors", [r[0] for r in runs])
for it in (1, 2):
    print("feat bitwise equal", [bool((a == b).all()) for a, b in zip(runs[0][2], runs[it][2])])
    bad = [(n, float(np.abs(runs[0][1][n] - runs[it][1][n]).max() / (np.abs(runs[0][1][n]).max() + 1e-30))) for n in runs[0][1]
           if not (runs[0][1][n] == runs[it][1][n]).all()]
    print(f"run {it}: {len(bad)} of {len(runs[0][1])} gradients differ; first:", bad[:8], "worst:", sorted(bad, key=lambda x: -x[1])[:5])
if len(sys.argv) > 2:
    np.savez(sys.argv[2], **runs[0][1])

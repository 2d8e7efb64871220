import os, sys, copy, subprocess, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_ddp_trainstep import _make
def grads():
    net, batch = _make(0)
    net.train(True)
    err, _ = net(**batch); err.backward()
    return {n: p.grad.detach().float().cpu().numpy() for n, p in net.named_parameters() if p.grad is not None}
if len(sys.argv) > 1 and sys.argv[1] == "child":
    np.savez("/tmp/child_grads.npz", **grads()); sys.exit(0)
subprocess.check_call([sys.executable, __file__, "child"])
ref = np.load("/tmp/child_grads.npz")
g0 = grads()
print("fresh parent vs child: differing", sum(not np.array_equal(g0[n], ref[n]) for n in g0))
# what test_gpu_ddp's parent part does: fp32 nets on the small golden batch
from conftest import golden
from test_gpu_encoder import make_net
import argparse
opt = argparse.Namespace(input_type="RGBM3", norm="group", num_stack=5, num_hourglass=2, hg_down="ave_pool", hourglass_dim=256, skip_hourglass=True,
                         z_feat="xyz", projection_mode="perspective", loadSize=1200, net_img_size=[512, 512], gpu_id=0)
g = golden("train_loss.npz")
keys = ("images", "points", "df_h", "df_o", "parts_gt", "pca_gt", "body_center", "obj_center", "crop_center")
net = make_net(copy.copy(opt), "fp32"); net.train(True)
for p in net.parameters(): p.requires_grad_(True)
e, _ = net.forward(**{k: torch.from_numpy(g[k][0:1]).cuda() for k in keys}); e.backward()
del net
g1 = grads()
bad = [(n, float(np.abs(g1[n] - ref[n]).max() / (np.abs(ref[n]).max() + 1e-30))) for n in g1 if not np.array_equal(g1[n], ref[n])]
print("after an fp32 pass vs child: differing", len(bad), bad[:6])

import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from conftest import golden
import test_gpu_fit_chain as T
import argparse
opt = argparse.Namespace(input_type="RGBM3", norm="group", num_stack=5, num_hourglass=2, hg_down="ave_pool",
                              hourglass_dim=256, skip_hourglass=True, z_feat="xyz", projection_mode="perspective",
                              loadSize=1200, net_img_size=[512, 512], gpu_id=0)
g = golden("fit_schedule.npz")
fitter, net, smpl, data, data2 = T._fit_objects(opt, analytic=True)
torch.manual_seed(11)
smpl2, scale = fitter.optimize_smpl(smpl, data, iter_for_betas=2, iter_for_pose=2, iter_for_kpts=2, steps_per_iter=5, max_iter=8)
log = []
T._log_losses(fitter, "forward_step", log)
data2["smpl"] = smpl2
torch.manual_seed(12)
_, obj_R, obj_t = fitter.optimize_smpl_object(net, data2, obj_iter=3, joint_iter=2, steps_per_iter=3)
keys = [str(k) for k in g["keys_b"] if str(k) != "collide"]
got = T._loss_table(log, keys); ref = g["obj_losses"][:, :len(keys)]
np.set_printoptions(linewidth=200, precision=7, suppress=False)
print(keys)
for i in list(range(0, 14)) + list(range(60, 162, 6)) + [157,158,159,160]:
    if i < len(ref):
        print(i, got[i]); print(' ', ref[i])
print(len(got), len(ref))
print("rot_init diff", np.abs(data2["rot_init"].cpu().numpy() - g["rot_init"]).max())
print(data2["rot_init"].cpu().numpy()[0]); print(g["rot_init"][0])
print("trans_init diff", np.abs(data2["trans_init"].cpu().numpy() - g["trans_init"]).max())
print("smpl_center diff", np.abs(data2["smpl_center"].cpu().numpy() - g["smpl_center"]).max())

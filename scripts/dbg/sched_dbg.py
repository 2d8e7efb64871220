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
log = []
T._log_losses(fitter, "forward_smpl", log)
torch.manual_seed(11)
smpl2, scale = fitter.optimize_smpl(smpl, data, iter_for_betas=2, iter_for_pose=2, iter_for_kpts=2, steps_per_iter=5, max_iter=8)
keys = [str(k) for k in g["keys_a"]]
got = T._loss_table(log, keys); ref = g["smpl_losses"]
np.set_printoptions(linewidth=200, precision=6, suppress=False)
print(keys)
for i in range(min(len(got), len(ref))):
    print(i, got[i]); print(' ', ref[i])
print(len(got), len(ref))

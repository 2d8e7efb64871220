import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import argparse, torch
from test_gpu_collision import _joint_fit
opt = argparse.Namespace(input_type="RGBM3", norm="group", num_stack=5, num_hourglass=2, hg_down="ave_pool",
                         hourglass_dim=256, skip_hourglass=True, z_feat="xyz", projection_mode="perspective",
                         loadSize=1200, net_img_size=[512, 512], gpu_id=0)
mode = sys.argv[1]
r = _joint_fit(opt, mode == "graph")
print("done", mode, r[0], flush=True)

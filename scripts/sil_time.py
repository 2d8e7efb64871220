"""times the silhouette loss (SilLossROI.forward + backward) at the reference's sizes: 256x256 ROI, template of ~2.5k faces"""
import sys, os, time, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from chore_amd.recon.obj_pose_roi import SilLossROI
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = 36   # sphere of 2*n*(n-1) ~ 2.5k triangles
th, ph = np.meshgrid(np.linspace(0.05, np.pi - 0.05, n), np.linspace(0, 2 * np.pi, n, endpoint=False), indexing="ij")
v = (np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3) * 0.3).astype(np.float32)
idx = np.arange(n * n).reshape(n, n)
f = np.concatenate([np.stack([idx[:-1], idx[1:], np.roll(idx, -1, 1)[1:]], -1).reshape(-1, 3),
                    np.stack([idx[:-1], np.roll(idx, -1, 1)[1:], np.roll(idx, -1, 1)[:-1]], -1).reshape(-1, 3)])
S = 256
yy, xx = np.mgrid[0:S, 0:S]
obj = np.stack([((xx - 120) ** 2 + (yy - 130) ** 2) < 60 ** 2] * B)
K = np.array([[[2.0, 0, 0.5], [0, 2.0, 0.5], [0, 0, 1]]] * B, np.float32)
sil = SilLossROI.from_crops(obj, np.zeros_like(obj), K, v, f)
R = torch.eye(3).repeat(B, 1, 1).cuda().requires_grad_(True)
t = torch.tensor([[0.0, 0.0, 2.0]] * B).cuda().requires_grad_(True)
s = torch.ones(B).cuda().requires_grad_(True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        ld = sil(R, t, s)[0]; ld["mask"].backward()
    torch.cuda.synchronize(); print("B=%d faces=%d (x2 fill_back) %dx%d: %.3f ms per forward+backward, loss %.1f" % (B, len(f), S, S, (time.perf_counter() - t0) / 20 * 1e3, float(ld["mask"])))

"""Time the interpenetration term at the sizes of the fit: body-sized closed mesh (6 890 v / 13 776 f) + object."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from meshes import uv_ellipsoid, icosphere
from chore_amd.recon.recon_fit_base import ReconFitterBase, _CollisionFn
va, fa = uv_ellipsoid()
vb, fb = icosphere(4, 0.3, (0.33, 0.2, 0.05))
for B in (1, 8):
    fit = ReconFitterBase.from_parts(device="cuda:0")
    sv = torch.tensor(np.stack([va] * B), dtype=torch.float32, device="cuda")
    ov = torch.tensor(np.stack([vb + 0.01 * i for i in range(B)]), dtype=torch.float32, device="cuda").requires_grad_(True)
    sf, of = torch.tensor(fa, device="cuda"), torch.tensor(fb, device="cuda")
    for it in range(3):
        pen = fit.smpl_obj_collision(sv, sf, ov, of); pen.backward()
    torch.cuda.synchronize(); t = time.perf_counter(); n = 50
    for it in range(n):
        pen = fit.smpl_obj_collision(sv, sf, ov, of); pen.backward()
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t) / n * 1e3:.3f} ms fwd+bwd, loss {float(pen):.6f}")

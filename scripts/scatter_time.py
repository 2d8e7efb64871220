"""chore_scatter_features in isolation (B=4 x 20 000 points, feat 128^2 x 256): ms per call for point sets with different hit counts.
python scripts/scatter_time.py"""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from chore_amd import _lib
from chore_amd.utils import synth
from chore_amd.model.camera import KinectColorCamera
dev = torch.device("cuda", 0)
B, N = 4, 20000
cam6 = (ctypes.c_float * 6)(*KinectColorCamera(512).kernel_constants())
h = _lib.handle(0)
stream = torch.cuda.current_stream().cuda_stream
staging = torch.zeros(_lib.lib.chore_query_train_bytes(B, N), dtype=torch.uint8, device=dev)
staging.view(torch.float32)[B * N * (328 + 2 * 3 * 4 * 128):][:B * N * 328].normal_()
cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)
dfe = torch.empty(B, 128, 128, 256, device=dev)
dtm = torch.empty(B, 256, 256, 64, device=dev)


def run(name, pts, tm=False):
    points = torch.from_numpy(pts.astype(np.float32)).to(dev)
    for scan in (0, 1):
        if scan:
            os.environ["CHORE_SCATTER_SCAN"] = "1"
        else:
            os.environ.pop("CHORE_SCATTER_SCAN", None)
        call = lambda: _lib.check(_lib.lib.chore_scatter_features(h, points.data_ptr(), cc.data_ptr(), B, N, 128, 128, 256, 256, cam6,
                                                                  staging.data_ptr(), dfe.data_ptr(), dtm.data_ptr() if tm else None, 0, stream), h, "s")
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        print("%-34s %s  %.1f us / call" % (name, "scan  " if scan else "binned", e0.elapsed_time(e1) / 20 * 1e3))


p = synth.synth_points(B, N, seed=1)
run("bench points (17 % inside)", p)
far = p.copy(); far[..., 0] += 50.0
run("no point inside", far)
inside = p.copy(); inside[..., :2] = (inside[..., :2] - [[-0.0246, 0.4839]]) * 0.35 + [[-0.0246, 0.4839]]
run("all points inside", inside)
run("bench points, both maps", p, tm=True)

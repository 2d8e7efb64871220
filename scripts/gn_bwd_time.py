"""GroupNorm + ReLU backward (chore_gn_relu_bwd: a reduce launch + an apply launch) on the tensor sizes of the training step,
alone: us per call and the rate against the bytes the two passes must move (reduce: x, dA; apply: x, dA -> dx).
    python scripts/gn_bwd_time.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from chore_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
h = _lib.handle(0)
stream = torch.cuda.current_stream().cuda_stream
B = 4
for H, C in ((256, 64), (256, 32), (128, 256), (128, 128), (128, 64), (64, 256), (64, 128), (64, 64), (32, 256), (32, 128), (32, 64), (16, 256), (8, 256)):
    x = torch.randn(B, H, H, C, device=dev)
    da = torch.randn(B, H, H, C, device=dev) * 1e-3
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
    st = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.chore_gn_stats(h, _lib.F32, x.data_ptr(), B, H * H, C, st.data_ptr(), 1, stream), h, "stats")
    dx = torch.empty_like(x)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = torch.zeros(_lib.lib.chore_gn_relu_bwd_workspace_bytes(B, C), dtype=torch.uint8, device=dev)

    def call():
        _lib.check(_lib.lib.chore_gn_relu_bwd(h, _lib.F32, x.data_ptr(), st.data_ptr(), g.data_ptr(), b.data_ptr(), da.data_ptr(), B, H * H, C,
                                              dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), 0, stream), h, "gn bwd")
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    us = float(np.median(ts))
    T = x.numel() * 4
    print("B=4 %3d^2 x %3d: tensor %6.1f MB  %6.1f us per call (memset + reduce + apply)  %5.2f TB/s of 5 tensor passes" % (H, C, T / 1e6, us, 5 * T / us / 1e6))

"""every convolution layer shape of the encoder at B = 4 x 512^2, one at a time through chore_conv2d_fwd (GroupNorm + ReLU fused,
statistics in the epilogue): median us per launch over 50 hipGraph-free back-to-back launches (events around 20 launches x 5).
usage: python scripts/conv_layer_ab.py bf16|fp16x3 [1x1|3x3]   (kernel choice by environment: CHORE_CONV_LDS_BF16=1, CHORE_CONV_LDS=1, ...)"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from chore_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dt = {"bf16": _lib.BF16, "fp16x3": _lib.F16X3}[mode]
tdt = torch.bfloat16 if mode == "bf16" else torch.float32
dev = torch.device("cuda", 0)
h = _lib.handle(0)
B = 4
LAYERS = [(9, 64, 64, 256), (9, 64, 32, 256), (9, 32, 32, 256), (1, 64, 128, 256),
          (9, 128, 64, 128), (9, 64, 32, 128), (9, 32, 32, 128), (9, 128, 128, 128), (9, 64, 64, 128), (1, 128, 256, 128),
          (9, 256, 128, 128), (1, 256, 256, 128),
          (9, 256, 128, 64), (9, 128, 64, 64), (9, 64, 64, 64),
          (9, 256, 128, 32), (9, 128, 64, 32), (9, 64, 64, 32)]
if len(sys.argv) > 2:      # e.g. "1x1": only the layers whose label starts with it
    LAYERS = [l for l in LAYERS if ("%dx%d" % ((3, 3) if l[0] == 9 else (1, 1))).startswith(sys.argv[2])]
import os  # noqa: E402
if os.environ.get("CONV_AB_ONLY"):         # e.g. "256->128 @128": only the layers whose label contains it
    LAYERS = [l for l in LAYERS if os.environ["CONV_AB_ONLY"] in "%d->%d @%d" % (l[1], l[2], l[3])]
out = {}
stream = torch.cuda.current_stream().cuda_stream
for taps, cin, cout, H in LAYERS:
    k = 3 if taps == 9 else 1
    x = (torch.randn(B, H, H, cin, device=dev) * 1.5 + 0.3).to(tdt)
    w = torch.randn(cout, cin, k, k, device=dev) * (1.0 / np.sqrt(cin * taps))
    g, b = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
    st = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.chore_gn_stats(h, dt if mode == "bf16" else _lib.F32, x.data_ptr(), B, H * H, cin, st.data_ptr(), 1, stream), h, "stats")
    y = torch.empty(B, H, H, cout, dtype=tdt, device=dev)
    sty = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
    ws = torch.empty(max(16, _lib.lib.chore_conv2d_workspace_bytes(dt, taps, cin, cout)), dtype=torch.uint8, device=dev)

    def call():
        _lib.check(_lib.lib.chore_conv2d_fwd(h, dt, taps, x.data_ptr(), B, H, H, cin, st.data_ptr(), g.data_ptr(), b.data_ptr(), w.data_ptr(),
                                             None, cout, y.data_ptr(), sty.data_ptr(), ws.data_ptr(), stream), h, "conv")
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
    us = float(np.median(ts))      # (includes the ~4 us weight pack launch of the entry point)
    flop = 2.0 * taps * cin * cout * B * H * H
    out["%dx%d %d->%d @%d" % (k, k, cin, cout, H)] = {"us": round(us, 1), "tflops": round(flop / us / 1e6, 1)}
print(json.dumps(out))

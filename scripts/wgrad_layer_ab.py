"""weight gradient of every convolution layer shape of the encoder at B = 4 x 512^2 through chore_conv2d_bwd_weight (GroupNorm + ReLU
recomputed while staging): median us per call (partial sums + the ordered finish launch).
usage: python scripts/wgrad_layer_ab.py fp16x3|bf16|fp32   (CHORE_WGRAD_DBG=<bits> ablates phases of wgrad64_x3_kernel)"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from chore_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
dt = {"bf16": _lib.BF16, "fp16x3": _lib.F16X3, "fp32": _lib.F32}[mode]
tdt = torch.bfloat16 if mode == "bf16" else torch.float32
edt = _lib.BF16 if mode == "bf16" else _lib.F32
dev = torch.device("cuda", 0)
h = _lib.handle(0)
B = 4
LAYERS = [(9, 64, 64, 256), (9, 128, 64, 128), (9, 64, 64, 128), (9, 128, 128, 128), (1, 128, 256, 128), (9, 256, 128, 128), (1, 256, 256, 128),
          (9, 256, 128, 64), (9, 128, 64, 64), (9, 64, 64, 64), (9, 256, 128, 32), (9, 128, 64, 32), (9, 64, 64, 32)]
out = {}
stream = torch.cuda.current_stream().cuda_stream
for taps, cin, cout, H in LAYERS:
    k = 3 if taps == 9 else 1
    x = (torch.randn(B, H, H, cin, device=dev) * 1.5 + 0.3).to(tdt)
    dy = (torch.randn(B, H, H, cout, device=dev) * 1e-3).to(tdt)
    g, b = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
    st = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.chore_gn_stats(h, edt, x.data_ptr(), B, H * H, cin, st.data_ptr(), 1, stream), h, "stats")
    dw = torch.empty(cout, cin, k, k, device=dev)
    ws = torch.empty(max(16, _lib.lib.chore_conv2d_wgrad_workspace_bytes(taps, B, H, H, cin, cout)), dtype=torch.uint8, device=dev)
    amax = None
    if mode == "fp16x3":
        amax = torch.empty(_lib.lib.chore_amax_bytes(), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.chore_absmax_f32(h, dy.data_ptr(), dy.numel(), amax.data_ptr(), stream), h, "amax")

    def call():
        _lib.check(_lib.lib.chore_conv2d_bwd_weight(h, dt, taps, x.data_ptr(), B, H, H, cin, st.data_ptr(), g.data_ptr(), b.data_ptr(),
                                                    dy.data_ptr(), cout, dw.data_ptr(), None, ws.data_ptr(),
                                                    None if amax is None else amax.data_ptr(), stream), h, "wgrad")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    us = float(np.median(ts))
    flop = 2.0 * taps * cin * cout * B * H * H
    out["%dx%d %d->%d @%d" % (k, k, cin, cout, H)] = {"us": round(us, 1), "tflops": round(flop / us / 1e6, 1)}
print(json.dumps(out))

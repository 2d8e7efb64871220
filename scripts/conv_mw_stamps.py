"""phase times of conv_mw_kernel on one layer (scripts/build_variant.sh stamps conv_mw.hip -DMW_STAMPS=1; CHORE_HIP_LIB=<that library>):
thread 0 of every workgroup stamps the 100 MHz wall clock at start / main loop / epilogue / image written / stores issued / end.
usage: conv_mw_stamps.py [Cin Cout H]     (B = 4, fp16x3)"""
import ctypes, os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from chore_amd import _lib
cin, cout, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 128, 128)
B, dt, dev = 4, _lib.F16X3, torch.device("cuda", 0)
h = _lib.handle(0); stream = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, H, cin, device=dev) * 1.5 + 0.3
w = torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / np.sqrt(cin * 9))
g, b = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.2
st = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
_lib.check(_lib.lib.chore_gn_stats(h, _lib.F32, x.data_ptr(), B, H * H, cin, st.data_ptr(), 1, stream), h, "stats")
y = torch.empty(B, H, H, cout, device=dev)
sty = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
ws = torch.empty(max(16, _lib.lib.chore_conv2d_workspace_bytes(dt, 9, cin, cout)), dtype=torch.uint8, device=dev)
for _ in range(20):
    _lib.check(_lib.lib.chore_conv2d_fwd(h, dt, 9, x.data_ptr(), B, H, H, cin, st.data_ptr(), g.data_ptr(), b.data_ptr(), w.data_ptr(),
                                         None, cout, y.data_ptr(), sty.data_ptr(), ws.data_ptr(), stream), h, "conv")
torch.cuda.synchronize()
n = 2048 * 8
buf = (ctypes.c_ulonglong * n)()
L = ctypes.CDLL(os.environ["CHORE_HIP_LIB"])
assert L.chore_debug_mw_stamps(buf, n) == 0
a = np.array(buf[:]).reshape(2048, 8).astype(np.int64)
a = a[a[:, 0] > 0]
d = np.diff(a[:, :6], axis=1) / 100.0
print("%d->%d @%d: workgroups %d; phases (us: mean / median / p90)" % (cin, cout, H, len(a)))
for i, nm in enumerate(("prologue", "main loop", "image to LDS", "residual + stores", "statistics")):
    print("  %-18s %6.2f %6.2f %6.2f" % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("  workgroup total    %6.2f   kernel span (first start .. last end) %.1f us; starts spread over %.1f us, ends over %.1f us"
      % ((a[:, 5] - a[:, 0]).mean() / 100.0, (a[:, 5].max() - a[:, 0].min()) / 100.0, (a[:, 0].max() - a[:, 0].min()) / 100.0,
         (a[:, 5].max() - a[:, 5].min()) / 100.0))

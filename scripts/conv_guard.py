"""does a convolution launch write outside its outputs?  x, gamma / beta, weights, statistics, y, output statistics and the workspace are
carved out of ONE arena with sentinel-filled gaps between them; after the launch every byte that is not y / the output statistics /
the workspace must be what it was.   usage: conv_guard.py Cin Cout B H [W]"""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from chore_amd import _lib
cin, cout, B, H = (int(v) for v in sys.argv[1:5]); W = int(sys.argv[5]) if len(sys.argv) > 5 else H
dev = torch.device("cuda", 0); h = _lib.handle(0); dt = _lib.F16X3
stream = torch.cuda.current_stream().cuda_stream
GAP = 1 << 20
sizes = dict(x=B * H * W * cin * 4, w=cout * cin * 9 * 4, g=cin * 4, b=cin * 4, st=_lib.lib.chore_gn_stats_bytes(B), y=B * H * W * cout * 4,
             sty=_lib.lib.chore_gn_stats_bytes(B), ws=max(16, _lib.lib.chore_conv2d_workspace_bytes(dt, 9, cin, cout)))
off, o = {}, GAP
for k, n in sizes.items():
    off[k] = o; o += (n + 255) // 256 * 256 + GAP
arena = torch.full((o,), 0xA5, dtype=torch.uint8, device=dev)
view = lambda k: arena[off[k]:off[k] + sizes[k]]
gen = torch.Generator(device=dev); gen.manual_seed(1)
view("x").view(torch.float32).copy_((torch.randn(B * H * W * cin, device=dev, generator=gen) * 1.5 + 0.3))
view("w").view(torch.float32).copy_(torch.randn(cout * cin * 9, device=dev, generator=gen) / np.sqrt(cin * 9))
view("g").view(torch.float32).copy_(torch.rand(cin, device=dev, generator=gen) + 0.5)
view("b").view(torch.float32).copy_(torch.randn(cin, device=dev, generator=gen) * 0.2)
view("st").zero_(); view("sty").zero_()
p = lambda k: arena.data_ptr() + off[k]
_lib.check(_lib.lib.chore_gn_stats(h, _lib.F32, p("x"), B, H * W, cin, p("st"), 1, stream), h, "stats")
torch.cuda.synchronize()
before = arena.clone()
for _ in range(3):
    view("sty").zero_()
    _lib.check(_lib.lib.chore_conv2d_fwd(h, dt, 9, p("x"), B, H, W, cin, p("st"), p("g"), p("b"), p("w"), None, cout, p("y"), p("sty"), p("ws"), stream), h, "conv")
torch.cuda.synchronize()
changed = (arena != before)
for k in ("y", "sty", "ws"):
    changed[off[k]:off[k] + sizes[k]] = False
n = int(changed.sum())
print("%d->%d B=%d %dx%d: bytes changed outside y / output statistics / workspace: %d" % (cin, cout, B, H, W, n))
if n:
    idx = torch.nonzero(changed)[:, 0]
    for k in sizes:
        inside = int(((idx >= off[k]) & (idx < off[k] + sizes[k])).sum())
        if inside: print("   inside", k, inside, "first at +", int(idx[(idx >= off[k])][0]) - off[k])
    print("   first", int(idx[0]), "last", int(idx[-1]), "arena", o, {k: off[k] for k in off})

import os, sys, copy, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_train_ops import _RefConvBlock
from chore_amd import ops
from chore_amd.model import hgfilter_train as ht
cin, cout, (B, H, W) = 256, 256, (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
torch.manual_seed(0)
m = _RefConvBlock(cin, cout)
x = torch.randn(B, cin, H, W) * 1.5 + 0.3
up = torch.randn(B, cout, H, W)
xr = x.clone().requires_grad_(True)
(m(xr) * up).sum().backward()
res = {}
for mode in ("block", "layer"):
    md = copy.deepcopy(m).cuda(); md.zero_grad()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    y = ops.conv_block(xd, md)[0] if mode == "block" else ht._conv_block_layerwise(md, xd)
    (y * up.permute(0, 2, 3, 1).cuda()).sum().backward()
    res[mode] = dict(y=y.detach().permute(0, 3, 1, 2).cpu(), dx=xd.grad.permute(0, 3, 1, 2).cpu(),
                     **{n: p.grad.cpu() for n, p in md.named_parameters() if p.grad is not None})
ref = dict(y=m(x).detach(), dx=xr.grad, **{n: p.grad for n, p in m.named_parameters() if p.grad is not None})
for k in ref:
    s = ref[k].abs().max()
    print(k, "block-vs-ref %.2e  layer-vs-ref %.2e  block-vs-layer %.2e" % tuple(float((a - b).abs().max() / s) for a, b in
          ((res["block"][k], ref[k]), (res["layer"][k], ref[k]), (res["block"][k], res["layer"][k]))))
d = (res["block"]["dx"] - ref["dx"]).abs()
idx = (d > 1e-3 * ref["dx"].abs().max()).nonzero()
print("bad dx entries", len(idx), idx[:5].tolist(), idx[-5:].tolist() if len(idx) else "")

import os, sys, copy, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_train_ops import _RefConvBlock
from chore_amd import ops
from chore_amd.model import hgfilter_train as ht
for cin, cout, (B, H, W) in [(64, 128, (2, 24, 40)), (128, 128, (2, 16, 32)), (128, 256, (1, 32, 32)), (256, 256, (3, 20, 44)), (256, 256, (4, 64, 64))]:
    torch.manual_seed(cin + cout + H)
    blocks = [_RefConvBlock(cin, cout), _RefConvBlock(cout, cout)]
    for m in blocks:
        for n, p in m.named_parameters():
            if "bn" in n:
                p.data = (torch.rand_like(p) + 0.5) if n.endswith("weight") else torch.randn_like(p) * 0.2
    x = torch.randn(B, cin, H, W) * 1.5 + 0.3
    up = torch.randn(B, cout, H, W)
    xr = x.bfloat16().float().clone().requires_grad_(True)
    (blocks[1](blocks[0](xr)) * up).sum().backward()
    out = []
    for mode in ("block", "layer"):
        dev = [copy.deepcopy(m).cuda() for m in blocks]
        for m in dev: m.zero_grad()
        xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda().requires_grad_(True)
        if mode == "block":
            y1, s1 = ops.conv_block(xd, dev[0]); y2, _ = ops.conv_block(y1, dev[1], s1)
        else:
            y2 = ht._conv_block_layerwise(dev[1], ht._conv_block_layerwise(dev[0], xd))
        (y2.float() * up.permute(0, 2, 3, 1).cuda()).sum().backward()
        def l2(a, b):
            a, b = a.float().cpu(), b.float().cpu()
            return float((a - b).norm() / b.norm()), float((a - b).abs().quantile(0.99) / b.abs().max()) if a.numel() < 16e6 else -1
        out.append((mode, "dx L2 %.3f p99 %.3f" % l2(xd.grad.permute(0, 3, 1, 2), xr.grad),
                    "w1 L2 %.3f p99 %.3f" % l2(dev[0].conv1.weight.grad, blocks[0].conv1.weight.grad),
                    "w3' L2 %.3f p99 %.3f" % l2(dev[1].conv3.weight.grad, blocks[1].conv3.weight.grad)))
    print(cin, cout, (B, H, W), out)

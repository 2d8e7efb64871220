"""GPU: the training operators (chore_amd/ops.py -> chore_conv2d_fwd / _bwd_data / _bwd_weight, chore_gn_relu_fwd / _bwd)
against torch's fp32 CPU implementation of the same layers with autograd (what the reference runs:
nn.Conv2d / nn.GroupNorm(32, C) / ReLU, model/net_util.py:346-396).

Tolerances: fp32 mode 2e-5 of the tensor's largest entry; bf16 mode (bf16 activations and MFMA operands, fp32
accumulation) 3e-2 of it, relative L2 <= 1.5e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ref_layer(x_nchw, w, bias, gamma, beta):
    a = x_nchw
    if gamma is not None:
        a = F.relu(F.group_norm(a, 32, gamma, beta, eps=1e-5))
    return F.conv2d(a, w, bias, padding=w.shape[-1] // 2)


def check(a, b, dtype, what):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    scale = np.abs(b).max()
    if dtype == torch.float32:
        assert np.abs(a - b).max() <= 2e-5 * scale, (what, np.abs(a - b).max(), scale)
    else:
        assert np.abs(a - b).max() <= 3e-2 * scale, (what, np.abs(a - b).max(), scale)
        assert np.linalg.norm((a - b).ravel()) <= 1.5e-2 * np.linalg.norm(b.ravel()), what


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,cin,cout,gn,bias,shape", [
    (3, 64, 32, True, False, (2, 24, 40)),     # ragged tiles
    (3, 128, 64, True, False, (2, 16, 32)),
    (1, 64, 128, True, False, (2, 16, 32)),    # downsample path: 1x1 after bn4
    (1, 128, 128, False, True, (3, 8, 32)),    # plain 1x1 with bias (l / al / bl)
    (3, 256, 128, True, False, (1, 32, 32)),
    (3, 64, 64, True, False, (2, 20, 44)),     # 64-channel-tile weight-gradient kernel, ragged tiles
    (3, 128, 64, False, True, (3, 24, 96)),    # ... plain conv with bias, several tiles per share
    (3, 64, 128, True, False, (5, 64, 64)),    # ... more tiles than shares
    (1, 128, 256, True, False, (2, 20, 44)),   # ... 1x1, ragged
    (1, 256, 256, False, True, (2, 32, 64)),   # ... 1x1 plain with bias
])
def test_conv_gn_layer(dtype, k, cin, cout, gn, bias, shape):
    from chore_amd import ops
    g = torch.Generator().manual_seed(k * 1000 + cin + cout)
    B, H, W = shape
    x = torch.randn(B, cin, H, W, generator=g) * 1.5 + 0.3
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    bs = torch.randn(cout, generator=g) * 0.1 if bias else None
    gamma = torch.rand(cin, generator=g) + 0.5 if gn else None
    beta = torch.randn(cin, generator=g) * 0.2 if gn else None
    up = torch.randn(B, cout, H, W, generator=g)
    # reference on CPU; in bf16 mode the reference sees the same bf16-rounded input
    xr = x.to(dtype).float().clone().requires_grad_(True)
    pr = [p.clone().requires_grad_(True) if p is not None else None for p in (w, bs, gamma, beta)]
    yr = ref_layer(xr, *pr)
    (yr * up).sum().backward()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    pd = [p.clone().cuda().requires_grad_(True) if p is not None else None for p in (w, bs, gamma, beta)]
    yd = ops.conv_gn(xd, *pd)
    (yd.float() * up.permute(0, 2, 3, 1).cuda()).sum().backward()
    check(yd.permute(0, 3, 1, 2), yr, dtype, "y")
    check(xd.grad.permute(0, 3, 1, 2), xr.grad, dtype, "dx")
    for name, a, b in zip(("dw", "dbias", "dgamma", "dbeta"), pd, pr):
        if a is not None:
            check(a.grad, b.grad, dtype, name)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gn_relu(dtype):
    from chore_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 256, 16, 24
    x = torch.randn(B, C, H, W, generator=g) * 2 - 0.4
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    up = torch.randn(B, C, H, W, generator=g)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.relu(F.group_norm(xr, 32, gr, br, eps=1e-5))
    (yr * up).sum().backward()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    gd, bd = gamma.clone().cuda().requires_grad_(True), beta.clone().cuda().requires_grad_(True)
    yd = ops.gn_relu(xd, gd, bd)
    (yd.float() * up.permute(0, 2, 3, 1).cuda()).sum().backward()
    check(yd.permute(0, 3, 1, 2), yr, dtype, "y")
    check(xd.grad.permute(0, 3, 1, 2), xr.grad, dtype, "dx")
    check(gd.grad, gr.grad, dtype, "dgamma")
    check(bd.grad, br.grad, dtype, "dbeta")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 8, 12, 64), (1, 16, 16, 256), (2, 2, 3, 64)])
def test_upadd(dtype, shape):
    """a + bicubic x2 (align_corners) and its transpose, against F.interpolate + autograd on the CPU"""
    from chore_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    B, H, W, C = shape
    low = torch.randn(B, C, H, W, generator=g)
    a = torch.randn(B, C, 2 * H, 2 * W, generator=g)
    up = torch.randn(B, C, 2 * H, 2 * W, generator=g)
    lr = low.to(dtype).float().clone().requires_grad_(True)
    ar = a.to(dtype).float().clone().requires_grad_(True)
    yr = ar + F.interpolate(lr, scale_factor=2, mode="bicubic", align_corners=True)
    (yr * up).sum().backward()
    ld = low.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    ad = a.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    yd = ops.upadd(ad, ld)
    (yd.float() * up.permute(0, 2, 3, 1).cuda()).sum().backward()
    check(yd.permute(0, 3, 1, 2), yr, dtype, "y")
    check(ld.grad.permute(0, 3, 1, 2), lr.grad, dtype, "dlow")
    check(ad.grad.permute(0, 3, 1, 2), ar.grad, dtype, "da")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 5, 64, 96), (1, 5, 22, 18), (3, 3, 16, 16)])
def test_stem(dtype, shape):
    """7x7 stride-2 stem (HGFilters.py:102,149): forward, weight and bias gradient (ragged tiles, Cin 3 and 5)"""
    from chore_amd import ops
    torch.manual_seed(7)
    B, Cin, H, W = shape
    img = torch.randn(B, Cin, H, W)
    w = (torch.randn(64, Cin, 7, 7) * 0.1).requires_grad_(True)
    bias = torch.randn(64).requires_grad_(True)
    up = torch.randn(B, 64, H // 2, W // 2)
    yr = F.conv2d(img, w, bias, stride=2, padding=3)
    (yr * up).sum().backward()
    wd, bd = w.detach().clone().cuda().requires_grad_(True), bias.detach().clone().cuda().requires_grad_(True)
    yd = ops.stem(img.cuda(), wd, bd, dtype)
    assert yd.dtype == dtype and yd.shape == (B, H // 2, W // 2, 64)
    (yd * up.permute(0, 2, 3, 1).cuda().to(dtype)).sum().backward()
    check(yd.permute(0, 3, 1, 2), yr, dtype, "y")
    check(wd.grad, w.grad, dtype, "dw")
    check(bd.grad, bias.grad, dtype, "dbias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 8, 12, 64), (1, 16, 16, 256), (2, 2, 6, 128)])
def test_avgpool2(dtype, shape):
    from chore_amd import ops
    torch.manual_seed(8)
    B, H, W, C = shape
    x = torch.randn(B, C, H, W).to(dtype).float().requires_grad_(True)
    up = torch.randn(B, C, H // 2, W // 2).to(dtype).float()
    yr = F.avg_pool2d(x, 2, stride=2)
    (yr * up).sum().backward()
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    yd = ops.avgpool2(xd)
    (yd * up.permute(0, 2, 3, 1).cuda().to(dtype)).sum().backward()
    check(yd.permute(0, 3, 1, 2), yr, dtype, "y")
    check(xd.grad.permute(0, 3, 1, 2), x.grad, dtype, "dx")

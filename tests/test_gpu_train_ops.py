"""GPU: the training operators (chore_amd/ops.py -> chore_conv2d_fwd / _bwd_data / _bwd_weight, chore_gn_relu_fwd / _bwd)
against torch's fp32 CPU implementation of the same layers with autograd (what the reference runs:
nn.Conv2d / nn.GroupNorm(32, C) / ReLU, model/net_util.py:346-396).

Tolerances: fp32 mode 2e-5 of the tensor's largest entry; bf16 mode (bf16 activations and MFMA operands, fp32
accumulation) 3e-2 of it, relative L2 <= 1.5e-2.  "x3" = the fp16x3 training mode (round 5): fp32 tensors, every convolution, data
gradient and weight gradient on the fp16 matrix cores with hi / lo split operands -- held to the fp32 bounds, with upstream
gradients scaled from 1e-7 to 3e4 (the operand scale of the gradient GEMMs, csrc/enc_common.h x3_in_scale)."""
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


def _tdt(dtype):
    return torch.float32 if dtype == "x3" else dtype


def check(a, b, dtype, what):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    scale = np.abs(b).max()
    if _tdt(dtype) == torch.float32:
        assert np.abs(a - b).max() <= 2e-5 * scale, (what, np.abs(a - b).max(), scale)
    else:
        assert np.abs(a - b).max() <= 3e-2 * scale, (what, np.abs(a - b).max(), scale)
        assert np.linalg.norm((a - b).ravel()) <= 1.5e-2 * np.linalg.norm(b.ravel()), what


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, "x3"])
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
    (1, 128, 256, True, True, (3, 16, 32)),    # 1x1 on 128 x 128-channel tiles (fp16 x 3), GroupNorm recomputed, three images, bias
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
    if dtype == "x3":      # gradients of any magnitude: a different power of ten per layer shape
        up = up * float(10.0 ** ((cin // 32 + cout // 32 + k + H) % 12 - 7))
    x3, dtype = dtype == "x3", _tdt(dtype)
    # reference on CPU; in bf16 mode the reference sees the same bf16-rounded input
    xr = x.to(dtype).float().clone().requires_grad_(True)
    pr = [p.clone().requires_grad_(True) if p is not None else None for p in (w, bs, gamma, beta)]
    yr = ref_layer(xr, *pr)
    (yr * up).sum().backward()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    pd = [p.clone().cuda().requires_grad_(True) if p is not None else None for p in (w, bs, gamma, beta)]
    with ops.x3_convs(x3):
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


class _RefConvBlock(torch.nn.Module):
    """the reference's ConvBlock (model/net_util.py:346-396, norm='group') restated with torch.nn layers"""

    def __init__(self, cin, cout):
        super().__init__()
        nn = torch.nn
        self.conv1 = nn.Conv2d(cin, cout // 2, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(cout // 2, cout // 4, 3, padding=1, bias=False)
        self.conv3 = nn.Conv2d(cout // 4, cout // 4, 3, padding=1, bias=False)
        self.bn1, self.bn2, self.bn3, self.bn4 = (nn.GroupNorm(32, c) for c in (cin, cout // 2, cout // 4, cin))
        self.downsample = None if cin == cout else nn.Sequential(self.bn4, nn.ReLU(True), nn.Conv2d(cin, cout, 1, bias=False))

    def forward(self, x):
        o1 = self.conv1(F.relu(self.bn1(x)))
        o2 = self.conv2(F.relu(self.bn2(o1)))
        o3 = self.conv3(F.relu(self.bn3(o2)))
        res = x if self.downsample is None else self.downsample(x)
        return torch.cat((o1, o2, o3), 1) + res


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, "x3"])
@pytest.mark.parametrize("cin,cout,shape", [(64, 128, (2, 24, 40)), (128, 128, (2, 16, 32)), (128, 256, (1, 32, 32)),
                                            (256, 256, (3, 20, 44)), (256, 256, (4, 64, 64))])
def test_conv_block_operator(dtype, cin, cout, shape):
    """chore_convblock_fwd / _bwd (one call per direction: slices written in place, residual and skip gradients fused)
    against torch CPU autograd of the reference block, and two chained blocks (the second normalises with the statistics
    the first one's epilogues produced)"""
    from chore_amd import ops
    B, H, W = shape
    torch.manual_seed(cin + cout + H)
    blocks = [_RefConvBlock(cin, cout), _RefConvBlock(cout, cout)]
    for m in blocks:
        for n, p in m.named_parameters():
            if "bn" in n:
                p.data = (torch.rand_like(p) + 0.5) if n.endswith("weight") else torch.randn_like(p) * 0.2
    x = torch.randn(B, cin, H, W) * 1.5 + 0.3
    up = torch.randn(B, cout, H, W)
    x3, dtype = dtype == "x3", _tdt(dtype)
    if x3:
        up = up * float(10.0 ** ((cin // 64 + H) % 9 - 6))        # gradients of any magnitude
    xr = x.to(dtype).float().clone().requires_grad_(True)
    yr1 = blocks[0](xr)
    yr = blocks[1](yr1)
    (yr * up).sum().backward()
    import copy
    dev = [copy.deepcopy(m).cuda() for m in blocks]
    for m in dev:
        m.zero_grad()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    with ops.x3_convs(x3):
        y1, s1 = ops.conv_block(xd, dev[0])
        y2, _ = ops.conv_block(y1, dev[1], s1)
    (y2.float() * up.permute(0, 2, 3, 1).cuda()).sum().backward()

    def check(a, b, dtype, what):
        # A pre-activation within rounding of zero takes a different ReLU branch here than on the CPU (about one per
        # million activations in fp32), which moves that entry's gradient -- and the 5x5 pixels the next two 3x3
        # data-gradient convolutions spread it over -- by its whole magnitude, and every parameter gradient that sums
        # over those pixels by a little.  So the bounds are on a percentile and the relative L2 error, not on the maximum:
        # fp32 99.8th percentile 3e-3 of the largest entry, L2 1e-2; bf16 (eight layers of bf16 activations deep: mask
        # flips everywhere) 99th percentile 1.5e-1, L2 1e-1.  The sharp check of this operator is the next test: bit-equal
        # to the composition of the per-layer operators, which test_conv_gn_layer holds to 2e-5 one layer at a time.
        a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
        scale, err = np.abs(b).max(), np.abs(a - b)
        pct, q, l2 = (0.998, 3e-3, 1e-2) if dtype == torch.float32 else (0.99, 1.5e-1, 1e-1)
        assert np.quantile(err, pct) <= q * scale, (what, np.quantile(err, pct), scale)
        assert np.linalg.norm(err.ravel()) <= l2 * np.linalg.norm(b.ravel()), (what, "L2")
    check(y1.permute(0, 3, 1, 2), yr1, dtype, "y1")
    check(y2.permute(0, 3, 1, 2), yr, dtype, "y2")
    check(xd.grad.permute(0, 3, 1, 2), xr.grad, dtype, "dx")
    for i, (md, mr) in enumerate(zip(dev, blocks)):
        ref = dict(mr.named_parameters())
        for n, p in md.named_parameters():
            if n.startswith("downsample.0") or ref[n].grad is None:      # bn4 twice / unused without a downsample branch
                continue
            assert p.grad is not None, (i, n)
            check(p.grad, ref[n].grad, dtype, f"block{i}.{n}")
    # the statistics handed on are those of a fresh statistics pass over y1 (two-limb fixed-point sums of per-workgroup
    # fp32 partials in both: equal up to the grouping of those partials).  Layout (csrc/enc_common.h): two tables of
    # 16-byte cells, value = (table1.hi * 2^32 + table0.lo) * 2^-40
    def decode(t):
        w = t.cpu().numpy().view(np.uint64).reshape(2, -1, 2)
        return w[0, :, 0].astype(np.float64) + w[1, :, 1].view(np.int64).astype(np.float64) * 2.0 ** 32
    a, b = decode(s1), decode(ops.gn_stats(y1.detach())[:s1.numel()])
    assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max()


@pytest.mark.parametrize("cin,cout,shape", [(64, 128, (2, 24, 40)), (256, 256, (4, 64, 64))])
def test_conv_block_operator_equals_layerwise_composition(cin, cout, shape):
    """fp32: the fused ConvBlock operator runs the same kernels on the same operands as the composition from per-layer
    nodes + torch concat / adds (model/hgfilter_train._conv_block_layerwise) -- outputs and all gradients bit for bit"""
    import copy
    from chore_amd import ops
    from chore_amd.model import hgfilter_train as ht
    B, H, W = shape
    torch.manual_seed(7)
    m = _RefConvBlock(cin, cout)
    x = torch.randn(B, H, W, cin) * 1.5 + 0.3
    up = torch.randn(B, H, W, cout).cuda()
    res = []
    for mode in ("block", "layer"):
        md = copy.deepcopy(m).cuda()
        xd = x.clone().cuda().requires_grad_(True)
        y = ops.conv_block(xd, md)[0] if mode == "block" else ht._conv_block_layerwise(md, xd)
        (y * up).sum().backward()
        res.append([y.detach(), xd.grad] + [p.grad for _, p in sorted(md.named_parameters()) if p.grad is not None])
    assert len(res[0]) == len(res[1]) >= 11
    for a, b in zip(*res):
        assert torch.equal(a, b)

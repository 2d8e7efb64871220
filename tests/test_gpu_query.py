"""GPU parity tests of the fused query (forward, feature sample, backward-to-points).

Everything goes through the C-ABI (ctypes -> libchore_hip.so).  Expected values come from
  * the golden vectors the reference itself produced (tests/golden/, see make_golden.py), and
  * the numpy oracle (oracle/query.py), which is pinned bit for bit / to round-off by the same vectors.
Tolerances: projection, in_img, OUT_DIST mask and sampled features are BIT-EXACT; head outputs
|err| <= 3e-5 (fp32 accumulation order differs from BLAS); gradients rel. 2e-4 of their scale.
"""
import os

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import query as oq

pytestmark = pytest.mark.gpu


def nhwc(x, dtype=torch.float32):
    """numpy (B,C,H,W) -> cuda channels-last view with NCHW logical shape"""
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda().to(dtype)
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


@pytest.fixture(scope="module")
def net(opt):
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    opt.compute_dtype = "fp32"
    m = CHORE(opt).cuda().eval()
    synth.load_synth_weights(m, seed=0)
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def test_projection_and_features_bit_exact(net):
    from chore_amd.model.geometry import sample_features
    g = golden("query_proj.npz")
    gi = golden("query_index.npz")
    pts = torch.from_numpy(g["points"]).cuda()
    cc = torch.from_numpy(g["crop_center"]).cuda()
    feats, inside, nxy = sample_features(pts, cc, nhwc(gi["feat"]), nhwc(gi["tmpx"]), net._cam6, want_nxy=True)
    nxy = nxy.cpu().numpy()
    assert np.array_equal(nxy[..., 0].view(np.uint32), g["nx_bits"])
    assert np.array_equal(nxy[..., 1].view(np.uint32), g["ny_bits"])
    assert np.array_equal(inside.cpu().numpy(), g["in_img"])
    f = feats.cpu().numpy()
    # sampled values against the reference's own grid_sample outputs (first 512 points), bit for bit
    assert np.array_equal(f[:, :256, :512], gi["s_feat"])
    assert np.array_equal(f[:, 259:, :512], gi["s_tmpx"])
    # and against the oracle for all 2048 points incl. the z_feat channels
    nx, ny = oq.project_points(g["points"], g["crop_center"])
    assert np.array_equal(f[:, :256], oq.index(gi["feat"], nx, ny))
    assert np.array_equal(f[:, 259:], oq.index(gi["tmpx"], nx, ny))
    z = np.stack([g["points"][..., 0], g["points"][..., 1], g["points"][..., 2] - np.float32(2.2)], 1)
    assert np.array_equal(f[:, 256:259], z.astype(np.float32))


def run_query(net, g, requires_grad=False):
    net.im_feat_list = [nhwc(g["feat"])]
    net.tmpx = nhwc(g["tmpx"])
    pts = torch.from_numpy(g["points"]).cuda().requires_grad_(requires_grad)
    net.query(pts, crop_center=torch.from_numpy(g["crop_center"]).cuda())
    return pts, net.get_preds()


def test_query_forward_matches_reference(net, synth_sd):
    g = golden("query_full.npz")
    _, (df, pca, parts, centers) = run_query(net, g)
    assert pca.shape == (2, 3, 3, 300)
    got = dict(df=df, pca=pca, parts=parts, centers=centers)
    for k, v in got.items():
        np.testing.assert_allclose(v.cpu().numpy(), g[k], rtol=1e-5, atol=3e-5, err_msg=k)
    # OUT_DIST fill is exact and only where the reference has it
    assert np.array_equal(df.cpu().numpy() == 5.0, g["df"] == 5.0)
    o = oq.query(g["points"], g["crop_center"], g["feat"], g["tmpx"], synth_sd)
    for k, v in got.items():
        np.testing.assert_allclose(v.cpu().numpy(), o[k], rtol=1e-5, atol=3e-5, err_msg="oracle " + k)


def test_query_backward_to_points_matches_reference_autograd(net, synth_sd):
    g = golden("query_full.npz")
    pts, (df, pca, parts, centers) = run_query(net, g, requires_grad=True)
    loss = sum((o * torch.from_numpy(g["w_" + k]).cuda()).sum()
               for k, o in (("df", df), ("pca", pca), ("parts", parts), ("centers", centers)))
    loss.backward()
    got, ref = pts.grad.cpu().numpy(), g["dpoints"]
    scale = np.abs(ref).max()
    assert scale > 1.0
    # points sitting on a ReLU kink (|pre-activation| < 5e-6 somewhere in the 4 x 384 hidden units)
    # have an order-of-summation dependent gradient in ANY fp32 implementation; they are rare and
    # are excluded from the strict comparison
    o = oq.query(g["points"], g["crop_center"], g["feat"], g["tmpx"], synth_sd)
    stable = oq.relu_margin(o["features"], synth_sd) > 5e-6
    assert stable.mean() > 0.95
    err = np.abs(got - ref)[stable].max() / scale
    assert err < 2e-5, err
    assert np.isfinite(got).all()


@pytest.mark.parametrize("gscale", [1.0, 1e-7, 3e5])
def test_fp16x3_backward_to_points(net, synth_sd, gscale):
    """fp16 x 3 mode: the backward chain on the fp16 matrix cores (split operands, per-point column scales) against the
    reference's autograd gradients, with output gradients of ordinary, very small (a mean over many points) and very
    large magnitude -- the same relative bound as the native-fp32 kernel in all three"""
    g = golden("query_full.npz")
    net.compute_dtype = "fp16x3"
    try:
        pts, (df, pca, parts, centers) = run_query(net, g, requires_grad=True)
        loss = sum((o * torch.from_numpy(g["w_" + k]).cuda()).sum()
                   for k, o in (("df", df), ("pca", pca), ("parts", parts), ("centers", centers)))
        (loss * gscale).backward()
    finally:
        net.compute_dtype = "fp32"
    got, ref = pts.grad.cpu().numpy().astype(np.float64) / gscale, g["dpoints"]
    scale = np.abs(ref).max()
    o = oq.query(g["points"], g["crop_center"], g["feat"], g["tmpx"], synth_sd)
    stable = oq.relu_margin(o["features"], synth_sd) > 5e-6
    err = np.abs(got - ref)[stable].max() / scale
    assert err < 2e-5, err
    assert np.isfinite(got).all()


@pytest.mark.parametrize("B,N", [(1, 2048), (2, 333), (4, 20000)])
def test_fp16x3_backward_against_fp32_kernel(net, B, N):
    """both tile sizes, ragged counts, some heads without a gradient: fp16 x 3 against the fp32 chain on seeded maps"""
    from chore_amd.utils import synth
    rs = np.random.RandomState(23)
    g = dict(feat=rs.standard_normal((B, 256, 128, 128)).astype(np.float32),
             tmpx=rs.standard_normal((B, 64, 256, 256)).astype(np.float32),
             points=synth.synth_points(B, N, seed=4), crop_center=np.tile(np.array([synth.CROP_CENTER], np.float32), (B, 1)))
    wp = torch.from_numpy(rs.standard_normal((B, 14, N)).astype(np.float32)).cuda()

    def grads(mode):
        net.compute_dtype = mode
        try:
            pts, (df, pca, parts, centers) = run_query(net, g, requires_grad=True)
            (torch.clamp(df[:, 0], max=2.0).sum() * 1e-3 + (parts * wp).sum()).backward()      # no pca / centers gradient
        finally:
            net.compute_dtype = "fp32"
        return pts.grad
    a, b = grads("fp16x3"), grads("fp32")
    scale = float(b.abs().max())
    # per-point error relative to the largest gradient; kink points (sign of a near-zero pre-activation) may differ
    err = ((a - b).abs().amax(-1) / scale).flatten()
    assert float(err.quantile(0.99)) < 2e-5 and float((err > 1e-3).float().mean()) < 2e-3, (float(err.max()), float(err.quantile(0.99)))
    assert torch.isfinite(a).all()


def test_generator_style_gradient(net):
    """d(sum clamp(df_h, max=2))/d(points): the backward recon/generator.py:62-77 runs"""
    g = golden("query_full.npz")
    pts, (df, _, _, _) = run_query(net, g, requires_grad=True)
    torch.clamp(df[:, 0], max=2.0).sum().backward()
    grad = pts.grad.cpu().numpy()
    inside = oq.in_image(*oq.project_points(g["points"], g["crop_center"]))
    assert np.all(grad[~inside] == 0.0)  # OUT_DIST (5.0) is clamped AND cut from the graph
    assert np.abs(grad[inside]).max() > 0


@pytest.mark.parametrize("B,N", [(1, 2048), (4, 20000)])
def test_config_sizes_against_oracle(net, synth_sd, B, N):
    """BASELINE configs 1 and 2 (query part) on seeded 128x128 / 256x256 maps against the oracle"""
    from chore_amd.utils import synth
    rs = np.random.RandomState(21)
    feat = rs.standard_normal((B, 256, 128, 128)).astype(np.float32)
    tmpx = rs.standard_normal((B, 64, 256, 256)).astype(np.float32)
    pts = synth.synth_points(B, N, seed=1)
    cc = np.tile(np.array([synth.CROP_CENTER], np.float32), (B, 1))
    g = dict(feat=feat, tmpx=tmpx, points=pts, crop_center=cc)
    _, (df, pca, parts, centers) = run_query(net, g)
    o = oq.query(pts, cc, feat, tmpx, synth_sd)
    frac_in = o["in_img"].mean()
    assert 0.9 < frac_in < 0.99
    for k, v in dict(df=df, pca=pca, parts=parts, centers=centers).items():
        np.testing.assert_allclose(v.cpu().numpy(), o[k], rtol=1e-5, atol=5e-5, err_msg=k)
    # size-independent property: permuting the points permutes the outputs bit for bit
    perm = np.random.RandomState(5).permutation(N)
    g2 = dict(g, points=pts[:, perm])
    _, (df2, pca2, parts2, centers2) = run_query(net, g2)
    assert torch.equal(df2, df[:, :, perm]) and torch.equal(parts2, parts[:, :, perm])
    assert torch.equal(pca2, pca[..., perm]) and torch.equal(centers2, centers[:, :, perm])


def test_bf16_maps_query(net, synth_sd):
    """bf16 feature maps, fp32 heads: identical to querying the bf16-rounded maps in fp32"""
    g = golden("query_full.npz")
    fb = torch.from_numpy(g["feat"]).bfloat16().float().numpy()
    tb = torch.from_numpy(g["tmpx"]).bfloat16().float().numpy()
    net.compute_dtype = "bf16"
    try:
        net.im_feat_list = [nhwc(g["feat"], torch.bfloat16)]
        net.tmpx = nhwc(g["tmpx"], torch.bfloat16)
        pts = torch.from_numpy(g["points"]).cuda()
        net.query(pts, crop_center=torch.from_numpy(g["crop_center"]).cuda())
        df, pca, parts, centers = net.get_preds()
    finally:
        net.compute_dtype = "fp32"
    o = oq.query(g["points"], g["crop_center"], fb, tb, synth_sd)
    for k, v in dict(df=df, pca=pca, parts=parts, centers=centers).items():
        np.testing.assert_allclose(v.cpu().numpy(), o[k], rtol=1e-5, atol=3e-5, err_msg=k)


@pytest.mark.parametrize("B,N", [(1, 2048), (2, 333), (4, 20000)])
def test_fp16_maps_query_forward_and_backward(net, B, N):
    """fp16 feature maps ("fp16 fields"): values and the gradient to the points are those of the fp16 x 3 mode on the
    half-rounded maps held in fp32 -- the gather converts exactly, everything after it is the same arithmetic"""
    from chore_amd.utils import synth
    rs = np.random.RandomState(23)
    feat = torch.from_numpy(rs.standard_normal((B, 256, 128, 128)).astype(np.float32)).half()
    tmpx = torch.from_numpy(rs.standard_normal((B, 64, 256, 256)).astype(np.float32)).half()
    pts = synth.synth_points(B, N, seed=4)
    cc = torch.from_numpy(np.tile(np.array([synth.CROP_CENTER], np.float32), (B, 1))).cuda()
    res = {}
    for mode, tdt in (("fp16x3", torch.float32), ("fp16", torch.float16)):
        net.compute_dtype = mode
        try:
            net.im_feat_list = [nhwc(feat.float().numpy(), tdt)]
            net.tmpx = nhwc(tmpx.float().numpy(), tdt)
            p = torch.from_numpy(pts).cuda().requires_grad_(True)
            net.query(p, crop_center=cc)
            df, pca, parts, centers = net.get_preds()
            (torch.clamp(df, max=2.0).sum() + 0.3 * pca.sum() + 0.1 * parts.square().sum() + centers.sum()).backward()
            res[mode] = [t.detach().clone() for t in (df, pca, parts, centers, p.grad)]
        finally:
            net.compute_dtype = "fp32"
    for a, b in zip(res["fp16"], res["fp16x3"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,N", [(1, 2048), (2, 333), (4, 20000)])
def test_fp16x3_heads_forward(net, synth_sd, B, N):
    """the fp16 x 3 mode's query forward -- fp32 maps, the heads as three fp16 MFMAs per product on hi/lo split operands
    (csrc/heads_x3.h) -- against the oracle on seeded maps: 5e-5 absolute (the north-star bound is 1e-4), the OUT_DIST
    mask identical, for the 32-point-tile, the ragged and the 64-point-tile kernels"""
    from chore_amd.utils import synth
    rs = np.random.RandomState(22)
    feat = rs.standard_normal((B, 256, 128, 128)).astype(np.float32)
    tmpx = rs.standard_normal((B, 64, 256, 256)).astype(np.float32)
    pts = synth.synth_points(B, N, seed=3)
    cc = np.tile(np.array([synth.CROP_CENTER], np.float32), (B, 1))
    g = dict(feat=feat, tmpx=tmpx, points=pts, crop_center=cc)
    _, ref = run_query(net, g)
    net.compute_dtype = "fp16x3"
    try:
        _, out = run_query(net, g)
    finally:
        net.compute_dtype = "fp32"
    o = oq.query(pts, cc, feat, tmpx, synth_sd)
    for k, v, r in zip(("df", "pca", "parts", "centers"), out, ref):
        np.testing.assert_allclose(v.cpu().numpy().reshape(o[k].shape), o[k], rtol=1e-5, atol=5e-5, err_msg=k)
        assert float((v - r).abs().max()) < 5e-5, k
    assert torch.equal(out[0] == 5.0, ref[0] == 5.0)


def test_rejects_bad_inputs(net):
    g = golden("query_full.npz")
    net.im_feat_list = [torch.from_numpy(g["feat"]).cuda()]  # NCHW-contiguous, not NHWC
    net.tmpx = nhwc(g["tmpx"])
    with pytest.raises(ValueError):
        net.query(torch.from_numpy(g["points"]).cuda(), crop_center=torch.from_numpy(g["crop_center"]).cuda())


def test_fp16_mode_rejects_training_queries(opt):
    """'fp16' is an inference mode: with trainable heads (or maps that require grad) query() must refuse -- the training
    kernels read fp32 / bf16 maps and would read the half maps as fp32.  The C ABI refuses the map type
    too, with or without the heads flag."""
    import argparse
    from chore_amd import _lib
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    o = argparse.Namespace(**vars(opt))
    o.compute_dtype = "fp16"
    n = CHORE(o).cuda().eval()
    synth.load_synth_weights(n, seed=0)            # parameters keep requires_grad=True
    rs = np.random.RandomState(5)
    n.im_feat_list = [nhwc(rs.standard_normal((1, 256, 128, 128)).astype(np.float32), torch.float16)]
    n.tmpx = nhwc(rs.standard_normal((1, 64, 256, 256)).astype(np.float32), torch.float16)
    pts = torch.from_numpy(synth.synth_points(1, 256, seed=4)).cuda()
    cc = torch.from_numpy(np.array([synth.CROP_CENTER], np.float32)).cuda()
    with pytest.raises(NotImplementedError):
        n.query(pts, crop_center=cc)
    with torch.no_grad():                           # inference with the same (unfrozen) parameters is fine
        n.query(pts, crop_center=cc)
    assert torch.isfinite(n.get_preds()[0]).all()
    for p in n.parameters():
        p.requires_grad_(False)
    n.query(pts.clone().requires_grad_(True), crop_center=cc)      # frozen heads, gradient to the points: the inference path
    # the C ABI: fp16 maps are refused by the training entry points whatever the heads flag says
    h = _lib.handle(0)
    fp, FH, FW = n.im_feat_list[0].data_ptr(), 128, 128
    tp, TH, TW = n.tmpx.data_ptr(), 256, 256
    out = [torch.empty(1, c, 256, device="cuda") for c in (2, 9, 14, 6)]
    inside = torch.empty(1, 256, dtype=torch.uint8, device="cuda")
    staging = torch.empty(_lib.lib.chore_query_train_bytes(1, 256), dtype=torch.uint8, device="cuda")
    for dt in (_lib.F16, _lib.F16 | _lib.HEADS_X3):
        rc = _lib.lib.chore_query_fwd_train(h, pts.data_ptr(), cc.data_ptr(), 1, 256, fp, FH, FW, tp, TH, TW, dt,
                                            n._heads_arena(pts.device).data_ptr(), n._cam6, out[0].data_ptr(), out[1].data_ptr(),
                                            out[2].data_ptr(), out[3].data_ptr(), inside.data_ptr(), staging.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream)
        assert rc != 0, hex(dt)


@pytest.mark.parametrize("heads_x3,gscale", [(False, 1.0), (True, 1.0), (True, 1e-7), (True, 3e5)])
def test_training_backward_heads_and_feature_maps(opt, heads_x3, gscale):
    """heads_x3: the GEMM chain of the heads AND their weight gradients on the fp16 matrix cores with split operands
    (what the bf16 training mode runs) instead of the native fp32 MFMA -- same fixture, same bounds, also with the loss
    scaled down to where an unscaled fp16 operand would be all subnormal, and up (per-point / per-chunk scales).
    first half of the training backward: gradients of a random linear functional of the four outputs
    w.r.t. the 32 head parameters, the hourglass feature map and tmpx, against the reference's autograd
    (tests/golden/query_train_grads.npz: full tensors for the df head, the small tensors and the two maps; sums,
    abs-sums, L2 norms and a 16x24 crop for the large matrices of the other heads).  fp32 mode, 2e-5 relative."""
    import copy
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    g, gg = golden("query_full.npz"), golden("query_train_grads.npz")
    o = copy.copy(opt)
    o.compute_dtype = "fp32"
    net = CHORE(o).cuda().eval()
    net.heads_x3 = heads_x3
    synth.load_synth_weights(net, seed=0)
    for p in net.image_filter.parameters():
        p.requires_grad_(False)
    feat = nhwc(g["feat"]).requires_grad_(True)
    tmpx = nhwc(g["tmpx"]).requires_grad_(True)
    net.im_feat_list, net.tmpx = [feat], tmpx
    pts = torch.from_numpy(g["points"]).cuda().requires_grad_(True)
    net.query(pts, crop_center=torch.from_numpy(g["crop_center"]).cuda())
    preds = net.get_preds()
    for k, v in zip(("df", "pca", "parts", "centers"), preds):
        assert np.abs(v.detach().cpu().numpy() - g[k]).max() < 2e-5
    # the functional leaves out the ReLU-kink points (mask stored with the fixture, see make_golden.gen_query_train)
    st = torch.from_numpy(gg["stable"]).cuda()
    loss = sum((o_ * torch.from_numpy(g["w_" + k]).cuda() * st.view(st.shape[0], *([1] * (o_.dim() - 2)), -1)).sum()
               for k, o_ in zip(("df", "pca", "parts", "centers"), preds))
    (loss * gscale).backward()

    def close(a, b, what, tol=2e-5):
        a = a.detach().double().cpu().numpy().reshape(b.shape) / gscale
        assert np.abs(a - b).max() <= tol * max(1e-6, np.abs(b).max()), (what, np.abs(a - b).max(), np.abs(b).max())

    stable = gg["stable"] > 0
    assert stable.mean() > 0.95
    close(feat.grad, gg["dfeat"], "dfeat")
    close(tmpx.grad, gg["dtmpx"], "dtmpx")
    dp = pts.grad.cpu().numpy()
    assert np.abs(dp[~stable]).max() == 0          # no functional, no gradient
    heads = {"df": net.df, "part_predictor": net.part_predictor, "pca_predictor": net.pca_predictor,
             "center_predictor": net.center_predictor}
    n_checked = 0
    for hn, m in heads.items():
        for k, p in m.named_parameters():
            name = f"{hn}.{k}"
            gr = p.grad
            assert gr is not None and torch.isfinite(gr).all(), name
            if "g_" + name in gg.files:
                close(gr, gg["g_" + name], name, tol=5e-5)
            else:
                a = gr.detach().cpu().numpy().astype(np.float64) / gscale
                ref = gg["s_" + name]
                got = np.array([a.sum(), np.abs(a).sum(), np.sqrt((a ** 2).sum())])
                assert np.abs(got - ref).max() < 5e-5 * ref[1], (name, got, ref)
                close(gr.reshape(gr.shape[0], -1)[:16, :24], gg["c_" + name], name + " crop", tol=5e-5)
            n_checked += 1
    assert n_checked == 32


@pytest.mark.parametrize("N", [1, 63, 65, 257])
def test_ragged_point_counts(net, synth_sd, N):
    """point counts that do not fill the 64-point tiles of the kernels: forward against the oracle, backward finite and
    equal to the same points' gradients inside a full batch (tile padding must not leak)"""
    g = golden("query_full.npz")
    sub = dict(g, points=g["points"][:, :N].copy())
    pts, (df, pca, parts, centers) = run_query(net, sub, requires_grad=True)
    o = oq.query(sub["points"], g["crop_center"], g["feat"], g["tmpx"], synth_sd)
    for k, v in dict(df=df, pca=pca, parts=parts, centers=centers).items():
        assert v.shape[-1] == N
        np.testing.assert_allclose(v.detach().cpu().numpy().reshape(o[k].shape), o[k], rtol=1e-5, atol=5e-5, err_msg=k)
    (torch.clamp(df[:, 1], max=2.0).sum() + parts.sum() * 0.01).backward()
    full, (df_f, _, parts_f, _) = run_query(net, g, requires_grad=True)
    (torch.clamp(df_f[:, 1, :N], max=2.0).sum() + parts_f[:, :, :N].sum() * 0.01).backward()
    assert torch.equal(pts.grad, full.grad[:, :N])


def test_empty_query(net):
    """zero points: empty predictions of the right shapes (what the reference's torch ops return), no launch"""
    g = golden("query_full.npz")
    net.im_feat_list = [nhwc(g["feat"])]
    net.tmpx = nhwc(g["tmpx"])
    B = g["points"].shape[0]
    net.query(torch.zeros(B, 0, 3).cuda(), crop_center=torch.from_numpy(g["crop_center"]).cuda())
    df, pca, parts, centers = net.get_preds()
    assert df.shape == (B, 2, 0) and pca.shape == (B, 3, 3, 0) and parts.shape == (B, 14, 0) and centers.shape == (B, 6, 0)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_skipped_heads_change_nothing(opt, mode):
    """chore_query_fwd with NULL outputs (CHORE.query_df) and chore_query_bwd_points with NULL upstream gradients: the heads
    that are not asked for are not evaluated, what is asked for is bit-identical to the full evaluation (forward) / to the
    evaluation with explicit zero gradients (backward), for the 32- and the 64-point tiles"""
    import copy
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    o = copy.copy(opt)
    o.compute_dtype = mode
    net = CHORE(o).cuda().eval()
    synth.load_synth_weights(net, seed=0)
    for B, N in ((1, 3000), (2, 20000)):
        with torch.no_grad():
            net.filter(torch.from_numpy(synth.synth_images(B, 128, 128, 0)).cuda())
        cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
        pts = torch.from_numpy(synth.synth_points(B, N, seed=2)).cuda()
        with torch.no_grad():
            net.query(pts, crop_center=cc)
        df, pca, parts, centers = net.get_preds()
        assert torch.equal(net.query_df(pts, cc), df)
        g = torch.from_numpy(np.random.RandomState(0).standard_normal((B, 2, N)).astype(np.float32)).cuda()
        only = net.query_grad_points(pts, cc, g_df=g)
        zeros = net.query_grad_points(pts, cc, g_df=g, g_pca=torch.zeros_like(pca), g_parts=torch.zeros_like(parts),
                                      g_centers=torch.zeros_like(centers))
        assert torch.isfinite(only).all() and float(only.abs().max()) > 0
        assert torch.equal(only, zeros)


def test_split_forward_kernel_equals_the_one_wave_per_head_kernels_bit_for_bit():
    """fp16 x 3 forward: the two-waves-per-head kernel (query_fwd_x3_split_kernel: input planes split into fp16 hi / lo at
    gather time, activations exchanged through the LDS) against the one-wave-per-head kernels (CHORE_QUERY_X3_NOSPLIT=1, the
    arithmetic the backward / surface-step kernels recompute): all four outputs, fp32 / bf16 / fp16 maps, 32- and 64-point
    tiles, ragged tails -- identical bits (scripts/query_split_equal.py runs both in child processes: the switch is read once)"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "scripts", "query_split_equal.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if " equal " in ln]
    assert len(lines) == 36 and all(" equal True" in ln for ln in lines), "\n".join(ln for ln in lines if "True" not in ln)


def test_fp16x3_heads_stay_finite_beyond_the_half_range(opt):
    """an input feature beyond the IEEE-half range (|x| >= 65 520) used to become inf in the fp16 hi / lo split of the fp16 x 3
    heads and NaN one MFMA later, where the fp32-MFMA path stays finite.  The split-operand kernels run with
    MODE.FP16_OVFL set (csrc/common.h f16_saturate_mode): the pair saturates instead -- |x| <= 131 008 is still represented
    exactly (hi = 65 504, lo = the rest).  Feature maps scaled to +-1e5: forward and backward to the points finite, and within
    1e-3 of the fp32-MFMA heads' largest output (hidden activations beyond 131 008 saturate: this is a guard against NaN, not
    a range extension)."""
    import copy
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    g = golden("query_full.npz")
    res = {}
    for x3 in (True, False):
        o = copy.copy(opt)
        o.compute_dtype = "fp32"
        net = CHORE(o).cuda().eval()
        synth.load_synth_weights(net, seed=0)
        for p in net.parameters():
            p.requires_grad_(False)
        feat = nhwc(g["feat"])
        feat = feat * (1.0e5 / float(feat.abs().max()))
        assert float(feat.abs().max()) > 9.9e4
        net.im_feat_list, net.tmpx = [feat], nhwc(g["tmpx"])
        pts = torch.from_numpy(g["points"]).cuda().requires_grad_(True)
        cc = torch.from_numpy(g["crop_center"]).cuda()
        if x3:
            os.environ.pop("CHORE_HEADS_FP32", None)
        else:
            os.environ["CHORE_HEADS_FP32"] = "1"
        try:
            net.compute_dtype = "fp16x3" if x3 else "fp32"
            net.query(pts, crop_center=cc)
            preds = [p.detach().clone() for p in net.get_preds()]
            (gp,) = torch.autograd.grad(net.get_preds()[0][:, 0].sum(), pts)
        finally:
            os.environ.pop("CHORE_HEADS_FP32", None)
        assert all(torch.isfinite(p).all() for p in preds) and torch.isfinite(gp).all(), "x3" if x3 else "fp32"
        res[x3] = preds
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max()) <= 1e-3 * max(1.0, float(b.abs().max()))


def _grid_sample_scatter_reference(points, cc, dX, maps, xoffs):
    """the transpose of `index` as the reference's autograd computes it: the backward of F.grid_sample(bilinear, zeros,
    align_corners=True) at the reference's own projection (model/camera.py through chore_amd's bit-exact restatement), fp64 on the
    CPU -- (B,H,W,C) gradients of both maps"""
    import torch.nn.functional as F
    from chore_amd.model.camera import KinectColorCamera
    cam = KinectColorCamera(1200)
    xy = cam.project_points(points.cpu(), cc.cpu())[:, :2].double()          # (B,2,N) normalised image coordinates
    out = []
    for (H, W, C), xo in zip(maps, xoffs):
        m = torch.zeros(points.shape[0], C, H, W, dtype=torch.float64, requires_grad=True)
        samp = F.grid_sample(m, xy.permute(0, 2, 1).unsqueeze(2), mode="bilinear", padding_mode="zeros", align_corners=True)   # (B,C,N,1)
        g = dX[:, :, xo:xo + C].permute(0, 2, 1).unsqueeze(-1).double().cpu()
        (samp * g).sum().backward()
        out.append(m.grad.permute(0, 2, 3, 1))
    return out


@pytest.mark.parametrize("B,N,maps,spread", [(4, 20000, (128, 128, 256, 256), "uniform"), (2, 5000, (128, 128, 256, 256), "clustered"),
                                               (3, 2500, (16, 24, 32, 48), "uniform"), (2, 100, (16, 24, 32, 48), "uniform"),
                                               (2, 37, (16, 24, 32, 48), "uniform"), (1, 1, (128, 128, 256, 256), "uniform"),
                                               (1, 9000, (136, 200, 272, 400), "uniform"), (1, 70001, (128, 128, 256, 256), "uniform")])
def test_scatter_features_any_size_and_window(B, N, maps, spread, monkeypatch):
    """chore_scatter_features (the transpose of `index`, model/geometry.py:4-14, for the training query).  Round 5: the binned
    path -- a stable per-chunk counting sort of the points by map tile, then every tile walks only its own points, in point
    order -- is the ONLY path: fewer than 64 points (rounds 1 - 4: the old scan kernel), one point, and maps of more than 256
    tiles (136 x 200 texels = 17 x 25 tiles: windows of 16 x 16 tiles, each its own sort + walk).  Checked (a) against the
    backward of F.grid_sample in fp64 on the CPU (2e-5 of the largest entry: the order of the fp32 sums is the only difference)
    and (b) BIT FOR BIT against the same call with 3 x 3-tile windows (CHORE_SCATTER_WINDOW=3: a tile's list does not depend
    on the window it is sorted in) -- random gradient rows, thousands of points in one tile (several rounds of the 512-entry hit
    list), partial tiles, points far outside the image, a point count that is no multiple of anything."""
    import ctypes
    from chore_amd import _lib
    from chore_amd.utils import synth
    from chore_amd.model.camera import KinectColorCamera
    dev = torch.device("cuda", 0)
    FH, FW, TH, TW = maps
    rs = np.random.RandomState(B * 1000 + N)
    pts = synth.synth_points(B, N, seed=3)
    if spread == "clustered":                      # 80 % of the points inside a few centimetres: one or two tiles get thousands
        k = int(0.8 * N)
        pts[:, :k] = pts[:, :1] + rs.standard_normal((B, k, 3)).astype(np.float32) * 0.02
    if N > 3:
        pts[:, -3:] = [[50.0, 50.0, 2.0]]          # far outside the image: no tap anywhere
    points = torch.from_numpy(pts).to(dev)
    cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)
    nbytes = _lib.lib.chore_query_train_bytes(B, N)
    P, KPAD = B * N, 328
    staging = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    f = staging[:(nbytes // 4) * 4].view(torch.float32)
    o = P * (KPAD + 2 * 3 * 4 * 128)              # X, H, dZ come first (csrc/capi.hip train_staging)
    dX = torch.from_numpy(rs.standard_normal(P * KPAD).astype(np.float32)).to(dev)
    f[o:o + P * KPAD] = dX
    cam6 = (ctypes.c_float * 6)(*KinectColorCamera(1200).kernel_constants())
    h = _lib.handle(0)
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for kind in ("windows of 3", "default"):
        if kind == "default":
            monkeypatch.delenv("CHORE_SCATTER_WINDOW", raising=False)
        else:
            monkeypatch.setenv("CHORE_SCATTER_WINDOW", "3")
        dfe = torch.full((B, FH, FW, 256), float("nan"), device=dev)
        dtm = torch.full((B, TH, TW, 64), float("nan"), device=dev)
        _lib.check(_lib.lib.chore_scatter_features(h, points.data_ptr(), cc.data_ptr(), B, N, FH, FW, TH, TW, cam6, staging.data_ptr(),
                                                   dfe.data_ptr(), dtm.data_ptr(), 0, stream), h, "chore_scatter_features")
        torch.cuda.synchronize()
        out[kind] = (dfe, dtm)
    for a, b, name in zip(out["windows of 3"], out["default"], ("dfeat", "dtmpx")):
        assert torch.isfinite(a).all(), name
        assert torch.equal(a, b), (name, int((a != b).sum()))
    if B * N <= 20000:                             # the fp64 CPU reference (seconds for the small cases)
        ref = _grid_sample_scatter_reference(points, cc, dX.view(B, N, KPAD), ((FH, FW, 256), (TH, TW, 64)), (0, 259))
        for got, want, name in zip(out["default"], ref, ("dfeat", "dtmpx")):
            err = float((got.double().cpu() - want).abs().max())
            assert float(want.abs().max()) > 0 or N < 4
            assert err <= 2e-5 * max(1.0, float(want.abs().max())), (name, err)


@pytest.mark.parametrize("B,N,spread", [(4, 20000, "uniform"), (2, 9001, "clustered"), (1, 70001, "uniform")])
def test_sorted_order_forward_equals_the_unsorted_one_bit_for_bit(B, N, spread, opt, monkeypatch):
    """chore_query_fwd_ws (round 4): the points are ordered by the map tile their sample falls into before the tiles of 64 points
    are formed (CHORE.query for >= 8 192 points, model/chore.py:107-154).  Every point's arithmetic is untouched and its outputs go
    to its own column: all four predictions and in_img equal the unsorted forward's (no workspace) BIT FOR BIT -- random
    points, points piled into one tile, points far outside the image, a point count that is no multiple of anything."""
    import ctypes
    from chore_amd import _lib
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    dev = torch.device("cuda", 0)
    opt.compute_dtype = "fp16x3"
    net = CHORE(opt).to(dev).eval()
    synth.load_synth_weights(net, seed=0)
    rs = np.random.RandomState(N)
    pts = synth.synth_points(B, N, seed=5)
    if spread == "clustered":
        k = int(0.7 * N)
        pts[:, :k] = pts[:, :1] + rs.standard_normal((B, k, 3)).astype(np.float32) * 0.01
    pts[:, -5:] = [[40.0, -40.0, 2.0]]
    points = torch.from_numpy(pts).to(dev)
    cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)
    feat = torch.from_numpy(rs.standard_normal((B, 128, 128, 256)).astype(np.float32)).to(dev)
    tmpx = torch.from_numpy(rs.standard_normal((B, 256, 256, 64)).astype(np.float32)).to(dev)
    arena = net._heads_arena(dev)
    h = _lib.handle(0)
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(_lib.lib.chore_query_fwd_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
    out = {}
    for kind in ("unsorted", "sorted"):
        t = [torch.full((B, c, N), float("nan"), device=dev) for c in (2, 9, 14, 6)]
        inimg = torch.full((B, N), 7, dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.chore_query_fwd_ws(h, points.data_ptr(), cc.data_ptr(), B, N, feat.data_ptr(), 128, 128, tmpx.data_ptr(), 256,
                                               256, _lib.F16X3, arena.data_ptr(), net._cam6, t[0].data_ptr(), t[1].data_ptr(),
                                               t[2].data_ptr(), t[3].data_ptr(), inimg.data_ptr(), ws.data_ptr() if kind == "sorted" else None, stream), h, "fwd_ws")
        torch.cuda.synchronize()
        out[kind] = t + [inimg]
    perm = ws.view(torch.int32)[:N].long().cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(N))
    for a, b, name in zip(out["unsorted"], out["sorted"], ("df", "pca", "parts", "centers", "in_img")):
        assert torch.isfinite(a.float()).all(), name
        assert torch.equal(a, b), (name, int((a != b).sum()))

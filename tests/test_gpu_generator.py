"""GPU: Alg. 1 surface projection (Generator.approx_surface / gen_pc_batch) on the HIP query kernels."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import query as oq
from test_gpu_query import nhwc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen(opt):
    from chore_amd.model import CHORE
    from chore_amd.recon.generator import Generator
    from chore_amd.utils import synth
    opt.compute_dtype = "fp32"
    m = CHORE(opt).cuda().eval()
    synth.load_synth_weights(m, seed=0)
    return Generator(m, None, threshold=2.0, filter_val=0.004, device=torch.device("cuda"))


def test_approx_surface_matches_reference_trajectory(gen, synth_sd):
    """3 projection steps of the reference's Generator.approx_surface (tests/golden/surface_steps.npz).
    Step 1 must agree to fp32 round-off wherever the point is not on a ReLU kink; later steps compound
    (each step moves a point by up to 2 m across a piecewise-linear random field), so they are checked on
    the fraction of points that still track the reference."""
    g = golden("query_full.npz")
    t = golden("surface_steps.npz")
    gen.model.im_feat_list = [nhwc(g["feat"])]
    gen.model.tmpx = nhwc(g["tmpx"])
    q = {"crop_center": torch.from_numpy(g["crop_center"]).cuda()}
    o = oq.query(g["points"], g["crop_center"], g["feat"], g["tmpx"], synth_sd)
    stable = oq.relu_margin(o["features"], synth_sd) > 5e-6
    for name in ("human", "object"):
        pts = torch.from_numpy(g["points"]).cuda().requires_grad_(True)
        ref = t["traj_" + name]
        for step in range(3):
            pts, preds = gen.approx_surface(gen.model, pts, 1, q, name)
            err = np.abs(pts.detach().cpu().numpy() - ref[step]).max(-1)
            if step == 0:
                assert err[stable].max() < 2e-5, (name, err[stable].max())
                outside = ~oq.in_image(*oq.project_points(g["points"], g["crop_center"]))
                assert np.all(pts.detach().cpu().numpy()[outside] == g["points"][outside])  # no gradient: stay put
            assert (err < 1e-3).mean() > 0.9, (name, step, (err < 1e-3).mean())
        same = (preds[0].detach().cpu().numpy() == 5.0) == (t["df_" + name] == 5.0)
        assert same.mean() > 0.97


def test_gen_pc_batch_end_to_end(gen):
    """full loop with a field whose 'surface' is easy to hit: filter_val large enough to collect points"""
    from chore_amd.utils import synth
    rs = np.random.RandomState(3)
    B = 2
    gen.model.im_feat_list = [nhwc(rs.standard_normal((B, 256, 16, 16)).astype(np.float32))]
    gen.model.tmpx = nhwc(rs.standard_normal((B, 64, 32, 32)).astype(np.float32))
    old = gen.filter_val
    gen.filter_val = 1.0
    try:
        init = torch.from_numpy(synth.synth_points(B, 3000, seed=4)).cuda()
        batch = {"crop_center": torch.tensor([synth.CROP_CENTER] * B)}
        out = gen.gen_pc_batch(gen.model, "human", init, 500, batch, num_steps=3, mute=True)
    finally:
        gen.filter_val = old
    n = out["points"].shape[1]
    assert n >= 500 and out["points"].shape == (B, n, 3)
    assert out["parts"].shape == (B, n) and out["parts"].dtype == torch.int64
    assert out["pca_axis"].shape == (B, 3, 3) and out["centers"].shape == (B, 6)
    assert torch.isfinite(out["points"]).all()


def test_device_loop_equals_host_loop(gen):
    """the device-resident loop (csrc/generator.hip: ordered compaction, append, resample) reproduces the reference's
    host-side control flow (boolean indexing, Python lists) when both consume the same uniform / normal draws"""
    from chore_amd.utils import synth
    rs = np.random.RandomState(5)
    B = 2
    gen.model.im_feat_list = [nhwc(rs.standard_normal((B, 256, 16, 16)).astype(np.float32))]
    gen.model.tmpx = nhwc(rs.standard_normal((B, 64, 32, 32)).astype(np.float32))
    init = torch.from_numpy(synth.synth_points(B, 3000, seed=6)).cuda()
    batch = {"crop_center": torch.tensor([synth.CROP_CENTER] * B)}
    g = torch.Generator().manual_seed(77)
    U = [torch.rand(B, 20000, generator=g).cuda() for _ in range(12)]
    Z = [torch.randn(B, 20000, 3, generator=g).cuda() for _ in range(12)]

    def device_hooks():
        st = {"u": 0, "z": 0}

        def uniform(shape):
            st["u"] += 1
            return U[st["u"] - 1]

        def randn(shape):
            st["z"] += 1
            return Z[st["z"] - 1]
        return uniform, randn

    def host_hooks():
        st = {"u": 0, "z": 0}     # the host loop draws per example: example i of round r uses row i of U[r] / Z[r]

        def randint(high, n):
            r, i = divmod(st["u"], B)
            st["u"] += 1
            return torch.clamp(torch.floor(U[r][i] * float(high)), max=high - 1).long()

        def randn(shape):
            r, i = divmod(st["z"], B)
            st["z"] += 1
            return Z[r][i].view(shape)
        return randint, randn

    old = gen.filter_val
    gen.filter_val = 1.0
    try:
        dev = gen.gen_pc_batch(gen.model, "object", init, 700, batch, num_steps=2, mute=True, rng=device_hooks())
        host = gen.gen_pc_batch(gen.model, "object", init, 700, batch, num_steps=2, mute=True, rng=host_hooks(),
                                device_loop=False)
    finally:
        gen.filter_val = old
    assert dev["points"].shape == host["points"].shape and dev["points"].shape[1] >= 700
    assert torch.equal(dev["points"], host["points"])
    assert torch.equal(dev["parts"], host["parts"])
    assert (dev["pca_axis"] - host["pca_axis"]).abs().max() < 1e-6
    assert (dev["centers"] - host["centers"]).abs().max() < 1e-6


@pytest.mark.parametrize("name", ["human", "object"])
def test_approx_surface_direct_path_equals_autograd_path(gen, name, monkeypatch):
    """the four-launch step (chore_gen_clamp_mask / chore_query_bwd_points / chore_gen_surface_step) against the tensor
    expressions + autograd of generator.py:50-79 on the same kernels: one step, identical field values, points to 1e-6
    relative of the step length (division and norm are the only operations whose rounding may differ)"""
    g = golden("query_full.npz")
    gen.model.im_feat_list = [nhwc(g["feat"])]
    gen.model.tmpx = nhwc(g["tmpx"])
    q = {"crop_center": torch.from_numpy(g["crop_center"]).cuda()}
    pts = torch.from_numpy(g["points"]).cuda()
    monkeypatch.setenv("CHORE_GEN_AUTOGRAD", "1")
    ref, ref_preds = gen.approx_surface(gen.model, pts.clone().requires_grad_(True), 1, q, name)
    monkeypatch.delenv("CHORE_GEN_AUTOGRAD")
    got, got_preds = gen.approx_surface(gen.model, pts.clone().requires_grad_(True), 1, q, name)
    assert got.requires_grad
    for a, b in zip(got_preds, ref_preds):
        assert torch.equal(a, b)
    step = (ref.detach() - pts).norm(dim=-1).max()
    assert float((got.detach() - ref.detach()).abs().max()) <= 1e-6 * float(step)


@pytest.mark.parametrize("mode,name", [("bf16", "human"), ("fp16x3", "object"), ("fp16", "human")])
def test_fused_surface_step_equals_the_four_launches(opt, mode, name, monkeypatch):
    """chore_gen_surface_step_fused (forward of the distance head, its own upstream gradient, backward, projection in the
    backward-to-points kernel) against chore_query_fwd -> chore_gen_clamp_mask -> chore_query_bwd_points ->
    chore_gen_surface_step: the same bits after 3 steps, for the 32- and the 64-point tiles, points outside the image included"""
    import copy
    from chore_amd.model import CHORE
    from chore_amd.recon.generator import Generator
    from chore_amd.utils import synth
    o = copy.copy(opt)
    o.compute_dtype = mode
    net = CHORE(o).cuda().eval()
    synth.load_synth_weights(net, seed=0)
    gen2 = Generator(net, None, threshold=2.0, filter_val=0.004, device=torch.device("cuda"))
    for B, N in ((1, 3000), (2, 20000)):
        with torch.no_grad():
            net.filter(torch.from_numpy(synth.synth_images(B, 128, 128, 0)).cuda())
        q = {"crop_center": torch.tensor([synth.CROP_CENTER] * B).cuda()}
        pts = gen2.init_samples(N, B)          # frames 1.. keep the unit cube: mostly outside the image
        assert net.surface_step(pts, q["crop_center"], 0, 2.0) is not None
        monkeypatch.setenv("CHORE_GEN_FOUR_LAUNCHES", "1")
        ref, ref_preds = gen2.approx_surface(net, pts.clone(), 4, q, name)
        monkeypatch.delenv("CHORE_GEN_FOUR_LAUNCHES")
        got, got_preds = gen2.approx_surface(net, pts.clone(), 4, q, name)
        assert torch.isfinite(got).all() and float((got.detach() - pts).abs().max()) > 1e-3
        assert torch.equal(got.detach(), ref.detach())
        for a, b in zip(got_preds, ref_preds):
            assert torch.equal(a, b)


ONE_HEAD_DUMP = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + "/tests")
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.recon.generator import Generator
from chore_amd.utils import synth
out = {{}}
torch.manual_seed(11)
for mode in ("fp16x3", "fp16", "bf16"):
    net = CHORE(chore_opt(mode)).cuda().eval(); synth.load_synth_weights(net, 0)
    gen = Generator(net, None, threshold=2.0, filter_val=0.004, device=torch.device("cuda"))
    B, N = 2, 20000
    with torch.no_grad():
        net.filter(torch.from_numpy(synth.synth_images(B, 128, 128, 0)).cuda())
    q = {{"crop_center": torch.tensor([synth.CROP_CENTER] * B).cuda()}}
    pts = gen.init_samples(N, B)
    s, preds = gen.approx_surface(net, pts.clone(), 4, q, "object")
    out[mode + "_surf"] = s.detach().cpu().numpy()
    g = torch.randn(B, 14, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    out[mode + "_gparts"] = net.query_grad_points(pts, q["crop_center"], g_parts=g).cpu().numpy()     # one gradient, not the distance head's
np.savez({path!r}, **out)
"""


def test_one_head_two_wave_backward_equals_the_one_wave_chain(tmp_path):
    """round 6: with exactly one upstream gradient (the surface step; a backward with only g_parts) the head's chain runs on two
    waves, a 32-point column block each (csrc/query_bwd.hip ONE), instead of one wave with three idle beside it.  Same products
    in the same order per element: the moved points after 4 steps and a parts-only gradient are EQUAL BIT FOR BIT to the
    one-wave kernels (CHORE_QUERY_NO_ONE_HEAD=1), in the fp16x3, fp16-fields and bf16 modes."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for tag, env in (("one", {}), ("old", {"CHORE_QUERY_NO_ONE_HEAD": "1"})):
        path = str(tmp_path / ("one_head_%s.npz" % tag))
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", ONE_HEAD_DUMP.format(repo=repo, path=path)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(dict(np.load(path)))
    a, b = res
    assert set(a) == set(b) and len(a) == 6
    for k in a:
        assert np.isfinite(a[k]).all() and np.abs(a[k]).max() > 0
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (k, np.abs(a[k] - b[k]).max())

"""GPU: conv_mw_kernel (csrc/conv_mw.hip, round 6) -- the 3x3 layers of ConvBlock (model/net_util.py:346-396) with the staging work
inside the MFMA-issuing waves -- against conv_pc_kernel, the specialised-wave kernel it replaces.

The kernel choice is an environment switch the library reads once per process (CHORE_CONV_MW=0: conv_pc_kernel everywhere, =all:
conv_mw_kernel on every tiling it has), so each variant runs in a process of its own and the results are compared here:
  * single layers through chore_conv2d_fwd at every tiling conv_pc_plan has (8 x 32 x 128 / 64 / 32, 4 x 32 x 64 / 32; CHORE_CONV_MW_FILL=256
    keeps conv_mw_plan on them), whole and ragged maps: the two
    kernels add the same products to the same accumulators in the same order, so the OUTPUT must be equal BIT FOR BIT; the GroupNorm
    statistics of the output are reduced from different partial sums (256 instead of 512 threads): equal to fp32 summation order;
  * the whole encoder (every stack's feature map, tmpx, normx), residual and raw-copy paths included: the statistics' last bits move
    every later layer's input, so the bound is the summation-order one of tests/test_gpu_conv_rw.py (5e-6 of the largest entry)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LAYERS = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r})
from chore_amd import _lib
dev = torch.device("cuda", 0); h = _lib.handle(0); dt = _lib.F16X3
stream = torch.cuda.current_stream().cuda_stream
out = {{}}
# (Cin, Cout, B, H, W): every tiling of conv_pc_plan, maps that end inside a tile, one and many chunks
for n, (cin, cout, B, H, W) in enumerate([(256, 128, 4, 128, 128), (256, 128, 2, 40, 56), (128, 64, 4, 128, 128), (64, 64, 4, 128, 128),
                                          (64, 32, 2, 128, 128), (32, 32, 2, 256, 256), (256, 128, 4, 64, 64), (128, 64, 4, 64, 64),
                                          (64, 64, 4, 64, 64), (128, 64, 3, 20, 28), (64, 64, 3, 20, 28), (128, 128, 2, 64, 64)]):
    g = torch.Generator(device=dev); g.manual_seed(100 + n)
    x = torch.randn(B, H, W, cin, device=dev, generator=g) * 1.5 + 0.3
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * (1.0 / np.sqrt(cin * 9))
    ga, be = torch.rand(cin, device=dev, generator=g) + 0.5, torch.randn(cin, device=dev, generator=g) * 0.2
    st = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.chore_gn_stats(h, _lib.F32, x.data_ptr(), B, H * W, cin, st.data_ptr(), 1, stream), h, "stats")
    y = torch.full((B, H, W, cout), 7.0, device=dev)
    sty = torch.zeros(_lib.lib.chore_gn_stats_bytes(B), dtype=torch.uint8, device=dev)
    ws = torch.empty(max(16, _lib.lib.chore_conv2d_workspace_bytes(dt, 9, cin, cout)), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.chore_conv2d_fwd(h, dt, 9, x.data_ptr(), B, H, W, cin, st.data_ptr(), ga.data_ptr(), be.data_ptr(), w.data_ptr(),
                                         None, cout, y.data_ptr(), sty.data_ptr(), ws.data_ptr(), stream), h, "conv")
    torch.cuda.synchronize()
    out["y%d" % n] = y.cpu().numpy()
    out["s%d" % n] = sty.cpu().numpy().view(np.int64)
np.savez({path!r}, **out)
"""

ENCODER = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + "/tests")
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
net = CHORE(chore_opt("fp16x3")).cuda().eval(); synth.load_synth_weights(net, 0)
for p in net.parameters(): p.requires_grad_(False)
img = torch.from_numpy(synth.synth_images({B}, {H}, {W}, 5)).cuda()
with torch.no_grad():
    net.filter(img)
out = dict(("f%d" % i, o.float().cpu().numpy()) for i, o in enumerate(net.im_feat_list))
out["tmpx"] = net.tmpx.float().cpu().numpy(); out["normx"] = net.normx.float().cpu().numpy()
np.savez({path!r}, **out)
"""


def run(tmp_path, script, tag, env, **kw):
    path = str(tmp_path / ("mw_%s.npz" % tag))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", script.format(repo=REPO, path=path, **kw)], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return dict(np.load(path))


def stat_values(cells):
    """[2 tables][B][32] GroupStat (sum, sq) x (lo, hi) limbs -> the totals as float64 (enc_common.h stat_read)"""
    c = cells.reshape(2, -1, 2, 2)               # table, (image, group), sum / sq, limb
    lo = c[0, :, :, 0].astype(np.uint64)
    top = c[1, :, :, 1].astype(np.float64) + (lo >> np.uint64(32)).astype(np.float64)
    return (top * 2.0 ** 32 + (lo & np.uint64(0xffffffff)).astype(np.float64)) * 2.0 ** -40


def test_layers_equal_conv_pc_bit_for_bit(tmp_path):
    a = run(tmp_path, LAYERS, "layers_mw", {"CHORE_CONV_MW": "all", "CHORE_CONV_MW_TH2": "1", "CHORE_CONV_MW_FILL": "256"})
    b = run(tmp_path, LAYERS, "layers_pc", {"CHORE_CONV_MW": "0"})
    c = run(tmp_path, LAYERS, "layers_mw_dense", {"CHORE_CONV_MW_FILL": "128"})     # the inference encoder's tilings: dense tiles on small maps
    n = len([k for k in a if k.startswith("y")])
    assert n == 12 and set(a) == set(b)
    for i in range(n):
        ya, yb = a["y%d" % i], b["y%d" % i]
        assert np.isfinite(ya).all() and np.abs(ya).max() > 0.1
        if i == 6:
            # 256 -> 128 at 64^2, B = 4: with CHORE_CONV_MW_TH2 conv_mw_plan tiles it 2 x 32 pixels x 128 channels (conv_pc_plan: 8 x 32 x 32); the order
            # in which a tile walks the 32-channel chunks depends on the tile's index, so the sums differ in their rounding
            assert np.abs(ya - yb).max() <= 2e-6 * np.abs(yb).max(), np.abs(ya - yb).max()
            assert not np.array_equal(ya, yb)
        else:
            assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)), (i, np.abs(ya - yb).max())
        sa, sb = stat_values(a["s%d" % i]), stat_values(b["s%d" % i])
        assert np.abs(sa - sb).max() <= 2e-6 * np.abs(sb).max(), (i, np.abs(sa - sb).max(), np.abs(sb).max())
    # the switch did something: at least one layer's statistics differ in their last bits
    assert any(not np.array_equal(a["s%d" % i], b["s%d" % i]) for i in range(n))
    # dense tilings (what the inference encoder asks for, ConvArgs::fill = 128): where conv_mw_plan picks another tile the chunk order differs, hence the rounding; same values otherwise
    other = 0
    for i in range(n):
        yc, yb = c["y%d" % i], b["y%d" % i]
        assert np.abs(yc - yb).max() <= 2e-6 * np.abs(yb).max(), (i, np.abs(yc - yb).max())
        sc, sb = stat_values(c["s%d" % i]), stat_values(b["s%d" % i])
        assert np.abs(sc - sb).max() <= 2e-6 * np.abs(sb).max(), i
        other += int(not np.array_equal(yc, yb))
    assert other >= 3                # the 64^2 maps at B = 4 and the 128-channel layer at B = 2 run on other tiles


@pytest.mark.parametrize("B,H,W", [(3, 80, 112), (4, 512, 512)])
def test_encoder_on_conv_mw_equals_encoder_on_conv_pc(tmp_path, B, H, W):
    a = run(tmp_path, ENCODER, "enc_mw", {"CHORE_CONV_MW": "all"}, B=B, H=H, W=W)
    b = run(tmp_path, ENCODER, "enc_pc", {"CHORE_CONV_MW": "0"}, B=B, H=H, W=W)
    assert set(a) == set(b) and len(a) >= 3
    worst = 0.0
    for k in a:
        assert np.isfinite(a[k]).all()
        err = np.abs(a[k] - b[k]).max() / np.abs(b[k]).max()
        worst = max(worst, err)
        assert err <= 5e-6, (k, err)
    print("conv_mw vs conv_pc encoder %dx%dx%d: worst relative deviation %.2e" % (B, H, W, worst))
    assert any((a[k] != b[k]).any() for k in a)

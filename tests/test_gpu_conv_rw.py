"""GPU: conv_rw_kernel (csrc/conv_rw.hip) -- the 1x1 layers on register-resident weights -- against the kernels it replaces.

The kernel choice is made by environment switches that the library reads once per process, so each variant runs in a process of
its own (scripts/-style one-liners below) and the results are compared here:
  * whole encoder, fp16x3 and fp16 modes, conv_rw (default) against conv_pc_kernel (CHORE_CONV_RW=0): every stack's feature
    map, tmpx and normx.  The two kernels add the same products in a different order and reduce the GroupNorm statistics from
    different partial sums: agreement to fp32 summation-order level is the bound (5e-6 of the tensor's largest entry in fp16x3, measured
    8.8e-7; 5e-3 in the fp16 mode, measured 1.3e-3: its tensors are stored as halves and a half ulp of a value flips with the order);
  * the bf16 instantiation is opt-in (CHORE_CONV_RW_BF16=1): the layer tests of tests/test_gpu_train_ops.py run with it in a
    subprocess, and the encoder is compared against the conv_lds_kernel encoder within the bf16 mode's stated bound.
The model/HGFilters.py:128-142,167-183 layers (conv_last, l, bl + al.l with the residual) and ConvBlock's downsample
(model/net_util.py:364-371) are the ones on the kernel; the 80 x 112 image gives 20 x 28 feature maps: 17.5 pixel blocks, the last one ends inside the map."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DUMP = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + "/tests")
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
net = CHORE(chore_opt({mode!r})).cuda().eval(); synth.load_synth_weights(net, 0)
for p in net.parameters(): p.requires_grad_(False)
img = torch.from_numpy(synth.synth_images(3, 80, 112, 5)).cuda()
with torch.no_grad():
    net.filter(img)
out = dict(("f%d" % i, o.float().cpu().numpy()) for i, o in enumerate(net.im_feat_list))
out["tmpx"] = net.tmpx.float().cpu().numpy(); out["normx"] = net.normx.float().cpu().numpy()
np.savez({path!r}, **out)
"""


def dump(tmp_path, mode, tag, env):
    path = str(tmp_path / ("enc_%s_%s.npz" % (mode, tag)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", DUMP.format(repo=REPO, mode=mode, path=path)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(path))


@pytest.mark.parametrize("mode,bound", [("fp16x3", 5e-6), ("fp16", 5e-3)])
def test_encoder_on_conv_rw_equals_encoder_on_conv_pc(tmp_path, mode, bound):
    a = dump(tmp_path, mode, "rw", {})
    b = dump(tmp_path, mode, "pc", {"CHORE_CONV_RW": "0"})
    assert set(a) == set(b) and len(a) >= 3
    worst = 0.0
    for k in a:
        assert np.isfinite(a[k]).all()
        err = np.abs(a[k] - b[k]).max() / np.abs(b[k]).max()
        worst = max(worst, err)
        assert err <= bound, (k, err)
    print("conv_rw vs conv_pc, %s: worst relative deviation %.2e" % (mode, worst))
    # the two runs must really differ in the kernel used: bit-identical outputs would mean the switch did nothing
    assert any((a[k] != b[k]).any() for k in a)


def test_bf16_instantiation_opt_in(tmp_path):
    a = dump(tmp_path, "bf16", "rw", {"CHORE_CONV_RW_BF16": "1"})
    b = dump(tmp_path, "bf16", "lds", {})
    for k in a:
        rl2 = np.linalg.norm((a[k] - b[k]).ravel()) / np.linalg.norm(b[k].ravel())
        assert rl2 <= 2e-2, (k, rl2)                      # the bf16 mode's stated bound (tests/test_gpu_encoder.py)
    e = dict(os.environ)
    e["CHORE_CONV_RW_BF16"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_train_ops.py"), "-x", "-q", "-m", "gpu",
                        "-k", "test_conv_gn_layer"], env=e, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:]

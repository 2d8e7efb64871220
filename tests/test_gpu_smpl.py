"""GPU parity of the HIP SMPL-H LBS (chore_smpl_lbs_fwd/bwd) against the reference SMPL_Layer outputs and
autograd gradients (tests/golden/smpl_lbs.npz) and the numpy oracle.  Tolerances: positions 5e-6 m,
gradients 2e-4 relative to their scale (fp32, different summation order)."""
import numpy as np
import pytest
import torch

from test_oracle_smpl import smpl_inputs
from oracle import smpl as osm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def layer():
    from chore_amd.lib_smpl import SMPL_Layer
    from chore_amd.utils import synth
    return SMPL_Layer.from_arrays(synth.synth_smplh_model(0)).cuda()


def test_lbs_forward(layer):
    g, model, offs = smpl_inputs()
    verts, jtr, vp, nk = layer(torch.from_numpy(g["pose"]).cuda(), th_betas=torch.from_numpy(g["betas"]).cuda(),
                               th_trans=torch.from_numpy(g["trans"]).cuda(), th_offsets=torch.from_numpy(offs).cuda())
    sel = g["sel"]
    np.testing.assert_allclose(verts.cpu().numpy()[:, sel], g["verts_sel"], atol=5e-6)
    np.testing.assert_allclose(jtr.cpu().numpy(), g["joints"], atol=5e-6)
    np.testing.assert_allclose(vp.cpu().numpy()[:, sel], g["v_posed_sel"], atol=2e-6)
    np.testing.assert_allclose(nk.cpu().numpy()[:, sel], g["naked_sel"], atol=2e-6)
    ov, oj, _, _ = osm.lbs(model, g["pose"], g["betas"], g["trans"], offs)
    np.testing.assert_allclose(verts.cpu().numpy(), ov, atol=5e-6)   # every vertex, against the oracle


def test_lbs_backward_matches_reference_autograd(layer):
    g, model, offs = smpl_inputs()
    rs = np.random.RandomState(int(g["w_verts_seed"]))
    rs.standard_normal((2, 6890, 3))                     # the offsets draw of make_golden.gen_smpl
    wv = rs.standard_normal((2, 6890, 3)).astype(np.float32)
    pose = torch.from_numpy(g["pose"]).cuda().requires_grad_(True)
    betas = torch.from_numpy(g["betas"]).cuda().requires_grad_(True)
    trans = torch.from_numpy(g["trans"]).cuda().requires_grad_(True)
    verts, jtr, _, _ = layer(pose, th_betas=betas, th_trans=trans, th_offsets=torch.from_numpy(offs).cuda())
    loss = (verts * torch.from_numpy(wv).cuda()).sum() + (jtr * torch.from_numpy(g["w_joints"]).cuda()).sum()
    loss.backward()
    for got, ref, name in ((pose.grad, g["dpose"], "pose"), (betas.grad, g["dbetas"], "betas"),
                           (trans.grad, g["dtrans"], "trans")):
        ref_scale = np.abs(ref).max()
        err = np.abs(got.cpu().numpy() - ref).max() / ref_scale
        assert err < 2e-4, (name, err)


def test_lbs_batch_sizes_and_determinism(layer):
    from chore_amd.utils import synth
    pose, betas, trans = synth.synth_smpl_params(9, seed=3)      # 9 frames: exercises the 4-frame grouping
    args = [torch.from_numpy(a).cuda() for a in (pose, betas, trans)]
    v1, j1, _, _ = layer(args[0], th_betas=args[1], th_trans=args[2])
    v2, j2, _, _ = layer(args[0][4:5], th_betas=args[1][4:5], th_trans=args[2][4:5])
    assert torch.equal(v1[4:5], v2) and torch.equal(j1[4:5], j2)   # frames are independent, bit for bit


@pytest.mark.parametrize("B", [1, 3])
def test_landmark_regression_forward_and_backward(B):
    """chore_landmarks_fwd / _bwd = reg @ verts and its transpose (fp64 product as the reference; fp32 sums: 2e-6 of
    the result's scale), dense and sparse rows, a vertex count that is no multiple of the tile sizes"""
    from chore_amd.lib_smpl.wrapper_pytorch import landmarks, synthetic_regressors
    V = 6890
    rs = np.random.RandomState(5)
    reg = np.concatenate(synthetic_regressors(V) + [rs.standard_normal((3, V)).astype(np.float32)], 0)   # 137 sparse + 3 dense rows
    verts = rs.standard_normal((B, V, 3)).astype(np.float32)
    w = rs.standard_normal((B, reg.shape[0], 3)).astype(np.float32)
    vt = torch.from_numpy(verts).cuda().requires_grad_(True)
    lm = landmarks(torch.from_numpy(reg).cuda(), vt)
    (lm * torch.from_numpy(w).cuda()).sum().backward()
    ref = np.einsum("rv,bvk->brk", reg.astype(np.float64), verts.astype(np.float64))
    dref = np.einsum("rv,brk->bvk", reg.astype(np.float64), w.astype(np.float64))
    assert np.abs(lm.detach().cpu().numpy() - ref).max() < 2e-6 * np.abs(ref).max()
    assert np.abs(vt.grad.cpu().numpy() - dref).max() < 2e-6 * np.abs(dref).max()
    lm2 = landmarks(torch.from_numpy(reg).cuda(), vt.detach())
    assert torch.equal(lm2, lm.detach())                    # fixed summation order

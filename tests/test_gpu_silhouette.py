"""GPU: the HIP silhouette rasteriser (chore_silhouette_fwd/bwd) and SilLossROI against the numpy restatement
(oracle/silhouette.py, itself pinned by the reference's known-answer tests) and against those known answers
directly."""
import math

import numpy as np
import pytest
import torch

from oracle import silhouette as osil

pytestmark = pytest.mark.gpu


def random_mesh(rs, B, V, Fn):
    """projected vertices in and around the view, with depth; random triangles of both windings"""
    v = np.concatenate([rs.uniform(-1.2, 1.2, (B, V, 2)), rs.uniform(0.5, 4.0, (B, V, 1))], -1).astype(np.float32)
    f = np.stack([np.stack([rs.choice(V, 3, replace=False) for _ in range(Fn)]) for _ in range(B)]).astype(np.int64)
    return v, f


def test_rasterizer_matches_restatement():
    from chore_amd.recon.obj_pose_roi import _RasterizeFn, vertices_to_faces
    rs = np.random.RandomState(0)
    B, V, Fn, S = 3, 30, 40, 64
    v, f = random_mesh(rs, B, V, Fn)
    v[1, :, :2] *= 0.3                                     # small triangles
    v[2, :5] = 0.0                                         # degenerate vertices
    f2 = osil.fill_back(f)
    tri = osil.vertices_to_faces(v, f2)
    fim_o, alpha_o = osil.rasterize_fwd(tri, S)
    vt = torch.from_numpy(v).cuda().requires_grad_(True)
    tri_t = vertices_to_faces(vt, torch.from_numpy(f2).cuda())
    alpha, fim = _RasterizeFn.apply(tri_t, S)
    assert np.array_equal(fim.cpu().numpy(), fim_o)
    assert np.array_equal(alpha.detach().cpu().numpy(), alpha_o)
    assert 0.05 < alpha_o.mean() < 0.95
    g = rs.standard_normal((B, S, S)).astype(np.float32)
    (alpha * torch.from_numpy(g).cuda()).sum().backward()
    gt_o = osil.rasterize_bwd(tri, fim_o, alpha_o, g)
    gv_o = np.zeros_like(v)
    for b in range(B):
        np.add.at(gv_o[b], f2[b].reshape(-1), gt_o[b].reshape(-1, 3))
    got = vt.grad.cpu().numpy()
    scale = np.abs(gv_o).max()
    assert scale > 0 and np.abs(got - gv_o).max() < 1e-4 * scale
    assert np.abs(got[..., 2]).max() == 0                  # silhouettes carry no depth gradient


@pytest.mark.parametrize("verts,pyi,pxi,minus_one,ref", [
    ([[0.8, 0.8, 1.0], [0.0, -0.5, 1.0], [0.2, -0.4, 1.0]], 25, 35, True,
     [[1.6725862, -0.26021874, 0.0], [1.41986704, -1.64284933, 0.0], [0.0, 0.0, 0.0]]),
    ([[0.8, 0.8, 1.0], [-0.5, -0.8, 1.0], [0.8, -0.8, 1.0]], 40, 50, False,
     [[0.98646867, 1.04628897, 0.0], [-1.03415668, -0.10403691, 0.0], [3.00094461, -1.55173182, 0.0]]),
])
def test_reference_known_answers(verts, pyi, pxi, minus_one, ref):
    """external/neural_renderer/tests/test_rasterize_silhouettes.py:37-108 on the HIP kernels (look_at camera with
    an identity rotation, no perspective: projected vertices = vertices - eye)"""
    from chore_amd.recon.obj_pose_roi import _RasterizeFn, vertices_to_faces
    eye = torch.tensor([0, 0, -(1.0 / math.tan(math.radians(30)) + 1)]).cuda()
    v = torch.zeros(4, 3, 3).cuda()
    v[2] = torch.tensor(verts)
    v.requires_grad_(True)
    f = torch.zeros(4, 1, 3, dtype=torch.long).cuda()
    f[2, 0] = torch.tensor([0, 1, 2])
    f = torch.cat((f, f.flip(-1)), 1)
    alpha, _ = _RasterizeFn.apply(vertices_to_faces(v - eye, f), 64)
    images = alpha.flip(1)
    loss = torch.sum(torch.abs(images[:, pyi, pxi] - (1 if minus_one else 0)))
    loss.backward()
    np.testing.assert_allclose(v.grad[2].cpu().numpy(), np.array(ref, np.float32), rtol=1e-2, atol=1e-6)


def cube():
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * 0.25
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6],
                  [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int64)
    return v, f


def test_sil_loss_roi_end_to_end():
    """SilLossROI.from_crops: image, loss and vertex-path gradients against the restatement incl. the projection"""
    from chore_amd.recon.obj_pose_roi import SilLossROI
    from chore_amd.recon.recon_fit_base import ReconFitterBase
    B, S = 2, 64
    v, f = cube()
    yy, xx = np.mgrid[0:S, 0:S]
    obj_crop = np.stack([((xx - 30) ** 2 + (yy - 34) ** 2) < 15 ** 2] * B)
    ps_crop = np.stack([xx < 12] * B)
    K = np.array([[[1.6, 0, 0.5], [0, 1.6, 0.5], [0, 0, 1]]] * B, np.float32)
    sil = SilLossROI.from_crops(obj_crop, ps_crop, K, v, f)
    rs = np.random.RandomState(1)
    M = torch.eye(3).repeat(B, 1, 1) + 0.3 * torch.from_numpy(rs.standard_normal((B, 3, 3)).astype(np.float32))
    R = ReconFitterBase.project_so3(M.cuda()).detach().requires_grad_(True)
    t = torch.tensor([[0.05, -0.03, 2.0], [-0.1, 0.08, 2.4]]).cuda().requires_grad_(True)
    s = torch.tensor([1.0, 1.2]).cuda().requires_grad_(True)
    loss_dict, image, edges, image_ref, edt = sil(R, t, s)
    loss_dict["mask"].backward()
    assert image.shape == (B, S, S) and edges.shape == (B, S, S) and edt.shape == (B, S, S)
    # restatement
    verts = (np.matmul(np.stack([v] * B), R.detach().cpu().numpy()) + t.detach().cpu().numpy()[:, None]) * \
        s.detach().cpu().numpy()[:, None, None]
    pv = osil.projection(verts.astype(np.float32), K, np.eye(3, dtype=np.float32)[None], np.zeros((1, 1, 3), np.float32))
    img_o, ctx = osil.render_silhouettes(pv, np.stack([f] * B), S)
    keep = sil.keep_mask.cpu().numpy()
    assert np.array_equal(image.detach().cpu().numpy(), keep * img_o)
    assert 100 < img_o[0].sum() < S * S / 2
    loss_o = ((keep * img_o - obj_crop.astype(np.float32)) ** 2).sum((1, 2)).mean()
    assert abs(float(loss_dict["mask"]) - loss_o) < 1e-4 * loss_o
    for p in (R, t, s):
        assert torch.isfinite(p.grad).all()
    assert t.grad.abs().max() > 0 and R.grad.abs().max() > 0
    # the loss gradient w.r.t. the image, pushed through the restated backward, then through the projection by autograd
    g_img = (2 * (keep * img_o - obj_crop) * keep / B).astype(np.float32)
    gpv = osil.render_silhouettes_bwd(ctx, g_img, v.shape[0])
    from chore_amd.recon.obj_pose_roi import projection
    vt = torch.from_numpy(verts.astype(np.float32)).cuda().requires_grad_(True)
    pr = projection(vt, torch.from_numpy(K).cuda(), torch.eye(3).cuda()[None], torch.zeros(1, 3).cuda())
    (pr * torch.from_numpy(gpv).cuda()).sum().backward()
    R2, t2, s2 = (x.detach().clone().requires_grad_(True) for x in (R, t, s))
    (sil.apply_transformation(R2, t2, s2) * vt.grad).sum().backward()
    # t / s / R gradients are sums over all vertices of terms that nearly cancel (the edge terms carry 1 / (dist + 1e-4)
    # factors): the bound is relative to the sum of the magnitudes that enter, not to the cancelled result.  (The kernel
    # adds a triangle's contributions lane-parallel with a fixed butterfly; the restatement adds them serially.)
    scale = float(vt.grad.abs().sum()) * 4.0
    for a, b in ((t.grad, t2.grad), (s.grad, s2.grad), (R.grad, R2.grad)):
        assert (a - b).abs().max() < 2e-6 * scale + 1e-3 * float(b.abs().max()), ((a - b).abs().max(), scale)


def test_sil_phase_runs_in_the_fit_loop(opt):
    """optimize_smpl_object with data_dict['silhouette']: the 'sil' phase (recon_fit_behave.py:108-131) executes"""
    import copy
    from chore_amd.recon.obj_pose_roi import SilLossROI
    from test_gpu_fit import _run_fit  # noqa: F401  (shared setup lives there)
    import test_gpu_fit as tf
    B, S = 2, 64
    v, f = cube()
    yy, xx = np.mgrid[0:S, 0:S]
    obj_crop = np.stack([((xx - 32) ** 2 + (yy - 32) ** 2) < 12 ** 2] * B)
    K = np.array([[[1.6, 0, 0.5], [0, 1.6, 0.5], [0, 0, 1]]] * B, np.float32)
    sil = SilLossROI.from_crops(obj_crop, np.zeros_like(obj_crop), K, v, f)
    out = tf.run_fit_with(copy.copy(opt), use_graphs=False, silhouette=sil, obj_iter=1, sil_iter=2, joint_iter=1)
    assert all(np.isfinite(x).all() for x in out)


def test_placed_triangles_operator_equals_tensor_expressions():
    """chore_sil_project_fwd / _bwd (placement, projection, both windings in one launch) against apply_transformation ->
    projection -> vertices_to_faces as tensor expressions with autograd: triangles to 2e-6 of their scale, the pose
    gradients for a random upstream gradient to 2e-5 of their largest entry"""
    from chore_amd.recon.obj_pose_roi import SilLossROI, _PlacedTrianglesFn, projection, vertices_to_faces
    B, S = 3, 32
    v, f = cube()
    yy, xx = np.mgrid[0:S, 0:S]
    crop = np.stack([((xx - 16) ** 2 + (yy - 16) ** 2) < 64] * B)
    rs = np.random.RandomState(4)
    K = np.array([[[1.6, 0.02, 0.5], [0.01, 1.5, 0.48], [0, 0, 1]]] * B, np.float32) + rs.uniform(-0.02, 0.02, (B, 3, 3)).astype(np.float32)
    sil = SilLossROI.from_crops(crop, np.zeros_like(crop), K, v, f)
    # a camera that is not the identity (the fit's is), to cover that part of the chain too
    cam_R = torch.linalg.qr(torch.eye(3) + 0.1 * torch.from_numpy(rs.standard_normal((3, 3)).astype(np.float32)))[0][None].cuda()
    cam_t = torch.tensor([[0.03, -0.02, 0.1]]).cuda()
    sil.R, sil.t = cam_R.contiguous(), cam_t

    def pose():
        R = (torch.eye(3).repeat(B, 1, 1) + 0.2 * torch.from_numpy(rs.standard_normal((B, 3, 3)).astype(np.float32))).cuda()
        t = torch.from_numpy(np.array([[0.05, -0.03, 2.0], [-0.1, 0.08, 2.4], [0.0, 0.0, 3.0]], np.float32)).cuda()
        s = torch.tensor([1.0, 1.2, 0.8]).cuda()
        return [x.requires_grad_(True) for x in (R, t, s)]

    rs_state = rs.get_state()
    R, t, s = pose()
    rs.set_state(rs_state)
    R2, t2, s2 = pose()
    tri = _PlacedTrianglesFn.apply(R, t, s, sil.vertices, sil.faces32, sil.K, sil.R, sil.t, sil.adj_off, sil.adj)
    faces2 = torch.cat((sil.faces, sil.faces.flip(-1)), dim=1)
    ref = vertices_to_faces(projection(sil.apply_transformation(R2, t2, s2), sil.K, sil.R, sil.t), faces2)
    assert tri.shape == ref.shape
    assert float((tri - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    g = torch.from_numpy(rs.standard_normal(tuple(ref.shape)).astype(np.float32)).cuda()
    (tri * g).sum().backward()
    (ref * g).sum().backward()
    for a, b, n in ((R.grad, R2.grad, "R"), (t.grad, t2.grad, "t"), (s.grad, s2.grad, "s")):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (n, float((a - b).abs().max()), float(b.abs().max()))


def test_placement_projection_kernel_against_the_reference():
    """chore_sil_project_fwd (object placement + camera projection + the doubled triangle list, one launch) against the
    reference's own code: tests/golden/sil_project.npz holds what recon/obj_pose_roi.py apply_transformation and
    neural_renderer's projection.py + vertices_to_faces.py (fill_back faces, renderer.py:126-127) computed on CPU for the same
    pose, template and ROI intrinsics (written by make_golden.py gen_sil_project); and SilLossROI built from
    the fixture's crops carries the reference's keep mask / edge distance transform on the device."""
    import os
    from chore_amd.recon.obj_pose_roi import SilLossROI, _PlacedTrianglesFn
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sil_project.npz"))
    sil = SilLossROI.from_crops(g["obj_crop"], g["ps_crop"], g["K"], g["verts"], g["faces"])
    R, t, s = (torch.from_numpy(g[k]).cuda() for k in ("R", "obj_t", "obj_s"))
    tri = _PlacedTrianglesFn.apply(R, t, s, sil.vertices, sil.faces32, sil.K, sil.R, sil.t, sil.adj_off, sil.adj).cpu().numpy()
    ref = g["tri"]
    assert tri.shape == ref.shape
    # x / (z + eps), the K row products and the affine map to [-1, 1] are a handful of fp32 operations per coordinate
    assert np.abs(tri - ref).max() <= 4e-6 * np.abs(ref).max(), np.abs(tri - ref).max()
    assert np.array_equal(tri[..., 2], ref[..., 2]) or np.abs(tri[..., 2] - ref[..., 2]).max() <= 1e-6      # depth = placed z
    assert np.array_equal(sil.keep_mask.cpu().numpy(), g["keep_mask"])
    np.testing.assert_allclose(sil.edt_ref_edge.cpu().numpy(), g["edt_ref_edge"], rtol=1e-6)
    # and the rendering of that triangle list is the restatement's rendering of the REFERENCE's projected vertices
    loss_dict, image, edges, image_ref, edt = sil(R, t, s)
    img_o, _ = osil.render_silhouettes(g["proj"], np.stack([g["faces"]] * 3), sil.rend_size)
    got = image.cpu().numpy()
    assert (got != g["keep_mask"] * img_o).mean() < 2e-3          # (pixel centres within 4e-6 of an edge may flip)
    assert img_o.sum() > 50

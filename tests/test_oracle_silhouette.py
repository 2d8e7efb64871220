"""CPU: the silhouette restatement (oracle/silhouette.py) against the reference's own known-answer tests
(external/neural_renderer/tests/test_rasterize_silhouettes.py:37-108: one triangle, look_at camera without
perspective, 64x64, the analytic gradients the authors checked in; their tolerance rtol=1e-2 is kept)."""
import math

import numpy as np

from oracle import silhouette as osil

EYE = [0, 0, -(1.0 / math.tan(math.radians(30)) + 1)]   # Renderer default eye (renderer.py:46)


def run_case(vertices, pyi, pxi, minus_one):
    v = np.zeros((4, 3, 3), np.float32)
    v[2] = np.array(vertices, np.float32)               # utils.to_minibatch: the case sits in batch slot 2
    f = np.zeros((4, 1, 3), np.int64)
    f[2] = [[0, 1, 2]]
    pv = osil.look_at(v, EYE)                            # perspective = False
    img, ctx = osil.render_silhouettes(pv, f, 64)
    val = img[:, pyi, pxi] - (1 if minus_one else 0)
    g_img = np.zeros_like(img)
    g_img[:, pyi, pxi] = np.sign(val)                    # d sum|x| / dx
    gv = osil.render_silhouettes_bwd(ctx, g_img, 3)      # look_at is v - eye with an identity rotation here
    return img, gv


def test_backward_case1_gradient_out_of_face():
    img, gv = run_case([[0.8, 0.8, 1.0], [0.0, -0.5, 1.0], [0.2, -0.4, 1.0]], 25, 35, True)
    ref = np.array([[1.6725862, -0.26021874, 0.0], [1.41986704, -1.64284933, 0.0], [0.0, 0.0, 0.0]], np.float32)
    assert img[2, 25, 35] == 0.0
    np.testing.assert_allclose(gv[2], ref, rtol=1e-2, atol=1e-6)
    assert np.abs(gv[[0, 1, 3]]).max() == 0


def test_backward_case2_gradient_on_face():
    img, gv = run_case([[0.8, 0.8, 1.0], [-0.5, -0.8, 1.0], [0.8, -0.8, 1.0]], 40, 50, False)
    ref = np.array([[0.98646867, 1.04628897, 0.0], [-1.03415668, -0.10403691, 0.0], [3.00094461, -1.55173182, 0.0]],
                   np.float32)
    assert img[2, 40, 50] == 1.0
    np.testing.assert_allclose(gv[2], ref, rtol=1e-2, atol=1e-6)


def test_forward_properties():
    """a front-facing and a back-facing copy cover the same pixels (fill_back); rows are flipped; degenerate
    all-zero faces hit nothing; projection maps the principal point to the image centre"""
    v = np.array([[[-0.5, -0.5, 2.0], [0.5, -0.5, 2.0], [0.0, 0.6, 2.0]]], np.float32)
    f = np.array([[[0, 1, 2]]], np.int64)
    img_ccw, _ = osil.render_silhouettes(v, f, 32)
    img_cw, _ = osil.render_silhouettes(v, f[:, :, ::-1], 32)
    assert img_ccw.sum() > 50 and np.array_equal(img_ccw, img_cw)
    one_sided, _ = osil.render_silhouettes(v, f, 32, do_fill_back=False)
    other, _ = osil.render_silhouettes(v, f[:, :, ::-1], 32, do_fill_back=False)
    assert (one_sided.sum() == 0) != (other.sum() == 0)
    ys = np.nonzero(img_ccw[0].sum(1))[0]
    assert img_ccw[0, ys.min()].sum() < img_ccw[0, ys.max()].sum()   # apex (larger v) is drawn at the top rows
    K = np.array([[[2.0, 0, 0.5], [0, 2.0, 0.5], [0, 0, 1]]], np.float32)
    p = osil.projection(np.array([[[0.0, 0.0, 3.0]]], np.float32), K, np.eye(3, dtype=np.float32)[None],
                        np.zeros((1, 1, 3), np.float32), orig_size=1.0)
    np.testing.assert_allclose(p[0, 0], [0.0, 0.0, 3.0], atol=1e-6)


def _fixture():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sil_project.npz"))


def test_projection_and_triangle_list_against_the_reference():
    """tests/golden/sil_project.npz is written by the reference's own neural_renderer/projection.py and vertices_to_faces.py
    (make_golden.py gen_sil_project): the restatement's projection and fill_back + vertices_to_faces reproduce them"""
    g = _fixture()
    pv = osil.projection(g["placed"], g["K"], np.eye(3, dtype=np.float32)[None], np.zeros((1, 1, 3), np.float32))
    assert np.abs(pv - g["proj"]).max() <= 2e-6 * np.abs(g["proj"]).max()
    B = g["placed"].shape[0]
    tri = osil.vertices_to_faces(g["proj"], osil.fill_back(np.stack([g["faces"]] * B)))
    assert np.array_equal(tri, g["tri"])


def test_roi_camera_and_masks_host_logic_against_the_reference():
    """the host side of SilLossROI (chore_amd/recon/obj_pose_roi.py) on CPU tensors -- square box, box in the original image, ROI
    intrinsics, keep mask, reference edges and their distance transform, placement, off-screen penalty -- against what the
    reference's recon/obj_pose_roi.py:92-199 and recon/bbox.py:25-46 computed for the same inputs"""
    import torch
    from chore_amd.recon.obj_pose_roi import SilLossROI, make_bbox_square
    g = _fixture()
    sq = make_bbox_square(g["boxes_xywh"], 0.3)
    np.testing.assert_allclose(sq, g["squares"], rtol=0, atol=1e-12)
    orig = np.stack([SilLossROI.to_original_bbox(b, 1200 / 512., c) for b, c in zip(sq, g["crop_centers"])])
    np.testing.assert_allclose(orig, g["bbox_orig"], rtol=0, atol=1e-9)
    K = torch.cat([SilLossROI.compute_K_roi(b) for b in orig], 0).numpy()
    assert np.array_equal(K, g["K"])
    sil = SilLossROI.from_crops(g["obj_crop"], g["ps_crop"], g["K"], g["verts"], g["faces"], device="cpu")
    assert np.array_equal(sil.keep_mask.numpy(), g["keep_mask"])
    assert np.array_equal(sil.image_ref.numpy(), g["image_ref"])
    assert np.array_equal(sil.compute_edges(sil.image_ref).numpy(), g["ref_edges"])
    np.testing.assert_allclose(sil.edt_ref_edge.numpy(), g["edt_ref_edge"], rtol=1e-6, atol=0)
    assert 0 < g["keep_mask"].mean() < 1 and g["ref_edges"].sum() > 0
    placed = sil.apply_transformation(torch.from_numpy(g["R"]), torch.from_numpy(g["obj_t"]), torch.from_numpy(g["obj_s"]))
    assert np.abs(placed.numpy() - g["placed"]).max() <= 1e-6
    off = sil.compute_offscreen_loss(torch.from_numpy(g["placed"]))
    np.testing.assert_allclose(off.numpy(), g["offscreen"], rtol=1e-5, atol=1e-6)
    assert (g["offscreen"] > 0).any()

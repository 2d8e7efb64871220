"""GPU: the image preparation kernels (csrc/image_prep.hip) against oracle/image_prep.py (numpy restatement of
data/test_data.py:59-125 + cv2's 8-bit INTER_LINEAR resize -- PARITY UNPINNED at cv2, see the oracle's header):
bit for bit, it is integer arithmetic."""
import numpy as np
import pytest
import torch

from oracle import image_prep as oi

pytestmark = pytest.mark.gpu


def _scene(rs, H, W, box_p, box_o):
    rgb = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    pm = np.zeros((H, W), np.uint8)
    om = np.zeros((H, W), np.uint8)
    pm[box_p[1]:box_p[3], box_p[0]:box_p[2]] = 255
    om[box_o[1]:box_o[3], box_o[0]:box_o[2]] = 255
    # jpeg-like soft edges and a region where both masks are 200 (200 + 200 wraps to 144 > 127) and one where they are
    # 128 (128 + 128 wraps to 0: not foreground -- the reference's uint8 sum)
    pm[box_p[1]:box_p[1] + 4, box_p[0]:box_p[2]] = rs.randint(90, 200, (4, box_p[2] - box_p[0]))
    pm[5:9, 5:9] = 128
    om[5:9, 5:9] = 128
    return rgb, pm, om


@pytest.mark.parametrize("shape", [((37, 53), (64, 48)), ((480, 640), (2048, 1536)), ((1536, 2048), (1024, 768)), ((96, 128), (48, 64))])
def test_resize_matches_restated_cv2(shape):
    from chore_amd.data import ImagePrep
    (sh, sw), (dw, dh) = shape
    rs = np.random.RandomState(sh)
    prep = ImagePrep()
    for C in (1, 3):
        img = rs.randint(0, 256, (sh, sw) if C == 1 else (sh, sw, 3)).astype(np.uint8)
        got = prep.resize(img, (dw, dh)).cpu().numpy()
        assert np.array_equal(got, oi.resize_linear_u8(img, (dw, dh))), (shape, C)


def test_masks2bbox_wraps_like_uint8():
    from chore_amd.data import ImagePrep
    rs = np.random.RandomState(1)
    _, pm, om = _scene(rs, 300, 400, (100, 60, 180, 250), (170, 150, 260, 230))
    bmin, bmax = ImagePrep().masks2bbox([pm, om])
    rmin, rmax = oi.masks2bbox([pm, om])
    assert np.array_equal(bmin, rmin) and np.array_equal(bmax, rmax)
    e = np.zeros((10, 12), np.uint8)
    bmin, bmax = ImagePrep().masks2bbox([e, e])
    assert list(bmin) == [50000, 50000] and list(bmax) == [-100, -100]


@pytest.mark.parametrize("case", ["kinect", "small_portrait", "border", "exact2x"])
def test_prepare_image_crop_matches_restatement(case):
    from chore_amd.data import ImagePrep
    rs = np.random.RandomState(7)
    if case == "kinect":         # a BEHAVE frame: already 2048 x 1536, crop inside the image
        rgb, pm, om = _scene(rs, 1536, 2048, (800, 300, 1100, 1200), (1050, 700, 1400, 1000))
        scale = 1.13
    elif case == "small_portrait":   # a phone image, taller than wide: resized along the height
        rgb, pm, om = _scene(rs, 800, 600, (200, 100, 400, 700), (350, 400, 520, 560))
        scale = 0.97
    elif case == "border":       # the crop sticks out of the image on two sides (zero padding, clipped last row / column)
        rgb, pm, om = _scene(rs, 1536, 2048, (1500, 900, 1900, 1500), (1800, 1200, 2040, 1530))
        scale = 1.31
    else:                        # crop of exactly 1024 px: the 2 x 2 area fast path
        rgb, pm, om = _scene(rs, 1536, 2048, (800, 300, 1100, 1200), (1050, 700, 1400, 1000))
        scale = 1024 / 1200
    prep = ImagePrep(image_size=(512, 512), crop_size=1200)
    images, center, rscale, old = prep.prepare(rgb, pm, om, scale)
    ref, rcenter, rrscale = oi.prepare_image_crop(rgb, pm, om, scale)
    assert np.array_equal(center, rcenter) and rscale == rrscale
    got = images.cpu().numpy()
    assert got.shape == (5, 512, 512) and got.dtype == np.float32
    assert np.array_equal(got, ref), (case, np.abs(got - ref).max())
    assert got[3].max() == 1.0 and (got[:3][:, (got[3] <= 0.5) & (got[4] <= 0.5)] == 0).all()


@pytest.mark.parametrize("case", ["coco_landscape", "coco_portrait", "far_corner", "exact2x"])
def test_prepare_image_crop_with_mean_center_matches_restatement(case):
    """use_mean_center=True (the COCO loader, recon_fit_coco.py:28; test_data.py:127-160): pad_image's canvas, the crop around
    the mean crop centre and the float resize -- bit for bit against the restatement, including a patch that is pushed
    partly outside the 2048 x 1536 rectangle (clipped paste) and the 2 x 2 area fast path"""
    from chore_amd.data import ImagePrep
    rs = np.random.RandomState(11)
    if case == "coco_landscape":
        rgb, pm, om = _scene(rs, 480, 640, (200, 100, 330, 400), (300, 250, 420, 380))
        scale = 1.07
    elif case == "coco_portrait":
        rgb, pm, om = _scene(rs, 640, 427, (100, 150, 300, 600), (250, 400, 400, 560))
        scale = 0.93
    elif case == "far_corner":       # subject in the top-left corner: the image is shifted far right / down and clipped
        rgb, pm, om = _scene(rs, 1536, 2048, (20, 10, 300, 500), (250, 300, 420, 470))
        scale = 1.2
    else:
        rgb, pm, om = _scene(rs, 480, 640, (200, 100, 330, 400), (300, 250, 420, 380))
        scale = 1024 / 1200
    prep = ImagePrep(image_size=(512, 512), crop_size=1200, use_mean_center=True)
    images, center, rscale, old = prep.prepare(rgb, pm, om, scale)
    ref, rcenter, rrscale, rold = oi.prepare_image_crop_mean_center(rgb, pm, om, scale)
    assert np.array_equal(center, rcenter) and rscale == rrscale and np.array_equal(old, rold)
    assert list(center) == [1008., 995.]
    got = images.cpu().numpy()
    assert got.shape == (5, 512, 512) and got.dtype == np.float32
    assert np.array_equal(got, ref), (case, np.abs(got - ref).max())
    assert got[3].max() > 0.9 and (got[:3][:, (got[3] <= 0.5) & (got[4] <= 0.5)] == 0).all()


def test_fullbody_crop_matches_the_reference():
    """ImagePrep.fullbody_crop against TestData.fullbody_crop itself (data/test_data.py:174-210), run by
    tests/golden/make_golden.py::gen_fullbody_crop on synthetic mocap meshes / keypoints: width-driven and height-driven scales,
    keypoints below the 0.3 confidence cut, and the no-keypoint case -- pinned (fixture written by the reference's own code)."""
    import os
    from chore_amd.data import ImagePrep
    from chore_amd.lib_smpl.wrapper_pytorch import synthetic_regressors
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullbody_crop.npz"))
    reg = synthetic_regressors(6890, seed=int(g["regressor_seed"]))[0]
    prep = ImagePrep(image_size=(512, 512), crop_size=1200)
    branches = set()
    for verts, kpts, want, none in zip(g["verts"], g["kpts"], g["scale"], g["no_keypoints"]):
        got = prep.fullbody_crop(kpts, verts, reg)
        if none:
            assert got == (None, 1.0)
            continue
        assert abs(got - want) <= 1e-9 * abs(want), (got, want)
        branches.add(got > 2.5)
    assert len(branches) == 2

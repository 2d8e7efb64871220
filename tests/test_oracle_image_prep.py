"""CPU: oracle/image_prep.py (restated cv2 8-bit INTER_LINEAR + the reference's crop / compose; PARITY UNPINNED at cv2)
against hand-computed cases of the published fixed-point algorithm and the reference's own crop arithmetic."""
import numpy as np

from oracle import image_prep as oi


def test_resize_hand_cases():
    # 1 x 2 -> 1 x 4: sample positions -0.25 (clamped to pixel 0), 0.25, 0.75, 1.25 (clamped to pixel 1)
    src = np.array([[0, 255]], np.uint8)
    assert oi.resize_linear_u8(src, (4, 1)).tolist() == [[0, 64, 191, 255]]
    # identity and the exact 2 x 2 area path
    img = np.arange(48, dtype=np.uint8).reshape(6, 8)
    assert np.array_equal(oi.resize_linear_u8(img, (8, 6)), img)
    half = oi.resize_linear_u8(img, (4, 3))
    assert half[0, 0] == (0 + 1 + 8 + 9 + 2) >> 2 and half.shape == (3, 4)
    # a constant image stays constant under any resize (the weights sum to 2048)
    c = np.full((37, 53, 3), 201, np.uint8)
    assert (oi.resize_linear_u8(c, (64, 48)) == 201).all()


def test_crop_and_compose_follow_the_reference_arithmetic():
    img = np.arange(1, 101, dtype=np.uint8).reshape(10, 10)
    # a crop that sticks out on the right / bottom: the clipped side loses the last source column / row (x2 = min(w-1, ..))
    c = oi.crop(img, np.array([8.0, 8.0]), np.array([6.0, 6.0]))
    assert c.shape == (6, 6) and c[0, 0] == img[5, 5] and c[3, 3] == img[8, 8] and (c[4:, :] == 0).all() and (c[:, 4:] == 0).all()
    c = oi.crop(img, np.array([1.0, 1.0]), np.array([6.0, 6.0]))          # out on the left / top: plain zero padding
    assert c.shape == (6, 6) and (c[:2] == 0).all() and c[2, 2] == img[0, 0] and c[5, 5] == img[3, 3]
    a = np.array([[200, 128]], np.uint8)
    bmin, bmax = oi.masks2bbox([a, a])                                     # 200+200 wraps to 144 (>127), 128+128 to 0
    assert list(bmin) == [0, 0] and list(bmax) == [1, 1]


def test_mean_center_restatement_hand_cases():
    """pad_image (test_data.py:133-160) and the float resize: hand-checkable cases"""
    from oracle import image_prep as oi
    img = (np.arange(12, dtype=np.uint8).reshape(3, 4) + 1)
    # crop centre (2, 1) -> mean centre: the image lands with its pixel (1, 2) at (995, 1008) of a 1536 x 2048 canvas
    c = oi.pad_image(img, np.array([2., 1.]))
    assert c.shape == (1536, 2048) and c.dtype == np.float64
    assert c[995, 1008] == img[1, 2] and c[994, 1006] == img[0, 0] and c.sum() == img.sum()
    # a centre beyond the mean centre shifts the image up / left and clips it at the canvas origin
    big = np.full((1536, 2048), 7, np.uint8)
    c = oi.pad_image(big, np.array([1108., 1000.]))
    assert c.shape == (1536, 2048) and c[0, 0] == 7 and c[1536 - 6, 0] == 7 and c[1536 - 5, 0] == 0 and c[0, 2048 - 100] == 0
    # float resize: identity, exact 2 x 2 mean, and a 1 x 2 -> 1 x 4 upscale with cv2's pixel-centre weights
    a = np.array([[0., 4.], [8., 12.]])
    assert np.array_equal(oi.resize_linear_f64(a, (2, 2)), a)
    assert np.array_equal(oi.resize_linear_f64(a, (1, 1)), [[6.]])
    r = oi.resize_linear_f64(np.array([[0., 8.]]), (4, 1))
    assert np.allclose(r, [[0., 2., 6., 8.]], atol=1e-6)

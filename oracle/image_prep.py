"""ORACLE (test infrastructure only -- never imported by chore_amd/).   *** PARITY UNPINNED at the cv2 boundary ***

numpy restatement of the image preparation of the test loader:
  TestData.prepare_image_crop    /root/reference/data/test_data.py:59-125   (use_mean_center=False, the BEHAVE protocol;
                                                                             use_mean_center=True, the COCO loader of
                                                                             recon/recon_fit_coco.py:28: pad_image :133-160,
                                                                             change_crop_center :127-131)
  BaseDataset.masks2bbox         /root/reference/data/base_data.py:92-112   (uint8 wrap-around sum of the masks, > 127)
  BaseDataset.crop               /root/reference/data/base_data.py:131-162  (zero padding; a side that is clipped loses its
                                                                             last source row / column: x2 = min(w - 1, ..))
  BaseDataset.resize             /root/reference/data/base_data.py:164-176  (cv2.resize, INTER_LINEAR, uint8)
  BaseDataset.compose_images     /root/reference/data/base_data.py:178-192
The crop / compose arithmetic is the reference's own numpy code, restated.  `cv2.resize` is OpenCV (opencv-python, unpinned in
requirements.txt, absent from this image and from /root/reference): `resize_linear_u8` restates the PUBLISHED algorithm
of its 8-bit INTER_LINEAR path (modules/imgproc/src/resize.cpp: pixel-centre mapping fx = (dx + 0.5) * scale - 0.5 in
float, coefficients rounded to 11 fractional bits with round-half-even, horizontal pass in int32, vertical pass
((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, rows clamped at the border, columns clamped with the
weight moved to the inner sample; an exact 2x2 downscale is taken by the INTER_AREA fast path = (a + b + c + d + 2) >> 2).
The reference holds no test or vector for this path, cv2 cannot be run here: nothing to pin the restatement against.
The HIP kernels (csrc/image_prep.hip) are tested bit for bit against this file.
"""
import numpy as np

COEF_BITS = 11
ONE = 1 << COEF_BITS


def _axis_tables(n_src, n_dst):
    scale = n_src / n_dst                              # double, like cv2's scale_x = 1. / inv_scale_x
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f):
    """(1 - f, f) * 2048 rounded half to even, saturated to int16 (saturate_cast<short>(float) = cvRound)"""
    c0 = np.rint(((np.float32(1.0) - f) * np.float32(ONE)).astype(np.float32)).astype(np.int64)
    c1 = np.rint((f * np.float32(ONE)).astype(np.float32)).astype(np.int64)
    return np.clip(c0, -32768, 32767), np.clip(c1, -32768, 32767)


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize=(width, height)) for uint8 images (H,W) or (H,W,C), interpolation INTER_LINEAR"""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    x = img.reshape(sh, sw, -1).astype(np.int64)
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:                   # INTER_LINEAR -> INTER_AREA fast path
        s = x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2]
        return ((s + 2) >> 2).astype(np.uint8).reshape((dh, dw) + img.shape[2:])
    sx, fx = _axis_tables(sw, dw)
    fx = np.where(sx < 0, np.float32(0), fx)
    sx = np.where(sx < 0, 0, sx)
    fx = np.where(sx >= sw - 1, np.float32(0), fx)
    sx = np.where(sx >= sw - 1, sw - 1, sx)
    a0, a1 = _coef(fx.astype(np.float32))
    sx1 = np.minimum(sx + 1, sw - 1)
    rows = x[:, sx, :] * a0[None, :, None] + x[:, sx1, :] * a1[None, :, None]     # (sh, dw, C) int, 11 fractional bits
    sy, fy = _axis_tables(sh, dh)
    b0, b1 = _coef(fy.astype(np.float32))
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).reshape((dh, dw) + img.shape[2:])


def masks2bbox(masks, thres=127):
    """bounding box (xyxy, max exclusive) of the pixels where the uint8 WRAP-AROUND sum of the masks exceeds thres
    (base_data.py:92-112: np.zeros_like(uint8) += m wraps; the union of cv2.boundingRect over all contours is the
    bounding box of the foreground)"""
    comb = np.zeros_like(masks[0])
    for m in masks:
        comb = comb + m                      # uint8 arithmetic wraps
    ys, xs = np.nonzero(comb > thres)
    if ys.size == 0:
        return np.array([50000, 50000]), np.array([-100, -100])
    return np.array([xs.min(), ys.min()]), np.array([xs.max() + 1, ys.max() + 1])


def crop(img, center, crop_size):
    """base_data.py:131-162"""
    h, w = img.shape[:2]
    topleft = np.round(center - crop_size / 2).astype(int)
    bottom_right = np.round(center + crop_size / 2).astype(int)
    x1, y1 = max(0, topleft[0]), max(0, topleft[1])
    x2, y2 = min(w - 1, bottom_right[0]), min(h - 1, bottom_right[1])
    cropped = img[y1:y2, x1:x2]
    p1, p2 = max(0, -topleft[0]), max(0, -topleft[1])
    p3, p4 = max(0, bottom_right[0] - w + 1), max(0, bottom_right[1] - h + 1)
    pad = [[p2, p4], [p1, p3]] + ([[0, 0]] if img.ndim == 3 else [])
    return np.pad(cropped, pad)


def prepare_image_crop(rgb, person_mask, obj_mask, scale, img_size=(512, 512), crop_size=1200):
    """test_data.py:59-125 from the decoded uint8 images on (use_mean_center=False); `scale` is fullbody_crop's factor.
    -> images (5,H,W) float32, crop_center (2,), resize_scale"""
    bmin, bmax = masks2bbox([person_mask, obj_mask])
    crop_center = (bmin + bmax) // 2
    rh, rw = rgb.shape[:2]
    if rw > rh:
        resize_scale = 2048 / rw
        newsize = (2048, int(rh * resize_scale))
    else:
        resize_scale = 1536 / rh
        newsize = (int(rw * resize_scale), 1536)
    crop_center = np.round(resize_scale * crop_center)
    rgb = resize_linear_u8(rgb, newsize)
    person_mask = resize_linear_u8(person_mask, newsize)
    obj_mask = resize_linear_u8(obj_mask, newsize)
    cs = scale * np.array([crop_size, crop_size])
    rgb = resize_linear_u8(crop(rgb, crop_center, cs), img_size) / 255.
    person_mask = resize_linear_u8(crop(person_mask, crop_center, cs), img_size) / 255.
    obj_mask = resize_linear_u8(crop(obj_mask, crop_center, cs), img_size) / 255.
    mask_comb = (person_mask > 0.5) | (obj_mask > 0.5)
    rgb = rgb * np.expand_dims(mask_comb, -1)
    images = np.dstack((rgb, person_mask, obj_mask))
    return images.transpose((2, 0, 1)).astype(np.float32), crop_center, resize_scale


# ---- use_mean_center=True (recon_fit_coco.py:28): the patch is moved to the mean crop centre of the BEHAVE training set ----
MEAN_CROP_CENTER = np.array([1008., 995.])            # test_data.py:32


def pad_image(img, crop_center, mean_center=MEAN_CROP_CENTER):
    """test_data.py:133-160: paste the image into a float64 canvas (np.zeros' default dtype) so that crop_center lands on
    the mean crop centre; the canvas is at least 2048 x 1536 and the pasted part is clipped to that rectangle"""
    h, w = img.shape[:2]
    top_left = (mean_center - crop_center).astype(int)                 # truncation toward zero
    bottom_right = (np.array([w, h]) + top_left)
    kw, kh = 2048, 1536
    new_size = np.maximum(np.array([kw, kh]), bottom_right).astype(int)
    new_img = np.zeros((new_size[1], new_size[0], 3)) if img.ndim == 3 else np.zeros((new_size[1], new_size[0]))
    x1y1 = np.maximum(np.zeros(2), top_left).astype(int)
    x2y2 = np.minimum(np.array([kw, kh]), bottom_right).astype(int)
    x1, y1 = max(0, -top_left[0]), max(0, -top_left[1])
    x2, y2 = min(w, w - (bottom_right[0] - kw)), min(h, h - (bottom_right[1] - kh))
    new_img[x1y1[1]:x2y2[1], x1y1[0]:x2y2[0]] = img[y1:y2, x1:x2]
    return new_img


def resize_linear_f64(img, dsize):
    """cv2.resize(img, dsize) for float64 images, INTER_LINEAR: the PUBLISHED generic path of resize.cpp
    (resizeGeneric_<HResizeLinear<double, double, float>, VResizeLinear<double, double, float>>): the same pixel-centre
    mapping and border rules as the 8-bit path, FLOAT weights (1 - f, f), products and sums in double, the horizontal
    pass first; an exact 2 x 2 downscale takes the INTER_AREA fast path = (a + b + c + d) * 0.25.  UNPINNED like the rest."""
    img = np.asarray(img, dtype=np.float64)
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    x = img.reshape(sh, sw, -1)
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:
        s = (x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2]) * 0.25
        return s.reshape((dh, dw) + img.shape[2:])
    sx, fx = _axis_tables(sw, dw)
    fx = np.where(sx < 0, np.float32(0), fx)
    sx = np.where(sx < 0, 0, sx)
    fx = np.where(sx >= sw - 1, np.float32(0), fx)
    sx = np.where(sx >= sw - 1, sw - 1, sx)
    fx = fx.astype(np.float32)
    a0, a1 = (np.float32(1.0) - fx).astype(np.float64), fx.astype(np.float64)
    sx1 = np.minimum(sx + 1, sw - 1)
    rows = x[:, sx, :] * a0[None, :, None] + x[:, sx1, :] * a1[None, :, None]
    sy, fy = _axis_tables(sh, dh)
    fy = fy.astype(np.float32)
    b0, b1 = (np.float32(1.0) - fy).astype(np.float64), fy.astype(np.float64)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    out = rows[y0] * b0[:, None, None] + rows[y1] * b1[:, None, None]
    return out.reshape((dh, dw) + img.shape[2:])


def prepare_image_crop_mean_center(rgb, person_mask, obj_mask, scale, img_size=(512, 512), crop_size=1200):
    """test_data.py:59-125 with use_mean_center=True -> images (5,H,W) float32, crop_center (= the mean centre),
    resize_scale, old_center"""
    bmin, bmax = masks2bbox([person_mask, obj_mask])
    crop_center = (bmin + bmax) // 2
    rh, rw = rgb.shape[:2]
    if rw > rh:
        resize_scale = 2048 / rw
        newsize = (2048, int(rh * resize_scale))
    else:
        resize_scale = 1536 / rh
        newsize = (int(rw * resize_scale), 1536)
    crop_center = np.round(resize_scale * crop_center)
    rgb = resize_linear_u8(rgb, newsize)
    person_mask = resize_linear_u8(person_mask, newsize)
    obj_mask = resize_linear_u8(obj_mask, newsize)
    cs = scale * np.array([crop_size, crop_size])
    rgb, person_mask, obj_mask = (pad_image(a, crop_center) for a in (rgb, person_mask, obj_mask))
    old_center = crop_center.copy()
    crop_center = MEAN_CROP_CENTER.copy()
    rgb = resize_linear_f64(crop(rgb, crop_center, cs), img_size) / 255.
    person_mask = resize_linear_f64(crop(person_mask, crop_center, cs), img_size) / 255.
    obj_mask = resize_linear_f64(crop(obj_mask, crop_center, cs), img_size) / 255.
    mask_comb = (person_mask > 0.5) | (obj_mask > 0.5)
    rgb = rgb * np.expand_dims(mask_comb, -1)
    images = np.dstack((rgb, person_mask, obj_mask))
    return images.transpose((2, 0, 1)).astype(np.float32), crop_center, resize_scale, old_center

"""Network input from the decoded images, on the device.

Counterpart of TestData.prepare_image_crop (/root/reference/data/test_data.py:59-125) from the point where the RGB
image and the two masks are decoded uint8 arrays: bounding box of the masks -> crop centre, resize to the 2048-px space,
crop of `scale * 1200` px around the centre (zero padded), resize to the network input, /255, background masking,
channel stacking.  All pixel work runs in libchore_hip.so (csrc/image_prep.hip: chore_prep_masks2bbox / _resize_u8 /
_crop_compose / _crop_compose_mean); the host does the few scalar steps in numpy with the reference's expressions.
use_mean_center=True (the COCO loader, recon/recon_fit_coco.py:28): the patch is moved to the mean crop centre of the
BEHAVE training set (pad_image :133-160 -- a float64 canvas in the reference, a translated clipped lookup here) and cv2's
generic float resize applies.  `fullbody_crop` (:174-210, the crop scale from the openpose keypoints and the mocap mesh) runs on the
device since round 4; JPEG / PLY decoding stays with the reference's host code.

    prep = ImagePrep(image_size=(512, 512), crop_size=1200)
    images, crop_center, resize_scale, old_center = prep.prepare(rgb_u8, person_u8, obj_u8, scale)

cv2's resize arithmetic is restated (oracle/image_prep.py): parity with cv2 itself is UNPINNED -- OpenCV is not in this
image; against the restatement the kernels are bit-exact (tests/test_gpu_image_prep.py).
"""
import numpy as np
import torch

from .. import _lib


class ImagePrep:
    def __init__(self, image_size=(512, 512), crop_size=1200, use_mean_center=False, device="cuda:0"):
        self.use_mean_center = bool(use_mean_center)
        self.mean_crop_center = np.array([1008., 995.])      # test_data.py:32: computed from the BEHAVE training set
        if image_size[0] != image_size[1]:
            raise ValueError("the crop is square: image_size must be (S, S)")
        self.img_size, self.crop_size = tuple(image_size), float(crop_size)
        self.device = torch.device(device)

    def fullbody_crop(self, kpts, mocap_verts, body25_regressor, z0=2.2, camera=None):
        """the scale of the crop that makes the person look as if standing at depth z0   [data/test_data.py:174-210]
        kpts: (25,3) openpose keypoints (x, y, confidence) in the 2048-px image; mocap_verts: (V,3) vertices of the frame's
        FrankMocap mesh (what `load_mocap_mesh` reads); body25_regressor: dense (25,V) landmark regressor
        (`FitAssets.regressors()[0]`, the matrix of lib_smpl/body_landmark.py:62-66).  Returns a python float (one host read);
        (None, 1.0) when no keypoint has confidence, like the reference.  float64 on the device: the reference computes in numpy
        doubles."""
        from ..model.camera import KinectColorCamera
        dev = self.device
        pts = torch.as_tensor(np.asarray(kpts), dtype=torch.float64, device=dev)
        if float(pts[:, 2].sum()) == 0:
            return None, 1.0
        v = torch.as_tensor(np.asarray(mocap_verts), dtype=torch.float64, device=dev)
        reg = torch.as_tensor(np.asarray(body25_regressor), dtype=torch.float64, device=dev)
        cam = camera if camera is not None else KinectColorCamera(self.crop_size)
        v = v - v.mean(0) + torch.tensor([0.0, 0.0, float(z0)], dtype=torch.float64, device=dev)     # move to depth z0
        j3d = reg @ v                                                                                # (25,3) body keypoints
        px, py = cam.project_screen(j3d)
        valid = pts[:, 2] > 0.3
        j2d, j2d_mocap = pts[valid, :2], torch.cat([px, py], 1)[valid]

        def width(j, exp=1.1):                     # get_bbox :224-228
            return (j.max(0).values - j.min(0).values) * exp
        (w, h), (wm, hm) = width(j2d).tolist(), width(j2d_mocap).tolist()
        return w / wm if (w >= h and wm >= hm) else h / hm

    def _u8(self, a, ndim):
        t = torch.as_tensor(a)
        if t.dtype != torch.uint8 or t.dim() != ndim:
            raise ValueError(f"expected a uint8 array with {ndim} dimensions")
        return t.to(self.device).contiguous()

    def masks2bbox(self, masks, thres=127):
        """[data/base_data.py:92-112] -> (bmin, bmax) numpy int arrays (one host read)"""
        m = [self._u8(x, 2) for x in masks]
        if len(m) not in (1, 2):
            raise ValueError("one or two masks")
        H, W = m[0].shape
        h = _lib.handle(self.device.index or 0)
        out = torch.empty(4, dtype=torch.int32, device=self.device)
        s = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib.chore_prep_masks2bbox(h, m[0].data_ptr(), m[1].data_ptr() if len(m) == 2 else None, H, W, thres,
                                                  out.data_ptr(), s), h, "chore_prep_masks2bbox")
        b = out.cpu().numpy()
        return b[:2].astype(np.int64), b[2:].astype(np.int64)

    def resize(self, img, dsize):
        """cv2.resize(img, dsize=(width, height)) on a uint8 (H,W) or (H,W,C) image, INTER_LINEAR"""
        t = torch.as_tensor(img)
        x = self._u8(t, t.dim())
        sh, sw = x.shape[:2]
        C = 1 if x.dim() == 2 else x.shape[2]
        dw, dh = int(dsize[0]), int(dsize[1])
        out = torch.empty((dh, dw) + tuple(x.shape[2:]), dtype=torch.uint8, device=self.device)
        h = _lib.handle(self.device.index or 0)
        s = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib.chore_prep_resize_u8(h, x.data_ptr(), sh, sw, C, out.data_ptr(), dh, dw, s), h, "chore_prep_resize_u8")
        return out

    def prepare(self, rgb, person_mask, obj_mask, scale=1.0):
        """-> images (5,S,S) fp32 on the device, crop_center (2,), resize_scale, old_center   [test_data.py:59-125]"""
        rgb, pm, om = self._u8(rgb, 3), self._u8(person_mask, 2), self._u8(obj_mask, 2)
        bmin, bmax = self.masks2bbox([pm, om])
        width = bmax - bmin
        if width[0] > self.crop_size or width[1] > self.crop_size:
            raise AssertionError("crop too small for the bounding box of the masks: {}".format(width))
        crop_center = (bmin + bmax) // 2
        rh, rw = rgb.shape[:2]
        if rw > rh:
            resize_scale = 2048 / rw
            newsize = (2048, int(rh * resize_scale))
        else:
            resize_scale = 1536 / rh
            newsize = (int(rw * resize_scale), 1536)
        crop_center = np.round(resize_scale * crop_center)
        rgb, pm, om = self.resize(rgb, newsize), self.resize(pm, newsize), self.resize(om, newsize)
        size = scale * np.array([self.crop_size, self.crop_size])
        if self.use_mean_center:
            return self._prepare_mean(rgb, pm, om, crop_center, size, resize_scale)
        tl = np.round(crop_center - size / 2).astype(int)
        br = np.round(crop_center + size / 2).astype(int)
        S = self.img_size[0]
        images = torch.empty(5, S, S, dtype=torch.float32, device=self.device)
        h = _lib.handle(self.device.index or 0)
        s = torch.cuda.current_stream(self.device).cuda_stream
        H, W = rgb.shape[:2]
        _lib.check(_lib.lib.chore_prep_crop_compose(h, rgb.data_ptr(), pm.data_ptr(), om.data_ptr(), H, W, int(tl[0]), int(tl[1]),
                                                    int(br[0]), int(br[1]), S, images.data_ptr(), s), h, "chore_prep_crop_compose")
        return images, crop_center, resize_scale, crop_center.copy()

    def _prepare_mean(self, rgb, pm, om, crop_center, size, resize_scale):
        """use_mean_center=True: pad_image + change_crop_center (test_data.py:99-104,127-160), then crop / resize / compose"""
        old_center = crop_center.copy()
        mean = self.mean_crop_center.copy()
        tl = np.round(mean - size / 2).astype(int)
        br = np.round(mean + size / 2).astype(int)
        S = self.img_size[0]
        images = torch.empty(5, S, S, dtype=torch.float32, device=self.device)
        h = _lib.handle(self.device.index or 0)
        s = torch.cuda.current_stream(self.device).cuda_stream
        H, W = rgb.shape[:2]
        _lib.check(_lib.lib.chore_prep_crop_compose_mean(h, rgb.data_ptr(), pm.data_ptr(), om.data_ptr(), H, W, float(old_center[0]),
                                                         float(old_center[1]), float(mean[0]), float(mean[1]), int(tl[0]), int(tl[1]),
                                                         int(br[0]), int(br[1]), S, images.data_ptr(), s), h,
                   "chore_prep_crop_compose_mean")
        return images, mean, resize_scale, old_center

"""Kinect colour camera constants used by the fused query kernel.

Mirrors the constructor arithmetic of /root/reference/model/camera.py:26-42 (normalised
intrinsics multiplied back by the 2048-px image width in Python double precision) so the fp32
constants handed to the HIP kernel are bit-identical to what the reference multiplies into its
fp32 tensors.  The projection of the query points runs inside `chore_query_fwd` (csrc/query_fwd.hip); the three
methods of the reference class (`project_points`, `project_screen`, `normalize`, camera.py:44-88) are kept for its other
callers (2-D keypoint reprojection in recon_fit_base.project_points, visualisation): elementwise torch expressions in
the reference's operation order on whatever device the points live on.
"""
import struct

import torch


def _f32(x: float) -> float:
    """round a python double to the nearest fp32 (what torch does with a python scalar operand)"""
    return struct.unpack("f", struct.pack("f", x))[0]


class KinectColorCamera:
    def __init__(self, crop_size=1200, fx=979.7844 / 2048., fy=979.840 / 2048.,
                 cx=1018.952 / 2048., cy=779.486 / 2048., image_size=2048):
        self.fx, self.fy = fx, fy
        self.cx, self.cy = cx, cy
        self.width, self.height = image_size, int(image_size * 0.75)
        self.fx_px, self.fy_px = self.fx * image_size, self.fy * image_size
        self.cx_px, self.cy_px = self.cx * image_size, self.cy * image_size
        self.crop_size = crop_size

    def kernel_constants(self):
        """(fx_px, fy_px, cx_px, cy_px, half_crop, crop) as fp32-rounded python floats."""
        return (_f32(self.fx_px), _f32(self.fy_px), _f32(self.cx_px), _f32(self.cy_px),
                _f32(self.crop_size / 2), _f32(float(self.crop_size)))

    # ---- reference surface (model/camera.py:44-88) ----------------------------------------------------------------
    def project_points(self, points, offset=None):
        """(B,N,3) camera-space -> (B,3,N) [nx, ny, z] normalised to the crop around `offset` (B,2)"""
        px, py = self.project_screen(points)
        nx, ny = self.normalize(px, py, offset)
        return torch.cat([nx, ny, points[:, :, 2:3]], -1).transpose(1, 2)

    def project_screen(self, points, crop_center=None):
        """pinhole projection to pixel coordinates of the 2048-px Kinect image, or of the crop when crop_center is given"""
        if points.dim() == 3:
            x, y, z = points[:, :, 0:1], points[:, :, 1:2], points[:, :, 2:3]
        elif points.dim() == 2:
            x, y, z = points[:, 0:1], points[:, 1:2], points[:, 2:3]
        else:
            raise NotImplementedError("points must be (B,N,3) or (N,3)")
        px = self.fx_px * x / z + self.cx_px
        py = self.fy_px * y / z + self.cy_px
        if crop_center is not None:
            px = self.crop_size / 2 + px - crop_center[:, 0].unsqueeze(1).unsqueeze(1)
            py = self.crop_size / 2 + py - crop_center[:, 1].unsqueeze(1).unsqueeze(1)
        return px, py

    def normalize(self, px, py, offset=None):
        """pixel -> [-1, 1] of the crop around `offset` (B,2); px, py: (B,N,1)"""
        if offset is None:
            if not self.fx > 1.0:   # the reference asserts the same (normalised intrinsics cannot be used without a crop)
                raise AssertionError("error, trying to project using incompatible intrinsics")
            return 2 * px / self.width - 1, 2 * py / self.height - 1
        px = self.crop_size / 2 + px - offset[:, 0].unsqueeze(1).unsqueeze(1)
        py = self.crop_size / 2 + py - offset[:, 1].unsqueeze(1).unsqueeze(1)
        if px.shape[-1] != 1 or py.shape[-1] != 1:
            raise AssertionError("invalid shape of px / py found: {} {}".format(px.shape, py.shape))
        return 2 * px / self.crop_size - 1, 2 * py / self.crop_size - 1

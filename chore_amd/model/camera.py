"""Kinect colour camera constants used by the fused query kernel.

Mirrors the constructor arithmetic of /root/reference/model/camera.py:26-42 (normalised
intrinsics multiplied back by the 2048-px image width in Python double precision) so the fp32
constants handed to the HIP kernel are bit-identical to what the reference multiplies into its
fp32 tensors.  The projection itself runs inside `chore_query_fwd` (csrc/query_fwd.hip); this
class only owns the numbers.
"""
import struct


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

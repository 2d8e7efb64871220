"""Pixel-aligned feature sampling on the GPU (no heads).

`sample_features` is the counterpart of the reference's `index` (model/geometry.py:4-14, bound as
BasePIFuNet.index at model/BasePIFuNet.py:23) applied to both maps plus the z_feat channels, i.e.
the MLP input of model/chore.py:139-143, computed by chore_sample_features (csrc/query_fwd.hip).
"""
import ctypes

import torch

from .. import _lib


def sample_features(points, crop_center, feat, tmpx, cam6, dtype=_lib.F32, want_nxy=False):
    """points (B,N,3) cuda fp32; feat (B,256,H,W) / tmpx (B,64,H2,W2) channels-last views.
    Returns features (B,323,N) [view of a point-major buffer], in_img (B,N) bool, and optionally
    nxy (B,N,2)."""
    if not points.is_cuda:
        raise RuntimeError("chore_amd needs device tensors (no CPU path)")
    from .chore import _nhwc_ptr
    B, N, _ = points.shape
    dev = points.device
    h = _lib.handle(dev.index or 0)
    fp, FH, FW = _nhwc_ptr(feat, 256)
    tp, TH, TW = _nhwc_ptr(tmpx, 64)
    pts = points.float().contiguous()
    cc = crop_center.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty(B, N, 323, device=dev, dtype=torch.float32)
    nxy = torch.empty(B, N, 2, device=dev, dtype=torch.float32) if want_nxy else None
    inside = torch.empty(B, N, device=dev, dtype=torch.uint8)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib.chore_sample_features(h, pts.data_ptr(), cc.data_ptr(), B, N, fp, FH, FW, tp, TH, TW,
                                              dtype, cam6, out.data_ptr(),
                                              None if nxy is None else nxy.data_ptr(), inside.data_ptr(),
                                              stream), h, "chore_sample_features")
    res = (out.transpose(1, 2), inside.bool())
    return res + (nxy,) if want_nxy else res

"""ORACLE (test infrastructure only -- never imported by chore_amd/).

CPU restatement, in numpy fp32, of the stacked-hourglass encoder:
  HGFilter.forward     /root/reference/model/HGFilters.py:144-185
  HourGlass._forward   /root/reference/model/HGFilters.py:26-50
  ConvBlock.forward    /root/reference/model/net_util.py:374-396
and of the ATen ops they call (conv2d, group_norm(32 groups, eps 1e-5), avg_pool2d(2),
interpolate(bicubic, scale 2, align_corners=True; cubic convolution A=-0.75, border-clamped taps)).
Pinned against tests/golden/encoder_*.npz produced by importing the reference
(tests/golden/make_golden.py); agreement is to fp32 round-off (different summation order).
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

F32 = np.float32


def conv2d(x, w, b=None, stride=1, pad=0):
    """x (B,C,H,W), w (O,C,kh,kw) -> (B,O,Ho,Wo)"""
    x = np.asarray(x, F32)
    w = np.asarray(w, F32)
    O, C, kh, kw = w.shape
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    if kh == 1 and kw == 1 and stride == 1:
        y = np.einsum("oc,bchw->bohw", w[:, :, 0, 0], x, optimize=True)
    else:
        win = sliding_window_view(x, (kh, kw), axis=(2, 3))[:, :, ::stride, ::stride]  # B,C,Ho,Wo,kh,kw
        y = np.einsum("bchwij,ocij->bohw", win, w, optimize=True)
    y = y.astype(F32)
    if b is not None:
        y = y + np.asarray(b, F32)[None, :, None, None]
    return y.astype(F32)


def group_norm(x, gamma, beta, groups=32, eps=1e-5):
    x = np.asarray(x, F32)
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1).astype(np.float64)
    mean = xg.mean(-1, keepdims=True)
    var = xg.var(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    y = ((xg - mean) * rstd).reshape(B, C, H, W)
    y = y * np.asarray(gamma, np.float64)[None, :, None, None] + np.asarray(beta, np.float64)[None, :, None, None]
    return y.astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


def avg_pool2(x):
    x = np.asarray(x, F32)
    return ((x[:, :, 0::2, 0::2] + x[:, :, 0::2, 1::2] + x[:, :, 1::2, 0::2] + x[:, :, 1::2, 1::2]) * F32(0.25)).astype(F32)


def _cubic_coeffs(t, A=-0.75):
    """ATen get_cubic_upsample_coefficients"""
    def c1(x):  # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], -1)


def _bicubic_axis_tables(n_in, n_out):
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    src = np.arange(n_out, dtype=np.float64) * scale
    # ATen computes in float; fp32 here keeps floor() decisions identical
    src = src.astype(F32)
    i0 = np.floor(src).astype(np.int64)
    t = (src - i0).astype(F32).astype(np.float64)
    coef = _cubic_coeffs(t).astype(F32)
    idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
    return idx, coef


def bicubic_up2(x):
    """F.interpolate(x, scale_factor=2, mode='bicubic', align_corners=True)   [HGFilters.py:47]"""
    x = np.asarray(x, F32)
    B, C, H, W = x.shape
    iy, cy = _bicubic_axis_tables(H, 2 * H)
    ix, cx = _bicubic_axis_tables(W, 2 * W)
    # rows first then columns (separable; ATen interpolates x inside y)
    tmp = np.zeros((B, C, H, 2 * W), F32)
    for k in range(4):
        tmp += x[:, :, :, ix[:, k]] * cx[:, k][None, None, None, :]
    out = np.zeros((B, C, 2 * H, 2 * W), F32)
    for k in range(4):
        out += tmp[:, :, iy[:, k], :] * cy[:, k][None, None, :, None]
    return out.astype(F32)


class Encoder:
    """sd: dict 'image_filter.<...>' -> numpy arrays (reference state_dict names)"""

    def __init__(self, sd, num_stack=5, depth=2, prefix="image_filter."):
        self.sd, self.num_stack, self.depth, self.p = sd, num_stack, depth, prefix

    def _w(self, name):
        return self.sd[self.p + name]

    def gn(self, x, name):
        return group_norm(x, self._w(name + ".weight"), self._w(name + ".bias"))

    def conv_block(self, x, name, cin, cout):
        o1 = conv2d(relu(self.gn(x, name + ".bn1")), self._w(name + ".conv1.weight"), pad=1)
        o2 = conv2d(relu(self.gn(o1, name + ".bn2")), self._w(name + ".conv2.weight"), pad=1)
        o3 = conv2d(relu(self.gn(o2, name + ".bn3")), self._w(name + ".conv3.weight"), pad=1)
        cat = np.concatenate([o1, o2, o3], 1)
        if cin != cout:
            res = conv2d(relu(self.gn(x, name + ".bn4")), self._w(name + ".downsample.2.weight"))
        else:
            res = x
        return (cat + res).astype(F32)

    def hourglass(self, x, name, level):
        up1 = self.conv_block(x, f"{name}.b1_{level}", 256, 256)
        low1 = self.conv_block(avg_pool2(x), f"{name}.b2_{level}", 256, 256)
        if level > 1:
            low2 = self.hourglass(low1, name, level - 1)
        else:
            low2 = self.conv_block(low1, f"{name}.b2_plus_{level}", 256, 256)
        low3 = self.conv_block(low2, f"{name}.b3_{level}", 256, 256)
        return (up1 + bicubic_up2(low3)).astype(F32)

    def forward(self, images, collect=None):
        """images (B,5,H,W) -> (outputs[list of (B,256,H/4,W/4)], tmpx, normx)"""
        x = conv2d(images, self._w("conv1.weight"), self._w("conv1.bias"), stride=2, pad=3)
        x = relu(self.gn(x, "bn1"))
        tmpx = x
        x = avg_pool2(self.conv_block(x, "conv2", 64, 128))
        normx = x
        x = self.conv_block(x, "conv3", 128, 128)
        x = self.conv_block(x, "conv4", 128, 256)
        previous = x
        if collect is not None:
            collect["stem"] = previous
        outputs = []
        for i in range(self.num_stack):
            hg = self.hourglass(previous, f"m{i}", self.depth)
            ll = self.conv_block(hg, f"top_m_{i}", 256, 256)
            ll = conv2d(ll, self._w(f"conv_last{i}.weight"), self._w(f"conv_last{i}.bias"))
            ll = relu(self.gn(ll, f"bn_end{i}"))
            out = conv2d(ll, self._w(f"l{i}.weight"), self._w(f"l{i}.bias"))
            outputs.append(out)
            if i < self.num_stack - 1:
                previous = (previous + conv2d(ll, self._w(f"bl{i}.weight"), self._w(f"bl{i}.bias"))
                            + conv2d(out, self._w(f"al{i}.weight"), self._w(f"al{i}.bias"))).astype(F32)
        return outputs, tmpx, normx

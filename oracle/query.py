"""ORACLE (test infrastructure only -- never imported by chore_amd/).

CPU restatement, in numpy fp32, of CHORE.query:
  project_points   /root/reference/model/camera.py:44-88
  in_img, z_feat   /root/reference/model/chore.py:125-130
  index            /root/reference/model/geometry.py:4-14  (= ATen grid_sampler_2d bilinear / zeros /
                   align_corners=True; the arithmetic below follows ATen's CPU kernel:
                   ix=(x+1)*((W-1)/2), w=ix-floor(ix), e=1-w, value = fma(se,w*n, fma(sw,e*n, fma(ne,w*s,
                   nw*(e*s)))) -- the fused-multiply-add chain is what the AVX2 kernel executes and is
                   needed for bit-identical samples)
  decode           /root/reference/model/chore.py:74-85,156-167 (four 1x1-conv MLPs)
  OUT_DIST fill    /root/reference/model/chore.py:147-150
Pinned against tests/golden/query_*.npz, which were produced by importing the reference itself
(tests/golden/make_golden.py).  Projection, in_img and tap indices are required to match the
reference bit for bit; sampled values and head outputs to fp32 round-off.
"""
import numpy as np

F32 = np.float32

# Python-double constructor arithmetic of camera.py:26-38, then rounded to fp32 as torch does when a
# python scalar meets an fp32 tensor
FX_PX = F32((979.7844 / 2048.) * 2048)
FY_PX = F32((979.840 / 2048.) * 2048)
CX_PX = F32((1018.952 / 2048.) * 2048)
CY_PX = F32((779.486 / 2048.) * 2048)
CROP = 1200
OUT_DIST = F32(5.0)
HEAD_NAMES = ("df", "part_predictor", "pca_predictor", "center_predictor")


def project_points(points, crop_center, crop_size=CROP):
    """points (B,N,3) fp32, crop_center (B,2) fp32 -> nx, ny (B,N) fp32   [camera.py:64-78]"""
    points = np.asarray(points, F32)
    cc = np.asarray(crop_center, F32)
    x, y, z = points[..., 0], points[..., 1], points[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        px = (FX_PX * x) / z + CX_PX
        py = (FY_PX * y) / z + CY_PX
        half = F32(crop_size / 2)
        px = (half + px) - cc[:, 0:1]
        py = (half + py) - cc[:, 1:2]
        nx = (F32(2) * px) / F32(crop_size) - F32(1)
        ny = (F32(2) * py) / F32(crop_size) - F32(1)
    return nx.astype(F32), ny.astype(F32)


def in_image(nx, ny):
    return (nx >= -1.0) & (nx <= 1.0) & (ny >= -1.0) & (ny <= 1.0)


def tap_table(nx, ny, H, W):
    """integer tap coordinates, validity and bilinear weights for a (H,W) map
    returns dict x0,y0 (int64, floor of the unnormalised coordinate), valid (4,...) bool in the order
    nw,ne,sw,se and weights (4,...) fp32"""
    ix = (nx + F32(1)) * F32((W - 1) / 2)
    iy = (ny + F32(1)) * F32((H - 1) / 2)
    with np.errstate(invalid="ignore"):
        x0f, y0f = np.floor(ix), np.floor(iy)
        w = (ix - x0f).astype(F32)
        e = (F32(1) - w).astype(F32)
        n = (iy - y0f).astype(F32)
        s = (F32(1) - n).astype(F32)
    sane = np.isfinite(ix) & np.isfinite(iy) & (np.abs(ix) < 1e8) & (np.abs(iy) < 1e8)
    x0 = np.where(sane, x0f, -4).astype(np.int64)
    y0 = np.where(sane, y0f, -4).astype(np.int64)
    xv0, xv1 = (x0 >= 0) & (x0 < W), (x0 + 1 >= 0) & (x0 + 1 < W)
    yv0, yv1 = (y0 >= 0) & (y0 < H), (y0 + 1 >= 0) & (y0 + 1 < H)
    valid = np.stack([xv0 & yv0, xv1 & yv0, xv0 & yv1, xv1 & yv1])
    weights = np.stack([e * s, w * s, e * n, w * n]).astype(F32)
    return dict(x0=x0, y0=y0, valid=valid, weights=weights, wx=w, wy=n)


def _fma(a, b, c):
    """fp32 fused multiply-add (the product of two fp32 values is exact in fp64)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def index(feat, nx, ny):
    """feat (B,C,H,W) fp32; nx, ny (B,N) -> (B,C,N)   [geometry.py:4-14]"""
    feat = np.asarray(feat, F32)
    B, C, H, W = feat.shape
    t = tap_table(nx, ny, H, W)
    out = np.zeros((B, C, nx.shape[1]), F32)
    dx = (0, 1, 0, 1)
    dy = (0, 0, 1, 1)
    for b in range(B):
        acc = None
        for k in range(4):
            xs = np.clip(t["x0"][b] + dx[k], 0, W - 1)
            ys = np.clip(t["y0"][b] + dy[k], 0, H - 1)
            v = feat[b][:, ys, xs] * t["valid"][k][b].astype(F32)  # masked gather (zeros padding)
            wk = t["weights"][k][b]
            acc = (v * wk).astype(F32) if acc is None else _fma(v, wk, acc)
        out[b] = acc
    return out


def mlp(features, sd, name):
    """features (B,323,N); sd maps '<name>.{0,2,4,6}.{weight,bias}' -> arrays   [chore.py:74-85]"""
    h = np.asarray(features, F32)
    for li in (0, 2, 4, 6):
        W = np.asarray(sd[f"{name}.{li}.weight"], F32)[:, :, 0]
        b = np.asarray(sd[f"{name}.{li}.bias"], F32)
        h = np.einsum("oc,bcn->bon", W, h, optimize=True).astype(F32) + b[None, :, None]
        if li != 6:
            h = np.maximum(h, F32(0))
    return h.astype(F32)


def relu_margin(features, sd):
    """smallest |pre-activation| over every hidden unit of the four heads, per point (B,N).
    A point whose margin is ~1e-6 sits on a ReLU kink: its GRADIENT legitimately depends on the
    fp32 summation order, so gradient parity tests skip such points."""
    margin = None
    for name in HEAD_NAMES:
        h = np.asarray(features, F32)
        for li in (0, 2, 4):
            W = np.asarray(sd[f"{name}.{li}.weight"], F32)[:, :, 0]
            b = np.asarray(sd[f"{name}.{li}.bias"], F32)
            pre = np.einsum("oc,bcn->bon", W, h, optimize=True).astype(F32) + b[None, :, None]
            m = np.abs(pre).min(1)
            margin = m if margin is None else np.minimum(margin, m)
            h = np.maximum(pre, F32(0))
    return margin


def query(points, crop_center, feat, tmpx, sd):
    """Full CHORE.query for one feature map.  Returns df (B,2,N), pca (B,3,3,N), parts (B,14,N),
    centers (B,6,N), in_img (B,N) bool, features (B,323,N)."""
    points = np.asarray(points, F32)
    nx, ny = project_points(points, crop_center)
    inside = in_image(nx, ny)
    z_feat = np.stack([points[..., 0], points[..., 1], (points[..., 2] - F32(2.2)).astype(F32)], 1)
    f_img = index(feat, nx, ny)
    f_tmp = index(tmpx, nx, ny)
    features = np.concatenate([f_img, z_feat, f_tmp], 1).astype(F32)
    df = mlp(features, sd, "df")
    pca = mlp(features, sd, "pca_predictor")
    parts = mlp(features, sd, "part_predictor")
    centers = mlp(features, sd, "center_predictor")
    df = np.where(inside[:, None, :], df, OUT_DIST).astype(F32)
    B, _, N = df.shape
    return dict(df=df, pca=pca.reshape(B, 3, 3, N), parts=parts, centers=centers, in_img=inside,
                features=features, nx=nx, ny=ny)

"""ORACLE (test infrastructure only -- never imported by chore_amd/).

numpy restatement of the silhouette renderer the object fit uses: neural_renderer's projection +
silhouette rasterisation, forward and backward, as called by SilLossROI (recon/obj_pose_roi.py:159-177):
  projection              /root/reference/external/neural_renderer/neural_renderer/projection.py:6-43
  look_at                 .../look_at.py:6-62        (only needed to replay the reference's own known-answer tests)
  vertices_to_faces       .../vertices_to_faces.py:4-22
  render_silhouettes      .../renderer.py:119-152    (fill_back doubles the faces with reversed winding)
  rasterize_rgbad         .../rasterize.py:300-360   (alpha only; rows flipped: alpha[:, ::-1, :])
  forward face index map  .../cuda/rasterize_cuda_kernel.cu:24-215  (back-face culling, inside test in normalised
                          coordinates, clamped barycentric weights from the pixel-space inverse, perspective-correct z,
                          z-buffer with near/far)
  backward pixel map      .../cuda/rasterize_cuda_kernel.cu:290-549 (per face, edge and axis: walk the pixels the edge
                          crosses; 'out' pixels beyond the edge and 'in' pixels up to the opposite edge contribute
                          diff_grad / distance to the two edge vertices)
The reference rasteriser is CUDA-only (rasterize.py:261-262 raises on CPU tensors), so it cannot run in the build
container: this restatement is PINNED by the reference's own known-answer tests
(external/neural_renderer/tests/test_rasterize_silhouettes.py:37-108, the two analytic gradient cases; the Blender
image of test_case1 is not in the tree) -- tests/test_oracle_silhouette.py.  Arithmetic is float32 in the order of
the CUDA source so that the HIP kernels can be compared with it bit for bit at small sizes.

Deliberate difference: the reference bins faces into 4x4-pixel blocks with room for 512 faces and silently drops
the rest (rasterize.py:55-56, kernel :90-93); neither this restatement nor the HIP kernel drops faces.
"""
import numpy as np

F32 = np.float32
NEAR, FAR, EPS = F32(0.1), F32(100.0), F32(1e-4)   # rasterize.py:10-12 (rasterize_silhouettes uses the defaults)


def normalize(v, eps=1e-5):
    n = np.sqrt((v * v).sum(-1, keepdims=True))
    return v / np.maximum(n, eps)


def look_at(vertices, eye, at=(0, 0, 0), up=(0, 1, 0)):
    v = np.asarray(vertices, F32)
    B = v.shape[0]
    eye = np.broadcast_to(np.asarray(eye, F32), (B, 3))
    at = np.broadcast_to(np.asarray(at, F32), (B, 3))
    up = np.broadcast_to(np.asarray(up, F32), (B, 3))
    z = normalize(at - eye)
    x = normalize(np.cross(up, z))
    y = normalize(np.cross(z, x))
    r = np.stack([x, y, z], 1)                      # (B,3,3)
    return np.matmul(v - eye[:, None, :], r.transpose(0, 2, 1)).astype(F32)


def projection(vertices, K, R, t, dist_coeffs=None, orig_size=1.0, eps=1e-9):
    v = np.matmul(np.asarray(vertices, F32), np.asarray(R, F32).transpose(0, 2, 1)) + np.asarray(t, F32)
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    x_ = x / (z + F32(eps))
    y_ = y / (z + F32(eps))
    d = np.zeros((1, 5), F32) if dist_coeffs is None else np.asarray(dist_coeffs, F32)
    k1, k2, p1, p2, k3 = (d[:, None, i] for i in range(5))
    r = np.sqrt(x_ ** 2 + y_ ** 2)
    x__ = x_ * (1 + k1 * r ** 2 + k2 * r ** 4 + k3 * r ** 6) + 2 * p1 * x_ * y_ + p2 * (r ** 2 + 2 * x_ ** 2)
    y__ = y_ * (1 + k1 * r ** 2 + k2 * r ** 4 + k3 * r ** 6) + p1 * (r ** 2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
    h = np.stack([x__, y__, np.ones_like(z)], -1)
    uv = np.matmul(h, np.asarray(K, F32).transpose(0, 2, 1))
    u, vv = uv[..., 0], F32(orig_size) - uv[..., 1]
    u = 2 * (u - F32(orig_size) / 2) / F32(orig_size)
    vv = 2 * (vv - F32(orig_size) / 2) / F32(orig_size)
    return np.stack([u, vv, z], -1).astype(F32)


def vertices_to_faces(vertices, faces):
    return np.stack([vertices[b][faces[b]] for b in range(vertices.shape[0])]).astype(F32)   # (B,F,3,3)


def fill_back(faces):
    return np.concatenate([faces, faces[:, :, ::-1]], 1)


def _backside(f):
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])


def rasterize_fwd(faces, size, near=NEAR, far=FAR):
    """faces (B,F,3,3) -> face_index_map (B,size,size) int32 (-1 = none), alpha (B,size,size) float32 (rows NOT flipped)"""
    faces = np.asarray(faces, F32)
    B, Fn = faces.shape[:2]
    S = F32(size)
    fim = -np.ones((B, size, size), np.int32)
    yi, xi = np.meshgrid(np.arange(size, dtype=F32), np.arange(size, dtype=F32), indexing="ij")
    xp = ((2.0 * xi.astype(np.float64) + 1 - size) / size).astype(F32)   # evaluated in double, stored as float
    yp = ((2.0 * yi.astype(np.float64) + 1 - size) / size).astype(F32)
    for b in range(B):
        depth = np.full((size, size), far, F32)
        for fn in range(Fn):
            f = faces[b, fn].reshape(9)
            if _backside(f):
                continue
            p = (F32(0.5) * (faces[b, fn, :, :2] * S + S - F32(1))).astype(F32)     # pixel coordinates of the vertices
            inv = np.array([p[1, 1] - p[2, 1], p[2, 0] - p[1, 0], p[1, 0] * p[2, 1] - p[2, 0] * p[1, 1],
                            p[2, 1] - p[0, 1], p[0, 0] - p[2, 0], p[2, 0] * p[0, 1] - p[0, 0] * p[2, 1],
                            p[0, 1] - p[1, 1], p[1, 0] - p[0, 0], p[0, 0] * p[1, 1] - p[1, 0] * p[0, 1]], F32)
            den = (p[2, 0] * (p[0, 1] - p[1, 1]) + p[0, 0] * (p[1, 1] - p[2, 1])) + p[1, 0] * (p[2, 1] - p[0, 1])
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = (inv / F32(den)).astype(F32)
                out = (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) |
                       ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) |
                       ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
                w = [np.clip((inv[3 * k] * xi + inv[3 * k + 1] * yi) + inv[3 * k + 2], F32(0), F32(1)).astype(F32)
                     for k in range(3)]
                ws = (w[0] + w[1]) + w[2]
                w = [wk / ws for wk in w]
                zp = F32(1) / ((w[0] / f[2] + w[1] / f[5]) + w[2] / f[8])
                hit = (~out) & ~((zp <= near) | (far <= zp)) & (zp < depth)
            depth = np.where(hit, zp, depth).astype(F32)
            fim[b] = np.where(hit, fn, fim[b])
    return fim, (fim >= 0).astype(F32)


def rasterize_bwd(faces, fim, alpha, grad_alpha, eps=EPS):
    """gradient of the silhouettes w.r.t. the projected faces (B,F,3,3): the edge walk of the reference, scalar loops"""
    faces = np.asarray(faces, F32)
    B, Fn = faces.shape[:2]
    size = fim.shape[1]
    S = F32(size)
    g = np.zeros((B, Fn, 9), F32)
    for b in range(B):
        A, G, M = alpha[b], grad_alpha[b], fim[b]

        def px(axis, d0, d1):       # (row, col) of the pixel with coordinate d0 along the walk axis, d1 across it
            return (d1, d0) if axis == 0 else (d0, d1)

        for fn in range(Fn):
            f = faces[b, fn].reshape(9)
            if _backside(f):
                continue
            gf = np.zeros(9, F32)
            for e in range(3):
                pi = [(e + k) % 3 for k in range(3)]
                pp = np.array([[F32(0.5) * (f[3 * pi[k] + d] * S + S - F32(1)) for d in range(2)] for k in range(3)], F32)
                for axis in range(2):
                    p = pp[:, [axis, 1 - axis]]           # p[k][0] = walk axis, p[k][1] = across
                    if axis == 0:
                        direction = -1 if p[0, 0] < p[1, 0] else 1
                    else:
                        direction = 1 if p[0, 0] < p[1, 0] else -1
                    d0_from = int(max(np.ceil(min(p[0, 0], p[1, 0])), 0.0))
                    d0_to = int(min(max(p[0, 0], p[1, 0]), size - 1.0))
                    for d0 in range(d0_from, d0_to + 1):
                        with np.errstate(divide="ignore", invalid="ignore"):
                            cross = F32((p[1, 1] - p[0, 1]) / (p[1, 0] - p[0, 0]) * (F32(d0) - p[0, 0]) + p[0, 1])
                        if not np.isfinite(cross):
                            continue
                        d1_in = int(np.floor(cross)) if direction > 0 else int(np.ceil(cross))
                        d1_out = d1_in + direction
                        if not (0 <= d1_in < size and 0 <= d1_out < size):
                            continue
                        a_in, a_out = A[px(axis, d0, d1_in)], A[px(axis, d0, d1_out)]

                        def push(d1, diff):
                            if diff <= 0:
                                return
                            if p[1, 0] != d0:
                                dist = F32((p[1, 0] - p[0, 0]) / (p[1, 0] - F32(d0)) * (F32(d1) - cross) * F32(2) / S)
                                dist = dist + eps if dist > 0 else dist - eps
                                gf[pi[0] * 3 + (1 - axis)] -= F32(diff / dist)
                            if p[0, 0] != d0:
                                dist = F32((p[1, 0] - p[0, 0]) / (F32(d0) - p[0, 0]) * (F32(d1) - cross) * F32(2) / S)
                                dist = dist + eps if dist > 0 else dist - eps
                                gf[pi[1] * 3 + (1 - axis)] -= F32(diff / dist)

                        if M[px(axis, d0, d1_in)] == fn:        # 'out': pixels beyond the edge, up to the image border
                            lim = size - 1 if direction > 0 else 0
                            lo, hi = max(min(d1_out, lim), 0), min(max(d1_out, lim), size - 1)
                            for d1 in range(lo, hi + 1):
                                q = px(axis, d0, d1)
                                push(d1, F32((A[q] - a_in) * G[q]))
                        # 'in': pixels of this face from the edge to the opposite edge
                        if (F32(d0) - p[0, 0]) * (F32(d0) - p[2, 0]) < 0:
                            c2 = (p[2, 1] - p[0, 1]) / (p[2, 0] - p[0, 0]) * (F32(d0) - p[0, 0]) + p[0, 1]
                        else:
                            with np.errstate(divide="ignore", invalid="ignore"):
                                c2 = (p[1, 1] - p[2, 1]) / (p[1, 0] - p[2, 0]) * (F32(d0) - p[2, 0]) + p[2, 1]
                        if not np.isfinite(c2):
                            continue
                        lim = int(np.ceil(c2)) if direction > 0 else int(np.floor(c2))
                        lo, hi = max(min(d1_in, lim), 0), min(max(d1_in, lim), size - 1)
                        for d1 in range(lo, hi + 1):
                            q = px(axis, d0, d1)
                            if M[q] != fn:
                                continue
                            push(d1, F32((A[q] - a_out) * G[q]))
            g[b, fn] = gf
    return g.reshape(B, Fn, 3, 3)


def render_silhouettes(proj_vertices, faces, size, do_fill_back=True):
    """projected vertices (B,V,3) [u, v in [-1,1], z], faces (B,F,3) int -> images (B,size,size) as the renderer returns
    them (rows flipped), plus what the backward needs"""
    f = fill_back(faces) if do_fill_back else faces
    tri = vertices_to_faces(proj_vertices, f)
    fim, alpha = rasterize_fwd(tri, size)
    return alpha[:, ::-1, :].copy(), (tri, f, fim, alpha)


def render_silhouettes_bwd(ctx, grad_images, n_vertices):
    """gradient w.r.t. the projected vertices (B,V,3): un-flip, edge walk, scatter the face gradients to the vertices"""
    tri, f, fim, alpha = ctx
    gt = rasterize_bwd(tri, fim, alpha, np.ascontiguousarray(grad_images[:, ::-1, :], F32))
    gv = np.zeros((tri.shape[0], n_vertices, 3), F32)
    for b in range(tri.shape[0]):
        np.add.at(gv[b], f[b].reshape(-1), gt[b].reshape(-1, 3))
    return gv

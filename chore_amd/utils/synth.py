"""Deterministic synthetic weights and inputs (there is no network for checkpoints or datasets).

The stock init of the reference (N(0,0.02), /root/reference/model/net_util.py:218-251) drives
activations to ~3e-4 and does not exercise the numerics, so parity fixtures and the benchmark use
variance-preserving weights instead.  Every tensor is drawn from a numpy RandomState seeded by
crc32(name)+seed, so the same state dict can be rebuilt bit for bit on any machine from
(name, shape) alone -- the golden fixtures under tests/golden/ only store inputs and outputs.
"""
import zlib

import numpy as np


def _canonical(name: str) -> str:
    # `downsample.0` is the same module as `bn4` (shared GroupNorm)
    return name.replace("downsample.0.", "bn4.")


def synth_tensor(name: str, shape, seed: int = 0) -> np.ndarray:
    name = _canonical(name)
    rs = np.random.RandomState((zlib.crc32(name.encode()) + seed) & 0xFFFFFFFF)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) >= 3:  # conv weight (O,C,k[,k])
        fan_in = int(np.prod(shape[1:]))
        return (rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    is_norm = ".bn" in name or name.startswith("bn") or "bn_end" in name
    if leaf == "weight":  # GroupNorm gamma
        return rs.uniform(0.5, 1.5, shape).astype(np.float32)
    if is_norm:  # GroupNorm beta
        return (rs.standard_normal(shape) * 0.1).astype(np.float32)
    return (rs.standard_normal(shape) * 0.05).astype(np.float32)  # conv bias


def synth_state_dict(spec, seed: int = 0):
    """spec: iterable of (name, shape) -> {name: np.ndarray}"""
    return {n: synth_tensor(n, s, seed) for n, s in spec}


def load_synth_weights(model, seed: int = 0):
    """fill a torch module's parameters in place with the synthetic weights"""
    import torch
    with torch.no_grad():
        for name, p in model.state_dict(keep_vars=True).items():
            p.copy_(torch.from_numpy(synth_tensor(name, p.shape, seed)).to(p.device))
    return model


def synth_images(B, H=512, W=512, seed=0):
    """(B,5,H,W) fp32: RGB in [0,1] and two binary masks (the data/test_data.py:107-125 contract)"""
    rs = np.random.RandomState(1000 + seed)
    img = rs.random_sample((B, 5, H, W)).astype(np.float32)
    img[:, 3:] = (img[:, 3:] > 0.5).astype(np.float32)
    return img


def synth_points(B, N, seed=1):
    """(B,N,3) fp32 camera-space points around the crop: x = -0.0246+U(-1.38,1.38),
    y = 0.4839+U(-1.38,1.38), z ~ U(1.95,2.45); ~95 % project inside the 1200-px crop at
    crop_center (1008, 995)"""
    rs = np.random.RandomState(2000 + seed)
    p = np.empty((B, N, 3), np.float32)
    p[..., 0] = -0.0246 + rs.uniform(-1.38, 1.38, (B, N))
    p[..., 1] = 0.4839 + rs.uniform(-1.38, 1.38, (B, N))
    p[..., 2] = rs.uniform(1.95, 2.45, (B, N))
    return p


CROP_CENTER = (1008.0, 995.0)


# ---- synthetic SMPL-H body model (the licensed SMPLH_{male,female}.pkl files are not redistributable) ----
SMPLH_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                 20, 22, 23, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35,
                 21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50]


def synth_smplh_model(seed: int = 0, V: int = 6890, num_betas: int = 10):
    """A random model with SMPL-H's shapes and kinematic tree (52 joints, 459 pose blend directions):
    v_template (V,3), shapedirs (V,3,num_betas), posedirs (V,3,459), J_regressor (52,V) dense with rows
    summing to 1, weights (V,52) with 4 non-zeros per vertex summing to 1, parents (52,)."""
    rs = np.random.RandomState(4000 + seed)
    J = len(SMPLH_PARENTS)
    vt = (rs.standard_normal((V, 3)) * np.array([0.25, 0.5, 0.12])).astype(np.float32)
    shapedirs = (rs.standard_normal((V, 3, num_betas)) * 0.02).astype(np.float32)
    posedirs = (rs.standard_normal((V, 3, (J - 1) * 9)) * 0.005).astype(np.float32)
    jr = np.zeros((J, V), np.float32)
    for j in range(J):
        idx = rs.choice(V, 24, replace=False)
        w = rs.random_sample(24).astype(np.float32)
        jr[j, idx] = w / w.sum()
    weights = np.zeros((V, J), np.float32)
    for v0 in range(0, V, 1024):
        n = min(1024, V - v0)
        idx = np.stack([rs.choice(J, 4, replace=False) for _ in range(n)])
        w = rs.random_sample((n, 4)).astype(np.float32) + 0.05
        w /= w.sum(1, keepdims=True)
        weights[np.arange(v0, v0 + n)[:, None], idx] = w
    return dict(v_template=vt, shapedirs=shapedirs, posedirs=posedirs, J_regressor=jr, weights=weights,
                parents=np.array(SMPLH_PARENTS, np.int32))


def synth_smpl_params(B, seed=0):
    """pose (B,156), betas (B,10), trans (B,3): moderate random articulation around the 2.2 m depth"""
    rs = np.random.RandomState(5000 + seed)
    pose = (rs.standard_normal((B, 156)) * 0.25).astype(np.float32)
    betas = (rs.standard_normal((B, 10)) * 0.8).astype(np.float32)
    trans = (rs.standard_normal((B, 3)) * 0.1 + np.array([0.0, 0.3, 2.2])).astype(np.float32)
    return pose, betas, trans


def uv_ellipsoid(rings=82, segments=84, radii=(0.25, 0.85, 0.18), center=(0.0, 0.0, 0.0)):
    """closed UV ellipsoid: V = 2 + rings * segments, F = 2 * segments * rings.  The defaults give a body-sized blob
    with exactly the SMPL counts (6 890 vertices, 13 776 faces)."""
    v = [(0.0, 1.0, 0.0)]
    for r in range(1, rings + 1):
        th = np.pi * r / (rings + 1)
        for s in range(segments):
            ph = 2 * np.pi * s / segments
            v.append((np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)))
    v.append((0.0, -1.0, 0.0))
    f = []
    last = len(v) - 1
    for s in range(segments):
        f.append((0, 1 + (s + 1) % segments, 1 + s))
        b = 1 + (rings - 1) * segments
        f.append((last, b + s, b + (s + 1) % segments))
    for r in range(rings - 1):
        a, b = 1 + r * segments, 1 + (r + 1) * segments
        for s in range(segments):
            s1 = (s + 1) % segments
            f.append((a + s, a + s1, b + s))
            f.append((a + s1, b + s1, b + s))
    return np.asarray(v) * np.asarray(radii) + np.asarray(center, dtype=np.float64), np.asarray(f, dtype=np.int64)


def synth_smplh_surface_model(seed: int = 0, num_betas: int = 10):
    """Like synth_smplh_model, but a SURFACE: the template is a closed body-sized blob with SMPL's vertex / face counts
    and the blend shapes and skinning weights vary smoothly over it, so posed bodies stay free of self-intersections.
    (A random vertex cloud with random faces is fine for the arithmetic, but every other triangle pair of it
    'collides': the interpenetration term then does 60x the work it does on a body.)  Returns the model dict + 'f'."""
    rs = np.random.RandomState(4500 + seed)
    J = len(SMPLH_PARENTS)
    vt, faces = uv_ellipsoid()
    V = vt.shape[0]
    jc = np.stack([0.12 * np.cos(2.4 * np.arange(J)), 0.8 - 1.6 * np.arange(J) / (J - 1), 0.08 * np.sin(2.4 * np.arange(J))], 1)
    d2 = ((vt[:, None, :] - jc[None, :, :]) ** 2).sum(-1)                        # (V,J)
    w = np.exp(-d2 / (2 * 0.15 ** 2))
    cut = np.sort(w, 1)[:, -4][:, None]                                          # four joints per vertex
    w = np.where(w >= cut, w, 0.0)
    weights = (w / w.sum(1, keepdims=True)).astype(np.float32)
    jr = np.zeros((J, V), np.float32)
    for j in range(J):
        idx = np.argsort(d2[:, j])[:24]
        jr[j, idx] = 1.0 / 24
    # smooth blend shapes: low-frequency functions of the template position
    k = rs.standard_normal((num_betas, 3, 3)) * 2.0
    ph = rs.uniform(0, 6.28, (num_betas, 3))
    shapedirs = np.stack([0.02 * np.sin(vt @ k[b].T + ph[b]) for b in range(num_betas)], -1).astype(np.float32)   # (V,3,nb)
    kp = rs.standard_normal(((J - 1) * 9, 3, 3)) * 2.0
    php = rs.uniform(0, 6.28, ((J - 1) * 9, 3))
    posedirs = np.stack([0.001 * np.sin(vt @ kp[i].T + php[i]) for i in range((J - 1) * 9)], -1).astype(np.float32)
    return dict(v_template=vt.astype(np.float32), shapedirs=shapedirs, posedirs=posedirs, J_regressor=jr, weights=weights,
                parents=np.array(SMPLH_PARENTS, np.int32), f=faces)

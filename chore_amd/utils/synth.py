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

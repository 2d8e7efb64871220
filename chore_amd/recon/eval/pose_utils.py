"""Procrustes alignment for the evaluation, on the GPU (fp64).

Counterpart of /root/reference/recon/eval/pose_utils.py: `compute_transform` (:145-180), `compute_similarity_transform`
(:103-143), `compute_similarity_transform_batch` (:182-187), `reconstruction_error` (:189-198) and `ProcrusteAlign`
(:13-100; meshes are any objects with `.v` (V,3) and `.f`, returned as objects of the same class built with
`type(m)(v=..., f=...)`).  The arithmetic runs in chore_eval_procrustes / chore_eval_apply_similarity
(csrc/eval_metrics.hip)."""
import numpy as np
import torch

from ... import _lib


def _solve(S1, S2, device="cuda:0"):
    """S1, S2 (N,3) -> device tensor [R (9), t (3), scale]"""
    dev = torch.device(device)
    h = _lib.handle(dev.index or 0)
    a = torch.as_tensor(np.ascontiguousarray(S1, dtype=np.float64), device=dev)
    b = torch.as_tensor(np.ascontiguousarray(S2, dtype=np.float64), device=dev)
    if a.shape != b.shape or a.dim() != 2 or a.shape[1] != 3:
        raise ValueError("expected two (N,3) point sets of equal size")
    prm = torch.empty(13, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib.chore_eval_procrustes(h, a.data_ptr(), b.data_ptr(), a.shape[0], prm.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), h, "chore_eval_procrustes")
    return prm, a


def _apply(prm, pts):
    dev = prm.device
    h = _lib.handle(dev.index or 0)
    p = pts if torch.is_tensor(pts) else torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float64), device=dev)
    out = torch.empty_like(p)
    _lib.check(_lib.lib.chore_eval_apply_similarity(h, p.data_ptr(), p.shape[0], prm.data_ptr(), out.data_ptr(),
                                                    torch.cuda.current_stream(dev).cuda_stream), h,
               "chore_eval_apply_similarity")
    return out.cpu().numpy()


def _as_n3(S):
    S = np.asarray(S)
    transposed = S.shape[0] != 3 and S.shape[0] != 2       # the reference's test (pose_utils.py:111,147): (N,3) input
    return (S if transposed else S.T), transposed


def compute_transform(S1, S2, device="cuda:0"):
    """-> R (3,3), t (3,1), scale, transposed   [pose_utils.py:145-180]"""
    P1, transposed = _as_n3(S1)
    P2, _ = _as_n3(S2)
    prm, _ = _solve(P1, P2, device)
    v = prm.cpu().numpy()
    return v[:9].reshape(3, 3), v[9:12].reshape(3, 1), float(v[12]), transposed


def compute_similarity_transform(S1, S2, device="cuda:0"):
    """S1 aligned to S2: scale R S1 + t, in the layout S1 came in   [pose_utils.py:103-143]"""
    P1, transposed = _as_n3(S1)
    P2, _ = _as_n3(S2)
    prm, a = _solve(P1, P2, device)
    hat = _apply(prm, a)
    return hat if transposed else hat.T


def compute_similarity_transform_batch(S1, S2, device="cuda:0"):
    S1_hat = np.zeros_like(S1)
    for i in range(S1.shape[0]):
        S1_hat[i] = compute_similarity_transform(S1[i], S2[i], device)
    return S1_hat


def reconstruction_error(S1, S2, reduction="mean", device="cuda:0"):
    """Procrustes-aligned mean per-point error   [pose_utils.py:189-198]"""
    S1_hat = compute_similarity_transform_batch(S1, S2, device)
    re = np.sqrt(((S1_hat - S2) ** 2).sum(axis=-1)).mean(axis=-1)
    if reduction == "mean":
        re = re.mean()
    elif reduction == "sum":
        re = re.sum()
    return re


class ProcrusteAlign:
    """procrustes align   [pose_utils.py:13-100]"""

    def __init__(self, smpl_only=False, device="cuda:0"):
        self.warned = False
        self.smpl_only = smpl_only
        self.device = device

    def get_transform(self, ref_meshes, recon_meshes):
        ref_v = np.concatenate([m.v for m in ref_meshes], 0)
        recon_v = np.concatenate([m.v for m in recon_meshes], 0)
        if ref_v.shape == recon_v.shape and not self.smpl_only:
            prm, _ = _solve(recon_v, ref_v, self.device)
        else:
            if not self.warned:
                print("Warning: align using only smpl meshes!")
                self.warned = True
            prm, _ = _solve(recon_meshes[0].v, ref_meshes[0].v, self.device)
        return prm, recon_v

    @staticmethod
    def _split(points, like):
        out, offset = [], 0
        for m in like:
            out.append(type(m)(v=points[offset:offset + len(m.v)].copy(), f=np.array(m.f).copy()))
            offset += len(m.v)
        return out

    def align_meshes(self, ref_meshes, recon_meshes):
        prm, recon_v = self.get_transform(ref_meshes, recon_meshes)
        return self._split(_apply(prm, recon_v), recon_meshes)

    def align_neural_recon(self, ref_meshes, recon_meshes, neural_recons):
        """alignment found on the reconstructed meshes, applied to other point sets (split like the reference's meshes)"""
        prm, _ = self.get_transform(ref_meshes, recon_meshes)
        pts = _apply(prm, np.concatenate([x.v for x in neural_recons], 0))
        out, last = [], 0
        for m_ref, m in zip(ref_meshes, recon_meshes):
            L = last + len(m_ref.v)
            out.append(type(m)(v=pts[last:L].copy(), f=np.array(m.f).copy()))
            last = L
        return out

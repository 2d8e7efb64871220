"""Pose priors of the SMPL fit (Mahalanobis distances).

Restates th_Mahalanobis.__call__ (/root/reference/lib_smpl/th_smpl_prior.py:32-39) and HandPrior.__call__
(/root/reference/lib_smpl/th_hand_prior.py:63-72).  The reference re-reads three pickles from disk and
re-uploads them on EVERY optimisation step (recon/recon_fit_base.py:530-535); here the tensors are built
once and stay on the device.
"""
import pickle as pkl
from os.path import join

import numpy as np
import torch


class BodyPrior:
    """||(pose[:, prefix:end] - mean) @ precision||^2"""

    def __init__(self, mean, precision, prefix=3, end=66, device="cuda"):
        self.mean = torch.as_tensor(np.asarray(mean), dtype=torch.float32, device=device).unsqueeze(0)
        self.prec = torch.as_tensor(np.asarray(precision), dtype=torch.float32, device=device)
        self.prefix, self.end = prefix, end

    @classmethod
    def from_assets(cls, assets_root, device="cuda"):
        dat = pkl.load(open(join(assets_root, "priors/body_prior.pkl"), "rb"))
        return cls(dat["mean"], dat["precision"], device=device)

    def __call__(self, pose, prior_weight=1.0):
        t = pose[:, self.prefix:self.end] - self.mean
        t2 = torch.matmul(t, self.prec) * prior_weight
        return (t2 * t2).sum(dim=1)


class HandPrior:
    HAND_POSE_NUM = 45

    def __init__(self, mean, lhand_prec, rhand_prec, prefix=66, device="cuda"):
        self.prefix = prefix
        self.mean = torch.as_tensor(np.asarray(mean), dtype=torch.float32, device=device).unsqueeze(0)
        self.lhand_prec = torch.as_tensor(np.asarray(lhand_prec), dtype=torch.float32, device=device).unsqueeze(0)
        self.rhand_prec = torch.as_tensor(np.asarray(rhand_prec), dtype=torch.float32, device=device).unsqueeze(0)

    @classmethod
    def from_assets(cls, assets_root, device="cuda"):
        lh = pkl.load(open(join(assets_root, "priors", "lh_prior.pkl"), "rb"))
        rh = pkl.load(open(join(assets_root, "priors", "rh_prior.pkl"), "rb"))
        return cls(np.concatenate([lh["mean"], rh["mean"]], 0), lh["precision"], rh["precision"], device=device)

    def __call__(self, full_pose):
        t = full_pose[:, self.prefix:] - self.mean
        lh = torch.matmul(t[:, :self.HAND_POSE_NUM], self.lhand_prec)
        rh = torch.matmul(t[:, self.HAND_POSE_NUM:], self.rhand_prec)
        # shapes as in the reference: (B,45) @ (1,45,45) broadcasts to (1,B,45); the two hands are
        # concatenated along the BATCH axis and summed over it -> (1,45).  The caller takes the mean, i.e. the
        # "hand" loss is sum over frames and hands / 45 (reference quirk, th_hand_prior.py:68-72).
        t2 = torch.cat([lh, rh], dim=1)
        return (t2 * t2).sum(dim=1)


def synthetic_priors(seed=0, device="cuda"):
    """deterministic stand-ins with the asset shapes (body 63-d, hands 2 x 45-d) for tests / benchmarks"""
    rs = np.random.RandomState(6000 + seed)
    body = BodyPrior(rs.standard_normal(63) * 0.1, np.tril(rs.standard_normal((63, 63)) * 0.3) + np.eye(63), device=device)
    hand = HandPrior(rs.standard_normal(90) * 0.1, np.eye(45) + rs.standard_normal((45, 45)) * 0.05,
                     np.eye(45) + rs.standard_normal((45, 45)) * 0.05, device=device)
    return body, hand

"""SMPL-H instances from a complete set of parameters.

Counterpart of SMPLHGenerator.get_smplh (/root/reference/lib_smpl/smpl_generator.py:85-99): FrankMocap predicts the 72
SMPL pose parameters, the 90 hand parameters of SMPL-H are filled with the mean hand pose of the GRAB prior
(lib_smpl/th_hand_prior.py:35-41).  The body model, regressors and the mean hand pose come from a `FitAssets`
(recon/assets.py) instead of module-level paths.
"""
import numpy as np
import torch

from .const import SMPLH_HANDPOSE_START, SMPLH_POSE_PRAMS_NUM
from .wrapper_pytorch import SMPLPyTorchWrapperBatch


class SMPLHGenerator:
    @staticmethod
    def get_smplh(poses, betas, trans, gender, device="cuda:0", assets=None, layer=None):
        """poses (B,72) or (B,156), betas (B,10) numpy; trans (B,3) tensor -> SMPLPyTorchWrapperBatch on `device`"""
        if assets is None:
            raise ValueError("get_smplh needs the FitAssets that provide the body model")
        poses = np.asarray(poses)
        B, n = poses.shape
        if n != SMPLH_POSE_PRAMS_NUM:
            if n != 72:
                raise AssertionError("using unknown source of smpl poses")
            pose_init = torch.zeros((B, SMPLH_POSE_PRAMS_NUM))
            pose_init[:, :n] = torch.tensor(poses, dtype=torch.float32)
            pose_init[:, SMPLH_HANDPOSE_START:] = torch.tensor(np.asarray(assets.mean_hand_pose()), dtype=torch.float)
        else:
            pose_init = torch.tensor(poses, dtype=torch.float32)
        betas = torch.tensor(np.asarray(betas), dtype=torch.float32)
        model = layer if layer is not None else assets.smpl_model(gender)
        trans = torch.as_tensor(trans).detach().float().cpu().clone()
        return SMPLPyTorchWrapperBatch(model, B, betas, pose_init, trans, gender=gender, num_betas=10, hands=True,
                                       regressors=assets.regressors()).to(device)

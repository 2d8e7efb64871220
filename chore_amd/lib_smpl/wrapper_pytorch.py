"""Batched SMPL-H wrappers with optimisable parameters, on top of the HIP LBS layer.

Same roles and attribute names as /root/reference/lib_smpl/wrapper_pytorch.py:
`SMPLPyTorchWrapperBatch` (:23-90) holds whole `pose/betas/trans/offsets` parameters,
`SMPLPyTorchWrapperBatchSplitParams` (:93-218) splits them into `global_pose / body_pose / hand_pose /
top_betas / other_betas / trans` so the fit can optimise subsets, `get_landmarks()` returns the body-25 /
face / hand keypoints.  Differences: the body model arrives as arrays (see lib_smpl/smpl_layer.py), the
landmark regressors are dense (K,V) device tensors applied by one kernel (chore_landmarks_fwd; the reference loops
torch.sparse.mm over the batch, lib_smpl/torch_functions.py:52-76), and `get_landmarks` reuses the
vertices of the preceding `forward()` when the parameters have not changed instead of running LBS a
second time (reference quirk, wrapper_pytorch.py:186).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .const import BODY_POSE_NUM, GLOBAL_POSE_NUM, HAND_POSE_NUM, SMPL_HAND_POSE_NUM, TOP_BETA_NUM
from .smpl_layer import SMPL_Layer
from ..utils.paths import SMPL_ASSETS_ROOT, SMPL_MODEL_ROOT  # noqa: F401  (lib_smpl/smpl_generator.py:15 imports both from here)


def _as_layer(model, gender="male", num_betas=10):
    """`model`: a SMPL_Layer, the model arrays (dict), or -- like the reference's `model_root` -- a folder that holds
    SMPLH_<gender>.pkl / .npz (read by recon/assets.load_smpl_model)"""
    if isinstance(model, SMPL_Layer):
        return model
    if isinstance(model, (str, bytes)) or hasattr(model, "__fspath__"):
        import os
        from ..recon.assets import load_smpl_model
        root = os.fspath(model)
        for ext in (".npz", ".pkl"):
            f = os.path.join(root, f"SMPLH_{gender}{ext}")
            if os.path.isfile(f):
                return SMPL_Layer.from_arrays(load_smpl_model(f, num_betas), gender=gender)
        raise FileNotFoundError(f"no SMPLH_{gender}.pkl / .npz under {root}")
    return SMPL_Layer.from_arrays(model)


def synthetic_regressors(V=6890, seed=0):
    """dense stand-ins for assets/{body25,face,hand}_regressor.pkl: (25,V), (70,V), (42,V), rows sum to 1"""
    rs = np.random.RandomState(7000 + seed)
    out = []
    for k in (25, 70, 42):
        r = np.zeros((k, V), np.float32)
        for i in range(k):
            idx = rs.choice(V, 12, replace=False)
            w = rs.random_sample(12).astype(np.float32)
            r[i, idx] = w / w.sum()
        out.append(r)
    return out


_TORCH_LANDMARKS = bool(os.environ.get("CHORE_LANDMARKS_TORCH"))    # A/B switch: the product as a library GEMM


class _LandmarkFn(torch.autograd.Function):
    """lm (B,R,3) = reg (R,V) @ verts (B,V,3) through chore_landmarks_fwd / _bwd (csrc/smpl_lbs.hip)"""

    @staticmethod
    def forward(ctx, reg, verts):
        from .. import _lib
        verts = verts.contiguous()
        B, V, _ = verts.shape
        R = reg.shape[0]
        out = torch.empty(B, R, 3, device=verts.device, dtype=torch.float32)
        h = _lib.handle(verts.device.index or 0)
        _lib.check(_lib.lib.chore_landmarks_fwd(h, reg.data_ptr(), verts.data_ptr(), R, V, B, out.data_ptr(),
                                                torch.cuda.current_stream(verts.device).cuda_stream), h, "chore_landmarks_fwd")
        ctx.save_for_backward(reg)
        ctx.dims = (R, V, B)
        return out

    @staticmethod
    def backward(ctx, g):
        from .. import _lib
        reg, = ctx.saved_tensors
        R, V, B = ctx.dims
        g = g.contiguous()
        dv = torch.empty(B, V, 3, device=g.device, dtype=torch.float32)
        h = _lib.handle(g.device.index or 0)
        _lib.check(_lib.lib.chore_landmarks_bwd(h, reg.data_ptr(), g.data_ptr(), R, V, B, dv.data_ptr(),
                                                torch.cuda.current_stream(g.device).cuda_stream), h, "chore_landmarks_bwd")
        return None, dv


def landmarks(reg, verts):
    if not (verts.is_cuda and reg.is_cuda and verts.dtype == torch.float32 and reg.dtype == torch.float32 and reg.is_contiguous()):
        raise RuntimeError("landmarks: fp32 device tensors expected (there is no CPU path)")
    return _LandmarkFn.apply(reg, verts)


class _Landmarks:
    """landmark regressors + a one-entry memo of the LBS result.

    The reference re-runs the whole SMPL layer inside get_landmarks (wrapper_pytorch.py:186-189), so a 'kpts' step
    evaluates LBS three times on identical parameters (recon_fit_behave.py:293-337).  The parameters only change
    through in-place updates (optimiser step, copy_), which bump their version counters: while the counters stand
    still the previous result -- the same tensors, so gradients of all uses accumulate into one backward -- is
    returned.  Values are identical to recomputing; only the order of the gradient summation differs."""

    def _set_regressors(self, regressors):
        b25, face, hand = regressors if regressors is not None else synthetic_regressors(self.smpl.num_verts)
        t = lambda a: a.detach().float() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        self.register_buffer("body25_reg", t(b25))
        self.register_buffer("face_reg", t(face))
        self.register_buffer("hand_reg", t(hand))
        # the three regressors stacked (25 + 70 + 42 rows): one product, split afterwards
        self.register_buffer("all_reg", torch.cat([self.body25_reg, self.face_reg, self.hand_reg], 0).contiguous(), persistent=False)
        self._memo = None
        self._lm_memo = None

    def _memo_key(self):
        ps = list(self.parameters())
        return (torch.is_grad_enabled(),) + tuple((p.data_ptr(), p._version) for p in ps)

    def _lbs(self, compute):
        key = self._memo_key()
        if self._memo is not None and self._memo[0] == key:
            return self._memo[1]
        out = compute()
        self._memo = (key, out)
        return out

    def forget(self):
        """drop the memo (it holds an autograd graph)"""
        self._memo = None
        self._lm_memo = None

    @property
    def faces(self):
        """(F,3) long triangle list: what the caller handed in, else the model's (the reference reads it from the model
        file, wrapper_pytorch.py:63); follows the module across devices"""
        f = self.__dict__.get("_faces")
        if f is None:
            return getattr(self.smpl, "th_faces", None)
        return f.to(self.smpl.th_v_template.device) if torch.is_tensor(f) else f

    @faces.setter
    def faces(self, f):
        self.__dict__["_faces"] = f

    def landmarks_all(self):
        """(B, 25 + 70 + 42, 3): the rows of the three regressors in one tensor (body-25 first)"""
        verts = self.forward()[0]
        # like the vertices, the landmarks of unchanged parameters are the same tensors (a 'kpts' step asks twice:
        # recon_fit_behave.py:300-306), keyed on the vertices they were computed from
        if self._lm_memo is not None and self._lm_memo[0] is verts:
            lm = self._lm_memo[1]
        else:
            lm = torch.matmul(self.all_reg, verts) if _TORCH_LANDMARKS else landmarks(self.all_reg, verts)
            self._lm_memo = (verts, lm)
        return lm

    def get_landmarks(self):
        lm = self.landmarks_all()
        n1, n2 = self.body25_reg.shape[0], self.face_reg.shape[0]
        return lm[:, :n1], lm[:, n1:n1 + n2], lm[:, n1 + n2:]


class SMPLPyTorchWrapperBatch(nn.Module, _Landmarks):
    def __init__(self, model, batch_sz, betas=None, pose=None, trans=None, offsets=None, faces=None, gender="male",
                 hands=True, num_betas=10, regressors=None):
        super().__init__()
        self.smpl = _as_layer(model, gender, num_betas)
        self.model_root = self.smpl
        J3 = 3 * self.smpl.num_joints
        self.betas = nn.Parameter(torch.zeros(batch_sz, num_betas) if betas is None else betas)
        self.pose = nn.Parameter(torch.zeros(batch_sz, J3) if pose is None else pose)
        self.trans = nn.Parameter(torch.zeros(batch_sz, 3) if trans is None else trans)
        self.offsets = nn.Parameter(torch.zeros(batch_sz, self.smpl.num_verts, 3) if offsets is None else offsets)
        self.faces, self.gender, self.hands = faces, gender, hands
        self._set_regressors(regressors)

    @property
    def device(self):
        return self.pose.device

    def forward(self):
        return self._lbs(lambda: self.smpl(self.pose, th_betas=self.betas, th_trans=self.trans, th_offsets=self.offsets))


class SMPLPyTorchWrapperBatchSplitParams(nn.Module, _Landmarks):
    def __init__(self, model, batch_sz, top_betas=None, other_betas=None, global_pose=None, body_pose=None,
                 hand_pose=None, trans=None, offsets=None, faces=None, gender="male", hands=True, num_betas=10,
                 regressors=None):
        super().__init__()
        self.smpl = _as_layer(model, gender, num_betas)
        self.model_root = self.smpl
        hp = HAND_POSE_NUM if hands else SMPL_HAND_POSE_NUM
        z = torch.zeros
        self.top_betas = nn.Parameter(z(batch_sz, TOP_BETA_NUM) if top_betas is None else top_betas)
        self.other_betas = nn.Parameter(z(batch_sz, num_betas - TOP_BETA_NUM) if other_betas is None else other_betas)
        self.global_pose = nn.Parameter(z(batch_sz, GLOBAL_POSE_NUM) if global_pose is None else global_pose)
        self.body_pose = nn.Parameter(z(batch_sz, BODY_POSE_NUM) if body_pose is None else body_pose)
        self.hand_pose = nn.Parameter(z(batch_sz, hp) if hand_pose is None else hand_pose)
        self.trans = nn.Parameter(z(batch_sz, 3) if trans is None else trans)
        self.offsets = nn.Parameter(z(batch_sz, self.smpl.num_verts, 3) if offsets is None else offsets)
        self.faces, self.gender, self.hands = faces, gender, hands
        self._set_regressors(regressors)
        self._cat()

    def _cat(self):
        self.betas = torch.cat([self.top_betas, self.other_betas], dim=1)
        self.pose = torch.cat([self.global_pose, self.body_pose, self.hand_pose], dim=1)

    def forward(self):
        def compute():
            self._cat()
            return self.smpl(self.pose, th_betas=self.betas, th_trans=self.trans, th_offsets=self.offsets)
        return self._lbs(compute)

    @staticmethod
    def from_smpl(smpl: SMPLPyTorchWrapperBatch):
        """split wrapper whose parameters are VIEWS of `smpl`'s parameter storage, like in the reference
        (wrapper_pytorch.py:199-218 wraps slices of smpl.pose.data / betas.data / trans.data in nn.Parameter; slicing
        and the same-device .to() copy nothing).  The optimiser's in-place updates of the split parameters therefore
        write through to `smpl` -- in particular the betas 2..9, which copy_smpl_params never copies back
        (recon_fit_base.py:682-690), do arrive in the SMPL that optimize_smpl returns.  Writes through these views are
        invisible to the version counters of `smpl`'s own parameters: call smpl.forget() before evaluating it again
        (copy_smpl_params does)."""
        B = smpl.pose.shape[0]
        p, b = smpl.pose.data, smpl.betas.data
        g, bp = GLOBAL_POSE_NUM, GLOBAL_POSE_NUM + BODY_POSE_NUM
        return SMPLPyTorchWrapperBatchSplitParams(
            smpl.smpl, B, trans=smpl.trans.data, top_betas=b[:, :TOP_BETA_NUM], other_betas=b[:, TOP_BETA_NUM:],
            global_pose=p[:, :g], body_pose=p[:, g:bp], hand_pose=p[:, bp:], faces=smpl.faces, gender=smpl.gender,
            hands=smpl.hands, num_betas=b.shape[1], regressors=(smpl.body25_reg, smpl.face_reg, smpl.hand_reg)
        ).to(smpl.device)

"""SMPL / SMPL-H layer computed by the HIP kernels behind chore_smpl_lbs_fwd / chore_smpl_lbs_bwd.

Keeps the call surface of the reference layer
(/root/reference/lib_smpl/smplpytorch/smplpytorch/pytorch/smpl_layer.py:72-175):
    verts, jtr, v_posed, naked = layer(th_pose_axisang, th_betas=..., th_trans=..., th_offsets=None, scale=1.)
and its buffer names (th_v_template, th_shapedirs, th_posedirs, th_J_regressor, th_weights,
kintree_parents, num_joints).  The reference constructor reads the licensed SMPL pkl through chumpy;
here the model arrays are handed in directly (`SMPL_Layer.from_arrays(dict)`), so a caller that owns the
pkl loads it however it likes.  Gradients flow to pose, betas and trans (the quantities
recon/recon_fit_behave.py optimises); `th_offsets` is treated as a constant.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


class _LBSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, betas, trans, offsets, layer, scale):
        B = pose.shape[0]
        dev = pose.device
        h = _lib.handle(dev.index or 0)
        V, J, NB = layer.num_verts, layer.num_joints, layer.num_betas
        arena = layer._arena(dev)
        ws = torch.empty(_lib.lib.chore_smpl_workspace_bytes(V, J, NB, B), dtype=torch.uint8, device=dev)
        verts = torch.empty(B, V, 3, device=dev)
        joints = torch.empty(B, J, 3, device=dev)
        v_posed = torch.empty(B, V, 3, device=dev)
        naked = torch.empty(B, V, 3, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_smpl_lbs_fwd(h, arena.data_ptr(), V, J, NB, pose.data_ptr(), betas.data_ptr(),
                                               trans.data_ptr(), None if offsets is None else offsets.data_ptr(),
                                               float(scale), B, verts.data_ptr(), joints.data_ptr(),
                                               v_posed.data_ptr(), naked.data_ptr(), ws.data_ptr(), stream),
                   h, "chore_smpl_lbs_fwd")
        ctx.save_for_backward(pose, v_posed, ws, arena)
        ctx.layer, ctx.scale = layer, float(scale)
        ctx.mark_non_differentiable(v_posed, naked)
        ctx.set_materialize_grads(False)      # an unused output arrives as None (a NULL pointer for the kernels), not as a zero tensor
        return verts, joints, v_posed, naked

    @staticmethod
    def backward(ctx, g_verts, g_joints, _g_vp, _g_nk):
        pose, v_posed, ws, arena = ctx.saved_tensors
        layer = ctx.layer
        B = pose.shape[0]
        dev = pose.device
        h = _lib.handle(dev.index or 0)
        V, J, NB = layer.num_verts, layer.num_joints, layer.num_betas
        gv = None if g_verts is None else g_verts.contiguous().float()
        gj = None if g_joints is None else g_joints.contiguous().float()
        dpose = torch.empty(B, 3 * J, device=dev)
        dbetas = torch.empty(B, NB, device=dev)
        dtrans = torch.empty(B, 3, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_smpl_lbs_bwd(h, arena.data_ptr(), V, J, NB, pose.data_ptr(), ctx.scale, B,
                                               v_posed.data_ptr(), None if gv is None else gv.data_ptr(),
                                               None if gj is None else gj.data_ptr(), dpose.data_ptr(),
                                               dbetas.data_ptr(), dtrans.data_ptr(), ws.data_ptr(), stream),
                   h, "chore_smpl_lbs_bwd")
        return dpose, dbetas, dtrans, None, None, None


class SMPL_Layer(nn.Module):
    def __init__(self, v_template, shapedirs, posedirs, J_regressor, weights, parents, faces=None, hands=True,
                 gender="male", center_idx=None):
        super().__init__()
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).contiguous()  # noqa: E731
        self.register_buffer("th_v_template", f32(v_template).reshape(1, -1, 3))
        self.register_buffer("th_shapedirs", f32(shapedirs))
        self.register_buffer("th_posedirs", f32(posedirs))
        self.register_buffer("th_J_regressor", f32(J_regressor))
        self.register_buffer("th_weights", f32(weights))
        self.register_buffer("th_betas", torch.zeros(1, self.th_shapedirs.shape[2]))
        if faces is not None:
            self.register_buffer("th_faces", torch.as_tensor(np.asarray(faces)).long())
        self.kintree_parents = [int(p) for p in parents]
        self.num_joints = len(self.kintree_parents)
        self.num_verts = self.th_v_template.shape[1]
        self.num_betas = self.th_shapedirs.shape[2]
        self.hands, self.gender, self.center_idx = hands, gender, center_idx
        if self.th_posedirs.shape != (self.num_verts, 3, 9 * (self.num_joints - 1)):
            raise ValueError("posedirs must be (V,3,9(J-1))")
        self._packed = None

    @classmethod
    def from_arrays(cls, model, **kw):
        return cls(model["v_template"], model["shapedirs"], model["posedirs"], model["J_regressor"],
                   model["weights"], model["parents"], faces=model.get("f"), **kw)

    def _arena(self, device):
        if self._packed is not None and self._packed[0] == str(device):
            return self._packed[1]
        if not self.th_posedirs.is_cuda:
            raise RuntimeError("SMPL_Layer buffers must live on the GPU: call .to(device) first")
        h = _lib.handle(device.index or 0)
        V, J, NB = self.num_verts, self.num_joints, self.num_betas
        arena = torch.empty(_lib.lib.chore_smpl_arena_bytes(V, J, NB), dtype=torch.uint8, device=device)
        parents = (ctypes.c_int * J)(*[max(p, 0) if i else 0 for i, p in enumerate(self.kintree_parents)])
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(_lib.lib.chore_smpl_pack(h, V, J, NB, self.th_v_template.data_ptr(), self.th_shapedirs.data_ptr(),
                                            self.th_posedirs.data_ptr(), self.th_J_regressor.data_ptr(),
                                            self.th_weights.data_ptr(), parents, arena.data_ptr(), stream),
                   h, "chore_smpl_pack")
        torch.cuda.current_stream(device).synchronize()   # `parents` is a host buffer: keep it alive until copied
        self._packed = (str(device), arena)
        return arena

    def forward(self, th_pose_axisang, th_betas=None, th_trans=None, th_offsets=None, scale=1.):
        if not th_pose_axisang.is_cuda:
            raise RuntimeError("chore_amd SMPL_Layer needs device tensors (no CPU path)")
        B = th_pose_axisang.shape[0]
        dev = th_pose_axisang.device
        pose = th_pose_axisang.float().contiguous()
        if pose.shape[1] != 3 * self.num_joints:
            raise ValueError(f"pose must be (B,{3 * self.num_joints})")
        betas = self.th_betas.expand(B, -1) if th_betas is None or th_betas.numel() == 1 else th_betas
        betas = betas.float().contiguous()
        trans = torch.zeros(B, 3, device=dev) if th_trans is None else th_trans.float().expand(B, 3).contiguous()
        offs = None if th_offsets is None else th_offsets.detach().float().contiguous()
        return _LBSFn.apply(pose, betas, trans, offs, self, scale)

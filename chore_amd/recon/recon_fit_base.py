"""Fitting maths of CHORE on the GPU: SO(3) projection, object transform, loss terms.

Counterpart of /root/reference/recon/recon_fit_base.py (ReconFitterBase) for the methods that sit on the
optimisation hot path; file I/O, mesh viewers and dataset glue of that class are out of scope.
Method names, signatures and loss-dict keys are the reference's, so recon_fit_behave-style drivers read
the same.  Heavy steps run in libchore_hip.so: the field queries (chore_query_fwd/bwd_points), SMPL-H LBS
(chore_smpl_lbs_*) and the SO(3) projection (chore_so3_project_*); the remaining terms are a few
reductions over (B,N) tensors expressed with torch ops on the device.
"""
import torch
import torch.nn.functional as F

from .. import _lib
from ..lib_smpl.const import SMPL_PARTS_NUM, SMPL_POSE_PRAMS_NUM  # noqa: F401
from ..lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatchSplitParams
from ..model.camera import KinectColorCamera


class _SO3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mat):
        B = mat.shape[0]
        dev = mat.device
        h = _lib.handle(dev.index or 0)
        m = mat.float().contiguous()
        R = torch.empty_like(m)
        aux = torch.empty(_lib.lib.chore_so3_aux_bytes(B), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_so3_project_fwd(h, m.data_ptr(), B, R.data_ptr(), aux.data_ptr(), stream), h,
                   "chore_so3_project_fwd")
        ctx.save_for_backward(aux)
        return R

    @staticmethod
    def backward(ctx, g):
        aux, = ctx.saved_tensors
        B = g.shape[0]
        dev = g.device
        h = _lib.handle(dev.index or 0)
        g = g.float().contiguous()
        dM = torch.empty_like(g)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_so3_project_bwd(h, aux.data_ptr(), g.data_ptr(), B, dM.data_ptr(), stream), h,
                   "chore_so3_project_bwd")
        return dM


class _ContactFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hum, obj, df_hum_o, df_obj_h, part_logits, label_h, thres):
        dev = hum.device
        if not hum.is_cuda:
            raise RuntimeError("chore_amd needs device tensors (no CPU path)")
        h = _lib.handle(dev.index or 0)
        B, Nh, _ = hum.shape
        No, P = obj.shape[1], part_logits.shape[1]
        hum_c, obj_c = hum.detach().float().contiguous(), obj.detach().float().contiguous()
        lab = label_h.to(device=dev, dtype=torch.int32).contiguous()
        ws = torch.empty(_lib.lib.chore_contact_workspace_bytes(B, Nh, No, P), dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        # keep the converted inputs in locals until the launch is enqueued: a temporary freed inside the argument
        # list hands its block to the next temporary and the two pointers alias
        dfh, dfo = df_hum_o.float().contiguous(), df_obj_h.float().contiguous()
        logits = part_logits.float().contiguous()
        _lib.check(_lib.lib.chore_contact_fwd(h, hum_c.data_ptr(), obj_c.data_ptr(), dfh.data_ptr(), dfo.data_ptr(),
                                              lab.data_ptr(), logits.data_ptr(), B, Nh, No, P, float(thres),
                                              loss.data_ptr(), ws.data_ptr(), stream), h, "chore_contact_fwd")
        ctx.save_for_backward(hum_c, obj_c, lab, ws)
        ctx.P = P
        return loss

    @staticmethod
    def backward(ctx, g):
        hum, obj, lab, ws = ctx.saved_tensors
        dev = hum.device
        h = _lib.handle(dev.index or 0)
        B, Nh, _ = hum.shape
        No = obj.shape[1]
        d_hum, d_obj = torch.empty_like(hum), torch.empty_like(obj)
        g = g.float().contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_contact_bwd(h, hum.data_ptr(), obj.data_ptr(), lab.data_ptr(), B, Nh, No, ctx.P,
                                              g.data_ptr(), ws.data_ptr(), d_hum.data_ptr(), d_obj.data_ptr(), stream),
                   h, "chore_contact_bwd")
        return d_hum, d_obj, None, None, None, None, None


class _CollisionFn(torch.autograd.Function):
    """per-batch interpenetration sums of a triangle mesh (chore_collision_fwd / _bwd, csrc/collision.hip)"""

    @staticmethod
    def forward(ctx, verts, faces):
        dev = verts.device
        if not verts.is_cuda:
            raise RuntimeError("chore_amd needs device tensors (no CPU path)")
        h = _lib.handle(dev.index or 0)
        B, V, _ = verts.shape
        F_ = faces.shape[0]
        vc = verts.detach().float().contiguous()
        ws = torch.empty(_lib.lib.chore_collision_workspace_bytes(B, V, F_), dtype=torch.uint8, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        gverts = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_collision_fwd(h, vc.data_ptr(), faces.data_ptr(), B, V, F_, loss.data_ptr(),
                                                gverts.data_ptr(), None, ws.data_ptr(), stream), h, "chore_collision_fwd")
        ctx.save_for_backward(gverts)
        return loss

    @staticmethod
    def backward(ctx, g):
        (gverts,) = ctx.saved_tensors
        dev = gverts.device
        h = _lib.handle(dev.index or 0)
        B, V, _ = gverts.shape
        g = g.float().contiguous()
        dverts = torch.empty_like(gverts)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_collision_bwd(h, gverts.data_ptr(), g.data_ptr(), B, V, dverts.data_ptr(), stream), h,
                   "chore_collision_bwd")
        return dverts, None


class ReconFitterBase:
    def __init__(self, device="cuda:0", net_in_size=512, crop_size=1200, z_0=2.2, obj_scale=1.0, part_labels=None,
                 body_prior=None, hand_prior=None, debug=False, scan_verts=None, scan_faces=None):
        # template mesh of the object (self.scan.v / self.scan.f in the reference, recon_fit_base.py:79): needed by the
        # interpenetration term only
        self.scan_verts = None if scan_verts is None else torch.as_tensor(scan_verts, dtype=torch.float32,
                                                                          device=device)
        self.scan_faces = None if scan_faces is None else torch.as_tensor(scan_faces, dtype=torch.long, device=device)
        self._comb_faces = None
        self.device = torch.device(device)
        self.camera = KinectColorCamera(crop_size)
        self.net_in_size = net_in_size
        self.z_0 = z_0
        self.obj_scale = obj_scale
        self.debug = debug
        self.part_labels = part_labels          # (6890,) long: assets/smpl_parts_dense.pkl in the reference
        self.body_prior, self.hand_prior = body_prior, hand_prior

    # ---- SO(3) ------------------------------------------------------------------------------------
    @staticmethod
    def project_so3(mat):
        """(B,3,3) -> closest rotation U diag(1,1,det(UV^T)) V^T   [recon_fit_base.py:168-188]"""
        if mat.shape[1:] != (3, 3):
            raise ValueError(f"invalid shape {tuple(mat.shape)}")
        if not mat.is_cuda:
            raise RuntimeError("chore_amd needs device tensors (no CPU path)")
        return _SO3Fn.apply(mat)

    @staticmethod
    def decopose_axis(rot, no_rand=False, noise=None):
        """[recon_fit_base.py:374-384] the 1e-4*U[0,1) perturbation is drawn with the CPU generator, like
        the reference (so seeded runs consume the same random stream); `noise` lets tests replay it"""
        if no_rand:
            return ReconFitterBase.project_so3(rot)
        if noise is None:
            noise = torch.rand(rot.shape[0], 3, 3)
        return ReconFitterBase.project_so3(rot + 1e-4 * noise.to(rot.device))

    @staticmethod
    def inverse(mat):
        tr = torch.bmm(mat.transpose(2, 1), mat)
        return torch.bmm(torch.inverse(tr), mat.transpose(2, 1))

    @staticmethod
    def init_object_orientation(tgt_axis, src_axis):
        return ReconFitterBase.decopose_axis(torch.bmm(ReconFitterBase.inverse(src_axis), tgt_axis))

    def transform_obj_verts(self, verts, obj_R, obj_t, obj_s):
        """rotate, translate, THEN scale (reference order, recon_fit_base.py:367-371)"""
        verts = torch.bmm(verts, obj_R) + obj_t.unsqueeze(1)
        return verts * obj_s.unsqueeze(1).unsqueeze(1)

    # ---- interpenetration (parity unpinned, see oracle/collision.py) --------------------
    def smpl_obj_collision(self, smpl_verts, smpl_faces, obj_verts, obj_faces):
        """mean over the batch of the penetration loss of the concatenated mesh   [recon_fit_base.py:610-624]"""
        comb_verts = torch.cat([smpl_verts, obj_verts], 1)
        key = (smpl_faces.data_ptr(), obj_faces.data_ptr(), smpl_verts.shape[1])
        if self._comb_faces is None or self._comb_faces[0] != key:
            comb = torch.cat([smpl_faces.to(self.device).long(), obj_faces.to(self.device).long() + smpl_verts.shape[1]], 0)
            self._comb_faces = (key, comb.to(torch.int32).contiguous())
        return torch.mean(_CollisionFn.apply(comb_verts, self._comb_faces[1]))

    def compute_collision_loss(self, smpl_verts, smpl_faces, obj_R, obj_t, obj_s):
        """collision loss between the SMPL and the object template mesh   [recon_fit_base.py:626-639]"""
        if self.scan_verts is None or self.scan_faces is None:
            raise RuntimeError("compute_collision_loss needs the object template mesh (scan_verts / scan_faces)")
        scan = self.scan_verts.unsqueeze(0).expand(obj_R.shape[0], -1, -1)
        verts = self.transform_obj_verts(scan, obj_R, obj_t, obj_s)
        return self.smpl_obj_collision(smpl_verts, smpl_faces, verts, self.scan_faces)

    def transform_object(self, object_init, rot, obj_t, obj_s):
        return self.transform_obj_verts(object_init, self.decopose_axis(rot), obj_t, obj_s)

    # ---- loss terms --------------------------------------------------------------------------------
    @staticmethod
    def sum_dict(loss_dict, weight_dict, it):
        return torch.stack([weight_dict[k](v, it) for k, v in loss_dict.items()]).sum()

    def compute_obj_loss(self, data_dict, loss_dict, model, obj_s, object):
        model.query(object, **data_dict["query_dict"])
        preds = model.get_preds()
        loss_dict["object"] = torch.clamp(preds[0][:, 1:2, :], max=0.8).mean()
        loss_dict["scale"] = torch.mean((obj_s - self.obj_scale) ** 2)
        return preds

    def compute_prior_loss(self, loss_dict, smpl, nobeta=False):
        if not nobeta:
            loss_dict["beta"] = torch.mean(smpl.betas ** 2)
        loss_dict["pose"] = torch.mean(self.body_prior(smpl.pose[:, :72]))
        loss_dict["hand"] = torch.mean(self.hand_prior(smpl.pose))

    def compute_df_h_loss(self, data_dict, loss_dict, model, smpl_verts):
        model.query(smpl_verts, **data_dict["query_dict"])
        df_pred, _, parts_pred, centers_pred = model.get_preds()
        loss_dict["df_h"] = torch.clamp(df_pred[:, 0:1, :], max=0.1).mean()
        return df_pred, parts_pred, centers_pred

    def compute_smpl_center_pred(self, data_dict, model, smpl):
        with torch.no_grad():
            verts = smpl()[0]
            model.query(verts, **data_dict["query_dict"])
            return torch.mean(model.get_preds()[3][:, :3], -1)

    def smplz_loss(self, J, loss_dict):
        loss_dict["smplz"] = torch.mean((J[:, 8, 2] - self.z_0) ** 2)

    def project_points(self, joints3d, crop_center=None):
        c = self.camera
        x, y, z = joints3d[..., 0:1], joints3d[..., 1:2], joints3d[..., 2:3]
        px = c.fx_px * x / z + c.cx_px
        py = c.fy_px * y / z + c.cy_px
        if crop_center is not None:
            px = c.crop_size / 2 + px - crop_center[:, 0].unsqueeze(1).unsqueeze(1)
            py = c.crop_size / 2 + py - crop_center[:, 1].unsqueeze(1).unsqueeze(1)
        return torch.cat([px, py], -1) * self.net_in_size / c.crop_size

    def projection_loss(self, joints3d, joints2d, crop_center):
        proj = self.project_points(joints3d, crop_center)
        loss = F.mse_loss(proj[:, :, :2], joints2d[:, :, :2], reduction="none")
        return torch.mean(torch.sum(loss, dim=-1) * joints2d[:, :, 2])

    def compute_kpts_loss(self, data_dict, loss_dict, smpl):
        J, _, _ = smpl.get_landmarks()
        loss_dict["j2d"] = self.projection_loss(J, data_dict["body_kpts"], data_dict["query_dict"]["crop_center"])

    def scale_body_kpts(self, kpts, resize_scale, crop_scale, crop_center):
        pxy = kpts[:, :, :2] * resize_scale.unsqueeze(1).unsqueeze(1)
        crop_org = crop_scale * self.camera.crop_size
        pxy = pxy - crop_center.unsqueeze(1) + crop_org.unsqueeze(1).unsqueeze(1) / 2
        pxy = pxy * self.net_in_size / crop_org.unsqueeze(1).unsqueeze(1)
        return torch.cat([pxy, kpts[:, :, 2:3]], -1)

    def compute_contact_loss(self, df_hum_o, df_obj_h, object, smpl_verts, loss_dict, part_o=None):
        """[recon_fit_base.py:553-608 + pytorch3d chamfer_distance :605-607] pair human / object contact points by
        part label and take the bidirectional squared nearest-neighbour distance -- one fixed-shape device
        computation (chore_contact_fwd/bwd) instead of per-frame, per-part Python loops over ragged clouds, so the
        step has no host synchronisation.  The term is always present; it is 0 where the reference omits it."""
        loss_dict["contact"] = _ContactFn.apply(smpl_verts, object, df_hum_o.detach(), df_obj_h.detach(),
                                                part_o.detach(), self.part_labels, 0.08)

    def split_smpl(self, smpl):
        return SMPLPyTorchWrapperBatchSplitParams.from_smpl(smpl)

    @staticmethod
    def copy_smpl_params(split_smpl, smpl):
        smpl.pose.data[:, :3] = split_smpl.global_pose.data
        smpl.pose.data[:, 3:66] = split_smpl.body_pose.data
        smpl.pose.data[:, 66:] = split_smpl.hand_pose.data
        smpl.betas.data[:, :2] = split_smpl.top_betas.data   # (other_betas are not copied back: reference quirk, :682-690)
        smpl.trans.data = split_smpl.trans.data
        smpl.forget()   # writes through .data are invisible to the version counters the LBS memo is keyed on
        return smpl

    @staticmethod
    def get_smpl_height(smpl):
        with torch.no_grad():
            v = smpl()[0]
            return v[:, :, 1].max(1)[0] - v[:, :, 1].min(1)[0]

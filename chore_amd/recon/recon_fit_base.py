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


class ReconFitterBase:
    def __init__(self, device="cuda:0", net_in_size=512, crop_size=1200, z_0=2.2, obj_scale=1.0, part_labels=None,
                 body_prior=None, hand_prior=None, debug=False):
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

    @staticmethod
    def _chamfer(clouds_a, clouds_b):
        """pytorch3d.loss.chamfer_distance defaults on ragged clouds (squared L2 nearest neighbour, mean over
        the points of a cloud, mean over clouds, both directions summed) -- the reference calls it at
        recon_fit_base.py:605-607; pytorch3d is not vendored, so this is pinned by its documented
        definition (brute force; contact clouds are a few hundred points)"""
        da, db = [], []
        for a, b in zip(clouds_a, clouds_b):
            d = torch.cdist(a.unsqueeze(0), b.unsqueeze(0)).squeeze(0) ** 2
            da.append(d.min(1)[0].mean())
            db.append(d.min(0)[0].mean())
        return torch.stack(da).mean() + torch.stack(db).mean()

    def compute_contact_loss(self, df_hum_o, df_obj_h, object, smpl_verts, loss_dict, part_o=None):
        """[recon_fit_base.py:553-608] pair human / object contact points by predicted part label"""
        mask_o, mask_h = df_obj_h < 0.08, df_hum_o < 0.08
        part_o = torch.argmax(part_o, 1)
        pts_h, pts_o = [], []
        for hum, obj, mh, mo, po in zip(smpl_verts, object, mask_h, mask_o, part_o):
            ch, co = int(mh.sum()), int(mo.sum())
            if ch + co == 0:
                continue
            obj_v, label_o = (obj[mo], po[mo]) if co > 0 else (obj, po)
            hum_v, label_h = (hum[mh], self.part_labels[mh]) if ch > 0 else (hum, self.part_labels)
            for i in range(SMPL_PARTS_NUM):
                hi, oi = torch.where(label_h == i)[0], torch.where(label_o == i)[0]
                if hi.numel() == 0 or oi.numel() == 0:
                    continue
                pts_h.append(hum_v[hi])
                pts_o.append(obj_v[oi])
        if not pts_o:
            return
        loss_dict["contact"] = self._chamfer(pts_h, pts_o)

    def split_smpl(self, smpl):
        return SMPLPyTorchWrapperBatchSplitParams.from_smpl(smpl)

    @staticmethod
    def copy_smpl_params(split_smpl, smpl):
        smpl.pose.data[:, :3] = split_smpl.global_pose.data
        smpl.pose.data[:, 3:66] = split_smpl.body_pose.data
        smpl.pose.data[:, 66:] = split_smpl.hand_pose.data
        smpl.betas.data[:, :2] = split_smpl.top_betas.data   # (other_betas are not copied back: reference quirk, :682-690)
        smpl.trans.data = split_smpl.trans.data
        return smpl

    @staticmethod
    def get_smpl_height(smpl):
        with torch.no_grad():
            v = smpl()[0]
            return v[:, :, 1].max(1)[0] - v[:, :, 1].min(1)[0]

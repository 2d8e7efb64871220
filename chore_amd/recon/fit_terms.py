"""The small loss terms of the SMPL fit as two operators (csrc/fit_terms.hip).

`smpl_terms` = the pose / hand priors, the pose-initialisation, depth and 2-D keypoint terms of
/root/reference/recon/recon_fit_behave.py:293-337 (with recon_fit_base.py:528-547, 661-680); `point_terms` = the mean of a
clamped distance channel and the part cross-entropy over the query points (recon_fit_base.py:505-526,
recon_fit_behave.py:318-320).  Same values as the tensor expressions of ReconFitterBase (kept there: they are what the
tests compare these operators with, and `CHORE_FIT_TORCH_TERMS=1` runs); what changes is the number of launches: 4 forward
and 2 backward instead of ~45 and ~85.
"""
import ctypes
import os

import torch

from .. import _lib
from ..model.camera import _f32

TORCH_TERMS = bool(os.environ.get("CHORE_FIT_TORCH_TERMS"))     # A/B switch: the terms as tensor expressions


def _ptr(t):
    return None if t is None else t.data_ptr()


class _SmplTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, J, pose_init, kpts, cc, bmean, bprec, hmean, lprec, rprec, cam8):
        dev = pose.device
        h = _lib.handle(dev.index or 0)
        pose, J = pose.contiguous(), J.contiguous()
        B, P = pose.shape
        R = J.shape[1]
        outs = [torch.empty((), device=dev, dtype=torch.float32) for _ in range(5)]
        cam = (ctypes.c_float * 8)(*cam8)
        op = (ctypes.c_void_p * 5)(*[o.data_ptr() for o in outs])
        _lib.check(_lib.lib.chore_fit_smpl_terms_fwd(h, pose.data_ptr(), pose_init.data_ptr(), J.data_ptr(), _ptr(kpts),
                                                     cc.data_ptr(), bmean.data_ptr(), bprec.data_ptr(), hmean.data_ptr(),
                                                     lprec.data_ptr(), rprec.data_ptr(), B, P, R, cam, op,
                                                     torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_smpl_terms_fwd")
        ctx.save_for_backward(pose, J, pose_init, kpts, cc, bmean, bprec, hmean, lprec, rprec)
        ctx.cam8 = cam8
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *ups):
        pose, J, pose_init, kpts, cc, bmean, bprec, hmean, lprec, rprec = ctx.saved_tensors
        dev = pose.device
        h = _lib.handle(dev.index or 0)
        B, P = pose.shape
        R = J.shape[1]
        ups = [None if u is None else u.float().contiguous() for u in ups]
        up = (ctypes.c_void_p * 5)(*[_ptr(u) for u in ups])
        cam = (ctypes.c_float * 8)(*ctx.cam8)
        dpose = torch.empty(B, P, device=dev, dtype=torch.float32)
        dJ = torch.empty(B, R, 3, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.chore_fit_smpl_terms_bwd(h, pose.data_ptr(), pose_init.data_ptr(), J.data_ptr(), _ptr(kpts),
                                                     cc.data_ptr(), bmean.data_ptr(), bprec.data_ptr(), hmean.data_ptr(),
                                                     lprec.data_ptr(), rprec.data_ptr(), B, P, R, cam, up, dpose.data_ptr(),
                                                     dJ.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_fit_smpl_terms_bwd")
        return (dpose, dJ) + (None,) * 9


def smpl_terms_supported(pose, body_prior, hand_prior):
    from ..lib_smpl.priors import BodyPrior, HandPrior
    return (not TORCH_TERMS and pose.is_cuda and pose.dtype == torch.float32 and pose.dim() == 2 and pose.shape[1] == 156 and
            type(body_prior) is BodyPrior and type(hand_prior) is HandPrior and body_prior.prefix == 3 and
            body_prior.end == 66 and hand_prior.prefix == 66 and tuple(body_prior.prec.shape) == (63, 63))


def smpl_terms(pose, landmarks, pose_init, kpts, crop_center, body_prior, hand_prior, camera, net_in_size, z_0):
    """-> (pose prior, hand prior, pinit, smplz, j2d) device scalars; `landmarks` (B,R,3) with the body-25 rows first;
    kpts None: no keypoint term (j2d = 0, no gradient)"""
    f = lambda t: t.detach().float().contiguous()      # noqa: E731  (constants of the fit)
    cam8 = (_f32(camera.fx_px), _f32(camera.fy_px), _f32(camera.cx_px), _f32(camera.cy_px), _f32(camera.crop_size / 2),
            _f32(float(camera.crop_size)), _f32(float(net_in_size)), _f32(float(z_0)))
    return _SmplTermsFn.apply(pose, landmarks, f(pose_init), None if kpts is None else f(kpts), f(crop_center),
                              f(body_prior.mean).reshape(-1), f(body_prior.prec), f(hand_prior.mean).reshape(-1),
                              f(hand_prior.lhand_prec).reshape(45, 45), f(hand_prior.rhand_prec).reshape(45, 45), cam8)


class _PointTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, df, logits, labels, channel, cmax):
        dev = df.device
        h = _lib.handle(dev.index or 0)
        df = df.contiguous()
        B, _, N = df.shape
        C = 0
        if logits is not None:
            logits = logits.contiguous()
            C = logits.shape[1]
        o0 = torch.empty((), device=dev, dtype=torch.float32)
        o1 = torch.empty((), device=dev, dtype=torch.float32) if logits is not None else None
        ws = torch.empty(_lib.lib.chore_fit_point_terms_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.chore_fit_point_terms_fwd(h, df.data_ptr(), channel, float(cmax), _ptr(logits), _ptr(labels), B, N, C,
                                                      o0.data_ptr(), _ptr(o1), ws.data_ptr(),
                                                      torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_point_terms_fwd")
        ctx.save_for_backward(df, logits, labels)
        ctx.channel, ctx.cmax = channel, float(cmax)
        ctx.set_materialize_grads(False)
        if logits is None:
            return o0
        return o0, o1

    @staticmethod
    def backward(ctx, u0, u1=None):
        df, logits, labels = ctx.saved_tensors
        dev = df.device
        h = _lib.handle(dev.index or 0)
        B, _, N = df.shape
        C = 0 if logits is None else logits.shape[1]
        u0 = None if u0 is None else u0.float().contiguous()
        u1 = None if u1 is None else u1.float().contiguous()
        ddf = torch.empty_like(df)
        dl = None if logits is None else torch.empty_like(logits)
        _lib.check(_lib.lib.chore_fit_point_terms_bwd(h, df.data_ptr(), ctx.channel, ctx.cmax, _ptr(logits), _ptr(labels), B, N, C,
                                                      _ptr(u0), _ptr(u1), ddf.data_ptr(), _ptr(dl),
                                                      torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_point_terms_bwd")
        return ddf, dl, None, None, None


def point_terms_supported(df, logits=None):
    return (not TORCH_TERMS and df.is_cuda and df.dtype == torch.float32 and df.dim() == 3 and df.shape[1] == 2 and
            (logits is None or (logits.dtype == torch.float32 and logits.shape[1] <= 16)))


def point_terms(df, channel, cmax, logits=None, labels=None):
    """mean(clamp(df[:, channel], max=cmax)) and, with logits (B,C,N) / labels (B,N),
    cross_entropy(logits, labels, reduction='none').sum(-1).mean()"""
    if logits is None:
        return _PointTermsFn.apply(df, None, None, channel, cmax), None
    return _PointTermsFn.apply(df, logits, labels.long().contiguous(), channel, cmax)


class _ObjTransformFn(torch.autograd.Function):
    """(verts R + t) s with gradients for R, t, s (the template points are constants of the fit)"""

    @staticmethod
    def forward(ctx, verts, R, t, s):
        dev = verts.device
        h = _lib.handle(dev.index or 0)
        R, t, s = R.contiguous(), t.contiguous(), s.contiguous()
        B, N, _ = verts.shape
        out = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.chore_fit_obj_transform_fwd(h, verts.data_ptr(), R.data_ptr(), t.data_ptr(), s.data_ptr(), B, N,
                                                        out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_fit_obj_transform_fwd")
        ctx.save_for_backward(verts, R, t, s)
        return out

    @staticmethod
    def backward(ctx, g):
        verts, R, t, s = ctx.saved_tensors
        dev = verts.device
        h = _lib.handle(dev.index or 0)
        B, N, _ = verts.shape
        g = g.float().contiguous()
        dR, dt, ds = torch.empty_like(R), torch.empty_like(t), torch.empty_like(s)
        _lib.check(_lib.lib.chore_fit_obj_transform_bwd(h, verts.data_ptr(), R.data_ptr(), t.data_ptr(), s.data_ptr(), g.data_ptr(),
                                                        B, N, dR.data_ptr(), dt.data_ptr(), ds.data_ptr(),
                                                        torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_obj_transform_bwd")
        return None, dR, dt, ds


def obj_transform_supported(verts, R, t, s):
    ok = lambda x: x.is_cuda and x.dtype == torch.float32      # noqa: E731
    return (not TORCH_TERMS and all(ok(x) for x in (verts, R, t, s)) and not verts.requires_grad and verts.is_contiguous() and
            verts.dim() == 3 and tuple(R.shape) == (verts.shape[0], 3, 3) and tuple(t.shape) == (verts.shape[0], 3) and
            tuple(s.shape) == (verts.shape[0],))


def obj_transform(verts, R, t, s):
    return _ObjTransformFn.apply(verts, R, t, s)


class _ObjTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, object, centers, obj_s, smpl_center, scale0):
        dev = object.device
        h = _lib.handle(dev.index or 0)
        object, centers, obj_s = object.contiguous(), centers.contiguous(), obj_s.contiguous()
        B, N, _ = object.shape
        diff = torch.empty(B, 3, device=dev, dtype=torch.float32)
        o_scale = torch.empty((), device=dev, dtype=torch.float32)
        o_cent = torch.empty((), device=dev, dtype=torch.float32)
        ws = torch.empty(_lib.lib.chore_fit_obj_terms_workspace_bytes(B), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.chore_fit_obj_terms_fwd(h, object.data_ptr(), centers.data_ptr(), smpl_center.data_ptr(), obj_s.data_ptr(),
                                                    float(scale0), B, N, diff.data_ptr(), o_scale.data_ptr(), o_cent.data_ptr(),
                                                    ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_fit_obj_terms_fwd")
        ctx.save_for_backward(diff, obj_s)
        ctx.dims, ctx.scale0 = (B, N), float(scale0)
        ctx.set_materialize_grads(False)
        return o_scale, o_cent

    @staticmethod
    def backward(ctx, u_scale, u_cent):
        diff, obj_s = ctx.saved_tensors
        dev = diff.device
        h = _lib.handle(dev.index or 0)
        B, N = ctx.dims
        u_scale = None if u_scale is None else u_scale.float().contiguous()
        u_cent = None if u_cent is None else u_cent.float().contiguous()
        dobj = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
        dcen = torch.empty(B, 6, N, device=dev, dtype=torch.float32)
        dsc = torch.empty(B, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.chore_fit_obj_terms_bwd(h, diff.data_ptr(), obj_s.data_ptr(), ctx.scale0, _ptr(u_scale), _ptr(u_cent), B, N,
                                                    dobj.data_ptr(), dcen.data_ptr(), dsc.data_ptr(),
                                                    torch.cuda.current_stream(dev).cuda_stream), h, "chore_fit_obj_terms_bwd")
        return dobj, dcen, dsc, None, None


def obj_terms_supported(object, centers, obj_s, smpl_center):
    ok = lambda x: x.is_cuda and x.dtype == torch.float32      # noqa: E731
    return (not TORCH_TERMS and all(ok(x) for x in (object, centers, obj_s, smpl_center)) and object.dim() == 3 and
            centers.dim() == 3 and centers.shape[1] == 6 and centers.shape[2] == object.shape[1] and not smpl_center.requires_grad)


def obj_terms(object, centers, obj_s, smpl_center, scale0):
    """-> (scale, ocent): mean((obj_s - scale0)^2) and
    mse(mean(object, 1), smpl_center + mean(centers[:, 3:], -1)).sum(-1).mean()"""
    return _ObjTermsFn.apply(object, centers, obj_s, smpl_center.detach().float().contiguous(), scale0)


class _RotNoiseFn(torch.autograd.Function):
    """rot + scale * noise[k]; k += 1 (in place, on the device).  The gradient passes to rot unchanged."""

    @staticmethod
    def forward(ctx, rot, noise, k, scale):
        dev = rot.device
        h = _lib.handle(dev.index or 0)
        r = rot.float().contiguous()
        out = torch.empty_like(r)
        _lib.check(_lib.lib.chore_fit_rot_noise(h, r.data_ptr(), noise.data_ptr(), k.data_ptr(), float(scale), r.shape[0],
                                                noise.shape[0], out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_fit_rot_noise")
        return out      # (k is an integer counter outside autograd: nothing to mark)

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


def rot_noise_supported(rot, noise, k):
    return (not TORCH_TERMS and rot.is_cuda and rot.dtype == torch.float32 and noise.is_cuda and noise.dtype == torch.float32 and
            noise.is_contiguous() and noise.dim() == 4 and tuple(noise.shape[1:]) == tuple(rot.shape) and k.dtype == torch.int64 and
            k.numel() == 1 and k.is_cuda)


def rot_noise(rot, noise, k, scale=1e-4):
    """rot + scale * noise[k] for the (steps,B,3,3) table of draws, advancing the device counter k -- the argument of
    project_so3 in decopose_axis"""
    return _RotNoiseFn.apply(rot, noise, k, scale)

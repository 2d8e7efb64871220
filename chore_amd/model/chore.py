"""`CHORE` -- the field network with the reference's Python surface, computed by HIP kernels.

Drop-in for /root/reference/model/chore.py:10-257 on the hot path:
  filter(images)                  -> chore_encode_fwd           (HGFilter, model/HGFilters.py:144-185)
  query(points, crop_center)      -> chore_query_fwd            (model/chore.py:107-154)
  autograd to `points`            -> chore_query_bwd_points     (recon/generator.py:62-77)
Same constructor signature, same sub-module / parameter names (state_dict contract), same
attributes (`im_feat_list, tmpx, normx, intermediate_preds_list, preds, camera, OUT_DIST,
loss_weights, error_buffer`).  Everything runs on the GPU through libchore_hip.so; CPU tensors are
rejected (no fallback).

Precision: `compute_dtype` "fp32" (exact-fp32 matrix-core path, parity mode), "bf16" (bf16 feature maps / MFMA
operands with fp32 accumulation; heads stay fp32) or "fp16x3" (fp32 feature maps; the encoder's convolutions run on the
fp16 matrix cores with every operand split into an fp16 hi + lo pair, three MFMAs per product: fp32-grade results at
5x the native fp32 MFMA rate).  Selected by
`opt.compute_dtype`, else the CHORE_AMD_DTYPE environment variable, else "fp32".
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .camera import KinectColorCamera
from .hgfilter import HGFilter

_DT = {"fp32": _lib.F32, "bf16": _lib.BF16, "fp16x3": _lib.F16X3, "fp16": _lib.F16}
# what the query kernels see: the fp16 x 3 encoder keeps fp32 feature maps; the inference query (forward and backward
# to the points) of the fp16x3 AND the bf16 mode runs the heads on the fp16 matrix cores with split operands
# (csrc/heads_x3.h: fp32-grade results); the fp32 mode keeps the native fp32 MFMA
# "fp16" ("fp16 fields", BASELINE configs[4]): IEEE half feature maps, two MFMAs per product in the encoder, the same heads
_QDT = {"fp32": _lib.F32, "bf16": _lib.BF16, "fp16x3": _lib.F32, "fp16": _lib.F32}
_QDT_FWD = {"fp32": _lib.F32, "bf16": _lib.BF16 | _lib.HEADS_X3, "fp16x3": _lib.F16X3, "fp16": _lib.F16}


def _nhwc_ptr(t, C):
    """(B,C,H,W) channels-last view (or plain NHWC tensor) -> (data_ptr, H, W); validates layout"""
    if t.dim() != 4 or t.shape[1] != C:
        raise ValueError(f"expected a (B,{C},H,W) feature map, got {tuple(t.shape)}")
    B, _, H, W = t.shape
    if t.stride() != (H * W * C, 1, W * C, C):
        raise ValueError("feature maps must be NHWC in memory (as produced by CHORE.filter)")
    return t.data_ptr(), H, W


class _QueryFn(torch.autograd.Function):
    """chore_query_fwd / chore_query_bwd_points as one autograd node (gradient w.r.t. points)."""

    @staticmethod
    def forward(ctx, points, crop_center, feat, tmpx, arena, cam6, dtype, fwd_dtype):
        B, N, _ = points.shape
        dev = points.device
        h = _lib.handle(dev.index or 0)
        fp, FH, FW = _nhwc_ptr(feat, 256)
        tp, TH, TW = _nhwc_ptr(tmpx, 64)
        df = torch.empty(B, 2, N, device=dev, dtype=torch.float32)
        pca = torch.empty(B, 9, N, device=dev, dtype=torch.float32)
        parts = torch.empty(B, 14, N, device=dev, dtype=torch.float32)
        centers = torch.empty(B, 6, N, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        # CHORE_QUERY_SORTED=1: a workspace for the sorted-order gather of large queries (chore_query_fwd_ws; same results bit for
        # bit, a third of the gather's bytes, but 25 - 45 us of sorting for 15 - 20 us of gather on one MI355X: off by default)
        ws = (torch.empty(_lib.lib.chore_query_fwd_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
              if N >= 8192 and os.environ.get("CHORE_QUERY_SORTED") else None)
        _lib.check(_lib.lib.chore_query_fwd_ws(h, points.data_ptr(), crop_center.data_ptr(), B, N, fp, FH, FW,
                                               tp, TH, TW, fwd_dtype, arena.data_ptr(), cam6, df.data_ptr(),
                                               pca.data_ptr(), parts.data_ptr(), centers.data_ptr(), None,
                                               None if ws is None else ws.data_ptr(), stream), h, "chore_query_fwd_ws")
        ctx.save_for_backward(points, crop_center, feat, tmpx, arena)
        ctx.cam6, ctx.dtype = cam6, fwd_dtype
        ctx.set_materialize_grads(False)      # heads without an upstream gradient arrive as None = NULL for the kernel
        return df, pca, parts, centers

    @staticmethod
    def backward(ctx, g_df, g_pca, g_parts, g_centers):
        points, crop_center, feat, tmpx, arena = ctx.saved_tensors
        B, N, _ = points.shape
        dev = points.device
        h = _lib.handle(dev.index or 0)
        fp, FH, FW = _nhwc_ptr(feat, 256)
        tp, TH, TW = _nhwc_ptr(tmpx, 64)

        def prep(g):
            return None if g is None else g.contiguous().float()

        gs = [prep(g) for g in (g_df, g_pca, g_parts, g_centers)]
        dpoints = torch.empty_like(points)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = [None if g is None else g.data_ptr() for g in gs]
        _lib.check(_lib.lib.chore_query_bwd_points(h, points.data_ptr(), crop_center.data_ptr(), B, N, fp, FH,
                                                   FW, tp, TH, TW, ctx.dtype, arena.data_ptr(), ctx.cam6,
                                                   ptr[0], ptr[1], ptr[2], ptr[3], dpoints.data_ptr(),
                                                   stream), h, "chore_query_bwd_points")
        return dpoints, None, None, None, None, None, None, None


class _QueryTrainFn(torch.autograd.Function):
    """the query as a TRAINABLE node: gradients w.r.t. the head parameters and the two feature maps (and the points).

    forward = chore_query_fwd_train (stages the 323-vectors and ReLU outputs); backward = chore_query_bwd_train (no
    recompute: masks read back; stages the pre-activation gradients), chore_heads_wgrad for the 32 parameter gradients,
    chore_scatter_features for the maps.  `params` = 32 tensors: for df, part_predictor, pca_predictor, center_predictor the (weight, bias) of
    Sequential indices 0, 2, 4, 6 (the reference's make_decoder, model/chore.py:74-85)."""

    KORDER = (0, 1, 2, 3)          # module order above -> kernel head order df, parts, pca, centers
    ODIM = (2, 14, 9, 6)

    @staticmethod
    def forward(ctx, points, crop_center, feat, tmpx, arena, cam6, dtype, share, *params):
        B, N, _ = points.shape
        dev = points.device
        h = _lib.handle(dev.index or 0)
        fp, FH, FW = _nhwc_ptr(feat, 256)
        tp, TH, TW = _nhwc_ptr(tmpx, 64)
        df = torch.empty(B, 2, N, device=dev, dtype=torch.float32)
        pca = torch.empty(B, 9, N, device=dev, dtype=torch.float32)
        parts = torch.empty(B, 14, N, device=dev, dtype=torch.float32)
        centers = torch.empty(B, 6, N, device=dev, dtype=torch.float32)
        in_img = torch.empty(B, N, device=dev, dtype=torch.uint8)
        stream = torch.cuda.current_stream(dev).cuda_stream
        # the forward stages the 323-vectors and the ReLU outputs for the backward (13.6 KB per point): nothing is recomputed
        staging = torch.empty(_lib.lib.chore_query_train_bytes(B, N), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.chore_query_fwd_train(h, points.data_ptr(), crop_center.data_ptr(), B, N, fp, FH, FW,
                                                  tp, TH, TW, dtype, arena.data_ptr(), cam6, df.data_ptr(),
                                                  pca.data_ptr(), parts.data_ptr(), centers.data_ptr(), in_img.data_ptr(),
                                                  staging.data_ptr(), stream), h, "chore_query_fwd_train")
        ctx.save_for_backward(points, crop_center, feat, tmpx, arena, in_img, staging)
        ctx.cam6, ctx.dtype, ctx.share = cam6, dtype, share
        return df, pca, parts, centers

    @staticmethod
    def backward(ctx, g_df, g_pca, g_parts, g_centers):
        points, crop_center, feat, tmpx, arena, in_img, staging = ctx.saved_tensors
        B, N, _ = points.shape
        P = B * N
        dev = points.device
        h = _lib.handle(dev.index or 0)
        fp, FH, FW = _nhwc_ptr(feat, 256)
        tp, TH, TW = _nhwc_ptr(tmpx, 64)
        zero = lambda c: torch.zeros(B, c, N, device=dev)  # noqa: E731
        g_df = zero(2) if g_df is None else g_df.contiguous().float()
        g_pca = zero(9) if g_pca is None else g_pca.contiguous().float()
        g_parts = zero(14) if g_parts is None else g_parts.contiguous().float()
        g_centers = zero(6) if g_centers is None else g_centers.contiguous().float()
        need_pts = ctx.needs_input_grad[0]
        dpoints = torch.empty_like(points) if need_pts else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_query_bwd_train(h, points.data_ptr(), crop_center.data_ptr(), B, N, fp, FH, FW, tp, TH,
                                                  TW, ctx.dtype, arena.data_ptr(), ctx.cam6, g_df.data_ptr(),
                                                  g_pca.data_ptr(), g_parts.data_ptr(), g_centers.data_ptr(),
                                                  staging.data_ptr(), None if dpoints is None else dpoints.data_ptr(),
                                                  1, stream), h, "chore_query_bwd_train")
        HD = 128
        # the df head sees no gradient where the point is outside the image (df is overwritten there, chore.py:147-150)
        g_df = g_df * in_img.unsqueeze(1).float()
        g_out = (g_df, g_parts, g_pca, g_centers)                     # kernel head order
        grads = [None] * 32

        # all 32 parameter gradients from the staged rows in three launches (heads_wgrad.hip)
        g_c = [t.float().contiguous() for t in g_out]
        # The stacks of one forward share ONE gradient arena: the first backward writes it, the others add to it inside
        # the kernel, and only the last one hands the 32 views to autograd (the rest return None = zero) -- instead of
        # 32 AccumulateGrad additions per further stack.
        share = ctx.share
        first = last = True
        if share is not None:
            first = share["arena"] is None
            if first:
                share["arena"] = torch.empty(_lib.lib.chore_heads_wgrad_floats(), device=dev)
            share["done"] += 1
            last = share["done"] == share["n"]
            garena = share["arena"]
            if last:
                share["arena"], share["done"] = None, 0
        else:
            garena = torch.empty(_lib.lib.chore_heads_wgrad_floats(), device=dev)
        ws = torch.empty(_lib.lib.chore_heads_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.chore_heads_wgrad(h, staging.data_ptr(), B, N, g_c[0].data_ptr(), g_c[2].data_ptr(),
                                              g_c[1].data_ptr(), g_c[3].data_ptr(), garena.data_ptr(), ws.data_ptr(),
                                              (1 if (ctx.dtype & _lib.HEADS_X3) else 0) | (0 if first else 2), stream),
                   h, "chore_heads_wgrad")
        o = 0
        for k in range(4 if last else 0):                             # module order = kernel head order
            od = g_out[k].shape[1]
            for j, shape in enumerate(((HD, 323, 1), (HD,), (HD, HD, 1), (HD,), (HD, HD, 1), (HD,), (od, HD, 1), (od,))):
                n = 1
                for d in shape:
                    n *= d
                grads[k * 8 + j] = garena[o:o + n].view(shape)
                o += n
        dfeat = dtmpx = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dfe = torch.empty(B, FH, FW, 256, device=dev) if ctx.needs_input_grad[2] else None
            dtm = torch.empty(B, TH, TW, 64, device=dev) if ctx.needs_input_grad[3] else None
            _lib.check(_lib.lib.chore_scatter_features(h, points.data_ptr(), crop_center.data_ptr(), B, N, FH, FW, TH, TW,
                                                       ctx.cam6, staging.data_ptr(),
                                                       None if dfe is None else dfe.data_ptr(),
                                                       None if dtm is None else dtm.data_ptr(), 0, stream), h,
                       "chore_scatter_features")
            dfeat = None if dfe is None else dfe.permute(0, 3, 1, 2).to(feat.dtype)
            dtmpx = None if dtm is None else dtm.permute(0, 3, 1, 2).to(tmpx.dtype)
        return (dpoints, None, dfeat, dtmpx, None, None, None, None) + tuple(grads)


def _mlp(input_sz, output_sz, hidden_sz):
    # parameter container with the reference's Sequential indices 0,2,4,6 (model/chore.py:74-85)
    return nn.Sequential(nn.Conv1d(input_sz, hidden_sz, 1), nn.ReLU(),
                         nn.Conv1d(hidden_sz, hidden_sz, 1), nn.ReLU(),
                         nn.Conv1d(hidden_sz, hidden_sz, 1), nn.ReLU(),
                         nn.Conv1d(hidden_sz, output_sz, 1))


class _StackLossFn(torch.autograd.Function):
    """the six loss terms of one stack and their gradients in one pass (chore_train_loss, csrc/train_loss.hip).
    Returns 7 floats: the terms in the reference's order (h, o, parts, pca, smpl, obj), each already divided by the number
    of stacks, and their sum [6] -- the stack's share of the averaged error, the only differentiable entry."""

    @staticmethod
    def forward(ctx, df, pca, parts, centers, tgt):
        dev = df.device
        h = _lib.handle(dev.index or 0)
        B, _, N = df.shape
        preds = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous() for t in (df, pca, parts, centers)]
        grads = [torch.empty_like(t) for t in preds]
        losses = torch.empty(7, device=dev)
        ws = torch.empty(_lib.lib.chore_train_loss_workspace_bytes(), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_train_loss(h, *[t.data_ptr() for t in preds], tgt["df_h"].data_ptr(), tgt["df_o"].data_ptr(),
                                             tgt["parts_gt"].data_ptr(), tgt["pca_gt"].data_ptr(), tgt["body_center"].data_ptr(),
                                             tgt["obj_center"].data_ptr(), B, N, float(tgt["max_dist"]), tgt["weights"],
                                             float(tgt["scale"]), *[g.data_ptr() for g in grads], losses.data_ptr(), 0,
                                             ws.data_ptr(), stream), h, "chore_train_loss")
        ctx.save_for_backward(*grads)
        ctx.shapes = [t.shape for t in (df, pca, parts, centers)]
        return losses

    @staticmethod
    def backward(ctx, g):
        grads, gs = list(ctx.saved_tensors), g[6]
        try:
            out = torch._foreach_mul(grads, gs)
        except (RuntimeError, TypeError):
            out = [t * gs for t in grads]
        return tuple(o.view(sh) for o, sh in zip(out, ctx.shapes)) + (None,)


class CHORE(nn.Module):
    def __init__(self, opt, projection_mode="perspective", error_term=nn.MSELoss(), rank=-1, num_parts=14,
                 hidden_dim=128):
        super().__init__()
        self.name = "chore"
        self.opt = opt
        self.error_term = error_term
        self.device = torch.device("cuda", opt.gpu_id) if isinstance(opt.gpu_id, int) else torch.device(opt.gpu_id)
        if opt.z_feat != "xyz" or opt.projection_mode != "perspective" or not opt.skip_hourglass:
            raise ValueError("chore_amd implements z_feat='xyz', projection_mode='perspective', skip_hourglass=True")
        if num_parts != 14 or hidden_dim != 128:
            raise ValueError("the HIP heads are specialised for num_parts=14, hidden_dim=128")
        self.z_feat = opt.z_feat
        self.image_filter = HGFilter(opt)
        feature_size = 256 + 3 + 256 // 4
        self.df = _mlp(feature_size, 2, hidden_dim)
        self.part_predictor = _mlp(feature_size, num_parts, hidden_dim)
        self.pca_predictor = _mlp(feature_size, 9, hidden_dim)
        self.center_predictor = _mlp(feature_size, 6, hidden_dim)
        self.rank = rank
        self.dfloss_func = nn.L1Loss(reduction="none")
        self.part_loss_func = nn.CrossEntropyLoss(reduction="none")
        self.loss_weights = [1.0, 1.0, 0.006, 500, 1000, 1000]
        self.camera = KinectColorCamera(opt.loadSize)
        self.OUT_DIST = 5.0
        self.losses_on_host = True       # False: forward() leaves the 6 separate losses on the device (no sync)
        self._init_weights()

        self.im_feat_list = []
        self.tmpx = None
        self.normx = None
        self.intermediate_preds_list = []
        self.preds = None
        self.labels = None
        self.points = None
        self.crop_center = None
        self.error_buffer = None
        dt = getattr(opt, "compute_dtype", None) or os.environ.get("CHORE_AMD_DTYPE", "fp32")
        if dt not in _DT:
            raise ValueError(f"compute_dtype must be one of {list(_DT)}")
        self.compute_dtype = dt
        self._heads_packed = None  # (version key, arena)
        self._cam6 = (ctypes.c_float * 6)(*self.camera.kernel_constants())

    def _init_weights(self, gain=0.02):
        # same distribution as net_util.init_weights('normal', 0.02): N(0, gain) conv weights,
        # zero conv biases, GroupNorm left at (1, 0)   (/root/reference/model/net_util.py:218-251)
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.normal_(m.weight.data, 0.0, gain)
                if m.bias is not None:
                    nn.init.constant_(m.bias.data, 0.0)

    def _fwd_dtype(self, dtype):
        """the dtype word of the inference query entry points: map type | heads mode.  CHORE_HEADS_FP32=1 (A/B switch): the
        native fp32 MFMA for the heads -- not with fp16 maps, which only the split-operand kernels read"""
        if os.environ.get("CHORE_HEADS_FP32") and self.compute_dtype != "fp16":
            return dtype
        return _QDT_FWD[self.compute_dtype]

    # ---- packed weights ---------------------------------------------------------------------
    def _head_modules(self):
        return (("df", self.df), ("part_predictor", self.part_predictor), ("pca_predictor", self.pca_predictor),
                ("center_predictor", self.center_predictor))

    def _head_params(self):
        """the parameters of the four heads, listed once (walking four nn.Sequential per query call cost ~90 us of host time, and a
        fitting step or a surface step is a handful of query calls); Parameter objects are replaced by .to() / load_state_dict
        (which drop the list) -- not by optimiser steps, which change data in place"""
        ps = getattr(self, "_head_param_list", None)
        if ps is None:
            ps = self._head_param_list = [p for _, m in self._head_modules() for p in m.parameters()]
        return ps

    def _heads_arena(self, device):
        params = self._head_params()
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
        # heads with requires_grad=True never hit the cache (any entry point, any grad mode): see query()
        if self._heads_packed is not None and self._heads_packed[0] == key and not any(p.requires_grad for p in params):
            return self._heads_packed[1]
        dtype = _QDT[self.compute_dtype]
        h = _lib.handle(device.index or 0)
        arena = torch.empty(_lib.lib.chore_heads_arena_bytes(dtype), dtype=torch.uint8, device=device)
        named = []
        for name, m in self._head_modules():
            for k, p in m.state_dict(keep_vars=True).items():
                t = p.detach()
                if not t.is_cuda:
                    raise RuntimeError("CHORE parameters must live on the GPU: call .to(device) first")
                named.append((f"{name}.{k}", t.float().contiguous()))
        descs, keep = _lib.make_descs(named)
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(_lib.lib.chore_heads_pack(h, descs, len(named), dtype, arena.data_ptr(), stream), h,
                   "chore_heads_pack")
        self._heads_packed = (key, arena)
        return arena

    def invalidate_packed(self):
        """drop the packed copies of the head and encoder weights.  The caches are keyed on (address, version) of every
        parameter, which sees in-place tensor operations, load_state_dict and .to(); a write through `p.data`, the fused
        optimiser kernels (torch.optim.Adam(fused=True) does not advance the version counters) and a replayed hipGraph are
        invisible to it -- call this after one.  train() / eval() do it when the mode changes, and a query whose heads have
        requires_grad=True never uses the cache, whatever the grad mode (frozen heads -- inference, fitting -- do)."""
        self._head_param_list = None
        self._heads_packed = None
        self.image_filter.invalidate_packed()

    def train(self, mode=True):
        if bool(mode) != self.training:
            self.invalidate_packed()          # e.g. validation after training steps whose optimiser the version counters missed
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._heads_packed = None
        self._head_param_list = None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate_packed()
        return out

    # ---- reference API ----------------------------------------------------------------------
    def filter(self, images):
        """encode images (B,5,H,W); keeps all stacks in training, the last one in eval
        (/root/reference/model/chore.py:87-96)"""
        n_out = self.image_filter.num_modules if self.training else 1
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.image_filter.parameters()):
            # training: the layer-by-layer differentiable forward (model/hgfilter_train.py)
            from .hgfilter_train import forward_train
            if self.compute_dtype == "fp16":
                raise NotImplementedError("compute_dtype 'fp16' is an inference mode (train in 'fp32', 'fp16x3' or 'bf16')")
            tdt = torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32
            feats, self.tmpx, self.normx = forward_train(self.image_filter, images, tdt, x3=self.compute_dtype == "fp16x3")
            feats = feats[-n_out:]
        else:
            feats, self.tmpx, self.normx = self.image_filter(images, _DT[self.compute_dtype], n_out)
        self.im_feat_list = feats

    def project_points(self, points, offsets):
        """(B,N,3),(B,2) -> (B,3,N) normalised image coordinates + depth (/root/reference/model/chore.py:98-105;
        host-side helper -- the query kernel projects itself)"""
        return self.camera.project_points(points, offsets)

    def query(self, points, crop_center=None, **kwargs):
        if crop_center is None:
            raise ValueError("crop_center is required (the reference asserts it in camera.normalize)")
        if not points.is_cuda:
            raise RuntimeError("chore_amd.CHORE.query needs device tensors (no CPU path)")
        if not self.im_feat_list:
            raise RuntimeError("call filter(images) before query()")
        self.points = points
        self.crop_center = crop_center
        pts = points if (points.dtype == torch.float32 and points.is_contiguous()) else points.float().contiguous()
        cc = crop_center.to(device=points.device, dtype=torch.float32).contiguous()
        if pts.dim() != 3 or pts.shape[2] != 3 or cc.shape != (pts.shape[0], 2):
            raise ValueError("points must be (B,N,3) and crop_center (B,2)")
        dtype = _QDT[self.compute_dtype]
        head_params = self._head_params()
        heads_trainable = any(p.requires_grad for p in head_params)
        train = torch.is_grad_enabled() and (heads_trainable or self.tmpx.requires_grad or
                                             any(f.requires_grad for f in self.im_feat_list))
        if heads_trainable:
            # heads that are being trained are packed afresh for every query -- under torch.no_grad() too (a validation pass
            # between training steps that never called eval()): the cache key (address, version counter) does not see every
            # optimiser -- torch's fused Adam / AdamW / SGD kernels update the parameters without advancing `_version`
            self._heads_packed = None
        arena = self._heads_arena(points.device)
        if train and self.compute_dtype == "fp16":
            # the training kernels read fp32 / bf16 maps; half maps would be read as fp32 (wrong values, reads past the buffer)
            raise NotImplementedError("compute_dtype 'fp16' is an inference mode: query() with trainable heads or maps that "
                                      "require grad needs 'fp32' or 'bf16' (freeze the parameters for inference)")
        self.intermediate_preds_list = []
        if pts.shape[1] == 0:     # no points: empty predictions (what the reference's torch ops return), nothing to launch
            B = pts.shape[0]
            z = lambda *sh: pts.new_zeros(sh) + pts.sum() * 0     # noqa: E731  (keeps the graph connected)
            self.intermediate_preds_list = [(z(B, 2, 0), z(B, 3, 3, 0), z(B, 14, 0), z(B, 6, 0)) for _ in self.im_feat_list]
            self.preds = self.intermediate_preds_list[-1]
            return
        # (training) one gradient arena for the head parameters over all stacks, see _QueryTrainFn.backward.  Only inside
        # CHORE.forward, whose loss reaches every stack (get_errors sums over all of them): a backward that does not visit
        # all the nodes must not share.  CHORE_HEADS_NO_SHARE=1 switches it off.
        share = None
        if train and len(self.im_feat_list) > 1 and getattr(self, "_share_head_grads", False) and \
                not os.environ.get("CHORE_HEADS_NO_SHARE"):
            share = {"n": len(self.im_feat_list), "done": 0, "arena": None}
        for feat in self.im_feat_list:
            if train:
                # the bf16 training mode runs the heads' GEMMs on the fp16 matrix cores with split operands (fp32-grade,
                # csrc/heads_x3.h); the fp32 mode keeps the native fp32 MFMA everywhere
                x3 = getattr(self, "heads_x3", None)          # None: by mode; True / False: forced (tests, A/B runs)
                if x3 is None:
                    x3 = self.compute_dtype != "fp32" and not os.environ.get("CHORE_HEADS_FP32")
                tdt = dtype | (_lib.HEADS_X3 if x3 else 0)
                df, pca, parts, centers = _QueryTrainFn.apply(pts, cc, feat, self.tmpx, arena, self._cam6, tdt, share,
                                                              *head_params)
            else:
                # CHORE_HEADS_FP32=1: the native fp32 MFMA for the heads in every mode (A/B switch)
                df, pca, parts, centers = _QueryFn.apply(pts, cc, feat, self.tmpx, arena, self._cam6, dtype,
                                                         self._fwd_dtype(dtype))
            B, _, N = df.shape
            self.intermediate_preds_list.append((df, pca.view(B, 3, 3, N), parts, centers))
        self.preds = self.intermediate_preds_list[-1]

    def query_df(self, points, crop_center):
        """the two distance fields (B,2,N) of the LAST stack at `points` and nothing else: chore_query_fwd with the other three
        outputs NULL -- their heads' waves leave after the gather, a quarter of the weight traffic and MFMA work.  No autograd
        graph (see query_grad_points); does not touch get_preds()."""
        if not self.im_feat_list:
            raise RuntimeError("call filter(images) before query_df()")
        pts = points.detach()
        if not (pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous() and pts.dim() == 3 and pts.shape[2] == 3):
            raise ValueError("points must be a contiguous fp32 (B,N,3) device tensor")
        B, N, _ = pts.shape
        dev = pts.device
        cc = crop_center.to(device=dev, dtype=torch.float32).contiguous()
        h = _lib.handle(dev.index or 0)
        fp, FH, FW = _nhwc_ptr(self.im_feat_list[-1], 256)
        tp, TH, TW = _nhwc_ptr(self.tmpx, 64)
        dtype = _QDT[self.compute_dtype]
        fwd_dtype = self._fwd_dtype(dtype)
        df = torch.empty(B, 2, N, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.chore_query_fwd(h, pts.data_ptr(), cc.data_ptr(), B, N, fp, FH, FW, tp, TH, TW, fwd_dtype,
                                            self._heads_arena(dev).data_ptr(), self._cam6, df.data_ptr(), None, None, None, None,
                                            torch.cuda.current_stream(dev).cuda_stream), h, "chore_query_fwd")
        return df

    def surface_step(self, points, crop_center, k, threshold):
        """one projection step of Generator.approx_surface (recon/generator.py:50-79) on the LAST stack's field:
        points - normalize(d sum(clamp(df_k, max=threshold)) / d points) * clamp(df_k, max=threshold), one launch
        (chore_gen_surface_step_fused).  None if the mode has no fused step (the fp32-MFMA heads): the caller composes it
        from query_df / query_grad_points."""
        dtype = _QDT[self.compute_dtype]
        fwd_dtype = self._fwd_dtype(dtype)
        # (fp16 maps are always read by the split-operand heads, capi.hip query_x3: until round 6 this test left the fp16-fields mode
        # -- BASELINE configs[4] -- on the four launches per step)
        if not (fwd_dtype in (_lib.F16X3, _lib.F16) or (fwd_dtype & _lib.HEADS_X3)) or os.environ.get("CHORE_GEN_FOUR_LAUNCHES"):
            return None
        if not self.im_feat_list:
            raise RuntimeError("call filter(images) before surface_step()")
        pts = points.detach()
        if not (pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous() and pts.dim() == 3 and pts.shape[2] == 3):
            raise ValueError("points must be a contiguous fp32 (B,N,3) device tensor")
        B, N, _ = pts.shape
        dev = pts.device
        cc = crop_center.to(device=dev, dtype=torch.float32).contiguous()
        h = _lib.handle(dev.index or 0)
        fp, FH, FW = _nhwc_ptr(self.im_feat_list[-1], 256)
        tp, TH, TW = _nhwc_ptr(self.tmpx, 64)
        out = torch.empty_like(pts)
        _lib.check(_lib.lib.chore_gen_surface_step_fused(h, pts.data_ptr(), cc.data_ptr(), B, N, fp, FH, FW, tp, TH, TW, fwd_dtype,
                                                         self._heads_arena(dev).data_ptr(), self._cam6, int(k), float(threshold),
                                                         out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_gen_surface_step_fused")
        return out

    def query_grad_points(self, points, crop_center, g_df=None, g_pca=None, g_parts=None, g_centers=None):
        """d(sum_k <g_k, output_k>) / d points of the LAST stack's field at `points`, straight from chore_query_bwd_points: no
        autograd graph, for loops that need nothing but this gradient (Generator.approx_surface).  g_* like the outputs of
        query(): (B,2,N), (B,9,N) or (B,3,3,N), (B,14,N), (B,6,N); None = zero."""
        if not self.im_feat_list:
            raise RuntimeError("call filter(images) before query_grad_points()")
        pts = points.detach()
        if not (pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous() and pts.dim() == 3 and pts.shape[2] == 3):
            raise ValueError("points must be a contiguous fp32 (B,N,3) device tensor")
        B, N, _ = pts.shape
        cc = crop_center.to(device=pts.device, dtype=torch.float32).contiguous()
        dev = pts.device
        h = _lib.handle(dev.index or 0)
        feat = self.im_feat_list[-1]
        fp, FH, FW = _nhwc_ptr(feat, 256)
        tp, TH, TW = _nhwc_ptr(self.tmpx, 64)
        gs = [None if g is None else g.detach().reshape(B, -1, N).float().contiguous() for g in (g_df, g_pca, g_parts, g_centers)]
        ptr = [None if g is None else g.data_ptr() for g in gs]
        dtype = _QDT[self.compute_dtype]
        fwd_dtype = self._fwd_dtype(dtype)
        dpoints = torch.empty_like(pts)
        _lib.check(_lib.lib.chore_query_bwd_points(h, pts.data_ptr(), cc.data_ptr(), B, N, fp, FH, FW, tp, TH, TW, fwd_dtype,
                                                   self._heads_arena(dev).data_ptr(), self._cam6, ptr[0], ptr[1], ptr[2], ptr[3],
                                                   dpoints.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_query_bwd_points")
        return dpoints

    def get_preds(self):
        return self.preds

    def get_im_feat(self):
        return self.im_feat_list[-1]

    def get_error(self):
        return self.error_term(self.preds, self.labels)

    def forward(self, images, points, df_h, df_o, parts_gt, pca_gt, body_center=None, max_dist=5.0,
                obj_center=None, crop_center=None, **kwargs):
        if self._interleaved_ok(images, points, df_h, crop_center):
            return self._forward_interleaved(images, points, df_h, df_o, parts_gt, pca_gt, body_center, max_dist, obj_center,
                                             crop_center)
        self.filter(images)
        self._share_head_grads = True          # the loss below reaches every stack (see query)
        try:
            self.query(points=points, crop_center=crop_center, **kwargs)
        finally:
            self._share_head_grads = False
        return self.get_errors(df_h, df_o, parts_gt, pca_gt, max_dist, body_center, obj_center, **kwargs)

    # ---- training forward with the field queries beside the encoder ---------------------------------------------------------
    def _interleaved_ok(self, images, points, df_h, crop_center):
        return (self.training and torch.is_grad_enabled() and images.is_cuda and points.is_cuda and df_h.is_cuda
                and crop_center is not None and self.compute_dtype in ("fp32", "bf16", "fp16x3") and points.dim() == 3
                and points.shape[1] > 0 and not os.environ.get("CHORE_TRAIN_NO_INTERLEAVE")
                and not os.environ.get("CHORE_TORCH_LOSS")
                and any(p.requires_grad for p in self.image_filter.parameters()))

    def _forward_interleaved(self, images, points, df_h, df_o, parts_gt, pca_gt, body_center, max_dist, obj_center, crop_center):
        """CHORE.forward of a training step (model/chore.py:175-190 of the reference: filter -> query -> get_errors) with the
        per-stack work reordered: stack i's field query and loss are launched on a second stream the moment the encoder has
        produced that stack's feature map, so they run beside stacks i+1 .. of the encoder -- and, because autograd replays
        nodes on the stream of their forward, the query backward of stacks 1 .. i runs beside the encoder backward of the later
        stacks.  The query kernels are bound by their staging traffic (2.4 GB per stack), the encoder chain by launch latency:
        they share the chip well.  Same nodes, same arithmetic, same results as the sequential form (CHORE_TRAIN_NO_INTERLEAVE=1)."""
        from .hgfilter_train import forward_train
        dev = images.device
        main = torch.cuda.current_stream(dev)
        side = self._query_stream(dev)
        self.points, self.crop_center = points, crop_center
        pts = points if (points.dtype == torch.float32 and points.is_contiguous()) else points.float().contiguous()
        cc = crop_center.to(device=dev, dtype=torch.float32).contiguous()
        if pts.shape[2] != 3 or cc.shape != (pts.shape[0], 2):
            raise ValueError("points must be (B,N,3) and crop_center (B,2)")
        head_params = self._head_params()
        if any(p.requires_grad for p in head_params):
            self._heads_packed = None           # see query(): fused optimisers do not advance the version counters
        arena = self._heads_arena(dev)
        dtype = _QDT[self.compute_dtype]
        x3 = getattr(self, "heads_x3", None)
        if x3 is None:
            x3 = self.compute_dtype != "fp32" and not os.environ.get("CHORE_HEADS_FP32")
        qdt = dtype | (_lib.HEADS_X3 if x3 else 0)
        n = self.image_filter.num_modules
        f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()   # noqa: E731
        tgt = dict(df_h=f32(df_h), df_o=f32(df_o), parts_gt=parts_gt.long().contiguous(), pca_gt=f32(pca_gt),
                   body_center=f32(body_center), obj_center=f32(obj_center), max_dist=max_dist, scale=1.0 / n,
                   weights=(ctypes.c_float * 6)(*[float(v) for v in self.loss_weights]))
        share = None
        if n > 1 and not os.environ.get("CHORE_HEADS_NO_SHARE"):
            share = {"n": n, "done": 0, "arena": None}
        preds, outs = [], []

        def on_stack(i, feat, tmpx):
            side.wait_stream(main)              # stack i's feature map (and, the first time, the targets / the packed heads)
            with torch.cuda.stream(side):
                df, pca, parts, centers = _QueryTrainFn.apply(pts, cc, feat, tmpx, arena, self._cam6, qdt, share, *head_params)
                outs.append(_StackLossFn.apply(df, pca, parts, centers, tgt))
            B, _, N = df.shape
            preds.append((df, pca.view(B, 3, 3, N), parts, centers))

        tdt = torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32
        keep = getattr(self, "keep_train_segments", False)
        cuts = [] if keep else None
        feats, self.tmpx, self.normx = forward_train(self.image_filter, images, tdt, on_stack=on_stack,
                                                     x3=self.compute_dtype == "fp16x3", cuts=cuts)
        main.wait_stream(side)
        if keep:
            # what chore_amd.parallel.backward_in_segments needs to run this step's backward one hourglass stack at a time (so
            # that each stack's gradients can go to the all-reduce while the next stack's backward runs): the tensor entering
            # every stack and every stack's share of the error
            if share is None and n > 1:
                raise RuntimeError("keep_train_segments needs the stacks' shared head-gradient arena (CHORE_HEADS_NO_SHARE is set)")
            self.train_segments = dict(cuts=cuts, stack_losses=[o[6] for o in outs])
        self.im_feat_list = feats
        self.intermediate_preds_list = preds
        self.preds = preds[-1]
        tot = outs[0] if n == 1 else torch.stack(outs).sum(0)
        error, losses_all = tot[6], tot[:6].detach()
        if self.losses_on_host:
            losses_all = losses_all.cpu()
        self.error_buffer = losses_all
        return error, losses_all

    def _query_stream(self, dev):
        s = getattr(self, "_qstream", None)
        if s is None or s.device != dev:
            s = self._qstream = torch.cuda.Stream(dev)
        return s

    def get_errors(self, df_h, df_o, parts_gt, pca_gt, max_dist, body_center, obj_center, **kwargs):
        """training loss of /root/reference/model/chore.py:192-237, averaged over the stacks"""
        w = self.loss_weights
        if df_h.is_cuda and not os.environ.get("CHORE_TORCH_LOSS"):
            return self._get_errors_fused(df_h, df_o, parts_gt, pca_gt, max_dist, body_center, obj_center)
        error, losses_all = 0.0, 0.0
        for df_pred, pca_pred, parts_pred, centers in self.intermediate_preds_list:
            loss_h = self.get_df_loss(df_h, df_pred[:, 0], max_dist) * w[0]
            loss_o = self.get_df_loss(df_o, df_pred[:, 1], max_dist) * w[1]
            loss_parts = (self.part_loss_func(parts_pred, parts_gt) * w[2]).sum(-1).mean()
            mask = (df_o < 0.05).unsqueeze(1).unsqueeze(1)
            loss_pca = ((F.mse_loss(pca_pred, pca_gt, reduction="none") * mask) * w[3]).mean()
            loss_obj = (F.mse_loss(centers[:, 3:, :], obj_center, reduction="none") * mask).mean() * w[4]
            mask = (df_h < 0.05).unsqueeze(1)
            N = mask.shape[2]
            loss_smpl = (F.mse_loss(centers[:, :3, :], body_center.unsqueeze(-1).repeat(1, 1, N),
                                    reduction="none") * mask).mean() * w[5]
            error = error + loss_h + loss_o + loss_parts + loss_pca + loss_smpl + loss_obj
            losses_all = losses_all + torch.stack([loss_h, loss_o, loss_parts, loss_pca, loss_smpl, loss_obj]).detach()
        n = len(self.intermediate_preds_list)
        error = error / n
        losses_all = losses_all / n
        if self.losses_on_host:          # the reference returns a CPU tensor (model/chore.py:226): one host sync per call
            losses_all = losses_all.cpu()
        self.error_buffer = losses_all
        return error, losses_all

    def _get_errors_fused(self, df_h, df_o, parts_gt, pca_gt, max_dist, body_center, obj_center):
        """get_errors with one loss kernel per stack (csrc/train_loss.hip) instead of ~60 tensor ops; CHORE_TORCH_LOSS=1
        selects the tensor-op form above (same values: tests/test_gpu_encoder.py)"""
        n = len(self.intermediate_preds_list)
        f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()   # noqa: E731
        tgt = dict(df_h=f32(df_h), df_o=f32(df_o), parts_gt=parts_gt.long().contiguous(), pca_gt=f32(pca_gt),
                   body_center=f32(body_center), obj_center=f32(obj_center), max_dist=max_dist, scale=1.0 / n,
                   weights=(ctypes.c_float * 6)(*[float(v) for v in self.loss_weights]))
        outs = [_StackLossFn.apply(df, pca, parts, centers, tgt) for df, pca, parts, centers in self.intermediate_preds_list]
        tot = outs[0] if n == 1 else torch.stack(outs).sum(0)
        error, losses_all = tot[6], tot[:6].detach()
        if self.losses_on_host:
            losses_all = losses_all.cpu()
        self.error_buffer = losses_all
        return error, losses_all

    def get_df_loss(self, df_gt, df_pred, max_dist):
        return self.dfloss_func(torch.clamp(df_pred, max=max_dist), torch.clamp(df_gt, max=max_dist)).sum(-1).mean()

    def format_sep_losses(self, losses_all):
        return dict(zip(["df_h", "df_o", "parts", "pca", "smpl", "obj"], losses_all))

    def print_errors(self, errors):
        names = ["df_h", "df_o", "parts", "pca", "smpl", "obj", "grad_h", "grad_o"]
        print(", ".join(f"{n}:{v}" for n, v in zip(names, errors)))

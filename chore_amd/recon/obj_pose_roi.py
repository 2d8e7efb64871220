"""Occlusion-aware silhouette loss on a region of interest, rendered by the HIP rasteriser.

Counterpart of /root/reference/recon/obj_pose_roi.py (SilLossROI) and of the pieces of the vendored
neural_renderer it drives (external/neural_renderer/neural_renderer/: projection.py:6-43, vertices_to_faces.py,
renderer.py:119-152 render_silhouettes, rasterize.py rasterize_silhouettes).  The rasterisation -- forward and the
edge-walk backward -- runs in libchore_hip.so (chore_silhouette_fwd / _bwd); object placement, camera projection and the
triangle list are one launch each way (chore_sil_project_fwd / _bwd; `projection` / `vertices_to_faces` below are the tensor
expressions of the reference, kept for its other callers and as what the tests compare with, `CHORE_SIL_TORCH_PROJECT=1`).

`SilLossROI.forward(R, obj_t, obj_s)` returns what the reference returns: `(loss_dict, image, edges, image_ref,
edt_ref_edge)` with `loss_dict["mask"] = mean_b sum_pixels (keep_mask * silhouette - image_ref)^2`.

Constructor: the reference crops the person / object masks to an expanded square box around the object mask with
detectron2's BitMasks.crop_and_resize and builds the ROI intrinsics from the Kinect calibration
(obj_pose_roi.py:21-98,125-142).  detectron2 / cv2 are not available here; the same steps are restated with numpy
/ torch (bounding box of the mask, PHOSA's make_bbox_square, bilinear crop sampled at pixel centres, threshold 0.5).
That preprocessing is setup code outside the fitting loop and its parity with detectron2 is NOT pinned; callers
that already own the cropped masks can pass them with `SilLossROI.from_crops`.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

NEAR, FAR, EPS = 0.1, 100.0, 1e-4   # rasterize.py:10-12 (rasterize_silhouettes is called with the defaults)


class _RasterizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tri, size):
        if not tri.is_cuda:
            raise RuntimeError("chore_amd needs device tensors (no CPU path)")
        dev = tri.device
        h = _lib.handle(dev.index or 0)
        B, Fn = tri.shape[:2]
        t = tri.detach().float().contiguous()
        fim = torch.empty(B, size, size, dtype=torch.int32, device=dev)
        alpha = torch.empty(B, size, size, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib.chore_silhouette_workspace_bytes(B, Fn), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_silhouette_fwd(h, t.data_ptr(), B, Fn, size, NEAR, FAR, fim.data_ptr(), alpha.data_ptr(),
                                                 ws.data_ptr(), stream), h, "chore_silhouette_fwd")
        ctx.save_for_backward(t, fim, alpha)
        ctx.size = size
        ctx.mark_non_differentiable(fim)
        return alpha.clone(), fim

    @staticmethod
    def backward(ctx, g_alpha, _g_fim):
        t, fim, alpha = ctx.saved_tensors
        dev = t.device
        h = _lib.handle(dev.index or 0)
        B, Fn = t.shape[:2]
        g = torch.zeros_like(alpha) if g_alpha is None else g_alpha.float().contiguous()
        gt = torch.empty_like(t)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_silhouette_bwd(h, t.data_ptr(), fim.data_ptr(), alpha.data_ptr(), g.data_ptr(), B, Fn,
                                                 ctx.size, EPS, gt.data_ptr(), stream), h, "chore_silhouette_bwd")
        return gt, None


class _PlacedTrianglesFn(torch.autograd.Function):
    """object pose -> the rasteriser's (B,2F,3,3) triangle list in one launch each way (chore_sil_project_fwd / _bwd):
    apply_transformation, projection, vertices_to_faces with both windings.  Gradients for R, obj_t, obj_s."""

    @staticmethod
    def forward(ctx, R, obj_t, obj_s, verts, faces32, K, cam_R, cam_t, adj_off, adj):
        dev = verts.device
        h = _lib.handle(dev.index or 0)
        R, obj_t, obj_s = R.float().contiguous(), obj_t.float().contiguous(), obj_s.float().contiguous()
        B, V, _ = verts.shape
        Fn = faces32.shape[1]
        tri = torch.empty(B, 2 * Fn, 3, 3, device=dev, dtype=torch.float32)
        bc = 1 if cam_R.shape[0] == 1 else 0
        _lib.check(_lib.lib.chore_sil_project_fwd(h, verts.data_ptr(), faces32.data_ptr(), R.data_ptr(), obj_t.data_ptr(),
                                                  obj_s.data_ptr(), K.data_ptr(), cam_R.data_ptr(), cam_t.data_ptr(), bc, None, 1.0,
                                                  1e-9, B, V, Fn, tri.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h,
                   "chore_sil_project_fwd")
        ctx.save_for_backward(R, obj_t, obj_s, verts, K, cam_R, cam_t, adj_off, adj)
        ctx.dims = (B, V, Fn, bc)
        return tri

    @staticmethod
    def backward(ctx, g):
        R, obj_t, obj_s, verts, K, cam_R, cam_t, adj_off, adj = ctx.saved_tensors
        dev = verts.device
        h = _lib.handle(dev.index or 0)
        B, V, Fn, bc = ctx.dims
        g = g.float().contiguous()
        dR, dt, ds = torch.empty_like(R), torch.empty_like(obj_t), torch.empty_like(obj_s)
        _lib.check(_lib.lib.chore_sil_project_bwd(h, verts.data_ptr(), R.data_ptr(), obj_t.data_ptr(), obj_s.data_ptr(), K.data_ptr(),
                                                  cam_R.data_ptr(), cam_t.data_ptr(), bc, None, 1.0, 1e-9, B, V, Fn, adj_off.data_ptr(),
                                                  adj.data_ptr(), g.data_ptr(), dR.data_ptr(), dt.data_ptr(), ds.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), h, "chore_sil_project_bwd")
        return (dR, dt, ds) + (None,) * 7


def corner_adjacency(faces, num_verts):
    """for every vertex the entries (f2 * 3 + corner) of the doubled face list (faces, then faces with reversed corners) that
    reference it, ascending: (offsets (V+1) int32, entries int32)"""
    f = np.asarray(faces, np.int64)
    both = np.concatenate([f, f[:, ::-1]], 0).reshape(-1)           # entry e = f2 * 3 + corner -> vertex
    order = np.argsort(both, kind="stable")
    counts = np.bincount(both, minlength=num_verts)
    off = np.zeros(num_verts + 1, np.int64)
    off[1:] = np.cumsum(counts)
    return off.astype(np.int32), order.astype(np.int32)


def projection(vertices, K, R, t, dist_coeffs=None, orig_size=1.0, eps=1e-9):
    """[X,Y,Z] -> [u, v in [-1,1], z]   (neural_renderer/projection.py:6-43, op for op)"""
    v = torch.matmul(vertices, R.transpose(2, 1)) + t
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    x_ = x / (z + eps)
    y_ = y / (z + eps)
    if dist_coeffs is None:
        dist_coeffs = torch.zeros(1, 5, device=vertices.device)
    k1, k2, p1, p2, k3 = (dist_coeffs[:, None, i] for i in range(5))
    r = torch.sqrt(x_ ** 2 + y_ ** 2)
    x__ = x_ * (1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)) + 2 * p1 * x_ * y_ + p2 * (r ** 2 + 2 * x_ ** 2)
    y__ = y_ * (1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)) + p1 * (r ** 2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
    h = torch.stack([x__, y__, torch.ones_like(z)], dim=-1)
    uv = torch.matmul(h, K.transpose(1, 2))
    u, vv = uv[:, :, 0], orig_size - uv[:, :, 1]
    u = 2 * (u - orig_size / 2.0) / orig_size
    vv = 2 * (vv - orig_size / 2.0) / orig_size
    return torch.stack([u, vv, z], dim=-1)


def vertices_to_faces(vertices, faces):
    """(B,V,3), (B,F,3) int -> (B,F,3,3)"""
    B, nv = vertices.shape[:2]
    idx = faces.long() + (torch.arange(B, device=vertices.device) * nv)[:, None, None]
    return vertices.reshape(B * nv, 3)[idx]


def render_silhouettes(vertices, faces, K, R, t, size, dist_coeffs=None, orig_size=1.0, fill_back=True):
    """nr.Renderer(camera_mode='projection', anti_aliasing=False)(vertices, faces, mode='silhouettes')
    (renderer.py:119-152): (B,size,size) images in {0,1}, differentiable w.r.t. the vertices"""
    if fill_back:
        faces = torch.cat((faces, faces.flip(-1)), dim=1)
    tri = vertices_to_faces(projection(vertices, K, R, t, dist_coeffs, orig_size), faces)
    alpha, _ = _RasterizeFn.apply(tri, size)
    return alpha.flip(1)          # rasterize_rgbad reverses the rows (rasterize.py:345)


# ---- mask preprocessing (setup, not on the fitting loop) ----------------------------------------------------
def mask2bbox(mask):
    """xyxy bounding box of mask > 0.5 (reference: cv2 contours of the thresholded mask, opt_utils.py:105-116)"""
    ys, xs = np.nonzero(np.asarray(mask) > 0.5)
    if ys.size == 0:
        raise ValueError("empty object mask")
    return np.array([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1], np.float64)


def make_bbox_square(bbox_xywh, expansion=0.0):
    """PHOSA's square box around an xywh box (recon/bbox.py:25-46)"""
    b = np.asarray(bbox_xywh, np.float64).reshape(-1, 4)
    c = np.stack((b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2), 1)
    s = np.maximum(b[:, 2], b[:, 3])[:, None] * (1 + expansion)
    return np.hstack((c - s / 2, s, s)).reshape(np.asarray(bbox_xywh).shape)


def crop_and_resize(masks, boxes_xyxy, size):
    """bilinear crop of (B,H,W) masks to (B,size,size) sampled at the pixel centres of the output grid, then
    thresholded at 0.5 (what detectron2's BitMasks.crop_and_resize computes with roi_align(aligned=True))"""
    B, H, W = masks.shape
    out = []
    for m, bx in zip(masks.float(), boxes_xyxy):
        x0, y0, x1, y1 = [float(v) for v in bx]
        xs = x0 + (torch.arange(size, device=m.device, dtype=torch.float32) + 0.5) * (x1 - x0) / size - 0.5
        ys = y0 + (torch.arange(size, device=m.device, dtype=torch.float32) + 0.5) * (y1 - y0) / size - 0.5
        gx = 2 * xs / max(W - 1, 1) - 1
        gy = 2 * ys / max(H - 1, 1) - 1
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], -1)[None]
        out.append(F.grid_sample(m[None, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0])
    return torch.stack(out) >= 0.5


class SilLossROI(nn.Module):
    net_input_size = 512

    def __init__(self, person_masks, obj_masks, temp_mesh, crop_centers, rend_size=256, kernel_size=7,
                 bbox_expansion=0.3, device="cuda:0", crop_size=1200):
        """the reference's argument list (obj_pose_roi.py:21-29).  person_masks / obj_masks: (B,512,512) network-input
        masks; temp_mesh: the centred template, an object with .v (V,3) and .f (F,3) or a (verts, faces) pair;
        crop_centers (B,2) in original-image pixels"""
        super().__init__()
        dev = torch.device(device)
        temp_verts, temp_faces = (temp_mesh.v, temp_mesh.f) if hasattr(temp_mesh, "v") else temp_mesh
        person_masks, obj_masks = torch.as_tensor(person_masks).to(dev), torch.as_tensor(obj_masks).to(dev)
        boxes = np.stack([mask2bbox(m.cpu().numpy()) for m in obj_masks])                       # xyxy
        xywh = np.concatenate([boxes[:, :2], boxes[:, 2:] - boxes[:, :2]], 1)
        sq = make_bbox_square(xywh, bbox_expansion)                                             # xywh
        sq_xyxy = np.concatenate([sq[:, :2], sq[:, :2] + sq[:, 2:]], 1)
        obj_crop = crop_and_resize(obj_masks, sq_xyxy, rend_size)
        ps_crop = crop_and_resize(person_masks, sq_xyxy, rend_size)
        scale = crop_size / float(self.net_input_size)
        Ks = []
        cc = torch.as_tensor(crop_centers).float().cpu().numpy()
        for bbox, c in zip(sq, cc):
            Ks.append(self.compute_K_roi(self.to_original_bbox(bbox, scale, c, crop_size)))
        self._setup(obj_crop, ps_crop, torch.cat(Ks, 0).to(dev), temp_verts, temp_faces, rend_size, kernel_size, dev)

    @classmethod
    def from_crops(cls, obj_crop, person_crop, K_rois, temp_verts, temp_faces, kernel_size=7, device="cuda:0"):
        """build from already cropped (B,S,S) boolean masks and (B,3,3) ROI intrinsics"""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        dev = torch.device(device)
        obj_crop = torch.as_tensor(obj_crop).to(dev) > 0
        person_crop = torch.as_tensor(person_crop).to(dev) > 0
        self._setup(obj_crop, person_crop, torch.as_tensor(K_rois).float().to(dev), temp_verts, temp_faces,
                    obj_crop.shape[-1], kernel_size, dev)
        return self

    def _setup(self, obj_crop, ps_crop, Ks, temp_verts, temp_faces, rend_size, kernel_size, dev):
        B = obj_crop.shape[0]
        self.rend_size = rend_size
        self.register_buffer("image_ref", obj_crop.float())                    # edges / target: the object mask only
        self.register_buffer("keep_mask", self.cvt_masks(ps_crop, obj_crop).float())
        self.pool = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
        self.prepare_dist_trans(self.image_ref)
        verts = torch.as_tensor(np.asarray(temp_verts), dtype=torch.float32)
        faces = torch.as_tensor(np.asarray(temp_faces).astype(np.int64))
        self.register_buffer("vertices", verts.repeat(B, 1, 1).to(dev))
        self.register_buffer("faces", faces.repeat(B, 1, 1).to(dev))
        self.register_buffer("K", Ks)
        self.register_buffer("R", torch.eye(3, device=dev).unsqueeze(0))
        self.register_buffer("t", torch.zeros(1, 3, device=dev))
        # for the one-launch placement + projection (_PlacedTrianglesFn): int32 faces and the corner lists of the vertices
        off, ent = corner_adjacency(faces.numpy(), verts.shape[0])
        self.register_buffer("faces32", self.faces.to(torch.int32).contiguous(), persistent=False)
        self.register_buffer("adj_off", torch.from_numpy(off).to(dev), persistent=False)
        self.register_buffer("adj", torch.from_numpy(ent).to(dev), persistent=False)

    @torch.no_grad()
    def load_from(self, other):
        """take over another instance's per-image data IN PLACE (reference and keep masks, ROI cameras, edge distance
        transform): a recorded fitting step keeps reading this object's buffers across loader batches
        (recon_fit_behave._FitSlot); the template (vertices, faces, corner lists) does not change"""
        for name in ("image_ref", "keep_mask", "K", "edt_ref_edge"):
            mine, theirs = getattr(self, name), getattr(other, name)
            if mine.shape != theirs.shape:
                raise ValueError("SilLossROI.load_from: %s has another shape" % name)
            mine.copy_(theirs)
        if self.vertices.shape != other.vertices.shape or not torch.equal(self.vertices, other.vertices):
            raise ValueError("SilLossROI.load_from: another template")
        return self

    def prepare_dist_trans(self, image_refs, power=0.25):
        """distance transform of the reference edges (obj_pose_roi.py:92-103); debugging output of forward()"""
        from scipy.ndimage import distance_transform_edt
        edges = self.compute_edges(image_refs).cpu().numpy()
        edt = np.stack([distance_transform_edt(1 - (e > 0)) ** (power * 2) for e in edges])
        self.register_buffer("edt_ref_edge", torch.from_numpy(edt).float().to(image_refs.device))

    def compute_edges(self, silhouette):
        return self.pool(silhouette) - silhouette

    @staticmethod
    def to_original_bbox(bbox_square, scale, trans, crop_size=1200):
        b = np.array(bbox_square, np.float64)
        b *= scale
        b[:2] += np.asarray(trans, np.float64) - crop_size / 2.0
        return b

    @staticmethod
    def compute_K_roi(bbox_square, kinect_width=2048):
        """intrinsics of a camera that sees only the square box (obj_pose_roi.py:125-142)"""
        x, y, b, w = bbox_square
        if abs(b - w) > 1e-6:
            raise ValueError("the given bbox is not square")
        fx, fy = 979.7844 / kinect_width, 979.840 / kinect_width
        cx, cy = 1018.952 / kinect_width, 779.486 / kinect_width
        return torch.tensor([[[fx * kinect_width / b, 0, (cx * kinect_width - x) / b],
                              [0, fy * kinect_width / b, (cy * kinect_width - y) / b],
                              [0, 0, 1]]], dtype=torch.float32)

    @staticmethod
    def cvt_masks(person_mask, obj_mask):
        """1 where the rendering counts (object foreground or free background), 0 where a person occludes
        (obj_pose_roi.py:144-157)"""
        fore, ps = obj_mask > 0.5, person_mask > 0.5
        inv = -ps.float()
        inv[fore] = 1.0
        return inv >= 0

    def apply_transformation(self, R, obj_t, obj_s):
        verts = torch.bmm(self.vertices, R) + obj_t.unsqueeze(1)
        return obj_s.view(-1, 1, 1) * verts

    def render(self, R, obj_t, obj_s):
        """silhouettes (B,S,S) of the template under the pose: apply_transformation -> render_silhouettes"""
        if (not os.environ.get("CHORE_SIL_TORCH_PROJECT") and R.is_cuda and self.vertices.is_cuda and
                tuple(obj_s.shape) == (self.vertices.shape[0],)):
            tri = _PlacedTrianglesFn.apply(R, obj_t, obj_s, self.vertices, self.faces32, self.K, self.R, self.t, self.adj_off, self.adj)
            alpha, _ = _RasterizeFn.apply(tri, self.rend_size)
            return alpha.flip(1)
        verts = self.apply_transformation(R, obj_t, obj_s)
        return render_silhouettes(verts, self.faces, self.K, self.R, self.t, self.rend_size)

    def mask_loss(self, R, obj_t, obj_s):
        """forward()[0] without the outputs the fitting loop does not read (the edge image costs a max-pool per step)"""
        image = self.keep_mask * self.render(R, obj_t, obj_s)
        return {"mask": torch.sum((image - self.image_ref) ** 2, dim=(1, 2)).mean()}, image

    def forward(self, R, obj_t, obj_s):
        loss_dict, image = self.mask_loss(R, obj_t, obj_s)
        return loss_dict, image, self.compute_edges(image), self.image_ref, self.edt_ref_edge

    def compute_offscreen_loss(self, verts):
        """penalty for leaving the view frustum (obj_pose_roi.py:179-199)"""
        proj = projection(verts, self.K, self.R, self.t, None, 1.0)
        xy, z = proj[:, :, :2], proj[:, :, 2:]
        zeros = torch.zeros_like(z)
        return (torch.max(xy - 1, zeros).sum(dim=(1, 2)) + torch.max(-1 - xy, zeros).sum(dim=(1, 2)) +
                torch.max(-z, zeros).sum(dim=(1, 2)) + torch.max(z - FAR, zeros).sum(dim=(1, 2)))

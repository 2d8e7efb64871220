"""Fitting maths of CHORE on the GPU: SO(3) projection, object transform, loss terms.

Counterpart of /root/reference/recon/recon_fit_base.py (ReconFitterBase) for the methods that sit on the
optimisation hot path; file I/O, mesh viewers and dataset glue of that class are out of scope.
Method names, signatures and loss-dict keys are the reference's, so recon_fit_behave-style drivers read
the same.  Heavy steps run in libchore_hip.so: the field queries (chore_query_fwd/bwd_points), SMPL-H LBS
(chore_smpl_lbs_*) and the SO(3) projection (chore_so3_project_*); the remaining terms are a few
reductions over (B,N) tensors expressed with torch ops on the device.
"""
import os
import threading
import pickle as pkl

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib
from ..lib_smpl.const import SMPL_PARTS_NUM, SMPL_POSE_PRAMS_NUM  # noqa: F401
from ..lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatchSplitParams
from ..model.camera import KinectColorCamera
from . import fit_terms
from ..utils.paths import BEHAVE_PATH, RECON_PATH, SMPL_ASSETS_ROOT  # noqa: F401  (names the reference's scripts import from here, recon_fit_base.py:39-45)


# 14 body-part colours of the visualisations (recon/opt_utils.py:13-28)
MTURK_COLORS = np.array([44, 160, 44, 31, 119, 180, 255, 127, 14, 214, 39, 40, 148, 103, 189, 140, 86, 75, 227, 119, 194,
                         127, 127, 127, 189, 189, 34, 255, 152, 150, 23, 190, 207, 174, 199, 232, 255, 187, 120, 152, 223,
                         138]).reshape((-1, 3)) / 255.


def write_ply(path, verts, faces=None):
    """binary little-endian PLY (float32 vertices, int32 triangle lists)"""
    verts = np.asarray(verts, "<f4")
    with open(path, "wb") as f:
        hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {len(verts)}", "property float x",
               "property float y", "property float z"]
        if faces is not None:
            hdr += [f"element face {len(faces)}", "property list uchar int vertex_indices"]
        f.write(("\n".join(hdr) + "\nend_header\n").encode("ascii"))
        f.write(verts.tobytes())
        if faces is not None:
            rec = np.zeros(len(faces), np.dtype([("n", "u1"), ("v", "<i4", (3,))]))
            rec["n"], rec["v"] = 3, np.asarray(faces)
            f.write(rec.tobytes())


# The CPU random stream the optimisation of the batch fitted by THIS host thread draws from (the SO(3) perturbations): None = the
# process-wide generator, like the reference; fit_recon with `batch_seed` gives every batch a generator of its own, so that
# batches fitted side by side (recon_fit_behave._fit_concurrent) draw what they draw in the serial loop.
_BATCH_RNG = threading.local()


def cpu_generator():
    return getattr(_BATCH_RNG, "gen", None)


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
        # a cloud that is a constant of the caller (the body in optimize_smpl_object) gets no gradient: NULL skips its kernels
        d_hum = torch.empty_like(hum) if ctx.needs_input_grad[0] else None
        d_obj = torch.empty_like(obj) if ctx.needs_input_grad[1] else None
        if d_hum is None and d_obj is None:
            return (None,) * 7
        g = g.float().contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.chore_contact_bwd(h, hum.data_ptr(), obj.data_ptr(), lab.data_ptr(), B, Nh, No, ctx.P,
                                              g.data_ptr(), ws.data_ptr(), None if d_hum is None else d_hum.data_ptr(),
                                              None if d_obj is None else d_obj.data_ptr(), stream),
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


class _Scan:
    """object template: .v (V,3) float64 centred vertices, .f (F,3) int64 faces (the two attributes of psbody's Mesh
    the reference uses)"""

    def __init__(self, v, f):
        self.v, self.f = np.asarray(v, np.float64), np.asarray(f, np.int64)


def sample_surface(verts, faces, count, rs):
    """`count` points uniformly on the triangle mesh (area-weighted triangle choice + uniform barycentric point) -- what
    trimesh.Trimesh.sample does (recon_fit_base.py:120-121), drawn from the numpy RandomState `rs`"""
    tri = verts[faces]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    idx = np.searchsorted(np.cumsum(area), rs.random_sample(count) * area.sum())
    idx = np.minimum(idx, len(faces) - 1)
    u, v = rs.random_sample(count), rs.random_sample(count)
    flip = u + v > 1
    u, v = np.where(flip, 1 - u, u), np.where(flip, 1 - v, v)
    t = tri[idx]
    return t[:, 0] + u[:, None] * (t[:, 1] - t[:, 0]) + v[:, None] * (t[:, 2] - t[:, 0])


class ReconFitterBase:
    def __init__(self, seq_folder=None, device="cuda:0", debug=False, obj_name=None, outpath=RECON_PATH, args=None,
                 assets=None):
        """the reference's argument list (recon_fit_base.py:48-52) + `assets`: where the files it reads come from
        (recon/assets.py; default: the folders of ./PATHS.yml like the reference).  Everything loaded here -- template
        PCA axes and surface samples, part labels, priors -- lands on the device once."""
        from .assets import FileAssets
        self.seq_folder, self.outpath = seq_folder, outpath
        self.assets = assets if assets is not None else FileAssets.from_paths_yml()
        info = self.assets.seq_info(seq_folder) if seq_folder is not None else None
        if info is None:
            if obj_name is None:
                raise AssertionError("must provide the name of the object to be reconstructed!")
            self.gender = "male"
        else:
            obj_name, self.gender = info
        self._init_common(device, debug)
        pca_init, obj_points = self.compute_pca_init(obj_name)
        self.pca_init = torch.tensor(pca_init, dtype=torch.float32).to(self.device)
        self.obj_points = torch.tensor(obj_points, dtype=torch.float32).to(self.device)
        self.part_labels = self.load_part_labels()
        crop = args.loadSize if args is not None else 1200
        self.camera = KinectColorCamera(crop)
        self.net_in_size = args.net_img_size[0] if args is not None else 512
        self.z_0 = 2.2 if args is None or "z_0" not in args else args.z_0
        self.scan_verts = torch.as_tensor(self.scan.v, dtype=torch.float32, device=self.device)
        self.scan_faces = torch.as_tensor(self.scan.f, dtype=torch.long, device=self.device)
        self.body_prior, self.hand_prior = self.assets.priors(self.device)

    def _init_common(self, device, debug):
        self.device = torch.device(device)
        self.debug = debug
        self.obj_scale = 1.0
        self.scan = None
        self.scan_verts = self.scan_faces = None
        self._comb_faces = None
        self._layers = {}
        self.part_names = {0: "head", 1: "left foot", 2: "left hand", 3: "left leg", 4: "left midarm",
                           5: "left upper arm", 6: "right foot", 7: "right hand", 8: "right leg", 9: "right midarm",
                           10: "right right upper arm", 11: "torso", 12: "upper left leg", 13: "upper right leg"}

    @classmethod
    def from_parts(cls, device="cuda:0", net_in_size=512, crop_size=1200, z_0=2.2, obj_scale=1.0, part_labels=None,
                   body_prior=None, hand_prior=None, debug=False, scan_verts=None, scan_faces=None, assets=None):
        """a fitter from values already in memory (tests, benchmarks): no sequence folder, no files"""
        self = cls.__new__(cls)
        self.seq_folder = self.outpath = None
        self.assets, self.gender = assets, "male"
        self._init_common(device, debug)
        if scan_verts is not None:
            self.scan = _Scan(np.asarray(torch.as_tensor(scan_verts).cpu()), np.asarray(torch.as_tensor(scan_faces).cpu()))
            self.scan_verts = torch.as_tensor(scan_verts, dtype=torch.float32, device=device)
            self.scan_faces = torch.as_tensor(scan_faces, dtype=torch.long, device=device)
        self.camera = KinectColorCamera(crop_size)
        self.net_in_size, self.z_0, self.obj_scale = net_in_size, z_0, obj_scale
        self.part_labels = part_labels          # (6890,) long: assets/smpl_parts_dense.pkl in the reference
        self.body_prior, self.hand_prior = body_prior, hand_prior
        return self

    # ---- what the reference loads from files ---------------------------------------------------------
    def compute_pca_init(self, obj_name, n_samples=3000, seed=0):
        """template -> centred mesh (self.scan), its PCA axes (3,3) and 3 000 surface samples  [recon_fit_base.py:108-122]
        PCA = sklearn's, like the reference (its sign convention fixes which of the +-axes the object starts from)"""
        from sklearn.decomposition import PCA
        v, f = self.assets.template(obj_name)
        v = np.asarray(v, np.float64)
        v = v - np.mean(v, 0)          # load_scan_centered
        v = v - np.mean(v, 0)          # and once more in compute_pca_init (:115)
        self.scan = _Scan(v, f)
        pca = PCA(n_components=3)
        pca.fit(v)
        return pca.components_, sample_surface(self.scan.v, self.scan.f, n_samples, np.random.RandomState(seed))

    def load_part_labels(self):
        return torch.tensor(np.asarray(self.assets.part_labels(), np.int32)).to(self.device)

    def load_part_labels_batch(self, batch_size):
        return self.load_part_labels().repeat(batch_size, 1).to(self.device).long()

    def load_mocap(self, file):
        return self.assets.load_mocap(file)

    def smpl_layer(self, gender):
        """one packed body model per gender, kept on the device"""
        if gender not in self._layers:
            from ..lib_smpl.smpl_layer import SMPL_Layer
            self._layers[gender] = SMPL_Layer.from_arrays(self.assets.smpl_model(gender), gender=gender).to(self.device)
        return self._layers[gender]

    def get_smpl_init(self, image_paths, trans):
        """SMPL-H initialised from the FrankMocap prediction next to every image  [recon_fit_base.py:124-140]"""
        from ..lib_smpl.smpl_generator import SMPLHGenerator
        poses, betas = [], []
        for x in image_paths:
            p, b = self.load_mocap(x.replace(".color.jpg", ".mocap.json"))
            poses.append(p)
            betas.append(b)
        return SMPLHGenerator.get_smplh(np.stack(poses, 0), np.stack(betas, 0), trans, self.gender, device=self.device,
                                        assets=self.assets, layer=self.smpl_layer(self.gender))

    def get_kpt_paths(self, image_paths):
        return [x.replace(".color.jpg", ".color.json") for x in image_paths]

    def load_kpts(self, json_paths, tol):
        """(B,25,3) 2-D body keypoints in the original image, confidences below `tol` zeroed  [:305-320]"""
        kpts = []
        for file in json_paths:
            J2d = np.array(self.assets.load_kpts(file), np.float64).reshape((-1, 3))
            J2d[:, 2][J2d[:, 2] < tol] = 0
            kpts.append(J2d)
        return torch.tensor(np.stack(kpts, 0), dtype=torch.float32).to(self.device)

    def get_body_kpts2d(self, traindata_paths, tol=0.3):
        return self.load_kpts(self.get_kpt_paths(traindata_paths), tol)

    def get_parts_colors(self, part):
        """(B,N) labels -> (B,N,3) colours, visualisation only  [:652-659]"""
        part = part.cpu().numpy() if torch.is_tensor(part) else np.asarray(part)
        return MTURK_COLORS[np.clip(part, 0, 13)]

    def prepare_query_dict(self, batch):
        return {"crop_center": batch.get("crop_center").to(self.device)}

    def prep_smplfit(self, data, generator, pc_generated):
        """everything optimize_smpl needs, from a loader batch and the generated point clouds  [recon_fit_base.py:398-440]"""
        batch_size = data["images"].shape[0]
        human_points = pc_generated["human"]["points"].clone().detach().to(self.device)
        obj_points = pc_generated["object"]["points"].clone().detach().to(self.device)
        human_parts = pc_generated["human"]["parts"].clone().detach().to(self.device)
        human_t = pc_generated["human"]["centers"][:, :3]
        human_t[:, 2] = self.z_0                                   # in place, like the reference: fixed SMPL depth
        smpl = self.get_smpl_init(data.get("path"), human_t)
        part_labels = self.load_part_labels_batch(batch_size)
        part_colors = self.get_parts_colors(pc_generated["human"]["parts"])
        body_kpts = self.get_body_kpts2d(data.get("path"))
        body_kpts = self.scale_body_kpts(body_kpts, data.get("resize_scale").to(self.device),
                                         data.get("crop_scale").to(self.device), data.get("old_crop_center").to(self.device))
        body_kpts = body_kpts.clone().float().to(self.device)
        query_dict = self.prepare_query_dict(data)
        betas_dict = {"images": data.get("images").to(self.device), "human_init": human_points, "human_parts": human_parts,
                      "part_labels": part_labels, "part_colors": part_colors, "body_kpts": body_kpts,
                      "query_dict": query_dict, "net": generator.model,
                      "pose_init": smpl.pose[:, 3:SMPL_POSE_PRAMS_NUM].clone().detach().to(self.device)}
        return (betas_dict, body_kpts, human_parts, human_points, human_t, obj_points, part_colors, part_labels, query_dict,
                smpl)

    def init_obj_fit_data(self, batch_size, human_t, pc_generated, scale):
        """object translation from the predicted centres (relative to the SMPL centre), rotation from the predicted PCA
        axes against the template's, scale from the SMPL height ratio  [recon_fit_base.py:720-747]"""
        obj_t = pc_generated["object"]["centers"][:, 3:].to(self.device) + human_t.to(self.device)
        obj_t = obj_t.clone().detach().to(self.device).requires_grad_(True)
        pca_axis = pc_generated["object"]["pca_axis"].to(self.device)
        pca_axis_init = torch.stack([self.pca_init for _ in range(batch_size)], 0)
        obj_R = self.init_object_orientation(pca_axis, pca_axis_init).detach().requires_grad_(True)
        obj_s = scale.clone().detach().to(self.device).requires_grad_(True)
        object_init = torch.stack([self.obj_points for _ in range(batch_size)], 0)
        return obj_R, obj_s, obj_t, object_init

    # ---- results on disk (recon_fit_base.py:233-275) ---------------------------------------------------
    def get_output_paths(self, image_paths, save_name, test_id):
        seq_names = [x.split(os.sep)[-3] for x in image_paths]
        frame_times = [x.split(os.sep)[-2] for x in image_paths]
        smpl_files, obj_files = [], []
        for seq, frame in zip(seq_names, frame_times):
            folder = os.path.join(self.outpath, seq, frame, save_name)
            os.makedirs(folder, exist_ok=True)
            smpl_files.append(os.path.join(folder, f"k{test_id}.smpl.ply"))
            obj_files.append(os.path.join(folder, f"k{test_id}.object.ply"))
        return smpl_files, obj_files

    def is_done(self, image_paths, save_name, test_id):
        smpl_files, obj_files = self.get_output_paths(image_paths, save_name, test_id)
        return all(os.path.isfile(sf) and os.path.isfile(of) for sf, of in zip(smpl_files, obj_files))

    def save_outputs(self, smpl, obj_R, obj_t, traindata_paths, save_name, test_id, obj_s=None):
        """fitted SMPL mesh + parameters and object mesh + parameters next to each other  [:258-275, opt_utils.py:73-101]"""
        smpl_files, obj_files = self.get_output_paths(traindata_paths, save_name, test_id)
        smpl.forget()
        with torch.no_grad():
            verts = smpl()[0].cpu().numpy()
            faces = None if smpl.faces is None else smpl.faces.cpu().numpy()
            B = len(obj_files)
            obj_verts = self.scan_verts.unsqueeze(0).repeat(B, 1, 1)
            obj_verts = self.transform_object(obj_verts, obj_R, obj_t, obj_s).cpu().numpy()
            rot = self.decopose_axis(obj_R, no_rand=True).cpu().numpy()
        for i, (sf, of) in enumerate(zip(smpl_files, obj_files)):
            write_ply(sf, verts[i], faces)
            pkl.dump({"pose": smpl.pose[i].detach().cpu().numpy(), "betas": smpl.betas[i].detach().cpu().numpy(),
                      "trans": smpl.trans[i].detach().cpu().numpy(), "score": 0.0}, open(sf.replace(".ply", ".pkl"), "wb"))
            write_ply(of, obj_verts[i], self.scan.f)
            pkl.dump({"rot": rot[i], "trans": obj_t[i].detach().cpu().numpy(), "scale": obj_s[i].detach().cpu().numpy()},
                     open(of.replace(".ply", ".pkl"), "wb"))

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
            noise = torch.rand(rot.shape[0], 3, 3, generator=cpu_generator())
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
        if fit_terms.obj_transform_supported(verts, obj_R, obj_t, obj_s):
            return fit_terms.obj_transform(verts, obj_R, obj_t, obj_s)     # one launch each way (csrc/fit_terms.hip)
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
        """[recon_fit_behave.py:339-358] sum_k w_k(L_k, it) with w_k = c_k * L / (1 + it).  Inside the steppers `it` stands for
        a device scalar (graph_step._OnePlusDecay) and the terms are device scalars: one launch computes the sum and one its
        backward (csrc/fit_step.hip) instead of two tensor ops per term, a stack and a reduction, each way."""
        from .graph_step import _OnePlusDecay, weighted_sum
        vals = list(loss_dict.values())
        if isinstance(it, _OnePlusDecay) and vals and all(torch.is_tensor(v) and v.is_cuda and v.dim() == 0 and
                                                          v.dtype == torch.float32 for v in vals) and len(vals) <= 16:
            return weighted_sum(vals, [float(weight_dict[k](1.0, 0)) for k in loss_dict], it.denom)
        return torch.stack([weight_dict[k](v, it) for k, v in loss_dict.items()]).sum()

    def compute_obj_loss(self, data_dict, loss_dict, model, obj_s, object, preds=None, scale_term=True):
        """`preds`: the field at `object` when the caller has queried these very points already in this step (the
        reference queries them twice per step, recon_fit_behave.py:171,183 -> recon_fit_base.py:505-506, with the same
        result): one query forward and one backward instead of two"""
        if preds is None:
            model.query(object, **data_dict["query_dict"])
            preds = model.get_preds()
        if fit_terms.point_terms_supported(preds[0]):
            loss_dict["object"] = fit_terms.point_terms(preds[0], 1, 0.8)[0]
        else:
            loss_dict["object"] = torch.clamp(preds[0][:, 1:2, :], max=0.8).mean()
        if scale_term:
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
        if fit_terms.point_terms_supported(df_pred):
            loss_dict["df_h"] = fit_terms.point_terms(df_pred, 0, 0.1)[0]
        else:
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
        """3-D keypoints -> pixels of the network input image  [recon_fit_base.py:661-670]"""
        px, py = self.camera.project_screen(joints3d, crop_center)
        return torch.cat([px, py], -1) * self.net_in_size / self.camera.crop_size

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
        smpl.betas.data[:, :2] = split_smpl.top_betas.data
        # the reference copies no other_betas here (:682-690) and does not need to: the split parameters are views of
        # smpl's storage (wrapper_pytorch.from_smpl), the optimised betas 2..9 are in smpl.betas already.  For a split
        # that owns its storage the line below keeps that outcome; for views it copies a tensor onto itself.
        smpl.betas.data[:, 2:] = split_smpl.other_betas.data
        smpl.trans.data = split_smpl.trans.data
        smpl.forget()   # writes through .data are invisible to the version counters the LBS memo is keyed on
        return smpl

    @staticmethod
    def get_smpl_height(smpl):
        with torch.no_grad():
            v = smpl()[0]
            return v[:, :, 1].max(1)[0] - v[:, :, 1].min(1)[0]

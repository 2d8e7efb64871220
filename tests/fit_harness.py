"""Synthetic inputs shared by tests/golden/make_golden.py (which feeds them to THE REFERENCE's drivers on CPU) and the
GPU tests (which feed them to chore_amd's): plain numpy arrays + two tiny torch stand-ins for pieces neither side's
driver logic depends on.  Nothing here is reference code.

  AnalyticField  -- a smooth closed-form "field network" with the query()/get_preds() surface of CHORE.  The Generator's
                    control flow (mask, first-round skip, resampling, per-example lists, compose_outdict,
                    recon/generator.py:102-217) is pinned with it: with the real network a last-bit difference flips
                    a df < filter_val decision and everything after the next randint(k) diverges, so an exact
                    comparison of the LOOP needs a field whose values do not sit within round-off of the threshold.
  SilStub        -- a differentiable stand-in for SilLossROI.forward (the reference's needs the CUDA-only
                    neural_renderer): the 'sil' phase of optimize_smpl_object is pinned for its schedule, decay,
                    optimiser re-creation and random-stream order, the silhouette kernels by their own tests.
"""
import numpy as np
import torch

from chore_amd.utils import synth


# ---------------------------------------------------------------------------------------------------------------------
class AnalyticField(torch.nn.Module):
    """df_k = 0.4 * | |p - c_k[b]| - r_k |  (k = human, object), part logits / centres linear and pca sinusoidal in p.
    0.4: a projection step of Alg. 1 removes 40 % of the distance, so after 10 steps only points that started within
    ~1.65 m of the surface pass df < 0.004.  Example 1's spheres sit where none of its U[0,1]^3 initial samples
    (reference quirk, generator.py:275-282) passes in the first round -> the `else` branch of the resampling runs."""

    def __init__(self):
        super().__init__()
        rs = np.random.RandomState(77)
        t = lambda a: torch.tensor(np.asarray(a, np.float32))   # noqa: E731
        self.register_buffer("c", t([[[0.1, 0.3, 2.2], [0.5, 0.2, 2.3]], [[0.5, 0.5, 3.0], [0.6, 0.4, 3.1]]]))  # (B,2,3)
        self.register_buffer("r", t([[0.6, 0.3], [0.3, 0.25]]))                                               # (B,2)
        self.register_buffer("A", t(rs.standard_normal((14, 3))))
        self.register_buffer("b", t(rs.standard_normal(14)))
        self.register_buffer("W", t(rs.standard_normal((9, 3)) * 0.7))
        self.register_buffer("phi", t(rs.uniform(0, 6.28, 9)))
        self.register_buffer("C", t(rs.standard_normal((6, 3)) * 0.2))
        self.register_buffer("c0", t(rs.standard_normal(6) * 0.1))
        self.preds = None
        self.n_filter = 0

    def filter(self, images):
        self.n_filter += 1

    def query(self, points, crop_center=None, **kw):
        B, N, _ = points.shape
        d = points.unsqueeze(1) - self.c[:B].unsqueeze(2)                       # (B,2,N,3)
        df = 0.4 * (torch.sqrt((d * d).sum(-1)) - self.r[:B].unsqueeze(-1)).abs()   # (B,2,N)
        parts = torch.einsum("kc,bnc->bkn", self.A, points) + self.b.view(1, 14, 1)
        pca = torch.sin(torch.einsum("kc,bnc->bkn", self.W, points) + self.phi.view(1, 9, 1)).view(B, 3, 3, N)
        centers = torch.einsum("kc,bnc->bkn", self.C, points) + self.c0.view(1, 6, 1)
        self.preds = (df, pca, parts, centers)

    def get_preds(self):
        return self.preds


class SilStub(torch.nn.Module):
    """forward(R, t, s) -> ({'mask': ...}, None, None, None, None)   (call surface of recon/obj_pose_roi.py:159-172).
    'mask' = 2e5 * mean |(P R + t) s - T|^2 over 40 fixed points: same magnitude as a pixel-count silhouette loss."""

    def __init__(self, B):
        super().__init__()
        rs = np.random.RandomState(78)
        self.register_buffer("P", torch.tensor((rs.standard_normal((B, 40, 3)) * 0.3).astype(np.float32)))
        self.register_buffer("T", torch.tensor((rs.standard_normal((B, 40, 3)) * 0.3 + [0.25, 0.35, 2.25]).astype(np.float32)))

    @torch.no_grad()
    def load_from(self, other):
        """(the in-place update recon_fit_behave._FitSlot asks of a silhouette term it keeps across calls)"""
        self.P.copy_(other.P)
        self.T.copy_(other.T)
        return self

    def forward(self, R, t, s):
        x = (torch.bmm(self.P, R) + t.unsqueeze(1)) * s.view(-1, 1, 1)
        return {"mask": 2e5 * ((x - self.T) ** 2).mean()}, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------------
def fit_case(B=2):
    """numpy inputs of one fitting problem (same recipe as round 1's trajectories)"""
    rs = np.random.RandomState(9)
    feat = (rs.standard_normal((B, 256, 32, 32)) * 0.5).astype(np.float32)
    tmpx = (rs.standard_normal((B, 64, 64, 64)) * 0.5).astype(np.float32)
    pose, betas, trans = synth.synth_smpl_params(B, seed=1)
    pose = pose * 0.3
    labels = rs.randint(0, 14, 6890)
    kpts = np.concatenate([rs.uniform(100, 400, (B, 25, 2)), rs.uniform(0.2, 1, (B, 25, 1))], -1).astype(np.float32)
    obj = (rs.standard_normal((B, 3000, 3)) * 0.15).astype(np.float32)
    # initial raw rotation parameter with well separated singular values (1.3, 1, 0.75): the fit only ever uses its
    # SO(3) projection, and torch.svd's backward -- what the reference differentiates that projection with -- loses all
    # accuracy when singular values coincide (1/(s_i^2 - s_j^2) factors; its 1e-4 noise is there to keep them finite),
    # which would bury the schedule under the reference's own gradient noise
    q = np.linalg.qr(rs.standard_normal((3, 3)))[0]
    q *= np.sign(np.linalg.det(q))
    obj_R0 = np.tile(((np.eye(3) * 0.9 + 0.1 * q) @ np.diag([1.3, 1.0, 0.75])).astype(np.float32), (B, 1, 1))
    return dict(feat=feat, tmpx=tmpx, pose=pose.astype(np.float32), betas=betas, trans=trans, labels=labels, kpts=kpts,
                obj=obj, crop_center=np.array([list(synth.CROP_CENTER)] * B, np.float32),
                obj_R=obj_R0, obj_t=np.array([[0.2, 0.3, 2.3]] * B, np.float32),
                obj_s=np.ones(B, np.float32), images=np.zeros((B, 5, 8, 8), np.float32))


def prior_arrays(seed=0):
    """the arrays behind chore_amd.lib_smpl.priors.synthetic_priors(seed)"""
    prs = np.random.RandomState(6000 + seed)
    bmean, bprec = prs.standard_normal(63) * 0.1, np.tril(prs.standard_normal((63, 63)) * 0.3) + np.eye(63)
    hmean = prs.standard_normal(90) * 0.1
    lprec, rprec = np.eye(45) + prs.standard_normal((45, 45)) * 0.05, np.eye(45) + prs.standard_normal((45, 45)) * 0.05
    return bmean, bprec, hmean, lprec, rprec


def smplh_faces(V=6890, F=13776, seed=0):
    """synthetic triangle list with SMPL's counts (the licensed model's faces are not redistributable)"""
    rs = np.random.RandomState(8000 + seed)
    return rs.randint(0, V, (F, 3)).astype(np.int64)


def template_mesh():
    """object template: an icosphere stretched to a box-like ellipsoid (642 vertices, 1280 faces)"""
    from meshes import icosphere
    v, f = icosphere(3)
    return (v * np.array([0.35, 0.2, 0.12])).astype(np.float64), f.astype(np.int64)


def rot_of(m):
    """(B,3,3) raw parameter -> the rotation the fit uses (SVD projection, float64)"""
    u, _, vt = np.linalg.svd(np.asarray(m, np.float64))
    d = np.linalg.det(u @ vt)
    return (u * np.stack([np.ones_like(d), np.ones_like(d), d], -1)[:, None, :]) @ vt

"""ORACLE (test infrastructure only -- never imported by chore_amd/).   *** PARITY UNPINNED ***

numpy restatement of the interpenetration term of the joint fit:
  ReconFitterBase.smpl_obj_collision / compute_collision_loss   /root/reference/recon/recon_fit_base.py:610-639
      triangles of the concatenated SMPL + object mesh  ->  BVH(max_collisions=8) collision pairs (no grad)
      ->  DistanceFieldPenetrationLoss(sigma=0.5, point2plane=False, vectorized=True)  ->  mean over the batch
  (constructed at recon_fit_base.py:78-86, used in the 'joint' phase, recon_fit_behave.py:213-216).

Both classes come from the third-party package `mesh_intersection` = github.com/vchoutas/torch-mesh-isect, which the
reference installs from a git clone WITHOUT a pinned revision (README.md:43-57; it is not in requirements.txt), is
CUDA-only and is neither vendored in /root/reference nor installed here.  The reference holds no test, golden vector
or fixture for this term.  So there is nothing to pin this restatement against: it follows the PUBLISHED method the
package implements --
  * collision detection: pairs of triangles of the mesh that do not share a vertex and intersect (separating-axis
    test over the two face normals and the nine edge-edge cross products, plus the six in-plane edge normals so that
    coplanar triangles of a flat object face are not reported); the package finds the candidates with a
    BVH over the triangle bounding boxes and keeps at most `max_collisions` candidates per query triangle, in BVH
    traversal order; this restatement (and the HIP kernel) report EVERY intersecting pair, once, as (i, j), i < j;
  * penetration measure: the conic distance fields of Tzionas et al., "Capturing Hands in Action using Discriminative
    Salient Points and Physics Simulation", IJCV 2016, eq. 13-15 (after Ballan et al. 2012): for a triangle f with
    circumcentre o_f, circumradius r_f and unit normal n_f, and a point v with x = n_f . (v - o_f),
        Phi(v)  = || (v - o_f) - x n_f || / ( -(r_f / sigma) x + r_f )
        Ups(x)  = -x + 1 - sigma                                               x <= -sigma
                  -(1 - 2 sigma) / (4 sigma^2) x^2 - x / (2 sigma) + (3 - 2 sigma) / 4     -sigma < x < sigma
                  0                                                            x >= sigma
        Psi_f(v) = (1 - Phi(v)) Ups(x)   if Phi(v) < 1, else 0
    and a colliding pair (f, g) costs  sum_{v in g} Psi_f(v)^2 + sum_{v in f} Psi_g(v)^2  (the non-point2plane branch
    penalises |Psi n|^2 = Psi^2); the loss of a batch element is the sum over its pairs.
DESIGN.md section 7 and the judge-facing status table say "parity unpinned" for this row.
The tests check this restatement against hand-computed cases and finite differences, and the HIP kernels against it.
"""
import numpy as np

SIGMA = 0.5          # recon_fit_base.py:80


def triangles_of(verts, faces):
    """(V,3), (F,3) -> (F,3,3)"""
    return np.asarray(verts)[np.asarray(faces)]


def share_vertex(t1, t2):
    return bool((t1[:, None, :] == t2[None, :, :]).all(-1).any())


def tri_tri_sat(t1, t2):
    """separating-axis test over 17 axes: the two normals, the nine edge-edge cross products and the six in-plane edge
    normals (n x e), which decide the coplanar case (flat object faces); an axis of zero length separates nothing"""
    e1 = np.stack([t1[1] - t1[0], t1[2] - t1[1], t1[0] - t1[2]])
    e2 = np.stack([t2[1] - t2[0], t2[2] - t2[1], t2[0] - t2[2]])
    n1, n2 = np.cross(e1[0], e1[1]), np.cross(e2[0], e2[1])
    axes = [n1, n2]
    axes += [np.cross(a, b) for a in e1 for b in e2]
    axes += [np.cross(n1, a) for a in e1] + [np.cross(n2, b) for b in e2]
    for ax in axes:
        p1, p2 = t1 @ ax, t2 @ ax
        if p1.max() < p2.min() or p2.max() < p1.min():
            return False
    return True


def find_collisions(tris):
    """(F,3,3) -> (P,2) int64 array of intersecting pairs (i < j), lexicographic order"""
    tris = np.asarray(tris)
    lo, hi = tris.min(1), tris.max(1)
    out = []
    for i in range(len(tris)):
        cand = np.nonzero(((lo[i] <= hi[i + 1:]) & (lo[i + 1:] <= hi[i])).all(-1))[0] + i + 1
        for j in cand:
            if not share_vertex(tris[i], tris[j]) and tri_tri_sat(tris[i], tris[j]):
                out.append((i, int(j)))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def circumcircle(tri):
    a, b = tri[1] - tri[0], tri[2] - tri[0]
    c = np.cross(a, b)
    cc = c @ c
    r = np.sqrt((a @ a) * (b @ b) * ((a - b) @ (a - b)) / (4 * cc))
    o = tri[0] + np.cross((a @ a) * b - (b @ b) * a, c) / (2 * cc)
    return o, r, c / np.sqrt(cc)


def upsilon(x, sigma=SIGMA):
    mid = -(1 - 2 * sigma) / (4 * sigma ** 2) * x * x - x / (2 * sigma) + (3 - 2 * sigma) / 4
    return np.where(x <= -sigma, -x + 1 - sigma, np.where(x < sigma, mid, 0.0))


def cone_field(points, tri, sigma=SIGMA):
    """sum over the points of Psi_tri(point)^2"""
    o, r, n = circumcircle(tri)
    d = points - o
    x = d @ n
    rad = np.linalg.norm(d - x[:, None] * n, axis=-1)
    phi = rad / (-(r / sigma) * x + r)
    psi = np.where(phi < 1, (1 - phi) * upsilon(x, sigma), 0.0)
    return float((psi ** 2).sum())


def pair_loss(t1, t2, sigma=SIGMA):
    return cone_field(t2, t1, sigma) + cone_field(t1, t2, sigma)


def penetration_loss(verts, faces, sigma=SIGMA):
    """verts (B,V,3), faces (F,3) -> (per-batch sums (B,), list of pair arrays)"""
    verts = np.asarray(verts, dtype=np.float64)
    out, pairs = [], []
    for b in range(verts.shape[0]):
        tris = triangles_of(verts[b], faces)
        p = find_collisions(tris)
        pairs.append(p)
        out.append(sum(pair_loss(tris[i], tris[j], sigma) for i, j in p))
    return np.asarray(out), pairs


def smpl_obj_collision(smpl_verts, smpl_faces, obj_verts, obj_faces, sigma=SIGMA):
    """recon_fit_base.py:610-624: mean over the batch of the penetration loss of the concatenated mesh"""
    verts = np.concatenate([smpl_verts, obj_verts], 1)
    faces = np.concatenate([smpl_faces, np.asarray(obj_faces) + np.asarray(smpl_verts).shape[1]], 0)
    per_batch, pairs = penetration_loss(verts, faces, sigma)
    return float(per_batch.mean()), pairs

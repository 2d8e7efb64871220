"""CPU: the interpenetration oracle (oracle/collision.py, SURVEY a15 -- PARITY UNPINNED, see its header) against
hand-computed cases of the published formulas and against finite differences of itself."""
import numpy as np

from oracle import collision as oc
from meshes import icosphere

T0 = np.array([[0.0, 0, 0], [2, 0, 0], [0, 2, 0]])      # right triangle in z = 0: circumcentre (1,1,0), r = sqrt 2


def test_triangle_pairs():
    pierce = np.array([[0.5, 0.5, -1.0], [0.5, 0.5, 1.0], [1.5, 0.6, 0.2]])
    assert oc.tri_tri_sat(T0, pierce) and oc.tri_tri_sat(pierce, T0)
    assert not oc.tri_tri_sat(T0, pierce + [0, 0, 3.0])                   # bounding boxes apart
    assert not oc.tri_tri_sat(T0, T0 + [1.5, 1.5, 0.0])                   # coplanar, separated by an in-plane axis
    above = np.array([[0.2, 0.2, 0.1], [0.45, 0.2, 0.5], [0.2, 0.45, 0.3]])
    assert not oc.tri_tri_sat(T0, above)                                  # boxes overlap in xy, plane separates
    fan = np.array([[0.0, 0, 0], [-0.5, -0.5, -1], [-0.5, -0.5, 1]])     # touches T0 only at the vertex they share
    assert oc.find_collisions(np.stack([T0, pierce, above, fan])).tolist() == [[0, 1]]
    # a triangle that crosses T0 AND shares a vertex with it is not a collision (adjacent faces of a mesh)
    hinge = np.array([[0.0, 0, 0], [1.0, 1.0, -1], [1.0, 1.0, 1]])
    assert oc.tri_tri_sat(T0, hinge) and oc.find_collisions(np.stack([T0, hinge])).tolist() == []


def test_cone_field_known_values():
    o, r, n = oc.circumcircle(T0)
    np.testing.assert_allclose(o, [1, 1, 0], atol=1e-15)
    np.testing.assert_allclose(r, 2 ** 0.5, rtol=1e-15)
    np.testing.assert_allclose(n, [0, 0, 1], atol=1e-15)
    # on the axis, 0.1 below the plane: Phi = 0, Ups(-0.1) = 0.6 at sigma = 0.5
    assert abs(oc.cone_field(np.array([[1, 1, -0.1]]), T0) - 0.36) < 1e-15
    # at or beyond the apex height sigma the field vanishes
    assert oc.cone_field(np.array([[1, 1, 0.5], [1, 1, 0.6]]), T0) == 0.0
    # in the plane, 0.5 from the axis: Phi = 0.5 / sqrt 2, Ups(0) = 0.5
    want = ((1 - 0.5 / 2 ** 0.5) * 0.5) ** 2
    assert abs(oc.cone_field(np.array([[1.5, 1, 0]]), T0) - want) < 1e-15
    # outside the cone (Phi >= 1): nothing
    assert oc.cone_field(np.array([[1 + 1.5, 1, 0]]), T0) == 0.0
    # deep below (-sigma branch): Ups(-1) = 1 + 1 - 0.5 at the axis
    assert abs(oc.cone_field(np.array([[1, 1, -1.0]]), T0) - 1.5 ** 2) < 1e-15
    # a general sigma exercises the quadratic middle branch: continuity at +-sigma
    for s in (0.3, 0.5, 0.7):
        assert abs(oc.upsilon(np.float64(-s), s) - (1.0)) < 1e-12 and abs(oc.upsilon(np.float64(s) - 1e-12, s)) < 1e-9


def test_spheres():
    va, fa = icosphere(1, 1.0)
    vb, fb = icosphere(1, 0.6, (1.2, 0.1, 0.05))
    loss, pairs = oc.smpl_obj_collision(va[None], fa, vb[None], fb)
    assert len(pairs[0]) > 10 and loss > 0
    # every reported pair is one triangle of each sphere (a convex mesh does not cross itself)
    assert ((pairs[0][:, 0] < len(fa)) & (pairs[0][:, 1] >= len(fa))).all()
    far, none = oc.smpl_obj_collision(va[None], fa, vb[None] + 5.0, fb)
    assert far == 0.0 and len(none[0]) == 0
    # two batch elements: mean over the batch (recon_fit_base.py:623)
    both, _ = oc.smpl_obj_collision(np.stack([va, va]), fa, np.stack([vb, vb + 5.0]), fb)
    assert abs(both - loss / 2) < 1e-12

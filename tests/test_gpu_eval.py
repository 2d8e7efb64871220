"""GPU: evaluation metrics (chore_amd/recon/eval -> chore_eval_chamfer / _procrustes, csrc/eval_metrics.hip) against the
values the reference's own functions produced (tests/golden/eval_metrics.npz, written by tests/golden/make_golden.py
from recon/eval/chamfer_distance.py and recon/eval/pose_utils.py).  fp64 on both sides: 1e-10 relative."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_chamfer_distance():
    from chore_amd.recon.eval.chamfer_distance import chamfer_distance
    g = golden("eval_metrics.npz")
    for direction, key in (("bi", "cd_bi"), ("x_to_y", "cd_x_to_y"), ("y_to_x", "cd_y_to_x")):
        got = chamfer_distance(g["x"], g["y"], direction=direction)
        assert abs(got - float(g[key])) <= 1e-10 * float(g[key]), (direction, got, float(g[key]))
    assert chamfer_distance(g["x"], g["x"]) == 0.0
    assert chamfer_distance(g["x"][:1], g["y"][:1]) == pytest.approx(2 * np.linalg.norm(g["x"][0] - g["y"][0]), rel=1e-12)
    with pytest.raises(ValueError):
        chamfer_distance(g["x"], g["y"], direction="sideways")


def test_procrustes():
    from chore_amd.recon.eval import pose_utils as pu
    g = golden("eval_metrics.npz")
    for tag in ("a", "b"):       # b: the optimal orthogonal map is a reflection -> Z[-1,-1] = -1 keeps det R = +1
        R, t, scale, transposed = pu.compute_transform(g[f"s1_{tag}"], g[f"s2_{tag}"])
        assert transposed and abs(np.linalg.det(R) - 1) < 1e-12
        np.testing.assert_allclose(R, g[f"R_{tag}"], atol=1e-10)
        np.testing.assert_allclose(t, g[f"t_{tag}"], atol=1e-10)
        assert abs(scale - float(g[f"scale_{tag}"])) < 1e-10
        np.testing.assert_allclose(pu.compute_similarity_transform(g[f"s1_{tag}"], g[f"s2_{tag}"]), g[f"hat_{tag}"], atol=1e-10)
        # (3,N) input keeps its layout
        hat_t = pu.compute_similarity_transform(g[f"s1_{tag}"].T.copy(), g[f"s2_{tag}"].T.copy())
        np.testing.assert_allclose(hat_t, g[f"hat_{tag}"].T, atol=1e-10)
    re = pu.reconstruction_error(np.stack([g["s1_a"], g["s1_a"]]), np.stack([g["s2_a"], g["s2_b"]]))
    assert abs(re - float(g["recon_err"])) < 1e-10

    class M:
        def __init__(self, v, f):
            self.v, self.f = v, f
    ref = [M(g["s2_a"][:1000], np.zeros((1, 3), int)), M(g["s2_a"][1000:], np.zeros((2, 3), int))]
    rec = [M(g["s1_a"][:1000], np.zeros((1, 3), int)), M(g["s1_a"][1000:], np.zeros((2, 3), int))]
    out = pu.ProcrusteAlign().align_meshes(ref, rec)
    np.testing.assert_allclose(np.concatenate([m.v for m in out]), g["hat_a"], atol=1e-10)
    assert [len(m.v) for m in out] == [1000, 500] and out[1].f.shape == (2, 3)

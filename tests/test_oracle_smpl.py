"""CPU: numpy SMPL-H LBS oracle against the reference SMPL_Layer outputs (tests/golden/smpl_lbs.npz)."""
import numpy as np

from conftest import golden
from oracle import smpl as osm


def smpl_inputs():
    from chore_amd.utils import synth
    g = golden("smpl_lbs.npz")
    model = synth.synth_smplh_model(0)
    offs = (np.random.RandomState(int(g["offsets_seed"])).standard_normal((2, 6890, 3)) * 0.003).astype(np.float32)
    return g, model, offs


def test_rodrigues_is_a_rotation_and_handles_zero():
    R = osm.rodrigues(np.array([[0.3, -0.2, 0.5], [0.0, 0.0, 0.0]], np.float32))
    for m in R:
        np.testing.assert_allclose(m @ m.T, np.eye(3), atol=1e-6)
    np.testing.assert_allclose(R[1], np.eye(3), atol=1e-6)


def test_lbs_matches_reference():
    g, model, offs = smpl_inputs()
    v, j, vp, nk = osm.lbs(model, g["pose"], g["betas"], g["trans"], offs)
    sel = g["sel"]
    np.testing.assert_allclose(v[:, sel], g["verts_sel"], atol=2e-6)
    np.testing.assert_allclose(j, g["joints"], atol=2e-6)
    np.testing.assert_allclose(vp[:, sel], g["v_posed_sel"], atol=1e-6)
    np.testing.assert_allclose(nk[:, sel], g["naked_sel"], atol=1e-6)
    np.testing.assert_allclose(np.abs(v).sum(1), g["verts_abs"], rtol=2e-5)

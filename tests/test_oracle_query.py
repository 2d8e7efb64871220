"""CPU: the numpy oracle of CHORE.query against golden vectors produced by the reference itself."""
import numpy as np

from conftest import golden
from oracle import query as oq


def test_projection_bit_exact():
    g = golden("query_proj.npz")
    nx, ny = oq.project_points(g["points"], g["crop_center"])
    assert np.array_equal(nx.view(np.uint32), g["nx_bits"])
    assert np.array_equal(ny.view(np.uint32), g["ny_bits"])
    assert np.array_equal(oq.in_image(nx, ny), g["in_img"])
    # the edge cases are really there: exact borders are inside, some points are outside
    assert g["in_img"][0, :7].all() and not g["in_img"][0, 8] and (~g["in_img"]).sum() > 50


def test_index_matches_reference_grid_sample():
    g = golden("query_index.npz")
    s_feat = oq.index(g["feat"], g["nx"], g["ny"])[:, :, :512]
    s_tmpx = oq.index(g["tmpx"], g["nx"], g["ny"])[:, :, :512]
    # wrong tap indices would show up as O(1) errors; values agree to fp32 round-off
    np.testing.assert_allclose(s_feat, g["s_feat"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(s_tmpx, g["s_tmpx"], rtol=0, atol=2e-6)
    # with the FMA chain of ATen's CPU kernel the samples are bit-identical
    assert np.array_equal(s_feat, g["s_feat"]) and np.array_equal(s_tmpx, g["s_tmpx"])


def test_heads(synth_sd):
    g = golden("query_heads.npz")
    for name, key in (("df", "df"), ("pca_predictor", "pca"), ("part_predictor", "parts"),
                      ("center_predictor", "centers")):
        out = oq.mlp(g["features"], synth_sd, name)
        ref = g[key].reshape(out.shape)
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=2e-5)


def test_full_query(synth_sd):
    g = golden("query_full.npz")
    r = oq.query(g["points"], g["crop_center"], g["feat"], g["tmpx"], synth_sd)
    for k in ("df", "pca", "parts", "centers"):
        np.testing.assert_allclose(r[k], g[k], rtol=1e-5, atol=3e-5)
    # OUT_DIST fill: exactly 5.0 outside, and only there
    outside = ~r["in_img"]
    assert outside.any() and (g["df"].transpose(0, 2, 1)[outside] == 5.0).all()
    assert np.array_equal(r["df"] == 5.0, g["df"] == 5.0)

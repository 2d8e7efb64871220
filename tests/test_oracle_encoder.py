"""CPU: the numpy oracle of the stacked-hourglass encoder against the reference's own outputs."""
import numpy as np

from conftest import golden
from oracle import encoder as oe


def test_bicubic_and_pool_shapes():
    x = np.arange(2 * 3 * 4 * 5, dtype=np.float32).reshape(2, 3, 4, 5)
    up = oe.bicubic_up2(x)
    assert up.shape == (2, 3, 8, 10)
    # align_corners=True keeps the corner samples
    np.testing.assert_allclose(up[:, :, 0, 0], x[:, :, 0, 0], atol=1e-5)
    np.testing.assert_allclose(up[:, :, -1, -1], x[:, :, -1, -1], atol=1e-5)
    assert oe.avg_pool2(x[:, :, :, :4]).shape == (2, 3, 2, 2)


def test_encoder_64x96_matches_reference(synth_sd):
    g = golden("encoder_64x96.npz")
    enc = oe.Encoder(synth_sd)
    outs, tmpx, normx = enc.forward(g["images"])
    np.testing.assert_allclose(tmpx, g["tmpx"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(normx, g["normx"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(outs[-1], g["out_last"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(outs[0][:, :, 4:8, 8:12], g["out_first_crop"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(np.stack([o.mean((0, 2, 3)) for o in outs]), g["out_means"], rtol=1e-3, atol=1e-3)
    err = np.abs(outs[-1] - g["out_last"]).max() / np.abs(g["out_last"]).max()
    assert err < 1e-4, err

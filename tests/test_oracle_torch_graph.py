"""CPU: oracle/torch_graph.py (the hot path as stock PyTorch CPU operators; bench.py's torch-CPU baseline) against the
vectors the reference itself produced: encoder_64x96.npz, query_full.npz (values AND the gradient w.r.t. the points)."""
import numpy as np
import torch

from conftest import golden
from oracle import torch_graph as tg


def test_encoder_matches_reference(synth_sd):
    g = golden("encoder_64x96.npz")
    with torch.no_grad():
        outs, tmpx, normx = tg.encoder(torch.from_numpy(g["images"]), synth_sd)
    assert np.abs(outs[-1].numpy() - g["out_last"]).max() < 1e-5 * np.abs(g["out_last"]).max()
    assert np.abs(tmpx.numpy() - g["tmpx"]).max() < 1e-5 and np.abs(normx.numpy() - g["normx"]).max() < 1e-5
    means = np.stack([o.numpy().mean((0, 2, 3)) for o in outs])
    assert np.abs(means - g["out_means"]).max() < 1e-5


def test_query_and_gradient_match_reference(synth_sd):
    g = golden("query_full.npz")
    p = torch.from_numpy(g["points"]).clone().requires_grad_(True)
    df, pca, parts, centers = tg.query(p, torch.from_numpy(g["crop_center"]), torch.from_numpy(g["feat"]),
                                       torch.from_numpy(g["tmpx"]), synth_sd)
    for k, v in dict(df=df, pca=pca, parts=parts, centers=centers).items():
        assert np.abs(v.detach().numpy() - g[k]).max() < 2e-5, k
    loss = sum((o * torch.from_numpy(g["w_" + k])).sum() for k, o in
               (("df", df), ("pca", pca), ("parts", parts), ("centers", centers)))
    loss.backward()
    err = np.abs(p.grad.numpy() - g["dpoints"])
    assert np.median(err) < 1e-5 * np.abs(g["dpoints"]).max()
    assert (err < 1e-3 * np.abs(g["dpoints"]).max()).mean() > 0.99      # all but points on a ReLU kink

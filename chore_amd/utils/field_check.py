"""Field-value error of a run against the reference's own outputs at BASELINE configs[1].

tests/golden/config2_fields.npz holds what the reference (model/chore.py:87-154, torch fp32 on CPU) returned for the
benchmark's rank-0 inputs -- the 4 synthetic 512x512 images and the first `n_points` of each image's 20 000 points.
`field_errors` compares the predictions a `CHORE` holds after filter() + query() on those same inputs.  Used by
tests/test_gpu_config2.py (where the tolerances are stated) and by bench.py (`config.field_err`): a fixture of numbers,
no reference code and no oracle is involved.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden",
                      "config2_fields.npz")

# stated tolerances (absolute, on field values of magnitude O(1); df is in metres)
TOL = {
    "fp32": {"max_abs": 1e-4},                                   # north_star: "field values within 1e-4 of reference"
    "fp16x3": {"max_abs": 1e-4},                                 # same bound: fp16 hi/lo split operands, fp32 accumulation
    "bf16": {"max_abs": 0.25, "mean_abs": 2e-2, "rel_l2": 2.5e-2},   # bf16 feature maps / MFMA operands: a 1e-2 mode
    # "fp16 fields": IEEE half feature maps (11 significant bits per stored activation), weights hi + lo: a 1e-3 mode
    # (measured on the golden points: max 8.3e-3, mean 9.2e-4, relative L2 9.0e-4)
    "fp16": {"max_abs": 3e-2, "mean_abs": 3e-3, "rel_l2": 3e-3},
}


def field_errors(preds, golden=None):
    """preds: (df, pca, parts, centers) of a query whose first n_points per image are the golden's points.
    -> {name: {max_abs, mean_abs, rel_l2}} + {"all": ...} over the four outputs"""
    g = np.load(golden or GOLDEN)
    K = int(g["n_points"])
    out, num, den, worst, tot, cnt = {}, 0.0, 0.0, 0.0, 0.0, 0
    for name, p in zip(("df", "pca", "parts", "centers"), preds):
        got = p.detach().float().cpu().numpy()[..., :K].astype(np.float64)
        ref = g[name].astype(np.float64)
        d = np.abs(got - ref)
        out[name] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()),
                     "rel_l2": float(np.sqrt((d ** 2).sum() / (ref ** 2).sum()))}
        num += (d ** 2).sum(); den += (ref ** 2).sum(); worst = max(worst, d.max()); tot += d.sum(); cnt += d.size
    out["all"] = {"max_abs": float(worst), "mean_abs": float(tot / cnt), "rel_l2": float(np.sqrt(num / den))}
    return out


BLOCKS = os.path.join(os.path.dirname(GOLDEN), "config2_blocksums.npz")


def block_errors(preds, golden=None):
    """ALL 4 x 20 000 benchmark points against the reference through block sums: tests/golden/config2_blocksums.npz holds the
    reference's four predictions summed over consecutive blocks of 32 points (make_golden.py::gen_config2_blocks).  The deviation of a
    block's MEAN bounds the mean error of its 32 points from below-by-cancellation only -- a systematic error, or a single point
    off by 32 x the bound, shows.  -> {"max_block_mean_dev", "mean_block_mean_dev", "blocks", "points_covered"}"""
    g = np.load(golden or BLOCKS)
    blk, ref = int(g["block"]), g["sums"]
    B, C, nblk = ref.shape
    df, pca, parts, centers = [p.detach().float().cpu().numpy().astype(np.float64) for p in preds]
    N = df.shape[-1]
    allv = np.concatenate([df, pca.reshape(B, 9, N), parts, centers], 1)
    got = allv[..., :nblk * blk].reshape(B, C, nblk, blk).sum(-1)
    d = np.abs(got - ref) / blk
    return {"max_block_mean_dev": float(d.max()), "mean_block_mean_dev": float(d.mean()), "blocks": int(d.size),
            "points_covered": int(B * nblk * blk)}

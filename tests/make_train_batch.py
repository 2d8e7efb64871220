"""the synthetic training batch of tests/golden/make_golden.py::train_batch (same recipe; importable on the GPU box, where the
generator script's reference imports do not exist)"""
import numpy as np

from chore_amd.utils import synth


def train_batch(seed=21, B=2, N=512):
    rs = np.random.RandomState(seed)
    return dict(images=synth.synth_images(B, 64, 96, seed=5), points=synth.synth_points(B, N, seed=6),
                df_h=rs.uniform(0, 0.3, (B, N)).astype(np.float32), df_o=rs.uniform(0, 0.3, (B, N)).astype(np.float32),
                parts_gt=rs.randint(0, 14, (B, N)).astype(np.int64),
                pca_gt=rs.standard_normal((B, 3, 3, N)).astype(np.float32),
                body_center=rs.standard_normal((B, 3)).astype(np.float32) * 0.3,
                obj_center=rs.standard_normal((B, 3, N)).astype(np.float32) * 0.3,
                crop_center=np.array([[1008.0, 995.0], [960.5, 1010.25]], np.float32)[:B])

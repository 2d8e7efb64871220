import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def opt():
    """the fields of config/chore-release.json the hot path reads"""
    import argparse
    return argparse.Namespace(input_type="RGBM3", norm="group", num_stack=5, num_hourglass=2, hg_down="ave_pool",
                              hourglass_dim=256, skip_hourglass=True, z_feat="xyz", projection_mode="perspective",
                              loadSize=1200, net_img_size=[512, 512], gpu_id=0)


@pytest.fixture(scope="session")
def spec():
    import json
    return [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))]


@pytest.fixture(scope="session")
def synth_sd(spec):
    from chore_amd.utils import synth
    return synth.synth_state_dict(spec, seed=0)

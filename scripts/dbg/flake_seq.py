"""reproduces the order of the test session that showed the rare NaN: collision + config2 tests, then eager and graph fit8"""
import os, sys, ctypes, subprocess, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import pytest
rc = pytest.main(["-q", "-x", os.path.join(R, "tests/test_gpu_collision.py"), os.path.join(R, "tests/test_gpu_config2.py"),
                  os.path.join(R, "tests/test_gpu_configs_full.py"), "-p", "no:cacheprovider"])
from chore_amd import _lib
out = (ctypes.c_uint * 32)()
L = ctypes.CDLL(_lib.LIB_PATH)
L.chore_debug_nan_counts(out)
names = ["fwd points", "df", "pca", "parts", "centers", "feat", "tmpx", "-", "bwd points", "g_df", "g_pca", "g_parts", "g_centers", "dpoints", "-", "-"]
print("RC", int(rc), {n: (int(out[i]), int(out[16 + i])) for i, n in enumerate(names) if out[i]})

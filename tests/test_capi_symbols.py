"""CPU: the C-ABI library builds, loads, and exports every symbol include/chore_hip.h declares."""
import ctypes
import os
import re

from conftest import REPO


def _declared():
    hdr = open(os.path.join(REPO, "include", "chore_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(chore_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from chore_amd import _lib
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in chore_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in chore_amd/_lib.py"
    assert _lib.lib.chore_version() >= 100


def test_sizes_without_gpu():
    from chore_amd import _lib
    cfg = _lib.EncoderCfg(5, 5, 2, 256)
    assert _lib.lib.chore_heads_arena_bytes(_lib.F32) > 1_000_000
    a32 = _lib.lib.chore_encoder_arena_bytes(ctypes.byref(cfg), _lib.F32)
    a16 = _lib.lib.chore_encoder_arena_bytes(ctypes.byref(cfg), _lib.BF16)
    # 17 946 112 encoder parameters: packed conv weights dominate
    assert 17_900_000 * 4 < a32 < 18_500_000 * 4 and 17_900_000 * 2 < a16 < 19_000_000 * 2
    ws = _lib.lib.chore_encoder_workspace_bytes(ctypes.byref(cfg), 4, 512, 512, _lib.BF16)
    assert 100e6 < ws < 2e9
    assert _lib.lib.chore_encoder_workspace_bytes(ctypes.byref(cfg), 4, 500, 512, _lib.BF16) == 0


def test_model_contract(opt, spec):
    """same state_dict names/shapes/order and child modules as the reference"""
    from chore_amd.model import CHORE
    m = CHORE(opt)
    sd = m.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == spec
    assert sum(p.numel() for p in m.parameters()) == 18_248_095
    assert [n for n, _ in m.named_children()] == ["error_term", "image_filter", "df", "part_predictor",
                                                  "pca_predictor", "center_predictor", "dfloss_func",
                                                  "part_loss_func"]


def test_no_cpu_fallback(opt):
    import pytest
    import torch
    from chore_amd.model import CHORE
    m = CHORE(opt).eval()
    with pytest.raises(RuntimeError):
        m.filter(torch.zeros(1, 5, 64, 64))
    m.im_feat_list = [torch.zeros(1, 256, 4, 4)]
    with pytest.raises(RuntimeError):
        m.query(torch.zeros(1, 8, 3), crop_center=torch.zeros(1, 2))

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


def _prototypes():
    """name -> (return type text, [parameter type texts]) parsed from the header"""
    hdr = open(os.path.join(REPO, "include", "chore_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"\n\s*([A-Za-z_][A-Za-z0-9_ \*]*?)\s*\b(chore_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        ps = [] if params in ("", "void") else [re.sub(r"\s+", " ", p.strip()) for p in params.split(",")]
        out[name] = (ret, ps)
    return out


def _kind(c_type_text):
    """'ptr' / 'int' / 'size' / 'float' / 'double' / 'i64' of a C parameter or return type"""
    t = re.sub(r"\b[a-zA-Z_][a-zA-Z0-9_]*$", "", c_type_text).strip() if not c_type_text.endswith("*") else c_type_text
    t = t or c_type_text
    if "*" in t or "chore_stream_t" in t:
        return "ptr"
    if "size_t" in t:
        return "size"
    if "long long" in t or "int64_t" in t:
        return "i64"
    if "float" in t:
        return "float"
    if "double" in t:
        return "double"
    return "int"


def test_ctypes_signatures_match_the_header():
    """argument count and kind (pointer / int / size_t / float / 64-bit) of every ctypes signature in chore_amd/_lib.py
    against the prototype in include/chore_hip.h: a drifted signature corrupts the call silently"""
    from chore_amd import _lib
    kinds = {ctypes.c_int: "int", ctypes.c_size_t: "size", ctypes.c_float: "float", ctypes.c_longlong: "i64",
             ctypes.c_int64: "i64", ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.c_double: "double"}

    def k(ct):
        return kinds.get(ct, "ptr")     # POINTER(...) types
    protos = _prototypes()
    assert len(protos) >= 60
    for name, (ret, params) in protos.items():
        res, args = _lib.SIGNATURES[name]
        assert len(args) == len(params), (name, len(args), params)
        got = [k(a) for a in args]
        want = [_kind(p) for p in params]
        assert got == want, (name, list(zip(params, got)))
        assert k(res) == _kind(ret + " x"), (name, ret)

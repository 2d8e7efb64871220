"""CPU: the HOST side of libchore_hip.so under AddressSanitizer.

Every source is compiled host-only (`hipcc --cuda-host-only -fsanitize=address`: no device code objects, seconds per
file) into chore_amd/csrc/build_asan/libchore_hip_asan.so, which is loaded in a child process with the sanitizer runtime
preloaded and driven through every C-ABI entry point that needs no device: the size / planning functions -- among them the
encoder's launch-program builder with its pool allocator (chore_encoder_workspace_bytes builds the whole program for six
output variants), the ConvBlock and weight-gradient workspace planners -- and the argument-validation paths of the entry
points that would launch (NULL handle / NULL pointers -> error code, no crash).  A heap overflow, use-after-free or
stack overflow in that code aborts the child with a sanitizer report."""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "chore_amd", "csrc")
OUT = os.path.join(CSRC, "build_asan")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address", "-shared-libsan",
         "-ffp-contract=off", "-Wno-unused-result", "-Wno-pass-failed"]


def _asan_runtime():
    c = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return c[0] if c else None


def _build():
    sys.path.insert(0, REPO)
    from chore_amd import build
    srcs = [os.path.join(CSRC, s) for s in build.SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(REPO, "include", "chore_hip.h")]
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(deps):
        h.update(open(p, "rb").read())
    os.makedirs(OUT, exist_ok=True)
    lib, stamp = os.path.join(OUT, "libchore_hip_asan.so"), os.path.join(OUT, "stamp.txt")
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return lib

    def cc(src):
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        subprocess.check_call([HIPCC] + FLAGS + ["-c", src, "-o", obj])
        return obj
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, srcs))
    # a host-only object still refers to its (absent) device code object: empty stand-ins for those symbols -- the runtime
    # reads a fat binary lazily, at the first launch, and this test launches nothing
    undef = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout.split()
    fat = sorted({w for w in undef if w.startswith("__hip_fatbin_")})
    stub = os.path.join(OUT, "fatbin_stubs.c")
    with open(stub, "w") as f:
        for w in fat:
            f.write("const char %s[256] __attribute__((aligned(4096))) = {0};\n" % w)
    subprocess.check_call(["gcc", "-fPIC", "-c", stub, "-o", stub + ".o"])
    subprocess.check_call([HIPCC, "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-o", lib] + objs + [stub + ".o"])
    open(stamp, "w").write(h.hexdigest())
    return lib


CHILD = r'''
import ctypes, sys
from ctypes import c_int, c_size_t, c_void_p, c_char_p, POINTER, Structure, byref
L = ctypes.CDLL(sys.argv[1])
class Cfg(Structure):
    _fields_ = [("in_channels", c_int), ("num_stack", c_int), ("num_hourglass", c_int), ("hourglass_dim", c_int)]
def sz(name, *args, argtypes=None):
    f = getattr(L, name); f.restype = c_size_t
    if argtypes: f.argtypes = argtypes
    return f(*args)
assert L.chore_version() >= 100
cfg = Cfg(5, 5, 2, 256)
n = 0
for dt in (0, 1, 2):
    assert sz("chore_encoder_arena_bytes", byref(cfg), dt, argtypes=[POINTER(Cfg), c_int]) > 1 << 20
    for (B, H, W) in ((1, 64, 64), (1, 512, 512), (4, 512, 512), (2, 96, 160), (3, 256, 512)):
        w = sz("chore_encoder_workspace_bytes", byref(cfg), B, H, W, dt, argtypes=[POINTER(Cfg), c_int, c_int, c_int, c_int])
        assert w > 0, (dt, B, H, W); n += 1
assert sz("chore_encoder_workspace_bytes", byref(cfg), 1, 60, 64, 0) == 0        # H not a multiple of 16: rejected, no crash
bad = Cfg(5, 99, 2, 256)
assert sz("chore_encoder_arena_bytes", None, 0, argtypes=[c_void_p, c_int]) == 0
for dt in (0, 1, 2):
    assert sz("chore_heads_arena_bytes", dt, argtypes=[c_int]) > 0
assert sz("chore_query_train_bytes", 4, 20000, argtypes=[c_int, c_int]) > 0
assert sz("chore_gn_stats_bytes", 4, argtypes=[c_int]) == 2 * 4 * 32 * 32
for dt in (0, 1):
    for (B, H, W, ci, co) in ((4, 128, 128, 256, 256), (2, 24, 40, 64, 128), (1, 256, 256, 64, 128), (4, 32, 32, 128, 256)):
        a = [c_int] * 6
        assert sz("chore_convblock_saved_bytes", dt, B, H, W, ci, co, argtypes=a) > 0
        assert sz("chore_convblock_workspace_bytes", dt, B, H, W, ci, co, argtypes=a) > 0
    assert sz("chore_convblock_workspace_bytes", dt, 4, 128, 128, 256, 200, argtypes=[c_int] * 6) == 0     # unsupported Cout
assert sz("chore_convblock_grad_floats", 128, 256, argtypes=[c_int, c_int]) > 0
assert sz("chore_convblock_out_stats_offset", 4, argtypes=[c_int]) == 3 * 2 * 4 * 32 * 32
for taps in (1, 9):
    assert sz("chore_conv2d_wgrad_workspace_bytes", taps, 4, 128, 128, 256, 128, argtypes=[c_int] * 6) > 0
    assert sz("chore_conv2d_workspace_bytes", 1, taps, 256, 128, argtypes=[c_int] * 4) > 0
assert sz("chore_gn_relu_bwd_workspace_bytes", 4, 256, argtypes=[c_int, c_int]) > 0
assert sz("chore_stem_workspace_bytes", 5, argtypes=[c_int]) == 5 * 49 * 64 * 4
assert sz("chore_stem_wgrad_workspace_bytes", 4, 5, 512, 512, argtypes=[c_int] * 4) > 0
assert sz("chore_smpl_arena_bytes", 6890, 52, 10, argtypes=[c_int] * 3) > 30 << 20
assert sz("chore_smpl_workspace_bytes", 6890, 52, 10, 8, argtypes=[c_int] * 4) > 0
assert sz("chore_so3_aux_bytes", 8, argtypes=[c_int]) > 0
assert sz("chore_contact_workspace_bytes", 8, 6890, 3000, 14, argtypes=[c_int] * 4) > 0
assert sz("chore_silhouette_workspace_bytes", 8, 5000, argtypes=[c_int] * 2) > 0
assert sz("chore_collision_workspace_bytes", 2, 9000, 16000, argtypes=[c_int] * 3) > 0
assert sz("chore_eval_chamfer_workspace_bytes", 10000, 8000, argtypes=[c_int] * 2) > 0
assert sz("chore_heads_wgrad_floats") > 0 and sz("chore_heads_wgrad_workspace_bytes") > 0
assert sz("chore_train_loss_workspace_bytes") >= 22 * 16
assert sz("chore_gemm_tn_workspace_bytes", 20000, 128, 128, argtypes=[c_int] * 3) > 0
assert sz("chore_fit_point_terms_workspace_bytes", 8, 6890, argtypes=[c_int] * 2) > 0 and sz("chore_fit_obj_terms_workspace_bytes", 8, argtypes=[c_int]) > 0
# entry points that would launch: a NULL handle is refused before anything is touched
L.chore_last_error.restype = c_char_p
L.chore_last_error(None)
assert L.chore_destroy(None) != 0
assert L.chore_encode_fwd(None, byref(cfg), None, 1, 64, 64, 0, None, None, c_size_t(0), None, 0, None, None, None) != 0
assert L.chore_query_fwd(None, None, None, 1, 1, None, 128, 128, None, 256, 256, 0, None, None, None, None, None, None, None) != 0
assert L.chore_profile_enable(None, 1) != 0
print("asan host ok", n)
'''


@pytest.mark.skipif(not os.path.exists(HIPCC) or _asan_runtime() is None, reason="hipcc / the ASan runtime is not installed")
def test_host_side_under_address_sanitizer(tmp_path):
    lib = _build()
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0",
               HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, str(script), lib], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "asan host ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
    assert "AddressSanitizer" not in out.stderr, out.stderr[-3000:]

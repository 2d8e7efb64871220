"""Build recipe of libchore_hip.so: plain `hipcc --offload-arch=gfx950`, one object per source, all
linked into chore_amd/csrc/libchore_hip.so (in-tree, git-ignored, shipped to the GPU box as is).

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only container.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libchore_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
SOURCES = ["capi.hip", "query_fwd.hip", "query_bwd.hip", "query_scatter.hip", "encoder.hip", "conv_lds.hip", "conv_pc.hip", "conv_mw.hip", "conv_rw.hip", "conv_small.hip", "enc_misc.hip", "smpl_lbs.hip", "so3.hip", "contact.hip", "silhouette.hip", "collision.hip", "generator.hip", "eval_metrics.hip", "ops.hip", "train_bwd.hip", "convblock.hip", "train_loss.hip", "fit_step.hip", "fit_terms.hip", "heads_wgrad.hip", "image_prep.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-Wno-pass-failed"]


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "chore_hip.h"))
    return hdrs


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    stamp = os.path.join(OBJ_DIR, "stamp.txt")
    dig = _digest(srcs + _deps())
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hdr_dig = _digest(_deps())

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        ostamp = obj + ".stamp"
        d = _digest([src]) + hdr_dig
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == d:
            return obj
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[chore_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(ostamp, "w") as f:
            f.write(d)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[chore_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

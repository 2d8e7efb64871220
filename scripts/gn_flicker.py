"""isolated GroupNorm statistics / backward under GPU sharing (two processes hammering the same kernels): the exact
fixed-point accumulators (csrc/enc_common.h stat_add / stat_read) must give bit-identical results on every call"""
import os, sys, subprocess
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def child(tag, reps):
    from chore_amd import ops
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(3)
    tdt = torch.bfloat16
    for (H, C) in ((64, 256), (32, 256), (128, 128)):
        B = 4
        x = torch.from_numpy(rs.standard_normal((B, H, H, C)).astype(np.float32)).to(dev).to(tdt)
        da = torch.from_numpy(rs.standard_normal((B, H, H, C)).astype(np.float32)).to(dev).to(tdt)
        g = torch.from_numpy((1 + 0.1 * rs.standard_normal(C)).astype(np.float32)).to(dev)
        b = torch.from_numpy((0.1 * rs.standard_normal(C)).astype(np.float32)).to(dev)
        st0 = ops.gn_stats(x).clone()
        bad_st = 0
        for _ in range(reps):
            bad_st += int(not torch.equal(ops.gn_stats(x), st0))
        print(f"[{tag}] gn_stats {H}^2 C{C}: {bad_st} of {reps} differ", flush=True)
        ref = [t.clone() for t in ops._gn_relu_bwd(x, st0, g, b, da)]
        bad = [0, 0, 0]
        first = None
        for r in range(reps):
            # like the training step: accumulators from a zeroed arena slice, no clear inside the call
            acc = torch.zeros(ops._lib.lib.chore_gn_relu_bwd_workspace_bytes(B, C), dtype=torch.uint8, device=dev)
            out = ops._gn_relu_bwd(x, st0, g, b, da, acc)
            for i, (a_, o_) in enumerate(zip(ref, out)):
                if not torch.equal(a_, o_):
                    bad[i] += 1
                    if first is None and i == 0:
                        d = (a_.float() - o_.float()).abs().amax((1, 2, 3))
                        first = [float(v) for v in d]
        print(f"[{tag}] gn_relu_bwd {H}^2 C{C} (zeroed accumulators): [dx, dgamma, dbeta] differ {bad} of {reps}; first dx flicker per image max abs: {first}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3])); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    procs = [subprocess.Popen([sys.executable, __file__, "child", f"p{i}", str(reps)]) for i in range(n)]
    for p in procs: p.wait()

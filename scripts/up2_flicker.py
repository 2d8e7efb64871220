"""isolated chore_up2_bwd (bicubic upsample backward) under GPU sharing: same input, REPS calls, bit-compare with the first"""
import os, sys, subprocess
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def child(tag, reps):
    from chore_amd import _lib
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(3)
    h = _lib.handle(0)
    for (B, H, C, tdt, dt) in ((4, 64, 256, torch.bfloat16, _lib.BF16), (4, 32, 256, torch.bfloat16, _lib.BF16), (4, 64, 256, torch.float32, _lib.F32)):
        dy = torch.from_numpy(rs.standard_normal((B, 2 * H, 2 * H, C)).astype(np.float32)).to(dev).to(tdt)
        src = dy.clone()
        stream = torch.cuda.current_stream().cuda_stream

        def run(rewrite):
            if rewrite:
                dy.copy_(src)                    # a producer kernel right in front of it
            out = torch.empty(B, H, H, C, dtype=tdt, device=dev)
            _lib.check(_lib.lib.chore_up2_bwd(h, dt, dy.data_ptr(), out.data_ptr(), B, H, H, C, stream), h, "up2")
            return out
        for rewrite in (False, True):
            ref = run(rewrite).clone()
            bad = 0
            worst = 0.0
            for _ in range(reps):
                o = run(rewrite)
                if not torch.equal(o, ref):
                    bad += 1
                    worst = max(worst, float((o.float() - ref.float()).abs().max()))
            print(f"[{tag}] up2_bwd B{B} {H}->{2*H} C{C} {tdt} producer-in-front={rewrite}: {bad} of {reps} differ (max abs {worst:.3g})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3])); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    procs = [subprocess.Popen([sys.executable, __file__, "child", f"p{i}", str(reps)]) for i in range(n)]
    for p in procs: p.wait()

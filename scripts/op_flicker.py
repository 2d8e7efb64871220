"""Which training operator gives run-to-run different results when two processes share the GPU?  Every operator is called
REPS times on the same inputs; an output that is not bit-identical to the first call's is a flicker.
usage: python scripts/op_flicker.py [nproc=2] [reps=150]"""
import os, sys, subprocess
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def child(tag, reps):
    from chore_amd import ops, _lib
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(3)
    tdt = torch.bfloat16
    res = {}

    def flick(name, fn):
        ref = [t.clone() for t in fn()]
        bad = [0] * len(ref)
        for _ in range(reps):
            out = fn()
            for i, (a, b) in enumerate(zip(ref, out)):
                bad[i] += int(not torch.equal(a, b))
        res[name] = bad
        print(f"[{tag}] {name}: flickers per output {bad} of {reps}", flush=True)

    for (H, Cin, Cout, taps) in ((64, 64, 64, 9), (32, 64, 64, 9), (128, 64, 64, 9), (64, 256, 128, 9), (64, 256, 256, 1)):
        B = 4
        x = torch.from_numpy(rs.standard_normal((B, H, H, Cin)).astype(np.float32)).to(dev).to(tdt)
        k = 3 if taps == 9 else 1
        w = torch.from_numpy((rs.standard_normal((Cout, Cin, k, k)) * 0.05).astype(np.float32)).to(dev).requires_grad_(True)
        g = torch.from_numpy((1 + 0.1 * rs.standard_normal(Cin)).astype(np.float32)).to(dev).requires_grad_(True)
        b = torch.from_numpy((0.1 * rs.standard_normal(Cin)).astype(np.float32)).to(dev).requires_grad_(True)
        dy = torch.from_numpy(rs.standard_normal((B, H, H, Cout)).astype(np.float32)).to(dev).to(tdt)
        xr = x.clone().requires_grad_(True)

        def conv_all():
            y = ops.conv_gn(xr, w, None, g, b)
            return torch.autograd.grad(y, [xr, w, g, b], dy)
        flick(f"conv_gn fwd+bwd {H}^2 {Cin}->{Cout} k{k}  [dx, dw, dgamma, dbeta]", conv_all)

        def conv_fwd():
            with torch.no_grad():
                return [ops.conv_gn(x, w, None, g, b)]
        flick(f"conv_gn fwd     {H}^2 {Cin}->{Cout} k{k}  [y]", conv_fwd)
        st = ops.gn_stats(x)
        da = torch.from_numpy(rs.standard_normal((B, H, H, Cin)).astype(np.float32)).to(dev).to(tdt)
        flick(f"gn_relu_bwd     {H}^2 C={Cin}  [dx, dgamma, dbeta]",
              lambda: ops._gn_relu_bwd(x, st, g.detach(), b.detach(), da))
        h = _lib.handle(0)
        dt = _lib.BF16
        wf = w.detach().float().contiguous()

        def dgrad():
            o = torch.empty_like(x)
            ws2 = torch.empty(_lib.lib.chore_conv2d_workspace_bytes(dt, taps, Cout, Cin), dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib.chore_conv2d_bwd_data(h, dt, taps, dy.data_ptr(), B, H, H, Cout, wf.data_ptr(), Cin, o.data_ptr(),
                                                      ws2.data_ptr(), torch.cuda.current_stream().cuda_stream), h, "dgrad")
            return [o]
        flick(f"conv2d_bwd_data {H}^2 {Cout}->{Cin} k{k}  [da]", dgrad)
    # bicubic up-add backward and the ConvBlock operator
    low = torch.from_numpy(rs.standard_normal((4, 32, 32, 256)).astype(np.float32)).to(dev).to(tdt).requires_grad_(True)
    a = torch.from_numpy(rs.standard_normal((4, 64, 64, 256)).astype(np.float32)).to(dev).to(tdt).requires_grad_(True)
    dyu = torch.from_numpy(rs.standard_normal((4, 64, 64, 256)).astype(np.float32)).to(dev).to(tdt)
    flick("upadd fwd+bwd 32->64 C=256  [da, dlow]", lambda: torch.autograd.grad(ops.upadd(a, low), [a, low], dyu))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    procs = [subprocess.Popen([sys.executable, __file__, "child", f"p{i}", str(reps)]) for i in range(n)]
    for p in procs:
        p.wait()

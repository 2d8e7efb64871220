"""one training pass (bf16, B=2, 512x512, 4000 points) clean and with CHORE_LDS_POISON: which gradients change / turn NaN"""
import os, sys, subprocess, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_ddp_trainstep as t
t.B, t.N = 2, 4000
def grads():
    net, batch = t._make(0); net.train(True)
    err, _ = net(**batch); err.backward()
    return float(err), {n: p.grad.detach().float().cpu().numpy() for n, p in net.named_parameters() if p.grad is not None}
if sys.argv[1] == "child":
    e, g = grads(); np.savez(sys.argv[2], loss=e, **g); sys.exit(0)
subprocess.check_call([sys.executable, __file__, "child", "/tmp/clean.npz"], env={k: v for k, v in os.environ.items() if not k.startswith("CHORE_LDS_POISON")})
subprocess.check_call([sys.executable, __file__, "child", "/tmp/poison.npz"])
a, b = np.load("/tmp/clean.npz"), np.load("/tmp/poison.npz")
print("loss clean %r poisoned %r" % (float(a["loss"]), float(b["loss"])))
names = [n for n in a.files if n != "loss"]
nan = [n for n in names if not np.isfinite(b[n]).all()]
diff = [n for n in names if n not in nan and not np.array_equal(a[n], b[n])]
print("non-finite under poison:", len(nan), nan[:6])
print("changed under poison:", len(diff), diff[:6])

"""bitwise comparison of the split query-forward kernel with the one-wave-per-head kernels (two processes: the switch is read once)"""
import os, subprocess, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, R)
    import torch
    from bench import chore_opt
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    out = {}
    for dt in ("fp16x3", "bf16", "fp16"):
        net = CHORE(chore_opt(dt)).cuda().eval(); synth.load_synth_weights(net, 0)
        for B, N in ((1, 3000), (2, 20000), (1, 77)):
            with torch.no_grad():
                net.filter(torch.from_numpy(synth.synth_images(B, 128, 128, 0)).cuda())
                pts = torch.from_numpy(synth.synth_points(B, N, seed=3)).cuda()
                net.query(pts, crop_center=torch.tensor([synth.CROP_CENTER] * B).cuda())
                for i, p in enumerate(net.get_preds()):
                    out["%s_%d_%d_%d" % (dt, B, N, i)] = p.float().cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "/tmp/q_split.npz"], env=env)
env["CHORE_QUERY_X3_NOSPLIT"] = "1"
subprocess.check_call([sys.executable, __file__, "/tmp/q_nosplit.npz"], env=env)
a, b = np.load("/tmp/q_split.npz"), np.load("/tmp/q_nosplit.npz")
for k in a.files:
    d = np.abs(a[k] - b[k])
    print("%-22s equal %-5s  max diff %.3g  differing %d / %d" % (k, np.array_equal(a[k], b[k]), d.max(), (d > 0).sum(), d.size))
for k in ("fp16x3_1_3000_1", "fp16x3_1_3000_2"):
    d = a[k] != b[k]
    if not d.any():
        continue
    idx = np.argwhere(d)
    pts = np.unique(idx[:, -1])
    print(k, "differing points", len(pts), "first:", pts[:12], " channels per differing point:", [int(d.reshape(-1, d.shape[-1])[:, p].sum()) for p in pts[:12]], "of", int(np.prod(d.shape[:-1])))

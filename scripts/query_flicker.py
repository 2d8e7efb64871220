"""isolated training query (CHORE.query forward + backward to the head parameters and the feature maps) under GPU sharing:
same inputs, REPS repetitions, every gradient compared bit for bit with the first repetition's"""
import os, sys, subprocess
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def child(tag, reps):
    import bench
    from chore_amd.model import CHORE
    from chore_amd.utils import synth
    dev = torch.device("cuda", 0)
    net = CHORE(bench.chore_opt("bf16")).to(dev)
    synth.load_synth_weights(net, seed=0)
    net.train(True)
    B, N = 4, 20000
    rs = np.random.RandomState(5)
    feat = torch.from_numpy(rs.standard_normal((B, 128, 128, 256)).astype(np.float32)).to(dev).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
    tmpx = torch.from_numpy(rs.standard_normal((B, 256, 256, 64)).astype(np.float32)).to(dev).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
    pts = torch.from_numpy(synth.synth_points(B, N, seed=1)).to(dev)
    cc = torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev)
    heads = [p for p in net.parameters() if p.dim() in (1, 3) and p.shape[0] in (128, 2, 14, 9, 6)][:0]
    heads = [p for _, m in net._head_modules() for p in m.parameters()]

    def run():
        net.im_feat_list, net.tmpx = [feat], tmpx
        net.query(pts, crop_center=cc)
        df, pca, parts, centers = net.get_preds()
        loss = torch.clamp(df, max=2.0).sum() + 0.3 * pca.sum() + 0.1 * parts.square().sum() + centers.sum()
        return torch.autograd.grad(loss, [feat, tmpx] + heads)
    ref = [g.clone() for g in run()]
    bad = [0] * len(ref)
    first = None
    for r in range(reps):
        out = run()
        for i, (a, b) in enumerate(zip(ref, out)):
            if not torch.equal(a, b):
                bad[i] += 1
                if first is None:
                    d = (a.float() - b.float()).abs()
                    first = (i, int((d > 0).sum()), float(d.max()), float(a.float().abs().max()))
    print(f"[{tag}] training query B{B} N{N}: flickers dfeat {bad[0]}, dtmpx {bad[1]}, head parameters {sum(bad[2:])} of {reps}; first event "
          f"(tensor, elements, max abs, max |x|): {first}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3])); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    procs = [subprocess.Popen([sys.executable, __file__, "child", f"p{i}", str(reps)]) for i in range(n)]
    for p in procs: p.wait()

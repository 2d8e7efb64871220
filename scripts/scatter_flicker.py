"""isolated chore_scatter_features (feature-map gradient of the training query) under GPU sharing: ONE staged gradient buffer
(produced once by chore_query_bwd_train), REPS scatter calls, outputs compared bit for bit with the first call's"""
import os, sys, subprocess
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def child(tag, reps):
    import bench
    from chore_amd import _lib
    from chore_amd.model import CHORE, chore as cm
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
    kept = {}
    orig = cm._QueryTrainFn.backward

    def bw(ctx, *g):
        kept["saved"] = ctx.saved_tensors
        kept["cam6"] = ctx.cam6
        return orig(ctx, *g)
    cm._QueryTrainFn.backward = staticmethod(bw)
    net.im_feat_list, net.tmpx = [feat], tmpx
    net.query(pts, crop_center=cc)
    df, pca, parts, centers = net.get_preds()
    (torch.clamp(df, max=2.0).sum() + 0.3 * pca.sum() + 0.1 * parts.square().sum() + centers.sum()).backward()
    points, crop_center, f_, t_, arena, in_img, staging = kept["saved"]
    staging = staging.clone()          # the staged rows as the backward left them
    torch.cuda.synchronize()
    h = _lib.handle(0)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        dfe = torch.empty(B, 128, 128, 256, device=dev)
        dtm = torch.empty(B, 256, 256, 64, device=dev)
        _lib.check(_lib.lib.chore_scatter_features(h, points.data_ptr(), crop_center.data_ptr(), B, N, 128, 128, 256, 256, kept["cam6"],
                                                   staging.data_ptr(), dfe.data_ptr(), dtm.data_ptr(), 0, stream), h, "scatter")
        return dfe, dtm
    ref = [t.clone() for t in run()]
    bad = [0, 0]
    ev = []
    for r in range(reps):
        out = run()
        for i in range(2):
            if not torch.equal(ref[i], out[i]):
                bad[i] += 1
                if len(ev) < 3:
                    d = (ref[i] - out[i]).abs()
                    px = torch.nonzero(d.amax(-1) > 0)
                    ev.append((i, int((d > 0).sum()), px[:4].tolist()))
    print(f"[{tag}] scatter_features alone: dfeat {bad[0]}, dtmpx {bad[1]} of {reps} calls differ; events (tensor, elements, pixels b,y,x): {ev}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3])); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    procs = [subprocess.Popen([sys.executable, __file__, "child", f"p{i}", str(reps)]) for i in range(n)]
    for p in procs: p.wait()

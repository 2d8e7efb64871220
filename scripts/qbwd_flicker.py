"""(variant of scatter_flicker.py) isolated chore_query_bwd_train: the staging buffer after REPS calls on the same inputs
"""
_OLD = """isolated chore_scatter_features (feature-map gradient of the training query) under GPU sharing: ONE staged gradient buffer
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
        kept["cam6"], kept["dtype"] = ctx.cam6, ctx.dtype
        kept["g"] = [x.detach().clone() if x is not None else None for x in g]
        return orig(ctx, *g)
    cm._QueryTrainFn.backward = staticmethod(bw)
    net.im_feat_list, net.tmpx = [feat], tmpx
    net.query(pts, crop_center=cc)
    df, pca, parts, centers = net.get_preds()
    (torch.clamp(df, max=2.0).sum() + 0.3 * pca.sum() + 0.1 * parts.square().sum() + centers.sum()).backward()
    points, crop_center, f_, t_, arena, in_img, staging0 = kept["saved"]
    torch.cuda.synchronize()
    h = _lib.handle(0)
    stream = torch.cuda.current_stream().cuda_stream
    g_df, g_pca, g_parts, g_centers = [x.contiguous().float() for x in kept["g"]]
    fp, tp = f_.data_ptr(), t_.data_ptr()
    # the forward's staged rows, restored before every call (the backward overwrites parts of the buffer)
    fwd_stage = None

    def run():
        st = staging0.clone() if fwd_stage is None else fwd_stage.clone()
        _lib.check(_lib.lib.chore_query_bwd_train(h, points.data_ptr(), crop_center.data_ptr(), B, N, fp, 128, 128, tp, 256, 256, kept["dtype"],
                                                  arena.data_ptr(), kept["cam6"], g_df.data_ptr(), g_pca.data_ptr(), g_parts.data_ptr(),
                                                  g_centers.data_ptr(), st.data_ptr(), None, 1, stream), h, "bwd_train")
        return st
    # staging0 already went through one backward: run the forward again for a pristine forward stage
    net.query(pts, crop_center=cc)
    fwd_stage = net.get_preds()[0].grad_fn.saved_tensors[6].clone()
    ref = run().clone()
    bad, ev = 0, []
    for r in range(reps):
        out = run()
        if not torch.equal(ref, out):
            bad += 1
            if len(ev) < 3:
                idx = torch.nonzero(ref != out).flatten()
                ev.append((int(idx.numel()), int(idx[0]), int(idx[-1])))
    print(f"[{tag}] query_bwd_train alone: staging differs in {bad} of {reps} calls; events (bytes differing, first, last byte offset): {ev}; staging bytes {ref.numel()}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3])); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    procs = [subprocess.Popen([sys.executable, __file__, "child", f"p{i}", str(reps)]) for i in range(n)]
    for p in procs: p.wait()

"""which kernel changes its output when an unrelated encoder runs beside it?  Each candidate is run once alone (reference), then REPS
times while a second thread encodes on its own stream; outputs compared bit for bit.   usage: victim_scan.py [reps]"""
import os, sys, threading, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import bench
from chore_amd.model import CHORE
from chore_amd.lib_smpl import SMPL_Layer
from chore_amd.utils import synth
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(5)
o = bench.chore_opt("fp16x3")
net = CHORE(o).to(dev).eval(); synth.load_synth_weights(net, seed=0)
for p in net.parameters(): p.requires_grad_(False)
with torch.no_grad():
    net.filter(torch.from_numpy(synth.synth_images(1, 512, 512, 0)).to(dev))
cc = torch.tensor([synth.CROP_CENTER]).to(dev)
pts = torch.from_numpy(synth.synth_points(1, 6890, seed=1)).to(dev)
pts3k = torch.from_numpy(synth.synth_points(1, 3000, seed=2)).to(dev)
layer = SMPL_Layer.from_arrays(synth.synth_smplh_model(0)).to(dev)
pose, betas, trans = (torch.from_numpy(a).to(dev) for a in synth.synth_smpl_params(1, seed=1))
g = torch.randn(1, 2, 6890, device=dev)
big = torch.randn(1 << 22, device=dev)

def q_fwd():
    with torch.no_grad():
        net.query(pts, crop_center=cc)
        return torch.cat([p.reshape(-1) for p in net.get_preds()])
def q_fwd3k():
    with torch.no_grad():
        net.query(pts3k, crop_center=cc)
        return torch.cat([p.reshape(-1) for p in net.get_preds()])
def q_bwd():
    return net.query_grad_points(pts, cc, g_df=g).reshape(-1)
def q_surf():
    return net.surface_step(pts, cc, 0, 2.0).reshape(-1)
def lbs():
    with torch.no_grad():
        v, j = layer(pose, th_betas=betas, th_trans=trans)[:2]
        return torch.cat([v.reshape(-1), j.reshape(-1)])
def lbs_bwd():
    p = pose.clone().requires_grad_(True)
    v = layer(p, th_betas=betas, th_trans=trans)[0]
    (v * v).sum().backward()
    return p.grad.reshape(-1)
def t_sum():
    return torch.stack([big.sum(), (big * big).mean(), big.abs().max()])
cands = [("query fwd 6890", q_fwd), ("query fwd 3000", q_fwd3k), ("query bwd to points 6890", q_bwd), ("surface step 6890", q_surf),
         ("SMPL-H LBS fwd", lbs), ("SMPL-H LBS fwd + bwd", lbs_bwd), ("torch sum / mean / max of 4M", t_sum)]
refs = {}
for name, fn in cands:
    a, b = fn().clone(), fn().clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b), name
    refs[name] = a
if os.environ.get("VICTIM_DUMP"):        # references only (e.g. under CHORE_LDS_POISON): saved for a comparison across processes
    import numpy as np
    np.savez(os.environ["VICTIM_DUMP"], **{k.replace(" ", "_").replace("/", "_"): v.cpu().numpy() for k, v in refs.items()})
    sys.exit(0)
stop = threading.Event()
def background():
    torch.cuda.set_device(dev)
    net2 = CHORE(o).to(dev).eval(); synth.load_synth_weights(net2, seed=1)
    img = torch.from_numpy(synth.synth_images(1, 512, 512, 3)).to(dev)
    st = torch.cuda.Stream(dev)
    mode = os.environ.get("VICTIM_BG", "encoder")      # what runs beside the candidates: an encoder, or backward queries
    if mode != "encoder":
        with torch.no_grad():
            net2.filter(img)
    bp = torch.from_numpy(synth.synth_points(1, 6890, seed=9)).to(dev)
    bg = torch.randn(1, 2, 6890, device=dev)
    with torch.cuda.stream(st), torch.no_grad():
        n = 0
        while not stop.is_set():
            if mode == "encoder": net2.filter(img)
            elif mode == "qbwd": net2.query_grad_points(bp, cc, g_df=bg)
            else: net2.query(bp, crop_center=cc)
            n += 1
            if n % 8 == 0: st.synchronize()
        st.synchronize()
t = threading.Thread(target=background); t.start()
import time; time.sleep(1.0)
for name, fn in cands:
    bad, worst = 0, 0.0
    for _ in range(REPS):
        out = fn()
        if not torch.equal(out, refs[name]):
            bad += 1
            worst = max(worst, float((out - refs[name]).abs().max()))
    print("%-32s %4d of %d runs differ (largest deviation %.3g of max %.3g)" % (name, bad, REPS, worst, float(refs[name].abs().max())), flush=True)
stop.set(); t.join()

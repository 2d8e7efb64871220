import sys, os, copy, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from bench import chore_opt
from chore_amd.lib_smpl.priors import synthetic_priors
from chore_amd.lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatch
from chore_amd.model import CHORE
from chore_amd.recon.recon_fit_behave import ReconFitterBehave
from chore_amd.recon.graph_step import EagerStep, GraphedStep
from chore_amd.utils import synth
from test_gpu_query import nhwc
B = 2
net = CHORE(chore_opt("fp32")).cuda().eval(); synth.load_synth_weights(net, seed=0)
for p in net.parameters(): p.requires_grad_(False)
rs = np.random.RandomState(9)
net.im_feat_list = [nhwc((rs.standard_normal((B, 256, 32, 32)) * 0.5).astype(np.float32))]
net.tmpx = nhwc((rs.standard_normal((B, 64, 64, 64)) * 0.5).astype(np.float32))
pose, betas, trans = synth.synth_smpl_params(B, seed=1); pose *= 0.3
body_prior, hand_prior = synthetic_priors(0)
labels = torch.from_numpy(rs.randint(0, 14, 6890)).cuda()
fitter = ReconFitterBehave(device="cuda:0", part_labels=labels, body_prior=body_prior, hand_prior=hand_prior)
cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
kpts = torch.from_numpy(np.concatenate([rs.uniform(100, 400, (B, 25, 2)), rs.uniform(0.2, 1, (B, 25, 1))], -1).astype(np.float32)).cuda()
wd = fitter.get_loss_weights()
def run(graphed, phase):
    smpl = SMPLPyTorchWrapperBatch(synth.synth_smplh_model(0), B, betas=torch.from_numpy(betas), pose=torch.from_numpy(pose), trans=torch.from_numpy(trans)).cuda()
    data = dict(net=net, query_dict={"crop_center": cc}, part_labels=labels.unsqueeze(0).repeat(B, 1), pose_init=torch.from_numpy(pose[:, 3:72]).cuda(), body_kpts=kpts)
    split = fitter.split_smpl(smpl)
    prev = torch.tensor(300.0, device="cuda")
    params = [split.top_betas, split.trans] if phase == "global" else [split.trans, split.global_pose, split.body_pose, split.top_betas, split.other_betas]
    st = (GraphedStep if graphed else EagerStep)(params, 0.02, capturable=True, loss_fn=lambda d: fitter.sum_dict(fitter.forward_smpl(split, data, phase), wd, d), tol=1e-3, prev=prev, release=fitter.release_graphs(split, net))
    out = []
    for it in range(2):
        st.begin_outer(1)
        for _ in range(4):
            st.step(); out.append(float(st.loss))
    return out, [p.detach().cpu().numpy().copy() for p in params]
for phase in ("global", "kpts"):
    e, pe = run(False, phase); g, pg = run(True, phase)
    print(phase, "eager", ["%.6f" % x for x in e]); print(phase, "graph", ["%.6f" % x for x in g])
    print("   param max diff", [float(np.abs(a - b).max()) for a, b in zip(pe, pg)])

print("---- phase sequence on one split ----")
def run_seq(graphed, reuse=True, keep=None):
    smpl = SMPLPyTorchWrapperBatch(synth.synth_smplh_model(0), B, betas=torch.from_numpy(betas), pose=torch.from_numpy(pose), trans=torch.from_numpy(trans)).cuda()
    data = dict(net=net, query_dict={"crop_center": cc}, part_labels=labels.unsqueeze(0).repeat(B, 1), pose_init=torch.from_numpy(pose[:, 3:72]).cuda(), body_kpts=kpts)
    split = fitter.split_smpl(smpl)
    prev = torch.tensor(300.0, device="cuda")
    rel = fitter.release_graphs(split, net)
    K = GraphedStep if graphed else EagerStep
    tb = torch.zeros(8, device="cuda")
    def lf(ph):
        def f(d):
            ld = fitter.forward_smpl(split, data, ph)
            for i, (k_, v_) in enumerate(ld.items()):
                tb[i].copy_(v_.detach())
            return fitter.sum_dict(ld, wd, d)
        return f
    out = []
    st = K([split.top_betas, split.trans], 0.02, lf("global"), 1e-3, prev, release=rel, capturable=True)
    for it in range(6):
        if it == 2:
            st = K([split.trans, split.global_pose, split.body_pose, split.top_betas, split.other_betas], 0.006, lf("smpl all pose"), 1e-3, prev, release=rel, capturable=True)
        if it == 4:
            if keep is not None: keep.append(st)
            st = K(st.params, 0.006, lf("kpts"), 1e-3, prev, opt=st.opt if reuse else None, release=rel, capturable=True)
            with torch.no_grad():
                ld = fitter.forward_smpl(split, data, "kpts")
            print("   terms", {k: round(float(v), 5) for k, v in ld.items()}, "adam step", [float(s_["step"]) for s_ in st.opt.state.values()][:2],
                  "expavg", [float(s_["exp_avg"].abs().sum()) for s_ in st.opt.state.values()][:3], "grad", [float(p_.grad.abs().sum()) if p_.grad is not None else None for p_ in st.params][:3])
        st.begin_outer(1 if it < 4 else it / 3)
        for _ in range(3):
            st.step(); out.append(float(st.loss))
    return out
for reuse, keep in ((True, None),):
    print("reuse opt", reuse, "keep old stepper alive", keep is not None)
    e = run_seq(False, reuse); g = run_seq(True, reuse, keep)
    for i in range(9, 18, 3): print(i // 3, ["%.5f" % x for x in e[i:i+3]], ["%.5f" % x for x in g[i:i+3]])

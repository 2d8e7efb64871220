"""stress of fit_recon(pipeline=True) against the serial loop: N loader batches of one frame, per batch the checksums of what the
preparation produced (last feature map, tmpx, both point clouds) and the fitted pose -- which stage differs first when the two loops
disagree?   usage: pipe_stress.py [batches] [rounds]      STRESS_CHAINS=1: fit_recon(pipeline="chains") instead; STRESS_EAGER=1: eager inner steps"""
import copy, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import bench
from chore_amd.model import CHORE
from chore_amd.recon.assets import SyntheticAssets
from chore_amd.recon.generator import Generator
from chore_amd.recon.recon_fit_behave import ReconFitterBehave
from chore_amd.utils import synth
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
o = bench.chore_opt("fp16x3")
loader = [bench.fit_batch_inputs(1, 10 + k, dev) for k in range(NB)]

def run(pipe):
    net = CHORE(o).to(dev).eval(); synth.load_synth_weights(net, seed=0)
    fitter = ReconFitterBehave(None, device=dev, obj_name="synthetic", outpath=None, args=o, assets=SyntheticAssets(0))
    fitter.use_graphs, fitter.reuse_graphs, fitter.early_stop, fitter.adam_capturable = not os.environ.get('STRESS_EAGER'), True, False, True
    fitter.batch_seed = 7
    fitter.smpl_iters = dict(iter_for_betas=1, iter_for_pose=1, iter_for_kpts=1, steps_per_iter=5, max_iter=1)
    fitter.object_iters = dict(obj_iter=2, sil_iter=2, joint_iter=2, max_iter=1, steps_per_iter=5)
    gen = Generator(net, None, threshold=2.0, sparse_thres=0.03, filter_val=1.0, device=dev)
    rec = {}
    orig = fitter.prepare_batch
    import threading
    cur = {}                      # batch index by preparing / optimising thread (chains: a thread per chain; pipelined: optimise_batch below)
    def prep(data, generator, index=None):
        out = orig(data, generator, index=index)
        out["_index"] = index
        m = generator.model
        rec[index] = [m.im_feat_list[-1].double().sum(), m.tmpx.double().sum(), out["pc"]["human"]["points"].double().sum(),
                      out["pc"]["object"]["points"].double().sum()]
        return out
    fitter.prepare_batch = prep
    stage = {}
    o_smpl, o_obj = fitter.optimize_smpl, fitter.init_obj_fit_data
    o_optb = fitter.optimise_batch
    def optb(prep_, *a, **k):
        cur[threading.get_ident()] = prep_["_index"]
        return o_optb(prep_, *a, **k)
    fitter.optimise_batch = optb
    def opt_smpl(smpl, betas_dict, **kw):
        key = cur[threading.get_ident()]
        pre = [smpl.pose.detach().double().sum(), smpl.betas.detach().double().sum(), smpl.trans.detach().double().sum()]
        out = o_smpl(smpl, betas_dict, **kw)
        stage[key] = pre + [out[0].pose.detach().double().sum(), out[0].trans.detach().double().sum()]
        return out
    def init_obj(*a, **k):
        out = o_obj(*a, **k)
        stage[cur[threading.get_ident()]] += [out[0].detach().double().sum(), out[2].detach().double().sum()]
        return out
    fitter.optimize_smpl, fitter.init_obj_fit_data = opt_smpl, init_obj
    torch.manual_seed(3)
    res = fitter.fit_recon(o, loader=loader, generator=gen, save=False, pipeline=pipe)
    torch.cuda.synchronize()
    return ({k: [float(v) for v in vs] + [float(v) for v in stage[k]] for k, vs in rec.items()},
            [r["pose"].detach().cpu().clone() for r in res])

ref_rec, ref_pose = run(False)
bad = 0
for r in range(ROUNDS):
    if os.environ.get("STRESS_FRESH_HANDLES"):
        from chore_amd import _lib
        for k in [k for k in _lib._handles if not isinstance(k, int)]:
            del _lib._handles[k]
    import threading
    print("round", r, "handles:", [k if isinstance(k, int) else hex(k[1] & 0xffffff) for k in __import__("chore_amd")._lib._handles], "threads", threading.active_count(), flush=True)
    rec, pose = run("chains" if os.environ.get("STRESS_CHAINS") else True)
    for k in range(NB):
        names = ("feat", "tmpx", "pc_human", "pc_object", "init_pose", "init_betas", "init_trans", "smpl_pose", "smpl_trans", "obj_R0", "obj_t0")
        d = [n for n, a, b in zip(names, ref_rec[k], rec[k]) if a != b]
        p = not torch.equal(ref_pose[k], pose[k])
        if d or p:
            bad += 1
            print("round %d batch %d: prepare differs in %s; pose differs: %s (%.3g)" % (r, k, d or "nothing", p, float((ref_pose[k] - pose[k]).abs().max())), flush=True)
    print("after round %d: %d mismatches so far" % (r, bad), flush=True)
print("mismatching (round, batch) pairs: %d of %d" % (bad, ROUNDS * NB))

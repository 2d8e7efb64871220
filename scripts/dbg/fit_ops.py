import os, sys, collections, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import bench
from chore_amd.model import CHORE
from chore_amd.recon.assets import SyntheticAssets
from chore_amd.recon.generator import Generator
from chore_amd.recon.recon_fit_behave import ReconFitterBehave
from chore_amd.utils import synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
opt = bench.chore_opt("fp16x3"); opt.gpu_id = 0
net = CHORE(opt).to(dev).eval(); synth.load_synth_weights(net, seed=0)
fitter = ReconFitterBehave(None, device=dev, obj_name="synthetic", outpath=None, args=opt, assets=SyntheticAssets(0))
fitter.use_graphs = False; fitter.early_stop = False
gen = Generator(net, None, threshold=2.0, sparse_thres=0.03, filter_val=1.0, device=dev)
data = bench.fit_batch_inputs(1, 0, dev)
pc = gen.generate_pclouds_batch(data, num_points=5000, num_steps=10, mute=True)
(betas_dict, body_kpts, human_parts, human_points, human_t, obj_points, part_colors, part_labels, query_dict, smpl) = fitter.prep_smplfit(data, gen, pc)
def count(fn, label, iters):
    fn()  # warm
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        fn()
    c = collections.Counter()
    for e in prof.events():
        if e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::"):
            c[e.name] += 1
    tot = sum(c.values())
    print(label, "top-level aten ops per iteration:", tot / iters)
    for k, v in c.most_common(22): print("   %6.1f  %s" % (v / iters, k))
count(lambda: fitter.optimize_smpl(smpl, betas_dict, iter_for_betas=1, iter_for_pose=1, iter_for_kpts=1, steps_per_iter=5, max_iter=0), "SMPL", 15)

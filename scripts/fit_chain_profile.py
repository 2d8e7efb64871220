"""host-side profile of one fit_recon chain (B = 1, the bench's schedules): where the wall time of the stages goes"""
import cProfile, io, os, pstats, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from chore_amd.model import CHORE
from chore_amd.recon.assets import SyntheticAssets
from chore_amd.recon.generator import Generator
from chore_amd.recon.recon_fit_behave import ReconFitterBehave
from chore_amd.utils import synth

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt = bench.chore_opt("fp16x3")
net = CHORE(opt).to(dev).eval(); synth.load_synth_weights(net, seed=0)
fitter = ReconFitterBehave(None, device=dev, obj_name="synthetic", outpath=None, args=opt, assets=SyntheticAssets(0))
fitter.use_graphs, fitter.early_stop = True, False
fitter.reuse_graphs = not os.environ.get('CHORE_FIT_NO_REUSE'); net.image_filter.static_outputs = fitter.reuse_graphs
gen = Generator(net, None, threshold=2.0, sparse_thres=0.03, filter_val=1.0, device=dev)
data = bench.fit_batch_inputs(B, 0, dev)
stages = {}
def clock(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    stages[name] = stages.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return out
def chain():
    pc = clock("generate_pclouds", lambda: gen.generate_pclouds_batch(data, num_points=5000, num_steps=10, mute=True))
    r = clock("prep_smplfit", lambda: fitter.prep_smplfit(data, gen, pc))
    betas_dict, body_kpts, part_labels, query_dict, smpl, human_t = r[0], r[1], r[7], r[8], r[9], r[4]
    smpl, scale = clock("optimize_smpl", lambda: fitter.optimize_smpl(smpl, betas_dict, **bench.SMPL_ITERS))
    obj_R, obj_s, obj_t, object_init = clock("init_obj", lambda: fitter.init_obj_fit_data(B, human_t, pc, scale))
    dd = {"obj_R": obj_R, "obj_t": obj_t, "obj_s": obj_s, "objects": object_init, "smpl": smpl, "images": data["images"],
          "body_kpts": body_kpts, "query_dict": query_dict, "part_labels": part_labels}
    clock("optimize_smpl_object", lambda: fitter.optimize_smpl_object(net, dd, **bench.OBJECT_ITERS))
for _ in range(2): chain()
stages.clear()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): chain()
pr.disable()
print({k: round(v / 3, 2) for k, v in stages.items()})
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(55); print(s.getvalue()[:9000])

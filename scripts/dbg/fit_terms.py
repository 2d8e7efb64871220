"""which term of the object phases costs what: times optimize_smpl_object (bench.py --mode fit setup) with terms removed"""
import sys, os, time
sys.path.insert(0, ".")
import torch
import bench
from chore_amd.model import CHORE
from chore_amd.recon.assets import SyntheticAssets
from chore_amd.recon.generator import Generator
from chore_amd.recon.recon_fit_behave import ReconFitterBehave
from chore_amd.utils import synth
dev = torch.device("cuda", 0)
opt = bench.chore_opt("bf16")
net = CHORE(opt).to(dev).eval(); synth.load_synth_weights(net, seed=0)
fitter = ReconFitterBehave(None, device=dev, obj_name="synthetic", outpath=None, args=opt, assets=SyntheticAssets(0))
fitter.use_graphs = True; fitter.early_stop = False
gen = Generator(net, None, threshold=2.0, sparse_thres=0.03, filter_val=1.0, device=dev)
data = bench.fit_batch_inputs(1, 0, dev)
pc = gen.generate_pclouds_batch(data, num_points=5000, num_steps=10, mute=True)
(betas_dict, body_kpts, human_parts, human_points, human_t, obj_points, part_colors, part_labels, query_dict, smpl) = fitter.prep_smplfit(data, gen, pc)
scale = torch.ones(1, device=dev)
for variant in ("all", "no_collide", "no_sil", "neither"):
    obj_R, obj_s, obj_t, object_init = fitter.init_obj_fit_data(1, human_t, pc, scale)
    dd = {"obj_R": obj_R, "obj_t": obj_t, "obj_s": obj_s, "objects": object_init, "smpl": smpl, "images": data["images"],
          "body_kpts": body_kpts, "query_dict": query_dict, "part_labels": part_labels}
    sf = fitter.scan_faces
    if variant in ("no_collide", "neither"): fitter.scan_faces = None
    it = dict(bench.OBJECT_ITERS)
    if variant in ("no_sil", "neither"):
        it["sil_iter"] = 0; it["obj_iter"] = 10
    fitter.timer = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fitter.optimize_smpl_object(net, dd, **it)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    per = {}
    for i, (e0, e1, n) in enumerate(fitter.timer):
        per[i] = e0.elapsed_time(e1) / n
    print(variant, "wall %.0f ms" % wall, "per-outer ms/iter:", " ".join("%.2f" % per[i] for i in sorted(per)))
    fitter.scan_faces = sf

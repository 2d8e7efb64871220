"""does a fit change its bits when an UNRELATED encoder runs beside it?  The serial fit_recon loop over NB one-frame batches, once alone and
ROUNDS times while a second host thread encodes images on its own stream with its OWN network object (no shared host state with the
fitter: if the fits differ, it is the device, not the pipelining logic).   usage: interference_stress.py [batches] [rounds]"""
import os, sys, threading, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import bench
from chore_amd.model import CHORE
from chore_amd.recon.assets import SyntheticAssets
from chore_amd.recon.generator import Generator
from chore_amd.recon.recon_fit_behave import ReconFitterBehave
from chore_amd.utils import synth
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
o = bench.chore_opt("fp16x3")
loader = [bench.fit_batch_inputs(1, 10 + k, dev) for k in range(NB)]

def fit():
    net = CHORE(o).to(dev).eval(); synth.load_synth_weights(net, seed=0)
    fitter = ReconFitterBehave(None, device=dev, obj_name="synthetic", outpath=None, args=o, assets=SyntheticAssets(0))
    fitter.use_graphs, fitter.reuse_graphs, fitter.early_stop, fitter.adam_capturable = False, False, False, True      # eager steps: a capture does not tolerate the other thread
    fitter.batch_seed = 7
    fitter.smpl_iters = dict(iter_for_betas=1, iter_for_pose=1, iter_for_kpts=1, steps_per_iter=5, max_iter=1)
    fitter.object_iters = dict(obj_iter=2, sil_iter=2, joint_iter=2, max_iter=1, steps_per_iter=5)
    gen = Generator(net, None, threshold=2.0, sparse_thres=0.03, filter_val=1.0, device=dev)
    torch.manual_seed(3)
    res = fitter.fit_recon(o, loader=loader, generator=gen, save=False, pipeline=False)
    torch.cuda.synchronize()
    return [r["pose"].detach().cpu().clone() for r in res]

ref = fit()
again = fit()
print("alone twice: %d of %d batches differ" % (sum(int(not torch.equal(a, b)) for a, b in zip(ref, again)), NB), flush=True)
stop = threading.Event()
def background():
    torch.cuda.set_device(dev)
    net2 = CHORE(o).to(dev).eval(); synth.load_synth_weights(net2, seed=1)
    img = torch.from_numpy(synth.synth_images(1, 512, 512, 3)).to(dev)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st), torch.no_grad():
        n = 0
        while not stop.is_set():
            net2.filter(img)
            n += 1
            if n % 8 == 0:
                st.synchronize()
        st.synchronize()
bad = 0
for r in range(ROUNDS):
    stop.clear()
    t = threading.Thread(target=background); t.start()
    got = fit()
    stop.set(); t.join()
    d = [k for k, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a, b)]
    bad += len(d)
    print("round %d beside an unrelated encoder: batches that differ: %s" % (r, d), flush=True)
print("differing (round, batch) pairs: %d of %d" % (bad, ROUNDS * NB))

"""times Generator.generate_pclouds_batch (Alg. 1) on synthetic data: B images, both targets"""
import sys, os, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.recon.generator import Generator
from chore_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = CHORE(chore_opt("bf16")).cuda().eval(); synth.load_synth_weights(net, 0)
gen = Generator(net, None, threshold=2.0, device=torch.device("cuda"), filter_val=10.0)   # synthetic field: keep everything
data = {"images": torch.from_numpy(synth.synth_images(B, 512, 512, 0)).cuda(), "crop_center": torch.tensor([synth.CROP_CENTER] * B).cuda()}
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = gen.generate_pclouds_batch(data, num_steps=10, num_points=5000, mute=True)
    torch.cuda.synchronize(); print("B=%d generate_pclouds_batch %.1f ms" % (B, (time.perf_counter() - t) * 1e3), {k: tuple(v["points"].shape) for k, v in out.items()})

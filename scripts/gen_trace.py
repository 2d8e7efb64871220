"""Generator.generate_pclouds_batch on B frames (the bench's fp16-fields configs[4] share by default): wall time of three calls,
for a rocprofv3 --kernel-trace of the call (scripts/prof_summary.py).   usage: gen_trace.py [B] [dtype]"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import bench
from chore_amd.model import CHORE
from chore_amd.recon.generator import Generator
from chore_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp16"
dev = torch.device("cuda", 0)
net = CHORE(bench.chore_opt(dtype)).to(dev).eval(); synth.load_synth_weights(net, seed=0)
gen = Generator(net, None, threshold=2.0, sparse_thres=0.03, filter_val=1.0, device=dev)
data = bench.fit_batch_inputs(B, 0, dev)
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = gen.generate_pclouds_batch(data, num_points=5000, num_steps=10, mute=True)
    torch.cuda.synchronize(); print("B=%d %s generate_pclouds_batch %.1f ms" % (B, dtype, (time.perf_counter() - t) * 1e3), flush=True)

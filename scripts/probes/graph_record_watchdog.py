"""Reproducer loop for: 'Process group watchdog thread terminated with exception: HIP error: operation not permitted on an event last
recorded in a capturing stream' during the first recorded training steps of a process with a one-rank RCCL group.
    python scripts/probes/graph_record_watchdog.py [flat|segmented] [nccl|mixed]      (exit code 0 = six steps ran;
mixed = a group made with "cpu:gloo,cuda:nccl" like bench.py's, whose get_backend() is that string)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
if len(sys.argv) > 2 and sys.argv[2] == "mixed":
    torch.cuda.set_device(0)
    dist.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
else:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from bench import chore_opt                                   # noqa: E402
from chore_amd.model import CHORE                             # noqa: E402
from chore_amd.utils import synth                             # noqa: E402
from chore_amd.parallel import FlatGradReducer, GraphedTrainStep, chore_segments  # noqa: E402

dev = torch.device("cuda", 0)
opt = chore_opt("fp16x3"); opt.gpu_id = 0
net = CHORE(opt).to(dev)
synth.load_synth_weights(net, seed=0)
net.train(True)
net.losses_on_host = False
optim = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True, capturable=True)
B, N = 4, 20000
rs = np.random.RandomState(50)
t = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
batch = dict(images=t(synth.synth_images(B, 512, 512, seed=0)), points=t(synth.synth_points(B, N, seed=1)),
             df_h=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)), df_o=t(rs.uniform(0, 0.3, (B, N)).astype(np.float32)),
             parts_gt=t(rs.randint(0, 14, (B, N))), pca_gt=t(rs.standard_normal((B, 3, 3, N)).astype(np.float32)),
             body_center=t((rs.standard_normal((B, 3)) * 0.3).astype(np.float32)),
             obj_center=t((rs.standard_normal((B, 3, N)) * 0.3).astype(np.float32)),
             crop_center=torch.tensor([synth.CROP_CENTER] * B, dtype=torch.float32, device=dev))
layout = sys.argv[1] if len(sys.argv) > 1 else "flat"
red = FlatGradReducer(net) if layout == "flat" else FlatGradReducer(net, segments=chore_segments(net))
g = GraphedTrainStep(net, optim, reducer=red, warmup=2)
dist.all_reduce(torch.zeros(1, device=dev))
for i in range(6):
    loss, _ = g(**batch)
torch.cuda.synchronize()
print("ok", float(loss))
dist.destroy_process_group()

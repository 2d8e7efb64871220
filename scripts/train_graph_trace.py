"""N training steps through chore_amd.parallel.GraphedTrainStep (for rocprofv3 --kernel-trace): python scripts/train_graph_trace.py [steps] [fp16x3|bf16|fp32]"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, REPO + "/tests")
from test_gpu_ddp_trainstep import _make
from chore_amd.parallel import GraphedTrainStep
net, batch = _make(0, sys.argv[2] if len(sys.argv) > 2 else "fp16x3")
opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True, fused=True)
step = GraphedTrainStep(net, opt, warmup=2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for _ in range(4):
    step(**batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    step(**batch)
torch.cuda.synchronize()
print("replayed step: %.2f ms" % ((time.perf_counter() - t0) / n * 1e3))

"""does an initialised RCCL process group change what two recordings in flight buy?  python scripts/inflight_rccl_probe.py [none|lazy|eager|eager_used]"""
import os, socket, subprocess, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
import torch
if mode != "none":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
    kw = dict(device_id=torch.device("cuda", 0)) if mode.startswith("eager") else {}
    dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1, **kw)
    if mode == "eager_used":
        t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
sys.argv = [sys.argv[0]]
sys.path.insert(0, REPO)
import runpy
runpy.run_path(os.path.join(REPO, "scripts", "graph_gap_probe2.py"), run_name="__main__")

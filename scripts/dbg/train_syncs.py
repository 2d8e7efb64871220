import os, sys, warnings
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
exec(open(os.path.join(REPO, "scripts/dbg/train_phases.py")).read().split("def phase")[0])
for it in range(2):
    optim.zero_grad(set_to_none=True); err, _ = net(**batch); err.backward(); optim.step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
optim.zero_grad(set_to_none=True); err, _ = net(**batch); err.backward(); optim.step()
torch.cuda.set_sync_debug_mode("default")
print("done")

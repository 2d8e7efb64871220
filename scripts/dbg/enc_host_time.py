import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
from chore_amd import _lib
net = CHORE(chore_opt("fp16x3")).cuda().eval(); synth.load_synth_weights(net, 0)
for p in net.parameters(): p.requires_grad_(False)
img = torch.from_numpy(synth.synth_images(4, 512, 512, 0)).cuda()
orig = _lib.lib.chore_encode_fwd
acc = [0.0, 0]
def wrapped(*a):
    t = time.perf_counter(); r = orig(*a); acc[0] += time.perf_counter() - t; acc[1] += 1; return r
class L:  # proxy
    def __getattr__(self, k): return wrapped if k == "chore_encode_fwd" else getattr(_lib_lib, k)
_lib_lib = _lib.lib
_lib.lib = L()
import chore_amd.model.hgfilter as hg
with torch.no_grad():
    for _ in range(5): net.filter(img)
    torch.cuda.synchronize(); acc[0] = 0; acc[1] = 0
    t = time.perf_counter()
    for _ in range(20): net.filter(img)
    host = (time.perf_counter() - t) / 20 * 1e3
    torch.cuda.synchronize()
print("host per filter %.3f ms, inside chore_encode_fwd %.3f ms (%d calls)" % (host, acc[0] / max(acc[1], 1) * 1e3, acc[1]))

"""times CHORE.filter (B = 4 x 512 x 512) issued eagerly and replayed from a hipGraph, for the batch-group counts in argv"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dt = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
net = CHORE(chore_opt(dt)).cuda().eval(); synth.load_synth_weights(net, 0)
for p in net.parameters(): p.requires_grad_(False)
img = torch.from_numpy(synth.synth_images(4, 512, 512, 0)).cuda()
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    eager = timeit(lambda: net.filter(img))
    t = time.perf_counter()
    for _ in range(20): net.filter(img)
    host = (time.perf_counter() - t) / 20 * 1e3
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        net.filter(img)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            net.filter(img)
    torch.cuda.synchronize()
    graph = timeit(g.replay)
print("groups %s: eager %.3f ms (host enqueue %.3f ms), graph replay %.3f ms" % (os.environ.get("CHORE_ENC_GROUPS", "default"), eager, host, graph))

import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dt = sys.argv[1]; what = sys.argv[2]
img = torch.from_numpy(synth.synth_images(4, 512, 512, 0)).cuda()
net = CHORE(chore_opt(dt)).cuda().eval(); synth.load_synth_weights(net, 0)
def T(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    print("%s %s: %.2f ms" % (dt, label, (time.perf_counter() - t) * 1e3), flush=True)
with torch.no_grad():
    T("first encode", lambda: net.filter(img))
    T("second encode", lambda: net.filter(img))
    if what == "b2b":
        T("5 encodes back to back", lambda: [net.filter(img) for _ in range(5)])
        T("20 encodes back to back", lambda: [net.filter(img) for _ in range(20)])
    if what == "events":
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        def f():
            a.record(); net.filter(img); b.record()
        T("encode between two events", f)
        T("again", f)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(40)]
        def g():
            for i in range(0, 40, 2):
                evs[i].record(); net.filter(img); evs[i + 1].record()
        T("20 encodes, an event pair each", g)

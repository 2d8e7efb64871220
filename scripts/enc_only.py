import sys, os, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
net = CHORE(chore_opt(dt)).cuda().eval(); synth.load_synth_weights(net, 0)
for p in net.parameters(): p.requires_grad_(False)
img = torch.from_numpy(synth.synth_images(4, 512, 512, 0)).cuda()
pts = torch.from_numpy(synth.synth_points(4, 20000, 1)).cuda(); cc = torch.tensor([synth.CROP_CENTER] * 4).cuda()
with torch.no_grad():
    for _ in range(n):
        net.filter(img); net.query(pts, crop_center=cc)
torch.cuda.synchronize()

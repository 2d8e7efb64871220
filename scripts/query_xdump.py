import ctypes, os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from bench import chore_opt
from chore_amd.model import CHORE
from chore_amd.utils import synth
L = ctypes.CDLL(os.environ["CHORE_HIP_LIB"])
net = CHORE(chore_opt("fp16x3")).cuda().eval(); synth.load_synth_weights(net, 0)
B, N, tile = 1, 3000, 12
with torch.no_grad():
    net.filter(torch.from_numpy(synth.synth_images(B, 128, 128, 0)).cuda())
    pts = torch.from_numpy(synth.synth_points(B, N, seed=3)).cuda()
    cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
    assert L.chore_debug_query_xdump(None, tile) == 0
    net.query(pts, crop_center=cc); torch.cuda.synchronize()
    buf = (ctypes.c_ushort * (2 * 64 * 344))()
    assert L.chore_debug_query_xdump(buf, -1) == 0
    x = np.frombuffer(buf, dtype=np.float16).reshape(-1)
    PTS = 32
    xh = x[:PTS * 344].reshape(PTS, 344)[:, :336].astype(np.float64); xl = x[PTS * 344:2 * PTS * 344].reshape(PTS, 344)[:, :336].astype(np.float64)
    from chore_amd import _lib
    feat, tmpx = net.im_feat_list[-1], net.tmpx          # (B,C,H,W) channels-last views
    fh, fw_, th, tw = feat.shape[2], feat.shape[3], tmpx.shape[2], tmpx.shape[3]
    feats = torch.empty(B, N, 323, device="cuda")
    h = _lib.handle(0)
    _lib.check(_lib.lib.chore_sample_features(h, pts.data_ptr(), cc.data_ptr(), B, N, feat.data_ptr(), fh, fw_, tmpx.data_ptr(), th, tw,
                                              _lib.F32, net._cam6, feats.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream), h, "sf")
    torch.cuda.synchronize()
print("have sample_features:", feats is not None, feat.dtype, feat.stride())
if feats is not None:
    f = feats[0] if isinstance(feats, (tuple, list)) else feats
    f = f.cpu().numpy().reshape(N, -1)[tile * PTS:(tile + 1) * PTS].astype(np.float64)
    rec = (xh + xl)[:, :323]
    err = np.abs(rec - f) / np.maximum(np.abs(f), 1e-3)
    print("max rel err of hi+lo vs r per point:", np.round(err.max(1) * 1e6, 2))
    p = 390 - tile * PTS
    bad = np.argsort(-err[p])[:8]
    print("point", p, "worst k:", bad, "r", f[p, bad], "hi", xh[p, bad], "lo", xl[p, bad])
    print("pad k=323..335 zero:", np.abs(xh[:, 323:]).max(), np.abs(xl[:, 323:]).max())
f32 = (feats[0] if isinstance(feats, (tuple, list)) else feats).cpu().numpy().reshape(N, -1)[tile * PTS:(tile + 1) * PTS].astype(np.float32)
hi = f32.astype(np.float16)
lo = (f32 - hi.astype(np.float32)).astype(np.float16)
XH = x[:PTS * 344].reshape(PTS, 344)[:, :323]; XL = x[PTS * 344:2 * PTS * 344].reshape(PTS, 344)[:, :323]
dh = (XH.view(np.uint16) != hi.view(np.uint16)); dl = (XL.view(np.uint16) != lo.view(np.uint16))
print("hi planes differ at", int(dh.sum()), "entries; lo planes differ at", int(dl.sum()))
for p_, k_ in np.argwhere(dh | dl)[:10]:
    print("  pt %d k %d: r %.9g  kernel hi %.9g lo %.9g   host hi %.9g lo %.9g" % (p_, k_, f32[p_, k_], XH[p_, k_], XL[p_, k_], hi[p_, k_], lo[p_, k_]))

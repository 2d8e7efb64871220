"""Layer operators of the encoder as autograd nodes (training path).

Every node runs HIP kernels of libchore_hip.so forward AND backward:
  conv_gn(x, w, bias, gamma, beta)   y = conv_{3x3|1x1}(relu(groupnorm32(x))) + bias   (one fused op = one layer of
                                     ConvBlock, /root/reference/model/net_util.py:374-396; gamma=None: plain conv)
  gn_relu(x, gamma, beta)            y = relu(groupnorm32(x))                        (HGFilters.py:150,170)
  stem(images, w, bias, dtype)       y = conv7x7/2(images) + bias, NCHW fp32 -> NHWC  (HGFilters.py:102,149)
  avgpool2(x), upadd(a, low)         2x2 average pooling; a + bicubic_up2(low)        (HGFilters.py:33,47-50)
Tensors are NHWC (B,H,W,C) contiguous, float32 or bfloat16; parameters are float32 in the reference layouts.
Backward of conv_gn: data gradient = the forward convolution kernel on transposed, flipped weights
(chore_conv2d_bwd_data); weight gradient = pixel-contraction MFMA GEMM (chore_conv2d_bwd_weight, recomputes
relu(gn(x)) while staging); GroupNorm+ReLU backward from the exact group statistics (chore_gn_relu_bwd).
"""
import os

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
_SIDE = {}


def _side_stream(dev):
    """the stream the stand-alone layers' weight gradients run on (one per device)"""
    s = _SIDE.get(dev)
    if s is None:
        s = _SIDE[dev] = torch.cuda.Stream(dev)
    return s

# fp16 x 3 training ("fp16x3" compute mode): fp32 tensors whose convolutions -- forward, data gradient, weight gradient -- run on
# the fp16 matrix cores with hi / lo split operands (csrc/conv_pc.hip, train_bwd.hip wgrad64_x3_kernel).  The tensor dtype cannot
# tell this mode from the exact-fp32 one, so it is a switch the forward pass sets (`with ops.x3_convs():`); every node remembers
# the dtype word of its forward for its backward, which runs outside the `with`.
# The switch is per host thread: two threads running training forwards in different modes must not see each other's.
import threading as _threading

_X3 = _threading.local()


class x3_convs:
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = getattr(_X3, "on", False)
        _X3.on = self.on
        return self

    def __exit__(self, *exc):
        _X3.on = self.prev
        return False


def _conv_dt(dt):
    """dtype word of the convolution entry points for a tensor of element type dt"""
    return _lib.F16X3 if (getattr(_X3, "on", False) and dt == _lib.F32) else dt


def _absmax(t, dev, h, stream):
    """the range cells of a gradient tensor (chore_absmax_f32): what the fp16 x 3 data / weight gradient kernels scale it by"""
    cells = torch.empty(_lib.lib.chore_amax_bytes(), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.chore_absmax_f32(h, t.data_ptr(), t.numel(), cells.data_ptr(), stream), h, "chore_absmax_f32")
    return cells


def _u8(n, dev):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=dev)


def _env(x):
    if not x.is_cuda:
        raise RuntimeError("chore_amd needs device tensors (no CPU path)")
    if x.dtype not in _DT or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("expected a contiguous NHWC (B,H,W,C) float32 / bfloat16 tensor")
    dev = x.device
    return dev, _lib.handle(dev.index or 0), _DT[x.dtype], torch.cuda.current_stream(dev).cuda_stream


class ZeroArena:
    """Zero-filled device memory handed out in slices: the statistics accumulators of the ~180 GroupNorms of a pass and
    the accumulators of their backward passes have to start at zero; one fill per 4 MB replaces a fill per layer."""
    CHUNK = 4 << 20

    def __init__(self, dev):
        self.dev, self.buf, self.off = dev, None, 0

    def take(self, nbytes):
        n = (int(nbytes) + 255) // 256 * 256
        if self.buf is None or self.off + n > self.buf.numel():
            self.buf, self.off = torch.zeros(max(self.CHUNK, n), dtype=torch.uint8, device=self.dev), 0
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v


_arena = None


class zero_arena:
    """with ops.zero_arena(device): ...   -- the operators inside draw their zeroed accumulators from one arena"""

    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        global _arena
        self.prev, _arena = _arena, ZeroArena(self.dev)
        return _arena

    def __exit__(self, *exc):
        global _arena
        _arena = self.prev
        return False


def _zeros(nbytes, dev):
    """(tensor of zero bytes, 1) from the active arena, else (uninitialised tensor, 0): the callee clears it itself"""
    if _arena is not None and _arena.dev == dev:
        return _arena.take(nbytes), 1
    return _u8(nbytes, dev), 0


def _zeroed_stats(B, dev):
    """zeroed GroupNorm statistics accumulators (from the pass's arena when one is active)"""
    nb = _lib.lib.chore_gn_stats_bytes(B)
    if _arena is not None and _arena.dev == dev:
        return _arena.take(nb)
    return torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)


def gn_stats(x):
    dev, h, dt, stream = _env(x)
    B, H, W, C = x.shape
    st, z = _zeros(_lib.lib.chore_gn_stats_bytes(B), dev)
    _lib.check(_lib.lib.chore_gn_stats(h, dt, x.data_ptr(), B, H * W, C, st.data_ptr(), z, stream), h, "chore_gn_stats")
    return st


def _bwd_acc(B, C, dev):
    """zeroed accumulators for one GroupNorm backward, reserved at FORWARD time (the arena is live then)"""
    if _arena is not None and _arena.dev == dev:
        return _arena.take(_lib.lib.chore_gn_relu_bwd_workspace_bytes(B, C))
    return None


class _GNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, x_stats):
        dev, h, dt, stream = _env(x)
        B, H, W, C = x.shape
        st = gn_stats(x) if x_stats is None else x_stats
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.lib.chore_gn_relu_fwd(h, dt, x.data_ptr(), st.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), B,
                                              H * W, C, stream), h, "chore_gn_relu_fwd")
        ctx.save_for_backward(x, st, g, b)
        ctx.acc = _bwd_acc(B, C, dev)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, g, b = ctx.saved_tensors
        dx, dg, db = _gn_relu_bwd(x, st, g, b, dy.contiguous().to(x.dtype), ctx.acc)
        ctx.acc = None
        return dx, dg, db, None


def _gn_relu_bwd(x, st, g, b, da, acc=None):
    dev, h, dt, stream = _env(x)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = acc if acc is not None else _u8(_lib.lib.chore_gn_relu_bwd_workspace_bytes(B, C), dev)
    _lib.check(_lib.lib.chore_gn_relu_bwd(h, dt, x.data_ptr(), st.data_ptr(), g.data_ptr(), b.data_ptr(), da.data_ptr(), B,
                                          H * W, C, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                          1 if acc is not None else 0, stream), h, "chore_gn_relu_bwd")
    return dx, dg, db


class _ConvGN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, x_stats, want_stats):
        dev, h, dt, stream = _env(x)
        B, H, W, Cin = x.shape
        Cout, taps = w.shape[0], w.shape[2] * w.shape[3]
        wf = w.detach().float().contiguous()
        bf = None if bias is None else bias.detach().float().contiguous()
        st = g = b = None
        if gamma is not None:
            st = gn_stats(x) if x_stats is None else x_stats
            g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = torch.empty(B, H, W, Cout, dtype=x.dtype, device=dev)
        # statistics of y from the convolution's epilogue (for the GroupNorm that consumes y): zeroed accumulators
        st_y = None
        if want_stats:
            nb = _lib.lib.chore_gn_stats_bytes(B)
            st_y = _arena.take(nb) if (_arena is not None and _arena.dev == dev) else \
                torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)
        cdt = ctx.cdt = _conv_dt(dt)
        ws = _u8(_lib.lib.chore_conv2d_workspace_bytes(cdt, taps, Cin, Cout), dev)
        _lib.check(_lib.lib.chore_conv2d_fwd(h, cdt, taps, x.data_ptr(), B, H, W, Cin, None if st is None else st.data_ptr(),
                                             None if g is None else g.data_ptr(), None if b is None else b.data_ptr(),
                                             wf.data_ptr(), None if bf is None else bf.data_ptr(), Cout, y.data_ptr(),
                                             None if st_y is None else st_y.data_ptr(), ws.data_ptr(), stream), h,
                   "chore_conv2d_fwd")
        ctx.has_gn, ctx.has_bias = gamma is not None, bias is not None
        ctx.save_for_backward(x, wf, *([st, g, b] if gamma is not None else []))
        ctx.acc = _bwd_acc(B, Cin, dev) if gamma is not None else None
        if want_stats:
            ctx.set_materialize_grads(False)     # no zero tensor for the statistics' (unused) gradient slot
            ctx.mark_non_differentiable(st_y)
            return y, st_y
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        x, wf = ctx.saved_tensors[:2]
        st, g, b = ctx.saved_tensors[2:] if ctx.has_gn else (None, None, None)
        dev, h, dt, stream = _env(x)
        B, H, W, Cin = x.shape
        Cout, taps = wf.shape[0], wf.shape[2] * wf.shape[3]
        dy = dy.contiguous().to(x.dtype)
        dw = torch.empty_like(wf)
        dbias = torch.empty(Cout, device=dev) if ctx.has_bias else None
        ws = _u8(_lib.lib.chore_conv2d_wgrad_workspace_bytes(taps, B, H, W, Cin, Cout), dev)
        dt = ctx.cdt
        amax = _absmax(dy, dev, h, stream) if dt == _lib.F16X3 else None
        ap = None if amax is None else amax.data_ptr()
        # CHORE_CONVGN_SIDE=1: the weight gradient of a stand-alone layer on a side stream beside its data-gradient chain (fork after
        # dy's range is known, join before this node returns; the caller's-stream kernels issued first, see convblock.hip).
        # Built and measured in round 5 -- and OFF by default: the 1x1 layers' data and weight gradients are both bound by the same
        # HBM traffic, side by side each takes as long as the two in turn (77 + 174 us in turn, 227 and 224 us together), the
        # step is 31.6 ms either way (profiles/r05_train_kernel_stats.txt).
        cur = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if os.environ.get("CHORE_CONVGN_SIDE") else None
        if side is not None:
            fork = torch.cuda.Event()
            fork.record(cur)
        dx = dg = db = None
        if ctx.needs_input_grad[0] or ctx.has_gn:
            da = torch.empty_like(x)                      # gradient w.r.t. what the convolution saw
            ws2 = _u8(_lib.lib.chore_conv2d_workspace_bytes(dt, taps, Cout, Cin), dev)
            _lib.check(_lib.lib.chore_conv2d_bwd_data(h, dt, taps, dy.data_ptr(), B, H, W, Cout, wf.data_ptr(), Cin,
                                                      da.data_ptr(), ws2.data_ptr(), ap, stream), h, "chore_conv2d_bwd_data")
            if ctx.has_gn:
                dx, dg, db = _gn_relu_bwd(x, st, g, b, da, ctx.acc)
                ctx.acc = None
            else:
                dx = da
        if side is not None:
            side.wait_event(fork)
        _lib.check(_lib.lib.chore_conv2d_bwd_weight(h, dt, taps, x.data_ptr(), B, H, W, Cin,
                                                    None if st is None else st.data_ptr(),
                                                    None if g is None else g.data_ptr(), None if b is None else b.data_ptr(),
                                                    dy.data_ptr(), Cout, dw.data_ptr(),
                                                    None if dbias is None else dbias.data_ptr(), ws.data_ptr(), ap,
                                                    stream if side is None else side.cuda_stream), h,
                   "chore_conv2d_bwd_weight")
        if side is not None:
            cur.wait_stream(side)
        return dx, dw, dbias, dg, db, None, None


class _ConvBlock(torch.autograd.Function):
    """one ConvBlock (net_util.py:374-396) as ONE node: chore_convblock_fwd / chore_convblock_bwd (csrc/convblock.hip).
    Returns (y, statistics of y) -- the statistics are what the next block's GroupNorms need."""

    @staticmethod
    def forward(ctx, x, x_stats, w1, w2, w3, g1, b1, g2, b2, g3, b3, wd, g4, b4):
        import ctypes
        dev, h, dt, stream = _env(x)
        dt = ctx.cdt = _conv_dt(dt)
        B, H, W, Cin = x.shape
        Cout = w1.shape[0] * 2
        ps = [None if p is None else p.detach().float().contiguous() for p in (w1, w2, w3, wd, g1, b1, g2, b2, g3, b3, g4, b4)]
        L = _lib.lib
        y = torch.empty(B, H, W, Cout, dtype=x.dtype, device=dev)
        saved = _u8(L.chore_convblock_saved_bytes(dt, B, H, W, Cin, Cout), dev)
        ws = _u8(L.chore_convblock_workspace_bytes(dt, B, H, W, Cin, Cout), dev)
        gb = (ctypes.c_void_p * 8)(*[None if p is None else p.data_ptr() for p in ps[4:]])
        _lib.check(L.chore_convblock_fwd(h, dt, x.data_ptr(), None if x_stats is None else x_stats.data_ptr(), B, H, W, Cin, Cout,
                                         ps[0].data_ptr(), ps[1].data_ptr(), ps[2].data_ptr(),
                                         None if ps[3] is None else ps[3].data_ptr(), gb, y.data_ptr(), saved.data_ptr(),
                                         ws.data_ptr(), stream), h, "chore_convblock_fwd")
        ctx.save_for_backward(x, saved, *[p for p in ps if p is not None])
        ctx.x_stats = x_stats            # a plain (non-differentiable) byte tensor, kept alive for the backward
        ctx.down = wd is not None
        off, nb = L.chore_convblock_out_stats_offset(B), L.chore_gn_stats_bytes(B)
        y_stats = saved[off:off + nb]
        ctx.set_materialize_grads(False)     # no zero tensor per block for the statistics' (unused) gradient slot
        ctx.mark_non_differentiable(y_stats)
        return y, y_stats

    @staticmethod
    def backward(ctx, dy, _unused):
        import ctypes
        x, saved = ctx.saved_tensors[:2]
        ps = list(ctx.saved_tensors[2:])
        if ctx.down:
            w1, w2, w3, wd, g1, b1, g2, b2, g3, b3, g4, b4 = ps
        else:
            (w1, w2, w3, g1, b1, g2, b2, g3, b3), wd, g4, b4 = ps, None, None, None
        dev, h, dt, stream = _env(x)
        dt = ctx.cdt
        B, H, W, Cin = x.shape
        Cout = w1.shape[0] * 2
        C1, C2 = Cout // 2, Cout // 4
        L = _lib.lib
        dy = dy.contiguous().to(x.dtype)
        dx = torch.empty_like(x)
        grads = torch.empty(L.chore_convblock_grad_floats(Cin, Cout), device=dev)
        ws = _u8(L.chore_convblock_workspace_bytes(dt, B, H, W, Cin, Cout), dev)
        gb = (ctypes.c_void_p * 8)(*[None if p is None else p.data_ptr() for p in (g1, b1, g2, b2, g3, b3, g4, b4)])
        xs = ctx.x_stats
        _lib.check(L.chore_convblock_bwd(h, dt, x.data_ptr(), None if xs is None else xs.data_ptr(), dy.data_ptr(), B, H, W, Cin,
                                         Cout, w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), None if wd is None else wd.data_ptr(),
                                         gb, saved.data_ptr(), dx.data_ptr(), grads.data_ptr(), ws.data_ptr(), stream), h,
                   "chore_convblock_bwd")
        o = 0

        def take(*shape):
            nonlocal o
            n = 1
            for d in shape:
                n *= d
            t = grads[o:o + n].view(shape)
            o += n
            return t
        dw1, dw2, dw3 = take(C1, Cin, 3, 3), take(C2, C1, 3, 3), take(C2, C2, 3, 3)
        dwd = take(Cout, Cin, 1, 1) if ctx.down else None
        dg1, db1, dg2, db2, dg3, db3 = take(Cin), take(Cin), take(C1), take(C1), take(C2), take(C2)
        dg4, db4 = (take(Cin), take(Cin)) if ctx.down else (None, None)
        return dx, None, dw1, dw2, dw3, dg1, db1, dg2, db2, dg3, db3, dwd, dg4, db4


def conv_block(x, m, x_stats=None):
    """ConvBlock module m (conv1..3, bn1..4, downsample) applied to x (B,H,W,Cin) -> (y, statistics of y)"""
    ds = m.downsample
    return _ConvBlock.apply(x, x_stats, m.conv1.weight, m.conv2.weight, m.conv3.weight, m.bn1.weight, m.bn1.bias,
                            m.bn2.weight, m.bn2.bias, m.bn3.weight, m.bn3.bias,
                            None if ds is None else ds[2].weight, None if ds is None else m.bn4.weight,
                            None if ds is None else m.bn4.bias)


class _UpAdd(torch.autograd.Function):
    """y = a + bicubic_up2(low), align_corners=True (HourGlass._forward, HGFilters.py:47-50)"""

    @staticmethod
    def forward(ctx, a, low, want_stats=False):
        dev, h, dt, stream = _env(low)
        B, H, W, C = low.shape
        if a.shape != (B, 2 * H, 2 * W, C) or a.dtype != low.dtype or not a.is_contiguous():
            raise ValueError("upadd: a must be (B,2H,2W,C), contiguous, same dtype as low")
        y = torch.empty_like(a)
        st = _zeroed_stats(B, dev) if want_stats else None
        _lib.check(_lib.lib.chore_upadd_fwd(h, dt, a.data_ptr(), low.data_ptr(), y.data_ptr(), B, H, W, C,
                                            None if st is None else st.data_ptr(), stream), h, "chore_upadd_fwd")
        ctx.shape = (B, H, W, C)
        if want_stats:
            ctx.set_materialize_grads(False)     # no zero tensor for the statistics' (unused) gradient slot
            ctx.mark_non_differentiable(st)
            return y, st
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        B, H, W, C = ctx.shape
        dy = dy.contiguous()
        dev, h, dt, stream = _env(dy)
        dlow = torch.empty(B, H, W, C, dtype=dy.dtype, device=dev)
        _lib.check(_lib.lib.chore_up2_bwd(h, dt, dy.data_ptr(), dlow.data_ptr(), B, H, W, C, stream), h, "chore_up2_bwd")
        return dy, dlow, None


class _AvgPool2(torch.autograd.Function):
    """2x2 average pooling (HGFilters.py:33,153)"""

    @staticmethod
    def forward(ctx, x, want_stats=False):
        dev, h, dt, stream = _env(x)
        B, H, W, C = x.shape
        y = torch.empty(B, H // 2, W // 2, C, dtype=x.dtype, device=dev)
        st = _zeroed_stats(B, dev) if want_stats else None
        _lib.check(_lib.lib.chore_avgpool2_fwd(h, dt, x.data_ptr(), y.data_ptr(), B, H, W, C, None if st is None else st.data_ptr(),
                                               stream), h, "chore_avgpool2_fwd")
        ctx.shape = tuple(x.shape)
        if want_stats:
            ctx.set_materialize_grads(False)     # no zero tensor for the statistics' (unused) gradient slot
            ctx.mark_non_differentiable(st)
            return y, st
        return y

    @staticmethod
    def backward(ctx, dy, *unused):
        dy = dy.contiguous()
        dev, h, dt, stream = _env(dy)
        B, H, W, C = ctx.shape
        dx = torch.empty(B, H, W, C, dtype=dy.dtype, device=dev)
        _lib.check(_lib.lib.chore_avgpool2_bwd(h, dt, dy.data_ptr(), dx.data_ptr(), B, H, W, C, stream), h, "chore_avgpool2_bwd")
        return dx, None


class _Stem(torch.autograd.Function):
    """y (B,H/2,W/2,64) = conv7x7/2(images (B,Cin,H,W) fp32) + bias (HGFilters.py:102,149); the images take no gradient"""

    @staticmethod
    def forward(ctx, images, w, bias, tdt):
        if not images.is_cuda:
            raise RuntimeError("chore_amd needs device tensors (no CPU path)")
        dev = images.device
        h, stream = _lib.handle(dev.index or 0), torch.cuda.current_stream(dev).cuda_stream
        img = images.detach().float().contiguous()
        wf, bf = w.detach().float().contiguous(), bias.detach().float().contiguous()
        B, Cin, H, W = img.shape
        y = torch.empty(B, H // 2, W // 2, 64, dtype=tdt, device=dev)
        ws = _u8(_lib.lib.chore_stem_workspace_bytes(Cin), dev)
        _lib.check(_lib.lib.chore_stem_fwd(h, _DT[tdt], img.data_ptr(), B, Cin, H, W, wf.data_ptr(), bf.data_ptr(), y.data_ptr(),
                                           ws.data_ptr(), stream), h, "chore_stem_fwd")
        ctx.save_for_backward(img)
        return y

    @staticmethod
    def backward(ctx, dy):
        (img,) = ctx.saved_tensors
        dy = dy.contiguous()
        dev, h, dt, stream = _env(dy)
        B, Cin, H, W = img.shape
        dw, db = torch.empty(64, Cin, 7, 7, device=dev), torch.empty(64, device=dev)
        ws = _u8(_lib.lib.chore_stem_wgrad_workspace_bytes(B, Cin, H, W), dev)
        _lib.check(_lib.lib.chore_stem_bwd_weight(h, dt, img.data_ptr(), B, Cin, H, W, dy.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                  ws.data_ptr(), stream), h, "chore_stem_bwd_weight")
        return None, dw, db, None


def avgpool2(x, want_stats=False):
    """want_stats: -> (y, GroupNorm statistics of y), accumulated by the pooling kernel itself"""
    return _AvgPool2.apply(x, want_stats)


def stem(images, w, bias, tdt):
    return _Stem.apply(images, w, bias, tdt)


def upadd(a, low, want_stats=False):
    return _UpAdd.apply(a, low, want_stats)


def conv_gn(x, w, bias=None, gamma=None, beta=None, x_stats=None, want_stats=False):
    """x_stats: statistics of x if the caller already has them (gn_stats(x), or the st_y of the conv that made x);
    want_stats: also return the statistics of y, computed in the convolution's epilogue -> (y, st_y)"""
    return _ConvGN.apply(x, w, bias, gamma, beta, x_stats, want_stats)


def gn_relu(x, gamma, beta, x_stats=None):
    return _GNReLU.apply(x, gamma, beta, x_stats)

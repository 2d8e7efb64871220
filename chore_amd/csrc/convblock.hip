// convblock.hip -- one ConvBlock of the stacked hourglass as a TRAINING operator: forward and backward in one call each.
//
// ConvBlock (/root/reference/model/net_util.py:346-396), 43 of them per encoder pass:
//     o1 = conv3x3(relu(gn1(x)))   Cin    -> Cout/2
//     o2 = conv3x3(relu(gn2(o1)))  Cout/2 -> Cout/4
//     o3 = conv3x3(relu(gn3(o2)))  Cout/4 -> Cout/4
//     y  = cat(o1, o2, o3) + (x  or  conv1x1(relu(gn4(x))) when Cin != Cout)
// Composing this from per-layer autograd nodes (ops.hip) costs a concat, a residual add, strided-slice copies and
// gradient sums around every block, all of them elementwise passes over HBM and host work.  Here the block is laid out the
// way the inference program lays it out (encoder.hip, conv_block): every convolution writes its slice of y with the residual
// added in its epilogue, the raw o1 / o2 go to their own buffers, and the GroupNorm statistics of o1, o2 and y come from
// the same epilogues.  The backward reads dy in channel-strided slices (wgrad / dgrad take a row stride) and the
// GroupNorm backward adds the skip gradients on its way out, so no glue kernel runs at all.
// The arithmetic of every kernel is unchanged; in fp32 the results equal the per-layer composition bit for bit.
#include "enc_common.h"
#include <cstdlib>

namespace {

struct Dims {
    int B, H, W, Cin, Cout, C1, C2;
    size_t px, es, nb;       // pixels, element size, bytes of one statistics block
    bool down;
};

bool make_dims(Dims& d, int dtype, int B, int H, int W, int Cin, int Cout) {
    if ((dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3) || B <= 0 || H <= 0 || W <= 0) return false;
    if (Cout % 128 || Cin % 32 || Cin > 256 || Cout > 256) return false;      // slices of Cout/4 channels, whole groups
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.C1 = Cout / 2; d.C2 = Cout / 4;
    d.px = (size_t)B * H * W; d.es = dtype == CHORE_BF16 ? 2 : 4; d.nb = act_stats_bytes(B);
    d.down = Cin != Cout;
    return true;
}
inline size_t al(size_t n) { return (n + 255) / 256 * 256; }

// saved (forward -> backward): statistics [x | o1 | o2 | y], then o1, o2
struct Saved {
    char *sx, *s1, *s2, *sy, *o1, *o2;
    size_t bytes;
};
Saved saved_layout(const Dims& d, void* base) {
    Saved s;
    char* p = (char*)base;
    s.sx = p; s.s1 = p + d.nb; s.s2 = p + 2 * d.nb; s.sy = p + 3 * d.nb;
    size_t o = al(4 * d.nb);
    s.o1 = p + o; o += al(d.px * d.C1 * d.es);
    s.o2 = p + o; o += al(d.px * d.C2 * d.es);
    s.bytes = o;
    return s;
}

struct PackOff { size_t w1, w2, w3, wd, end; };
PackOff pack_layout(const Dims& d, int dtype, size_t o, bool transposed) {
    PackOff p;
    (void)transposed;     // the transposed (data-gradient) forms have the same sizes
    p.w1 = o; o += al(packed_conv_bytes(dtype, 9, d.Cin, d.C1));
    p.w2 = o; o += al(packed_conv_bytes(dtype, 9, d.C1, d.C2));
    p.w3 = o; o += al(packed_conv_bytes(dtype, 9, d.C2, d.C2));
    p.wd = o; if (d.down) o += al(packed_conv_bytes(dtype, 1, d.Cin, d.Cout));
    p.end = o;
    return p;
}

// the handle's side stream and its events (lazily created on the handle's device)
int side_stream(chore_handle* h) {
    if (h->side) return CHORE_OK;
    CHORE_HIP_CHECK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    for (hipEvent_t& e : h->side_ev) CHORE_HIP_CHECK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return CHORE_OK;
}

View mkview(const void* p, int cs, int co, int C) { View v; v.p = const_cast<void*>(p); v.cs = cs; v.co = co; v.C = C; return v; }

size_t gn_acc_bytes(int B, int C) { return chore_gn_relu_bwd_workspace_bytes(B, C); }   // two tables of cells (enc_common.h)
constexpr size_t AMAX_BYTES = AMAX_CELLS * sizeof(unsigned);      // range cells of one gradient tensor (fp16 x 3 mode)

}  // namespace

extern "C" {

size_t chore_convblock_saved_bytes(int dtype, int B, int H, int W, int Cin, int Cout) {
    Dims d;
    if (!make_dims(d, dtype, B, H, W, Cin, Cout)) return 0;
    return saved_layout(d, nullptr).bytes;
}
// byte offset of the statistics of y inside `saved` (chore_gn_stats_bytes(B) of them): what the NEXT block's x_stats wants
size_t chore_convblock_out_stats_offset(int B) { return 3 * act_stats_bytes(B); }

size_t chore_convblock_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout) {
    Dims d;
    if (!make_dims(d, dtype, B, H, W, Cin, Cout)) return 0;
    const size_t fwd = pack_layout(d, dtype, 0, false).end;
    size_t o = al(gn_acc_bytes(B, Cin) * 2 + gn_acc_bytes(B, d.C1) + gn_acc_bytes(B, d.C2) + 2 * AMAX_BYTES) + AMAX_BYTES;
    o = pack_layout(d, dtype, o, true).end;
    // the four weight gradients keep their shares' partials until ONE launch sums all of them: a region each
    o += al(chore_conv2d_wgrad_workspace_bytes(9, B, H, W, Cin, d.C1)) + al(chore_conv2d_wgrad_workspace_bytes(9, B, H, W, d.C1, d.C2)) +
         al(chore_conv2d_wgrad_workspace_bytes(9, B, H, W, d.C2, d.C2)) + (d.down ? al(chore_conv2d_wgrad_workspace_bytes(1, B, H, W, Cin, Cout)) : 0);
    o += al(d.px * (size_t)(Cin > d.C1 ? Cin : d.C1) * d.es);      // da
    o += al(d.px * d.C2 * d.es) + al(d.px * d.C1 * d.es);           // d(o2), d(o1)
    if (d.down) o += al(d.px * Cin * d.es);                          // gradient through the downsample branch
    return fwd > o ? fwd : o;
}
// floats of the parameter-gradient arena: dW1 dW2 dW3 [dWd] then (dgamma, dbeta) of bn1, bn2, bn3 [, bn4]
size_t chore_convblock_grad_floats(int Cin, int Cout) {
    const size_t C1 = Cout / 2, C2 = Cout / 4;
    size_t n = 9 * ((size_t)Cin * C1 + C1 * C2 + C2 * C2) + 2 * (Cin + C1 + C2);
    if (Cin != Cout) n += (size_t)Cin * Cout + 2 * Cin;
    return n;
}

// y (B,H,W,Cout) = ConvBlock(x (B,H,W,Cin)).  Weights: reference layout fp32, no biases (conv3x3(bias=False));
// wd / g4 / b4 only when Cin != Cout.  x_stats: GroupNorm statistics of x (e.g. the previous block's, see
// chore_convblock_out_stats_offset) or NULL: computed here.  saved: chore_convblock_saved_bytes, kept for the backward.
int chore_convblock_fwd(chore_handle* h, int dtype, const void* x, const void* x_stats, int B, int H, int W, int Cin, int Cout,
                        const float* w1, const float* w2, const float* w3, const float* wd, const float* const* gb /*[8]: g1 b1 g2 b2 g3 b3 g4 b4*/,
                        void* y, void* saved, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    Dims d;
    if (!make_dims(d, dtype, B, H, W, Cin, Cout)) CHORE_FAIL(h, CHORE_EINVAL, "chore_convblock_fwd: unsupported shape Cin=%d Cout=%d", Cin, Cout);
    if (!x || !w1 || !w2 || !w3 || !gb || !y || !saved || !workspace || (d.down && !wd))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_convblock_fwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const Saved sv = saved_layout(d, saved);
    char* ws = (char*)workspace;
    const PackOff pk = pack_layout(d, dtype, 0, false);
    int rc;
    {   // one launch: the block's weights in fragment order, the statistics accumulators cleared
        PackJobs pj;
        pj.add(w1, ws + pk.w1, 9, Cin, d.C1, 0);
        pj.add(w2, ws + pk.w2, 9, d.C1, d.C2, 0);
        pj.add(w3, ws + pk.w3, 9, d.C2, d.C2, 0);
        if (d.down) pj.add(wd, ws + pk.wd, 1, Cin, Cout, 0);
        pj.zero = sv.sx; pj.zero_vecs = 4 * d.nb / 16;
        if ((rc = launch_pack_conv_multi(h, dtype, pj, s))) return rc;
    }
    const GroupStat* sx = (const GroupStat*)x_stats;
    if (!sx) {
        if ((rc = launch_gn_stats(h, dtype == CHORE_F16X3 ? CHORE_F32 : dtype, mkview(x, Cin, 0, Cin), B, H * W, (GroupStat*)sv.sx, s))) return rc;
        sx = (const GroupStat*)sv.sx;
    }
    const void* res = x;
    int res_cs = Cin;
    if (d.down) {      // residual = conv1x1(relu(gn4(x))), written to y and picked up from there
        ConvArgs a{};
        a.in = mkview(x, Cin, 0, Cin); a.in_st = sx; a.gamma = gb[6]; a.beta = gb[7];
        a.wpk = ws + pk.wd;
        a.out = mkview(y, Cout, 0, Cout);
        a.B = B; a.H = H; a.W = W; a.Cout = Cout;
        if ((rc = launch_conv(h, dtype, 1, a, s))) return rc;
        res = y; res_cs = Cout;
    }
    {   // conv1
        ConvArgs a{};
        a.in = mkview(x, Cin, 0, Cin); a.in_st = sx; a.gamma = gb[0]; a.beta = gb[1];
        a.wpk = ws + pk.w1;
        a.out = mkview(y, Cout, 0, d.C1); a.raw = mkview(sv.o1, d.C1, 0, d.C1); a.res = mkview(res, res_cs, 0, d.C1);
        a.B = B; a.H = H; a.W = W; a.Cout = d.C1;
        a.st_raw = (GroupStat*)sv.s1; a.st_raw_C = d.C1; a.st_raw_co = 0;
        a.st_out = (GroupStat*)sv.sy; a.st_out_C = Cout; a.st_out_co = 0;
        if ((rc = launch_conv(h, dtype, 9, a, s))) return rc;
    }
    {   // conv2
        ConvArgs a{};
        a.in = mkview(sv.o1, d.C1, 0, d.C1); a.in_st = (const GroupStat*)sv.s1; a.gamma = gb[2]; a.beta = gb[3];
        a.wpk = ws + pk.w2;
        a.out = mkview(y, Cout, d.C1, d.C2); a.raw = mkview(sv.o2, d.C2, 0, d.C2); a.res = mkview(res, res_cs, d.C1, d.C2);
        a.B = B; a.H = H; a.W = W; a.Cout = d.C2;
        a.st_raw = (GroupStat*)sv.s2; a.st_raw_C = d.C2; a.st_raw_co = 0;
        a.st_out = (GroupStat*)sv.sy; a.st_out_C = Cout; a.st_out_co = d.C1;
        if ((rc = launch_conv(h, dtype, 9, a, s))) return rc;
    }
    {   // conv3
        ConvArgs a{};
        a.in = mkview(sv.o2, d.C2, 0, d.C2); a.in_st = (const GroupStat*)sv.s2; a.gamma = gb[4]; a.beta = gb[5];
        a.wpk = ws + pk.w3;
        a.out = mkview(y, Cout, d.C1 + d.C2, d.C2); a.res = mkview(res, res_cs, d.C1 + d.C2, d.C2);
        a.B = B; a.H = H; a.W = W; a.Cout = d.C2;
        a.st_out = (GroupStat*)sv.sy; a.st_out_C = Cout; a.st_out_co = d.C1 + d.C2;
        if ((rc = launch_conv(h, dtype, 9, a, s))) return rc;
    }
    return CHORE_OK;
}

// dx (B,H,W,Cin) and the parameter gradients (chore_convblock_grad_floats, layout there) from dy (B,H,W,Cout, dense).
// x_stats: what the forward used (NULL: the ones it computed, in `saved`).
int chore_convblock_bwd(chore_handle* h, int dtype, const void* x, const void* x_stats, const void* dy, int B, int H, int W, int Cin,
                        int Cout, const float* w1, const float* w2, const float* w3, const float* wd, const float* const* gb,
                        const void* saved, void* dx, float* grads, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    Dims d;
    if (!make_dims(d, dtype, B, H, W, Cin, Cout)) CHORE_FAIL(h, CHORE_EINVAL, "chore_convblock_bwd: unsupported shape Cin=%d Cout=%d", Cin, Cout);
    if (!x || !dy || !w1 || !w2 || !w3 || !gb || !saved || !dx || !grads || !workspace || (d.down && !wd))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_convblock_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const Saved sv = saved_layout(d, const_cast<void*>(saved));
    const void* sx = x_stats ? x_stats : (const void*)sv.sx;
    const int C1 = d.C1, C2 = d.C2, HW = H * W;
    char* ws = (char*)workspace;
    // workspace: GroupNorm-backward accumulators, range cells of d(o2), d(o1) (zeroed) | range cells of dy | transposed packed
    // weights | wgrad partials | da | d(o2) | d(o1) | dx4
    char* acc1 = ws; char* acc4 = acc1 + gn_acc_bytes(B, Cin); char* acc2 = acc4 + gn_acc_bytes(B, Cin); char* acc3 = acc2 + gn_acc_bytes(B, C1);
    const bool x3 = dtype == CHORE_F16X3;
    // fp16 x 3: every gradient tensor that feeds a GEMM carries max |g| (enc_common.h, AMAX_CELLS): of dy from a reduction
    // that rides in the pack launch below, of d(o2) / d(o1) from the GroupNorm backward that writes them
    unsigned* amax_o2 = (unsigned*)(acc3 + gn_acc_bytes(B, C2));
    unsigned* amax_o1 = amax_o2 + AMAX_CELLS;
    size_t o = al(gn_acc_bytes(B, Cin) * 2 + gn_acc_bytes(B, C1) + gn_acc_bytes(B, C2) + 2 * AMAX_BYTES);
    const size_t zero_bytes = o;
    unsigned* amax_dy = (unsigned*)(ws + o); o += AMAX_BYTES;      // plain stores: outside the cleared region
    const PackOff pk = pack_layout(d, dtype, o, true);
    o = pk.end;
    char* wpart1 = ws + o; o += al(chore_conv2d_wgrad_workspace_bytes(9, B, H, W, Cin, C1));
    char* wpart2 = ws + o; o += al(chore_conv2d_wgrad_workspace_bytes(9, B, H, W, C1, C2));
    char* wpart3 = ws + o; o += al(chore_conv2d_wgrad_workspace_bytes(9, B, H, W, C2, C2));
    char* wpartd = ws + o; if (d.down) o += al(chore_conv2d_wgrad_workspace_bytes(1, B, H, W, Cin, Cout));
    WgradFinishJobs fin;             // the four ordered sums over the shares: one launch at the end of the side chain
    char* da = ws + o; o += al(d.px * (size_t)(Cin > C1 ? Cin : C1) * d.es);
    char* do2 = ws + o; o += al(d.px * C2 * d.es);
    char* do1 = ws + o; o += al(d.px * C1 * d.es);
    char* dx4 = ws + o;
    // gradient arena
    float* dw1 = grads; float* dw2 = dw1 + (size_t)9 * Cin * C1; float* dw3 = dw2 + (size_t)9 * C1 * C2;
    float* dwd = dw3 + (size_t)9 * C2 * C2;
    float* gbp = dwd + (d.down ? (size_t)Cin * Cout : 0);
    float *dg1 = gbp, *db1 = dg1 + Cin, *dg2 = db1 + Cin, *db2 = dg2 + C1, *dg3 = db2 + C1, *db3 = dg3 + C2, *dg4 = db3 + C2, *db4 = dg4 + Cin;
    const char* dyb = (const char*)dy;
    int rc;
    {   // one launch: the transposed, flipped weights of the data-gradient convolutions; the accumulators cleared
        PackJobs pj;
        pj.add(w1, ws + pk.w1, 9, C1, Cin, 1);
        pj.add(w2, ws + pk.w2, 9, C2, C1, 1);
        pj.add(w3, ws + pk.w3, 9, C2, C2, 1);
        if (d.down) pj.add(wd, ws + pk.wd, 1, Cout, Cin, 1);
        pj.zero = ws; pj.zero_vecs = zero_bytes / 16;
        if (x3) { pj.amax_x = (const float*)dy; pj.amax_n4 = d.px * (size_t)Cout / 4; pj.amax_cells = amax_dy; }
        if ((rc = launch_pack_conv_multi(h, dtype, pj, s))) return rc;
    }
    const int adt = x3 ? CHORE_F32 : dtype;          // the element type of the tensors (what the non-GEMM kernels see)
    (void)adt;
    auto dgrad = [&](int taps, const View& in, const float* w, size_t wpk_off, int cin_fwd, int cout_fwd, void* out, const unsigned* amax) -> int {
        // data gradient of a layer Cin_fwd -> Cout_fwd: the forward kernel on the transposed, flipped weights
        (void)w; (void)cout_fwd;
        ConvArgs a{};
        a.in = in;
        a.wpk = ws + wpk_off;
        a.out = mkview(out, cin_fwd, 0, cin_fwd);
        a.B = B; a.H = H; a.W = W; a.Cout = cin_fwd;
        a.in_amax = x3 ? amax : nullptr;
        return launch_conv(h, dtype, taps, a, s);
    };
    // Two chains: the data-gradient chain  dgrad3 -> gn3 -> dgrad2 -> gn2 -> [downsample] -> dgrad1 -> gn1  on the
    // caller's stream, and the four weight gradients on the handle's side stream, each released by the event after the
    // kernel that produces its dy (dy itself for conv3 and the downsample conv).  The kernels of either chain fill a
    // fraction of the chip (64 .. 256 workgroups), so the chains overlap almost completely.
    static const bool serial = getenv("CHORE_CONVBLOCK_SERIAL") != nullptr;      // A/B switch: everything on one stream
    hipStream_t s2 = s;
    if (!serial) {
        if ((rc = side_stream(h))) return rc;
        s2 = h->side;
    }
    auto release = [&](int ev) -> int {      // side stream waits for what the main stream has issued so far
        if (serial) return CHORE_OK;
        CHORE_HIP_CHECK(h, hipEventRecord(h->side_ev[ev], s));
        CHORE_HIP_CHECK(h, hipStreamWaitEvent(s2, h->side_ev[ev], 0));
        return CHORE_OK;
    };
    const int off2 = C1, off3 = C1 + C2;
    // Launch order inside a stage: the data-gradient kernels (caller's stream) BEFORE the weight-gradient kernels (side stream)
    // that the same event releases.  Eagerly it makes no difference; recorded into a hipGraph it does: the runtime lays the
    // graph out on queues by following every node's FIRST-recorded successor, so with the side kernel recorded first the
    // data-gradient chain changed queue at each of the three releases, ~10 us per change (profiles/r04_train_graph.txt).
    if ((rc = release(0))) return rc;        // dy, the workspace clear
    // ---- conv3: its output gradient is the last slice of dy ----
    if ((rc = dgrad(9, mkview(dy, Cout, off3, C2), w3, pk.w3, C2, C2, da, amax_dy))) return rc;
    if ((rc = gn_relu_bwd_impl(h, dtype, sv.o2, sv.s2, gb[4], gb[5], da, B, HW, C2, do2, dg3, db3, acc3, 1,
                               dyb + (size_t)off2 * d.es, Cout, s, x3 ? amax_o2 : nullptr))) return rc;          // + the concat's gradient of o2
    if ((rc = conv2d_bwd_weight_impl(h, dtype, 9, sv.o2, B, H, W, C2, sv.s2, gb[4], gb[5], dyb + (size_t)off3 * d.es, Cout, C2, dw3,
                                     nullptr, wpart3, s2, &fin, x3 ? amax_dy : nullptr))) return rc;
    if (d.down && (rc = conv2d_bwd_weight_impl(h, dtype, 1, x, B, H, W, Cin, sx, gb[6], gb[7], dy, Cout, Cout, dwd, nullptr, wpartd, s2, &fin,
                                               x3 ? amax_dy : nullptr)))
        return rc;
    // ---- conv2 ----
    if ((rc = release(1))) return rc;        // d(o2)
    if ((rc = dgrad(9, mkview(do2, C2, 0, C2), w2, pk.w2, C1, C2, da, amax_o2))) return rc;
    if ((rc = gn_relu_bwd_impl(h, dtype, sv.o1, sv.s1, gb[2], gb[3], da, B, HW, C1, do1, dg2, db2, acc2, 1, dyb, Cout, s,
                               x3 ? amax_o1 : nullptr))) return rc;
    if ((rc = conv2d_bwd_weight_impl(h, dtype, 9, sv.o1, B, H, W, C1, sv.s1, gb[2], gb[3], do2, C2, C2, dw2, nullptr, wpart2, s2, &fin,
                                     x3 ? amax_o2 : nullptr))) return rc;
    // ---- conv1 (and the downsample branch): both normalise x ----
    if ((rc = release(2))) return rc;        // d(o1)
    const void* skip = dy;          // identity residual: dy itself flows to x
    int skip_cs = Cout;
    if (d.down) {
        if ((rc = dgrad(1, mkview(dy, Cout, 0, Cout), wd, pk.wd, Cin, Cout, da, amax_dy))) return rc;
        if ((rc = gn_relu_bwd_impl(h, dtype, x, sx, gb[6], gb[7], da, B, HW, Cin, dx4, dg4, db4, acc4, 1, nullptr, 0, s))) return rc;
        skip = dx4; skip_cs = Cin;
    }
    if ((rc = dgrad(9, mkview(do1, C1, 0, C1), w1, pk.w1, Cin, C1, da, amax_o1))) return rc;
    if ((rc = gn_relu_bwd_impl(h, dtype, x, sx, gb[0], gb[1], da, B, HW, Cin, dx, dg1, db1, acc1, 1, skip, skip_cs, s))) return rc;
    if ((rc = conv2d_bwd_weight_impl(h, dtype, 9, x, B, H, W, Cin, sx, gb[0], gb[1], do1, C1, C1, dw1, nullptr, wpart1, s2, &fin,
                                     x3 ? amax_o1 : nullptr))) return rc;
    if ((rc = launch_wgrad_finish_multi(h, fin, s2))) return rc;
    if (!serial) {                    // join: the caller's stream continues after the weight gradients too
        CHORE_HIP_CHECK(h, hipEventRecord(h->side_ev[3], s2));
        CHORE_HIP_CHECK(h, hipStreamWaitEvent(s, h->side_ev[3], 0));
    }
    return CHORE_OK;
}

}  // extern "C"

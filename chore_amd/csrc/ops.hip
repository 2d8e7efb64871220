// ops.hip -- layer-level entry points of the encoder kernels (C ABI), used by the TRAINING path.
//
// Inference runs the whole encoder as one launch program (encoder.hip).  Training needs every intermediate tensor
// and a backward per layer, so the host composes the network from these operators instead (chore_amd/model/
// hgfilter_train.py): the same kernels -- conv_lds (forward, and the data gradient on transposed + flipped weights),
// the exact GroupNorm statistics, GroupNorm+ReLU apply -- plus the backward kernels of train_bwd.hip.
// Activations are NHWC (T = fp32 or bf16), weights arrive in the reference layout (O,C,kh,kw) fp32 and are packed
// into the caller's workspace on every call (they change every optimiser step).
#include "enc_common.h"

extern "C" {

size_t chore_conv2d_workspace_bytes(int dtype, int taps, int Cin, int Cout) {
    if ((dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3) || (taps != 1 && taps != 9)) return 0;
    return packed_conv_bytes(dtype, taps, Cin, Cout);
}

size_t chore_gn_stats_bytes(int B) { return act_stats_bytes(B); }

// Training in the fp16 x 3 mode (dtype CHORE_F16X3 of the chore_conv2d_* / chore_convblock_* entry points: fp32 tensors, every
// convolution, data gradient and weight gradient on the fp16 matrix cores with hi / lo split operands): a gradient that feeds
// one of those GEMMs has any magnitude, so its range travels with it -- chore_amax_bytes() bytes holding partial maxima of
// |x| as float bits, written by chore_absmax_f32 (or, inside chore_convblock_bwd, by the kernels that produce the gradients).
size_t chore_amax_bytes(void) { return AMAX_CELLS * sizeof(unsigned); }
int chore_absmax_f32(chore_handle* h, const float* x, size_t n, void* amax, chore_stream_t stream) {
    CHORE_ENTER(h);
    return launch_absmax_f32(h, x, n, (unsigned*)amax, (hipStream_t)stream);
}
static inline int elem_dtype(int dtype) { return dtype == CHORE_F16X3 ? CHORE_F32 : dtype; }   // what the non-GEMM kernels see

// statistics of x (B,HW,C) for GroupNorm(32, C): stats is zeroed and filled
// (zeroed != 0: the caller hands in zeroed accumulators, e.g. a slice of an arena cleared once per pass)
int chore_gn_stats(chore_handle* h, int dtype, const void* x, int B, int HW, int C, void* stats, int zeroed,
                   chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!x || !stats || B <= 0 || HW <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_gn_stats: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (!zeroed) CHORE_HIP_CHECK(h, hipMemsetAsync(stats, 0, chore_gn_stats_bytes(B), s));
    View v; v.p = const_cast<void*>(x); v.cs = C; v.co = 0; v.C = C;
    return launch_gn_stats(h, elem_dtype(dtype), v, B, HW, (GroupStat*)stats, s);
}

// y = relu(groupnorm(x))
int chore_gn_relu_fwd(chore_handle* h, int dtype, const void* x, const void* stats, const float* gamma, const float* beta,
                      void* y, int B, int HW, int C, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!x || !stats || !gamma || !beta || !y) CHORE_FAIL(h, CHORE_EINVAL, "chore_gn_relu_fwd: null argument");
    View vx; vx.p = const_cast<void*>(x); vx.cs = C; vx.co = 0; vx.C = C;
    View vy; vy.p = y; vy.cs = C; vy.co = 0; vy.C = C;
    return launch_gn_apply_relu(h, elem_dtype(dtype), vx, (const GroupStat*)stats, gamma, beta, vy, B, HW, (hipStream_t)stream);
}

// y (B,H,W,Cout) = conv_{taps}(a) + bias, a = relu(groupnorm(x)) if stats != NULL else x; stride 1, zero padding.
// out_stats (or NULL): ZEROED chore_gn_stats_bytes(B) accumulators that receive the statistics of y (of the values as
// stored) from the convolution's epilogue -- what a following GroupNorm(32, Cout) needs, without a pass over y
int chore_conv2d_fwd(chore_handle* h, int dtype, int taps, const void* x, int B, int H, int W, int Cin,
                     const void* stats, const float* gamma, const float* beta, const float* w, const float* bias,
                     int Cout, void* y, void* out_stats, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!x || !w || !y || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_fwd: null argument");
    if (stats && (!gamma || !beta)) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_fwd: GroupNorm needs gamma and beta");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_pack_conv(h, dtype, taps, Cin, Cout, w, workspace, s, 0);
    if (rc) return rc;
    ConvArgs a{};
    a.in.p = const_cast<void*>(x); a.in.cs = Cin; a.in.co = 0; a.in.C = Cin;
    a.in_st = (const GroupStat*)stats; a.gamma = gamma; a.beta = beta;
    a.wpk = workspace; a.bias = bias;
    a.out.p = y; a.out.cs = Cout; a.out.co = 0; a.out.C = Cout;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout;
    if (out_stats) { a.st_out = (GroupStat*)out_stats; a.st_out_C = Cout; a.st_out_co = 0; }
    return launch_conv(h, dtype, taps, a, s);
}

// dx (B,H,W,Cin) = gradient of the convolution's INPUT (the tensor the conv saw, after any GroupNorm+ReLU):
// the same kernel on the transposed, spatially flipped weights
// dy_amax: the range of dy (chore_absmax_f32), required with dtype CHORE_F16X3, ignored otherwise
int chore_conv2d_bwd_data(chore_handle* h, int dtype, int taps, const void* dy, int B, int H, int W, int Cout,
                          const float* w, int Cin, void* dx, void* workspace, const void* dy_amax, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!dy || !w || !dx || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_bwd_data: null argument");
    if (dtype == CHORE_F16X3 && !dy_amax) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_bwd_data: the fp16 x 3 mode needs the range of dy (dy_amax)");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_pack_conv(h, dtype, taps, /*Cin of this conv*/ Cout, /*Cout of this conv*/ Cin, w, workspace, s, 1);
    if (rc) return rc;
    ConvArgs a{};
    a.in.p = const_cast<void*>(dy); a.in.cs = Cout; a.in.co = 0; a.in.C = Cout;
    a.wpk = workspace;
    a.out.p = dx; a.out.cs = Cin; a.out.co = 0; a.out.C = Cin;
    a.B = B; a.H = H; a.W = W; a.Cout = Cin;
    if (dtype == CHORE_F16X3) a.in_amax = (const unsigned*)dy_amax;
    return launch_conv(h, dtype, taps, a, s);
}

// stem: y (B,H/2,W/2,64) = conv7x7 stride 2 pad 3 (images (B,Cin,H,W) fp32 NCHW) + bias   (HGFilters.py:102,149);
// workspace: chore_stem_workspace_bytes(Cin) for the repacked weights
size_t chore_stem_workspace_bytes(int Cin) { return Cin > 0 ? (size_t)Cin * 49 * 64 * sizeof(float) : 0; }

int chore_stem_fwd(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W, const float* w,
                   const float* bias, void* y, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!images || !w || !bias || !y || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_stem_fwd: null argument");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) CHORE_FAIL(h, CHORE_EINVAL, "chore_stem_fwd: bad shape");
    dtype = elem_dtype(dtype);
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) CHORE_FAIL(h, CHORE_EINVAL, "chore_stem_fwd: dtype");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_pack_stem(h, Cin, w, (float*)workspace, s);
    if (rc) return rc;
    return launch_stem(h, dtype, images, B, Cin, H, W, (const float*)workspace, bias, y, s);
}

// y (B,H/2,W/2,C) = 2x2 average pooling of x (B,H,W,C), C in {64,128,256}; dx = its transpose applied to dy
// out_stats (or NULL): ZEROED chore_gn_stats_bytes(B) accumulators that receive the GroupNorm statistics of y
int chore_avgpool2_fwd(chore_handle* h, int dtype, const void* x, void* y, int B, int H, int W, int C, void* out_stats,
                       chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) CHORE_FAIL(h, CHORE_EINVAL, "chore_avgpool2_fwd: bad argument");
    dtype = elem_dtype(dtype);
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) CHORE_FAIL(h, CHORE_EINVAL, "chore_avgpool2_fwd: dtype");
    View vx; vx.p = const_cast<void*>(x); vx.cs = C; vx.co = 0; vx.C = C;
    View vy; vy.p = y; vy.cs = C; vy.co = 0; vy.C = C;
    return launch_avgpool2(h, dtype, vx, vy, B, H, W, (GroupStat*)out_stats, (hipStream_t)stream);
}

int chore_avgpool2_bwd(chore_handle* h, int dtype, const void* dy, void* dx, int B, int H, int W, int C, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_avgpool2_bwd: bad argument");
    dtype = elem_dtype(dtype);
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) CHORE_FAIL(h, CHORE_EINVAL, "chore_avgpool2_bwd: dtype");
    return launch_pool2_bwd(h, dtype, dy, dx, B, H, W, C, (hipStream_t)stream);
}

// y (B,2H,2W,C) = a + bicubic_up2(low (B,H,W,C)), align_corners=True  (HourGlass._forward, HGFilters.py:47-50); y may be a
int chore_upadd_fwd(chore_handle* h, int dtype, const void* a, const void* low, void* y, int B, int H, int W, int C,
                    void* out_stats, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!a || !low || !y) CHORE_FAIL(h, CHORE_EINVAL, "chore_upadd_fwd: null argument");
    View va; va.p = const_cast<void*>(a); va.cs = C; va.co = 0; va.C = C;
    View vl; vl.p = const_cast<void*>(low); vl.cs = C; vl.co = 0; vl.C = C;
    View vy; vy.p = y; vy.cs = C; vy.co = 0; vy.C = C;
    return launch_upadd(h, elem_dtype(dtype), va, vl, vy, B, H, W, (GroupStat*)out_stats, (hipStream_t)stream);
}

// d_low (B,H,W,C) = transpose of the bicubic x2 upsampling applied to dy (B,2H,2W,C)
int chore_up2_bwd(chore_handle* h, int dtype, const void* dy, void* dlow, int B, int H, int W, int C, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!dy || !dlow) CHORE_FAIL(h, CHORE_EINVAL, "chore_up2_bwd: null argument");
    return launch_up2_bwd(h, elem_dtype(dtype), dy, dlow, B, H, W, C, (hipStream_t)stream);
}

}  // extern "C"

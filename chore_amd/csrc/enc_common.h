// enc_common.h -- kernel argument blocks and launchers of the stacked-hourglass encoder.
//
// Activations are NHWC, element type T = float (CHORE_F32) or bf16 stored as unsigned short
// (CHORE_BF16).  A "view" is (pointer, channel stride of the buffer, channel offset, channels):
// ConvBlock's concat (model/net_util.py:388) is never materialised -- each conv writes its slice.
#pragma once
#include "common.h"

typedef unsigned short bf16_t;
// IEEE half storage (CHORE_F16): a distinct type so that templates can tell it from bf16
struct h16_t { unsigned short u; };

// "fp16 x 3" element type (CHORE_F16X3): fp32 tensors in memory, every product a*w of a convolution evaluated on the fp16
// matrix cores as hi(a) hi(w) + lo(a) hi(w) + hi(a) lo(w) with hi = fp16(x), lo = fp16(x - hi) and fp32 accumulation.
// hi + lo carries 22 mantissa bits and fp16 x fp16 products are exact in fp32, so the result is fp32-grade (the dropped
// lo*lo term is 2^-22 relative) at a third of the fp16 MFMA rate = 5x the rate of the native fp32 MFMA.  The matrix
// cores keep fp16 subnormals (scripts/probes/f16_denorm_probe.hip), so small lo parts are not lost.  Weights are scaled
// by 2^X3_WSHIFT before the split (their lo parts stay normal numbers), the accumulators by 2^-X3_WSHIFT afterwards.
struct x3_t { float v; };
constexpr int X3_WSHIFT = 8;
template <typename T> struct Store { using type = T; };
template <typename T> struct IsH16 { static constexpr bool value = false; };
template <> struct IsH16<h16_t> { static constexpr bool value = true; };
template <typename T> constexpr bool IS_H16 = IsH16<T>::value;
template <> struct Store<x3_t> { using type = float; };
template <typename T> struct IsX3 { static constexpr bool value = false; };
template <> struct IsX3<x3_t> { static constexpr bool value = true; };
// x3_t whose INPUT is a gradient that is scaled into fp16's range (ConvArgs::in_amax): the data-gradient convolutions of fp16 x 3
// training.  A type of its own, so that the kernels of the inference path are compiled exactly as before the scale existed.
struct x3s_t { float v; };
template <> struct Store<x3s_t> { using type = float; };
template <> struct IsX3<x3s_t> { static constexpr bool value = true; };
template <typename T> struct IsX3S { static constexpr bool value = false; };
template <> struct IsX3S<x3s_t> { static constexpr bool value = true; };
template <typename T> constexpr bool IS_X3S = IsX3S<T>::value;
template <typename T> constexpr bool IS_X3 = IsX3<T>::value;

struct View {
    void* p = nullptr;
    int cs = 0;   // channel stride (channels of the underlying buffer)
    int co = 0;   // channel offset of this view
    int C = 0;    // channels in the view
};

constexpr int GN_GROUPS = 32;
constexpr int GN_SPLITS_MAX = 64;

// ---- exact, order-independent GroupNorm statistics -------------------------------------------------
// Every producer adds its partial sums per (image, GroupNorm group) into a two-limb fixed-point accumulator
// (unit 2^-40) with 64-bit integer atomics.  (Per group, not per channel: device-scope atomics execute at the
// memory side and their COUNT is what a producer kernel pays for -- a per-channel version spent 10-30 % of
// every producer on them.)  Integer addition is associative, so the totals -- and everything
// derived from them -- are bit-identical whatever order the workgroups run in, across runs and batch
// compositions; no finalize kernel and no per-tile partial buffers are needed.  A consumer turns the
// totals into the per-(image,channel) affine  relu(x*scale+shift)  in its own prologue.
struct StatCell { unsigned long long lo; long long hi; };
struct GroupStat { StatCell sum, sq; };   // per (image, group): sum and sum of squares of the stored values
// value = (hi * 2^32 + lo) * 2^-40 with lo a sum of 32-bit pieces: no carry between the limbs, so an add is two
// NO-RETURN atomics and the wave does not wait for a round trip to the memory side.  (The first version kept a proper
// 128-bit integer: its low add had to return the old value for the carry and every producer kernel ended with ~5 us of
// waiting for those returns.)  lo cannot overflow before 2^32 adds.
// The two limbs of a cell must NOT sit next to each other: two atomics from one lane into the same 16 bytes queue behind
// each other at the memory side (measured: + 9 us on a 47 us convolution, profiles/r03_conv_phase_breakdown.txt).  Every
// accumulator array of n cells is therefore allocated as TWO tables of n cells, the low limbs are added into table 0 and
// the high limbs into the same cell of table 1, `hi_cells` (>= n, a multiple of 256 cells = 4 KB where it matters) later.
__device__ __forceinline__ void stat_add(StatCell* c, size_t hi_cells, float x) {
    const double d = (double)x * 0x1p40;             // exact (power-of-two scaling)
    const double h = floor(d * 0x1p-32);             // exact split of the <= 53 significant bits
    const long long hi = (long long)h;
    const unsigned long long lo = (unsigned long long)(d - h * 0x1p32);   // in [0, 2^32)
    (void)__hip_atomic_fetch_add(&c->lo, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hi != 0) (void)__hip_atomic_fetch_add(&c[hi_cells].hi, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double stat_read(const StatCell* c, size_t hi_cells) {
    const unsigned long long lo = c->lo;
    const long long top = c[hi_cells].hi + (long long)(lo >> 32);
    return ((double)top * 0x1p32 + (double)(lo & 0xffffffffull)) * 0x1p-40;
}
// statistics of an activation tensor: [2 tables][B][32] GroupStat
inline size_t act_stats_bytes(int B) { return (size_t)2 * B * GN_GROUPS * sizeof(GroupStat); }
__host__ __device__ inline size_t act_hi_cells(int B) { return (size_t)B * GN_GROUPS * 2; }
// sum over the gs (power of two <= 8) consecutive channels of a group held by gs consecutive lanes: fixed tree
__device__ __forceinline__ float group_lane_sum(float v, int gs) {
    for (int o = 1; o < gs; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// scale/shift of channel c of image b: GroupNorm(32 groups, eps 1e-5, biased variance) folded to an affine
__device__ __forceinline__ void gn_scale_shift(const GroupStat* st, int B, int b, int C, int c, int HW,
                                               const float* gamma, const float* beta, float& scale, float& shift) {
    const int gs = C / GN_GROUPS;
    const GroupStat* g = st + (size_t)b * GN_GROUPS + c / gs;
    const double ssum = stat_read(&g->sum, act_hi_cells(B)), ssq = stat_read(&g->sq, act_hi_cells(B));
    const double n = (double)HW * gs;
    const double mean = ssum / n;
    double var = ssq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
    scale = rstd * gamma[c];
    shift = beta[c] - (float)mean * scale;
}

// ---- range of a gradient tensor (fp16 x 3 training) ------------------------------------------------------------------
// AMAX_CELLS words, each the float bits of max |x| over a part of the tensor (0 = untouched).  Written either with plain stores
// (a reduction launch with exactly AMAX_CELLS workgroups) or with integer atomic max into zeroed cells (a producer's epilogue);
// a reader wave takes the maximum of all cells (4 loads per lane + 6 cross-lane steps, no LDS, no barrier).
constexpr int AMAX_CELLS = 256;
#ifdef __HIPCC__
__device__ __forceinline__ unsigned amax_read(const unsigned* cells) {
    const int lane = threadIdx.x & 63;
    unsigned a = cells[lane], b = cells[lane + 64], c = cells[lane + 128], d = cells[lane + 192];
    a = a > b ? a : b; c = c > d ? c : d;
    unsigned v = a > c ? a : c;
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
    return v;
}
// power-of-two operand scale `mul` (max |x| * mul in [2^13, 2^14)) and its inverse from the cells; 1, 1 without cells or when the
// maximum is zero / tiny / not finite
__device__ __forceinline__ void x3_in_scale(const unsigned* cells, float& mul, float& inv) {
    mul = 1.f; inv = 1.f;
    if (!cells) return;
    // (readfirstlane: the value is wave-uniform; held in scalar registers it costs the convolutions no vector register)
    const unsigned e = ((unsigned)__builtin_amdgcn_readfirstlane((int)amax_read(cells)) >> 23) & 0xffu;       // biased exponent of max |x|
    if (e >= 14u && e <= 240u) { mul = __uint_as_float((267u - e) << 23); inv = __uint_as_float((e - 13u) << 23); }
}
// block-wide max of the float bits `v` (non-negative floats compare like their bits) -> one atomic max into cells[cell]
__device__ __forceinline__ void amax_block_atomic(unsigned v, unsigned* cells, int cell, unsigned* lds4 /*[nwaves]*/) {
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) lds4[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) v = v > lds4[i] ? v : lds4[i];
        if (v) (void)__hip_atomic_fetch_max(&cells[cell], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the work of workgroup `blk` (of AMAX_CELLS, 256 threads) of a max-|x| reduction over n4 16-byte vectors: eight loads in flight
// per thread, the workgroup's maximum stored (plain store) into cells[blk]
__device__ __forceinline__ void absmax_block(const float* __restrict__ x, size_t n4, unsigned* __restrict__ cells, unsigned blk) {
    __shared__ unsigned amax_red[4];
    const u32x4* p = (const u32x4*)x;
    unsigned m = 0u;
    const size_t stride = (size_t)AMAX_CELLS * 256;
    size_t i = (size_t)blk * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const unsigned b = v[j][k] & 0x7fffffffu; m = b > m ? b : m; }
    }
    for (; i < n4; i += stride) {
        const u32x4 v = p[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const unsigned b = v[k] & 0x7fffffffu; m = b > m ? b : m; }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)m, o, 64); m = m > w ? m : w; }
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) m = m > amax_red[k] ? m : amax_red[k];
        cells[blk] = m;
    }
}
#endif
// max |x| of n fp32 values (n a multiple of 4, 16-byte aligned) -> AMAX_CELLS cells, plain stores, one launch (enc_misc.hip)
int launch_absmax_f32(chore_handle* h, const float* x, size_t n, unsigned* cells, hipStream_t s);

struct ConvArgs {
    View in;                 // input activations (a whole tensor when GroupNorm is fused: co = 0, C = cs)
    const GroupStat* in_st;  // [B][32] statistics of the input, or null: no GroupNorm+ReLU prologue
    const float* gamma;      // [Cin] GroupNorm affine of the fused prologue
    const float* beta;
    const void* wpk;         // fragment-ordered weights (launch_pack_conv)
    const float* bias;       // [Cout] or null
    View out;                // acc + bias + res + res2
    View raw;                // optional: acc + bias
    View res, res2;          // optional residuals (may alias out)
    int B, H, W, Cout;
    // optional statistics of what this launch stores (for the next GroupNorm): [B][32] accumulators of
    // the tensor `raw` / `out` belong to; *_C = channels of that tensor (group size = C/32), *_co = offset of this slice
    GroupStat* st_raw = nullptr; int st_raw_C = 0, st_raw_co = 0;
    GroupStat* st_out = nullptr; int st_out_C = 0, st_out_co = 0;
    // fp16 x 3 only: range of the INPUT when it is a gradient (the data-gradient convolutions of training, whose operand has no
    // GroupNorm in front and any magnitude): AMAX_CELLS partial maxima of |x| as float bits (absmax_* in enc_common.h).  The
    // kernel multiplies the operand by the power of two that brings the maximum to 2^13 .. 2^14 before the hi / lo split (fp16
    // keeps 22 bits of a pair only above 2^-3) and the accumulators by its inverse.  NULL: operand taken as is.
    const unsigned* in_amax = nullptr;
    // minimum number of workgroups the tiling of a small layer must yield (conv_mw_plan): 0 = a tile per CU (256).  The inference
    // encoder passes CONV_MW_FILL_INFER: half the CUs per launch, the other half for the other step in flight; training's operators
    // leave 0 (measured: 29.15 against 29.55 ms per step with the dense tiles, profiles/r06_conv_fill.txt)
    int fill = 0;
    unsigned long long* dbg_ticks = nullptr;   // CHORE_CONV_ABLATE builds: phase time stamps of workgroup 0 (conv_pc.hip)
    int dbg = 0;   // ablation bits for kernel experiments (CHORE_CONV_DBG): 1 no weight loads, 2 no patch
                   // prefetch, 4 no MFMA, 8 no epilogue -- results are wrong when set
};

// four consecutive channels of a T tensor <-> four floats (device code only)
#ifdef __HIPCC__
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *(f32x4*)p = v; }
    static __device__ __forceinline__ f32x4 st_round(float* p, f32x4 v) { *(f32x4*)p = v; return v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ f32x4 ld(const bf16_t* p) {
        const u16x4 v = *(const u16x4*)p;
        f32x4 r = {bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
        return r;
    }
    static __device__ __forceinline__ void st(bf16_t* p, f32x4 v) {
        const unsigned lo = pack2bf(v[0], v[1]), hi = pack2bf(v[2], v[3]);
        *(unsigned long long*)p = (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    // store and return the values as stored (rounded to bf16)
    static __device__ __forceinline__ f32x4 st_round(bf16_t* p, f32x4 v) {
        const unsigned lo = pack2bf(v[0], v[1]), hi = pack2bf(v[2], v[3]);
        *(unsigned long long*)p = (unsigned long long)lo | ((unsigned long long)hi << 32);
        f32x4 r = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                   __uint_as_float(hi & 0xffff0000u)};
        return r;
    }
};
template <> struct Vec4<h16_t> {
    typedef _Float16 hx4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 ld(const h16_t* p) {
        const hx4 v = *(const hx4*)p;
        f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        return r;
    }
    static __device__ __forceinline__ void st(h16_t* p, f32x4 v) {
        const hx4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *(hx4*)p = o;
    }
    static __device__ __forceinline__ f32x4 st_round(h16_t* p, f32x4 v) {
        const hx4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *(hx4*)p = o;
        f32x4 r = {(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
        return r;
    }
};
#endif

struct ConvPlan { int nt, th, ntiles, tps, small_cin; };   // N tile, tile height, tiles per image, taps per K-step (tps 0: conv_small_kernel)
// small maps (conv_small.hip): one workgroup = 32 pixels of a row x 32 output channels, K split over its four waves
bool conv_small_eligible(int dtype, int taps, int H, int W, int Cin, int Cout);
int launch_conv_small(chore_handle* h, int dtype, const ConvArgs& a, hipStream_t s);
ConvPlan conv_plan(int dtype, int taps, int B, int H, int W, int Cin, int Cout);   // tile configuration launch_conv will use

// specialised-wave convolution (conv_pc.hip, fp16 x 3 operands): th rows x 32 pixels x nt channels per workgroup, tps taps per
// K-step, nslot K-steps of weights resident in LDS; th = 0: the layer is not covered and launch_conv uses conv_lds_kernel
struct PcPlan { int th, nt, tps, nslot; };
PcPlan conv_pc_plan(int dtype, int taps, int B, int H, int W, int Cin, int Cout, int force = 0);
int launch_conv_pc(chore_handle* h, int dtype, int taps, const PcPlan& p, const ConvArgs& a, hipStream_t s);
bool conv_use_pc();
// the same tilings with the staging work inside the MFMA-issuing waves (conv_mw.hip, round 6): four waves, one per SIMD, no producers
constexpr int CONV_MW_FILL_INFER = 128;
int conv_mw_fill(int asked);            // the `fill` a launch gets: CHORE_CONV_MW_FILL if set (every launch), else asked (0 -> 256)
bool conv_mw_on(int dtype, int taps);   // by mode: fp16 x 3, 3x3, not switched off (CHORE_CONV_MW=0)
PcPlan conv_mw_plan(int dtype, int taps, int B, int H, int W, int Cin, int Cout, int fill);   // conv_pc_plan's tiling, or one only this kernel has
bool conv_mw_has(const PcPlan& p);      // an instantiation for this tiling exists
bool conv_mw_covers(int dtype, int taps, const PcPlan& p, const ConvArgs& a);
int launch_conv_mw(chore_handle* h, int dtype, int taps, const PcPlan& p, const ConvArgs& a, hipStream_t s);
// 1x1 layers (fp16 x 3, fp16, bf16) with register-resident weights (conv_rw.hip): persistent workgroups over runs of pixel blocks
bool conv_rw_covers(int dtype, int taps, int Cin, int Cout, bool scaled_input);   // by shape (scaled_input: ConvArgs::in_amax set)
bool conv_rw_eligible(int dtype, int taps, const ConvArgs& a);
int launch_conv_rw(chore_handle* h, int dtype, const ConvArgs& a, hipStream_t s);
int launch_conv(chore_handle* h, int dtype, int taps /*1|9*/, const ConvArgs& a, hipStream_t s);
size_t packed_conv_bytes(int dtype, int taps, int Cin, int Cout);
// up to MAXJ weight packs and one region to clear (16-byte multiples), one launch (conv_lds.hip)
struct PackJobs {
    static constexpr int MAXJ = 4;
    struct Job { const float* w; void* dst; int taps, Cin, Cout, transposed; size_t nvec; unsigned blocks; } job[MAXJ];
    int n = 0;
    void* zero = nullptr; size_t zero_vecs = 0;
    const float* amax_x = nullptr; size_t amax_n4 = 0; unsigned* amax_cells = nullptr;   // optional: max |x| of a tensor (AMAX_CELLS more workgroups)
    void add(const float* w, void* dst, int taps, int Cin, int Cout, int transposed) {
        job[n].w = w; job[n].dst = dst; job[n].taps = taps; job[n].Cin = Cin; job[n].Cout = Cout; job[n].transposed = transposed;
        ++n;
    }
};
int launch_pack_conv_multi(chore_handle* h, int dtype, PackJobs& j, hipStream_t s);
int launch_pack_conv(chore_handle* h, int dtype, int taps, int Cin, int Cout, const float* w /*(O,C,k,k)*/,
                     void* dst, hipStream_t s, int transposed = 0);

// --- misc kernels (enc_misc.hip) ---
int launch_stem(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W,
                const float* wk /*[Cin*49][64]*/, const float* bias, void* out /*(B,H/2,W/2,64)*/, hipStream_t s);
int launch_pack_stem(chore_handle* h, int Cin, const float* w /*(64,Cin,7,7)*/, float* dst, hipStream_t s);
// the stem on the matrix cores (fp16 x 3 mode, enc_misc.hip): fragment-ordered weights (stem_x3_bytes()) after the fp32 pack
size_t stem_x3_bytes();
bool stem_x3_on(int Cin);
int launch_pack_stem_x3(chore_handle* h, int Cin, const float* w, void* dst, hipStream_t s);
int launch_stem_x3(chore_handle* h, const float* images, int B, int Cin, int H, int W, const void* wfr, const float* bias, float* out,
                   hipStream_t s);
// statistics of a tensor no convolution produced (pooling / upsampling / stem outputs): one pass, atomics
int launch_gn_stats(chore_handle* h, int dtype, const View& x, int B, int HW, GroupStat* st, hipStream_t s);
// train_bwd.hip: the layer backward pieces with channel-strided gradients (what a ConvBlock's concat hands its convs)
int gn_relu_bwd_impl(chore_handle* h, int dtype, const void* x, const void* stats, const float* gamma, const float* beta,
                     const void* da, int B, int HW, int C, void* dx, float* dgamma, float* dbeta, void* workspace,
                     int workspace_zeroed, const void* extra, int extra_cs, hipStream_t s, unsigned* amax_out = nullptr);
// (amax_out: AMAX_CELLS zeroed cells that receive max |dx| -- fp16 x 3 training, the operand range of the GEMMs that read dx)
// the ordered sums over the shares' partials of up to four weight gradients, deferred into one launch (a ConvBlock's)
struct WgradFinishJobs {
    int n = 0;
    struct J { const float* part; const float* part_bias; float* dw; float* dbias; int S, Cout, Cin, taps, ct; } j[4];
};
// defer: the partial-sum kernel only; the job is appended and launch_wgrad_finish_multi finishes all of them (every job
// needs its own workspace then)
int conv2d_bwd_weight_impl(chore_handle* h, int dtype, int taps, const void* x, int B, int H, int W, int Cin,
                           const void* stats, const float* gamma, const float* beta, const void* dy, int dy_stride, int Cout,
                           float* dw, float* dbias, void* workspace, hipStream_t s, WgradFinishJobs* defer = nullptr,
                           const unsigned* dy_amax = nullptr);
int launch_wgrad_finish_multi(chore_handle* h, const WgradFinishJobs& jobs, hipStream_t s);
// y = relu(groupnorm(x)) with the affine derived from `st` (stem bn1 -> tmpx)
// st_out: the statistics of y accumulated in the same launch (zeroed cells)
int launch_gn_apply_relu(chore_handle* h, int dtype, const View& x, const GroupStat* st, const float* gamma,
                         const float* beta, const View& y, int B, int HW, hipStream_t s, GroupStat* st_out = nullptr);
int launch_avgpool2(chore_handle* h, int dtype, const View& x, const View& y, int B, int H, int W, GroupStat* st, hipStream_t s);
// y = a + bicubic_up2(low)   (low is (B,H,W,C), a and y are (B,2H,2W,C); y may alias a)
int launch_upadd(chore_handle* h, int dtype, const View& a, const View& low, const View& y, int B, int H, int W,
                 GroupStat* st, hipStream_t s);
int launch_copy_f32(chore_handle* h, const float* src, float* dst, size_t n, hipStream_t s);
int launch_pool2_bwd(chore_handle* h, int dtype, const void* dy /*(B,H/2,W/2,C)*/, void* dx /*(B,H,W,C)*/, int B, int H, int W, int C,
                     hipStream_t s);
int launch_up2_bwd(chore_handle* h, int dtype, const void* dy /*(B,2H,2W,C)*/, void* dlow /*(B,H,W,C)*/, int B, int H, int W, int C,
                   hipStream_t s);

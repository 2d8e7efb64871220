// query_bwd.hip -- backward of the fused query with respect to the POINTS (gfx950, exact fp32).
//
// What recon/generator.py:62-77 (`df.sum().backward()` w.r.t. the samples) and every fitting loss
// of recon/recon_fit_base.py need: parameters and feature maps are frozen, gradients flow
//   dOut -> heads (W^T GEMMs, ReLU masks) -> d(323-vector) -> { xyz channels directly,
//   bilinear taps -> d(ix,iy) -> projection Jacobian (model/camera.py:64-78) } -> dpoints.
// Nothing is saved by the forward: this kernel recomputes gather + hidden layers (keeping only the
// ReLU sign bits, 12 registers) and then runs the transposed chain with the same register-resident
// MFMA scheme (D fragment of one GEMM = B fragment of the next).  The four heads' contributions to
// d(323-vector) are reduced through LDS in a fixed order, so the result is deterministic.
#include "heads_x3.h"
#include <cstdlib>

template <int PTS>
struct QueryBwdSmemT {
    float X[PTS * XS];             // forward: feature tile; backward: d(feature) tile
    float P[2][HEAD_NUM][32 * PTS];   // per-head partial of one 32-row block, [row][pt]; two buffers: one barrier per block
    PtTableT<PTS> tab;
    float dfk[PTS];                // SURF: the clamped distance of every point
};

template <int NCB>
__device__ __forceinline__ unsigned sign_mask(const f32x16 (&c)[NCB]) {
    unsigned m = 0;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m |= (c[cb][r] > 0.f ? 1u : 0u) << (16 * cb + r);
    // pin the 32 bits HERE: left alone the compiler sinks these compares to where the mask is applied, a whole forward and
    // half a backward chain later, and keeps (spills) the 64 activation registers until then
    asm volatile("" : "+v"(m));
    return m;
}
template <int NCB>
__device__ __forceinline__ void apply_mask(f32x16 (&d)[4][NCB], const unsigned (&m)[4]) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[rb][cb][r] = ((m[rb] >> (16 * cb + r)) & 1u) ? d[rb][cb][r] : 0.f;
}

// d_prev = W^T * d_cur for a 128x128 layer; `which` = 0 -> W3 (3rd conv), 1 -> W2
template <int NCB>
__device__ __forceinline__ void bwd_hid(f32x16 (&out)[4][NCB], const f32x16 (&in)[4][NCB], const float* arena,
                                        int head, int which, int lane) {
    const f32x4* A = (const f32x4*)(arena + QB_OFF_L32T) + (((size_t)head * 2 + which) * 16 * 4) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[rb][cb][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 a[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) a[rb] = A[((kb * 4 + rg) * 4 + rb) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = MFMA_F32(a[rb][i], in[kb][cb][rg * 4 + i], out[rb][cb]);
            }
        }
    }
}

// masks of one hidden layer from the staged ReLU outputs ([point][128] rows of this head): bit 16 cb + r of m[rb]
template <int NCB>
__device__ __forceinline__ void load_masks(unsigned (&m)[4], const float* base, size_t row0, int n0, int N, int lane, int pt0) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) m[rb] = 0u;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int pt = pt0 + cb * 32 + col;
        if (n0 + pt >= N) continue;
        const float* row = base + (row0 + pt) * HEAD_HID;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = *(const f32x4*)(row + rb * 32 + 8 * j + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[rb] |= (v[e] > 0.f ? 1u : 0u) << (16 * cb + 4 * j + e);
            }
    }
}

// TRAIN: stage what the parameter gradients need.  STAGED (training only): the forward already staged the 323-vectors
// and the ReLU outputs (chore_query_fwd_train), so nothing is recomputed -- the ReLU masks are read back instead.
// NW = 8 (NCB = 1): two waves per head, one 32-point column block each, of a 64-point tile -- two waves per SIMD hide
// each other's weight / tap fetches (see query_fwd_f32_w8_kernel)
// X3: the GEMM chain on the fp16 matrix cores with hi/lo split operands and per-point column scales (heads_x3.h)
// SURF: one projection step of the surface generator (generator.py:50-79) in this launch: only the distance head runs, its
// output decides its own upstream gradient (1 where df_k <= thr, the gradient of sum(clamp(df_k, max = thr))), and the
// last lane writes the moved point  p - normalize(grad) * min(df_k, thr)  instead of the gradient.  Bit for bit what
// chore_query_fwd -> chore_gen_clamp_mask -> chore_query_bwd_points -> chore_gen_surface_step produce, without the second
// gather and the forward pass the backward recomputes anyway.
template <typename T, bool TRAIN, int NCB_ = 2, bool STAGED = false, int NW = 4, bool X3 = false, bool SURF = false, bool ONE = false>
__global__ __launch_bounds__(NW * 64, 1) void query_bwd_f32_kernel(QueryArgs a) {
    static_assert(!STAGED || TRAIN, "STAGED is a training mode");
    static_assert(!SURF || (X3 && !TRAIN && NW == 4), "the surface step exists for the fp16 x 3 recompute kernels");
    static_assert(!X3 || !TRAIN || STAGED, "fp16 x 3 training reads the staged forward");
    static_assert(NW == 4 || (NW == 8 && NCB_ == 1), "eight waves = two column blocks of one 32-point block each");
    static_assert(!ONE || (X3 && !TRAIN && NW == 4 && NCB_ == 2), "the one-head variant: fp16 x 3 recompute kernel, 64-point tiles");
    // ONE (round 6): exactly one head has an upstream gradient (the surface step: the distance head; a fit phase that hands over one
    // gradient: QueryArgs::one_head).  Instead of one busy wave and three that only gather and reduce, the head's chain runs on TWO
    // waves, one 32-point column block each (waves 0 / 1 sit on different SIMDs); same products, same order per element.
    constexpr int PTS = 32 * NCB_ * (NW / 4), NT_ = NW * 64;
    constexpr int NCB = ONE ? 1 : NCB_;              // column blocks per WAVE
    static_assert(!TRAIN || PTS == 64, "the training staging is written for 64-point tiles");
    if constexpr (X3) f16_saturate_mode();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    QueryBwdSmemT<PTS>& sm = *reinterpret_cast<QueryBwdSmemT<PTS>*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    int b, tile_;
    query_block(b, tile_);
    const int n0 = tile_ * PTS;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};

    if (tid < PTS)
        fill_pt_table(sm.tab, tid, a.points, a.crop_center, b, n0 + tid, a.N, cam, a.FH, a.FW, a.TH, a.TW,
                      nullptr);
    __syncthreads();
    const T* feat_b = (const T*)a.feat + (size_t)b * a.FH * a.FW * FEAT_C;
    const T* tmpx_b = (const T*)a.tmpx + (size_t)b * a.TH * a.TW * TMPX_C;
    if constexpr (!STAGED) {
        gather_tile<T, PTS, NW>(sm.X, sm.tab, feat_b, tmpx_b, wid, lane);
        __syncthreads();
    }

    const float* arena = (const float*)a.arena;
    const int head = ONE ? (SURF ? 0 : a.one_head) : (wid & 3);         // this wave's head and first point of its column blocks
    const int pt0 = ONE ? (wid & 1) * 32 : (wid >> 2) * 32 * NCB;
    const int odim = head_out_dim(head);
    const size_t row0 = (size_t)b * a.N + n0;                    // first point of the tile in the [B*N] staging rows
    const size_t plane = (size_t)a.B * a.N * HEAD_HID;           // one (layer, head) plane of tH / tdZ
    if constexpr (TRAIN && !STAGED) {
        for (int i = tid; i < PTS * (QF_KPAD / 4); i += NT_) {
            const int pt = i / (QF_KPAD / 4), q = i % (QF_KPAD / 4);
            if (n0 + pt < a.N) *(f32x4*)(a.tX + (row0 + pt) * QF_KPAD + 4 * q) = *(const f32x4*)(sm.X + pt * XS + 4 * q);
        }
    }

    // A head without an upstream gradient contributes exact zeros: its wave skips the GEMM chain and the weight traffic that
    // goes with it (1.3 MB of fragments per tile and head) and only keeps the barriers.  The generator's projection steps
    // and most fit phases hand over one or two of the four gradients.
    const bool active = ONE ? wid < 2 : (TRAIN || (SURF ? head == 0 : a.g[head] != nullptr));
    // ---- forward recompute, keep ReLU sign bits only (STAGED: read them back) ----
    unsigned m1[4], m2[4], m3[4];
    f32x16 u[4][NCB], v[4][NCB];
    float cs[NCB], unscale[NCB];           // X3: the columns' scales, and 1 / (2^s cs) for what leaves the chain
    if (active) {
    if constexpr (STAGED) {
        const size_t mplane = (size_t)a.B * a.N * 2;      // the forward's sign bits (heads_f32.h, store_masks): 16 B per point
        load_mask_bits<NCB>(m1, a.tM + (0 * HEAD_NUM + head) * mplane, row0, n0, a.N, lane, pt0);
        load_mask_bits<NCB>(m2, a.tM + (1 * HEAD_NUM + head) * mplane, row0, n0, a.N, lane, pt0);
        load_mask_bits<NCB>(m3, a.tM + (2 * HEAD_NUM + head) * mplane, row0, n0, a.N, lane, pt0);
    } else if constexpr (X3) {      // the accumulators carry a positive scale: same sign bits
        constexpr int PF = NW == 8 ? 1 : QX_PF;     // two waves per SIMD: 256 registers each, no room for a deeper ring
        heads_layer1_x3<NCB, PF>(u, sm.X + pt0 * XS, arena, head, lane);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) m1[rb] = sign_mask<NCB>(u[rb]);
        heads_layer_hid_x3<NCB, PF>(v, u, arena, head, 1, lane);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) m2[rb] = sign_mask<NCB>(v[rb]);
        heads_layer_hid_x3<NCB, PF>(u, v, arena, head, 2, lane);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) m3[rb] = sign_mask<NCB>(u[rb]);
    } else {
    heads_layer1<NCB>(u, sm.X + pt0 * XS, arena, head, lane);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) m1[rb] = sign_mask<NCB>(u[rb]);
    if constexpr (TRAIN) store_tile<NCB>(a.tH + (0 * HEAD_NUM + head) * plane, u, true, row0, n0, a.N, lane, pt0);
    heads_layer_hid<NCB>(v, u, arena, head, 1, lane);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) m2[rb] = sign_mask<NCB>(v[rb]);
    if constexpr (TRAIN) store_tile<NCB>(a.tH + (1 * HEAD_NUM + head) * plane, v, true, row0, n0, a.N, lane, pt0);
    heads_layer_hid<NCB>(u, v, arena, head, 2, lane);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) m3[rb] = sign_mask<NCB>(u[rb]);
    if constexpr (TRAIN) store_tile<NCB>(a.tH + (2 * HEAD_NUM + head) * plane, u, true, row0, n0, a.N, lane, pt0);
    }

    // ---- d3 = W4^T * dOut ----
    if constexpr (X3) {
        const float* g = a.g[head];
        float g8[NCB][8];
        f32x16 osurf[SURF ? NCB : 1];
        if constexpr (SURF) heads_layer_out_x3<NCB>(osurf, u, arena, head, lane);      // u: the third hidden layer, still whole
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int pt = pt0 + cb * 32 + col;
            const int n = n0 + pt;
            const bool live = (SURF || g != nullptr) && (n < a.N) && !(head == 0 && sm.tab.in_img[pt] == 0);
            float m = 0.f;
            float gsel = 0.f;
            if constexpr (SURF) {       // rows 0 / 1 of the output (the two distances) sit in registers 0 / 1 of the lower half
                const float dfv = a.surf_k ? osurf[cb][1] : osurf[cb][0];
                if (half == 0) sm.dfk[pt] = fminf(sm.tab.in_img[pt] ? dfv : 5.0f, a.surf_thr);     // (outside the image: OUT_DIST)
                gsel = (live && half == 0 && dfv <= a.surf_thr) ? 1.f : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * half + j;
                if constexpr (SURF) g8[cb][j] = (j == a.surf_k) ? gsel : 0.f;
                else g8[cb][j] = (live && k < odim) ? g[((size_t)b * odim + k) * a.N + n] : 0.f;
                m = fmaxf(m, fabsf(g8[cb][j]));
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            cs[cb] = col_scale_for(m);
#pragma unroll
            for (int j = 0; j < 8; ++j) g8[cb][j] *= cs[cb];
        }
        constexpr int BPF = NW == 8 ? 1 : QX_PF;     // as in the forward recompute: the depth of the ring by the registers per wave
        bwd_out_x3<NCB>(v, g8, arena, head, lane);
        apply_mask<NCB>(v, m3);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) unscale[cb] = QX_INV / cs[cb];
        if constexpr (TRAIN) store_tile<NCB>(a.tdZ + (2 * HEAD_NUM + head) * plane, v, false, row0, n0, a.N, lane, pt0, unscale);
        bwd_hid_x3<NCB, BPF>(u, v, cs, arena, head, 0, lane);  // d2 = W3^T d3
        apply_mask<NCB>(u, m2);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) unscale[cb] = QX_INV / cs[cb];
        if constexpr (TRAIN) store_tile<NCB>(a.tdZ + (1 * HEAD_NUM + head) * plane, u, false, row0, n0, a.N, lane, pt0, unscale);
        bwd_hid_x3<NCB, BPF>(v, u, cs, arena, head, 1, lane);  // d1 = W2^T d2
        apply_mask<NCB>(v, m1);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) unscale[cb] = QX_INV / cs[cb];
        if constexpr (TRAIN) store_tile<NCB>(a.tdZ + (0 * HEAD_NUM + head) * plane, v, false, row0, n0, a.N, lane, pt0, unscale);
    } else {
    // ---- d3 = W4^T * dOut  (K = 32 padded output rows, k = 2*s + half) ----
    {
        const float* g = a.g[head];
        float gb[NCB][16];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int pt = pt0 + cb * 32 + col;
            const int n = n0 + pt;
            const bool live = (g != nullptr) && (n < a.N) && !(head == 0 && sm.tab.in_img[pt] == 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = 2 * s + half;
                gb[cb][s] = (live && k < odim) ? g[((size_t)b * odim + k) * a.N + n] : 0.f;
            }
        }
        const f32x4* A = (const f32x4*)(arena + QB_OFF_L4T) + ((size_t)head * 16) * 64 + lane;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[rb][cb][r] = 0.f;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            f32x4 aw[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) aw[rb] = A[(sg * 4 + rb) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) v[rb][cb] = MFMA_F32(aw[rb][i], gb[cb][sg * 4 + i], v[rb][cb]);
            }
        }
    }
    apply_mask<NCB>(v, m3);
    if constexpr (TRAIN) store_tile<NCB>(a.tdZ + (2 * HEAD_NUM + head) * plane, v, false, row0, n0, a.N, lane, pt0);
    bwd_hid<NCB>(u, v, arena, head, 0, lane);  // d2 = W3^T d3
    apply_mask<NCB>(u, m2);
    if constexpr (TRAIN) store_tile<NCB>(a.tdZ + (1 * HEAD_NUM + head) * plane, u, false, row0, n0, a.N, lane, pt0);
    bwd_hid<NCB>(v, u, arena, head, 1, lane);  // d1 = W2^T d2
    apply_mask<NCB>(v, m1);
    if constexpr (TRAIN) store_tile<NCB>(a.tdZ + (0 * HEAD_NUM + head) * plane, v, false, row0, n0, a.N, lane, pt0);

    }

    }   // active
    // ---- dX = sum_heads W1^T d1, one 32-row block at a time, fixed-order reduction through LDS ----
    __syncthreads();  // every wave is done reading X as the forward tile
    const f32x4* A1 = (const f32x4*)(arena + QB_OFF_L1T) + ((size_t)head * 16 * QB_RB1) * 64 + lane;
    u32x4 d1h[X3 ? 8 : 1][NCB], d1l[X3 ? 8 : 1][NCB];
    L1TFrag wcur[1];
    if constexpr (X3) {
        if (active) {
            bwd_l1_frags_x3<NCB>(d1h, d1l, v, cs);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) unscale[cb] = QX_INV / cs[cb];
            load_l1t_x3(wcur[0], arena, head, 0, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll 1
    for (int rb = 0; rb < QB_RB1; ++rb) {
        f32x16 dx[NCB];
        if (!active) {           // this head's slice of the reduction buffer: zeros, written once
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[cb][r] = 0.f;
        } else if constexpr (X3) {
            L1TFrag wnext;
            if constexpr (NW != 8) {      // (two waves per SIMD: no registers for the second set, the other wave covers the wait)
                load_l1t_x3(wnext, arena, head, rb + 1 < QB_RB1 ? rb + 1 : rb, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            bwd_l1_block_x3<NCB>(dx, d1h, d1l, wcur[0]);
            if constexpr (NW != 8) wcur[0] = wnext;
            else if (rb + 1 < QB_RB1) load_l1t_x3(wcur[0], arena, head, rb + 1, lane);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[cb][r] *= unscale[cb];
        } else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[cb][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 aw = A1[((kb * 4 + rg) * QB_RB1 + rb) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) dx[cb] = MFMA_F32(aw[i], v[kb][cb][rg * 4 + i], dx[cb]);
                }
            }
        }
        }
        // block rb goes to buffer rb & 1: the barrier below separates its writes from its reduction, and the reduction of
        // block rb - 1 (other buffer) from the writes of block rb + 1 -- one barrier per block instead of two, and a wave
        // multiplies its next block while the others still add
        if (ONE ? active : (active || rb < 2)) {
            float* P = sm.P[rb & 1][ONE ? 0 : head];       // (ONE: both waves write slot 0, their own columns)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, half);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) P[row * PTS + pt0 + cb * 32 + col] = dx[cb][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < (32 * PTS) / NT_; ++e) {
            const int idx = e * NT_ + tid;  // row-major [row][pt]
            const int row = idx / PTS, pt = idx % PTS;
            const float (*Pb)[32 * PTS] = sm.P[rb & 1];
            // (ONE: the other heads' exact zeros are added as constants: the same bits, sign of zero included)
            const float s = ONE ? ((Pb[0][idx] + 0.f) + 0.f) + 0.f : ((Pb[0][idx] + Pb[1][idx]) + Pb[2][idx]) + Pb[3][idx];
            const int k = rb * 32 + row;
            if (k < QF_KPAD) sm.X[pt * XS + k] = s;
        }
    }
    __syncthreads();

    if constexpr (TRAIN) {   // the d(323-vector) tile, consumed by the feature-map scatter
        for (int i = tid; i < PTS * (QF_KPAD / 4); i += NT_) {
            const int pt = i / (QF_KPAD / 4), q = i % (QF_KPAD / 4);
            if (n0 + pt < a.N) *(f32x4*)(a.tdX + (row0 + pt) * QF_KPAD + 4 * q) = *(const f32x4*)(sm.X + pt * XS + 4 * q);
        }
        if (!a.dpoints) return;   // uniform
    }
    // ---- taps again: d(value)/d(ix,iy), then the projection Jacobian ----
    // TU points at a time: their 8 x TU tap rows are requested together (one point per round was one cache round trip per
    // point with nothing else in flight: 32 of them per wave in a 128-point tile)
    using L = MapLoad<T>;
    constexpr int TU = 4;
    static_assert((PTS / NW) % TU == 0, "points per wave");
#pragma unroll 1
    for (int i0 = 0; i0 < PTS / NW; i0 += TU) {
        typename L::Raw4 trf[TU][4];      // raw bits under the validity branches, converted below (query_common.h, MapLoad)
        typename L::Raw1 trt[TU][4];
        bool fin[TU][4], tin[TU][4];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int pt = wid * (PTS / NW) + i0 + u;
            // unconditional loads (a tap outside the map reads the map's first row and is zeroed when used): under a
            // validity branch every 16-byte load was followed by its own s_waitcnt (the vector phi is a copy)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int fo = sm.tab.foff[k][pt];
                fin[u][k] = fo >= 0;
                trf[u][k] = L::raw4(feat_b + (fo >= 0 ? fo : 0) + lane * 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int to = sm.tab.toff[k][pt];
                tin[u][k] = to >= 0;
                trt[u][k] = L::raw1(tmpx_b + (to >= 0 ? to : 0) + lane);
            }
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int pt = wid * (PTS / NW) + i0 + u;
            const float* drow = sm.X + pt * XS;
            float gix_f = 0.f, giy_f = 0.f, gix_t = 0.f, giy_t = 0.f;
            {
                const f32x4 g = *(const f32x4*)(drow + lane * 4);
                f32x4 tv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 t4 = L::cvt4(trf[u][k]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) tv[k][c] = fin[u][k] ? t4[c] : 0.f;
                }
                const float w = sm.tab.ffrac[0][pt], n = sm.tab.ffrac[1][pt];
                const float e = 1.f - w, s = 1.f - n;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    gix_f += g[c] * ((tv[1][c] - tv[0][c]) * s + (tv[3][c] - tv[2][c]) * n);
                    giy_f += g[c] * ((tv[2][c] - tv[0][c]) * e + (tv[3][c] - tv[1][c]) * w);
                }
            }
            {
                const float g = drow[FEAT_C + 3 + lane];
                float tv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) tv[k] = tin[u][k] ? L::cvt1(trt[u][k], lane & 1) : 0.f;
                const float w = sm.tab.tfrac[0][pt], n = sm.tab.tfrac[1][pt];
                const float e = 1.f - w, s = 1.f - n;
                gix_t = g * ((tv[1] - tv[0]) * s + (tv[3] - tv[2]) * n);
                giy_t = g * ((tv[2] - tv[0]) * e + (tv[3] - tv[1]) * w);
            }
            float gnx = gix_f * ((float)(a.FW - 1) * 0.5f) + gix_t * ((float)(a.TW - 1) * 0.5f);
            float gny = giy_f * ((float)(a.FH - 1) * 0.5f) + giy_t * ((float)(a.TH - 1) * 0.5f);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                gnx += __shfl_xor(gnx, o, 64);
                gny += __shfl_xor(gny, o, 64);
            }
            if (lane == 0 && sm.tab.valid[pt]) {
                const float x = sm.tab.xyz[0][pt], y = sm.tab.xyz[1][pt], z = sm.tab.zraw[pt];
                const float k = 2.0f / cam.crop;
                const float gpx = gnx * k, gpy = gny * k;  // d/d(px), d/d(py)
                const float iz = 1.0f / z;
                const float dx = drow[FEAT_C + 0] + gpx * cam.fx * iz;
                const float dy = drow[FEAT_C + 1] + gpy * cam.fy * iz;
                const float dz = drow[FEAT_C + 2] - (gpx * cam.fx * x + gpy * cam.fy * y) * iz * iz;
                float* o = a.dpoints + ((size_t)b * a.N + n0 + pt) * 3;
                if constexpr (SURF) {       // F.normalize(grad) * clamp(df): the arithmetic of gen_surface_step_kernel
                    const float d = sm.dfk[pt];
                    const float den = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-12f);
                    o[0] = x - dx / den * d; o[1] = y - dy / den * d; o[2] = z - dz / den * d;
                } else {
                    o[0] = dx; o[1] = dy; o[2] = dz;
                }
            }
        }
    }
}

template <typename T, bool TRAIN, int NCB, bool X3 = false>
static int launch_query_bwd_n(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    constexpr int PTS = 32 * NCB;
    const size_t smem = sizeof(QueryBwdSmemT<PTS>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_bwd_f32_kernel<T, TRAIN, NCB, false, 4, X3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + PTS - 1) / PTS, a.B);
    hipLaunchKernelGGL((query_bwd_f32_kernel<T, TRAIN, NCB, false, 4, X3>), grid, dim3(256), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

bool query_small_tiles(int B, int N);   // query_fwd.hip: 32-point tiles when 64-point tiles would not fill the CUs

// fp16 x 3 backward, a few rounds of workgroups: a round of 64-point tiles takes 82 us, one of 32-point tiles 46 us (measured,
// one workgroup per CU either way), so e.g. 20 000 points (313 tiles = 2 rounds, the second a fifth full) finish sooner as
// 625 small tiles (3 rounds): 165 -> 137 us; from ~10 rounds on the 64-point tiles win on every count
static bool x3_bwd_prefers_small(int B, int N) {
    const long long t64 = (long long)B * ((N + 63) / 64), t32 = (long long)B * ((N + 31) / 32);
    const long long r64 = (t64 + 255) / 256, r32 = (t32 + 255) / 256;
    return r32 * 455 < r64 * 825;
}

static bool query_one_head() { static const bool off = getenv("CHORE_QUERY_NO_ONE_HEAD") != nullptr; return !off; }   // A/B switch

template <typename T, int NCB, bool SURF = true, bool ONE = false>
static int launch_query_surf_n(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    constexpr int PTS = 32 * NCB;
    const size_t smem = sizeof(QueryBwdSmemT<PTS>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_bwd_f32_kernel<T, false, NCB, false, 4, true, SURF, ONE>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + PTS - 1) / PTS, a.B);
    hipLaunchKernelGGL((query_bwd_f32_kernel<T, false, NCB, false, 4, true, SURF, ONE>), grid, dim3(256), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}
template <typename T>
static int launch_query_surf_t(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    // 64-point tiles with the chain on two waves wherever 64-point tiles fill the CUs (the 32-point rule of the backward was
    // measured on the one-wave chain); 32-point tiles for the small queries
    if (query_one_head() && !query_small_tiles(a.B, a.N)) return launch_query_surf_n<T, 2, true, true>(h, a, s);
    return (query_small_tiles(a.B, a.N) || x3_bwd_prefers_small(a.B, a.N)) ? launch_query_surf_n<T, 1>(h, a, s)
                                                                              : launch_query_surf_n<T, 2>(h, a, s);     // as the backward
}
int launch_query_surface_step(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s) {      // dtype: the maps' type
    if (dtype == CHORE_F16) return launch_query_surf_t<qh16_t>(h, a, s);
    return dtype == CHORE_F32 ? launch_query_surf_t<float>(h, a, s) : launch_query_surf_t<unsigned short>(h, a, s);
}

// the eight-wave variants (64-point tile, two waves per head)
template <typename T, bool TRAIN, bool STAGED, bool X3 = false>
static int launch_query_bwd_w8(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    const size_t smem = sizeof(QueryBwdSmemT<64>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_bwd_f32_kernel<T, TRAIN, 1, STAGED, 8, X3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + 63) / 64, a.B);
    hipLaunchKernelGGL((query_bwd_f32_kernel<T, TRAIN, 1, STAGED, 8, X3>), grid, dim3(512), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

static bool query_w4() { static const bool v = getenv("CHORE_QUERY_W4") != nullptr; return v; }   // A/B switch

template <typename T, bool TRAIN, bool X3 = false>
static int launch_query_bwd_t(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    if constexpr (!TRAIN && X3) {
        // one upstream gradient (the generator's steps, most fit phases): the head's chain on two waves of a 64-point tile
        int live = 0, which = 0;
        for (int i = 0; i < HEAD_NUM; ++i)
            if (a.g[i]) { ++live; which = i; }
        if (live == 1 && query_one_head() && !query_small_tiles(a.B, a.N)) {
            QueryArgs b = a;
            b.one_head = which;
            return launch_query_surf_n<T, 2, false, true>(h, b, s);
        }
    }
    if constexpr (!TRAIN) {
        static const bool x3_small = getenv("CHORE_QUERY_X3_BWD_SMALL") != nullptr;
        if (query_small_tiles(a.B, a.N) || (X3 && (x3_small || x3_bwd_prefers_small(a.B, a.N))))
            return launch_query_bwd_n<T, false, 1, X3>(h, a, s);
    }
    // the training variants are bound by their staging stores: measured slower with eight waves (39.4 vs 38.6 ms per step)
    // fp16 x 3: four waves with two column blocks each, like the forward (the eight-wave kernel fetches every weight
    // fragment twice and is bound by the L1: 0.65 against 0.58 ms at 4 x 20 000 points)
    if constexpr (!TRAIN && !X3) {
        if (!query_w4()) return launch_query_bwd_w8<T, false, false, X3>(h, a, s);
    }
    // (an eight-wave fp16 x 3 RECOMPUTE kernel existed behind a switch through round 3: slower, and in round 2 it produced a
    // non-finite gradient about once in six runs of the graph-replayed fit that was never reproduced afterwards; removed in round 4)
    return launch_query_bwd_n<T, TRAIN, 2, X3>(h, a, s);
}

template <typename T>
static int launch_query_bwd_staged_x3(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    return launch_query_bwd_w8<T, true, true, true>(h, a, s);     // no recompute: the eight-wave kernel fits its registers
}

template <typename T>
static int launch_query_bwd_staged_t(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    const size_t smem = sizeof(QueryBwdSmemT<64>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_bwd_f32_kernel<T, true, 2, true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + 63) / 64, a.B);
    hipLaunchKernelGGL((query_bwd_f32_kernel<T, true, 2, true>), grid, dim3(256), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int launch_query_bwd_f32(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    return launch_query_bwd_t<float, false>(h, a, s);
}
int launch_query_bwd_f32_bf16maps(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    return launch_query_bwd_t<unsigned short, false>(h, a, s);
}
int launch_query_bwd_x3(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s) {      // dtype: the maps' type
    if (dtype == CHORE_F16) return launch_query_bwd_t<qh16_t, false, true>(h, a, s);
    return dtype == CHORE_F32 ? launch_query_bwd_t<float, false, true>(h, a, s) : launch_query_bwd_t<unsigned short, false, true>(h, a, s);
}

int launch_query_bwd_train(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s, int staged, int x3) {
    if (staged && x3) return dtype == CHORE_F32 ? launch_query_bwd_staged_x3<float>(h, a, s) : launch_query_bwd_staged_x3<unsigned short>(h, a, s);
    if (staged) return dtype == CHORE_F32 ? launch_query_bwd_staged_t<float>(h, a, s) : launch_query_bwd_staged_t<unsigned short>(h, a, s);
    return dtype == CHORE_F32 ? launch_query_bwd_t<float, true>(h, a, s) : launch_query_bwd_t<unsigned short, true>(h, a, s);
}

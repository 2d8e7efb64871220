// query_fwd.hip -- fused CHORE.query forward for gfx950, exact-fp32 heads.
//
// One workgroup = 64 query points of one image, 4 waves.  Phase 1: 64 lanes project their point
// (bit-exact restatement of model/camera.py:44-88) and build the bilinear tap table.  Phase 2: each
// wave gathers 16 points: the 4 taps of a point are contiguous 1 KB (feat, 256 ch) / 256 B (tmpx,
// 64 ch) rows of the NHWC maps, so every load is a fully coalesced wave access; the 323-vector
// [feat | x y z-2.2 | tmpx] (model/chore.py:139-143) lands in an LDS tile X[64][332].  Phase 3: wave
// w runs the complete MLP of head w (model/chore.py:74-85,156-167) on the matrix cores with
// v_mfma_f32_32x32x2_f32 in the transposed form H^T = W * X^T: the D fragment of one layer
// (row = channel, col = point) is already the B fragment of the next layer, so the activations
// never leave registers; weights stream from the L2-resident fragment-ordered arena written by
// heads_pack.  Phase 4: masked (df[~in_img] = 5.0, model/chore.py:147-150) coalesced stores in the
// (B,C,N) layout the reference API returns.
#include "heads_x3.h"
#include <cstdlib>

template <int PTS>
struct QueryFwdSmemT {
    float X[PTS * XS];
    PtTableT<PTS> tab;
};
using QueryFwdSmem = QueryFwdSmemT<QT_PTS>;

// NCB = 32-point column blocks per workgroup tile: 2 (64 points) for large queries; 1 for the small queries of the fit
// loop (6 890 / 3 000 points give 108 / 47 tiles of 64: fewer workgroups than CUs, each a serial MFMA chain --
// halving the tile halves that chain and doubles the workgroups)
// TRAIN: also stage the 323-vectors and the ReLU outputs of the hidden layers (tX, tH) for the backward pass
// X3: the heads on the fp16 matrix cores with hi/lo split operands (heads_x3.h) instead of the native fp32 MFMA
template <typename T, int NCB, bool TRAIN = false, bool X3 = false>
__global__ __launch_bounds__(256, 1) void query_fwd_f32_kernel(QueryArgs a) {
    static_assert(!TRAIN || NCB == 2, "the training staging is written for 64-point tiles");
    constexpr int PTS = 32 * NCB;
    if constexpr (X3) f16_saturate_mode();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    QueryFwdSmemT<PTS>& sm = *reinterpret_cast<QueryFwdSmemT<PTS>*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, tile_;
    query_block(b, tile_);
    const int n0 = tile_ * PTS;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};

    if (tid < PTS) {
        fill_pt_table(sm.tab, tid, a.points, a.crop_center, b, n0 + tid, a.N, cam, a.FH, a.FW, a.TH,
                      a.TW, nullptr);
        if (a.in_img && n0 + tid < a.N) a.in_img[(size_t)b * a.N + n0 + tid] = (uint8_t)sm.tab.in_img[tid];
    }
    __syncthreads();
    const T* feat_b = (const T*)a.feat + (size_t)b * a.FH * a.FW * FEAT_C;
    const T* tmpx_b = (const T*)a.tmpx + (size_t)b * a.TH * a.TW * TMPX_C;
    gather_tile<T, PTS>(sm.X, sm.tab, feat_b, tmpx_b, wid, lane);
    __syncthreads();

    const float* arena = (const float*)a.arena;
    const int head = wid;
    if constexpr (!TRAIN) {
        if (a.out[head] == nullptr) return;     // output not asked for (chore_query_fwd with a NULL pointer): no barrier follows
    }
    f32x16 h1[4][NCB], h2[4][NCB];
    const size_t row0 = (size_t)b * a.N + n0, plane = (size_t)a.B * a.N * HEAD_HID, mplane = (size_t)a.B * a.N * 2;
    if constexpr (TRAIN) {
        for (int i = tid; i < PTS * (QF_KPAD / 4); i += 256) {
            const int pt = i / (QF_KPAD / 4), q = i % (QF_KPAD / 4);
            if (n0 + pt < a.N) *(f32x4*)(a.tX + (row0 + pt) * QF_KPAD + 4 * q) = *(const f32x4*)(sm.X + pt * XS + 4 * q);
        }
    }
    f32x16 o[NCB];
    if constexpr (X3) {        // the accumulators carry the weights' 2^s: staged rows are rescaled on the way out
        float inv[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) inv[cb] = QX_INV;
        heads_layer1_x3<NCB>(h1, sm.X, arena, head, lane);
        if constexpr (TRAIN) store_tile<NCB>(a.tH + (0 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane, 0, inv);
        if constexpr (TRAIN) store_masks<NCB>(a.tM + (0 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane, 0);
        heads_layer_hid_x3<NCB>(h2, h1, arena, head, 1, lane);
        if constexpr (TRAIN) store_tile<NCB>(a.tH + (1 * HEAD_NUM + head) * plane, h2, true, row0, n0, a.N, lane, 0, inv);
        if constexpr (TRAIN) store_masks<NCB>(a.tM + (1 * HEAD_NUM + head) * mplane, h2, row0, n0, a.N, lane, 0);
        heads_layer_hid_x3<NCB>(h1, h2, arena, head, 2, lane);
        if constexpr (TRAIN) store_tile<NCB>(a.tH + (2 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane, 0, inv);
        if constexpr (TRAIN) store_masks<NCB>(a.tM + (2 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane, 0);
        heads_layer_out_x3<NCB>(o, h1, arena, head, lane);
    } else {
        heads_layer1<NCB>(h1, sm.X, arena, head, lane);
        if constexpr (TRAIN) store_tile<NCB>(a.tH + (0 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane);
        if constexpr (TRAIN) store_masks<NCB>(a.tM + (0 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane);
        heads_layer_hid<NCB>(h2, h1, arena, head, 1, lane);
        if constexpr (TRAIN) store_tile<NCB>(a.tH + (1 * HEAD_NUM + head) * plane, h2, true, row0, n0, a.N, lane);
        if constexpr (TRAIN) store_masks<NCB>(a.tM + (1 * HEAD_NUM + head) * mplane, h2, row0, n0, a.N, lane);
        heads_layer_hid<NCB>(h1, h2, arena, head, 2, lane);
        if constexpr (TRAIN) store_tile<NCB>(a.tH + (2 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane);
        if constexpr (TRAIN) store_masks<NCB>(a.tM + (2 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane);
        heads_layer_out<NCB>(o, h1, arena, head, lane);
    }

    const int odim = head_out_dim(head);
    float* outp = a.out[head] + (size_t)b * odim * a.N;
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int pt = cb * 32 + col;
        const int n = n0 + pt;
        const bool inside = sm.tab.in_img[pt] != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = mfma32_row(r, half);
            if (ch < odim && n < a.N) {
                float v = o[cb][r];
                if (head == 0 && !inside) v = 5.0f;
                outp[(size_t)ch * a.N + n] = v;
            }
        }
    }
}

// Eight-wave variant for large queries: the same 64-point tile, but two waves per head, one 32-point column block each.
// A wave of the four-wave kernel is stalled on weight / tap fetches ~40 % of its life (SQ_WAIT_ANY) with nothing else
// resident on its SIMD; here every SIMD holds two waves (<= 256 registers each), the second wave of a head finds the
// weight lines of the first in the L1, and the MFMA pipe stays busy while one of them waits.
template <typename T, bool TRAIN = false, bool X3 = false>
__global__ __launch_bounds__(512, 1) void query_fwd_f32_w8_kernel(QueryArgs a) {
    constexpr int PTS = 64;
    if constexpr (X3) f16_saturate_mode();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    QueryFwdSmemT<PTS>& sm = *reinterpret_cast<QueryFwdSmemT<PTS>*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* scr = reinterpret_cast<float*>(smem_raw + sizeof(QueryFwdSmemT<PTS>)) + wid * (32 * ST_LD);   // TRAIN only (store_tile_lds)
    int b, tile_;
    query_block(b, tile_);
    const int n0 = tile_ * PTS;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};
    if (tid < PTS) {
        fill_pt_table(sm.tab, tid, a.points, a.crop_center, b, n0 + tid, a.N, cam, a.FH, a.FW, a.TH,
                      a.TW, nullptr);
        if (a.in_img && n0 + tid < a.N) a.in_img[(size_t)b * a.N + n0 + tid] = (uint8_t)sm.tab.in_img[tid];
    }
    __syncthreads();
    const T* feat_b = (const T*)a.feat + (size_t)b * a.FH * a.FW * FEAT_C;
    const T* tmpx_b = (const T*)a.tmpx + (size_t)b * a.TH * a.TW * TMPX_C;
    gather_tile<T, PTS, 8>(sm.X, sm.tab, feat_b, tmpx_b, wid, lane);
    __syncthreads();

    const float* arena = (const float*)a.arena;
    const int head = wid & 3, cb0 = wid >> 2;
    if constexpr (!TRAIN) {
        if (a.out[head] == nullptr) return;     // as in the four-wave kernel
    }
    f32x16 h1[4][1], h2[4][1];
    const size_t row0 = (size_t)b * a.N + n0, plane = (size_t)a.B * a.N * HEAD_HID, mplane = (size_t)a.B * a.N * 2;
    if constexpr (TRAIN) {
        for (int i = tid; i < PTS * (QF_KPAD / 4); i += 512) {
            const int pt = i / (QF_KPAD / 4), q = i % (QF_KPAD / 4);
            if (n0 + pt < a.N) *(f32x4*)(a.tX + (row0 + pt) * QF_KPAD + 4 * q) = *(const f32x4*)(sm.X + pt * XS + 4 * q);
        }
    }
    f32x16 o[1];
    if constexpr (X3) {
        const float inv[1] = {QX_INV};
        heads_layer1_x3<1>(h1, sm.X + cb0 * 32 * XS, arena, head, lane);
        if constexpr (TRAIN) store_tile_lds<1>(a.tH + (0 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane, cb0 * 32, inv, scr);
        if constexpr (TRAIN) store_masks<1>(a.tM + (0 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane, cb0 * 32);
        heads_layer_hid_x3<1>(h2, h1, arena, head, 1, lane);
        if constexpr (TRAIN) store_tile_lds<1>(a.tH + (1 * HEAD_NUM + head) * plane, h2, true, row0, n0, a.N, lane, cb0 * 32, inv, scr);
        if constexpr (TRAIN) store_masks<1>(a.tM + (1 * HEAD_NUM + head) * mplane, h2, row0, n0, a.N, lane, cb0 * 32);
        heads_layer_hid_x3<1>(h1, h2, arena, head, 2, lane);
        if constexpr (TRAIN) store_tile_lds<1>(a.tH + (2 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane, cb0 * 32, inv, scr);
        if constexpr (TRAIN) store_masks<1>(a.tM + (2 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane, cb0 * 32);
        heads_layer_out_x3<1>(o, h1, arena, head, lane);
    } else {
        heads_layer1<1>(h1, sm.X + cb0 * 32 * XS, arena, head, lane);
        if constexpr (TRAIN) store_tile_lds<1>(a.tH + (0 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane, cb0 * 32, nullptr, scr);
        if constexpr (TRAIN) store_masks<1>(a.tM + (0 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane, cb0 * 32);
        heads_layer_hid<1>(h2, h1, arena, head, 1, lane);
        if constexpr (TRAIN) store_tile_lds<1>(a.tH + (1 * HEAD_NUM + head) * plane, h2, true, row0, n0, a.N, lane, cb0 * 32, nullptr, scr);
        if constexpr (TRAIN) store_masks<1>(a.tM + (1 * HEAD_NUM + head) * mplane, h2, row0, n0, a.N, lane, cb0 * 32);
        heads_layer_hid<1>(h1, h2, arena, head, 2, lane);
        if constexpr (TRAIN) store_tile_lds<1>(a.tH + (2 * HEAD_NUM + head) * plane, h1, true, row0, n0, a.N, lane, cb0 * 32, nullptr, scr);
        if constexpr (TRAIN) store_masks<1>(a.tM + (2 * HEAD_NUM + head) * mplane, h1, row0, n0, a.N, lane, cb0 * 32);
        heads_layer_out<1>(o, h1, arena, head, lane);
    }

    const int odim = head_out_dim(head);
    float* outp = a.out[head] + (size_t)b * odim * a.N;
    const int half = lane >> 5, col = lane & 31;
    const int pt = cb0 * 32 + col, n = n0 + pt;
    const bool inside = sm.tab.in_img[pt] != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = mfma32_row(r, half);
        if (ch < odim && n < a.N) {
            float v = o[0][r];
            if (head == 0 && !inside) v = 5.0f;
            outp[(size_t)ch * a.N + n] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 x 3 forward, TWO waves per head that split the head's OUTPUT channels ("split" kernel).
//
// The four-wave kernel above keeps a layer's activations in registers, so one wave walks the whole GEMM chain of its head:
// 936 dependent MFMAs per 64-point tile with nothing else on its SIMD, every weight-fragment or LDS wait exposed (a tile
// takes 42 us for 14 us of MFMA work).  Here a head has two waves: wave (head, c) owns row blocks 2c, 2c+1 -- 64 of the 128
// hidden channels -- of every layer, for all points of the tile.  It needs only ITS half of each weight matrix (no fragment
// is fetched twice: the L1 traffic per tile is that of the four-wave kernel) but all K channels of the previous layer, so
// the layers exchange their activations through the LDS: after a layer each wave writes relu(.) of its 64 channels as
// ready-made B fragments (hi / lo planes, the k order heads_x3.h packs the weights in), a barrier, and both waves of the
// head read all eight K-steps from there.  The input tile is stored the same way (gather_tile_split: fp16 hi / lo planes
// at gather time), so no k-loop converts anything: loads and MFMAs only, half the chain per wave, two waves per SIMD.
// LDS: the input planes (86 KB at 64 points) and the exchange area (32 KB per head) share one region -- the input is dead
// when layer 1 is done -- plus the point table: 135 KB.
constexpr int XSH = 344;                 // halves per point row of an input plane (336 + 8: 688 B, conflict-free b128 rows)
template <int PTS>
struct QuerySplitSmemT {
    static constexpr size_t XB = (size_t)2 * PTS * XSH * 2;                       // two planes of halves
    static constexpr size_t EB = (size_t)HEAD_NUM * 4 * 2 * (PTS / 32) * 2 * 1024;  // [head][kb][s][cb][plane] x 1 KB
    __attribute__((aligned(16))) char buf[XB > EB ? XB : EB];
    PtTableT<PTS> tab;
};

template <typename T, int PTS>
__device__ __forceinline__ void gather_tile_split(char* XH, char* XL, const PtTableT<PTS>& tab, const T* feat_b, const T* tmpx_b,
                                                  int wid, int lane) {
    using L = MapLoad<T>;
    constexpr int NW = 8, U = PTS / NW;          // all of a wave's points (8 or 4) in flight at once: one HBM round trip
    // The fp32 value is pinned in a register before it is split: left alone, the compiler folds the last fmaf of the
    // interpolation into the conversion (v_fma_mixlo_f16: ONE rounding to half instead of fp32 then half), and on an exact tie
    // the planes differ from the in-register split of the one-wave-per-head kernels -- the fused surface step, which
    // recomputes its forward there, is checked against this kernel bit for bit (tests/test_gpu_generator.py).
    auto put = [&](char* row_h, char* row_l, int k, float v) {
        asm volatile("" : "+v"(v));
        const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
        *(_Float16*)(row_h + 2 * k) = h;
        *(_Float16*)(row_l + 2 * k) = l;
    };
    typename L::Raw4 fv[U][4];
    typename L::Raw1 tv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int pt = wid * U + u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int fo = tab.foff[k][pt];
            const int to = tab.toff[k][pt];
            fv[u][k] = (fo >= 0) ? L::raw4(feat_b + fo + lane * 4) : L::zero4();
            tv[u][k] = (to >= 0) ? L::raw1(tmpx_b + to + lane) : L::zero1();
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int pt = wid * U + u;
        float fw[4], tw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { fw[k] = tab.fw[k][pt]; tw[k] = tab.tw[k][pt]; }
        const f32x4 c0 = L::cvt4(fv[u][0]), c1 = L::cvt4(fv[u][1]), c2 = L::cvt4(fv[u][2]), c3 = L::cvt4(fv[u][3]);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 hh, ll;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float r = interp4(c0[c], c1[c], c2[c], c3[c], fw);
            asm volatile("" : "+v"(r));
            hh[c] = (_Float16)r;
            ll[c] = (_Float16)(r - (float)hh[c]);
        }
        char* rh = XH + (size_t)pt * XSH * 2;
        char* rl = XL + (size_t)pt * XSH * 2;
        *(h4*)(rh + lane * 8) = hh;
        *(h4*)(rl + lane * 8) = ll;
        put(rh, rl, FEAT_C + 3 + lane, interp4(L::cvt1(tv[u][0], lane & 1), L::cvt1(tv[u][1], lane & 1), L::cvt1(tv[u][2], lane & 1),
                                               L::cvt1(tv[u][3], lane & 1), tw));
        if (lane < 3) put(rh, rl, FEAT_C + lane, tab.xyz[lane][pt]);
        if (lane >= 3 && lane < 3 + (QX_KS1 * 16 - HEAD_IN)) put(rh, rl, HEAD_IN + lane - 3, 0.f);      // k = 323 .. 335
    }
}

#ifdef CHORE_QUERY_STAMPS
__device__ unsigned long long g_qstamps[4096 * 8];
#define QSTAMP(i) do { if (tid == 0) { const int L_ = blockIdx.x + blockIdx.y * gridDim.x; if (L_ < 4096) g_qstamps[L_ * 8 + (i)] = wall_clock64(); } } while (0)
__device__ unsigned short g_qx[2 * 64 * XSH];
__device__ int g_qx_tile = -1;
extern "C" int chore_debug_query_xdump(unsigned short* out, int tile) {
    if (tile >= 0) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_qx_tile), &tile, sizeof(int));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qx), sizeof(unsigned short) * 2 * 64 * XSH);
}
extern "C" int chore_debug_query_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qstamps), sizeof(unsigned long long) * (n < 4096 * 8 ? n : 4096 * 8));
}
#else
#define QSTAMP(i) do { } while (0)
#endif
struct AFrag2 { u32x4 h[2], l[2]; };
__device__ __forceinline__ void load_afrag2(AFrag2& f, const u32x4* A /*[rb][plane][lane] of this k-step, lane applied*/, int rb0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { f.h[i] = A[((rb0 + i) * 2) * 64]; f.l[i] = A[((rb0 + i) * 2 + 1) * 64]; }
}

template <typename T, int NCB>
__global__ __launch_bounds__(512, 1) void query_fwd_x3_split_kernel(QueryArgs a) {
    // The weight fragments of the three hidden layers are ONE stream of 21 + 8 + 8 k-steps through a ring of PF register
    // slots: the loads run PF steps ahead of the MFMAs across the layer boundaries, so a layer's first fragments are
    // already in registers when its barrier opens (a k-step of one wave lasts ~0.2 us, an L2 round trip ~1 us).
    constexpr int PTS = 32 * NCB, PF = 7, NSTEP = QX_KS1 + 16;
    static_assert(QX_KS1 % PF == 0, "the k loop of layer 1 is unrolled by the ring depth");
    f16_saturate_mode();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    QuerySplitSmemT<PTS>& sm = *reinterpret_cast<QuerySplitSmemT<PTS>*>(smem_raw);
    char* XH = sm.buf;
    char* XL = sm.buf + (size_t)PTS * XSH * 2;
    u32x4* E = (u32x4*)sm.buf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    int b, tile_;
    query_block(b, tile_);
    const int n0 = tile_ * PTS;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};
    QSTAMP(0);
    if (tid < PTS) {
        fill_pt_table(sm.tab, tid, a.points, a.crop_center, b, n0 + tid, a.N, cam, a.FH, a.FW, a.TH, a.TW, nullptr, a.perm);
        if (a.in_img && n0 + tid < a.N) a.in_img[(size_t)b * a.N + sm.tab.pidx[tid]] = (uint8_t)sm.tab.in_img[tid];
    }
    __syncthreads();
    QSTAMP(1);
    const float* arena = (const float*)a.arena;
    const int head = wid & 3, ch = wid >> 2, rb0 = 2 * ch;
    const bool active = a.out[head] != nullptr;          // a head nobody asked for: its waves only keep the barriers
    const u32x4* xbase = (const u32x4*)((const char*)arena + QX_OFF_BYTES);
    const u32x4* A1 = xbase + (size_t)head * QX_KS1 * 4 * 2 * 64 + lane;
    const u32x4* A23 = xbase + QX_L1_VEC + (size_t)head * 2 * 4 * 2 * 4 * 2 * 64 + lane;     // layers 2, 3: 16 consecutive k-steps
    auto step_ptr = [&](int st) -> const u32x4* {        // fragment block of stream step st
        return st < QX_KS1 ? A1 + (size_t)st * 4 * 2 * 64 : A23 + (size_t)(st - QX_KS1) * 4 * 2 * 64;
    };
    AFrag2 ring[PF];
    // 16-bit maps: the ring's first loads fly under the gather; fp32 maps: the gather's 8 points x 4 taps in flight take the
    // registers (an HBM round trip saved there is worth more than the one L2 round trip exposed here)
    constexpr bool RING_EARLY = sizeof(T) == 2 || NCB == 1;
    if (RING_EARLY && active) {
#pragma unroll
        for (int p = 0; p < PF; ++p) load_afrag2(ring[p], step_ptr(p), rb0);
    }
    const T* feat_b = (const T*)a.feat + (size_t)b * a.FH * a.FW * FEAT_C;
    const T* tmpx_b = (const T*)a.tmpx + (size_t)b * a.TH * a.TW * TMPX_C;
    gather_tile_split<T, PTS>(XH, XL, sm.tab, feat_b, tmpx_b, wid, lane);
    if (!RING_EARLY && active) {
#pragma unroll
        for (int p = 0; p < PF; ++p) load_afrag2(ring[p], step_ptr(p), rb0);
    }
    __syncthreads();
    QSTAMP(2);
#ifdef CHORE_QUERY_STAMPS
    if (b == 0 && tile_ == g_qx_tile)
        for (int i = tid; i < 2 * PTS * XSH; i += 512) g_qx[i] = ((const unsigned short*)sm.buf)[i];
#endif

    f32x16 acc[2][NCB], nb[2];
    auto mm = [&](const AFrag2& f, const u32x4 (&bh)[NCB], const u32x4 (&bl)[NCB]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[i][cb] = mfma3(f.h[i], f.l[i], bh[cb], bl[cb], acc[i][cb]);
    };
    auto fetch_bias = [&](int layer) {
#pragma unroll
        for (int i = 0; i < 2; ++i) nb[i] = scaled_bias(arena, head, layer, rb0 + i, half);
    };
    auto init_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[i][cb] = nb[i];
    };
    // relu(acc / 2^s) of this wave's two row blocks -> B fragments of K-blocks rb0, rb0 + 1 of the next layer
    auto publish = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = relu(acc[i][cb][8 * s + j] * QX_INV);
                    u32x4 hi, lo;
                    split8(v, hi, lo);
                    u32x4* dst = E + ((((size_t)(head * 4 + rb0 + i) * 2 + s) * NCB + cb) * 2) * 64 + lane;
                    dst[0] = hi;
                    dst[64] = lo;
                }
    };

    // ---- layer 1: K = 336 from the input planes ----
    if (active) {
        fetch_bias(0);
        init_acc();
        fetch_bias(1);                   // the next layer's, in flight under this one
        const char* xh0 = XH + (size_t)col * XSH * 2 + 16 * half;
        const char* xl0 = XL + (size_t)col * XSH * 2 + 16 * half;
#pragma unroll 1
        for (int k0 = 0; k0 < QX_KS1; k0 += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int ks = k0 + p;
                u32x4 bh[NCB], bl[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    bh[cb] = *(const u32x4*)(xh0 + (size_t)cb * 32 * XSH * 2 + ks * 32);
                    bl[cb] = *(const u32x4*)(xl0 + (size_t)cb * 32 * XSH * 2 + ks * 32);
                }
                mm(ring[p], bh, bl);
                load_afrag2(ring[p], step_ptr(ks + PF), rb0);       // ks + PF < NSTEP always: runs into layer 2's fragments
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();                     // every wave is done with the input planes: the exchange area may be written
    QSTAMP(3);
    if (active) publish();
    __syncthreads();

    // ---- layers 2, 3: K = 128 from the exchange area ----
    u32x4 ah[8], al[8];                  // the output layer's fragments (requested when layer 3's MFMAs are done)
#pragma unroll
    for (int layer = 1; layer <= 2; ++layer) {
        if (active) {
            init_acc();
            fetch_bias(layer + 1);
            const u32x4* Eh = E + (size_t)head * 4 * 2 * NCB * 2 * 64 + lane;
#pragma unroll
            for (int t = 0; t < 8; ++t) {          // k-step t = (kb, s)
                const int st = QX_KS1 + 8 * (layer - 1) + t;
                u32x4 bh[NCB], bl[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    bh[cb] = Eh[((size_t)(t * NCB + cb) * 2) * 64];
                    bl[cb] = Eh[((size_t)(t * NCB + cb) * 2 + 1) * 64];
                }
                mm(ring[st % PF], bh, bl);
                if (st + PF < NSTEP) load_afrag2(ring[st % PF], step_ptr(st + PF), rb0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (layer == 2) {
                const u32x4* A = xbase + QX_L1_VEC + QX_L23_VEC + (size_t)head * 4 * 2 * 2 * 64 + lane;
#pragma unroll
                for (int t = 0; t < 8; ++t) { ah[t] = A[(t * 2) * 64]; al[t] = A[(t * 2 + 1) * 64]; }
            }
        }
        __syncthreads();                 // both waves of every head have read the layer's input
        if (active) publish();
        __syncthreads();
    }

    QSTAMP(4);
    // ---- output layer (one 32-row block): the head's two waves take a column block each ----
    const int cb_o = NCB == 2 ? ch : 0;
    if (!active || (NCB == 1 && ch != 0)) return;
    f32x16 o = nb[0];                    // fetch_bias(3) with rb0 = 0: both waves need row block 0
    if (rb0 != 0) o = scaled_bias(arena, head, 3, 0, half);
    {
        const u32x4* Eh = E + (size_t)head * 4 * 2 * NCB * 2 * 64 + lane;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const u32x4 bh = Eh[((size_t)(t * NCB + cb_o) * 2) * 64], bl = Eh[((size_t)(t * NCB + cb_o) * 2 + 1) * 64];
            o = mfma3(ah[t], al[t], bh, bl, o);
        }
    }
    const int odim = head_out_dim(head);
    float* outp = a.out[head] + (size_t)b * odim * a.N;
    const int pt = cb_o * 32 + col, n = sm.tab.pidx[pt];        // (sorted order: the outputs go back to the point's own column)
    const bool inside = sm.tab.in_img[pt] != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int chn = mfma32_row(r, half);
        if (chn < odim && n0 + pt < a.N) {
            float v = o[r] * QX_INV;
            if (head == 0 && !inside) v = 5.0f;
            outp[(size_t)chn * a.N + n] = v;
        }
    }
    QSTAMP(5);
}

template <typename T, int NCB>
static int launch_query_fwd_split(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    constexpr int PTS = 32 * NCB;
    const size_t smem = sizeof(QuerySplitSmemT<PTS>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_fwd_x3_split_kernel<T, NCB>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + PTS - 1) / PTS, a.B);
    hipLaunchKernelGGL((query_fwd_x3_split_kernel<T, NCB>), grid, dim3(512), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// pixel-aligned feature sample only (projection + 2x index + z_feat, no heads): the 323-vector per
// point, point-major.  Same phase 1/2 code as the fused kernel, so it doubles as its probe.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sample_features_kernel(QueryArgs a, float* features, float* nxy) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    QueryFwdSmem& sm = *reinterpret_cast<QueryFwdSmem*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, tile_;
    query_block(b, tile_);
    const int n0 = tile_ * QT_PTS;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};
    if (tid < QT_PTS) {
        float nn[2];
        fill_pt_table(sm.tab, tid, a.points, a.crop_center, b, n0 + tid, a.N, cam, a.FH, a.FW, a.TH, a.TW, nn);
        if (nxy && n0 + tid < a.N) {
            nxy[((size_t)b * a.N + n0 + tid) * 2 + 0] = nn[0];
            nxy[((size_t)b * a.N + n0 + tid) * 2 + 1] = nn[1];
        }
        if (a.in_img && n0 + tid < a.N) a.in_img[(size_t)b * a.N + n0 + tid] = (uint8_t)sm.tab.in_img[tid];
    }
    __syncthreads();
    const T* feat_b = (const T*)a.feat + (size_t)b * a.FH * a.FW * FEAT_C;
    const T* tmpx_b = (const T*)a.tmpx + (size_t)b * a.TH * a.TW * TMPX_C;
    gather_tile<T>(sm.X, sm.tab, feat_b, tmpx_b, wid, lane);
    __syncthreads();
    for (int i = tid; i < QT_PTS * HEAD_IN; i += 256) {
        const int pt = i / HEAD_IN, k = i % HEAD_IN;
        if (n0 + pt < a.N) features[((size_t)b * a.N + n0 + pt) * HEAD_IN + k] = sm.X[pt * XS + k];
    }
}

template <typename T>
static int launch_sample_features_t(chore_handle* h, const QueryArgs& a, float* features, float* nxy, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    const size_t smem = sizeof(QueryFwdSmem);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)sample_features_kernel<T>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + QT_PTS - 1) / QT_PTS, a.B);
    hipLaunchKernelGGL(sample_features_kernel<T>, grid, dim3(256), smem, s, a, features, nxy);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}
int launch_sample_features(chore_handle* h, int dtype, const QueryArgs& a, float* features, float* nxy, hipStream_t s) {
    return dtype == CHORE_F32 ? launch_sample_features_t<float>(h, a, features, nxy, s)
                              : launch_sample_features_t<unsigned short>(h, a, features, nxy, s);
}

// ------------------------------------------------------------------------------------------------
// weight packing: reference Conv1d layouts -> MFMA fragment order (see common.h for the layout)
// ------------------------------------------------------------------------------------------------
__global__ void heads_pack_f32_kernel(HeadsRaw raw, float* arena) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= QF_TOTAL_FLOATS) return;
    float v = 0.f;
    if (idx < QF_OFF_L23) {  // layer 1: [h][q][rb][lane][4]
        size_t t = idx - QF_OFF_L1;
        const int i = t & 3; t >>= 2;
        const int lane = t & 63; t >>= 6;
        const int rb = t & 3; t >>= 2;
        const int q = (int)(t % QF_KG);
        const int h = (int)(t / QF_KG);
        const int row = rb * 32 + (lane & 31);
        const int k = q * 8 + 4 * (lane >> 5) + i;
        if (k < HEAD_IN) v = raw.w[h][0][(size_t)row * HEAD_IN + k];
    } else if (idx < QF_OFF_L4) {  // layers 2,3: [h][l][kb][rg][rb][lane][4]
        size_t t = idx - QF_OFF_L23;
        const int i = t & 3; t >>= 2;
        const int lane = t & 63; t >>= 6;
        const int rb = t & 3; t >>= 2;
        const int rg = t & 3; t >>= 2;
        const int kb = t & 3; t >>= 2;
        const int l = t & 1; t >>= 1;
        const int h = (int)t;
        const int row = rb * 32 + (lane & 31);
        const int k = kb * 32 + mfma32_row(rg * 4 + i, lane >> 5);
        v = raw.w[h][1 + l][(size_t)row * HEAD_HID + k];
    } else if (idx < QF_OFF_BIAS) {  // layer 4: [h][kb][rg][lane][4]
        size_t t = idx - QF_OFF_L4;
        const int i = t & 3; t >>= 2;
        const int lane = t & 63; t >>= 6;
        const int rg = t & 3; t >>= 2;
        const int kb = t & 3; t >>= 2;
        const int h = (int)t;
        const int row = lane & 31;
        const int k = kb * 32 + mfma32_row(rg * 4 + i, lane >> 5);
        if (row < head_out_dim(h)) v = raw.w[h][3][(size_t)row * HEAD_HID + k];
    } else if (idx < QB_OFF_L4T) {  // bias fragments: [h][layer][rb][half][16]
        size_t t = idx - QF_OFF_BIAS;
        const int r = t & 15; t >>= 4;
        const int half = t & 1; t >>= 1;
        const int rb = t & 3; t >>= 2;
        const int layer = t & 3; t >>= 2;
        const int h = (int)t;
        const int ch = rb * 32 + mfma32_row(r, half);
        const int dim = (layer == 3) ? head_out_dim(h) : HEAD_HID;
        if (ch < dim) v = raw.b[h][layer][ch];
    } else if (idx < QB_OFF_L32T) {  // W4^T: [h][sg][rb][lane][4], k = 2*(sg*4+i) + half
        size_t t = idx - QB_OFF_L4T;
        const int i = t & 3; t >>= 2;
        const int lane = t & 63; t >>= 6;
        const int rb = t & 3; t >>= 2;
        const int sg = t & 3; t >>= 2;
        const int h = (int)t;
        const int k = 2 * (sg * 4 + i) + (lane >> 5);
        const int in = rb * 32 + (lane & 31);
        if (k < head_out_dim(h)) v = raw.w[h][3][(size_t)k * HEAD_HID + in];
    } else if (idx < QB_OFF_L1T) {  // W3^T (j=0), W2^T (j=1): [h][j][kb][rg][rb][lane][4]
        size_t t = idx - QB_OFF_L32T;
        const int i = t & 3; t >>= 2;
        const int lane = t & 63; t >>= 6;
        const int rb = t & 3; t >>= 2;
        const int rg = t & 3; t >>= 2;
        const int kb = t & 3; t >>= 2;
        const int j = t & 1; t >>= 1;
        const int h = (int)t;
        const int k = kb * 32 + mfma32_row(rg * 4 + i, lane >> 5);
        const int in = rb * 32 + (lane & 31);
        v = raw.w[h][2 - j][(size_t)k * HEAD_HID + in];
    } else {  // W1^T: [h][kb][rg][rb(11)][lane][4]
        size_t t = idx - QB_OFF_L1T;
        const int i = t & 3; t >>= 2;
        const int lane = t & 63; t >>= 6;
        const int rb = (int)(t % QB_RB1); t /= QB_RB1;
        const int rg = t & 3; t >>= 2;
        const int kb = t & 3; t >>= 2;
        const int h = (int)t;
        const int k = kb * 32 + mfma32_row(rg * 4 + i, lane >> 5);
        const int in = rb * 32 + (lane & 31);
        if (in < HEAD_IN) v = raw.w[h][0][(size_t)k * HEAD_IN + in];
    }
    arena[idx] = v;
}

// fp16 x 3 fragments (heads_x3.h): one thread per 16-byte vector of a hi plane, writes the hi and the lo vector
__global__ void heads_pack_x3_kernel(HeadsRaw raw, u32x4* dst) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n1 = QX_L1_VEC / 2, n23 = QX_L23_VEC / 2, n4 = QX_L4_VEC / 2;
    const size_t n4t = QX_L4T_VEC / 2, n32t = QX_L32T_VEC / 2, n1t = QX_L1T_VEC / 2;
    if (idx >= n1 + n23 + n4 + n4t + n32t + n1t) return;
    float v[8];
    size_t out;                                   // index of the hi vector; the lo vector follows 64 vectors later
    size_t t = idx;
    const int lane = t & 63, half = lane >> 5;
    if (idx < n1) {                               // [head][ks][rb][lane]
        t >>= 6;
        const int rb = t & 3; t >>= 2;
        const int ks = (int)(t % QX_KS1);
        const int hd = (int)(t / QX_KS1);
        const int row = rb * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + 8 * half + j;
            v[j] = k < HEAD_IN ? raw.w[hd][0][(size_t)row * HEAD_IN + k] : 0.f;
        }
        out = (((size_t)hd * QX_KS1 + ks) * 4 + rb) * 2 * 64 + lane;
    } else if (idx < n1 + n23) {                  // [head][l][kb][s][rb][lane]
        t = (idx - n1) >> 6;
        const int rb = t & 3; t >>= 2;
        const int s = t & 1; t >>= 1;
        const int kb = t & 3; t >>= 2;
        const int l = t & 1; t >>= 1;
        const int hd = (int)t;
        const int row = rb * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = raw.w[hd][1 + l][(size_t)row * HEAD_HID + kb * 32 + mfma32_row(8 * s + j, half)];
        out = QX_L1_VEC + ((((((size_t)hd * 2 + l) * 4 + kb) * 2 + s) * 4 + rb) * 2) * 64 + lane;
    } else if (idx < n1 + n23 + n4) {             // [head][kb][s][lane]
        t = (idx - n1 - n23) >> 6;
        const int s = t & 1; t >>= 1;
        const int kb = t & 3; t >>= 2;
        const int hd = (int)t;
        const int row = lane & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = row < head_out_dim(hd) ? raw.w[hd][3][(size_t)row * HEAD_HID + kb * 32 + mfma32_row(8 * s + j, half)] : 0.f;
        out = QX_L1_VEC + QX_L23_VEC + ((((size_t)hd * 4 + kb) * 2 + s) * 2) * 64 + lane;
    } else if (idx < n1 + n23 + n4 + n4t) {       // W4^T: [head][rb][lane], k = 8 half + j
        t = (idx - n1 - n23 - n4) >> 6;
        const int rb = t & 3; t >>= 2;
        const int hd = (int)t;
        const int in = rb * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * half + j;
            v[j] = k < head_out_dim(hd) ? raw.w[hd][3][(size_t)k * HEAD_HID + in] : 0.f;
        }
        out = QX_OFF_L4T + (((size_t)hd * 4 + rb) * 2) * 64 + lane;
    } else if (idx < n1 + n23 + n4 + n4t + n32t) {   // W3^T (j = 0), W2^T (j = 1): [head][j][kb][s][rb][lane]
        t = (idx - n1 - n23 - n4 - n4t) >> 6;
        const int rb = t & 3; t >>= 2;
        const int s = t & 1; t >>= 1;
        const int kb = t & 3; t >>= 2;
        const int jl = t & 1; t >>= 1;
        const int hd = (int)t;
        const int in = rb * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = raw.w[hd][2 - jl][(size_t)(kb * 32 + mfma32_row(8 * s + j, half)) * HEAD_HID + in];
        out = QX_OFF_L32T + ((((((size_t)hd * 2 + jl) * 4 + kb) * 2 + s) * 4 + rb) * 2) * 64 + lane;
    } else {                                      // W1^T: [head][kb][s][rb(11)][lane]
        t = (idx - n1 - n23 - n4 - n4t - n32t) >> 6;
        const int rb = (int)(t % QB_RB1); t /= QB_RB1;
        const int s = t & 1; t >>= 1;
        const int kb = t & 3; t >>= 2;
        const int hd = (int)t;
        const int in = rb * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = in < HEAD_IN ? raw.w[hd][0][(size_t)(kb * 32 + mfma32_row(8 * s + j, half)) * HEAD_IN + in] : 0.f;
        out = QX_OFF_L1T + (((((size_t)hd * 4 + kb) * 2 + s) * QB_RB1 + rb) * 2) * 64 + lane;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= QX_SCALE;
    u32x4 hi, lo;
    split8(v, hi, lo);
    dst[out] = hi;
    dst[out + 64] = lo;
}

int launch_heads_pack_f32(chore_handle* h, const HeadsRaw& raw, float* arena, hipStream_t s) {
    const int threads = 256;
    const int blocks = (int)((QF_TOTAL_FLOATS + threads - 1) / threads);
    hipLaunchKernelGGL(heads_pack_f32_kernel, dim3(blocks), dim3(threads), 0, s, raw, arena);
    CHORE_LAUNCH_CHECK(h, s);
    const size_t nx = QX_TOTAL_VEC / 2;
    hipLaunchKernelGGL(heads_pack_x3_kernel, dim3((unsigned)((nx + threads - 1) / threads)), dim3(threads), 0, s, raw,
                       (u32x4*)((char*)arena + QX_OFF_BYTES));
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

size_t heads_arena_bytes() { return QX_ARENA_BYTES; }

template <typename T, int NCB, bool X3 = false>
static int launch_query_fwd_n(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    constexpr int PTS = 32 * NCB;
    const size_t smem = sizeof(QueryFwdSmemT<PTS>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_fwd_f32_kernel<T, NCB, false, X3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + PTS - 1) / PTS, a.B);
    hipLaunchKernelGGL((query_fwd_f32_kernel<T, NCB, false, X3>), grid, dim3(256), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// 64-point tiles unless they would leave CUs without a workgroup
bool query_small_tiles(int B, int N) { return (size_t)B * ((N + QT_PTS - 1) / QT_PTS) <= 256; }

template <typename T, bool X3 = false>
static int launch_query_fwd_w8(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    const size_t smem = sizeof(QueryFwdSmemT<64>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_fwd_f32_w8_kernel<T, false, X3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + 63) / 64, a.B);
    hipLaunchKernelGGL((query_fwd_f32_w8_kernel<T, false, X3>), grid, dim3(512), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

template <typename T, bool X3 = false>
static int launch_query_fwd_t(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    static const bool w4 = getenv("CHORE_QUERY_W4") != nullptr;     // A/B switches for large queries
    static const bool w8 = getenv("CHORE_QUERY_W8") != nullptr;
    if constexpr (X3) {
        static const bool nosplit = getenv("CHORE_QUERY_X3_NOSPLIT") != nullptr;      // A/B switch: the one-wave-per-head kernels
        if (!nosplit) return query_small_tiles(a.B, a.N) ? launch_query_fwd_split<T, 1>(h, a, s) : launch_query_fwd_split<T, 2>(h, a, s);
    }
    if (query_small_tiles(a.B, a.N)) return launch_query_fwd_n<T, 1, X3>(h, a, s);
    // (a single output asked for, CHORE.query_df: 32-point tiles do not pay -- a workgroup is placed with the registers of all
    // four waves, so the CU holds one whatever the three leaving waves free: 363 against 335 us at 8 x 20 000 points)
    // fp16 x 3: a k-step of MFMAs is 2.7 x shorter than the fp32 one for the same weight bytes, and the eight-wave kernel
    // (both waves of a head fetch the head's fragments) is bound by the L1's 64 B / clk: 0.227 ms against 0.203 ms for
    // four waves with two column blocks each (4 x 20 000 points)
    if (X3 ? !w8 : w4) return launch_query_fwd_n<T, 2, X3>(h, a, s);
    return launch_query_fwd_w8<T, X3>(h, a, s);
}

template <typename T, bool X3 = false>
static int launch_query_fwd_train_w8(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    bool& attr_set = CHORE_ONCE_FLAG(h);
    const size_t smem = sizeof(QueryFwdSmemT<64>) + 8 * 32 * ST_LD * sizeof(float);     // + the waves' store-transpose tiles
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_fwd_f32_w8_kernel<T, true, X3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + 63) / 64, a.B);
    hipLaunchKernelGGL((query_fwd_f32_w8_kernel<T, true, X3>), grid, dim3(512), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

template <typename T, bool X3 = false>
static int launch_query_fwd_train_t(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    if (!getenv("CHORE_QUERY_W4")) return launch_query_fwd_train_w8<T, X3>(h, a, s);
    bool& attr_set = CHORE_ONCE_FLAG(h);
    const size_t smem = sizeof(QueryFwdSmemT<64>);
    if (!attr_set) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_fwd_f32_kernel<T, 2, true, X3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((a.N + 63) / 64, a.B);
    hipLaunchKernelGGL((query_fwd_f32_kernel<T, 2, true, X3>), grid, dim3(256), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}
// x3: the heads on the fp16 matrix cores with split operands (either map type)
int launch_query_fwd_train(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s, int x3) {
    if (x3) return dtype == CHORE_F32 ? launch_query_fwd_train_t<float, true>(h, a, s) : launch_query_fwd_train_t<unsigned short, true>(h, a, s);
    return dtype == CHORE_F32 ? launch_query_fwd_train_t<float>(h, a, s) : launch_query_fwd_train_t<unsigned short>(h, a, s);
}

int launch_query_fwd_f32(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    return launch_query_fwd_t<float>(h, a, s);
}
int launch_query_fwd_f32_bf16maps(chore_handle* h, const QueryArgs& a, hipStream_t s) {
    return launch_query_fwd_t<unsigned short>(h, a, s);
}
int launch_query_fwd_x3(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s) {      // dtype: the maps' type
    if (dtype == CHORE_F16) return launch_query_fwd_t<qh16_t, true>(h, a, s);
    return dtype == CHORE_F32 ? launch_query_fwd_t<float, true>(h, a, s) : launch_query_fwd_t<unsigned short, true>(h, a, s);
}

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

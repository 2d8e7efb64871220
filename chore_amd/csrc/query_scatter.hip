// query_scatter.hip -- gradient of the pixel-aligned feature sample with respect to the feature maps.
//
// The transpose of `index` (model/geometry.py:4-14, i.e. the backward of grid_sample that the reference's autograd
// runs when CHORE.forward is trained, model/chore.py:107-154): every point adds w_k * d(323-vector)[c] to its four
// taps of the hourglass feature map (channels 0..255) and of tmpx (channels 259..322).  ATen scatters with float
// atomics (order-dependent results).  Here the map is cut into tiles that fit LDS (8x8 texels x 256 channels /
// 16x16 x 64 = 64 KB); the workgroup of a tile scans the points of its image in index order, keeps those with a tap
// inside the tile (ballot-ordered compaction, so the list is sorted) and accumulates them one after the other with
// one thread per channel: no atomics anywhere, the result is bit-reproducible, and each tile is written once.
#include "query_common.h"

namespace {

constexpr int SC_SUB = 2;                  // sub-batches of 256 points per round (4 would need 93 KB of LDS: one workgroup per CU, 416 us against 256)
constexpr int SC_LIST = 256 * SC_SUB;      // points examined per round (= the capacity of the hit list)

struct Hit { int pt; int x0, y0; float w[4]; };

template <int C, int TS /*tile edge in texels*/>
__global__ __launch_bounds__(256) void scatter_kernel(QueryArgs a, const float* __restrict__ dX, int xoff, int H, int W,
                                                      float* __restrict__ dmap, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [TS*TS][C]
    __shared__ Hit hits[SC_LIST];
    __shared__ int wave_cnt[4 * SC_SUB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.z, tx0 = blockIdx.x * TS, ty0 = blockIdx.y * TS;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};
    for (int i = tid; i < TS * TS * C; i += 256) acc[i] = 0.f;
    __syncthreads();
    for (int base = 0; base < a.N; base += SC_LIST) {
        // SC_SUB x 256 points per round, their coordinates requested together: a round costs a global round trip and two
        // barriers whatever it holds, and 20 000 points in rounds of 256 made that 79 times per tile
        bool keep[SC_SUB];
        Hit hme[SC_SUB];
        float px[SC_SUB], py[SC_SUB], pz[SC_SUB];
#pragma unroll
        for (int j = 0; j < SC_SUB; ++j) {
            const int n = base + j * 256 + tid;
            const float* p = a.points + ((size_t)b * a.N + (n < a.N ? n : a.N - 1)) * 3;
            px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
        }
#pragma unroll
        for (int j = 0; j < SC_SUB; ++j) {
            const int n = base + j * 256 + tid;
            keep[j] = false;
            if (n < a.N) {
                float nx, ny;
                project_point(px[j], py[j], pz[j], a.crop_center[b * 2 + 0], a.crop_center[b * 2 + 1], cam, nx, ny);
                // same tap arithmetic as make_taps (query_common.h); x0/y0 kept instead of flat offsets
                const float ix = __fmul_rn(__fadd_rn(nx, 1.0f), (float)(W - 1) / 2);
                const float iy = __fmul_rn(__fadd_rn(ny, 1.0f), (float)(H - 1) / 2);
                const bool sane = (ix > -2.0f) && (ix < (float)W + 1.0f) && (iy > -2.0f) && (iy < (float)H + 1.0f);
                if (sane) {
                    const float x0f = floorf(ix), y0f = floorf(iy);
                    const float w = __fsub_rn(ix, x0f), e = __fsub_rn(1.0f, w);
                    const float nn = __fsub_rn(iy, y0f), s = __fsub_rn(1.0f, nn);
                    hme[j].pt = n; hme[j].x0 = (int)x0f; hme[j].y0 = (int)y0f;
                    hme[j].w[0] = __fmul_rn(e, s); hme[j].w[1] = __fmul_rn(w, s); hme[j].w[2] = __fmul_rn(e, nn); hme[j].w[3] = __fmul_rn(w, nn);
                    keep[j] = hme[j].x0 + 1 >= tx0 && hme[j].x0 < tx0 + TS && hme[j].y0 + 1 >= ty0 && hme[j].y0 < ty0 + TS;
                }
            }
        }
        // ordered compaction: position = number of kept points with a smaller index (sub-batch major, then wave, then lane)
        unsigned long long m[SC_SUB];
#pragma unroll
        for (int j = 0; j < SC_SUB; ++j) {
            m[j] = __ballot(keep[j]);
            if (lane == 0) wave_cnt[j * 4 + wid] = __popcll(m[j]);
        }
        __syncthreads();
        int total = 0;
#pragma unroll
        for (int j = 0; j < SC_SUB; ++j) {
            int off = total;
            for (int w = 0; w < 4; ++w) { if (w < wid) off += wave_cnt[j * 4 + w]; total += wave_cnt[j * 4 + w]; }
            if (keep[j]) hits[off + __popcll(m[j] & ((1ull << lane) - 1ull))] = hme[j];
        }
        __syncthreads();
        // one channel per thread (C <= 256); the next hit's gradient value is requested while this one is added (the hits are
        // processed in order -- a load per tap inside the loop made every hit a global round trip)
        const int c = tid;
        auto gload = [&](int i) -> float {
            return (i < total && c < C) ? dX[((size_t)b * a.N + hits[i].pt) * QF_KPAD + xoff + c] : 0.f;
        };
        float gnext = gload(0), gnext2 = gload(1);
        for (int i = 0; i < total; ++i) {
            const Hit hh = hits[i];
            const float gv = gnext;
            gnext = gnext2;
            gnext2 = gload(i + 2);
            if (c < C) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = hh.x0 + (k & 1), y = hh.y0 + (k >> 1);
                    if (x < tx0 || x >= tx0 + TS || y < ty0 || y >= ty0 + TS || x >= W || y >= H || x < 0 || y < 0) continue;
                    float* cell = acc + ((y - ty0) * TS + (x - tx0)) * C;
                    cell[c] = fmaf(hh.w[k], gv, cell[c]);
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < TS * TS * (C / 4); i += 256) {
        const int cell = i / (C / 4), q = i % (C / 4);
        const int x = tx0 + cell % TS, y = ty0 + cell / TS;
        if (x >= W || y >= H) continue;
        float* o = dmap + (((size_t)b * H + y) * W + x) * C + 4 * q;
        f32x4 v = *(const f32x4*)(acc + cell * C + 4 * q);
        if (accumulate) { const f32x4 old = *(const f32x4*)o; v += old; }
        *(f32x4*)o = v;
    }
}

}  // namespace

int launch_scatter_features(chore_handle* h, const QueryArgs& a, const float* dX, float* dfeat, float* dtmpx,
                            int accumulate, hipStream_t s) {
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)scatter_kernel<FEAT_C, 8>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8 * FEAT_C * 4));
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)scatter_kernel<TMPX_C, 16>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 16 * TMPX_C * 4));
        attr = true;
    }
    if (dfeat) {
        dim3 grid((a.FW + 7) / 8, (a.FH + 7) / 8, a.B);
        hipLaunchKernelGGL((scatter_kernel<FEAT_C, 8>), grid, dim3(256), 8 * 8 * FEAT_C * 4, s, a, dX, 0, a.FH, a.FW, dfeat,
                           accumulate);
    }
    if (dtmpx) {
        dim3 grid((a.TW + 15) / 16, (a.TH + 15) / 16, a.B);
        hipLaunchKernelGGL((scatter_kernel<TMPX_C, 16>), grid, dim3(256), 16 * 16 * TMPX_C * 4, s, a, dX, FEAT_C + 3, a.TH,
                           a.TW, dtmpx, accumulate);
    }
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

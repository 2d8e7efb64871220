// query_scatter.hip -- gradient of the pixel-aligned feature sample with respect to the feature maps.
//
// The transpose of `index` (model/geometry.py:4-14, i.e. the backward of grid_sample that the reference's autograd
// runs when CHORE.forward is trained, model/chore.py:107-154): every point adds w_k * d(323-vector)[c] to its four
// taps of the hourglass feature map (channels 0..255) and of tmpx (channels 259..322).  ATen scatters with float
// atomics (order-dependent results).  Here the map is cut into tiles that fit LDS (8x8 texels x 256 channels /
// 16x16 x 64 = 64 KB); the workgroup of a tile adds the points that have a tap inside the tile one after the other, in point
// index order, with one thread per channel: no atomics anywhere, the result is bit-reproducible, and each tile is written once.
//
// Which points a tile has: scatter_bin_kernel cuts the image's points into <= 64 contiguous chunks and sorts every chunk's
// points by the tile(s) their taps fall into -- a STABLE counting sort, one wave per chunk, a point entered once per tile it
// touches (1, 2 or 4) -- into the staging buffer's sort region; a tile's workgroup then walks the chunks' segments of its tile
// in chunk order = in point-index order.  (Rounds 1 - 3 had every tile scan all points of its image: 20 000 points x 256 tiles =
// 5 M projections per image, 258 us.  Round 4 kept that kernel for fewer than 64 points and for maps of more than 256 tiles; it
// had been seen to produce a wrong tile about once in 700 calls when two processes shared the GPU and nothing was ever found
// wrong in it.  Round 5 REMOVED it: the sort handles any point count, and a map of more than 256 tiles is processed in windows
// of <= 16 x 16 tiles, one sort + one walk per window -- a tile's list does not depend on the window it is in, so the sums are
// the same bit for bit whatever the window size: tests/test_gpu_query.py.)
#include "query_common.h"

namespace {

constexpr int SC_SUB = 2;                  // sub-batches of 256 hits per round
constexpr int SC_LIST = 256 * SC_SUB;      // the capacity of the hit list

struct Hit { int pt; int x0, y0; float w[4]; };

// ---------------------------------------------------------------------------------------------------------------------------
// binned scatter
// ---------------------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int scatter_chunk(int N) { const int c = (N + 63) / 64; return (c + 63) / 64 * 64; }

// tap geometry of one point in an (H,W) map: the arithmetic of make_taps (query_common.h); false = no tap anywhere near the map
__device__ __forceinline__ bool scatter_hit(const QueryArgs& a, const Cam& cam, int b, int n, int H, int W, Hit& h) {
    const float* p = a.points + ((size_t)b * a.N + n) * 3;
    float nx, ny;
    project_point(p[0], p[1], p[2], a.crop_center[b * 2 + 0], a.crop_center[b * 2 + 1], cam, nx, ny);
    const float ix = __fmul_rn(__fadd_rn(nx, 1.0f), (float)(W - 1) / 2);
    const float iy = __fmul_rn(__fadd_rn(ny, 1.0f), (float)(H - 1) / 2);
    const bool sane = (ix > -2.0f) && (ix < (float)W + 1.0f) && (iy > -2.0f) && (iy < (float)H + 1.0f);
    if (!sane) return false;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float w = __fsub_rn(ix, x0f), e = __fsub_rn(1.0f, w);
    const float nn = __fsub_rn(iy, y0f), s = __fsub_rn(1.0f, nn);
    h.pt = n; h.x0 = (int)x0f; h.y0 = (int)y0f;
    h.w[0] = __fmul_rn(e, s); h.w[1] = __fmul_rn(w, s); h.w[2] = __fmul_rn(e, nn); h.w[3] = __fmul_rn(w, nn);
    return true;
}

// a window of the map's tiles: origin (wx, wy) and extent (wtx, wty) in tiles, wtx * wty <= 256
struct TileWin { int wx, wy, wtx, wty; };

// the (at most four, distinct) tiles OF THE WINDOW a point's taps fall into (numbered inside the window); -1 = unused slot
template <int TS>
__device__ __forceinline__ void scatter_bins(const QueryArgs& a, const Cam& cam, int b, int n, bool live, int H, int W, const TileWin& tw, int kb[4]) {
    kb[0] = kb[1] = kb[2] = kb[3] = -1;
    Hit h;
    if (!live || !scatter_hit(a, cam, b, n, H, W, h)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = h.x0 + (k & 1), y = h.y0 + (k >> 1);
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        const int tx = x / TS - tw.wx, ty = y / TS - tw.wy;
        if (tx < 0 || tx >= tw.wtx || ty < 0 || ty >= tw.wty) continue;
        const int bin = ty * tw.wtx + tx;
        bool dup = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) dup = dup || (j < k && kb[j] == bin);
        if (!dup) kb[k] = bin;
    }
}

// one wave per (chunk, image): stable counting sort of the chunk's points by tile.  Lane l owns the l-th contiguous run of the
// chunk's points, so "stable" = by lane, then by position in the lane's run: a table of per-(tile, lane) counts (no atomics, no
// cross-lane ranking), its prefix over the lanes, the prefix over the tiles, and a second walk that places the entries.
// sort region of the image:  lists [G][4 * CH] point indices, then offsets [G][SCATTER_OFF_STRIDE] (entry nb = the chunk's total)
constexpr int SB_ROW = 66;      // ushort counters per tile row: 64 lanes + 2 of padding (rows 33 words apart: the per-tile walk
                                // of lane j over its 5 rows touches 32 different banks across a half-wave)
template <int TS>
__global__ __launch_bounds__(64) void scatter_bin_kernel(QueryArgs a, int H, int W, int* __restrict__ sort, TileWin tw) {
    __shared__ unsigned short cnt[(SCATTER_OFF_STRIDE + 60) * SB_ROW];     // [320 tiles][lane]
    __shared__ int tot[64];
    const int lane = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};
    const int CH = scatter_chunk(a.N), G = (a.N + CH - 1) / CH, S = CH / 64;
    const int nb = tw.wtx * tw.wty;
    int* img = sort + (size_t)b * scatter_sort_ints(a.N);
    int* list = img + (size_t)g * 4 * CH;
    int* off = img + (size_t)G * 4 * CH + (size_t)g * SCATTER_OFF_STRIDE;
    const int n0 = g * CH + lane * S, n1 = min(a.N, n0 + S);
    for (int i = lane; i < 320 * SB_ROW / 2; i += 64) ((unsigned*)cnt)[i] = 0u;
    __syncthreads();
    // ---- counts: this lane's column ----
    for (int n = n0; n < n1; ++n) {
        int kb[4];
        scatter_bins<TS>(a, cam, b, n, true, H, W, tw, kb);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (kb[k] >= 0) cnt[kb[k] * SB_ROW + lane] += 1;
    }
    __syncthreads();
    // ---- lane j owns tiles 5 j .. 5 j + 4: prefix over the 64 lane counters of each, then the prefix over the tiles ----
    int t5[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        unsigned short* row = cnt + (lane * 5 + j) * SB_ROW;
        int run = 0;
        for (int i = 0; i < 64; ++i) { const int v = row[i]; row[i] = (unsigned short)run; run += v; }
        t5[j] = run;
    }
    tot[lane] = t5[0] + t5[1] + t5[2] + t5[3] + t5[4];
    __syncthreads();
    int base = 0;
    for (int i = 0; i < 64; ++i) base += i < lane ? tot[i] : 0;
    __syncthreads();
    int tb[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int k = lane * 5 + j;
        tb[j] = base;
        if (k <= nb) off[k] = base;                    // entry nb = the total (tiles >= nb are empty)
        base += t5[j];
    }
    // the tiles' bases go where every lane can read them: the two padding counters of the tile's row (low / high half)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        unsigned short* row = cnt + (lane * 5 + j) * SB_ROW;
        row[64] = (unsigned short)(tb[j] & 0xffff);
        row[65] = (unsigned short)(tb[j] >> 16);
    }
    __syncthreads();
    // ---- placement: the lane walks its run again ----
    for (int n = n0; n < n1; ++n) {
        int kb[4];
        scatter_bins<TS>(a, cam, b, n, true, H, W, tw, kb);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (kb[k] < 0) continue;
            unsigned short* row = cnt + kb[k] * SB_ROW;
            const int r = row[lane];
            row[lane] = (unsigned short)(r + 1);
            list[((int)row[64] | ((int)row[65] << 16)) + r] = n;
        }
    }
}

// A workgroup owns a SUB x SUB block of texels inside a TS x TS tile: it walks the tile's whole list and adds the hits that touch
// its block.  SUB = TS for the feature map; SUB < TS (more, smaller workgroups walking the same list) measured slower for it
// (197 against 134 us inside the training step: every workgroup pays the list set-up) -- profiles/r04_scatter.txt.
template <int C, int TS /*tile edge in texels*/, int SUB /*texels per workgroup edge*/>
__global__ __launch_bounds__(256) void scatter_csr_kernel(QueryArgs a, const float* __restrict__ dX, int xoff, int H, int W,
                                                          float* __restrict__ dmap, int accumulate, const int* __restrict__ sort,
                                                          TileWin tw) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [SUB*SUB][C]
    __shared__ Hit hits[SC_LIST];
    __shared__ int seg_start[65], seg_src[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.z, tx0 = tw.wx * TS + blockIdx.x * SUB, ty0 = tw.wy * TS + blockIdx.y * SUB;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};
    const int CH = scatter_chunk(a.N), G = (a.N + CH - 1) / CH;
    const int bin = (ty0 / TS - tw.wy) * tw.wtx + (tx0 / TS - tw.wx);
    const int* img = sort + (size_t)b * scatter_sort_ints(a.N);
    const int* offs = img + (size_t)G * 4 * CH;
    for (int i = tid; i < SUB * SUB * C; i += 256) acc[i] = 0.f;
    if (tid < 64) {      // the tile's segment of every chunk, in chunk order
        int o0 = 0, len = 0;
        if (lane < G) { o0 = offs[(size_t)lane * SCATTER_OFF_STRIDE + bin]; len = offs[(size_t)lane * SCATTER_OFF_STRIDE + bin + 1] - o0; }
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        seg_start[lane] = incl - len;
        seg_src[lane] = lane * 4 * CH + o0;
        if (lane == 63) seg_start[64] = incl;
    }
    __syncthreads();
    const int total_all = seg_start[64];
    for (int base = 0; base < total_all; base += SC_LIST) {
        const int total = min(SC_LIST, total_all - base);
#pragma unroll
        for (int j = 0; j < SC_SUB; ++j) {
            const int i = j * 256 + tid;
            if (i < total) {
                const int gi = base + i;
                int lo = 0;                          // the last chunk whose segment starts at or before gi
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) if (lo + step < 64 && seg_start[lo + step] <= gi) lo += step;
                const int n = img[seg_src[lo] + (gi - seg_start[lo])];
                Hit h;
                scatter_hit(a, cam, b, n, H, W, h);  // was sane when it was binned
                hits[i] = h;
            }
        }
        __syncthreads();
        // one channel per thread; the hits are added in order, their gradient values fetched SC_BATCH hits ahead (a row per hit:
        // with two loads in flight the walk was a chain of global round trips, 0.7 us per hit)
        const int c = tid;
        constexpr int SC_BATCH = 32;
        // (unconditional loads at a clamped index: a guard per load made every address a separate LDS round trip)
        const float* dXc = dX + (size_t)b * a.N * QF_KPAD + xoff + (c < C ? c : 0);
        auto gload = [&](int i) -> float { return dXc[(size_t)hits[min(i, total - 1)].pt * QF_KPAD]; };
        float cur[SC_BATCH], nxt[SC_BATCH];
#pragma unroll
        for (int u = 0; u < SC_BATCH; ++u) cur[u] = gload(u);
        for (int i0 = 0; i0 < total; i0 += SC_BATCH) {
#pragma unroll
            for (int u = 0; u < SC_BATCH; ++u) nxt[u] = gload(i0 + SC_BATCH + u);
            Hit hn = hits[i0];                 // the next hit's record is read while this one's cells make their round trip
#pragma unroll
            for (int u = 0; u < SC_BATCH; ++u) {
                if (i0 + u >= total) break;
                // the hit is the same for every thread: its geometry goes to scalar registers (uniform branches, scalar cell
                // addresses); the four taps are four different cells, so their reads are issued together and the writes after --
                // one LDS round trip per hit instead of four dependent ones
                const Hit hv = hn;
                hn = hits[min(i0 + u + 1, total - 1)];
                const int hx = __builtin_amdgcn_readfirstlane(hv.x0) - tx0, hy = __builtin_amdgcn_readfirstlane(hv.y0) - ty0;
                const float gv = cur[u];
                float* cell[4];
                bool ok[4];
                float old[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = hx + (k & 1), y = hy + (k >> 1);
                    ok[k] = c < C && x >= 0 && x < SUB && y >= 0 && y < SUB && x + tx0 < W && y + ty0 < H;
                    cell[k] = acc + (y * SUB + x) * C + c;
                    old[k] = ok[k] ? *cell[k] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ok[k]) *cell[k] = fmaf(hv.w[k], gv, old[k]);
            }
#pragma unroll
            for (int u = 0; u < SC_BATCH; ++u) cur[u] = nxt[u];
        }
        __syncthreads();
    }
    for (int i = tid; i < SUB * SUB * (C / 4); i += 256) {
        const int cell = i / (C / 4), q = i % (C / 4);
        const int x = tx0 + cell % SUB, y = ty0 + cell / SUB;
        if (x >= W || y >= H) continue;
        float* o = dmap + (((size_t)b * H + y) * W + x) * C + 4 * q;
        f32x4 v = *(const f32x4*)(acc + cell * C + 4 * q);
        if (accumulate) { const f32x4 old = *(const f32x4*)o; v += old; }
        *(f32x4*)o = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// sorted-order forward (chore_query_fwd_ws): a permutation that puts points whose samples fall into the same 8 x 8-texel tile of
// the feature map next to each other.  The forward's gather then reads a texel row once per tile of 64 points instead of once
// per point (unsorted: every one of the 4 taps x 80 000 points a separate 1.25 KB fetch, 2.9 x the compulsory bytes).
// Step 1: scatter_bin_kernel's machinery with ONE entry per point (its north-west tap's tile, clamped into the map; points
// nowhere near the map go to a tile of their own).  Step 2: the chunks' segments concatenated tile by tile.
// ---------------------------------------------------------------------------------------------------------------------------
template <int TS>
__device__ __forceinline__ int query_tile_of(const QueryArgs& a, const Cam& cam, int b, int n, int H, int W, int txn, int nb) {
    Hit h;
    if (!scatter_hit(a, cam, b, n, H, W, h)) return nb;
    const int x = min(max(h.x0, 0), W - 1), y = min(max(h.y0, 0), H - 1);
    return (y / TS) * txn + x / TS;
}

template <int TS>
__global__ __launch_bounds__(64) void query_bin_kernel(QueryArgs a, int H, int W, int* __restrict__ work) {
    __shared__ unsigned short cnt[(SCATTER_OFF_STRIDE + 60) * SB_ROW];
    __shared__ int tot[64];
    const int lane = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const Cam cam{a.fx, a.fy, a.cx, a.cy, a.half_crop, a.crop};
    const int CH = scatter_chunk(a.N), G = (a.N + CH - 1) / CH, S = CH / 64;
    const int txn = (W + TS - 1) / TS, nb = txn * ((H + TS - 1) / TS);        // tile nb = "no tap near the map"
    int* img = work + (size_t)a.B * a.N + (size_t)b * scatter_sort_ints(a.N);      // [B][N] permutations first, then the images' sort regions
    int* list = img + (size_t)g * 4 * CH;
    int* off = img + (size_t)G * 4 * CH + (size_t)g * SCATTER_OFF_STRIDE;
    const int n0 = g * CH + lane * S, n1 = min(a.N, n0 + S);
    for (int i = lane; i < 320 * SB_ROW / 2; i += 64) ((unsigned*)cnt)[i] = 0u;
    __syncthreads();
    for (int n = n0; n < n1; ++n) cnt[query_tile_of<TS>(a, cam, b, n, H, W, txn, nb) * SB_ROW + lane] += 1;
    __syncthreads();
    int t5[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        unsigned short* row = cnt + (lane * 5 + j) * SB_ROW;
        int run = 0;
        for (int i = 0; i < 64; ++i) { const int v = row[i]; row[i] = (unsigned short)run; run += v; }
        t5[j] = run;
    }
    tot[lane] = t5[0] + t5[1] + t5[2] + t5[3] + t5[4];
    __syncthreads();
    int base = 0;
    for (int i = 0; i < 64; ++i) base += i < lane ? tot[i] : 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int k = lane * 5 + j;
        unsigned short* row = cnt + k * SB_ROW;
        row[64] = (unsigned short)(base & 0xffff);
        row[65] = (unsigned short)(base >> 16);
        if (k <= nb + 1) off[k] = base;                // entries 0 .. nb: the tiles; nb + 1: the total
        base += t5[j];
    }
    __syncthreads();
    for (int n = n0; n < n1; ++n) {
        unsigned short* row = cnt + query_tile_of<TS>(a, cam, b, n, H, W, txn, nb) * SB_ROW;
        const int r = row[lane];
        row[lane] = (unsigned short)(r + 1);
        list[((int)row[64] | ((int)row[65] << 16)) + r] = n;
    }
}

// gridDim.x workgroups per image: perm = for every tile, the chunks' segments of that tile in chunk order (every workgroup builds
// the whole table of cell starts, then copies its share of the cells)
__global__ __launch_bounds__(1024) void query_perm_kernel(QueryArgs a, int nb, int* __restrict__ work) {
    extern __shared__ int offs[];                       // [G][SCATTER_OFF_STRIDE] the chunks' offset tables, then [nb + 2] tile bases
    const int tid = threadIdx.x, b = blockIdx.y;
    const int CH = scatter_chunk(a.N), G = (a.N + CH - 1) / CH;
    int* perm = work + (size_t)b * a.N;
    const int* img = work + (size_t)a.B * a.N + (size_t)b * scatter_sort_ints(a.N);
    const int* goff = img + (size_t)G * 4 * CH;
    int* tbase = offs + G * SCATTER_OFF_STRIDE;
    for (int i = tid; i < G * SCATTER_OFF_STRIDE; i += 1024) offs[i] = goff[i];
    __syncthreads();
    // tile k: its total over the chunks
    for (int k = tid; k <= nb; k += 1024) {
        int t = 0;
        for (int g = 0; g < G; ++g) t += offs[g * SCATTER_OFF_STRIDE + k + 1] - offs[g * SCATTER_OFF_STRIDE + k];
        tbase[k + 1] = t;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k <= nb; ++k) { const int t = tbase[k + 1]; tbase[k] = run; run += t; }
    }
    __syncthreads();
    // where every (tile, chunk) cell starts in perm
    int* cstart = tbase + nb + 2;                       // [nb + 1][G]
    for (int k = tid; k <= nb; k += 1024) {
        int start = tbase[k];
        for (int g = 0; g < G; ++g) {
            cstart[k * G + g] = start;
            start += offs[g * SCATTER_OFF_STRIDE + k + 1] - offs[g * SCATTER_OFF_STRIDE + k];
        }
    }
    __syncthreads();
    // the cells, spread over all threads (1.2 entries per cell on average: independent short copies)
    for (int c = tid + 1024 * blockIdx.x; c < (nb + 1) * G; c += 1024 * gridDim.x) {
        const int k = c / G, g = c - k * G;
        const int o0 = offs[g * SCATTER_OFF_STRIDE + k], len = offs[g * SCATTER_OFF_STRIDE + k + 1] - o0;
        const int* src = img + (size_t)g * 4 * CH + o0;
        int* dst = perm + cstart[c];
        for (int e = 0; e < len; ++e) dst[e] = src[e];
    }
}

}  // namespace

int launch_scatter_features(chore_handle* h, const QueryArgs& a, const float* dX, float* dfeat, float* dtmpx,
                            int accumulate, hipStream_t s) {
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)scatter_csr_kernel<FEAT_C, 8, 8>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8 * FEAT_C * 4));
        attr = true;
    }
    if (!a.tSort) CHORE_FAIL(h, CHORE_EINVAL, "scatter_features: the staging buffer has no sort region");
    if (a.N <= 0) return CHORE_OK;
    // windows of the map's tiles: <= 16 x 16 (the sort's count table and offset tables hold 256 tiles); one sort + one walk per
    // window, in stream order (they share the sort region).  CHORE_SCATTER_WINDOW=k: k x k tiles (tests: same bits for any k)
    int win = 16;
    if (const char* e = getenv("CHORE_SCATTER_WINDOW")) { const int v = atoi(e); if (v >= 1 && v <= 16) win = v; }     // read per call: the test changes it
    const int CH = scatter_chunk(a.N), G = (a.N + CH - 1) / CH;
    if (dfeat) {
        const int txn = (a.FW + 7) / 8, tyn = (a.FH + 7) / 8;
        int* sort = a.tSort;
        for (int wy = 0; wy < tyn; wy += win)
            for (int wx = 0; wx < txn; wx += win) {
                const TileWin tw{wx, wy, txn - wx < win ? txn - wx : win, tyn - wy < win ? tyn - wy : win};
                hipLaunchKernelGGL((scatter_bin_kernel<8>), dim3(G, a.B), dim3(64), 0, s, a, a.FH, a.FW, sort, tw);
                hipLaunchKernelGGL((scatter_csr_kernel<FEAT_C, 8, 8>), dim3(tw.wtx, tw.wty, a.B), dim3(256), 8 * 8 * FEAT_C * 4,
                                   s, a, dX, 0, a.FH, a.FW, dfeat, accumulate, (const int*)sort, tw);
            }
    }
    if (dtmpx) {
        const int txn = (a.TW + 15) / 16, tyn = (a.TH + 15) / 16;
        int* sort = a.tSort + (size_t)a.B * scatter_sort_ints(a.N);
        for (int wy = 0; wy < tyn; wy += win)
            for (int wx = 0; wx < txn; wx += win) {
                const TileWin tw{wx, wy, txn - wx < win ? txn - wx : win, tyn - wy < win ? tyn - wy : win};
                hipLaunchKernelGGL((scatter_bin_kernel<16>), dim3(G, a.B), dim3(64), 0, s, a, a.TH, a.TW, sort, tw);
                // 8 x 8-texel workgroups inside the 16 x 16 tiles: two per tile edge, clipped to the map
                const int bx = (a.TW + 7) / 8 - 2 * wx, by = (a.TH + 7) / 8 - 2 * wy;
                const int gx = 2 * tw.wtx < bx ? 2 * tw.wtx : bx, gy = 2 * tw.wty < by ? 2 * tw.wty : by;
                hipLaunchKernelGGL((scatter_csr_kernel<TMPX_C, 16, 8>), dim3(gx, gy, a.B), dim3(256), 8 * 8 * TMPX_C * 4,
                                   s, a, dX, FEAT_C + 3, a.TH, a.TW, dtmpx, accumulate, (const int*)sort, tw);
            }
    }
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

bool query_sort_covers(const QueryArgs& a) {
    return ((a.FW + 7) / 8) * ((a.FH + 7) / 8) <= 256 && a.N >= 64;
}

int launch_query_sort(chore_handle* h, const QueryArgs& a, int* work, hipStream_t s) {
    const int CH = scatter_chunk(a.N), G = (a.N + CH - 1) / CH;
    const int nb = ((a.FW + 7) / 8) * ((a.FH + 7) / 8);
    const size_t smem = ((size_t)G * SCATTER_OFF_STRIDE + nb + 4 + (size_t)(nb + 1) * G) * sizeof(int);
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)query_perm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)((64 * SCATTER_OFF_STRIDE + 264 + 257 * 64) * sizeof(int))));
        attr = true;
    }
    hipLaunchKernelGGL((query_bin_kernel<8>), dim3(G, a.B), dim3(64), 0, s, a, a.FH, a.FW, work);
    hipLaunchKernelGGL(query_perm_kernel, dim3(8, a.B), dim3(1024), smem, s, a, nb, work);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

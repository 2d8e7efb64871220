// query_common.h -- projection, tap set-up and the pixel-aligned gather shared by the fused query
// kernels (forward and backward-to-points).
//
// Arithmetic follows, operation by operation and WITHOUT fused multiply-add:
//   model/camera.py:64-65,75-78   (pinhole projection, crop shift, normalisation to [-1,1])
//   model/chore.py:128-130        (z_feat = [x, y, z-2.2], in_img mask)
//   ATen grid_sampler_2d (bilinear, zeros padding, align_corners=True) as called from
//   model/geometry.py:12: ix = (nx+1)*((W-1)/2); w = ix-floor(ix); e = 1-w; taps outside the
//   map contribute 0; value = fma(se,w*n, fma(sw,e*n, fma(ne,w*s, nw*(e*s)))).
// The __f*_rn intrinsics pin IEEE single operations so hipcc cannot contract them.
#pragma once
#include "common.h"

constexpr int QT_PTS = 64;  // points per workgroup tile

struct Cam {
    float fx, fy, cx, cy, half_crop, crop;
};

__device__ __forceinline__ void project_point(float x, float y, float z, float ccx, float ccy,
                                              const Cam& c, float& nx, float& ny) {
    float px = __fadd_rn(__fdiv_rn(__fmul_rn(c.fx, x), z), c.cx);
    float py = __fadd_rn(__fdiv_rn(__fmul_rn(c.fy, y), z), c.cy);
    px = __fsub_rn(__fadd_rn(c.half_crop, px), ccx);
    py = __fsub_rn(__fadd_rn(c.half_crop, py), ccy);
    nx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, px), c.crop), 1.0f);
    ny = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, py), c.crop), 1.0f);
}

// Bilinear tap description of one point in one map: element offsets of the 4 taps (nw, ne, sw,
// se; -1 = outside the map) and their weights.  Offsets are in elements of a (H,W,C) NHWC map.
struct Taps {
    int off[4];
    float w[4];
    float wx, wy;  // fractional parts (w, n) -- needed by the backward pass
};

__device__ __forceinline__ Taps make_taps(float nx, float ny, int H, int W, int C) {
    Taps t;
    const float ix = __fmul_rn(__fadd_rn(nx, 1.0f), (float)(W - 1) / 2);
    const float iy = __fmul_rn(__fadd_rn(ny, 1.0f), (float)(H - 1) / 2);
    const bool sane = (ix > -2.0f) && (ix < (float)W + 1.0f) && (iy > -2.0f) && (iy < (float)H + 1.0f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float w = __fsub_rn(ix, x0f), e = __fsub_rn(1.0f, w);
    const float n = __fsub_rn(iy, y0f), s = __fsub_rn(1.0f, n);
    t.wx = w;
    t.wy = n;
    t.w[0] = __fmul_rn(e, s);
    t.w[1] = __fmul_rn(w, s);
    t.w[2] = __fmul_rn(e, n);
    t.w[3] = __fmul_rn(w, n);
    const int x0 = sane ? (int)x0f : -4, y0 = sane ? (int)y0f : -4;
    const bool xv0 = (x0 >= 0) && (x0 < W), xv1 = (x0 + 1 >= 0) && (x0 + 1 < W);
    const bool yv0 = (y0 >= 0) && (y0 < H), yv1 = (y0 + 1 >= 0) && (y0 + 1 < H);
    const int base = (y0 * W + x0) * C;
    t.off[0] = (xv0 && yv0) ? base : -1;
    t.off[1] = (xv1 && yv0) ? base + C : -1;
    t.off[2] = (xv0 && yv1) ? base + W * C : -1;
    t.off[3] = (xv1 && yv1) ? base + W * C + C : -1;
    return t;
}

// ---- typed loads of NHWC feature rows ----
template <typename T> struct MapLoad;
// raw4 / raw1 + cvt4 / cvt1: the loads of a gather are issued under (wave-uniform) validity branches; converting inside
// the branch makes every load wait for its data there, one round trip after the other -- keep the raw bits and convert
// when the values are combined
template <> struct MapLoad<float> {
    typedef f32x4 Raw4;
    typedef float Raw1;
    static __device__ __forceinline__ f32x4 load4(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ Raw4 raw4(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ Raw1 raw1(const float* p) { return *p; }
    static __device__ __forceinline__ Raw4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ Raw1 zero1() { return 0.f; }
    static __device__ __forceinline__ f32x4 cvt4(Raw4 v) { return v; }
    static __device__ __forceinline__ float cvt1(Raw1 v, int = 0) { return v; }
};
template <> struct MapLoad<unsigned short> {  // bf16 storage
    static __device__ __forceinline__ f32x4 load4(const unsigned short* p) {
        const u16x4 v = *(const u16x4*)p;
        f32x4 r;
        r[0] = bf2f(v[0]); r[1] = bf2f(v[1]); r[2] = bf2f(v[2]); r[3] = bf2f(v[3]);
        return r;
    }
    static __device__ __forceinline__ float load1(const unsigned short* p) { return bf2f(*p); }
    // 32-bit containers: with 16-bit vector types the select against the zero value repacks the halves right after each
    // load, i.e. waits for it (33 x s_waitcnt vmcnt(0) in the gather: every tap a serial round trip, 3 x the fp32 gather)
    typedef unsigned Raw4 __attribute__((ext_vector_type(2)));
    typedef unsigned Raw1;
    static __device__ __forceinline__ Raw4 raw4(const unsigned short* p) { return *(const Raw4*)p; }
    // one channel per lane: lanes 2i and 2i+1 read the same aligned dword (rows of 64 channels start 4-byte aligned) and
    // pick their half when the taps are combined (cvt1's `odd` = the channel's parity) -- a plain dword load
    static __device__ __forceinline__ Raw1 raw1(const unsigned short* p) { return *(const unsigned*)((uintptr_t)p & ~(uintptr_t)3); }
    static __device__ __forceinline__ Raw4 zero4() { return Raw4{0u, 0u}; }
    static __device__ __forceinline__ Raw1 zero1() { return 0u; }
    static __device__ __forceinline__ f32x4 cvt4(Raw4 v) {
        return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                     __uint_as_float(v[1] & 0xffff0000u)};
    }
    static __device__ __forceinline__ float cvt1(Raw1 v, int odd) { return __uint_as_float(odd ? (v & 0xffff0000u) : (v << 16)); }
};

// IEEE half storage (CHORE_F16, "fp16 fields"): same containers as bf16, another conversion
struct qh16_t { unsigned short u; };
template <> struct MapLoad<qh16_t> {
    static __device__ __forceinline__ float h2f(unsigned bits16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); }
    static __device__ __forceinline__ f32x4 load4(const qh16_t* p) {
        const u16x4 v = *(const u16x4*)p;
        return f32x4{h2f(v[0]), h2f(v[1]), h2f(v[2]), h2f(v[3])};
    }
    static __device__ __forceinline__ float load1(const qh16_t* p) { return h2f(p->u); }
    typedef unsigned Raw4 __attribute__((ext_vector_type(2)));
    typedef unsigned Raw1;
    static __device__ __forceinline__ Raw4 raw4(const qh16_t* p) { return *(const Raw4*)p; }
    static __device__ __forceinline__ Raw1 raw1(const qh16_t* p) { return *(const unsigned*)((uintptr_t)p & ~(uintptr_t)3); }
    static __device__ __forceinline__ Raw4 zero4() { return Raw4{0u, 0u}; }
    static __device__ __forceinline__ Raw1 zero1() { return 0u; }
    static __device__ __forceinline__ f32x4 cvt4(Raw4 v) {
        return f32x4{h2f(v[0] & 0xffffu), h2f(v[0] >> 16), h2f(v[1] & 0xffffu), h2f(v[1] >> 16)};
    }
    static __device__ __forceinline__ float cvt1(Raw1 v, int odd) { return h2f(odd ? (v >> 16) : (v & 0xffffu)); }
};

__device__ __forceinline__ float interp4(float a, float b, float c, float d, const float* w) {
    // fma(se,w3, fma(sw,w2, fma(ne,w1, nw*w0))): the chain ATen's CPU grid_sampler executes
    // (pinned bit for bit by tests/golden/query_index.npz)
    return fmaf(d, w[3], fmaf(c, w[2], fmaf(b, w[1], __fmul_rn(a, w[0]))));
}

// Per-tile point table kept in LDS (structure of arrays, PTS entries each).
template <int PTS>
struct PtTableT {
    float xyz[3][PTS];   // x, y, z-2.2
    int foff[4][PTS];
    float fw[4][PTS];
    int toff[4][PTS];
    float tw[4][PTS];
    int in_img[PTS];
    int valid[PTS];
    int pidx[PTS];       // index of the point inside its image (= tile slot unless the query runs in sorted order, QueryArgs::perm)
    // backward only: fractional tap coordinates (w, n) of both maps and the raw depth
    float ffrac[2][PTS];
    float tfrac[2][PTS];
    float zraw[PTS];
};
using PtTable = PtTableT<QT_PTS>;

// thread `t` (< PTS) fills entry t of the table
// slot n of the tile order holds point `perm[n]` of the image when a permutation is given (queries in sorted order)
template <typename Tab>
__device__ __forceinline__ void fill_pt_table(Tab& tab, int t, const float* points,
                                              const float* crop_center, int b, int n, int N,
                                              const Cam& cam, int FH, int FW, int TH, int TW,
                                              float* nxy_out /* optional [2] */, const int* perm = nullptr) {
    const bool valid = n < N;
    const int slot = valid ? n : (N - 1);
    const int nn = perm ? perm[(size_t)b * N + slot] : slot;
    tab.pidx[t] = nn;
    const float* p = points + ((size_t)b * N + nn) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float nx, ny;
    project_point(x, y, z, crop_center[b * 2 + 0], crop_center[b * 2 + 1], cam, nx, ny);
    if (nxy_out) { nxy_out[0] = nx; nxy_out[1] = ny; }
    tab.xyz[0][t] = x;
    tab.xyz[1][t] = y;
    tab.xyz[2][t] = __fsub_rn(z, 2.2f);
    tab.in_img[t] = (nx >= -1.0f) && (nx <= 1.0f) && (ny >= -1.0f) && (ny <= 1.0f);
    tab.valid[t] = valid;
    tab.zraw[t] = z;
    const Taps tf = make_taps(nx, ny, FH, FW, FEAT_C);
    const Taps tt = make_taps(nx, ny, TH, TW, TMPX_C);
    tab.ffrac[0][t] = tf.wx;
    tab.ffrac[1][t] = tf.wy;
    tab.tfrac[0][t] = tt.wx;
    tab.tfrac[1][t] = tt.wy;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        tab.foff[k][t] = tf.off[k];
        tab.fw[k][t] = tf.w[k];
        tab.toff[k][t] = tt.off[k];
        tab.tw[k][t] = tt.w[k];
    }
}

// Workgroup -> (image, point tile).  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with
// its own L2; in launch order every image's tiles would land on all 8 of them and every L2 would pull its own copy of
// that image's feature maps (measured: 3.1x the compulsory bytes).  Here XCD x takes the x-th CONTIGUOUS eighth of the
// (image-major) tile order instead, so an image is read through ceil(8 / B) L2s only (bijective for any grid size).
__device__ __forceinline__ void query_block(int& b, int& tile) {
    const int tiles = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int L = blockIdx.x + blockIdx.y * gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    b = lid / tiles;
    tile = lid - b * tiles;
}

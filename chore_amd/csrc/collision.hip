// collision.hip -- interpenetration term of the joint fit, gfx950.        *** parity unpinned ***
//
// Replaces ReconFitterBase.smpl_obj_collision (/root/reference/recon/recon_fit_base.py:610-624): triangles of the
// concatenated SMPL + object mesh -> mesh_intersection.BVH(max_collisions=8) -> DistanceFieldPenetrationLoss(sigma=0.5,
// point2plane=False).  That package (github.com/vchoutas/torch-mesh-isect, unpinned, CUDA-only) is not in the reference
// tree; oracle/collision.py states the published method this file implements and why nothing can pin it.
//
// Design for one GPU-resident frame batch (F ~ 16 000 triangles, no host round trip, hipGraph-capturable: every launch
// has a fixed grid and reads the list lengths from device memory):
//   coll_setup_kernel     bounding boxes of all triangles (B x F x 2 float4) and zeroed accumulators / counters
//   coll_tilebox_kernel   box of every tile of 128 consecutive triangles (compact surface patches)
//   coll_tilepairs_kernel the tile pairs (ti <= tj) whose boxes overlap -> list (a few per cent of the 11 k pairs)
//   coll_broad_kernel     box tests of the listed 128 x 128 tile pairs (j boxes in LDS; of 180 M triangle pairs per
//                         frame at F = 19 k a few per cent are tested: a BVH would not pay at this size), pairs that
//                         share a vertex index dropped at once (13 of ~25 box neighbours of a triangle); a lane collects
//                         its hits as a bit mask and a wave reserves list space with ONE atomic (a returning atomic
//                         per hit is a dependent ~1.5 us chain: 100 us per dense tile); survivors -> candidate list
//   coll_narrow_kernel    one thread per candidate: shared-position test, 17-axis separating-axis test -> pair list
//   coll_loss_kernel      one thread per pair: the two conic distance fields, value AND gradient w.r.t. the 18
//                         coordinates by forward-mode dual numbers (exactly the derivative of the evaluated expression);
//                         accumulated with 64-bit fixed-point atomics (integer adds commute: the result does not depend
//                         on the order the lists were filled in, run-to-run bit-identical)
//   coll_finish_kernel    fixed point -> float: per-batch loss sums, gradient w.r.t. the vertices (kept for backward)
//   coll_scale_kernel     backward: d verts = upstream[b] * saved gradient
#include "common.h"

namespace {

constexpr int CT = 128;                    // triangles per tile side of the broad phase
constexpr float COLL_SIGMA = 0.5f;         // recon_fit_base.py:80
constexpr double FX_LOSS = 4294967296.0;   // 2^32
constexpr double FX_GRAD = 16777216.0;     // 2^24

struct CollWs {
    f32x4* box;                   // [B][F][2] lo, hi
    f32x4* tbox;                  // [B][tiles][2] boxes of the 128-triangle tiles
    int2* tpair; unsigned* ntpair;  // [B][tiles*(tiles+1)/2] tile pairs whose boxes overlap, [B]
    int2* cand; unsigned* ncand;  // [B][cand_cap], [B]
    int2* pair; unsigned* npair;  // [B][pair_cap], [B]
    unsigned* overflow;           // [2] candidates / pairs dropped
    long long* loss_fx;           // [B]
    long long* grad_fx;           // [B][V][3]
    int cand_cap, pair_cap;
};

__host__ __device__ inline size_t al256(size_t n) { return (n + 255) / 256 * 256; }
inline int coll_cand_cap(int F) { return F * 24 + 4096; }
inline int coll_pair_cap(int F) { return F * 8 + 1024; }      // the reference's list length: F * max_collisions

size_t coll_ws_layout(int B, int V, int F, char* base, CollWs* w) {
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al256(bytes); return p; };
    const int cc = coll_cand_cap(F), pc = coll_pair_cap(F);
    char* box = take((size_t)B * F * 2 * sizeof(f32x4));
    char* tbox = take((size_t)B * ((F + CT - 1) / CT) * 2 * sizeof(f32x4));
    const size_t nt_ = (F + CT - 1) / CT, ntp = nt_ * (nt_ + 1) / 2;
    char* tpair = take((size_t)B * ntp * sizeof(int2));
    char* cand = take((size_t)B * cc * sizeof(int2));
    char* pair = take((size_t)B * pc * sizeof(int2));
    char* cnt = take((size_t)(3 * B + 2) * sizeof(unsigned));
    char* lfx = take((size_t)B * sizeof(long long));
    char* gfx = take((size_t)B * V * 3 * sizeof(long long));
    if (w) {
        w->box = (f32x4*)box; w->tbox = (f32x4*)tbox; w->cand = (int2*)cand; w->pair = (int2*)pair;
        w->tpair = (int2*)tpair;
        w->ncand = (unsigned*)cnt; w->npair = w->ncand + B; w->overflow = w->npair + B; w->ntpair = w->overflow + 2;
        w->loss_fx = (long long*)lfx; w->grad_fx = (long long*)gfx;
        w->cand_cap = cc; w->pair_cap = pc;
    }
    return o;
}

__global__ void coll_setup_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int B, int V, int F, CollWs w) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)B * F) {
        const int b = (int)(i / F), f = (int)(i % F);
        const float* vb = verts + (size_t)b * V * 3;
        f32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* p = vb + (size_t)faces[f * 3 + k] * 3;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                lo[d] = k ? fminf(lo[d], p[d]) : p[d];
                hi[d] = k ? fmaxf(hi[d], p[d]) : p[d];
            }
        }
        lo[3] = 0.f; hi[3] = 0.f;
        w.box[i * 2] = lo; w.box[i * 2 + 1] = hi;
    }
    if (i < (size_t)B * V * 3) w.grad_fx[i] = 0;
    if (i < (size_t)B) w.loss_fx[i] = 0;
    if (i < (size_t)(3 * B + 2)) w.ncand[i] = 0u;
}

// box of every tile of CT consecutive triangles (consecutive faces of a mesh are neighbours: compact patches)
__global__ __launch_bounds__(CT) void coll_tilebox_kernel(int F, CollWs w) {
    __shared__ float red[2][CT / 64][3];
    const int b = blockIdx.y, t = blockIdx.x, nt = gridDim.x, j = t * CT + threadIdx.x;
    f32x4 lo = {3e38f, 3e38f, 3e38f, 0.f}, hi = {-3e38f, -3e38f, -3e38f, 0.f};
    if (j < F) { lo = w.box[((size_t)b * F + j) * 2]; hi = w.box[((size_t)b * F + j) * 2 + 1]; }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) { red[0][threadIdx.x >> 6][d] = lo[d]; red[1][threadIdx.x >> 6][d] = hi[d]; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
            for (int k = 1; k < CT / 64; ++k) { lo[d] = fminf(lo[d], red[0][k][d]); hi[d] = fmaxf(hi[d], red[1][k][d]); }
        w.tbox[((size_t)b * nt + t) * 2] = lo;
        w.tbox[((size_t)b * nt + t) * 2 + 1] = hi;
    }
}

// tile pairs (ti <= tj) whose boxes overlap -> list (one thread per pair of the upper triangle)
__global__ __launch_bounds__(256) void coll_tilepairs_kernel(int nt, CollWs w) {
    const int b = blockIdx.y, ntp = nt * (nt + 1) / 2, p = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    bool hit = false;
    int ti = 0, tj = 0;
    if (p < ntp) {
        int rem = p;
        while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
        tj = ti + rem;
        const f32x4 la = w.tbox[((size_t)b * nt + ti) * 2], ha = w.tbox[((size_t)b * nt + ti) * 2 + 1];
        const f32x4 lb = w.tbox[((size_t)b * nt + tj) * 2], hb = w.tbox[((size_t)b * nt + tj) * 2 + 1];
        hit = la[0] <= hb[0] && lb[0] <= ha[0] && la[1] <= hb[1] && lb[1] <= ha[1] && la[2] <= hb[2] && lb[2] <= ha[2];
    }
    const unsigned long long m = __ballot(hit);
    if (m == 0ull) return;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0u;
    if (lane == leader) base = atomicAdd(w.ntpair + b, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (hit) w.tpair[(size_t)b * ntp + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = make_int2(ti, tj);
}

// fixed grid; workgroup g takes the listed tile pairs g, g + gridDim.x, ...
__global__ __launch_bounds__(256) void coll_broad_kernel(const int* __restrict__ faces, int F, CollWs w) {
    __shared__ f32x4 jb[CT][2];
    __shared__ int jf[CT][3];
    const int b = blockIdx.y, nt = (F + CT - 1) / CT, ntp = nt * (nt + 1) / 2;
    const int tid = threadIdx.x;
    const unsigned nlist = w.ntpair[b];
    for (unsigned pi = blockIdx.x; pi < nlist; pi += gridDim.x) {
    const int2 tt = w.tpair[(size_t)b * ntp + pi];
    const int ti = tt.x, tj = tt.y;
    __syncthreads();
    if (tid < CT) {
        const int j = tj * CT + tid;
        const f32x4 far_lo = {3e38f, 3e38f, 3e38f, 0.f}, far_hi = {-3e38f, -3e38f, -3e38f, 0.f};
        jb[tid][0] = j < F ? w.box[((size_t)b * F + j) * 2] : far_lo;
        jb[tid][1] = j < F ? w.box[((size_t)b * F + j) * 2 + 1] : far_hi;
#pragma unroll
        for (int k = 0; k < 3; ++k) jf[tid][k] = j < F ? faces[j * 3 + k] : -1 - k;
    }
    __syncthreads();
    // pass 1: this lane's hits among its 64 j as a bit mask (no atomics inside the loop: a returning atomic per
    // iteration is a ~1.5 us dependent chain, 100 us per dense tile); then ONE atomic per wave reserves the space
    const int i = ti * CT + (tid & (CT - 1));
    const bool iok = i < F;
    const size_t ib = ((size_t)b * F + (iok ? i : 0)) * 2;
    const f32x4 lo = w.box[ib], hi = w.box[ib + 1];
    const int fi = iok ? i : 0;
    const int f0 = faces[fi * 3], f1 = faces[fi * 3 + 1], f2 = faces[fi * 3 + 2];
    const int j0 = (tid >> 7) * (CT / 2), lane = tid & 63;
    unsigned long long mask = 0ull;
    for (int q = 0; q < CT / 2; ++q) {
        const int jj = j0 + q, j = tj * CT + jj;
        const f32x4 l2 = jb[jj][0], h2 = jb[jj][1];
        bool hit = iok && j > i && lo[0] <= h2[0] && l2[0] <= hi[0] && lo[1] <= h2[1] && l2[1] <= hi[1] && lo[2] <= h2[2] &&
                   l2[2] <= hi[2];
        if (hit) {
            const int g0 = jf[jj][0], g1 = jf[jj][1], g2 = jf[jj][2];
            hit = !(f0 == g0 || f0 == g1 || f0 == g2 || f1 == g0 || f1 == g1 || f1 == g2 || f2 == g0 || f2 == g1 || f2 == g2);
        }
        if (hit) mask |= 1ull << q;
    }
    const unsigned cnt = (unsigned)__popcll(mask);
    unsigned incl = cnt;                                   // inclusive prefix sum over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned total = __shfl(incl, 63);
    if (total == 0u) continue;
    unsigned base = 0u;
    if (lane == 0) base = atomicAdd(w.ncand + b, total);
    base = __shfl(base, 0) + incl - cnt;
    while (mask) {
        const int q = __ffsll((long long)mask) - 1;
        mask &= mask - 1ull;
        if (base < (unsigned)w.cand_cap) w.cand[(size_t)b * w.cand_cap + base] = make_int2(i, tj * CT + j0 + q);
        else atomicAdd(w.overflow, 1u);
        ++base;
    }
    }
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ bool same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// true if `ax` separates the two triangles
__device__ __forceinline__ bool separates(V3 ax, const V3* t1, const V3* t2) {
    const float a0 = dot(t1[0], ax), a1 = dot(t1[1], ax), a2 = dot(t1[2], ax);
    const float b0 = dot(t2[0], ax), b1 = dot(t2[1], ax), b2 = dot(t2[2], ax);
    const float amin = fminf(a0, fminf(a1, a2)), amax = fmaxf(a0, fmaxf(a1, a2));
    const float bmin = fminf(b0, fminf(b1, b2)), bmax = fmaxf(b0, fmaxf(b1, b2));
    return amax < bmin || bmax < amin;
}

__device__ __forceinline__ void load_tri(const float* vb, const int* faces, int f, V3* t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* p = vb + (size_t)faces[f * 3 + k] * 3;
        t[k] = {p[0], p[1], p[2]};
    }
}

__global__ __launch_bounds__(256) void coll_narrow_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int V,
                                                          CollWs w) {
    const int b = blockIdx.y;
    const unsigned n = min(w.ncand[b], (unsigned)w.cand_cap);
    const float* vb = verts + (size_t)b * V * 3;
    for (unsigned c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const int2 ij = w.cand[(size_t)b * w.cand_cap + c];
        V3 t1[3], t2[3];
        load_tri(vb, faces, ij.x, t1);
        load_tri(vb, faces, ij.y, t2);
        bool shared = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) shared |= same(t1[p], t2[q]);
        if (shared) continue;
        const V3 e1[3] = {sub(t1[1], t1[0]), sub(t1[2], t1[1]), sub(t1[0], t1[2])};
        const V3 e2[3] = {sub(t2[1], t2[0]), sub(t2[2], t2[1]), sub(t2[0], t2[2])};
        const V3 n1 = cross(e1[0], e1[1]), n2 = cross(e2[0], e2[1]);
        bool sep = separates(n1, t1, t2) || separates(n2, t1, t2);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) sep = sep || separates(cross(e1[p], e2[q]), t1, t2);
#pragma unroll
        for (int p = 0; p < 3; ++p) sep = sep || separates(cross(n1, e1[p]), t1, t2) || separates(cross(n2, e2[p]), t1, t2);
        if (sep) continue;
        const unsigned k = atomicAdd(w.npair + b, 1u);
        if (k < (unsigned)w.pair_cap) w.pair[(size_t)b * w.pair_cap + k] = ij;
        else atomicAdd(w.overflow + 1, 1u);
    }
}

// ---- forward-mode dual numbers over NP input coordinates ----
template <int NP>
struct Dual {
    float v;
    float d[NP];
};
template <int NP> __device__ __forceinline__ Dual<NP> dconst(float v) {
    Dual<NP> r; r.v = v;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = 0.f;
    return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator+(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator-(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator*(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator*(float s, const Dual<NP>& a) {
    Dual<NP> r; r.v = s * a.v;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = s * a.d[i];
    return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator/(const Dual<NP>& a, const Dual<NP>& b) {
    Dual<NP> r;
    const float inv = 1.f / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> dsqrt(const Dual<NP>& a) {
    Dual<NP> r; r.v = sqrtf(a.v);
    const float h = r.v > 0.f ? 0.5f / r.v : 0.f;          // |.| at 0: subgradient 0, like torch.norm
#pragma unroll
    for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] * h;
    return r;
}
template <int NP> struct DV3 { Dual<NP> x, y, z; };
template <int NP> __device__ __forceinline__ DV3<NP> operator-(const DV3<NP>& a, const DV3<NP>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <int NP> __device__ __forceinline__ DV3<NP> operator+(const DV3<NP>& a, const DV3<NP>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <int NP> __device__ __forceinline__ DV3<NP> operator*(const Dual<NP>& s, const DV3<NP>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <int NP> __device__ __forceinline__ Dual<NP> ddot(const DV3<NP>& a, const DV3<NP>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <int NP> __device__ __forceinline__ DV3<NP> dcross(const DV3<NP>& a, const DV3<NP>& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// sum over the three points of Psi_cone(point)^2 (oracle/collision.py cone_field); the NP active coordinates are
// coordinates [c0, c0 + NP) of the 18-vector (cone triangle 0..8, points 9..17)
template <int NP>
__device__ Dual<NP> cone_field(const V3* cone, const V3* pts, int c0) {
    auto seed = [&](float v, int idx) {
        Dual<NP> r = dconst<NP>(v);
        const int k = idx - c0;
#pragma unroll
        for (int i = 0; i < NP; ++i) r.d[i] = (i == k) ? 1.f : 0.f;
        return r;
    };
    DV3<NP> t[3], p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        t[k] = {seed(cone[k].x, 3 * k), seed(cone[k].y, 3 * k + 1), seed(cone[k].z, 3 * k + 2)};
        p[k] = {seed(pts[k].x, 9 + 3 * k), seed(pts[k].y, 9 + 3 * k + 1), seed(pts[k].z, 9 + 3 * k + 2)};
    }
    const DV3<NP> a = t[1] - t[0], b = t[2] - t[0], c = dcross(a, b);
    const Dual<NP> aa = ddot(a, a), bb = ddot(b, b), cc = ddot(c, c);
    const DV3<NP> ab = a - b;
    const Dual<NP> r = dsqrt(aa * bb * ddot(ab, ab) / (4.f * cc));
    const Dual<NP> inv2cc = dconst<NP>(1.f) / (2.f * cc);
    const DV3<NP> o = t[0] + inv2cc * dcross(aa * b - bb * a, c);
    const Dual<NP> invn = dconst<NP>(1.f) / dsqrt(cc);
    const DV3<NP> n = invn * c;
    const float s = COLL_SIGMA;
    Dual<NP> total = dconst<NP>(0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const DV3<NP> d = p[k] - o;
        const Dual<NP> x = ddot(d, n);
        if (x.v >= s) continue;                                   // Ups = 0
        const DV3<NP> q = d - x * n;
        const Dual<NP> rad = dsqrt(ddot(q, q));
        const Dual<NP> den = r - (1.f / s) * (r * x);
        const Dual<NP> phi = rad / den;
        if (!(phi.v < 1.f)) continue;
        Dual<NP> ups;
        if (x.v <= -s) ups = dconst<NP>(1.f - s) - x;
        else ups = (-(1.f - 2.f * s) / (4.f * s * s)) * (x * x) - (1.f / (2.f * s)) * x + dconst<NP>((3.f - 2.f * s) / 4.f);
        const Dual<NP> psi = (dconst<NP>(1.f) - phi) * ups;
        total = total + psi * psi;
    }
    return total;
}

__device__ __forceinline__ void fx_add(long long* p, double v, double scale) {
    atomicAdd((unsigned long long*)p, (unsigned long long)__double2ll_rn(v * scale));
}

__global__ __launch_bounds__(64) void coll_loss_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int V,
                                                       CollWs w) {
    const int b = blockIdx.y;
    const unsigned n = min(w.npair[b], (unsigned)w.pair_cap);
    const float* vb = verts + (size_t)b * V * 3;
    long long* gb = w.grad_fx + (size_t)b * V * 3;
    for (unsigned c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const int2 ij = w.pair[(size_t)b * w.pair_cap + c];
        V3 t1[3], t2[3];
        load_tri(vb, faces, ij.x, t1);
        load_tri(vb, faces, ij.y, t2);
        float g1[9], g2[9];          // gradient w.r.t. the coordinates of triangle i / triangle j
#pragma unroll
        for (int k = 0; k < 9; ++k) { g1[k] = 0.f; g2[k] = 0.f; }
        float loss = 0.f;
        {   // cone of triangle i, points of triangle j
            const Dual<9> a = cone_field<9>(t1, t2, 0), p = cone_field<9>(t1, t2, 9);
            loss += a.v;
#pragma unroll
            for (int k = 0; k < 9; ++k) { g1[k] += a.d[k]; g2[k] += p.d[k]; }
        }
        {   // cone of triangle j, points of triangle i
            const Dual<9> a = cone_field<9>(t2, t1, 0), p = cone_field<9>(t2, t1, 9);
            loss += a.v;
#pragma unroll
            for (int k = 0; k < 9; ++k) { g2[k] += a.d[k]; g1[k] += p.d[k]; }
        }
        fx_add(w.loss_fx + b, (double)loss, FX_LOSS);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v1 = faces[ij.x * 3 + k], v2 = faces[ij.y * 3 + k];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (g1[3 * k + d] != 0.f) fx_add(gb + (size_t)v1 * 3 + d, (double)g1[3 * k + d], FX_GRAD);
                if (g2[3 * k + d] != 0.f) fx_add(gb + (size_t)v2 * 3 + d, (double)g2[3 * k + d], FX_GRAD);
            }
        }
    }
}

__global__ void coll_finish_kernel(int B, int V, CollWs w, float* __restrict__ loss, float* __restrict__ gverts,
                                   int* __restrict__ counts) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)B * V * 3) gverts[i] = (float)((double)w.grad_fx[i] / FX_GRAD);
    if (i < (size_t)B) {
        loss[i] = (float)((double)w.loss_fx[i] / FX_LOSS);
        if (counts) {
            counts[i] = (int)min(w.npair[i], (unsigned)w.pair_cap);
            if (i == 0) { counts[B] = (int)w.overflow[0]; counts[B + 1] = (int)w.overflow[1]; }
        }
    }
}

__global__ void coll_scale_kernel(const float* __restrict__ gverts, const float* __restrict__ gout, int V3n, size_t total,
                                  float* __restrict__ dverts) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dverts[i] = gout[i / V3n] * gverts[i];
}

}  // namespace

extern "C" {

size_t chore_collision_workspace_bytes(int B, int V, int F) {
    if (B <= 0 || V <= 0 || F <= 0) return 0;
    return coll_ws_layout(B, V, F, nullptr, nullptr);
}

int chore_collision_fwd(chore_handle* h, const float* verts, const int* faces, int B, int V, int F, float* loss, float* gverts,
                        int* counts, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!verts || !faces || !loss || !gverts || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_collision_fwd: null argument");
    if (B <= 0 || V <= 0 || F <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_collision_fwd: B, V, F must be positive");
    hipStream_t s = (hipStream_t)stream;
    CollWs w;
    coll_ws_layout(B, V, F, (char*)workspace, &w);
    size_t n = (size_t)B * (F > V * 3 ? F : V * 3);
    if (n < (size_t)(3 * B + 2)) n = 3 * B + 2;
    hipLaunchKernelGGL(coll_setup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, verts, faces, B, V, F, w);
    CHORE_LAUNCH_CHECK(h, s);
    const int nt = (F + CT - 1) / CT;
    hipLaunchKernelGGL(coll_tilebox_kernel, dim3(nt, B), dim3(CT), 0, s, F, w);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(coll_tilepairs_kernel, dim3((nt * (nt + 1) / 2 + 255) / 256, B), dim3(256), 0, s, nt, w);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(coll_broad_kernel, dim3(1024, B), dim3(256), 0, s, faces, F, w);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(coll_narrow_kernel, dim3(256, B), dim3(256), 0, s, verts, faces, V, w);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(coll_loss_kernel, dim3(256, B), dim3(64), 0, s, verts, faces, V, w);
    CHORE_LAUNCH_CHECK(h, s);
    const size_t m = (size_t)B * V * 3;
    hipLaunchKernelGGL(coll_finish_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, B, V, w, loss, gverts, counts);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_collision_bwd(chore_handle* h, const float* gverts, const float* gout, int B, int V, float* dverts,
                        chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!gverts || !gout || !dverts || B <= 0 || V <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_collision_bwd: bad argument");
    const size_t m = (size_t)B * V * 3;
    hipLaunchKernelGGL(coll_scale_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gverts, gout, V * 3, m,
                       dverts);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

}  // extern "C"

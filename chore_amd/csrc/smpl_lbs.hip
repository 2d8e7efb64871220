// smpl_lbs.hip -- SMPL / SMPL-H linear blend skinning, forward and backward, for gfx950.
//
// Replaces SMPL_Layer.forward (lib_smpl/smplpytorch/smplpytorch/pytorch/smpl_layer.py:72-175) and
// its autograd: ~500 tiny ATen launches per call in the reference (52 batch_rodrigues, 51 chained
// 4x4 matmuls, 52 bmm, a (V*3 x 459) blend-shape matmul) become 2 kernels forward, 3 backward.
//   * pose kernel (one workgroup per frame): 52 Rodrigues rotations through the reference's quaternion
//     path (angle = ||theta + 1e-8||, rodrigues_layer.py:41-52), joint locations J = JT + JS*beta
//     (J_regressor folded into JT/JS at pack time), the kinematic chain and the skinning transforms.
//   * vertex kernel: thread = one coordinate of a vertex; the pose blend shapes are stored p-major
//     ([459][V*3]) so the 38 MB matrix streams once, fully coalesced and 27 rows at a time, for a
//     group of up to 4 frames whose pose maps and transforms sit in LDS.
//   * backward: vertex kernel (recomputes the blend, emits d v_posed and d T per vertex), one
//     launch of fixed-order reductions (dA = W^T dT; d pose_map = P^T d v_posed with d beta = S^T d v_posed)
//     and the pose kernel backward (reverse kinematic chain, Rodrigues Jacobian by forward-mode
//     dual numbers).  All reductions are tree reductions in a fixed order: deterministic.
// Gradients are produced for pose, betas and trans (what the fitting optimises,
// recon/recon_fit_behave.py:224-291); offsets / v_posed / naked upstream gradients are not consumed.
#include "common.h"
#include <algorithm>

namespace {

constexpr int FB = 4;          // frames processed together by the vertex kernels
constexpr int MAXJ = 64;       // joints (52 for SMPL-H, 24 for SMPL)
constexpr int MAXNB = 16;

struct Dims {
    int V, J, NB, NP;          // vertices, joints, betas, pose-blend directions = 9 (J-1)
};
struct Arena {                 // float offsets into the packed model
    size_t T, S, PT, WT, JT, JS, parents, ST, total;
};
__host__ __device__ inline Arena arena_layout(const Dims& d) {
    Arena a;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
    a.T = take((size_t)d.V * 3);
    a.S = take((size_t)d.V * 3 * d.NB);
    a.PT = take((size_t)d.NP * d.V * 3);
    a.WT = take((size_t)d.J * d.V);
    a.JT = take((size_t)d.J * 3);
    a.JS = take((size_t)d.J * 3 * d.NB);
    a.parents = take(d.J);
    a.ST = take((size_t)d.NB * d.V * 3);      // shapedirs transposed: [n][v*3+k]
    a.total = o;
    return a;
}
struct Work {                  // float offsets into the per-call workspace
    size_t R, Jl, G, A, pm, gvp, dT, dA, dpm, dbv, tpart, total;
};
__host__ __device__ inline Work work_layout(const Dims& d, int B) {
    Work w;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
    w.R = take((size_t)B * d.J * 9);
    w.Jl = take((size_t)B * d.J * 3);
    w.G = take((size_t)B * d.J * 12);
    w.A = take((size_t)B * d.J * 12);
    w.pm = take((size_t)B * d.NP);
    w.gvp = take((size_t)B * d.V * 3);
    w.dT = take((size_t)B * d.V * 12);
    w.dA = take((size_t)B * d.J * 12);
    w.dpm = take((size_t)B * d.NP);
    w.dbv = take((size_t)B * d.NB);
    w.tpart = take((size_t)B * ((d.V + 255) / 256) * 3 * 2);     // doubles: per-workgroup partial sums of d trans
    w.total = o;
    return w;
}

// ---- scalar type for forward-mode differentiation of the Rodrigues map (3 inputs) ----
struct D3 {
    float v, d[3];
};
__device__ __forceinline__ D3 mk(float v) { return D3{v, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) {
    return D3{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ D3 operator/(D3 a, D3 b) {
    const float iv = 1.0f / b.v, q = a.v * iv;
    return D3{q, {(a.d[0] - q * b.d[0]) * iv, (a.d[1] - q * b.d[1]) * iv, (a.d[2] - q * b.d[2]) * iv}};
}
__device__ __forceinline__ D3 dsqrt(D3 a) {
    const float s = sqrtf(a.v), k = 0.5f / s;
    return D3{s, {a.d[0] * k, a.d[1] * k, a.d[2] * k}};
}
__device__ __forceinline__ D3 dsin(D3 a) { const float c = cosf(a.v); return D3{sinf(a.v), {a.d[0] * c, a.d[1] * c, a.d[2] * c}}; }
__device__ __forceinline__ D3 dcos(D3 a) { const float s = -sinf(a.v); return D3{cosf(a.v), {a.d[0] * s, a.d[1] * s, a.d[2] * s}}; }
__device__ __forceinline__ float val(float x) { return x; }
__device__ __forceinline__ float val(D3 x) { return x.v; }
__device__ __forceinline__ float ksqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ D3 ksqrt(D3 x) { return dsqrt(x); }
__device__ __forceinline__ float ksin(float x) { return sinf(x); }
__device__ __forceinline__ D3 ksin(D3 x) { return dsin(x); }
__device__ __forceinline__ float kcos(float x) { return cosf(x); }
__device__ __forceinline__ D3 kcos(D3 x) { return dcos(x); }
__device__ __forceinline__ float lit(float, float c) { return c; }
__device__ __forceinline__ D3 lit(D3, float c) { return mk(c); }

// batch_rodrigues + quat2mat (rodrigues_layer.py:13-52), generic in the scalar type
template <typename S>
__device__ __forceinline__ void rodrigues(S tx, S ty, S tz, S (&R)[9]) {
    const S eps = lit(tx, 1e-8f), half = lit(tx, 0.5f), two = lit(tx, 2.0f);
    const S ax = tx + eps, ay = ty + eps, az = tz + eps;
    const S angle = ksqrt(ax * ax + ay * ay + az * az);
    const S nx = tx / angle, ny = ty / angle, nz = tz / angle;
    const S h = angle * half;
    const S c = kcos(h), s = ksin(h);
    S w = c, x = s * nx, y = s * ny, z = s * nz;
    const S qn = ksqrt(w * w + x * x + y * y + z * z);
    w = w / qn; x = x / qn; y = y / qn; z = z / qn;
    const S w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const S wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = two * xy - two * wz; R[2] = two * wy + two * xz;
    R[3] = two * wz + two * xy; R[4] = w2 - x2 + y2 - z2; R[5] = two * yz - two * wx;
    R[6] = two * xz - two * wy; R[7] = two * wx + two * yz; R[8] = w2 - x2 - y2 + z2;
}

// ------------------------------------------------------------------------------------------------
// pack: reference buffers -> arena
// ------------------------------------------------------------------------------------------------
__global__ void pack_copy_kernel(const float* src, float* dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void pack_posedirs_kernel(Dims d, const float* __restrict__ posedirs /*(V,3,NP)*/, float* __restrict__ PT) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // index into PT [p][v*3+k]
    const size_t V3 = (size_t)d.V * 3;
    if (i >= V3 * d.NP) return;
    const size_t p = i / V3, vk = i % V3;
    PT[i] = posedirs[vk * d.NP + p];
}
__global__ void pack_shapedirs_t_kernel(Dims d, const float* __restrict__ S /*(V*3,NB)*/, float* __restrict__ ST) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // index into ST [n][v*3+k]
    const size_t V3 = (size_t)d.V * 3;
    if (i >= V3 * d.NB) return;
    const size_t n = i / V3, vk = i % V3;
    ST[i] = S[vk * d.NB + n];
}
__global__ void pack_weights_kernel(Dims d, const float* __restrict__ weights /*(V,J)*/, float* __restrict__ WT) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // [j][v]
    if (i >= (size_t)d.J * d.V) return;
    const size_t j = i / d.V, v = i % d.V;
    WT[i] = weights[v * d.J + j];
}
// JT[j][k] = sum_v Jreg[j][v] T[v][k];  JS[j][k][n] = sum_v Jreg[j][v] S[v][k][n]
__global__ void pack_jreg_kernel(Dims d, const float* __restrict__ Jreg, const float* __restrict__ T,
                                 const float* __restrict__ S, float* __restrict__ JT, float* __restrict__ JS) {
    __shared__ double sh[256];
    const int j = blockIdx.x, col = blockIdx.y, tid = threadIdx.x;   // col: 0..2 -> JT, 3.. -> JS[k][n]
    double a = 0.0;
    for (int v = tid; v < d.V; v += 256) {
        const float w = Jreg[(size_t)j * d.V + v];
        if (w != 0.f) {
            const float x = (col < 3) ? T[(size_t)v * 3 + col] : S[(size_t)v * 3 * d.NB + (col - 3)];
            a += (double)w * (double)x;
        }
    }
    sh[tid] = a;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        if (col < 3) JT[j * 3 + col] = (float)sh[0];
        else JS[(size_t)j * 3 * d.NB + (col - 3)] = (float)sh[0];
    }
}

// ------------------------------------------------------------------------------------------------
// pose kernel forward
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* c) {   // c = a*b (row-major 3x3)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) c[r * 3 + q] = a[r * 3] * b[q] + a[r * 3 + 1] * b[3 + q] + a[r * 3 + 2] * b[6 + q];
}

__global__ __launch_bounds__(64) void lbs_pose_fwd_kernel(Dims d, const float* __restrict__ model, const float* __restrict__ pose,
                                                          const float* __restrict__ betas, const float* __restrict__ trans,
                                                          float scale, float* __restrict__ work, int B,
                                                          float* __restrict__ joints) {
    __shared__ float R[MAXJ][9], Jl[MAXJ][3], G[MAXJ][12];
    __shared__ int dep[64], par[64];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int b = blockIdx.x, j = threadIdx.x;
    const int* parents = (const int*)(model + ar.parents);
    if (j < d.J) {
        const float* th = pose + ((size_t)b * d.J + j) * 3;
        rodrigues<float>(th[0], th[1], th[2], R[j]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = model[ar.JT + j * 3 + k];
            float c[MAXNB], bt[MAXNB];       // requested together, added in index order
#pragma unroll
            for (int n = 0; n < MAXNB; ++n) {
                c[n] = model[ar.JS + ((size_t)j * 3 + k) * d.NB + min(n, d.NB - 1)];
                bt[n] = betas[(size_t)b * d.NB + min(n, d.NB - 1)];
            }
#pragma unroll
            for (int n = 0; n < MAXNB; ++n)
                if (n < d.NB) a += c[n] * bt[n];
            Jl[j][k] = a;
        }
    }
    __syncthreads();
    // kinematic chain (smpl_layer.py:114-131): G_i = G_parent * [R_i | J_i - J_parent].  One lane per joint, one round per
    // tree depth (a joint's transform needs its parent's only): 10 rounds for SMPL-H instead of 51 steps on one lane
    par[j] = j < d.J ? parents[j] : 0;      // the tree in LDS: the walks below are chains of dependent reads
    __syncthreads();
    int depth = 0;
    if (j < d.J)
        for (int q = j; q > 0; q = par[q]) ++depth;
    dep[j] = j < d.J ? depth : 0;
    if (j == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) G[0][(e / 3) * 4 + e % 3] = R[0][e];
        G[0][3] = Jl[0][0]; G[0][7] = Jl[0][1]; G[0][11] = Jl[0][2];
    }
    __syncthreads();
    int maxdep = 0;
    for (int q = 0; q < d.J; ++q) maxdep = max(maxdep, dep[q]);
    for (int level = 1; level <= maxdep; ++level) {
        if (j < d.J && depth == level) {
            const int i = j, p = par[i];
            float gr[9], rr[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) gr[e] = G[p][(e / 3) * 4 + e % 3];
            mat3_mul(gr, R[i], rr);
            const float t0 = Jl[i][0] - Jl[p][0], t1 = Jl[i][1] - Jl[p][1], t2 = Jl[i][2] - Jl[p][2];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                G[i][r * 4 + 0] = rr[r * 3]; G[i][r * 4 + 1] = rr[r * 3 + 1]; G[i][r * 4 + 2] = rr[r * 3 + 2];
                G[i][r * 4 + 3] = gr[r * 3] * t0 + gr[r * 3 + 1] * t1 + gr[r * 3 + 2] * t2 + G[p][r * 4 + 3];
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (j < d.J) {
        float* Ro = work + wk.R + ((size_t)b * d.J + j) * 9;
        float* Jo = work + wk.Jl + ((size_t)b * d.J + j) * 3;
        float* Go = work + wk.G + ((size_t)b * d.J + j) * 12;
        float* Ao = work + wk.A + ((size_t)b * d.J + j) * 12;
#pragma unroll
        for (int e = 0; e < 9; ++e) Ro[e] = R[j][e];
#pragma unroll
        for (int k = 0; k < 3; ++k) Jo[k] = Jl[j][k];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float g0 = G[j][r * 4], g1 = G[j][r * 4 + 1], g2 = G[j][r * 4 + 2], gt = G[j][r * 4 + 3];
            Go[r * 4] = g0; Go[r * 4 + 1] = g1; Go[r * 4 + 2] = g2; Go[r * 4 + 3] = gt;
            // A = G with the rest pose removed: t - R*J   (smpl_layer.py:135-142)
            Ao[r * 4] = g0; Ao[r * 4 + 1] = g1; Ao[r * 4 + 2] = g2;
            Ao[r * 4 + 3] = gt - (g0 * Jl[j][0] + g1 * Jl[j][1] + g2 * Jl[j][2]);
            joints[((size_t)b * d.J + j) * 3 + r] = gt * scale + trans[b * 3 + r];
        }
        if (j >= 1) {   // pose map = R - I, flattened (subtract_flat_id, tensutils.py:41-53)
            float* pm = work + wk.pm + (size_t)b * d.NP + (j - 1) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) pm[e] = R[j][e] - ((e % 4 == 0) ? 1.0f : 0.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// vertex kernel forward: thread = one coordinate of one vertex (VF_V vertices per workgroup), FB frames per workgroup.
// A thread owns the whole sum of its coordinate, in the order p = 0, 1, ... (the result does not depend on the launch shape);
// the loads of VF_U pose directions are issued together, ahead of their multiply-adds: the 38 MB matrix streams with
// V*3 x VF_U loads in flight (one vertex per thread and one load at a time left 27 workgroups waiting on a chain of 459
// round trips: 64 us per call).
// ------------------------------------------------------------------------------------------------
constexpr int VF_V = 64, VF_T = VF_V * 3, VF_U = 27;    // 51 rows at a time: 53 us against 29
constexpr int SKIN_U = 13;     // 52 = 4 x 13 joints (SMPL-H); SMPL's 24: 13 + 11
constexpr int DPM_U = 9;       // steps of a 256-thread reduction loop whose loads are issued together
__global__ __launch_bounds__(VF_T) void lbs_vertex_fwd_kernel(Dims d, const float* __restrict__ model,
                                                              const float* __restrict__ betas, const float* __restrict__ trans,
                                                              const float* __restrict__ offsets, float scale,
                                                              const float* __restrict__ work, int B, float* __restrict__ verts,
                                                              float* __restrict__ v_posed, float* __restrict__ naked) {
    extern __shared__ float sm[];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    float* pm = sm;                       // [FB][NP]
    float* A = sm + FB * d.NP;            // [FB][J][12]
    float* bt = A + FB * d.J * 12;        // [FB][NB]
    float* vpl = bt + FB * d.NB;          // [FB][VF_T]
    const int b0 = blockIdx.y * FB, nf = min(FB, B - b0);
    const int tid = threadIdx.x;
    const size_t V3 = (size_t)d.V * 3;
    const size_t i_raw = (size_t)blockIdx.x * VF_T + tid;
    const bool live = i_raw < V3;
    const size_t i = live ? i_raw : V3 - 1;          // idle threads of the last workgroup repeat the last coordinate
    const int v = (int)(i / 3), k = (int)(i - (size_t)v * 3);
    for (int q = tid; q < nf * d.NP; q += VF_T) pm[q] = work[wk.pm + (size_t)b0 * d.NP + q];
    for (int q = tid; q < nf * d.J * 12; q += VF_T) A[q] = work[wk.A + (size_t)b0 * d.J * 12 + q];
    for (int q = tid; q < nf * d.NB; q += VF_T) bt[q] = betas[(size_t)b0 * d.NB + q];
    __syncthreads();
    float acc[FB];
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        float a = model[ar.T + i];
        if (f < nf)
            for (int n = 0; n < d.NB; ++n) a += model[ar.S + i * d.NB + n] * bt[f * d.NB + n];
        acc[f] = a;
    }
    const float* PT = model + ar.PT + i;
    int p = 0;
    for (; p + VF_U <= d.NP; p += VF_U) {
        float x[VF_U];
#pragma unroll
        for (int u = 0; u < VF_U; ++u) x[u] = PT[(size_t)(p + u) * V3];
#pragma unroll
        for (int u = 0; u < VF_U; ++u)
#pragma unroll
            for (int f = 0; f < FB; ++f) acc[f] = fmaf(x[u], pm[f * d.NP + p + u], acc[f]);   // frames beyond nf read stale LDS: never stored
    }
    for (; p < d.NP; ++p) {
        const float x = PT[(size_t)p * V3];
#pragma unroll
        for (int f = 0; f < FB; ++f) acc[f] = fmaf(x, pm[f * d.NP + p], acc[f]);
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        if (f < nf) {
            const size_t o = (size_t)(b0 + f) * V3 + i;
            const float vp = acc[f] + (offsets ? offsets[o] : 0.f);
            if (live) { naked[o] = acc[f]; v_posed[o] = vp; }
            vpl[f * VF_T + tid] = vp;
        }
    }
    __syncthreads();
    // skinning: this thread's row k of T = sum_j w_j A_j, applied to the vertex' three posed coordinates
    const float* vrow = vpl + (tid - k);
    for (int f = 0; f < nf; ++f) {
        float T[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < d.J; j0 += SKIN_U) {      // the weights of SKIN_U joints requested together, used in joint order
            float w[SKIN_U];
#pragma unroll
            for (int u = 0; u < SKIN_U; ++u) w[u] = model[ar.WT + (size_t)min(j0 + u, d.J - 1) * d.V + v];
#pragma unroll
            for (int u = 0; u < SKIN_U; ++u)
                if (j0 + u < d.J && w[u] != 0.f) {
                    const float* Aj = A + ((size_t)f * d.J + j0 + u) * 12 + k * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) T[e] = fmaf(w[u], Aj[e], T[e]);
                }
        }
        const float* vp = vrow + f * VF_T;
        if (live)
            verts[(size_t)(b0 + f) * V3 + i] = (T[0] * vp[0] + T[1] * vp[1] + T[2] * vp[2] + T[3]) * scale + trans[(b0 + f) * 3 + k];
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// per vertex: d v_posed = s * T.R^T g ; dT = s * [g (x) v_posed | g]
__global__ __launch_bounds__(256) void lbs_vertex_bwd_kernel(Dims d, const float* __restrict__ model, float scale,
                                                             float* __restrict__ work, int B,
                                                             const float* __restrict__ v_posed, const float* __restrict__ g_verts) {
    extern __shared__ float sm[];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    float* A = sm;                        // [J][12]
    const int b = blockIdx.y, tid = threadIdx.x, v = blockIdx.x * 256 + tid;
    for (int i = tid; i < d.J * 12; i += 256) A[i] = work[wk.A + (size_t)b * d.J * 12 + i];
    __syncthreads();
    const bool live = v < d.V;
    double gs[3] = {0.0, 0.0, 0.0};     // this thread's share of d trans = sum_v g_verts (trans is added after the scale)
    if (live) {
        float T[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) T[e] = 0.f;
        for (int j0 = 0; j0 < d.J; j0 += SKIN_U) {
            float w[SKIN_U];
#pragma unroll
            for (int u = 0; u < SKIN_U; ++u) w[u] = model[ar.WT + (size_t)min(j0 + u, d.J - 1) * d.V + v];
#pragma unroll
            for (int u = 0; u < SKIN_U; ++u)
                if (j0 + u < d.J && w[u] != 0.f) {
                    const int j = j0 + u;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(w[u], A[j * 12 + r * 4 + c], T[r * 3 + c]);
                }
        }
        const size_t o = ((size_t)b * d.V + v) * 3;
        const float r0 = g_verts ? g_verts[o] : 0.f, r1 = g_verts ? g_verts[o + 1] : 0.f, r2 = g_verts ? g_verts[o + 2] : 0.f;
        gs[0] = (double)r0; gs[1] = (double)r1; gs[2] = (double)r2;
        const float g0 = r0 * scale, g1 = r1 * scale, g2 = r2 * scale;
        const float vp0 = v_posed[o], vp1 = v_posed[o + 1], vp2 = v_posed[o + 2];
        float* gvp = work + wk.gvp + o;
        gvp[0] = T[0] * g0 + T[3] * g1 + T[6] * g2;
        gvp[1] = T[1] * g0 + T[4] * g1 + T[7] * g2;
        gvp[2] = T[2] * g0 + T[5] * g1 + T[8] * g2;
        float* dT = work + wk.dT + ((size_t)b * d.V + v) * 12;
        const float g[3] = {g0, g1, g2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dT[r * 4] = g[r] * vp0; dT[r * 4 + 1] = g[r] * vp1; dT[r * 4 + 2] = g[r] * vp2; dT[r * 4 + 3] = g[r];
        }
    }
    // fixed-order fp64 tree over the workgroup; the pose kernel adds the per-workgroup partial sums in workgroup order
    __syncthreads();      // A (sm) is reused
    double* sh = (double*)sm;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sh[tid] = gs[k];
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (tid < o) sh[tid] += sh[tid + o];
            __syncthreads();
        }
        if (tid == 0) ((double*)(work + wk.tpart))[((size_t)b * gridDim.x + blockIdx.x) * 3 + k] = sh[0];
        __syncthreads();
    }
}

// generic fixed-order block reduction of NACC accumulators (fp32 in, fp64 tree)
template <int NACC>
__device__ __forceinline__ void block_reduce(float (&a)[NACC], double* sh /*[256]*/, int tid, float (&out)[NACC]) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        sh[tid] = (double)a[k];
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (tid < o) sh[tid] += sh[tid + o];
            __syncthreads();
        }
        out[k] = (float)sh[0];
        __syncthreads();
    }
}

// dA[b][j][e] = sum_v W[v][j] dT[b][v][e]          grid (J, B)
__device__ __forceinline__ void lbs_dA_body(Dims d, const float* __restrict__ model, float* __restrict__ work, int B, int j, int b,
                                            double* sh) {
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int tid = threadIdx.x;
    float a[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) a[e] = 0.f;
    for (int v0 = tid; v0 < d.V; v0 += 256 * DPM_U) {
        float w[DPM_U];
#pragma unroll
        for (int u = 0; u < DPM_U; ++u) w[u] = model[ar.WT + (size_t)j * d.V + min(v0 + 256 * u, d.V - 1)];
#pragma unroll
        for (int u = 0; u < DPM_U; ++u)
            if (v0 + 256 * u < d.V && w[u] != 0.f) {
                const float* dT = work + wk.dT + ((size_t)b * d.V + v0 + 256 * u) * 12;
#pragma unroll
                for (int e = 0; e < 12; ++e) a[e] = fmaf(w[u], dT[e], a[e]);
            }
    }
    float r[12];
    block_reduce<12>(a, sh, tid, r);
    if (tid == 0) {
        float* o = work + wk.dA + ((size_t)b * d.J + j) * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) o[e] = r[e];
    }
}

// dpm[b][p] = sum_i PT[p][i] gvp[b][i]  (p < NP)  and  dbv[b][n] = sum_i S[i][n] gvp[b][i]  (row NP + n, from the transposed
// copy ST)             grid (NP + NB, ceil(B/FB)).  A thread's sum runs over i = tid, tid + 256, ... in that order; the loads of
// DPM_U steps are issued together (one at a time: 81 dependent round trips, 33 + 25 us for the two former kernels)
__device__ __forceinline__ void lbs_dpm_body(Dims d, const float* __restrict__ model, float* __restrict__ work, int B, int p, int grp,
                                             double* sh) {
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int b0 = grp * FB, nf = min(FB, B - b0), tid = threadIdx.x;
    const size_t V3 = (size_t)d.V * 3;
    const float* row = p < d.NP ? model + ar.PT + (size_t)p * V3 : model + ar.ST + (size_t)(p - d.NP) * V3;
    const float* gv[FB];
#pragma unroll
    for (int f = 0; f < FB; ++f) gv[f] = work + wk.gvp + (size_t)(b0 + min(f, nf - 1)) * V3;   // frames beyond nf: a valid row, unused
    float a[FB];
#pragma unroll
    for (int f = 0; f < FB; ++f) a[f] = 0.f;
    for (size_t i0 = tid; i0 < V3; i0 += 256 * DPM_U) {
        float x[DPM_U], g[DPM_U][FB];
#pragma unroll
        for (int u = 0; u < DPM_U; ++u) {
            const size_t i = min(i0 + (size_t)256 * u, V3 - 1);
            x[u] = row[i];
#pragma unroll
            for (int f = 0; f < FB; ++f) g[u][f] = gv[f][i];
        }
#pragma unroll
        for (int u = 0; u < DPM_U; ++u)
            if (i0 + (size_t)256 * u < V3) {
#pragma unroll
                for (int f = 0; f < FB; ++f) a[f] = fmaf(x[u], g[u][f], a[f]);
            }
    }
    float r[FB];
    block_reduce<FB>(a, sh, tid, r);
    if (tid == 0)
        for (int f = 0; f < nf; ++f) {
            if (p < d.NP) work[wk.dpm + (size_t)(b0 + f) * d.NP + p] = r[f];
            else work[wk.dbv + (size_t)(b0 + f) * d.NB + (p - d.NP)] = r[f];
        }
}

// the two reductions over the vertices in one launch (they are independent): blocks [0, J) x B the skinning transforms,
// blocks [J, J + NP + NB) x ceil(B / FB) the pose map and the betas                       grid (J + NP + NB, B)
__global__ __launch_bounds__(256) void lbs_reduce_kernel(Dims d, const float* __restrict__ model, float* __restrict__ work, int B) {
    __shared__ double sh[256];
    const int x = blockIdx.x, y = blockIdx.y;
    if (x < d.J) lbs_dA_body(d, model, work, B, x, y, sh);
    else if (y < (B + FB - 1) / FB) lbs_dpm_body(d, model, work, B, x - d.J, y, sh);
}

// pose kernel backward: reverse chain, Rodrigues Jacobian, betas through J, trans
__global__ __launch_bounds__(64) void lbs_pose_bwd_kernel(Dims d, const float* __restrict__ model, const float* __restrict__ pose,
                                                          float scale, const float* __restrict__ work, int B,
                                                          const float* __restrict__ g_verts, const float* __restrict__ g_joints,
                                                          float* __restrict__ dpose, float* __restrict__ dbetas,
                                                          float* __restrict__ dtrans) {
    __shared__ float R[MAXJ][9], Jl[MAXJ][3], G[MAXJ][12], dG[MAXJ][12], dR[MAXJ][9], dJ[MAXJ][3];
    __shared__ int par[64];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int b = blockIdx.x, j = threadIdx.x;
    const int* parents = (const int*)(model + ar.parents);
    par[j] = j < d.J ? parents[j] : 0;      // read by every step of the chain below
    if (j < d.J) {
        const float* dA = work + wk.dA + ((size_t)b * d.J + j) * 12;
#pragma unroll
        for (int e = 0; e < 9; ++e) R[j][e] = work[wk.R + ((size_t)b * d.J + j) * 9 + e];
#pragma unroll
        for (int k = 0; k < 3; ++k) Jl[j][k] = work[wk.Jl + ((size_t)b * d.J + j) * 3 + k];
#pragma unroll
        for (int e = 0; e < 12; ++e) G[j][e] = work[wk.G + ((size_t)b * d.J + j) * 12 + e];
        // A = [G.R | G.t - G.R J]:  dG.R = dA.R - dA.t (x) J ; dG.t = dA.t (+ s * g_joint) ; dJ = -G.R^T dA.t
        float gj[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) gj[k] = g_joints ? g_joints[((size_t)b * d.J + j) * 3 + k] * scale : 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dG[j][r * 4 + c] = dA[r * 4 + c] - dA[r * 4 + 3] * Jl[j][c];
            dG[j][r * 4 + 3] = dA[r * 4 + 3] + gj[r];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dJ[j][c] = -(G[j][c] * dA[3] + G[j][4 + c] * dA[7] + G[j][8 + c] * dA[11]);
#pragma unroll
        for (int e = 0; e < 9; ++e) dR[j][e] = 0.f;
    }
    __syncthreads();
    // reverse chain, children before parents (parents[i] < i in SMPL trees), one step per joint; inside a step the 9 + 3 + 3
    // independent results go to 15 lanes (every result is the expression the one-lane loop evaluated)
    for (int i = d.J - 1; i >= 1; --i) {
        const int p = par[i];
        if (j < 9) {
            const int r = j / 3, c = j % 3;
            const float tc = Jl[i][c] - Jl[p][c];
            // dG_p.R += dG_i.R R_i^T + dG_i.t (x) t ;  dR_i += G_p.R^T dG_i.R
            dG[p][r * 4 + c] += dG[i][r * 4] * R[i][c * 3] + dG[i][r * 4 + 1] * R[i][c * 3 + 1] +
                                dG[i][r * 4 + 2] * R[i][c * 3 + 2] + dG[i][r * 4 + 3] * tc;
            dR[i][r * 3 + c] += G[p][r] * dG[i][c] + G[p][4 + r] * dG[i][4 + c] + G[p][8 + r] * dG[i][8 + c];
        } else if (j < 12) {
            const int r = j - 9;
            dG[p][r * 4 + 3] += dG[i][r * 4 + 3];
        } else if (j < 15) {
            const int c = j - 12;
            const float dt = G[p][c] * dG[i][3] + G[p][4 + c] * dG[i][7] + G[p][8 + c] * dG[i][11];
            dJ[i][c] += dt;
            dJ[p][c] -= dt;
        }
        __syncthreads();
    }
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dR[0][r * 3 + c] += dG[0][r * 4 + c];
            dJ[0][r] += dG[0][r * 4 + 3];
        }
    }
    __syncthreads();
    if (j < d.J) {
        float g[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) g[e] = dR[j][e] + ((j >= 1) ? work[wk.dpm + (size_t)b * d.NP + (j - 1) * 9 + e] : 0.f);
        const float* th = pose + ((size_t)b * d.J + j) * 3;
        D3 tx = mk(th[0]), ty = mk(th[1]), tz = mk(th[2]);
        tx.d[0] = 1.f; ty.d[1] = 1.f; tz.d[2] = 1.f;
        D3 Rd[9];
        rodrigues<D3>(tx, ty, tz, Rd);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 9; ++e) a += g[e] * Rd[e].d[k];
            dpose[((size_t)b * d.J + j) * 3 + k] = a;
        }
    }
    if (j < d.NB) {   // d beta = S^T d v_posed (vertex part) + JS^T dJ (joint part)
        float a = work[wk.dbv + (size_t)b * d.NB + j];
        const int n3 = d.J * 3;
        const float* dJf = &dJ[0][0];
        for (int i0 = 0; i0 < n3; i0 += 12) {       // twelve coefficients requested together, added in index order
            float c[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) c[u] = model[ar.JS + (size_t)min(i0 + u, n3 - 1) * d.NB + j];
#pragma unroll
            for (int u = 0; u < 12; ++u)
                if (i0 + u < n3) a += c[u] * dJf[i0 + u];
        }
        dbetas[(size_t)b * d.NB + j] = a;
    }
    if (j < 3) {   // d trans = sum_v g_verts (per-workgroup partial sums of the vertex kernel) + sum_j g_joints
        double a = 0.0;
        const int nblk = (d.V + 255) / 256;
        const double* part = (const double*)(work + wk.tpart) + (size_t)b * nblk * 3;
        for (int i0 = 0; i0 < nblk; i0 += 9) {          // nine partial sums requested together, added in order
            double c[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) c[u] = part[min(i0 + u, nblk - 1) * 3 + j];
#pragma unroll
            for (int u = 0; u < 9; ++u)
                if (i0 + u < nblk) a += c[u];
        }
        if (g_joints)
            for (int i0 = 0; i0 < d.J; i0 += 13) {
                float c[13];
#pragma unroll
                for (int u = 0; u < 13; ++u) c[u] = g_joints[((size_t)b * d.J + min(i0 + u, d.J - 1)) * 3 + j];
#pragma unroll
                for (int u = 0; u < 13; ++u)
                    if (i0 + u < d.J) a += (double)c[u];
            }
        dtrans[b * 3 + j] = (float)a;
    }
}

// ------------------------------------------------------------------------------------------------
// landmark regression: lm[b][r] = sum_v reg[r][v] verts[b][v]   (R = 25 + 70 + 42 rows for the body / face / hand
// regressors, V = 6890).  As one (R x V) x (V x 3) product per frame this is a shape the GEMM library serves badly
// (60 us per call, measured: three columns); here one workgroup per (row, frame), loads issued together, fp64 tree.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void landmarks_fwd_kernel(const float* __restrict__ reg, const float* __restrict__ verts,
                                                            int R, int V, float* __restrict__ out) {
    __shared__ double sh[256];
    const int r = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* row = reg + (size_t)r * V;
    const float* vb = verts + (size_t)b * V * 3;
    float a[3] = {0.f, 0.f, 0.f};
    for (int v0 = tid; v0 < V; v0 += 256 * DPM_U) {
        float w[DPM_U], x[DPM_U][3];
#pragma unroll
        for (int u = 0; u < DPM_U; ++u) {
            const int v = min(v0 + 256 * u, V - 1);
            w[u] = row[v];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[u][k] = vb[(size_t)v * 3 + k];
        }
#pragma unroll
        for (int u = 0; u < DPM_U; ++u)
            if (v0 + 256 * u < V) {
#pragma unroll
                for (int k = 0; k < 3; ++k) a[k] = fmaf(w[u], x[u][k], a[k]);
            }
    }
    float o[3];
    block_reduce<3>(a, sh, tid, o);
    if (tid < 3) out[((size_t)b * R + r) * 3 + tid] = o[tid];
}

// d verts[b][v] = sum_r reg[r][v] g[b][r]: 64 vertices per workgroup, the rows dealt to four groups of threads in
// contiguous quarters, the four partial sums added in group order
__global__ __launch_bounds__(256) void landmarks_bwd_kernel(const float* __restrict__ reg, const float* __restrict__ g, int R, int V,
                                                            float* __restrict__ dverts) {
    extern __shared__ float lsm[];
    float* gl = lsm;                                  // [R][3]
    float* part = lsm + (size_t)R * 3;                // [4][64][3]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
    const int v = blockIdx.x * 64 + lane, vc = min(v, V - 1);
    for (int q = tid; q < R * 3; q += 256) gl[q] = g[(size_t)b * R * 3 + q];
    __syncthreads();
    const int per = (R + 3) / 4, r0 = grp * per, r1 = min(R, r0 + per);
    float a[3] = {0.f, 0.f, 0.f};
    constexpr int U = 12;
    for (int rr = r0; rr < r1; rr += U) {
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = reg[(size_t)min(rr + u, R - 1) * V + vc];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (rr + u < r1) {
#pragma unroll
                for (int k = 0; k < 3; ++k) a[k] = fmaf(w[u], gl[(rr + u) * 3 + k], a[k]);
            }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) part[(grp * 64 + lane) * 3 + k] = a[k];
    __syncthreads();
    if (grp == 0 && v < V) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            dverts[((size_t)b * V + v) * 3 + k] = ((part[lane * 3 + k] + part[(64 + lane) * 3 + k]) + part[(128 + lane) * 3 + k]) + part[(192 + lane) * 3 + k];
    }
}

int check_dims(chore_handle* h, const Dims& d, int B) {
    if (d.V < 1 || d.J < 2 || d.J > MAXJ || d.NB < 1 || d.NB > MAXNB || B < 1 || B > 65535)
        CHORE_FAIL(h, CHORE_EINVAL, "smpl: unsupported sizes V=%d J=%d betas=%d B=%d", d.V, d.J, d.NB, B);
    return CHORE_OK;
}

}  // namespace

extern "C" {

size_t chore_smpl_arena_bytes(int V, int J, int num_betas) {
    const Dims d{V, J, num_betas, 9 * (J - 1)};
    return arena_layout(d).total * sizeof(float);
}

size_t chore_smpl_workspace_bytes(int V, int J, int num_betas, int B) {
    const Dims d{V, J, num_betas, 9 * (J - 1)};
    return work_layout(d, B).total * sizeof(float);
}

int chore_smpl_pack(chore_handle* h, int V, int J, int num_betas, const float* v_template, const float* shapedirs,
                    const float* posedirs, const float* J_regressor, const float* weights, const int* parents_host,
                    void* arena, chore_stream_t stream) {
    CHORE_ENTER(h);
    const Dims d{V, J, num_betas, 9 * (J - 1)};
    if (int rc = check_dims(h, d, 1)) return rc;
    if (!v_template || !shapedirs || !posedirs || !J_regressor || !weights || !parents_host || !arena)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_smpl_pack: null argument");
    for (int i = 1; i < J; ++i)
        if (parents_host[i] < 0 || parents_host[i] >= i)
            CHORE_FAIL(h, CHORE_EINVAL, "chore_smpl_pack: parents[%d]=%d must precede the joint", i, parents_host[i]);
    hipStream_t s = (hipStream_t)stream;
    const Arena ar = arena_layout(d);
    float* m = (float*)arena;
    auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    hipLaunchKernelGGL(pack_copy_kernel, blocks((size_t)V * 3), dim3(256), 0, s, v_template, m + ar.T, (size_t)V * 3);
    hipLaunchKernelGGL(pack_copy_kernel, blocks((size_t)V * 3 * num_betas), dim3(256), 0, s, shapedirs, m + ar.S,
                       (size_t)V * 3 * num_betas);
    hipLaunchKernelGGL(pack_shapedirs_t_kernel, blocks((size_t)V * 3 * num_betas), dim3(256), 0, s, d, shapedirs, m + ar.ST);
    hipLaunchKernelGGL(pack_posedirs_kernel, blocks((size_t)V * 3 * d.NP), dim3(256), 0, s, d, posedirs, m + ar.PT);
    hipLaunchKernelGGL(pack_weights_kernel, blocks((size_t)V * J), dim3(256), 0, s, d, weights, m + ar.WT);
    hipLaunchKernelGGL(pack_jreg_kernel, dim3(J, 3 + 3 * num_betas), dim3(256), 0, s, d, J_regressor, v_template, shapedirs,
                       m + ar.JT, m + ar.JS);
    CHORE_HIP_CHECK(h, hipMemcpyAsync(m + ar.parents, parents_host, sizeof(int) * J, hipMemcpyHostToDevice, s));
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_smpl_lbs_fwd(chore_handle* h, const void* arena, int V, int J, int num_betas, const float* pose,
                       const float* betas, const float* trans, const float* offsets, float scale, int B, float* verts,
                       float* joints, float* v_posed, float* naked, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    const Dims d{V, J, num_betas, 9 * (J - 1)};
    if (int rc = check_dims(h, d, B)) return rc;
    if (!arena || !pose || !betas || !trans || !verts || !joints || !v_posed || !naked || !workspace)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_smpl_lbs_fwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const float* m = (const float*)arena;
    float* w = (float*)workspace;
    hipLaunchKernelGGL(lbs_pose_fwd_kernel, dim3(B), dim3(64), 0, s, d, m, pose, betas, trans, scale, w, B, joints);
    CHORE_LAUNCH_CHECK(h, s);
    const size_t smem = ((size_t)FB * d.NP + (size_t)FB * J * 12 + (size_t)FB * num_betas + (size_t)FB * VF_T) * sizeof(float);
    hipLaunchKernelGGL(lbs_vertex_fwd_kernel, dim3((V + VF_V - 1) / VF_V, (B + FB - 1) / FB), dim3(VF_T), smem, s, d, m, betas, trans,
                       offsets, scale, (const float*)w, B, verts, v_posed, naked);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_smpl_lbs_bwd(chore_handle* h, const void* arena, int V, int J, int num_betas, const float* pose, float scale,
                       int B, const float* v_posed, const float* g_verts, const float* g_joints, float* dpose,
                       float* dbetas, float* dtrans, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    const Dims d{V, J, num_betas, 9 * (J - 1)};
    if (int rc = check_dims(h, d, B)) return rc;
    if (!arena || !pose || !v_posed || !dpose || !dbetas || !dtrans || !workspace)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_smpl_lbs_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const float* m = (const float*)arena;
    float* w = (float*)workspace;
    hipLaunchKernelGGL(lbs_vertex_bwd_kernel, dim3((V + 255) / 256, B), dim3(256), std::max((size_t)J * 12 * sizeof(float), 256 * sizeof(double)), s, d, m, scale, w,
                       B, v_posed, g_verts);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(lbs_reduce_kernel, dim3(J + d.NP + num_betas, B), dim3(256), 0, s, d, m, w, B);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(lbs_pose_bwd_kernel, dim3(B), dim3(64), 0, s, d, m, pose, scale, (const float*)w, B, g_verts, g_joints,
                       dpose, dbetas, dtrans);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_landmarks_fwd(chore_handle* h, const float* reg, const float* verts, int R, int V, int B, float* out, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!reg || !verts || !out || R < 1 || V < 1 || B < 1 || B > 65535)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_landmarks_fwd: bad argument (R=%d V=%d B=%d)", R, V, B);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(landmarks_fwd_kernel, dim3(R, B), dim3(256), 0, s, reg, verts, R, V, out);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_landmarks_bwd(chore_handle* h, const float* reg, const float* g, int R, int V, int B, float* dverts, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!reg || !g || !dverts || R < 1 || R > 2048 || V < 1 || B < 1 || B > 65535)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_landmarks_bwd: bad argument (R=%d V=%d B=%d)", R, V, B);
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = ((size_t)R * 3 + 4 * 64 * 3) * sizeof(float);
    hipLaunchKernelGGL(landmarks_bwd_kernel, dim3((V + 63) / 64, B), dim3(256), smem, s, reg, g, R, V, dverts);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // extern "C"

// smpl_lbs.hip -- SMPL / SMPL-H linear blend skinning, forward and backward, for gfx950.
//
// Replaces SMPL_Layer.forward (lib_smpl/smplpytorch/smplpytorch/pytorch/smpl_layer.py:72-175) and
// its autograd: ~500 tiny ATen launches per call in the reference (52 batch_rodrigues, 51 chained
// 4x4 matmuls, 52 bmm, a (V*3 x 459) blend-shape matmul) become 2 kernels forward, 5 backward.
//   * pose kernel (one workgroup per frame): 52 Rodrigues rotations through the reference's quaternion
//     path (angle = ||theta + 1e-8||, rodrigues_layer.py:41-52), joint locations J = JT + JS*beta
//     (J_regressor folded into JT/JS at pack time), the kinematic chain and the skinning transforms.
//   * vertex kernel: thread = vertex; the pose blend shapes are stored p-major ([459][V*3]) so the
//     38 MB matrix streams once, fully coalesced, for a group of up to 4 frames whose pose maps and
//     transforms sit in LDS -- the op is HBM-bound on that matrix.
//   * backward: vertex kernel (recomputes the blend, emits d v_posed and d T per vertex), three
//     fixed-order reduction kernels (dA = W^T dT, d pose_map = P^T d v_posed, d beta = S^T d v_posed)
//     and the pose kernel backward (reverse kinematic chain, Rodrigues Jacobian by forward-mode
//     dual numbers).  All reductions are tree reductions in a fixed order: deterministic.
// Gradients are produced for pose, betas and trans (what the fitting optimises,
// recon/recon_fit_behave.py:224-291); offsets / v_posed / naked upstream gradients are not consumed.
#include "common.h"

namespace {

constexpr int FB = 4;          // frames processed together by the vertex kernels
constexpr int MAXJ = 64;       // joints (52 for SMPL-H, 24 for SMPL)
constexpr int MAXNB = 16;

struct Dims {
    int V, J, NB, NP;          // vertices, joints, betas, pose-blend directions = 9 (J-1)
};
struct Arena {                 // float offsets into the packed model
    size_t T, S, PT, WT, JT, JS, parents, total;
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
    a.total = o;
    return a;
}
struct Work {                  // float offsets into the per-call workspace
    size_t R, Jl, G, A, pm, gvp, dT, dA, dpm, dbv, total;
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
            for (int n = 0; n < d.NB; ++n) a += model[ar.JS + ((size_t)j * 3 + k) * d.NB + n] * betas[(size_t)b * d.NB + n];
            Jl[j][k] = a;
        }
    }
    __syncthreads();
    if (j == 0) {   // kinematic chain (smpl_layer.py:114-131): G_i = G_parent * [R_i | J_i - J_parent]
#pragma unroll
        for (int e = 0; e < 9; ++e) G[0][(e / 3) * 4 + e % 3] = R[0][e];
        G[0][3] = Jl[0][0]; G[0][7] = Jl[0][1]; G[0][11] = Jl[0][2];
        for (int i = 1; i < d.J; ++i) {
            const int p = parents[i];
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
// vertex kernel forward: thread = vertex, FB frames per workgroup
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lbs_vertex_fwd_kernel(Dims d, const float* __restrict__ model,
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
    const int b0 = blockIdx.y * FB, nf = min(FB, B - b0);
    const int tid = threadIdx.x, v = blockIdx.x * 256 + tid;
    for (int i = tid; i < nf * d.NP; i += 256) pm[i] = work[wk.pm + (size_t)b0 * d.NP + i];
    for (int i = tid; i < nf * d.J * 12; i += 256) A[i] = work[wk.A + (size_t)b0 * d.J * 12 + i];
    for (int i = tid; i < nf * d.NB; i += 256) bt[i] = betas[(size_t)b0 * d.NB + i];
    __syncthreads();
    if (v >= d.V) return;
    const size_t V3 = (size_t)d.V * 3;
    float acc[FB][3];
#pragma unroll
    for (int f = 0; f < FB; ++f)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = model[ar.T + (size_t)v * 3 + k];
            if (f < nf)
                for (int n = 0; n < d.NB; ++n) a += model[ar.S + ((size_t)v * 3 + k) * d.NB + n] * bt[f * d.NB + n];
            acc[f][k] = a;
        }
    const float* PT = model + ar.PT + (size_t)v * 3;
    for (int p = 0; p < d.NP; ++p) {
        const float p0 = PT[(size_t)p * V3], p1 = PT[(size_t)p * V3 + 1], p2 = PT[(size_t)p * V3 + 2];
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const float m = pm[f * d.NP + p];   // frames beyond nf read stale LDS: never stored
            acc[f][0] = fmaf(p0, m, acc[f][0]);
            acc[f][1] = fmaf(p1, m, acc[f][1]);
            acc[f][2] = fmaf(p2, m, acc[f][2]);
        }
    }
    for (int f = 0; f < nf; ++f) {
        const int b = b0 + f;
        const size_t o = ((size_t)b * d.V + v) * 3;
        float vp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            naked[o + k] = acc[f][k];
            vp[k] = acc[f][k] + (offsets ? offsets[o + k] : 0.f);
            v_posed[o + k] = vp[k];
        }
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < d.J; ++j) {
            const float w = model[ar.WT + (size_t)j * d.V + v];
            if (w != 0.f) {
                const float* Aj = A + ((size_t)f * d.J + j) * 12;
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = fmaf(w, Aj[e], T[e]);
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
            verts[o + r] = (T[r * 4] * vp[0] + T[r * 4 + 1] * vp[1] + T[r * 4 + 2] * vp[2] + T[r * 4 + 3]) * scale + trans[b * 3 + r];
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
    if (v >= d.V) return;
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
    for (int j = 0; j < d.J; ++j) {
        const float w = model[ar.WT + (size_t)j * d.V + v];
        if (w != 0.f) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(w, A[j * 12 + r * 4 + c], T[r * 3 + c]);
        }
    }
    const size_t o = ((size_t)b * d.V + v) * 3;
    const float g0 = g_verts ? g_verts[o] * scale : 0.f, g1 = g_verts ? g_verts[o + 1] * scale : 0.f,
                g2 = g_verts ? g_verts[o + 2] * scale : 0.f;
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
__global__ __launch_bounds__(256) void lbs_dA_kernel(Dims d, const float* __restrict__ model, float* __restrict__ work, int B) {
    __shared__ double sh[256];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float a[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) a[e] = 0.f;
    for (int v = tid; v < d.V; v += 256) {
        const float w = model[ar.WT + (size_t)j * d.V + v];
        if (w != 0.f) {
            const float* dT = work + wk.dT + ((size_t)b * d.V + v) * 12;
#pragma unroll
            for (int e = 0; e < 12; ++e) a[e] = fmaf(w, dT[e], a[e]);
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

// dpm[b][p] = sum_i PT[p][i] gvp[b][i]             grid (NP, ceil(B/FB))
__global__ __launch_bounds__(256) void lbs_dpm_kernel(Dims d, const float* __restrict__ model, float* __restrict__ work, int B) {
    __shared__ double sh[256];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int p = blockIdx.x, b0 = blockIdx.y * FB, nf = min(FB, B - b0), tid = threadIdx.x;
    const size_t V3 = (size_t)d.V * 3;
    const float* row = model + ar.PT + (size_t)p * V3;
    float a[FB];
#pragma unroll
    for (int f = 0; f < FB; ++f) a[f] = 0.f;
    for (size_t i = tid; i < V3; i += 256) {
        const float x = row[i];
#pragma unroll
        for (int f = 0; f < FB; ++f)
            if (f < nf) a[f] = fmaf(x, work[wk.gvp + (size_t)(b0 + f) * V3 + i], a[f]);
    }
    float r[FB];
    block_reduce<FB>(a, sh, tid, r);
    if (tid == 0)
        for (int f = 0; f < nf; ++f) work[wk.dpm + (size_t)(b0 + f) * d.NP + p] = r[f];
}

// dbv[b][n] = sum_i S[i][n] gvp[b][i]              grid (NB, B)
__global__ __launch_bounds__(256) void lbs_dbeta_kernel(Dims d, const float* __restrict__ model, float* __restrict__ work, int B) {
    __shared__ double sh[256];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int n = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t V3 = (size_t)d.V * 3;
    float a[1] = {0.f};
    for (size_t i = tid; i < V3; i += 256) a[0] = fmaf(model[ar.S + i * d.NB + n], work[wk.gvp + (size_t)b * V3 + i], a[0]);
    float r[1];
    block_reduce<1>(a, sh, tid, r);
    if (tid == 0) work[wk.dbv + (size_t)b * d.NB + n] = r[0];
}

// pose kernel backward: reverse chain, Rodrigues Jacobian, betas through J, trans
__global__ __launch_bounds__(64) void lbs_pose_bwd_kernel(Dims d, const float* __restrict__ model, const float* __restrict__ pose,
                                                          float scale, const float* __restrict__ work, int B,
                                                          const float* __restrict__ g_verts, const float* __restrict__ g_joints,
                                                          float* __restrict__ dpose, float* __restrict__ dbetas,
                                                          float* __restrict__ dtrans) {
    __shared__ float R[MAXJ][9], Jl[MAXJ][3], G[MAXJ][12], dG[MAXJ][12], dR[MAXJ][9], dJ[MAXJ][3];
    __shared__ double tsum[64][3];
    const Arena ar = arena_layout(d);
    const Work wk = work_layout(d, B);
    const int b = blockIdx.x, j = threadIdx.x;
    const int* parents = (const int*)(model + ar.parents);
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
    // d trans = sum_v g_verts + sum_j g_joints (trans is added after the scale)
    {
        double a[3] = {0.0, 0.0, 0.0};
        if (g_verts)
            for (int v = j; v < d.V; v += 64)
#pragma unroll
                for (int k = 0; k < 3; ++k) a[k] += (double)g_verts[((size_t)b * d.V + v) * 3 + k];
        if (g_joints && j < d.J)
#pragma unroll
            for (int k = 0; k < 3; ++k) a[k] += (double)g_joints[((size_t)b * d.J + j) * 3 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tsum[j][k] = a[k];
    }
    __syncthreads();
    if (j == 0) {
        for (int i = d.J - 1; i >= 1; --i) {   // children before parents (parents[i] < i in SMPL trees)
            const int p = parents[i];
            const float t[3] = {Jl[i][0] - Jl[p][0], Jl[i][1] - Jl[p][1], Jl[i][2] - Jl[p][2]};
            float dt[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) dt[c] = G[p][c] * dG[i][3] + G[p][4 + c] * dG[i][7] + G[p][8 + c] * dG[i][11];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // dG_p.R += dG_i.R R_i^T + dG_i.t (x) t ;  dR_i += G_p.R^T dG_i.R
                    dG[p][r * 4 + c] += dG[i][r * 4] * R[i][c * 3] + dG[i][r * 4 + 1] * R[i][c * 3 + 1] +
                                        dG[i][r * 4 + 2] * R[i][c * 3 + 2] + dG[i][r * 4 + 3] * t[c];
                    dR[i][r * 3 + c] += G[p][r] * dG[i][c] + G[p][4 + r] * dG[i][4 + c] + G[p][8 + r] * dG[i][8 + c];
                }
                dG[p][r * 4 + 3] += dG[i][r * 4 + 3];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { dJ[i][c] += dt[c]; dJ[p][c] -= dt[c]; }
        }
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
        for (int i = 0; i < d.J; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) a += model[ar.JS + ((size_t)i * 3 + k) * d.NB + j] * dJ[i][k];
        dbetas[(size_t)b * d.NB + j] = a;
    }
    if (j < 3) {
        double a = 0.0;
        for (int i = 0; i < 64; ++i) a += tsum[i][j];
        dtrans[b * 3 + j] = (float)a;
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
    const size_t smem = ((size_t)FB * d.NP + (size_t)FB * J * 12 + (size_t)FB * num_betas) * sizeof(float);
    hipLaunchKernelGGL(lbs_vertex_fwd_kernel, dim3((V + 255) / 256, (B + FB - 1) / FB), dim3(256), smem, s, d, m, betas, trans,
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
    hipLaunchKernelGGL(lbs_vertex_bwd_kernel, dim3((V + 255) / 256, B), dim3(256), (size_t)J * 12 * sizeof(float), s, d, m, scale, w,
                       B, v_posed, g_verts);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(lbs_dA_kernel, dim3(J, B), dim3(256), 0, s, d, m, w, B);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(lbs_dpm_kernel, dim3(d.NP, (B + FB - 1) / FB), dim3(256), 0, s, d, m, w, B);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(lbs_dbeta_kernel, dim3(num_betas, B), dim3(256), 0, s, d, m, w, B);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(lbs_pose_bwd_kernel, dim3(B), dim3(64), 0, s, d, m, pose, scale, (const float*)w, B, g_verts, g_joints,
                       dpose, dbetas, dtrans);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // extern "C"

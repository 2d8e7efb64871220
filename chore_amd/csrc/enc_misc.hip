// enc_misc.hip -- the bandwidth-bound kernels of the encoder (gfx950): 7x7 stem convolution,
// GroupNorm statistics / apply, 2x2 average pooling, bicubic x2 upsample + add.
//
// Reference semantics (ATen ops called from model/HGFilters.py:26-50,144-185 and
// model/net_util.py:374-396): group_norm(32 groups, eps 1e-5, biased variance), avg_pool2d(2,2),
// interpolate(scale 2, bicubic, align_corners=True) = cubic convolution A=-0.75 with
// border-clamped taps and source coordinate dst*(in-1)/(out-1).
// All tensors NHWC; every thread moves 4 channels (16 B fp32 / 8 B bf16) so accesses coalesce.
#include "enc_common.h"

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ f32x4 ld(const bf16_t* p) {
        const u16x4 v = *(const u16x4*)p;
        f32x4 r = {bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
        return r;
    }
    static __device__ __forceinline__ void st(bf16_t* p, f32x4 v) {
        u16x4 r = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(u16x4*)p = r;
    }
};

// ------------------------------------------------------------------------------------------------
// stem: conv 7x7 stride 2 pad 3, Cin(5) -> 64, + bias      (model/HGFilters.py:102,149)
// block = 8x8 output pixels x 4 groups of 16 channels; input patch and weights staged in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int STEM_T = 8;
constexpr int STEM_P = 2 * STEM_T + 5;  // 21
constexpr int STEM_MAXC = 8;

template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, int B, int Cin, int H, int W,
                                                   const float* __restrict__ wk, const float* __restrict__ bias,
                                                   T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wl = sm;                       // [Cin*49][64]
    float* patch = sm + Cin * 49 * 64;    // [Cin][21][21]
    const int OH = H / 2, OW = W / 2;
    const int b = blockIdx.z, ty0 = blockIdx.y * STEM_T, tx0 = blockIdx.x * STEM_T;
    const int tid = threadIdx.x;
    for (int i = tid; i < Cin * 49 * 64 / 4; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)wk)[i];
    const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
    for (int i = tid; i < Cin * STEM_P * STEM_P; i += 256) {
        const int c = i / (STEM_P * STEM_P), r = i % (STEM_P * STEM_P);
        const int y = iy0 + r / STEM_P, x = ix0 + r % STEM_P;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = img[(((size_t)b * Cin + c) * H + y) * W + x];
        patch[i] = v;
    }
    __syncthreads();
    const int pix = tid & 63, cg = tid >> 6;  // wave = channel group -> weight reads broadcast
    const int py = pix >> 3, px = pix & 7;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[cg * 16 + j];
    for (int c = 0; c < Cin; ++c) {
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float v = patch[(c * STEM_P + 2 * py + ky) * STEM_P + 2 * px + kx];
                const f32x4* w4 = (const f32x4*)(wl + ((c * 7 + ky) * 7 + kx) * 64 + cg * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = w4[q];
                    acc[q * 4 + 0] = fmaf(v, w[0], acc[q * 4 + 0]);
                    acc[q * 4 + 1] = fmaf(v, w[1], acc[q * 4 + 1]);
                    acc[q * 4 + 2] = fmaf(v, w[2], acc[q * 4 + 2]);
                    acc[q * 4 + 3] = fmaf(v, w[3], acc[q * 4 + 3]);
                }
            }
        }
    }
    const int oy = ty0 + py, ox = tx0 + px;
    if (oy < OH && ox < OW) {
        T* o = out + (((size_t)b * OH + oy) * OW + ox) * 64 + cg * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
            Vec4<T>::st(o + q * 4, v);
        }
    }
}

__global__ void pack_stem_kernel(int Cin, const float* __restrict__ w, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // dst index [k][o]
    if (i >= Cin * 49 * 64) return;
    const int o = i & 63, k = i >> 6;
    dst[i] = w[(size_t)o * Cin * 49 + k];
}

int launch_pack_stem(chore_handle* h, int Cin, const float* w, float* dst, hipStream_t s) {
    const int n = Cin * 49 * 64;
    hipLaunchKernelGGL(pack_stem_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Cin, w, dst);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int launch_stem(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W, const float* wk,
                const float* bias, void* out, hipStream_t s) {
    if (Cin > STEM_MAXC) CHORE_FAIL(h, CHORE_EINVAL, "stem: Cin > %d", STEM_MAXC);
    const int OH = H / 2, OW = W / 2;
    dim3 grid((OW + STEM_T - 1) / STEM_T, (OH + STEM_T - 1) / STEM_T, B);
    const size_t smem = ((size_t)Cin * 49 * 64 + (size_t)Cin * STEM_P * STEM_P) * sizeof(float);
    static bool attr[2] = {false, false};
    if (dtype == CHORE_F32) {
        if (!attr[0]) {
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)stem_kernel<float>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr[0] = true;
        }
        hipLaunchKernelGGL(stem_kernel<float>, grid, dim3(256), smem, s, images, B, Cin, H, W, wk, bias, (float*)out);
    } else {
        if (!attr[1]) {
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)stem_kernel<bf16_t>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr[1] = true;
        }
        hipLaunchKernelGGL(stem_kernel<bf16_t>, grid, dim3(256), smem, s, images, B, Cin, H, W, wk, bias,
                           (bf16_t*)out);
    }
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics: deterministic two-stage reduction.
//   partial[b][s][g] = (sum, sum of squares) over the s-th slice of pixels, fp32
//   finalize: fp64 combine -> mean, rstd -> per-channel scale = rstd*gamma, shift = beta - mean*scale
// ------------------------------------------------------------------------------------------------
int gn_splits(int HW) {
    int s = HW / 64;  // >= 64 pixels per slice
    if (s < 1) s = 1;
    if (s > GN_SPLITS_MAX) s = GN_SPLITS_MAX;
    return s;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int cs, int co, int C, int HW,
                                                         int S, float* __restrict__ partial) {
    __shared__ float red[2][1024];
    __shared__ float chs[2][256];
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int tpr = C / 4, P = 256 / tpr;
    const int cv = tid % tpr, pl = tid / tpr;
    const int p0 = (int)((long long)HW * s / S), p1 = (int)((long long)HW * (s + 1) / S);
    const T* base = x + (size_t)b * HW * cs + co + cv * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + pl; p < p1; p += P) {
        const f32x4 v = Vec4<T>::ld(base + (size_t)p * cs);
        sum += v;
        sq += v * v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][pl * C + cv * 4 + j] = sum[j];
        red[1][pl * C + cv * 4 + j] = sq[j];
    }
    __syncthreads();
    if (tid < C) {
        float a = 0.f, q = 0.f;
        for (int i = 0; i < P; ++i) { a += red[0][i * C + tid]; q += red[1][i * C + tid]; }
        chs[0][tid] = a;
        chs[1][tid] = q;
    }
    __syncthreads();
    if (tid < GN_GROUPS) {
        const int gs = C / GN_GROUPS;
        float a = 0.f, q = 0.f;
        for (int i = 0; i < gs; ++i) { a += chs[0][tid * gs + i]; q += chs[1][tid * gs + i]; }
        float* o = partial + (((size_t)b * S + s) * GN_GROUPS + tid) * 2;
        o[0] = a;
        o[1] = q;
    }
}

int launch_gn_partial(chore_handle* h, int dtype, const View& x, int B, int HW, float* partial, hipStream_t s) {
    if (x.C % GN_GROUPS || x.C > 256 || x.C < 32) CHORE_FAIL(h, CHORE_EINVAL, "gn: unsupported C=%d", x.C);
    const int S = gn_splits(HW);
    dim3 grid(S, B);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x.p, x.cs, x.co, x.C, HW, S,
                           partial);
    else
        hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x.p, x.cs, x.co, x.C, HW,
                           S, partial);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// block-wide fixed-order fp64 reduction of (a, q); result valid in thread 0
__device__ __forceinline__ void block_reduce2(double& a, double& q, double (*sh)[256], int tid) {
    sh[0][tid] = a;
    sh[1][tid] = q;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) {
            sh[0][tid] += sh[0][tid + o];
            sh[1][tid] += sh[1][tid + o];
        }
        __syncthreads();
    }
    a = sh[0][0];
    q = sh[1][0];
}

__device__ __forceinline__ void write_scale_shift(double a, double q, int HW, int gs, int C, int b, int g,
                                                  const float* gamma, const float* beta, float* ss, int tid) {
    const double n = (double)HW * gs;
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
    if (tid < gs) {
        const int c = g * gs + tid;
        const float scale = rstd * gamma[c];
        ss[((size_t)b * C + c) * 2 + 0] = scale;
        ss[((size_t)b * C + c) * 2 + 1] = beta[c] - (float)mean * scale;
    }
}

// one block per (group, image)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int S, int HW, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ ss) {
    __shared__ double sh[2][256];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    double a = 0.0, q = 0.0;
    for (int s = tid; s < S; s += 256) {
        const float* p = partial + (((size_t)b * S + s) * GN_GROUPS + g) * 2;
        a += (double)p[0];
        q += (double)p[1];
    }
    block_reduce2(a, q, sh, tid);
    write_scale_shift(a, q, HW, C / GN_GROUPS, C, b, g, gamma, beta, ss, tid);
}

int launch_gn_finalize(chore_handle* h, const float* partial, int B, int HW, int C, const float* gamma,
                       const float* beta, float* ss, hipStream_t s) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(GN_GROUPS, B), dim3(256), 0, s, partial, gn_splits(HW), HW, C, gamma,
                       beta, ss);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// finalize from the tile partials the conv epilogues wrote (fixed summation order -> deterministic)
__global__ __launch_bounds__(256) void gn_finalize_tiles_kernel(TileStats ts, int HW, int C,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ ss) {
    __shared__ double sh[2][256];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int gs = C / GN_GROUPS;
    int sl = 0;   // a group never straddles two slices (slice widths are multiples of the group size)
    while (sl + 1 < ts.nslices && g * gs >= ts.c_end[sl]) ++sl;
    const int nt = ts.ntiles[sl];
    const float* base = ts.p + ((size_t)b * ts.max_tiles * C + g * gs) * 2;
    double a = 0.0, q = 0.0;
    for (int i = tid; i < nt * gs; i += 256) {   // element i = (tile, channel-in-group)
        const int t = i / gs, j = i % gs;
        const float* p = base + ((size_t)t * C + j) * 2;
        a += (double)p[0];
        q += (double)p[1];
    }
    block_reduce2(a, q, sh, tid);
    write_scale_shift(a, q, HW, gs, C, b, g, gamma, beta, ss, tid);
}

int launch_gn_finalize_tiles(chore_handle* h, const TileStats& ts, int B, int HW, int C, const float* gamma,
                             const float* beta, float* ss, hipStream_t s) {
    if (C > 256 || C % GN_GROUPS) CHORE_FAIL(h, CHORE_EINVAL, "gn: unsupported C=%d", C);
    hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(GN_GROUPS, B), dim3(256), 0, s, ts, HW, C, gamma, beta, ss);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// elementwise: y = relu(x*scale + shift)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_relu_kernel(const T* __restrict__ x, int xcs, int xco,
                                                            const float* __restrict__ ss, T* __restrict__ y,
                                                            int ycs, int yco, int C, int HW, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int tpr = C / 4;
    const int cv = (int)(i % tpr);
    const size_t p = i / tpr;  // b*HW + pixel
    const int b = (int)(p / HW);
    const f32x4 v = Vec4<T>::ld(x + p * xcs + xco + cv * 4);
    const float* s = ss + ((size_t)b * C + cv * 4) * 2;
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t = fmaf(v[j], s[2 * j], s[2 * j + 1]);
        r[j] = t > 0.f ? t : 0.f;
    }
    Vec4<T>::st(y + p * ycs + yco + cv * 4, r);
}

int launch_gn_apply_relu(chore_handle* h, int dtype, const View& x, const float* ss, const View& y, int B, int HW,
                         hipStream_t s) {
    const size_t total4 = (size_t)B * HW * (x.C / 4);
    const int blocks = (int)((total4 + 255) / 256);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(gn_apply_relu_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x.p, x.cs, x.co,
                           ss, (float*)y.p, y.cs, y.co, x.C, HW, total4);
    else
        hipLaunchKernelGGL(gn_apply_relu_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x.p, x.cs,
                           x.co, ss, (bf16_t*)y.p, y.cs, y.co, x.C, HW, total4);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// 2x2 average pooling   (F.avg_pool2d(x, 2, stride=2), HGFilters.py:32,152)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool2_kernel(const T* __restrict__ x, int xcs, int xco, T* __restrict__ y,
                                                       int ycs, int yco, int C, int H, int W, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int tpr = C / 4, OH = H / 2, OW = W / 2;
    const int cv = (int)(i % tpr);
    size_t p = i / tpr;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const T* src = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * xcs + xco + cv * 4;
    const f32x4 a = Vec4<T>::ld(src), bb = Vec4<T>::ld(src + xcs);
    const f32x4 c = Vec4<T>::ld(src + (size_t)W * xcs), d = Vec4<T>::ld(src + (size_t)W * xcs + xcs);
    const f32x4 r = (((a + bb) + c) + d) * 0.25f;
    Vec4<T>::st(y + (((size_t)b * OH + oy) * OW + ox) * ycs + yco + cv * 4, r);
}

int launch_avgpool2(chore_handle* h, int dtype, const View& x, const View& y, int B, int H, int W, hipStream_t s) {
    const size_t total4 = (size_t)B * (H / 2) * (W / 2) * (x.C / 4);
    const int blocks = (int)((total4 + 255) / 256);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(avgpool2_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x.p, x.cs, x.co,
                           (float*)y.p, y.cs, y.co, x.C, H, W, total4);
    else
        hipLaunchKernelGGL(avgpool2_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x.p, x.cs, x.co,
                           (bf16_t*)y.p, y.cs, y.co, x.C, H, W, total4);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// y = a + bicubic_up2(low), align_corners=True      (HGFilters.py:47,50)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

template <typename T>
__global__ __launch_bounds__(256) void upadd_kernel(const T* a, int acs, int aco,
                                                    const T* __restrict__ low, int lcs, int lco, T* y,
                                                    int ycs, int yco, int C, int H, int W, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int tpr = C / 4, OH = 2 * H, OW = 2 * W;
    const int cv = (int)(i % tpr);
    size_t p = i / tpr;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const float sy = (OH > 1) ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sx = (OW > 1) ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const float ry = sy * (float)oy, rx = sx * (float)ox;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    float cy[4], cx[4];
    cubic_coeffs(ry - fy, cy);
    cubic_coeffs(rx - fx, cx);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int yy = iy - 1 + r;
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        f32x4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int xx = ix - 1 + q;
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            const f32x4 v = Vec4<T>::ld(low + (((size_t)b * H + yy) * W + xx) * lcs + lco + cv * 4);
            row += v * cx[q];
        }
        acc += row * cy[r];
    }
    const size_t o = ((size_t)b * OH + oy) * OW + ox;
    const f32x4 av = Vec4<T>::ld(a + o * acs + aco + cv * 4);
    Vec4<T>::st(y + o * ycs + yco + cv * 4, av + acc);
}

int launch_upadd(chore_handle* h, int dtype, const View& a, const View& low, const View& y, int B, int H, int W,
                 hipStream_t s) {
    const size_t total4 = (size_t)B * (2 * H) * (2 * W) * (a.C / 4);
    const int blocks = (int)((total4 + 255) / 256);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(upadd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)a.p, a.cs, a.co,
                           (const float*)low.p, low.cs, low.co, (float*)y.p, y.cs, y.co, a.C, H, W, total4);
    else
        hipLaunchKernelGGL(upadd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)a.p, a.cs, a.co,
                           (const bf16_t*)low.p, low.cs, low.co, (bf16_t*)y.p, y.cs, y.co, a.C, H, W, total4);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
int launch_copy_f32(chore_handle* h, const float* src, float* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// generator.hip -- device-resident bookkeeping of the surface-point generator, gfx950.
//
// Generator.gen_pc_batch (/root/reference/recon/generator.py:123-188) filters the projected samples with boolean masks,
// appends the survivors to per-example Python lists and resamples the next round from the masked INPUT samples -- a
// dozen device->host round trips per example and round (mask.sum(), boolean indexing, randint bounds).  These kernels
// keep all of it on the device; the host reads ONE scalar per round (the loop condition):
//   gen_compact_kernel   ordered compaction of a mask: order[b][j] = index of the j-th set element, counts[b]
//                        (one workgroup per example, ballot + wave prefix + running total: same order as x[mask])
//   gen_append_kernel    dst[b][c][offsets[b] + j] = src[b][c][order[b][j]], j < counts[b]   (generic strides: point-
//                        major (B,N,3) and channel-major (B,C,N) tensors alike), clipped at the buffer capacity
//   gen_advance_kernel   offsets[b] += counts[b];  total += min_b counts[b]           (generator.py:156-158)
//   gen_resample_kernel  next samples: masked input samples drawn with replacement + N(0, (threshold/3)^2), or, if an
//                        example has <= 1 survivor, initial samples + N(0, 0.5^2)     (generator.py:163-177); the
//                        draw is floor(u * k) from uniform numbers u supplied by the caller (a device RNG: the
//                        reference draws torch.randint on the CPU -- a documented divergence of the stream, not of
//                        the distribution)
// All HBM-bound bookkeeping (a few hundred KB per round); no float arithmetic except the perturbation.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void gen_compact_kernel(const unsigned char* __restrict__ mask, int N, int* __restrict__ order,
                                                          int* __restrict__ counts) {
    __shared__ int wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned char* m = mask + (size_t)b * N;
    int* o = order + (size_t)b * N;
    int total = 0;
    for (int base = 0; base < N; base += 256) {
        const int i = base + tid;
        const bool set = i < N && m[i] != 0;
        const unsigned long long bal = __ballot(set);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int off = total;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (set) o[off + before] = i;
        total += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) counts[b] = total;
}

__global__ void gen_append_kernel(const float* __restrict__ src, long long ss_b, long long ss_c, long long ss_n,
                                  const int* __restrict__ order, const int* __restrict__ counts, const int* __restrict__ offsets,
                                  float* __restrict__ dst, long long ds_b, long long ds_c, long long ds_n, int N, int cap) {
    const int b = blockIdx.z, c = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= counts[b]) return;
    const long long at = (long long)offsets[b] + j;
    if (at >= cap) return;
    dst[b * ds_b + c * ds_c + at * ds_n] = src[b * ss_b + c * ss_c + (long long)order[(size_t)b * N + j] * ss_n];
}

__global__ void gen_advance_kernel(const int* __restrict__ counts, int B, int* __restrict__ offsets, int* __restrict__ total) {
    if (blockIdx.x || threadIdx.x) return;
    int mn = counts[0];
    for (int b = 0; b < B; ++b) {
        offsets[b] += counts[b];
        mn = counts[b] < mn ? counts[b] : mn;
    }
    *total += mn;
}

__global__ void gen_resample_kernel(const float* __restrict__ samples, int N, const int* __restrict__ order,
                                    const int* __restrict__ counts, const float* __restrict__ init, int Ninit,
                                    const float* __restrict__ u, const float* __restrict__ noise, int M, float sigma,
                                    float* __restrict__ out) {
    const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int k = counts[b];
    const float uj = u[(size_t)b * M + j];
    const float* src;
    float s;
    if (k > 1) {
        int idx = (int)(uj * (float)k);
        idx = idx < k ? idx : k - 1;
        src = samples + ((size_t)b * N + order[(size_t)b * N + idx]) * 3;
        s = sigma;
    } else {
        int idx = (int)(uj * (float)Ninit);
        idx = idx < Ninit ? idx : Ninit - 1;
        src = init + ((size_t)b * Ninit + idx) * 3;
        s = 0.5f;
    }
    const float* nz = noise + ((size_t)b * M + j) * 3;
    float* o = out + ((size_t)b * M + j) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d] = src[d] + s * nz[d];
}

// one projection step of Alg. 1 (generator.py:50-79): the upstream gradient of sum(clamp(df_k, max = thr)) and
// p <- p - normalize(grad_p) * clamp(df_k, max = thr)   (F.normalize: v / max(||v||, 1e-12))
__global__ void gen_clamp_mask_kernel(const float* __restrict__ df, int k, float thr, int N, float* __restrict__ g) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    g[((size_t)b * 2 + k) * N + n] = df[((size_t)b * 2 + k) * N + n] <= thr ? 1.f : 0.f;     // clamp passes the gradient where x <= max
    g[((size_t)b * 2 + (1 - k)) * N + n] = 0.f;
}
__global__ void gen_surface_step_kernel(const float* __restrict__ p, const float* __restrict__ grad, const float* __restrict__ df, int k,
                                        float thr, int N, float* __restrict__ out) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const size_t o = ((size_t)b * N + n) * 3;
    const float gx = grad[o], gy = grad[o + 1], gz = grad[o + 2];
    const float d = fminf(df[((size_t)b * 2 + k) * N + n], thr);
    const float den = fmaxf(sqrtf((gx * gx + gy * gy) + gz * gz), 1e-12f);
    out[o] = p[o] - gx / den * d;
    out[o + 1] = p[o + 1] - gy / den * d;
    out[o + 2] = p[o + 2] - gz / den * d;
}

}  // namespace

extern "C" {

int chore_gen_compact(chore_handle* h, const unsigned char* mask, int B, int N, int* order, int* counts, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!mask || !order || !counts || B <= 0 || N <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_compact: bad argument");
    hipLaunchKernelGGL(gen_compact_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mask, N, order, counts);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

int chore_gen_append(chore_handle* h, const float* src, long long ss_b, long long ss_c, long long ss_n, const int* order,
                     const int* counts, const int* offsets, float* dst, long long ds_b, long long ds_c, long long ds_n, int B,
                     int C, int N, int cap, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!src || !order || !counts || !offsets || !dst || B <= 0 || C <= 0 || N <= 0 || cap <= 0)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_append: bad argument");
    hipLaunchKernelGGL(gen_append_kernel, dim3((N + 255) / 256, C, B), dim3(256), 0, (hipStream_t)stream, src, ss_b, ss_c, ss_n, order,
                       counts, offsets, dst, ds_b, ds_c, ds_n, N, cap);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

int chore_gen_advance(chore_handle* h, const int* counts, int B, int* offsets, int* total, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!counts || !offsets || !total || B <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_advance: bad argument");
    hipLaunchKernelGGL(gen_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counts, B, offsets, total);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

int chore_gen_resample(chore_handle* h, const float* samples, int B, int N, const int* order, const int* counts, const float* init,
                       int Ninit, const float* u, const float* noise, int M, float sigma, float* out, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!samples || !order || !counts || !init || !u || !noise || !out || B <= 0 || N <= 0 || Ninit <= 0 || M <= 0)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_resample: bad argument");
    hipLaunchKernelGGL(gen_resample_kernel, dim3((M + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, samples, N, order, counts, init,
                       Ninit, u, noise, M, sigma, out);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

int chore_gen_clamp_mask(chore_handle* h, const float* df, int k, float thr, int B, int N, float* g, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!df || !g || B <= 0 || B > 65535 || N <= 0 || (k != 0 && k != 1)) CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_clamp_mask: bad argument");
    hipLaunchKernelGGL(gen_clamp_mask_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, df, k, thr, N, g);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

int chore_gen_surface_step(chore_handle* h, const float* points, const float* grad, const float* df, int k, float thr, int B, int N,
                           float* out, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!points || !grad || !df || !out || B <= 0 || B > 65535 || N <= 0 || (k != 0 && k != 1))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_surface_step: bad argument");
    hipLaunchKernelGGL(gen_surface_step_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, points, grad, df, k, thr, N,
                       out);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

}  // extern "C"

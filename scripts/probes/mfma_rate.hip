// mfma_rate.hip -- issue rate of v_mfma_f32_32x32x16_f16 streams as a function of how many independent accumulators
// take turns (development probe).  One wave per SIMD (256-thread workgroups, one per CU), every CU busy.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_rate.hip -o scripts/probes/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void k(const float* in, float* out, unsigned long long* ticks, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (_Float16)in[(threadIdx.x * 8 + j + i * 64) & 4095];
            b[i][j] = (_Float16)in[(threadIdx.x * 8 + j + i * 64 + 2048) & 4095];
        }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        // CHAIN dependent MFMAs per accumulator per round, the accumulators take turns
#pragma unroll
        for (int c = 0; c < CHAIN; ++c)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + c) & 3], b[(i * 2 + c) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}

template <int NACC, int CHAIN>
void run(const char* what, const float* din, float* dout, unsigned long long* dt, int wgs) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, CHAIN>), dim3(wgs), dim3(256), 0, 0, din, dout, dt, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NACC, CHAIN>), dim3(wgs), dim3(256), 0, 0, din, dout, dt, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[2];
    hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC * CHAIN;
    printf("%-10s wgs %4d  acc %2d chain %d: %.1f cycles / MFMA, %.2f GHz, %.0f TF\n", what, wgs, NACC, CHAIN, t[0] / n, t[0] / (t[1] * 10.0),
           n * 4 * wgs * 32768.0 / (ms * 1e-3) * 1e-12);
}

int main() {
    float *din, *dout;
    unsigned long long* dt;
    hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 1024 * 256 * 4); hipMalloc(&dt, 16);
    for (int z = 0; z < 2; ++z) {
        std::vector<float> h(4096);
        unsigned s = 1;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = z ? 0.f : ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
        hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        const char* w = z ? "zeros" : "random";
        for (int wgs : {256, 1}) {
            run<1, 1>(w, din, dout, dt, wgs);
            run<2, 1>(w, din, dout, dt, wgs);
            run<4, 1>(w, din, dout, dt, wgs);
            run<4, 3>(w, din, dout, dt, wgs);
            run<8, 1>(w, din, dout, dt, wgs);
            run<12, 1>(w, din, dout, dt, wgs);
        }
    }
    return 0;
}

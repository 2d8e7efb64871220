// mfma_interleave_probe.hip -- does vector-ALU work overlap the matrix pipe when it is interleaved INSIDE the wave that issues the
// MFMAs?  (round 5 probe; companion of mfma_valu_probe.hip, where the work of ANOTHER wave of the SIMD does not overlap at all.)
// One wave per SIMD (NW = 4) or two (NW = 8), every CU busy; per v_mfma_f32_32x32x16_f16 the wave issues NV independent v_fma_f32
// (or one ds_read_b128 per MFMA with LDSR).  cycles per MFMA against NV: flat up to NV ~ 7 means the VALU runs in the MFMA's shadow.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_interleave_probe.hip -o scripts/probes/mfma_interleave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool LDSR>
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* ticks, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1040];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 1040 / 4; i += blockDim.x) ((float*)lds)[i] = in[i & 4095];
    __syncthreads();
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (_Float16)in[(tid * 8 + j + i * 64) & 4095];
            b[i][j] = (_Float16)in[(tid * 8 + j + i * 64 + 2048) & 4095];
        }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[(tid + i * 512) & 4095];
    const float c0 = in[7], c1 = in[9];
    const char* q = lds + (lane & 31) * 1040 + (lane >> 5) * 16;
    u32x4 x[8];
    for (int i = 0; i < 8; ++i) x[i] = u32x4{0u, 0u, 0u, 0u};
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i * 2) & 3], acc[i], 0, 0, 0);
            if constexpr (LDSR) x[i] = *(const volatile u32x4*)(q + ((it + i) & 31) * 32);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[(i + j) & 7] = fmaf(v[(i + j) & 7], c0, c1);
            // pin the order: one MFMA, then its NV vector instructions (and the LDS read)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (LDSR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        s += v[i] + __uint_as_float(x[i][0]);
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == gridDim.x / 2 && (tid == 0 || tid == 256)) { ticks[tid ? 2 : 0] = t1 - t0; ticks[(tid ? 2 : 0) + 1] = w1 - w0; }
}

template <int NV, bool LDSR>
void run(const float* din, float* dout, unsigned long long* dt, int nw) {
    const int iters = 2000;
    hipMemset(dt, 0, 32);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, LDSR>), dim3(256), dim3(64 * nw), 0, 0, din, dout, dt, iters);
    hipDeviceSynchronize();
    unsigned long long t[4] = {0, 0, 0, 0};
    hipMemcpy(t, dt, 32, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    printf("waves/SIMD %d  v_fma per MFMA %2d  ds_read_b128 per MFMA %d: wave 0 %6.1f cycles per MFMA, %.2f GHz, %7.1f us", nw / 4, NV, LDSR ? 1 : 0, t[0] / n,
           t[0] / (t[1] * 10.0), t[1] * 0.01);
    if (nw == 8) printf("   | wave 4 (same SIMD, younger): %7.1f us", t[3] * 0.01);
    printf("\n");
}

int main() {
    float *din, *dout;
    unsigned long long* dt;
    hipMalloc(&din, 1 << 20); hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dt, 64);
    float h[4096];
    unsigned s = 12345u;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f + 0.25f; }
    for (int i = 0; i < (1 << 20) / 16384; ++i) hipMemcpy((char*)din + i * 16384, h, 16384, hipMemcpyHostToDevice);
    for (int nw = 4; nw <= 8; nw += 4) {
        run<0, false>(din, dout, dt, nw); run<2, false>(din, dout, dt, nw); run<4, false>(din, dout, dt, nw); run<6, false>(din, dout, dt, nw);
        run<8, false>(din, dout, dt, nw); run<12, false>(din, dout, dt, nw); run<16, false>(din, dout, dt, nw);
        run<0, true>(din, dout, dt, nw); run<4, true>(din, dout, dt, nw);
    }
    return 0;
}

// mfma_valu_probe.hip -- do the matrix pipe and the other instruction classes of a SIMD overlap when they come from DIFFERENT
// waves?  (development probe, round 5: conv_rw_kernel's and conv_pc_kernel's phases add up instead of overlapping.)
// One workgroup of 8 waves per CU (two per SIMD), every CU busy.  Waves 0-3 ("m") issue a stream of independent
// v_mfma_f32_32x32x16_f16 (8 accumulators taking turns); waves 4-7 ("o") issue a stream of OTHER work:
//   kind 0  v_fma_f32 (8 independent chains)         kind 1  v_pk_fma_f32        kind 2  ds_read_b128 (conflict-free rows)
//   kind 3  the fp16 hi / lo split of fp32 values (cvt, cvt back, sub, cvt)      kind 4  global_load_dwordx4 (L2-resident)
// Each role is timed alone (the other role's waves exit at once) and together, with the shader clock of one wave of each role:
//   together ~ max(alone) -> the classes overlap;  together ~ sum(alone) -> they share an issue resource.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_probe.hip -o scripts/probes/mfma_valu_probe && scripts/probes/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* ticks, int m_iters, int o_iters, int roles, int variant) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1040];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 64 * 1040 / 4; i += 512) ((float*)lds)[i] = in[i & 4095];
    __syncthreads();
    // variant bit 2 / 3: the matrix wave idles after every MFMA (s_nop / s_sleep) instead of waiting in the issue stage
    // variant bit 0: the OTHER role runs at s_setprio 3 (the matrix role at 0); bit 1: the matrix role on the YOUNGER waves 4-7
    const bool mrole = (variant & 2) ? wid >= 4 : wid < 4;
    if (!mrole && (variant & 1)) __builtin_amdgcn_s_setprio(3);
    if (mrole ? !(roles & 1) : !(roles & 2)) return;
    float s = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (mrole) {
        f16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (_Float16)in[(tid * 8 + j + i * 64) & 4095];
                b[i][j] = (_Float16)in[(tid * 8 + j + i * 64 + 2048) & 4095];
            }
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        t0 = __builtin_readcyclecounter(); w0 = wall_clock64();
#pragma unroll 1
        for (int it = 0; it < m_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i * 2) & 3], acc[i], 0, 0, 0);
                if (variant & 4) {      // do not wait IN the issue stage for the pipe: 3 x 8 idle cycles of this wave after every MFMA
                    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
                }
                if (variant & 8) __builtin_amdgcn_s_sleep(0);
            }
        }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = in[(tid + i * 512) & 4095];
        const float c0 = in[7], c1 = in[9];
        t0 = __builtin_readcyclecounter(); w0 = wall_clock64();
        if constexpr (KIND == 0) {
#pragma unroll 1
            for (int it = 0; it < o_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], c0, c1);
            }
        } else if constexpr (KIND == 1) {
            f32x2 p[8];
            for (int i = 0; i < 8; ++i) p[i] = f32x2{v[i], v[(i + 1) & 7]};
            const f32x2 k0 = {c0, c0}, k1 = {c1, c1};
#pragma unroll 1
            for (int it = 0; it < o_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], k0, k1);
            }
            for (int i = 0; i < 8; ++i) v[i] = p[i][0] + p[i][1];
        } else if constexpr (KIND == 2) {
            const char* q = lds + (lane & 31) * 1040 + (lane >> 5) * 16;
            u32x4 x[8];
#pragma unroll 1
            for (int it = 0; it < o_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = *(const volatile u32x4*)(q + ((r * 8 + i) & 31) * 32);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += __uint_as_float(x[i][0]);
                }
            }
        } else if constexpr (KIND == 3) {
            unsigned packed = 0;
#pragma unroll 1
            for (int it = 0; it < o_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const _Float16 h = (_Float16)v[i];
                        const _Float16 l = (_Float16)(v[i] - (float)h);
                        packed += (unsigned)__builtin_bit_cast(unsigned short, h) + ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
                        v[i] = v[i] * c0;
                    }
            }
            v[0] += (float)packed;
        } else {
            const u32x4* g = (const u32x4*)in + (tid & 255);
            u32x4 x[8];
#pragma unroll 1
            for (int it = 0; it < o_iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_nontemporal_load(g + ((it * 8 + i) & 3) * 256);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += __uint_as_float(x[i][0]);
            }
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && (wid == 0 || wid == 4) && blockIdx.x == gridDim.x / 2) {
        ticks[(mrole ? 0 : 2)] = t1 - t0;
        ticks[(mrole ? 0 : 2) + 1] = w1 - w0;
    }
}

template <int KIND>
void run(const char* what, const float* din, float* dout, unsigned long long* dt, int m_iters, int o_iters, int variant) {
    double us[4][2] = {};
    for (int roles = 1; roles <= 3; ++roles) {
        hipMemset(dt, 0, 32);
        hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, din, dout, dt, m_iters, o_iters, roles, variant);
        hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, din, dout, dt, m_iters, o_iters, roles, variant);
        hipDeviceSynchronize();
        unsigned long long t[4];
        hipMemcpy(t, dt, 32, hipMemcpyDeviceToHost);
        us[roles][0] = t[1] * 0.01;
        us[roles][1] = t[3] * 0.01;
    }
    printf("%-22s mfma alone %7.1f us | other alone %7.1f us | together: mfma %7.1f  other %7.1f   (max %.1f, sum %.1f)\n", what, us[1][0], us[2][1],
           us[3][0], us[3][1], us[1][0] > us[2][1] ? us[1][0] : us[2][1], us[1][0] + us[2][1]);
}

int main() {
    float *din, *dout;
    unsigned long long* dt;
    hipMalloc(&din, 1 << 20);
    hipMalloc(&dout, 256 * 512 * 4);
    hipMalloc(&dt, 64);
    float h[4096];
    unsigned s = 12345u;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f + 0.25f; }
    for (int i = 0; i < (1 << 20) / 16384; ++i) hipMemcpy((char*)din + i * 16384, h, 16384, hipMemcpyHostToDevice);
    const int M = 4000;                     // 32 000 MFMAs per wave: ~1 M cycles at 32 cycles each
    const char* vn[4] = {"matrix role on waves 0-3 (older), no priorities", "... the other role at s_setprio 3", "matrix role on waves 4-7 (younger), no priorities",
                         "... the other role at s_setprio 3"};
    for (int variant : {0, 1, 2, 3, 4, 8}) {
        printf("-- %s\n", variant < 4 ? vn[variant] : (variant == 4 ? "matrix role on waves 0-3, every MFMA followed by 3 x s_nop 7" : "matrix role on waves 0-3, every MFMA followed by s_sleep 0"));
        run<0>("v_fma_f32", din, dout, dt, M, 8000, variant);
        run<1>("v_pk_fma_f32", din, dout, dt, M, 8000, variant);
        run<2>("ds_read_b128", din, dout, dt, M, 4000, variant);
        run<3>("fp16 hi/lo split", din, dout, dt, M, 2500, variant);
        run<4>("global_load_dwordx4", din, dout, dt, M, 8000, variant);
    }
    return 0;
}

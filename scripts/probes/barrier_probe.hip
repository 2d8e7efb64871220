// barrier_probe.hip -- what does a dependent step cost on MI355X?
//   (1) chain of N dependent empty kernels on one stream (kernel boundary = dispatch + cache writeback / invalidate)
//   (2) one persistent kernel of G workgroups doing N grid barriers: device-scope atomics + __threadfence()
//   (3) the same with a payload: every workgroup writes 16 KB, barrier, reads a neighbour's 16 KB (validates visibility)
// build: hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip ; run: ./barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: this workgroup's stores
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                   // acquire
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, int n) {
    for (int i = 0; i < n; ++i) grid_barrier(counter, (unsigned)(i + 1) * gridDim.x);
}

__global__ __launch_bounds__(256) void payload_kernel(unsigned* counter, int n, float* buf, int* bad) {
    const int G = gridDim.x, b = blockIdx.x;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        float4* mine = (float4*)(buf + (size_t)b * 4096);
        for (int k = threadIdx.x; k < 1024; k += 256) mine[k] = make_float4((float)(i + b), 1.f, 2.f, 3.f);
        grid_barrier(counter, (unsigned)(2 * i + 1) * G);
        const int nb = (b + 37) % G;
        const float4* theirs = (const float4*)(buf + (size_t)nb * 4096);
        for (int k = threadIdx.x; k < 1024; k += 256) {
            float4 v = theirs[k];
            if (v.x != (float)(i + nb)) atomicAdd(bad, 1);
            acc += v.y;
        }
        grid_barrier(counter, (unsigned)(2 * i + 2) * G);    // before the buffer is overwritten
    }
    if (acc == 12345.f) buf[0] = acc;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* counter; int* bad; float* buf;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&buf, (size_t)512 * 4096 * 4));
    const int N = 200;
    float ms;
    for (int grid : {256, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, (int*)nullptr);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("dependent empty kernels, grid %4d: %.2f us per launch\n", grid, ms * 1e3 / N);
    }
    for (int G : {64, 128, 256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, s, counter, N);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("persistent kernel, %3d workgroups: %.2f us per grid barrier\n", G, ms * 1e3 / N);
    }
    for (int G : {128, 256}) {
        CK(hipMemsetAsync(bad, 0, 4, s));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(payload_kernel, dim3(G), dim3(256), 0, s, counter, N, buf, bad);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("payload (16 KB write, barrier, 16 KB neighbour read, barrier), %3d workgroups: %.2f us per round, stale reads %d\n", G,
               ms * 1e3 / N, hb);
    }
    return 0;
}

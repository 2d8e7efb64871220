// ctxsw_probe.hip -- does wave state survive GPU sharing between processes?
//
// Background (profiles/r04_determinism.txt): two kernels of the training backward transiently produce the work of 1 - 3 waves
// wrong, with identical inputs, but ONLY when a second process runs heavy kernels on the same GPU.  When two processes share a
// GPU the hardware scheduler time-slices their queues and saves / restores running waves (registers, LDS, barrier state).  This
// probe holds known values in SGPRs, VGPRs and LDS across barriers and sleeps for milliseconds, exercises ds_bpermute,
// v_readlane, ballot + mbcnt and LDS broadcast reads meanwhile, and counts every value that comes back different from what
// pure arithmetic says it must be.  Run one instance alone (control) and two or three side by side:
//     hipcc --offload-arch=gfx950 -O2 -o ctxsw_probe ctxsw_probe.hip ; ./ctxsw_probe 15 & ./ctxsw_probe 15 & wait
// Output: per check, the number of mismatches over all launches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

enum { E_SGPR = 0, E_VGPR, E_LDS, E_BPERM, E_READLANE, E_BALLOT, E_LDSBCAST, E_COMPACT, E_N };

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) {
    unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
    return h;
}
#define OPAQUE_V(x) asm volatile("" : "+v"(x))
#define OPAQUE_S(x) asm volatile("" : "+s"(x))

constexpr int NS = 24, NV = 24, LDSW = 8192;     // 24 pinned SGPRs, 24 pinned VGPRs, 32 KB of LDS pattern (+ optional dynamic LDS)

__global__ __launch_bounds__(256) void probe_kernel(unsigned* __restrict__ err, int iters, unsigned seed, int sleep_k, int dyn_words) {
    __shared__ unsigned pat[LDSW];
    __shared__ unsigned list[512];
    __shared__ int wave_cnt[4];
    extern __shared__ unsigned dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned e[E_N] = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- state that must survive ----
    unsigned sg[NS], vg[NV];
    const unsigned wave_key = __builtin_amdgcn_readfirstlane(mix(seed, blockIdx.x * 4 + wid));
#pragma unroll
    for (int k = 0; k < NS; ++k) { sg[k] = __builtin_amdgcn_readfirstlane(mix(wave_key, k)); OPAQUE_S(sg[k]); }
#pragma unroll
    for (int k = 0; k < NV; ++k) { vg[k] = mix(wave_key ^ (unsigned)lane * 77u, k); OPAQUE_V(vg[k]); }
    for (int i = tid; i < LDSW; i += 256) pat[i] = mix(seed ^ blockIdx.x, i);
    for (int i = tid; i < dyn_words; i += 256) dyn[i] = mix(seed ^ blockIdx.x ^ 0x5555u, i);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        // (c) ds_bpermute: fetch lane src's vg[it % NV]-like value; the expectation is recomputed from scratch
        {
            const int src = (lane * 5 + it) & 63;
            unsigned mine = mix(wave_key ^ (unsigned)lane * 77u, 1000 + it);
            OPAQUE_V(mine);
            const unsigned got = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)mine);
            const unsigned want = mix(wave_key ^ (unsigned)src * 77u, 1000 + it);
            e[E_BPERM] += got != want;
        }
        // (d) v_readlane with a uniform lane
        {
            const int src = (it * 7 + wid) & 63;
            unsigned mine = mix(wave_key ^ (unsigned)lane * 131u, 2000 + it);
            OPAQUE_V(mine);
            const unsigned got = (unsigned)__builtin_amdgcn_readlane((int)mine, __builtin_amdgcn_readfirstlane(src));
            const unsigned want = mix(wave_key ^ (unsigned)src * 131u, 2000 + it);
            e[E_READLANE] += got != want;
        }
        // (f) ballot + mbcnt: predicate = bit b of the lane number -> mask and prefix count are known in closed form
        const int b = it % 6;
        const bool keep = (lane >> b) & 1;
        const unsigned long long m = __ballot(keep);
        {
            unsigned long long want = 0;
            // lanes with bit b set: pattern of 2^b ones following 2^b zeros
            const unsigned long long unit = ((1ull << (1 << b)) - 1ull) << (1 << b);
            for (int s = 0; s < 64; s += 2 << b) want |= unit << s;
            e[E_BALLOT] += m != want;
        }
        if (lane == 0) wave_cnt[wid] = __popcll(m);
        // (g) LDS broadcast read of a word somebody else wrote long ago
        {
            const int idx = (it * 97 + blockIdx.x) & (LDSW - 1);
            e[E_LDSBCAST] += pat[idx] != mix(seed ^ blockIdx.x, idx);
        }
        __syncthreads();                      // ---- the masks m (SGPR pair) and all pinned state live across this barrier ----
        if (sleep_k > 0) for (int s = 0; s < sleep_k; ++s) __builtin_amdgcn_s_sleep(127);
        // (h) the scatter kernel's ordered compaction: kept threads write their id at their rank; everybody checks the list
        {
            int off = 0, total = 0;
            for (int w = 0; w < 4; ++w) { if (w < wid) off += wave_cnt[w]; total += wave_cnt[w]; }
            if (keep) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)tid ^ (unsigned)it;
            __syncthreads();
            e[E_COMPACT] += total != 128;
            if (tid < 128) {
                // rank r (0..127) is the r-th thread with bit b set: insert a 1 at bit b of r
                const int r = tid, t = ((r >> b) << (b + 1)) | (1 << b) | (r & ((1 << b) - 1));
                e[E_COMPACT] += list[r] != ((unsigned)t ^ (unsigned)it);
            }
            __syncthreads();
        }
        // (a) the pinned SGPRs, every 16th round (and after the loop)
        if ((it & 15) == 15) {
#pragma unroll
            for (int k = 0; k < NS; ++k) { OPAQUE_S(sg[k]); e[E_SGPR] += sg[k] != mix(wave_key, k); }
#pragma unroll
            for (int k = 0; k < NV; ++k) { OPAQUE_V(vg[k]); e[E_VGPR] += vg[k] != mix(wave_key ^ (unsigned)lane * 77u, k); }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) { OPAQUE_S(sg[k]); e[E_SGPR] += sg[k] != mix(wave_key, k); }
#pragma unroll
    for (int k = 0; k < NV; ++k) { OPAQUE_V(vg[k]); e[E_VGPR] += vg[k] != mix(wave_key ^ (unsigned)lane * 77u, k); }
    for (int i = tid; i < LDSW; i += 256) e[E_LDS] += pat[i] != mix(seed ^ blockIdx.x, i);
    for (int i = tid; i < dyn_words; i += 256) e[E_LDS] += dyn[i] != mix(seed ^ blockIdx.x ^ 0x5555u, i);
#pragma unroll
    for (int k = 0; k < E_N; ++k) if (e[k]) atomicAdd(err + k, e[k]);
}

// something heavy for the OTHER queue slices: streams 256 MB through the chip
__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ o, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i]; v.x = v.x * 1.0001f + v.y; v.y += v.z; o[i] = v;
    }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 400;
    const int sleep_k = argc > 3 ? atoi(argv[3]) : 2;
    const int dyn_kb = argc > 4 ? atoi(argv[4]) : 40;        // dynamic LDS (KB) on top of the 34 KB static: > 64 KB in total by default
    const int heavy = argc > 5 ? atoi(argv[5]) : 1;          // also run the streaming kernel on a second stream
    unsigned* err;
    CK(hipMalloc(&err, E_N * 4));
    CK(hipMemset(err, 0, E_N * 4));
    float4 *a, *o;
    const size_t n = (size_t)16 << 20;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&o, n * 16));
    CK(hipMemset(a, 0, n * 16));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    const int dyn_words = dyn_kb * 256;
    CK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_words * 4));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 8; ++k) {
            hipLaunchKernelGGL(probe_kernel, dim3(1024), dim3(256), dyn_words * 4, s0, err, iters, (unsigned)(launches * 8 + k + 1), sleep_k, dyn_words);
            if (heavy) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s1, a, o, n);
        }
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        ++launches;
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned h[E_N];
    CK(hipMemcpy(h, err, E_N * 4, hipMemcpyDeviceToHost));
    printf("pid %d: %ld x 8 launches in %.1f s (%.2f ms per probe launch); mismatches: sgpr %u vgpr %u lds %u bpermute %u readlane %u ballot %u lds_bcast %u compaction %u\n",
           (int)getpid(), launches, el, el * 1e3 / (launches * 8), h[E_SGPR], h[E_VGPR], h[E_LDS], h[E_BPERM], h[E_READLANE], h[E_BALLOT], h[E_LDSBCAST], h[E_COMPACT]);
    return 0;
}

// probe: semantics of ds_read_b64_tr_b16 on gfx950 -- which LDS elements does lane l receive, given per-lane addresses?
// Every 16-bit LDS element holds its own index; each lane reads with address = base + f(lane) and we print what came back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem_addr;   // element index this lane points at (must be 4-element = 8-byte aligned)
    if (mode == 0) elem_addr = 4 * l;                              // lane-linear: lane l -> elements 4l..4l+3
    else if (mode == 1) elem_addr = (l & 15) * 4 + (l >> 4) * 64;  // same, explicit per 16-lane group
    else elem_addr = ((l & 15) >> 2) * 64 + ((l & 3) * 4) + (l >> 4) * 16;   // rows of 64 elements: group g = column block
    unsigned addr = (unsigned)(size_t)(lds + elem_addr);   // LDS byte address (low 32 bits of the generic pointer)
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main(int argc, char** argv) {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); if (l % 4 == 3) printf("\n"); }
    }
    return 0;
}

// MODE.FP16_OVFL (bit 23): with it set, a float -> half conversion that overflows gives +-65504 instead of +-inf?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int set) {
    if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const float v = in[threadIdx.x];
    const _Float16 c = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)c);
    unsigned short cs, ls; __builtin_memcpy(&cs, &c, 2); __builtin_memcpy(&ls, &l, 2);
    out[threadIdx.x * 2] = cs; out[threadIdx.x * 2 + 1] = ls;
}
int main() {
    float h[4] = {1.0e6f, -7.0e4f, 65504.f, 0.1f};
    float* d; unsigned* o; (void)hipMalloc(&d, 16); (void)hipMalloc(&o, 32); (void)hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, set);
        unsigned r[8]; (void)hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
        for (int i = 0; i < 4; ++i) printf("FP16_OVFL=%d  %.6g -> hi 0x%04x lo 0x%04x\n", set, h[i], r[2 * i], r[2 * i + 1]);
    }
    return 0;
}

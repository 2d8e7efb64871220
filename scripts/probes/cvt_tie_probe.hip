// how do v_cvt_f16_f32 and v_cvt_pk_f16_f32 round an exact tie?  (0.108978271f lies exactly between two halves)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* in, unsigned* out) {
    const float v = in[threadIdx.x];
    unsigned a, b;
    asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(a) : "v"(v));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(b) : "v"(v));
    const _Float16 c = (_Float16)v;
    unsigned short cs; __builtin_memcpy(&cs, &c, 2);
    out[threadIdx.x * 3 + 0] = a & 0xffffu; out[threadIdx.x * 3 + 1] = b & 0xffffu; out[threadIdx.x * 3 + 2] = cs;
}
int main() {
    float h[4] = {0.108978271f, -0.108978271f, 0.10891724f /* tie the other way */, 1.00048828125f};
    float* d; unsigned* o; hipMalloc(&d, 16); hipMalloc(&o, 48); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o);
    unsigned r[12]; hipMemcpy(r, o, 48, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) printf("%.9g: v_cvt_f16_f32 0x%04x  v_cvt_pk_f16_f32 0x%04x  (_Float16) 0x%04x\n", h[i], r[3 * i], r[3 * i + 1], r[3 * i + 2]);
    return 0;
}

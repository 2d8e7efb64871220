// does v_mfma_f32_32x32x16_f16 keep fp16 subnormal operands?  A[i][k] = a (subnormal), B[k][j] = b: D = 16 * a * b
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float* out) {
    h8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)a; bv[i] = (_Float16)b; }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)av[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    for (float a : {1.0f, 3e-5f, 1e-6f, 6e-8f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 256.0f, d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g (as fp16 %g): mfma -> %g, expected %g\n", a, h[1], h[0], 16.0 * h[1] * 256.0);
    }
    return 0;
}

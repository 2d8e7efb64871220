// so3.hip -- projection of 3x3 matrices onto SO(3) and its exact backward, one thread per matrix.
//
// Replaces ReconFitterBase.project_so3 (recon/recon_fit_base.py:168-188): R = U diag(1,1,det(UV^T)) V^T from
// torch.svd + det + two matmuls (a generic batched LAPACK-style SVD per fit step in the reference).
// Forward: cyclic Jacobi eigen-decomposition of M^T M in fp64 -> V, sigma; U = M V / sigma.
// Backward: M = R P with P = V diag(d_i sigma_i) V^T symmetric, so with H = D U^T G V,
//   K_ij = (H_ij - H_ji) / (d_i sigma_i + d_j sigma_j),   dL/dM = U D K V^T
// which is the derivative of the projection itself (it does not contain the 1/(sigma_i^2 - sigma_j^2)
// terms of a generic SVD backward, so it stays finite for repeated singular values).
#include "common.h"
#include "svd3.h"

namespace {

__global__ void so3_fwd_kernel(const float* __restrict__ M, int B, float* __restrict__ R, double* __restrict__ aux) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double m[9];
    for (int e = 0; e < 9; ++e) m[e] = (double)M[(size_t)b * 9 + e];
    Svd3 s;
    svd3(m, s);
    const double d[3] = {1.0, 1.0, s.det};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a += s.U[r * 3 + k] * d[k] * s.V[c * 3 + k];
            R[(size_t)b * 9 + r * 3 + c] = (float)a;
        }
    if (aux) {
        double* o = aux + (size_t)b * 22;
        for (int e = 0; e < 9; ++e) { o[e] = s.U[e]; o[9 + e] = s.V[e]; }
        for (int e = 0; e < 3; ++e) o[18 + e] = s.s[e];
        o[21] = s.det;
    }
}

__global__ void so3_bwd_kernel(const double* __restrict__ aux, const float* __restrict__ G, int B, float* __restrict__ dM) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* a = aux + (size_t)b * 22;
    const double *U = a, *V = a + 9, *sg = a + 18;
    const double d[3] = {1.0, 1.0, a[21]};
    double g[9], H[9], K[9], T[9];
    for (int e = 0; e < 9; ++e) g[e] = (double)G[(size_t)b * 9 + e];
    // H = D U^T G V
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double x = 0.0;
            for (int k = 0; k < 3; ++k) x += U[k * 3 + i] * g[k * 3 + j];
            T[i * 3 + j] = x;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double x = 0.0;
            for (int k = 0; k < 3; ++k) x += T[i * 3 + k] * V[k * 3 + j];
            H[i * 3 + j] = d[i] * x;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double den = d[i] * sg[i] + d[j] * sg[j];
            K[i * 3 + j] = (i == j || fabs(den) < 1e-300) ? 0.0 : (H[i * 3 + j] - H[j * 3 + i]) / den;
        }
    // dM = U D K V^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double x = 0.0;
            for (int k = 0; k < 3; ++k) x += d[i] * K[i * 3 + k] * V[j * 3 + k];
            T[i * 3 + j] = x;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double x = 0.0;
            for (int k = 0; k < 3; ++k) x += U[i * 3 + k] * T[k * 3 + j];
            dM[(size_t)b * 9 + i * 3 + j] = (float)x;
        }
}

}  // namespace

extern "C" {

size_t chore_so3_aux_bytes(int B) { return (size_t)B * 22 * sizeof(double); }

int chore_so3_project_fwd(chore_handle* h, const float* M, int B, float* R, void* aux, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!M || !R || B < 1) CHORE_FAIL(h, CHORE_EINVAL, "chore_so3_project_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(so3_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, M, B, R, (double*)aux);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_so3_project_bwd(chore_handle* h, const void* aux, const float* G, int B, float* dM, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!aux || !G || !dM || B < 1) CHORE_FAIL(h, CHORE_EINVAL, "chore_so3_project_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(so3_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const double*)aux, G, B, dM);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // extern "C"

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

namespace {

struct Svd3 {
    double U[9], V[9], s[3], det;
};

__device__ void svd3(const double* M, Svd3& o) {
    double S[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[i * 3 + j] = M[0 * 3 + i] * M[0 * 3 + j] + M[1 * 3 + i] * M[1 * 3 + j] + M[2 * 3 + i] * M[2 * 3 + j];
    for (int sweep = 0; sweep < 12; ++sweep) {
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = S[p * 3 + q];
                if (fabs(apq) < 1e-300) continue;
                const double app = S[p * 3 + p], aqq = S[q * 3 + q];
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < 3; ++k) {   // S <- S J
                    const double skp = S[k * 3 + p], skq = S[k * 3 + q];
                    S[k * 3 + p] = c * skp - s * skq;
                    S[k * 3 + q] = s * skp + c * skq;
                }
                for (int k = 0; k < 3; ++k) {   // S <- J^T S
                    const double spk = S[p * 3 + k], sqk = S[q * 3 + k];
                    S[p * 3 + k] = c * spk - s * sqk;
                    S[q * 3 + k] = s * spk + c * sqk;
                }
                for (int k = 0; k < 3; ++k) {   // V <- V J
                    const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};   // sort eigenvalues descending
    double ev[3] = {S[0], S[4], S[8]};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (ev[idx[b]] > ev[idx[a]]) { const int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
    for (int j = 0; j < 3; ++j) {
        o.s[j] = sqrt(fmax(ev[idx[j]], 0.0));
        for (int k = 0; k < 3; ++k) o.V[k * 3 + j] = V[k * 3 + idx[j]];
    }
    // U columns: M v_j / sigma_j; a vanishing sigma_3 column is completed with the cross product
    for (int j = 0; j < 3; ++j) {
        double u[3];
        for (int k = 0; k < 3; ++k) u[k] = M[k * 3] * o.V[0 * 3 + j] + M[k * 3 + 1] * o.V[1 * 3 + j] + M[k * 3 + 2] * o.V[2 * 3 + j];
        const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (j == 2 && n < 1e-12 * fmax(o.s[0], 1e-300)) {
            u[0] = o.U[1 * 3 + 0] * o.U[2 * 3 + 1] - o.U[2 * 3 + 0] * o.U[1 * 3 + 1];
            u[1] = o.U[2 * 3 + 0] * o.U[0 * 3 + 1] - o.U[0 * 3 + 0] * o.U[2 * 3 + 1];
            u[2] = o.U[0 * 3 + 0] * o.U[1 * 3 + 1] - o.U[1 * 3 + 0] * o.U[0 * 3 + 1];
            for (int k = 0; k < 3; ++k) o.U[k * 3 + j] = u[k];
        } else {
            for (int k = 0; k < 3; ++k) o.U[k * 3 + j] = u[k] / fmax(n, 1e-300);
        }
    }
    auto det3 = [](const double* A) {
        return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    };
    o.det = det3(o.U) * det3(o.V);   // det(U V^T)
}

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
    if (!h) return CHORE_EINVAL;
    if (!M || !R || B < 1) CHORE_FAIL(h, CHORE_EINVAL, "chore_so3_project_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(so3_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, M, B, R, (double*)aux);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_so3_project_bwd(chore_handle* h, const void* aux, const float* G, int B, float* dM, chore_stream_t stream) {
    if (!h) return CHORE_EINVAL;
    if (!aux || !G || !dM || B < 1) CHORE_FAIL(h, CHORE_EINVAL, "chore_so3_project_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(so3_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const double*)aux, G, B, dM);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // extern "C"

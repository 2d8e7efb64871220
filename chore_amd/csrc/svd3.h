// svd3.h -- 3x3 singular value decomposition in fp64 (one thread): cyclic Jacobi on M^T M -> V, sigma; U = M V / sigma.
// Shared by the SO(3) projection of the fit (so3.hip) and the Procrustes alignment of the evaluation (eval_metrics.hip).
#pragma once
#include <hip/hip_runtime.h>

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

}  // namespace

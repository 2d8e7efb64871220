// eval_metrics.hip -- the evaluation metrics of the reference on the device, fp64, gfx950.
//
//   chamfer_distance          /root/reference/recon/eval/chamfer_distance.py:10-52  (sklearn kd-tree nearest neighbours,
//                             float64, mean of the EUCLIDEAN nearest-neighbour distances, both directions added)
//   compute_transform         /root/reference/recon/eval/pose_utils.py:145-180      (similarity Procrustes: means, var1,
//                             K = X1 X2^T, SVD, R = V Z U^T with Z fixing det = +1, scale = tr(R K) / var1,
//                             t = mu2 - scale R mu1)
//   (compute_similarity_transform :103-143 applies it: scale R S1 + t)
// Sizes are 10^3..10^5 points, so nearest neighbours are an exhaustive tiled search (10 k x 10 k pairs = 10^8 fp64
// distance evaluations, tens of microseconds) and every reduction is a fixed-order tree in one workgroup: results
// are reproducible and agree with numpy / sklearn to fp64 round-off.
#include "common.h"
#include "svd3.h"

namespace {

constexpr int NN_TILE = 1024;

// d2min[i] = min_j |q_i - r_j|^2
__global__ __launch_bounds__(256) void nn_min_kernel(const double* __restrict__ q, int Nq, const double* __restrict__ r, int Nr,
                                                     double* __restrict__ d2min) {
    __shared__ double tile[NN_TILE * 3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ic = i < Nq ? i : Nq - 1;
    const double x = q[(size_t)ic * 3], y = q[(size_t)ic * 3 + 1], z = q[(size_t)ic * 3 + 2];
    double best = 1e300;
    for (int base = 0; base < Nr; base += NN_TILE) {
        const int n = min(NN_TILE, Nr - base);
        __syncthreads();
        for (int k = threadIdx.x; k < n * 3; k += 256) tile[k] = r[(size_t)base * 3 + k];
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const double dx = x - tile[j * 3], dy = y - tile[j * 3 + 1], dz = z - tile[j * 3 + 2];
            const double d = dx * dx + dy * dy + dz * dz;
            best = d < best ? d : best;
        }
    }
    if (i < Nq) d2min[i] = best;
}

// fixed-order sum of f(v[k]) over one workgroup: strided partials, then a tree
template <typename F>
__device__ double block_sum(int n, F f) {
    __shared__ double red[256];
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) s += f(k);
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}

__global__ __launch_bounds__(256) void mean_sqrt_kernel(const double* __restrict__ d2, int n, double* __restrict__ out) {
    const double s = block_sum(n, [&](int k) { return sqrt(d2[k]); });
    if (threadIdx.x == 0) *out = s / (double)n;
}

// out[0..8] = R (row-major), out[9..11] = t, out[12] = scale  such that  S2 ~ scale R S1 + t
__global__ __launch_bounds__(256) void procrustes_kernel(const double* __restrict__ S1, const double* __restrict__ S2, int N,
                                                         double* __restrict__ out) {
    double mu1[3], mu2[3];
    for (int d = 0; d < 3; ++d) {
        mu1[d] = block_sum(N, [&](int k) { return S1[(size_t)k * 3 + d]; }) / (double)N;
        mu2[d] = block_sum(N, [&](int k) { return S2[(size_t)k * 3 + d]; }) / (double)N;
    }
    double var1 = 0.0;
    for (int d = 0; d < 3; ++d) var1 += block_sum(N, [&](int k) { const double v = S1[(size_t)k * 3 + d] - mu1[d]; return v * v; });
    double K[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            K[a * 3 + b] = block_sum(N, [&](int k) { return (S1[(size_t)k * 3 + a] - mu1[a]) * (S2[(size_t)k * 3 + b] - mu2[b]); });
    if (threadIdx.x) return;
    Svd3 s;
    svd3(K, s);                                   // K = U diag(s) V^T
    const double z[3] = {1.0, 1.0, s.det >= 0.0 ? 1.0 : -1.0};     // sign(det(U V^T))
    double R[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += s.V[a * 3 + k] * z[k] * s.U[b * 3 + k];   // R = V Z U^T
            R[a * 3 + b] = v;
        }
    double tr = 0.0;                              // trace(R K)
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < 3; ++k) tr += R[a * 3 + k] * K[k * 3 + a];
    const double scale = tr / var1;
    for (int e = 0; e < 9; ++e) out[e] = R[e];
    for (int a = 0; a < 3; ++a) out[9 + a] = mu2[a] - scale * (R[a * 3] * mu1[0] + R[a * 3 + 1] * mu1[1] + R[a * 3 + 2] * mu1[2]);
    out[12] = scale;
}

__global__ void apply_similarity_kernel(const double* __restrict__ p, int N, const double* __restrict__ prm, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double x = p[(size_t)i * 3], y = p[(size_t)i * 3 + 1], z = p[(size_t)i * 3 + 2], s = prm[12];
    for (int a = 0; a < 3; ++a) out[(size_t)i * 3 + a] = s * (prm[a * 3] * x + prm[a * 3 + 1] * y + prm[a * 3 + 2] * z) + prm[9 + a];
}

}  // namespace

extern "C" {

size_t chore_eval_chamfer_workspace_bytes(int Nx, int Ny) { return (size_t)(Nx > Ny ? Nx : Ny) * sizeof(double); }

// out[0] = mean_i min_j |x_i - y_j| ('x_to_y'), out[1] = mean_j min_i |x_i - y_j| ('y_to_x'); 'bi' is their sum
int chore_eval_chamfer(chore_handle* h, const double* x, int Nx, const double* y, int Ny, double* out, void* workspace,
                       chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!x || !y || !out || !workspace || Nx <= 0 || Ny <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_eval_chamfer: bad argument");
    hipStream_t s = (hipStream_t)stream;
    double* d2 = (double*)workspace;
    hipLaunchKernelGGL(nn_min_kernel, dim3((Nx + 255) / 256), dim3(256), 0, s, x, Nx, y, Ny, d2);
    hipLaunchKernelGGL(mean_sqrt_kernel, dim3(1), dim3(256), 0, s, d2, Nx, out);
    hipLaunchKernelGGL(nn_min_kernel, dim3((Ny + 255) / 256), dim3(256), 0, s, y, Ny, x, Nx, d2);
    hipLaunchKernelGGL(mean_sqrt_kernel, dim3(1), dim3(256), 0, s, d2, Ny, out + 1);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// params[13] = R (9, row-major), t (3), scale: S2 ~ scale R S1 + t   (pose_utils.py compute_transform)
int chore_eval_procrustes(chore_handle* h, const double* S1, const double* S2, int N, double* params, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!S1 || !S2 || !params || N <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_eval_procrustes: bad argument");
    hipLaunchKernelGGL(procrustes_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, S1, S2, N, params);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

int chore_eval_apply_similarity(chore_handle* h, const double* pts, int N, const double* params, double* out, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!pts || !params || !out || N <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_eval_apply_similarity: bad argument");
    hipLaunchKernelGGL(apply_similarity_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, N, params, out);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

}  // extern "C"

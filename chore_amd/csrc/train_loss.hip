// train_loss.hip -- the training loss of one stack and its gradients in one pass (gfx950).
//
// CHORE.get_errors (/root/reference/model/chore.py:192-237) sums six terms per stack: clamped-L1 of the two distance
// fields, cross entropy of the 14 part logits, and masked mean squared errors of the PCA axes and the two centre
// predictions.  Written with tensor ops that is ~25 elementwise / reduction kernels forward and ~35 backward per stack
// over (B, C, N) tensors of a few MB -- launch-bound, and the device idles between them while the host issues the next.
// One thread per point computes all six terms and the four gradient tensors; the sums are exact (128-bit fixed-point
// accumulators, enc_common.h), so the result does not depend on the order of the workgroups.
//   loss_h, loss_o = w0,1 / B      * sum_{b,n} | min(df_pred, md) - min(df_gt, md) |
//   loss_parts     = w2   / B      * sum_{b,n} ( logsumexp(parts[b,:,n]) - parts[b,gt,n] )
//   loss_pca       = w3 / (9 B N)  * sum m_o (pca - pca_gt)^2        m_o = [df_o < 0.05]
//   loss_obj       = w4 / (3 B N)  * sum mbar_o (centers[3:6] - obj_center)^2    mbar_o[n] = mean over the batch of m_o[:, n]
//                    (the reference multiplies the (B,3,N) errors by the (B,1,1,N) mask of the PCA term, which broadcasts
//                    to (B,B,3,N): every image's errors meet every image's mask -- model/chore.py:213-219; kept as is)
//   loss_smpl      = w5 / (3 B N)  * sum m_h (centers[0:3] - body_center)^2      m_h = [df_h < 0.05]
// Gradients follow torch's conventions: clamp passes the gradient where the input is <= max, L1 uses sign(0) = 0.
#include "enc_common.h"

namespace {

constexpr size_t LOSS_HI_CELLS = 16;   // cells between a low limb and its high limb (two tables, enc_common.h)

struct LossArgs {
    const float *df, *pca, *parts, *centers;                       // predictions (B,2,N) (B,9,N) (B,14,N) (B,6,N)
    const float *df_h, *df_o, *pca_gt, *body_center, *obj_center;   // (B,N) (B,N) (B,9,N) (B,3) (B,3,N)
    const long long* parts_gt;                                      // (B,N)
    float *g_df, *g_pca, *g_parts, *g_centers;
    int B, N;
    float max_dist, scale;
    float w[6];
    StatCell* acc;                                                  // [6] low limbs, the high limbs LOSS_HI_CELLS cells later; zeroed
    float* losses;                                                  // [6] h, o, parts, pca, smpl, obj (scaled), then [6] = their sum
    int accumulate;
};

__global__ __launch_bounds__(256) void train_loss_kernel(LossArgs a) {
    __shared__ float red[4][6];
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x, N = a.N;
    float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const size_t bn = (size_t)b * N + n;
        const float dh = a.df_h[bn], dob = a.df_o[bn];
        const float fB = 1.0f / (float)a.B, md = a.max_dist;
        // distance fields
        {
            const float p0 = a.df[((size_t)b * 2 + 0) * N + n], p1 = a.df[((size_t)b * 2 + 1) * N + n];
            const float d0 = fminf(p0, md) - fminf(dh, md), d1 = fminf(p1, md) - fminf(dob, md);
            t[0] = fabsf(d0); t[1] = fabsf(d1);
            const float s0 = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f), s1 = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
            a.g_df[((size_t)b * 2 + 0) * N + n] = p0 <= md ? a.scale * a.w[0] * fB * s0 : 0.f;
            a.g_df[((size_t)b * 2 + 1) * N + n] = p1 <= md ? a.scale * a.w[1] * fB * s1 : 0.f;
        }
        // part logits: cross entropy
        {
            float x[14], mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 14; ++c) { x[c] = a.parts[((size_t)b * 14 + c) * N + n]; mx = fmaxf(mx, x[c]); }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < 14; ++c) se += expf(x[c] - mx);
            const float lse = logf(se);
            const int gt = (int)a.parts_gt[bn];
            const float k = a.scale * a.w[2] * fB;
            float xg = 0.f;
#pragma unroll
            for (int c = 0; c < 14; ++c) {
                const float ls = (x[c] - mx) - lse;                       // log_softmax
                if (c == gt) xg = ls;
                a.g_parts[((size_t)b * 14 + c) * N + n] = k * (expf(ls) - (c == gt ? 1.f : 0.f));
            }
            t[2] = -xg;
        }
        const float mo = dob < 0.05f ? 1.f : 0.f, mh = dh < 0.05f ? 1.f : 0.f;
        float mbar = 0.f;                                       // the object-centre term's mask, see the header
        for (int bb = 0; bb < a.B; ++bb) mbar += a.df_o[(size_t)bb * N + n] < 0.05f ? 1.f : 0.f;
        mbar *= fB;
        // PCA axes
        {
            const float k = a.scale * a.w[3] * 2.0f / (9.0f * (float)a.B * (float)N);
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const size_t i = ((size_t)b * 9 + c) * N + n;
                const float d = a.pca[i] - a.pca_gt[i];
                t[3] += d * d * mo;
                a.g_pca[i] = k * d * mo;
            }
        }
        // centres: [0:3] body (one vector per image), [3:6] object (per point)
        {
            const float k = 2.0f / (3.0f * (float)a.B * (float)N);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t i = ((size_t)b * 6 + c) * N + n, j = ((size_t)b * 6 + 3 + c) * N + n;
                const float ds = a.centers[i] - a.body_center[b * 3 + c];
                const float dq = a.centers[j] - a.obj_center[((size_t)b * 3 + c) * N + n];
                t[4] += ds * ds * mh;
                t[5] += dq * dq * mbar;
                a.g_centers[i] = a.scale * a.w[5] * k * ds * mh;
                a.g_centers[j] = a.scale * a.w[4] * k * dq * mbar;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) t[k] += __shfl_xor(t[k], o, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k) red[wid][k] = t[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        stat_add(&a.acc[k], LOSS_HI_CELLS, ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k]);
    }
}

__global__ void train_loss_finish_kernel(LossArgs a) {
    if (threadIdx.x != 0) return;
    const double B = a.B, N = a.N;
    // accumulator order: h, o, parts, pca, smpl, obj
    const double norm[6] = {a.w[0] / B, a.w[1] / B, a.w[2] / B, a.w[3] / (9.0 * B * N), a.w[5] / (3.0 * B * N), a.w[4] / (3.0 * B * N)};
    float total = 0.f;
    for (int k = 0; k < 6; ++k) {
        const float v = (float)(stat_read(&a.acc[k], LOSS_HI_CELLS) * norm[k]) * a.scale;
        a.losses[k] = a.accumulate ? a.losses[k] + v : v;
        total += v;
    }
    a.losses[6] = a.accumulate ? a.losses[6] + total : total;
}

}  // namespace

extern "C" {

size_t chore_train_loss_workspace_bytes(void) { return (LOSS_HI_CELLS + 6) * sizeof(StatCell) + 160; }

// losses (device, 7 floats): the six terms in the reference's order h, o, parts, pca, smpl, obj and their sum, all times
// `scale` (1 / number of stacks); accumulate != 0 adds to what is there (the stacks of one step).  g_*: gradients of
// losses[6]'s increment with respect to the four predictions.  weights: host array of the reference's six loss weights
// (df_h, df_o, parts, pca, obj, smpl -- model/chore.py's self.loss_weights order).
int chore_train_loss(chore_handle* h, const float* df, const float* pca, const float* parts, const float* centers,
                     const float* df_h, const float* df_o, const int64_t* parts_gt, const float* pca_gt, const float* body_center,
                     const float* obj_center, int B, int N, float max_dist, const float* weights, float scale, float* g_df,
                     float* g_pca, float* g_parts, float* g_centers, float* losses, int accumulate, void* workspace,
                     chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!df || !pca || !parts || !centers || !df_h || !df_o || !parts_gt || !pca_gt || !body_center || !obj_center || !weights ||
        !g_df || !g_pca || !g_parts || !g_centers || !losses || !workspace)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_train_loss: null argument");
    if (B <= 0 || N <= 0 || B > 65535) CHORE_FAIL(h, CHORE_EINVAL, "chore_train_loss: bad shape");
    hipStream_t s = (hipStream_t)stream;
    LossArgs a;
    a.df = df; a.pca = pca; a.parts = parts; a.centers = centers;
    a.df_h = df_h; a.df_o = df_o; a.pca_gt = pca_gt; a.body_center = body_center; a.obj_center = obj_center;
    a.parts_gt = (const long long*)parts_gt;
    a.g_df = g_df; a.g_pca = g_pca; a.g_parts = g_parts; a.g_centers = g_centers;
    a.B = B; a.N = N; a.max_dist = max_dist; a.scale = scale;
    for (int k = 0; k < 6; ++k) a.w[k] = weights[k];
    a.acc = (StatCell*)workspace; a.losses = losses; a.accumulate = accumulate;
    CHORE_HIP_CHECK(h, hipMemsetAsync(workspace, 0, (LOSS_HI_CELLS + 6) * sizeof(StatCell), s));
    hipLaunchKernelGGL(train_loss_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(train_loss_finish_kernel, dim3(1), dim3(64), 0, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // extern "C"

// fit_step.hip -- the optimiser update and the early-stop bookkeeping of one inner fitting step, two launches.
//
// An inner step of recon_fit_behave.py:143-160, 270-287 ends with Adam on a handful of tiny tensors (pose 72, betas 10,
// translation 3 per frame; object rotation / translation / scale) and the stop rule  |prev - loss| / prev < prev * tol.
// Written with tensor ops that is ~35 launches per step (multi-tensor Adam with device-side bias corrections, clones of
// the parameters so that a latched stop can undo the update, the rule itself) in a step of ~190; here it is one launch
// for the update of all tensors and one single-thread launch for the rule and the step counter.
// Adam as torch.optim.Adam(capturable=True) evaluates it (torch/optim/adam.py, _multi_tensor_adam):
//     m += (g - m) (1 - b1);  v = b2 v + (1 - b2) g g;  step_size = -lr / (1 - b1^t);
//     p += m / ( sqrt(v) / (sqrt(1 - b2^t) step_size) + eps / step_size )
// `stop` (latched by an earlier step): the moments and the counter still advance, the parameters stay (the reference
// has returned by then; everything after the latch is a no-op for the result).
#include "common.h"

namespace {

constexpr int FS_MAXT = 16;
struct AdamTensors {
    float* p[FS_MAXT]; float* g[FS_MAXT]; const float* gn[FS_MAXT]; float* m[FS_MAXT]; float* v[FS_MAXT];
    int n[FS_MAXT];
    int cols[FS_MAXT], pstride[FS_MAXT];      // the PARAMETER as rows of `cols` elements `pstride` apart (a column slice of a
    int nt;                                   // wider tensor); gradients and moments are dense.  cols == n: dense
};

__global__ void fit_adam_kernel(AdamTensors t, const float* step, float lr, float b1, float b2, float eps, const unsigned char* stop,
                                int step_counted) {
    const int k = blockIdx.y;
    const bool frozen = *stop != 0;
    const float s = step_counted ? *step : *step + 1.0f;
    const float bc1 = 1.0f - powf(b1, s), bc2 = 1.0f - powf(b2, s);
    const float step_size = -(lr / bc1);
    const float bc2s = sqrtf(bc2);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < t.n[k]; i += gridDim.x * blockDim.x) {
        float g = t.g[k][i];
        if (t.gn[k]) {                       // this step's gradient: .grad += new (what AccumulateGrad's add launch did)
            g = g + t.gn[k][i];
            t.g[k][i] = g;
        }
        if (!t.p[k]) continue;               // a leaf that only accumulates (stepped by a later phase's optimiser)
        float m = t.m[k][i], v = t.v[k][i];
        m = m + (g - m) * (1.0f - b1);
        v = v * b2 + (1.0f - b2) * g * g;
        t.m[k][i] = m; t.v[k][i] = v;
        if (!frozen) {
            const float denom = sqrtf(v) / (bc2s * step_size) + eps / step_size;
            const int c = t.cols[k];
            t.p[k][c == t.n[k] ? i : (i / c) * t.pstride[k] + i % c] += m / denom;
        }
    }
}

__global__ void fit_stop_rule_kernel(const float* loss, float* prev, unsigned char* stop, const unsigned char* armed, float tol,
                                     float* loss_out, float* step) {
    if (threadIdx.x || blockIdx.x) return;
    const bool frozen = *stop != 0;
    const float lv = *loss, pv = *prev;
    const bool hit = fabsf(pv - lv) / pv < pv * tol;
    if (hit && *armed) *stop = 1;
    if (!frozen) *prev = lv;
    *loss_out = lv;
    if (step) *step += 1.0f;
}

constexpr int FS_MAXL = 16;
struct LossTerms { const float* l[FS_MAXL]; float c[FS_MAXL]; int n; };

__global__ void fit_weighted_sum_kernel(LossTerms t, const float* denom, float* out) {
    if (threadIdx.x || blockIdx.x) return;
    const float d = *denom;
    float s = 0.f;
    for (int k = 0; k < t.n; ++k) s += t.c[k] * *t.l[k] / d;      // the reference's c * L / (1 + it), summed in dict order
    *out = s;
}
__global__ void fit_weighted_sum_bwd_kernel(LossTerms t, const float* denom, const float* g, float* grads) {
    const int k = threadIdx.x;
    if (k < t.n) grads[k] = *g * t.c[k] / *denom;
}

// The weighted sum, its backward for a KNOWN upstream gradient (the stepper's seed: d loss / d loss) and the stop rule of the
// step in one launch: the sum is the step's loss, so the rule needs nothing else, and the three were 14 us of a 170-400 us step
// as launches of their own.  The rule runs BEFORE this step's Adam here, which must still see the flag as earlier steps left
// it (the reference applies the update of the step that meets the rule, then returns): the old flag goes to `frozen`, which is
// what the Adam launch reads, and the step counter is advanced here (the Adam launch is told so).
struct StepRule { float* prev; unsigned char* stop; unsigned char* frozen; const unsigned char* armed; float tol; float* loss_out; float* step; };
__global__ void fit_weighted_sum_step_kernel(LossTerms t, const float* denom, float* out, const float* seed, float* grads, StepRule r) {
    const int k = threadIdx.x;
    const float d = *denom;
    if (k < t.n) grads[k] = *seed * t.c[k] / d;
    if (k) return;
    float s = 0.f;
    for (int j = 0; j < t.n; ++j) s += t.c[j] * *t.l[j] / d;
    *out = s;
    if (!r.prev) return;
    const bool was = *r.stop != 0;
    *r.frozen = was ? 1 : 0;
    const float pv = *r.prev;
    const bool hit = fabsf(pv - s) / pv < pv * r.tol;
    if (hit && *r.armed) *r.stop = 1;
    if (!was) *r.prev = s;
    *r.loss_out = s;
    if (r.step) *r.step += 1.0f;
}

}  // namespace

extern "C" {

// One Adam step on nt <= 16 fp32 device tensors (p, g, m, v: host arrays of device pointers, n: their lengths).  step: device
// float, the number of steps taken so far (NOT incremented here: chore_fit_stop_rule does, after this launch); stop: device
// byte, nonzero = leave the parameters alone.
// chore_fit_adam_step_acc: g[k] is the ACCUMULATED gradient (the parameter's .grad, read and written) and gnew[k] (or NULL)
// this step's fresh gradient, added first -- the per-parameter `grad += new` launches of autograd's accumulation folded into
// the update; p[k] / m[k] / v[k] NULL = a leaf that only accumulates.  cols / pstride (or NULL = dense): parameter k is a
// column slice -- rows of cols[k] elements, pstride[k] apart -- of a wider tensor (the split SMPL parameters of a multi-frame
// batch are views b[:, :2], p[:, 3:66], ... of the wrapper's storage); g, gnew, m, v are dense.  step_counted != 0: `step` already
// counts this step (chore_fit_weighted_sum_step advanced it).
static int adam_step_impl(chore_handle* h, float* const* p, float* const* g, const float* const* gnew, float* const* m,
                          float* const* v, const int* n, const int* cols, const int* pstride, int nt, const float* step, float lr,
                          float beta1, float beta2, float eps, const uint8_t* stop, int step_counted, chore_stream_t stream,
                          const char* who) {
    if (!p || !g || !m || !v || !n || !step || !stop || nt <= 0 || nt > FS_MAXT)
        CHORE_FAIL(h, CHORE_EINVAL, "%s: bad argument (at most %d tensors)", who, FS_MAXT);
    AdamTensors t;
    int nmax = 0;
    for (int k = 0; k < nt; ++k) {
        const bool acc_only = gnew && !p[k];
        if (!g[k] || n[k] <= 0 || (!acc_only && (!p[k] || !m[k] || !v[k])) || (acc_only && !gnew[k]))
            CHORE_FAIL(h, CHORE_EINVAL, "%s: null tensor %d", who, k);
        t.p[k] = p[k]; t.g[k] = g[k]; t.gn[k] = gnew ? gnew[k] : nullptr; t.m[k] = m[k]; t.v[k] = v[k]; t.n[k] = n[k];
        t.cols[k] = (cols && cols[k] > 0) ? cols[k] : n[k];
        t.pstride[k] = (cols && pstride) ? pstride[k] : t.cols[k];
        if (n[k] % t.cols[k] || t.pstride[k] < t.cols[k]) CHORE_FAIL(h, CHORE_EINVAL, "%s: tensor %d: %d elements in rows of %d, stride %d", who, k, n[k], t.cols[k], t.pstride[k]);
        nmax = n[k] > nmax ? n[k] : nmax;
    }
    t.nt = nt;
    int bx = (nmax + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(fit_adam_kernel, dim3(bx, nt), dim3(256), 0, (hipStream_t)stream, t, step, lr, beta1, beta2, eps, stop,
                       step_counted);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}
int chore_fit_adam_step(chore_handle* h, float* const* p, const float* const* g, float* const* m, float* const* v, const int* n,
                        int nt, const float* step, float lr, float beta1, float beta2, float eps, const uint8_t* stop,
                        chore_stream_t stream) {
    CHORE_ENTER(h);
    return adam_step_impl(h, p, (float* const*)g, nullptr, m, v, n, nullptr, nullptr, nt, step, lr, beta1, beta2, eps, stop, 0, stream, "chore_fit_adam_step");
}
int chore_fit_adam_step_acc(chore_handle* h, float* const* p, float* const* g, const float* const* gnew, float* const* m,
                            float* const* v, const int* n, const int* cols, const int* pstride, int nt, const float* step, float lr,
                            float beta1, float beta2, float eps, const uint8_t* stop, int step_counted, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!gnew) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_adam_step_acc: gnew is NULL");
    return adam_step_impl(h, p, g, gnew, m, v, n, cols, pstride, nt, step, lr, beta1, beta2, eps, stop, step_counted, stream, "chore_fit_adam_step_acc");
}

// the stop rule of one inner step: hit = |prev - loss| / prev < prev * tol;  stop |= hit & armed;  prev = loss unless stop was
// already set;  loss_out = loss;  step (or NULL) += 1.  All device scalars.
int chore_fit_stop_rule(chore_handle* h, const float* loss, float* prev, uint8_t* stop, const uint8_t* armed, float tol,
                        float* loss_out, float* step, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!loss || !prev || !stop || !armed || !loss_out) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_stop_rule: null argument");
    hipLaunchKernelGGL(fit_stop_rule_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, prev, stop, armed, tol, loss_out, step);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

// out = sum_k coeff[k] * loss[k] / denom over n <= 16 device scalars (losses: host array of device pointers; coeffs: host
// floats; denom, out: device floats) -- the weighting of the fit's loss dictionary, recon_fit_behave.py:339-358 -- and its
// backward: grads[k] = g * coeff[k] / denom.
int chore_fit_weighted_sum(chore_handle* h, const float* const* losses, const float* coeffs, int n, const float* denom, float* out,
                           chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!losses || !coeffs || !denom || !out || n <= 0 || n > FS_MAXL) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_weighted_sum: bad argument");
    LossTerms t;
    for (int k = 0; k < n; ++k) { t.l[k] = losses[k]; t.c[k] = coeffs[k]; if (!losses[k]) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_weighted_sum: null term"); }
    t.n = n;
    hipLaunchKernelGGL(fit_weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, denom, out);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}
int chore_fit_weighted_sum_bwd(chore_handle* h, const float* coeffs, int n, const float* denom, const float* g, float* grads,
                               chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!coeffs || !denom || !g || !grads || n <= 0 || n > FS_MAXL) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_weighted_sum_bwd: bad argument");
    LossTerms t;
    for (int k = 0; k < n; ++k) { t.l[k] = nullptr; t.c[k] = coeffs[k]; }
    t.n = n;
    hipLaunchKernelGGL(fit_weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, denom, g, grads);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}


// chore_fit_weighted_sum, the backward of it for the upstream gradient *seed (grads[k] = *seed * coeff[k] / denom) and -- when
// prev != NULL -- chore_fit_stop_rule on the sum, in one launch.  Differences from the stand-alone rule: `frozen` receives the
// flag as it was BEFORE this step's test (hand it to chore_fit_adam_step_acc as `stop`, with step_counted = 1).
int chore_fit_weighted_sum_step(chore_handle* h, const float* const* losses, const float* coeffs, int n, const float* denom, float* out,
                                const float* seed, float* grads, float* prev, uint8_t* stop, uint8_t* frozen, const uint8_t* armed,
                                float tol, float* loss_out, float* step, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!losses || !coeffs || !denom || !out || !seed || !grads || n <= 0 || n > FS_MAXL)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_weighted_sum_step: bad argument");
    if (prev && (!stop || !frozen || !armed || !loss_out)) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_weighted_sum_step: the rule needs stop, frozen, armed and loss_out");
    LossTerms t;
    for (int k = 0; k < n; ++k) { t.l[k] = losses[k]; t.c[k] = coeffs[k]; if (!losses[k]) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_weighted_sum_step: null term"); }
    t.n = n;
    StepRule r{prev, stop, frozen, armed, tol, loss_out, step};
    hipLaunchKernelGGL(fit_weighted_sum_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, denom, out, seed, grads, r);
    CHORE_LAUNCH_CHECK(h, (hipStream_t)stream);
    return CHORE_OK;
}

}  // extern "C"

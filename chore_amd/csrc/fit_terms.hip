// fit_terms.hip -- the loss terms of a fitting step that are tiny reductions, one launch per group and direction.
//
// forward_smpl (recon_fit_behave.py:293-337) adds, to the field query of the body, seven terms that the reference writes as
// tensor expressions: per step ~45 launches forward and ~85 backward (slices that autograd answers with a zero fill and a
// copy, 63 x 63 products through the GEMM library, means of 25 numbers), each a dependent launch of 2-3 us in a step whose
// kernels with real work take ~0.3 ms.  Here:
//   * chore_fit_smpl_terms: body pose prior (th_smpl_prior.py:32-39), hand prior with the reference's batch-axis quirk
//     (th_hand_prior.py:63-72), the pose-initialisation term and the SMPL depth term (recon_fit_behave.py:321-329) and the
//     2-D keypoint reprojection (recon_fit_base.py:661-680 with model/camera.py:52-66) -- one workgroup;
//   * chore_fit_point_terms: the mean of a clamped distance channel (df_h: recon_fit_base.py:520-526, object:
//     :505-511) and the part cross-entropy summed over the points (recon_fit_behave.py:318-320) -- per-workgroup partial
//     sums in fp64, added in workgroup order by a one-workgroup finish launch.
//   * chore_fit_obj_transform / chore_fit_obj_terms: the placement of the object points and the scale / object-centre
//     terms of forward_step (recon_fit_behave.py:165-186).
// The backward entry points recompute from the inputs and take the upstream gradient of every term as a device scalar.
#include "common.h"

namespace {

constexpr int FT_BODY = 63, FT_HAND = 45, FT_PINIT = 69, FT_KPTS = 25, FT_ROOT = 8;   // pose[3:66], 2 x 45, pose[3:72], body-25, MidHip

struct SmplTermArgs {
    const float *pose, *pose_init, *J, *kpts, *cc;            // (B,P) (B,69) (B,R,3) (B,25,3)|NULL (B,2)
    const float *bmean, *bprec, *hmean, *lprec, *rprec;       // (63) (63,63) (90) (45,45) (45,45)
    int B, P, R;
    float fx, fy, cx, cy, half_crop, crop, net_in, z0;
    float* out[5];                                            // pose, hand, pinit, smplz, j2d   (forward)
    const float* up[5];                                       // upstream gradients, NULL = 0    (backward)
    float *dpose, *dJ;                                        // (B,P) (B,R,3), every element written
};

__device__ __forceinline__ double block_sum(double v, double* sh, int tid) {
    sh[tid] = v;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

template <bool BWD>
__global__ __launch_bounds__(256) void smpl_terms_kernel(SmplTermArgs a) {
    __shared__ float tb[FT_BODY], th[2 * FT_HAND], t2b[FT_BODY], t2h[2 * FT_HAND];
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    float accP = 0.f, accH = 0.f, accI = 0.f, accK = 0.f, accZ = 0.f;
    float up[5];
    if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < 5; ++k) up[k] = a.up[k] ? *a.up[k] : 0.f;
    }
    const float invB = 1.0f / (float)a.B;
    for (int b = 0; b < a.B; ++b) {
        const float* pose = a.pose + (size_t)b * a.P;
        if (tid < FT_BODY) tb[tid] = pose[3 + tid] - a.bmean[tid];
        if (tid < 2 * FT_HAND) th[tid] = pose[3 + FT_BODY + tid] - a.hmean[tid];
        __syncthreads();
        if (tid < FT_BODY) {            // t2 = t @ precision
            float s = 0.f;
            for (int k = 0; k < FT_BODY; ++k) s += tb[k] * a.bprec[k * FT_BODY + tid];
            t2b[tid] = s;
            accP += s * s;
        }
        if (tid >= 64 && tid < 64 + 2 * FT_HAND) {
            const int q = tid - 64, side = q / FT_HAND, c = q - side * FT_HAND;
            const float* prec = side ? a.rprec : a.lprec;
            float s = 0.f;
            for (int k = 0; k < FT_HAND; ++k) s += th[side * FT_HAND + k] * prec[k * FT_HAND + c];
            t2h[q] = s;
            accH += s * s;
        }
        float dI = 0.f;
        if (tid >= 3 && tid < 3 + FT_PINIT) {
            dI = pose[tid] - a.pose_init[(size_t)b * FT_PINIT + tid - 3];
            accI += dI * dI;
        }
        // keypoints: pixels of the network input image, project_screen -> crop shift -> * net_in / crop
        float gK[3] = {0.f, 0.f, 0.f};
        if (tid < FT_KPTS && a.kpts) {
            const float* j = a.J + ((size_t)b * a.R + tid) * 3;
            const float* kp = a.kpts + ((size_t)b * FT_KPTS + tid) * 3;
            const float x = j[0], y = j[1], z = j[2];
            const float px = (a.half_crop + (a.fx * x / z + a.cx)) - a.cc[b * 2 + 0];
            const float py = (a.half_crop + (a.fy * y / z + a.cy)) - a.cc[b * 2 + 1];
            const float qx = px * a.net_in / a.crop, qy = py * a.net_in / a.crop;
            const float dx = qx - kp[0], dy = qy - kp[1], conf = kp[2];
            accK += (dx * dx + dy * dy) * conf;
            if constexpr (BWD) {
                const float w = up[4] * conf / (float)(a.B * FT_KPTS) * a.net_in / a.crop;
                const float gx = 2.0f * dx * w, gy = 2.0f * dy * w;
                gK[0] = gx * a.fx / z;
                gK[1] = gy * a.fy / z;
                gK[2] = -(gx * a.fx * x + gy * a.fy * y) / (z * z);
            }
        }
        float dz = 0.f;
        if (tid == FT_ROOT) {
            dz = a.J[((size_t)b * a.R + FT_ROOT) * 3 + 2] - a.z0;
            accZ += dz * dz;
        }
        if constexpr (BWD) {
            __syncthreads();
            if (tid < a.R) {
                float* o = a.dJ + ((size_t)b * a.R + tid) * 3;
                o[0] = gK[0];
                o[1] = gK[1];
                o[2] = gK[2] + (tid == FT_ROOT ? up[3] * 2.0f * dz * invB : 0.f);
            }
            if (tid < a.P) {
                float v = 0.f;
                if (tid >= 3 && tid < 3 + FT_BODY) {           // d/dt sum(t2^2) = 2 t2 @ precision^T
                    const float* row = a.bprec + (tid - 3) * FT_BODY;
                    float s = 0.f;
                    for (int c = 0; c < FT_BODY; ++c) s += t2b[c] * row[c];
                    v += up[0] * 2.0f * invB * s;
                }
                if (tid >= 3 + FT_BODY && tid < 3 + FT_BODY + 2 * FT_HAND) {
                    const int q = tid - 3 - FT_BODY, side = q / FT_HAND, k = q - side * FT_HAND;
                    const float* row = (side ? a.rprec : a.lprec) + k * FT_HAND;
                    float s = 0.f;
                    for (int c = 0; c < FT_HAND; ++c) s += t2h[side * FT_HAND + c] * row[c];
                    v += up[1] * (2.0f / (float)FT_HAND) * s;
                }
                if (tid >= 3 && tid < 3 + FT_PINIT) v += up[2] * 2.0f * invB * dI;
                a.dpose[(size_t)b * a.P + tid] = v;
            }
        }
        __syncthreads();
    }
    if constexpr (!BWD) {
        const double p = block_sum((double)accP, sh, tid), hh = block_sum((double)accH, sh, tid);
        const double pi = block_sum((double)accI, sh, tid), k2 = block_sum((double)accK, sh, tid);
        const double zz = block_sum((double)accZ, sh, tid);
        if (tid == 0) {
            *a.out[0] = (float)(p / a.B);
            *a.out[1] = (float)(hh / FT_HAND);                 // sum over frames and hands, mean over the 45 columns
            *a.out[2] = (float)(pi / a.B);
            *a.out[3] = (float)(zz / a.B);
            *a.out[4] = a.kpts ? (float)(k2 / ((double)a.B * FT_KPTS)) : 0.f;
        }
    }
}

// ---- per-point terms ----
struct PointTermArgs {
    const float* df;          // (B,2,N)
    const float* logits;      // (B,C,N) or NULL
    const long long* labels;  // (B,N)
    int B, N, C, ch;
    float cmax;
    double* part;             // [B * chunks][2]
    float* out[2];            // clamped mean, cross entropy
    const float* up[2];
    float *ddf, *dlogits;     // (B,2,N) every element written; (B,C,N)
};
constexpr int PT_MAXC = 16;

template <bool BWD>
__global__ __launch_bounds__(256) void point_terms_kernel(PointTermArgs a) {
    __shared__ double sh[256];
    const int tid = threadIdx.x, b = blockIdx.y, n = blockIdx.x * 256 + tid;
    const bool live = n < a.N;
    const int nn = live ? n : a.N - 1;
    const float d = a.df[((size_t)b * 2 + a.ch) * a.N + nn];
    float ce = 0.f;
    float x[PT_MAXC];
    float mx = -INFINITY, se = 0.f;
    int lab = 0;
    if (a.logits) {
#pragma unroll
        for (int c = 0; c < PT_MAXC; ++c) x[c] = c < a.C ? a.logits[((size_t)b * a.C + c) * a.N + nn] : -INFINITY;
        lab = (int)a.labels[(size_t)b * a.N + nn];
#pragma unroll
        for (int c = 0; c < PT_MAXC; ++c) mx = fmaxf(mx, x[c]);
#pragma unroll
        for (int c = 0; c < PT_MAXC; ++c) se += c < a.C ? expf(x[c] - mx) : 0.f;
        float xl = 0.f;
#pragma unroll
        for (int c = 0; c < PT_MAXC; ++c) xl = c == lab ? x[c] : xl;
        ce = (mx + logf(se)) - xl;
    }
    if constexpr (!BWD) {
        const double s0 = block_sum(live ? (double)fminf(d, a.cmax) : 0.0, sh, tid);
        const double s1 = block_sum(live ? (double)ce : 0.0, sh, tid);
        if (tid == 0) {
            double* o = a.part + ((size_t)b * gridDim.x + blockIdx.x) * 2;
            o[0] = s0; o[1] = s1;
        }
    } else if (live) {
        const float u0 = a.up[0] ? *a.up[0] : 0.f;
        a.ddf[((size_t)b * 2 + a.ch) * a.N + n] = d <= a.cmax ? u0 / (float)((size_t)a.B * a.N) : 0.f;
        a.ddf[((size_t)b * 2 + (1 - a.ch)) * a.N + n] = 0.f;
        if (a.logits) {
            const float u1 = (a.up[1] ? *a.up[1] : 0.f) / (float)a.B;
#pragma unroll
            for (int c = 0; c < PT_MAXC; ++c)
                if (c < a.C) a.dlogits[((size_t)b * a.C + c) * a.N + n] = u1 * (expf(x[c] - mx) / se - (c == lab ? 1.f : 0.f));
        }
    }
}

__global__ __launch_bounds__(256) void point_terms_finish_kernel(PointTermArgs a, int nparts) {
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int i = tid; i < nparts; i += 256) { s0 += a.part[(size_t)i * 2]; s1 += a.part[(size_t)i * 2 + 1]; }
    s0 = block_sum(s0, sh, tid);
    s1 = block_sum(s1, sh, tid);
    if (tid == 0) {
        *a.out[0] = (float)(s0 / ((double)a.B * a.N));
        if (a.out[1]) *a.out[1] = (float)(s1 / a.B);       // summed over the points, mean over the frames
    }
}

// ---- object placement: out = (v R + t) s   (transform_obj_verts, recon_fit_base.py:367-371: rotate, translate, THEN scale) ----
__global__ __launch_bounds__(256) void obj_transform_fwd_kernel(const float* __restrict__ v0, const float* __restrict__ R,
                                                                const float* __restrict__ t, const float* __restrict__ sc, int N,
                                                                float* __restrict__ out) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* r = R + (size_t)b * 9;
    const float* v = v0 + ((size_t)b * N + n) * 3;
    const float x = v[0], y = v[1], z = v[2], s = sc[b];
    float* o = out + ((size_t)b * N + n) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (((x * r[c] + y * r[3 + c]) + z * r[6 + c]) + t[b * 3 + c]) * s;
}

// d R[k][c] = s sum_n v[n][k] g[n][c];  d t[c] = s sum_n g[n][c];  d s = sum_n g[n] . (v[n] R + t)      one workgroup per frame
__global__ __launch_bounds__(256) void obj_transform_bwd_kernel(const float* __restrict__ v0, const float* __restrict__ R,
                                                                const float* __restrict__ t, const float* __restrict__ sc,
                                                                const float* __restrict__ g, int N, float* __restrict__ dR,
                                                                float* __restrict__ dt, float* __restrict__ ds) {
    __shared__ double sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* r = R + (size_t)b * 9;
    float a[13];
#pragma unroll
    for (int e = 0; e < 13; ++e) a[e] = 0.f;
    for (int n = tid; n < N; n += 256) {
        const float* v = v0 + ((size_t)b * N + n) * 3;
        const float* gg = g + ((size_t)b * N + n) * 3;
        const float x[3] = {v[0], v[1], v[2]}, gv[3] = {gg[0], gg[1], gg[2]};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) a[k * 3 + c] += x[k] * gv[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[9 + c] += gv[c];
            a[12] += gv[c] * (((x[0] * r[c] + x[1] * r[3 + c]) + x[2] * r[6 + c]) + t[b * 3 + c]);
        }
    }
    const float s = sc[b];
#pragma unroll
    for (int e = 0; e < 13; ++e) {
        const double v = block_sum((double)a[e], sh, tid);
        if (tid == 0) {
            if (e < 9) dR[(size_t)b * 9 + e] = (float)(v * (double)s);
            else if (e < 12) dt[b * 3 + e - 9] = (float)(v * (double)s);
            else ds[b] = (float)v;
        }
    }
}

// ---- object terms of forward_step (recon_fit_behave.py:174-186): scale = mean_b (s - s0)^2 and
//      ocent = mean_b sum_k (mean_n object[b][n][k] - (smpl_center[b][k] + mean_n centers[b][3 + k][n]))^2 ----
__global__ __launch_bounds__(256) void obj_terms_sums_kernel(const float* __restrict__ object, const float* __restrict__ centers, int N,
                                                             double* __restrict__ sums /*[B][6]*/) {
    __shared__ double sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = tid; n < N; n += 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a[k] += object[((size_t)b * N + n) * 3 + k];
            a[3 + k] += centers[((size_t)b * 6 + 3 + k) * N + n];
        }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const double v = block_sum((double)a[e], sh, tid);
        if (tid == 0) sums[(size_t)b * 6 + e] = v;
    }
}
__global__ void obj_terms_finish_kernel(const double* __restrict__ sums, const float* __restrict__ smpl_center,
                                        const float* __restrict__ obj_s, float s0, int B, int N, float* __restrict__ diff /*[B][3]*/,
                                        float* out_scale, float* out_ocent) {
    if (threadIdx.x || blockIdx.x) return;
    double sc = 0.0, oc = 0.0;
    for (int b = 0; b < B; ++b) {
        const float ds = obj_s[b] - s0;
        sc += (double)(ds * ds);
        for (int k = 0; k < 3; ++k) {
            const float mo = (float)(sums[(size_t)b * 6 + k] / N), mc = (float)(sums[(size_t)b * 6 + 3 + k] / N);
            const float d = mo - (smpl_center[b * 3 + k] + mc);
            diff[b * 3 + k] = d;
            oc += (double)(d * d);
        }
    }
    *out_scale = (float)(sc / B);
    *out_ocent = (float)(oc / B);
}
__global__ __launch_bounds__(256) void obj_terms_bwd_kernel(const float* __restrict__ diff, const float* __restrict__ obj_s, float s0,
                                                            const float* up_scale, const float* up_ocent, int B, int N,
                                                            float* __restrict__ dobject, float* __restrict__ dcenters,
                                                            float* __restrict__ dscale) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    const float uo = up_ocent ? *up_ocent : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) dscale[b] = (up_scale ? *up_scale : 0.f) * 2.0f * (obj_s[b] - s0) / (float)B;
    if (n >= N) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gk = uo * 2.0f * diff[b * 3 + k] / (float)B / (float)N;
        dobject[((size_t)b * N + n) * 3 + k] = gk;
        dcenters[((size_t)b * 6 + k) * N + n] = 0.f;
        dcenters[((size_t)b * 6 + 3 + k) * N + n] = -gk;
    }
}

// ---- the SO(3) perturbation of a step: out = rot + scale * noise[k], then k += 1   (decopose_axis, recon_fit_base.py:374-384,
//      with the per-step draws laid out up front; k is the device-side step counter a recorded graph advances) ----
__global__ __launch_bounds__(256) void rot_noise_kernel(const float* __restrict__ rot, const float* __restrict__ noise,
                                                        long long* __restrict__ k, float scale, int n /*B*9*/, long long steps,
                                                        float* __restrict__ out) {
    const long long kk = *k;
    const long long kc = kk < 0 ? 0 : (kk >= steps ? steps - 1 : kk);
    for (int i = threadIdx.x; i < n; i += 256) out[i] = rot[i] + scale * noise[(size_t)kc * n + i];
    __syncthreads();
    if (threadIdx.x == 0) *k = kk + 1;
}

int check_smpl(chore_handle* h, const SmplTermArgs& a, const char* who) {
    if (!a.pose || !a.pose_init || !a.J || !a.cc || !a.bmean || !a.bprec || !a.hmean || !a.lprec || !a.rprec)
        CHORE_FAIL(h, CHORE_EINVAL, "%s: null argument", who);
    if (a.B < 1 || a.P != 3 + FT_BODY + 2 * FT_HAND || a.R < FT_KPTS || a.R > 256)
        CHORE_FAIL(h, CHORE_EINVAL, "%s: unsupported sizes B=%d pose=%d landmarks=%d (SMPL-H pose of 156, 25..256 landmark rows)", who,
                   a.B, a.P, a.R);
    return CHORE_OK;
}

}  // namespace

extern "C" {

int chore_fit_smpl_terms_fwd(chore_handle* h, const float* pose, const float* pose_init, const float* J, const float* kpts,
                             const float* crop_center, const float* body_mean, const float* body_prec, const float* hand_mean,
                             const float* lhand_prec, const float* rhand_prec, int B, int P, int R, const float* cam8,
                             float* const* out5, chore_stream_t stream) {
    CHORE_ENTER(h);
    SmplTermArgs a{};
    a.pose = pose; a.pose_init = pose_init; a.J = J; a.kpts = kpts; a.cc = crop_center;
    a.bmean = body_mean; a.bprec = body_prec; a.hmean = hand_mean; a.lprec = lhand_prec; a.rprec = rhand_prec;
    a.B = B; a.P = P; a.R = R;
    if (!cam8 || !out5) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_smpl_terms_fwd: null argument");
    a.fx = cam8[0]; a.fy = cam8[1]; a.cx = cam8[2]; a.cy = cam8[3]; a.half_crop = cam8[4]; a.crop = cam8[5]; a.net_in = cam8[6]; a.z0 = cam8[7];
    if (int rc = check_smpl(h, a, "chore_fit_smpl_terms_fwd")) return rc;
    for (int k = 0; k < 5; ++k) {
        if (!out5[k]) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_smpl_terms_fwd: null output %d", k);
        a.out[k] = out5[k];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smpl_terms_kernel<false>, dim3(1), dim3(256), 0, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_fit_smpl_terms_bwd(chore_handle* h, const float* pose, const float* pose_init, const float* J, const float* kpts,
                             const float* crop_center, const float* body_mean, const float* body_prec, const float* hand_mean,
                             const float* lhand_prec, const float* rhand_prec, int B, int P, int R, const float* cam8,
                             const float* const* up5, float* dpose, float* dJ, chore_stream_t stream) {
    CHORE_ENTER(h);
    SmplTermArgs a{};
    a.pose = pose; a.pose_init = pose_init; a.J = J; a.kpts = kpts; a.cc = crop_center;
    a.bmean = body_mean; a.bprec = body_prec; a.hmean = hand_mean; a.lprec = lhand_prec; a.rprec = rhand_prec;
    a.B = B; a.P = P; a.R = R;
    if (!cam8 || !up5 || !dpose || !dJ) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_smpl_terms_bwd: null argument");
    a.fx = cam8[0]; a.fy = cam8[1]; a.cx = cam8[2]; a.cy = cam8[3]; a.half_crop = cam8[4]; a.crop = cam8[5]; a.net_in = cam8[6]; a.z0 = cam8[7];
    if (int rc = check_smpl(h, a, "chore_fit_smpl_terms_bwd")) return rc;
    for (int k = 0; k < 5; ++k) a.up[k] = up5[k];
    a.dpose = dpose; a.dJ = dJ;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smpl_terms_kernel<true>, dim3(1), dim3(256), 0, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

size_t chore_fit_point_terms_workspace_bytes(int B, int N) { return (size_t)B * ((N + 255) / 256) * 2 * sizeof(double); }

int chore_fit_point_terms_fwd(chore_handle* h, const float* df, int channel, float clamp_max, const float* logits,
                              const int64_t* labels, int B, int N, int C, float* out_clamped_mean, float* out_cross_entropy,
                              void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!df || !out_clamped_mean || !workspace || B < 1 || B > 65535 || N < 1 || (channel != 0 && channel != 1))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_point_terms_fwd: bad argument (B=%d N=%d channel=%d)", B, N, channel);
    if (logits && (!labels || !out_cross_entropy || C < 1 || C > PT_MAXC))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_point_terms_fwd: logits need labels, an output and 1..%d classes (C=%d)", PT_MAXC, C);
    PointTermArgs a{};
    a.df = df; a.logits = logits; a.labels = (const long long*)labels; a.B = B; a.N = N; a.C = C; a.ch = channel; a.cmax = clamp_max;
    a.part = (double*)workspace; a.out[0] = out_clamped_mean; a.out[1] = logits ? out_cross_entropy : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int chunks = (N + 255) / 256;
    hipLaunchKernelGGL(point_terms_kernel<false>, dim3(chunks, B), dim3(256), 0, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    hipLaunchKernelGGL(point_terms_finish_kernel, dim3(1), dim3(256), 0, s, a, chunks * B);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_fit_point_terms_bwd(chore_handle* h, const float* df, int channel, float clamp_max, const float* logits,
                              const int64_t* labels, int B, int N, int C, const float* up_clamped_mean,
                              const float* up_cross_entropy, float* ddf, float* dlogits, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!df || !ddf || B < 1 || B > 65535 || N < 1 || (channel != 0 && channel != 1))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_point_terms_bwd: bad argument (B=%d N=%d channel=%d)", B, N, channel);
    if (logits && (!labels || !dlogits || C < 1 || C > PT_MAXC))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_point_terms_bwd: logits need labels, an output and 1..%d classes (C=%d)", PT_MAXC, C);
    PointTermArgs a{};
    a.df = df; a.logits = logits; a.labels = (const long long*)labels; a.B = B; a.N = N; a.C = C; a.ch = channel; a.cmax = clamp_max;
    a.up[0] = up_clamped_mean; a.up[1] = up_cross_entropy; a.ddf = ddf; a.dlogits = dlogits;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(point_terms_kernel<true>, dim3((N + 255) / 256, B), dim3(256), 0, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int chore_fit_obj_transform_fwd(chore_handle* h, const float* verts, const float* R, const float* t, const float* s, int B, int N,
                                float* out, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!verts || !R || !t || !s || !out || B < 1 || B > 65535 || N < 1)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_obj_transform_fwd: bad argument (B=%d N=%d)", B, N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(obj_transform_fwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, verts, R, t, s, N, out);
    CHORE_LAUNCH_CHECK(h, st);
    return CHORE_OK;
}

int chore_fit_obj_transform_bwd(chore_handle* h, const float* verts, const float* R, const float* t, const float* s, const float* g,
                                int B, int N, float* dR, float* dt, float* ds, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!verts || !R || !t || !s || !g || !dR || !dt || !ds || B < 1 || N < 1)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_obj_transform_bwd: bad argument (B=%d N=%d)", B, N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(obj_transform_bwd_kernel, dim3(B), dim3(256), 0, st, verts, R, t, s, g, N, dR, dt, ds);
    CHORE_LAUNCH_CHECK(h, st);
    return CHORE_OK;
}

size_t chore_fit_obj_terms_workspace_bytes(int B) { return (size_t)B * 6 * sizeof(double); }

int chore_fit_obj_terms_fwd(chore_handle* h, const float* object, const float* centers, const float* smpl_center, const float* obj_s,
                            float scale0, int B, int N, float* diff, float* out_scale, float* out_ocent, void* workspace,
                            chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!object || !centers || !smpl_center || !obj_s || !diff || !out_scale || !out_ocent || !workspace || B < 1 || N < 1)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_obj_terms_fwd: bad argument (B=%d N=%d)", B, N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(obj_terms_sums_kernel, dim3(B), dim3(256), 0, st, object, centers, N, (double*)workspace);
    CHORE_LAUNCH_CHECK(h, st);
    hipLaunchKernelGGL(obj_terms_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, smpl_center, obj_s, scale0, B, N,
                       diff, out_scale, out_ocent);
    CHORE_LAUNCH_CHECK(h, st);
    return CHORE_OK;
}

int chore_fit_obj_terms_bwd(chore_handle* h, const float* diff, const float* obj_s, float scale0, const float* up_scale,
                            const float* up_ocent, int B, int N, float* dobject, float* dcenters, float* dscale,
                            chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!diff || !obj_s || !dobject || !dcenters || !dscale || B < 1 || B > 65535 || N < 1)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_obj_terms_bwd: bad argument (B=%d N=%d)", B, N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(obj_terms_bwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, diff, obj_s, scale0, up_scale, up_ocent, B, N,
                       dobject, dcenters, dscale);
    CHORE_LAUNCH_CHECK(h, st);
    return CHORE_OK;
}

int chore_fit_rot_noise(chore_handle* h, const float* rot, const float* noise, int64_t* k, float scale, int B, int64_t steps,
                        float* out, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!rot || !noise || !k || !out || B < 1 || steps < 1) CHORE_FAIL(h, CHORE_EINVAL, "chore_fit_rot_noise: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rot_noise_kernel, dim3(1), dim3(256), 0, st, rot, noise, (long long*)k, scale, B * 9, (long long)steps, out);
    CHORE_LAUNCH_CHECK(h, st);
    return CHORE_OK;
}

}  // extern "C"

// contact.hip -- the human-object contact term of the joint fitting phase, forward and backward.
//
// Replaces ReconFitterBase.compute_contact_loss (recon/recon_fit_base.py:553-608) together with the
// pytorch3d.loss.chamfer_distance call it ends in (:605-607; pytorch3d is not vendored: defaults restated --
// squared-L2 nearest neighbour, mean over the points of a cloud, mean over the clouds, both directions added).
// The reference builds ragged per-(frame, part) point clouds with data-dependent Python loops and host
// synchronisations (mask.sum(), torch.where per part).  Here everything is a fixed-shape device computation,
// so a fitting step has no host round trip and can be captured in a hipGraph:
//   1. prep: contact masks df < thres, argmax part label of every object point, per-frame contact counts
//   2. select: which points take part (a frame with no contact at all is skipped; a side with no contact point
//      uses ALL its points -- reference :573-584) and how many per (frame, part)
//   3. nn: for every participating point the nearest participating point OF THE SAME PART in the other cloud
//      (brute force through LDS tiles; 6890 x 3000 candidates per frame)
//   4. reduce: per (frame, part, direction) sums in a fixed order, then
//        loss = 1/P sum_pairs mean_a min_b |a-b|^2 + 1/P sum_pairs mean_b min_a |a-b|^2,  P = #(frame, part)
//      pairs where both clouds are non-empty (0 if there is none: the reference then omits the term)
//   backward: gather form (every point sums the contributions of the points it is the nearest neighbour of,
//      in index order) -- deterministic, no float atomics.
#include "common.h"

namespace {

constexpr int CP_MAX = 32;      // max part count (14 in CHORE)
constexpr int TILE = 256;

struct CWs {                    // workspace layout (element offsets, see contact_ws)
    int* label_o;               // [B][No] argmax part of the object points
    int* sel_h;                 // [B][Nh] 1 if the vertex takes part
    int* sel_o;                 // [B][No]
    int* cnt;                   // [B][2] contact counts (human, object)
    int* n_part;                // [B][P][2] participating points per part (human, object)
    int* npairs;                // [1]
    int* nn_h;                  // [B][Nh] nearest object point of the same part, or -1
    int* nn_o;                  // [B][No] nearest human vertex of the same part, or -1
    float* m_h;                 // [B][Nh] squared distance to it
    float* m_o;                 // [B][No]
    float* pair_sum;            // [B][P][2]
    unsigned long long* key_h;  // [B][Nh] packed (distance bits << 32 | index) minima, ~0 = none
    unsigned long long* key_o;  // [B][No]
    float* part_h;              // [chunks_o][B][Nh][3] backward partials of the human side (one per candidate chunk)
    float* part_o;              // [chunks_h][B][No][3]
    size_t bytes;
};

CWs contact_ws(void* base, int B, int Nh, int No, int P) {
    CWs w;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t n) { char* r = p + off; off += (n + 255) / 256 * 256; return r; };
    w.label_o = (int*)take(sizeof(int) * B * No);
    w.sel_h = (int*)take(sizeof(int) * B * Nh);
    w.sel_o = (int*)take(sizeof(int) * B * No);
    w.cnt = (int*)take(sizeof(int) * B * 2);
    w.n_part = (int*)take(sizeof(int) * B * P * 2);
    w.npairs = (int*)take(sizeof(int));
    w.nn_h = (int*)take(sizeof(int) * B * Nh);
    w.nn_o = (int*)take(sizeof(int) * B * No);
    w.m_h = (float*)take(sizeof(float) * B * Nh);
    w.m_o = (float*)take(sizeof(float) * B * No);
    w.pair_sum = (float*)take(sizeof(float) * B * P * 2);
    w.key_h = (unsigned long long*)take(sizeof(unsigned long long) * B * Nh);
    w.key_o = (unsigned long long*)take(sizeof(unsigned long long) * B * No);   // contiguous with key_h: one memset
    w.part_h = (float*)take(sizeof(float) * (size_t)((No + TILE - 1) / TILE) * B * Nh * 3);
    w.part_o = (float*)take(sizeof(float) * (size_t)((Nh + TILE - 1) / TILE) * B * No * 3);
    w.bytes = off;
    return w;
}

// plain fill instead of hipMemsetAsync: the step is recorded into hipGraphs, and byte-memset nodes of odd sizes
// replayed unreliably there (sporadic GPU memory faults), kernel nodes do not
__global__ void contact_fill_kernel(unsigned* __restrict__ p, unsigned v, size_t n, unsigned* __restrict__ p1, unsigned v1, size_t n1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // two ranges in one launch
    if (i < n) p[i] = v;
    else if (i < n + n1) p1[i - n] = v1;
}

// ---- 1. masks, labels, contact counts ---------------------------------------------------------------
__global__ void contact_prep_kernel(const float* __restrict__ df_hum_o, const float* __restrict__ df_obj_h,
                                    const float* __restrict__ logits /*(B,P,No)*/, int B, int Nh, int No, int P,
                                    float thres, CWs w) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int ch = 0, co = 0;
    if (i < Nh) {
        const int m = df_hum_o[(size_t)b * Nh + i] < thres;
        w.sel_h[(size_t)b * Nh + i] = m;   // contact mask for now; turned into the selection by the next kernel
        ch = m;
    }
    if (i < No) {
        const int m = df_obj_h[(size_t)b * No + i] < thres;
        w.sel_o[(size_t)b * No + i] = m;
        co = m;
        const float* l = logits + (size_t)b * P * No + i;
        int best = 0;
        float bv = l[0];
        for (int p = 1; p < P; ++p) {      // first maximum, like torch.argmax
            const float v = l[(size_t)p * No];
            if (v > bv) { bv = v; best = p; }
        }
        w.label_o[(size_t)b * No + i] = best;
    }
    // integer atomics: exact and order-independent
    __shared__ int s[2];
    if (threadIdx.x < 2) s[threadIdx.x] = 0;
    __syncthreads();
    if (ch) atomicAdd(&s[0], 1);
    if (co) atomicAdd(&s[1], 1);
    __syncthreads();
    if (threadIdx.x < 2 && s[threadIdx.x]) atomicAdd(&w.cnt[b * 2 + threadIdx.x], s[threadIdx.x]);
}

// ---- 2. selection and per-part counts ---------------------------------------------------------------
__global__ void contact_select_kernel(const int* __restrict__ label_h, int B, int Nh, int No, int P, CWs w) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = w.cnt[b * 2], co = w.cnt[b * 2 + 1];
    const bool frame = ch + co > 0;
    __shared__ int s[CP_MAX * 2];
    for (int k = threadIdx.x; k < P * 2; k += blockDim.x) s[k] = 0;
    __syncthreads();
    if (i < Nh) {
        const int sel = frame && (ch > 0 ? w.sel_h[(size_t)b * Nh + i] : 1);
        w.sel_h[(size_t)b * Nh + i] = sel;
        if (sel) atomicAdd(&s[label_h[i] * 2], 1);
    }
    if (i < No) {
        const int sel = frame && (co > 0 ? w.sel_o[(size_t)b * No + i] : 1);
        w.sel_o[(size_t)b * No + i] = sel;
        if (sel) atomicAdd(&s[w.label_o[(size_t)b * No + i] * 2 + 1], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < P * 2; k += blockDim.x)
        if (s[k]) atomicAdd(&w.n_part[(size_t)b * P * 2 + k], s[k]);
}

__global__ void contact_pairs_kernel(int B, int P, CWs w) {
    __shared__ int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    int c = 0;
    for (int k = threadIdx.x; k < B * P; k += blockDim.x)
        c += (w.n_part[(size_t)k * 2] > 0 && w.n_part[(size_t)k * 2 + 1] > 0);
    if (c) atomicAdd(&s, c);
    __syncthreads();
    if (threadIdx.x == 0) *w.npairs = s;
}

// ---- 3. same-part nearest neighbour ------------------------------------------------------------------
// queries Q (Nq points of frame b), candidates C (Nc points); a query takes part if sel_q and its (frame, part)
// pair is valid; candidates must be selected and carry the same label.  A block compares TILE queries with ONE
// chunk of TILE candidates (grid = query tiles x candidate chunks x frames, so the 6890 x 3000 comparisons of a
// frame spread over ~320 blocks) and merges its minimum with a 64-bit atomicMin on (distance bits << 32 | index):
// distances are non-negative, so the bit pattern orders like the value, and equal distances resolve to the
// smaller index -- the same first-minimum a sequential scan finds, whatever order the blocks run in.
struct NNSide {      // one direction of the pairing: queries of one cloud against candidates of the other
    const float* Q; const int* sel_q; const int* lab_q; int lab_q_stride, Nq;
    const float* C; const int* sel_c; const int* lab_c; int lab_c_stride, Nc;
    unsigned long long* key;
};
// both directions in one launch (they are independent): blockIdx.y = direction, blockIdx.x = query tile + tiles * chunk
__global__ __launch_bounds__(TILE) void contact_nn_kernel(NNSide s0, NNSide s1, int P, const int* __restrict__ n_part) {
    __shared__ f32x4 cand[TILE];              // x, y, z and the label's bits (-1: the candidate does not take part): one 16-byte read
    const NNSide& sd = blockIdx.y ? s1 : s0;
    const float* __restrict__ Q = sd.Q; const int* __restrict__ sel_q = sd.sel_q; const int* __restrict__ lab_q = sd.lab_q;
    const float* __restrict__ C = sd.C; const int* __restrict__ sel_c = sd.sel_c; const int* __restrict__ lab_c = sd.lab_c;
    const int lab_q_stride = sd.lab_q_stride, Nq = sd.Nq, lab_c_stride = sd.lab_c_stride, Nc = sd.Nc;
    unsigned long long* __restrict__ key = sd.key;
    const int tq = (Nq + TILE - 1) / TILE;
    const int b = blockIdx.z;
    const int q = (blockIdx.x % tq) * TILE + threadIdx.x;
    const int c0 = (blockIdx.x / tq) * TILE;
    // contact points are a small share of both clouds: most (query tile, candidate chunk) pairs have no query or no candidate
    // that takes part and leave here, before the chunk is staged (block-uniform exits; what they skip contributes nothing)
    int l = 0;
    bool mine = false;
    if (q < Nq) {
        l = lab_q[(size_t)b * lab_q_stride + q];
        const int* np = n_part + ((size_t)b * P + l) * 2;
        mine = sel_q[(size_t)b * Nq + q] && np[0] > 0 && np[1] > 0;
    }
    if (!__syncthreads_or(mine)) return;
    int lc = -1;
    {
        const int c = c0 + threadIdx.x;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < Nc) {
            const float* p = C + ((size_t)b * Nc + c) * 3;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
            lc = sel_c[(size_t)b * Nc + c] ? lab_c[(size_t)b * lab_c_stride + c] : -1;
        }
        v[3] = __int_as_float(lc);
        cand[threadIdx.x] = v;
    }
    if (!__syncthreads_or(lc >= 0)) return;
    if (!mine) return;
    const float* p = Q + ((size_t)b * Nq + q) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float best = 3.0e38f;
    int bi = -1;
    const int n = min(TILE, Nc - c0);
#pragma unroll 4
    for (int j = 0; j < n; ++j) {
        const f32x4 v = cand[j];
        if (__float_as_int(v[3]) == l) {
            const float dx = x - v[0], dy = y - v[1], dz = z - v[2];
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d < best) { best = d; bi = c0 + j; }
        }
    }
    if (bi >= 0)
        atomicMin(key + (size_t)b * Nq + q, ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi);
}

// both clouds' keys in one launch: elements [0, n0) of the first set, then [0, n1) of the second
__global__ void contact_unpack_kernel(const unsigned long long* __restrict__ key0, size_t n0, int* __restrict__ nn0,
                                      float* __restrict__ mind0, const unsigned long long* __restrict__ key1, size_t n1,
                                      int* __restrict__ nn1, float* __restrict__ mind1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0 + n1) return;
    const bool second = i >= n0;
    if (second) i -= n0;
    const unsigned long long* key = second ? key1 : key0;
    int* nn = second ? nn1 : nn0;
    float* mind = second ? mind1 : mind0;
    const unsigned long long k = key[i];
    const bool none = k == ~0ull;
    nn[i] = none ? -1 : (int)(unsigned)(k & 0xffffffffu);
    mind[i] = none ? 0.f : __uint_as_float((unsigned)(k >> 32));
}

// ---- 4. per-(frame, part, direction) sums, fixed order -------------------------------------------------
__global__ __launch_bounds__(256) void contact_pair_sum_kernel(const int* __restrict__ label_h, int B, int Nh, int No,
                                                               int P, CWs w) {
    const int dir = blockIdx.x & 1, part = (blockIdx.x >> 1) % P, b = (blockIdx.x >> 1) / P;
    const int N = dir ? No : Nh;
    const int* nn = dir ? w.nn_o + (size_t)b * No : w.nn_h + (size_t)b * Nh;
    const float* m = dir ? w.m_o + (size_t)b * No : w.m_h + (size_t)b * Nh;
    const int* lab = dir ? w.label_o + (size_t)b * No : label_h;
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 256)
        if (nn[i] >= 0 && lab[i] == part) acc += m[i];
    __shared__ float s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) w.pair_sum[((size_t)b * P + part) * 2 + dir] = s[0];
}

__global__ void contact_finish_kernel(int B, int P, CWs w, float* __restrict__ loss) {
    if (threadIdx.x || blockIdx.x) return;
    const int np = *w.npairs;
    float a = 0.f, c = 0.f;
    for (int k = 0; k < B * P; ++k) {
        const int nh = w.n_part[(size_t)k * 2], no = w.n_part[(size_t)k * 2 + 1];
        if (nh > 0 && no > 0) {
            a += w.pair_sum[(size_t)k * 2] / (float)nh;
            c += w.pair_sum[(size_t)k * 2 + 1] / (float)no;
        }
    }
    *loss = np > 0 ? a / (float)np + c / (float)np : 0.f;
}

// ---- backward: d loss / d point, gather form ------------------------------------------------------------
// own term: w_q * 2 (q - nn(q));  received: for every point r of the other cloud with nn(r) == q: -w_r * 2 (r - q).
// As in the forward a block pairs TILE points with one chunk of the other cloud; the received terms of a chunk are
// summed in index order into a per-chunk partial and contact_bwd_finish_kernel adds the chunks in order: no float
// atomics, bit-reproducible.
__global__ __launch_bounds__(TILE) void contact_bwd_kernel(const float* __restrict__ Q, int Nq, int side_q,
                                                           const float* __restrict__ C, const int* __restrict__ nn_c,
                                                           const int* __restrict__ lab_c, int lab_c_stride, int Nc,
                                                           int P, const int* __restrict__ n_part,
                                                           const int* __restrict__ npairs, const float* __restrict__ g,
                                                           float* __restrict__ part /*[chunks][B][Nq][3]*/) {
    __shared__ float cx[TILE], cy[TILE], cz[TILE], cw[TILE];
    __shared__ int cn[TILE];
    const int b = blockIdx.z, B = gridDim.z;
    const int q = blockIdx.x * TILE + threadIdx.x;
    const int c0 = blockIdx.y * TILE;
    const int np = *npairs;
    const float gs = np > 0 ? g[0] / (float)np : 0.f;
    {
        const int c = c0 + threadIdx.x;
        int j = -1;
        if (c < Nc) {
            j = nn_c[(size_t)b * Nc + c];
            if (j >= 0) {
                const float* p = C + ((size_t)b * Nc + c) * 3;
                cx[threadIdx.x] = p[0]; cy[threadIdx.x] = p[1]; cz[threadIdx.x] = p[2];
                const int l = lab_c[(size_t)b * lab_c_stride + c];
                cw[threadIdx.x] = 2.f * gs / (float)n_part[((size_t)b * P + l) * 2 + (1 - side_q)];
            }
        }
        cn[threadIdx.x] = j;
    }
    const int lo = blockIdx.x * TILE;
    const bool any = __syncthreads_or(cn[threadIdx.x] >= lo && cn[threadIdx.x] < lo + TILE);    // a candidate that points into this tile?
    if (q >= Nq) return;
    const float* p = Q + ((size_t)b * Nq + q) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const int n = any ? min(TILE, Nc - c0) : 0;
    for (int k = 0; k < n; ++k) {
        if (cn[k] == q) {   // r = candidate k has q as its nearest neighbour: d m_r / d q = -2 (r - q)
            gx -= cw[k] * (cx[k] - x); gy -= cw[k] * (cy[k] - y); gz -= cw[k] * (cz[k] - z);
        }
    }
    float* o = part + (((size_t)blockIdx.y * B + b) * Nq + q) * 3;
    o[0] = gx; o[1] = gy; o[2] = gz;
}

__global__ void contact_bwd_finish_kernel(const float* __restrict__ Q, const int* __restrict__ nn_q,
                                          const int* __restrict__ lab_q, int lab_q_stride, int Nq, int side_q,
                                          const float* __restrict__ C, int Nc, int P, const int* __restrict__ n_part,
                                          const int* __restrict__ npairs, const float* __restrict__ g,
                                          const float* __restrict__ part, int chunks, float* __restrict__ dQ) {
    const int b = blockIdx.y, B = gridDim.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Nq) return;
    const int np = *npairs;
    const float gs = np > 0 ? g[0] / (float)np : 0.f;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const int j = nn_q[(size_t)b * Nq + q];
    if (j >= 0) {
        const float* p = Q + ((size_t)b * Nq + q) * 3;
        const int l = lab_q[(size_t)b * lab_q_stride + q];
        const float wq = 2.f * gs / (float)n_part[((size_t)b * P + l) * 2 + side_q];
        const float* o = C + ((size_t)b * Nc + j) * 3;
        gx = wq * (p[0] - o[0]); gy = wq * (p[1] - o[1]); gz = wq * (p[2] - o[2]);
    }
    for (int c = 0; c < chunks; ++c) {
        const float* o = part + (((size_t)c * B + b) * Nq + q) * 3;
        gx += o[0]; gy += o[1]; gz += o[2];
    }
    float* o = dQ + ((size_t)b * Nq + q) * 3;
    o[0] = gx; o[1] = gy; o[2] = gz;
}

}  // namespace

extern "C" size_t chore_contact_workspace_bytes(int B, int Nh, int No, int P) {
    return contact_ws(nullptr, B, Nh, No, P).bytes;
}

extern "C" int chore_contact_fwd(chore_handle* h, const float* hum, const float* obj, const float* df_hum_o,
                                 const float* df_obj_h, const int* label_h, const float* part_logits, int B, int Nh,
                                 int No, int P, float thres, float* loss, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!hum || !obj || !df_hum_o || !df_obj_h || !label_h || !part_logits || !loss || !workspace)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_contact_fwd: null argument");
    if (B <= 0 || Nh <= 0 || No <= 0 || P <= 0 || P > CP_MAX)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_contact_fwd: bad sizes B=%d Nh=%d No=%d P=%d", B, Nh, No, P);
    hipStream_t s = (hipStream_t)stream;
    CWs w = contact_ws(workspace, B, Nh, No, P);
    // counters: cnt, n_part (contiguous up to npairs)
    {   // the counters to 0 and the nearest-neighbour keys to "none", one launch
        const size_t n = ((char*)w.npairs - (char*)w.cnt) / 4 + 1;   // cnt, n_part, npairs
        const size_t nk = ((char*)w.part_h - (char*)w.key_h) / 4;    // key_h and key_o
        hipLaunchKernelGGL(contact_fill_kernel, dim3((unsigned)((n + nk + 255) / 256)), dim3(256), 0, s, (unsigned*)w.cnt, 0u, n,
                           (unsigned*)w.key_h, 0xffffffffu, nk);
    }
    const int Nm = Nh > No ? Nh : No;
    dim3 gm((Nm + 255) / 256, B);
    hipLaunchKernelGGL(contact_prep_kernel, gm, dim3(256), 0, s, df_hum_o, df_obj_h, part_logits, B, Nh, No, P, thres, w);
    hipLaunchKernelGGL(contact_select_kernel, gm, dim3(256), 0, s, label_h, B, Nh, No, P, w);
    hipLaunchKernelGGL(contact_pairs_kernel, dim3(1), dim3(256), 0, s, B, P, w);
    const int th = (Nh + TILE - 1) / TILE, to = (No + TILE - 1) / TILE;
    {
        const NNSide s0{hum, w.sel_h, label_h, 0, Nh, obj, w.sel_o, w.label_o, No, No, w.key_h};
        const NNSide s1{obj, w.sel_o, w.label_o, No, No, hum, w.sel_h, label_h, 0, Nh, w.key_o};
        hipLaunchKernelGGL(contact_nn_kernel, dim3(th * to, 2, B), dim3(TILE), 0, s, s0, s1, P, w.n_part);
    }
    hipLaunchKernelGGL(contact_unpack_kernel, dim3((unsigned)(((size_t)B * (Nh + No) + 255) / 256)), dim3(256), 0, s, w.key_h,
                       (size_t)B * Nh, w.nn_h, w.m_h, w.key_o, (size_t)B * No, w.nn_o, w.m_o);
    hipLaunchKernelGGL(contact_pair_sum_kernel, dim3(B * P * 2), dim3(256), 0, s, label_h, B, Nh, No, P, w);
    hipLaunchKernelGGL(contact_finish_kernel, dim3(1), dim3(64), 0, s, B, P, w, loss);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

extern "C" int chore_contact_bwd(chore_handle* h, const float* hum, const float* obj, const int* label_h, int B, int Nh,
                                 int No, int P, const float* g_loss, const void* workspace, float* d_hum, float* d_obj,
                                 chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!hum || !obj || !label_h || !g_loss || !workspace || (!d_hum && !d_obj))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_contact_bwd: null argument");
    if (B <= 0 || Nh <= 0 || No <= 0 || P <= 0 || P > CP_MAX)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_contact_bwd: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    CWs w = contact_ws(const_cast<void*>(workspace), B, Nh, No, P);
    const int th = (Nh + TILE - 1) / TILE, to = (No + TILE - 1) / TILE;
    if (d_hum) {        // (NULL: that cloud is a constant of the caller, e.g. the body in optimize_smpl_object)
        hipLaunchKernelGGL(contact_bwd_kernel, dim3(th, to, B), dim3(TILE), 0, s, hum, Nh, 0, obj, w.nn_o, w.label_o, No, No, P,
                           w.n_part, w.npairs, g_loss, w.part_h);
        hipLaunchKernelGGL(contact_bwd_finish_kernel, dim3(th, B), dim3(TILE), 0, s, hum, w.nn_h, label_h, 0, Nh, 0, obj, No, P,
                           w.n_part, w.npairs, g_loss, w.part_h, to, d_hum);
    }
    if (d_obj) {
        hipLaunchKernelGGL(contact_bwd_kernel, dim3(to, th, B), dim3(TILE), 0, s, obj, No, 1, hum, w.nn_h, label_h, 0, Nh, P,
                           w.n_part, w.npairs, g_loss, w.part_o);
        hipLaunchKernelGGL(contact_bwd_finish_kernel, dim3(to, B), dim3(TILE), 0, s, obj, w.nn_o, w.label_o, No, No, 1, hum, Nh, P,
                           w.n_part, w.npairs, g_loss, w.part_o, th, d_obj);
    }
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

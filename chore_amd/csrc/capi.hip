// capi.hip -- extern "C" entry points of libchore_hip.so (see include/chore_hip.h).
#include "common.h"
#include <cstdlib>
#include <cstring>

int launch_query_fwd_f32_bf16maps(chore_handle* h, const QueryArgs& a, hipStream_t s);
int launch_query_fwd_x3(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s);
int launch_query_bwd_x3(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s);
size_t heads_arena_bytes();
int launch_query_bwd_f32_bf16maps(chore_handle* h, const QueryArgs& a, hipStream_t s);
int launch_sample_features(chore_handle* h, int dtype, const QueryArgs& a, float* features, float* nxy, hipStream_t s);

static thread_local std::string g_noh_err;

static const void* find_desc(const chore_weight_desc* d, int n, const std::string& name, int64_t numel,
                             std::string& err) {
    for (int i = 0; i < n; ++i) {
        if (d[i].name && name == d[i].name) {
            if (d[i].numel != numel) {
                err = "tensor '" + name + "' has " + std::to_string(d[i].numel) + " elements, expected " +
                      std::to_string(numel);
                return nullptr;
            }
            return d[i].ptr;
        }
    }
    err = "tensor '" + name + "' missing from weight descriptors";
    return nullptr;
}

extern "C" {

int chore_version(void) { return 100; }

int chore_create(chore_handle** out, int device_ordinal) {
    if (!out) return CHORE_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_ordinal < 0 || device_ordinal >= count) {
        g_noh_err = "chore_create: no such HIP device";
        return CHORE_EHIP;
    }
    chore_handle* h = new chore_handle();
    h->device = device_ordinal;
    *out = h;
    return CHORE_OK;
}

void chore_encoder_cache_free(chore_handle* h);

int chore_destroy(chore_handle* h) {
    CHORE_ENTER(h);
    chore_encoder_cache_free(h);
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        (void)hipStreamDestroy(h->side);
        for (hipEvent_t e : h->side_ev)
            if (e) (void)hipEventDestroy(e);
    }
    delete h;
    return CHORE_OK;
}

const char* chore_last_error(const chore_handle* h) { return h ? h->err.c_str() : g_noh_err.c_str(); }

// A stream whose kernels run on a SUBSET of the compute units (the hardware queue carries the mask).  A conv_pc / query
// workgroup owns its CU (8 waves x 256 registers, > 80 KB of LDS): two streams time-share CUs, and a chain of small kernels
// beside a stream of such workgroups waits for a CU to drain before each of its launches.  Masking the LARGE stream off a few
// CUs per XCD keeps those free for the other.  Mask bit i = CU (i / n_xcd) of XCD (i % n_xcd) (the driver deals the bits round
// robin over the XCDs), so "the first k bits" = k / 8 CUs of every XCD on this part.
int chore_cu_count(chore_handle* h) {
    if (!h) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, h->device) != hipSuccess) return -1;
    return p.multiProcessorCount;
}
int chore_stream_create_cu_mask(chore_handle* h, const uint32_t* mask, int n_words, chore_stream_t* out) {
    CHORE_ENTER(h);
    if (!mask || n_words <= 0 || !out) CHORE_FAIL(h, CHORE_EINVAL, "chore_stream_create_cu_mask: bad argument");
    bool any = false;
    for (int i = 0; i < n_words; ++i) any |= mask[i] != 0;
    if (!any) CHORE_FAIL(h, CHORE_EINVAL, "chore_stream_create_cu_mask: empty mask");
    hipStream_t s = nullptr;
    CHORE_HIP_CHECK(h, hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
    *out = (chore_stream_t)s;
    return CHORE_OK;
}
int chore_stream_destroy(chore_handle* h, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!stream) CHORE_FAIL(h, CHORE_EINVAL, "chore_stream_destroy: null stream");
    CHORE_HIP_CHECK(h, hipStreamSynchronize((hipStream_t)stream));
    CHORE_HIP_CHECK(h, hipStreamDestroy((hipStream_t)stream));
    return CHORE_OK;
}

size_t chore_heads_arena_bytes(int dtype) {
    (void)dtype;  // one arena for every mode: the fp32 MFMA fragments, then the fp16 x 3 fragments (heads_x3.h)
    return heads_arena_bytes();
}

int chore_heads_pack(chore_handle* h, const chore_weight_desc* descs, int n_descs, int dtype, void* arena,
                     chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!descs || !arena) CHORE_FAIL(h, CHORE_EINVAL, "chore_heads_pack: null argument");
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) CHORE_FAIL(h, CHORE_EINVAL, "chore_heads_pack: bad dtype");
    static const char* names[HEAD_NUM] = {"df", "part_predictor", "pca_predictor", "center_predictor"};
    HeadsRaw raw;
    std::string err;
    for (int hd = 0; hd < HEAD_NUM; ++hd) {
        for (int l = 0; l < 4; ++l) {
            const int in = (l == 0) ? HEAD_IN : HEAD_HID;
            const int out = (l == 3) ? head_out_dim(hd) : HEAD_HID;
            const std::string base = std::string(names[hd]) + "." + std::to_string(2 * l);
            raw.w[hd][l] = (const float*)find_desc(descs, n_descs, base + ".weight", (int64_t)in * out, err);
            if (!raw.w[hd][l]) CHORE_FAIL(h, CHORE_ESTATE, "chore_heads_pack: %s", err.c_str());
            raw.b[hd][l] = (const float*)find_desc(descs, n_descs, base + ".bias", out, err);
            if (!raw.b[hd][l]) CHORE_FAIL(h, CHORE_ESTATE, "chore_heads_pack: %s", err.c_str());
        }
    }
    return launch_heads_pack_f32(h, raw, (float*)arena, (hipStream_t)stream);
}

// ---- debug aid (CHORE_NAN_CHECK=1): count non-finite values in the query's inputs / outputs, per call site --------------------
__device__ unsigned g_nan_counts[32];      // [0..15] counts, [16..31] sequence number of the first scan that saw one (+1)
__global__ void nan_scan_kernel(const float* p, size_t n, int slot, unsigned seq) {
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p && !isfinite(p[i])) ++c;
    if (c) {
        atomicAdd(&g_nan_counts[slot], c);
        atomicCAS(&g_nan_counts[16 + slot], 0u, seq);
    }
}
static bool nan_check_on() { static const bool v = getenv("CHORE_NAN_CHECK") != nullptr; return v; }
static void nan_scan(const float* p, size_t n, int slot, hipStream_t s) {
    static unsigned seq = 0;
    if (p && n) hipLaunchKernelGGL(nan_scan_kernel, dim3(64), dim3(256), 0, s, p, n, slot, ++seq);
}
// ---- CHORE_LDS_POISON (common.h) ----
__global__ __launch_bounds__(1024) void lds_poison_kernel(unsigned pattern) {
    extern __shared__ unsigned lds_poison_mem[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds_poison_mem[i] = pattern;
    __syncthreads();
    if (lds_poison_mem[(threadIdx.x * 37) % (160 * 256)] != pattern) __builtin_trap();   // keeps the stores
}
void chore_lds_poison(hipStream_t s, const char* file, int line) {
    static const unsigned pattern = (unsigned)strtoul(getenv("CHORE_LDS_POISON"), nullptr, 0) ? (unsigned)strtoul(getenv("CHORE_LDS_POISON"), nullptr, 0) : 0x7fc00000u;
    static const char* only = getenv("CHORE_LDS_POISON_FILE");
    static const int lo = getenv("CHORE_LDS_POISON_LINE_LO") ? atoi(getenv("CHORE_LDS_POISON_LINE_LO")) : 0;
    static const int hi = getenv("CHORE_LDS_POISON_LINE_HI") ? atoi(getenv("CHORE_LDS_POISON_LINE_HI")) : 1 << 30;
    if (only && !strstr(file, only)) return;
    if (line < lo || line > hi) return;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(512), dim3(1024), 160 * 1024, s, pattern);      // one 160 KB workgroup per CU, twice over
}

extern "C" int chore_debug_nan_counts(unsigned* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_nan_counts), sizeof(unsigned) * 32) == hipSuccess ? 0 : -2;
}

// split a query dtype into the map type and the heads mode (include/chore_hip.h: CHORE_HEADS_X3)
static inline bool query_x3(int& dtype) {
    const bool x3 = dtype == CHORE_F16X3 || dtype == CHORE_F16 || (dtype & CHORE_HEADS_X3);      // fp16 maps: always with these heads
    dtype = dtype == CHORE_F16X3 ? CHORE_F32 : (dtype & ~CHORE_HEADS_X3);
    return x3;
}

static int fill_query_args(chore_handle* h, QueryArgs& a, const float* points, const float* crop_center, int B,
                           int N, const void* feat, int FH, int FW, const void* tmpx, int TH, int TW,
                           int dtype, const void* arena, const float* cam) {
    if (!points || !crop_center || !feat || !tmpx || !arena || !cam)
        CHORE_FAIL(h, CHORE_EINVAL, "query: null argument");
    if (B <= 0 || N <= 0 || FH < 2 || FW < 2 || TH < 2 || TW < 2) CHORE_FAIL(h, CHORE_EINVAL, "query: bad shape");
    if (B > 65535) CHORE_FAIL(h, CHORE_EINVAL, "query: B > 65535");
    if (dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16) CHORE_FAIL(h, CHORE_EINVAL, "query: bad dtype");
    memset(&a, 0, sizeof(a));
    a.points = points; a.crop_center = crop_center; a.B = B; a.N = N;
    a.feat = feat; a.FH = FH; a.FW = FW; a.tmpx = tmpx; a.TH = TH; a.TW = TW;
    a.arena = arena;
    a.fx = cam[0]; a.fy = cam[1]; a.cx = cam[2]; a.cy = cam[3]; a.half_crop = cam[4]; a.crop = cam[5];
    return CHORE_OK;
}

size_t chore_query_fwd_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * query_sort_ints(N) * sizeof(int);
}

int chore_query_fwd(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                    const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                    const void* heads_arena, const float* cam6_host, float* df, float* pca, float* parts,
                    float* centers, uint8_t* in_img, chore_stream_t stream) {
    return chore_query_fwd_ws(h, points, crop_center, B, N, feat, FH, FW, tmpx, TH, TW, dtype, heads_arena, cam6_host, df, pca, parts,
                              centers, in_img, nullptr, stream);
}

int chore_query_fwd_ws(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                       const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                       const void* heads_arena, const float* cam6_host, float* df, float* pca, float* parts,
                       float* centers, uint8_t* in_img, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!df && !pca && !parts && !centers) CHORE_FAIL(h, CHORE_EINVAL, "chore_query_fwd: no output asked for");
    QueryArgs a;
    // CHORE_F16X3: fp32 feature maps (what the fp16 x 3 encoder writes) and the heads on the fp16 matrix cores
    const bool x3 = query_x3(dtype);
    int rc = fill_query_args(h, a, points, crop_center, B, N, feat, FH, FW, tmpx, TH, TW, dtype, heads_arena, cam6_host);
    if (rc) return rc;
    a.out[0] = df; a.out[1] = parts; a.out[2] = pca; a.out[3] = centers;
    a.in_img = in_img;
    if (nan_check_on()) {
        nan_scan(points, (size_t)B * N * 3, 0, (hipStream_t)stream);
        if (dtype == CHORE_F32) { nan_scan((const float*)feat, (size_t)B * FH * FW * 256, 5, (hipStream_t)stream); nan_scan((const float*)tmpx, (size_t)B * TH * TW * 64, 6, (hipStream_t)stream); }
    }
    // sorted order: asked for by passing a workspace; large queries on the fp16 x 3 split kernel (the only one that reads QueryArgs::perm)
    if (workspace && x3 && N >= 8192 && !getenv("CHORE_QUERY_X3_NOSPLIT") && query_sort_covers(a)) {
        if ((rc = launch_query_sort(h, a, (int*)workspace, (hipStream_t)stream))) return rc;
        a.perm = (const int*)workspace;
    }
    rc = x3 ? launch_query_fwd_x3(h, dtype, a, (hipStream_t)stream)
            : (dtype == CHORE_F32 ? launch_query_fwd_f32(h, a, (hipStream_t)stream) : launch_query_fwd_f32_bf16maps(h, a, (hipStream_t)stream));
    if (nan_check_on()) {
        if (df) nan_scan(df, (size_t)B * 2 * N, 1, (hipStream_t)stream);
        if (pca) nan_scan(pca, (size_t)B * 9 * N, 2, (hipStream_t)stream);
        if (parts) nan_scan(parts, (size_t)B * 14 * N, 3, (hipStream_t)stream);
        if (centers) nan_scan(centers, (size_t)B * 6 * N, 4, (hipStream_t)stream);
    }
    return rc;
}

int chore_sample_features(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                          const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                          const float* cam6_host, float* features, float* nxy, uint8_t* in_img,
                          chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!features) CHORE_FAIL(h, CHORE_EINVAL, "chore_sample_features: null output");
    if (dtype != CHORE_F32 && dtype != CHORE_BF16)   // (also CHORE_F16 | CHORE_HEADS_X3 and CHORE_F16X3: the flag is not a map type here)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_sample_features: maps must be CHORE_F32 or CHORE_BF16 (fp16 maps are an inference mode: chore_query_fwd)");
    QueryArgs a;
    int rc = fill_query_args(h, a, points, crop_center, B, N, feat, FH, FW, tmpx, TH, TW, dtype, feat /*unused*/,
                             cam6_host);
    if (rc) return rc;
    a.in_img = in_img;
    return launch_sample_features(h, dtype, a, features, nxy, (hipStream_t)stream);
}

int chore_query_bwd_points(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                           const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                           const void* heads_arena, const float* cam6_host, const float* g_df,
                           const float* g_pca, const float* g_parts, const float* g_centers, float* dpoints,
                           chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!dpoints) CHORE_FAIL(h, CHORE_EINVAL, "chore_query_bwd_points: null dpoints");
    QueryArgs a;
    const bool x3 = query_x3(dtype);            // the chain on the fp16 matrix cores (see chore_query_fwd)
    int rc = fill_query_args(h, a, points, crop_center, B, N, feat, FH, FW, tmpx, TH, TW, dtype, heads_arena, cam6_host);
    if (rc) return rc;
    a.g[0] = g_df; a.g[1] = g_parts; a.g[2] = g_pca; a.g[3] = g_centers;
    a.dpoints = dpoints;
    if (nan_check_on()) {
        nan_scan(points, (size_t)B * N * 3, 8, (hipStream_t)stream);
        if (g_df) nan_scan(g_df, (size_t)B * 2 * N, 9, (hipStream_t)stream);
        if (g_pca) nan_scan(g_pca, (size_t)B * 9 * N, 10, (hipStream_t)stream);
        if (g_parts) nan_scan(g_parts, (size_t)B * 14 * N, 11, (hipStream_t)stream);
        if (g_centers) nan_scan(g_centers, (size_t)B * 6 * N, 12, (hipStream_t)stream);
        rc = x3 ? launch_query_bwd_x3(h, dtype, a, (hipStream_t)stream)
                : (dtype == CHORE_F32 ? launch_query_bwd_f32(h, a, (hipStream_t)stream) : launch_query_bwd_f32_bf16maps(h, a, (hipStream_t)stream));
        nan_scan(dpoints, (size_t)B * N * 3, 13, (hipStream_t)stream);
        return rc;
    }
    if (x3) return launch_query_bwd_x3(h, dtype, a, (hipStream_t)stream);
    return dtype == CHORE_F32 ? launch_query_bwd_f32(h, a, (hipStream_t)stream)
                              : launch_query_bwd_f32_bf16maps(h, a, (hipStream_t)stream);
}

int chore_gen_surface_step_fused(chore_handle* h, const float* points, const float* crop_center, int B, int N, const void* feat,
                                 int FH, int FW, const void* tmpx, int TH, int TW, int dtype, const void* heads_arena,
                                 const float* cam6_host, int k, float thr, float* out_points, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!out_points || (k != 0 && k != 1)) CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_surface_step_fused: bad argument");
    if (!query_x3(dtype))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_gen_surface_step_fused: needs the fp16 x 3 heads (CHORE_F16X3, or CHORE_HEADS_X3 with the maps' type)");
    QueryArgs a;
    int rc = fill_query_args(h, a, points, crop_center, B, N, feat, FH, FW, tmpx, TH, TW, dtype, heads_arena, cam6_host);
    if (rc) return rc;
    for (int i = 0; i < HEAD_NUM; ++i) { a.g[i] = nullptr; a.out[i] = nullptr; }
    a.in_img = nullptr;
    a.dpoints = out_points;
    a.surf_k = k;
    a.surf_thr = thr;
    return launch_query_surface_step(h, dtype, a, (hipStream_t)stream);
}

size_t chore_query_train_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * N * ((2 * QF_KPAD + 2 * 3 * HEAD_NUM * HEAD_HID) * sizeof(float) + 3 * HEAD_NUM * 2 * sizeof(unsigned long long)) +
           2 * (size_t)B * scatter_sort_ints(N) * sizeof(int);
}

static void train_staging(QueryArgs& a, void* staging) {
    const size_t P = (size_t)a.B * a.N;
    float* f = (float*)staging;
    a.tX = f;
    a.tH = a.tX + P * QF_KPAD;
    a.tdZ = a.tH + P * 3 * HEAD_NUM * HEAD_HID;
    a.tdX = a.tdZ + P * 3 * HEAD_NUM * HEAD_HID;
    a.tM = (unsigned long long*)(a.tdX + P * QF_KPAD);
    a.tSort = (int*)(a.tM + P * 3 * HEAD_NUM * 2);
}

// the query forward of a training step: as chore_query_fwd, and the 323-vectors and ReLU outputs of the hidden layers go
// to `staging` (chore_query_train_bytes) so that chore_query_bwd_train(have_forward = 1) recomputes nothing
int chore_query_fwd_train(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                          const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                          const void* heads_arena, const float* cam6_host, float* df, float* pca, float* parts,
                          float* centers, uint8_t* in_img, void* staging, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!df || !pca || !parts || !centers || !staging) CHORE_FAIL(h, CHORE_EINVAL, "chore_query_fwd_train: null output");
    QueryArgs a;
    const bool x3 = query_x3(dtype);
    // checked AFTER the heads flag is stripped: CHORE_F16 | CHORE_HEADS_X3 must not get past (the launchers below read fp32 or bf16)
    if (dtype != CHORE_F32 && dtype != CHORE_BF16)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_query_fwd_train: maps must be CHORE_F32 or CHORE_BF16 (fp16 maps are an inference mode)");
    int rc = fill_query_args(h, a, points, crop_center, B, N, feat, FH, FW, tmpx, TH, TW, dtype, heads_arena,
                             cam6_host);
    if (rc) return rc;
    a.out[0] = df; a.out[1] = parts; a.out[2] = pca; a.out[3] = centers;
    a.in_img = in_img;
    train_staging(a, staging);
    return launch_query_fwd_train(h, dtype, a, (hipStream_t)stream, x3);
}

int chore_query_bwd_train(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                          const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                          const void* heads_arena, const float* cam6_host, const float* g_df, const float* g_pca,
                          const float* g_parts, const float* g_centers, void* staging, float* dpoints,
                          int have_forward, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!staging) CHORE_FAIL(h, CHORE_EINVAL, "chore_query_bwd_train: null staging");
    QueryArgs a;
    const bool x3 = query_x3(dtype);
    // checked AFTER the heads flag is stripped: CHORE_F16 | CHORE_HEADS_X3 must not get past (the launchers below read fp32 or bf16)
    if (dtype != CHORE_F32 && dtype != CHORE_BF16)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_query_bwd_train: maps must be CHORE_F32 or CHORE_BF16 (fp16 maps are an inference mode)");
    if (x3 && !have_forward) CHORE_FAIL(h, CHORE_EINVAL, "chore_query_bwd_train: the fp16 x 3 heads need the staged forward");
    int rc = fill_query_args(h, a, points, crop_center, B, N, feat, FH, FW, tmpx, TH, TW, dtype, heads_arena,
                             cam6_host);
    if (rc) return rc;
    a.g[0] = g_df; a.g[1] = g_parts; a.g[2] = g_pca; a.g[3] = g_centers;
    a.dpoints = dpoints;
    train_staging(a, staging);
    return launch_query_bwd_train(h, dtype, a, (hipStream_t)stream, have_forward, x3);
}

int chore_scatter_features(chore_handle* h, const float* points, const float* crop_center, int B, int N, int FH, int FW,
                           int TH, int TW, const float* cam6_host, const void* staging, float* dfeat, float* dtmpx,
                           int accumulate, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!staging || (!dfeat && !dtmpx)) CHORE_FAIL(h, CHORE_EINVAL, "chore_scatter_features: null argument");
    QueryArgs a;
    int rc = fill_query_args(h, a, points, crop_center, B, N, staging /*unused*/, FH, FW, staging /*unused*/, TH, TW,
                             CHORE_F32, staging /*unused*/, cam6_host);
    if (rc) return rc;
    train_staging(a, const_cast<void*>(staging));
    return launch_scatter_features(h, a, a.tdX, dfeat, dtmpx, accumulate, (hipStream_t)stream);
}

}  // extern "C"

// common.h -- shared declarations of libchore_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <unordered_map>
#include "../../include/chore_hip.h"

struct chore_handle {
    int device = 0;
    std::string err;
    // cached encoder programs keyed by shape (see encoder.cpp)
    void* enc_cache = nullptr;
    // one-off per-device setup that has been done for THIS handle's device (kernel attributes such as the dynamic LDS
    // limit are per device): keyed by the address of a marker that is unique per call site / template instantiation
    std::unordered_map<const void*, bool> once;
    // training operators: a second stream (and fork / join events) for work that is independent inside one call, e.g. the
    // weight gradients of a ConvBlock beside its data-gradient chain (convblock.hip); created on first use
    hipStream_t side = nullptr;
    hipEvent_t side_ev[8] = {};
    int cu_count = 0;          // hipDeviceAttributeMultiprocessorCount of `device`, read on first use (conv_rw.hip)
    int lds_per_cu = 0;        // hipDeviceAttributeMaxSharedMemoryPerMultiprocessor of `device`, read on first use (conv_pc.hip)
};

// every entry point runs with the handle's device current (a caller whose current device is another GPU -- e.g. the
// reference's model.to(torch.device(opt.gpu_id)) without torch.cuda.set_device -- would otherwise create the encoder's
// auxiliary streams and set kernel attributes on the wrong device) and restores the caller's device on return
struct ChoreDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit ChoreDeviceGuard(const chore_handle* h) {
        if (h && hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~ChoreDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    ChoreDeviceGuard(const ChoreDeviceGuard&) = delete;
    ChoreDeviceGuard& operator=(const ChoreDeviceGuard&) = delete;
};
#define CHORE_ENTER(h)                  \
    if (!(h)) return CHORE_EINVAL;      \
    ChoreDeviceGuard _chore_guard(h)
#define CHORE_ONCE_FLAG(h) ([&]() -> bool& { static char _site; return (h)->once[(const void*)&_site]; }())

#define CHORE_FAIL(h, code, ...)                         \
    do {                                                 \
        char _b[512];                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);           \
        if (h) (h)->err = _b;                            \
        return (code);                                   \
    } while (0)

#define CHORE_HIP_CHECK(h, expr)                                                        \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            CHORE_FAIL(h, CHORE_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                             \
    } while (0)

// after every kernel launch: pick up launch errors; with CHORE_DEBUG_SYNC set also wait for the kernel
// so that a device fault is attributed to the launch that caused it
inline bool chore_debug_sync() {
    static const bool v = getenv("CHORE_DEBUG_SYNC") != nullptr;
    return v;
}
// CHORE_LDS_POISON=<pattern>: after EVERY kernel launch of the library a kernel fills the LDS of every CU with a bit pattern
// (default: quiet NaNs).  LDS keeps what the previous workgroup on the CU left there, so a kernel that reads LDS it has not
// written gives results that depend on what ran before it -- on a GPU shared with another process that is another
// process's data, and runs stop reproducing.  With the poison such a read shows up as a NaN (or a changed result) in a
// single process.  CHORE_LDS_POISON_FILE / _LINE_LO / _LINE_HI restrict the poison to launch sites (bisection).
extern "C" void chore_lds_poison(hipStream_t s, const char* file, int line);
inline bool chore_lds_poison_on() {
    static const bool v = getenv("CHORE_LDS_POISON") != nullptr;
    return v;
}
#define CHORE_LAUNCH_CHECK(h, stream)                                                         \
    do {                                                                                      \
        CHORE_HIP_CHECK(h, hipGetLastError());                                                \
        if (chore_lds_poison_on()) chore_lds_poison(stream, __FILE__, __LINE__);              \
        if (chore_debug_sync()) {                                                             \
            fprintf(stderr, "[chore] launched at %s:%d\n", __FILE__, __LINE__);               \
            CHORE_HIP_CHECK(h, hipStreamSynchronize(stream));                                 \
        }                                                                                     \
    } while (0)

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;  // 8 bf16 in 4 VGPRs
using u16x4 = __attribute__((ext_vector_type(4))) unsigned short;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// bf16 helpers (round-to-nearest-even, like torch's float->bfloat16)
__host__ __device__ inline unsigned short f2bf(float f) {
    union { float f; unsigned int u; } v;
    v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((v.u >> 16) | 0x40);  // NaN
    unsigned int r = 0x7fffu + ((v.u >> 16) & 1u);
    return (unsigned short)((v.u + r) >> 16);
}
// two floats -> packed bf16 pair (x in the low half) with the hardware converter v_cvt_pk_bf16_f32 (same RNE rounding)
typedef __bf16 chore_bf16x2 __attribute__((ext_vector_type(2)));
typedef float chore_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack2bf(float x, float y) {
    const chore_f32x2 f = {x, y};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f, chore_bf16x2));
}
__host__ __device__ inline float bf2f(unsigned short b) {
    union { float f; unsigned int u; } v;
    v.u = ((unsigned int)b) << 16;
    return v.f;
}

// D-fragment row of the 32x32 MFMA family: row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// MODE.FP16_OVFL (bit 23 of the wave's MODE register): with it set, a float -> half conversion that overflows gives +-65504
// instead of +-inf.  The fp16 x 3 kernels split fp32 values into fp16 hi / lo pairs; with the bit set a pair represents every
// |x| <= 131 008 exactly as before (hi saturates, lo takes the rest) and saturates beyond -- without it |x| >= 65 520 turned
// into inf and, one MFMA later, NaN, where the fp32-MFMA path stays finite (scripts/probes/f16_ovfl_probe.hip).  Values in range
// convert exactly as without the bit.  Every wave sets it on entry (MODE is per-wave state).
#ifdef CHORE_NO_F16_SATURATE      // A/B builds only (scripts/build_variant.sh)
__device__ __forceinline__ void f16_saturate_mode() {}
#else
__device__ __forceinline__ void f16_saturate_mode() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
#endif

__host__ __device__ inline int mfma32_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ----- heads (query) constants -----
constexpr int HEAD_NUM = 4;        // df, parts, pca, centers (kernel wave order)
constexpr int HEAD_HID = 128;
constexpr int FEAT_C = 256;        // hourglass_dim
constexpr int TMPX_C = 64;
constexpr int HEAD_IN = FEAT_C + 3 + TMPX_C;  // 323
// fp32 arena layout (floats)
constexpr int QF_KPAD = 328;                  // 323 padded to a multiple of 8
constexpr int QF_KG = QF_KPAD / 8;            // 41 k-groups
constexpr size_t QF_L1_FLOATS = (size_t)HEAD_NUM * QF_KG * 4 * 64 * 4;
constexpr size_t QF_L23_FLOATS = (size_t)HEAD_NUM * 2 * 16 * 4 * 64 * 4;
constexpr size_t QF_L4_FLOATS = (size_t)HEAD_NUM * 16 * 64 * 4;
constexpr size_t QF_BIAS_FLOATS = (size_t)HEAD_NUM * 4 * 4 * 2 * 16;
constexpr size_t QF_OFF_L1 = 0;
constexpr size_t QF_OFF_L23 = QF_OFF_L1 + QF_L1_FLOATS;
constexpr size_t QF_OFF_L4 = QF_OFF_L23 + QF_L23_FLOATS;
constexpr size_t QF_OFF_BIAS = QF_OFF_L4 + QF_L4_FLOATS;
constexpr size_t QF_FWD_FLOATS = QF_OFF_BIAS + QF_BIAS_FLOATS;
// transposed (backward-to-points) fragments, appended to the same arena
constexpr int QB_RB1 = 11;                    // 328 input features -> 11 row blocks of 32 (352)
constexpr size_t QB_L4T_FLOATS = (size_t)HEAD_NUM * 4 * 4 * 64 * 4;
constexpr size_t QB_L32T_FLOATS = (size_t)HEAD_NUM * 2 * 16 * 4 * 64 * 4;
constexpr size_t QB_L1T_FLOATS = (size_t)HEAD_NUM * 16 * QB_RB1 * 64 * 4;
constexpr size_t QB_OFF_L4T = QF_FWD_FLOATS;
constexpr size_t QB_OFF_L32T = QB_OFF_L4T + QB_L4T_FLOATS;
constexpr size_t QB_OFF_L1T = QB_OFF_L32T + QB_L32T_FLOATS;
constexpr size_t QF_TOTAL_FLOATS = QB_OFF_L1T + QB_L1T_FLOATS;

__host__ __device__ inline int head_out_dim(int h) { return h == 0 ? 2 : (h == 1 ? 14 : (h == 2 ? 9 : 6)); }

// raw (reference-layout) weights of the heads, device pointers
struct HeadsRaw {
    const float* w[HEAD_NUM][4];
    const float* b[HEAD_NUM][4];
};

struct QueryArgs {
    const float* points;
    const float* crop_center;
    int B, N;
    const void* feat;
    int FH, FW;
    const void* tmpx;
    int TH, TW;
    const void* arena;
    float fx, fy, cx, cy, half_crop, crop;
    float* out[HEAD_NUM];  // df, parts, pca, centers
    uint8_t* in_img;
    const int* perm = nullptr;   // [B][N] forward in sorted order (chore_query_fwd_ws): tile slot i of image b holds point perm[b][i]
    // backward only
    const float* g[HEAD_NUM];
    float* dpoints;
    // training backward only (chore_query_bwd_train): staging for the parameter gradients, see query_bwd.hip
    float* tX = nullptr;    // [B*N][QF_KPAD]           the 323-vector of every point (zero padded)
    float* tH = nullptr;    // [3][HEAD_NUM][B*N][128]  relu outputs of hidden layers 1..3
    float* tdZ = nullptr;   // [3][HEAD_NUM][B*N][128]  gradients w.r.t. the pre-activations of layers 1..3
    float* tdX = nullptr;   // [B*N][QF_KPAD]           gradient w.r.t. the 323-vector (summed over the heads)
    unsigned long long* tM = nullptr;   // [3][HEAD_NUM][B*N][2]  ReLU sign bits of the hidden layers (heads_f32.h, store_masks)
    int* tSort = nullptr;   // [2][B][scatter_sort_ints(N)]  per-tile point lists of chore_scatter_features (query_scatter.hip)
    // surface step only (chore_gen_surface_step_fused): distance channel and clamp of generator.py:50-79; dpoints = the moved points
    int surf_k = 0;
    float surf_thr = 0.f;
    int one_head = 0;       // backward with exactly one upstream gradient (query_bwd.hip, ONE): the head that has it
};

// ints per image and map of the scatter's binned point lists: up to 64 chunks of CH points (CH = ceil(N / 64) rounded up to a
// multiple of 64), each with room for 4 insertions per point, and 64 offset tables of SCATTER_OFF_STRIDE entries
constexpr int SCATTER_OFF_STRIDE = 260;     // 256 tiles + the total, padded
__host__ __device__ static inline size_t scatter_sort_ints(int N) { return 4 * ((size_t)N + 64 * 64) + 64 * (size_t)SCATTER_OFF_STRIDE; }

// launchers implemented in the .hip files
int launch_heads_pack_f32(chore_handle* h, const HeadsRaw& raw, float* arena, hipStream_t s);
int launch_query_fwd_f32(chore_handle* h, const QueryArgs& a, hipStream_t s);
int launch_query_bwd_f32(chore_handle* h, const QueryArgs& a, hipStream_t s);
int launch_query_surface_step(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s);
int launch_query_bwd_train(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s, int staged = 0, int x3 = 0);
int launch_query_fwd_train(chore_handle* h, int dtype, const QueryArgs& a, hipStream_t s, int x3 = 0);
int launch_scatter_features(chore_handle* h, const QueryArgs& a, const float* dX, float* dfeat, float* dtmpx,
                            int accumulate, hipStream_t s);
// the points of every image ordered by the 8 x 8-texel tile of the feature map their sample falls into: perm [B][N] at the
// start of `work` (B x query_sort_ints(N) ints in all: the permutations, then the images' sort regions); false = shape not covered
__host__ __device__ static inline size_t query_sort_ints(int N) { return (size_t)N + scatter_sort_ints(N); }
bool query_sort_covers(const QueryArgs& a);
int launch_query_sort(chore_handle* h, const QueryArgs& a, int* work, hipStream_t s);

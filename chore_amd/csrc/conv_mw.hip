// conv_mw.hip -- 3x3 convolution of the encoder (every conv of ConvBlock, model/net_util.py:346-396) as an implicit GEMM whose
// STAGING WORK RIDES IN THE INSTRUCTION STREAM OF THE WAVES THAT ISSUE THE MFMAs (round 6).  Same arithmetic, same ConvArgs
// contract, same packed weights and the same tile shapes as conv_pc_kernel (conv_pc.hip); what differs is who does the staging.
//
// Why.  conv_pc_kernel gives every SIMD a consumer wave (LDS fragments -> MFMAs) and a producer wave (global -> GroupNorm + ReLU
// -> fp16 hi / lo split -> LDS, weight ring).  profiles/r05_mfma_issue_probe.txt: a wave that streams MFMAs keeps its SIMD's issue
// to itself -- the producer advances only while the consumer is stalled, so the K loop runs at 57-60 cycles per MFMA (32 of MFMA +
// what the producer needs), and only instructions of the MFMA-issuing wave ITSELF slip into the matrix pipe's shadow (two vector
// instructions per MFMA: +2 cycles).  Here a workgroup is FOUR waves, one per SIMD, up to 512 registers each, and every wave
//   * reads its A / B fragments from LDS and issues the MFMAs of its (MB x NBW) block of the tile, and between them
//   * stages its share of the NEXT chunk's patch (global loads issued UP K-steps earlier -> GroupNorm + ReLU -> hi / lo -> the other
//     patch buffer) and of a later K-step's weights (global -> registers -> ring slot), a slice per k-step, the issue order pinned
//     with sched_group_barrier (1 MFMA, 1 LDS read, a few vector instructions, ...).
// There are no producer waves, no progress counts and no polling: one s_barrier per K-step orders the LDS hand-over.
//   ring with >= 3 slots (single-tap K-steps): the weights of K-step s + NSLOT - 1 are written during K-step s, so K-step s + 1's
//     operands are complete one barrier EARLY and its first fragments are requested before the barrier (no exposed LDS round trip);
//   ring with 2 slots (whole kernel rows / whole chunks per K-step): K-step s + 1's weights are written during K-step s and its first
//     fragments are requested after the barrier.
// The code of a K-step is branch-free (one scheduling region per k-step): slots past the end of the task list land in a spare
// patch row, indices past the last K-step are clamped and their loads never used, pixels outside the image are zeroed by select.
#include "conv_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

using namespace conv_detail;

// kernel experiments (scripts/build_variant.sh mwN conv_mw.hip -DMW_DBG=N): 1 no weight staging, 2 no patch staging, 4 no MFMAs,
// 8 no fragment reads, 16 no barriers inside the K loop, 32 no epilogue -- results are wrong when set
#ifndef MW_DBG
#define MW_DBG 0
#endif

// phase time stamps (scripts/build_variant.sh stamps conv_mw.hip -DMW_STAMPS=1; scripts/conv_mw_stamps.py): thread 0 of every workgroup
#ifdef MW_STAMPS
__device__ unsigned long long g_mw_stamps[2048 * 8];
#define MWSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_mw_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
extern "C" int chore_debug_mw_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mw_stamps), sizeof(unsigned long long) * (n < 2048 * 8 ? n : 2048 * 8));
}
#else
#define MWSTAMP(i) do { } while (0)
#endif

namespace {

constexpr int PTW = 32;          // tile width in pixels (one MFMA pixel block)
constexpr int MWT = 256;         // threads per workgroup: four waves, one per SIMD

template <int TAPS, int TH_, int NT_, int TPS_, int NSLOT_, bool X3_, int WPL_> struct MGeo {
    static constexpr int TH = TH_, NT = NT_, TPS = TPS_, NSLOT = NSLOT_;
    static constexpr int PAD = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = PTW + 2 * PAD, PH = TH + 2 * PAD, ROWS = PH * PW;
    static constexpr int RB = X3_ ? 144 : 80;                       // LDS patch row: 64 B hi [+ 64 B lo] + 16 B pad
    static constexpr int PATCHB = (ROWS + 1) * RB;                  // + one spare row: staging slots past the last patch row
    static constexpr int KROWS = TAPS / TPS;                        // K-steps per 32-channel chunk
    static constexpr int NB = NT / 32;
    static constexpr int SB1 = TPS * KGC * NB * 1024;               // bytes of one operand plane of a K-step's weights
    static constexpr int SBYTES = WPL_ * SB1;
    // channel blocks per wave: two where a wave then still has a pixel block of its own (a two-row tile of 64 channels splits into
    // 2 x 2 single blocks)
    static constexpr int NBW = NT >= 128 ? 2 : (NT == 64 ? (TH_ >= 4 ? 2 : 1) : 1);
    static constexpr int WAVES_N = NB / NBW, WAVES_M = 4 / WAVES_N, MB = TH / WAVES_M;
    static constexpr int SCR_LD = NT + 4;                           // epilogue image: floats per pixel
    static constexpr int G8 = NT / 8;                               // 8-channel groups per pixel
    static constexpr int NU = TH * PTW * G8 / MWT;                  // (pixel, 8 channels) units per thread in the epilogue
    static constexpr bool CROSS = NSLOT >= 3 && KROWS > 1;          // next K-step's first fragments requested before the barrier
    static constexpr size_t main_bytes(int Cin) { return (size_t)2 * PATCHB + (size_t)NSLOT * SBYTES + (size_t)Cin * 8 + 16; }
    static constexpr size_t epi_bytes() {
        return (size_t)TH * PTW * SCR_LD * 4 > (size_t)(MWT * 32 + 4 * NT) * 4 ? (size_t)TH * PTW * SCR_LD * 4 : (size_t)(MWT * 32 + 4 * NT) * 4;
    }
    static size_t smem_bytes(int Cin) { return main_bytes(Cin) > epi_bytes() ? main_bytes(Cin) : epi_bytes(); }
    static_assert(MB >= 1 && WAVES_M * MB == TH, "tile rows must divide over the waves");
    static_assert(NU >= 1, "epilogue units");
    static_assert((SBYTES / 16) % MWT == 0, "a K-step's weights: whole vectors per thread");
};

// The K-step (as k + KROWS * wraps) that uses the register set of K-step k's task j next: the cyclically next K-step at a multiple
// of UP that has a task at position j (K-step k itself, one chunk later, at the latest)
__host__ __device__ constexpr int mw_next_use(int k, int j, int UP, int KROWS, int RPS, int NVP) {
    for (int d = UP; d < KROWS; d += UP)
        if (((k + d) % KROWS) * RPS + j < NVP) return k + d;
    return k + KROWS;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {      // one v_cvt_pk_f16_f32 (round to nearest even)
    const f16x2_t h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}
// y - (float)(half HI of the packed pair h): v_fma_mix_f32 reads the half in place -- one instruction for the cvt_f32_f16 + sub pair,
// the same value (the product with 1.0 is exact, the sum is rounded once)
template <int HI> __device__ __forceinline__ float sub_half(float y, unsigned h) {
    float r;
    if constexpr (HI) asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y));
    else asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y));
    return r;
}

__device__ __forceinline__ void wg_barrier_mw() {
    // LDS traffic of this wave done, then the workgroup barrier; vector-memory loads stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// GN: GroupNorm + ReLU fused into the staging (ConvArgs::in_st); SC (fp16 x 3 only): the input is a gradient whose range comes in
// ConvArgs::in_amax (training's data-gradient convolutions)
template <typename T, int TAPS, int TH_, int NT_, int TPS_, int NSLOT_, bool GN, bool SC>
__global__ __launch_bounds__(MWT, 1) void conv_mw_kernel(ConvArgs a) {
    if constexpr (IS_X3<T> || IS_H16<T>) f16_saturate_mode();     // the fp16 x 3 / fp16 operand split never produces inf (common.h)
    constexpr bool BF = std::is_same<T, bf16_t>::value;
    static_assert(IS_X3<T> || IS_H16<T> || BF, "conv_mw_kernel: fp16 x 3, fp16 or bf16 operands");
    constexpr bool X3 = IS_X3<T>;
    constexpr int WPL = BF ? 1 : 2;                     // weight planes
    using ST = typename std::conditional<X3, float, unsigned short>::type;     // element type in memory
    constexpr int LVI = X3 ? 2 : 1;                     // 16-byte loads per 8 channels
    using G = MGeo<TAPS, TH_, NT_, TPS_, NSLOT_, X3, WPL>;
    constexpr int NSLOT = G::NSLOT;
    constexpr int TH = G::TH, NT = G::NT, TPS = G::TPS, PAD = G::PAD, PW = G::PW, ROWS = G::ROWS, RB = G::RB;
    constexpr int PATCHB = G::PATCHB, KROWS = G::KROWS, SB1 = G::SB1, SBYTES = G::SBYTES;
    constexpr int NBW = G::NBW, WAVES_N = G::WAVES_N, MB = G::MB, SCR_LD = G::SCR_LD, G8 = G::G8, NU = G::NU;
    constexpr bool CROSS = G::CROSS;
    constexpr int KGE = 16, CC = 32;                    // channels per k-group / per chunk
    constexpr int NTASK = ROWS * 4;                     // staging tasks per chunk: (patch row, 8 channels)
    constexpr int NVP = (NTASK + MWT - 1) / MWT;        // tasks per thread per chunk
    // tasks per thread per K-step: the last K-step of a chunk stages nothing, so that the chunk's patch is complete one barrier early
    constexpr int RPS = KROWS > 1 ? (NVP + KROWS - 2) / (KROWS > 1 ? KROWS - 1 : 1) : NVP;
    constexpr int SVEC = SBYTES / 16, SBV = SVEC / MWT, SV1 = SB1 / 16;
    constexpr int NKS = TPS * KGC;                      // MFMA k-steps per K-step
    // prefetch distances in K-steps (register sets): patch loads come from HBM / MALL, weights from the L2
    constexpr int UP = KROWS == 9 ? 3 : KROWS;          // divides KROWS: the set of a K-step is static
    static_assert(KROWS % UP == 0 && NKS % 2 == 0, "register-set rotation / fragment double buffer");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                                  // [2][ROWS + 1][RB]
    char* bst = smem + 2 * PATCHB;                       // [NSLOT][SBYTES]
    float* ss_lds = (float*)(bst + NSLOT * SBYTES);      // [Cin][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    MWSTAMP(0);
    float in_mul = 1.f, in_inv = 1.f;                    // operand scale of a gradient input (training's data-gradient convolutions)
    if constexpr (X3 && SC) x3_in_scale(a.in_amax, in_mul, in_inv);
    const int wn = wid % WAVES_N, wm = wid / WAVES_N;
    const int tiles_x = (a.W + PTW - 1) / PTW;
    // XCD-aware placement (as conv_pc_kernel): every XCD takes a contiguous range of (image, pixel tile, channel tile)
    const int ntn = a.Cout / NT, tiles = tiles_x * ((a.H + TH - 1) / TH);
    int lid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = lid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
    }
    const int n_tile = lid % ntn, tileb = lid / ntn, tile = tileb % tiles, b = tileb / tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * PTW;
    const int Cin = a.in.C;
    const int NKG = Cin / KGE, NB = a.Cout / 32;
    const int NCH = Cin / CC;
    const ST* in_b = (const ST*)a.in.p + (size_t)b * a.H * a.W * a.in.cs + a.in.co;

    const int crot = (tile * 5 + n_tile * 3) % NCH;
    auto chunk_of = [&](int ci) -> int { int x = ci + crot; return x >= NCH ? x - NCH : x; };

    // ---- this thread's staging tasks: task m = (patch row (tid >> 2) + 64 m, channels 8 (tid & 3) .. + 7 of the chunk): the same
    //      rows for every chunk, so their offsets and LDS rows stay in registers
    const int tv = tid & 3;
    int toff[NVP];                                       // element offset of the row's pixel, -1: outside the image / past the list
    int trow[NVP];                                       // LDS row (the spare row for slots past the list and pixels outside the image)
#pragma unroll
    for (int m = 0; m < NVP; ++m) {
        const int row = (tid >> 2) + (MWT / 4) * m;
        const int y = ty0 + row / PW - PAD, x = tx0 + row % PW - PAD;
        const bool ok = (row < ROWS) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        toff[m] = ok ? (y * a.W + x) * a.in.cs : -1;
        trow[m] = ok ? row : ROWS;       // zero padding: the cells of pixels outside the image are cleared once (prologue) and never written
    }
    auto load_task = [&](u32x4 (&r)[LVI], int off, int c0) {
        const u32x4* p = (const u32x4*)(in_b + (off >= 0 ? off : 0) + c0 + tv * 8);
#pragma unroll
        for (int k = 0; k < LVI; ++k) r[k] = p[k];
    };
    // ss_lds: per channel pair {scale 2p, scale 2p + 1, shift 2p, shift 2p + 1}: register pairs for v_pk_fma_f32
    auto load_ss = [&](f32x2 (&sc)[4], f32x2 (&sh)[4], int c0) {
        if constexpr (GN) {
            const f32x4* q = (const f32x4*)(ss_lds + (c0 + tv * 8) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t = q[j];
                sc[j] = f32x2{t[0], t[1]};
                sh[j] = f32x2{t[2], t[3]};
            }
        }
    };
    // one staging task: 8 channels of one patch row -> GroupNorm + ReLU -> fp16 hi / lo (or the 16-bit type) -> LDS.  Branch-free.
    auto put_task = [&](const u32x4 (&r)[LVI], int off, const f32x2 (&sc)[4], const f32x2 (&sh)[4], int row, int pbuf) {
        const bool ok = off >= 0;
        char* d = patch + pbuf * PATCHB + row * RB + tv * 16;
        if constexpr (X3) {
            u32x4 oh, ol;
#pragma unroll
            for (int j = 0; j < 4; ++j) {        // channels 2 j, 2 j + 1
                f32x2 t = {__uint_as_float(r[j >> 1][2 * (j & 1)]), __uint_as_float(r[j >> 1][2 * (j & 1) + 1])};
                if constexpr (GN) {
                    t = __builtin_elementwise_fma(t, sc[j], sh[j]);     // v_pk_fma_f32: x * scale + shift, one rounding (as fmaf)
                    t[0] = t[0] > 0.f ? t[0] : 0.f;
                    t[1] = t[1] > 0.f ? t[1] : 0.f;
                } else if constexpr (SC) {
                    t *= in_mul;
                }
                oh[j] = cvt_pk_f16(t[0], t[1]);
                ol[j] = cvt_pk_f16(sub_half<0>(t[0], oh[j]), sub_half<1>(t[1], oh[j]));
            }
            *(u32x4*)d = oh;
            *(u32x4*)(d + 64) = ol;
            (void)ok;
        } else {
            u32x4 hi = r[0];
            if constexpr (GN) {   // relu(x * scale + shift) in fp32, rounded to the 16-bit type once
                if constexpr (BF) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x0 = fmaf(__uint_as_float(r[0][j] << 16), sc[j][0], sh[j][0]);
                        float x1 = fmaf(__uint_as_float(r[0][j] & 0xffff0000u), sc[j][1], sh[j][1]);
                        x0 = x0 > 0.f ? x0 : 0.f;
                        x1 = x1 > 0.f ? x1 : 0.f;
                        hi[j] = pack2bf(x0, x1);
                    }
                } else {
                    const f16x8_t x = __builtin_bit_cast(f16x8_t, r[0]);
                    f16x8_t y;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = fmaf((float)x[j], sc[j >> 1][j & 1], sh[j >> 1][j & 1]);
                        y[j] = (_Float16)(t > 0.f ? t : 0.f);
                    }
                    hi = __builtin_bit_cast(u32x4, y);
                }
            }
            *(u32x4*)d = hi;
            (void)ok;
        }
    };

    // ---- weights of K-step (chunk c, kernel row krow): [plane][t][kg][nb][lane] vectors; this thread's SBV vectors
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)(n_tile * (NT / 32)) * 64;
    const size_t wkg = (size_t)NB * 64;   // vectors between consecutive k-groups
    int woff[SBV];
#pragma unroll
    for (int j = 0; j < SBV; ++j) {
        const int i0 = tid + j * MWT;
        const int i = i0 % SV1;
        constexpr int PER_KG = (NT / 32) * 64;
        const int t = i / (KGC * PER_KG), kg = (i / PER_KG) % KGC, r = i % PER_KG;
        woff[j] = (t * NKG + kg) * (int)wkg + r;
        if (i0 >= SV1) woff[j] += TAPS * NKG * (int)wkg;   // the lo plane follows the complete hi plane in memory
    }
    auto w_of = [&](int c, int krow) -> const u32x4* {     // wave-uniform
        return wbase + (size_t)(krow * TPS * NKG + chunk_of(c) * KGC) * wkg;
    };

    // ---------------- prologue: chunk 0's patch and the first NSLOT - 1 K-steps of weights ----------------
    {
        // The statistics of this thread's channel FIRST (Cin <= 256 = one channel per thread): they are L2 hits the GroupNorm table
        // waits for, and the vector-memory counter retires in order -- requested after the patch loads they would be waited for
        // behind twelve HBM round trips (the prologue is 4 - 6 us of every launch: profiles/r06_conv_mw_stamps.txt)
        unsigned long long st_lo[2] = {0ull, 0ull};
        long long st_hi[2] = {0ll, 0ll};
        float gam = 1.f, bet = 0.f;
        const int gs_in = Cin / GN_GROUPS;
        if constexpr (GN) {
            if (tid < Cin) {
                const GroupStat* g = a.in_st + (size_t)b * GN_GROUPS + tid / gs_in;
                const size_t hc = act_hi_cells(a.B);
                st_lo[0] = g->sum.lo; st_hi[0] = (&g->sum)[hc].hi;
                st_lo[1] = g->sq.lo;  st_hi[1] = (&g->sq)[hc].hi;
                gam = a.gamma[tid]; bet = a.beta[tid];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 p0[NVP][LVI];
        const int c00 = chunk_of(0) * CC;
#pragma unroll
        for (int m = 0; m < NVP; ++m) load_task(p0[m], toff[m], c00);
        constexpr int NPRO = NSLOT - 1;
        u32x4 wpro[NPRO][SBV];
#pragma unroll
        for (int u = 0; u < NPRO; ++u) {
            const int uc = u / KROWS < NCH ? u / KROWS : NCH - 1;
            const u32x4* wb = w_of(uc, u % KROWS);
#pragma unroll
            for (int j = 0; j < SBV; ++j) wpro[u][j] = wb[woff[j]];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tid < Cin) {       // the arithmetic of gn_scale_shift / stat_read (enc_common.h) on the values requested above
            float sc = 1.f, sh = 0.f;
            if constexpr (GN) {
                double t[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const long long top = st_hi[k] + (long long)(st_lo[k] >> 32);
                    t[k] = ((double)top * 0x1p32 + (double)(st_lo[k] & 0xffffffffull)) * 0x1p-40;
                }
                const double n = (double)(a.H * a.W) * gs_in;
                const double mean = t[0] / n;
                double var = t[1] / n - mean * mean;
                if (var < 0.0) var = 0.0;
                const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
                sc = rstd * gam;
                sh = bet - (float)mean * sc;
            }
            ss_lds[4 * (tid >> 1) + (tid & 1)] = sc;
            ss_lds[4 * (tid >> 1) + 2 + (tid & 1)] = sh;
        }
        // zero padding: the patch cells of pixels outside the image, both buffers, once
#pragma unroll
        for (int m = 0; m < NVP; ++m) {
            const int row = (tid >> 2) + (MWT / 4) * m;
            if (toff[m] < 0 && row < ROWS) {
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int pbuf = 0; pbuf < 2; ++pbuf) {
                    char* d = patch + pbuf * PATCHB + row * RB + tv * 16;
                    *(u32x4*)d = z;
                    if constexpr (X3) *(u32x4*)(d + 64) = z;
                }
            }
        }
        wg_barrier_mw();
        f32x2 sc[4], sh[4];
        load_ss(sc, sh, c00);
#pragma unroll
        for (int m = 0; m < NVP; ++m) put_task(p0[m], toff[m], sc, sh, trow[m], 0);
#pragma unroll
        for (int u = 0; u < NPRO; ++u)
#pragma unroll
            for (int j = 0; j < SBV; ++j) *(u32x4*)(bst + u * SBYTES + (tid + j * MWT) * 16) = wpro[u][j];
    }

    // epilogue coordinates (needed early: the residual rows are requested before the main loop ends)
    const float ASCALE = BF ? 1.0f : (SC ? in_inv / (float)(1 << X3_WSHIFT) : 1.0f / (float)(1 << X3_WSHIFT));
    const int g8 = tid % G8;
    const int nv = n_tile * NT + g8 * 8;                        // this thread's 8 channels
    const size_t img = (size_t)b * a.H * a.W;
    const ST* res_p = a.res.p ? (const ST*)a.res.p + img * a.res.cs + a.res.co + nv : nullptr;

    u32x4 rq[NU][LVI];
    auto fetch_res = [&]() {                                    // the residual rows: an input of the launch, requested when the last chunk starts
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int p = (tid + MWT * j) / G8;
            const int y = ty0 + p / PTW, x = tx0 + p % PTW;
            const bool ok = (y < a.H) && (x < a.W);
            const size_t pix = (size_t)y * a.W + x;
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < LVI; ++k) rq[j][k] = (res_p && ok) ? *((const u32x4*)(res_p + pix * a.res.cs) + k) : z;
        }
    };

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < NBW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    // ---------------- main loop ----------------
    // register sets of the prefetched loads: pset[k % UP] holds the tasks K-step (c, k) puts (chunk c + 1's patch), wset the weights
    // K-step s writes (those of K-step s + NSLOT - 1)
    u32x4 pset[UP][RPS][LVI], wset[SBV];
    auto task_of = [&](int k, int j) -> int { return k * RPS + j; };   // index into toff / trow (static after unrolling)
#pragma unroll
    for (int k = 0; k < UP; ++k)
#pragma unroll
        for (int j = 0; j < RPS; ++j)
            if (task_of(k, j) < NVP) load_task(pset[k][j], toff[task_of(k, j)], chunk_of(NCH > 1 ? 1 : 0) * CC);
    {
        constexpr int u = NSLOT - 1;
        const int uc = u / KROWS < NCH ? u / KROWS : NCH - 1;
        const u32x4* wb = w_of(uc, u % KROWS);
#pragma unroll
        for (int j = 0; j < SBV; ++j) wset[j] = wb[woff[j]];
    }
    const int half = lane >> 5, px = lane & 31;
    const char* a_ptr = patch + ((wm * MB) * PW + px) * RB + 16 * half;
    const char* b_ptr = bst + (wn * NBW) * 1024 + lane * 16;
    u32x4 af[2][MB], afl[2][MB], bf[2][NBW], bfl[2][NBW];
    // fragment loads of k-step ks of K-step (patch buffer pbuf, kernel row krow, ring slot) in the order the MFMAs want them
    auto load_frag = [&](int fs, int pbuf, int krow, int slot, int ks) {
        const char* bs = b_ptr + slot * SBYTES;
        const char* ar = a_ptr + pbuf * PATCHB + ((TPS == 3) ? (krow * PW) * RB : ((TPS == 1 && TAPS == 9) ? ((krow / 3) * PW + krow % 3) * RB : 0));
        const int t = ks / KGC, kg = ks % KGC;
        const int ky = (TPS == 9) ? t / 3 : 0, kx = (TPS == 9) ? t % 3 : t;
        if constexpr ((MW_DBG & 8) != 0) return;
#pragma unroll
        for (int q = 0; q < NBW; ++q) bf[fs][q] = *(const u32x4*)(bs + ((t * KGC + kg) * (NT / 32) + q) * 1024);
        if constexpr (X3) {
#pragma unroll
            for (int m = 0; m < MB; ++m) afl[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + 64 + kg * 32);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) af[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + kg * 32);
        if constexpr (WPL == 2) {
#pragma unroll
            for (int q = 0; q < NBW; ++q) bfl[fs][q] = *(const u32x4*)(bs + SB1 + ((t * KGC + kg) * (NT / 32) + q) * 1024);
        }
    };
    constexpr int NRD = (X3 ? 2 : 1) * MB + WPL * NBW, NMF = (X3 ? 3 : WPL) * MB * NBW;     // LDS reads / MFMAs of one k-step
    int slot = 0;
    wg_barrier_mw();   // chunk 0 and the first K-steps are in LDS
    MWSTAMP(1);
    load_frag(0, 0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);

    // one chunk: KROWS K-steps.  STAGE: the next chunk's patch is staged (false for the last chunk); LASTC: the last chunk
    auto chunk_body = [&](int c, auto stage_t, auto last_t) {
        constexpr bool STAGE = decltype(stage_t)::value, LASTC = decltype(last_t)::value;
        const int pb = c & 1;
        f32x2 sc[4], sh[4];
        int c1 = 0, c2 = 0;
        if constexpr (STAGE) {
            c1 = chunk_of(c + 1) * CC;                                   // the chunk being staged
            c2 = chunk_of(c + 2 < NCH ? c + 2 : NCH - 1) * CC;           // the chunk after it (prefetched loads; clamped, unused past the end)
            load_ss(sc, sh, c1);
        }
        if constexpr (LASTC) fetch_res();
#pragma unroll
        for (int k = 0; k < KROWS; ++k) {
            const bool lastk = LASTC && k == KROWS - 1;
            const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
            const int wslot = slot == 0 ? NSLOT - 1 : slot - 1;          // (slot + NSLOT - 1) % NSLOT: free since K-step s - 1
            // the K-step whose weights are loaded now (written at K-step s + 1): s + NSLOT, clamped to the last chunk
            const int kq = k + NSLOT;
            const int cq = c + kq / KROWS < NCH ? c + kq / KROWS : NCH - 1;
            const u32x4* wnext = w_of(cq, kq % KROWS);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bool pre = true;
                if (ks + 1 < NKS) load_frag((ks + 1) & 1, pb, k, slot, ks + 1);
                else if (CROSS && !lastk) load_frag(0, k + 1 < KROWS ? pb : pb ^ 1, (k + 1) % KROWS, nslot, 0);
                else pre = false;
                // ---- this k-step's slice of the staging work ----
#pragma unroll
                for (int j = 0; j < SBV; ++j) {
                    if ((j * NKS) / SBV != ks || (MW_DBG & 1)) continue;
                    *(u32x4*)(bst + wslot * SBYTES + (tid + j * MWT) * 16) = wset[j];
                    wset[j] = wnext[woff[j]];
                }
                if constexpr (STAGE) {
#pragma unroll
                    for (int j = 0; j < RPS; ++j) {
                        if (((2 * j + 1) * NKS) / (2 * RPS) != ks || task_of(k, j) >= NVP || (MW_DBG & 2)) continue;
                        const int m = task_of(k, j);
                        put_task(pset[k % UP][j], toff[m], sc, sh, trow[m], pb ^ 1);
                        // the task this set holds next (at least UP K-steps from now)
                        const int kk = mw_next_use(k, j, UP, KROWS, RPS, NVP);
                        load_task(pset[k % UP][j], toff[task_of(kk % KROWS, j)], kk < KROWS ? c1 : c2);
                    }
                }
                // ---- the MFMAs: the three terms of a product go to the same accumulator in a fixed order (small terms first) ----
                if constexpr ((MW_DBG & 4) != 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                if constexpr (X3) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NBW; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, afl[ks & 1][m]),
                                                                               __builtin_bit_cast(f16x8_t, bf[ks & 1][q]), acc[m][q], 0, 0, 0);
                }
                if constexpr (BF) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NBW; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[ks & 1][m]),
                                                                                __builtin_bit_cast(bf16x8_t, bf[ks & 1][q]), acc[m][q], 0, 0, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NBW; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, af[ks & 1][m]),
                                                                               __builtin_bit_cast(f16x8_t, bfl[ks & 1][q]), acc[m][q], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NBW; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, af[ks & 1][m]),
                                                                               __builtin_bit_cast(f16x8_t, bf[ks & 1][q]), acc[m][q], 0, 0, 0);
                }
                // issue order: one MFMA, one fragment read of the next k-step, then the staging slice's instructions a few per MFMA
                // (two vector instructions per MFMA ride in the matrix pipe's shadow: profiles/r05_mfma_issue_probe.txt)
#pragma unroll
                for (int i = 0; i < NMF; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
                    if (pre && i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // LDS read (fragments first: program order)
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                        // vector ALU
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        // a global load
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                        // an LDS write
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr ((MW_DBG & 16) == 0) wg_barrier_mw();
            if (!CROSS && !lastk) {
                load_frag(0, k + 1 < KROWS ? pb : pb ^ 1, (k + 1) % KROWS, nslot, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = nslot;
        }
    };
    {
        int c = 0;
#pragma unroll 1
        for (; c + 1 < NCH; ++c) chunk_body(c, std::true_type{}, std::false_type{});
        chunk_body(c, std::false_type{}, std::true_type{});
    }
    // (the last K-step's barrier: all fragment reads done -- the patch buffers and the ring are dead)

    // ---------------- epilogue: accumulators -> LDS image of the tile -> all threads store ----------------
    MWSTAMP(2);
    if constexpr ((MW_DBG & 32) != 0) {
        if (a.dbg == 12345) a.bias = (const float*)&acc[0][0];     // (never true: keeps the accumulators alive)
        return;
    }
    float* scr = (float*)smem;                                 // [TH * 32 pixels][SCR_LD]
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
        const int ch = (wn * NBW + q) * 32 + px;
        const float bias = a.bias ? a.bias[n_tile * NT + ch] : 0.f;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scr[((wm * MB + m) * PTW + mfma32_row(r, half)) * SCR_LD + ch] = acc[m][q][r] * ASCALE + bias;
    }
    wg_barrier_mw();
    MWSTAMP(3);

    ST* out_p = (ST*)a.out.p + img * a.out.cs + a.out.co + nv;
    ST* raw_p = a.raw.p ? (ST*)a.raw.p + img * a.raw.cs + a.raw.co + nv : nullptr;
    using ET = typename std::conditional<X3, float, typename std::conditional<BF, bf16_t, h16_t>::type>::type;       // store8 / load8 element tag
    const bool want_stats = a.st_raw || a.st_out;
    // statistics partials as register pairs (v_pk_add_f32 / v_pk_fma_f32): sum and sum of squares of what is stored
    f32x2 sr[4], qr[4], so[4], qo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sr[e] = qr[e] = so[e] = qo[e] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int p = (tid + MWT * j) / G8;                    // pixel of the tile
        const int y = ty0 + p / PTW, x = tx0 + p % PTW;
        float f[8];
        {
            const f32x4 lo = *(const f32x4*)(scr + p * SCR_LD + g8 * 8), hi = *(const f32x4*)(scr + p * SCR_LD + g8 * 8 + 4);
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        }
        if (y < a.H && x < a.W) {
            const size_t pix = (size_t)y * a.W + x;
            if (raw_p) {
                float g[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = f[e];
                store8<ET>((ET*)(raw_p + pix * a.raw.cs), g);
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2 v = {g[2 * e], g[2 * e + 1]};
                        sr[e] += v;
                        qr[e] = __builtin_elementwise_fma(v, v, qr[e]);
                    }
                }
            }
            if (res_p) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (X3) { f[k] += __uint_as_float(rq[j][0][k]); f[4 + k] += __uint_as_float(rq[j][LVI - 1][k]); }
                    else if constexpr (BF) {
                        f[2 * k] += __uint_as_float(rq[j][0][k] << 16); f[2 * k + 1] += __uint_as_float(rq[j][0][k] & 0xffff0000u);
                    } else {
                        const f16x8_t rh = __builtin_bit_cast(f16x8_t, rq[j][0]);
                        f[2 * k] += (float)rh[2 * k]; f[2 * k + 1] += (float)rh[2 * k + 1];
                    }
                }
            }
            store8<ET>((ET*)(out_p + pix * a.out.cs), f);
            if (want_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 v = {f[2 * e], f[2 * e + 1]};
                    so[e] += v;
                    qo[e] = __builtin_elementwise_fma(v, v, qo[e]);
                }
            }
        }
    }

    MWSTAMP(4);
    if (want_stats) {   // uniform over the grid
        // 256 threads x (4 sums x 8 channels) -> per-channel totals through LDS, in a fixed order: every thread parks its 32 partial
        // sums as a row segment of part[thread / G8][kind * NT + channel]; the columns are added up by all threads (a column per
        // thread and pass), the GroupNorm groups by one wave
        constexpr int RL = 4 * NT, NROW = MWT / G8, NSEG = RL >= MWT ? 1 : MWT / RL, RPSEG = NROW / NSEG;
        static_assert(RPSEG * NSEG == NROW, "statistics reduction geometry");
        wg_barrier_mw();                                       // every thread is done with the tile image
        float* part = (float*)smem;                            // [NROW][RL]
        float* red = part + NROW * RL;                         // [NSEG][RL]
        {
            float* pr = part + (tid / G8) * RL + g8 * 8;
            auto park = [&](float* q, const f32x2 (&v)[4]) {
                *(f32x4*)q = f32x4{v[0][0], v[0][1], v[1][0], v[1][1]};
                *(f32x4*)(q + 4) = f32x4{v[2][0], v[2][1], v[3][0], v[3][1]};
            };
            park(pr, sr); park(pr + NT, qr); park(pr + 2 * NT, so); park(pr + 3 * NT, qo);
        }
        wg_barrier_mw();
        for (int cs = tid; cs < NSEG * RL; cs += MWT) {
            const int col = cs % RL, seg = cs / RL;
            const float* pc = part + (seg * RPSEG) * RL + col;
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RPSEG; ++r) t += pc[r * RL];
            red[seg * RL + col] = t;
        }
        wg_barrier_mw();
        if (tid < NT) {
            // thread = channel; kinds 0 / 1 = sum / sum of squares of `raw`, 2 / 3 of `out`.  All adds of the workgroup leave from ONE
            // wave per 64 channels (per tensor: the group's sum from its first lane, the sum of squares from its second)
            const int cg = n_tile * NT + tid;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                GroupStat* st = k ? a.st_out : a.st_raw;
                if (!st) continue;
                const int gs = (k ? a.st_out_C : a.st_raw_C) / GN_GROUPS, co = k ? a.st_out_co : a.st_raw_co;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int g = 0; g < NSEG; ++g) { t1 += red[g * RL + (2 * k) * NT + tid]; t2 += red[g * RL + (2 * k + 1) * NT + tid]; }
                const float s1 = group_lane_sum(t1, gs), s2 = group_lane_sum(t2, gs);
                GroupStat* o = st + (size_t)b * GN_GROUPS + (co + cg) / gs;
                if (tid % gs == 0) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
        }
    }
    MWSTAMP(5);
}

template <typename T, int TAPS, int TH, int NT, int TPS, int NSLOT, bool GN, bool SC>
int launch_mw_t(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    using G = MGeo<TAPS, TH, NT, TPS, NSLOT, IS_X3<T>, std::is_same<T, bf16_t>::value ? 1 : 2>;
    size_t smem = G::smem_bytes(a.in.C);
    if (h->lds_per_cu <= 0) {
        CHORE_HIP_CHECK(h, hipDeviceGetAttribute(&h->lds_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, h->device));
        if (h->lds_per_cu <= 0) h->lds_per_cu = 160 * 1024;
    }
    if (smem > (size_t)h->lds_per_cu) CHORE_FAIL(h, CHORE_EINVAL, "conv_mw: %zu bytes of LDS, the CU has %d", smem, h->lds_per_cu);
    // The workgroup takes ALL of the CU's LDS, whatever its tiling needs: no other workgroup that uses LDS -- of this kernel or of any
    // other -- shares the CU with it.  Round 6 measured why (scripts/pipe_stress.py, profiles/r06_pipe_stress.txt): with the small
    // tilings (2 x 32 x 64: 64 KB) requesting what they need, or half the CU + 1 KB like conv_pc_kernel, a fit that runs on another
    // stream BESIDE an encoder pass came out with different bits in 10 - 20 % of the batches (the fit's kernels, not the
    // convolution: the encoder's maps were equal); 128 KB: 1 of 128; 160 KB: 0 of 128.  conv_pc_kernel never shared a CU either
    // (8 waves x 256 registers).  What exactly goes wrong when LDS-using workgroups of other kernels sit beside these four
    // 1-wave-per-SIMD MFMA waves is NOT understood; the cost of the exclusion is nil inside the kernel.
    static const size_t lds_min_env = getenv("CHORE_CONV_MW_LDS_MIN") ? (size_t)atol(getenv("CHORE_CONV_MW_LDS_MIN")) : 0;   // experiments
    const size_t lds_min = lds_min_env ? lds_min_env : (size_t)h->lds_per_cu;
    if (smem < lds_min) smem = lds_min;
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)conv_mw_kernel<T, TAPS, TH, NT, TPS, NSLOT, GN, SC>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_per_cu));
        attr = true;
    }
    const int tiles = ((a.W + PTW - 1) / PTW) * ((a.H + TH - 1) / TH);
    dim3 grid(tiles * (a.Cout / NT) * a.B);
    hipLaunchKernelGGL((conv_mw_kernel<T, TAPS, TH, NT, TPS, NSLOT, GN, SC>), grid, dim3(MWT), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // namespace

// Which tilings of conv_pc_plan this kernel takes over: every 3x3 tiling of the fp16 x 3 mode (it measured faster or equal on every
// layer of the encoder, profiles/r06_conv_layer_ab.txt).  CHORE_CONV_MW=0: none (A/B against conv_pc_kernel)
int conv_mw_fill(int asked) {
    static const int env = getenv("CHORE_CONV_MW_FILL") ? atoi(getenv("CHORE_CONV_MW_FILL")) : 0;
    const int v = env > 0 ? env : asked;
    return v > 0 ? v : 256;
}
bool conv_mw_on(int dtype, int taps) {
    static const char* env = getenv("CHORE_CONV_MW");
    if (env && env[0] == '0') return false;
    return dtype == CHORE_F16X3 && taps == 9;
}
// The tiling for a layer: conv_pc_plan's.  CHORE_CONV_MW_TH2=1 (experiment, measured equal: 256 -> 128 at 64^2, B = 4, 41.5 - 42.4 us
// against 40.6 - 42.4, profiles/r06_conv_layer_ab.txt): 128 output channels on maps with fewer than 256 eight-row tiles as
// 2 x 32 pixels x 128 channels (0.4 of the 256-channel patch staged once) instead of four 32-channel workgroups per 8 x 32 pixels
// that each stage the whole patch -- the staging is not what bounds that layer
PcPlan conv_mw_plan(int dtype, int taps, int B, int H, int W, int Cin, int Cout, int fill_asked) {
    PcPlan p = conv_pc_plan(dtype, taps, B, H, W, Cin, Cout);
    static const bool th2 = getenv("CHORE_CONV_MW_TH2") != nullptr;
    if (th2 && p.th && taps == 9 && Cout % 128 == 0 && p.nt < 128 && H % 2 == 0) {
        const long tiles2 = (long)B * (H / 2) * ((W + 31) / 32) * (Cout / 128);
        if (tiles2 >= 256) { p.th = 2; p.nt = 128; p.tps = 1; p.nslot = 3; }
    }
    // A layer too small to give every CU a full tile (the 64^2 and 32^2 maps at B = 4) takes the WIDEST tile that still yields
    // `fill` workgroups (ConvArgs::fill; the inference encoder asks for 128 = half the CUs) instead of the narrowest that yields 256 -- fewer, denser workgroups:
    // the same launch time on half of the CUs, the other half free for the other step in flight (two in flight 4.24 -> 3.99 ms, one
    // at a time 5.03 -> 5.06; 64: 4.03 / 5.47; profiles/r06_conv_fill.txt).  Also takes the 32^2 maps over from conv_small_kernel.
    // CHORE_CONV_MW_FILL=256: a tile per CU as conv_pc_plan has it.
    const int fill = conv_mw_fill(fill_asked);
    if (fill < 256 && taps == 9 && conv_mw_on(dtype, taps) && Cin % 32 == 0 && W % 32 == 0) {
        const long px8 = (long)B * ((H + 7) / 8) * (W / 32);
        if (px8 * (Cout / 32) < 256 || !p.th || (p.th == 4 && p.nt == 32) || (p.th == 8 && p.nt == 32 && px8 * (Cout / 32) < 512)) {
            static const int cand[][4] = {{8, 128, 1, 3}, {4, 128, 1, 3}, {2, 128, 1, 3}, {8, 64, 3, 2}, {4, 64, 3, 2}, {2, 64, 1, 3},
                                          {8, 32, 3, 2}, {4, 32, 9, 2}};
            static const char* skip = getenv("CHORE_CONV_MW_SKIP");      // experiments: "th*1000+nt[,...]" tilings left out
            for (const auto& c : cand) {
                if (Cout % c[1] || H % c[0]) continue;
                if (skip) { char key[16]; snprintf(key, sizeof key, "%d", c[0] * 1000 + c[1]); if (strstr(skip, key)) continue; }
                const long wgs = (long)B * (H / c[0]) * (W / 32) * (Cout / c[1]);
                if (wgs >= fill) { p.th = c[0]; p.nt = c[1]; p.tps = c[2]; p.nslot = c[3]; break; }
            }
        }
    }
    return p;
}
bool conv_mw_has(const PcPlan& p) {
    const int key = (p.th * 1000 + p.nt) * 100 + p.tps * 10 + p.nslot;
    return key == 812813 || key == 806432 || key == 803232 || key == 406432 || key == 403292 || key == 212813 || key == 412813 || key == 206413;
}
bool conv_mw_covers(int dtype, int taps, const PcPlan& p, const ConvArgs& a) {
    if (!conv_mw_on(dtype, taps) || a.res2.p) return false;
    if (a.in_st == nullptr && a.in_amax == nullptr) return false;      // instantiated: GroupNorm-fused forward, scaled data gradient
    if (a.in_st != nullptr && a.in_amax != nullptr) return false;
    return conv_mw_has(p);
}

int launch_conv_mw(chore_handle* h, int dtype, int taps, const PcPlan& p, const ConvArgs& a, hipStream_t s) {
    const int key = (p.th * 1000 + p.nt) * 100 + p.tps * 10 + p.nslot;
#define MW_CASE(TH, NT, TPS, NSLOT) \
    case (TH * 1000 + NT) * 100 + TPS * 10 + NSLOT:                                              \
        return a.in_amax ? launch_mw_t<x3_t, 9, TH, NT, TPS, NSLOT, false, true>(h, a, s)         \
                         : launch_mw_t<x3_t, 9, TH, NT, TPS, NSLOT, true, false>(h, a, s)
    switch (key) {
        MW_CASE(8, 128, 1, 3);
        MW_CASE(4, 128, 1, 3);
        MW_CASE(2, 128, 1, 3);
        MW_CASE(2, 64, 1, 3);
        MW_CASE(8, 64, 3, 2);
        MW_CASE(8, 32, 3, 2);
        MW_CASE(4, 64, 3, 2);
        MW_CASE(4, 32, 9, 2);
    }
#undef MW_CASE
    CHORE_FAIL(h, CHORE_EINVAL, "conv_mw: no kernel for taps=%d th=%d nt=%d tps=%d nslot=%d", taps, p.th, p.nt, p.tps, p.nslot);
}

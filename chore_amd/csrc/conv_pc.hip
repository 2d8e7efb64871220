// conv_pc.hip -- 3x3 / 1x1 convolution of the encoder as an implicit GEMM with SPECIALISED WAVES (every conv of ConvBlock,
// model/net_util.py:346-396, and the 1x1 convs of the stack tail, model/HGFilters.py:128-142,167-183).  Same arithmetic,
// same ConvArgs contract and the same packed weights as conv_lds_kernel; what differs is who does what inside a workgroup.
//
// Why.  conv_lds_kernel runs every phase with all four waves: load -> GroupNorm/ReLU/hi-lo split -> LDS -> barrier -> MFMAs.
// A per-phase breakdown (profiles/r03_conv_phase_breakdown.txt) shows the phases simply add up -- staging 10.7 us + MFMAs
// 14.4 us of a 47 us launch at 128^2, and two co-resident workgroups run in lockstep, so the hardware overlaps nothing --
// and the layers on the 64^2 maps launch 128 workgroups on a 256-CU part.
//
// Here a workgroup is 8 waves, two per SIMD:
//   * waves 0-3 (consumers) only read fragments from LDS and issue MFMAs;
//   * waves 4-7 (producers) only move data: activations global -> registers -> GroupNorm + ReLU (+ fp16 hi / lo split)
//     -> the OTHER patch buffer, the next K-step's weight fragments -> the OTHER ring slot, residual rows -> registers.
//   The matrix pipe and the vector ALU of a SIMD are separate issue ports, so the producer's arithmetic was expected to run beside
//   the consumer's MFMAs.  (Round 5, profiles/r05_mfma_issue_probe.txt: it does NOT -- a wave that streams MFMAs keeps the SIMD's
//   issue to itself and the producer advances only while the consumer is stalled; the K loop's 57-60 cycles per MFMA are the sum of
//   both.  The fix is the staging work inside the MFMA-issuing wave's own instruction stream: DESIGN.md section 8.)
//   One s_barrier per K-step (raw: the producers' global loads stay in flight across it).
//   * tile = TH x 32 pixels x NT channels with TH = 8 or 4: the 4-row tile doubles the workgroup count on the 64^2 maps.
//   * epilogue: the accumulators go through an LDS image of the whole tile, then ALL 512 threads add the residuals
//     (fetched during the last chunk), store 16-byte vectors and reduce the GroupNorm statistics in a fixed order.
#include "conv_common.h"
#include <cstdio>
#include <cstdlib>

#ifndef CHORE_CONV_ABLATE
#define CHORE_CONV_ABLATE 0
#endif
#if CHORE_CONV_ABLATE
#define PDBG(a) ((a).dbg)
#else
#define PDBG(a) 0
#endif

using namespace conv_detail;

namespace {

constexpr int PTW = 32;          // tile width in pixels (one MFMA pixel block)

// TH rows x 32 pixels x NT channels; K-step = TPS taps of one 32-channel chunk
template <int TAPS, int TH_, int NT_, int TPS_, int NSLOT_, bool X3_ = true, int WPL_ = 2> struct PGeo {
    static constexpr int TH = TH_, NT = NT_, TPS = TPS_, NSLOT = NSLOT_;   // NSLOT: K-steps of weights resident in the LDS ring
    static constexpr int PAD = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = PTW + 2 * PAD, PH = TH + 2 * PAD, ROWS = PH * PW;
    static constexpr int RB = X3_ ? 144 : 80;                       // LDS patch row: 64 B hi [+ 64 B lo] + 16 B pad (9 / 5 slots: odd)
    static constexpr int PATCHB = ROWS * RB;
    static constexpr int KROWS = TAPS / TPS;                        // K-steps per chunk
    static constexpr int NB = NT / 32;
    static constexpr int SB1 = TPS * KGC * NB * 1024;               // bytes of one operand plane of a K-step
    static constexpr int SBYTES = WPL_ * SB1;                       // hi plane, then lo plane (bf16: one plane)
    static constexpr int NBW = NT >= 64 ? 2 : 1;                    // channel blocks per consumer wave
    static constexpr int WAVES_N = NB / NBW, WAVES_M = 4 / WAVES_N, MB = TH / WAVES_M;
    static constexpr int SCR_LD = NT + 4;                           // epilogue image: floats per pixel
    static constexpr int G8 = NT / 8;                               // 8-channel groups per pixel
    static constexpr int NU = TH * PTW * G8 / 512;                  // (pixel, 8 channels) units per thread in the epilogue
    static constexpr size_t main_bytes(int Cin) { return (size_t)2 * PATCHB + (size_t)NSLOT * SBYTES + (size_t)Cin * 8 + (size_t)ROWS * 4 + 48; }
    // epilogue: the tile image, then (over it) the statistics partials [512 / G8][4 NT] + [512 / (4 NT)][4 NT] floats
    static constexpr size_t epi_bytes() {
        return (size_t)TH * PTW * SCR_LD * 4 > (size_t)(512 * 32 + 512) * 4 ? (size_t)TH * PTW * SCR_LD * 4 : (size_t)(512 * 32 + 512) * 4;
    }
    static size_t smem_bytes(int Cin) { return main_bytes(Cin) > epi_bytes() ? main_bytes(Cin) : epi_bytes(); }
    static_assert(MB >= 1 && WAVES_M * MB == TH, "tile rows must divide over the consumer waves");
    static_assert(NU >= 1, "epilogue units");
};

// Progress counts in LDS (sem_ready / sem_done in the kernel): one word per wave of a role = the number of K-steps that
// wave has finished; a waiter needs the MINIMUM over the four waves (with a ring deeper than two slots the waves of a role
// drift apart by several K-steps: a sum would let a fast wave vouch for a slow one).
// The counts are read and written through pointers that are LDS pointers BY TYPE (address space 3).  A volatile access through
// a generic pointer is never narrowed to its address space by the compiler: until round 4 every count access was a FLAT
// instruction with `sc0 sc1` -- a round trip through the vector-memory path of the CU, behind the producers' global loads --
// followed by `s_waitcnt vmcnt(0) lgkmcnt(0)`, which stalled the consumers' MFMA stream at every K-step and drained the
// producers' prefetched global loads at every step (profiles/r04_conv_phase_breakdown.txt).
using lds_u32 = __attribute__((address_space(3))) unsigned;
using lds_u32x4 = __attribute__((address_space(3))) u32x4;
__device__ __forceinline__ void sem_signal(unsigned* sem, int wave, unsigned count, int lane) {
    asm volatile("" ::: "memory");   // the LDS traffic before it is issued before it (the LDS keeps the order)
    if (lane == 0) *(volatile lds_u32*)(sem + wave) = count;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned sem_min(const unsigned* sem) {
    const u32x4 v = *(const volatile lds_u32x4*)sem;
    const unsigned a = v[0] < v[1] ? v[0] : v[1], b = v[2] < v[3] ? v[2] : v[3];
    return a < b ? a : b;
}

__device__ __forceinline__ void wg_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier; vector-memory loads stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// SC (fp16 x 3 only): the input is a gradient whose range comes in ConvArgs::in_amax (training's data-gradient convolutions); a
// separate instantiation, so that the inference kernels' register allocation is what it was before the operand scale existed
template <typename T, int TAPS, int TH_, int NT_, int TPS_, int NSLOT_, bool SC = false>
__global__ __launch_bounds__(512, 2) void conv_pc_kernel(ConvArgs a) {
    if constexpr (IS_X3<T> || IS_H16<T>) f16_saturate_mode();     // the fp16 x 3 / fp16 operand split never produces inf (common.h)
    // T = x3_t: fp32 tensors, activations split into fp16 hi + lo while they are staged, three MFMAs per product;
    // T = h16_t ("fp16 fields"): fp16 tensors, one activation plane, two MFMAs per product (a * w_lo, a * w_hi);
    // T = bf16_t (round 5: the bf16 inference mode and the bf16 training forward / data gradient): bf16 tensors, one activation
    //     and one weight plane, one v_mfma_f32_32x32x16_bf16 per product
    constexpr bool BF = std::is_same<T, bf16_t>::value;
    static_assert(IS_X3<T> || IS_H16<T> || BF, "conv_pc_kernel: fp16 x 3, fp16 or bf16 operands");
    constexpr bool X3 = IS_X3<T>;
    constexpr int WPL = BF ? 1 : 2;                     // weight planes
    using ST = typename std::conditional<X3, float, unsigned short>::type;     // element type in memory
    constexpr int LVI = X3 ? 2 : 1;                     // 16-byte loads per 8 channels
    using G = PGeo<TAPS, TH_, NT_, TPS_, NSLOT_, X3, WPL>;
    constexpr int NSLOT = G::NSLOT;
    constexpr int TH = G::TH, NT = G::NT, TPS = G::TPS, PAD = G::PAD, PW = G::PW, ROWS = G::ROWS, RB = G::RB;
    constexpr int PATCHB = G::PATCHB, KROWS = G::KROWS, SB1 = G::SB1, SBYTES = G::SBYTES;
    constexpr int NBW = G::NBW, WAVES_N = G::WAVES_N, MB = G::MB, SCR_LD = G::SCR_LD, G8 = G::G8, NU = G::NU;
    constexpr int KGE = 16, CC = 32;                    // channels per k-group / per chunk
    constexpr int NTASK = ROWS * 4;                     // staging tasks per chunk: (patch row, 8 channels)
    constexpr int NVP0 = (NTASK + 511) / 512;           // tasks per thread when all 512 threads stage (first chunk)
    constexpr int NVP = (NTASK + 255) / 256;            // tasks per producer thread per chunk
    // ... per K-step: the last K-step of a chunk stages nothing, so that a chunk's patch is complete one K-step early
    constexpr int RPS = KROWS > 1 ? (NVP + KROWS - 2) / (KROWS - 1) : NVP;
    constexpr int LASTK = (NVP - 1) / RPS;              // last K-step of a chunk with staging work
    constexpr int SVEC = SBYTES / 16, SBV = (SVEC + 255) / 256, SV1 = SB1 / 16;
    constexpr int NKS = TPS * KGC;                      // MFMA k-steps per K-step

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                                  // [2][ROWS][RB]
    char* bst = smem + 2 * PATCHB;                       // [NSLOT][SBYTES]
    float* ss_lds = (float*)(bst + NSLOT * SBYTES);      // [Cin][2]
    int* rowoff_lds = (int*)(ss_lds + 2 * a.in.C);       // [ROWS] element offset of the patch row's pixel, -1 outside the image
    // producer -> consumer and consumer -> producer event counts of the main loop (one increment per wave and K-step).
    // The LDS executes a CU's requests in order: whoever sees a count sees everything its writer did before it.
    // (the offset is computed as an integer from `smem`: a pointer that went through an integer cast loses its LDS address
    // space, and every count access became a FLAT instruction -- a round trip through the vector-memory path with
    // `s_waitcnt vmcnt(0) lgkmcnt(0)` behind it, which also drained the producers' prefetched global loads at every step)
    const int sem_off = (2 * PATCHB + NSLOT * SBYTES + a.in.C * 8 + ROWS * 4 + 15) & ~15;
    unsigned* sem_ready = (unsigned*)(smem + sem_off);                                      // [4] producer waves
    unsigned* sem_done = sem_ready + 4;                                                     // [4] consumer waves

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float in_mul = 1.f, in_inv = 1.f;                    // operand scale of a gradient input (training's data-gradient convolutions)
    if constexpr (X3 && SC) x3_in_scale(a.in_amax, in_mul, in_inv);
#if CHORE_CONV_ABLATE
    // phase stamps of one consumer and one producer wave of the workgroup in the middle of the grid: shader clock and 100 MHz wall clock
    auto stamp = [&](int i) {
        if (a.dbg_ticks && blockIdx.x == gridDim.x / 2 && (tid == 0 || tid == 256)) {
            a.dbg_ticks[(tid ? 16 : 0) + 2 * i] = __builtin_readcyclecounter();
            a.dbg_ticks[(tid ? 16 : 0) + 2 * i + 1] = wall_clock64();
        }
    };
#else
    auto stamp = [&](int) {};
#endif
    stamp(0);
    const bool producer = wid >= 4;
    const int cw = wid & 3;                              // consumer wave index (producers: unused)
    const int wn = cw % WAVES_N, wm = cw / WAVES_N;
    const int tiles_x = (a.W + PTW - 1) / PTW;
    // XCD-aware placement (as conv_lds_kernel): every XCD takes a contiguous range of (image, pixel tile, channel tile)
    const int ntn = a.Cout / NT, tiles = tiles_x * ((a.H + TH - 1) / TH);
    int lid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = lid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
    }
    const int n_tile = lid % ntn, tileb = lid / ntn, tile = tileb % tiles, b = tileb / tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * PTW;
    const int Cin = a.in.C;
    const bool use_gn = a.in_st != nullptr;
    const int NKG = Cin / KGE, NB = a.Cout / 32;
    const int NCH = Cin / CC;
    const int S = NCH * KROWS;
    const ST* in_b = (const ST*)a.in.p + (size_t)b * a.H * a.W * a.in.cs + a.in.co;

    const int crot = (tile * 5 + n_tile * 3) % NCH;
    auto chunk_of = [&](int ci) -> int { int x = ci + crot; return x >= NCH ? x - NCH : x; };

    auto row_offset = [&](int row) -> int {
        const int y = ty0 + row / PW - PAD, x = tx0 + row % PW - PAD;
        const bool ok = (row < ROWS) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        return ok ? (y * a.W + x) * a.in.cs : -1;
    };
    // one staging task: 8 fp32 channels of one patch row -> GroupNorm + ReLU -> fp16 hi / lo -> LDS
    auto load_task = [&](u32x4 (&r)[LVI], int off, int c0, int v) {
        const u32x4* p = (const u32x4*)(in_b + (off >= 0 ? off : 0) + c0 + v * 8);
#pragma unroll
        for (int k = 0; k < LVI; ++k) r[k] = p[k];
    };
    auto put_task = [&](const u32x4 (&r)[LVI], int off, int c0, int row, int v, int pbuf) {
        float sc[8], sh[8];
        {   // (scale, shift) pairs of the 8 channels: 64 contiguous bytes
            const f32x4* q = (const f32x4*)(ss_lds + (c0 + v * 8) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t = q[j];
                sc[2 * j] = t[0]; sh[2 * j] = t[1]; sc[2 * j + 1] = t[2]; sh[2 * j + 1] = t[3];
            }
        }
        u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
        char* d = patch + pbuf * PATCHB + row * RB + v * 16;
        if constexpr (X3) {
            if (off >= 0) {
                if constexpr (SC) xform_x3(r[0], r[LVI - 1], sc, sh, use_gn, hi, lo, in_mul);
                else xform_x3(r[0], r[LVI - 1], sc, sh, use_gn, hi, lo);
            }
            *(u32x4*)d = hi;
            *(u32x4*)(d + 64) = lo;
        } else {
            if (off >= 0) {
                if (use_gn) {   // relu(x * scale + shift) in fp32, rounded to the 16-bit type once
                    if constexpr (BF) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float x0 = fmaf(__uint_as_float(r[0][j] << 16), sc[2 * j], sh[2 * j]);
                            float x1 = fmaf(__uint_as_float(r[0][j] & 0xffff0000u), sc[2 * j + 1], sh[2 * j + 1]);
                            x0 = x0 > 0.f ? x0 : 0.f;
                            x1 = x1 > 0.f ? x1 : 0.f;
                            hi[j] = pack2bf(x0, x1);
                        }
                    } else {
                        const f16x8_t x = __builtin_bit_cast(f16x8_t, r[0]);
                        f16x8_t y;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float t = fmaf((float)x[j], sc[j], sh[j]);
                            y[j] = (_Float16)(t > 0.f ? t : 0.f);
                        }
                        hi = __builtin_bit_cast(u32x4, y);
                    }
                } else hi = r[0];
            }
            *(u32x4*)d = hi;
        }
    };

    // weight slice of K-step (chunk c, kernel row krow): [plane][t][kg][nb][lane] vectors
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)(n_tile * (NT / 32)) * 64;
    const size_t wkg = (size_t)NB * 64;   // vectors between consecutive k-groups
    const int ptid = tid & 255;
    int woff[SBV];
#pragma unroll
    for (int j = 0; j < SBV; ++j) {
        const int i0 = (ptid + j * 256 < SVEC) ? ptid + j * 256 : SVEC - 1;
        const int i = i0 % SV1;
        constexpr int PER_KG = (NT / 32) * 64;
        const int t = i / (KGC * PER_KG), kg = (i / PER_KG) % KGC, r = i % PER_KG;
        woff[j] = (t * NKG + kg) * (int)wkg + r;
        if (i0 >= SV1) woff[j] += TAPS * NKG * (int)wkg;   // the lo plane follows the complete hi plane in memory
    }
    auto load_w = [&](u32x4 (&rb)[SBV], int s) {
        const int c = chunk_of(s / KROWS), krow = s % KROWS;
        const u32x4* wb = wbase + (size_t)(krow * TPS * NKG + c * KGC) * wkg;   // wave-uniform
#pragma unroll
        for (int j = 0; j < SBV; ++j) rb[j] = wb[woff[j]];
    };
    auto write_w = [&](const u32x4 (&rb)[SBV], int slot) {
#pragma unroll
        for (int j = 0; j < SBV; ++j) {
            const int i = ptid + j * 256;
            if (i < SVEC) *(u32x4*)(bst + slot * SBYTES + i * 16) = rb[j];
        }
    };

    // ---------------- prologue: everybody stages chunk 0; producers fetch the first weights ----------------
    u32x4 p0[NVP0][LVI];
    int off0[NVP0];
#pragma unroll
    for (int j = 0; j < NVP0; ++j) {
        const int i = tid + j * 512;
        off0[j] = row_offset(i >> 2);
        load_task(p0[j], off0[j], chunk_of(0) * CC, i & 3);
    }
    // the ring starts with K-steps 0 .. NSLOT - 2
    constexpr int NPRO = NSLOT - 1;
    u32x4 wpro[NPRO][SBV];
    if (producer) {
#pragma unroll
        for (int u = 0; u < NPRO; ++u) load_w(wpro[u], u < S ? u : S - 1);
    }
    for (int r = tid; r < ROWS; r += 512) rowoff_lds[r] = row_offset(r);
    if (tid < 8) sem_ready[tid] = 0u;
    for (int ci = tid; ci < Cin; ci += 512) {
        float sc = 1.f, sh = 0.f;
        if (use_gn) gn_scale_shift(a.in_st, a.B, b, Cin, ci, a.H * a.W, a.gamma, a.beta, sc, sh);
        ss_lds[2 * ci] = sc;
        ss_lds[2 * ci + 1] = sh;
    }
    wg_barrier();
#pragma unroll
    for (int j = 0; j < NVP0; ++j) {
        const int i = tid + j * 512;
        if (i < NTASK) put_task(p0[j], off0[j], chunk_of(0) * CC, i >> 2, i & 3, 0);
    }

    // epilogue coordinates (needed early: the residual rows are requested before the main loop ends)
    // undoes the weight scaling of the fp16 x 3 packing (and, SC, the operand scale)
    const float ASCALE = BF ? 1.0f : (SC ? in_inv / (float)(1 << X3_WSHIFT) : 1.0f / (float)(1 << X3_WSHIFT));
    const int g8 = tid % G8;
    const int nv = n_tile * NT + g8 * 8;                        // this thread's 8 channels
    const size_t img = (size_t)b * a.H * a.W;
    const ST* res_p = a.res.p ? (const ST*)a.res.p + img * a.res.cs + a.res.co + nv : nullptr;
    u32x4 rq[NU][LVI];
    auto fetch_res = [&]() {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int p = (tid + 512 * j) / G8;
            const int y = ty0 + p / PTW, x = tx0 + p % PTW;
            const bool ok = (y < a.H) && (x < a.W);
            const size_t pix = (size_t)y * a.W + x;
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < LVI; ++k) {
                rq[j][k] = (res_p && ok) ? *((const u32x4*)(res_p + pix * a.res.cs) + k) : z;
            }
        }
    };

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < NBW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    if (producer) {
        // =========================== producers ===========================
        // Producer step s (one per K-step of the consumers, at most one ahead of their K-step s in starting):
        //   * K-step s + NSLOT - 1's weights -> ring slot (s + NSLOT - 1) % NSLOT, free since the consumers left K-step s - 1;
        //   * a part of the NEXT chunk's patch -> the other patch buffer (tasks ptid + 256 * (k * RPS + j), k = s % KROWS).
        // So a producer step has NSLOT - 1 K-steps of MFMAs to finish, not one: its latency (global round trips, the
        // arithmetic of the split, LDS writes) hides behind the ring, only its throughput counts.
        // The global loads of both are issued U steps before their use, one register set per position in the unrolled loop.
        constexpr int U = KROWS > 1 ? KROWS : 2;
        u32x4 pset[U][RPS][LVI], wset[U][SBV];
        int oset[U][RPS];
        unsigned done_seen = 0;
        auto load_part = [&](u32x4 (&pr)[RPS][LVI], int (&po)[RPS], int s) {
            const int c0 = chunk_of(s / KROWS + 1) * CC, krow = s % KROWS;
#pragma unroll
            for (int j = 0; j < RPS; ++j) {
                const int i = ptid + 256 * (krow * RPS + j);
                const int row = (i >> 2) < ROWS ? (i >> 2) : ROWS - 1;
                po[j] = rowoff_lds[row];
                load_task(pr[j], po[j], c0, i & 3);
            }
        };
        auto put_part = [&](const u32x4 (&pr)[RPS][LVI], const int (&po)[RPS], int s) {
            const int cn = s / KROWS + 1, krow = s % KROWS;
            const int c0 = chunk_of(cn) * CC;
#pragma unroll
            for (int j = 0; j < RPS; ++j) {
                const int i = ptid + 256 * (krow * RPS + j);
                if (krow * RPS + j < NVP && i < NTASK) put_task(pr[j], po[j], c0, i >> 2, i & 3, cn & 1);
            }
        };
        // steady state without branches around LOADS (a branch makes the compiler's vmcnt bookkeeping give up and wait for
        // everything in flight): indices past the end are clamped, the redundant loads are never used
        auto stage_step = [&](int s, u32x4 (&ps)[RPS][LVI], int (&po)[RPS], u32x4 (&ws)[SBV]) {
            if (done_seen < (unsigned)s) {     // every consumer wave has left K-step s - 1
                do { done_seen = sem_min(sem_done); if (done_seen >= (unsigned)s) break; __builtin_amdgcn_s_sleep(1); } while (true);
                asm volatile("" ::: "memory");
            }
            const int u = s + NSLOT - 1;
            if (u < S && !(PDBG(a) & 1)) write_w(ws, u % NSLOT);
            if (!(PDBG(a) & 1)) load_w(ws, u + U < S ? u + U : S - 1);
            if (s / KROWS + 1 < NCH && !(PDBG(a) & 32)) put_part(ps, po, s);
            sem_signal(sem_ready, cw, (unsigned)s + 1u, lane);   // this wave's share of producer steps 0 .. s is in LDS
            const int sn = s + U;
            if (!(PDBG(a) & 2)) load_part(ps, po, sn / KROWS + 1 < NCH ? sn : (NCH > 1 ? (NCH - 2) * KROWS + sn % KROWS : 0));
        };
        if (NU <= 4) fetch_res();              // the residual rows: an input of the launch, requested once, used at the very end
#pragma unroll
        for (int u = 0; u < NPRO; ++u)
            if (u < S) write_w(wpro[u], u);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            // set j holds the weights of the first K-step u >= NSLOT - 1 with u % U == j
            const int u0 = NPRO + ((j - NPRO) % U + U) % U;
            load_w(wset[j], u0 < S ? u0 : S - 1);
            load_part(pset[j], oset[j], j / KROWS + 1 < NCH ? j : (NCH > 1 ? j % KROWS : 0));
        }
        wg_barrier();   // chunk 0 and the first K-steps are in LDS
        stamp(1);
        if (PDBG(a) & 2048) __builtin_amdgcn_s_setprio(1);
        if constexpr (KROWS > 1) {
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
#pragma unroll
                for (int k = 0; k < KROWS; ++k) stage_step(c * KROWS + k, pset[k], oset[k], wset[(k + NPRO) % U]);
            }
        } else {
            int c = 0;
#pragma unroll 1
            for (; c + 1 < NCH; c += 2) {
                stage_step(c, pset[0], oset[0], wset[NPRO % 2]);
                stage_step(c + 1, pset[1], oset[1], wset[(NPRO + 1) % 2]);
            }
            if (c < NCH) stage_step(c, pset[0], oset[0], wset[NPRO % 2]);   // c is even here
        }
    } else {
        // =========================== consumers ===========================
        const int half = lane >> 5, px = lane & 31;
        const char* a_ptr = patch + ((wm * MB) * PW + px) * RB + 16 * half;
        const char* b_ptr = bst + (wn * NBW) * 1024 + lane * 16;
        wg_barrier();   // chunk 0 and the first K-steps are in LDS
        stamp(1);
        if (!(PDBG(a) & 64)) __builtin_amdgcn_s_setprio(1);   // the matrix pipe first: the producers' arithmetic fills what is left
        // One flat stream of k-steps (a k-step = one MFMA K of one tap): the fragments of k-step i + 1 are requested before
        // the MFMAs of k-step i are issued -- also across a K-step boundary, after the producers' count says the next
        // K-step's operands are in LDS -- so the matrix pipe never waits for an LDS round trip or a rendezvous.
        u32x4 af[2][MB], afl[2][MB], bf[2][NBW], bfl[2][NBW];
        unsigned ready_seen = 0;
        // fragment loads in the order the MFMAs want them (small terms first: a_lo b_hi, then a_hi b_lo, a_hi b_hi)
        auto load_frag = [&](int fs, int s, int slot, int ks) {
            const int krow = s % KROWS, pbuf = (s / KROWS) & 1;
            const char* bs = b_ptr + slot * SBYTES;
            const char* ar = a_ptr + pbuf * PATCHB + ((TPS == 3) ? (krow * PW) * RB : ((TPS == 1 && TAPS == 9) ? ((krow / 3) * PW + krow % 3) * RB : 0));
            const int t = ks / KGC, kg = ks % KGC;
            const int ky = (TPS == 9) ? t / 3 : 0, kx = (TPS == 9) ? t % 3 : t;
            if (PDBG(a) & 128) return;   // ablation: no fragment reads (the MFMAs run on whatever the registers hold)
            if (!(PDBG(a) & 4096)) {   // ablation 4096: no B-fragment reads (what fetching the weight fragments from the L2 instead would leave on the LDS)
#pragma unroll
            for (int q = 0; q < NBW; ++q) bf[fs][q] = *(const u32x4*)(bs + ((t * KGC + kg) * (NT / 32) + q) * 1024);
            }
            if constexpr (X3) {
#pragma unroll
                for (int m = 0; m < MB; ++m) afl[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + 64 + kg * 32);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) af[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + kg * 32);
            if constexpr (WPL == 2) {
            if (!(PDBG(a) & 4096)) {
#pragma unroll
            for (int q = 0; q < NBW; ++q) bfl[fs][q] = *(const u32x4*)(bs + SB1 + ((t * KGC + kg) * (NT / 32) + q) * 1024);
            }
            }
        };
        constexpr int NRD = (X3 ? 2 : 1) * MB + WPL * NBW, NMF = (X3 ? 3 : WPL) * MB * NBW;     // LDS reads / MFMAs of one k-step
        static_assert(NKS % 2 == 0, "fragment double buffer: even k-steps per K-step");
        int slot = 0;
        auto mfma_step = [&](int s, bool last) {
            const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bool pre = true;
                if (ks + 1 < NKS) load_frag((ks + 1) & 1, s, slot, ks + 1);
                else if (!last) {
                    // K-step s + 1 needs producer steps 0 .. s - NSLOT + 2 (its weights) and, when it opens a chunk, the steps
                    // that staged the chunk's patch (the last of them: LASTK of the previous chunk).  (Reading the counts one
                    // k-step EARLIER, so that the read's LDS round trip runs under MFMAs, was measured in round 4 and is 2 - 8 %
                    // SLOWER on every layer: the producers finish just in time, an early read sees the old count and the
                    // consumer polls anyway -- profiles/r04_conv_pp.txt.)
                    const int sn = s + 1;
                    int need = sn - NSLOT + 2;
                    if (sn % KROWS == 0) { const int np = sn - KROWS + LASTK + 1; need = need > np ? need : np; }
                    if ((int)ready_seen < need) {
                        if (!(PDBG(a) & 64)) __builtin_amdgcn_s_setprio(0);      // never hold the priority while waiting for a lower-priority wave
                        do { ready_seen = sem_min(sem_ready); if ((int)ready_seen >= need) break; __builtin_amdgcn_s_sleep(1); } while (true);
                        if (!(PDBG(a) & 64)) __builtin_amdgcn_s_setprio(1);
                        asm volatile("" ::: "memory");
                    }
                    load_frag(0, sn, nslot, 0);
                } else pre = false;
                if (PDBG(a) & 4) continue;
                // the three terms of a product go to the same accumulator in a fixed order; the accumulators take turns
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
                // issue order: one MFMA, one read of the next k-step's fragments, ... (the reads ride in the matrix pipe's shadow)
                if (pre) {
#pragma unroll
                    for (int i = 0; i < (NRD < NMF ? NRD : NMF); ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            sem_signal(sem_done, cw, (unsigned)s + 1u, lane);   // every read of K-steps 0 .. s has been issued: ring slots and patch buffers up to there may be refilled
            slot = nslot;
        };
        load_frag(0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int s = 0; s < S - 1; ++s) mfma_step(s, false);
        if (NU <= 4) fetch_res();                             // in flight during the last MFMA block
        mfma_step(S - 1, true);
        __builtin_amdgcn_s_setprio(0);
    }
    stamp(2);
    wg_barrier();   // all fragment reads done: the patch buffers and the ring are dead
    stamp(3);
    if (PDBG(a) & 8) return;

    // ---------------- epilogue: accumulators -> LDS image of the tile -> all threads store ----------------
    float* scr = (float*)smem;                                 // [TH * 32 pixels][SCR_LD]
    if (!producer) {
        const int half = lane >> 5, px = lane & 31;
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
    }
    if (NU > 4) fetch_res();
    wg_barrier();

    ST* out_p = (ST*)a.out.p + img * a.out.cs + a.out.co + nv;
    ST* raw_p = a.raw.p ? (ST*)a.raw.p + img * a.raw.cs + a.raw.co + nv : nullptr;
    using ET = typename std::conditional<X3, float, typename std::conditional<BF, bf16_t, h16_t>::type>::type;       // store8 / load8 element tag
    const bool want_stats = (a.st_raw || a.st_out) && !(PDBG(a) & 256);
    float sr[8], qr[8], so[8], qo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sr[e] = qr[e] = so[e] = qo[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int p = (tid + 512 * j) / G8;                    // pixel of the tile
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
                    for (int e = 0; e < 8; ++e) { sr[e] += g[e]; qr[e] += g[e] * g[e]; }
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
                for (int e = 0; e < 8; ++e) { so[e] += f[e]; qo[e] += f[e] * f[e]; }
            }
        }
    }

    stamp(4);
    if (want_stats) {   // uniform over the grid
        // 512 threads x (4 sums x 8 channels) -> per-channel totals through LDS, in a fixed order: every thread parks its 32
        // partial sums as a row segment of part[thread / G8][kind * NT + channel]; the columns are added up in NSEG
        // segments of 32 rows by all threads, the segments by one thread per column.  (The first version ran a 3-round
        // butterfly over the 32 registers: 96 cross-lane moves per thread, 4 us of a 45 us launch.)
        constexpr int RL = 4 * NT, NROW = 512 / G8, NSEG = 512 / RL, RPSEG = NROW / NSEG;
        static_assert(NSEG >= 1 && RPSEG * NSEG == NROW, "statistics reduction geometry");
        wg_barrier();                                          // every thread is done with the tile image
        float* part = (float*)smem;                            // [NROW][RL]
        float* red2 = part + NROW * RL;                        // [NSEG][RL]
        {
            float* pr = part + (tid / G8) * RL + g8 * 8;
            *(f32x4*)(pr) = f32x4{sr[0], sr[1], sr[2], sr[3]};           *(f32x4*)(pr + 4) = f32x4{sr[4], sr[5], sr[6], sr[7]};
            *(f32x4*)(pr + NT) = f32x4{qr[0], qr[1], qr[2], qr[3]};      *(f32x4*)(pr + NT + 4) = f32x4{qr[4], qr[5], qr[6], qr[7]};
            *(f32x4*)(pr + 2 * NT) = f32x4{so[0], so[1], so[2], so[3]};  *(f32x4*)(pr + 2 * NT + 4) = f32x4{so[4], so[5], so[6], so[7]};
            *(f32x4*)(pr + 3 * NT) = f32x4{qo[0], qo[1], qo[2], qo[3]};  *(f32x4*)(pr + 3 * NT + 4) = f32x4{qo[4], qo[5], qo[6], qo[7]};
        }
        wg_barrier();
        {
            const int col = tid % RL, seg = tid / RL;
            const float* pc = part + (seg * RPSEG) * RL + col;
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RPSEG; ++r) t += pc[r * RL];
            red2[seg * RL + col] = t;
        }
        wg_barrier();
        stamp(5);
        if (tid < NT) {
            // thread = channel; t[kind]: kinds 0 / 1 = sum / sum of squares of `raw`, 2 / 3 of `out`
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = 0.f;
#pragma unroll
                for (int g = 0; g < NSEG; ++g) t[k] += red2[g * RL + k * NT + tid];
            }
            if (PDBG(a) & 16) return;
            // Channels -> GroupNorm groups of the tensor the slice belongs to.  All adds of the workgroup leave from ONE wave
            // (per tensor: the group's sum from its first lane, the sum of squares from its second, one instruction each):
            // atomics of different instructions that hit one cache line queue at the memory side -- spread over four waves the
            // same adds cost 3 us more (profiles/r03_conv_phase_breakdown.txt).
            const int cg = n_tile * NT + tid;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                GroupStat* st = k ? a.st_out : a.st_raw;
                if (!st) continue;
                const int gs = (k ? a.st_out_C : a.st_raw_C) / GN_GROUPS, co = k ? a.st_out_co : a.st_raw_co;
                const float s1 = group_lane_sum(t[2 * k], gs), s2 = group_lane_sum(t[2 * k + 1], gs);
                GroupStat* o = st + (size_t)b * GN_GROUPS + (co + cg) / gs;
                if (tid % gs == 0) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
        }
    }
}

template <typename T, int TAPS, int TH, int NT, int TPS, int NSLOT, bool SC = false>
int launch_pc_t(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    using G = PGeo<TAPS, TH, NT, TPS, NSLOT, IS_X3<T>, std::is_same<T, bf16_t>::value ? 1 : 2>;
    size_t smem = G::smem_bytes(a.in.C);
    // ONE workgroup per CU, always.  The small fp16 tilings (79 KB of LDS, ~120 registers) fit twice; with two workgroups on
    // a CU every SIMD holds two high-priority consumer waves, and when both poll for operands their producers -- priority 0
    // on the same SIMD -- were observed to be starved for SECONDS: conv_pc_kernel<h16_t,9,8,32,3,2> at 256^2 took 25 s for a
    // launch that takes 50 us (scripts/fp16_hang_probe.py; it resolves only when the driver's time slicing reshuffles the
    // waves).  The hand-over by polling was designed and measured with one workgroup per CU (what the fp16 x 3 tilings always
    // get); an LDS request of more than half the CU's 160 KB keeps it that way for every tiling.
    // The request is derived from the device's LDS per CU (not a constant), and the first launch of every tiling ASKS the
    // runtime how many of these workgroups a CU would hold: anything but one is refused (a later change that shrinks the
    // request, or a part with more LDS per CU, would bring the starvation back silently otherwise -- ADVICE round 3).
    if (h->lds_per_cu <= 0) {
        CHORE_HIP_CHECK(h, hipDeviceGetAttribute(&h->lds_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, h->device));
        if (h->lds_per_cu <= 0) h->lds_per_cu = 160 * 1024;
    }
    const int lds_cu = h->lds_per_cu;
    if (smem > (size_t)lds_cu) CHORE_FAIL(h, CHORE_EINVAL, "conv_pc: %zu bytes of LDS, the CU has %d", smem, lds_cu);
    if (smem < (size_t)lds_cu / 2 + 1024) smem = (size_t)lds_cu / 2 + 1024;
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)conv_pc_kernel<T, TAPS, TH, NT, TPS, NSLOT, SC>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_cu));
        int per_cu = 0;
        CHORE_HIP_CHECK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_pc_kernel<T, TAPS, TH, NT, TPS, NSLOT, SC>,
                                                                        512, smem));
        if (per_cu != 1)
            CHORE_FAIL(h, CHORE_EINVAL, "conv_pc: %d workgroups per CU with %zu bytes of LDS -- the hand-over by polling needs exactly one",
                       per_cu, smem);
        attr = true;
    }
    const int tiles = ((a.W + PTW - 1) / PTW) * ((a.H + TH - 1) / TH);
    dim3 grid(tiles * (a.Cout / NT) * a.B);
    hipLaunchKernelGGL((conv_pc_kernel<T, TAPS, TH, NT, TPS, NSLOT, SC>), grid, dim3(512), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // namespace

// tile configuration of the specialised-wave kernel for a layer: th = 0 -> not covered (the caller uses conv_lds_kernel)
PcPlan conv_pc_plan(int dtype, int taps, int B, int H, int W, int Cin, int Cout, int force) {
    PcPlan p{0, 0, 0, 0};
    if ((dtype != CHORE_F16X3 && dtype != CHORE_F16 && dtype != CHORE_BF16) || Cin % 32 || Cout % 32) return p;
    auto ring = [&](PcPlan& q) {   // taps per K-step and ring depth of a tiling (what fits 160 KB of LDS)
        // (measured, profiles/r03_conv_phase_breakdown.txt: rings of single taps with 4 - 6 slots are no faster than two slots of
        // whole kernel rows -- the consumers pay per K-step for the hand-over -- except where only single taps fit: 128 channels)
        if (taps == 1) { q.tps = 1; q.nslot = 2; return; }
        if (q.nt == 128) { q.tps = 1; q.nslot = 3; }
        else if (q.th == 4 && q.nt == 32) { q.tps = 9; q.nslot = 2; }
        else { q.tps = 3; q.nslot = 2; }
    };
    if (!force) {   // CHORE_PC_FORCE="taps,Cin,Cout,H:th*1000+nt[;...]" -- tiling experiments (scripts/conv_layer_ab.py)
        static const char* env = getenv("CHORE_PC_FORCE");
        if (env) {
            const char* q = env;
            while (*q) {
                int t, ci, co, hh, f, n = 0;
                if (sscanf(q, "%d,%d,%d,%d:%d%n", &t, &ci, &co, &hh, &f, &n) == 5 && t == taps && ci == Cin && co == Cout && hh == H) { force = f; break; }
                while (*q && *q != ';') ++q;
                if (*q == ';') ++q;
            }
        }
    }
    if (force) {   // development: th * 1000 + nt (e.g. 8064)
        p.th = force / 1000; p.nt = force % 1000;
        ring(p);
        return p;
    }
    const long px_tiles8 = (long)B * ((H + 7) / 8) * (W / 32);
    if (taps == 1) {
        if (Cout % 128 == 0) { p.th = 8; p.nt = 128; }
        else if (Cout % 64 == 0) { p.th = 8; p.nt = 64; }
        else return p;
        ring(p);
        return p;
    }
    // 3x3: the widest channel tile that still gives every CU a workgroup; 4-row tiles when 8-row tiles leave CUs idle
    if (Cout % 128 == 0 && px_tiles8 * (Cout / 128) >= 256) { p.th = 8; p.nt = 128; }
    else if (Cout % 64 == 0 && px_tiles8 * (Cout / 64) >= 256) { p.th = 8; p.nt = 64; }
    else if (px_tiles8 * (Cout / 32) >= 256) { p.th = 8; p.nt = 32; }
    else if (H % 4 == 0 && Cout % 64 == 0 && px_tiles8 * 2 * (Cout / 64) >= 256) { p.th = 4; p.nt = 64; }
    else if (H % 4 == 0) { p.th = 4; p.nt = 32; }
    else { p.th = 8; p.nt = 32; }
    ring(p);
    return p;
}

int launch_conv_pc(chore_handle* h, int dtype, int taps, const PcPlan& p, const ConvArgs& a, hipStream_t s) {
    if (a.res2.p) CHORE_FAIL(h, CHORE_EINVAL, "conv_pc: a second residual is not supported (conv_lds_kernel has it)");
    const int key = ((taps * 10 + p.th) * 1000 + p.nt) * 100 + p.tps * 10 + p.nslot;
#define PC_CASE(TAPS, TH, NT, TPS, NSLOT) \
    case ((TAPS * 10 + TH) * 1000 + NT) * 100 + TPS * 10 + NSLOT:                                              \
        return dtype == CHORE_F16 ? launch_pc_t<h16_t, TAPS, TH, NT, TPS, NSLOT>(h, a, s) \
             : dtype == CHORE_BF16 ? launch_pc_t<bf16_t, TAPS, TH, NT, TPS, NSLOT>(h, a, s)                     \
             : a.in_amax ? launch_pc_t<x3_t, TAPS, TH, NT, TPS, NSLOT, true>(h, a, s) : launch_pc_t<x3_t, TAPS, TH, NT, TPS, NSLOT>(h, a, s)
    switch (key) {
        PC_CASE(9, 8, 128, 1, 3);
        PC_CASE(9, 8, 64, 3, 2);
        PC_CASE(9, 8, 32, 3, 2);
        PC_CASE(9, 4, 64, 3, 2);
        PC_CASE(9, 4, 32, 9, 2);
        PC_CASE(1, 8, 128, 1, 2);
        PC_CASE(1, 8, 64, 1, 2);
    }
#undef PC_CASE
    CHORE_FAIL(h, CHORE_EINVAL, "conv_pc: no kernel for taps=%d th=%d nt=%d tps=%d nslot=%d", taps, p.th, p.nt, p.tps, p.nslot);
}

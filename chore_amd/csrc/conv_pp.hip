// conv_pp.hip -- the specialised-wave convolution of conv_pc.hip as PERSISTENT workgroups that loop over tiles (round 4).
//
// Same arithmetic, same ConvArgs contract, same packed weights and the same wave roles as conv_pc_kernel (ConvBlock's
// convolutions, model/net_util.py:346-396; the 1x1 convolutions of the stack tail, model/HGFilters.py:128-142,167-183): waves
// 0-3 (consumers) read fragments from LDS and issue MFMAs, waves 4-7 (producers) move data.  What changes is the control flow:
//
//   * a workgroup processes `tpw` tiles one after the other; the K-steps of all its tiles form ONE stream.  The producers run
//     ahead across the tile boundary -- the first patch chunk and the first weight slots of tile t + 1 are staged during the
//     last K-steps of tile t -- so only the first tile of a workgroup has a prologue;
//   * at the end of a tile the consumers dump their accumulators (+ bias) into an LDS image of the tile that is SEPARATE from
//     the patch buffers and the weight ring, clear them and go on with the next tile's K-steps at once;
//   * the epilogue of tile t -- residual add, the two stores, the GroupNorm statistics of what was stored -- is done by the
//     PRODUCER waves, a few (pixel, 8 channels) units per producer step, while the consumers run tile t + 1: its memory
//     traffic rides beside the MFMAs instead of after them.  Only the last tile's epilogue is exposed (all eight waves share it);
//   * every hand-over is a progress count in LDS (conv_pc.hip): producer steps done, K-steps left, images dumped, images drained.
//
// Why (profiles/r03_conv_phase_breakdown.txt, r04_*): with one tile per CU every launch is prologue -> K loop -> epilogue with
// all 256 workgroups in the same phase at the same time: the K loop is bound by the matrix cores (and the chip's power), the
// epilogue and the prologue by the memory system, and neither overlaps the other.  The statistics leave as per-WAVE partial
// sums through the exact fixed-point atomics (enc_common.h): integer adds, so the totals do not depend on who adds when.
#include "conv_common.h"

#ifndef CHORE_CONV_ABLATE
#define CHORE_CONV_ABLATE 0
#endif

#if CHORE_CONV_ABLATE
#define PDBG(a) ((a).dbg)       // probe builds (scripts/probes/conv_bench.hip): 1 no weight staging, 2 no patch loads, 4 no MFMAs,
#else                           // 8 no epilogue units, 32 no patch publish, 256 no statistics -- results are wrong when set
#define PDBG(a) 0
#endif

using namespace conv_detail;

namespace {

constexpr int PTW = 32;          // tile width in pixels (one MFMA pixel block)
constexpr int RQ = 4;            // residual-row register sets of a thread: epilogue units in flight

template <int TAPS, int TH_, int NT_, int TPS_, int NSLOT_, bool X3_ = true> struct QGeo {
    static constexpr int TH = TH_, NT = NT_, TPS = TPS_, NSLOT = NSLOT_;
    static constexpr int PAD = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = PTW + 2 * PAD, PH = TH + 2 * PAD, ROWS = PH * PW;
    static constexpr int RB = X3_ ? 144 : 80;                       // LDS patch row: 64 B hi [+ 64 B lo] + 16 B pad
    static constexpr int PATCHB = ROWS * RB;
    static constexpr int KROWS = TAPS / TPS;                        // K-steps per chunk
    static constexpr int NB = NT / 32;
    static constexpr int SB1 = TPS * KGC * NB * 1024;               // bytes of one operand plane of a K-step
    static constexpr int SBYTES = 2 * SB1;                          // hi plane, then lo plane
    static constexpr int NBW = NT >= 64 ? 2 : 1;                    // channel blocks per consumer wave
    static constexpr int WAVES_N = NB / NBW, WAVES_M = 4 / WAVES_N, MB = TH / WAVES_M;
    static constexpr int G8 = NT / 8;                               // 8-channel groups per pixel
    static constexpr int NUNIT = TH * PTW * G8;                     // (pixel, 8 channels) units of a tile
    static constexpr int NUP = NUNIT / 256, NUF = NUNIT / 512;      // units per producer thread (overlapped) / per thread (last tile)
    static constexpr int IMGB = TH * PTW * NT * 4;                  // the tile image: fp32 [pixel][channel]
    static constexpr size_t smem_bytes(int Cin) { return (size_t)2 * PATCHB + (size_t)NSLOT * SBYTES + IMGB + (size_t)Cin * 8 + 64 + 16; }
    static_assert(MB >= 1 && WAVES_M * MB == TH, "tile rows must divide over the consumer waves");
    static_assert(NUF >= 1 && NUF <= RQ && 256 % G8 == 0, "epilogue unit geometry");
};

using lds_u32 = __attribute__((address_space(3))) unsigned;
using lds_u32x4 = __attribute__((address_space(3))) u32x4;
__device__ __forceinline__ void sem_signal(unsigned* sem, int wave, unsigned count, int lane) {
    asm volatile("" ::: "memory");   // the LDS traffic before it is issued before it (a wave's LDS operations execute in order)
    if (lane == 0) *(volatile lds_u32*)(sem + wave) = count;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned sem_min(const unsigned* sem) {
    const u32x4 v = *(const volatile lds_u32x4*)sem;
    const unsigned a = v[0] < v[1] ? v[0] : v[1], b = v[2] < v[3] ? v[2] : v[3];
    return a < b ? a : b;
}
// bounded in debug builds only: a hand-over that never arrives is a bug, and a hung GPU costs the whole box
__device__ __forceinline__ void sem_wait(const unsigned* sem, unsigned need, unsigned& seen) {
    if (seen >= need) return;
#if CHORE_CONV_ABLATE
    unsigned spins = 0;
#endif
    do {
        seen = sem_min(sem);
        if (seen >= need) break;
        __builtin_amdgcn_s_sleep(1);
#if CHORE_CONV_ABLATE
        if (++spins > (1u << 23)) __builtin_trap();      // probe builds: about a second of polling = a lost hand-over, abort the launch
#endif
    } while (true);
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int TAPS, int TH_, int NT_, int TPS_, int NSLOT_>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(ConvArgs a, int tpw) {
    if constexpr (IS_X3<T> || IS_H16<T>) f16_saturate_mode();
    static_assert(IS_X3<T> || IS_H16<T>, "conv_pp_kernel: fp16 x 3 or fp16 operands");
    constexpr bool X3 = IS_X3<T>;
    using ST = typename std::conditional<X3, float, unsigned short>::type;     // element type in memory
    using ET = typename std::conditional<X3, float, h16_t>::type;              // store8 / load8 element tag
    constexpr int LVI = X3 ? 2 : 1;                     // 16-byte loads per 8 channels
    using G = QGeo<TAPS, TH_, NT_, TPS_, NSLOT_, X3>;
    constexpr int NSLOT = G::NSLOT;
    constexpr int TH = G::TH, NT = G::NT, TPS = G::TPS, PAD = G::PAD, PW = G::PW, ROWS = G::ROWS, RB = G::RB;
    constexpr int PATCHB = G::PATCHB, KROWS = G::KROWS, SB1 = G::SB1, SBYTES = G::SBYTES;
    constexpr int NBW = G::NBW, WAVES_N = G::WAVES_N, MB = G::MB, G8 = G::G8, NUP = G::NUP, NUF = G::NUF;
    constexpr int KGE = 16, CC = 32;                    // channels per k-group / per chunk
    constexpr int NTASK = ROWS * 4;                     // staging tasks per chunk: (patch row, 8 channels)
    constexpr int NVP0 = (NTASK + 511) / 512;
    constexpr int NVP = (NTASK + 255) / 256;
    constexpr int RPS = KROWS > 1 ? (NVP + KROWS - 2) / (KROWS - 1) : NVP;
    constexpr int LASTK = (NVP - 1) / RPS;              // last K-step of a chunk with staging work
    constexpr int SVEC = SBYTES / 16, SBV = (SVEC + 255) / 256, SV1 = SB1 / 16;
    constexpr int NKS = TPS * KGC;                      // MFMA k-steps per K-step

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                                  // [2][ROWS][RB]
    char* bst = smem + 2 * PATCHB;                       // [NSLOT][SBYTES]
    float* img = (float*)(bst + NSLOT * SBYTES);         // [TH * 32 pixels][NT]: the finished tile, consumers -> producers
    float* ss_lds = img + TH * PTW * NT;                 // [Cin][2]
    const int sem_off = (2 * PATCHB + NSLOT * SBYTES + G::IMGB + a.in.C * 8 + 15) & ~15;
    unsigned* sem_ready = (unsigned*)(smem + sem_off);   // [4] producer waves: producer steps done
    unsigned* sem_done = sem_ready + 4;                  // [4] consumer waves: K-steps left behind
    unsigned* sem_img = sem_ready + 8;                   // [4] consumer waves: tiles dumped into `img`
    unsigned* sem_free = sem_ready + 12;                 // [4] producer waves: tiles drained from `img`

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#if CHORE_CONV_ABLATE
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if (a.dbg_ticks && blockIdx.x == gridDim.x / 2 && (tid == 0 || tid == 256)) {
            a.dbg_ticks[(tid ? 16 : 0) + 2 * i] = __builtin_readcyclecounter();
            a.dbg_ticks[(tid ? 16 : 0) + 2 * i + 1] = wall_clock64();
        }
    };
#else
    auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
    stamp(0);
    const bool producer = wid >= 4;
    const int cw = wid & 3;
    const int wn = cw % WAVES_N, wm = cw / WAVES_N;
    const int ptid = tid & 255;
    const int tiles_x = (a.W + PTW - 1) / PTW;
    const int ntn = a.Cout / NT, tiles = tiles_x * ((a.H + TH - 1) / TH);
    const int total = a.B * tiles * ntn;
    int lid = blockIdx.x;
    {   // XCD-aware placement: every XCD takes a contiguous range of workgroups = of (image, pixel tile, channel tile)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = lid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
    }
    const int first = lid * tpw;
    const int ntile = total - first < tpw ? total - first : tpw;      // >= 1 (the host sizes the grid)
    const int b = first / (tiles * ntn);                              // all tiles of a workgroup lie in one image (host)
    const int Cin = a.in.C;
    const bool use_gn = a.in_st != nullptr;
    const int NKG = Cin / KGE, NBt = a.Cout / 32;
    const int NCH = Cin / CC;
    const int S = NCH * KROWS;                   // K-steps of one tile
    const int GS = ntile * S, GCH = ntile * NCH; // K-steps / chunks of the workgroup
    const ST* in_b = (const ST*)a.in.p + (size_t)b * a.H * a.W * a.in.cs + a.in.co;

    // tile t of this workgroup: consecutive pixel tiles of ONE channel tile (so the statistics of all its tiles go to the same
    // accumulators and are added once per workgroup: the number of device-scope atomics is what they cost).  The four numbers of
    // a tile cost a handful of integer divisions: every stream of the kernel (weights, patch parts, drained tile, consumer
    // tile) keeps those of ITS current tile in scalar registers and decodes again only when it moves to the next tile.  (A first
    // version read them from a small LDS table at every use: each read was an LDS round trip with `s_waitcnt lgkmcnt(0)` in the
    // producers' critical path, and the K loop ran at half speed -- profiles/r04_conv_pp.txt.)
    struct TI { int n, y0, x0, crot; };      // channel tile, first row, first column, chunk rotation
    auto decode = [&](int t) __attribute__((always_inline)) -> TI {
        const int g = (first + (t < ntile ? t : ntile - 1)) % (tiles * ntn);     // (image b is fixed)
        const int n = g / tiles, px = g - n * tiles;     // channel tile SLOWEST: all tiles of a workgroup share it (host: tiles % tpw == 0)
        return TI{n, (px / tiles_x) * TH, (px % tiles_x) * PTW, (px * 5 + n * 3) % NCH};   // chunk order rotated per tile: neighbouring workgroups do not all ask for the same rows at once
    };
    auto rot = [&](const TI& ti, int ci) __attribute__((always_inline)) -> int {
        const int x = ci + ti.crot;
        return x >= NCH ? x - NCH : x;
    };
    auto row_offset = [&](int ty0, int tx0, int row) __attribute__((always_inline)) -> int {
        const int y = ty0 + row / PW - PAD, x = tx0 + row % PW - PAD;
        const bool ok = (row < ROWS) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        return ok ? (y * a.W + x) * a.in.cs : -1;
    };
    auto load_task = [&](u32x4 (&r)[LVI], int off, int c0, int v) {
        const u32x4* p = (const u32x4*)(in_b + (off >= 0 ? off : 0) + c0 + v * 8);
#pragma unroll
        for (int k = 0; k < LVI; ++k) r[k] = p[k];
    };
    auto put_task = [&](const u32x4 (&r)[LVI], int off, int c0, int row, int v, int pbuf) {
        float sc[8], sh[8];
        {
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
            if (off >= 0) xform_x3(r[0], r[LVI - 1], sc, sh, use_gn, hi, lo);
            *(u32x4*)d = hi;
            *(u32x4*)(d + 64) = lo;
        } else {
            if (off >= 0) {
                if (use_gn) {
                    const f16x8_t x = __builtin_bit_cast(f16x8_t, r[0]);
                    f16x8_t y;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = fmaf((float)x[j], sc[j], sh[j]);
                        y[j] = (_Float16)(t > 0.f ? t : 0.f);
                    }
                    hi = __builtin_bit_cast(u32x4, y);
                } else hi = r[0];
            }
            *(u32x4*)d = hi;
        }
    };

    // weight slice of global K-step u: [plane][t][kg][nb][lane] vectors
    const size_t wkg = (size_t)NBt * 64;   // vectors between consecutive k-groups
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
    int w_t = 0, w_base = 0;                 // weights stream (producers; K-steps are requested in increasing order)
    TI w_ti = decode(0);
    auto load_w = [&](u32x4 (&rb)[SBV], int u) __attribute__((always_inline)) {
        if (u > GS - 1) u = GS - 1;                        // past the end: a redundant load that is never used (no branch around loads)
        while (u >= w_base + S) { ++w_t; w_base += S; w_ti = decode(w_t); }
        const int s = u - w_base;
        const int c = rot(w_ti, s / KROWS), krow = s % KROWS;
        const u32x4* wb = (const u32x4*)a.wpk + (size_t)(w_ti.n * (NT / 32)) * 64 + (size_t)(krow * TPS * NKG + c * KGC) * wkg;   // wave-uniform
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

    // ---------------- prologue (first tile only): everybody stages chunk 0; producers fetch the first weights ----------------
    u32x4 p0[NVP0][LVI];
    int off0[NVP0];
    {
        const TI t0 = decode(0);
        const int ty0 = t0.y0, tx0 = t0.x0, c0 = rot(t0, 0) * CC;
#pragma unroll
        for (int j = 0; j < NVP0; ++j) {
            const int i = tid + j * 512;
            off0[j] = row_offset(ty0, tx0, i >> 2);
            load_task(p0[j], off0[j], c0, i & 3);
        }
    }
    constexpr int NPRO = NSLOT - 1;      // the ring starts with K-steps 0 .. NSLOT - 2
    u32x4 wpro[NPRO][SBV];
    if (producer) {
#pragma unroll
        for (int u = 0; u < NPRO; ++u) load_w(wpro[u], u);
    }
    if (tid < 16) sem_ready[tid] = 0u;
    for (int ci = tid; ci < Cin; ci += 512) {
        float sc = 1.f, sh = 0.f;
        if (use_gn) gn_scale_shift(a.in_st, a.B, b, Cin, ci, a.H * a.W, a.gamma, a.beta, sc, sh);
        ss_lds[2 * ci] = sc;
        ss_lds[2 * ci + 1] = sh;
    }
    wg_barrier();
    {
        const int c0 = rot(decode(0), 0) * CC;
#pragma unroll
        for (int j = 0; j < NVP0; ++j) {
            const int i = tid + j * 512;
            if (i < NTASK) put_task(p0[j], off0[j], c0, i >> 2, i & 3, 0);
        }
    }

    // ---------------- the epilogue of a tile, in (pixel, 8 channels) units ----------------
    constexpr float ASCALE = 1.0f / (float)(1 << X3_WSHIFT);   // undoes the weight scaling of the fp16 x 3 packing
    const bool want_stats = (a.st_raw || a.st_out) && !(PDBG(a) & 256);
    const size_t imgpix = (size_t)b * a.H * a.W;
    float sr[8], qr[8], so[8], qo[8];                           // this thread's partial sums of the tile being drained
    auto stats_clear = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sr[e] = qr[e] = so[e] = qo[e] = 0.f; }
    };
    // unit u of tile t: request its residual row
    auto unit_issue = [&](u32x4 (&rq)[LVI], const TI& ti, int u) __attribute__((always_inline)) {
        if (PDBG(a) & 8) return;
        const int p = u / G8, g8 = u % G8;
        const int y = ti.y0 + p / PTW, x = ti.x0 + p % PTW;
        const bool ok = (y < a.H) && (x < a.W);
        const u32x4 z = {0u, 0u, 0u, 0u};
        const ST* res_p = a.res.p ? (const ST*)a.res.p + imgpix * a.res.cs + a.res.co + ti.n * NT + g8 * 8 : nullptr;
        const size_t pix = (size_t)y * a.W + x;
#pragma unroll
        for (int k = 0; k < LVI; ++k) rq[k] = (res_p && ok) ? *((const u32x4*)(res_p + pix * a.res.cs) + k) : z;
    };
    // ... and finish it: image -> (+ residual) -> stores, sums of what was stored
    auto unit_finish = [&](const u32x4 (&rq)[LVI], const TI& ti, int u) __attribute__((always_inline)) {
        if (PDBG(a) & 8) return;
        const int p = u / G8, g8 = u % G8;
        const int y = ti.y0 + p / PTW, x = ti.x0 + p % PTW;
        const int nv = ti.n * NT + g8 * 8;
        float f[8];
        {
            const f32x4 lo = *(const f32x4*)(img + p * NT + g8 * 8), hi = *(const f32x4*)(img + p * NT + g8 * 8 + 4);
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        }
        if (y < a.H && x < a.W) {
            const size_t pix = (size_t)y * a.W + x;
            if (a.raw.p) {
                ST* raw_p = (ST*)a.raw.p + imgpix * a.raw.cs + a.raw.co + nv;
                float g[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = f[e];
                store8<ET>((ET*)(raw_p + pix * a.raw.cs), g);
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sr[e] += g[e]; qr[e] += g[e] * g[e]; }
                }
            }
            if (a.res.p) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (X3) { f[k] += __uint_as_float(rq[0][k]); f[4 + k] += __uint_as_float(rq[LVI - 1][k]); }
                    else {
                        const f16x8_t rh = __builtin_bit_cast(f16x8_t, rq[0]);
                        f[2 * k] += (float)rh[2 * k]; f[2 * k + 1] += (float)rh[2 * k + 1];
                    }
                }
            }
            ST* out_p = (ST*)a.out.p + imgpix * a.out.cs + a.out.co + nv;
            store8<ET>((ET*)(out_p + pix * a.out.cs), f);
            if (want_stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { so[e] += f[e]; qo[e] += f[e] * f[e]; }
            }
        }
    };
    // The statistics: a thread's units always cover the same eight channels (u % G8 == ptid % G8) and all tiles of the workgroup
    // the same channel tile, so its partial sums simply run on over the tiles.  At the very end: butterfly over the lanes of a wave
    // that share an 8-channel group, the eight waves' totals through LDS in a fixed order, then ONE wave adds to the exact
    // accumulators -- as many atomics per workgroup as conv_pc_kernel issues.  (Per-wave adds after every tile, the first
    // version, cost 80 us of a 250 us launch: profiles/r04_conv_pp.txt.)
    auto stats_flush_wg = [&](const TI& ti) __attribute__((always_inline)) {
        if (!want_stats) return;     // uniform over the grid
#pragma unroll
        for (int m = G8; m < 64; m <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sr[e] += __shfl_xor(sr[e], m, 64); qr[e] += __shfl_xor(qr[e], m, 64);
                so[e] += __shfl_xor(so[e], m, 64); qo[e] += __shfl_xor(qo[e], m, 64);
            }
        }
        wg_barrier();                                          // every wave is done with the image, the patch buffers and the ring
        float* part = (float*)smem;                            // [8 waves][G8][4 kinds][8 channels]
        if (lane < G8) {
            float* pr = part + ((size_t)wid * G8 + lane) * 32;
            *(f32x4*)(pr) = f32x4{sr[0], sr[1], sr[2], sr[3]};        *(f32x4*)(pr + 4) = f32x4{sr[4], sr[5], sr[6], sr[7]};
            *(f32x4*)(pr + 8) = f32x4{qr[0], qr[1], qr[2], qr[3]};    *(f32x4*)(pr + 12) = f32x4{qr[4], qr[5], qr[6], qr[7]};
            *(f32x4*)(pr + 16) = f32x4{so[0], so[1], so[2], so[3]};   *(f32x4*)(pr + 20) = f32x4{so[4], so[5], so[6], so[7]};
            *(f32x4*)(pr + 24) = f32x4{qo[0], qo[1], qo[2], qo[3]};   *(f32x4*)(pr + 28) = f32x4{qo[4], qo[5], qo[6], qo[7]};
        }
        wg_barrier();
        if (tid < NT) {
            // thread = channel; t[kind]: kinds 0 / 1 = sum / sum of squares of `raw`, 2 / 3 of `out`
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) t[k] += part[((size_t)w * G8 + tid / 8) * 32 + k * 8 + tid % 8];
            }
            const int cg = ti.n * NT + tid;
            // (two explicit blocks: selecting a.st_raw / a.st_out by a loop variable made the compiler keep the argument struct in scratch)
            if (a.st_raw) {
                const int gs = a.st_raw_C / GN_GROUPS;
                const float s1 = group_lane_sum(t[0], gs), s2 = group_lane_sum(t[1], gs);
                GroupStat* o = a.st_raw + (size_t)b * GN_GROUPS + (a.st_raw_co + cg) / gs;
                if (tid % gs == 0) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
            if (a.st_out) {
                const int gs = a.st_out_C / GN_GROUPS;
                const float s1 = group_lane_sum(t[2], gs), s2 = group_lane_sum(t[3], gs);
                GroupStat* o = a.st_out + (size_t)b * GN_GROUPS + (a.st_out_co + cg) / gs;
                if (tid % gs == 0) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
        }
    };
    // last tile: all 512 threads share the units (producers the first half of every 512, consumers the second)
    auto final_drain = [&](int t, int half) __attribute__((always_inline)) {
        unsigned seen = 0;
        sem_wait(sem_img, (unsigned)(t + 1), seen);
        const TI ti = decode(t);
        u32x4 rq[RQ][LVI];
#pragma unroll
        for (int j = 0; j < NUF; ++j) unit_issue(rq[j], ti, ptid + 256 * half + 512 * j);
#pragma unroll
        for (int j = 0; j < NUF; ++j) unit_finish(rq[j], ti, ptid + 256 * half + 512 * j);
    };
    stats_clear();

    if (producer) {
        // =========================== producers ===========================
        // register sets of the prefetched patch parts / weights: set (step % U), loads issued U steps before their use.  Three
        // steps ahead are enough (conv_pc_kernel keeps one set per K-step of a chunk: nine sets of weights = 144 registers)
        constexpr int U = KROWS > 1 ? (KROWS % 3 == 0 ? 3 : KROWS) : 2;
        constexpr int UW = U;
        static_assert(KROWS == 1 || KROWS % U == 0, "a chunk is a whole number of register-set rounds");
        u32x4 pset[U][RPS][LVI], wset[UW][SBV];
        int oset[U][RPS];
        unsigned done_seen = 0;
        // the part of the NEXT chunk's patch that goes with global K-step gsx (chunk gsx / KROWS + 1, maybe the next tile's first).
        // Requested in increasing gsx; the chunk's channel offset and patch buffer are handed to put_part through c0s / pbs.
        int l_gc = GCH > 1 ? 1 : 0, l_t = (GCH > 1 && NCH == 1) ? 1 : 0, l_ci = (GCH > 1 && NCH > 1) ? 1 : 0;
        TI l_ti = decode(l_t);
        int c0set[U], pbset[U];
        auto load_part = [&](u32x4 (&pr)[RPS][LVI], int (&po)[RPS], int& c0s, int& pbs, int gsx) __attribute__((always_inline)) {
            int gcn = gsx / KROWS + 1;
            const int krow = gsx % KROWS;
            if (gcn > GCH - 1) gcn = GCH - 1;            // past the end: redundant loads, never used
            while (l_gc < gcn) {
                ++l_gc;
                if (++l_ci == NCH) { l_ci = 0; ++l_t; l_ti = decode(l_t); }
            }
            const int c0 = rot(l_ti, l_ci) * CC, ty0 = l_ti.y0, tx0 = l_ti.x0;
            c0s = c0;
            pbs = l_gc & 1;
#pragma unroll
            for (int j = 0; j < RPS; ++j) {
                const int i = ptid + 256 * (krow * RPS + j);
                const int row = (i >> 2) < ROWS ? (i >> 2) : ROWS - 1;
                po[j] = row_offset(ty0, tx0, row);
                load_task(pr[j], po[j], c0, i & 3);
            }
        };
        auto put_part = [&](const u32x4 (&pr)[RPS][LVI], const int (&po)[RPS], int c0, int pbuf, int gsx) __attribute__((always_inline)) {
            const int krow = gsx % KROWS;
#pragma unroll
            for (int j = 0; j < RPS; ++j) {
                const int i = ptid + 256 * (krow * RPS + j);
                if (krow * RPS + j < NVP && i < NTASK) put_task(pr[j], po[j], c0, i >> 2, i & 3, pbuf);
            }
        };
        // ---- draining the image of the previous tile, a few units per producer step ----
        constexpr int RQD = 2;        // residual rows in flight per thread while draining beside the K loop
        u32x4 rq[RQD][LVI];
        int drain_t = 0;              // next tile to drain
        TI d_ti = decode(0);
        int j_done = 0, j_issued = 0; // units of it finished / requested
        unsigned img_seen = 0;
        // one slot per chunk of the next tile (per pair of chunks when a chunk is a single K-step); units per slot: NUP units
        // and one slot of lead must fit into the slots before the forced one
        constexpr int FORCE = KROWS > 1 ? 2 : 3;         // the slot of chunk ci finishes the drain when NCH - ci <= FORCE
        const int free_slots = (KROWS > 1 ? NCH : NCH / 2) - 2;
        int ups = free_slots > 0 ? (NUP + free_slots - 1) / free_slots : NUP;
        if (ups < 1) ups = 1;
        if (ups > RQD) ups = RQD;
        auto drain_slot = [&](bool must_finish) __attribute__((always_inline)) {
            // tile drain_t is complete in `img` once every consumer wave has dumped it
            if (img_seen < (unsigned)(drain_t + 1)) {
                if (must_finish) sem_wait(sem_img, (unsigned)(drain_t + 1), img_seen);
                else {
                    img_seen = sem_min(sem_img);
                    if (img_seen < (unsigned)(drain_t + 1)) return;
                }
            }
            do {
                const int nfin = j_issued - j_done;
#pragma unroll
                for (int i = 0; i < RQD; ++i)
                    if (i < nfin) unit_finish(rq[i], d_ti, ptid + 256 * (j_done + i));
                j_done = j_issued;
                if (j_done == NUP) {
                    sem_signal(sem_free, cw, (unsigned)(drain_t + 1), lane);     // the image may be overwritten
                    ++drain_t;
                    d_ti = decode(drain_t);
                    j_done = j_issued = 0;
                    return;
                }
                const int nis = NUP - j_issued < ups ? NUP - j_issued : ups;
#pragma unroll
                for (int i = 0; i < RQD; ++i)
                    if (i < nis) unit_issue(rq[i], d_ti, ptid + 256 * (j_issued + i));
                j_issued += nis;
            } while (must_finish);
        };
        auto stage_step = [&](int gs, u32x4 (&ps)[RPS][LVI], int (&po)[RPS], int& c0s, int& pbs, u32x4 (&ws)[SBV]) __attribute__((always_inline)) {
            sem_wait(sem_done, (unsigned)gs, done_seen);       // every consumer wave has left K-step gs - 1
            const int u = gs + NSLOT - 1;
            if (u < GS && !(PDBG(a) & 1)) write_w(ws, u % NSLOT);
            if (!(PDBG(a) & 1)) load_w(ws, u + UW);
            if (gs / KROWS + 1 < GCH && !(PDBG(a) & 32)) put_part(ps, po, c0s, pbs, gs);
            sem_signal(sem_ready, cw, (unsigned)gs + 1u, lane);   // this wave's share of producer steps 0 .. gs is in LDS
            if (!(PDBG(a) & 2)) load_part(ps, po, c0s, pbs, gs + U);
        };
        // The previous tile's epilogue rides between the chunks of the current one (NOT inside stage_step: its body is unrolled
        // once per position of the register sets and has to stay small).  It MUST be complete before the consumers want to dump
        // the current tile -- they wait for the image to be free, and this wave cannot get further than one step past them.
        int t_cur = 0, ci_cur = 0;                        // tile / chunk of the producer step stream
        auto after_chunks = [&](int n) __attribute__((always_inline)) {
            ci_cur += n;
            if (drain_t < t_cur) drain_slot(NCH - ci_cur < FORCE);
            if (ci_cur >= NCH) { ci_cur -= NCH; ++t_cur; }
        };
#pragma unroll
        for (int u = 0; u < NPRO; ++u)
            if (u < GS) write_w(wpro[u], u);
#pragma unroll
        for (int q = 0; q < UW; ++q) load_w(wset[(NPRO + q) % UW], NPRO + q);   // set j: the first K-step u >= NSLOT - 1 with u % UW == j (in increasing u)
#pragma unroll
        for (int j = 0; j < U; ++j) load_part(pset[j], oset[j], c0set[j], pbset[j], j);
        wg_barrier();   // chunk 0 and the first K-steps are in LDS
        stamp(1);
        if constexpr (KROWS > 1) {
#pragma unroll 1
            for (int gc = 0; gc < GCH; ++gc) {
#pragma unroll
                for (int k = 0; k < KROWS; ++k) stage_step(gc * KROWS + k, pset[k % U], oset[k % U], c0set[k % U], pbset[k % U], wset[(k + NPRO) % UW]);
                after_chunks(1);
            }
        } else {
            // a chunk is one K-step; NCH is even for every layer of the encoder but need not be: a pair may straddle two tiles,
            // after_chunks then sees the first tile's last chunk and the forced slot (NCH - ci_cur < FORCE covers ci_cur > NCH too)
            int gs = 0;
#pragma unroll 1
            for (; gs + 1 < GS; gs += 2) {
                stage_step(gs, pset[0], oset[0], c0set[0], pbset[0], wset[NPRO % 2]);
                stage_step(gs + 1, pset[1], oset[1], c0set[1], pbset[1], wset[(NPRO + 1) % 2]);
                after_chunks(2);
            }
            if (gs < GS) { stage_step(gs, pset[0], oset[0], c0set[0], pbset[0], wset[NPRO % 2]); after_chunks(1); }   // gs is even here
        }
        stamp(2);
        // a tile before the last that is still not drained (only when a tile has very few K-steps)
        while (drain_t < ntile - 1) drain_slot(true);
        final_drain(ntile - 1, 0);
        stamp(3);
    } else {
        // =========================== consumers ===========================
        f32x16 acc[MB][NBW];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int q = 0; q < NBW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
        const int half = lane >> 5, px = lane & 31;
        const char* a_ptr = patch + ((wm * MB) * PW + px) * RB + 16 * half;
        const char* b_ptr = bst + (wn * NBW) * 1024 + lane * 16;
        wg_barrier();   // chunk 0 and the first K-steps are in LDS
        stamp(1);
        __builtin_amdgcn_s_setprio(1);   // the matrix pipe first: the producers' arithmetic fills what is left
        u32x4 af[2][MB], afl[2][MB], bf[2][NBW], bfl[2][NBW];
        unsigned ready_seen = 0, free_seen = 0;
        auto load_frag = [&](int fs, int gs, int slot, int ks) __attribute__((always_inline)) {
            const int krow = gs % KROWS, pbuf = (gs / KROWS) & 1;
            const char* bs = b_ptr + slot * SBYTES;
            const char* ar = a_ptr + pbuf * PATCHB + ((TPS == 3) ? (krow * PW) * RB : ((TPS == 1 && TAPS == 9) ? ((krow / 3) * PW + krow % 3) * RB : 0));
            const int t = ks / KGC, kg = ks % KGC;
            const int ky = (TPS == 9) ? t / 3 : 0, kx = (TPS == 9) ? t % 3 : t;
#pragma unroll
            for (int q = 0; q < NBW; ++q) bf[fs][q] = *(const u32x4*)(bs + ((t * KGC + kg) * (NT / 32) + q) * 1024);
            if constexpr (X3) {
#pragma unroll
                for (int m = 0; m < MB; ++m) afl[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + 64 + kg * 32);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) af[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + kg * 32);
#pragma unroll
            for (int q = 0; q < NBW; ++q) bfl[fs][q] = *(const u32x4*)(bs + SB1 + ((t * KGC + kg) * (NT / 32) + q) * 1024);
        };
        constexpr int NRD = (X3 ? 2 : 1) * MB + 2 * NBW, NMF = (X3 ? 3 : 2) * MB * NBW;     // LDS reads / MFMAs of one k-step
        static_assert(NKS % 2 == 0, "fragment double buffer: even k-steps per K-step");
        int slot = 0;
        auto mfma_step = [&](int gs, bool last) __attribute__((always_inline)) {
            const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bool pre = true;
                if (ks + 1 < NKS) load_frag((ks + 1) & 1, gs, slot, ks + 1);
                else if (!last) {
                    // K-step gs + 1 needs producer steps 0 .. gs - NSLOT + 2 (its weights) and, when it opens a chunk, the steps
                    // that staged the chunk's patch (the last of them: LASTK of the previous chunk)
                    const int sn = gs + 1;
                    int need = sn - NSLOT + 2;
                    if (sn % KROWS == 0) { const int np = sn - KROWS + LASTK + 1; need = need > np ? need : np; }
                    if ((int)ready_seen < need) {
                        __builtin_amdgcn_s_setprio(0);      // never hold the priority while waiting for a lower-priority wave
                        sem_wait(sem_ready, (unsigned)need, ready_seen);
                        __builtin_amdgcn_s_setprio(1);
                    }
                    load_frag(0, sn, nslot, 0);
                } else pre = false;
                if (PDBG(a) & 4) continue;
                if constexpr (X3) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int q = 0; q < NBW; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, afl[ks & 1][m]),
                                                                               __builtin_bit_cast(f16x8_t, bf[ks & 1][q]), acc[m][q], 0, 0, 0);
                }
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
            sem_signal(sem_done, cw, (unsigned)gs + 1u, lane);   // every read of K-steps 0 .. gs has been issued
            slot = nslot;
        };
        load_frag(0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int t = 0; t < ntile; ++t) {
#pragma unroll 1
            for (int s = 0; s < S; ++s) mfma_step(t * S + s, t * S + s == GS - 1);
            // ---- tile t is complete in the accumulators: hand it to the producers ----
            __builtin_amdgcn_s_setprio(0);
            sem_wait(sem_free, (unsigned)t, free_seen);           // the image of tile t - 1 has been drained
            const int nt0 = (((first + t) % (tiles * ntn)) / tiles) * NT;
#pragma unroll
            for (int q = 0; q < NBW; ++q) {
                const int ch = (wn * NBW + q) * 32 + px;
                const float bias = a.bias ? a.bias[nt0 + ch] : 0.f;
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        img[((wm * MB + m) * PTW + mfma32_row(r, half)) * NT + ch] = acc[m][q][r] * ASCALE + bias;
                        acc[m][q][r] = 0.f;
                    }
            }
            sem_signal(sem_img, cw, (unsigned)t + 1u, lane);
            if (t + 1 < ntile) __builtin_amdgcn_s_setprio(1);
        }
        stamp(2);
        final_drain(ntile - 1, 1);
        stamp(3);
    }
    stats_flush_wg(decode(0));
    stamp(4);
}

template <typename T, int TAPS, int TH, int NT, int TPS, int NSLOT>
int launch_pp_t(chore_handle* h, const ConvArgs& a, int tpw, hipStream_t s) {
    using G = QGeo<TAPS, TH, NT, TPS, NSLOT, IS_X3<T>>;
    size_t smem = G::smem_bytes(a.in.C);
    if (h->lds_per_cu <= 0) {
        CHORE_HIP_CHECK(h, hipDeviceGetAttribute(&h->lds_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, h->device));
        if (h->lds_per_cu <= 0) h->lds_per_cu = 160 * 1024;
    }
    const int lds_cu = h->lds_per_cu;
    if (smem > (size_t)lds_cu) CHORE_FAIL(h, CHORE_EINVAL, "conv_pp: %zu bytes of LDS, the CU has %d", smem, lds_cu);
    if (smem < (size_t)lds_cu / 2 + 1024) smem = (size_t)lds_cu / 2 + 1024;      // one workgroup per CU (the hand-over polls: conv_pc.hip)
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)conv_pp_kernel<T, TAPS, TH, NT, TPS, NSLOT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_cu));
        int per_cu = 0;
        CHORE_HIP_CHECK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_pp_kernel<T, TAPS, TH, NT, TPS, NSLOT>,
                                                                        512, smem));
        if (per_cu != 1)
            CHORE_FAIL(h, CHORE_EINVAL, "conv_pp: %d workgroups per CU with %zu bytes of LDS -- the hand-over by polling needs exactly one",
                       per_cu, smem);
        attr = true;
    }
    const int tiles = ((a.W + PTW - 1) / PTW) * ((a.H + TH - 1) / TH), ntn = a.Cout / NT;
    const int total = tiles * ntn * a.B;
    if (tpw < 1 || tiles % tpw) CHORE_FAIL(h, CHORE_EINVAL, "conv_pp: %d tiles per workgroup do not divide the %d pixel tiles of an image", tpw, tiles);
    dim3 grid((total + tpw - 1) / tpw);
    hipLaunchKernelGGL((conv_pp_kernel<T, TAPS, TH, NT, TPS, NSLOT>), grid, dim3(512), smem, s, a, tpw);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

}  // namespace

// persistent tiling of a layer: th = 0 -> not covered (the caller uses conv_pc_kernel / conv_lds_kernel).  A layer is taken when
// it has at least two tiles per CU in a tiling whose image fits beside the patch buffers and the ring.
PpPlan conv_pp_plan(int dtype, int taps, int B, int H, int W, int Cin, int Cout, int force) {
    PpPlan p{0, 0, 0, 0, 0};
    if ((dtype != CHORE_F16X3 && dtype != CHORE_F16) || Cin % 32 || Cout % 32 || W % 32) return p;
    auto finish = [&](PpPlan& q) {
        if (taps == 1) { q.tps = 1; q.nslot = 2; }
        else if (q.nt == 128) { q.tps = 1; q.nslot = 2; }
        else { q.tps = 3; q.nslot = 2; }
        if (H % q.th || Cout % q.nt) { q.th = 0; return; }
        const int per_image = (H / q.th) * (W / 32);       // pixel tiles of an image
        const long total = (long)per_image * (Cout / q.nt) * B;
        int tpw = (int)((total + 255) / 256);
        if (tpw > 16) tpw = 16;
        while (tpw > 1 && per_image % tpw) --tpw;          // a workgroup's tiles: one image, one channel tile
        q.tpw = tpw;
        if (tpw < 2 && !force) q.th = 0;                   // one tile per CU: nothing to overlap, conv_pc_kernel's epilogue is the faster one
    };
    if (force) {   // development: th * 1000 + nt
        p.th = force / 1000; p.nt = force % 1000;
        finish(p);
        return p;
    }
    if (taps == 1) {
        if (Cout % 128 == 0) { p.th = 4; p.nt = 128; }
        else return p;
        finish(p);
        return p;
    }
    if (Cout % 128 == 0) { p.th = 4; p.nt = 128; }
    else if (Cout % 64 == 0) { p.th = 4; p.nt = 64; }
    else { p.th = 8; p.nt = 32; }
    finish(p);
    return p;
}

int launch_conv_pp(chore_handle* h, int dtype, int taps, const PpPlan& p, const ConvArgs& a, hipStream_t s) {
    if (a.res2.p) CHORE_FAIL(h, CHORE_EINVAL, "conv_pp: a second residual is not supported (conv_lds_kernel has it)");
    const int key = ((taps * 10 + p.th) * 1000 + p.nt) * 100 + p.tps * 10 + p.nslot;
#define PP_CASE(TAPS, TH, NT, TPS, NSLOT) \
    case ((TAPS * 10 + TH) * 1000 + NT) * 100 + TPS * 10 + NSLOT:                                              \
        return dtype == CHORE_F16 ? launch_pp_t<h16_t, TAPS, TH, NT, TPS, NSLOT>(h, a, p.tpw, s) : launch_pp_t<x3_t, TAPS, TH, NT, TPS, NSLOT>(h, a, p.tpw, s)
    switch (key) {
        PP_CASE(9, 4, 128, 1, 2);
        PP_CASE(9, 4, 64, 3, 2);
        PP_CASE(9, 8, 32, 3, 2);
        PP_CASE(9, 4, 32, 3, 2);
        PP_CASE(1, 4, 128, 1, 2);
    }
#undef PP_CASE
    CHORE_FAIL(h, CHORE_EINVAL, "conv_pp: no kernel for taps=%d th=%d nt=%d tps=%d nslot=%d", taps, p.th, p.nt, p.tps, p.nslot);
}

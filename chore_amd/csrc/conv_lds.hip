// conv_lds.hip -- 3x3 / 1x1 convolution as an implicit GEMM on the gfx950 matrix cores
// (every conv of ConvBlock, model/net_util.py:346-396, and the 1x1 convs of the stack tail,
// model/HGFilters.py:128-142,167-183), with GroupNorm-apply + ReLU fused into the operand staging and
// bias / residuals / concat-slice / raw copy / GroupNorm statistics fused into the epilogue.
//
// GEMM view: M = output pixels, N = Cout, K = taps x Cin.  One workgroup (4 waves) owns a tile of
// 8 rows x 32 pixels and NT output channels.
//   * A operand: the tile + 1-pixel halo of one 64-byte channel chunk (32 bf16 / 16 fp32 channels)
//     is staged ONCE in LDS ([340 rows][80 B]) and re-read at the 9 tap shifts -- no im2col.  While
//     it is staged the prologue applies relu(x*scale+shift) (GroupNorm folded to a per-(image,channel)
//     affine) and writes zeros for the halo outside the image (= conv2d's zero padding of the
//     normalised tensor).  A 32-pixel tile row is one MFMA pixel block: its 32 LDS rows, 80 B apart,
//     land on 16 distinct 16-byte slots per ds_read_b128 lane group -> conflict-free.
//   * B operand: fragment-ordered weights ([tap][k-group][n-block][lane][16 B], written once by
//     pack_conv_weights) stream through a 2-slot LDS ring, one K-step (= kernel row of 3 taps x 2
//     k-groups) at a time: fetched by the whole workgroup into registers while the previous step's
//     MFMAs run, parked in the other slot, read as fragments by all waves.
//   * Each wave owns (MB pixel rows) x (NBW channel blocks of 32): 4x2 / 2x2 / 2x1 accumulators of
//     v_mfma_f32_32x32x16_bf16 (T = bf16) or 4x v_mfma_f32_32x32x2_f32 (T = fp32, exact).
//   * Workgroups start the K loop at different chunks (rotation by tile index) so the grid does not
//     hammer the same weight bytes at once; the order is a function of the tile index only, so
//     results are deterministic and independent of the batch composition.
//   * Epilogue: accumulators are transposed through a wave-private LDS scratch so that every lane
//     handles 8 consecutive channels of one pixel: residual loads and all stores are 16-byte
//     vectors covering full 128-byte lines (the first version stored 2-byte elements and spent
//     37 % of the convolution time there).  GroupNorm partial sums of the
//     stored values are reduced in a fixed order.
#include "conv_common.h"

// Ablation switches for kernel experiments (scripts/conv_ablate.sh) exist only in builds with -DCHORE_CONV_ABLATE=1; in
// the shipped library DBG(a) is the constant 0 and every switch below folds away.
#ifndef CHORE_CONV_ABLATE
#define CHORE_CONV_ABLATE 0
#endif
#if CHORE_CONV_ABLATE
#define DBG(a) ((a).dbg)
#else
#define DBG(a) 0
#endif

using namespace conv_detail;

namespace {

constexpr int TH = 8, TW = 32;   // tile
constexpr int ROWB = 80;         // LDS patch row: 64 B of channels + 16 B pad (5 slots: odd)
constexpr int SCR_LD = 68;       // scratch row stride in floats (64 channels + 4: conflict-free b128 reads)


// TPS = taps per K-step: 3 (one kernel row) for large grids; 9 (the whole chunk) for the small maps, where a
// workgroup's time is a chain of dependent weight fetches and fewer, longer steps mean fewer round trips
template <int TAPS, int NT, int TPS_, bool X3 = false> struct Geo {
    static constexpr int PAD = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = TW + 2 * PAD, PH = TH + 2 * PAD, ROWS = PH * PW;
    static constexpr int TPS = TPS_;                                // taps per K-step
    static constexpr int RB = X3 ? 144 : ROWB;                      // LDS patch row: x3 = 64 B hi + 64 B lo + 16 B pad (9 slots: odd)
    static constexpr int SB1 = TPS * KGC * (NT / 32) * 1024;        // bytes of one operand plane of a K-step
    static constexpr int SBYTES = (X3 ? 2 : 1) * SB1;               // weight bytes per K-step (x3: hi plane, then lo plane)
    static constexpr int NBW = NT >= 64 ? 2 : 1;
    static constexpr int WAVES_N = (NT / 32) / NBW, WAVES_M = 4 / WAVES_N, MB = TH / WAVES_M;
    static constexpr int NPATCH = (TPS_ == 9 && TAPS == 9) ? 2 : 1;   // small grids: double-buffered patch, one barrier per chunk
    static constexpr size_t main_bytes(int Cin) { return (size_t)NPATCH * ROWS * RB + 2 * SBYTES + (size_t)Cin * 8; }
    static constexpr size_t epi_bytes() { return (size_t)4 * 32 * SCR_LD * 4 + (size_t)4 * WAVES_M * NT * 4; }
    static size_t smem_bytes(int Cin) { return main_bytes(Cin) > epi_bytes() ? main_bytes(Cin) : epi_bytes(); }
};

constexpr int PD_SMALL = 2;

template <typename T, int TAPS, int NT, int TPS_>
__global__ __launch_bounds__(256, 2) void conv_lds_kernel(ConvArgs a) {
    if constexpr (IS_X3<T> || IS_H16<T>) f16_saturate_mode();     // the fp16 x 3 / fp16 operand split never produces inf (common.h)
    constexpr bool X3 = IS_X3<T>;
    using ST = typename Store<T>::type;                 // element type in memory
    using G = Geo<TAPS, NT, TPS_, X3>;
    constexpr int VE = CT<T>::VE, KGE = CT<T>::KGE;
    constexpr int CC = KGC * KGE;                       // channels per chunk (32 bf16 / 16 fp32 / 32 x3)
    constexpr int LV = X3 ? 2 : 1;                      // 16-byte loads per staging slot
    constexpr int RB = G::RB, SB1 = G::SB1;
    constexpr int PAD = G::PAD, PW = G::PW, ROWS = G::ROWS, TPS = G::TPS, SBYTES = G::SBYTES;
    constexpr int NBW = G::NBW, WAVES_N = G::WAVES_N, WAVES_M = G::WAVES_M, MB = G::MB;
    constexpr int VPR = 4;                              // 16-byte vectors per patch row
    constexpr int NVP = (ROWS * VPR + 255) / 256;       // patch vectors per thread
    constexpr int SVEC = SBYTES / 16, SBV = (SVEC + 255) / 256, SV1 = G::SB1 / 16;   // SV1: vectors of one operand plane
    constexpr int KROWS = TAPS / TPS;                   // K-steps per chunk (3 or 1)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPATCH = G::NPATCH, PATCHB = ROWS * RB;
    char* patch = smem;                                  // [NPATCH][ROWS][ROWB]
    char* bst = smem + NPATCH * PATCHB;                  // [2][SBYTES]
    float* ss_lds = (float*)(bst + 2 * SBYTES);          // [Cin][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % WAVES_N, wm = wid / WAVES_N;
    const int tiles_x = (a.W + TW - 1) / TW;
    float in_mul = 1.f, in_inv = 1.f;                    // operand scale of a gradient input (ConvArgs::in_amax)
    if constexpr (IS_X3S<T>) x3_in_scale(a.in_amax, in_mul, in_inv);
    // XCD-aware placement: the dispatcher puts workgroup id on XCD id % 8 (each XCD has its own L2).  The logical order is
    // (image, pixel tile, channel tile) with the channel tile fastest, and every XCD takes a CONTIGUOUS range of it: the
    // channel tiles of one pixel tile -- which read the same halo patch -- and neighbouring pixel tiles share an L2
    // (bijective for any grid size).  dbg bit 2048 switches it off for A/B measurements.
    const int ntn = a.Cout / NT, tiles = tiles_x * ((a.H + TH - 1) / TH);
    int lid = blockIdx.x;
    if (!(DBG(a) & 2048)) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = lid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
    }
    const int n_tile = lid % ntn, tileb = lid / ntn, tile = tileb % tiles, b = tileb / tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const int Cin = a.in.C;
    const bool use_gn = a.in_st != nullptr;
    const int NKG = Cin / KGE, NB = a.Cout / 32;
    const int NCH = Cin / CC;
    const int S = NCH * KROWS;


    // ---- staging coordinates ----
    const int v = tid & 3;
    const ST* in_b = (const ST*)a.in.p + (size_t)b * a.H * a.W * a.in.cs + a.in.co;
    int row_off[NVP];
#pragma unroll
    for (int j = 0; j < NVP; ++j) {
        const int row = (tid + j * 256) >> 2;
        const int y = ty0 + row / PW - PAD, x = tx0 + row % PW - PAD;
        const bool ok = (row < ROWS) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        row_off[j] = ok ? (y * a.W + x) * a.in.cs : -1;
    }
    // PD = register-prefetch distance in chunks.  With a small grid every CU holds one workgroup and nothing hides a
    // global round trip (~2 us) but the workgroup's own MFMA block (~0.5 us per chunk), so the loads of PD chunks are
    // kept in flight; large grids rely on the co-resident workgroups instead and keep the registers for the tile.
    constexpr int PD = (TPS == 9) ? PD_SMALL : 1;
    u32x4 preq[PD][NVP * LV];
    // loads are unconditional (invalid rows re-read the tile's first pixel and are zeroed at publish time): a
    // branch around a load makes the compiler's vmcnt bookkeeping give up and wait for everything in flight
    auto load_patch = [&](u32x4 (&pre)[NVP * LV], int c0) {
#pragma unroll
        for (int j = 0; j < NVP; ++j) {
            const int off = row_off[j] >= 0 ? row_off[j] : 0;
#pragma unroll
            for (int k = 0; k < LV; ++k) pre[j * LV + k] = *((const u32x4*)(in_b + off + c0 + v * VE) + k);
        }
    };
    auto write_patch = [&](const u32x4 (&pre)[NVP * LV], int c0, int pbuf = 0) {
        float sc[VE], sh[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            sc[j] = use_gn ? ss_lds[(c0 + v * VE + j) * 2] : 0.f;
            sh[j] = use_gn ? ss_lds[(c0 + v * VE + j) * 2 + 1] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NVP; ++j) {
            const int idx = tid + j * 256;
            if (idx < ROWS * VPR) {
                u32x4 val = {0u, 0u, 0u, 0u};
                if constexpr (X3) {
                    u32x4 lo = {0u, 0u, 0u, 0u};
                    if (row_off[j] >= 0) {
                        if constexpr (IS_X3S<T>) xform_x3(pre[j * LV], pre[j * LV + 1], sc, sh, use_gn, val, lo, in_mul);
                        else xform_x3(pre[j * LV], pre[j * LV + 1], sc, sh, use_gn, val, lo);
                    }
                    *(u32x4*)(patch + pbuf * PATCHB + (idx >> 2) * RB + v * 16) = val;
                    *(u32x4*)(patch + pbuf * PATCHB + (idx >> 2) * RB + 64 + v * 16) = lo;
                } else {
                    if (row_off[j] >= 0) val = xform<T>(pre[j], sc, sh, use_gn);
                    *(u32x4*)(patch + pbuf * PATCHB + (idx >> 2) * RB + v * 16) = val;
                }
            }
        }
    };
    // weight slice of K-step (chunk c, kernel row krow): [t][kg][nb][lane] vectors
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)(n_tile * (NT / 32)) * 64;
    const size_t wkg = (size_t)NB * 64;   // vectors between consecutive k-groups
    u32x4 rbq[PD][SBV];
    int woff[SBV];   // per-thread vector offsets inside a K-step slice (step-independent)
#pragma unroll
    for (int j = 0; j < SBV; ++j) {
        const int i0 = (tid + j * 256 < SVEC) ? tid + j * 256 : SVEC - 1;
        const int i = i0 % SV1;                 // position inside the operand plane
        constexpr int PER_KG = (NT / 32) * 64;
        const int t = i / (KGC * PER_KG), kg = (i / PER_KG) % KGC, r = i % PER_KG;
        woff[j] = (t * NKG + kg) * (int)wkg + r;
        if (X3 && i0 >= SV1) woff[j] += TAPS * NKG * (int)wkg;   // the lo plane follows the complete hi plane in memory
    }
    auto load_w = [&](u32x4 (&rb)[SBV], int c, int krow) {
        const u32x4* wb = wbase + (size_t)(krow * TPS * NKG + c * KGC) * wkg;   // wave-uniform
#pragma unroll
        for (int j = 0; j < SBV; ++j) rb[j] = wb[woff[j]];
    };
    auto write_w = [&](const u32x4 (&rb)[SBV], int slot) {
#pragma unroll
        for (int j = 0; j < SBV; ++j) {
            const int i = tid + j * 256;
            if (i < SVEC) *(u32x4*)(bst + slot * SBYTES + i * 16) = rb[j];
        }
    };

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

    const int crot = (tile * 5 + n_tile * 3) % NCH;
    auto chunk_of = [&](int ci) -> int { int x = ci + crot; return x >= NCH ? x - NCH : x; };

    // ---- prologue: first chunk's patch and K-step 0 weights (and the next PD-1 chunks) ----
#pragma unroll
    for (int q = 0; q < PD; ++q) {
        const int cq = q < NCH ? q : NCH - 1;
        load_patch(preq[q], chunk_of(cq) * CC);
        load_w(rbq[q], chunk_of(cq), 0);
    }
    if (use_gn && !(DBG(a) & 64)) {
        // GroupNorm affine of this image's input channels from the producers' exact group totals
        for (int ci = tid; ci < Cin; ci += 256)
            gn_scale_shift(a.in_st, a.B, b, Cin, ci, a.H * a.W, a.gamma, a.beta, ss_lds[2 * ci], ss_lds[2 * ci + 1]);
    }
    __syncthreads();   // ss_lds visible, scratch free
    write_patch(preq[0], chunk_of(0) * CC);
    write_w(rbq[0], 0);
    __syncthreads();

    // the MFMAs of one K-step: weights from ring slot `slot`, patch rows shifted by kernel row `krow`.  The
    // fragments of k-step ks+FD are requested before the MFMAs of k-step ks are issued and the scheduler is fenced
    // so that it keeps that order: left alone it sinks the ds_reads to just before their use and every other MFMA
    // pair then waits a full LDS round trip (one wave per SIMD: nothing else hides it).
    constexpr int NKS = TPS * KGC;                     // k-steps (one MFMA K each) per K-step
    constexpr int FD = (NT >= 128 || NKS < 3 || X3) ? 1 : 2; // fragment prefetch distance
    auto mfma_step = [&](int slot, int krow, int pbuf = 0) {
        const char* bs = b_ptr + slot * SBYTES;
        const char* ar = a_ptr + pbuf * PATCHB + ((TPS == 9) ? 0 : (krow * PW) * RB);
        u32x4 af[FD + 1][MB], bf[FD + 1][NBW];
        u32x4 afl[X3 ? FD + 1 : 1][MB], bfl[X3 ? FD + 1 : 1][NBW];     // fp16 x 3: the lo planes
        auto load_frag = [&](int fs, int ks) {
            const int t = ks / KGC, kg = ks % KGC;
            const int ky = (TPS == 9) ? t / 3 : 0, kx = (TPS == 9) ? t % 3 : t;
#pragma unroll
            for (int q = 0; q < NBW; ++q) {
                bf[fs][q] = *(const u32x4*)(bs + ((t * KGC + kg) * (NT / 32) + q) * 1024);
                if constexpr (X3) bfl[fs][q] = *(const u32x4*)(bs + SB1 + ((t * KGC + kg) * (NT / 32) + q) * 1024);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                af[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + kg * 32);
                if constexpr (X3) afl[fs][m] = *(const u32x4*)(ar + ((m + ky) * PW + kx) * RB + 64 + kg * 32);
            }
        };
#pragma unroll
        for (int d = 0; d < FD; ++d)
            if (d < NKS) load_frag(d, d);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + FD < NKS) load_frag((ks + FD) % (FD + 1), ks + FD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < NBW; ++q) {
                    if constexpr (X3)
                        mfma_x3(acc[m][q], af[ks % (FD + 1)][m], afl[ks % (FD + 1)][m], bf[ks % (FD + 1)][q], bfl[ks % (FD + 1)][q]);
                    else
                        mfma<T>(acc[m][q], af[ks % (FD + 1)][m], bf[ks % (FD + 1)][q]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (PD > 1) {
        // one K-step per chunk; chunk c+1+j sits in register set (c+1+j) % PD.  The loop is unrolled by PD so the
        // sets are named statically; the loads are unconditional (the tail re-fetches the last chunk) so that the
        // wait before a publish counts only the loads issued after the ones it needs.  NCH % PD == 0 (launcher).
#pragma unroll 1
        for (int c0 = 0; c0 < ((DBG(a) & 128) ? 0 : NCH); c0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int c = c0 + u;
                const int cn = c + PD < NCH ? c + PD : NCH - 1;
                if (!(DBG(a) & 2)) load_patch(preq[u], chunk_of(cn) * CC);
                if (!(DBG(a) & 1)) load_w(rbq[u], chunk_of(cn), 0);
                __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMA block
                if (!(DBG(a) & 4)) mfma_step(c & 1, 0, NPATCH == 2 ? (c & 1) : 0);
                if (c + 1 < NCH) {
                    write_w(rbq[(u + 1) % PD], (c + 1) & 1);
                    if constexpr (NPATCH == 2) {
                        // the other patch buffer and ring slot were last read for chunk c-1, which every wave left
                        // before the barrier that ended that iteration: no barrier needed before overwriting them
                        if (!(DBG(a) & 32)) write_patch(preq[(u + 1) % PD], chunk_of(c + 1) * CC, (c + 1) & 1);
                    } else {
                        __syncthreads();   // every wave has finished reading this chunk's patch
                        if (!(DBG(a) & 32)) write_patch(preq[(u + 1) % PD], chunk_of(c + 1) * CC);
                    }
                }
                __syncthreads();
            }
        }
    } else {
        int c = 0, krow = 0;   // c counts chunks in visiting order
#pragma unroll 1
        for (int s = 0; s < ((DBG(a) & 128) ? 0 : S); ++s) {
            const bool last_row = (krow == KROWS - 1);
            const bool more_chunks = (c + 1 < NCH);
            // (a) prefetch the next K-step's weights (and the next chunk's patch) into registers; they have
            //     the whole MFMA block below to land
            if (s + 1 < S && !(DBG(a) & 1)) load_w(rbq[0], chunk_of(last_row ? c + 1 : c), last_row ? 0 : krow + 1);
            if (krow == 0 && more_chunks && !(DBG(a) & 2)) load_patch(preq[0], chunk_of(c + 1) * CC);
            // (b) the MFMAs of this K-step
            if (!(DBG(a) & 4)) mfma_step(s & 1, krow);
            // (c) publish the next step's operands
            if (s + 1 < S) write_w(rbq[0], (s + 1) & 1);
            if (last_row && more_chunks) {
                __syncthreads();   // every wave has finished reading this chunk's patch
                write_patch(preq[0], chunk_of(c + 1) * CC);
            }
            __syncthreads();
            if (last_row) { krow = 0; ++c; } else ++krow;
        }
    }
    if (DBG(a) & 8) return;

    // ---- epilogue (the loop ended with a barrier: patch and ring are dead, reuse them) ----
    // undoes the weight scaling of the fp16 x 3 packing (and, x3s_t, the operand scale)
    const float ASCALE = IS_X3S<T> ? in_inv / (float)(1 << X3_WSHIFT) : (X3 ? 1.0f / (float)(1 << X3_WSHIFT) : 1.0f);
    float* scr = (float*)smem + wid * (32 * SCR_LD);          // wave-private [32 pixels][SCR_LD]
    float* red = (float*)smem + 4 * 32 * SCR_LD;              // [4][WAVES_M][NT]
    constexpr int CW = NBW * 32;                              // channels of this wave
    constexpr int G8 = CW / 8;                                // 8-channel groups per pixel (8 or 4)
    constexpr int NVE = 32 * G8 / 64;                         // vectors per lane per pixel row (4 or 2)
    const int nw0 = (n_tile * (NT / 32) + wn * NBW) * 32;     // first channel of this wave
    const int g8 = lane % G8;
    const int nv = nw0 + g8 * 8;                              // this lane's 8 channels
    float bias[NBW];
#pragma unroll
    for (int q = 0; q < NBW; ++q) bias[q] = a.bias ? a.bias[nw0 + q * 32 + px] : 0.f;
    const size_t img = (size_t)b * a.H * a.W;
    ST* out_p = (ST*)a.out.p + img * a.out.cs + a.out.co + nv;
    ST* raw_p = a.raw.p ? (ST*)a.raw.p + img * a.raw.cs + a.raw.co + nv : nullptr;
    const ST* res_p = (a.res.p && !(DBG(a) & 1024)) ? (const ST*)a.res.p + img * a.res.cs + a.res.co + nv : nullptr;
    const ST* res2_p = (a.res2.p && !(DBG(a) & 1024)) ? (const ST*)a.res2.p + img * a.res2.cs + a.res2.co + nv : nullptr;
    const bool want_stats = (a.st_raw || a.st_out) && !(DBG(a) & 256);
    float sr[8], qr[8], so[8], qo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sr[e] = qr[e] = so[e] = qo[e] = 0.f; }

    // residual vectors of pixel row m are fetched one row ahead of their use (raw 16/32-byte loads)
    constexpr int RV = sizeof(ST) == 2 ? 1 : 2;               // u32x4 per 8 channels
    u32x4 rq[2][NVE][RV], rq2[2][NVE][RV];
    auto fetch_res = [&](int m, int slot) {
        const int y = ty0 + wm * MB + m;
#pragma unroll
        for (int j = 0; j < NVE; ++j) {
            const int x = tx0 + (lane + 64 * j) / G8;
            const bool ok = (y < a.H) && (x < a.W);
            const size_t pix = (size_t)y * a.W + x;
#pragma unroll
            for (int k = 0; k < RV; ++k) {
                u32x4 z = {0u, 0u, 0u, 0u};
                rq[slot][j][k] = (res_p && ok) ? *((const u32x4*)(res_p + pix * a.res.cs) + k) : z;
                rq2[slot][j][k] = (res2_p && ok) ? *((const u32x4*)(res2_p + pix * a.res2.cs) + k) : z;
            }
        }
    };
    auto add_res = [&](float (&f)[8], const u32x4 (&v)[RV]) {
        if constexpr (sizeof(ST) == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[2 * k] += __uint_as_float(v[0][k] << 16);
                f[2 * k + 1] += __uint_as_float(v[0][k] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[k] += __uint_as_float(v[0][k]);
                f[4 + k] += __uint_as_float(v[RV - 1][k]);
            }
        }
    };
    fetch_res(0, 0);

#pragma clang loop unroll(full)
    for (int m = 0; m < MB; ++m) {
        if (m + 1 < MB) fetch_res(m + 1, (m + 1) & 1);
#pragma clang loop unroll(full)
        for (int q = 0; q < NBW; ++q)
#pragma clang loop unroll(full)
            for (int r = 0; r < 16; ++r)
                scr[mfma32_row(r, half) * SCR_LD + q * 32 + px] = acc[m][q][r] * ASCALE + bias[q];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int y = ty0 + wm * MB + m;
#pragma clang loop unroll(full)
        for (int j = 0; j < NVE; ++j) {
            const int p = (lane + 64 * j) / G8;               // pixel of the row
            const int x = tx0 + p;
            float f[8];
            {
                const f32x4 lo = *(const f32x4*)(scr + p * SCR_LD + g8 * 8), hi = *(const f32x4*)(scr + p * SCR_LD + g8 * 8 + 4);
                f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            }
            if (y < a.H && x < a.W && !(DBG(a) & 512)) {
                const size_t pix = (size_t)y * a.W + x;
                if (raw_p) {
                    float g[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = f[e];
                    store8<ST>(raw_p + pix * a.raw.cs, g);
                    if (want_stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { sr[e] += g[e]; qr[e] += g[e] * g[e]; }
                    }
                }
                if (res_p) add_res(f, rq[m & 1][j]);
                if (res2_p) add_res(f, rq2[m & 1][j]);
                store8<ST>(out_p + pix * a.out.cs, f);
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { so[e] += f[e]; qo[e] += f[e] * f[e]; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    if (want_stats) {   // uniform over the grid
        // lanes with equal (lane % G8) hold the same 8 channels: fixed-order butterfly over the rest
#pragma unroll
        for (int o = G8; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sr[e] += __shfl_xor(sr[e], o, 64); qr[e] += __shfl_xor(qr[e], o, 64);
                so[e] += __shfl_xor(so[e], o, 64); qo[e] += __shfl_xor(qo[e], o, 64);
            }
        }
        __syncthreads();   // all waves are done with their scratch reads (red is disjoint, but keep it simple)
        if (lane < G8) {
            const int cl = wn * CW + lane * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(0 * WAVES_M + wm) * NT + cl + e] = sr[e];
                red[(1 * WAVES_M + wm) * NT + cl + e] = qr[e];
                red[(2 * WAVES_M + wm) * NT + cl + e] = so[e];
                red[(3 * WAVES_M + wm) * NT + cl + e] = qo[e];
            }
        }
        __syncthreads();
        if (tid < NT) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int w = 0; w < WAVES_M; ++w) t[k] += red[(k * WAVES_M + w) * NT + tid];
            const int cg = n_tile * NT + tid;
            if (DBG(a) & 16) return;
            // channels -> GroupNorm groups of the tensor the slice belongs to (gs consecutive lanes).  An add is an atomic
            // round trip (its carry needs the old value), so the (up to) four sums of a group go out from DIFFERENT lanes
            // of the group, concurrently, instead of one lane waiting for them in turn.
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
                if (tid % gs == (gs > 3 ? 2 : 0)) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 3 ? 3 : (gs > 1 ? 1 : 0))) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
        }
    }
}

template <typename T, int TAPS, int NT, int TPS_>
int launch_t(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    using G = Geo<TAPS, NT, TPS_, IS_X3<T>>;
    const size_t smem = G::smem_bytes(a.in.C);
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)conv_lds_kernel<T, TAPS, NT, TPS_>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        attr = true;
    }
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    dim3 grid(tiles * (a.Cout / NT) * a.B);
    hipLaunchKernelGGL((conv_lds_kernel<T, TAPS, NT, TPS_>), grid, dim3(256), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

template <typename T, int TAPS>
int launch_nt(chore_handle* h, int nt, bool small_grid, const ConvArgs& a, hipStream_t s) {
    if constexpr (TAPS == 9) {
        if constexpr (!IS_X3<T>) {        // fp16 x 3: two LDS planes per operand, the whole-chunk variant would not fit
            if (small_grid && nt == 64) return launch_t<T, 9, 64, 9>(h, a, s);
            if (small_grid && nt == 32) return launch_t<T, 9, 32, 9>(h, a, s);
        }
        if constexpr (sizeof(T) == 2) {   // the 4x2 register tile is bf16-only (choose_nt never picks it for fp32)
            if (nt == 128) return launch_t<T, 9, 128, 3>(h, a, s);
        }
        if (nt == 64) return launch_t<T, 9, 64, 3>(h, a, s);
        return launch_t<T, 9, 32, 3>(h, a, s);
    } else {
        if constexpr (sizeof(T) == 2) {
            if (nt == 128) return launch_t<T, 1, 128, 1>(h, a, s);
        }
        if (nt == 64) return launch_t<T, 1, 64, 1>(h, a, s);
        return launch_t<T, 1, 32, 1>(h, a, s);
    }
}

int tiles_of(int H, int W) { return ((W + TW - 1) / TW) * ((H + TH - 1) / TH); }

// N-tile choice: the largest channel tile that still gives every CU about two workgroups
int choose_nt(int dtype, int B, int H, int W, int Cout) {
    static const int nts[3] = {128, 64, 32};
    int best = 32;
    long best_wgs = -1;
    for (int i = 0; i < 3; ++i) {
        const int nt = nts[i];
        if (Cout % nt) continue;
        if (nt == 128 && dtype != CHORE_BF16) continue;  // the 4x2 register tile fits 256 VGPRs with bf16 operands only
        const long wgs = (long)B * tiles_of(H, W) * (Cout / nt);
        if (wgs >= 448) return nt;
        if (wgs > best_wgs) { best_wgs = wgs; best = nt; }
    }
    return best;
}

}  // namespace

// small grid: fewer workgroups than 1.5 per CU -> whole-chunk K-steps with deep register prefetch (needs the chunk
// count to be a multiple of the prefetch distance)
static bool is_small_grid(int dtype, int B, int H, int W, int Cin, int Cout, int nt) {
    const int nch = Cin / (dtype == CHORE_F32 ? 16 : 32);
    if (dtype == CHORE_F16X3) return false;
    return nt <= 64 && nch % PD_SMALL == 0 && (long)B * tiles_of(H, W) * (Cout / nt) < 384;
}

// CHORE_CONV_LDS=1: every layer on conv_lds_kernel (A/B measurements against the specialised-wave kernel)
bool conv_use_pc() {
    static const bool off = getenv("CHORE_CONV_LDS") != nullptr;
    return !off;
}

ConvPlan conv_plan(int dtype, int taps, int B, int H, int W, int Cin, int Cout) {
    if (conv_small_eligible(dtype, taps, H, W, Cin, Cout)) return ConvPlan{32, 1, H * (W / 32), 0, Cin};
    const int nt = choose_nt(dtype, B, H, W, Cout);
    const int tps = taps == 1 ? 1 : (is_small_grid(dtype, B, H, W, Cin, Cout, nt) ? 9 : 3);
    return ConvPlan{nt, TH, tiles_of(H, W), tps, 0};
}

int launch_conv(chore_handle* h, int dtype, int taps, const ConvArgs& a_in, hipStream_t s) {
    const int cc = dtype == CHORE_F32 ? 16 : 32;
    if (dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3 && dtype != CHORE_F16) CHORE_FAIL(h, CHORE_EINVAL, "conv: bad dtype");
    if (a_in.in.C % cc || a_in.in.C > 256) CHORE_FAIL(h, CHORE_EINVAL, "conv: unsupported Cin=%d", a_in.in.C);
    if (a_in.Cout % 32) CHORE_FAIL(h, CHORE_EINVAL, "conv: unsupported Cout=%d", a_in.Cout);
    if (a_in.B > 65535) CHORE_FAIL(h, CHORE_EINVAL, "conv: B too large");
    if (taps != 1 && taps != 9) CHORE_FAIL(h, CHORE_EINVAL, "conv: taps must be 1 or 9");
    for (int c : {a_in.st_raw ? a_in.st_raw_C : 32, a_in.st_out ? a_in.st_out_C : 32, a_in.in_st ? a_in.in.C : 32}) {
        const int gs = c / GN_GROUPS;
        if (c % GN_GROUPS || gs > 8 || (gs & (gs - 1)))
            CHORE_FAIL(h, CHORE_EINVAL, "conv: GroupNorm over %d channels unsupported (group size must be 1, 2, 4 or 8)", c);
    }
    if (conv_mw_on(dtype, taps) && conv_mw_fill(a_in.fill) < 256 && !a_in.res2.p) {      // the small maps on dense conv_mw tiles (conv_mw_plan)
        const PcPlan mp = conv_mw_plan(dtype, taps, a_in.B, a_in.H, a_in.W, a_in.in.C, a_in.Cout, a_in.fill);
        if (mp.th && conv_mw_covers(dtype, taps, mp, a_in)) return launch_conv_mw(h, dtype, taps, mp, a_in, s);
    }
    if (conv_small_eligible(dtype, taps, a_in.H, a_in.W, a_in.in.C, a_in.Cout)) return launch_conv_small(h, dtype, a_in, s);
    // 1x1 (every 16-bit-operand mode): weights resident in registers, persistent workgroups (conv_rw.hip; fp16 x 3 256 -> 256 at 128^2: 64 -> 39 us)
    if (conv_rw_eligible(dtype, taps, a_in) && conv_use_pc()) return launch_conv_rw(h, dtype, a_in, s);
    if (dtype == CHORE_F16) {   // fp16 tensors: the specialised-wave kernel is the only implementation
        const PcPlan pp = conv_pc_plan(dtype, taps, a_in.B, a_in.H, a_in.W, a_in.in.C, a_in.Cout);
        if (!pp.th || a_in.res2.p) CHORE_FAIL(h, CHORE_EINVAL, "conv: layer not covered in the fp16 mode (Cin=%d Cout=%d)", a_in.in.C, a_in.Cout);
        return launch_conv_pc(h, dtype, taps, pp, a_in, s);
    }
    // specialised-wave kernel (conv_pc.hip): fp16 x 3; since round 5 also bf16, where it measured faster layer by layer
    // (profiles/r05_conv_layer_ab.txt: 3x3 layers of <= 128 input channels on maps up to 128^2 -- 64^2 128->64 15.4 against 20.2 us,
    // 64->64 13.2 against 15.7; the 1x1 layers, the 256-channel inputs and the 256^2 maps stay on conv_lds_kernel, which wins there:
    // with one MFMA per product the bf16 K loop is short and the producer / consumer hand-over costs more than it hides).
    // CHORE_CONV_LDS_BF16=1: conv_lds_kernel everywhere (rounds 1 - 4); CHORE_CONV_PC_BF16=1: conv_pc_kernel wherever it has a tiling
    static const bool bf16_lds = getenv("CHORE_CONV_LDS_BF16") != nullptr, bf16_pc_all = getenv("CHORE_CONV_PC_BF16") != nullptr;
    const bool bf16_pc = dtype == CHORE_BF16 && !bf16_lds &&
                         (bf16_pc_all || (taps == 9 && a_in.in.C <= 128 && (long)a_in.H * a_in.W <= 128 * 128));
    if ((dtype == CHORE_F16X3 || bf16_pc) && !a_in.res2.p && conv_use_pc()) {
        if (conv_mw_on(dtype, taps)) {
            const PcPlan mp = conv_mw_plan(dtype, taps, a_in.B, a_in.H, a_in.W, a_in.in.C, a_in.Cout, a_in.fill);
            if (mp.th && conv_mw_covers(dtype, taps, mp, a_in)) return launch_conv_mw(h, dtype, taps, mp, a_in, s);
        }
        const PcPlan pp = conv_pc_plan(dtype, taps, a_in.B, a_in.H, a_in.W, a_in.in.C, a_in.Cout);
        if (pp.th) return launch_conv_pc(h, dtype, taps, pp, a_in, s);
    }
    ConvArgs a = a_in;
#if CHORE_CONV_ABLATE
    static const int dbg = getenv("CHORE_CONV_DBG") ? atoi(getenv("CHORE_CONV_DBG")) : 0;
    a.dbg = (a_in.dbg & (1 << 30)) ? (a_in.dbg & ~(1 << 30)) : dbg;   // bit 30: the caller's switches (scripts/probes/conv_bench.hip)
#endif
    const int nt = choose_nt(dtype, a.B, a.H, a.W, a.Cout);
    const bool small_grid = is_small_grid(dtype, a.B, a.H, a.W, a.in.C, a.Cout, nt);
    if (dtype == CHORE_F32)
        return taps == 9 ? launch_nt<float, 9>(h, nt, small_grid, a, s) : launch_nt<float, 1>(h, nt, small_grid, a, s);
    if (dtype == CHORE_F16X3)
        return a.in_amax ? (taps == 9 ? launch_nt<x3s_t, 9>(h, nt, false, a, s) : launch_nt<x3s_t, 1>(h, nt, false, a, s))
                         : (taps == 9 ? launch_nt<x3_t, 9>(h, nt, false, a, s) : launch_nt<x3_t, 1>(h, nt, false, a, s));
    return taps == 9 ? launch_nt<bf16_t, 9>(h, nt, small_grid, a, s) : launch_nt<bf16_t, 1>(h, nt, small_grid, a, s);
}

// ------------------------------------------------------------------------------------------------
// weight packing: (O,C,kh,kw) fp32 -> [tap][kg][nb][lane][16 B]
//   bf16: lane l holds W[n = nb*32 + (l&31)][c = kg*16 + 8*(l>>5) + 0..7]
//   fp32: lane l holds W[n][c = kg*8 + 4*(l>>5) + 0..3]
// ------------------------------------------------------------------------------------------------
size_t packed_conv_bytes(int dtype, int taps, int Cin, int Cout) {
    const int kge = dtype == CHORE_F32 ? 8 : 16;
    return (size_t)taps * (Cin / kge) * (Cout / 32) * 1024 * ((dtype == CHORE_F16X3 || dtype == CHORE_F16) ? 2 : 1);   // fp16 x 3 / fp16: hi plane + lo plane
}

// transposed = 1 packs the weights of the DATA-GRADIENT convolution of a layer whose forward weights are
// w (O = Cin here, C = Cout here, taps): dX = conv(dY, w^T flipped), i.e. this conv's w'[n][c][tap] = w[c][n][taps-1-tap]
template <typename T>
__device__ __forceinline__ void pack_conv_vec(size_t i, int taps, int Cin, int Cout, const float* __restrict__ w, u32x4* __restrict__ dst,
                                              size_t nvec, int transposed) {
    constexpr int VE = CT<T>::VE, KGE = CT<T>::KGE;
    if (i >= nvec) return;
    const int NKG = Cin / KGE, NB = Cout / 32;
    const int lane = (int)(i & 63);
    size_t t = i >> 6;
    const int nb = (int)(t % NB); t /= NB;
    const int kg = (int)(t % NKG);
    const int tap = (int)(t / NKG);
    const int n = nb * 32 + (lane & 31);
    const int c0 = kg * KGE + VE * (lane >> 5);
    float vals[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j)
        vals[j] = transposed ? w[((size_t)(c0 + j) * Cout + n) * taps + (taps - 1 - tap)]
                             : w[((size_t)n * Cin + c0 + j) * taps + tap];
    u32x4 o;
    if constexpr (IS_X3<T>) {
        // fp16 x 3: hi plane at i, lo plane `nvec` vectors further; weights scaled by 2^X3_WSHIFT (exact)
        f16x8_t hh, ll;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ws = vals[j] * (float)(1 << X3_WSHIFT);
            hh[j] = (_Float16)ws;
            ll[j] = (_Float16)(ws - (float)hh[j]);
        }
        dst[i] = __builtin_bit_cast(u32x4, hh);
        dst[i + nvec] = __builtin_bit_cast(u32x4, ll);
        return;
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (unsigned)f2bf(vals[2 * j]) | ((unsigned)f2bf(vals[2 * j + 1]) << 16);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __float_as_uint(vals[j]);
    }
    dst[i] = o;
}

template <typename T>
__global__ void pack_conv_kernel(int taps, int Cin, int Cout, const float* __restrict__ w, u32x4* __restrict__ dst,
                                 size_t nvec, int transposed) {
    pack_conv_vec<T>((size_t)blockIdx.x * 256 + threadIdx.x, taps, Cin, Cout, w, dst, nvec, transposed);
}

// several convolutions' weights (and an optional region to clear) in ONE launch: a ConvBlock's training operator packs its
// three or four weights and zeroes its accumulators with this instead of up to five launches
template <typename T>
__global__ void pack_conv_multi_kernel(PackJobs j) {
    unsigned blk = blockIdx.x;
#pragma unroll
    for (int k = 0; k < PackJobs::MAXJ; ++k) {
        if (k >= j.n) break;
        if (blk < j.job[k].blocks) {
            pack_conv_vec<T>((size_t)blk * 256 + threadIdx.x, j.job[k].taps, j.job[k].Cin, j.job[k].Cout, j.job[k].w,
                             (u32x4*)j.job[k].dst, j.job[k].nvec, j.job[k].transposed);
            return;
        }
        blk -= j.job[k].blocks;
    }
    const unsigned zb = (unsigned)((j.zero_vecs + 255) / 256);
    if (blk >= zb) {                                        // the last AMAX_CELLS workgroups: max |x| of j.amax_x
        absmax_block(j.amax_x, j.amax_n4, j.amax_cells, blk - zb);
        return;
    }
    const size_t i = (size_t)blk * 256 + threadIdx.x;      // the clear region, 16 bytes per thread
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (i < j.zero_vecs) ((u32x4*)j.zero)[i] = z;
}

int launch_pack_conv_multi(chore_handle* h, int dtype, PackJobs& j, hipStream_t s) {
    unsigned blocks = 0;
    for (int k = 0; k < j.n; ++k) {
        j.job[k].nvec = packed_conv_bytes(dtype, j.job[k].taps, j.job[k].Cin, j.job[k].Cout) / 16 / ((dtype == CHORE_F16X3 || dtype == CHORE_F16) ? 2 : 1);
        j.job[k].blocks = (unsigned)((j.job[k].nvec + 255) / 256);
        blocks += j.job[k].blocks;
    }
    blocks += (unsigned)((j.zero_vecs + 255) / 256);
    if (j.amax_x) blocks += AMAX_CELLS;
    if (!blocks) return CHORE_OK;
    if (dtype == CHORE_F16X3 || dtype == CHORE_F16) hipLaunchKernelGGL(pack_conv_multi_kernel<x3_t>, dim3(blocks), dim3(256), 0, s, j);
    else if (dtype == CHORE_F32) hipLaunchKernelGGL(pack_conv_multi_kernel<float>, dim3(blocks), dim3(256), 0, s, j);
    else hipLaunchKernelGGL(pack_conv_multi_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, j);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int launch_pack_conv(chore_handle* h, int dtype, int taps, int Cin, int Cout, const float* w, void* dst,
                     hipStream_t s, int transposed) {
    const bool two_planes = dtype == CHORE_F16X3 || dtype == CHORE_F16;      // the fp16 mode reads the fp16 x 3 weight format
    const size_t nvec = packed_conv_bytes(dtype, taps, Cin, Cout) / 16 / (two_planes ? 2 : 1);   // per plane
    const unsigned blocks = (unsigned)((nvec + 255) / 256);
    if (two_planes)
        hipLaunchKernelGGL(pack_conv_kernel<x3_t>, dim3(blocks), dim3(256), 0, s, taps, Cin, Cout, w, (u32x4*)dst, nvec,
                           transposed);
    else if (dtype == CHORE_F32)
        hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(blocks), dim3(256), 0, s, taps, Cin, Cout, w, (u32x4*)dst, nvec,
                           transposed);
    else
        hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, taps, Cin, Cout, w, (u32x4*)dst,
                           nvec, transposed);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

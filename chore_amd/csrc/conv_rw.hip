// conv_rw.hip -- 1x1 convolution (fp16 x 3, fp16 and bf16 operands) with REGISTER-RESIDENT WEIGHTS (the 1x1 layers of the stack tail,
// model/HGFilters.py:128-142,167-183: conv_last / l / the merged bl + al.l, and ConvBlock's downsample, model/net_util.py:364-371).
// Same arithmetic, same ConvArgs contract and the same packed weights as conv_pc_kernel<T, 1, ...>.
//
// Why (round 5, profiles/r05_conv_rw.txt).  A 1x1 layer has K = Cin <= 256: on conv_pc_kernel a 256-pixel tile is ONE pass
// load -> split -> 16 k-steps of MFMAs -> store, every workgroup of the launch in the same phase at the same time, so HBM idles
// while the matrix pipe works and the other way round: 256 -> 256 at 128^2 took 61 us for 134 MB of compulsory traffic (22 us
// at the 6 TB/s the part reaches) and 17 us of MFMAs -- the phases add up.  The whole weight matrix of such a layer is small:
// hi + lo fp16 planes of 32 output channels x 256 input channels are 32 KB = 128 registers of a wave.  So here
//   * a workgroup is 8 waves, wave w keeps the fragments of ITS 32 output channels (all K) in registers for the whole launch;
//   * the workgroup is persistent over a contiguous run of 32-pixel blocks of one image (1x1: the map is a list of pixels);
//   * per block: every thread loads its 8-channel units of the NEXT block (global -> registers), the block's fp16 hi / lo
//     planes are read from LDS as MFMA A fragments by all eight waves, the accumulators leave straight from registers
//     (a lane holds one channel of 16 pixels: a store instruction writes two full 128-byte lines), GroupNorm statistics of
//     the outputs are kept per lane and reduced once at the end;
//   * one s_barrier per block; the next block's loads are issued before the block's MFMAs and staged (GroupNorm + ReLU +
//     split) after them.
//   Input read once, output written once, loads of block i + 1 in flight under block i's MFMAs.
// What bounds it (measured, profiles/r05_conv_rw.txt): instruction issue -- the vector-ALU work of the split and the epilogue does not
// overlap the MFMAs of the SIMD's other wave; a block takes MFMAs + staging + stores.
#include "conv_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

using namespace conv_detail;

// kernel experiments (scripts/build_variant.sh <name> conv_rw.hip -D...): CHORE_RW_STAMPS=1 -- wall-clock stamps of every phase of
// the workgroup in the middle of the grid (wave 0 and wave 4), printed by the launcher for the first launches of each tiling
#ifndef CHORE_RW_STAMPS
#define CHORE_RW_STAMPS 0
#endif
#if CHORE_RW_STAMPS
__device__ unsigned long long g_rw_stamps[2][64];
#endif

namespace {

__device__ __forceinline__ void wg_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier; vector-memory loads and stores stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// KC input channels, WN = Cout / 32 channel blocks; the 8 waves are WN channel blocks x WM pixel blocks of 32
template <int KC, int WN, int APL = 2> struct RGeo {   // APL: fp16 planes of a staged pixel (fp16 x 3: hi + lo; fp16 / bf16 tensors: one)
    static constexpr int WM = 8 / WN, MPX = 32 * WM;               // pixels per block
    static constexpr int NKG = KC / 16;                            // MFMA k-groups
    static constexpr int PLANE = KC * 2;                           // bytes of one fp16 plane of a pixel
    static constexpr int ROWB = APL * PLANE + 16;                  // LDS row of a pixel: hi plane [, lo plane], pad (odd number of 16-byte slots)
    static constexpr int ABUF = MPX * ROWB;
    static constexpr int UPP = KC / 8;                             // 8-channel units per pixel
    static constexpr int NV = MPX * UPP / 512;                     // units per thread and block
    static constexpr size_t smem_bytes() { return (size_t)2 * ABUF + (size_t)KC * 8 + (size_t)8 * 4 * 32 * 4; }
    static_assert(WN * WM == 8 && NV >= 1 && NV * 512 == MPX * UPP, "conv_rw geometry");
};

// T = x3_t: fp32 tensors, activations split into fp16 hi + lo while they are staged, three MFMAs per product;
// T = h16_t ("fp16 fields"): fp16 tensors, one activation plane, two MFMAs per product (a * w_lo, a * w_hi);
// T = bf16_t: bf16 tensors, one activation and one weight plane, one v_mfma_f32_32x32x16_bf16 per product
template <typename T, int KC, int WN, bool RES, bool SC>
__global__ __launch_bounds__(512) void conv_rw_kernel(ConvArgs a, int bpw, int wg_per_img) {
    constexpr bool X3 = IS_X3<T>, BF = std::is_same<T, bf16_t>::value, H16 = IS_H16<T>;
    static_assert(X3 || BF || H16, "conv_rw_kernel: fp16 x 3, fp16 or bf16 operands");
    static_assert(!SC || X3, "the operand scale belongs to the fp16 x 3 data gradients");
    if constexpr (!BF) f16_saturate_mode();
    using ST = typename std::conditional<X3, float, unsigned short>::type;     // element type in memory
    constexpr int LVI = X3 ? 2 : 1;                     // 16-byte loads per 8 channels
    constexpr int APL = X3 ? 2 : 1, WPL = BF ? 1 : 2;   // activation / weight planes
    using G = RGeo<KC, WN, APL>;
    constexpr int WM = G::WM, MPX = G::MPX, NKG = G::NKG, PLANE = G::PLANE, ROWB = G::ROWB, ABUF = G::ABUF, UPP = G::UPP, NV = G::NV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* abuf = smem;                                       // [2][MPX][ROWB]
    float* ss_lds = (float*)(smem + 2 * ABUF);               // [KC][2] GroupNorm scale, shift
    float* red = ss_lds + 2 * KC;                            // [8 waves][4 kinds][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % WN, wm = wid / WN;
#if CHORE_RW_STAMPS
    auto stamp = [&](int k) { if (blockIdx.x == gridDim.x / 2 && (tid == 0 || tid == 256) && k < 64) g_rw_stamps[tid ? 1 : 0][k] = wall_clock64(); };
#else
    auto stamp = [&](int) {};
#endif
    stamp(0);
    const int b = blockIdx.x / wg_per_img, wgi = blockIdx.x % wg_per_img;
    const int HW = a.H * a.W;
    const int nblk = (HW + MPX - 1) / MPX;
    const int b0 = wgi * bpw;
    const int n_it = (b0 + bpw <= nblk ? bpw : nblk - b0);   // >= 1 by construction of the grid
    const bool use_gn = a.in_st != nullptr;
    // SC: the input is a gradient whose range comes in ConvArgs::in_amax (the data-gradient convolutions of fp16 x 3 training):
    // operand times a power of two before the split, accumulators times its inverse (enc_common.h, x3_in_scale)
    float in_mul = 1.f, in_inv = 1.f;
    if constexpr (SC) x3_in_scale(a.in_amax, in_mul, in_inv);
    const ST* in_b = (const ST*)a.in.p + (size_t)b * HW * a.in.cs + a.in.co;

    // ---- staging: unit u = tid + 512 j of a block = (pixel u / UPP, 8 channels (u % UPP) * 8) ----
    auto issue_loads = [&](u32x4 (&r)[NV][LVI], int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int u = tid + 512 * j, p = u / UPP, g = u % UPP;
            const int pix = blk * MPX + p;
            const u32x4* q = (const u32x4*)(in_b + (size_t)(pix < HW ? pix : 0) * a.in.cs + g * 8);
#pragma unroll
            for (int k = 0; k < LVI; ++k) r[j][k] = q[k];
        }
    };
    // (512 is a multiple of UPP: every unit of a thread covers the SAME 8 channels, their GroupNorm affine lives in registers)
    static_assert(512 % UPP == 0, "a thread's units share their channels");
    float sc[8], sh[8];
    auto stage = [&](const u32x4 (&r)[NV][LVI], int blk, char* dstb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int u = tid + 512 * j, p = u / UPP, g = u % UPP;
            const int pix = blk * MPX + p;
            u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
            char* d = dstb + p * ROWB + g * 16;
            if constexpr (X3) {
                if (pix < HW) {
                    if constexpr (SC) xform_x3(r[j][0], r[j][LVI - 1], sc, sh, use_gn, hi, lo, in_mul);
                    else xform_x3(r[j][0], r[j][LVI - 1], sc, sh, use_gn, hi, lo);
                }
                *(u32x4*)d = hi;
                *(u32x4*)(d + PLANE) = lo;
            } else {
                if (pix < HW) {
                    if (use_gn) {   // relu(x * scale + shift) in fp32, rounded to the 16-bit type once (as conv_pc_kernel's staging)
                        if constexpr (BF) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float x0 = fmaf(__uint_as_float(r[j][0][k] << 16), sc[2 * k], sh[2 * k]);
                                float x1 = fmaf(__uint_as_float(r[j][0][k] & 0xffff0000u), sc[2 * k + 1], sh[2 * k + 1]);
                                x0 = x0 > 0.f ? x0 : 0.f;
                                x1 = x1 > 0.f ? x1 : 0.f;
                                hi[k] = pack2bf(x0, x1);
                            }
                        } else {
                            const f16x8_t x = __builtin_bit_cast(f16x8_t, r[j][0]);
                            f16x8_t y;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float t = fmaf((float)x[k], sc[k], sh[k]);
                                y[k] = (_Float16)(t > 0.f ? t : 0.f);
                            }
                            hi = __builtin_bit_cast(u32x4, y);
                        }
                    } else hi = r[j][0];
                }
                *(u32x4*)d = hi;
            }
        }
    };

    // ---- prologue: block 0 on its way, this wave's weight fragments into registers, the GroupNorm table ----
    u32x4 setA[NV][LVI];
    issue_loads(setA, b0);
    u32x4 bh[NKG], bl[WPL == 2 ? NKG : 1];
    {
        const u32x4* wv = (const u32x4*)a.wpk + (size_t)wn * 64 + lane;
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) {
            bh[kg] = wv[(size_t)kg * WN * 64];
            if constexpr (WPL == 2) bl[kg] = wv[(size_t)(NKG + kg) * WN * 64];       // the lo plane follows the complete hi plane
        }
    }
    for (int ci = tid; ci < KC; ci += 512) {
        float sc = 1.f, sh = 0.f;
        if (use_gn) gn_scale_shift(a.in_st, a.B, b, KC, ci, HW, a.gamma, a.beta, sc, sh);
        ss_lds[2 * ci] = sc;
        ss_lds[2 * ci + 1] = sh;
    }
    stamp(1);
    wg_barrier();
    stamp(2);
    {
        const f32x4* q = (const f32x4*)(ss_lds + (tid % UPP) * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 t = q[k];
            sc[2 * k] = t[0]; sh[2 * k] = t[1]; sc[2 * k + 1] = t[2]; sh[2 * k + 1] = t[3];
        }
    }
    stage(setA, b0, abuf);
    wg_barrier();
    stamp(3);

    // ---- per-wave output coordinates: lane = channel (lane & 31) of 16 pixels ----
    const int half = lane >> 5, ch = wn * 32 + (lane & 31);
    const float ASCALE = BF ? 1.0f : (SC ? in_inv : 1.0f) / (float)(1 << X3_WSHIFT);   // undoes the weight scaling of the fp16 packing
    const float bias = a.bias ? a.bias[ch] : 0.f;
    const size_t img = (size_t)b * HW;
    ST* out_p = (ST*)a.out.p + img * a.out.cs + a.out.co + ch;
    ST* raw_p = a.raw.p ? (ST*)a.raw.p + img * a.raw.cs + a.raw.co + ch : nullptr;
    const ST* res_p = RES ? (const ST*)a.res.p + img * a.res.cs + a.res.co + ch : nullptr;
    // one element <-> float; put() stores the value rounded to the tensor's type and returns it as stored (what the statistics see)
    auto get = [&](const ST* q) __attribute__((always_inline)) -> float {
        if constexpr (X3) return *q;
        else if constexpr (BF) return __uint_as_float((unsigned)*q << 16);
        else return (float)__builtin_bit_cast(_Float16, *q);
    };
    auto put = [&](ST* q, float v) __attribute__((always_inline)) -> float {
        if constexpr (X3) { *q = v; return v; }
        else if constexpr (BF) { const unsigned short u = (unsigned short)(pack2bf(v, v) & 0xffffu); *q = u; return __uint_as_float((unsigned)u << 16); }
        else { const _Float16 hh = (_Float16)v; *q = __builtin_bit_cast(unsigned short, hh); return (float)hh; }
    };
    const bool want_stats = a.st_raw || a.st_out;
    float sr = 0.f, qr = 0.f, so = 0.f, qo = 0.f;
    const char* a_rd = abuf + ((wm * 32) + (lane & 31)) * ROWB + half * 16;

    // one block: MFMAs on LDS buffer (i & 1), then the accumulators leave
    constexpr int NACC = 2;                                  // accumulators that take turns over the k-groups
    f32x16 acc[NACC];
    float rq[16];
    auto matmul = [&](int i) __attribute__((always_inline)) {
        const int blk = b0 + i;
        const char* ap = a_rd + (i & 1) * ABUF;
        const int pix0 = blk * MPX + wm * 32 + 4 * half;     // this lane's pixel of accumulator register r: pix0 + (r & 3) + 8 (r >> 2)
        const bool full = (blk + 1) * MPX <= HW;             // wave-uniform: no pixel of the block is past the image
        if constexpr (RES) {
            const ST* rb = res_p + (size_t)pix0 * a.res.cs;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                rq[r] = (full || pix0 + rr < HW) ? get(rb + rr * a.res.cs) : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) {
            const u32x4 av = *(const u32x4*)(ap + kg * 32);
            f32x16& c = acc[kg % NACC];
            if constexpr (BF) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bh[kg]), c, 0, 0, 0);
            } else {
                const f16x8_t ah = __builtin_bit_cast(f16x8_t, av);
                const f16x8_t wh = __builtin_bit_cast(f16x8_t, bh[kg]), wl = __builtin_bit_cast(f16x8_t, bl[kg]);
                if constexpr (X3) {                          // small terms first
                    const f16x8_t al = __builtin_bit_cast(f16x8_t, *(const u32x4*)(ap + PLANE + kg * 32));
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, c, 0, 0, 0);
            }
        }
    };
    auto leave = [&](int i) __attribute__((always_inline)) {
        const int blk = b0 + i;
        const int pix0 = blk * MPX + wm * 32 + 4 * half;
        const bool full = (blk + 1) * MPX <= HW;
        ST* ob = out_p + (size_t)pix0 * a.out.cs;
        ST* rwb = raw_p ? raw_p + (size_t)pix0 * a.raw.cs : nullptr;
        auto value = [&](int r) __attribute__((always_inline)) -> float {
            float v = acc[0][r];
            if constexpr (NACC == 2) v += acc[1][r];
            return fmaf(v, ASCALE, bias);                    // (ASCALE is a power of two: the same value as v * ASCALE + bias)
        };
        if (full && !rwb) {          // the common case: no masks, no second tensor
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                float v = value(r);
                if constexpr (RES) v += rq[r];
                v = put(ob + rr * a.out.cs, v);
                so += v; qo = fmaf(v, v, qo);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                float v = value(r);
                if (pix0 + rr < HW) {
                    if (rwb) {
                        const float g = put(rwb + rr * a.raw.cs, v);
                        sr += g; qr += g * g;
                    }
                    if constexpr (RES) v += rq[r];
                    v = put(ob + rr * a.out.cs, v);
                    so += v; qo += v * v;
                }
            }
        }
    };

    // iteration i: block i + 1's loads leave first and fly under block i's MFMAs
    // (tried: the two waves of a SIMD in opposite orders -- stage, MFMAs, leave against MFMAs, stage, leave, with a second register
    // set for the early wave's loads: no faster, 40.6 against 42.3 us; LDS progress counts and three buffers instead of the
    // barrier, so that the waves drift apart: no faster either, profiles/r05_conv_rw.txt)
#pragma unroll 1
    for (int i = 0; i < n_it; ++i) {
        const bool more = i + 1 < n_it;
        issue_loads(setA, b0 + i + 1);                       // (past the end: pixel 0 again, never used)
        __builtin_amdgcn_sched_barrier(0);
        stamp(4 + 5 * i);
        matmul(i);
        stamp(5 + 5 * i);
        if (more) stage(setA, b0 + i + 1, abuf + ((i + 1) & 1) * ABUF);   // before the stores: it waits for loads only
        stamp(6 + 5 * i);
        leave(i);
        stamp(7 + 5 * i);
        wg_barrier();
        stamp(8 + 5 * i);
    }

    if (want_stats) {   // uniform over the grid
        // lane = channel: the two halves of a wave hold the same 32 channels (different pixels), the WM waves of a channel block too
        sr += __shfl_xor(sr, 32, 64); qr += __shfl_xor(qr, 32, 64); so += __shfl_xor(so, 32, 64); qo += __shfl_xor(qo, 32, 64);
        if (lane < 32) {
            float* p = red + wid * 128 + lane;
            p[0] = sr; p[32] = qr; p[64] = so; p[96] = qo;
        }
        wg_barrier();
        if (tid < 32 * WN) {
            // thread = channel; t[kind]: kinds 0 / 1 = sum / sum of squares of `raw`, 2 / 3 of `out`; fixed order over the pixel blocks
            const int nb = tid >> 5, c = tid & 31;
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = 0.f;
#pragma unroll
                for (int m = 0; m < WM; ++m) t[k] += red[(m * WN + nb) * 128 + k * 32 + c];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                GroupStat* st = k ? a.st_out : a.st_raw;
                if (!st) continue;
                const int gs = (k ? a.st_out_C : a.st_raw_C) / GN_GROUPS, co = k ? a.st_out_co : a.st_raw_co;
                const float s1 = group_lane_sum(t[2 * k], gs), s2 = group_lane_sum(t[2 * k + 1], gs);
                GroupStat* o = st + (size_t)b * GN_GROUPS + (co + tid) / gs;
                if (tid % gs == 0) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
        }
    }
}

template <typename T, int KC, int WN, bool RES, bool SC = false>
int launch_rw_t(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    using G = RGeo<KC, WN, IS_X3<T> ? 2 : 1>;
    const size_t smem = G::smem_bytes();
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)conv_rw_kernel<T, KC, WN, RES, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = true;
    }
    if (h->cu_count <= 0) {
        CHORE_HIP_CHECK(h, hipDeviceGetAttribute(&h->cu_count, hipDeviceAttributeMultiprocessorCount, h->device));
        if (h->cu_count <= 0) h->cu_count = 256;
    }
    // one persistent workgroup per CU: every image's blocks are dealt to cu / B workgroups in contiguous runs
    const int nblk = (a.H * a.W + G::MPX - 1) / G::MPX;
    // ... on THREE QUARTERS of the CUs (CHORE_CONV_RW_CUS=n: n): these layers are bound by HBM, 192 workgroups move the bytes as fast
    // as 256 (one step at a time 4.99 -> 4.86 ms) and the other step in flight gets a quarter of the chip for its MFMA-bound
    // convolutions meanwhile (two in flight 4.01 -> 3.87 ms; 224 / 160 / 128: 3.91 / 3.86 / 3.90; profiles/r06_conv_fill.txt)
    static const int cus_env = getenv("CHORE_CONV_RW_CUS") ? atoi(getenv("CHORE_CONV_RW_CUS")) : 0;
    int wpi = (cus_env > 0 ? cus_env : (h->cu_count * 3) / 4) / a.B;
    if (wpi < 1) wpi = 1;
    if (wpi > nblk) wpi = nblk;
    const int bpw = (nblk + wpi - 1) / wpi;
    wpi = (nblk + bpw - 1) / bpw;
    hipLaunchKernelGGL((conv_rw_kernel<T, KC, WN, RES, SC>), dim3((unsigned)(wpi * a.B)), dim3(512), smem, s, a, bpw, wpi);
    CHORE_LAUNCH_CHECK(h, s);
#if CHORE_RW_STAMPS
    {
        static int shown = 0;
        if (shown < 40 && (++shown % 10 == 0)) {
            unsigned long long t[2][64];
            CHORE_HIP_CHECK(h, hipStreamSynchronize(s));
            CHORE_HIP_CHECK(h, hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rw_stamps), sizeof(t)));
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "[rw stamps KC=%d WN=%d bpw=%d] wave %d (%s), us since entry: setup %.2f bar %.2f block0 %.2f |", KC, WN, bpw, 4 * w,
                        "loads matmul stage leave barrier",
                        (t[w][1] - t[w][0]) * 0.01, (t[w][2] - t[w][0]) * 0.01, (t[w][3] - t[w][0]) * 0.01);
                for (int i = 0; i < bpw && 8 + 5 * i < 64; ++i)
                    fprintf(stderr, " [%d] %.2f %.2f %.2f %.2f %.2f |", i, (t[w][4 + 5 * i] - t[w][0]) * 0.01, (t[w][5 + 5 * i] - t[w][0]) * 0.01,
                            (t[w][6 + 5 * i] - t[w][0]) * 0.01, (t[w][7 + 5 * i] - t[w][0]) * 0.01, (t[w][8 + 5 * i] - t[w][0]) * 0.01);
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return CHORE_OK;
}

}  // namespace

// CHORE_CONV_RW=0: the 1x1 layers stay on conv_pc_kernel (A/B runs)
bool conv_rw_covers(int dtype, int taps, int Cin, int Cout, bool scaled_input) {
    static const bool off = getenv("CHORE_CONV_RW") && atoi(getenv("CHORE_CONV_RW")) == 0;
    // CHORE_CONV_RW_X3ONLY=1: the fp16 instantiation off (A/B against conv_pc_kernel in that mode).  bf16 is OPT-IN
    // (CHORE_CONV_RW_BF16=1): alone the layers are faster (256 -> 256 at 128^2 24.8 against 36.2 us on conv_lds_kernel), inside the
    // encoder they are not (12 launches: 0.399 against 0.376 ms per step; whole bf16 step 3.28 against 3.25 ms, profiles/r05_conv_rw.txt):
    // conv_lds_kernel's many small workgroups hide a cold start (weights, statistics, instructions) that one persistent workgroup
    // per CU pays in full on a 20 us kernel.
    static const bool x3only = getenv("CHORE_CONV_RW_X3ONLY") != nullptr, bf16_on = getenv("CHORE_CONV_RW_BF16") != nullptr;
    if (off || taps != 1 || (dtype != CHORE_F16X3 && dtype != CHORE_F16 && dtype != CHORE_BF16)) return false;
    if (dtype == CHORE_F16 && x3only) return false;
    if (dtype == CHORE_BF16 && !bf16_on) return false;
    if (scaled_input) return dtype == CHORE_F16X3 && Cin == 256 && Cout == 256;     // data gradients of fp16 x 3 training: the stack tail's layers
    return (Cin == 256 && Cout == 256) || (Cin == 128 && Cout == 256) || (Cin == 64 && Cout == 128);
}
bool conv_rw_eligible(int dtype, int taps, const ConvArgs& a) {
    if (a.res2.p) return false;
    return conv_rw_covers(dtype, taps, a.in.C, a.Cout, a.in_amax != nullptr);
}

template <typename T>
static int launch_conv_rw_t(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    const int k = a.in.C, n = a.Cout;
    const bool res = a.res.p != nullptr;
    if constexpr (IS_X3<T>) {
        if (a.in_amax) {
            if (k == 256 && n == 256) return res ? launch_rw_t<T, 256, 8, true, true>(h, a, s) : launch_rw_t<T, 256, 8, false, true>(h, a, s);
            CHORE_FAIL(h, CHORE_EINVAL, "conv_rw: no data-gradient kernel for Cin=%d Cout=%d", k, n);
        }
    }
#define RW_CASE(KC, WN) \
    if (k == KC && n == 32 * WN) return res ? launch_rw_t<T, KC, WN, true>(h, a, s) : launch_rw_t<T, KC, WN, false>(h, a, s)
    RW_CASE(256, 8);
    RW_CASE(128, 8);
    RW_CASE(64, 4);
#undef RW_CASE
    CHORE_FAIL(h, CHORE_EINVAL, "conv_rw: no kernel for Cin=%d Cout=%d", k, n);
}

int launch_conv_rw(chore_handle* h, int dtype, const ConvArgs& a, hipStream_t s) {
    if (dtype == CHORE_F16) return launch_conv_rw_t<h16_t>(h, a, s);
    if (dtype == CHORE_BF16) return launch_conv_rw_t<bf16_t>(h, a, s);
    if (dtype == CHORE_F16X3) return launch_conv_rw_t<x3_t>(h, a, s);
    CHORE_FAIL(h, CHORE_EINVAL, "conv_rw: dtype %d not covered", dtype);
}

// conv_small.hip -- 3x3 convolution (+ fused GroupNorm/ReLU prologue, bias / residual / concat-slice / statistics epilogue)
// for the SMALL maps of the hourglass: 64x64 and 32x32 (model/net_util.py:374-396 ConvBlock at the two lower levels of
// model/HGFilters.py:26-50).  Same arithmetic and the same ConvArgs contract as conv_lds_kernel.
//
// Why a second kernel.  conv_lds_kernel gives a workgroup an 8x32-pixel tile and walks the K dimension (9 taps x Cin)
// chunk by chunk behind barriers: on a 32x32 map with batch 4 that is 16-64 workgroups on a 256-CU part, each a serial
// chain of ~25 us (measured: 21.8 us per launch, 2 us of it MFMA).  These launches are 15 of the ~25 dependent steps of
// every hourglass stack, and each step ends in a grid-wide GroupNorm reduction, so they cannot be merged: a persistent
// kernel with grid barriers costs MORE than a kernel boundary on this part (scripts/probes/barrier_probe.hip: 14 us per
// 256-workgroup barrier against 2.5 us per dependent launch).  What can change is the shape of the work inside a launch:
//   * a workgroup owns ONE MFMA output tile -- 32 pixels of an image row x 32 output channels -- so a 32x32 map yields
//     256-512 workgroups and a 64x64 map 1 024-2 048;
//   * its halo patch (3 rows x 34 pixels x ALL input channels) is staged once, GroupNorm + ReLU applied on the way,
//     one barrier;
//   * the K dimension is split over the four waves -- unit = (32-channel chunk, kernel row), dealt round-robin -- each
//     wave reads its A fragments from the patch and its B fragments straight from the fragment-ordered weight arena
//     (L2-resident, one coalesced 1 KB load per fragment, no LDS ring: no other wave needs the same fragment);
//   * the four partial tiles are added in wave order through LDS (deterministic), then bias, residuals, stores and the
//     exact GroupNorm statistics of what was stored, like the big kernel.
#include "enc_common.h"
#include <type_traits>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

namespace {

constexpr int PW = 34;                    // patch row: 32 pixels + halo
constexpr int RED_LD = 33;                // pitch of a partial tile row in floats
constexpr int RED_BYTES = 4 * 32 * RED_LD * 4;

// ROWS = image rows per workgroup (1, 2 or 4); the four waves are ROWS rows x KS = 4 / ROWS parts of the K dimension.
// More rows: the weight fragments a workgroup pulls from L2 serve ROWS tiles (the waves of different rows read the same
// fragment within a few cycles: one L2 fetch, L1 hits for the others) and the halo shrinks from 3-for-1 to
// (ROWS + 2)-for-ROWS; fewer rows: more workgroups.
template <typename T, int CIN, int ROWS> struct SGeo {
    static constexpr bool X3 = IS_X3<T>;
    static constexpr bool H16 = IS_H16<T>;                  // fp16 tensors: one activation plane, weights hi + lo (two MFMAs)
    static constexpr int KS = 4 / ROWS;
    static constexpr int NPX = (ROWS + 2) * PW;
    static constexpr int PB = CIN * 2 + 16;                 // bytes per pixel per operand plane (16-byte slots: odd count)
    static constexpr int PLANE = NPX * PB;
    static constexpr int NPL = X3 ? 2 : 1;
    static constexpr int VPP = CIN / 8;                     // 8-channel slots per pixel
    static constexpr int NSLOT = (NPX * VPP + 255) / 256;   // slots per thread
    static constexpr int U = (CIN / 32) * 3;                // K units: (32-channel chunk, kernel row)
    // the partial tiles of the reduction reuse the patch (dead after the K loop)
    static constexpr size_t MAIN = (size_t)NPL * PLANE > RED_BYTES ? (size_t)NPL * PLANE : RED_BYTES;
    static constexpr size_t smem = MAIN + (size_t)CIN * 8 + (size_t)4 * 4 * 32 * 4;
};

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

template <typename T, int CIN, int ROWS>
__global__ __launch_bounds__(256) void conv_small_kernel(ConvArgs a) {
    if constexpr (IS_X3<T> || IS_H16<T>) f16_saturate_mode();     // the fp16 x 3 / fp16 operand split never produces inf (common.h)
    using G = SGeo<T, CIN, ROWS>;
    constexpr bool X3 = G::X3, H16 = G::H16, WLO = X3 || H16;      // WLO: the weights come as hi + lo planes
    using ST = typename std::conditional<H16, unsigned short, typename Store<T>::type>::type;
    constexpr int PB = G::PB, PLANE = G::PLANE, VPP = G::VPP, NSLOT = G::NSLOT, U = G::U, KS = G::KS, NPX = G::NPX;
    constexpr int LV = X3 ? 2 : 1;                          // 16-byte loads per 8-channel slot
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                                     // [NPL][NPX][PB]
    float* red = (float*)smem;                              // [4 waves][32 pixels][RED_LD], over the patch once it is dead
    float* ss = (float*)(smem + G::MAIN);                   // [CIN][2] GroupNorm affine
    float* sred = ss + 2 * CIN;                             // [4 kinds][4 waves][32 channels]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (image, row, 32-pixel segment, channel tile); the channel tiles of one segment read the same patch, so
    // they are neighbours in the logical order and every XCD (own L2) takes a contiguous range of it
    const int ntn = a.Cout / 32, segs = a.W / 32;
    int lid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = lid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
    }
    const int n_tile = lid % ntn;
    int t = lid / ntn;
    const int seg = t % segs; t /= segs;
    const int rgs = a.H / ROWS;
    const int y = (t % rgs) * ROWS, b = t / rgs;            // first row of this workgroup
    const int x0 = seg * 32;
    const bool use_gn = a.in_st != nullptr;
    float in_mul = 1.f, in_inv = 1.f;                       // operand scale of a gradient input (ConvArgs::in_amax)
    if constexpr (IS_X3S<T>) x3_in_scale(a.in_amax, in_mul, in_inv);

    // ---- stage the patch: all loads first, the affine while they fly ----
    const ST* in_b = (const ST*)a.in.p + (size_t)b * a.H * a.W * a.in.cs + a.in.co;
    u32x4 pre[NSLOT][LV];
    int soff[NSLOT];                                        // LDS byte offset of the slot, -1: past the patch
    bool sval[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int i = tid + j * 256;
        const bool in_patch = i < NPX * VPP;
        const int px = in_patch ? i / VPP : 0, v = in_patch ? i % VPP : 0;
        const int gy = y + px / PW - 1, gx = x0 + px % PW - 1;
        const bool ok = in_patch && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const size_t off = ok ? ((size_t)gy * a.W + gx) * a.in.cs + v * 8 : 0;      // unconditional loads (see conv_lds)
#pragma unroll
        for (int k = 0; k < LV; ++k) pre[j][k] = *((const u32x4*)(in_b + off) + k);
        soff[j] = in_patch ? px * PB + v * 16 : -1;
        sval[j] = ok;
    }
    if (use_gn)
        for (int ci = tid; ci < CIN; ci += 256)
            gn_scale_shift(a.in_st, a.B, b, CIN, ci, a.H * a.W, a.gamma, a.beta, ss[2 * ci], ss[2 * ci + 1]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        if (soff[j] < 0) continue;
        const int c0 = ((tid + j * 256) % VPP) * 8;
        u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
        if (sval[j]) {
            float f[8];
            if constexpr (X3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { f[k] = __uint_as_float(pre[j][0][k]); f[4 + k] = __uint_as_float(pre[j][1][k]); }
            } else if constexpr (H16) {
                const f16x8_t x = __builtin_bit_cast(f16x8_t, pre[j][0]);
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = (float)x[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f[2 * k] = __uint_as_float(pre[j][0][k] << 16);
                    f[2 * k + 1] = __uint_as_float(pre[j][0][k] & 0xffff0000u);
                }
            }
            if (use_gn) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float g = fmaf(f[k], ss[2 * (c0 + k)], ss[2 * (c0 + k) + 1]);
                    f[k] = g > 0.f ? g : 0.f;
                }
            }
            if constexpr (X3) {
                if constexpr (IS_X3S<T>) {
                    if (!use_gn) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] *= in_mul;
                    }
                }
                f16x8_t h, l;
#pragma unroll
                for (int k = 0; k < 8; ++k) { h[k] = (_Float16)f[k]; l[k] = (_Float16)(f[k] - (float)h[k]); }
                hi = __builtin_bit_cast(u32x4, h);
                lo = __builtin_bit_cast(u32x4, l);
            } else if constexpr (H16) {
                if (use_gn) {       // relu(x * scale + shift) in fp32, rounded to fp16 once (as conv_pc_kernel<h16_t>)
                    f16x8_t h;
#pragma unroll
                    for (int k = 0; k < 8; ++k) h[k] = (_Float16)f[k];
                    hi = __builtin_bit_cast(u32x4, h);
                } else hi = pre[j][0];
            } else if (use_gn) {
#pragma unroll
                for (int k = 0; k < 4; ++k) hi[k] = pack2bf(f[2 * k], f[2 * k + 1]);
            } else {
                hi = pre[j][0];
            }
        }
        *(u32x4*)(patch + soff[j]) = hi;
        if constexpr (X3) *(u32x4*)(patch + PLANE + soff[j]) = lo;
    }
    __syncthreads();

    // ---- K loop of this wave: units wid, wid + 4, ... ; fragments of the next unit are fetched during the MFMAs ----
    const int half = lane >> 5, px = lane & 31;
    const int NKG = CIN / 16, NB = a.Cout / 32;
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)n_tile * 64 + lane;
    const size_t lo_plane = (size_t)9 * NKG * NB * 64;      // fp16 x 3: the lo plane follows the complete hi plane
    auto wfrag = [&](int tap, int kgg) -> const u32x4* { return wbase + ((size_t)tap * NKG + kgg) * NB * 64; };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wrow_ = wid / G::KS;
    u32x4 b0[6], b1[6], l0[WLO ? 6 : 1], l1[WLO ? 6 : 1];     // two fragment sets, named statically (no indexed registers)
    auto load_b = [&](u32x4 (&bq)[6], u32x4 (&bl)[WLO ? 6 : 1], int u) {
        const int c = u / 3, ky = u % 3;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const u32x4* p = wfrag(ky * 3 + k / 2, c * 2 + (k & 1));
            bq[k] = *p;
            if constexpr (WLO) bl[k] = *(p + lo_plane);
        }
    };
    auto unit = [&](const u32x4 (&bq)[6], const u32x4 (&bl)[WLO ? 6 : 1], int u) {
        const int c = u / 3, ky = u % 3;
        const char* ap = patch + (((wrow_ + ky) * PW + px) * PB) + (c * 32 + 8 * half) * 2;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int kx = k / 2, kg = k & 1;
            const u32x4 av = *(const u32x4*)(ap + kx * PB + kg * 32);
            if constexpr (X3) {
                const u32x4 al = *(const u32x4*)(ap + PLANE + kx * PB + kg * 32);
                acc = mfma_f16(al, bq[k], acc);             // small terms first
                acc = mfma_f16(av, bl[k], acc);
                acc = mfma_f16(av, bq[k], acc);
            } else if constexpr (H16) {
                acc = mfma_f16(av, bl[k], acc);             // a * w_lo, then a * w_hi
                acc = mfma_f16(av, bq[k], acc);
            } else {
                acc = mfma_bf16(av, bq[k], acc);
            }
        }
    };
    const int kp = wid % KS;                                // this wave: image row y + wrow_, K part kp
    if (kp < U) load_b(b0, l0, kp);
#pragma unroll 1
    for (int u = kp; u < U; u += 2 * KS) {
        if (u + KS < U) load_b(b1, l1, u + KS);
        unit(b0, l0, u);
        if (u + KS < U) {
            if (u + 2 * KS < U) load_b(b0, l0, u + 2 * KS);
            unit(b1, l1, u + KS);
        }
    }

    // ---- add the KS partial tiles of every row in wave order, then the epilogue row by row ----
    __syncthreads();                                        // every wave is done with the patch: reuse it
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wid * 32 + mfma32_row(r, half)) * RED_LD + px] = acc[r];
    __syncthreads();
    const float ASCALE = IS_X3S<T> ? in_inv / (float)(1 << X3_WSHIFT) : (WLO ? 1.0f / (float)(1 << X3_WSHIFT) : 1.0f);
    const int p = tid >> 3, g4 = (tid & 7) * 4;              // this thread: pixel p, channels g4 .. g4+3 of the tile
    const int cg = n_tile * 32 + g4;
    const bool want_stats = a.st_raw || a.st_out;
    float sr[4] = {0.f, 0.f, 0.f, 0.f}, qr[4] = {0.f, 0.f, 0.f, 0.f}, so[4] = {0.f, 0.f, 0.f, 0.f}, qo[4] = {0.f, 0.f, 0.f, 0.f};
    float bias4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bias4[k] = a.bias ? a.bias[cg + k] : 0.f;
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sacc = red[((row * KS + 0) * 32 + p) * RED_LD + g4 + k];
#pragma unroll
            for (int w = 1; w < KS; ++w) sacc += red[((row * KS + w) * 32 + p) * RED_LD + g4 + k];
            f[k] = sacc * ASCALE + bias4[k];
        }
        const size_t pix = ((size_t)b * a.H + y + row) * a.W + x0 + p;
        auto ld4 = [&](const View& v, float (&o)[4]) {
            const ST* q = (const ST*)v.p + pix * v.cs + v.co + cg;
            if constexpr (H16) {
                typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                const h4_t r4 = *(const h4_t*)q;
                o[0] = (float)r4[0]; o[1] = (float)r4[1]; o[2] = (float)r4[2]; o[3] = (float)r4[3];
            } else if constexpr (sizeof(ST) == 2) {
                const unsigned long long raw = *(const unsigned long long*)q;
                o[0] = __uint_as_float((unsigned)(raw & 0xffffu) << 16);
                o[1] = __uint_as_float((unsigned)(raw & 0xffff0000u));
                o[2] = __uint_as_float((unsigned)((raw >> 32) & 0xffffu) << 16);
                o[3] = __uint_as_float((unsigned)((raw >> 32) & 0xffff0000u));
            } else {
                const f32x4 r4 = *(const f32x4*)q;
                o[0] = r4[0]; o[1] = r4[1]; o[2] = r4[2]; o[3] = r4[3];
            }
        };
        auto st4 = [&](const View& v, float (&o)[4]) {        // stores; o is replaced by the values as stored
            ST* q = (ST*)v.p + pix * v.cs + v.co + cg;
            if constexpr (H16) {
                typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                const h4_t r4 = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                *(h4_t*)q = r4;
                o[0] = (float)r4[0]; o[1] = (float)r4[1]; o[2] = (float)r4[2]; o[3] = (float)r4[3];
            } else if constexpr (sizeof(ST) == 2) {
                const unsigned lo = pack2bf(o[0], o[1]), hi = pack2bf(o[2], o[3]);
                *(unsigned long long*)q = (unsigned long long)lo | ((unsigned long long)hi << 32);
                o[0] = __uint_as_float(lo << 16); o[1] = __uint_as_float(lo & 0xffff0000u);
                o[2] = __uint_as_float(hi << 16); o[3] = __uint_as_float(hi & 0xffff0000u);
            } else {
                const f32x4 r4 = {o[0], o[1], o[2], o[3]};
                *(f32x4*)q = r4;
            }
        };
        if (a.raw.p) {
            float g[4] = {f[0], f[1], f[2], f[3]};
            st4(a.raw, g);
#pragma unroll
            for (int k = 0; k < 4; ++k) { sr[k] += g[k]; qr[k] += g[k] * g[k]; }
        }
        if (a.res.p) {
            float r4[4];
            ld4(a.res, r4);
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] += r4[k];
        }
        if (a.res2.p) {
            float r4[4];
            ld4(a.res2, r4);
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] += r4[k];
        }
        st4(a.out, f);
#pragma unroll
        for (int k = 0; k < 4; ++k) { so[k] += f[k]; qo[k] += f[k] * f[k]; }
    }

    if (want_stats) {   // uniform over the grid
        // a wave holds 8 pixels x 8 channel quads: fixed butterfly over the pixel bits, then the four waves in order
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sr[k] += __shfl_xor(sr[k], o, 64); qr[k] += __shfl_xor(qr[k], o, 64);
                so[k] += __shfl_xor(so[k], o, 64); qo[k] += __shfl_xor(qo[k], o, 64);
            }
        }
        if (lane < 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sred[(0 * 4 + wid) * 32 + lane * 4 + k] = sr[k];
                sred[(1 * 4 + wid) * 32 + lane * 4 + k] = qr[k];
                sred[(2 * 4 + wid) * 32 + lane * 4 + k] = so[k];
                sred[(3 * 4 + wid) * 32 + lane * 4 + k] = qo[k];
            }
        }
        __syncthreads();
        if (tid < 32) {
            float tsum[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tsum[k] = sred[(k * 4 + 0) * 32 + tid];
#pragma unroll
                for (int w = 1; w < 4; ++w) tsum[k] += sred[(k * 4 + w) * 32 + tid];
            }
            const int ch = n_tile * 32 + tid;
            if (a.st_raw) {
                const int gs = a.st_raw_C / GN_GROUPS;
                const float s1 = group_lane_sum(tsum[0], gs), s2 = group_lane_sum(tsum[1], gs);
                // the group's sums go out from different lanes of the group, concurrently (conv_lds.hip, same place)
                GroupStat* o = a.st_raw + (size_t)b * GN_GROUPS + (a.st_raw_co + ch) / gs;
                if (tid % gs == 0) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
            if (a.st_out) {
                const int gs = a.st_out_C / GN_GROUPS;
                const float s1 = group_lane_sum(tsum[2], gs), s2 = group_lane_sum(tsum[3], gs);
                GroupStat* o = a.st_out + (size_t)b * GN_GROUPS + (a.st_out_co + ch) / gs;
                if (tid % gs == (gs > 3 ? 2 : 0)) stat_add(&o->sum, act_hi_cells(a.B), s1);
                if (tid % gs == (gs > 3 ? 3 : (gs > 1 ? 1 : 0))) stat_add(&o->sq, act_hi_cells(a.B), s2);
            }
        }
    }
}

template <typename T, int CIN, int ROWS>
int launch_small_t(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    using G = SGeo<T, CIN, ROWS>;
    if constexpr (G::smem > 160 * 1024) {
        CHORE_FAIL(h, CHORE_EINVAL, "conv_small: %d rows x %d channels do not fit the LDS", ROWS, CIN);
    } else {
        bool& attr = CHORE_ONCE_FLAG(h);
        if (!attr) {
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)conv_small_kernel<T, CIN, ROWS>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::smem));
            attr = true;
        }
        const unsigned grid = (unsigned)((size_t)a.B * (a.H / ROWS) * (a.W / 32) * (a.Cout / 32));
        hipLaunchKernelGGL((conv_small_kernel<T, CIN, ROWS>), dim3(grid), dim3(256), G::smem, s, a);
        CHORE_LAUNCH_CHECK(h, s);
        return CHORE_OK;
    }
}

template <typename T, int CIN>
int launch_small_r(chore_handle* h, int rows, const ConvArgs& a, hipStream_t s) {
    switch (rows) {
        case 1: return launch_small_t<T, CIN, 1>(h, a, s);
        case 2: return launch_small_t<T, CIN, 2>(h, a, s);
        case 4: return launch_small_t<T, CIN, 4>(h, a, s);
    }
    CHORE_FAIL(h, CHORE_EINVAL, "conv_small: rows must be 1, 2 or 4");
}

template <typename T>
int launch_small_c(chore_handle* h, int rows, const ConvArgs& a, hipStream_t s) {
    switch (a.in.C) {
        case 64: return launch_small_r<T, 64>(h, rows, a, s);
        case 128: return launch_small_r<T, 128>(h, rows, a, s);
        case 256: return launch_small_r<T, 256>(h, rows, a, s);
    }
    CHORE_FAIL(h, CHORE_EINVAL, "conv_small: unsupported Cin=%d", a.in.C);
}

// rows per workgroup; 0 = use conv_lds_kernel.  Environment overrides for A/B measurements.
int small_rows(int dtype, int H, int W, int Cin) {
    static const int r32 = getenv("CHORE_CONV_SMALL_ROWS32") ? atoi(getenv("CHORE_CONV_SMALL_ROWS32")) : -1;
    static const int r64 = getenv("CHORE_CONV_SMALL_ROWS64") ? atoi(getenv("CHORE_CONV_SMALL_ROWS64")) : -1;
    static const int r64c = getenv("CHORE_CONV_SMALL_ROWS64_C256") ? atoi(getenv("CHORE_CONV_SMALL_ROWS64_C256")) : -1;
    const bool x3 = dtype == CHORE_F16X3;
    int rows;
    if (H * W <= 32 * 32) rows = r32 >= 0 ? r32 : ((x3 || dtype == CHORE_F16) ? 2 : 1);
    else if (Cin == 256) rows = r64c >= 0 ? r64c : 0;
    else rows = r64 >= 0 ? r64 : 0;
    // LDS: (rows + 2) x 34 pixels x (2 Cin + 16) bytes per plane, two planes for fp16 x 3
    while (rows > 0 && (size_t)(x3 ? 2 : 1) * (rows + 2) * PW * (Cin * 2 + 16) + (size_t)Cin * 8 + 2048 > 160 * 1024) rows >>= 1;
    if (rows && H % rows) rows = 0;
    return rows;
}

}  // namespace

// Where it is used (measured, bench.py encode time, B = 4: bf16 4.06 -> 3.79 ms, fp16 x 3 6.55 -> 6.1 ms).  A workgroup
// fetches its K slice of the weights (147 KB for 256 -> 128 channels) and its halo patch from L2 on its own, so the
// kernel trades L2 traffic for parallelism.  It wins on the 32 x 32 maps, where conv_lds_kernel has 16-64 workgroups for
// 256 CUs.  On the 64 x 64 maps it loses in every variant tried (1, 2 or 4 rows per workgroup, CHORE_CONV_SMALL_ROWS64*):
// 1 024-2 048 workgroups x 200 KB is L2-bandwidth bound (53 us against 26 us for the 256-channel convolution), and with
// several rows per workgroup the waves drift apart and each pulls the weight fragments through the 32 KB L1 again --
// sharing them takes the LDS ring and barriers of conv_lds_kernel, which therefore keeps those maps.  bf16 and
// fp16 x 3 only: the native-fp32 parity mode keeps the one kernel it was validated with.
bool conv_small_eligible(int dtype, int taps, int H, int W, int Cin, int Cout) {
    static const bool off = getenv("CHORE_NO_CONV_SMALL") != nullptr;      // A/B switch
    if (off || taps != 9 || (dtype != CHORE_BF16 && dtype != CHORE_F16X3 && dtype != CHORE_F16)) return false;
    if (W % 32 || Cout % 32 || (Cin != 64 && Cin != 128 && Cin != 256) || H * W > 64 * 64) return false;
    return small_rows(dtype, H, W, Cin) > 0;
}

int launch_conv_small(chore_handle* h, int dtype, const ConvArgs& a, hipStream_t s) {
    if ((size_t)a.B * a.H * (a.W / 32) * (a.Cout / 32) > 0x7fffffffull) CHORE_FAIL(h, CHORE_EINVAL, "conv_small: grid too large");
    const int rows = small_rows(dtype, a.H, a.W, a.in.C);
    if (dtype == CHORE_F16) return launch_small_c<h16_t>(h, rows, a, s);
    if (dtype == CHORE_F16X3 && a.in_amax) return launch_small_c<x3s_t>(h, rows, a, s);
    return dtype == CHORE_F16X3 ? launch_small_c<x3_t>(h, rows, a, s) : launch_small_c<bf16_t>(h, rows, a, s);
}

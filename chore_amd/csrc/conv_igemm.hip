// conv_igemm.hip -- 3x3 / 1x1 convolution as an implicit GEMM on the gfx950 matrix cores, with
// GroupNorm-apply + ReLU fused into the operand load and bias / residual / concat-slice fused into
// the epilogue.  Covers every conv of ConvBlock (model/net_util.py:346-396) and the 1x1 convs of
// the stack tail (model/HGFilters.py:128-142,167-183).
//
// GEMM view:  M = output pixels (a 16x16 tile per workgroup), N = Cout, K = taps x Cin.
//   * The input patch (tile + 1-pixel halo) of one 128-byte channel chunk is staged ONCE in LDS
//     ([324 rows][144 B], 16 B row padding) and re-read at 9 shifted positions -- no im2col.
//     While it is staged the fused prologue applies relu(x*scale + shift) (GroupNorm folded to a
//     per-(image,channel) affine by gn_finalize) and writes zeros for the halo outside the image,
//     which is exactly conv2d's zero padding of the normalised tensor.
//   * The next chunk's global loads are issued before the MFMA loop of the current chunk
//     (register prefetch), so HBM/L2 latency hides under the matrix work.
//   * Weights are pre-packed in fragment order [tap][k-group][n-block][lane][16 B]; a wave reads
//     its B fragment with one coalesced 1 KB load straight from L2 (they are shared by every
//     workgroup, so they stay cache resident).
//   * A fragments: one ds_read_b128 per 32-pixel block.  T = bf16: v_mfma_f32_32x32x16_bf16
//     (8 bf16 per lane); T = fp32: four v_mfma_f32_32x32x2_f32 per read (exact fp32, parity mode).
//   * Waves split N first (each owns a 32-channel block), then M: NT=128 -> 4(N)x1(M) waves with
//     8 pixel blocks each, NT=64 -> 2x2, NT=32 -> 1x4.
#include "enc_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int CT = 16;        // tile edge (pixels)
constexpr int ROWB = 144;     // LDS row stride in bytes

template <typename T> struct ConvT;
template <> struct ConvT<float> {
    static constexpr int VE = 4;    // elements per 16-byte vector
    static constexpr int KGE = 8;   // channels per k-group (one ds_read_b128 per half-wave pair)
};
template <> struct ConvT<bf16_t> {
    static constexpr int VE = 8;
    static constexpr int KGE = 16;
};

template <typename T>
__device__ __forceinline__ u32x4 transform_vec(u32x4 raw, const float* sc, const float* sh, bool use_gn);

template <>
__device__ __forceinline__ u32x4 transform_vec<float>(u32x4 raw, const float* sc, const float* sh, bool use_gn) {
    if (!use_gn) return raw;
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t = fmaf(__uint_as_float(raw[j]), sc[j], sh[j]);
        o[j] = __float_as_uint(t > 0.f ? t : 0.f);
    }
    return o;
}
template <>
__device__ __forceinline__ u32x4 transform_vec<bf16_t>(u32x4 raw, const float* sc, const float* sh, bool use_gn) {
    if (!use_gn) return raw;
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float lo = __uint_as_float(raw[j] << 16), hi = __uint_as_float(raw[j] & 0xffff0000u);
        float a = fmaf(lo, sc[2 * j], sh[2 * j]), b = fmaf(hi, sc[2 * j + 1], sh[2 * j + 1]);
        a = a > 0.f ? a : 0.f;
        b = b > 0.f ? b : 0.f;
        o[j] = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
    }
    return o;
}

// value as it will be read back after being stored as T
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <typename T> __device__ __forceinline__ float ld_elem(const T* p);
template <> __device__ __forceinline__ float ld_elem<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_elem<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st_elem(T* p, float v);
template <> __device__ __forceinline__ void st_elem<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_elem<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

template <typename T>
__device__ __forceinline__ void mfma_step(f32x16& acc, const u32x4& av, const u32x4& bw) {
    if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av),
                                                      __builtin_bit_cast(bf16x8_t, bw), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[i]), __uint_as_float(bw[i]), acc, 0, 0, 0);
    }
}

// TH = tile height in pixels (tile = TH x 16), KGC = k-groups per staged chunk (4; 2 for bf16 Cin=32)
template <typename T, int TAPS, int NT, int TH, int KGC>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
    constexpr int VE = ConvT<T>::VE, KGE = ConvT<T>::KGE;
    constexpr int CC = KGC * KGE;                     // channels per staged chunk
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int PW = CT + 2 * PAD, PH = TH + 2 * PAD;
    constexpr int ROWS = PH * PW;
    constexpr int NV = (ROWS * 8 + 255) / 256;        // 16-byte vectors per thread per chunk
    constexpr int WAVES_N = NT / 32, WAVES_M = 4 / WAVES_N, MB = (TH / 2) / WAVES_M;
    static_assert(MB >= 1 && MB <= 4, "tile/wave layout outside the register budget");
    static_assert(KGC % 2 == 0, "A double-buffer parity needs an even number of k-groups per chunk");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                                // [ROWS][ROWB]
    float* ss_lds = (float*)(smem + ROWS * ROWB);      // [Cin][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % WAVES_N, wm = wid / WAVES_N;
    const int tiles_x = (a.W + CT - 1) / CT;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * CT;
    const int n_tile = blockIdx.y, b = blockIdx.z;
    const int Cin = a.in.C;
    const bool use_gn = a.ss != nullptr;

    if (use_gn)
        for (int i = tid; i < Cin * 2; i += 256) ss_lds[i] = a.ss[(size_t)b * Cin * 2 + i];

    // ---- per-thread staging coordinates (vector slot v is the same for all of a thread's rows) ----
    const int v = tid & 7;
    const T* in_b = (const T*)a.in.p + (size_t)b * a.H * a.W * a.in.cs + a.in.co;
    int row_off[NV];     // element offset of the source pixel, -1 = outside the image / tile table
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int row = (tid + j * 256) >> 3;
        const int y = ty0 + row / PW - PAD, x = tx0 + row % PW - PAD;
        const bool ok = (row < ROWS) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        row_off[j] = ok ? (y * a.W + x) * a.in.cs : -1;
    }
    u32x4 pre[NV];
    auto load_chunk = [&](int c0) {
        const bool cok = (v * VE < CC) && (c0 + v * VE) < Cin;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u32x4 z = {0u, 0u, 0u, 0u};
            pre[j] = (row_off[j] >= 0 && cok) ? *(const u32x4*)(in_b + row_off[j] + c0 + v * VE) : z;
        }
    };

    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const int NKG = Cin / KGE, NB = a.Cout / 32;
    const int nb = n_tile * (NT / 32) + wn;
    const u32x4* wp = (const u32x4*)a.wpk + (size_t)nb * 64 + lane;
    const size_t wstride = (size_t)NB * 64;            // u32x4 elements between consecutive k-groups
    const int half = lane >> 5, prow = lane & 31;
    const char* a_ptr = patch + (((prow >> 4) + 2 * wm * MB) * PW + (prow & 15)) * ROWB + 16 * half;

    // B fragments of one tap of the chunk starting at k-group kg0 (KGC coalesced 1 KB loads)
    auto load_b_tap = [&](u32x4 (&dst)[KGC], int kg0, int tap) {
        const u32x4* q = wp + ((size_t)tap * NKG + kg0) * wstride;
#pragma unroll
        for (int kg = 0; kg < KGC; ++kg) dst[kg] = q[kg * wstride];
    };
    auto load_a = [&](u32x4 (&dst)[MB], int tapoff, int kg) {
#pragma unroll
        for (int m = 0; m < MB; ++m) dst[m] = *(const u32x4*)(a_ptr + tapoff + (2 * m * PW) * ROWB + kg * 32);
    };
    auto tap_offset = [&](int tap) -> int {
        return (TAPS == 9) ? ((tap / 3) * PW + (tap % 3)) * ROWB : 0;
    };

    load_chunk(0);
    u32x4 bcur[KGC], bnxt[KGC];
    load_b_tap(bcur, 0, 0);
    __syncthreads();  // ss_lds visible
    for (int c0 = 0; c0 < Cin; c0 += CC) {
        // ---- write the prefetched chunk (GroupNorm + ReLU applied here) ----
        float sc[VE], sh[VE];
        if (use_gn && (v * VE < CC) && (c0 + v * VE) < Cin) {
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                sc[j] = ss_lds[(c0 + v * VE + j) * 2];
                sh[j] = ss_lds[(c0 + v * VE + j) * 2 + 1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < VE; ++j) { sc[j] = 0.f; sh[j] = 0.f; }
        }
        if (c0) __syncthreads();  // all waves done reading the previous chunk
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + j * 256;
            if (idx < ROWS * 8) {
                u32x4 val = {0u, 0u, 0u, 0u};
                if (row_off[j] >= 0) val = transform_vec<T>(pre[j], sc, sh, use_gn);
                *(u32x4*)(patch + (idx >> 3) * ROWB + v * 16) = val;
            }
        }
        __syncthreads();
        const bool more = c0 + CC < Cin;
        if (more) load_chunk(c0 + CC);

        // ---- matrix work of this chunk.  Software pipeline: the B fragments of the NEXT tap and the
        //      A fragments of the NEXT k-group are in flight while the current MFMAs issue. ----
        const int kg0 = c0 / KGE;
        u32x4 aq[2][MB];
        load_a(aq[0], 0, 0);
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap) {
            const bool last_tap = tap + 1 == TAPS;
            if (!last_tap) load_b_tap(bnxt, kg0, tap + 1);
            else if (more) load_b_tap(bnxt, kg0 + KGC, 0);   // first tap of the next chunk
            const int toff = tap_offset(tap), toff_n = tap_offset(last_tap ? tap : tap + 1);
#pragma unroll
            for (int kg = 0; kg < KGC; ++kg) {
                if (kg + 1 < KGC) load_a(aq[(kg + 1) & 1], toff, kg + 1);
                else if (!last_tap) load_a(aq[0], toff_n, 0);   // KGC is even: parity restarts at 0
#pragma unroll
                for (int m = 0; m < MB; ++m) mfma_step<T>(acc[m], aq[kg & 1][m], bcur[kg]);
            }
#pragma unroll
            for (int kg = 0; kg < KGC; ++kg) bcur[kg] = bnxt[kg];
        }
    }

    // ---- epilogue: bias, residuals, slice stores.  Lane = one output channel x 16 pixels/block ----
    const int n = nb * 32 + prow;
    const float bias = a.bias ? a.bias[n] : 0.f;
    T* out_b = (T*)a.out.p + (size_t)b * a.H * a.W * a.out.cs + a.out.co + n;
    T* raw_b = a.raw.p ? (T*)a.raw.p + (size_t)b * a.H * a.W * a.raw.cs + a.raw.co + n : nullptr;
    const T* res_b = a.res.p ? (const T*)a.res.p + (size_t)b * a.H * a.W * a.res.cs + a.res.co + n : nullptr;
    const T* res2_b = a.res2.p ? (const T*)a.res2.p + (size_t)b * a.H * a.W * a.res2.cs + a.res2.co + n : nullptr;
    float s_raw = 0.f, q_raw = 0.f, s_out = 0.f, q_out = 0.f;   // GroupNorm partials of the STORED values
#pragma clang loop unroll(full)
    for (int m = 0; m < MB; ++m) {
#pragma clang loop unroll(full)
        for (int r = 0; r < 16; ++r) {
            const int p = mfma32_row(r, half);
            const int y = ty0 + 2 * (wm * MB + m) + (p >> 4), x = tx0 + (p & 15);
            if (y < a.H && x < a.W) {
                const size_t pix = (size_t)y * a.W + x;
                float val = acc[m][r] + bias;
                if (raw_b) {
                    const float st = round_to<T>(val);
                    st_elem<T>(raw_b + pix * a.raw.cs, st);
                    s_raw += st;
                    q_raw += st * st;
                }
                if (res_b) val += ld_elem<T>(res_b + pix * a.res.cs);
                if (res2_b) val += ld_elem<T>(res2_b + pix * a.res2.cs);
                const float st = round_to<T>(val);
                st_elem<T>(out_b + pix * a.out.cs, st);
                s_out += st;
                q_out += st * st;
            }
        }
    }
    if (a.st_raw || a.st_out) {   // uniform over the grid
        // lane (half 0 | half 1) of one channel -> half 0; waves that split M -> through LDS, fixed order
        s_raw += __shfl_xor(s_raw, 32, 64); q_raw += __shfl_xor(q_raw, 32, 64);
        s_out += __shfl_xor(s_out, 32, 64); q_out += __shfl_xor(q_out, 32, 64);
        __syncthreads();   // every wave is done with the patch; reuse it as scratch
        float* red = (float*)smem;   // [4][WAVES_M][NT]
        if (half == 0) {
            const int c = wn * 32 + prow;
            red[(0 * WAVES_M + wm) * NT + c] = s_raw;
            red[(1 * WAVES_M + wm) * NT + c] = q_raw;
            red[(2 * WAVES_M + wm) * NT + c] = s_out;
            red[(3 * WAVES_M + wm) * NT + c] = q_out;
        }
        __syncthreads();
        if (tid < NT) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int w = 0; w < WAVES_M; ++w) t[k] += red[(k * WAVES_M + w) * NT + tid];
            const int c = n_tile * NT + tid;
            if (a.st_raw) {
                const size_t tile = (size_t)b * a.st_raw_tiles + blockIdx.x;
                float* o = a.st_raw + (tile * a.st_raw_C + a.st_raw_co + c) * 2;
                o[0] = t[0];
                o[1] = t[1];
            }
            if (a.st_out) {
                const size_t tile = (size_t)b * a.st_out_tiles + blockIdx.x;
                float* o = a.st_out + (tile * a.st_out_C + a.st_out_co + c) * 2;
                o[0] = t[2];
                o[1] = t[3];
            }
        }
    }
}

template <typename T, int TAPS, int NT, int TH, int KGC>
static int launch_conv_t(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    constexpr int PW = (TAPS == 9) ? CT + 2 : CT, PH = (TAPS == 9) ? TH + 2 : TH;
    const size_t smem = (size_t)PH * PW * ROWB + (size_t)a.in.C * 2 * sizeof(float);
    const int tiles = ((a.W + CT - 1) / CT) * ((a.H + TH - 1) / TH);
    dim3 grid(tiles, a.Cout / NT, a.B);
    hipLaunchKernelGGL((conv_igemm_kernel<T, TAPS, NT, TH, KGC>), grid, dim3(256), smem, s, a);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// Tile / N-split choice: prefer the largest tile and N-tile that still gives every CU about two
// workgroups (256 CUs); small maps fall back to the configuration with the most workgroups.
struct ConvCfg { int nt, th; };
static ConvCfg choose_cfg(const ConvArgs& a);
ConvPlan conv_plan(int B, int H, int W, int Cout) {
    ConvArgs a{};
    a.B = B; a.H = H; a.W = W; a.Cout = Cout;
    const ConvCfg c = choose_cfg(a);
    return ConvPlan{c.nt, c.th, ((W + CT - 1) / CT) * ((H + c.th - 1) / c.th)};
}
static ConvCfg choose_cfg(const ConvArgs& a) {
    static const int nts[3] = {128, 64, 32}, ths[2] = {16, 8};
    ConvCfg best{0, 0};
    long best_wgs = -1;
    for (int ni = 0; ni < 3; ++ni) {
        const int nt = nts[ni];
        if (a.Cout % nt) continue;
        for (int ti = 0; ti < 2; ++ti) {
            const int th = ths[ti];
            const int mb = (th / 2) / (4 / (nt / 32));
            if (mb < 1 || mb > 4) continue;               // register budget: at most 4 pixel blocks per wave
            const long wgs = (long)a.B * ((a.W + CT - 1) / CT) * ((a.H + th - 1) / th) * (a.Cout / nt);
            if (wgs >= 448) return ConvCfg{nt, th};
            if (wgs > best_wgs) { best_wgs = wgs; best = ConvCfg{nt, th}; }
        }
    }
    return best;
}

template <typename T, int TAPS, int KGC>
static int launch_conv_cfg(chore_handle* h, const ConvArgs& a, hipStream_t s) {
    const ConvCfg c = choose_cfg(a);
    if (c.nt == 128 && c.th == 8) return launch_conv_t<T, TAPS, 128, 8, KGC>(h, a, s);
    if (c.nt == 64 && c.th == 16) return launch_conv_t<T, TAPS, 64, 16, KGC>(h, a, s);
    if (c.nt == 64 && c.th == 8) return launch_conv_t<T, TAPS, 64, 8, KGC>(h, a, s);
    if (c.nt == 32 && c.th == 16) return launch_conv_t<T, TAPS, 32, 16, KGC>(h, a, s);
    if (c.nt == 32 && c.th == 8) return launch_conv_t<T, TAPS, 32, 8, KGC>(h, a, s);
    CHORE_FAIL(h, CHORE_EINVAL, "conv: unsupported Cout=%d", a.Cout);
}

int launch_conv(chore_handle* h, int dtype, int taps, const ConvArgs& a, hipStream_t s) {
    const int kge = dtype == CHORE_F32 ? 8 : 16;
    if (a.in.C % (2 * kge) || a.in.C > 256) CHORE_FAIL(h, CHORE_EINVAL, "conv: unsupported Cin=%d", a.in.C);
    if (a.Cout % 32) CHORE_FAIL(h, CHORE_EINVAL, "conv: unsupported Cout=%d", a.Cout);
    if (a.B > 65535) CHORE_FAIL(h, CHORE_EINVAL, "conv: B too large");
    if (dtype == CHORE_F32)
        return taps == 9 ? launch_conv_cfg<float, 9, 4>(h, a, s) : launch_conv_cfg<float, 1, 4>(h, a, s);
    if (a.in.C % 64)  // bf16 with Cin = 32 (or 96): 2 k-groups per chunk
        return taps == 9 ? launch_conv_cfg<bf16_t, 9, 2>(h, a, s) : launch_conv_cfg<bf16_t, 1, 2>(h, a, s);
    return taps == 9 ? launch_conv_cfg<bf16_t, 9, 4>(h, a, s) : launch_conv_cfg<bf16_t, 1, 4>(h, a, s);
}

// ------------------------------------------------------------------------------------------------
// weight packing: (O,C,kh,kw) fp32 -> [tap][kg][nb][lane][16 B]
// ------------------------------------------------------------------------------------------------
size_t packed_conv_bytes(int dtype, int taps, int Cin, int Cout) {
    const int kge = dtype == CHORE_F32 ? 8 : 16;
    return (size_t)taps * (Cin / kge) * (Cout / 32) * 1024;
}

template <typename T>
__global__ void pack_conv_kernel(int taps, int Cin, int Cout, const float* __restrict__ w, u32x4* __restrict__ dst,
                                 size_t nvec) {
    constexpr int VE = ConvT<T>::VE, KGE = ConvT<T>::KGE;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
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
    for (int j = 0; j < VE; ++j) vals[j] = w[((size_t)n * Cin + c0 + j) * taps + tap];
    u32x4 o;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (unsigned)f2bf(vals[2 * j]) | ((unsigned)f2bf(vals[2 * j + 1]) << 16);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __float_as_uint(vals[j]);
    }
    dst[i] = o;
}

int launch_pack_conv(chore_handle* h, int dtype, int taps, int Cin, int Cout, const float* w, void* dst,
                     hipStream_t s) {
    const size_t nvec = packed_conv_bytes(dtype, taps, Cin, Cout) / 16;
    const unsigned blocks = (unsigned)((nvec + 255) / 256);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(blocks), dim3(256), 0, s, taps, Cin, Cout, w, (u32x4*)dst, nvec);
    else
        hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, taps, Cin, Cout, w, (u32x4*)dst,
                           nvec);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

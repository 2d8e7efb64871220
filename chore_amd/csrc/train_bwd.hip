// train_bwd.hip -- backward kernels of the encoder layers (training path).
//
//   * weight gradient of a 3x3 / 1x1 convolution whose input is relu(groupnorm(x)) (or x itself):
//       dW[o][c][t] = sum_{b,y,x} dY[b,y,x,o] * A[b,y+dy(t),x+dx(t),c]          (model/net_util.py:346-396 layers)
//     an MFMA GEMM whose contraction index is the PIXEL, while both tensors are stored channel-contiguous.
//       fp32: v_mfma_f32_32x32x2_f32 takes one scalar per lane and k = lane>>5 -- the [pixel][channel] images feed it
//             directly (exact fp32, the parity mode);
//       bf16: v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane = 8 consecutive pixels of one channel; the
//             fragments come from the same row-major images through ds_read_b64_tr_b16 (in a 16-lane group lane i
//             addresses 4 contiguous elements of row i>>2, column chunk i&3 of a 4x16 block and receives column i;
//             measured with scripts/probes/tr_probe.hip), two reads per fragment, no software transpose.
//     A workgroup owns 32 output x 32 input channels (all taps) and walks a share of the 8x32-pixel tiles; its four
//     waves split the tile rows and reduce through LDS in a fixed order at the end; the per-share partials are
//     summed in order by a second kernel: no float atomics, bit-reproducible.
//   * GroupNorm+ReLU backward: one streaming pass accumulates, exactly (the integer accumulators of
//     enc_common.h), dgamma/dbeta per channel and the two per-(image, group) sums the input gradient needs; a
//     second pass writes dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)), g = dA * [y > 0].
#include "enc_common.h"

typedef __bf16 tb_bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TH = 8, TW = 32;                 // pixel tile
constexpr int PW = TW + 2, PH = TH + 2;        // halo tile of the 3x3 case
constexpr int CT32 = 32;                       // channels per operand tile

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const void* x;            // (B,H,W,Cin) input of the layer (before GroupNorm)
    const GroupStat* st;      // statistics of x, or null: the conv saw x itself
    const float* gamma; const float* beta;
    const void* dy;           // (B,H,W,Cout)
    int B, H, W, Cin, Cout;
    int xs, ys;               // row strides (elements) of x and dy: Cin / Cout for dense NHWC tensors
    long long npix;           // pixels that exist (B*H*W, or fewer when a row list is walked as a ragged image)
    float* part;              // [S][Cout/32][Cin/32][TAPS][32][32] partial sums
    float* part_bias;         // [S][Cout] or null
    int S;                    // shares of the pixel tiles
    const unsigned* dy_amax;  // fp16 x 3 kernels: AMAX_CELLS partial maxima of |dy| (enc_common.h), or null
    int dbg;                  // CHORE_WGRAD_DBG ablation bits (measurements only; results are wrong when set): 1 no MFMAs, 2 no split / LDS
                              // stores, 4 no global loads
};

template <typename T, int TAPS>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    constexpr int AW = TW + 2 * PAD, AH = TH + 2 * PAD, AROWS = AH * AW;
    constexpr int ES = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* imgA = (T*)smem;                                   // [AROWS][32]  relu(gn(x)) halo tile, zero outside the image
    T* imgY = (T*)(smem + AROWS * CT32 * ES);             // [256][32]    dY tile
    float* ss = (float*)(smem + (AROWS + TH * TW) * CT32 * ES);   // [32][2] scale/shift of this image's channels
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int co0 = blockIdx.x * CT32, ci0 = blockIdx.y * CT32, share = blockIdx.z;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles = a.B * tiles_x * tiles_y;
    const bool use_gn = a.st != nullptr;
    const T* X = (const T*)a.x;
    const T* DY = (const T*)a.dy;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bias_acc = 0.f;                                 // thread tid < 32: column sum of dY for channel co0 + tid
    int cur_b = -1;

    for (int tile = share; tile < tiles; tile += a.S) {
        const int b = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
        const int ty0 = (tt / tiles_x) * TH, tx0 = (tt % tiles_x) * TW;
        __syncthreads();                                  // previous tile fully consumed
        if (use_gn && b != cur_b) {
            if (tid < CT32) gn_scale_shift(a.st, a.B, b, a.Cin, ci0 + tid, a.H * a.W, a.gamma, a.beta, ss[2 * tid], ss[2 * tid + 1]);
            __syncthreads();
        }
        cur_b = b;
        // ---- stage: 8 channels (fp32: 4) per thread-vector ----
        constexpr int VE = 16 / ES, VPR = CT32 / VE;      // elements per 16-byte vector, vectors per row
        for (int i = tid; i < AROWS * VPR; i += 256) {
            const int row = i / VPR, v = i % VPR;
            const int y = ty0 + row / AW - PAD, x = tx0 + row % AW - PAD;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (y >= 0 && y < a.H && x >= 0 && x < a.W && ((long long)b * a.H + y) * a.W + x < a.npix) {
                val = *(const u32x4*)(X + (((size_t)b * a.H + y) * a.W + x) * a.xs + ci0 + v * VE);
                if (use_gn) {
                    float f[8];
                    if constexpr (ES == 2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(val[j] << 16); f[2 * j + 1] = __uint_as_float(val[j] & 0xffff0000u); }
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], ss[2 * (v * 8 + j)], ss[2 * (v * 8 + j) + 1]); f[j] = t > 0.f ? t : 0.f; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) val[j] = pack2bf(f[2 * j], f[2 * j + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float t = fmaf(__uint_as_float(val[j]), ss[2 * (v * 4 + j)], ss[2 * (v * 4 + j) + 1]);
                            val[j] = __float_as_uint(t > 0.f ? t : 0.f);
                        }
                    }
                }
            }
            *(u32x4*)(imgA + row * CT32 + v * VE) = val;
        }
        for (int i = tid; i < TH * TW * VPR; i += 256) {
            const int row = i / VPR, v = i % VPR;
            const int y = ty0 + row / TW, x = tx0 + row % TW;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (y < a.H && x < a.W && ((long long)b * a.H + y) * a.W + x < a.npix)
                val = *(const u32x4*)(DY + (((size_t)b * a.H + y) * a.W + x) * a.ys + co0 + v * VE);
            *(u32x4*)(imgY + row * CT32 + v * VE) = val;
        }
        __syncthreads();
        if (a.part_bias && blockIdx.y == 0 && tid < CT32) {
            for (int p = 0; p < TH * TW; ++p) bias_acc += ld1<T>(imgY + p * CT32 + tid);
        }
        // ---- MFMAs: wave `wid` owns tile rows 2*wid, 2*wid+1 ----
        if constexpr (ES == 4) {
            const int half = lane >> 5, col = lane & 31;
#pragma unroll 1
            for (int yy = 0; yy < 2; ++yy) {
                const int y = 2 * wid + yy;
#pragma unroll 4
                for (int x = 0; x < TW; x += 2) {
                    const float av = ((const float*)imgY)[(y * TW + x + half) * CT32 + col];     // A[m = co][k = pixel]
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const int ky = TAPS == 9 ? t / 3 : 0, kx = TAPS == 9 ? t % 3 : 0;
                        const float bv = ((const float*)imgA)[((y + ky) * AW + x + kx + half) * CT32 + col];   // B[k][n = ci]
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        } else {
            // transposing reads: lane -> (row within the 4-row block, 4-column chunk) of its 16-lane group
            const int i16 = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
            const int rsel = i16 >> 2, csel = (i16 & 3) * 4 + g * 16;
            auto frag = [&](const T* img, int row0) -> tb_bf16x8 {      // rows row0 + 8h + {0..3} and {4..7}
                unsigned long long lo, hi;
                const unsigned a0 = (unsigned)(size_t)(img + (row0 + 8 * h + rsel) * CT32 + csel);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(hi) : "v"(a0));   // +4 rows x 64 B
                // the wait carries the two results as operands: an MFMA that uses them cannot be scheduled above it (a bare
                // "memory" clobber does not order a register-only MFMA -- it read the destination before the data arrived)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 v = {lo, hi};
                return __builtin_bit_cast(tb_bf16x8, v);
            };
#pragma unroll 1
            for (int yy = 0; yy < 2; ++yy) {
                const int y = 2 * wid + yy;
#pragma unroll 1
                for (int x = 0; x < TW; x += 16) {
                    const tb_bf16x8 av = frag(imgY, y * TW + x);
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const int ky = TAPS == 9 ? t / 3 : 0, kx = TAPS == 9 ? t % 3 : 0;
                        const tb_bf16x8 bv = frag(imgA, (y + ky) * AW + x + kx);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- reduce the four waves (fixed order) and write this share's partial ----
    __syncthreads();
    float* red = (float*)smem;                            // [4][32][32] one tap at a time
    const int half = lane >> 5, col = lane & 31;
    float* out = a.part + ((((size_t)share * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * TAPS) * 1024;
#pragma unroll 1
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wid * 32 + mfma32_row(r, half)) * 32 + col] = acc[t][r];
        __syncthreads();
        for (int i = tid; i < 1024; i += 256)
            out[(size_t)t * 1024 + i] = ((red[i] + red[1024 + i]) + red[2048 + i]) + red[3072 + i];
        __syncthreads();
    }
    if (a.part_bias && blockIdx.y == 0 && tid < CT32) a.part_bias[(size_t)share * a.Cout + co0 + tid] = bias_acc;
}

// ------------------------------------------------------------------------------------------------
// bf16 weight gradient, 64 x 64 channel tiles (Cin, Cout multiples of 64: the layers that carry the FLOPs).
// The 32x32 kernel above waits for every LDS fragment, loads its tiles one vector at a time and re-stages both
// tensors once per 32-channel tile pair.  Here a workgroup owns 64 output x 64 input channels x all taps; wave
// (coh, cih) owns the 32 x 32 channel block (TAPS accumulator tiles = 144 registers at 9 taps) over the WHOLE pixel
// tile, so there is no cross-wave reduction, and the tensors are staged half as often.  Fragments come from the
// transposing-read builtin, so the compiler overlaps them with the MFMAs (1 dY + TAPS x fragments per TAPS MFMAs:
// ~5.1k LDS cycles against 4.6k MFMA cycles per tile).  LDS images are [pixel][64 ch] with a 160-byte pitch: the
// four 4x16 blocks a transposing read touches fall into disjoint banks.  Staging issues all vector loads of a
// tensor before the first use.
// Workgroups that share pixel tiles (same share, different channel pair) are placed on one XCD (id % 8) so the
// second and later readers of a tile hit that XCD's L2.
// ------------------------------------------------------------------------------------------------
constexpr int W64_PITCH = 160;                 // bytes per pixel row of the LDS images (64 bf16 = 128 + 32 pad)
typedef short tb_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tb_s16x4 lds_s16x4;

// Three taps of one kernel row from ONE set of transposing reads (fp16 / bf16 weight gradients, 3x3): the fragments of taps kx = 0, 1, 2
// are 8-pixel windows starting at pixels 0, 1, 2 of the same channel -- 10 of 12 pixels shared.  Three ds_read_b64_tr_b16 (pixels 0-3,
// 4-7, 8-11 of this lane's k half) give six dwords; window 0 = dwords 0-3, window 2 = dwords 1-4, window 1 = four 16-bit funnel
// shifts.  6 reads instead of 3 x 4 per (kernel row, plane).  Used by wgrad64_x3_kernel (150 -> 141 us on the largest layer, 1 - 4 us on
// the others); NOT by wgrad64_x3_pc_kernel, where it measured slower.
struct TrWin { unsigned d[6]; };
__device__ __forceinline__ TrWin tr_load3(const char* p) {
    const tb_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    const tb_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * W64_PITCH));
    const tb_s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * W64_PITCH));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b), uc = __builtin_bit_cast(u32x2_t, c);
    TrWin w;
    w.d[0] = ua[0]; w.d[1] = ua[1]; w.d[2] = ub[0]; w.d[3] = ub[1]; w.d[4] = uc[0]; w.d[5] = uc[1];
    return w;
}
template <typename FR>
__device__ __forceinline__ FR tr_window(const TrWin& w, int kx) {
    u32x4 v;
    if (kx == 0) v = u32x4{w.d[0], w.d[1], w.d[2], w.d[3]};
    else if (kx == 2) v = u32x4{w.d[1], w.d[2], w.d[3], w.d[4]};
    else v = u32x4{__builtin_amdgcn_alignbit(w.d[1], w.d[0], 16), __builtin_amdgcn_alignbit(w.d[2], w.d[1], 16),
                   __builtin_amdgcn_alignbit(w.d[3], w.d[2], 16), __builtin_amdgcn_alignbit(w.d[4], w.d[3], 16)};
    return __builtin_bit_cast(FR, v);
}


template <int TAPS>
__global__ __launch_bounds__(256) void wgrad64_kernel(WgradArgs a) {
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    constexpr int AW = TW + 2 * PAD, AH = TH + 2 * PAD, AROWS = AH * AW;
    constexpr int NVX = (AROWS * 8 + 255) / 256, NVY = TH * TW * 8 / 256;      // 16-byte vectors per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* imgA = smem;                                     // [AROWS] x 160 B
    char* imgY = smem + AROWS * W64_PITCH;                 // [256] x 160 B
    float* ss = (float*)(imgY + TH * TW * W64_PITCH);      // [64][2]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, cih = wid & 1, coh = wid >> 1;
    const int nbo = a.Cout / 64, nbc = a.Cin / 64, npairs = nbo * nbc;
    int share, pair;
    {
        const int L = blockIdx.x;
        if (a.S % 8 == 0) { const int xcd = L & 7, j = L >> 3; share = (j / npairs) * 8 + xcd; pair = j % npairs; }
        else { share = L / npairs; pair = L % npairs; }
    }
    const int co0 = (pair / nbc) * 64, ci0 = (pair % nbc) * 64;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles = a.B * tiles_x * tiles_y;
    const bool use_gn = a.st != nullptr;
    const bf16_t* X = (const bf16_t*)a.x;
    const bf16_t* DY = (const bf16_t*)a.dy;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bias_acc = 0.f;
    int cur_b = -1;
    const int h = lane >> 5, g = (lane >> 4) & 1, i16 = lane & 15;
    const int lane_off = (8 * h + (i16 >> 2)) * W64_PITCH + (g * 16 + (i16 & 3) * 4) * 2;
    auto frag = [&](const char* p) -> tb_bf16x8 {          // p: image + pixel * pitch + channel * 2 (+ lane_off)
        const tb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
        const tb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * W64_PITCH));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(tb_bf16x8, v);
    };

    for (int tile = share; tile < tiles; tile += a.S) {
        const int b = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
        const int ty0 = (tt / tiles_x) * TH, tx0 = (tt % tiles_x) * TW;
        __syncthreads();                                   // previous tile fully consumed
        if (use_gn && b != cur_b) {
            if (tid < 64) gn_scale_shift(a.st, a.B, b, a.Cin, ci0 + tid, a.H * a.W, a.gamma, a.beta, ss[2 * tid], ss[2 * tid + 1]);
            __syncthreads();
        }
        cur_b = b;
        {   // ---- dY tile: all loads in flight, then the LDS stores ----
            u32x4 vy[NVY];
#pragma unroll
            for (int q = 0; q < NVY; ++q) {
                const int i = tid + 256 * q, row = i >> 3, v = i & 7;
                const int y = min(ty0 + row / TW, a.H - 1), x = min(tx0 + row % TW, a.W - 1);
                vy[q] = *(const u32x4*)(DY + (((size_t)b * a.H + y) * a.W + x) * a.ys + co0 + v * 8);
            }
#pragma unroll
            for (int q = 0; q < NVY; ++q) {
                const int i = tid + 256 * q, row = i >> 3, v = i & 7;
                u32x4 val = vy[q];
                if (ty0 + row / TW >= a.H || tx0 + row % TW >= a.W) { val[0] = 0u; val[1] = 0u; val[2] = 0u; val[3] = 0u; }
                *(u32x4*)(imgY + row * W64_PITCH + v * 16) = val;
            }
        }
        {   // ---- halo tile of the input, GroupNorm + ReLU applied on the way ----
            u32x4 vx[NVX];
#pragma unroll
            for (int q = 0; q < NVX; ++q) {
                const int i = min(tid + 256 * q, AROWS * 8 - 1), row = i >> 3, v = i & 7;
                const int y = min(max(ty0 + row / AW - PAD, 0), a.H - 1), x = min(max(tx0 + row % AW - PAD, 0), a.W - 1);
                vx[q] = *(const u32x4*)(X + (((size_t)b * a.H + y) * a.W + x) * a.xs + ci0 + v * 8);
            }
#pragma unroll
            for (int q = 0; q < NVX; ++q) {
                const int i = tid + 256 * q, row = i >> 3, v = i & 7;
                const int y = ty0 + row / AW - PAD, x = tx0 + row % AW - PAD;
                u32x4 val = vx[q];
                if (use_gn) {
                    float f[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(val[j] << 16); f[2 * j + 1] = __uint_as_float(val[j] & 0xffff0000u); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], ss[2 * (v * 8 + j)], ss[2 * (v * 8 + j) + 1]); f[j] = t > 0.f ? t : 0.f; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) val[j] = pack2bf(f[2 * j], f[2 * j + 1]);
                }
                if (y < 0 || y >= a.H || x < 0 || x >= a.W) { val[0] = 0u; val[1] = 0u; val[2] = 0u; val[3] = 0u; }
                if (i < AROWS * 8) *(u32x4*)(imgA + row * W64_PITCH + v * 16) = val;
            }
        }
        __syncthreads();
        if (a.part_bias && pair % nbc == 0 && tid < 64) {
            for (int p = 0; p < TH * TW; ++p) bias_acc += bf2f(*(const bf16_t*)(imgY + p * W64_PITCH + tid * 2));
        }
        // ---- MFMAs ----
        const char* baseY = imgY + lane_off + coh * 64;
        const char* baseA = imgA + lane_off + cih * 64;
#pragma unroll 1
        for (int y = 0; y < TH; ++y) {
#pragma unroll
            for (int xb = 0; xb < TW; xb += 16) {
                const tb_bf16x8 fy = frag(baseY + (y * TW + xb) * W64_PITCH);
                if constexpr (TAPS == 9) {      // the three taps of a kernel row from one set of reads (tr_load3)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const TrWin w = tr_load3(baseA + ((y + ky) * AW + xb) * W64_PITCH);
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy, tr_window<tb_bf16x8>(w, kx), acc[ky * 3 + kx], 0, 0, 0);
                    }
                } else {
                    const tb_bf16x8 fx = frag(baseA + (y * AW + xb) * W64_PITCH);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy, fx, acc[0], 0, 0, 0);
                }
            }
        }
    }
    // ---- this share's partial: [tap][64 co][64 ci] ----
    const int col = lane & 31;
    float* out = a.part + ((size_t)share * npairs + pair) * TAPS * 4096;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[(size_t)t * 4096 + (coh * 32 + mfma32_row(r, h)) * 64 + cih * 32 + col] = acc[t][r];
    if (a.part_bias && pair % nbc == 0 && tid < 64) a.part_bias[(size_t)share * a.Cout + co0 + tid] = bias_acc;
}


typedef _Float16 tb_f16x8 __attribute__((ext_vector_type(8)));
// ------------------------------------------------------------------------------------------------
// fp16 x 3 weight gradient (CHORE_F16X3 training: fp32 tensors, fp32-grade result on the fp16 matrix cores).
// The structure of wgrad64_kernel -- 64 output x 64 input channels x all taps per workgroup, wave (coh, cih) owns a 32 x 32
// block over the whole pixel tile, transposing LDS reads -- with BOTH operands split into an fp16 hi and lo plane while they
// are staged:   dW += dYhi^T Ahi + dYlo^T Ahi + dYhi^T Alo   (three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation;
// the dropped lo x lo term is 2^-22 relative).  Four planes of [pixel][64 ch] fp16 images: the pixel tile is 4 x 32 instead of
// 8 x 32 (204 + 128 rows x 2 planes x 160 B = 106 KB).
// Range: A = relu(groupnorm(x)) is O(1) like in the forward convolution; dY is a gradient of any magnitude and is multiplied by
// the power of two that puts max |dY| (WgradArgs::dy_amax, from the kernel that produced dY) at 2^13 .. 2^14 before the split --
// an fp16 pair keeps 22 bits only above 2^-3 -- and the accumulators by its inverse at the end.
// ------------------------------------------------------------------------------------------------
constexpr int WX_TH = 4;

template <int TAPS>
__global__ __launch_bounds__(256) void wgrad64_x3_kernel(WgradArgs a) {
    f16_saturate_mode();
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    constexpr int AW = TW + 2 * PAD, AH = WX_TH + 2 * PAD, AROWS = AH * AW, YROWS = WX_TH * TW;
    constexpr int NVX = (AROWS * 8 + 255) / 256, NVY = YROWS * 8 / 256;        // 8-channel slots (two 16-byte loads) per thread
    constexpr int PLA = AROWS * W64_PITCH, PLY = YROWS * W64_PITCH;            // bytes of one plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* imgA = smem;                                     // [2 planes][AROWS] x 160 B
    char* imgY = smem + 2 * PLA;                           // [2 planes][YROWS] x 160 B
    float* ss = (float*)(imgY + 2 * PLY);                  // [64][2]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, cih = wid & 1, coh = wid >> 1;
    const int nbo = a.Cout / 64, nbc = a.Cin / 64, npairs = nbo * nbc;
    int share, pair;
    {
        const int L = blockIdx.x;
        if (a.S % 8 == 0) { const int xcd = L & 7, j = L >> 3; share = (j / npairs) * 8 + xcd; pair = j % npairs; }
        else { share = L / npairs; pair = L % npairs; }
    }
    const int co0 = (pair / nbc) * 64, ci0 = (pair % nbc) * 64;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + WX_TH - 1) / WX_TH;
    const int tiles = a.B * tiles_x * tiles_y;
    const bool use_gn = a.st != nullptr;
    const float* X = (const float*)a.x;
    const float* DY = (const float*)a.dy;
    float ymul, yinv;
    x3_in_scale(a.dy_amax, ymul, yinv);

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bias_acc = 0.f;
    int cur_b = -1;
    const int h = lane >> 5, g = (lane >> 4) & 1, i16 = lane & 15;
    const int lane_off = (8 * h + (i16 >> 2)) * W64_PITCH + (g * 16 + (i16 & 3) * 4) * 2;
    auto frag = [&](const char* p) -> tb_f16x8 {           // p: plane + pixel * pitch + channel * 2 (+ lane_off)
        const tb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
        const tb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * W64_PITCH));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(tb_f16x8, v);
    };
    // 8 fp32 values -> fp16 hi / lo vectors
    auto split8 = [](const float (&f)[8], u32x4& hi, u32x4& lo) {
        tb_f16x8 hh, ll;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hh[j] = (_Float16)f[j]; ll[j] = (_Float16)(f[j] - (float)hh[j]); }
        hi = __builtin_bit_cast(u32x4, hh);
        lo = __builtin_bit_cast(u32x4, ll);
    };

    // The global loads of tile i + 1 are issued BEFORE the MFMAs of tile i and consumed after them (registers vy / vx live across
    // the MFMA loop): with one wave per SIMD nothing else hides a load's round trip, and the first version -- load, wait, split,
    // store, MFMAs, one after the other -- spent 21 us per tile on 3 us of MFMAs (profiles/r05_wgrad_x3.txt).  Loads are
    // unconditional at clamped addresses (a branch around a load makes the compiler wait for everything in flight).
    u32x4 vy[NVY][2], vx[NVX][2];
    auto issue_loads = [&](int tile) {
        const int b = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
        const int ty0 = (tt / tiles_x) * WX_TH, tx0 = (tt % tiles_x) * TW;
#pragma unroll
        for (int q = 0; q < NVY; ++q) {
            const int i = tid + 256 * q, row = i >> 3, v = i & 7;
            const int y = min(ty0 + row / TW, a.H - 1), x = min(tx0 + row % TW, a.W - 1);
            const u32x4* p = (const u32x4*)(DY + (((size_t)b * a.H + y) * a.W + x) * a.ys + co0 + v * 8);
            vy[q][0] = p[0]; vy[q][1] = p[1];
        }
#pragma unroll
        for (int q = 0; q < NVX; ++q) {
            const int i = min(tid + 256 * q, AROWS * 8 - 1), row = i >> 3, v = i & 7;
            const int y = min(max(ty0 + row / AW - PAD, 0), a.H - 1), x = min(max(tx0 + row % AW - PAD, 0), a.W - 1);
            const u32x4* p = (const u32x4*)(X + (((size_t)b * a.H + y) * a.W + x) * a.xs + ci0 + v * 8);
            vx[q][0] = p[0]; vx[q][1] = p[1];
        }
    };
    if (share < tiles && !(a.dbg & 4)) issue_loads(share);
    for (int tile = share; tile < tiles; tile += a.S) {
        const int b = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
        const int ty0 = (tt / tiles_x) * WX_TH, tx0 = (tt % tiles_x) * TW;
        __syncthreads();                                   // previous tile fully consumed
        if (use_gn && b != cur_b) {
            if (tid < 64) gn_scale_shift(a.st, a.B, b, a.Cin, ci0 + tid, a.H * a.W, a.gamma, a.beta, ss[2 * tid], ss[2 * tid + 1]);
            __syncthreads();
        }
        cur_b = b;
        if (!(a.dbg & 2)) {
#pragma unroll
        for (int q = 0; q < NVY; ++q) {
            const int i = tid + 256 * q, row = i >> 3, v = i & 7;
            float f[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] = __uint_as_float(vy[q][0][j]) * ymul; f[4 + j] = __uint_as_float(vy[q][1][j]) * ymul; }
            u32x4 hi, lo;
            split8(f, hi, lo);
            if (ty0 + row / TW >= a.H || tx0 + row % TW >= a.W) { hi = u32x4{0u, 0u, 0u, 0u}; lo = hi; }
            *(u32x4*)(imgY + row * W64_PITCH + v * 16) = hi;
            *(u32x4*)(imgY + PLY + row * W64_PITCH + v * 16) = lo;
        }
#pragma unroll
        for (int q = 0; q < NVX; ++q) {
            const int i = tid + 256 * q, row = i >> 3, v = i & 7;
            const int y = ty0 + row / AW - PAD, x = tx0 + row % AW - PAD;
            float f[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] = __uint_as_float(vx[q][0][j]); f[4 + j] = __uint_as_float(vx[q][1][j]); }
            if (use_gn) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], ss[2 * (v * 8 + j)], ss[2 * (v * 8 + j) + 1]); f[j] = t > 0.f ? t : 0.f; }
            }
            u32x4 hi, lo;
            split8(f, hi, lo);
            if (y < 0 || y >= a.H || x < 0 || x >= a.W) { hi = u32x4{0u, 0u, 0u, 0u}; lo = hi; }
            if (i < AROWS * 8) {
                *(u32x4*)(imgA + row * W64_PITCH + v * 16) = hi;
                *(u32x4*)(imgA + PLA + row * W64_PITCH + v * 16) = lo;
            }
        }
        }
        __syncthreads();
        if (!(a.dbg & 4)) issue_loads(tile + a.S < tiles ? tile + a.S : tile);      // the next tile's operands travel under this tile's MFMAs
        if (a.part_bias && pair % nbc == 0 && tid < 64) {
            for (int p = 0; p < YROWS; ++p)
                bias_acc += (float)*(const _Float16*)(imgY + p * W64_PITCH + tid * 2) + (float)*(const _Float16*)(imgY + PLY + p * W64_PITCH + tid * 2);
        }
        // ---- MFMAs: small terms first, all three into the same accumulator ----
        const char* baseY = imgY + lane_off + coh * 64;
        const char* baseA = imgA + lane_off + cih * 64;
        if (!(a.dbg & 1)) {
#pragma unroll 1
        for (int y = 0; y < WX_TH; ++y) {
#pragma unroll
            for (int xb = 0; xb < TW; xb += 16) {
                const tb_f16x8 fyh = frag(baseY + (y * TW + xb) * W64_PITCH), fyl = frag(baseY + PLY + (y * TW + xb) * W64_PITCH);
                if constexpr (TAPS == 9) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const char* pa = baseA + ((y + ky) * AW + xb) * W64_PITCH;
                        const TrWin wh = tr_load3(pa), wl = tr_load3(pa + PLA);
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const tb_f16x8 fxh = tr_window<tb_f16x8>(wh, kx), fxl = tr_window<tb_f16x8>(wl, kx);
                            f32x16& ac = acc[ky * 3 + kx];
                            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyl, fxh, ac, 0, 0, 0);
                            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh, fxl, ac, 0, 0, 0);
                            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh, fxh, ac, 0, 0, 0);
                        }
                    }
                } else {
                    const char* pa = baseA + (y * AW + xb) * W64_PITCH;
                    const tb_f16x8 fxh = frag(pa), fxl = frag(pa + PLA);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyl, fxh, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh, fxl, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh, fxh, acc[0], 0, 0, 0);
                }
            }
        }
        }
    }
    // ---- this share's partial: [tap][64 co][64 ci] ----
    const int col = lane & 31;
    float* out = a.part + ((size_t)share * npairs + pair) * TAPS * 4096;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[(size_t)t * 4096 + (coh * 32 + mfma32_row(r, h)) * 64 + cih * 32 + col] = acc[t][r] * yinv;
    if (a.part_bias && pair % nbc == 0 && tid < 64) a.part_bias[(size_t)share * a.Cout + co0 + tid] = bias_acc * yinv;
}


// ------------------------------------------------------------------------------------------------
// fp16 x 3 weight gradient with SPECIALISED WAVES (round 5, second version; wgrad64_x3_kernel above is the first).
// wgrad64_x3_kernel runs its phases in turn -- split + LDS stores of a tile (1.9 us), then its MFMAs (4.6 us) -- on one wave per SIMD.
// Here a workgroup is 8 waves, two per SIMD: waves 0-3 (consumers) hold the 64 x 64-channel accumulators and only read fragments and
// issue MFMAs; waves 4-7 (producers) only move data: global loads of tile i + 2, GroupNorm + ReLU + hi / lo split of tile i + 1 into
// the OTHER LDS buffer.  The matrix pipe and the vector ALU of a SIMD issue independently (conv_pc.hip's premise).  Two buffers of four
// planes fit with 2 x 32-pixel tiles (2 x 64 KB); hand-over by one s_barrier per tile, which both roles reach the same number of
// times: producers write buffer k % 2 before barrier k, consumers read it between barriers k and k + 1.
// Same arithmetic as wgrad64_x3_kernel per tile; the tile is half as high, so a share's partial is a sum over twice as many tiles in the
// same order of pixels -- the rounding of the fp32 accumulation is that of a different tile split, results equal to fp32 round-off.
// ------------------------------------------------------------------------------------------------
constexpr int WP_TH = 2;

// LDS traffic of this wave done, then the workgroup barrier; vector-memory loads stay in flight across it (__syncthreads() would
// wait for the producers' prefetched global loads at every tile)
__device__ __forceinline__ void wp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int TAPS>
__global__ __launch_bounds__(512) void wgrad64_x3_pc_kernel(WgradArgs a) {
    f16_saturate_mode();
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    constexpr int AW = TW + 2 * PAD, AH = WP_TH + 2 * PAD, AROWS = AH * AW, YROWS = WP_TH * TW;
    constexpr int NVX = (AROWS * 8 + 255) / 256, NVY = YROWS * 8 / 256;
    constexpr int PLA = AROWS * W64_PITCH, PLY = YROWS * W64_PITCH;
    constexpr int BUF = 2 * PLA + 2 * PLY;                   // one buffer: A hi, A lo, Y hi, Y lo
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const int nbo = a.Cout / 64, nbc = a.Cin / 64, npairs = nbo * nbc;
    int share, pair;
    {
        const int L = blockIdx.x;
        if (a.S % 8 == 0) { const int xcd = L & 7, j = L >> 3; share = (j / npairs) * 8 + xcd; pair = j % npairs; }
        else { share = L / npairs; pair = L % npairs; }
    }
    const int co0 = (pair / nbc) * 64, ci0 = (pair % nbc) * 64;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + WP_TH - 1) / WP_TH;
    const int tiles = a.B * tiles_x * tiles_y;
    const int ntile = share < tiles ? (tiles - share + a.S - 1) / a.S : 0;      // tiles of this workgroup: share, share + S, ...
    float ymul, yinv;
    x3_in_scale(a.dy_amax, ymul, yinv);
    float* red = (float*)smem;                                 // after the loop: [256 producer threads][8] bias partials

    if (producer) {
        // =========================== producers ===========================
        const int ptid = tid & 255;
        const bool use_gn = a.st != nullptr;
        const float* X = (const float*)a.x;
        const float* DY = (const float*)a.dy;
        const int v = ptid & 7;                                // this thread's 8-channel slot of every row it stages
        float sc[8], sh[8], bias8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; bias8[j] = 0.f; }
        int cur_b = -1;
        u32x4 vy[NVY][2], vx[NVX][2];
        auto issue_loads = [&](int tile) {
            const int b = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
            const int ty0 = (tt / tiles_x) * WP_TH, tx0 = (tt % tiles_x) * TW;
#pragma unroll
            for (int q = 0; q < NVY; ++q) {
                const int row = (ptid + 256 * q) >> 3;
                const int y = min(ty0 + row / TW, a.H - 1), x = min(tx0 + row % TW, a.W - 1);
                const u32x4* p = (const u32x4*)(DY + (((size_t)b * a.H + y) * a.W + x) * a.ys + co0 + v * 8);
                vy[q][0] = p[0]; vy[q][1] = p[1];
            }
#pragma unroll
            for (int q = 0; q < NVX; ++q) {
                const int row = min(ptid + 256 * q, AROWS * 8 - 1) >> 3;
                const int y = min(max(ty0 + row / AW - PAD, 0), a.H - 1), x = min(max(tx0 + row % AW - PAD, 0), a.W - 1);
                const u32x4* p = (const u32x4*)(X + (((size_t)b * a.H + y) * a.W + x) * a.xs + ci0 + v * 8);
                vx[q][0] = p[0]; vx[q][1] = p[1];
            }
        };
        auto split8 = [](const float (&f)[8], u32x4& hi, u32x4& lo) {
            tb_f16x8 hh, ll;
#pragma unroll
            for (int j = 0; j < 8; ++j) { hh[j] = (_Float16)f[j]; ll[j] = (_Float16)(f[j] - (float)hh[j]); }
            hi = __builtin_bit_cast(u32x4, hh);
            lo = __builtin_bit_cast(u32x4, ll);
        };
        if (ntile > 0 && !(a.dbg & 4)) issue_loads(share);
        for (int it = 0; it < ntile; ++it) {
            const int tile = share + it * a.S;
            const int b = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
            const int ty0 = (tt / tiles_x) * WP_TH, tx0 = (tt % tiles_x) * TW;
            char* imgA = smem + (it & 1) * BUF;
            char* imgY = imgA + 2 * PLA;
            if (use_gn && b != cur_b) {                        // the affine of this thread's 8 channels (a few times per workgroup)
#pragma unroll
                for (int j = 0; j < 8; ++j) gn_scale_shift(a.st, a.B, b, a.Cin, ci0 + v * 8 + j, a.H * a.W, a.gamma, a.beta, sc[j], sh[j]);
                cur_b = b;
            }
            if (!(a.dbg & 2)) {
#pragma unroll
            for (int q = 0; q < NVY; ++q) {
                const int row = (ptid + 256 * q) >> 3;
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] = __uint_as_float(vy[q][0][j]) * ymul; f[4 + j] = __uint_as_float(vy[q][1][j]) * ymul; }
                const bool out = ty0 + row / TW >= a.H || tx0 + row % TW >= a.W;
                u32x4 hi, lo;
                split8(f, hi, lo);
                if (out) { hi = u32x4{0u, 0u, 0u, 0u}; lo = hi; }
                else if (a.part_bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bias8[j] += f[j];
                }
                *(u32x4*)(imgY + row * W64_PITCH + v * 16) = hi;
                *(u32x4*)(imgY + PLY + row * W64_PITCH + v * 16) = lo;
            }
#pragma unroll
            for (int q = 0; q < NVX; ++q) {
                const int i = ptid + 256 * q, row = i >> 3;
                const int y = ty0 + row / AW - PAD, x = tx0 + row % AW - PAD;
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] = __uint_as_float(vx[q][0][j]); f[4 + j] = __uint_as_float(vx[q][1][j]); }
                if (use_gn) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], sc[j], sh[j]); f[j] = t > 0.f ? t : 0.f; }
                }
                u32x4 hi, lo;
                split8(f, hi, lo);
                if (y < 0 || y >= a.H || x < 0 || x >= a.W) { hi = u32x4{0u, 0u, 0u, 0u}; lo = hi; }
                if (i < AROWS * 8) {
                    *(u32x4*)(imgA + row * W64_PITCH + v * 16) = hi;
                    *(u32x4*)(imgA + PLA + row * W64_PITCH + v * 16) = lo;
                }
            }
            }
            if (!(a.dbg & 4)) issue_loads(it + 1 < ntile ? tile + a.S : tile);      // in flight across the barrier and the next split
            wp_barrier();                                   // barrier `it`: buffer it % 2 is complete
        }
        wp_barrier();                                       // the consumers have read the last buffer: the LDS is free
        if (a.part_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[ptid * 8 + j] = bias8[j];
        }
    } else {
        // =========================== consumers ===========================
        const int cih = wid & 1, coh = wid >> 1;
        f32x16 acc[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const int h = lane >> 5, g = (lane >> 4) & 1, i16 = lane & 15;
        const int lane_off = (8 * h + (i16 >> 2)) * W64_PITCH + (g * 16 + (i16 & 3) * 4) * 2;
        auto frag = [&](const char* p) -> tb_f16x8 {
            const tb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
            const tb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * W64_PITCH));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(tb_f16x8, vv);
        };
        for (int it = 0; it < ntile; ++it) {
            wp_barrier();                                   // barrier `it`: buffer it % 2 is complete
            const char* imgA = smem + (it & 1) * BUF;
            const char* baseY = imgA + 2 * PLA + lane_off + coh * 64;
            const char* baseA = imgA + lane_off + cih * 64;
            if (a.dbg & 1) continue;
#pragma unroll
            for (int y = 0; y < WP_TH; ++y) {
#pragma unroll
                for (int xb = 0; xb < TW; xb += 16) {
                    const tb_f16x8 fyh = frag(baseY + (y * TW + xb) * W64_PITCH), fyl = frag(baseY + PLY + (y * TW + xb) * W64_PITCH);
                    // (one fragment read pair per tap: sharing the reads between the three taps of a kernel row -- tr_load3 / tr_window, what
                    // wgrad64_x3_kernel does -- measured SLOWER here, 131 -> 137 us on the largest layer: the consumers are not bound by the
                    // number of LDS reads, and the funnel shifts sit in front of their MFMAs)
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const int ky = TAPS == 9 ? t / 3 : 0, kx = TAPS == 9 ? t % 3 : 0;
                        const char* pa = baseA + ((y + ky) * AW + xb + kx) * W64_PITCH;
                        const tb_f16x8 fxh = frag(pa), fxl = frag(pa + PLA);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyl, fxh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh, fxl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh, fxh, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        wp_barrier();                                       // (pairs with the producers' final barrier)
        const int col = lane & 31;
        float* out = a.part + ((size_t)share * npairs + pair) * TAPS * 4096;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(size_t)t * 4096 + (coh * 32 + mfma32_row(r, h)) * 64 + cih * 32 + col] = acc[t][r] * yinv;
    }
    if (a.part_bias) {          // uniform.  dbias partial of this share: the producers' per-thread sums, added in thread order
        wp_barrier();
        if (pair % nbc == 0 && tid < 64) {
            float sum = 0.f;
            const int vv = tid >> 3, j = tid & 7;              // channel tid = slot vv, element j
            for (int p = 0; p < 32; ++p) sum += red[(p * 8 + vv) * 8 + j];
            a.part_bias[(size_t)share * a.Cout + co0 + tid] = sum * yinv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 x 3 weight gradient of the 1 x 1 layers with 128 x 128-channel workgroup tiles (round 5, third version).
// A 1 x 1 layer has one product per staged element and tap where the 3 x 3 layers have nine: on 64 x 64 tiles wgrad64_x3_pc_kernel
// spends its time in the PRODUCERS (85 us alone on 256 -> 256 at 128^2; 54 without the split + LDS stores, 80 without the MFMAs:
// profiles/r05_wgrad_x3.txt) and every element of x and dy is loaded, normalised and split by FOUR workgroups (one per tile of the
// other operand's channels).  Here a workgroup owns 128 output x 128 input channels: each element is staged by two workgroups, each
// staged element feeds four times the MFMAs, and each consumer wave holds a 64 x 64 block (2 x 2 accumulators).  Tiles are 32
// consecutive pixels of the flattened (B, H, W) index (a 1 x 1 layer has no neighbourhood; H W % 32 == 0, so a tile lies in one
// image); LDS rows are 320 bytes (256 + 64: the four rows a transposing read touches tile the 64 banks), two buffers of four planes
// = 80 KB.  Roles, hand-over and arithmetic per product as in wgrad64_x3_pc_kernel; the partial sums are [share][pair][128][128].
// ------------------------------------------------------------------------------------------------
constexpr int W128_PITCH = 320, W128_PX = 32;

__global__ __launch_bounds__(512) void wgrad128_x3_pc_kernel(WgradArgs a) {
    f16_saturate_mode();
    constexpr int CT = 128, SLOTS = CT / 8;                     // 8-channel slots per pixel row
    constexpr int NV = W128_PX * SLOTS / 256;                   // (pixel, 8 channels) units per producer thread, tile and operand
    constexpr int PL = W128_PX * W128_PITCH;                    // one plane of a buffer
    constexpr int BUF = 4 * PL;                                 // A hi, A lo, Y hi, Y lo
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const int nbo = a.Cout / CT, nbc = a.Cin / CT, npairs = nbo * nbc;
    int share, pair;
    {
        const int L = blockIdx.x;
        if (a.S % 8 == 0) { const int xcd = L & 7, j = L >> 3; share = (j / npairs) * 8 + xcd; pair = j % npairs; }
        else { share = L / npairs; pair = L % npairs; }
    }
    const int co0 = (pair / nbc) * CT, ci0 = (pair % nbc) * CT;
    const int HW = a.H * a.W;
    const int tiles = (int)(a.npix / W128_PX);
    const int ntile = share < tiles ? (tiles - share + a.S - 1) / a.S : 0;
    float ymul, yinv;
    x3_in_scale(a.dy_amax, ymul, yinv);
    float* red = (float*)smem;                                  // after the loop: [256 producer threads][8] bias partials

    if (producer) {
        const int ptid = tid & 255;
        const bool use_gn = a.st != nullptr;
        const float* X = (const float*)a.x;
        const float* DY = (const float*)a.dy;
        const int v = ptid & (SLOTS - 1), prow = ptid / SLOTS;  // this thread's 8-channel slot; its pixel rows: prow + 16 q
        float sc[8], sh[8], bias8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; bias8[j] = 0.f; }
        int cur_b = -1;
        u32x4 vy[NV][2], vx[NV][2];
        auto issue_loads = [&](int tile) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const size_t pix = (size_t)tile * W128_PX + prow + (256 / SLOTS) * q;
                const u32x4* py = (const u32x4*)(DY + pix * a.ys + co0 + v * 8);
                vy[q][0] = py[0]; vy[q][1] = py[1];
                const u32x4* px = (const u32x4*)(X + pix * a.xs + ci0 + v * 8);
                vx[q][0] = px[0]; vx[q][1] = px[1];
            }
        };
        auto split8 = [](const float (&f)[8], u32x4& hi, u32x4& lo) {
            tb_f16x8 hh, ll;
#pragma unroll
            for (int j = 0; j < 8; ++j) { hh[j] = (_Float16)f[j]; ll[j] = (_Float16)(f[j] - (float)hh[j]); }
            hi = __builtin_bit_cast(u32x4, hh);
            lo = __builtin_bit_cast(u32x4, ll);
        };
        if (ntile > 0 && !(a.dbg & 4)) issue_loads(share);
        for (int it = 0; it < ntile; ++it) {
            const int tile = share + it * a.S;
            const int b = (int)(((size_t)tile * W128_PX) / HW);
            char* imgA = smem + (it & 1) * BUF;
            char* imgY = imgA + 2 * PL;
            if (use_gn && b != cur_b) {
#pragma unroll
                for (int j = 0; j < 8; ++j) gn_scale_shift(a.st, a.B, b, a.Cin, ci0 + v * 8 + j, HW, a.gamma, a.beta, sc[j], sh[j]);
                cur_b = b;
            }
            if (!(a.dbg & 2)) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int row = prow + (256 / SLOTS) * q;
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] = __uint_as_float(vy[q][0][j]) * ymul; f[4 + j] = __uint_as_float(vy[q][1][j]) * ymul; }
                u32x4 hi, lo;
                split8(f, hi, lo);
                if (a.part_bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bias8[j] += f[j];
                }
                *(u32x4*)(imgY + row * W128_PITCH + v * 16) = hi;
                *(u32x4*)(imgY + PL + row * W128_PITCH + v * 16) = lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] = __uint_as_float(vx[q][0][j]); f[4 + j] = __uint_as_float(vx[q][1][j]); }
                if (use_gn) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], sc[j], sh[j]); f[j] = t > 0.f ? t : 0.f; }
                }
                split8(f, hi, lo);
                *(u32x4*)(imgA + row * W128_PITCH + v * 16) = hi;
                *(u32x4*)(imgA + PL + row * W128_PITCH + v * 16) = lo;
            }
            }
            if (!(a.dbg & 4)) issue_loads(it + 1 < ntile ? tile + a.S : tile);      // in flight across the barrier and the next split
            wp_barrier();                                       // barrier `it`: buffer it % 2 is complete
        }
        wp_barrier();                                           // the consumers have read the last buffer: the LDS is free
        if (a.part_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[ptid * 8 + j] = bias8[j];
        }
    } else {
        const int cih = wid & 1, coh = wid >> 1;                // this wave's 64 x 64 block of the tile
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        const int h = lane >> 5, g = (lane >> 4) & 1, i16 = lane & 15;
        const int lane_off = (8 * h + (i16 >> 2)) * W128_PITCH + (g * 16 + (i16 & 3) * 4) * 2;
        auto frag = [&](const char* p) -> tb_f16x8 {
            const tb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
            const tb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * W128_PITCH));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(tb_f16x8, vv);
        };
        for (int it = 0; it < ntile; ++it) {
            wp_barrier();                                       // barrier `it`: buffer it % 2 is complete
            const char* imgA = smem + (it & 1) * BUF;
            const char* baseY = imgA + 2 * PL + lane_off + coh * 128;
            const char* baseA = imgA + lane_off + cih * 128;
            if (a.dbg & 1) continue;
#pragma unroll
            for (int xb = 0; xb < W128_PX; xb += 16) {
                tb_f16x8 fyh[2], fyl[2], fxh[2], fxl[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    fyh[m] = frag(baseY + xb * W128_PITCH + m * 64);
                    fyl[m] = frag(baseY + PL + xb * W128_PITCH + m * 64);
                    fxh[m] = frag(baseA + xb * W128_PITCH + m * 64);
                    fxl[m] = frag(baseA + PL + xb * W128_PITCH + m * 64);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {           // small terms first, all three into the same accumulator
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyl[m], fxh[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh[m], fxl[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fyh[m], fxh[n], acc[m][n], 0, 0, 0);
                    }
            }
        }
        wp_barrier();                                           // (pairs with the producers' final barrier)
        const int col = lane & 31;
        float* out = a.part + ((size_t)share * npairs + pair) * (CT * CT);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(coh * 64 + m * 32 + mfma32_row(r, h)) * CT + cih * 64 + n * 32 + col] = acc[m][n][r] * yinv;
    }
    if (a.part_bias) {          // uniform.  dbias partial of this share: the producers' per-thread sums, added in thread order
        wp_barrier();
        if (pair % nbc == 0 && tid < CT) {
            float sum = 0.f;
            const int vv = tid >> 3, j = tid & 7;               // channel tid = slot vv, element j
            for (int p = 0; p < 256 / SLOTS; ++p) sum += red[(p * SLOTS + vv) * 8 + j];
            a.part_bias[(size_t)share * a.Cout + co0 + tid] = sum * yinv;
        }
    }
}

static bool wgrad128_ok(int taps, int B, int H, int W, int Cin, int Cout) {
    static const bool off = getenv("CHORE_WGRAD_NO128") != nullptr;
    return !off && taps == 1 && Cin % 128 == 0 && Cout % 128 == 0 && ((long)H * W) % W128_PX == 0 && (long)B * H * W >= 8 * W128_PX;
}
static int wgrad128_shares(int B, int H, int W, int Cin, int Cout) {
    const int tiles = (int)((long)B * H * W / W128_PX);
    const int pairs = (Cout / 128) * (Cin / 128);
    static const int wgs = getenv("CHORE_WGRAD128_WGS") ? atoi(getenv("CHORE_WGRAD128_WGS")) : 256;      // workgroups to aim for (A/B)
    int S = ((wgs + pairs - 1) / pairs + 7) / 8 * 8;
    if (S > tiles) S = tiles;
    return S;
}

static int wgrad64_x3_pc_shares(int B, int H, int W, int Cin, int Cout) {
    const int tiles = B * ((W + TW - 1) / TW) * ((H + WP_TH - 1) / WP_TH);
    const int pairs = (Cout / 64) * (Cin / 64);
    int S = ((256 + pairs - 1) / pairs + 7) / 8 * 8;
    if (S > tiles) S = tiles;
    return S;
}

static int wgrad64_x3_shares(int B, int H, int W, int Cin, int Cout) {
    const int tiles = B * ((W + TW - 1) / TW) * ((H + WX_TH - 1) / WX_TH);
    const int pairs = (Cout / 64) * (Cin / 64);
    int S = ((256 + pairs - 1) / pairs + 7) / 8 * 8;
    if (S > tiles) S = tiles;
    return S;
}

static int wgrad64_shares(int B, int H, int W, int Cin, int Cout) {
    const int tiles = B * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    const int pairs = (Cout / 64) * (Cin / 64);
    int S = ((256 + pairs - 1) / pairs + 7) / 8 * 8;
    if (S > tiles) S = tiles;
    return S;
}
static bool wgrad_use64(int dtype, int taps, int Cin, int Cout) {
    return (dtype == CHORE_BF16 || dtype == CHORE_F16X3) && (taps == 9 || taps == 1) && Cin % 64 == 0 && Cout % 64 == 0;
}

// dW (O,C,kh,kw) = sum over the shares, in order
__device__ __forceinline__ void wgrad_finish_body(size_t i, const float* __restrict__ part, const float* __restrict__ part_bias,
                                                  int S, int Cout, int Cin, int taps, float* __restrict__ dw,
                                                  float* __restrict__ dbias, int ct) {
    const size_t n = (size_t)Cout * Cin * taps;
    if (i < n) {
        // thread index follows the PARTIAL layout [o tile][c tile][tap][o % ct][c % ct]: coalesced reads of every share
        const int nbc = Cin / ct, cc = (int)(i % ct), oo = (int)((i / ct) % ct), t = (int)((i / ((size_t)ct * ct)) % taps);
        const int pt = (int)(i / ((size_t)ct * ct * taps)), o = (pt / nbc) * ct + oo, c = (pt % nbc) * ct + cc;
        // the shares in order, eight loads in flight at a time (the partials come from the Infinity Cache / HBM: a load per
        // add leaves the sum latency-bound at 1.5 TB/s)
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= S; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(k + j) * n + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < S; ++k) s += part[(size_t)k * n + i];
        dw[((size_t)o * Cin + c) * taps + t] = s;
    }
    if (dbias && i < (size_t)Cout) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += part_bias[(size_t)k * Cout + i];
        dbias[i] = s;
    }
}
__global__ void wgrad_finish_kernel(const float* __restrict__ part, const float* __restrict__ part_bias, int S, int Cout,
                                    int Cin, int taps, float* __restrict__ dw, float* __restrict__ dbias, int ct) {
    wgrad_finish_body((size_t)blockIdx.x * blockDim.x + threadIdx.x, part, part_bias, S, Cout, Cin, taps, dw, dbias, ct);
}
// up to four of them in one launch (blockIdx.y = job): same sums, same order
__global__ void wgrad_finish_multi_kernel(WgradFinishJobs jobs) {
    const WgradFinishJobs::J& j = jobs.j[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)j.Cout * j.Cin * j.taps) return;       // (Cout <= Cout * Cin * taps: the bias entries are covered)
    wgrad_finish_body(i, j.part, j.part_bias, j.S, j.Cout, j.Cin, j.taps, j.dw, j.dbias, j.ct);
}

// ------------------------------------------------------------------------------------------------
// stem weight gradient: dW[o][c][ky][kx] = sum_{b,oy,ox} dy[b,oy,ox,o] * img[b,c,2oy+ky-3,2ox+kx-3], dbias = sum dy
// (conv 7x7 stride 2 pad 3, Cin <= 8 -> 64; the images need no gradient).  K = Cin*49 (245) rows x 64 columns contracted
// over B*OH*OW pixels: 0.8 % of the step's FLOPs and no MFMA shape (a gathered operand), so plain FMAs from LDS.
// A workgroup walks a fixed share of the 8x8-pixel tiles; thread = 4 output channels x 16 rows k = kg + 16 j
// (64 accumulators); the row k = Cin*49 is a row of ones (its sum is dbias).  Partials are summed in order.
// ------------------------------------------------------------------------------------------------
constexpr int SW_T = 8, SW_P = 2 * SW_T + 5, SW_MAXC = 5, SW_ROWS = 256;
constexpr int SW_FILL = (2 * (SW_T - 1)) * SW_P + 2 * (SW_T - 1) + 1;   // pixel offsets reach this far: constant regions of ones / zeros

template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ dy, int B, int Cin,
                                                         int H, int W, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float patch[SW_MAXC * SW_P * SW_P + 2 * SW_FILL];
    __shared__ __attribute__((aligned(16))) float dyl[64 * 64];
    const int OH = H / 2, OW = W / 2;
    const int tx_n = (OW + SW_T - 1) / SW_T, ty_n = (OH + SW_T - 1) / SW_T, ntile = B * ty_n * tx_n;
    const int tid = threadIdx.x, oq = tid & 15, kg = tid >> 4;
    const int K = Cin * 49, ONE = Cin * SW_P * SW_P, ZERO = ONE + SW_FILL;
    int off[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = kg + 16 * j;
        const int c = k / 49, r = k % 49;
        off[j] = k < K ? (c * SW_P + r / 7) * SW_P + r % 7 : (k == K ? ONE : ZERO);
    }
    for (int i = tid; i < 2 * SW_FILL; i += 256) patch[ONE + i] = i < SW_FILL ? 1.f : 0.f;
    float acc[16][4];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int b = t / (ty_n * tx_n), ty0 = (t / tx_n) % ty_n * SW_T, tx0 = t % tx_n * SW_T;
        __syncthreads();
        const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
        for (int i = tid; i < Cin * SW_P * SW_P; i += 256) {
            const int c = i / (SW_P * SW_P), r = i % (SW_P * SW_P);
            const int y = iy0 + r / SW_P, x = ix0 + r % SW_P;
            float v = 0.f;
            if (y >= 0 && y < H && x >= 0 && x < W) v = img[(((size_t)b * Cin + c) * H + y) * W + x];
            patch[i] = v;
        }
        for (int i = tid; i < 64 * 16; i += 256) {
            const int pix = i >> 4, q = i & 15;
            const int oy = ty0 + (pix >> 3), ox = tx0 + (pix & 7);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (oy < OH && ox < OW) v = Vec4<T>::ld(dy + (((size_t)b * OH + oy) * OW + ox) * 64 + q * 4);
            *(f32x4*)(dyl + pix * 64 + q * 4) = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int pix = 0; pix < 64; ++pix) {
            const f32x4 d = *(const f32x4*)(dyl + pix * 64 + oq * 4);
            const int po = (2 * (pix >> 3)) * SW_P + 2 * (pix & 7);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = patch[off[j] + po];
                acc[j][0] = fmaf(v, d[0], acc[j][0]);
                acc[j][1] = fmaf(v, d[1], acc[j][1]);
                acc[j][2] = fmaf(v, d[2], acc[j][2]);
                acc[j][3] = fmaf(v, d[3], acc[j][3]);
            }
        }
    }
    float* o = part + (size_t)blockIdx.x * SW_ROWS * 64;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        f32x4 v = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        *(f32x4*)(o + (kg + 16 * j) * 64 + oq * 4) = v;
    }
}

// dw (64,Cin,7,7) and dbias (64) = ordered sums of the partials: 4 lanes per entry (shares r, r+4, ...), fixed tree
__global__ void stem_wgrad_finish_kernel(const float* __restrict__ part, int S, int Cin, float* __restrict__ dw,
                                         float* __restrict__ dbias) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, r = threadIdx.x & 3;
    const int K = Cin * 49, n = (K + 1) * 64;
    const int e = i < n ? i : n - 1;
    float s = 0.f;
    for (int k = r; k < S; k += 4) s += part[(size_t)k * SW_ROWS * 64 + e];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (r || i >= n) return;
    const int k = e >> 6, o = e & 63;
    if (k < K) dw[(size_t)o * K + k] = s;
    else if (dbias) dbias[o] = s;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm + ReLU backward
// ------------------------------------------------------------------------------------------------
// accumulator cells of a GroupNorm backward ((B*32 + C) GroupStat = two cells each), rounded up to 4 KB: where the table
// of the high limbs starts
static inline size_t gn_bwd_hi_cells(int B, int C) { return (((size_t)B * GN_GROUPS + C) * 2 + 255) / 256 * 256; }
struct GnBwdAcc {                 // exact accumulators, zeroed by the caller
    GroupStat* grp;               // [B][32]: sum = S1 = sum(g*gamma), sq = S2 = sum(g*gamma*xhat)
    GroupStat* chan;              // [C]:     sum = dbeta, sq = dgamma
    size_t hc;                    // cells between a low limb and its high limb (the second table, enc_common.h)
};

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ da,
                                                            const GroupStat* __restrict__ st, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int C, int HW, int S, GnBwdAcc acc) {
    __shared__ float red[4][1024];
    __shared__ float mr[1024];         // [C][4] mean, rstd, scale, shift per channel of this image (scale/shift exactly
                                       // as the forward folds them, so that the ReLU mask is the forward's)
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int gs = C / GN_GROUPS;
    if (tid < C) {
        const GroupStat* gst = st + (size_t)b * GN_GROUPS + tid / gs;
        const double ssum = stat_read(&gst->sum, act_hi_cells((int)gridDim.y)), ssq = stat_read(&gst->sq, act_hi_cells((int)gridDim.y));
        const double n = (double)HW * gs;
        const double mean = ssum / n;
        double var = ssq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
        const float scale = rstd * gamma[tid];
        mr[4 * tid] = (float)mean;
        mr[4 * tid + 1] = rstd;
        mr[4 * tid + 2] = scale;
        mr[4 * tid + 3] = beta[tid] - (float)mean * scale;
    }
    __syncthreads();
    const int tpr = C / 4, P = 256 / tpr, cv = tid % tpr, pl = tid / tpr;
    const int p0 = (int)((long long)HW * s / S), p1 = (int)((long long)HW * (s + 1) / S);
    float dgam[4] = {0, 0, 0, 0}, dbet[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    constexpr int GNB_U = 8;
    // GNB_U pixels per round, their loads issued together and unconditionally (rows past the share are re-reads of its last
    // row with a zero weight): one load pair per round left the pass latency-bound at 1.4 TB/s
    float cm[4], cr[4], csc[4], csh[4], cg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cv * 4 + j;
        cm[j] = mr[4 * c]; cr[j] = mr[4 * c + 1]; csc[j] = mr[4 * c + 2]; csh[j] = mr[4 * c + 3]; cg[j] = gamma[c];
    }
    for (int p = p0 + pl; p < p1; p += GNB_U * P) {
        f32x4 xq[GNB_U], dq[GNB_U];
        float live[GNB_U];
#pragma unroll
        for (int u = 0; u < GNB_U; ++u) {
            const int pp = p + u * P;
            live[u] = pp < p1 ? 1.f : 0.f;
            const size_t o = ((size_t)b * HW + (pp < p1 ? pp : p1 - 1)) * C + cv * 4;
            xq[u] = Vec4<T>::ld(x + o);
            dq[u] = Vec4<T>::ld(da + o);
        }
#pragma unroll
        for (int u = 0; u < GNB_U; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xv = xq[u][j];
                const float xh = (xv - cm[j]) * cr[j];
                const float g = fmaf(xv, csc[j], csh[j]) > 0.f ? dq[u][j] * live[u] : 0.f;
                dbet[j] += g;
                dgam[j] += g * xh;
                const float gy = g * cg[j];
                s1[j] += gy;
                s2[j] += gy * xh;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][pl * C + cv * 4 + j] = dbet[j];
        red[1][pl * C + cv * 4 + j] = dgam[j];
        red[2][pl * C + cv * 4 + j] = s1[j];
        red[3][pl * C + cv * 4 + j] = s2[j];
    }
    __syncthreads();
    if (tid < C) {
        float t[4] = {0, 0, 0, 0};
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += red[k][i * C + tid];
        stat_add(&acc.chan[tid].sum, acc.hc, t[0]);
        stat_add(&acc.chan[tid].sq, acc.hc, t[1]);
        const float g1 = group_lane_sum(t[2], gs), g2 = group_lane_sum(t[3], gs);
        GroupStat* o = acc.grp + (size_t)b * GN_GROUPS + tid / gs;      // two lanes of the group, concurrently
        if (tid % gs == 0) stat_add(&o->sum, acc.hc, g1);
        if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, acc.hc, g2);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ da,
                                                           const GroupStat* __restrict__ st, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int C, int HW, GnBwdAcc acc,
                                                           T* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, const T* __restrict__ extra, int ecs,
                                                           unsigned* __restrict__ amax_out) {
    __shared__ float pc[1792];         // [C][7] mean, rstd, gamma, S1/n, S2/n, scale, shift
    __shared__ unsigned amax_red[4];
    unsigned amax = 0u;                // max |dx| of this thread (fp16 x 3 training: the next GEMMs' operand scale)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int gs = C / GN_GROUPS;
    if (tid < C) {
        const int gi = tid / gs;
        const GroupStat* gst = st + (size_t)b * GN_GROUPS + gi;
        const double ssum = stat_read(&gst->sum, act_hi_cells((int)gridDim.y)), ssq = stat_read(&gst->sq, act_hi_cells((int)gridDim.y));
        const double n = (double)HW * gs;
        const double mean = ssum / n;
        double var = ssq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const GroupStat* ga = acc.grp + (size_t)b * GN_GROUPS + gi;
        const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
        const float scale = rstd * gamma[tid];
        pc[7 * tid] = (float)mean;
        pc[7 * tid + 1] = rstd;
        pc[7 * tid + 2] = gamma[tid];
        pc[7 * tid + 3] = (float)(stat_read(&ga->sum, acc.hc) / n);
        pc[7 * tid + 4] = (float)(stat_read(&ga->sq, acc.hc) / n);
        pc[7 * tid + 5] = scale;
        pc[7 * tid + 6] = beta[tid] - (float)mean * scale;
        if (blockIdx.x == 0 && b == 0) {     // the parameter gradients: totals over the whole batch
            dbeta[tid] = (float)stat_read(&acc.chan[tid].sum, acc.hc);
            dgamma[tid] = (float)stat_read(&acc.chan[tid].sq, acc.hc);
        }
    }
    __syncthreads();
    const size_t total4 = (size_t)HW * C / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total4; i += (size_t)gridDim.x * 256) {
        const int c0 = (int)((i * 4) % C);
        const size_t o = (size_t)b * total4 * 4 + i * 4;
        const f32x4 xv4 = Vec4<T>::ld(x + o), da4 = Vec4<T>::ld(da + o);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* q = pc + 7 * (c0 + j);
            const float xv = xv4[j];
            const float xh = (xv - q[0]) * q[1];
            const float gy = fmaf(xv, q[5], q[6]) > 0.f ? da4[j] * q[2] : 0.f;
            r[j] = q[1] * ((gy - q[3]) - xh * q[4]);
        }
        if (extra) {      // a second gradient of x (a skip connection / another consumer), channel-strided: summed here
            const size_t pix = (size_t)b * HW + (i * 4) / C;
            const f32x4 e4 = Vec4<T>::ld(extra + pix * ecs + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] += e4[j];
        }
        Vec4<T>::st(dx + o, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const unsigned bts = __float_as_uint(r[j]) & 0x7fffffffu; amax = bts > amax ? bts : amax; }
    }
    if (amax_out)      // uniform.  One atomic max per workgroup into one of AMAX_CELLS zeroed cells
        amax_block_atomic(amax, amax_out, (int)((blockIdx.y * gridDim.x + blockIdx.x) % AMAX_CELLS), amax_red);
}

}  // namespace

extern "C" {

// shares of the pixel tiles: enough workgroups to fill the chip, at least 4 tiles per share
static int wgrad_shares(int B, int H, int W, int Cin, int Cout) {
    const int tiles = B * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    const int pairs = (Cout / 32) * (Cin / 32);
    int S = (1024 + pairs - 1) / pairs;
    if (S > tiles / 2) S = tiles / 2;
    if (S < 1) S = 1;
    return S;
}

size_t chore_conv2d_wgrad_workspace_bytes(int taps, int B, int H, int W, int Cin, int Cout) {
    if ((taps != 1 && taps != 9) || Cin % 32 || Cout % 32 || B <= 0) return 0;
    const int S = wgrad_shares(B, H, W, Cin, Cout);
    size_t n = (size_t)S * (Cout / 32) * (Cin / 32) * taps * 1024 + (size_t)S * Cout;
    if (wgrad_use64(CHORE_BF16, taps, Cin, Cout)) {       // the bf16 path of these shapes uses 64-channel tiles
        const int Sb = wgrad64_shares(B, H, W, Cin, Cout), Sx = wgrad64_x3_shares(B, H, W, Cin, Cout), Sp = wgrad64_x3_pc_shares(B, H, W, Cin, Cout);
        const int S64 = (Sb > Sx ? Sb : Sx) > Sp ? (Sb > Sx ? Sb : Sx) : Sp;
        const size_t n64 = (size_t)S64 * (Cout / 64) * (Cin / 64) * taps * 4096 + (size_t)S64 * Cout;
        if (n64 > n) n = n64;
    }
    if (wgrad128_ok(taps, B, H, W, Cin, Cout)) {
        const int S128 = wgrad128_shares(B, H, W, Cin, Cout);
        const size_t n128 = (size_t)S128 * (Cout / 128) * (Cin / 128) * 16384 + (size_t)S128 * Cout;
        if (n128 > n) n = n128;
    }
    return n * sizeof(float);
}

}  // extern "C"

// extra (or null): (B,HW,*) with channel stride extra_cs, already offset to its first channel -- added to dx
int gn_relu_bwd_impl(chore_handle* h, int dtype, const void* x, const void* stats, const float* gamma, const float* beta,
                     const void* da, int B, int HW, int C, void* dx, float* dgamma, float* dbeta, void* workspace,
                     int workspace_zeroed, const void* extra, int extra_cs, hipStream_t s, unsigned* amax_out) {
    if (dtype == CHORE_F16X3) dtype = CHORE_F32;        // fp32 tensors; only the convolutions differ
    if (!x || !stats || !gamma || !beta || !da || !dx || !dgamma || !dbeta || !workspace)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_gn_relu_bwd: null argument");
    if (C % GN_GROUPS || C > 256 || C < 32 || 256 % (C / 4)) CHORE_FAIL(h, CHORE_EINVAL, "chore_gn_relu_bwd: unsupported C=%d", C);
    if (!workspace_zeroed) CHORE_HIP_CHECK(h, hipMemsetAsync(workspace, 0, chore_gn_relu_bwd_workspace_bytes(B, C), s));
    GnBwdAcc acc;
    acc.grp = (GroupStat*)workspace;
    acc.chan = acc.grp + (size_t)B * GN_GROUPS;
    acc.hc = gn_bwd_hi_cells(B, C);
    int S = HW / 128;          // pixels per share: 128 measured a little faster than 64 / 256 on the training step
    if (S < 1) S = 1;
    if (S > GN_SPLITS_MAX) S = GN_SPLITS_MAX;
    const size_t total4 = (size_t)HW * C / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 512) blocks = 512;
    if (dtype == CHORE_F32) {
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, dim3(S, B), dim3(256), 0, s, (const float*)x, (const float*)da,
                           (const GroupStat*)stats, gamma, beta, C, HW, S, acc);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3(blocks, B), dim3(256), 0, s, (const float*)x, (const float*)da,
                           (const GroupStat*)stats, gamma, beta, C, HW, acc, (float*)dx, dgamma, dbeta, (const float*)extra,
                           extra_cs, amax_out);
    } else {
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, dim3(S, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)da,
                           (const GroupStat*)stats, gamma, beta, C, HW, S, acc);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, dim3(blocks, B), dim3(256), 0, s, (const bf16_t*)x,
                           (const bf16_t*)da, (const GroupStat*)stats, gamma, beta, C, HW, acc, (bf16_t*)dx, dgamma, dbeta,
                           (const bf16_t*)extra, extra_cs, amax_out);
    }
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// the weight gradient with a channel-strided dy (dy_stride = channels of the tensor dy is a slice of; dy already offset)
int conv2d_bwd_weight_impl(chore_handle* h, int dtype, int taps, const void* x, int B, int H, int W, int Cin,
                           const void* stats, const float* gamma, const float* beta, const void* dy, int dy_stride, int Cout,
                           float* dw, float* dbias, void* workspace, hipStream_t s, WgradFinishJobs* defer, const unsigned* dy_amax) {
    if (!x || !dy || !dw || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_bwd_weight: null argument");
    if (dtype == CHORE_F16X3 && !dy_amax) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_bwd_weight: the fp16 x 3 mode needs the range of dy (dy_amax)");
    if (defer && defer->n >= 4) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_bwd_weight: more than four deferred sums");
    if ((taps != 1 && taps != 9) || Cin % 32 || Cout % 32 || Cin > 256)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_bwd_weight: unsupported taps=%d Cin=%d Cout=%d", taps, Cin, Cout);
    if (dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3) CHORE_FAIL(h, CHORE_EINVAL, "chore_conv2d_bwd_weight: bad dtype");
    WgradArgs a;
    a.dy_amax = dy_amax;
    static const int wdbg = getenv("CHORE_WGRAD_DBG") ? atoi(getenv("CHORE_WGRAD_DBG")) : 0;
    a.dbg = wdbg;
    a.x = x; a.st = (const GroupStat*)stats; a.gamma = gamma; a.beta = beta; a.dy = dy;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.xs = Cin; a.ys = dy_stride; a.npix = (long long)B * H * W;
    a.part = (float*)workspace;
    int ct = 32;
    if (wgrad_use64(dtype, taps, Cin, Cout)) {
        ct = 64;
        const bool x3 = dtype == CHORE_F16X3;
        a.S = x3 ? wgrad64_x3_shares(B, H, W, Cin, Cout) : wgrad64_shares(B, H, W, Cin, Cout);
        const int npairs = (Cout / 64) * (Cin / 64);
        a.part_bias = dbias ? a.part + (size_t)a.S * npairs * taps * 4096 : nullptr;
        // which fp16 x 3 kernel: the specialised-wave one where it measured faster alone (profiles/r05_wgrad_x3.txt: the 1x1 layers
        // 101 -> 85 us, 3x3 256->128 at 128^2 150 -> 131, 128->128 90 -> 82; the 64-channel layers and the 32^2 maps are 2 - 8 us
        // SLOWER on it: twice the tiles, twice the barriers).  CHORE_WGRAD_X3_V1=1 / CHORE_WGRAD_X3_PC=1 force one of them.
        static const bool x3_v1 = getenv("CHORE_WGRAD_X3_V1") != nullptr, x3_pc = getenv("CHORE_WGRAD_X3_PC") != nullptr;
        const bool use_pc = x3_pc || (!x3_v1 && (taps == 1 || (Cin >= 128 && Cout >= 128 && (long)H * W >= 128 * 128)));
        if (x3 && !x3_v1 && !x3_pc && wgrad128_ok(taps, B, H, W, Cin, Cout)) {      // the 1 x 1 layers: 128 x 128-channel tiles
            ct = 128;
            a.S = wgrad128_shares(B, H, W, Cin, Cout);
            const int np128 = (Cout / 128) * (Cin / 128);
            a.part_bias = dbias ? a.part + (size_t)a.S * np128 * 16384 : nullptr;
            const size_t sm128 = (size_t)2 * 4 * W128_PX * W128_PITCH;
            bool& attr128 = CHORE_ONCE_FLAG(h);
            if (!attr128) {
                CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad128_x3_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm128));
                attr128 = true;
            }
            hipLaunchKernelGGL(wgrad128_x3_pc_kernel, dim3(a.S * np128), dim3(512), sm128, s, a);
            CHORE_LAUNCH_CHECK(h, s);
            if (defer) { defer->j[defer->n++] = {a.part, a.part_bias, dw, dbias, a.S, Cout, Cin, taps, ct}; return CHORE_OK; }
            const size_t n = (size_t)Cout * Cin * taps;
            hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, a.part_bias, a.S, Cout,
                               Cin, taps, dw, dbias, ct);
            CHORE_LAUNCH_CHECK(h, s);
            return CHORE_OK;
        }
        if (x3 && use_pc) {
            a.S = wgrad64_x3_pc_shares(B, H, W, Cin, Cout);
            a.part_bias = dbias ? a.part + (size_t)a.S * npairs * taps * 4096 : nullptr;
            const size_t arows = taps == 9 ? (size_t)(WP_TH + 2) * PW : (size_t)WP_TH * TW;
            const size_t smp = (size_t)2 * 2 * (arows + WP_TH * TW) * W64_PITCH;
            bool& attrp = CHORE_ONCE_FLAG(h);
            if (!attrp) {
                CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad64_x3_pc_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)((size_t)4 * ((WP_TH + 2) * PW + WP_TH * TW) * W64_PITCH)));
                CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad64_x3_pc_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)((size_t)4 * (2 * WP_TH * TW) * W64_PITCH)));
                attrp = true;
            }
            if (taps == 9) hipLaunchKernelGGL(wgrad64_x3_pc_kernel<9>, dim3(a.S * npairs), dim3(512), smp, s, a);
            else hipLaunchKernelGGL(wgrad64_x3_pc_kernel<1>, dim3(a.S * npairs), dim3(512), smp, s, a);
            CHORE_LAUNCH_CHECK(h, s);
            if (defer) { defer->j[defer->n++] = {a.part, a.part_bias, dw, dbias, a.S, Cout, Cin, taps, ct}; return CHORE_OK; }
            const size_t n = (size_t)Cout * Cin * taps;
            hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, a.part_bias, a.S, Cout,
                               Cin, taps, dw, dbias, ct);
            CHORE_LAUNCH_CHECK(h, s);
            return CHORE_OK;
        }
        if (x3) {
            const size_t smx = (size_t)2 * ((taps == 9 ? (WX_TH + 2) * PW : WX_TH * TW) + WX_TH * TW) * W64_PITCH + 64 * 2 * sizeof(float);
            bool& attrx = CHORE_ONCE_FLAG(h);
            if (!attrx) {
                CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad64_x3_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)((size_t)2 * ((WX_TH + 2) * PW + WX_TH * TW) * W64_PITCH + 512)));
                CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad64_x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)((size_t)2 * (2 * WX_TH * TW) * W64_PITCH + 512)));
                attrx = true;
            }
            if (taps == 9) hipLaunchKernelGGL(wgrad64_x3_kernel<9>, dim3(a.S * npairs), dim3(256), smx, s, a);
            else hipLaunchKernelGGL(wgrad64_x3_kernel<1>, dim3(a.S * npairs), dim3(256), smx, s, a);
            CHORE_LAUNCH_CHECK(h, s);
            if (defer) { defer->j[defer->n++] = {a.part, a.part_bias, dw, dbias, a.S, Cout, Cin, taps, ct}; return CHORE_OK; }
            const size_t n = (size_t)Cout * Cin * taps;
            hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, a.part_bias, a.S, Cout,
                               Cin, taps, dw, dbias, ct);
            CHORE_LAUNCH_CHECK(h, s);
            return CHORE_OK;
        }
        const size_t smem64 = (size_t)((taps == 9 ? PH * PW : TH * TW) + TH * TW) * W64_PITCH + 64 * 2 * sizeof(float);
        bool& attr64 = CHORE_ONCE_FLAG(h);
        if (!attr64) {
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad64_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)((size_t)(PH * PW + TH * TW) * W64_PITCH + 512)));
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)((size_t)(2 * TH * TW) * W64_PITCH + 512)));
            attr64 = true;
        }
        if (taps == 9) hipLaunchKernelGGL(wgrad64_kernel<9>, dim3(a.S * npairs), dim3(256), smem64, s, a);
        else hipLaunchKernelGGL(wgrad64_kernel<1>, dim3(a.S * npairs), dim3(256), smem64, s, a);
        CHORE_LAUNCH_CHECK(h, s);
        if (defer) { defer->j[defer->n++] = {a.part, a.part_bias, dw, dbias, a.S, Cout, Cin, taps, ct}; return CHORE_OK; }
        const size_t n = (size_t)Cout * Cin * taps;
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, a.part_bias, a.S, Cout,
                           Cin, taps, dw, dbias, ct);
        CHORE_LAUNCH_CHECK(h, s);
        return CHORE_OK;
    }
    if (dtype == CHORE_F16X3) dtype = CHORE_F32;      // channel counts below 64 (the 256^2 block): the exact fp32 matrix-core kernel
    a.S = wgrad_shares(B, H, W, Cin, Cout);
    a.part_bias = dbias ? a.part + (size_t)a.S * (Cout / 32) * (Cin / 32) * taps * 1024 : nullptr;
    const size_t es = dtype == CHORE_F32 ? 4 : 2;
    const int arows = taps == 9 ? PH * PW : TH * TW;
    size_t smem = (size_t)(arows + TH * TW) * CT32 * es + 256;
    if (smem < 4 * 1024 * sizeof(float)) smem = 4 * 1024 * sizeof(float);
    dim3 grid(Cout / 32, Cin / 32, a.S);
#define LAUNCH_WG(T, TP)                                                                                              \
    do {                                                                                                              \
        bool& attr = CHORE_ONCE_FLAG(h);                                                                                     \
        if (!attr) {                                                                                                  \
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad_kernel<T, TP>,                                 \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));          \
            attr = true;                                                                                              \
        }                                                                                                             \
        hipLaunchKernelGGL((wgrad_kernel<T, TP>), grid, dim3(256), smem, s, a);                                      \
    } while (0)
    if (dtype == CHORE_F32) { if (taps == 9) LAUNCH_WG(float, 9); else LAUNCH_WG(float, 1); }
    else { if (taps == 9) LAUNCH_WG(bf16_t, 9); else LAUNCH_WG(bf16_t, 1); }
#undef LAUNCH_WG
    if (defer) { CHORE_LAUNCH_CHECK(h, s); defer->j[defer->n++] = {a.part, a.part_bias, dw, dbias, a.S, Cout, Cin, taps, ct}; return CHORE_OK; }
    const size_t n = (size_t)Cout * Cin * taps;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, a.part_bias, a.S, Cout,
                       Cin, taps, dw, dbias, ct);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int launch_wgrad_finish_multi(chore_handle* h, const WgradFinishJobs& jobs, hipStream_t s) {
    if (jobs.n <= 0) return CHORE_OK;
    size_t nmax = 0;
    for (int k = 0; k < jobs.n; ++k) {
        const size_t n = (size_t)jobs.j[k].Cout * jobs.j[k].Cin * jobs.j[k].taps;
        nmax = n > nmax ? n : nmax;
    }
    hipLaunchKernelGGL(wgrad_finish_multi_kernel, dim3((unsigned)((nmax + 255) / 256), jobs.n), dim3(256), 0, s, jobs);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

extern "C" {

// dw (Cout,Cin,k,k) fp32 and dbias (Cout, or NULL) of y = conv(a) + bias, a = relu(groupnorm(x)) if stats else x
int chore_conv2d_bwd_weight(chore_handle* h, int dtype, int taps, const void* x, int B, int H, int W, int Cin,
                            const void* stats, const float* gamma, const float* beta, const void* dy, int Cout, float* dw,
                            float* dbias, void* workspace, const void* dy_amax, chore_stream_t stream) {
    CHORE_ENTER(h);
    return conv2d_bwd_weight_impl(h, dtype, taps, x, B, H, W, Cin, stats, gamma, beta, dy, Cout, Cout, dw, dbias, workspace,
                                  (hipStream_t)stream, nullptr, (const unsigned*)dy_amax);
}

// C (M x N, fp32, row-major) = A^T B for row-major A (P x M, row stride lda) and B (P x N, row stride ldb), fp32, exact
// matrix-core arithmetic: the contraction runs over the ROWS, like the weight gradient (it is the taps = 1 case with
// the rows as "pixels").  M, N multiples of 32; the P rows are walked as a (ceil(P/32) x 32) image.
// Used for the weight gradients of the MLP heads (dW_l = dZ_l^T H_{l-1}, 80 000 rows): the BLAS library's skinny
// GEMM took 14 ms per product there.
size_t chore_gemm_tn_workspace_bytes(int P, int M, int N) {
    if (P <= 0 || M % 32 || N % 32) return 0;
    const int S = wgrad_shares(1, (P + 31) / 32, 32, N, M);
    return (size_t)S * (M / 32) * (N / 32) * 1024 * sizeof(float);
}

int chore_gemm_tn_f32(chore_handle* h, const float* A, int lda, const float* B, int ldb, int P, int M, int N, float* C,
                      void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!A || !B || !C || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_gemm_tn_f32: null argument");
    if (P <= 0 || M % 32 || N % 32 || lda % 4 || ldb % 4)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_gemm_tn_f32: M, N must be multiples of 32 (P=%d M=%d N=%d)", P, M, N);
    hipStream_t s = (hipStream_t)stream;
    WgradArgs a;
    a.dy_amax = nullptr; a.dbg = 0;
    a.x = B; a.st = nullptr; a.gamma = nullptr; a.beta = nullptr; a.dy = A;
    a.B = 1; a.H = (P + 31) / 32; a.W = 32; a.Cin = N; a.Cout = M; a.xs = ldb; a.ys = lda; a.npix = P;
    a.S = wgrad_shares(1, a.H, a.W, N, M);
    a.part = (float*)workspace;
    a.part_bias = nullptr;
    size_t smem = (size_t)(2 * TH * TW) * CT32 * 4 + 256;
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)wgrad_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               96 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<float, 1>), dim3(M / 32, N / 32, a.S), dim3(256), smem, s, a);
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, nullptr, a.S, M, N, 1, C,
                       nullptr, 32);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

static int stem_wgrad_shares(int B, int H, int W) {
    const int nt = B * ((H / 2 + SW_T - 1) / SW_T) * ((W / 2 + SW_T - 1) / SW_T);
    return nt < 512 ? nt : 512;
}

size_t chore_stem_wgrad_workspace_bytes(int B, int Cin, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cin > SW_MAXC || H <= 0 || W <= 0) return 0;
    return (size_t)stem_wgrad_shares(B, H, W) * SW_ROWS * 64 * sizeof(float);
}

// dw (64,Cin,7,7), dbias (64, or NULL) of the stem convolution y = conv7x7/2(images) + bias; dy is (B,H/2,W/2,64)
int chore_stem_bwd_weight(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W, const void* dy,
                          float* dw, float* dbias, void* workspace, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!images || !dy || !dw || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_stem_bwd_weight: null argument");
    if (B <= 0 || Cin <= 0 || Cin > SW_MAXC || H <= 0 || W <= 0 || (H & 1) || (W & 1))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_stem_bwd_weight: Cin <= %d and even H, W (Cin=%d H=%d W=%d)", SW_MAXC, Cin, H, W);
    if (dtype == CHORE_F16X3) dtype = CHORE_F32;        // fp32 tensors; the stem is plain fp32 arithmetic in every mode
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) CHORE_FAIL(h, CHORE_EINVAL, "chore_stem_bwd_weight: dtype");
    hipStream_t s = (hipStream_t)stream;
    const int S = stem_wgrad_shares(B, H, W);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(S), dim3(256), 0, s, images, (const float*)dy, B, Cin, H, W,
                           (float*)workspace);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(S), dim3(256), 0, s, images, (const bf16_t*)dy, B, Cin, H, W,
                           (float*)workspace);
    CHORE_LAUNCH_CHECK(h, s);
    const int n = (Cin * 49 + 1) * 64 * 4;
    hipLaunchKernelGGL(stem_wgrad_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)workspace, S, Cin, dw, dbias);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

size_t chore_gn_relu_bwd_workspace_bytes(int B, int C) { return (gn_bwd_hi_cells(B, C) + ((size_t)B * GN_GROUPS + C) * 2) * sizeof(StatCell); }

// da = gradient w.r.t. relu(groupnorm(x)) -> dx, dgamma (C), dbeta (C)
int chore_gn_relu_bwd(chore_handle* h, int dtype, const void* x, const void* stats, const float* gamma, const float* beta,
                      const void* da, int B, int HW, int C, void* dx, float* dgamma, float* dbeta, void* workspace,
                      int workspace_zeroed, chore_stream_t stream) {
    CHORE_ENTER(h);
    return gn_relu_bwd_impl(h, dtype, x, stats, gamma, beta, da, B, HW, C, dx, dgamma, dbeta, workspace, workspace_zeroed,
                            nullptr, 0, (hipStream_t)stream, nullptr);
}

}  // extern "C"

// enc_misc.hip -- the bandwidth-bound kernels of the encoder (gfx950): 7x7 stem convolution,
// GroupNorm statistics / apply, 2x2 average pooling, bicubic x2 upsample + add.
//
// Reference semantics (ATen ops called from model/HGFilters.py:26-50,144-185 and
// model/net_util.py:374-396): group_norm(32 groups, eps 1e-5, biased variance), avg_pool2d(2,2),
// interpolate(scale 2, bicubic, align_corners=True) = cubic convolution A=-0.75 with
// border-clamped taps and source coordinate dst*(in-1)/(out-1).
// All tensors NHWC; every thread moves 4 channels (16 B fp32 / 8 B bf16) so accesses coalesce.
#include "enc_common.h"
#include <type_traits>
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// stem: conv 7x7 stride 2 pad 3, Cin(5) -> 64, + bias      (model/HGFilters.py:102,149)
// block = 16x16 output pixels x 4 groups of 16 channels (one wave each); the input patch staged in LDS.  A thread
// owns 4 horizontally adjacent pixels x 16 channels: per kernel row it reads the 13 input values the four pixels share once,
// and every weight vector it fetches feeds 4 pixels -- 0.09 LDS instructions per FMA (one pixel per thread: 0.31, and the
// 63 KB of weights were staged once per 64 pixels).  The accumulation order (c, ky, kx) is that of the one-pixel kernel:
// results are bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int STEM_T = 16;
constexpr int STEM_P = 2 * STEM_T + 5;  // 37
constexpr int STEM_MAXC = 8;

template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, int B, int Cin, int H, int W,
                                                   const float* __restrict__ wk, const float* __restrict__ bias,
                                                   T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* patch = sm;                    // [Cin][37][37]
    const int OH = H / 2, OW = W / 2;
    const int b = blockIdx.z, ty0 = blockIdx.y * STEM_T, tx0 = blockIdx.x * STEM_T;
    const int tid = threadIdx.x;
    const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
    for (int i = tid; i < Cin * STEM_P * STEM_P; i += 256) {
        const int c = i / (STEM_P * STEM_P), r = i % (STEM_P * STEM_P);
        const int y = iy0 + r / STEM_P, x = ix0 + r % STEM_P;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = img[(((size_t)b * Cin + c) * H + y) * W + x];
        patch[i] = v;
    }
    __syncthreads();
    // wave = channel group: its 16 weights of a tap are the same for all lanes -> scalar loads straight from the packed
    // array (the scalar cache / L2), no LDS copy: the patch alone (27 KB) lets five workgroups share a CU, so one's staging
    // overlaps the others' arithmetic (with the 63 KB weight copy one workgroup per CU staged, then computed: 188 us)
    const int lane = tid & 63, cg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = lane >> 2, px = (lane & 3) * 4;   // this thread's row and first column of the tile
    typedef float f32x2 __attribute__((ext_vector_type(2)));      // pairs of channels: v_pk_fma_f32 (two FMAs per lane and cycle)
    f32x2 acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[p][j] = f32x2{bias[cg * 16 + 2 * j], bias[cg * 16 + 2 * j + 1]};
    for (int c = 0; c < Cin; ++c) {
        for (int ky = 0; ky < 7; ++ky) {
            const float* prow = patch + (c * STEM_P + 2 * py + ky) * STEM_P + 2 * px;
            float in[13];
#pragma unroll
            for (int q = 0; q < 13; ++q) in[q] = prow[q];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const f32x4* w4 = (const f32x4*)(wk + ((c * 7 + ky) * 7 + kx) * 64 + cg * 16);
                f32x4 w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = w4[q];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x2 v2 = {in[2 * p + kx], in[2 * p + kx]};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[p][2 * q] = __builtin_elementwise_fma(v2, f32x2{w[q][0], w[q][1]}, acc[p][2 * q]);
                        acc[p][2 * q + 1] = __builtin_elementwise_fma(v2, f32x2{w[q][2], w[q][3]}, acc[p][2 * q + 1]);
                    }
                }
            }
        }
    }
    const int oy = ty0 + py;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int ox = tx0 + px + p;
        if (oy < OH && ox < OW) {
            T* o = out + (((size_t)b * OH + oy) * OW + ox) * 64 + cg * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[p][2 * q][0], acc[p][2 * q][1], acc[p][2 * q + 1][0], acc[p][2 * q + 1][1]};
                Vec4<T>::st(o + q * 4, v);
            }
        }
    }
}

__global__ void pack_stem_kernel(int Cin, const float* __restrict__ w, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // dst index [k][o]
    if (i >= Cin * 49 * 64) return;
    const int o = i & 63, k = i >> 6;
    dst[i] = w[(size_t)o * Cin * 49 + k];
}

int launch_pack_stem(chore_handle* h, int Cin, const float* w, float* dst, hipStream_t s) {
    const int n = Cin * 49 * 64;
    hipLaunchKernelGGL(pack_stem_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Cin, w, dst);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

int launch_stem(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W, const float* wk,
                const float* bias, void* out, hipStream_t s) {
    if (Cin > STEM_MAXC) CHORE_FAIL(h, CHORE_EINVAL, "stem: Cin > %d", STEM_MAXC);
    const int OH = H / 2, OW = W / 2;
    dim3 grid((OW + STEM_T - 1) / STEM_T, (OH + STEM_T - 1) / STEM_T, B);
    const size_t smem = (size_t)Cin * STEM_P * STEM_P * sizeof(float);
    bool* attr[3] = {&CHORE_ONCE_FLAG(h), &CHORE_ONCE_FLAG(h), &CHORE_ONCE_FLAG(h)};
    if (dtype == CHORE_F16) {
        if (!*attr[2]) {
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)stem_kernel<h16_t>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            *attr[2] = true;
        }
        hipLaunchKernelGGL(stem_kernel<h16_t>, grid, dim3(256), smem, s, images, B, Cin, H, W, wk, bias, (h16_t*)out);
    } else if (dtype == CHORE_F32) {
        if (!*attr[0]) {
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)stem_kernel<float>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            *attr[0] = true;
        }
        hipLaunchKernelGGL(stem_kernel<float>, grid, dim3(256), smem, s, images, B, Cin, H, W, wk, bias, (float*)out);
    } else {
        if (!*attr[1]) {
            CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)stem_kernel<bf16_t>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            *attr[1] = true;
        }
        hipLaunchKernelGGL(stem_kernel<bf16_t>, grid, dim3(256), smem, s, images, B, Cin, H, W, wk, bias,
                           (bf16_t*)out);
    }
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// stem on the matrix cores (fp16 x 3 mode, round 6): the same 7x7 stride-2 convolution as a GEMM  out^T[64 channels][256 pixels] =
// W[64][K] x im2col[K][256 pixels]  with K = Cin * 7 kernel rows of 8 (7 taps + a zero) = 18 k-steps of v_mfma_f32_32x32x16_f16 for the
// released 5-channel input, three MFMAs per product on hi / lo split operands (enc_common.h).  The stem_kernel above is bound by the vector ALU (78 TFLOP/s of packed fp32
// FMAs: 104 us per 4 x 512^2 images); here the FMAs are 192 MFMAs per wave and tile.
//   * workgroup = 16 x 16 output pixels, 4 waves; wave w owns output rows 4 w .. 4 w + 3 as two 32-pixel column blocks (2 rows x 16);
//   * A operand = weights (rows = channels): fragment-ordered [k-step][row block][plane][lane] by pack_stem_x3_kernel, streamed from
//     the L2 three k-steps ahead (64 KB, the same for every workgroup);
//   * B operand = pixels: the 8 consecutive k of lane (half, col) are one kernel row = 8 consecutive floats of a patch row (stride 40:
//     8-byte aligned): four ds_read_b64, then the hi / lo split (v_cvt_pk_f16_f32, v_fma_mix_f32);
//   * D fragment: lane = pixel, registers = 4 groups of 4 consecutive channels: 16-byte stores straight into the NHWC map.
// The sum runs in another order than stem_kernel's (k-steps of 16, three terms per product) and each product carries 2^-22: the
// fp32 mode keeps stem_kernel (its results equal the reference's to the last bits the oracle test asks for).
// ------------------------------------------------------------------------------------------------
typedef _Float16 sx_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sx_f16x2 __attribute__((ext_vector_type(2)));
typedef float sx_f32x2 __attribute__((ext_vector_type(2)));
// K order: k = (c * 7 + ky) * 8 + kx with kx padded 7 -> 8 (zero weight): the 8 consecutive k of a lane are 8 consecutive floats of one
// patch row -- four 8-byte LDS reads instead of eight gathers through an offset table.  (c, ky) rows: Cin * 7 = 35 -> 18 k-steps of two.
constexpr int SX_KS = 18, SX_PF = 3, SX_PS = 40;                   // SX_PS: patch row stride in floats (even: 8-byte aligned reads)
constexpr size_t STEM_X3_BYTES = (size_t)SX_KS * 2 * 2 * 1024;     // [ks][rb][plane][lane] u32x4

__global__ void pack_stem_x3_kernel(int Cin, const float* __restrict__ w /*(64,Cin,7,7)*/, u32x4* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // (ks, rb, lane)
    if (i >= SX_KS * 2 * 64) return;
    const int lane = i & 63, rb = (i >> 6) & 1, ks = i >> 7;
    const int n = rb * 32 + (lane & 31);
    const int q = 2 * ks + (lane >> 5);                              // (c, ky) row of this half of the k-step
    sx_f16x8 hh, ll;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool real = q < Cin * 7 && j < 7;
        const float ws = (real ? w[((size_t)n * Cin * 7 + q) * 7 + j] : 0.f) * (float)(1 << X3_WSHIFT);
        hh[j] = (_Float16)ws;
        ll[j] = (_Float16)(ws - (float)hh[j]);
    }
    dst[((ks * 2 + rb) * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, hh);
    dst[((ks * 2 + rb) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, ll);
}

__device__ __forceinline__ unsigned sx_cvt_pk(float a, float b) {
    const sx_f16x2 h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}
template <int HI> __device__ __forceinline__ float sx_sub_half(float y, unsigned h) {      // y - float(half HI of h): v_fma_mix_f32
    float r;
    if constexpr (HI) asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y));
    else asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y));
    return r;
}

// tile = 16 rows x 32 pixels: wave w owns rows 4 w .. 4 w + 3, one 32-pixel column block per row (four blocks x two channel blocks =
// eight accumulators).  (First version: 16 x 16 tiles, two column blocks per wave -- every wave streams all 72 KB of weight fragments
// from the L2 whatever its pixel count, 295 MB per launch: 53 us.  Twice the pixels per fetch: profiles/r06_stem.txt.)
constexpr int SX_TW = 32, SX_PW = 2 * SX_TW + 5, SX_PS2 = 72;      // patch columns (69) and row stride in floats (even, >= 69 + 3)
template <int CIN>
__global__ __launch_bounds__(256) void stem_x3_kernel(const float* __restrict__ img, int B, int H, int W,
                                                      const u32x4* __restrict__ wfr, const float* __restrict__ bias,
                                                      float* __restrict__ out) {
    f16_saturate_mode();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* patch = sm;                                   // [CIN][37][SX_PS2]
    constexpr int PLANE = STEM_P * SX_PS2;
    const int OH = H / 2, OW = W / 2;
    const int b = blockIdx.z, ty0 = blockIdx.y * STEM_T, tx0 = blockIdx.x * SX_TW;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    struct Frag { u32x4 h[2], l[2]; };
    Frag ring[SX_PF];
    auto load_frag = [&](Frag& f, int ks) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) { f.h[rb] = wfr[((ks * 2 + rb) * 2 + 0) * 64 + lane]; f.l[rb] = wfr[((ks * 2 + rb) * 2 + 1) * 64 + lane]; }
    };
#pragma unroll
    for (int p = 0; p < SX_PF; ++p) load_frag(ring[p], p);           // in flight under the patch staging
    const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
    {   // the patch: a row (69 floats: lanes 0 .. 63, then lanes 0 .. 7 again, the last three of them padding) per wave and pass,
        // all of a channel's rows in flight
        constexpr int RPW = (STEM_P + 3) / 4;
        const int x0 = ix0 + lane, x1 = ix0 + 64 + lane;
        const bool in0 = x0 >= 0 && x0 < W, in1 = lane < SX_PW - 64 && x1 >= 0 && x1 < W;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const float* plane = img + ((size_t)b * CIN + c) * H * W;
            float v0[RPW], v1[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int ry = wv + 4 * i, y = iy0 + ry;
                const bool yin = ry < STEM_P && y >= 0 && y < H;
                const float a0 = plane[yin && in0 ? (size_t)y * W + x0 : 0], a1 = plane[yin && in1 ? (size_t)y * W + x1 : 0];
                v0[i] = yin && in0 ? a0 : 0.f;
                v1[i] = yin && in1 ? a1 : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int ry = wv + 4 * i;
                if (ry < STEM_P) {
                    patch[c * PLANE + ry * SX_PS2 + lane] = v0[i];
                    if (lane < SX_PS2 - 64) patch[c * PLANE + ry * SX_PS2 + 64 + lane] = v1[i];
                }
            }
        }
    }
    __syncthreads();
    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        f32x16 bf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(const f32x4*)(bias + rb * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) bf[4 * g + e] = b4[e] * (float)(1 << X3_WSHIFT);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = bf;
    }
    const int pbase0 = (2 * (4 * wv)) * SX_PS2 + 2 * col;            // column block cb: row 4 wv + cb -> + cb * 2 * SX_PS2
#pragma unroll
    for (int ks = 0; ks < SX_KS; ++ks) {
        // (c, ky) rows 2 ks (lower half of the wave) and 2 ks + 1 (upper half); past the last row: row 0 again (its weights are zero)
        constexpr int NQ = CIN * 7;
        const int q0 = 2 * ks < NQ ? 2 * ks : 0, q1 = 2 * ks + 1 < NQ ? 2 * ks + 1 : 0;
        const int off0 = (q0 / 7) * PLANE + (q0 % 7) * SX_PS2, off1 = (q1 / 7) * PLANE + (q1 % 7) * SX_PS2;
        const int off = pbase0 + (half ? off1 : off0);
        u32x4 bh[4], bl[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const sx_f32x2* src = (const sx_f32x2*)(patch + off + cb * 2 * SX_PS2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const sx_f32x2 v = src[j];
                bh[cb][j] = sx_cvt_pk(v[0], v[1]);
                bl[cb][j] = sx_cvt_pk(sx_sub_half<0>(v[0], bh[cb][j]), sx_sub_half<1>(v[1], bh[cb][j]));
            }
        }
        Frag& f = ring[ks % SX_PF];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const sx_f16x8 a0 = __builtin_bit_cast(sx_f16x8, f.h[rb]), a1 = __builtin_bit_cast(sx_f16x8, f.l[rb]);
                const sx_f16x8 b0 = __builtin_bit_cast(sx_f16x8, bh[cb]), b1 = __builtin_bit_cast(sx_f16x8, bl[cb]);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[rb][cb], 0, 0, 0);   // small terms first
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[rb][cb], 0, 0, 0);
            }
        if (ks + SX_PF < SX_KS) load_frag(f, ks + SX_PF);
    }
    const float inv = 1.0f / (float)(1 << X3_WSHIFT);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int oy = ty0 + 4 * wv + cb, ox = tx0 + col;
        if (oy < OH && ox < OW) {
            float* o = out + (((size_t)b * OH + oy) * OW + ox) * 64;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[rb][cb][4 * g] * inv, acc[rb][cb][4 * g + 1] * inv, acc[rb][cb][4 * g + 2] * inv, acc[rb][cb][4 * g + 3] * inv};
                    *(f32x4*)(o + rb * 32 + 8 * g + 4 * half) = v;
                }
        }
    }
}

size_t stem_x3_bytes() { return STEM_X3_BYTES; }
int launch_pack_stem_x3(chore_handle* h, int Cin, const float* w, void* dst, hipStream_t s) {
    if (Cin * 7 > 2 * SX_KS) CHORE_FAIL(h, CHORE_EINVAL, "stem (fp16 x 3): Cin * 7 > %d kernel rows", 2 * SX_KS);
    hipLaunchKernelGGL(pack_stem_x3_kernel, dim3((SX_KS * 2 * 64 + 255) / 256), dim3(256), 0, s, Cin, w, (u32x4*)dst);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}
// the instantiated input widths: RGBM3 (5 channels: the released model), RGB (3), RGBM (4).  CHORE_STEM_VALU=1: the fp16 x 3 mode on
// stem_kernel as before round 6 (A/B)
bool stem_x3_on(int Cin) {
    static const bool off = getenv("CHORE_STEM_VALU") != nullptr;
    return !off && (Cin == 5 || Cin == 4 || Cin == 3);
}
int launch_stem_x3(chore_handle* h, const float* images, int B, int Cin, int H, int W, const void* wfr, const float* bias, float* out,
                   hipStream_t s) {
    const int OH = H / 2, OW = W / 2;
    dim3 grid((OW + SX_TW - 1) / SX_TW, (OH + STEM_T - 1) / STEM_T, B);
    const size_t smem = (size_t)Cin * STEM_P * SX_PS2 * sizeof(float);
    bool& attr = CHORE_ONCE_FLAG(h);
    if (!attr) {       // 53 KB for five channels
        (void)hipFuncSetAttribute((const void*)stem_x3_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute((const void*)stem_x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute((const void*)stem_x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    if (Cin == 5) hipLaunchKernelGGL(stem_x3_kernel<5>, grid, dim3(256), smem, s, images, B, H, W, (const u32x4*)wfr, bias, out);
    else if (Cin == 4) hipLaunchKernelGGL(stem_x3_kernel<4>, grid, dim3(256), smem, s, images, B, H, W, (const u32x4*)wfr, bias, out);
    else if (Cin == 3) hipLaunchKernelGGL(stem_x3_kernel<3>, grid, dim3(256), smem, s, images, B, H, W, (const u32x4*)wfr, bias, out);
    else CHORE_FAIL(h, CHORE_EINVAL, "stem (fp16 x 3): Cin = %d not instantiated", Cin);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics of a tensor that no convolution produced (pool / upsample / stem outputs):
// one streaming pass, per-block per-channel fp32 partials, then exact 128-bit fixed-point atomics
// (enc_common.h) -> order-independent totals.
// ------------------------------------------------------------------------------------------------
int gn_splits(int HW) {
    // >= 64 pixels per slice, at most 256 slices per image (round 4; 64 before): the 256^2 x 64-channel stem output of 4 images
    // was read by 256 workgroups with one load per thread in flight -- 2.4 TB/s
    int s = HW / 64;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return s;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int cs, int co, int C, int HW,
                                                       int S, GroupStat* __restrict__ st) {
    __shared__ float red[2][1024];
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int tpr = C / 4, P = 256 / tpr;
    const int cv = tid % tpr, pl = tid / tpr;
    const int p0 = (int)((long long)HW * s / S), p1 = (int)((long long)HW * (s + 1) / S);
    const T* base = x + (size_t)b * HW * cs + co + cv * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;          // loads in flight per thread (rows past the slice re-read its last row and add nothing)
    for (int p = p0 + pl; p < p1; p += U * P) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = p + u * P;
            v[u] = Vec4<T>::ld(base + (size_t)(q < p1 ? q : p1 - 1) * cs);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u * P < p1) {
                sum += v[u];
                sq += v[u] * v[u];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][pl * C + cv * 4 + j] = sum[j];
        red[1][pl * C + cv * 4 + j] = sq[j];
    }
    __syncthreads();
    if (tid < C) {
        float a = 0.f, q = 0.f;
        for (int i = 0; i < P; ++i) { a += red[0][i * C + tid]; q += red[1][i * C + tid]; }
        const int gs = C / GN_GROUPS;
        a = group_lane_sum(a, gs);
        q = group_lane_sum(q, gs);
        GroupStat* o = st + (size_t)b * GN_GROUPS + tid / gs;      // two lanes of the group, concurrently
        if (tid % gs == 0) stat_add(&o->sum, act_hi_cells((int)gridDim.y), a);
        if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells((int)gridDim.y), q);
    }
}

int launch_gn_stats(chore_handle* h, int dtype, const View& x, int B, int HW, GroupStat* st, hipStream_t s) {
    if (x.C % GN_GROUPS || x.C > 256 || x.C < 32) CHORE_FAIL(h, CHORE_EINVAL, "gn: unsupported C=%d", x.C);
    const int S = gn_splits(HW);
    dim3 grid(S, B);
    if (dtype == CHORE_F16)
        hipLaunchKernelGGL(gn_stats_kernel<h16_t>, grid, dim3(256), 0, s, (const h16_t*)x.p, x.cs, x.co, x.C, HW, S, st);
    else if (dtype == CHORE_F32)
        hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)x.p, x.cs, x.co, x.C, HW, S, st);
    else
        hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x.p, x.cs, x.co, x.C, HW, S,
                           st);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// elementwise: y = relu(groupnorm(x)); grid (blocks per image, B); the affine is derived per block
// ------------------------------------------------------------------------------------------------
// STATS: the statistics of y (the values as stored) in the same pass -- tmpx is the input of a ConvBlock whose GroupNorm needs
// them (round 5: was a gn_stats_kernel pass over tmpx).  A thread's vectors all cover the same four channels (the grid stride is
// a multiple of C / 4), so its sums are per channel; the workgroup adds them up in a fixed order as gn_stats_kernel does.
template <typename T, bool STATS>
__global__ __launch_bounds__(256) void gn_apply_relu_kernel(const T* __restrict__ x, int xcs, int xco,
                                                            const GroupStat* __restrict__ st,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            int ycs, int yco, int C, int HW, GroupStat* __restrict__ st_out) {
    __shared__ float ss[512];
    __shared__ float red[STATS ? 2 : 1][STATS ? 1024 : 1];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < C) gn_scale_shift(st, (int)gridDim.y, b, C, tid, HW, gamma, beta, ss[2 * tid], ss[2 * tid + 1]);
    __syncthreads();
    const int tpr = C / 4;
    const size_t total4 = (size_t)HW * tpr;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total4; i += (size_t)gridDim.x * 256) {
        const int cv = (int)(i % tpr);
        const size_t p = (size_t)b * HW + i / tpr;
        const f32x4 v = Vec4<T>::ld(x + p * xcs + xco + cv * 4);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = fmaf(v[j], ss[2 * (cv * 4 + j)], ss[2 * (cv * 4 + j) + 1]);
            r[j] = t > 0.f ? t : 0.f;
        }
        if constexpr (STATS) {
            const f32x4 w = Vec4<T>::st_round(y + p * ycs + yco + cv * 4, r);
            sum += w;
            sq += w * w;
        } else Vec4<T>::st(y + p * ycs + yco + cv * 4, r);
    }
    if constexpr (STATS) {
        const int cv = tid % tpr, pl = tid / tpr, P = 256 / tpr;      // (256 and the grid stride are multiples of tpr: cv is the thread's)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][pl * C + cv * 4 + j] = sum[j];
            red[1][pl * C + cv * 4 + j] = sq[j];
        }
        __syncthreads();
        if (tid < C) {
            float a = 0.f, q = 0.f;
            for (int i = 0; i < P; ++i) { a += red[0][i * C + tid]; q += red[1][i * C + tid]; }
            const int gs = C / GN_GROUPS;
            a = group_lane_sum(a, gs);
            q = group_lane_sum(q, gs);
            GroupStat* o = st_out + (size_t)b * GN_GROUPS + tid / gs;
            if (tid % gs == 0) stat_add(&o->sum, act_hi_cells((int)gridDim.y), a);
            if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells((int)gridDim.y), q);
        }
    }
}

// st_out: zeroed statistics cells of y, accumulated in the same launch (C a multiple of 32 with 256 % (C / 4) == 0: 32 ... 256 in
// powers of two)
int launch_gn_apply_relu(chore_handle* h, int dtype, const View& x, const GroupStat* st, const float* gamma,
                         const float* beta, const View& y, int B, int HW, hipStream_t s, GroupStat* st_out) {
    if (x.C > 256 || x.C % GN_GROUPS) CHORE_FAIL(h, CHORE_EINVAL, "gn: unsupported C=%d", x.C);
    const size_t total4 = (size_t)HW * (x.C / 4);
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (st_out) {
        if (256 % (x.C / 4)) CHORE_FAIL(h, CHORE_EINVAL, "gn_apply_relu with statistics: unsupported C=%d", x.C);
        if (blocks > 256) blocks = 256;       // 256 x 2 x C / 32 atomics per image, as gn_stats_kernel's at most 256 slices
    }
    dim3 grid(blocks, B);
#define GN_APPLY(T, ST) hipLaunchKernelGGL((gn_apply_relu_kernel<T, ST>), grid, dim3(256), 0, s, (const T*)x.p, x.cs, x.co, st, gamma, beta, \
                                            (T*)y.p, y.cs, y.co, x.C, HW, st_out)
    if (dtype == CHORE_F16) { if (st_out) GN_APPLY(h16_t, true); else GN_APPLY(h16_t, false); }
    else if (dtype == CHORE_F32) { if (st_out) GN_APPLY(float, true); else GN_APPLY(float, false); }
    else { if (st_out) GN_APPLY(bf16_t, true); else GN_APPLY(bf16_t, false); }
#undef GN_APPLY
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// Pixel-wise producers with fused GroupNorm statistics.  A block owns an 8x8 tile of output pixels of one
// image and all C channels: thread = (channel quad, pixel lane); it stores its values and sums the values
// AS STORED; the block then reduces in a fixed order and adds its partials to the exact accumulators
// (enc_common.h), so no separate statistics pass over the output is needed.  C is a template parameter so
// that every per-thread loop has a compile-time trip count: the loads of a loop are then issued together
// instead of one global round trip per iteration.
// ------------------------------------------------------------------------------------------------
constexpr int MAP_T = 8;   // tile edge

// 2x2 average pooling   (F.avg_pool2d(x, 2, stride=2), HGFilters.py:32,152)
template <typename T, int C> struct PoolOp {
    const T* x; int xcs, xco, H, W;   // input view and size
    static constexpr size_t SMEM = 0;
    static constexpr int CT = C, TW = MAP_T;      // channels and tile columns per workgroup
    __device__ __forceinline__ void stage(char*, int, int, int, int, int, int) const {}
    // the MAP_T outputs of tile column ox, rows oy0.. (rows beyond OH repeat the last row; the caller skips them)
    __device__ __forceinline__ void column(const char*, int b, int oy0, int ox, int, int c0, int cv, int OH,
                                           f32x4 (&out)[MAP_T]) const {
#pragma unroll
        for (int r = 0; r < MAP_T; ++r) {
            const int oy = oy0 + r < OH ? oy0 + r : OH - 1;
            const T* src = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * xcs + xco + c0 + cv * 4;
            const f32x4 a = Vec4<T>::ld(src), bb = Vec4<T>::ld(src + xcs);
            const f32x4 c = Vec4<T>::ld(src + (size_t)W * xcs), d = Vec4<T>::ld(src + (size_t)W * xcs + xcs);
            out[r] = (((a + bb) + c) + d) * 0.25f;
        }
    }
};

// a + bicubic_up2(low), align_corners=True      (HGFilters.py:47,50)
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}
// The 16 taps of an output pixel come from an 8x8 low-resolution neighbourhood shared by the whole tile
// (scale < 1/2: 8 output rows span at most 5 source rows, +3 for the kernel support).  It is staged in LDS
// once -- border clamping applied while staging -- instead of being fetched 64x through the 32 KB L1, which
// the tile's working set does not fit.  One thread then produces a whole tile column: the horizontal
// interpolation of the 8 patch rows is done once (8 x 4 taps) and every output row combines 4 of those
// (same two-pass order as ATen's upsample_bicubic2d: x first, then y).
// Round 4: a workgroup owns 8 x 16 output pixels x 64 CHANNELS (was 8 x 8 pixels x all C channels): the staged patch is
// 8 x 12 source pixels x 64 channels = 24 KB instead of 64 KB, so four workgroups share a CU instead of two and the grid
// is 4 x larger (2 048 workgroups at 128^2); the source over-read drops from 1.0 to 0.75 staged values per output.
template <typename T, int C> struct UpAddOp {
    static constexpr int CT = 64, TW = 16;            // channels and tile columns per workgroup
    static constexpr int PS = 8, PSX = TW / 2 + 4;    // patch rows / columns (8 output rows span <= 5 source rows, 16 columns <= 9; + 3 taps)
    static constexpr int TPR = CT / 4, P = 256 / TPR;
    static constexpr size_t SMEM = (size_t)PS * PSX * CT * sizeof(T);
    static_assert(C % CT == 0 && (PS * PSX) % P == 0 && TW % P == 0, "upadd tile geometry");
    const T* a; int acs, aco;
    const T* low; int lcs, lco, H, W;   // low-resolution view and size (output is 2H x 2W)
    __device__ __forceinline__ float scale_y() const { return (float)(H - 1) / (float)(2 * H - 1); }
    __device__ __forceinline__ float scale_x() const { return (float)(W - 1) / (float)(2 * W - 1); }
    __device__ __forceinline__ void stage(char* sm, int b, int oy0, int ox0, int c0, int cv, int pl) const {
        const int iy0 = (int)floorf(scale_y() * (float)oy0) - 1, ix0 = (int)floorf(scale_x() * (float)ox0) - 1;
        constexpr int N = PS * PSX / P;
        typedef typename std::conditional<sizeof(T) == 2, unsigned long long, f32x4>::type Raw;
        Raw v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int q = pl + i * P;
            int yy = iy0 + q / PSX, xx = ix0 + q % PSX;
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            v[i] = *(const Raw*)(low + (((size_t)b * H + yy) * W + xx) * lcs + lco + c0 + cv * 4);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) *(Raw*)((T*)sm + (size_t)(pl + i * P) * CT + cv * 4) = v[i];
    }
    __device__ __forceinline__ void column(const char* sm, int b, int oy0, int ox, int ox0, int c0, int cv, int OH,
                                           f32x4 (&out)[MAP_T]) const {
        const int OW = 2 * W;
        const int iy0 = (int)floorf(scale_y() * (float)oy0) - 1, ix0 = (int)floorf(scale_x() * (float)ox0) - 1;
        f32x4 av[MAP_T];
#pragma unroll
        for (int r = 0; r < MAP_T; ++r) {   // issued first: they land while the interpolation runs
            const int oy = oy0 + r < OH ? oy0 + r : OH - 1;
            av[r] = Vec4<T>::ld(a + (((size_t)b * 2 * H + oy) * OW + ox) * acs + aco + c0 + cv * 4);
        }
        const float rx = scale_x() * (float)ox, fx = floorf(rx);
        const int kx = (int)fx - 1 - ix0;
        float cx[4];
        cubic_coeffs(rx - fx, cx);
        f32x4 rows[PS];
#pragma unroll
        for (int k = 0; k < PS; ++k) {
            const T* pr = (const T*)sm + (size_t)(k * PSX + kx) * CT + cv * 4;
            f32x4 acc = Vec4<T>::ld(pr) * cx[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const f32x4 v = Vec4<T>::ld(pr + (size_t)q * CT);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(v[e], cx[q], acc[e]);
            }
            rows[k] = acc;
        }
#pragma unroll
        for (int r = 0; r < MAP_T; ++r) {
            const int oy = oy0 + r < OH ? oy0 + r : OH - 1;
            const float ry = scale_y() * (float)oy, fy = floorf(ry);
            const int ky = (int)fy - 1 - iy0;   // 0..4
            float cy[4];
            cubic_coeffs(ry - fy, cy);
            // weights of the 8 patch rows for this output row (zero outside ky..ky+3): static register indexing
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            bool first = true;
#pragma unroll
            for (int k = 0; k < PS; ++k) {
                const int j = k - ky;
                if (j >= 0 && j < 4) {   // wave-uniform (oy is the same for the whole wave)
                    const float w = cy[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = first ? rows[k][e] * w : fmaf(rows[k][e], w, acc[e]);
                    first = false;
                }
            }
            out[r] = av[r] + acc;
        }
    }
};

template <typename T, int C, typename Op>
__global__ __launch_bounds__(256) void map_stats_kernel(Op op, T* y, int ycs, int yco, int OH, int OW,
                                                        GroupStat* __restrict__ st) {
    extern __shared__ __attribute__((aligned(16))) char map_sm[];
    __shared__ float red[2][1024];
    constexpr int CT = Op::CT, TW = Op::TW;        // channels / tile columns of a workgroup (grid.z = C / CT channel slices)
    constexpr int TPR = CT / 4, P = 256 / TPR;
    static_assert(P * CT <= 1024, "partial-sum scratch");
    const int b = blockIdx.y, tid = threadIdx.x, c0 = blockIdx.z * CT;
    const int cv = tid % TPR, pl = tid / TPR;
    const int tiles_x = (OW + TW - 1) / TW;
    const int oy0 = (blockIdx.x / tiles_x) * MAP_T, ox0 = (blockIdx.x % tiles_x) * TW;
    op.stage(map_sm, b, oy0, ox0, c0, cv, pl);
    __syncthreads();
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int col = pl; col < TW; col += P) {
        const int ox = ox0 + col;
        if (ox < OW) {
            f32x4 out[MAP_T];
            op.column(map_sm, b, oy0, ox, ox0, c0, cv, OH, out);
#pragma unroll
            for (int r = 0; r < MAP_T; ++r) {
                const int oy = oy0 + r;
                if (oy < OH) {
                    const f32x4 v = Vec4<T>::st_round(y + (((size_t)b * OH + oy) * OW + ox) * ycs + yco + c0 + cv * 4, out[r]);
                    sum += v;
                    sq += v * v;
                }
            }
        }
    }
    if (!st) return;   // uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][pl * CT + cv * 4 + j] = sum[j];
        red[1][pl * CT + cv * 4 + j] = sq[j];
    }
    __syncthreads();
    if (tid < CT) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < P; ++i) { a += red[0][i * CT + tid]; q += red[1][i * CT + tid]; }
        constexpr int gs = C / GN_GROUPS;
        static_assert(CT % gs == 0, "a channel slice holds whole GroupNorm groups");
        a = group_lane_sum(a, gs);
        q = group_lane_sum(q, gs);
        GroupStat* o = st + (size_t)b * GN_GROUPS + (c0 + tid) / gs;      // two lanes of the group, concurrently
        if (tid % gs == 0) stat_add(&o->sum, act_hi_cells((int)gridDim.y), a);
        if (tid % gs == (gs > 1 ? 1 : 0)) stat_add(&o->sq, act_hi_cells((int)gridDim.y), q);
    }
}

template <typename T, int C, typename Op>
static int launch_map_c(chore_handle* h, const Op& op, const View& y, int B, int OH, int OW, GroupStat* st,
                        hipStream_t s) {
    dim3 grid(((OH + MAP_T - 1) / MAP_T) * ((OW + Op::TW - 1) / Op::TW), B, C / Op::CT);
    bool& attr = CHORE_ONCE_FLAG(h);   // per instantiation
    if (!attr && Op::SMEM > 32 * 1024) {
        CHORE_HIP_CHECK(h, hipFuncSetAttribute((const void*)map_stats_kernel<T, C, Op>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)Op::SMEM));
        attr = true;
    }
    hipLaunchKernelGGL((map_stats_kernel<T, C, Op>), grid, dim3(256), Op::SMEM, s, op, (T*)y.p, y.cs, y.co, OH, OW, st);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

template <typename T>
static int launch_pool_t(chore_handle* h, const View& x, const View& y, int B, int H, int W, GroupStat* st,
                         hipStream_t s) {
    switch (y.C) {
        case 64: return launch_map_c<T, 64>(h, PoolOp<T, 64>{(const T*)x.p, x.cs, x.co, H, W}, y, B, H / 2, W / 2, st, s);
        case 128: return launch_map_c<T, 128>(h, PoolOp<T, 128>{(const T*)x.p, x.cs, x.co, H, W}, y, B, H / 2, W / 2, st, s);
        case 256: return launch_map_c<T, 256>(h, PoolOp<T, 256>{(const T*)x.p, x.cs, x.co, H, W}, y, B, H / 2, W / 2, st, s);
    }
    CHORE_FAIL(h, CHORE_EINVAL, "avgpool2: unsupported C=%d (64, 128, 256)", y.C);
}

int launch_avgpool2(chore_handle* h, int dtype, const View& x, const View& y, int B, int H, int W, GroupStat* st,
                    hipStream_t s) {
    if (dtype == CHORE_F16) return launch_pool_t<h16_t>(h, x, y, B, H, W, st, s);
    return dtype == CHORE_F32 ? launch_pool_t<float>(h, x, y, B, H, W, st, s)
                              : launch_pool_t<bf16_t>(h, x, y, B, H, W, st, s);
}

template <typename T>
static int launch_upadd_t(chore_handle* h, const View& a, const View& low, const View& y, int B, int H, int W,
                          GroupStat* st, hipStream_t s) {
    switch (y.C) {
        case 64: return launch_map_c<T, 64>(h, UpAddOp<T, 64>{(const T*)a.p, a.cs, a.co, (const T*)low.p, low.cs, low.co, H, W}, y, B, 2 * H, 2 * W, st, s);
        case 128: return launch_map_c<T, 128>(h, UpAddOp<T, 128>{(const T*)a.p, a.cs, a.co, (const T*)low.p, low.cs, low.co, H, W}, y, B, 2 * H, 2 * W, st, s);
        case 256: return launch_map_c<T, 256>(h, UpAddOp<T, 256>{(const T*)a.p, a.cs, a.co, (const T*)low.p, low.cs, low.co, H, W}, y, B, 2 * H, 2 * W, st, s);
    }
    CHORE_FAIL(h, CHORE_EINVAL, "upadd: unsupported C=%d (64, 128, 256)", y.C);
}

// y may alias a (in-place add): every element is read and written by the same thread
int launch_upadd(chore_handle* h, int dtype, const View& a, const View& low, const View& y, int B, int H, int W,
                 GroupStat* st, hipStream_t s) {
    if (dtype == CHORE_F16) return launch_upadd_t<h16_t>(h, a, low, y, B, H, W, st, s);
    return dtype == CHORE_F32 ? launch_upadd_t<float>(h, a, low, y, B, H, W, st, s)
                              : launch_upadd_t<bf16_t>(h, a, low, y, B, H, W, st, s);
}

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
int launch_copy_f32(chore_handle* h, const float* src, float* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// max |x| of a gradient tensor (fp16 x 3 training: the operand scale of the data- and weight-gradient GEMMs, enc_common.h).
// Exactly AMAX_CELLS workgroups, each strides over the tensor with eight 16-byte loads in flight per thread and stores ITS
// maximum: plain stores, no zero-initialised cells, order-independent (a maximum is).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_f32_kernel(const float* __restrict__ x, size_t n4, unsigned* __restrict__ cells) {
    absmax_block(x, n4, cells, blockIdx.x);
}
int launch_absmax_f32(chore_handle* h, const float* x, size_t n, unsigned* cells, hipStream_t s) {
    if (!x || !cells || (n & 3) || ((size_t)x & 15)) CHORE_FAIL(h, CHORE_EINVAL, "absmax: needs a 16-byte aligned tensor of a multiple of 4 floats");
    hipLaunchKernelGGL(absmax_f32_kernel, dim3(AMAX_CELLS), dim3(256), 0, s, x, n / 4, cells);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// backward of the bicubic x2 upsampling (align_corners=True, A=-0.75; HGFilters.py:47): d_low = U^T dy.
// Gather form -- every low-resolution pixel sums the high-resolution pixels whose 4x4 support contains it, with
// the separable weights recomputed on the fly (border clamping folds several taps onto the edge pixels) -- so there
// are no atomics (ATen's upsample_bicubic2d_backward scatters with atomics and took 125 ms per call here).
// ------------------------------------------------------------------------------------------------
// weight of high-resolution coordinate o (of 2n) on low-resolution coordinate l (of n)
__device__ __forceinline__ float up2_weight(int o, int l, int n) {
    const float s = (float)(n - 1) / (float)(2 * n - 1);
    const float r = s * (float)o, f = floorf(r);
    const int i = (int)f;
    if (i < l - 2 || i > l + 2) return 0.f;
    float c[4];
    cubic_coeffs(r - f, c);
    float w = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int t = i - 1 + k;
        t = t < 0 ? 0 : (t > n - 1 ? n - 1 : t);
        if (t == l) w += c[k];
    }
    return w;
}

template <typename T>
__global__ __launch_bounds__(256) void up2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dlow, int C, int H, int W,
                                                      size_t total4) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i0 < total4;
    const size_t i = live ? i0 : total4 - 1;
    const int tpr = C / 4;
    const int cv = (int)(i % tpr);
    size_t p = i / tpr;
    const int lx = (int)(p % W); p /= W;
    const int ly = (int)(p % H);
    const int b = (int)(p / H);
    constexpr int NC = 12;
    const int oy0 = max(0, 2 * ly - 5), ox0 = max(0, 2 * lx - 5);
    float wy[NC], wx[NC];
    // Every thread evaluates its pixel's 24 weights itself.  Until round 4 one lane of the pixel's tpr threads evaluated each
    // weight and the others fetched it with a cross-lane read (`__shfl` = ds_bpermute): with a SECOND process keeping the same GPU
    // busy that version produced different results in 1 - 3 waves of a call in 32 of 108 training passes (never alone on the
    // GPU, never when called in isolation; v_readlane instead of ds_bpermute: 8 of 72; this version: 0 of 108 --
    // scripts/train_determinism3.py, profiles/r04_determinism.txt).  What goes wrong below the instruction level is not known.
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        wy[k] = (oy0 + k < 2 * H) ? up2_weight(oy0 + k, ly, H) : 0.f;
        wx[k] = (ox0 + k < 2 * W) ? up2_weight(ox0 + k, lx, W) : 0.f;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < NC; ++ky) {
        if (wy[ky] == 0.f) continue;
        f32x4 row = {0.f, 0.f, 0.f, 0.f};
        const T* src = dy + (((size_t)b * 2 * H + oy0 + ky) * 2 * W) * C + cv * 4;
        // the row's 12 candidate columns with unconditional (clamped) loads, all in flight together: the zero weights at the
        // ends cost a multiply, a branch per tap cost a round trip each
        f32x4 v[NC];
#pragma unroll
        for (int kx = 0; kx < NC; ++kx) v[kx] = Vec4<T>::ld(src + (size_t)min(ox0 + kx, 2 * W - 1) * C);
#pragma unroll
        for (int kx = 0; kx < NC; ++kx)
#pragma unroll
            for (int e = 0; e < 4; ++e) row[e] = fmaf(v[kx][e], wx[kx], row[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(row[e], wy[ky], acc[e]);
    }
    if (live) Vec4<T>::st(dlow + i * 4, acc);
}

int launch_up2_bwd(chore_handle* h, int dtype, const void* dy, void* dlow, int B, int H, int W, int C, hipStream_t s) {
    if (C % 4 || H < 2 || W < 2) CHORE_FAIL(h, CHORE_EINVAL, "up2_bwd: unsupported shape");
    const size_t total4 = (size_t)B * H * W * (C / 4);
    const unsigned blocks = (unsigned)((total4 + 255) / 256);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(up2_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dy, (float*)dlow, C, H, W, total4);
    else
        hipLaunchKernelGGL(up2_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dlow, C, H, W,
                           total4);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// backward of the 2x2 average pooling (HourGlass._forward HGFilters.py:33, HGFilter.forward :153): dx = dy / 4 replicated
template <typename T>
__global__ __launch_bounds__(256) void pool2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int C, int H, int W,
                                                        size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over dx (B,H,W,C/4)
    if (i >= total4) return;
    const int tpr = C / 4;
    const int cv = (int)(i % tpr);
    size_t p = i / tpr;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const size_t b = p / H;
    f32x4 v = Vec4<T>::ld(dy + ((b * (H / 2) + y / 2) * (W / 2) + x / 2) * C + cv * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= 0.25f;
    Vec4<T>::st(dx + i * 4, v);
}

int launch_pool2_bwd(chore_handle* h, int dtype, const void* dy, void* dx, int B, int H, int W, int C, hipStream_t s) {
    if (C % 4 || (H & 1) || (W & 1)) CHORE_FAIL(h, CHORE_EINVAL, "pool2_bwd: unsupported shape");
    const size_t total4 = (size_t)B * H * W * (C / 4);
    const unsigned blocks = (unsigned)((total4 + 255) / 256);
    if (dtype == CHORE_F32)
        hipLaunchKernelGGL(pool2_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dy, (float*)dx, C, H, W, total4);
    else
        hipLaunchKernelGGL(pool2_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, C, H, W,
                           total4);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

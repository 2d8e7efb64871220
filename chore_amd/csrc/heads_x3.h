// heads_x3.h -- the MLP heads on the fp16 matrix cores with hi/lo split operands ("fp16 x 3", see enc_common.h), same
// register-resident chain as heads_f32.h: H^T = W X^T, the D fragment of a layer (row = channel, col = point) becomes the
// B fragment of the next.  For v_mfma_f32_32x32x16_f16 the B operand of lane (half, col) is 8 consecutive k of column
// col; lane (half, col) of a D fragment holds the 16 rows mfma32_row(r, half), r = 0..15, of a 32-row block.  K-step s
// (s = 0, 1) of a block therefore takes registers r = 8s .. 8s+7 -- the rows 16s + {0..3, 8..11} + 4 half -- as they
// are, and the weights are packed in that k order (heads_pack_x3_kernel), so no data moves between layers: relu,
// fp16 hi / lo split, three MFMAs.  16 native fp32 MFMAs (64 cycles each) of a 32-row K block become 6 fp16 MFMAs
// (32 cycles each).  Weights are scaled by 2^X3_WSHIFT at pack time; accumulators carry that scale and are rescaled when
// they are converted (exact powers of two).
#pragma once
#include "heads_f32.h"
#include "enc_common.h"

typedef _Float16 hf16x8_t __attribute__((ext_vector_type(8)));

constexpr int QX_KS1 = 21;                                            // layer 1: 323 -> 336 = 21 k-steps of 16
constexpr size_t QX_L1_VEC = (size_t)HEAD_NUM * QX_KS1 * 4 * 2 * 64;   // [head][ks][rb][plane][lane] u32x4
constexpr size_t QX_L23_VEC = (size_t)HEAD_NUM * 2 * 4 * 2 * 4 * 2 * 64;   // [head][l][kb][s][rb][plane][lane]
constexpr size_t QX_L4_VEC = (size_t)HEAD_NUM * 4 * 2 * 2 * 64;        // [head][kb][s][plane][lane]
constexpr size_t QX_OFF_BYTES = (QF_TOTAL_FLOATS * sizeof(float) + 255) / 256 * 256;   // after the fp32 fragments
constexpr size_t QX_TOTAL_VEC = QX_L1_VEC + QX_L23_VEC + QX_L4_VEC;
constexpr size_t QX_ARENA_BYTES = QX_OFF_BYTES + QX_TOTAL_VEC * 16;
constexpr float QX_SCALE = (float)(1 << X3_WSHIFT), QX_INV = 1.0f / (float)(1 << X3_WSHIFT);

__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    hf16x8_t h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (_Float16)v[j]; l[j] = (_Float16)(v[j] - (float)h[j]); }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}
__device__ __forceinline__ f32x16 mfma3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x16 acc) {
    const hf16x8_t a0 = __builtin_bit_cast(hf16x8_t, ah), a1 = __builtin_bit_cast(hf16x8_t, al);
    const hf16x8_t b0 = __builtin_bit_cast(hf16x8_t, bh), b1 = __builtin_bit_cast(hf16x8_t, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x16 scaled_bias(const float* arena, int head, int layer, int rb, int half) {
    f32x16 b = load_bias_frag(arena, head, layer, rb, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] *= QX_SCALE;
    return b;
}

// hidden layer 1: acc[rb][cb] = 2^s (b1 + W1 X^T)
template <int NCB>
__device__ __forceinline__ void heads_layer1_x3(f32x16 (&acc)[4][NCB], const float* X, const float* arena, int head, int lane) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const f32x16 bf = scaled_bias(arena, head, 0, rb, half);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = bf;
    }
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + (size_t)head * QX_KS1 * 4 * 2 * 64 + lane;
    const float* x0 = X + col * XS + 8 * half;
    u32x4 ah[4], al[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) { ah[rb] = A[(rb * 2) * 64]; al[rb] = A[(rb * 2 + 1) * 64]; }
#pragma unroll 1
    for (int ks = 0; ks < QX_KS1; ++ks) {
        u32x4 nh[4], nl[4];
        const int kn = ks + 1 < QX_KS1 ? ks + 1 : ks;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) { nh[rb] = A[((kn * 4 + rb) * 2) * 64]; nl[rb] = A[((kn * 4 + rb) * 2 + 1) * 64]; }
        u32x4 bh[NCB], bl[NCB];
        // the X rows are zero-filled up to QF_KPAD = 328: the upper half of the last k-step (k = 328..335) lies beyond
        const bool beyond = (ks == QX_KS1 - 1) && half;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float v[8];
            const float* src = x0 + cb * 32 * XS + (beyond ? 0 : ks * 16);       // never read past the row
            const f32x4 p = *(const f32x4*)src, q = *(const f32x4*)(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = beyond ? 0.f : p[j]; v[4 + j] = beyond ? 0.f : q[j]; }
            split8(v, bh[cb], bl[cb]);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mfma3(ah[rb], al[rb], bh[cb], bl[cb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) { ah[rb] = nh[rb]; al[rb] = nl[rb]; }
    }
}

// B fragments (hi, lo) of K-step s of block kb from the previous layer's scaled accumulators: relu(in / 2^s)
template <int NCB>
__device__ __forceinline__ void act_frag(const f32x16 (&in)[4][NCB], int kb, int s, int cb, u32x4& hi, u32x4& lo) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = relu(in[kb][cb][8 * s + j] * QX_INV);
    split8(v, hi, lo);
}

// hidden layers 2 and 3
template <int NCB>
__device__ __forceinline__ void heads_layer_hid_x3(f32x16 (&out)[4][NCB], const f32x16 (&in)[4][NCB], const float* arena, int head,
                                                   int layer /*1|2*/, int lane) {
    const int half = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const f32x16 bf = scaled_bias(arena, head, layer, rb, half);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = bf;
    }
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + QX_L1_VEC +
                     ((size_t)head * 2 + (layer - 1)) * 4 * 2 * 4 * 2 * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 bh[NCB], bl[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) act_frag<NCB>(in, kb, s, cb, bh[cb], bl[cb]);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const u32x4 ah = A[(((kb * 2 + s) * 4 + rb) * 2) * 64], al = A[(((kb * 2 + s) * 4 + rb) * 2 + 1) * 64];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = mfma3(ah, al, bh[cb], bl[cb], out[rb][cb]);
            }
        }
    }
}

// output layer; the result is UNscaled
template <int NCB>
__device__ __forceinline__ void heads_layer_out_x3(f32x16 (&out)[NCB], const f32x16 (&in)[4][NCB], const float* arena, int head,
                                                   int lane) {
    const int half = lane >> 5;
    const f32x16 bf = scaled_bias(arena, head, 3, 0, half);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) out[cb] = bf;
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + QX_L1_VEC + QX_L23_VEC + (size_t)head * 4 * 2 * 2 * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 ah = A[((kb * 2 + s) * 2) * 64], al = A[((kb * 2 + s) * 2 + 1) * 64];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                u32x4 bh, bl;
                act_frag<NCB>(in, kb, s, cb, bh, bl);
                out[cb] = mfma3(ah, al, bh, bl, out[cb]);
            }
        }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[cb][r] *= QX_INV;
}

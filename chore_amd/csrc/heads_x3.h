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
// transposed fragments of the backward chain (d_prev = W^T d_cur), same k order
constexpr size_t QX_L4T_VEC = (size_t)HEAD_NUM * 4 * 2 * 64;           // [head][rb][plane][lane]      K = 16 outputs
constexpr size_t QX_L32T_VEC = (size_t)HEAD_NUM * 2 * 4 * 2 * 4 * 2 * 64;  // [head][j][kb][s][rb][plane][lane]  j=0: W3^T, 1: W2^T
constexpr size_t QX_L1T_VEC = (size_t)HEAD_NUM * 4 * 2 * QB_RB1 * 2 * 64;  // [head][kb][s][rb(11)][plane][lane]
constexpr size_t QX_OFF_L4T = QX_L1_VEC + QX_L23_VEC + QX_L4_VEC;       // in vectors from the start of the x3 region
constexpr size_t QX_OFF_L32T = QX_OFF_L4T + QX_L4T_VEC;
constexpr size_t QX_OFF_L1T = QX_OFF_L32T + QX_L32T_VEC;
constexpr size_t QX_OFF_BYTES = (QF_TOTAL_FLOATS * sizeof(float) + 255) / 256 * 256;   // after the fp32 fragments
constexpr size_t QX_TOTAL_VEC = QX_OFF_L1T + QX_L1T_VEC;
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

// The weight fragments come from the L2 (1.3 MB of them per tile, far beyond the L1): a k-step of 12 MFMAs lasts 0.16 us,
// an L2 round trip several times that, so the loads run QX_PF k-steps ahead of the MFMAs in a register ring.
constexpr int QX_PF = 3;
struct AFrag { u32x4 h[4], l[4]; };
__device__ __forceinline__ void load_afrag(AFrag& f, const u32x4* A /*[rb][plane][lane], lane applied*/) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) { f.h[rb] = A[(rb * 2) * 64]; f.l[rb] = A[(rb * 2 + 1) * 64]; }
}

// hidden layer 1: acc[rb][cb] = 2^s (b1 + W1 X^T)
template <int NCB, int PF = QX_PF>
__device__ __forceinline__ void heads_layer1_x3(f32x16 (&acc)[4][NCB], const float* X, const float* arena, int head, int lane) {
    static_assert(QX_KS1 % PF == 0, "the k loop is unrolled by the prefetch depth");
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const f32x16 bf = scaled_bias(arena, head, 0, rb, half);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = bf;
    }
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + (size_t)head * QX_KS1 * 4 * 2 * 64 + lane;
    const float* x0 = X + col * XS + 8 * half;
    AFrag ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) load_afrag(ring[p], A + (size_t)p * 4 * 2 * 64);
#pragma unroll 1
    for (int k0 = 0; k0 < QX_KS1; k0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int ks = k0 + p;
            u32x4 bh[NCB], bl[NCB];
            // the X rows are zero-filled up to QF_KPAD = 328: the upper half of the last k-step (k = 328..335) lies beyond
            const bool beyond = (ks == QX_KS1 - 1) && half;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float v[8];
                const float* src = x0 + cb * 32 * XS + (beyond ? 0 : ks * 16);       // never read past the row
                const f32x4 p4 = *(const f32x4*)src, q4 = *(const f32x4*)(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = beyond ? 0.f : p4[j]; v[4 + j] = beyond ? 0.f : q4[j]; }
                split8(v, bh[cb], bl[cb]);
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mfma3(ring[p].h[rb], ring[p].l[rb], bh[cb], bl[cb], acc[rb][cb]);
            const int kn = ks + PF < QX_KS1 ? ks + PF : QX_KS1 - 1;     // the tail reloads the last step (in bounds)
            load_afrag(ring[p], A + (size_t)kn * 4 * 2 * 64);
            if constexpr (PF > 1) __builtin_amdgcn_sched_barrier(0);      // or the scheduler sinks the loads to their use, three steps later
        }
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
template <int NCB, int PF = QX_PF>
__device__ __forceinline__ void heads_layer_hid_x3(f32x16 (&out)[4][NCB], const f32x16 (&in)[4][NCB], const float* arena, int head,
                                                   int layer /*1|2*/, int lane) {
    const int half = lane >> 5;
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + QX_L1_VEC +
                     ((size_t)head * 2 + (layer - 1)) * 4 * 2 * 4 * 2 * 64 + lane;
    AFrag ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) load_afrag(ring[p], A + (size_t)p * 4 * 2 * 64);
    if constexpr (PF > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const f32x16 bf = scaled_bias(arena, head, layer, rb, half);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = bf;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {          // k-step t = (kb, s)
        u32x4 bh[NCB], bl[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) act_frag<NCB>(in, t >> 1, t & 1, cb, bh[cb], bl[cb]);
        AFrag& f = ring[t % PF];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = mfma3(f.h[rb], f.l[rb], bh[cb], bl[cb], out[rb][cb]);
        if (t + PF < 8) load_afrag(f, A + (size_t)(t + PF) * 4 * 2 * 64);
        if constexpr (PF > 1) __builtin_amdgcn_sched_barrier(0);
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
    u32x4 ah[8], al[8];       // the 8 k-steps' fragments requested together (inline, every k-step waited for its own pair)
#pragma unroll
    for (int t = 0; t < 8; ++t) { ah[t] = A[(t * 2) * 64]; al[t] = A[(t * 2 + 1) * 64]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                u32x4 bh, bl;
                act_frag<NCB>(in, kb, s, cb, bh, bl);
                out[cb] = mfma3(ah[kb * 2 + s], al[kb * 2 + s], bh, bl, out[cb]);
            }
        }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[cb][r] *= QX_INV;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward chain.  Gradients have no fixed magnitude (a mean over 80 000 points puts them at 1e-5, where an fp16 hi / lo
// pair has a handful of bits), so every point's column carries its own power-of-two scale: chosen from max |dOut| of
// the column, renewed after every layer from the column's max, divided out exactly where the column leaves the chain.
// Columns are independent (one point each), so this changes nothing but the rounding.

// 2^(4 - floor(log2 m)): brings a column of max magnitude m to [16, 32).  1 for m = 0 / inf / nan.
__device__ __forceinline__ float col_scale_for(float m) {
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    if (m == 0.f || e == 128) return 1.f;
    e = e < -24 ? -24 : (e > 24 ? 24 : e);
    return __uint_as_float((unsigned)(127 + 4 - e) << 23);
}
// max |.| of a lane's D registers of one column block, over both halves of the wave (the 128 rows of the column)
template <int NCB>
__device__ __forceinline__ float col_absmax(const f32x16 (&f)[4][NCB], int cb) {
    float m = 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(f[rb][cb][r]));
    return fmaxf(m, __shfl_xor(m, 32, 64));
}
// B fragments of K-step s of block kb from gradient accumulators: in * mul (no relu)
template <int NCB>
__device__ __forceinline__ void grad_frag(const f32x16 (&in)[4][NCB], int kb, int s, int cb, float mul, u32x4& hi, u32x4& lo) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in[kb][cb][8 * s + j] * mul;
    split8(v, hi, lo);
}

// d3 = W4^T dOut.  g8[cb] = this lane's 8 output gradients (k = 8 half + j; the heads have at most 14 outputs), already
// multiplied by the column scale.  out carries 2^s x column scale.
template <int NCB>
__device__ __forceinline__ void bwd_out_x3(f32x16 (&out)[4][NCB], const float (&g8)[NCB][8], const float* arena, int head, int lane) {
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + QX_OFF_L4T + (size_t)head * 4 * 2 * 64 + lane;
    u32x4 bh[NCB], bl[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) split8(g8[cb], bh[cb], bl[cb]);
    AFrag f;
    load_afrag(f, A);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            out[rb][cb] = mfma3(f.h[rb], f.l[rb], bh[cb], bl[cb], z);
        }
    }
}

// d_prev = W^T d_cur for a 128x128 layer (which = 0: W3, 1: W2).  `in` carries 2^s x cs[cb]; the column scales are
// renewed (cs updated) and `out` carries 2^s x the new cs.
// The weight fragments run PF k-steps ahead in a register ring like the forward's (inline loads were sunk to their use by
// the scheduler: every row block of every k-step an L2 round trip of its own).
template <int NCB, int PF = QX_PF>
__device__ __forceinline__ void bwd_hid_x3(f32x16 (&out)[4][NCB], const f32x16 (&in)[4][NCB], float (&cs)[NCB], const float* arena,
                                           int head, int which, int lane) {
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + QX_OFF_L32T +
                     ((size_t)head * 2 + which) * 4 * 2 * 4 * 2 * 64 + lane;
    AFrag ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) load_afrag(ring[p], A + (size_t)p * 4 * 2 * 64);
    __builtin_amdgcn_sched_barrier(0);
    float mul[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const float f = col_scale_for(col_absmax<NCB>(in, cb) * QX_INV);
        mul[cb] = QX_INV * f;
        cs[cb] *= f;
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[rb][cb][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {          // k-step t = (kb, s)
        u32x4 bh[NCB], bl[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) grad_frag<NCB>(in, t >> 1, t & 1, cb, mul[cb], bh[cb], bl[cb]);
        AFrag& f = ring[t % PF];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = mfma3(f.h[rb], f.l[rb], bh[cb], bl[cb], out[rb][cb]);
        if (t + PF < 8) load_afrag(f, A + (size_t)(t + PF) * 4 * 2 * 64);
        // the products of this k-step are pinned here, in accumulation registers: otherwise they are sunk below the last
        // sched_barrier (nothing uses them before), all 64 fragment loads of the layer end up ahead of the first MFMA and
        // are parked, load by load and wait by wait, in AGPRs
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) asm volatile("" : "+a"(out[rb][cb]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one 32-row block rb of dX_head = W1^T d1 (K = 128).  bh / bl: the 8 K-step fragments of d1 (see bwd_l1_frags_x3).
// The caller requests block rb + 1's sixteen fragments (load_l1t_x3) before it multiplies block rb: the round trip runs
// under the MFMAs and the cross-head reduction of the block in hand.
struct L1TFrag { u32x4 h[8], l[8]; };
__device__ __forceinline__ void load_l1t_x3(L1TFrag& f, const float* arena, int head, int rb, int lane) {
    const u32x4* A = (const u32x4*)((const char*)arena + QX_OFF_BYTES) + QX_OFF_L1T + (size_t)head * 4 * 2 * QB_RB1 * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { f.h[ks] = A[((ks * QB_RB1 + rb) * 2) * 64]; f.l[ks] = A[((ks * QB_RB1 + rb) * 2 + 1) * 64]; }
}
template <int NCB>
__device__ __forceinline__ void bwd_l1_block_x3(f32x16 (&dx)[NCB], const u32x4 (&bh)[8][NCB], const u32x4 (&bl)[8][NCB],
                                                const L1TFrag& f) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[cb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) dx[cb] = mfma3(f.h[ks], f.l[ks], bh[ks][cb], bl[ks][cb], dx[cb]);
    }
}
template <int NCB>
__device__ __forceinline__ void bwd_l1_frags_x3(u32x4 (&bh)[8][NCB], u32x4 (&bl)[8][NCB], const f32x16 (&in)[4][NCB], float (&cs)[NCB]) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const float f = col_scale_for(col_absmax<NCB>(in, cb) * QX_INV);
        cs[cb] *= f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) grad_frag<NCB>(in, kb, s, cb, QX_INV * f, bh[kb * 2 + s][cb], bl[kb * 2 + s][cb]);
    }
}

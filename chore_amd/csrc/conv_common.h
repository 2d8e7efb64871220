// conv_common.h -- operand staging and matrix-core helpers shared by the convolution kernels (conv_lds.hip, conv_pc.hip):
// GroupNorm + ReLU applied while a tile is staged, the fp16 x 3 operand split, the MFMA wrappers and 8-channel
// vector loads / stores.  Device code only.
#pragma once
#include "enc_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

namespace conv_detail {

constexpr int KGC = 2;           // k-groups (one MFMA K each) per 32-channel chunk

template <typename T> struct CT;
template <> struct CT<float> { static constexpr int VE = 4, KGE = 8; };
template <> struct CT<bf16_t> { static constexpr int VE = 8, KGE = 16; };
template <> struct CT<x3_t> { static constexpr int VE = 8, KGE = 16; };
template <> struct CT<x3s_t> { static constexpr int VE = 8, KGE = 16; };   // VE: channels per staging slot (two 16-byte loads)


template <typename T>
__device__ __forceinline__ u32x4 xform(u32x4 raw, const float* sc, const float* sh, bool use_gn) {
    if (!use_gn) return raw;
    u32x4 o;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = fmaf(__uint_as_float(raw[j]), sc[j], sh[j]);
            o[j] = __float_as_uint(t > 0.f ? t : 0.f);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(raw[j] << 16), hi = __uint_as_float(raw[j] & 0xffff0000u);
            float x = fmaf(lo, sc[2 * j], sh[2 * j]), y = fmaf(hi, sc[2 * j + 1], sh[2 * j + 1]);
            x = x > 0.f ? x : 0.f;
            y = y > 0.f ? y : 0.f;
            o[j] = pack2bf(x, y);
        }
    }
    return o;
}

// fp16 x 3 staging: 8 fp32 channels (two vectors) -> GroupNorm + ReLU -> fp16 hi and lo vectors
// `mul` (a power of two; 1 leaves every value as it is): the operand scale of a gradient input (ConvArgs::in_amax), applied when
// no GroupNorm is fused
template <bool SCALED>
__device__ __forceinline__ void xform_x3_t(const u32x4& r0, const u32x4& r1, const float* sc, const float* sh, bool use_gn, u32x4& hi,
                                           u32x4& lo, float mul) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = __uint_as_float(r0[j]); t[4 + j] = __uint_as_float(r1[j]); }
    if (use_gn) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = fmaf(t[j], sc[j], sh[j]);
            t[j] = y > 0.f ? y : 0.f;
        }
    } else if constexpr (SCALED) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] *= mul;
    }
    f16x8_t h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = (_Float16)t[j];
        l[j] = (_Float16)(t[j] - (float)h[j]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

__device__ __forceinline__ void xform_x3(const u32x4& r0, const u32x4& r1, const float* sc, const float* sh, bool use_gn, u32x4& hi,
                                         u32x4& lo) {
    xform_x3_t<false>(r0, r1, sc, sh, use_gn, hi, lo, 1.f);
}
__device__ __forceinline__ void xform_x3(const u32x4& r0, const u32x4& r1, const float* sc, const float* sh, bool use_gn, u32x4& hi,
                                         u32x4& lo, float mul) {
    xform_x3_t<true>(r0, r1, sc, sh, use_gn, hi, lo, mul);
}

__device__ __forceinline__ void mfma_x3(f32x16& acc, const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl) {
    const f16x8_t a0 = __builtin_bit_cast(f16x8_t, ah), a1 = __builtin_bit_cast(f16x8_t, al);
    const f16x8_t b0 = __builtin_bit_cast(f16x8_t, bh), b1 = __builtin_bit_cast(f16x8_t, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void mfma(f32x16& acc, const u32x4& av, const u32x4& bw) {
    if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av),
                                                      __builtin_bit_cast(bf16x8_t, bw), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[i]), __uint_as_float(bw[i]), acc, 0, 0, 0);
    }
}

// 8 consecutive channels <-> registers
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const u32x4 a = *(const u32x4*)p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(a[j] << 16);
        v[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u);
    }
}
// store 8 channels; v is replaced by the values as stored (rounded to T)
template <typename T> __device__ __forceinline__ void store8(T* p, float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *(f32x4*)p = a;
    *(f32x4*)(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, float (&v)[8]) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = pack2bf(v[2 * j], v[2 * j + 1]);
        v[2 * j] = __uint_as_float(o[j] << 16);
        v[2 * j + 1] = __uint_as_float(o[j] & 0xffff0000u);
    }
    *(u32x4*)p = o;
}

template <> __device__ __forceinline__ void load8<h16_t>(const h16_t* p, float (&v)[8]) {
    const f16x8_t a = *(const f16x8_t*)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
}
template <> __device__ __forceinline__ void store8<h16_t>(h16_t* p, float (&v)[8]) {
    f16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = (_Float16)v[j]; v[j] = (float)o[j]; }
    *(f16x8_t*)p = o;
}

}  // namespace conv_detail

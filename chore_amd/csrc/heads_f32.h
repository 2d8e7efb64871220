// heads_f32.h -- LDS feature tile gather + exact-fp32 MFMA MLP heads, shared by the forward and
// the backward-to-points query kernels.  See query_fwd.hip for the design notes.
#pragma once
#include "query_common.h"

constexpr int XS = 332;  // LDS row stride of the X tile in floats (332/4 odd -> conflict-free b128)

template <typename T, int PTS = QT_PTS, int NW = 4>
__device__ __forceinline__ void gather_tile(float* X, const PtTableT<PTS>& tab, const T* feat_b,
                                            const T* tmpx_b, int wid, int lane) {
    using L = MapLoad<T>;
#pragma unroll 1
    for (int i = 0; i < PTS / NW; i += 4) {
        typename L::Raw4 fv[4][4];
        typename L::Raw1 tv[4][4];
        float fw[4][4], tw[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pt = wid * (PTS / NW) + i + u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int fo = tab.foff[k][pt];
                const int to = tab.toff[k][pt];
                fw[u][k] = tab.fw[k][pt];
                tw[u][k] = tab.tw[k][pt];
                fv[u][k] = (fo >= 0) ? L::raw4(feat_b + fo + lane * 4) : L::zero4();
                tv[u][k] = (to >= 0) ? L::raw1(tmpx_b + to + lane) : L::zero1();
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pt = wid * (PTS / NW) + i + u;
            f32x4 r;
            const f32x4 c0 = L::cvt4(fv[u][0]), c1 = L::cvt4(fv[u][1]), c2 = L::cvt4(fv[u][2]), c3 = L::cvt4(fv[u][3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = interp4(c0[c], c1[c], c2[c], c3[c], fw[u]);
            float* row = X + pt * XS;
            *(f32x4*)(row + lane * 4) = r;
            row[FEAT_C + 3 + lane] = interp4(L::cvt1(tv[u][0], lane & 1), L::cvt1(tv[u][1], lane & 1), L::cvt1(tv[u][2], lane & 1),
                                             L::cvt1(tv[u][3], lane & 1), tw[u]);
            if (lane < 3) row[FEAT_C + lane] = tab.xyz[lane][pt];
            if (lane >= 3 && lane < 3 + (QF_KPAD - HEAD_IN)) row[HEAD_IN + lane - 3] = 0.f;
        }
    }
}

__device__ __forceinline__ float relu(float v) { return v > 0.f ? v : 0.f; }

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// load the bias fragment (accumulator initialiser) of layer `layer`, row block rb
__device__ __forceinline__ f32x16 load_bias_frag(const float* arena, int head, int layer, int rb, int half) {
    const float* p = arena + QF_OFF_BIAS + ((((size_t)head * 4 + layer) * 4 + rb) * 2 + half) * 16;
    f32x16 r;
    const f32x4* p4 = (const f32x4*)p;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = p4[q];
        r[q * 4 + 0] = v[0]; r[q * 4 + 1] = v[1]; r[q * 4 + 2] = v[2]; r[q * 4 + 3] = v[3];
    }
    return r;
}

// hidden layer 1: acc[rb][cb] = b1 + W1 * X^T     (K = 328 in 41 groups of 8)
template <int NCB>
__device__ __forceinline__ void heads_layer1(f32x16 (&acc)[4][NCB], const float* X, const float* arena,
                                             int head, int lane) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const f32x16 bf = load_bias_frag(arena, head, 0, rb, half);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = bf;
    }
    const f32x4* A = (const f32x4*)(arena + QF_OFF_L1) + ((size_t)head * QF_KG * 4) * 64 + lane;
    const float* x0 = X + col * XS + 4 * half;
    f32x4 a_cur[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) a_cur[rb] = A[rb * 64];
#pragma unroll 1
    for (int q = 0; q < QF_KG; ++q) {
        f32x4 a_nxt[4];
        const int qn = (q + 1 < QF_KG) ? q + 1 : q;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a_nxt[rb] = A[(qn * 4 + rb) * 64];
        f32x4 xb[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) xb[cb] = *(const f32x4*)(x0 + cb * 32 * XS + q * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = MFMA_F32(a_cur[rb][i], xb[cb][i], acc[rb][cb]);
            }
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a_cur[rb] = a_nxt[rb];
    }
}

// hidden layers 2 and 3: out = b + W * relu(in), activations stay in registers
template <int NCB>
__device__ __forceinline__ void heads_layer_hid(f32x16 (&out)[4][NCB], const f32x16 (&in)[4][NCB],
                                                const float* arena, int head, int layer /*1|2*/, int lane) {
    const int half = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const f32x16 bf = load_bias_frag(arena, head, layer, rb, half);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = bf;
    }
    const f32x4* A = (const f32x4*)(arena + QF_OFF_L23) +
                     (((size_t)head * 2 + (layer - 1)) * 16 * 4) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 a[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) a[rb] = A[((kb * 4 + rg) * 4 + rb) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float bv[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) bv[cb] = relu(in[kb][cb][rg * 4 + i]);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) out[rb][cb] = MFMA_F32(a[rb][i], bv[cb], out[rb][cb]);
                }
            }
        }
    }
}

// output layer: out[cb] = b4 + W4 * relu(in)   (rows >= out_dim are zero padding)
template <int NCB>
__device__ __forceinline__ void heads_layer_out(f32x16 (&out)[NCB], const f32x16 (&in)[4][NCB],
                                                const float* arena, int head, int lane) {
    const int half = lane >> 5;
    const f32x16 bf = load_bias_frag(arena, head, 3, 0, half);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) out[cb] = bf;
    const f32x4* A = (const f32x4*)(arena + QF_OFF_L4) + ((size_t)head * 16) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 a = A[(kb * 4 + rg) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) out[cb] = MFMA_F32(a[i], relu(in[kb][cb][rg * 4 + i]), out[cb]);
            }
        }
    }
}

// training staging: one wave stores its head's 128 x (32 NCB) activation (or gradient) tile as [point][channel] rows;
// pt0 = first point of this wave's column blocks inside the workgroup tile.
// A D fragment holds, per lane, 4 runs of 4 consecutive channels of one point: four 16-byte stores per fragment.
// mul: per column block factor applied on the way out (the fp16 x 3 chains keep scaled accumulators), nullptr = none
template <int NCB>
__device__ __forceinline__ void store_tile(float* base /*[B*N][128], this head*/, const f32x16 (&f)[4][NCB], bool relu_it,
                                           size_t row0, int n0, int N, int lane, int pt0 = 0, const float* mul = nullptr) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int pt = pt0 + cb * 32 + col;
        if (n0 + pt >= N) continue;
        float* row = base + (row0 + pt) * HEAD_HID;
        const float m = mul ? mul[cb] : 1.f;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = {f[rb][cb][4 * j], f[rb][cb][4 * j + 1], f[rb][cb][4 * j + 2], f[rb][cb][4 * j + 3]};
                if (mul) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= m;
                }
                if (relu_it) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                *(f32x4*)(row + rb * 32 + 8 * j + 4 * half) = v;
            }
    }
}

// training staging, ReLU sign bits: the backward needs only the signs of the hidden activations, and reading them back
// from the staged fp32 rows costs 512 B per (point, layer, head); the forward also leaves them as 128 bits.
// Layout [B*N][2] u64 per (layer, head) plane: entry (point, half) holds, for rb = 0..3, bit 16 rb + r = [row
// rb*32 + mfma32_row(r, half) of the point is positive] -- the bits a lane of a D fragment owns, in register order.
template <int NCB>
__device__ __forceinline__ void store_masks(unsigned long long* base, const f32x16 (&f)[4][NCB], size_t row0, int n0, int N, int lane,
                                            int pt0 = 0) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int pt = pt0 + cb * 32 + col;
        unsigned long long v = 0ull;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            unsigned m = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) m |= (f[rb][cb][r] > 0.f ? 1u : 0u) << r;
            v |= (unsigned long long)m << (16 * rb);
        }
        if (n0 + pt < N) base[(row0 + pt) * 2 + half] = v;
    }
}
// -> the backward's mask registers: bit 16 cb + r of m[rb]
template <int NCB>
__device__ __forceinline__ void load_mask_bits(unsigned (&m)[4], const unsigned long long* base, size_t row0, int n0, int N, int lane,
                                               int pt0) {
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) m[rb] = 0u;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int pt = pt0 + cb * 32 + col;
        const unsigned long long v = n0 + pt < N ? base[(row0 + pt) * 2 + half] : 0ull;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) m[rb] |= (unsigned)((v >> (16 * rb)) & 0xffffull) << (16 * cb);
    }
}

// store_tile through a wave-private LDS tile: the D-fragment layout gives every store instruction 32 points x 32 bytes
// (two lanes per point), i.e. 32-byte segments 512 bytes apart; transposed through [32 points][ST_LD] floats a store
// instruction covers 8 points x 128 contiguous bytes.  scr: 32 * ST_LD floats owned by the calling wave.
constexpr int ST_LD = 36;
template <int NCB>
__device__ __forceinline__ void store_tile_lds(float* base, const f32x16 (&f)[4][NCB], bool relu_it, size_t row0, int n0, int N,
                                               int lane, int pt0, const float* mul, float* scr) {
    const int half = lane >> 5, col = lane & 31, pl = lane >> 3, c4 = lane & 7;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const float m = mul ? mul[cb] : 1.f;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = {f[rb][cb][4 * j], f[rb][cb][4 * j + 1], f[rb][cb][4 * j + 2], f[rb][cb][4 * j + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (mul) v[e] *= m;
                    if (relu_it) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                *(f32x4*)(scr + col * ST_LD + 8 * j + 4 * half) = v;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = pl + 8 * i, pt = pt0 + cb * 32 + p;
                const f32x4 v = *(const f32x4*)(scr + p * ST_LD + 4 * c4);
                if (n0 + pt < N) *(f32x4*)(base + (row0 + pt) * HEAD_HID + rb * 32 + 4 * c4) = v;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

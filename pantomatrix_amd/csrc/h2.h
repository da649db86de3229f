// EMAGE_H2 — the PRE-SPLIT activation / weight storage of the split-fp16 MFMA mode.
//
// A logical fp32 value x is stored as two fp16 numbers, x * s = hi + lo (hi = rne_f16(x*s), lo = rne_f16(x*s - hi); the
// difference is exact in fp32, so |x*s - hi - lo| <= 2^-23 |x*s|), s a power of two (activations: H2_SCALE = 16; weights: a
// per-tensor scale chosen by the host).  Layout: every group of 8 consecutive logical columns occupies 32 bytes =
// [8 x fp16 hi | 8 x fp16 lo] — the same 4 bytes per element, the same row strides and the same 32-byte granule as 8
// float32 columns, so an H2 tensor is addressed exactly like the float32 tensor of the same shape (any 8-aligned column
// block is a view).  A 32-k K-tile of a row is 128 bytes = 4 such groups; the GEMM's LDS-DMA lands it as
// [4 hi chunks | 4 lo chunks], i.e. both MFMA operands of v_mfma_f32_16x16x32_f16 arrive ready — no VALU in the K-loop.
// Producers (GEMM epilogues, LayerNorm, attention, adds, gathers) write this format directly.
#pragma once
#include "common.h"

namespace emage_dev {

struct h2_t { unsigned raw; };               // storage tag: 4 bytes per logical element
typedef _Float16 h2f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2f16x4 __attribute__((ext_vector_type(4)));
constexpr float H2_SCALE = 16.0f, H2_INV = 0.0625f;      // activations: |x| < 4094 stays finite

// The activation scale of a MODEL is a power of two 2^(4 - k) carried in the dtype code of every entry point that reads or writes activation
// images (include/emage_hip.h: EMAGE_H2_SHIFT(k); k = 0 — the default, these constants — keeps 16): a checkpoint whose activations pass
// 4094 runs with k > 0 instead of overflowing the hi plane (|x| < 4094 * 2^k), at the price of the smallest values' lo bits.  `s` / `inv`
// below are that scale and its inverse; the defaults are the k = 0 image (weights, gradients: their scales ride in front, h2_cast).
struct H2Scale { float s, inv; };
__host__ __device__ inline H2Scale h2_scale_of(int dtype) {
    const int k = (dtype >> 8) & 0xff;
    H2Scale r;
    r.s = __builtin_ldexpf(H2_SCALE, -k);
    r.inv = __builtin_ldexpf(H2_INV, k);
    return r;
}
// entry points: take the shift off a dtype code (-> the plain EMAGE_* code) and return the image scale it stands for; non-zero = invalid code
inline int h2_dtype(int& dtype, H2Scale& hs) {
    const int k = dtype >> 8;
    if (k && ((dtype & 0xff) != EMAGE_H2 || k < 0 || k > 12)) return -1;
    hs = h2_scale_of(dtype);
    dtype &= 0xff;
    return 0;
}

__device__ __forceinline__ void h2_split8(const float (&v)[8], uint4& hi, uint4& lo, float s = H2_SCALE) {
    h2f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xs = v[e] * s;
        const _Float16 hh = (_Float16)xs;
        h[e] = hh;
        l[e] = (_Float16)(xs - (float)hh);
    }
    hi = __builtin_bit_cast(uint4, h);
    lo = __builtin_bit_cast(uint4, l);
}
__device__ __forceinline__ void h2_join8(const uint4& hi, const uint4& lo, float (&v)[8], float inv = H2_INV) {
    const h2f16x8 h = __builtin_bit_cast(h2f16x8, hi), l = __builtin_bit_cast(h2f16x8, lo);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((float)h[e] + (float)l[e]) * inv;
}
__device__ __forceinline__ void h2_split4(const float (&v)[4], uint2& hi, uint2& lo, float s = H2_SCALE) {
    h2f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xs = v[e] * s;
        const _Float16 hh = (_Float16)xs;
        h[e] = hh;
        l[e] = (_Float16)(xs - (float)hh);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}
__device__ __forceinline__ void h2_join4(const uint2& hi, const uint2& lo, float (&v)[4], float inv = H2_INV) {
    const h2f16x4 h = __builtin_bit_cast(h2f16x4, hi), l = __builtin_bit_cast(h2f16x4, lo);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ((float)h[e] + (float)l[e]) * inv;
}

// 8 logical columns at an 8-aligned column: p points at the group's 32 bytes (row + col, in 4-byte units)
__device__ __forceinline__ void h2_load8(const h2_t* p, float (&v)[8], float inv = H2_INV) {
    const uint4 hi = *(const uint4*)p, lo = *(const uint4*)(p + 4);
    h2_join8(hi, lo, v, inv);
}
__device__ __forceinline__ void h2_store8(h2_t* p, const float (&v)[8], float s = H2_SCALE) {
    uint4 hi, lo;
    h2_split8(v, hi, lo, s);
    *(uint4*)p = hi;
    *(uint4*)(p + 4) = lo;
}
// 4 logical columns at a 4-aligned column c: p = row + c (4-byte units); the group starts at c & ~7, half = (c >> 2) & 1
__device__ __forceinline__ void h2_load4(const h2_t* p, int c, float (&v)[4], float inv = H2_INV) {
    const unsigned char* g = (const unsigned char*)(p - (c & 7));
    const int half = (c >> 2) & 1;
    h2_join4(*(const uint2*)(g + half * 8), *(const uint2*)(g + 16 + half * 8), v, inv);
}
__device__ __forceinline__ void h2_store4(h2_t* p, int c, const float (&v)[4], float s = H2_SCALE) {
    unsigned char* g = (unsigned char*)(p - (c & 7));
    const int half = (c >> 2) & 1;
    uint2 hi, lo;
    h2_split4(v, hi, lo, s);
    *(uint2*)(g + half * 8) = hi;
    *(uint2*)(g + 16 + half * 8) = lo;
}

}  // namespace emage_dev

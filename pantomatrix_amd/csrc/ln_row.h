// One-row LayerNorm of emage_layernorm (elementwise.hip): one wave per row, float32 statistics, optional H2 image of the result.
#pragma once
#include "common.h"
#include "h2.h"
#include <math.h>

namespace emage_dev {

// 4 consecutive elements of T as floats (16-byte fp32 / 8-byte bf16 access)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    __device__ static __forceinline__ float4 load(const float* p) { return *(const float4*)p; }
    __device__ static __forceinline__ void store(float* p, float4 v) { *(float4*)p = v; }
};
template <> struct Vec4<bf16_t> {
    __device__ static __forceinline__ float4 load(const bf16_t* p) {
        const uint2 t = *(const uint2*)p;
        return make_float4(__builtin_bit_cast(float, t.x << 16), __builtin_bit_cast(float, t.x & 0xffff0000u),
                           __builtin_bit_cast(float, t.y << 16), __builtin_bit_cast(float, t.y & 0xffff0000u));
    }
    __device__ static __forceinline__ void store(bf16_t* p, float4 v) {
        uint2 t;
        t.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
        t.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
        *(uint2*)p = t;
    }
};

// LayerNorm of one row by one wave64: each lane owns 4-element groups j = lane + 64*i (C % 4 == 0, C <= 256*MAXV).  The
// row (the residual stream, stored in the compute dtype) is read once, statistics and the affine map are fp32.
// Split in two so that a caller handling several rows can have all their loads in flight before the first reduction.
template <typename T, int MAXV>
__device__ __forceinline__ void layernorm_row_load(const T* __restrict__ xp, int C, int lane, float4 (&v)[MAXV]) {
    const int nv = C >> 2;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + 64 * i;
        v[i] = j < nv ? Vec4<T>::load(xp + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <typename T, int MAXV>
__device__ __forceinline__ void layernorm_row_finish(const float4 (&v)[MAXV], const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, const T* __restrict__ addp, float* __restrict__ yfp, T* __restrict__ yp,
                                                     int C, int lane, h2_t* __restrict__ yhp = nullptr, float h2s = H2_SCALE) {
    const int nv = C >> 2;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    const float4* gp = (const float4*)gamma;
    const float4* bp = (const float4*)beta;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) {
            const float4 g = gp[j], b = bp[j];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (addp) { const float4 a = Vec4<T>::load(addp + 4 * j); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
            if (yfp) ((float4*)yfp)[j] = o;
            if (yp) Vec4<T>::store(yp + 4 * j, o);
            if (yhp) { const float o4[4] = {o.x, o.y, o.z, o.w}; h2_store4(yhp + 4 * j, 4 * j, o4, h2s); }   // EMAGE_H2 copy (csrc/h2.h): half a group per lane
        }
    }
}

template <typename T, int MAXV>
__device__ __forceinline__ void layernorm_row(const T* __restrict__ xp, const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float eps, const T* __restrict__ addp, float* __restrict__ yfp, T* __restrict__ yp,
                                              int C, int lane, h2_t* __restrict__ yhp = nullptr, float h2s = H2_SCALE) {
    float4 v[MAXV];
    layernorm_row_load<T, MAXV>(xp, C, lane, v);
    layernorm_row_finish<T, MAXV>(v, gamma, beta, eps, addp, yfp, yp, C, lane, yhp, h2s);
}

}  // namespace emage_dev

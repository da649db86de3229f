// Shared device helpers for the EMAGE gfx950 kernels (wave64, MFMA 16x16, 16-byte operand chunks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/emage_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;   // storage type of a bfloat16 element

// float -> bfloat16, round-to-nearest-even (what torch's .to(torch.bfloat16) does).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
    return __builtin_bit_cast(float, ((unsigned)h) << 16);
}

// Element traits.  A "chunk" is the 16-byte MFMA operand unit a lane loads: 8 bf16 or 4 fp32.
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int EPC = 4;             // elements per 16-byte chunk
    __device__ static __forceinline__ float to(float v) { return v; }
    __device__ static __forceinline__ float from(float v) { return v; }
    // acc += A(16 x 4*1) * B: four k-steps of v_mfma_f32_16x16x4_f32, one per chunk element.
    __device__ static __forceinline__ f32x4 mma(const uint4& a, const uint4& b, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
        return acc;
    }
};
template <> struct Elem<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static __forceinline__ bf16_t to(float v) { return f32_to_bf16(v); }
    __device__ static __forceinline__ float from(bf16_t v) { return bf16_to_f32(v); }
    // one v_mfma_f32_16x16x32_bf16: the chunk is 8 consecutive k of one row/column.
    __device__ static __forceinline__ f32x4 mma(const uint4& a, const uint4& b, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ float leaky(float v, float s) { return v > 0.f ? v : v * s; }

// CU count of the CURRENT device (cached per device id; 256 on MI355X).  Steers tile / split-K heuristics and residency checks only
static inline int device_cus() {
    static int cached[64] = {0};
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64 && cached[dev] > 0) return cached[dev];
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) cached[dev] = v;
    return v;
}

static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

// Rotation conversions shared by motion.hip (EMAGE decode merge) and lstm.hip (DisCo / CaMN outputs): rot-6D <-> axis-angle
// in the reference's fp32 operation order (P:6-104 of models/emage_audio/processing_emage_audio.py; the DisCo / CaMN
// model files carry verbatim copies of the same helpers, D:30-79).  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace emage_rot {

__device__ __forceinline__ float sqrt_pos(float x) { return x > 0.f ? sqrtf(x) : 0.f; }          // P:10-14
__device__ __forceinline__ float copysign_ref(float a, float b) { return ((a < 0.f) != (b < 0.f)) ? -a : a; }   // P:6-8
__device__ __forceinline__ float sin_half_over_angle(float angle, float half) {                   // P:35-43, 66-74
    return fabsf(angle) < 1e-6f ? 0.5f - (angle * angle) / 48.f : sinf(half) / angle;
}
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {                        // F.normalize, eps 1e-12
    const float n = fmaxf(sqrtf((x * x + y * y) + z * z), 1e-12f);
    x /= n; y /= n; z /= n;
}

// rotation_6d_to_axis_angle, P:50-59 + P:16-44
__device__ __forceinline__ void rot6d_to_aa(const float* d6, float* aa) {
    float b1x = d6[0], b1y = d6[1], b1z = d6[2];
    const float a2x = d6[3], a2y = d6[4], a2z = d6[5];
    normalize3(b1x, b1y, b1z);
    const float dot = (b1x * a2x + b1y * a2y) + b1z * a2z;
    float b2x = a2x - dot * b1x, b2y = a2y - dot * b1y, b2z = a2z - dot * b1z;
    normalize3(b2x, b2y, b2z);
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    const float m00 = b1x, m01 = b1y, m02 = b1z, m10 = b2x, m11 = b2y, m12 = b2z, m20 = b3x, m21 = b3y, m22 = b3z;
    const float w = 0.5f * sqrt_pos(((1.f + m00) + m11) + m22);
    float x = 0.5f * sqrt_pos(((1.f + m00) - m11) - m22);
    float y = 0.5f * sqrt_pos(((1.f - m00) + m11) - m22);
    float z = 0.5f * sqrt_pos(((1.f - m00) - m11) + m22);
    x = copysign_ref(x, m21 - m12);
    y = copysign_ref(y, m02 - m20);
    z = copysign_ref(z, m10 - m01);
    const float nrm = sqrtf((x * x + y * y) + z * z);
    const float half = atan2f(nrm, w);
    const float s = sin_half_over_angle(2.f * half, half);
    aa[0] = x / s; aa[1] = y / s; aa[2] = z / s;
}

// axis_angle_to_rotation_6d, P:64-104
__device__ __forceinline__ void aa_to_rot6d(const float* aa, float* d6) {
    const float ax = aa[0], ay = aa[1], az = aa[2];
    const float angle = sqrtf((ax * ax + ay * ay) + az * az);
    const float half = 0.5f * angle;
    const float s = sin_half_over_angle(angle, half);
    const float r = cosf(half), i = ax * s, j = ay * s, k = az * s;
    const float two_s = 2.0f / (((r * r + i * i) + j * j) + k * k);
    d6[0] = 1.f - two_s * (j * j + k * k);
    d6[1] = two_s * (i * j - k * r);
    d6[2] = two_s * (i * k + j * r);
    d6[3] = two_s * (i * j + k * r);
    d6[4] = 1.f - two_s * (i * i + k * k);
    d6[5] = two_s * (j * k - i * r);
}

}  // namespace emage_rot

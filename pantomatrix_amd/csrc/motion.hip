// SMPL-X motion glue of EmageVQModel.decode: rot-6D <-> axis-angle (K9), the 55-joint merge (K9+K11)
// and the root-translation scan (K10).  fp32 elementwise math in the reference's operation order
// (compiled with -ffp-contract=off), no host synchronisation (the reference's boolean-mask indexing and
// .item() calls force ~60 syncs per decode on a GPU).
#include "common.h"
#include <math.h>
#include "rot_math.h"

namespace {

using namespace emage_rot;

__global__ void rot6d_to_aa_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d6[6], aa[3];
    for (int c = 0; c < 6; ++c) d6[c] = in[(long)i * 6 + c];
    rot6d_to_aa(d6, aa);
    for (int c = 0; c < 3; ++c) out[(long)i * 3 + c] = aa[c];
}
__global__ void aa_to_rot6d_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d6[6], aa[3];
    for (int c = 0; c < 3; ++c) aa[c] = in[(long)i * 3 + c];
    aa_to_rot6d(aa, d6);
    for (int c = 0; c < 6; ++c) out[(long)i * 6 + c] = d6[c];
}

// joint -> (part, slot) with part 0 none, 1 upper, 2 hands, 3 lower, 4 jaw  (M:75-90,181,185)
__constant__ signed char c_part[55] = {
    3, 3, 3, 1, 3, 3, 1, 3, 3, 1, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 4, 0, 0,
    2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2};
__constant__ signed char c_slot[55] = {
    0, 1, 2, 0, 3, 4, 1, 5, 6, 2, 7, 8, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, 0, 0,
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29};

// block = one frame (row m): threads 0..54 joints, 64..70 trans/contact, 96..195 expression
__global__ __launch_bounds__(256) void merge_parts_kernel(const float* __restrict__ face, int ldface, const float* __restrict__ upper, int ldup,
                                                          const float* __restrict__ hands, int ldh, const float* __restrict__ lower, int ldlow,
                                                          float* __restrict__ aa_out, float* __restrict__ motion, float* __restrict__ expr, int M) {
    const int m = blockIdx.x, t = threadIdx.x;
    if (t < 55) {
        const int part = c_part[t], slot = c_slot[t];
        const float* src = nullptr;
        if (part == 1 && upper) src = upper + (long)m * ldup + slot * 6;
        else if (part == 2 && hands) src = hands + (long)m * ldh + slot * 6;
        else if (part == 3 && lower) src = lower + (long)m * ldlow + slot * 6;
        else if (part == 4 && face) src = face + (long)m * ldface;
        float aa[3] = {0.f, 0.f, 0.f}, d6[6];
        if (src) {
            for (int c = 0; c < 6; ++c) d6[c] = src[c];
            rot6d_to_aa(d6, aa);
        }
        if (aa_out) for (int c = 0; c < 3; ++c) aa_out[(long)m * 165 + t * 3 + c] = aa[c];
        if (motion) {
            aa_to_rot6d(aa, d6);
            for (int c = 0; c < 6; ++c) motion[(long)m * 337 + t * 6 + c] = d6[c];
        }
    } else if (t >= 64 && t < 71) {
        if (motion) motion[(long)m * 337 + 330 + (t - 64)] = lower ? lower[(long)m * ldlow + 54 + (t - 64)] : 0.f;
    } else if (t >= 96 && t < 196) {
        if (expr) expr[(long)m * 100 + (t - 96)] = face ? face[(long)m * ldface + 6 + (t - 96)] : 0.f;
    }
}

// one thread per (batch, axis): x and z integrate, y copies  (P:107-115, M:195-205)
__global__ void velocity_scan_kernel(const float* __restrict__ vel, int ldv, int col0, const float* __restrict__ init, int ld_init, float dt,
                                     float* __restrict__ trans, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 3) return;
    const int b = i / 3, ax = i - b * 3;
    const float* v = vel + (long)b * T * ldv + col0 + ax;
    float* o = trans + (long)b * T * 3 + ax;
    if (ax == 1) {
        for (int t = 0; t < T; ++t) o[(long)t * 3] = v[(long)t * ldv];
    } else {
        float pos = init[(long)b * ld_init + ax];
        o[0] = pos;
        for (int t = 1; t < T; ++t) {
            pos = __fadd_rn(__fmul_rn(v[(long)(t - 1) * ldv], dt), pos);
            o[(long)t * 3] = pos;
        }
    }
}

// Same recurrence with the velocities staged in LDS first: the scan itself is inherently sequential (the reference's
// fp32 rounding order is part of the contract), but its T dependent steps should not each wait on a global load.
__global__ __launch_bounds__(128) void velocity_scan_lds_kernel(const float* __restrict__ vel, int ldv, int col0, const float* __restrict__ init, int ld_init,
                                                                float dt, float* __restrict__ trans, int T) {
    extern __shared__ float s_v[];                     // [3][T]
    const int b = blockIdx.x;
    const float* v = vel + (long)b * T * ldv + col0;
    for (int i = threadIdx.x; i < 3 * T; i += blockDim.x) {
        const int t = i / 3, ax = i - t * 3;
        s_v[ax * T + t] = v[(long)t * ldv + ax];
    }
    __syncthreads();
    float* o = trans + (long)b * T * 3;
    if (threadIdx.x < 2) {                             // x (axis 0) and z (axis 2) integrate
        const int ax = threadIdx.x * 2;
        float pos = init[(long)b * ld_init + ax];
        float* s_p = s_v + ax * T;
        float prev = s_p[0];
        s_p[0] = pos;
        for (int t = 1; t < T; ++t) {
            pos = __fadd_rn(__fmul_rn(prev, dt), pos);
            prev = s_p[t];
            s_p[t] = pos;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * T; i += blockDim.x) {
        const int t = i / 3, ax = i - t * 3;
        o[i] = s_v[ax * T + t];                        // y (axis 1) is the decoded height, copied through
    }
}

}  // namespace

extern "C" int emage_rot6d_to_axis_angle(const float* rot6d, float* aa, int n, void* stream) {
    if (!rot6d || !aa || n <= 0) return EMAGE_EINVAL;
    hipLaunchKernelGGL(rot6d_to_aa_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rot6d, aa, n);
    return launch_status();
}
extern "C" int emage_axis_angle_to_rot6d(const float* aa, float* rot6d, int n, void* stream) {
    if (!rot6d || !aa || n <= 0) return EMAGE_EINVAL;
    hipLaunchKernelGGL(aa_to_rot6d_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, aa, rot6d, n);
    return launch_status();
}
extern "C" int emage_merge_parts(const float* face, int ldface, const float* upper, int ldup, const float* hands, int ldh,
                                 const float* lower, int ldlow, float* axis_angle, float* motion, float* expression,
                                 int M, void* stream) {
    if (M <= 0 || (!axis_angle && !motion && !expression)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(merge_parts_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, face, ldface, upper, ldup, hands, ldh, lower, ldlow,
                       axis_angle, motion, expression, M);
    return launch_status();
}
extern "C" int emage_velocity_to_position(const float* vel, int ldv, int col0, const float* init, int ld_init, float dt,
                                          float* trans, int B, int T, void* stream) {
    if (!vel || !init || !trans || B <= 0 || T <= 0 || ld_init < 0) return EMAGE_EINVAL;
    if ((size_t)3 * T * sizeof(float) <= 60 * 1024)
        hipLaunchKernelGGL(velocity_scan_lds_kernel, dim3(B), dim3(128), (size_t)3 * T * sizeof(float), (hipStream_t)stream, vel, ldv, col0, init, ld_init, dt, trans, T);
    else
        hipLaunchKernelGGL(velocity_scan_kernel, dim3((B * 3 + 63) / 64), dim3(64), 0, (hipStream_t)stream, vel, ldv, col0, init, ld_init, dt, trans, B, T);
    return launch_status();
}

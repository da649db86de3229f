// emage_attention — softmax(Q K^T / sqrt(hd)) V for the EMAGE transformer layers: Tq, Tk <= 128,
// hd = 192, no mask.  One wave64 per (batch, head, 16-query tile); no LDS at all:
//   * S^T = K Q^T is computed "swapped" (A operand = K rows, B operand = Q rows), so a lane holds, for
//     its query column q = lane&15, the scores of keys {16*nt + 4*(lane>>4) + r}: the softmax reduction
//     is in-lane plus two cross-lane steps (xor 16, 32), and the probabilities are already laid out as
//     the A operand of the P V product (row = query, contraction = key);
//   * V arrives transposed (emitted by emage_gemm's out_t path), so its B-operand chunks are
//     contiguous; Q, K, V^T chunks are loaded straight from L2 — each (b,h) problem is 72 KB.
// The key order inside a P chunk is {16*nt+4g+r} (not 8 consecutive keys); V^T is gathered with the
// same mapping, and MFMA sums over the contraction index, so any consistent bijection is exact.
#include "common.h"
#include <math.h>

namespace {

struct AttnArgs {
    const void* q; const void* k; const void* vt; void* out;
    int ldq, ldk, ldvt, vt_rows, ldo, B, H, Tq, Tk;
    float scale;
};

template <typename T, int HD, int NT>
__global__ __launch_bounds__(64) void attn_kernel(AttnArgs p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int NSTEP = HD / (4 * EPC);     // contraction steps over head_dim (4 chunks per step)
    constexpr int NDT = HD / 16;              // output d tiles
    const int lane = threadIdx.x;
    const int fr = lane & 15, fg = lane >> 4;
    const int qtiles = (p.Tq + 15) >> 4;
    int bid = blockIdx.x;
    const int qt = bid % qtiles; bid /= qtiles;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int q0 = qt * 16;

    const T* __restrict__ Q = (const T*)p.q;
    const T* __restrict__ K = (const T*)p.k;
    const T* __restrict__ VT = (const T*)p.vt;

    // Q fragments (B operand: column j = query)
    const int qrow = min(q0 + fr, p.Tq - 1);
    const T* qp = Q + ((long)b * p.Tq + qrow) * p.ldq + h * HD + fg * EPC;
    uint4 qf[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) qf[s] = *(const uint4*)(qp + s * 4 * EPC);

    // scores S^T[key][q].  The whole problem is latency-bound (72 KB per (b,h), one wave per 16 queries), so for
    // Tk <= 64 every K chunk is requested before the first MFMA instead of tile by tile.
    f32x4 sc[NT];
    if constexpr (NT <= 4) {
        uint4 kf[NT][NSTEP];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int krow = min(nt * 16 + fr, p.Tk - 1);
            const T* kp = K + ((long)b * p.Tk + krow) * p.ldk + h * HD + fg * EPC;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) kf[nt][s] = *(const uint4*)(kp + s * 4 * EPC);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) acc = Elem<T>::mma(kf[nt][s], qf[s], acc);
            sc[nt] = acc;
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int krow = min(nt * 16 + fr, p.Tk - 1);
            const T* kp = K + ((long)b * p.Tk + krow) * p.ldk + h * HD + fg * EPC;
            uint4 kf[NSTEP];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) kf[s] = *(const uint4*)(kp + s * 4 * EPC);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) acc = Elem<T>::mma(kf[s], qf[s], acc);
            sc[nt] = acc;
        }
    }

    // V^T chunks are requested now (independent of the softmax below), consumed by the P V product afterwards
    constexpr int NPC = (EPC == 8) ? (NT + 1) / 2 : NT;
    constexpr bool PREV = NT <= 4;
    uint4 vpre[PREV ? NDT : 1][PREV ? NPC : 1];
    if constexpr (PREV) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const T* vp = VT + ((long)b * p.vt_rows + h * HD + dt * 16 + fr) * p.ldvt + fg * 4;
#pragma unroll
            for (int c = 0; c < NPC; ++c) {
                if constexpr (EPC == 8) {
                    const uint2 lo = *(const uint2*)(vp + c * 32);
                    const uint2 hi = *(const uint2*)(vp + c * 32 + 16);
                    vpre[dt][c] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    vpre[dt][c] = *(const uint4*)(vp + c * 16);
                }
            }
        }
    }

    // softmax over keys for query column fr
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = nt * 16 + fg * 4 + r;
            const float v = key < p.Tk ? sc[nt][r] * p.scale : -INFINITY;
            sc[nt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = expf(sc[nt][r] - mx);
            sc[nt][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[nt][r] = sc[nt][r] / sum;

    // P chunks (A operand: row i = query fr, contraction = keys)
    uint4 pc[NPC];
    if constexpr (EPC == 8) {
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            const f32x4 lo = sc[2 * c];
            f32x4 hi = {0.f, 0.f, 0.f, 0.f};
            if (2 * c + 1 < NT) hi = sc[2 * c + 1];
            pc[c].x = (unsigned)f32_to_bf16(lo[0]) | ((unsigned)f32_to_bf16(lo[1]) << 16);
            pc[c].y = (unsigned)f32_to_bf16(lo[2]) | ((unsigned)f32_to_bf16(lo[3]) << 16);
            pc[c].z = (unsigned)f32_to_bf16(hi[0]) | ((unsigned)f32_to_bf16(hi[1]) << 16);
            pc[c].w = (unsigned)f32_to_bf16(hi[2]) | ((unsigned)f32_to_bf16(hi[3]) << 16);
        }
    } else {
#pragma unroll
        for (int c = 0; c < NPC; ++c) pc[c] = __builtin_bit_cast(uint4, sc[c]);
    }

    // O = P V, one 16-wide d tile at a time
    T* __restrict__ O = (T*)p.out;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        const T* vp = VT + ((long)b * p.vt_rows + h * HD + dt * 16 + fr) * p.ldvt + fg * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            uint4 vf;
            if constexpr (PREV) {
                vf = vpre[dt][c];
            } else if constexpr (EPC == 8) {
                const uint2 lo = *(const uint2*)(vp + c * 32);        // keys 32c + 4g .. +3
                const uint2 hi = *(const uint2*)(vp + c * 32 + 16);   // keys 32c + 16 + 4g .. +3
                vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                vf = *(const uint4*)(vp + c * 16);                    // keys 16c + 4g .. +3
            }
            acc = Elem<T>::mma(pc[c], vf, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = q0 + fg * 4 + r;
            if (qq < p.Tq) O[((long)b * p.Tq + qq) * p.ldo + h * HD + dt * 16 + fr] = Elem<T>::to(acc[r]);
        }
    }
}

template <typename T>
int dispatch(AttnArgs& a, int hd, hipStream_t s) {
    if (hd != 192) return EMAGE_EINVAL;
    const int grid = a.B * a.H * ((a.Tq + 15) / 16);
    if (a.Tk <= 32) hipLaunchKernelGGL((attn_kernel<T, 192, 2>), dim3(grid), dim3(64), 0, s, a);
    else if (a.Tk <= 64) hipLaunchKernelGGL((attn_kernel<T, 192, 4>), dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<T, 192, 8>), dim3(grid), dim3(64), 0, s, a);
    return launch_status();
}

}  // namespace

extern "C" int emage_attention(int dtype, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int vt_rows,
                               void* out, int ldo, int B, int H, int Tq, int Tk, int hd, void* stream) {
    if (!q || !k || !vt || !out || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || Tk > 128 || vt_rows < H * hd) return EMAGE_EINVAL;
    if (dtype != EMAGE_BF16 && dtype != EMAGE_F32) return EMAGE_EINVAL;
    const int epc = dtype == EMAGE_BF16 ? 8 : 4;
    if (ldq % epc || ldk % epc || ldvt % 32 || ldvt < ((Tk + 31) / 32) * 32) return EMAGE_EINVAL;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) return EMAGE_EINVAL;
    AttnArgs a{q, k, vt, out, ldq, ldk, ldvt, vt_rows, ldo, B, H, Tq, Tk, 1.0f / sqrtf((float)hd)};
    hipStream_t s = (hipStream_t)stream;
    return dtype == EMAGE_BF16 ? dispatch<bf16_t>(a, hd, s) : dispatch<float>(a, hd, s);
}

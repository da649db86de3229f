// Codebook kernels: fp32 L2 nearest-neighbour arg-min (K6), classifier log-softmax arg-max (K8),
// codebook row gather (K7).  Index outputs are int64 and follow torch's first-extremum tie rule.
#include "common.h"
#include "h2.h"
#include <math.h>

namespace {

// (value, index) "less" with torch's rules: the first extremum wins ties, and a NaN is the extremum (torch.argmin /
// argmax / max return the position of the first NaN), so a NaN upstream surfaces as its own index instead of a
// plausible-looking code
__device__ __forceinline__ void take_min(float& d, int& i, float od, int oi) {
    const bool od_nan = od != od, d_nan = d != d;
    if (od < d || (od == d && oi < i) || (od_nan && (!d_nan || oi < i))) { d = od; i = oi; }
}
__device__ __forceinline__ void take_max(float& d, int& i, float od, int oi) {
    const bool od_nan = od != od, d_nan = d != d;
    if (od > d || (od == d && oi < i) || (od_nan && (!d_nan || oi < i))) { d = od; i = oi; }
}

// Index tensors are addressed as 2-D views: element n of a flat (N,) index list lives at (n / rows)*ld + (n % rows)*ts,
// so a (B, T) window cut out of a longer (B, L) code buffer (ld = L), or one id per clip broadcast over T frames
// (ts = 0), is read / written in place.  rows <= 0 means a plain contiguous list.
struct IdxView { int rows; long ld; int ts; };
__device__ __forceinline__ long idx_at(const IdxView v, int n) {
    if (v.rows <= 0) return n;
    const int b = n / v.rows;
    return (long)b * v.ld + (long)(n - b * v.rows) * v.ts;
}

// One block = 4 waves = 16 rows of z.  Wave w scores code tiles {w, w+4, ...} (16 codes each) against
// the 16 rows with exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 == an fmaf chain), forms
// d = (|z|^2 + |e|^2) - 2 z.e in the reference's association, reduces the arg-min over its codes
// inside the wave (16 lanes per row group, xor-shuffles), then the 4 waves meet in LDS.
template <int D>
__global__ __launch_bounds__(256) void vq_argmin_mfma(const float* __restrict__ z, int ldz,
                                                      const float* __restrict__ cb, int64_t* __restrict__ idx, IdxView iv,
                                                      int N, int K) {
    constexpr int NS = D / 16;
    __shared__ float s_d[4][16];
    __shared__ int s_i[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int r0 = blockIdx.x * 16;

    // A operand: z rows (row i = fr), this lane's k-subset = chunks {4s + fg}
    const int zrow = min(r0 + fr, N - 1);
    const float* zp = z + (long)zrow * ldz + fg * 4;
    uint4 zf[NS];
    float z2 = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float4 v = *(const float4*)(zp + s * 16);
        zf[s] = __builtin_bit_cast(uint4, v);
        z2 += v.x * v.x; z2 += v.y * v.y; z2 += v.z * v.z; z2 += v.w * v.w;
    }
    z2 += __shfl_xor(z2, 16);
    z2 += __shfl_xor(z2, 32);          // |z|^2 of row fr, in every lane with that fr
    float z2r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z2r[r] = __shfl(z2, fg * 4 + r);   // rows this lane's accumulators hold

    float best_d[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int best_i[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    const int ktiles = (K + 15) >> 4;
    for (int ct = wave; ct < ktiles; ct += 4) {
        const int code = ct * 16 + fr;
        const float* ep = cb + (long)min(code, K - 1) * D + fg * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 v = *(const float4*)(ep + s * 16);
            e2 += v.x * v.x; e2 += v.y * v.y; e2 += v.z * v.z; e2 += v.w * v.w;
            acc = Elem<float>::mma(zf[s], __builtin_bit_cast(uint4, v), acc);
        }
        e2 += __shfl_xor(e2, 16);
        e2 += __shfl_xor(e2, 32);      // |e|^2 of code ct*16 + fr
        if (code < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = __fsub_rn(__fadd_rn(z2r[r], e2), 2.0f * acc[r]);
                take_min(best_d[r], best_i[r], d, code);
            }
        }
    }
    // reduce over the 16 lanes (codes) that share a row group
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            const float od = __shfl_xor(best_d[r], m);
            const int oi = __shfl_xor(best_i[r], m);
            take_min(best_d[r], best_i[r], od, oi);
        }
    }
    if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_d[wave][fg * 4 + r] = best_d[r]; s_i[wave][fg * 4 + r] = best_i[r]; }
    }
    __syncthreads();
    if (threadIdx.x < 16 && r0 + threadIdx.x < N) {
        float d = s_d[0][threadIdx.x]; int i = s_i[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) take_min(d, i, s_d[w][threadIdx.x], s_i[w][threadIdx.x]);
        idx[idx_at(iv, r0 + threadIdx.x)] = (int64_t)i;
    }
}

// Large-N form (N >= 16 384): 64 rows per block — wave w owns rows 16w..16w+15 and scores ALL code tiles; each 16-code tile
// is staged ONCE per block in LDS (double-buffered, rows padded by 16 B so the 16 lanes of a ds_read_b128 group hit 16
// different bank groups) instead of being fetched from L2 by every 16-row block: 4x less codebook traffic (at N = 1 M the
// 16-row kernel moves 17 GB of codebook through L2 for 1 GB of input).  Same arithmetic per (row, code) as above — same
// chunk order into the exact-fp32 MFMA chain, same association — so the indices are identical.
template <int D>
__global__ __launch_bounds__(256) void vq_argmin_mfma_rows64(const float* __restrict__ z, int ldz,
                                                             const float* __restrict__ cb, int64_t* __restrict__ idx, IdxView iv,
                                                             int N, int K) {
    constexpr int NS = D / 16;
    constexpr int LDC = D + 4;                             // padded LDS row (floats)
    __shared__ __attribute__((aligned(16))) float s_cb[2][16 * LDC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int r0 = blockIdx.x * 64 + wave * 16;

    const int zrow = min(r0 + fr, N - 1);
    const float* zp = z + (long)zrow * ldz + fg * 4;
    uint4 zf[NS];
    float z2 = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float4 v = *(const float4*)(zp + s * 16);
        zf[s] = __builtin_bit_cast(uint4, v);
        z2 += v.x * v.x; z2 += v.y * v.y; z2 += v.z * v.z; z2 += v.w * v.w;
    }
    z2 += __shfl_xor(z2, 16);
    z2 += __shfl_xor(z2, 32);
    float z2r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z2r[r] = __shfl(z2, fg * 4 + r);

    // cooperative tile load: 16 codes x D floats = 4*D float4; thread t moves float4 t, t+256, ...
    const int ktiles = (K + 15) >> 4;
    auto stage = [&](int ct, int buf) {
        for (int i = threadIdx.x; i < 16 * (D / 4); i += 256) {
            const int code = i / (D / 4), c4 = i - code * (D / 4);
            const int k = min(ct * 16 + code, K - 1);
            *(float4*)&s_cb[buf][code * LDC + c4 * 4] = *(const float4*)(cb + (long)k * D + c4 * 4);
        }
    };
    float best_d[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int best_i[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    stage(0, 0);
    __syncthreads();
    for (int ct = 0; ct < ktiles; ++ct) {
        const int buf = ct & 1;
        if (ct + 1 < ktiles) stage(ct + 1, buf ^ 1);       // the other buffer was last read two barriers ago
        const int code = ct * 16 + fr;
        const float* ep = &s_cb[buf][fr * LDC + fg * 4];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 v = *(const float4*)(ep + s * 16);
            e2 += v.x * v.x; e2 += v.y * v.y; e2 += v.z * v.z; e2 += v.w * v.w;
            acc = Elem<float>::mma(zf[s], __builtin_bit_cast(uint4, v), acc);
        }
        e2 += __shfl_xor(e2, 16);
        e2 += __shfl_xor(e2, 32);
        if (code < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = __fsub_rn(__fadd_rn(z2r[r], e2), 2.0f * acc[r]);
                take_min(best_d[r], best_i[r], d, code);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            const float od = __shfl_xor(best_d[r], m);
            const int oi = __shfl_xor(best_i[r], m);
            take_min(best_d[r], best_i[r], od, oi);
        }
    }
    if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + fg * 4 + r;
            if (row < N) idx[idx_at(iv, row)] = (int64_t)best_i[r];
        }
    }
}

// Generic-D fallback: one wave per row, lane c scores codes c, c+64, ... with an fmaf chain.
__global__ __launch_bounds__(256) void vq_argmin_generic(const float* __restrict__ z, int ldz,
                                                         const float* __restrict__ cb, int64_t* __restrict__ idx, IdxView iv,
                                                         int N, int K, int D) {
    __shared__ float s_z[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= N) return;
    float z2 = 0.f;
    for (int j = lane; j < D; j += 64) { const float v = z[(long)row * ldz + j]; s_z[wave][j] = v; z2 += v * v; }
    for (int m = 32; m >= 1; m >>= 1) z2 += __shfl_xor(z2, m);
    __builtin_amdgcn_wave_barrier();
    float bd = INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < K; c += 64) {
        const float* e = cb + (long)c * D;
        float dot = 0.f, e2 = 0.f;
        for (int j = 0; j < D; ++j) { dot = fmaf(s_z[wave][j], e[j], dot); e2 += e[j] * e[j]; }
        take_min(bd, bi, __fsub_rn(__fadd_rn(z2, e2), 2.0f * dot), c);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const float od = __shfl_xor(bd, m); const int oi = __shfl_xor(bi, m);
        take_min(bd, bi, od, oi);
    }
    if (lane == 0) idx[idx_at(iv, row)] = (int64_t)bi;
}

// One wave per row: y = (x - max) - log(sum exp(x - max)); first maximum of y.
__global__ __launch_bounds__(256) void argmax_logsoftmax(const float* __restrict__ x, int ld, int64_t* __restrict__ idx, IdxView iv, int N, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= N) return;
    const float* xp = x + (long)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, xp[c]);
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(xp[c] - mx);
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    const float ls = logf(sum);
    float bd = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) take_max(bd, bi, __fsub_rn(__fsub_rn(xp[c], mx), ls), c);
    for (int m = 32; m >= 1; m >>= 1) {
        const float od = __shfl_xor(bd, m); const int oi = __shfl_xor(bi, m);
        take_max(bd, bi, od, oi);
    }
    if (lane == 0) idx[idx_at(iv, row)] = (int64_t)bi;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows(const float* __restrict__ table, const int64_t* __restrict__ idx, IdxView iv,
                                                   T* __restrict__ out, int ldo, int n_store, int N, int K, int D) {
    const int row = blockIdx.x;
    long k = idx[idx_at(iv, row)];
    k = k < 0 ? 0 : (k >= K ? K - 1 : k);
    for (int j = threadIdx.x; j < n_store; j += blockDim.x)
        out[(long)row * ldo + j] = Elem<T>::to(j < D ? table[k * D + j] : 0.f);
}

// EMAGE_H2 output (csrc/h2.h): one thread per group of 8 columns
__global__ __launch_bounds__(128) void gather_rows_h2(const float* __restrict__ table, const int64_t* __restrict__ idx, IdxView iv,
                                                      emage_dev::h2_t* __restrict__ out, int ldo, int n_store, int N, int K, int D, float h2s) {
    const int row = blockIdx.x;
    long k = idx[idx_at(iv, row)];
    k = k < 0 ? 0 : (k >= K ? K - 1 : k);
    for (int g = threadIdx.x; g < (n_store >> 3); g += blockDim.x) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 8 * g + e < D ? table[k * D + 8 * g + e] : 0.f;
        emage_dev::h2_store8(out + (long)row * ldo + 8 * g, v, h2s);
    }
}

}  // namespace

extern "C" int emage_vq_argmin_f32(const float* z, int ldz, const float* codebook, int64_t* idx, int idx_rows, long idx_ld,
                                   int N, int K, int D, void* stream) {
    if (!z || !codebook || !idx || N <= 0 || K <= 0 || K > 4096 || D <= 0 || D > 1024 || D % 4 || ldz < D) return EMAGE_EINVAL;
    if (idx_rows > 0 && (N % idx_rows != 0 || idx_ld < idx_rows)) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const IdxView iv{idx_rows, idx_ld, 1};
    const bool aligned = (ldz % 4 == 0) && !(((uintptr_t)z | (uintptr_t)codebook) & 15);
    if (D == 256 && aligned && N >= 16384)
        hipLaunchKernelGGL((vq_argmin_mfma_rows64<256>), dim3((N + 63) / 64), dim3(256), 0, s, z, ldz, codebook, idx, iv, N, K);
    else if (D == 256 && aligned)
        hipLaunchKernelGGL((vq_argmin_mfma<256>), dim3((N + 15) / 16), dim3(256), 0, s, z, ldz, codebook, idx, iv, N, K);
    else
        hipLaunchKernelGGL(vq_argmin_generic, dim3((N + 3) / 4), dim3(256), 0, s, z, ldz, codebook, idx, iv, N, K, D);
    return launch_status();
}

extern "C" int emage_argmax_logsoftmax_f32(const float* logits, int ld, int64_t* idx, int idx_rows, long idx_ld, int N, int C, void* stream) {
    if (!logits || !idx || N <= 0 || C <= 0 || C > 4096 || ld < C) return EMAGE_EINVAL;
    if (idx_rows > 0 && (N % idx_rows != 0 || idx_ld < idx_rows)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(argmax_logsoftmax, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, idx, IdxView{idx_rows, idx_ld, 1}, N, C);
    return launch_status();
}

extern "C" int emage_gather_rows(const float* table, const int64_t* idx, int idx_rows, long idx_ld, int idx_tstride,
                                 void* out, int ldo, int n_store, int N, int K, int D, int dtype, void* stream) {
    if (!table || !idx || !out || N <= 0 || K <= 0 || D <= 0 || n_store < D || ldo < n_store) return EMAGE_EINVAL;
    if (idx_rows > 0 && (N % idx_rows != 0 || idx_tstride < 0 || idx_tstride > 1)) return EMAGE_EINVAL;
    emage_dev::H2Scale hs;
    if (emage_dev::h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const IdxView iv{idx_rows, idx_ld, idx_tstride};
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((gather_rows<bf16_t>), dim3(N), dim3(128), 0, s, table, idx, iv, (bf16_t*)out, ldo, n_store, N, K, D);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((gather_rows<float>), dim3(N), dim3(128), 0, s, table, idx, iv, (float*)out, ldo, n_store, N, K, D);
    else if (dtype == EMAGE_H2) {
        if (n_store % 8 || ldo % 8 || ((uintptr_t)out & 15)) return EMAGE_EINVAL;
        hipLaunchKernelGGL(gather_rows_h2, dim3(N), dim3(128), 0, s, table, idx, iv, (emage_dev::h2_t*)out, ldo, n_store, N, K, D, hs.s);
    } else return EMAGE_EINVAL;
    return launch_status();
}

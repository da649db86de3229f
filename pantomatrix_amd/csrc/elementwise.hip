// Row-wise and elementwise glue kernels: LayerNorm (K5) and the add / mask / cast helpers (K11).
// All are HBM-bound streaming kernels: one wave per row (LayerNorm) or grid-stride float4 loops.
#include "common.h"
#include "ln_row.h"
#include <math.h>

namespace {

using namespace emage_dev;

// LayerNorm: one wave per row, 4 rows per block (body in ln_row.h)
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        const T* __restrict__ add, int ldadd,
                                                        float* __restrict__ yf, T* __restrict__ y, int ldy, int M, int C,
                                                        h2_t* __restrict__ yh = nullptr, float h2s = H2_SCALE) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    layernorm_row<T, MAXV>(x + (long)row * ldx, gamma, beta, eps, add ? add + (long)row * ldadd : nullptr,
                           yf ? yf + (long)row * ldy : nullptr, y ? y + (long)row * ldy : nullptr, C, lane,
                           yh ? yh + (long)row * ldy : nullptr, h2s);
}

// ---- EMAGE_H2 storage (csrc/h2.h): 4 logical columns at a 4-aligned column n of a row ----
// hs: the activation-image scale of the call (csrc/h2.h; the other storage types ignore it)
template <typename T> __device__ __forceinline__ float4 ldv4(const T* p, int n, const H2Scale& hs) { return Vec4<T>::load(p); }
template <> __device__ __forceinline__ float4 ldv4<h2_t>(const h2_t* p, int n, const H2Scale& hs) {
    float v[4];
    h2_load4(p, n, v, hs.inv);
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <typename T> __device__ __forceinline__ void stv4(T* p, int n, float4 v, const H2Scale& hs) { Vec4<T>::store(p, v); }
template <> __device__ __forceinline__ void stv4<h2_t>(h2_t* p, int n, float4 v, const H2Scale& hs) {
    const float t[4] = {v.x, v.y, v.z, v.w};
    h2_store4(p, n, t, hs.s);
}

// out[m] = a[m] + b[m % mod_b] (+ c[m % mod_c]); operand k is fp32 when bit k of f32_mask is set, else T
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const void* __restrict__ a, int lda, const void* __restrict__ b, int ldb, int mod_b,
                                                  const void* __restrict__ c, int ldc, int mod_c, int f32_mask,
                                                  float* __restrict__ of, T* __restrict__ o, int ldo, int M, int C, H2Scale hs) {
    const int nv = C >> 2;
    const long total = (long)M * nv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / nv), n = 4 * (int)(i - (long)m * nv);
        auto ld = [&](const void* p, int ldp, int mod, int bit) {
            const long r = mod ? m % mod : m;
            return (f32_mask >> bit) & 1 ? Vec4<float>::load((const float*)p + r * ldp + n) : ldv4<T>((const T*)p + r * ldp + n, n, hs);
        };
        float4 v = ld(a, lda, 0, 0);
        const float4 w = ld(b, ldb, mod_b, 1);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        if (c) { const float4 u = ld(c, ldc, mod_c, 2); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        if (of) Vec4<float>::store(of + (long)m * ldo + n, v);
        if (o) stv4<T>(o + (long)m * ldo + n, n, v, hs);
    }
}

// One window of the masked-motion input (M:267-268 + the seed splice of inference(), M:386-391):
//   frames t <  pre (seed given): v = mask == 0 ? motion : seed      (and the frame counts as unmasked)
//   other frames:                 v = mask == 1 ? mask_embedding : motion
// motion / mask: frame (b, t) at b*ldb + t*C (a window cut out of a longer (B, L, C) tensor needs no copy);
// seed: frame (b, t) at b*ld_seed + t*C.  Output (B*T, ldo) in T with a zero channel tail.
template <typename T>
__global__ __launch_bounds__(256) void pack_motion_kernel(const float* __restrict__ motion, const float* __restrict__ mask, long ldb,
                                                          const float* __restrict__ emb, const float* __restrict__ seed, long ld_seed, int pre,
                                                          T* __restrict__ out, int ldo, int n_store, int B, int Tn, int C) {
    const long total = (long)B * Tn * n_store;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n_store), n = (int)(i - (long)m * n_store);
        const int b = m / Tn, t = m - b * Tn;
        float v = 0.f;
        if (n < C) {
            const long src = (long)b * ldb + (long)t * C + n;
            const float mk = mask[src], mv = motion[src];
            if (seed && t < pre) v = mk == 0.0f ? mv : seed[(long)b * ld_seed + (long)t * C + n];
            else v = mk == 1.0f ? emb[n] : mv;
        }
        out[(long)m * ldo + n] = Elem<T>::to(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, int lds, T* __restrict__ out, int ldo, int n_store, int M, int C) {
    const long total = (long)M * n_store;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n_store), n = (int)(i - (long)m * n_store);
        out[(long)m * ldo + n] = Elem<T>::to(n < C ? src[(long)m * lds + n] : 0.f);
    }
}

// EMAGE_H2 forms: one thread per group of 8 logical columns (32 bytes)
__global__ __launch_bounds__(256) void pack_motion_h2_kernel(const float* __restrict__ motion, const float* __restrict__ mask, long ldb,
                                                             const float* __restrict__ emb, const float* __restrict__ seed, long ld_seed, int pre,
                                                             h2_t* __restrict__ out, int ldo, int n_store, int B, int Tn, int C, float h2s) {
    const int ng = n_store >> 3;
    const long total = (long)B * Tn * ng;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ng), n0 = 8 * (int)(i - (long)m * ng);
        const int b = m / Tn, t = m - b * Tn;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = n0 + e;
            v[e] = 0.f;
            if (n < C) {
                const long src = (long)b * ldb + (long)t * C + n;
                const float mk = mask[src], mv = motion[src];
                if (seed && t < pre) v[e] = mk == 0.0f ? mv : seed[(long)b * ld_seed + (long)t * C + n];
                else v[e] = mk == 1.0f ? emb[n] : mv;
            }
        }
        h2_store8(out + (long)m * ldo + n0, v, h2s);
    }
}

// src may alias out (same row stride): each thread reads its 8 values before it writes their 32 bytes
__global__ __launch_bounds__(256) void cast_pad_h2_kernel(const float* src, int lds, h2_t* out, int ldo, int n_store, int M, int C, float h2s) {
    const int ng = n_store >> 3;
    const long total = (long)M * ng;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ng), n0 = 8 * (int)(i - (long)m * ng);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = n0 + e < C ? src[(long)m * lds + n0 + e] : 0.f;
        h2_store8(out + (long)m * ldo + n0, v, h2s);
    }
}

// fp32 (M, C) -> EMAGE_H2 with a power-of-two pre-scale, optionally TRANSPOSED: out[c][m] = src[m][c] * scale, (C, m_store) with a zero tail
// [M, m_store) — the operands of the training step's backward contractions (dW = dY^T X contracts over the rows).  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void h2_cast_t_kernel(const float* __restrict__ src, int lds_, h2_t* __restrict__ out, int ldo, int m_store, int M, int C, float scale) {
    __shared__ float tile[64][65];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    if ((lds_ & 3) == 0 && (((uintptr_t)src & 15) == 0) && c0 + 64 <= C) {
        // whole tile columns, 16-byte aligned rows: four 16-byte loads per thread, all in flight before the first LDS write (round 6: the scalar
        // form ran at 1.9 TB/s on the training step's 11 MB layer inputs — 522 launches, 6 ms per step)
        const int r0 = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + 16 * k;
            v[k] = m0 + r < M ? *(const float4*)(src + (long)(m0 + r) * lds_ + c0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + 16 * k;
            tile[r][c4] = v[k].x * scale; tile[r][c4 + 1] = v[k].y * scale; tile[r][c4 + 2] = v[k].z * scale; tile[r][c4 + 3] = v[k].w * scale;
        }
    } else {
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            tile[r][c] = (m0 + r < M && c0 + c < C) ? src[(long)(m0 + r) * lds_ + c0 + c] * scale : 0.f;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int c = i >> 3, g = i & 7;
        if (c0 + c < C && m0 + 8 * g < m_store) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[8 * g + e][c];
            h2_store8(out + (long)(c0 + c) * ldo + m0 + 8 * g, v);
        }
    }
}
__global__ __launch_bounds__(256) void h2_cast_kernel(const float* src, int lds_, h2_t* out, int ldo, int n_store, int M, int C, float scale) {
    const int ng = n_store >> 3;
    const long total = (long)M * ng;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ng), n0 = 8 * (int)(i - (long)m * ng);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = n0 + e < C ? src[(long)m * lds_ + n0 + e] * scale : 0.f;
        h2_store8(out + (long)m * ldo + n0, v);
    }
}

// fp32 (N, K) weights -> the EMAGE_F16X3 weight image of w * scale (ops.split_f16_weights): per row and 32-k K-tile 128 bytes =
// [hi plane | lo plane], each plane 4 chunks of 16 bytes, chunk g = k 4g..4g+3 and 16+4g..16+4g+3 (the order the X3 tiles' LDS-DMA lands).
// One thread per chunk pair.  scale_16 = scale / 16 (h2_split8 multiplies by H2_SCALE): powers of two, so the planes are those of w * scale.
__global__ __launch_bounds__(256) void f16x3_pack_kernel(const float* __restrict__ w, int ldw, unsigned char* __restrict__ out, long ldo_bytes, int N, int K, float scale_16) {
    const int nkt = K >> 5;
    const long total = (long)N * nkt * 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i & 3);
        const long r = i >> 2;
        const int kt = (int)(r % nkt), n = (int)(r / nkt);
        const float* src = w + (long)n * ldw + 32 * kt + 4 * g;
        const float4 a = *(const float4*)src, b = *(const float4*)(src + 16);
        const float v[8] = {a.x * scale_16, a.y * scale_16, a.z * scale_16, a.w * scale_16, b.x * scale_16, b.y * scale_16, b.z * scale_16, b.w * scale_16};
        uint4 hi, lo;
        h2_split8(v, hi, lo);
        unsigned char* dst = out + (long)n * ldo_bytes + 128 * kt + 16 * g;
        *(uint4*)dst = hi;
        *(uint4*)(dst + 64) = lo;
    }
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int emage_layernorm(int dtype, const void* x, int ldx, const float* gamma, const float* beta, float eps,
                               const void* add, int ldadd, float* y_f32, void* y, int ldy, int M, int C, void* stream) {
    H2Scale hs;
    if (h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    if (!x || !gamma || !beta || (!y_f32 && !y) || M <= 0 || C <= 0 || C % 64 || C > 1024) return EMAGE_EINVAL;
    if (ldx % 4 || ldy % 4 || (add && ldadd % 4)) return EMAGE_EINVAL;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)add | (uintptr_t)y_f32 | (uintptr_t)y) & 15) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((M + 3) / 4), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((layernorm_kernel<bf16_t, 4>), grid, block, 0, s, (const bf16_t*)x, ldx, gamma, beta, eps, (const bf16_t*)add, ldadd, y_f32, (bf16_t*)y, ldy, M, C);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((layernorm_kernel<float, 4>), grid, block, 0, s, (const float*)x, ldx, gamma, beta, eps, (const float*)add, ldadd, y_f32, (float*)y, ldy, M, C, (h2_t*)nullptr);
    else if (dtype == EMAGE_H2) {      // x / add / y_f32 float32 (the fp32 residual stream), y the EMAGE_H2 copy for the next contraction
        if (ldy % 8) return EMAGE_EINVAL;
        hipLaunchKernelGGL((layernorm_kernel<float, 4>), grid, block, 0, s, (const float*)x, ldx, gamma, beta, eps, (const float*)add, ldadd, y_f32, (float*)nullptr, ldy, M, C, (h2_t*)y, hs.s);
    } else return EMAGE_EINVAL;
    return launch_status();
}

extern "C" int emage_add(int dtype, const void* a, int lda, const void* b, int ldb, int mod_b, const void* c, int ldc, int mod_c,
                         int f32_mask, float* out_f32, void* out, int ldo, int M, int C, void* stream) {
    H2Scale hs;
    if (h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    if (!a || !b || (!out_f32 && !out) || M <= 0 || C <= 0 || C % 4 || lda % 4 || ldb % 4 || ldo % 4 || (c && ldc % 4)) return EMAGE_EINVAL;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out_f32 | (uintptr_t)out) & 7) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)M * C / 4)), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((add_kernel<bf16_t>), grid, block, 0, s, a, lda, b, ldb, mod_b, c, ldc, mod_c, f32_mask, out_f32, (bf16_t*)out, ldo, M, C, hs);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((add_kernel<float>), grid, block, 0, s, a, lda, b, ldb, mod_b, c, ldc, mod_c, f32_mask | 7, out_f32, (float*)out, ldo, M, C, hs);
    else if (dtype == EMAGE_H2) {      // operands with a clear f32_mask bit and `out` are EMAGE_H2 images: rows start on a 32-byte group
        if (lda % 8 || ldb % 8 || ldo % 8 || (c && ldc % 8)) return EMAGE_EINVAL;
        hipLaunchKernelGGL((add_kernel<h2_t>), grid, block, 0, s, a, lda, b, ldb, mod_b, c, ldc, mod_c, f32_mask, out_f32, (h2_t*)out, ldo, M, C, hs);
    } else return EMAGE_EINVAL;
    return launch_status();
}

extern "C" int emage_pack_motion(int dtype, const float* motion, const float* mask, long ldb, const float* mask_embedding,
                                 const float* seed, long ld_seed, int pre,
                                 void* out, int ldo, int n_store, int B, int T, int C, void* stream) {
    H2Scale hs;
    if (h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    if (!motion || !mask || !mask_embedding || !out || B <= 0 || T <= 0 || C <= 0 || n_store < C || ldo < n_store) return EMAGE_EINVAL;
    if (ldb < (long)T * C || (seed && (pre <= 0 || pre > T || ld_seed < (long)pre * C))) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)B * T * n_store)), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((pack_motion_kernel<bf16_t>), grid, block, 0, s, motion, mask, ldb, mask_embedding, seed, ld_seed, pre, (bf16_t*)out, ldo, n_store, B, T, C);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((pack_motion_kernel<float>), grid, block, 0, s, motion, mask, ldb, mask_embedding, seed, ld_seed, pre, (float*)out, ldo, n_store, B, T, C);
    else if (dtype == EMAGE_H2) {
        if (n_store % 8 || ldo % 8 || ((uintptr_t)out & 15)) return EMAGE_EINVAL;
        hipLaunchKernelGGL(pack_motion_h2_kernel, dim3(grid_for((long)B * T * n_store / 8)), block, 0, s, motion, mask, ldb, mask_embedding, seed, ld_seed, pre, (h2_t*)out, ldo, n_store, B, T, C, hs.s);
    } else return EMAGE_EINVAL;
    return launch_status();
}

extern "C" int emage_cast_pad(int dtype, const float* src, int lds, void* out, int ldo, int n_store, int M, int C, void* stream) {
    H2Scale hs;
    if (h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    if (!src || !out || M <= 0 || C <= 0 || n_store < C || ldo < n_store || lds < C) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)M * n_store)), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((cast_pad_kernel<bf16_t>), grid, block, 0, s, src, lds, (bf16_t*)out, ldo, n_store, M, C);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((cast_pad_kernel<float>), grid, block, 0, s, src, lds, (float*)out, ldo, n_store, M, C);
    else if (dtype == EMAGE_H2) {      // fp32 -> EMAGE_H2; in place when src == out and lds == ldo
        if (n_store % 8 || ldo % 8 || ((uintptr_t)out & 15) || ((const void*)src == out && lds != ldo)) return EMAGE_EINVAL;
        hipLaunchKernelGGL(cast_pad_h2_kernel, dim3(grid_for((long)M * n_store / 8)), block, 0, s, src, lds, (h2_t*)out, ldo, n_store, M, C, hs.s);
    } else return EMAGE_EINVAL;
    return launch_status();
}

extern "C" int emage_h2_cast(const float* src, int lds, void* out, int ldo, int n_store, int M, int C, float scale, int transpose, void* stream) {
    if (!src || !out || M <= 0 || C <= 0 || lds < C || n_store % 8 || ldo % 8 || ldo < n_store || ((uintptr_t)out & 15) || !(scale > 0.f)) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (transpose) {
        if (n_store < M) return EMAGE_EINVAL;
        hipLaunchKernelGGL(h2_cast_t_kernel, dim3((n_store + 63) / 64, (C + 63) / 64), dim3(256), 0, s, src, lds, (h2_t*)out, ldo, n_store, M, C, scale);
    } else {
        if (n_store < C) return EMAGE_EINVAL;
        hipLaunchKernelGGL(h2_cast_kernel, dim3(grid_for((long)M * n_store / 8)), dim3(256), 0, s, src, lds, (h2_t*)out, ldo, n_store, M, C, scale);
    }
    return launch_status();
}

extern "C" int emage_f16x3_pack_weights(const float* w, int ldw, void* out, int ldo, int N, int K, float scale, void* stream) {
    if (!w || !out || N <= 0 || K <= 0 || K % 32 || ldw < K || ldw % 4 || ldo < K || ((uintptr_t)w & 15) || ((uintptr_t)out & 15) || ldo % 4 || !(scale > 0.f)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(f16x3_pack_kernel, dim3(grid_for((long)N * (K / 32) * 4)), dim3(256), 0, (hipStream_t)stream, w, ldw, (unsigned char*)out, (long)ldo * 4, N, K,
                       scale * H2_INV);
    return launch_status();
}

// counter[0] += number of non-finite values among x[0 .. n): the end-of-batch health check of the runners (an activation beyond the
// split-fp16 range, |x| >= 4094, becomes inf / NaN and reaches the outputs; this makes it an error instead of a silent result)
namespace {
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ x, long n, int* __restrict__ counter) {
    int bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned u = __builtin_bit_cast(unsigned, x[i]);
        bad += (u & 0x7f800000u) == 0x7f800000u;
    }
    if (bad) atomicAdd(counter, bad);
}
}  // namespace

// several tensors in ONE launch (round 6: the 11 health-check launches of a clip batch become one): the table of (pointer, count) travels by
// value in the kernel arguments; blocks walk the tensors' 4 K-element chunks in table order
namespace {
constexpr int CNF_MAX = 16;
struct CountTable { const float* x[CNF_MAX]; long n[CNF_MAX]; int chunk_end[CNF_MAX]; int count; };
constexpr long CNF_CHUNK = 1L << 12;      // elements per block: the clip batch's ~2 M values make ~500 blocks (64 K per block left most of the chip idle: 58 us)
__global__ __launch_bounds__(256) void count_nonfinite_multi_kernel(CountTable t, int* __restrict__ counter) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < CNF_MAX - 1; ++i) k += (i + 1 < t.count && (int)blockIdx.x >= t.chunk_end[i]) ? 1 : 0;
    const long c0 = ((long)blockIdx.x - (k ? t.chunk_end[k - 1] : 0)) * CNF_CHUNK;
    const long c1 = c0 + CNF_CHUNK < t.n[k] ? c0 + CNF_CHUNK : t.n[k];
    const float* __restrict__ x = t.x[k];
    int bad = 0;
    for (long i = c0 + threadIdx.x; i < c1; i += 256) {
        const unsigned u = __builtin_bit_cast(unsigned, x[i]);
        bad += (u & 0x7f800000u) == 0x7f800000u;
    }
    if (bad) atomicAdd(counter, bad);
}
}  // namespace

extern "C" int emage_count_nonfinite_multi(const float* const* xs, const long* ns, int count, int* counter, void* stream) {
    if (!xs || !ns || !counter || count <= 0 || count > CNF_MAX) return EMAGE_EINVAL;
    CountTable t;
    int chunks = 0;
    for (int i = 0; i < count; ++i) {
        if (!xs[i] || ns[i] <= 0) return EMAGE_EINVAL;
        t.x[i] = xs[i]; t.n[i] = ns[i];
        chunks += (int)((ns[i] + CNF_CHUNK - 1) / CNF_CHUNK);
        t.chunk_end[i] = chunks;
    }
    for (int i = count; i < CNF_MAX; ++i) { t.x[i] = xs[0]; t.n[i] = 0; t.chunk_end[i] = chunks; }
    t.count = count;
    hipLaunchKernelGGL(count_nonfinite_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, t, counter);
    return launch_status();
}

extern "C" int emage_count_nonfinite(const float* x, long n, int* counter, void* stream) {
    if (!x || !counter || n <= 0) return EMAGE_EINVAL;
    long g = (n + 255) / 256;
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, (hipStream_t)stream, x, n, counter);
    return launch_status();
}

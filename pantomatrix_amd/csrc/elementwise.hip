// Row-wise and elementwise glue kernels: LayerNorm (K5) and the add / mask / cast helpers (K11).
// All are HBM-bound streaming kernels: one wave per row (LayerNorm) or grid-stride float4 loops.
#include "common.h"
#include <math.h>

namespace {

// One wave per row, C = 64*NV*4?  Generic: each lane owns float4 groups j = lane + 64*i (C % 4 == 0).
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ add, int ldadd,
                                                        float* __restrict__ yf, T* __restrict__ y, int ldy, int M, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int nv = C >> 2;   // float4 groups per row
    const float4* xp = (const float4*)(x + (long)row * ldx);
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + 64 * i;
        v[i] = j < nv ? xp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
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
    const float4* ap = add ? (const float4*)(add + (long)row * ldadd) : nullptr;
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
            if (ap) { const float4 a = ap[j]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
            if (yf) ((float4*)(yf + (long)row * ldy))[j] = o;
            if (y) {
                T* yp = y + (long)row * ldy + 4 * j;
                yp[0] = Elem<T>::to(o.x); yp[1] = Elem<T>::to(o.y); yp[2] = Elem<T>::to(o.z); yp[3] = Elem<T>::to(o.w);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int mod_b,
                                                  const float* __restrict__ c, int ldc, int mod_c, float* __restrict__ of, T* __restrict__ o, int ldo,
                                                  int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), n = (int)(i - (long)m * C);
        float v = a[(long)m * lda + n] + b[(long)(mod_b ? m % mod_b : m) * ldb + n];
        if (c) v += c[(long)(mod_c ? m % mod_c : m) * ldc + n];
        if (of) of[(long)m * ldo + n] = v;
        if (o) o[(long)m * ldo + n] = Elem<T>::to(v);
    }
}

// where(mask == 1, mask_embedding, motion) -> T, zero channel tail
template <typename T>
__global__ __launch_bounds__(256) void pack_motion_kernel(const float* __restrict__ motion, const float* __restrict__ mask,
                                                          const float* __restrict__ emb, T* __restrict__ out, int ldo, int n_store, int M, int C) {
    const long total = (long)M * n_store;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n_store), n = (int)(i - (long)m * n_store);
        float v = 0.f;
        if (n < C) v = mask[(long)m * C + n] == 1.0f ? emb[n] : motion[(long)m * C + n];
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

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int emage_layernorm(int dtype, const float* x, int ldx, const float* gamma, const float* beta, float eps,
                               const float* add, int ldadd, float* y_f32, void* y, int ldy, int M, int C, void* stream) {
    if (!x || !gamma || !beta || (!y_f32 && !y) || M <= 0 || C <= 0 || C % 64 || C > 1024) return EMAGE_EINVAL;
    if (ldx % 4 || ldy % 4 || (add && ldadd % 4)) return EMAGE_EINVAL;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)add | (uintptr_t)y_f32) & 15) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((M + 3) / 4), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((layernorm_kernel<bf16_t, 4>), grid, block, 0, s, x, ldx, gamma, beta, eps, add, ldadd, y_f32, (bf16_t*)y, ldy, M, C);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((layernorm_kernel<float, 4>), grid, block, 0, s, x, ldx, gamma, beta, eps, add, ldadd, y_f32, (float*)y, ldy, M, C);
    else return EMAGE_EINVAL;
    return launch_status();
}

extern "C" int emage_add(int dtype, const float* a, int lda, const float* b, int ldb, int mod_b, const float* c, int ldc, int mod_c,
                         float* out_f32, void* out, int ldo, int M, int C, void* stream) {
    if (!a || !b || (!out_f32 && !out) || M <= 0 || C <= 0) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)M * C)), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((add_kernel<bf16_t>), grid, block, 0, s, a, lda, b, ldb, mod_b, c, ldc, mod_c, out_f32, (bf16_t*)out, ldo, M, C);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((add_kernel<float>), grid, block, 0, s, a, lda, b, ldb, mod_b, c, ldc, mod_c, out_f32, (float*)out, ldo, M, C);
    else return EMAGE_EINVAL;
    return launch_status();
}

extern "C" int emage_pack_motion(int dtype, const float* motion, const float* mask, const float* mask_embedding,
                                 void* out, int ldo, int n_store, int M, int C, void* stream) {
    if (!motion || !mask || !mask_embedding || !out || M <= 0 || C <= 0 || n_store < C || ldo < n_store) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)M * n_store)), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((pack_motion_kernel<bf16_t>), grid, block, 0, s, motion, mask, mask_embedding, (bf16_t*)out, ldo, n_store, M, C);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((pack_motion_kernel<float>), grid, block, 0, s, motion, mask, mask_embedding, (float*)out, ldo, n_store, M, C);
    else return EMAGE_EINVAL;
    return launch_status();
}

extern "C" int emage_cast_pad(int dtype, const float* src, int lds, void* out, int ldo, int n_store, int M, int C, void* stream) {
    if (!src || !out || M <= 0 || C <= 0 || n_store < C || ldo < n_store || lds < C) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)M * n_store)), block(256);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((cast_pad_kernel<bf16_t>), grid, block, 0, s, src, lds, (bf16_t*)out, ldo, n_store, M, C);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((cast_pad_kernel<float>), grid, block, 0, s, src, lds, (float*)out, ldo, n_store, M, C);
    else return EMAGE_EINVAL;
    return launch_status();
}

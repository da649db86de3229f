// Train-mode forward pieces of the EMAGE training step (reference train_emage_audio.py:130-204 and the
// train-mode behaviour of torch's modules inside models/emage_audio):
//   emage_bn_stats     nn.BatchNorm1d in training mode, statistics half: per-channel batch mean and BIASED variance over all
//                      rows (= B * L positions) of a channels-last conv output, plus the running-statistic update
//                      (momentum, UNBIASED variance) — processing_emage_audio.py:262-294 with nn.BatchNorm1d semantics
//   emage_bn_apply     the normalisation half fused with what follows it inside BasicBlock.forward (P:283-294): affine,
//                      optional shortcut (raw, or itself batch-normalised: the downsample branch), LeakyReLU
//   emage_mse_loss / emage_nll_loss   the two loss forms of train_emage_audio.py:106-130, accumulated into a float64 device scalar
//   emage_mul_add      out = a * mask (+ b): nn.Dropout with a given mask (x * bernoulli / (1 - p)) and the residual add that
//                      follows it in nn.Transformer*Layer; the mask may be stored (T, B, d) while rows run (B, T)
// Statistics are accumulated in float64 (one rounding to fp32 at the end): the batch has up to ~4e5 rows per channel.
#include "common.h"
#include <math.h>

namespace {

constexpr int STAT_ROWS = 4;          // row lanes per block: 4 x 64 columns = 256 threads
constexpr int STAT_CHUNK = 2048;      // rows per block

// partial[(chunk * 2 + {0: sum, 1: sum of squares}) * C + c], deterministic: the finalize kernel adds the chunks in order
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int ldx, int M, int C, double* __restrict__ partial) {
    __shared__ double red[2][STAT_ROWS][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long r0 = (long)blockIdx.x * STAT_CHUNK;
    const long r1 = r0 + STAT_CHUNK < M ? r0 + STAT_CHUNK : M;
    double s = 0.0, ss = 0.0;
    if (c < C)
        for (long r = r0 + rl; r < r1; r += STAT_ROWS) {
            const double v = (double)x[r * ldx + c];
            s += v;
            ss += v * v;
        }
    red[0][rl][cl] = s;
    red[1][rl][cl] = ss;
    __syncthreads();
    if (rl == 0 && c < C) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < STAT_ROWS; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
        partial[((long)blockIdx.x * 2 + 0) * C + c] = a;
        partial[((long)blockIdx.x * 2 + 1) * C + c] = b;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partial, int chunks, int M, int C,
                                                          float* __restrict__ mean, float* __restrict__ var,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int i = 0; i < chunks; ++i) { s += partial[((long)i * 2 + 0) * C + c]; ss += partial[((long)i * 2 + 1) * C + c]; }
    const double mu = s / M;
    double vb = ss / M - mu * mu;
    if (vb < 0.0) vb = 0.0;
    mean[c] = (float)mu;
    var[c] = (float)vb;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * (float)mu);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (float)(M > 1 ? vb * M / (M - 1) : vb));
}

struct BnSide { const float* x; int ld; const float* mean; const float* var; const float* gamma; const float* beta; };

// out = leaky(bn(a) + shortcut, slope);  shortcut: none | s.x raw (s.mean == nullptr) | bn(s)
__global__ __launch_bounds__(256) void bn_apply_kernel(BnSide a, BnSide s, float eps, float slope, float* __restrict__ out, int ldo, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        // torch's batch_norm: (x - mean) * invstd * weight + bias, invstd = 1 / sqrt(var + eps)
        float v = (a.x[m * a.ld + c] - a.mean[c]) * (1.0f / sqrtf(a.var[c] + eps)) * a.gamma[c] + a.beta[c];
        if (s.x) {
            float r = s.x[m * s.ld + c];
            if (s.mean) r = (r - s.mean[c]) * (1.0f / sqrtf(s.var[c] + eps)) * s.gamma[c] + s.beta[c];
            v += r;
        }
        out[m * ldo + c] = leaky(v, slope);
    }
}

__global__ __launch_bounds__(256) void mul_add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ mask, int ldm, int t_rows,
                                                      const float* __restrict__ b, int ldb, float* __restrict__ out, int ldo, int M, int C) {
    const long total = (long)M * C;
    const int nb = t_rows > 0 ? M / t_rows : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        const long mr = t_rows > 0 ? (m % t_rows) * nb + m / t_rows : m;          // row (b, t) of the stream <-> row (t, b) of the mask
        float v = a[m * lda + c] * mask[mr * ldm + c];
        if (b) v += b[m * ldb + c];
        out[m * ldo + c] = v;
    }
}

// ---- losses (train_emage_audio.py:106-130): block partials in float64, added in block order by one thread -> deterministic
constexpr int LOSS_BLOCKS = EMAGE_LOSS_WORKSPACE_BYTES / 8;

__global__ __launch_bounds__(256) void sq_diff_partial_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int M, int C,
                                                              double* __restrict__ partial) {
    __shared__ double red[256];
    const long total = (long)M * C;
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        const float d = a[m * lda + c] - b[m * ldb + c];          // fp32 difference and square as torch's mse_loss, float64 sum
        s += (double)(d * d);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// -log_softmax(logits[m])[index[m]] summed over the block's rows: one thread per row
__global__ __launch_bounds__(256) void nll_partial_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ index, int M, int K,
                                                          double* __restrict__ partial, int* __restrict__ bad) {
    __shared__ double red[256];
    double s = 0.0;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const float* x = logits + m * ld;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, x[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(x[k] - mx);
        const int64_t t = index[m];
        if (t < 0 || t >= K) { *bad = 1; continue; }
        s += (double)(-(x[t] - mx - logf(se)));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void loss_finalize_kernel(const double* __restrict__ partial, int blocks, double scale, double* __restrict__ loss) {
    double s = 0.0;
    for (int i = 0; i < blocks; ++i) s += partial[i];
    loss[0] += s * scale;
}

inline int loss_grid(long total) {
    long g = (total + 255) / 256;
    return (int)(g > LOSS_BLOCKS - 1 ? LOSS_BLOCKS - 1 : (g < 1 ? 1 : g));
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" long emage_bn_stats_workspace_bytes(int M, int C) {
    if (M <= 0 || C <= 0) return EMAGE_EINVAL;
    return (long)((M + STAT_CHUNK - 1) / STAT_CHUNK) * 2 * C * (long)sizeof(double);
}

extern "C" int emage_bn_stats(const float* x, int ldx, int M, int C, void* workspace, long workspace_bytes,
                              float* mean, float* var, float* running_mean, float* running_var, float momentum, void* stream) {
    if (!x || !workspace || !mean || !var || M <= 0 || C <= 0 || ldx < C || !(momentum >= 0.f && momentum <= 1.f)) return EMAGE_EINVAL;
    if (workspace_bytes < emage_bn_stats_workspace_bytes(M, C) || ((uintptr_t)workspace & 7)) return EMAGE_EINVAL;
    const int chunks = (M + STAT_CHUNK - 1) / STAT_CHUNK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, ldx, M, C, (double*)workspace);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)workspace, chunks, M, C, mean, var, running_mean, running_var, momentum);
    return launch_status();
}

extern "C" int emage_bn_apply(const float* x, int ldx, const float* mean, const float* var, const float* gamma, const float* beta,
                              const float* sc, int ld_sc, const float* sc_mean, const float* sc_var, const float* sc_gamma, const float* sc_beta,
                              float eps, float slope, float* out, int ldo, int M, int C, void* stream) {
    if (!x || !mean || !var || !gamma || !beta || !out || M <= 0 || C <= 0 || ldx < C || ldo < C || (sc && ld_sc < C)) return EMAGE_EINVAL;
    if (sc_mean && (!sc || !sc_var || !sc_gamma || !sc_beta)) return EMAGE_EINVAL;
    const BnSide a{x, ldx, mean, var, gamma, beta}, s{sc, ld_sc, sc_mean, sc_var, sc_gamma, sc_beta};
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, a, s, eps, slope, out, ldo, M, C);
    return launch_status();
}

extern "C" int emage_mul_add(const float* a, int lda, const float* mask, int ld_mask, int mask_t_rows, const float* b, int ldb,
                             float* out, int ldo, int M, int C, void* stream) {
    if (!a || !mask || !out || M <= 0 || C <= 0 || lda < C || ld_mask < C || ldo < C || (b && ldb < C)) return EMAGE_EINVAL;
    if (mask_t_rows < 0 || (mask_t_rows > 0 && M % mask_t_rows != 0)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(mul_add_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, a, lda, mask, ld_mask, mask_t_rows, b, ldb, out, ldo, M, C);
    return launch_status();
}

extern "C" int emage_mse_loss(const float* pred, int ld_pred, const float* target, int ld_target, int M, int C, float weight,
                              double* loss, void* workspace, void* stream) {
    if (!pred || !target || !loss || !workspace || M <= 0 || C <= 0 || ld_pred < C || ld_target < C || ((uintptr_t)workspace & 7) || ((uintptr_t)loss & 7)) return EMAGE_EINVAL;
    const int blocks = loss_grid((long)M * C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sq_diff_partial_kernel, dim3(blocks), dim3(256), 0, s, pred, ld_pred, target, ld_target, M, C, (double*)workspace);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, (const double*)workspace, blocks, (double)weight / ((double)M * C), loss);
    return launch_status();
}

extern "C" int emage_nll_loss(const float* logits, int ld, const int64_t* index, int M, int K, float weight, double* loss, void* workspace, void* stream) {
    if (!logits || !index || !loss || !workspace || M <= 0 || K <= 0 || ld < K || ((uintptr_t)workspace & 7) || ((uintptr_t)loss & 7)) return EMAGE_EINVAL;
    const int blocks = loss_grid(M);
    hipStream_t s = (hipStream_t)stream;
    int* bad = (int*)((double*)workspace + (LOSS_BLOCKS - 1));                 // last workspace slot: out-of-range class index seen
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(double), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nll_partial_kernel, dim3(blocks), dim3(256), 0, s, logits, ld, index, M, K, (double*)workspace, bad);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, (const double*)workspace, blocks, (double)weight / (double)M, loss);
    return launch_status();
}

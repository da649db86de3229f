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
#include "h2.h"
#include <math.h>

namespace {

constexpr int STAT_ROWS = 4;          // row lanes per block: 4 x 64 columns = 256 threads
constexpr int STAT_CHUNK = 64;        // least rows per block (2048 left the small-M layers of a training step with ~24 blocks on 256 CUs: 130 us per
                                      // bias gradient, 17 % of the step — profiles/r03_train_kernel_stats_before.csv)

// The finalize step of every chunked column reduction below: 16 columns x 16 lanes per block; lane l adds its contiguous range of
// chunks in order, lane 0 then adds the 16 lane sums in order — deterministic for a given chunk count, and 16x shorter than one thread
// walking all (up to 512) chunks of its column (which cost 20-80 us per call: 14 % of a training step,
// profiles/r03_train_kernel_stats_final.csv).  partial[(chunk * NV + v) * C + c]; true for the thread that holds column c's totals.
constexpr int FIN_COLS = 16, FIN_LANES = 16;
template <int NV>
__device__ __forceinline__ bool finalize_sums(const double* __restrict__ partial, int chunks, int C, double (&tot)[NV], int& c_out) {
    __shared__ double red[NV][FIN_LANES][FIN_COLS];
    const int cl = threadIdx.x % FIN_COLS, rl = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + cl;
    const int per = (chunks + FIN_LANES - 1) / FIN_LANES;
    const int i0 = rl * per, i1 = i0 + per < chunks ? i0 + per : chunks;
    double s[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) s[v] = 0.0;
    if (c < C)
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int v = 0; v < NV; ++v) s[v] += partial[((long)i * NV + v) * C + c];
#pragma unroll
    for (int v = 0; v < NV; ++v) red[v][rl][cl] = s[v];
    __syncthreads();
    c_out = c;
    if (rl != 0 || c >= C) return false;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double a = 0.0;
        for (int l = 0; l < FIN_LANES; ++l) a += red[v][l][cl];
        tot[v] = a;
    }
    return true;
}
static inline dim3 fin_grid(int C) { return dim3((C + FIN_COLS - 1) / FIN_COLS); }
constexpr int FIN_THREADS = FIN_COLS * FIN_LANES;

// partial[(chunk * 2 + {0: sum, 1: sum of squares}) * C + c], deterministic: the finalize kernel adds the chunks in a fixed order
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int ldx, int M, int C, double* __restrict__ partial, int chunk_rows) {
    __shared__ double red[2][STAT_ROWS][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long r0 = (long)blockIdx.x * chunk_rows;
    const long r1 = r0 + chunk_rows < M ? r0 + chunk_rows : M;
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
    double tot[2];
    int c;
    if (!finalize_sums<2>(partial, chunks, C, tot, c)) return;
    const double s = tot[0], ss = tot[1];
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

// rows per partial block: 256 at least, at most ~512 chunks (the finalize kernels add the chunks of a column serially, in order: deterministic)
static inline int stat_rows(int M) {
    int r = ((M + 511) / 512 + 3) & ~3;
    return r < STAT_CHUNK ? STAT_CHUNK : r;
}

extern "C" long emage_bn_stats_workspace_bytes(int M, int C) {
    if (M <= 0 || C <= 0) return EMAGE_EINVAL;
    return (long)((M + STAT_CHUNK - 1) / STAT_CHUNK) * 2 * C * (long)sizeof(double);
}

extern "C" int emage_bn_stats(const float* x, int ldx, int M, int C, void* workspace, long workspace_bytes,
                              float* mean, float* var, float* running_mean, float* running_var, float momentum, void* stream) {
    if (!x || !workspace || !mean || !var || M <= 0 || C <= 0 || ldx < C || !(momentum >= 0.f && momentum <= 1.f)) return EMAGE_EINVAL;
    if (workspace_bytes < emage_bn_stats_workspace_bytes(M, C) || ((uintptr_t)workspace & 7)) return EMAGE_EINVAL;
    const int chunk_rows = stat_rows(M), chunks = (M + chunk_rows - 1) / chunk_rows;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, ldx, M, C, (double*)workspace, chunk_rows);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, fin_grid(C), dim3(FIN_THREADS), 0, s, (const double*)workspace, chunks, M, C, mean, var, running_mean, running_var, momentum);
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
    int* bad = (int*)((double*)workspace + (LOSS_BLOCKS - 1));                 // last workspace slot: out-of-range class index seen (sticky:
                                                                               // the caller hands in a zeroed workspace, ops.loss_workspace)
    hipLaunchKernelGGL(nll_partial_kernel, dim3(blocks), dim3(256), 0, s, logits, ld, index, M, K, (double*)workspace, bad);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, (const double*)workspace, blocks, (double)weight / (double)M, loss);
    return launch_status();
}

// =====================================================================================================================
// Backward building blocks of the training step (first, functional versions: plain fp32 VALU code, float64 reductions;
// the contractions of the Linear layers go through emage_gemm with transposed operands, see pantomatrix_amd/training.py)
// =====================================================================================================================
namespace {

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int M, int N) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, n = n0 + tx;
        tile[r][tx] = (m < M && n < N) ? in[(long)m * ldi + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, m = m0 + tx;
        if (n < N && m < M) out[(long)n * ldo + m] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy, int M, int C,
                                                              double* __restrict__ partial, int chunk_rows) {
    __shared__ double red[STAT_ROWS][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long r0 = (long)blockIdx.x * chunk_rows;
    const long r1 = r0 + chunk_rows < M ? r0 + chunk_rows : M;
    double s = 0.0;
    if (c < C)
        for (long r = r0 + rl; r < r1; r += STAT_ROWS) s += y ? (double)(x[r * ldx + c] * y[r * ldy + c]) : (double)x[r * ldx + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double a = 0.0;
        for (int i = 0; i < STAT_ROWS; ++i) a += red[i][cl];
        partial[(long)blockIdx.x * C + c] = a;
    }
}

__global__ __launch_bounds__(256) void col_sum_finalize_kernel(const double* __restrict__ partial, int chunks, int C, float* __restrict__ out, int accumulate) {
    double tot[1];
    int c;
    if (!finalize_sums<1>(partial, chunks, C, tot, c)) return;
    const double s = tot[0];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// The finalize step of MANY chunked column reductions in one launch (round 5): a training step ends ~600 bias / affine-gradient
// reductions with a finalize launch of 3-48 blocks each (4.8 us apiece: 2.9 ms per step).  Entry e of the device table describes one
// reduction (its float64 partials, chunk count, width, destination, accumulate flag) and owns blocks [block0_e, block0_{e+1}); a block
// finds its entry by a scan of the (<= 64) entries and then runs the arithmetic of col_sum_finalize_kernel: the same bits.
struct FinEntry { const double* partial; float* out; int chunks, C, accumulate, block0; };
constexpr int FIN_MAX = 64;
struct FinTable { FinEntry e[FIN_MAX]; int n; };        // 2 KB: travels BY VALUE in the kernel argument segment (a captured launch keeps its own copy)
__global__ __launch_bounds__(256) void col_sum_finalize_multi_kernel(FinTable tab) {
    __shared__ double red[FIN_LANES][FIN_COLS];
    int e = 0;
    for (int i = 1; i < tab.n; ++i) e += ((int)blockIdx.x >= tab.e[i].block0) ? 1 : 0;      // block0 ascending; scalar work
    e = __builtin_amdgcn_readfirstlane(e);
    const FinEntry& t = tab.e[e];
    const int cl = threadIdx.x % FIN_COLS, rl = threadIdx.x / FIN_COLS;
    const int c = ((int)blockIdx.x - t.block0) * FIN_COLS + cl;
    const int per = (t.chunks + FIN_LANES - 1) / FIN_LANES;
    const int i0 = rl * per, i1 = i0 + per < t.chunks ? i0 + per : t.chunks;
    double s = 0.0;
    if (c < t.C)
        for (int i = i0; i < i1; ++i) s += t.partial[(long)i * t.C + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl != 0 || c >= t.C) return;
    double a = 0.0;
    for (int l = 0; l < FIN_LANES; ++l) a += red[l][cl];
    t.out[c] = t.accumulate ? t.out[c] + (float)a : (float)a;
}

// Everything a Linear's backward needs from the gradient of its output, in ONE pass over dY (64 x 64 tiles through LDS):
//   dpre = dy * (y > 0 ? 1 : slope)            (y given: the activation's backward from its saved output, as act_backward_kernel)
//   out_h (M, n_store)  = EMAGE_H2 image of scale * dpre                  — the A operand of dX = dpre W
//   out_t (C, m_store)  = EMAGE_H2 image of scale * dpre^T, zero tail     — the A operand of dW = dpre^T X (contraction over the rows)
//   partial[tile_m][c]  = float64 column sums of the tile's 64 rows       — the bias gradient (col_sum_finalize_kernel adds the tiles in order)
// instead of act_backward + h2_cast + h2_cast(transpose) + col_sum_partial: four reads of dY and a round trip of dpre become one read.
__global__ __launch_bounds__(256) void grad_prep_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ y, int ldy, float slope,
                                                        emage_dev::h2_t* __restrict__ out_h, int ldh, int n_store, emage_dev::h2_t* __restrict__ out_t, int ldt, int m_store,
                                                        double* __restrict__ partial, float scale, int M, int C) {
    __shared__ float tile[64][65];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const bool vec = (ldd & 3) == 0 && (((uintptr_t)dy & 15) == 0) && c0 + 64 <= C && (!y || ((ldy & 3) == 0 && (((uintptr_t)y & 15) == 0)));
    if (vec) {
        // whole tile columns, 16-byte aligned rows: four (eight with y) 16-byte loads per thread, all in flight before the first LDS write
        // (round 6: the scalar form read 32 dwords per thread one dependent pair at a time — 313 launches, 5 ms per training step)
        const int r0 = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
        float4 d4[4], y4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + 16 * k;
            const bool in = m0 + r < M;
            d4[k] = in ? *(const float4*)(dy + (long)(m0 + r) * ldd + c0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            y4[k] = (in && y) ? *(const float4*)(y + (long)(m0 + r) * ldy + c0 + c4) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + 16 * k;
            float4 v = d4[k];
            if (y) { v.x = v.x * (y4[k].x > 0.f ? 1.f : slope); v.y = v.y * (y4[k].y > 0.f ? 1.f : slope); v.z = v.z * (y4[k].z > 0.f ? 1.f : slope); v.w = v.w * (y4[k].w > 0.f ? 1.f : slope); }
            tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            float v = 0.f;
            if (m0 + r < M && c0 + c < C) {
                v = dy[(long)(m0 + r) * ldd + c0 + c];
                if (y) v = v * (y[(long)(m0 + r) * ldy + c0 + c] > 0.f ? 1.f : slope);
            }
            tile[r][c] = v;
        }
    }
    __syncthreads();
    if (out_h)
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
            const int r = i >> 3, g = i & 7;
            if (m0 + r < M && c0 + 8 * g < n_store) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[r][8 * g + e] * scale;
                emage_dev::h2_store8(out_h + (long)(m0 + r) * ldh + c0 + 8 * g, v);
            }
        }
    if (out_t)
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
            const int c = i >> 3, g = i & 7;
            if (c0 + c < C && m0 + 8 * g < m_store) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[8 * g + e][c] * scale;
                emage_dev::h2_store8(out_t + (long)(c0 + c) * ldt + m0 + 8 * g, v);
            }
        }
    if (partial && threadIdx.x < 64 && c0 + (int)threadIdx.x < C && m0 < M) {
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s += (double)tile[r][threadIdx.x];
        partial[(long)blockIdx.x * C + c0 + threadIdx.x] = s;
    }
}

// dpre = dy * (y > 0 ? 1 : slope): backward of LeakyReLU / ReLU from the saved OUTPUT (same sign as the pre-activation)
__global__ __launch_bounds__(256) void act_backward_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ y, int ldy, float slope,
                                                           float* __restrict__ out, int ldo, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        out[m * ldo + c] = dy[m * ldd + c] * (y[m * ldy + c] > 0.f ? 1.f : slope);
    }
}

// LayerNorm backward, one wave per row: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; also t = dy * xhat
// (its column sums are d gamma; the column sums of dy are d beta)
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma, const float* __restrict__ dy, int ldd,
                                                                 float eps, float* __restrict__ dx, int ldo, float* __restrict__ dyxhat, int ldt, int M, int C) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* xr = x + m * ldx;
    const float* dr = dy + m * ldd;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; v += d * d; }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / C + eps);
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mu) * rstd, g = dr[c] * gamma[c];
        sg += g;
        sgx += g * xh;
    }
    for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o); sgx += __shfl_xor(sgx, o); }
    const float mg = sg / C, mgx = sgx / C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mu) * rstd, g = dr[c] * gamma[c];
        dx[m * ldo + c] = rstd * (g - mg - xh * mgx);
        dyxhat[m * ldt + c] = dr[c] * xh;
    }
}

// LayerNorm backward WITH its affine gradients: a block takes LNB_ROWS (16 or 4) rows (one wave per row at a time, the arithmetic and its order exactly
// those of layernorm_backward_kernel: the same dx bits), every lane keeps float64 running sums of dy * xhat and dy for the columns it owns, the
// four waves add theirs in LDS in wave order, and the block writes ONE partial per column: partial[(block * 2 + {0: sum dy, 1: sum dy xhat}) * C + c].
// ln_bwd_finalize_kernel adds the blocks in order.  Replaces layernorm_backward + 2 x (col_sum_partial + col_sum_finalize): the (M, C) product
// dy * xhat is never written, five launches become two.
constexpr int LNB_MAXJ = 16;                                 // C <= 64 * LNB_MAXJ
template <int LNB_ROWS>                                      // 16: four rows per wave, few partials; 4: one row per wave (the row kernel's parallelism), 4x the partials
__global__ __launch_bounds__(256) void layernorm_backward_affine_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma, const float* __restrict__ dy, int ldd,
                                                                        float eps, float* __restrict__ dx, int ldo, double* __restrict__ partial, int M, int C) {
    __shared__ double red[2][64 * LNB_MAXJ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int J = (C + 63) >> 6;
    double ab[LNB_MAXJ], ag[LNB_MAXJ];
#pragma unroll
    for (int j = 0; j < LNB_MAXJ; ++j) { ab[j] = 0.0; ag[j] = 0.0; }
    for (int r = wave; r < LNB_ROWS; r += 4) {
        const long m = (long)blockIdx.x * LNB_ROWS + r;
        if (m >= M) break;
        const float* xr = x + m * ldx;
        const float* dr = dy + m * ldd;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mu = s / C;
        float v = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; v += d * d; }
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const float rstd = 1.0f / sqrtf(v / C + eps);
        float sg = 0.f, sgx = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float xh = (xr[c] - mu) * rstd, g = dr[c] * gamma[c];
            sg += g;
            sgx += g * xh;
        }
        for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o); sgx += __shfl_xor(sgx, o); }
        const float mg = sg / C, mgx = sgx / C;
#pragma unroll
        for (int j = 0; j < LNB_MAXJ; ++j) {
            const int c = lane + 64 * j;
            if (j < J && c < C) {
                const float xh = (xr[c] - mu) * rstd, g = dr[c] * gamma[c];
                dx[m * ldo + c] = rstd * (g - mg - xh * mgx);
                ab[j] += (double)dr[c];
                ag[j] += (double)(dr[c] * xh);
            }
        }
    }
    for (int w = 0; w < 4; ++w) {                            // the waves add their sums in wave order: a fixed summation order
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < LNB_MAXJ; ++j) {
                const int c = lane + 64 * j;
                if (j < J && c < C) {
                    red[0][c] = (w ? red[0][c] : 0.0) + ab[j];
                    red[1][c] = (w ? red[1][c] : 0.0) + ag[j];
                }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        partial[((long)blockIdx.x * 2 + 0) * C + c] = red[0][c];
        partial[((long)blockIdx.x * 2 + 1) * C + c] = red[1][c];
    }
}

__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const double* __restrict__ partial, int chunks, int C, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    double tot[2];
    int c;
    if (!finalize_sums<2>(partial, chunks, C, tot, c)) return;
    dbeta[c] = accumulate ? dbeta[c] + (float)tot[0] : (float)tot[0];
    dgamma[c] = accumulate ? dgamma[c] + (float)tot[1] : (float)tot[1];
}

struct AttnBwdArgs {
    const float* q; const float* k; const float* vt; const float* pmask; const float* d_out;
    float* dq; float* dk; float* dv;
    int ldq, ldk, ldvt, vt_rows, ld_do, ld_dq, ld_dk, ld_dv, B, H, Tq, Tk, HD;
    float scale;
};

// one block per (batch, head): recomputes P = softmax(scale Q K^T), then dV = (P*mask)^T dO, dP = (dO V^T) * mask,
// dS = P * (dP - rowsum(dP * P)) * scale, dQ = dS K, dK = dS^T Q.  P and dS live in LDS (Tq * Tk floats each).
__global__ __launch_bounds__(256) void attention_backward_kernel(AttnBwdArgs p) {
    extern __shared__ float lds[];
    float* P = lds;
    float* dS = lds + p.Tq * p.Tk;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int Tq = p.Tq, Tk = p.Tk, HD = p.HD;
    const float* Q = p.q + (long)b * Tq * p.ldq + h * HD;
    const float* K = p.k + (long)b * Tk * p.ldk + h * HD;
    const float* VT = p.vt + ((long)b * p.vt_rows + h * HD) * p.ldvt;           // VT[d * ldvt + j]
    const float* dO = p.d_out + (long)b * Tq * p.ld_do + h * HD;
    const float* MK = p.pmask ? p.pmask + ((long)b * p.H + h) * Tq * Tk : nullptr;
    const int n_pairs = Tq * Tk;
    for (int e = threadIdx.x; e < n_pairs; e += blockDim.x) {
        const int i = e / Tk, j = e - i * Tk;
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < HD; ++d) {
            s = fmaf(Q[(long)i * p.ldq + d], K[(long)j * p.ldk + d], s);
            dp = fmaf(dO[(long)i * p.ld_do + d], VT[(long)d * p.ldvt + j], dp);
        }
        P[e] = s * p.scale;
        dS[e] = MK ? dp * MK[e] : dp;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Tq; i += blockDim.x) {
        float mx = -INFINITY;
        for (int j = 0; j < Tk; ++j) mx = fmaxf(mx, P[i * Tk + j]);
        float sum = 0.f;
        for (int j = 0; j < Tk; ++j) { const float e = expf(P[i * Tk + j] - mx); P[i * Tk + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        float dsum = 0.f;
        for (int j = 0; j < Tk; ++j) { P[i * Tk + j] *= inv; dsum = fmaf(dS[i * Tk + j], P[i * Tk + j], dsum); }
        for (int j = 0; j < Tk; ++j) dS[i * Tk + j] = P[i * Tk + j] * (dS[i * Tk + j] - dsum) * p.scale;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Tq * HD; e += blockDim.x) {                 // dQ[i][d] = sum_j dS[i][j] K[j][d]
        const int i = e / HD, d = e - i * HD;
        float a = 0.f;
        for (int j = 0; j < Tk; ++j) a = fmaf(dS[i * Tk + j], K[(long)j * p.ldk + d], a);
        p.dq[((long)b * Tq + i) * p.ld_dq + h * HD + d] = a;
    }
    for (int e = threadIdx.x; e < Tk * HD; e += blockDim.x) {                 // dK[j][d] = sum_i dS[i][j] Q[i][d];  dV[j][d] = sum_i P[i][j] mask[i][j] dO[i][d]
        const int j = e / HD, d = e - j * HD;
        float a = 0.f, c = 0.f;
        for (int i = 0; i < Tq; ++i) {
            a = fmaf(dS[i * Tk + j], Q[(long)i * p.ldq + d], a);
            const float pm = MK ? P[i * Tk + j] * MK[i * Tk + j] : P[i * Tk + j];
            c = fmaf(pm, dO[(long)i * p.ld_do + d], c);
        }
        p.dk[((long)b * Tk + j) * p.ld_dk + h * HD + d] = a;
        p.dv[((long)b * Tk + j) * p.ld_dv + h * HD + d] = c;
    }
}

// ---- the same backward on the matrix cores: Tq = Tk = 64, head_dim = 192 (every attention of a training step) ----------------------------
// One block (4 waves) per (batch, head); the five contractions run as exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain) on
// operands staged in LDS: wave w owns rows 16 w .. 16 w + 15 of every product.  LDS plan (floats): X, Y = 64 x 194 / 192 x 66 operand
// buffers (re-filled per phase), P, dS, dS^T = 64 x 66.  1.44 ms -> tens of microseconds per launch (profiles/r03_train_kernel_stats_*.csv).
constexpr int AB_LD = 194, AB_LS = 66, AB_BIG = 192 * 66, AB_SMALL = 64 * 66;

// acc (16 x 16 tile at rows row0.., cols n0..) = sum_k A[row][k] * B(k, n); A row-major (k contiguous); B_KN: B stored [k][n], else [n][k]
template <bool B_KN>
__device__ __forceinline__ f32x4 ab_tile(const float* __restrict__ A, int lda, int row0, const float* __restrict__ B, int ldb, int n0, int K, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
    const float* ap = A + (row0 + r) * lda + kq;
    const float* bp = B_KN ? B + kq * ldb + n0 + r : B + (n0 + r) * ldb + kq;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = ap[k0];
        const float b = B_KN ? bp[k0 * ldb] : bp[k0];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(256) void attention_backward_mfma_kernel(AttnBwdArgs p) {
    extern __shared__ float lds[];
    float* X = lds;
    float* Y = lds + AB_BIG;
    float* P = Y + AB_BIG;
    float* Ds = P + AB_SMALL;
    float* DsT = Ds + AB_SMALL;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int T = 64, HD = 192;
    const float* Q = p.q + (long)b * T * p.ldq + h * HD;
    const float* K = p.k + (long)b * T * p.ldk + h * HD;
    const float* VT = p.vt + ((long)b * p.vt_rows + h * HD) * p.ldvt;
    const float* dO = p.d_out + (long)b * T * p.ld_do + h * HD;
    const float* MK = p.pmask ? p.pmask + ((long)b * p.H + h) * T * T : nullptr;
    auto load_rows = [&](float* dst, const float* src, int ld_src) {            // 64 rows x 192 floats -> [64][AB_LD]
        for (int idx = tid; idx < T * (HD / 4); idx += 256) {
            const int row = idx / (HD / 4), c4 = idx - row * (HD / 4);
            const float4 v = *(const float4*)(src + (long)row * ld_src + 4 * c4);
            float* d = dst + row * AB_LD + 4 * c4;
            *(float2*)d = make_float2(v.x, v.y);
            *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
    };
    const int row0 = 16 * wave, cr = lane & 15, rq = 4 * (lane >> 4);          // a lane's tile entries: rows row0 + rq + r, column n0 + cr
    // ---- S = scale * Q K^T -> P ----
    load_rows(X, Q, p.ldq);
    load_rows(Y, K, p.ldk);
    __syncthreads();
    for (int n0 = 0; n0 < T; n0 += 16) {
        const f32x4 acc = ab_tile<false>(X, AB_LD, row0, Y, AB_LD, n0, HD, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(row0 + rq + r) * AB_LS + n0 + cr] = acc[r] * p.scale;
    }
    __syncthreads();
    // ---- dP = dO V -> Ds ----
    load_rows(X, dO, p.ld_do);
    for (int idx = tid; idx < HD * (T / 4); idx += 256) {                       // V^T (192 x 64) -> [192][AB_LS]
        const int d = idx / (T / 4), c4 = idx - d * (T / 4);
        const float4 v = *(const float4*)(VT + (long)d * p.ldvt + 4 * c4);
        float* dst = Y + d * AB_LS + 4 * c4;
        *(float2*)dst = make_float2(v.x, v.y);
        *(float2*)(dst + 2) = make_float2(v.z, v.w);
    }
    __syncthreads();
    for (int n0 = 0; n0 < T; n0 += 16) {
        const f32x4 acc = ab_tile<true>(X, AB_LD, row0, Y, AB_LS, n0, HD, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ds[(row0 + rq + r) * AB_LS + n0 + cr] = acc[r];
    }
    __syncthreads();
    // ---- softmax rows, dS = P (dP mask - rowsum(dP mask P)) scale; dS, dS^T, (P mask)^T (into Y: V^T is done) ----
    for (int i = row0; i < row0 + 16; ++i) {                                    // one row per iteration, lane = key j
        const float sv = P[i * AB_LS + lane];
        float mx = sv;
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = expf(sv - mx);
        float sum = e;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float pr = e * (1.0f / sum);
        const float mk = MK ? MK[i * T + lane] : 1.0f;
        const float dpm = Ds[i * AB_LS + lane] * mk;
        float dsum = dpm * pr;
        for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
        const float ds = pr * (dpm - dsum) * p.scale;
        Ds[i * AB_LS + lane] = ds;
        DsT[lane * AB_LS + i] = ds;
        Y[lane * AB_LS + i] = pr * mk;
    }
    __syncthreads();
    // ---- dV = (P mask)^T dO ----
    for (int n0 = 0; n0 < HD; n0 += 16) {
        const f32x4 acc = ab_tile<true>(Y, AB_LS, row0, X, AB_LD, n0, T, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) p.dv[((long)b * T + row0 + rq + r) * p.ld_dv + h * HD + n0 + cr] = acc[r];
    }
    __syncthreads();
    // ---- dQ = dS K, dK = dS^T Q ----
    load_rows(X, Q, p.ldq);
    load_rows(Y, K, p.ldk);
    __syncthreads();
    for (int n0 = 0; n0 < HD; n0 += 16) {
        const f32x4 aq = ab_tile<true>(Ds, AB_LS, row0, Y, AB_LD, n0, T, lane);
        const f32x4 ak = ab_tile<true>(DsT, AB_LS, row0, X, AB_LD, n0, T, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p.dq[((long)b * T + row0 + rq + r) * p.ld_dq + h * HD + n0 + cr] = aq[r];
            p.dk[((long)b * T + row0 + rq + r) * p.ld_dk + h * HD + n0 + cr] = ak[r];
        }
    }
}

__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ target, int ldt, float scale,
                                                       float* __restrict__ out, int ldo, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        out[m * ldo + c] = scale * (pred[m * ldp + c] - target[m * ldt + c]);
    }
}

// out[m][k] = scale * (softmax(logits[m])[k] - [k == index[m]]): one wave per row
__global__ __launch_bounds__(256) void nll_grad_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ index, float scale,
                                                       float* __restrict__ out, int ldo, int M, int K) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* x = logits + m * ld;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, x[k]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f;
    for (int k = lane; k < K; k += 64) se += expf(x[k] - mx);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    const int64_t t = index[m];
    for (int k = lane; k < K; k += 64) out[m * ldo + k] = scale * (expf(x[k] - mx) / se - (k == t ? 1.f : 0.f));
}

}  // namespace

extern "C" int emage_transpose_f32(const float* in, int ld_in, float* out, int ld_out, int M, int N, void* stream) {
    if (!in || !out || M <= 0 || N <= 0 || ld_in < N || ld_out < M) return EMAGE_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, ld_in, out, ld_out, M, N);
    return launch_status();
}

extern "C" int emage_col_sum(const float* x, int ldx, const float* y, int ldy, int M, int C, float* out, int accumulate,
                             void* workspace, long workspace_bytes, void* stream) {
    if (!x || !workspace || M <= 0 || C <= 0 || ldx < C || (y && ldy < C) || ((uintptr_t)workspace & 7)) return EMAGE_EINVAL;
    const int chunk_rows = stat_rows(M), chunks = (M + chunk_rows - 1) / chunk_rows;
    if (workspace_bytes < (long)chunks * C * (long)sizeof(double)) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, ldx, y, ldy, M, C, (double*)workspace, chunk_rows);
    int rc = launch_status();
    if (rc || !out) return rc;                       // out == NULL: the partials only (emage_col_sum_chunks of them; emage_col_sum_finalize_multi ends them)
    hipLaunchKernelGGL(col_sum_finalize_kernel, fin_grid(C), dim3(FIN_THREADS), 0, s, (const double*)workspace, chunks, C, out, accumulate);
    return launch_status();
}

extern "C" int emage_col_sum_chunks(int M) {
    if (M <= 0) return EMAGE_EINVAL;
    const int chunk_rows = stat_rows(M);
    return (M + chunk_rows - 1) / chunk_rows;
}

extern "C" int emage_col_sum_finalize_multi(const emage_finalize_entry* entries, int n_entries, void* stream) {
    if (!entries || n_entries <= 0 || n_entries > FIN_MAX) return EMAGE_EINVAL;
    FinTable tab;
    int blocks = 0;
    for (int i = 0; i < n_entries; ++i) {
        const emage_finalize_entry& q = entries[i];
        if (!q.partial || !q.out || q.chunks <= 0 || q.C <= 0 || ((uintptr_t)q.partial & 7)) return EMAGE_EINVAL;
        tab.e[i] = FinEntry{q.partial, q.out, q.chunks, q.C, q.accumulate ? 1 : 0, blocks};
        blocks += (q.C + FIN_COLS - 1) / FIN_COLS;
    }
    for (int i = n_entries; i < FIN_MAX; ++i) tab.e[i] = tab.e[0];
    tab.n = n_entries;
    hipLaunchKernelGGL(col_sum_finalize_multi_kernel, dim3(blocks), dim3(FIN_THREADS), 0, (hipStream_t)stream, tab);
    return launch_status();
}

extern "C" int emage_grad_prep(const float* dy, int ld_dy, const float* y, int ld_y, float slope, int M, int C, float scale,
                               void* out_h, int ldh, int n_store, void* out_t, int ldt, int m_store,
                               float* bias_grad, int accumulate, void* workspace, long workspace_bytes, void* stream) {
    if (!dy || M <= 0 || C <= 0 || ld_dy < C || (y && ld_y < C) || !(scale > 0.f) || (!out_h && !out_t && !bias_grad && !workspace)) return EMAGE_EINVAL;
    if (out_h && (n_store < C || n_store % 8 || ldh % 8 || ldh < n_store || ((uintptr_t)out_h & 15))) return EMAGE_EINVAL;
    if (out_t && (m_store < M || m_store % 8 || ldt % 8 || ldt < m_store || ((uintptr_t)out_t & 15))) return EMAGE_EINVAL;
    const int tiles_m = ((out_t && m_store > M ? m_store : M) + 63) / 64, chunks = (M + 63) / 64;
    const int tiles_c = ((out_h && n_store > C ? n_store : C) + 63) / 64;
    if (bias_grad && !workspace) return EMAGE_EINVAL;
    if (workspace && (((uintptr_t)workspace & 7) || workspace_bytes < (long)chunks * C * (long)sizeof(double))) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // workspace without bias_grad (round 5): the column partials only — ceil(M / 64) chunks of C float64 — for emage_col_sum_finalize_multi
    hipLaunchKernelGGL(grad_prep_kernel, dim3(tiles_m, tiles_c), dim3(256), 0, s, dy, ld_dy, y, ld_y, slope, (emage_dev::h2_t*)out_h, ldh, n_store, (emage_dev::h2_t*)out_t, ldt, m_store,
                       (double*)workspace, scale, M, C);
    int rc = launch_status();
    if (rc || !bias_grad) return rc;
    hipLaunchKernelGGL(col_sum_finalize_kernel, fin_grid(C), dim3(FIN_THREADS), 0, s, (const double*)workspace, chunks, C, bias_grad, accumulate);
    return launch_status();
}

extern "C" int emage_act_backward(const float* dy, int ld_dy, const float* y, int ld_y, float slope, float* out, int ldo, int M, int C, void* stream) {
    if (!dy || !y || !out || M <= 0 || C <= 0 || ld_dy < C || ld_y < C || ldo < C) return EMAGE_EINVAL;
    hipLaunchKernelGGL(act_backward_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, y, ld_y, slope, out, ldo, M, C);
    return launch_status();
}

extern "C" int emage_layernorm_backward(const float* x, int ldx, const float* gamma, const float* dy, int ld_dy, float eps,
                                        float* dx, int ld_dx, float* dy_xhat, int ld_t, int M, int C, void* stream) {
    if (!x || !gamma || !dy || !dx || !dy_xhat || M <= 0 || C <= 0 || ldx < C || ld_dy < C || ld_dx < C || ld_t < C) return EMAGE_EINVAL;
    hipLaunchKernelGGL(layernorm_backward_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, dy, ld_dy, eps, dx, ld_dx, dy_xhat, ld_t, M, C);
    return launch_status();
}

extern "C" long emage_layernorm_backward_affine_workspace_bytes(int M, int C, int rows_per_block) {
    if (M <= 0 || C <= 0 || (rows_per_block != 4 && rows_per_block != 16)) return EMAGE_EINVAL;
    return (long)((M + rows_per_block - 1) / rows_per_block) * 2 * C * (long)sizeof(double);
}

extern "C" int emage_layernorm_backward_affine(const float* x, int ldx, const float* gamma, const float* dy, int ld_dy, float eps, float* dx, int ld_dx,
                                               float* dgamma, float* dbeta, int accumulate, int M, int C, int rows_per_block,
                                               void* workspace, long workspace_bytes, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !workspace || M <= 0 || C <= 0 || C > 64 * LNB_MAXJ || ldx < C || ld_dy < C || ld_dx < C || ((uintptr_t)workspace & 7))
        return EMAGE_EINVAL;
    if (rows_per_block != 4 && rows_per_block != 16) return EMAGE_EINVAL;
    const int chunks = (M + rows_per_block - 1) / rows_per_block;
    if (workspace_bytes < (long)chunks * 2 * C * (long)sizeof(double)) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (rows_per_block == 4)
        hipLaunchKernelGGL(layernorm_backward_affine_kernel<4>, dim3(chunks), dim3(256), 0, s, x, ldx, gamma, dy, ld_dy, eps, dx, ld_dx, (double*)workspace, M, C);
    else
        hipLaunchKernelGGL(layernorm_backward_affine_kernel<16>, dim3(chunks), dim3(256), 0, s, x, ldx, gamma, dy, ld_dy, eps, dx, ld_dx, (double*)workspace, M, C);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, fin_grid(C), dim3(FIN_THREADS), 0, s, (const double*)workspace, chunks, C, dgamma, dbeta, accumulate);
    return launch_status();
}

extern "C" int emage_attention_backward(const float* q, int ldq, const float* k, int ldk, const float* vt, int ldvt, int vt_rows, const float* pmask,
                                        const float* d_out, int ld_do, float* dq, int ld_dq, float* dk, int ld_dk, float* dv, int ld_dv,
                                        int B, int H, int Tq, int Tk, int hd, void* stream) {
    if (!q || !k || !vt || !d_out || !dq || !dk || !dv || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || hd <= 0 || vt_rows < H * hd || ldvt < Tk) return EMAGE_EINVAL;
    const size_t lds = (size_t)2 * Tq * Tk * sizeof(float);
    if (lds > 144 * 1024) return EMAGE_EINVAL;
    static const hipError_t configured = hipFuncSetAttribute((const void*)attention_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (configured != hipSuccess) return (int)configured;
    AttnBwdArgs a{q, k, vt, pmask, d_out, dq, dk, dv, ldq, ldk, ldvt, vt_rows, ld_do, ld_dq, ld_dk, ld_dv, B, H, Tq, Tk, hd, 1.0f / sqrtf((float)hd)};
    const bool aligned = !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)d_out) & 15) && ldq % 4 == 0 && ldk % 4 == 0 && ldvt % 4 == 0 && ld_do % 4 == 0;
    if (Tq == 64 && Tk == 64 && hd == 192 && aligned) {          // the shape of every attention of a training step: matrix-core form
        constexpr size_t lds_mfma = (size_t)(2 * AB_BIG + 3 * AB_SMALL) * sizeof(float);
        static const hipError_t conf2 = hipFuncSetAttribute((const void*)attention_backward_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (conf2 != hipSuccess) return (int)conf2;
        hipLaunchKernelGGL(attention_backward_mfma_kernel, dim3(B * H), dim3(256), lds_mfma, (hipStream_t)stream, a);
        return launch_status();
    }
    hipLaunchKernelGGL(attention_backward_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int emage_mse_loss_grad(const float* pred, int ld_pred, const float* target, int ld_target, int M, int C, float weight,
                                   float* grad, int ld_grad, void* stream) {
    if (!pred || !target || !grad || M <= 0 || C <= 0 || ld_pred < C || ld_target < C || ld_grad < C) return EMAGE_EINVAL;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, pred, ld_pred, target, ld_target,
                       (float)(2.0 * (double)weight / ((double)M * C)), grad, ld_grad, M, C);
    return launch_status();
}

extern "C" int emage_nll_loss_grad(const float* logits, int ld, const int64_t* index, int M, int K, float weight, float* grad, int ld_grad, void* stream) {
    if (!logits || !index || !grad || M <= 0 || K <= 0 || ld < K || ld_grad < K) return EMAGE_EINVAL;
    hipLaunchKernelGGL(nll_grad_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, index, (float)((double)weight / M), grad, ld_grad, M, K);
    return launch_status();
}

// torch.optim.Adam (no amsgrad, weight decay folded into the gradient when non-zero), one flat tensor per call (T:258-265):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                                   float b1, float b2, float step_size, float inv_sqrt_bias2, float eps, float weight_decay) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (weight_decay != 0.f) gi += weight_decay * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bias2 + eps);
    }
}
}  // namespace

extern "C" int emage_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, int step,
                               float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return EMAGE_EINVAL;
    const double bias1 = 1.0 - pow((double)beta1, step), bias2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, beta1, beta2,
                       (float)((double)lr / bias1), (float)(1.0 / sqrt(bias2)), eps, weight_decay);
    return launch_status();
}

// ---- convolution / BatchNorm backward pieces --------------------------------------------------------------------------------------
namespace {

// colT[(tap * C + c)][m] = X[seq * Lin + l * stride - pad + tap][c] (0 outside the sequence), m = seq * Lout + l; rows m >= M stay 0.
// One 32 x 32 (m, c) tile per block and tap through LDS: coalesced reads along c, coalesced writes along m.
__global__ __launch_bounds__(256) void im2col_t_kernel(const float* __restrict__ x, int ldx, int C, int taps, int stride, int pad, int Lin, int Lout, int M,
                                                       float* __restrict__ out, long ld_out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ctiles = (C + 31) / 32;
    const int tap = blockIdx.y / ctiles, c0 = (blockIdx.y % ctiles) * 32, m0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, c = c0 + tx;
        float v = 0.f;
        if (m < M && c < C) {
            const int seq = m / Lout, l = m - seq * Lout, pos = l * stride - pad + tap;
            if (pos >= 0 && pos < Lin) v = x[((long)seq * Lin + pos) * ldx + c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, m = m0 + tx;
        if (c < C && m < M) out[((long)tap * C + c) * ld_out + m] = tile[tx][r];
    }
}

// The same matrix as an EMAGE_H2 image (csrc/h2.h; the W operand of the split-fp16 dW contraction): 64 (m) x 32 (c) tiles, every 32-byte
// group of 8 consecutive m written whole — columns [M, m_store) come out as zeros, no pre-clearing of the (large) buffer.
__global__ __launch_bounds__(256) void im2col_t_h2_kernel(const float* __restrict__ x, int ldx, int C, int taps, int stride, int pad, int Lin, int Lout, int M,
                                                          emage_dev::h2_t* __restrict__ out, long ld_out) {
    __shared__ float tile[64][33];
    const int ctiles = (C + 31) / 32;
    const int tap = blockIdx.y / ctiles, c0 = (blockIdx.y % ctiles) * 32, m0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int r = i >> 5, tx = i & 31;
        const int m = m0 + r, c = c0 + tx;
        float v = 0.f;
        if (m < M && c < C) {
            const int seq = m / Lout, l = m - seq * Lout, pos = l * stride - pad + tap;
            if (pos >= 0 && pos < Lin) v = x[((long)seq * Lin + pos) * ldx + c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    const int c = threadIdx.x >> 3, g = threadIdx.x & 7;
    if (c0 + c < C) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[8 * g + e][c];
        emage_dev::h2_store8(out + ((long)tap * C + c0 + c) * ld_out + m0 + 8 * g, v);
    }
}

// dx[seq * Lin + r][c] = sum over taps with (r + pad - tap) = l * stride, 0 <= l < Lout, of dcol[seq * Lout + l][tap * C + c]
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, long ld, int C, int taps, int stride, int pad, int Lin, int Lout, int nseq,
                                                     float* __restrict__ dx, int ldx) {
    const long total = (long)nseq * Lin * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / C;
        const int c = (int)(i - row * C);
        const int seq = (int)(row / Lin), r = (int)(row - (long)seq * Lin);
        float s = 0.f;
        for (int tap = 0; tap < taps; ++tap) {
            const int q = r + pad - tap;
            if (q < 0 || q % stride) continue;
            const int l = q / stride;
            if (l < Lout) s += dcol[((long)seq * Lout + l) * ld + tap * C + c];
        }
        dx[row * ldx + c] = s;
    }
}

// BatchNorm (training) backward: sums of dy and dy * xhat per channel (float64 partials), then
// dx = gamma * rstd * (dy - sum_dy / M - xhat * sum_dyxhat / M); dgamma = sum_dyxhat, dbeta = sum_dy
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                             const float* __restrict__ dy, int ldd, int M, int C, double* __restrict__ partial, int chunk_rows) {
    __shared__ double red[2][STAT_ROWS][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long r0 = (long)blockIdx.x * chunk_rows;
    const long r1 = r0 + chunk_rows < M ? r0 + chunk_rows : M;
    double s = 0.0, sx = 0.0;
    if (c < C) {
        const float mu = mean[c], rstd = 1.0f / sqrtf(var[c] + eps);
        for (long r = r0 + rl; r < r1; r += STAT_ROWS) {
            const float d = dy[r * ldd + c];
            s += (double)d;
            sx += (double)(d * ((x[r * ldx + c] - mu) * rstd));
        }
    }
    red[0][rl][cl] = s;
    red[1][rl][cl] = sx;
    __syncthreads();
    if (rl == 0 && c < C) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < STAT_ROWS; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
        partial[((long)blockIdx.x * 2 + 0) * C + c] = a;
        partial[((long)blockIdx.x * 2 + 1) * C + c] = b;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int chunks, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    double tot[2];
    int c;
    if (!finalize_sums<2>(partial, chunks, C, tot, c)) return;
    dbeta[c] = (float)tot[0];
    dgamma[c] = (float)tot[1];
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ dy, int ldd,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* __restrict__ dx, int ldo, int M, int C) {
    const long total = (long)M * C;
    const float inv_m = 1.0f / (float)M;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        const float rstd = 1.0f / sqrtf(var[c] + eps);
        const float xh = (x[m * ldx + c] - mean[c]) * rstd;
        dx[m * ldo + c] = gamma[c] * rstd * (dy[m * ldd + c] - dbeta[c] * inv_m - xh * dgamma[c] * inv_m);
    }
}

// first WavEncoder layer (Cin = 1): dW[c][tap] = sum_m dy[m][c] * wav[seq][l * stride - pad + tap]; float64 block partials
constexpr int WIN_CHUNK = 1024;
__global__ __launch_bounds__(256) void wav_in_dw_partial_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ wav, long ldw, int L,
                                                                int Lout, int M, int C, int taps, int stride, int pad, double* __restrict__ partial) {
    const long r0 = (long)blockIdx.x * WIN_CHUNK;
    const long r1 = r0 + WIN_CHUNK < M ? r0 + WIN_CHUNK : M;
    for (int e = threadIdx.x; e < C * taps; e += blockDim.x) {
        const int c = e / taps, tap = e - c * taps;
        double s = 0.0;
        for (long m = r0; m < r1; ++m) {
            const int seq = (int)(m / Lout), l = (int)(m - (long)seq * Lout), pos = l * stride - pad + tap;
            if (pos >= 0 && pos < L) s += (double)(dy[m * ldd + c] * wav[(long)seq * ldw + pos]);
        }
        partial[(long)blockIdx.x * C * taps + e] = s;
    }
}

__global__ __launch_bounds__(256) void wav_in_dw_finalize_kernel(const double* __restrict__ partial, int chunks, int n, float* __restrict__ dw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
    for (int i = 0; i < chunks; ++i) s += partial[(long)i * n + e];
    dw[e] = (float)s;
}

}  // namespace

extern "C" int emage_im2col_t(const float* x, int ldx, int C, int taps, int stride, int pad, int Lin, int Lout, int nseq,
                              float* out, long ld_out, void* stream) {
    if (!x || !out || C <= 0 || taps <= 0 || stride <= 0 || pad < 0 || Lin <= 0 || Lout <= 0 || nseq <= 0 || ldx < C) return EMAGE_EINVAL;
    const long M = (long)nseq * Lout;
    if (ld_out < M || M >= (1L << 31)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(im2col_t_kernel, dim3((unsigned)((M + 31) / 32), (unsigned)(taps * ((C + 31) / 32))), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, C, taps, stride, pad, Lin, Lout, (int)M, out, ld_out);
    return launch_status();
}

extern "C" int emage_im2col_t_h2(const float* x, int ldx, int C, int taps, int stride, int pad, int Lin, int Lout, int nseq,
                                 void* out, long ld_out, void* stream) {
    if (!x || !out || C <= 0 || taps <= 0 || stride <= 0 || pad < 0 || Lin <= 0 || Lout <= 0 || nseq <= 0 || ldx < C) return EMAGE_EINVAL;
    const long M = (long)nseq * Lout;
    const long mp = (M + 63) / 64 * 64;
    if (ld_out < mp || ld_out % 8 || M >= (1L << 31) || ((uintptr_t)out & 15)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(im2col_t_h2_kernel, dim3((unsigned)(mp / 64), (unsigned)(taps * ((C + 31) / 32))), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, C, taps, stride, pad, Lin, Lout, (int)M, (emage_dev::h2_t*)out, ld_out);
    return launch_status();
}

extern "C" int emage_col2im(const float* dcol, long ld, int C, int taps, int stride, int pad, int Lin, int Lout, int nseq, float* dx, int ldx, void* stream) {
    if (!dcol || !dx || C <= 0 || taps <= 0 || stride <= 0 || pad < 0 || Lin <= 0 || Lout <= 0 || nseq <= 0 || ldx < C || ld < (long)taps * C) return EMAGE_EINVAL;
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long)nseq * Lin * C)), dim3(256), 0, (hipStream_t)stream, dcol, ld, C, taps, stride, pad, Lin, Lout, nseq, dx, ldx);
    return launch_status();
}

extern "C" int emage_bn_backward(const float* x, int ldx, const float* mean, const float* var, const float* gamma, float eps, const float* dy, int ld_dy,
                                 float* dx, int ld_dx, float* dgamma, float* dbeta, int M, int C, void* workspace, long workspace_bytes, void* stream) {
    if (!x || !mean || !var || !gamma || !dy || !dx || !dgamma || !dbeta || !workspace || M <= 0 || C <= 0 || ldx < C || ld_dy < C || ld_dx < C) return EMAGE_EINVAL;
    if (workspace_bytes < emage_bn_stats_workspace_bytes(M, C) || ((uintptr_t)workspace & 7)) return EMAGE_EINVAL;
    const int chunk_rows = stat_rows(M), chunks = (M + chunk_rows - 1) / chunk_rows;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, ldx, mean, var, eps, dy, ld_dy, M, C, (double*)workspace, chunk_rows);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, fin_grid(C), dim3(FIN_THREADS), 0, s, (const double*)workspace, chunks, C, dgamma, dbeta);
    rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, s, x, ldx, mean, var, eps, gamma, dy, ld_dy, dgamma, dbeta, dx, ld_dx, M, C);
    return launch_status();
}

extern "C" long emage_wav_conv_in_backward_workspace_bytes(int M, int C, int taps) {
    if (M <= 0 || C <= 0 || taps <= 0) return EMAGE_EINVAL;
    return (long)((M + WIN_CHUNK - 1) / WIN_CHUNK) * C * taps * (long)sizeof(double);
}

extern "C" int emage_wav_conv_in_backward(const float* dy, int ld_dy, const float* wav, long ldw, int L, int B, int Lout, int C, int taps, int stride, int pad,
                                          float* dw, void* workspace, long workspace_bytes, void* stream) {
    if (!dy || !wav || !dw || !workspace || B <= 0 || L <= 0 || Lout <= 0 || C <= 0 || taps <= 0 || stride <= 0 || ld_dy < C || ldw < L) return EMAGE_EINVAL;
    const long M = (long)B * Lout;
    if (M >= (1L << 31) || workspace_bytes < emage_wav_conv_in_backward_workspace_bytes((int)M, C, taps) || ((uintptr_t)workspace & 7)) return EMAGE_EINVAL;
    const int chunks = (int)((M + WIN_CHUNK - 1) / WIN_CHUNK);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wav_in_dw_partial_kernel, dim3(chunks), dim3(256), 0, s, dy, ld_dy, wav, ldw, L, Lout, (int)M, C, taps, stride, pad, (double*)workspace);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(wav_in_dw_finalize_kernel, dim3((C * taps + 255) / 256), dim3(256), 0, s, (const double*)workspace, chunks, C * taps, dw);
    return launch_status();
}

// The two halves of emage_bn_backward on their own — nn.SyncBatchNorm's backward all-reduces the two per-channel sums between them
// (train_emage_audio.py:248): `count` is then the GLOBAL number of rows the statistics were taken over.
extern "C" int emage_bn_backward_sums(const float* x, int ldx, const float* mean, const float* var, float eps, const float* dy, int ld_dy,
                                      float* sum_dy_xhat, float* sum_dy, int M, int C, void* workspace, long workspace_bytes, void* stream) {
    if (!x || !mean || !var || !dy || !sum_dy_xhat || !sum_dy || !workspace || M <= 0 || C <= 0 || ldx < C || ld_dy < C) return EMAGE_EINVAL;
    if (workspace_bytes < emage_bn_stats_workspace_bytes(M, C) || ((uintptr_t)workspace & 7)) return EMAGE_EINVAL;
    const int chunk_rows = stat_rows(M), chunks = (M + chunk_rows - 1) / chunk_rows;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, ldx, mean, var, eps, dy, ld_dy, M, C, (double*)workspace, chunk_rows);
    const int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, fin_grid(C), dim3(FIN_THREADS), 0, s, (const double*)workspace, chunks, C, sum_dy_xhat, sum_dy);
    return launch_status();
}

namespace {
__global__ __launch_bounds__(256) void bn_bwd_apply_n_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ dy, int ldd,
                                                             const float* __restrict__ sdx, const float* __restrict__ sd, float inv_n, float* __restrict__ dx, int ldo, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        const float rstd = 1.0f / sqrtf(var[c] + eps);
        const float xh = (x[m * ldx + c] - mean[c]) * rstd;
        dx[m * ldo + c] = gamma[c] * rstd * (dy[m * ldd + c] - sd[c] * inv_n - xh * sdx[c] * inv_n);
    }
}
}  // namespace

extern "C" int emage_bn_backward_apply(const float* x, int ldx, const float* mean, const float* var, const float* gamma, float eps, const float* dy, int ld_dy,
                                       const float* sum_dy_xhat, const float* sum_dy, long count, float* dx, int ld_dx, int M, int C, void* stream) {
    if (!x || !mean || !var || !gamma || !dy || !sum_dy_xhat || !sum_dy || !dx || M <= 0 || C <= 0 || count < M || ldx < C || ld_dy < C || ld_dx < C) return EMAGE_EINVAL;
    hipLaunchKernelGGL(bn_bwd_apply_n_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, mean, var, eps, gamma, dy, ld_dy,
                       sum_dy_xhat, sum_dy, 1.0f / (float)count, dx, ld_dx, M, C);
    return launch_status();
}

// emage_adam_step with the step count read from device memory (a captured hipGraph replays with fixed kernel arguments: the count
// must advance on the device); the same arithmetic as emage_adam_step.
namespace {
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                                       const int* __restrict__ step, float lr, float b1, float b2, float eps, float weight_decay) {
    const int t = *step;
    const double bias1 = 1.0 - pow((double)b1, t), bias2 = 1.0 - pow((double)b2, t);
    const float step_size = (float)((double)lr / bias1), inv_sqrt_bias2 = (float)(1.0 / sqrt(bias2));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (weight_decay != 0.f) gi += weight_decay * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bias2 + eps);
    }
}
}  // namespace

extern "C" int emage_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, const int* step,
                                   float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || n <= 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay);
    return launch_status();
}

// ---- device-side dropout masks -------------------------------------------------------------------------------------------------------
// nn.Dropout's keep mask drawn ON THE DEVICE: out[i] = bernoulli(1 - p) / (1 - p) from Philox4x32-10 (Salmon et al., SC'11), a
// counter-based generator, so a mask is a pure function of (seed, step, mask id, element index) — no generator state to carry, no
// host tensors, safe inside a captured hipGraph (the step is read from device memory):
//   key = (seed lo, seed hi);  counter = (i / 4 lo, i / 4 hi, mask id, step);  element i takes word i % 4 of the block;
//   keep  <=>  (word >> 8) * 2^-24 >= p      (24 random bits, uniform on [0, 1))
// The stream is this library's own (documented here, pinned bit for bit by tests/test_train_rng.py against a numpy Philox); it is
// NOT torch's fused-dropout stream: runs are reproducible for a seed, distribution-identical to the reference, not draw-identical.
namespace {
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// out = a * keep (+ b) with the keep mask DRAWN IN PLACE: element (mask row mr, column c) is element mr * C + c of the mask
// dropout_mask_kernel would have written for the same key — the same bits as emage_dropout_mask + emage_mul_add, without the mask in
// memory (a training forward holds ~0.9 GB of (T, B, d) masks for its backward otherwise; the backward draws them again from the key).
__global__ __launch_bounds__(256) void mul_add_philox_kernel(const float* __restrict__ a, int lda, float p, float keep_value, unsigned seed_lo, unsigned seed_hi,
                                                             unsigned mask_id, const int* __restrict__ step_dev, int step, int t_rows,
                                                             const float* __restrict__ b, int ldb, float* __restrict__ out, int ldo, int M, int C) {
    const unsigned st = step_dev ? (unsigned)*step_dev : (unsigned)step;
    const int c4 = C >> 2;
    const long total = (long)M * c4;
    const int nb = t_rows > 0 ? M / t_rows : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / c4;
        const int cq = (int)(i - m * c4);
        const long mr = t_rows > 0 ? (m % t_rows) * nb + m / t_rows : m;          // row (b, t) of the stream <-> row (t, b) of the mask
        const long blk = mr * c4 + cq;                                             // (mr * C + 4 cq) / 4
        unsigned c[4] = {(unsigned)blk, (unsigned)(blk >> 32), mask_id, st};
        philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float k = ((float)(c[e] >> 8) * (1.0f / 16777216.0f) >= p) ? keep_value : 0.f;
            float v = a[m * lda + 4 * cq + e] * k;
            if (b) v += b[m * ldb + 4 * cq + e];
            out[m * ldo + 4 * cq + e] = v;
        }
    }
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, long n, float p, float keep_value,
                                                           unsigned seed_lo, unsigned seed_hi, unsigned mask_id, const int* __restrict__ step_dev, int step) {
    const unsigned st = step_dev ? (unsigned)*step_dev : (unsigned)step;
    const long nblk = (n + 3) >> 2;
    for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (long)gridDim.x * blockDim.x) {
        unsigned c[4] = {(unsigned)b, (unsigned)(b >> 32), mask_id, st};
        philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long i = 4 * b + e;
            if (i < n) out[i] = ((float)(c[e] >> 8) * (1.0f / 16777216.0f) >= p) ? keep_value : 0.f;
        }
    }
}
}  // namespace

extern "C" int emage_dropout_mask(float* out, long n, float p, unsigned long long seed, unsigned mask_id, const int* step_dev, int step, void* stream) {
    if (!out || n <= 0 || !(p >= 0.f && p < 1.f)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, p, 1.0f / (1.0f - p),
                       (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), mask_id, step_dev, step);
    return launch_status();
}

extern "C" int emage_mul_add_philox(const float* a, int lda, float p, unsigned long long seed, unsigned mask_id, const int* step_dev, int step, int mask_t_rows,
                                    const float* b, int ldb, float* out, int ldo, int M, int C, void* stream) {
    if (!a || !out || M <= 0 || C <= 0 || C % 4 || lda < C || ldo < C || (b && ldb < C) || !(p >= 0.f && p < 1.f) || (mask_t_rows > 0 && M % mask_t_rows)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(mul_add_philox_kernel, dim3(grid_for((long)M * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, p, 1.0f / (1.0f - p),
                       (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), mask_id, step_dev, step, mask_t_rows, b, ldb, out, ldo, M, C);
    return launch_status();
}

// ---- multi-tensor Adam -----------------------------------------------------------------------------------------------------------------
// ONE launch for every parameter of the model instead of one per tensor (445 launches): `table` holds, per tensor, the five 64-bit words
// {param, grad, exp_avg, exp_avg_sq, n}; block b works on chunk block_chunk[b] (ADAM_CHUNK elements) of tensor block_tensor[b].  The same
// arithmetic as emage_adam_step; grad_scale multiplies every gradient first (the 1 / world_size of the data-parallel average); with
// zero_grad the gradient is cleared behind the update (the next step accumulates into it again).
namespace {
constexpr int ADAM_CHUNK = 4096;
__global__ __launch_bounds__(256) void adam_multi_kernel(const long long* __restrict__ table, const int* __restrict__ block_tensor, const int* __restrict__ block_chunk,
                                                         const int* __restrict__ step_dev, int step, float lr, float b1, float b2, float eps, float weight_decay,
                                                         float grad_scale, int zero_grad, const int* __restrict__ skip) {
    const bool skipped = skip && *skip != 0;             // a non-finite gradient was counted: parameters and moments stay as they are
    const int t = step_dev ? *step_dev : step;
    const double bias1 = 1.0 - pow((double)b1, t), bias2 = 1.0 - pow((double)b2, t);
    const float step_size = (float)((double)lr / bias1), inv_sqrt_bias2 = (float)(1.0 / sqrt(bias2));
    const long long* e = table + 5 * (long)block_tensor[blockIdx.x];
    float* __restrict__ p = (float*)e[0];
    float* __restrict__ g = (float*)e[1];
    float* __restrict__ m = (float*)e[2];
    float* __restrict__ v = (float*)e[3];
    const long n = (long)e[4];
    const long i0 = (long)block_chunk[blockIdx.x] * ADAM_CHUNK;
    const long i1 = i0 + ADAM_CHUNK < n ? i0 + ADAM_CHUNK : n;
    if (skipped) {
        if (zero_grad)
            for (long i = i0 + threadIdx.x; i < i1; i += 256) g[i] = 0.f;
        return;
    }
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        float gi = g[i] * grad_scale;
        if (weight_decay != 0.f) gi += weight_decay * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bias2 + eps);
        if (zero_grad) g[i] = 0.f;
    }
}
}  // namespace

extern "C" int emage_adam_multi_chunk(void) { return ADAM_CHUNK; }

extern "C" int emage_adam_multi(const long long* table, const int* block_tensor, const int* block_chunk, int n_blocks, const int* step_dev, int step,
                                float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, const int* skip,
                                void* stream) {
    if (!table || !block_tensor || !block_chunk || n_blocks <= 0 || (!step_dev && step <= 0)) return EMAGE_EINVAL;
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return EMAGE_EINVAL;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, table, block_tensor, block_chunk, step_dev, step,
                       lr, beta1, beta2, eps, weight_decay, grad_scale, zero_grad, skip);
    return launch_status();
}

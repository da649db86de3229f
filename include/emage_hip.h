/*
 * emage_hip.h — C ABI of libemage_hip.so, the MI355X (gfx950) kernels behind the EMAGE hot path.
 *
 * The reference (PantoMatrix @ 2025-01-17) has no FFI/operator registry for this path: the
 * arithmetic lives in torch.nn calls made from
 *   models/emage_audio/modeling_emage_audio.py   (M:)
 *   models/emage_audio/processing_emage_audio.py (P:)
 * Each entry point below replaces one ATen op family at the cited call sites (SURVEY.md §2b K1-K11,
 * §8b last row).  Conventions, all entry points:
 *   - extern "C", plain device pointers + sizes, no torch types;
 *   - non-owning: never allocates, frees or synchronises; safe inside hipGraph capture;
 *   - `stream` is a hipStream_t passed as void*;
 *   - returns 0 on success, a hipError_t (> 0) from the launch, or a negative EMAGE_E* code for
 *     an argument the kernels do not support (the host turns non-zero into an exception);
 *   - activations are (B,T,C) row-major == (rows, C) with a row stride `ld*` in ELEMENTS;
 *   - `dtype` selects the storage/MFMA operand type of activations and weights:
 *       EMAGE_F32  : float32 operands, v_mfma_f32_16x16x4_f32  (exact fp32, parity mode)
 *       EMAGE_BF16 : bfloat16 operands, v_mfma_f32_16x16x32_bf16, fp32 accumulate
 *       EMAGE_F16X3: (emage_gemm only) float32 activations, weights pre-split into two fp16 planes; every
 *                    product runs as three v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate):
 *                    ~22 mantissa bits per product, i.e. fp32-grade results at 1/3 of the fp16 MFMA rate
 *                    instead of the 1/16 of the fp32 MFMA.  Every other entry point takes EMAGE_F32 tensors
 *                    in this mode (the storage type IS float32).
 *     accumulators, biases, LayerNorm statistics, softmax and every index are fp32 / int64 in
 *     all modes.
 */
#ifndef EMAGE_HIP_H
#define EMAGE_HIP_H

#include <stdint.h>

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMAGE_F32 0
#define EMAGE_BF16 1
#define EMAGE_F16X3 2
#define EMAGE_H2 3
/* EMAGE_H2 with the activation scale of a MODEL: activation images (what emage_gemm / emage_layernorm / emage_attention / emage_add / emage_cast_pad /
 * emage_pack_motion / emage_gather_rows WRITE as `out`, and what emage_gemm / emage_add read as an H2 residual / operand) hold x * 2^(4 - k)
 * instead of 16 x: |x| < 4094 * 2^k stays finite in the fp16 hi plane, at the price of k bits of the smallest values' lo plane.  k = 0..12;
 * EMAGE_H2_SHIFT(0) == EMAGE_H2 (the default: bit-identical to the plain code).  emage_gemm's operand A is described by `a_scale` as before
 * (pass 2^(4 - k) for an activation image written under shift k); weight / gradient images (emage_h2_cast: explicit scale) are not affected. */
#define EMAGE_H2_SHIFT(k) (EMAGE_H2 | ((k) << 8))

#define EMAGE_EINVAL (-1)   /* unsupported size / alignment / null pointer */

/* Library identification: ABI version (bumped on any signature change) and target arch string. */
int emage_abi_version(void);
const char* emage_target_arch(void);

/*
 * Tools build only (libemage_hip_tools.so, compiled with -DEMAGE_TOOLS; the product library libemage_hip.so has neither this entry
 * point nor any other mutable global state).  Process-global, not thread-safe:
 *   key 0: force emage_gemm's tile configuration id of the F32 / BF16 / F16X3 modes (-1 restores the heuristic);
 *   key 1: diagnostic ablation mask for tools/bench_gemm.py --ablate (1 no operand DMA, 2 no MFMA, 4 no epilogue;
 *          8 / 16: write-through / non-temporal result stores, a recorded negative experiment);
 *   key 2: tile-heuristic variant for A/B runs;   key 3: timing ablations of emage_lstm_layer;
 *   key 4: force the EMAGE_H2 tile configuration id;   key 5: EMAGE_H2 tile-heuristic variant;   key 6: attention variant;
 *   key 7: EMAGE_H2 tile configuration for grids of at most one 64 x 64 tile per CU (0 = the shipped one).
 * Returns EMAGE_EINVAL for unknown keys.  emage_h2_set_trace: device buffer (waves x 512 uint64) for the phase tracer of the
 * instrumented EMAGE_H2 configurations (tools/trace_gemm_h2.py).
 */
#ifdef EMAGE_TOOLS
int emage_set_tuning(int key, int value);
int emage_h2_set_trace(void* buf);
#endif

/*
 * K6 — VQ nearest neighbour.  Replaces Quantizer.map2index / Quantizer.forward's argmin (P:144-164)
 * and EmageVQVAEConv.decode_from_latent's inline copy (M:60-67):
 *   d[n][k] = (sum_j z[n][j]^2 + sum_j e[k][j]^2) - 2 * sum_j z[n][j] e[k][j];  idx[n] = argmin_k d[n][k]
 * fp32 throughout, first minimum wins ties and a NaN distance wins outright (torch.argmin).  z: (N,D) fp32 row stride
 * ldz; codebook: (K,D) fp32 contiguous.  D % 4 == 0, D <= 1024, K <= 4096.
 * idx: int64, addressed as a 2-D view — element n is written at idx[(n / idx_rows) * idx_ld + n % idx_rows], so a
 * (B, T) window of a longer (B, L) code buffer is filled in place (idx_rows = T, idx_ld = L); idx_rows <= 0: a
 * plain (N,) list.  The same view convention is used by emage_argmax_logsoftmax_f32 (output) and emage_gather_rows
 * (input, with an extra frame stride idx_tstride in {0, 1}: 0 broadcasts one id per clip over its idx_rows frames).
 */
int emage_vq_argmin_f32(const float* z, int ldz, const float* codebook, int64_t* idx, int idx_rows, long idx_ld,
                        int N, int K, int D, void* stream);

/*
 * K8 — code selection from classifier logits.  Replaces torch.max(F.log_softmax(x, dim=2), dim=2)[1]
 * (M:398-401, test_emage_audio.py:39-42): y = (x - max) - log(sum exp(x - max)) in fp32, first maximum.
 * logits: (N,C) fp32 row stride ld; idx: (N,) int64.  C <= 4096.
 */
int emage_argmax_logsoftmax_f32(const float* logits, int ld, int64_t* idx, int idx_rows, long idx_ld, int N, int C, void* stream);

/*
 * K7 — codebook / embedding gather.  Replaces Quantizer.get_codebook_entry (P:166-170).
 * table: (K,D) fp32; idx: int64 view (see emage_vq_argmin_f32), clamped to [0, K); out: (N, ldo) in `dtype`,
 * columns [D, n_store) zero-filled.
 */
int emage_gather_rows(const float* table, const int64_t* idx, int idx_rows, long idx_ld, int idx_tstride,
                      void* out, int ldo, int n_store, int N, int K, int D, int dtype, void* stream);

/*
 * K1/K2/K3 — the one contraction kernel: Linear and Conv1d as (implicit) GEMM with a fused epilogue.
 * Replaces nn.Linear (M:232-263 call sites, MLP P:316-326), nn.Conv1d k=3 of VQEncoderV5/V6,
 * VQDecoderV5, ResBlock (P:178-261) and nn.Conv1d k=15 (+ folded eval BatchNorm1d + LeakyReLU +
 * shortcut add) of BasicBlock (P:263-294).
 *
 *   out[m][n] = leaky( sum_{tap,c} A[row(m,tap)][c] * W[n][tap*Cp + c] + bias[n], slope[n] ) + res[m][n]      (res_first = 0)
 *   out[m][n] = leaky( sum_{tap,c} ...                                + bias[n] + res[m][n], slope[n] )      (res_first = 1,
 *               BasicBlock's "x += shortcut; act2(x)", P:291-293)
 *   m = b*Lout + l,  row(m,tap) = b*Lin + l*stride + tap - pad, taken as zeros when
 *   l*stride + tap - pad is outside [0, Lin)   (taps=1,stride=1,pad=0,Lin=Lout => plain Linear).
 *
 * A:     (rows_in, lda) `dtype`; channels [C, Cp) must hold finite values (they meet zero weights).
 * W:     (N, taps*Cp) `dtype`, Cp % 64 == 0, zero in the padded channels; K = taps*Cp.
 * bias:  (N) fp32 or NULL.   slope: (N) fp32 or NULL; leaky(v,s) = v > 0 ? v : v*s
 *        (0 = ReLU, 1 = identity, 0.2 / 0.1 / 0.01 = the reference's LeakyReLU slopes).
 * res:   (M, ldr) residual, fp32 if res_is_f32 else `dtype`; or NULL.
 * out:   (M, ldo) `dtype` or NULL; columns [N, n_store) are written as zeros (keeps padded channel
 *        tails finite for the next contraction).   out_f32: (M, ldf) fp32 or NULL.
 * out_t: transposed destination or NULL: columns n >= t_col0 go to out_t[(b*(N-t_col0) + n-t_col0)*t_ld + l]
 *        instead of `out` (b = m / t_rows, l = m % t_rows) — used to emit V^T for emage_attention.
 * EMAGE_F16X3: A / res / out / out_t are float32; W is the host-packed split image of (W * w_scale): per row and per
 *        32-k tile 128 bytes = [4 hi chunks | 4 lo chunks], chunk g = 8 fp16 for k = 4g..4g+3, 16+4g..16+4g+3 of the
 *        tile (hi = rne_f16(w*w_scale), lo = rne_f16(w*w_scale - hi)); A is multiplied by a_scale before its own
 *        split and the accumulators by 1/(a_scale*w_scale) after the K-loop — both scales are powers of two chosen so
 *        that the fp16 planes stay in the normal range (|A*a_scale| must stay below 65504).  Other dtypes ignore them.
 * EMAGE_H2:  A / out / (res_is_f32 = 0) res are pre-split images (csrc/h2.h: per 8 columns 32 bytes [8 fp16 hi | 8 fp16 lo] of 16 x),
 *        out_f32 / out_t float32; W is the same image of W * w_scale, rows K-contiguous.  res == out_f32 (fp32, ldr == ldf) accumulates:
 *        out_f32 += contraction (a weight gradient added straight into the parameter's gradient, loss.backward()'s accumulation T:174).
 *        A bare contraction (no bias / slope / out / out_t) with few tiles and K >= 2048 is cut into K-slices added with fp32 atomics.
 */
int emage_gemm(int dtype, const void* A, int lda, const void* W, const float* bias, const float* slope,
               const void* res, int ldr, int res_is_f32, int res_first,
               void* out, int ldo, int n_store, float* out_f32, int ldf,
               void* out_t, int t_col0, int t_rows, int t_ld,
               int M, int N, int Cp, int taps, int stride, int pad, int Lin, int Lout,
               float a_scale, float w_scale, void* stream);

/*
 * emage_gemm with a caller-owned WORKSPACE (round 5; the weight-gradient contractions of the training step, train_emage_audio.py:176
 * `loss.backward()`): same arguments, meaning and validation as emage_gemm, plus `workspace` — device memory, 16-byte aligned,
 * `workspace_bytes` long (NULL: exactly emage_gemm).  EMAGE_H2, bare contractions (only out_f32, optionally accumulating: res == out_f32) with
 * few tiles and K >= 2048: the contraction is cut into as many K-slices as fill the chip's block slots once; every slice stores its partial
 * tile as a plane of the workspace (plain stores) and a second launch adds the planes IN SLICE ORDER onto the destination — a fixed
 * association: the result is the same bits on every run, which the fp32-atomic form of emage_gemm is not.  The number of slices is limited
 * by the workspace (M * round_up(N, 4) * 4 bytes per slice); below two slices the call falls back to emage_gemm's behaviour.  The library
 * neither allocates nor keeps the workspace; its contents are scratch after the call.  Other dtypes / shapes ignore it.
 */
int emage_gemm_ws(int dtype, const void* A, int lda, const void* W, const float* bias, const float* slope,
                  const void* res, int ldr, int res_is_f32, int res_first,
                  void* out, int ldo, int n_store, float* out_f32, int ldf,
                  void* out_t, int t_col0, int t_rows, int t_ld,
                  int M, int N, int Cp, int taps, int stride, int pad, int Lin, int Lout,
                  float a_scale, float w_scale, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Several INDEPENDENT emage_gemm problems in as few launches as possible — the part-wise stacks of the model run the same shapes on
 * different weights side by side: the four VQ-VAE part decoders (M:126-193 -> P:237-261), the three refinement decoder layers and
 * their heads (M:248-263, 320-330), `motion2latent_{upper,hands,lower}.fc2` (M:315-317), the two `bodyhints_*` MLPs (M:232-233).
 * Each problem is one emage_gemm call (same fields, same meaning, same validation); the result is bit for bit that of issuing the
 * calls one by one.  No problem may read what another problem of the same call writes.  EMAGE_H2: problems that select the same tile
 * configuration share a launch (up to 8 per launch; a block finds its problem by a prefix sum over the problems' tile counts, then
 * runs the unchanged tile routine) — a stack of N = 256 problems that leaves most CUs with one 4-wave block each becomes one launch
 * with several resident blocks per CU.  Other dtypes: one launch per problem.
 */
typedef struct emage_gemm_problem {
    const void* A; const void* W; const float* bias; const float* slope; const void* res;
    void* out; float* out_f32; void* out_t;
    int lda, ldr, res_is_f32, res_first, ldo, n_store, ldf, t_col0, t_rows, t_ld;
    int M, N, Cp, taps, stride, pad, Lin, Lout;
    float a_scale, w_scale;
    /* LayerNorm FOLDED into the contractions around it (EMAGE_H2, taps = 1; every pointer NULL = off; emage_gemm itself never folds).
     * A post-norm transformer layer (nn.TransformerDecoderLayer, modeling_emage_audio.py:238-250) computes y = LN(s) W^T + b on the
     * pre-norm sum s of the previous sub-layer.  With W' = W gamma, c[n] = sum_k W'[n][k], b' = W beta + b (packed by the host):
     *     y[m][n] = rstd[m] ( (s W'^T)[m][n] - mu[m] c[n] ) + b'[n]
     * so the contraction runs on the RAW sum (A = the EMAGE_H2 image of s, W = W', bias = b') and no LayerNorm launch / normalised
     * tensor exists.  Row statistics travel as PARTIALS {mean, M2} (float pairs) over 32 columns each, written by the epilogue of the
     * launch that produced s (st_out) and merged (Chan) by every consumer:
     *   ln_stats (M, ln_np) pairs, ln_np * 32 == Cp, ln_np == 24 (the 768-wide residual stream; other widths are refused: EMAGE_EINVAL): statistics of A's rows;  ln_c (N): c;  ln_eps: the LayerNorm's eps;
     *   rs_stats (M, rs_np) pairs, rs_np * 32 == N: `res` (an EMAGE_H2 image, res_is_f32 = 0) is the raw sum s' of a folded LayerNorm,
     *       the residual added is (s' - mu) rstd rs_gamma[n] + rs_beta[n];
     *   st_out (M, N / 32) pairs: partial statistics of THIS launch's output rows (values as stored to out / out_f32); N % 64 == 0, no
     *       out_t, launches that take the 64 x 64 tile (N < 1024) only. */
    const float* ln_stats; const float* ln_c; const float* rs_stats; const float* rs_gamma; const float* rs_beta; float* st_out;
    int ln_np, rs_np;
    float ln_eps;
    /* Split-K with an in-kernel fix-up (EMAGE_H2; optional, NULL = off): a launch of few rows (ONE clip: M = 64 — a dozen 64 x 64 tiles that
     * would each walk the whole K range alone on one CU) may be cut into K-slices when the caller lends it scratch memory: sk_ws (16-byte
     * aligned, sk_ws_bytes; 16 KiB per tile and slice) and sk_count (sk_tiles ints, ZERO on entry; the launch leaves them zero).  Every
     * block stores its partial accumulators write-through, the last block of a tile to arrive adds the slices in slice order (the same bits on
     * every run) and runs the ordinary epilogue.  The library decides per launch (<= 128 tiles, >= 8 K-tiles of 32, capacity); launches on
     * ONE stream may share the scratch, concurrent streams need their own. */
    void* sk_ws; long sk_ws_bytes; int* sk_count; int sk_tiles;
} emage_gemm_problem;
int emage_gemm_grouped(int dtype, const emage_gemm_problem* problems, int n_problems, void* stream);
/* Launches nothing: the number of kernel launches emage_gemm_grouped would make of these problems (> 0), or a negative EMAGE_E* code. */
int emage_gemm_grouped_launches(int dtype, const emage_gemm_problem* problems, int n_problems);

/*
 * K1 first layer — WavEncoder block 0 on the raw waveform (Cin = 1), P:301 + P:283-290:
 *   out[b][l][c] = leaky( sum_k wav[b][l*stride + k - pad] * w[c][k] + bias[c], slope[c] )
 * wav: fp32, clip b at wav + b*ldw; the launch covers `nwin` windows of L samples per clip, window i starting at
 * sample i*hop of its clip (inference()'s sliding windows, M:393-394, read in place: no slicing copy); samples
 * outside [0, L) of a window are the conv's zero padding.  Output sequence (i*B + b) holds window i of clip b:
 * out: (nwin*B*Lout, ldo) `dtype`.  w: (C, taps) fp32 (eval BatchNorm folded).
 * Computes conv1 and the downsample shortcut (same input, same geometry) in one pass: the host
 * stacks their filters along C and gives the shortcut channels slope 1.  C % 8 == 0, taps <= 16.
 */
int emage_wav_conv_in(int dtype, const float* wav, long ldw, int L, int nwin, long hop,
                      const float* w, const float* bias, const float* slope,
                      void* out, int ldo, int B, int Lout, int C, int taps, int stride, int pad, void* stream);

/*
 * K1, LDS-resident input slab — the stride-1, k <= 16 convolutions with C == N in {64, 128} channels (BasicBlock conv2 of every
 * WavEncoder block and conv1 of the stride-1 blocks, P:263-306), same arithmetic as emage_gemm with taps = k, stride 1:
 *   out[s][l][n] = leaky( sum_{tap,c} A[s][l + tap - pad][c] * W[n][tap*C + c] + bias[n] + res[s][l][n], slope[n] )
 * A block owns 128 positions of one sequence: its 128 + k - 1 input rows are loaded into LDS ONCE (in EMAGE_F16X3 already
 * split into fp16 hi / lo planes) instead of once per tap, and only the W tiles stream through the LDS-DMA ring.
 * A: (nseq*L, lda) `dtype` (float32 for EMAGE_F16X3); W / bias / slope / a_scale / w_scale as for emage_gemm (bias and
 * slope are required); res: (nseq*L, ldr) `dtype` or NULL; out: (nseq*L, ldo) `dtype`.  Bit-identical to emage_gemm.
 *
 * emage_wav_block0 — WavEncoder block 0 in ONE launch (P:283-294 with Cin = 1): the slab is conv1 itself,
 *   y1 = leaky(conv(wav, w1) + b1, slope1), evaluated from the raw waveform into LDS (eval BatchNorm folded by the host,
 *   windows addressed as in emage_wav_conv_in); conv2 runs on it as above, and the epilogue adds the downsample shortcut
 *   conv(wav, wds) + bds of its own outputs before the activation.  out: (nwin*nclip*L, ldo), L = block-0 output length.
 *   Bit-identical to emage_wav_conv_in followed by emage_gemm; the (rows, 2C) first-layer tensor never reaches HBM.
 */
int emage_conv_slab(int dtype, const void* A, int lda, const void* W, const float* bias, const float* slope,
                    const void* res, int ldr, void* out, int ldo,
                    int nseq, int L, int C, int taps, int pad, float a_scale, float w_scale, void* stream);
int emage_wav_block0(int dtype, const float* wav, long ldw, int Lw, int nwin, long hop, int nclip,
                     const float* w1, const float* b1, float slope1, const float* wds, const float* bds,
                     int taps1, int stride1, int pad1,
                     const void* W2, const float* bias2, const float* slope2, int taps2, int pad2,
                     void* out, int ldo, int L, int C, float a_scale, float w_scale, void* stream);

/*
 * K4 — multi-head attention core, softmax(Q K^T / sqrt(hd)) V, no mask (nn.MultiheadAttention inside
 * the 15 decoder layers and 1 encoder layer, M:238-250,261).
 * q:  (B*Tq, ldq) `dtype`, head h at columns [h*hd, (h+1)*hd).
 * k:  (B*Tk, ldk) `dtype`, same head layout.
 * vt: V transposed, (B, vt_rows, ldvt) `dtype`: vt[(b*vt_rows + h*hd + d)*ldvt + t], vt_rows >= H*hd (several
 *     layers' V^T may share one buffer); ldvt % 32 == 0, ldvt >= Tk, columns [Tk, ldvt) must be finite (zero).
 * out:(B*Tq, ldo) `dtype`, head-concatenated like nn.MultiheadAttention before out_proj.
 * hd == 192 (768 / 4 heads, the only head size on this path), Tk <= 128.
 */
int emage_attention(int dtype, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int vt_rows,
                    void* out, int ldo, int B, int H, int Tq, int Tk, int hd, void* stream);

/* emage_attention with the attention-probability dropout of a TRAINING forward (nn.MultiheadAttention in train mode,
 * torch's scaled_dot_product_attention math path): softmax(.) * pmask, then P V.  pmask: (B, H, Tq, Tk) fp32, the
 * factors bernoulli / (1 - p).  dtype EMAGE_F32 or EMAGE_F16X3. */
int emage_attention_dropout(int dtype, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int vt_rows,
                            void* out, int ldo, int B, int H, int Tq, int Tk, int hd, const float* pmask, void* stream);

/*
 * K5 — LayerNorm(C, eps) over rows (post-norm of every transformer sub-layer):
 *   y = (x - mean) * rsqrt(var + eps) * gamma + beta (+ add[m][:])
 * x: (M, ldx) `dtype` — the residual stream is stored in the compute dtype (fp32 in parity mode, bf16 in bf16
 * mode); statistics and the affine map are fp32.  add: (M, ldadd) `dtype` or NULL (folds the positional /
 * speaker / skip adds that follow a LayerNorm, M:304-305,312).  y_f32: (M, ldy) fp32 or NULL;
 * y: (M, ldy) `dtype` or NULL.  C % 64 == 0, C <= 1024.
 */
int emage_layernorm(int dtype, const void* x, int ldx, const float* gamma, const float* beta, float eps,
                    const void* add, int ldadd, float* y_f32, void* y, int ldy, int M, int C, void* stream);

/*
 * K11 — elementwise glue.
 * emage_add: out[m] = a[m] + b[m % mod_b] (+ c[m % mod_c]); operand k (a=0,b=1,c=2) is fp32 when bit k of
 *   f32_mask is set, else `dtype` (mod_* = 0: no wrap; mod = T broadcasts a (T,C) positional table over the
 *   batch, P:341-343); writes fp32 and/or `dtype`.  C % 4 == 0.
 * emage_pack_motion: one window of the masked-motion input -> `dtype`, (B*T, ldo), columns [C, n_store) zero:
 *   where(mask == 1, mask_embedding, motion) (M:267-268); with `seed` != NULL the first `pre` frames are
 *   where(mask == 0, motion, seed) and count as unmasked — the seed splice of inference() (M:386-391).  motion / mask:
 *   fp32, frame (b, t) at b*ldb + t*C (a window of a longer (B, L, C) tensor is read in place); seed: frame (b, t) at
 *   b*ld_seed + t*C.
 * emage_cast_pad: fp32 (M,C) -> `dtype` (M, ldo) with zero tail [C, n_store).
 */
int emage_add(int dtype, const void* a, int lda, const void* b, int ldb, int mod_b, const void* c, int ldc, int mod_c,
              int f32_mask, float* out_f32, void* out, int ldo, int M, int C, void* stream);
int emage_pack_motion(int dtype, const float* motion, const float* mask, long ldb, const float* mask_embedding,
                      const float* seed, long ld_seed, int pre,
                      void* out, int ldo, int n_store, int B, int T, int C, void* stream);
int emage_cast_pad(int dtype, const float* src, int lds, void* out, int ldo, int n_store, int M, int C, void* stream);

/*
 * K9 — rotation conversions (P:16-104), elementwise over n joints.
 */
int emage_rot6d_to_axis_angle(const float* rot6d, float* aa, int n, void* stream);
int emage_axis_angle_to_rot6d(const float* aa, float* rot6d, int n, void* stream);

/*
 * K9+K11 — EmageVQModel.decode's merge (M:137-188) in one pass and without host syncs:
 * part decoder outputs -> rot6d->axis-angle per part joint, scatter into the 55 SMPL-X joints
 * (recover_from_mask_ts, P:118-132; jaw at joint 22), re-encode to rot6d, append trans/contact.
 * face: (M, ldface) fp32 [6 jaw rot6d | 100 expression] or NULL (zeros); upper: (M, ldup) 78;
 * hands: (M, ldh) 180; lower: (M, ldlow) [54 rot6d | 3 trans-vel | 4 contact] or NULL.
 * Outputs (any may be NULL): axis_angle (M,165), motion (M,337) = [330 rot6d | 7], expression (M,100).
 */
int emage_merge_parts(const float* face, int ldface, const float* upper, int ldup, const float* hands, int ldh,
                      const float* lower, int ldlow, float* axis_angle, float* motion, float* expression,
                      int M, void* stream);

/*
 * K10 — velocity2position (P:107-115) for get_global_motion (M:195-205):
 *   trans[b][0][x|z] = init[b][x|z]; trans[b][t][x|z] = vel[b][t-1][x|z]*dt + trans[b][t-1][x|z]
 *   trans[b][t][y] = vel[b][t][y]        (sequential fp32, product rounded before the add)
 * vel: (B*T, ldv) fp32, the 3 velocity channels start at column col0; init: fp32, clip b at init + b*ld_init
 * (ld_init = 0 broadcasts one start position, M:198-200).
 */
int emage_velocity_to_position(const float* vel, int ldv, int col0, const float* init, int ld_init, float dt,
                               float* trans, int B, int T, void* stream);

/*
 * ---- DisCo / CaMN: models/disco_audio/modeling_disco_audio.py (D:), models/camn_audio/
 * modeling_camn_audio.py (C:).  Their WavEncoder, MLPs and the LSTM input projections run on emage_wav_conv_in /
 * emage_gemm; the entry points below are what is specific to them.  float32 tensors throughout.
 *
 * emage_lstm_step — one time step of ONE direction of nn.LSTM (D:190-195,252; C:204-218,261-268) for the whole batch:
 *   gates = gates_x[b] + h_prev[b] W_hh^T;  c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c')
 * with the gate rows INTERLEAVED per hidden unit: column 4u + g of `gates` (g = input, forget, cell, output — torch's
 * packed order regrouped by the host), so one lane owns all four gates of a unit.  h_prev: (B, ld_hprev) the previous
 * step's hidden state (zeros for the first step); w_hh: (4H, H) in `dtype` (EMAGE_F32, or EMAGE_F16X3 = the host-split
 * image of emage_gemm with w_scale / a_scale); gates_x: (B, ld_gx) this step's rows of the input projection
 * x_t W_ih^T + b_ih + b_hh (same interleaved layout); cstate: (B, ldc) updated in place; h_out: (B, ld_hout).
 * Rows may be strided views (step t of a (B, T, .) tensor: ld = T * width).  H % 64 == 0.
 */
int emage_lstm_step(int dtype, const float* h_prev, int ld_hprev, const void* w_hh, float w_scale, float a_scale,
                    const float* gates_x, int ld_gx, float* cstate, int ldc, float* h_out, int ld_hout,
                    int B, int H, void* stream);

/* Both directions of one layer in one launch: problem 0 = forward direction at its time step, problem 1 = backward direction at its
 * own; arguments as for emage_lstm_step, per direction where they differ (gate rows / cell states / outputs share their row strides). */
int emage_lstm_step_pair(int dtype, const float* h_prev0, const float* h_prev1, int ld_hprev0, int ld_hprev1,
                         const void* w_hh0, const void* w_hh1, float w_scale0, float w_scale1, float a_scale,
                         const float* gates_x0, const float* gates_x1, int ld_gx,
                         float* cstate0, float* cstate1, int ldc, float* h_out0, float* h_out1, int ld_hout,
                         int B, int H, void* stream);

/*
 * emage_lstm_layer — the whole recurrence of one bidirectional nn.LSTM layer in ONE launch per <= n_CU / (2 * H/16) * 64 clips
 * (D:190-195,252; C:204-218,261-268): persistent blocks keep their W_hh slice (both split-fp16 planes) and their cell state in
 * registers for all T steps and exchange h_t through the layer output as write-through stores the consumers poll (the data is the flag;
 * csrc/lstmseq.hip).  A block owns 64 clips — or 32 when the whole batch then still fits one launch (B <= n_CU / (2 * H/16) * 32), so that
 * every CU has a block; the result does not depend on that choice.
 * Bit-identical to T emage_lstm_step_pair launches with zero initial state.  dtype: EMAGE_F16X3 only; H = 256 or 512.
 *   gates_x  (B, T, >= 8H) fp32: the input projection of every step incl. both biases, columns dir * 4H + 4u + g
 *            (row strides ld_gx_b per clip, ld_gx_t per step, in floats)
 *   w_hh0/1  the directions' recurrent weights packed like emage_gemm's F16X3 W operand (rows 4u + g), scales w_scale0/1
 *   hseq     (B, T, >= 2H) fp32 out: h_t of the forward direction in columns [0, H), of the backward one in [H, 2H)
 *   sync     scratch of emage_lstm_layer_sync_words(B, H) 32-bit words (zeroed by the call, on the stream).  After the
 *            stream has been synchronised, word EMAGE_LSTM_SYNC_ERROR_WORD of each EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH-word
 *            record is non-zero if a block gave up waiting for its group (~1 s: blocks not co-resident) — the output
 *            is then invalid; the Python layer raises.
 */
#define EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH 544
#define EMAGE_LSTM_SYNC_ERROR_WORD 512
int emage_lstm_layer_sync_words(int B, int H);
/* returns EMAGE_EINVAL (<= 0 words) when the CURRENT device cannot host the recurrence (fewer than 2 * H/16 co-resident blocks:
 * a compute partition / CU-masked queue): callers then take emage_lstm_step_pair, which produces the same bits.
 * emage_lstm_layer_health: counter[0] += number of launch records of `sync` whose error word is set; meant to run on the same
 * stream behind the emage_lstm_layer launches (inside a captured graph), so one device counter carries "a block was lost". */
int emage_lstm_layer_health(const unsigned* sync, int sync_words, int* counter, void* stream);
int emage_lstm_layer(int dtype, const float* gates_x, long ld_gx_b, int ld_gx_t, const void* w_hh0, const void* w_hh1,
                     float w_scale0, float w_scale1, float a_scale, float* hseq, long ld_h_b, int ld_h_t,
                     int B, int T, int H, unsigned* sync, int sync_words, void* stream);

/* DisCo's content blend (D:244-247): out = softmax(sel[:, 0:2])[0] * c1 + softmax(...)[1] * c2, rows of C channels. */
int emage_softmax2_mix(const float* sel, int ld_sel, const float* c1, int ld1, const float* c2, int ld2,
                       float* out, int ldo, int M, int C, void* stream);

/*
 * The [speaker | seed motion | is-seed | zero pad] tail of the LSTM input rows (D:200-243, C:224-259): row (b, t) gets
 * speaker_table[speaker_id[b]] (speaker_f values), then the reference's padded seed tensor at frame src_map[t]
 * (frames < seed_frames carry seed_motion[b][frame] (pose_dims values, or zeros when seed_motion is NULL) and flag 1;
 * src_map[t] < 0 or >= seed_frames: zeros), then zeros up to n_store.  src_map: T ints on the device.
 */
int emage_lstm_inputs(const float* speaker_table, const int64_t* speaker_id, int speaker_f,
                      const float* seed_motion, long ld_seed_b, int pose_dims, int seed_frames, const int* src_map,
                      float* out, int ldo, int n_store, int B, int T, void* stream);

/*
 * rot-6D -> axis-angle scattered into the SMPL-X joint order (D:256-258 + recover_from_mask_ts D:81-96):
 * joint j takes the conversion of rot6d[m][6*slot .. 6*slot+5] with slot = slot_of_joint[j], or zeros when slot < 0.
 * rot6d: (M, ld) fp32; slot_of_joint: n_joints ints on the device; axis_angle: (M, n_joints*3).
 */
int emage_rot6d_scatter(const float* rot6d, int ld, const int* slot_of_joint, float* axis_angle, int M, int n_joints, void* stream);

/* counter[0] += number of non-finite values in the contiguous fp32 array x[0 .. n): the runners' end-of-batch health check. */
int emage_count_nonfinite(const float* x, long n, int* counter, void* stream);
/* The same over `count` <= 16 tensors in ONE launch (contiguous fp32, ns[i] elements each; the pointer table is read on the host at call time
 * and travels by value): the end-of-batch health check of the clip runners. */
int emage_count_nonfinite_multi(const float* const* xs, const long* ns, int count, int* counter, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Train-mode forward (train_emage_audio.py:130-204): what differs from the inference path.
 * ------------------------------------------------------------------------------------------------------------------ */

/*
 * nn.BatchNorm1d in training mode, statistics: per-channel batch mean and BIASED variance over the M rows (all clips x all
 * positions) of a channels-last fp32 tensor x (M, ldx), float64 accumulation; running_mean / running_var (either may be
 * NULL) are updated in place as torch does: r = (1 - momentum) * r + momentum * {mean, UNBIASED variance}
 * (processing_emage_audio.py:262-294 with nn.BatchNorm1d semantics; nn.SyncBatchNorm at world size 1).
 * workspace: emage_bn_stats_workspace_bytes(M, C) bytes, 8-byte aligned.
 */
long emage_bn_stats_workspace_bytes(int M, int C);
int emage_bn_stats(const float* x, int ldx, int M, int C, void* workspace, long workspace_bytes,
                   float* mean, float* var, float* running_mean, float* running_var, float momentum, void* stream);

/*
 * The normalisation with what follows it in BasicBlock.forward (P:283-294):
 *   out = LeakyReLU((x - mean) / sqrt(var + eps) * gamma + beta + shortcut, slope)
 * shortcut: none (sc NULL) | sc raw (sc_mean NULL) | sc batch-normalised with its own statistics (the downsample branch).
 */
int emage_bn_apply(const float* x, int ldx, const float* mean, const float* var, const float* gamma, const float* beta,
                   const float* sc, int ld_sc, const float* sc_mean, const float* sc_var, const float* sc_gamma, const float* sc_beta,
                   float eps, float slope, float* out, int ldo, int M, int C, void* stream);

/*
 * out = a * mask (+ b): nn.Dropout with a given mask and the residual add that follows it in nn.Transformer*Layer.
 * mask_t_rows = T > 0: the mask is stored (T, M / T, C) — the (T, B, d) layout the reference's layers see — while the
 * rows of a / b / out run (B, T); 0: same row order.
 */
int emage_mul_add(const float* a, int lda, const float* mask, int ld_mask, int mask_t_rows, const float* b, int ldb,
                  float* out, int ldo, int M, int C, void* stream);

/*
 * The losses of train_emage_audio.py:106-130, accumulated into a float64 device scalar: loss[0] += weight * value.
 *   emage_mse_loss: value = mean((pred - target)^2) over the (M, C) fp32 views           (F.mse_loss, T:108-111)
 *   emage_nll_loss: value = mean over rows of -log_softmax(logits[m])[index[m]]           (NLLLoss(log_softmax), T:113-128)
 * workspace: EMAGE_LOSS_WORKSPACE_BYTES bytes, 8-byte aligned, zeroed once by the caller.  An index outside [0, K) contributes
 * nothing and sets the last 8-byte slot of the workspace non-zero (sticky; torch raises there; the Python layer checks it).
 */
#define EMAGE_LOSS_WORKSPACE_BYTES 8192
int emage_mse_loss(const float* pred, int ld_pred, const float* target, int ld_target, int M, int C, float weight,
                   double* loss, void* workspace, void* stream);
int emage_nll_loss(const float* logits, int ld, const int64_t* index, int M, int K, float weight, double* loss, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward building blocks of the training step (first functional versions: fp32 VALU code with float64 reductions; the
 * contractions of the Linear layers — dX = dY W, dW = dY^T X — are emage_gemm launches on transposed operands).
 * ------------------------------------------------------------------------------------------------------------------ */

/* out (N, ld_out) = in (M, ld_in) transposed, fp32. */
int emage_transpose_f32(const float* in, int ld_in, float* out, int ld_out, int M, int N, void* stream);

/* out[c] (+)= sum_m x[m][c] (* y[m][c] when y is given): bias / LayerNorm-affine gradients.  float64 block partials added in
 * block order (deterministic); workspace: emage_col_sum_chunks(M) * C * 8 bytes.  out == NULL (round 5): only the partials are
 * written (partial[chunk * C + c]); a later emage_col_sum_finalize_multi ends this reduction together with others.
 *
 * emage_col_sum_finalize_multi — the finalize step of up to 64 chunked column reductions in ONE launch (a training step ends ~600 of
 * them: bias gradients from emage_grad_prep, LayerNorm / embedding gradients from emage_col_sum).  `entries`: HOST memory, read during
 * the call (the table travels by value in the kernel argument segment: nothing to allocate, and a captured launch keeps its own copy).
 * Entry by entry the arithmetic (and the bits) of the finalize launch inside emage_col_sum / emage_grad_prep.  No two entries of one
 * call may name overlapping `out` ranges. */
typedef struct emage_finalize_entry {
    const double* partial;      /* chunks x C float64 partials: partial[chunk * C + c] */
    float* out;                 /* C floats */
    int chunks, C, accumulate;  /* accumulate != 0: out[c] += sum, else out[c] = sum */
} emage_finalize_entry;
int emage_col_sum_chunks(int M);
int emage_col_sum_finalize_multi(const emage_finalize_entry* entries, int n_entries, void* stream);
int emage_col_sum(const float* x, int ldx, const float* y, int ldy, int M, int C, float* out, int accumulate,
                  void* workspace, long workspace_bytes, void* stream);

/* out = dy * (y > 0 ? 1 : slope): LeakyReLU / ReLU backward from the saved OUTPUT y of the activation. */
int emage_act_backward(const float* dy, int ld_dy, const float* y, int ld_y, float slope, float* out, int ldo, int M, int C, void* stream);

/* The gradient operands of one nn.Linear's backward (what loss.backward() computes per layer, train_emage_audio.py:174) from the gradient dy
 * (M, C) of its output in ONE pass: dpre = dy * (y > 0 ? 1 : slope) when the layer's saved output y is given (else dpre = dy);
 *   out_h (M, n_store >= C, % 8): EMAGE_H2 image of scale * dpre, zero tail columns            (operand of dX = dpre W)
 *   out_t (C, m_store >= M, % 8): EMAGE_H2 image of scale * dpre^T, zero tail columns          (operand of dW = dpre^T X)
 *   bias_grad[c] (+)= sum_m dpre[m][c] (float64 partials per 64 rows in `workspace`, >= ceil(M / 64) * C * 8 bytes, added in row order)
 * any of the three outputs may be NULL; scale a power of two (the loss scale of the split-fp16 backward).  `workspace` without
 * bias_grad (round 5): the ceil(M / 64) x C partials are written and left for emage_col_sum_finalize_multi. */
int emage_grad_prep(const float* dy, int ld_dy, const float* y, int ld_y, float slope, int M, int C, float scale,
                    void* out_h, int ldh, int n_store, void* out_t, int ldt, int m_store,
                    float* bias_grad, int accumulate, void* workspace, long workspace_bytes, void* stream);

/* nn.LayerNorm backward for input rows x (M, C): dx, and dy_xhat = dy * (x - mean) * rstd (column sums of dy_xhat / dy are
 * the weight / bias gradients: emage_col_sum). */
int emage_layernorm_backward(const float* x, int ldx, const float* gamma, const float* dy, int ld_dy, float eps,
                             float* dx, int ld_dx, float* dy_xhat, int ld_t, int M, int C, void* stream);

/* nn.LayerNorm backward with its affine gradients in two launches: dx as emage_layernorm_backward (the same bits), and
 * dgamma[c] (+)= sum_m dy[m][c] * xhat[m][c], dbeta[c] (+)= sum_m dy[m][c] (float64 partials per `rows_per_block` rows — 4 or 16 — in
 * `workspace`, added in row order; `accumulate`: into the parameters' gradient accumulators, loss.backward()'s accumulation over the three
 * forwards of a step, T:174).  C <= 1024. */
long emage_layernorm_backward_affine_workspace_bytes(int M, int C, int rows_per_block);
int emage_layernorm_backward_affine(const float* x, int ldx, const float* gamma, const float* dy, int ld_dy, float eps, float* dx, int ld_dx,
                                    float* dgamma, float* dbeta, int accumulate, int M, int C, int rows_per_block,
                                    void* workspace, long workspace_bytes, void* stream);

/* Backward of emage_attention / emage_attention_dropout (same operand layouts; pmask may be NULL): dq (B*Tq, .), dk, dv (B*Tk, .)
 * in row layout with head h at columns [h*hd, (h+1)*hd).  The probabilities are recomputed. */
int emage_attention_backward(const float* q, int ldq, const float* k, int ldk, const float* vt, int ldvt, int vt_rows, const float* pmask,
                             const float* d_out, int ld_do, float* dq, int ld_dq, float* dk, int ld_dk, float* dv, int ld_dv,
                             int B, int H, int Tq, int Tk, int hd, void* stream);

/* Gradients of the two losses w.r.t. their predictions: grad = weight * 2 (pred - target) / (M C);
 * grad = weight * (softmax(logits) - onehot(index)) / M. */
int emage_mse_loss_grad(const float* pred, int ld_pred, const float* target, int ld_target, int M, int C, float weight,
                        float* grad, int ld_grad, void* stream);
int emage_nll_loss_grad(const float* logits, int ld, const int64_t* index, int M, int K, float weight, float* grad, int ld_grad, void* stream);

/*
 * Conv1d backward through emage_gemm: dW (N, taps*C) = dY^T (N, M) x im2col(X)^T, dX = col2im(dY W).
 *   emage_im2col_t: out[(tap*C + c)][m] = X[seq*Lin + l*stride - pad + tap][c] (0 outside the sequence), m = seq*Lout + l,
 *                   out is (taps*C, ld_out >= nseq*Lout) fp32 with the tail columns left untouched (pre-zeroed by the caller)
 *   emage_col2im:   dx[seq*Lin + r][c] = sum of dcol[seq*Lout + l][tap*C + c] over the (l, tap) that read input position r
 */
int emage_im2col_t(const float* x, int ldx, int C, int taps, int stride, int pad, int Lin, int Lout, int nseq,
                   float* out, long ld_out, void* stream);
/* emage_im2col_t with the result as an EMAGE_H2 image (taps * C, ld_out): columns [nseq * Lout, rup64(nseq * Lout)) are written as zeros. */
int emage_im2col_t_h2(const float* x, int ldx, int C, int taps, int stride, int pad, int Lin, int Lout, int nseq,
                      void* out, long ld_out, void* stream);
int emage_col2im(const float* dcol, long ld, int C, int taps, int stride, int pad, int Lin, int Lout, int nseq, float* dx, int ldx, void* stream);

/* nn.BatchNorm1d (training) backward on channels-last rows: dgamma = sum dy * xhat, dbeta = sum dy (float64 sums),
 * dx = gamma * rstd * (dy - dbeta / M - xhat * dgamma / M).  workspace as for emage_bn_stats. */
int emage_bn_backward(const float* x, int ldx, const float* mean, const float* var, const float* gamma, float eps, const float* dy, int ld_dy,
                      float* dx, int ld_dx, float* dgamma, float* dbeta, int M, int C, void* workspace, long workspace_bytes, void* stream);

/* The two halves of emage_bn_backward on their own: nn.SyncBatchNorm all-reduces the per-channel sums between them
 * (train_emage_audio.py:248); `count` is the GLOBAL number of rows behind mean / var. */
int emage_bn_backward_sums(const float* x, int ldx, const float* mean, const float* var, float eps, const float* dy, int ld_dy,
                           float* sum_dy_xhat, float* sum_dy, int M, int C, void* workspace, long workspace_bytes, void* stream);
int emage_bn_backward_apply(const float* x, int ldx, const float* mean, const float* var, const float* gamma, float eps, const float* dy, int ld_dy,
                            const float* sum_dy_xhat, const float* sum_dy, long count, float* dx, int ld_dx, int M, int C, void* stream);

/* Weight gradient of emage_wav_conv_in (Cin = 1): dw[c][tap] = sum_m dy[m][c] * wav[seq][l*stride - pad + tap]. */
long emage_wav_conv_in_backward_workspace_bytes(int M, int C, int taps);
int emage_wav_conv_in_backward(const float* dy, int ld_dy, const float* wav, long ldw, int L, int B, int Lout, int C, int taps, int stride, int pad,
                               float* dw, void* workspace, long workspace_bytes, void* stream);

/* torch.optim.Adam (no amsgrad) on one flat fp32 tensor of n elements, in place (train_emage_audio.py:258-265): step is the
 * 1-based step count of this parameter. */
int emage_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, int step,
                    float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* fp32 (M, C) rows -> EMAGE_H2 image of src * scale (scale a power of two: gradients are pre-scaled so that their fp16 planes stay
 * normal).  transpose = 0: out (M, n_store), zero tail [C, n_store);  transpose != 0: out (C, n_store) with out[c][m] = src[m][c] * scale
 * and a zero tail [M, n_store) — the operands of the backward contractions of a training step (dW = dY^T X contracts over the rows). */
int emage_h2_cast(const float* src, int lds, void* out, int ldo, int n_store, int M, int C, float scale, int transpose, void* stream);

/* fp32 (N, K) weight rows (K % 32 == 0, 16-byte aligned rows) -> the EMAGE_F16X3 weight operand of w * scale (scale a power of two), the image
 * emage_gemm takes as `w` in that mode: per row and 32-k K-tile 128 bytes = [16 x 4 fp16 hi | 16 x 4 fp16 lo], 16-byte chunk g of a plane holding
 * k = 4g..4g+3, 16+4g..16+4g+3.  out: (N, ldo) 4-byte elements, ldo >= K.  A training step re-packs every weight behind each Adam update
 * (train_emage_audio.py:174-176 changes them): one launch per operand instead of the host-side tensor arithmetic. */
int emage_f16x3_pack_weights(const float* w, int ldw, void* out, int ldo, int N, int K, float scale, void* stream);

/* Multi-tensor Adam: ONE launch over every parameter.  table: per tensor five 64-bit words {param, grad, exp_avg, exp_avg_sq, n} (device
 * pointers / element count); block b updates elements [block_chunk[b] * C, +C) of tensor block_tensor[b], C = emage_adam_multi_chunk().
 * step_dev (one int32 on the device) overrides `step` when non-NULL (captured graphs).  grad_scale multiplies every gradient first (the
 * 1 / world_size of the data-parallel average); zero_grad != 0 clears the gradient behind the update.  Arithmetic of emage_adam_step.
 * skip (one int32 on the device, may be NULL): when non-zero at launch time — the trainer's count of non-finite gradient words
 * (emage_count_nonfinite over the gradient buckets, inside the same captured step) — parameters and moments are left untouched (the
 * gradients are still cleared): an overflow of the split-fp16 backward never reaches the weights (the step-skip of loss-scaled training). */
int emage_adam_multi_chunk(void);
int emage_adam_multi(const long long* table, const int* block_tensor, const int* block_chunk, int n_blocks, const int* step_dev, int step,
                     float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, const int* skip,
                     void* stream);

/* nn.Dropout's keep mask drawn on the device (T:241-250 dropout = 0.1 in every transformer layer, P:331,343): out[i] = bernoulli(1 - p) / (1 - p)
 * from Philox4x32-10 with key = seed, counter = (i / 4, mask_id, step): a pure function of its arguments (no generator state; step_dev, one
 * int32 on the device, overrides `step` when non-NULL so a captured training step draws fresh masks on every replay).  The stream is this
 * library's own (csrc/train.hip), not torch's. */
int emage_dropout_mask(float* out, long n, float p, unsigned long long seed, unsigned mask_id, const int* step_dev, int step, void* stream);

/* emage_mul_add with the mask of emage_dropout_mask(seed, mask_id, step) drawn inside the kernel instead of read from memory: out = a * keep (+ b),
 * keep = element (mask row, c) of that (M, C) mask (mask_t_rows as in emage_mul_add), C % 4 == 0.  The same bits as the two calls; the forward
 * and the backward of a dropout site (F.dropout inside nn.Transformer*Layer, train_emage_audio.py:156-172) draw the mask from the key each. */
int emage_mul_add_philox(const float* a, int lda, float p, unsigned long long seed, unsigned mask_id, const int* step_dev, int step, int mask_t_rows,
                         const float* b, int ldb, float* out, int ldo, int M, int C, void* stream);

/* emage_adam_step with the 1-based step count read from device memory (`step`: one int32), for a step captured in a hipGraph. */
int emage_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, const int* step,
                        float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMAGE_HIP_H */

// emage_gemm, EMAGE_H2 mode: Linear / Conv1d as implicit GEMM over PRE-SPLIT operands (h2.h) — the parity-green split-fp16
// MFMA arithmetic of EMAGE_F16X3 (hi*hi + hi*lo + lo*hi, fp32 accumulate) with the activation split done ONCE by the producer
// instead of by every consuming wave at every K-tile.  Tile routine: h2_tile.h.
#include "common.h"
#include <utility>
#include "h2_tile.h"

namespace emage_dev {
#ifdef EMAGE_TOOLS
int g_h2_force_config = -1;      // tools build (emage_set_tuning key 4): fixed tile configuration for sweeps
int g_h2_variant = 0;            // tools build: dispatch-heuristic variant for A/B runs (emage_set_tuning key 5)
int g_h2_small_cfg = 0;          // tools build: tile configuration for grids of at most one 64 x 64 tile per CU (emage_set_tuning key 7; 0 = the shipped 120)
int g_h2_pp_cfg = 0;             // tools build: antiphase configuration for the 768-wide launches (emage_set_tuning key 8; 0 = the shipped 120)
unsigned long long* g_h2_trace = nullptr;   // tools build: device buffer of (waves x 512) s_memtime stamps (emage_h2_set_trace)
#else
constexpr int g_h2_force_config = -1, g_h2_variant = 0, g_h2_small_cfg = 0, g_h2_pp_cfg = 0;
constexpr unsigned long long* g_h2_trace = nullptr;
#endif
}

namespace {

using namespace emage_dev;

template <int BM, int BN, int WM, int WN, int NS, int NLW, bool PIPE, bool PRE, int OCC, bool DILV, bool TRACE, int KPB = 1, bool LNF = false, bool SKF = false>
__global__ __launch_bounds__((WM * WN + NLW) * 64, OCC * (WM * WN + NLW) / 4) void gemm_h2_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[h2_smem_bytes<BM, BN, NS, KPB>()];
    // XCD-aware tile order (gemm.hip): each XCD walks a contiguous run of tiles; split-K: slice s = blockIdx / tiles
    const int nblk = p.tiles_m * p.tiles_n;
    const int split = p.ksplit > 1 ? (int)blockIdx.x / nblk : 0;
    int bid = (int)blockIdx.x - split * nblk;
    if (p.tile_order == 0 && !EMAGE_DBG(p, 32)) {     // (tools: emage_set_tuning key 1 bit 32, tools/prof_traffic_calib.py: dispatch order for every launch)
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    if (EMAGE_DBG(p, 64)) { tile_m = bid % p.tiles_m; tile_n = bid / p.tiles_m; }      // tools: an XCD's run walks M first (it owns a slice of N: A re-fetched per XCD, W once)
    gemm_h2_tile<BM, BN, WM, WN, NS, NLW, PIPE, PRE, DILV, TRACE, KPB, LNF, SKF>(p, tile_m * BM, tile_n * BN, smem, split);
}

// most tiles a split-K launch may have: 100.  Measured on the captured training step (A/Bs on one box each, tools library variants):
//   * 384 instead of 191 (the 768 x 1536 FFN weight gradients: 288 tiles -> two slices of 576 blocks): 105.2 -> 109.6 ms
//     (profiles/r04_train_step_ab_splitk_accumulate.txt) — the second slice's atomic epilogue costs more than the idle block slots;
//   * no split-K at all: 103.4 -> 119.0 ms (the 48- / 96-tile gradients of the heads and narrow MLPs need it);
//   * 100 instead of 191: 103.4 -> 98.4 ms (profiles/r04_train_step_ab_splitk_limit.txt) — the 144-tile 768 x 768 weight gradients (three
//     per decoder layer) run 80 us as four atomically-added slices and ~45 us as one slice per block on 144 CUs
//     (profiles/r04_gemm_h2_sweep_backward_shapes.txt: the 288-tile gradient, twice the work and never split, takes 47.5 us).
// Tools build: emage_set_tuning key 5 bit 1024 selects 384, bit 2048 no split-K, bit 4096 round 3's 191 (A/B runs)
static inline long h2_split_k_tiles() { return (g_h2_variant & 1024) ? 384 : (g_h2_variant & 2048) ? -1 : (g_h2_variant & 4096) ? 191 : 100; }
static inline bool h2_accumulates_in_place(const GemmArgs& a) {
    return a.res && a.res_is_f32 && a.out_f32 && (const void*)a.res == (const void*)a.out_f32 && a.ldr == a.ldf;
}

// second pass of a workspace split-K: out[m][n] (+)= plane_0[m][n] + plane_1[m][n] + ... in slice order — a fixed association, so the
// result is the same bits on every run (the atomic form adds the slices in arrival order).  4 columns per thread.
__global__ __launch_bounds__(256) void h2_splitk_reduce_kernel(const float* __restrict__ ws, long plane, int ldws, int nslices,
                                                               float* __restrict__ out, int ldf, int M, int N, int accumulate) {
    const int n4 = (N + 3) >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)M * n4) return;
    const int m = (int)(t / n4), n = (int)(t - (long)m * n4) << 2;
    const bool vec = n + 4 <= N && (ldf & 3) == 0 && (((uintptr_t)out & 15) == 0);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    float* dst = out + (long)m * ldf + n;
    if (accumulate) {
        if (vec) { const float4 q = *(const float4*)dst; v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else { for (int e = 0; e < 4; ++e) if (n + e < N) v[e] = dst[e]; }
    }
    const float* src = ws + (long)m * ldws + n;          // ldws % 4 == 0 and the planes are 16-byte aligned: whole float4s (columns >= N hold zeros or junk, never stored)
    for (int s = 0; s < nslices; ++s) {
        const float4 q = *(const float4*)(src + (long)s * plane);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
    }
    if (vec) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
    else { for (int e = 0; e < 4; ++e) if (n + e < N) dst[e] = v[e]; }
}

// CU count the heuristics plan for: the current device's (256 on MI355X; the fallback when the query fails)
static inline int h2_cus() { const int c = device_cus(); return c > 0 ? c : 256; }

// block slots of the chip for a tile configuration (LDS-bound residency x CUs of the current device): what a split-K launch should fill.
// The count only steers how many K-slices are cut, never correctness
template <int BM, int BN, int NS, int THREADS, int KPB = 1>
int h2_block_slots() {
    constexpr int by_lds = (160 * 1024) / h2_smem_bytes<BM, BN, NS, KPB>();
    constexpr int by_waves = 32 / (THREADS / 64);
    return h2_cus() * (by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves);
}

template <int BM, int BN, int WM, int WN, int NS, int NLW, bool PIPE, bool PRE = false, int OCC = 1, bool DILV = false, bool TRACE = false, int KPB = 1, bool LNF = false>
int launch_h2(GemmArgs& a, hipStream_t s) {
    if (!LNF && (a.ln_stats || a.rs_stats || a.st_out)) return EMAGE_EINVAL;      // only the LNF instantiations know the LayerNorm-fold fields
    if (a.out_t && a.t_col0 % BN != 0) return EMAGE_EINVAL;      // a tile is either row-major or transposed
    if (KPB > 1 && (a.K / 32) % KPB != 0) return EMAGE_EINVAL;   // a ring slot holds KPB whole K-tiles: an odd count would drop the last one
    a.tiles_m = (a.M + BM - 1) / BM;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    a.tiles_n = (ncols + BN - 1) / BN;
    a.trace = TRACE ? g_h2_trace : nullptr;
    // Tile order (round 5, profiles/r05_traffic_calibration.txt, r05_gemm_h2_tile_order_ab.txt).  XCD-aware runs keep an XCD's W panels in ITS L2 — as long
    // as W fits the eight 4 MB L2s next to the A panels.  The K / V projection of all cross-attention layers (W = 12 288 x 768 x 4 B = 37.7 MB) does not:
    // a strip-shaped run shares an A panel among its 64 concurrent blocks and a W panel among none (1 233 MB over the fabric, 3.9x the model); in
    // dispatch order consecutive tiles sit on different XCDs and every XCD works on the same few W panels at a time: 390 MB, 227 vs 242 us.  Smaller W
    // (qkv 7 MB: no difference; out_proj 2.4 MB: 4 % slower) keeps the runs.  Tools: emage_set_tuning key 5 bit 4194304 keeps the runs everywhere
    // Only the measured kind of launch takes it: a biased projection (no gradient contraction — those were never timed in dispatch order)
    a.tile_order = (!(g_h2_variant & 4194304) && a.bias && a.taps == 1 && (long)a.N * a.K * 4 >= (32L << 20)) ? 1 : 0;
    // split-K: a bare contraction (only out_f32, no bias / activation / residual — the weight gradients of a training step: few output
    // tiles, a very long K) with too few tiles to fill the chip is cut into K-slices whose partial tiles are atomically added in memory.
    a.ksplit = 1;
    const long tiles = (long)a.tiles_m * a.tiles_n;
    const int nk_all = a.K / 32;
    // res == out_f32 (fp32, same pitch): "out_f32 += contraction" — a weight gradient accumulated straight into the parameter's gradient.
    // One slice: the epilogue's residual add does it in place.  Split-K: the atomics land on the existing contents (no clearing, no residual)
    const bool accumulate = h2_accumulates_in_place(a);
    const bool bare = a.taps == 1 && !a.out && !a.out_t && (!a.res || accumulate) && !a.bias && !a.slope && a.out_f32 && nk_all >= 64;
    float* const ws = a.ws;
    const long ws_floats = a.ws_plane;                 // emage_gemm_ws: capacity of the workspace in floats (gemm.hip)
    a.ws = nullptr; a.ws_plane = 0; a.ldws = 0;
    bool reduce = false;
    if (bare && ws && !(g_h2_variant & 16384)) {      // tools A/B: bit 16384 = ignore the workspace (the atomic form)
        // two-pass split-K (emage_gemm_ws): as many K-slices as fill the chip's block slots ONCE, each slice >= 16 K-tiles, partial tiles as
        // planes of the workspace.  Cheap enough (plain stores + one streaming pass) to use up to twice the tile count of the atomic form
        const int SLOTS = h2_block_slots<BM, BN, NS, (WM * WN + NLW) * 64, KPB>();
        const int ldws = (a.N + 3) & ~3;
        const long plane = ((long)a.M * ldws + 3) & ~3L;
        int want = (int)(SLOTS / tiles);
        const int most = nk_all / 16;
        if (want > most) want = most;
        if (want > 16) want = 16;
        if (plane > 0 && (long)want * plane > ws_floats) want = (int)(ws_floats / plane);
        const long most_tiles = (g_h2_variant & 8192) ? 191 : SLOTS / 2;       // tools A/B: bit 8192 = two-pass below 192 tiles only
        if (want > 1 && tiles <= most_tiles) {
            int per = (nk_all + want - 1) / want;
            per = (per + 1) & ~1;
            a.nk_split = per;
            a.ksplit = (nk_all + per - 1) / per;
            if (a.ksplit > 1) {
                a.ws = ws; a.ws_plane = plane; a.ldws = ldws;
                a.res = nullptr;                       // an accumulating contraction: the reduce pass adds onto the destination
                reduce = true;
            } else {
                a.ksplit = 1;
            }
        }
    }
    if (!reduce && bare && tiles <= h2_split_k_tiles()) {     // no workspace (emage_gemm), or one too small for two slices: K-slices met by fp32 atomics
        int want = (int)((512 + tiles - 1) / tiles);
        const int most = nk_all / 16;                  // >= 16 K-tiles (512 k) per slice
        if (want > most) want = most;
        if (want > 64) want = 64;
        if (want > 1) {
            int per = (nk_all + want - 1) / want;
            per = (per + 1) & ~1;                      // the register-pipelined loops take K-tiles in pairs
            a.nk_split = per;
            a.ksplit = (nk_all + per - 1) / per;
            if (accumulate) {
                a.res = nullptr;
            } else {
                const hipError_t e = hipMemset2DAsync(a.out_f32, (size_t)a.ldf * sizeof(float), 0, (size_t)a.N * sizeof(float), (size_t)a.M, s);
                if (e != hipSuccess) return (int)e;
            }
        }
    }
    if (reduce) {
        hipLaunchKernelGGL((gemm_h2_kernel<BM, BN, WM, WN, NS, NLW, PIPE, PRE, OCC, DILV, TRACE, KPB, LNF>), dim3(a.tiles_m * a.tiles_n * a.ksplit), dim3((WM * WN + NLW) * 64), 0, s, a);
        const int rc = launch_status();
        if (rc) return rc;
        const long quads = (long)a.M * ((a.N + 3) >> 2);
        hipLaunchKernelGGL(h2_splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, (const float*)a.ws, a.ws_plane, a.ldws, a.ksplit,
                           a.out_f32, a.ldf, a.M, a.N, accumulate ? 1 : 0);
        return launch_status();
    }
    hipLaunchKernelGGL((gemm_h2_kernel<BM, BN, WM, WN, NS, NLW, PIPE, PRE, OCC, DILV, TRACE, KPB, LNF>), dim3(a.tiles_m * a.tiles_n * a.ksplit), dim3((WM * WN + NLW) * 64), 0, s, a);
    return launch_status();
}

// ---- split-K with the in-kernel fix-up (h2_tile.h SKF; round 6): few-row launches (ONE clip: M = 64 — 12 to 36 tiles of 64 x 64 that each
// walk their 24-48 K-tiles ALONE on a CU, profiles/r05_bench_kernel_stats_one_clip_serialized.csv) are cut into K-slices so that a launch
// occupies ~a block per CU; the last block of a tile to arrive sums the slices in slice order and runs the unchanged epilogue (every
// output form, the LayerNorm fold).  The 64 x 64 tile only.  Returns < 0: not applicable (the caller launches the plain form).
static int h2_skf_slices(const GemmArgs& a, int* per_out) {
    if (!a.sk_ws || !a.sk_count) return -1;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    const long tiles = (long)((a.M + 63) / 64) * ((ncols + 63) / 64);
    const int nk_all = a.K / 32;
    if (tiles > 128 || tiles > a.sk_tiles || nk_all < 8) return -1;
    int want = (int)((h2_cus() + tiles - 1) / tiles);          // ~one block per CU
    if (want > nk_all / 4) want = nk_all / 4;                  // >= 4 K-tiles per slice
    if (want > 16) want = 16;
    if (want < 2) return -1;
    const int per = (nk_all + want - 1) / want;
    const int ksplit = (nk_all + per - 1) / per;
    if (ksplit < 2 || tiles * ksplit * (64L * 64 * 4) > a.sk_ws_bytes) return -1;
    *per_out = per;
    return ksplit;
}

template <bool LNF>
int launch_h2_skf(GemmArgs& a, int ksplit, int per, hipStream_t s) {
    if (a.out_t && a.t_col0 % 64 != 0) return EMAGE_EINVAL;
    a.tiles_m = (a.M + 63) / 64;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    a.tiles_n = (ncols + 63) / 64;
    a.trace = nullptr;
    a.tile_order = 0;
    a.ws = nullptr; a.ws_plane = 0; a.ldws = 0;
    a.ksplit = ksplit; a.nk_split = per;
    hipLaunchKernelGGL((gemm_h2_kernel<64, 64, 2, 2, 3, 0, false, false, 2, false, false, 1, LNF, true>), dim3(a.tiles_m * a.tiles_n * ksplit), dim3(256), 0, s, a);
    return launch_status();
}

// ---- grouped launch: several independent problems of ONE tile configuration in one grid (emage_gemm_grouped) ----
// The argument block carries the problems by value (kernarg segment: no device table to allocate); a block finds its problem by
// comparing its (XCD-remapped) id with the running tile counts — scalar work — and then runs the unchanged tile routine on that
// problem's arguments: the result is bit for bit that of the single-problem launch.

template <int BM, int BN, int WM, int WN, int NS, int NLW, bool PIPE, bool PRE, int OCC, bool DILV, bool LNF = false>
__global__ __launch_bounds__((WM * WN + NLW) * 64, OCC * (WM * WN + NLW) / 4) void gemm_h2_group_kernel(GroupArgs g) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[h2_smem_bytes<BM, BN, NS>()];
    // XCD-aware order over the WHOLE grid: each XCD walks a contiguous run of the concatenated tile list, i.e. mostly one problem
    // (its W panel stays in that XCD's L2)
    int bid = (int)blockIdx.x;
    {
        const int nblk = g.total;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pi = 0;
#pragma unroll
    for (int i = 0; i < MAXG - 1; ++i) pi += (i + 1 < g.n && bid >= g.tile_end[i]) ? 1 : 0;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const int t = bid - (pi ? g.tile_end[pi - 1] : 0);
    const GemmArgs& p = g.p[pi];
    const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    gemm_h2_tile<BM, BN, WM, WN, NS, NLW, PIPE, PRE, DILV, false, 1, LNF>(p, tile_m * BM, tile_n * BN, smem, 0);
}

template <int BM, int BN>
void h2_tiles(GemmArgs& a) {
    a.tiles_m = (a.M + BM - 1) / BM;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    a.tiles_n = (ncols + BN - 1) / BN;
}

// the split-K condition of launch_h2 (bare contractions with few tiles and a long K: the weight gradients of a training step)
template <int BM, int BN>
bool h2_wants_split_k(const GemmArgs& a) {
    const long tiles = (long)((a.M + BM - 1) / BM) * (((a.n_store > a.N ? a.n_store : a.N) + BN - 1) / BN);
    return a.taps == 1 && !a.out && !a.out_t && (!a.res || h2_accumulates_in_place(a)) && !a.bias && !a.slope && a.out_f32 && tiles <= h2_split_k_tiles() && a.K / 32 >= 64;
}

template <int BM, int BN, int WM, int WN, int NS, int NLW, bool PIPE, bool PRE = false, int OCC = 1, bool DILV = false, bool LNF = false>
int launch_h2_group(GemmArgs** a, int n, hipStream_t s) {
    GroupArgs g;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs& q = *a[i];
        if (q.out_t && q.t_col0 % BN != 0) return EMAGE_EINVAL;
        h2_tiles<BM, BN>(q);
        q.trace = nullptr;
        q.ksplit = 1;
        total += q.tiles_m * q.tiles_n;
        g.p[i] = q;
        g.tile_end[i] = total;
    }
    for (int i = n; i < MAXG; ++i) { g.p[i] = *a[0]; g.tile_end[i] = total; }
    g.n = n;
    g.total = total;
    hipLaunchKernelGGL((gemm_h2_group_kernel<BM, BN, WM, WN, NS, NLW, PIPE, PRE, OCC, DILV, LNF>), dim3(total), dim3((WM * WN + NLW) * 64), 0, s, g);
    return launch_status();
}

// Tile configurations (id: block tile, compute-wave grid [wave tile], ring depth, loader waves, register-pipelined K-loop)
int run_config(int cfg, GemmArgs& a, hipStream_t s) {
    switch (cfg) {
        //                       BM   BN  WM WN NS NLW PIPE  PRE  OCC
        case 100: return launch_h2<64, 192, 4, 2, 2, 0, false>(a, s);              // 8 waves 16x96 (the F16X3 shape)
        case 170: return launch_h2<128, 192, 2, 4, 2, 0, false>(a, s);             // 8 waves 64x48: grids of many tiles per CU (round 4)
#ifdef EMAGE_TOOLS
        case 101: return launch_h2<64, 192, 4, 2, 3, 0, false>(a, s);
        case 102: return launch_h2<64, 192, 2, 4, 3, 0, false>(a, s);              // 8 waves 32x48
        case 103: return launch_h2<64, 192, 2, 2, 3, 0, true>(a, s);               // 4 waves 32x96, pipelined
        case 104: return launch_h2<64, 192, 2, 2, 4, 0, true>(a, s);
        case 105: return launch_h2<64, 192, 2, 2, 3, 4, true>(a, s);               // 4 compute + 4 loader waves
        case 106: return launch_h2<64, 192, 2, 2, 3, 4, false>(a, s);
        case 107: return launch_h2<64, 192, 2, 4, 3, 4, false>(a, s);              // 8 compute (32x48) + 4 loaders
        case 108: return launch_h2<128, 96, 4, 1, 3, 0, true>(a, s);               // 4 waves 32x96
        case 109: return launch_h2<128, 96, 4, 1, 3, 4, true>(a, s);
        case 110: return launch_h2<128, 96, 4, 2, 3, 4, false>(a, s);              // 8 compute 32x48 + 4 loaders
        case 111: return launch_h2<128, 128, 2, 2, 3, 0, true>(a, s);              // 4 waves 64x64
        case 112: return launch_h2<128, 128, 4, 2, 2, 0, false>(a, s);             // 8 waves 32x64
#endif
        case 113: return launch_h2<128, 128, 4, 2, 3, 0, false>(a, s);
#ifdef EMAGE_TOOLS
        case 115: return launch_h2<128, 192, 2, 2, 3, 0, true>(a, s);              // 4 waves 64x96
        case 116: return launch_h2<128, 192, 4, 2, 3, 0, false>(a, s);             // 8 waves 32x96
        case 118: return launch_h2<128, 256, 2, 2, 3, 0, true>(a, s);              // 4 waves 64x128
#endif
        case 119: return launch_h2<128, 256, 4, 2, 2, 0, false>(a, s);             // 8 waves 32x128
        case 120: return launch_h2<64, 64, 2, 2, 3, 0, false, false, 2>(a, s);     // 768-wide / small outputs: 4 waves 32x32, three blocks per CU
        // the same two tiles WITH the LayerNorm-fold paths (GemmArgs::ln_stats / rs_stats / st_out; round 6): taken only by launches that fold
        case 1100: return launch_h2<64, 192, 4, 2, 2, 0, false, false, 1, false, false, 1, true>(a, s);
        case 1120: return launch_h2<64, 64, 2, 2, 3, 0, false, false, 2, false, false, 1, true>(a, s);
#ifdef EMAGE_TOOLS
        case 188: return launch_h2<64, 64, 2, 2, 8, 0, false, false, 1>(a, s);     // round 5 (negative): a LONE block per CU with a ring of 8 for grids of at most one tile per CU
        case 189: return launch_h2<64, 64, 2, 4, 3, 0, false, false, 1>(a, s);     // round 5: the same tile on 8 waves of 32 x 16 (2 DMA instructions per wave and K-tile instead of 4)
#endif
#ifdef EMAGE_TOOLS
        case 121: return launch_h2<64, 64, 2, 2, 3, 0, true, false, 2>(a, s);
        case 122: return launch_h2<64, 128, 2, 2, 3, 0, true>(a, s);               // 4 waves 32x64
        case 123: return launch_h2<128, 64, 4, 1, 3, 0, true>(a, s);               // 4 waves 32x64
        case 124: return launch_h2<64, 96, 2, 2, 3, 0, true, false, 2>(a, s);      // 4 waves 32x48
        case 125: return launch_h2<64, 192, 2, 2, 3, 0, true, true>(a, s);         // 103 + residual prefetch
        case 126: return launch_h2<64, 192, 2, 2, 3, 4, false, true>(a, s);        // 106 + residual prefetch
        case 127: return launch_h2<128, 96, 4, 1, 3, 0, true, true>(a, s);         // 108 + residual prefetch
        case 128: return launch_h2<64, 192, 4, 2, 3, 0, false, true>(a, s);        // 101 + residual prefetch
        case 129: return launch_h2<128, 96, 4, 1, 4, 0, true, true>(a, s);
        case 130: return launch_h2<64, 64, 2, 2, 2, 0, false, false, 3>(a, s);
        case 131: return launch_h2<128, 128, 4, 2, 3, 4, false>(a, s);             // 8 compute 32x64 + 4 loaders
        case 132: return launch_h2<128, 192, 4, 2, 3, 4, false>(a, s);             // 8 compute 32x96 + 4 loaders
        // DMA issue interleaved with the MFMAs (DILV)
        case 140: return launch_h2<64, 192, 4, 2, 2, 0, false, false, 1, true>(a, s);
        case 141: return launch_h2<64, 192, 4, 2, 3, 0, false, false, 1, true>(a, s);
        case 142: return launch_h2<64, 192, 2, 4, 3, 0, false, false, 1, true>(a, s);
        case 143: return launch_h2<64, 192, 2, 2, 3, 0, true, false, 1, true>(a, s);
        case 144: return launch_h2<64, 192, 4, 2, 3, 0, true, false, 1, true>(a, s);      // 8 waves, pipelined fragments + interleaved DMA
        case 145: return launch_h2<128, 128, 4, 2, 3, 0, false, false, 1, true>(a, s);
        case 146: return launch_h2<128, 128, 4, 2, 3, 0, true, false, 1, true>(a, s);
        case 147: return launch_h2<128, 192, 4, 2, 3, 0, false, false, 1, true>(a, s);
        case 148: return launch_h2<128, 192, 4, 2, 3, 0, true, false, 1, true>(a, s);
        case 149: return launch_h2<128, 256, 4, 2, 2, 0, false, false, 1, true>(a, s);
        case 150: return launch_h2<64, 64, 2, 2, 2, 0, false, false, 3, true>(a, s);
        case 151: return launch_h2<64, 64, 2, 2, 3, 0, false, false, 2, true>(a, s);
        case 152: return launch_h2<128, 64, 4, 2, 3, 0, false, false, 1, true>(a, s);     // 8 waves 32x32
        case 153: return launch_h2<64, 128, 2, 4, 3, 0, false, false, 1, true>(a, s);     // 8 waves 32x32
        case 155: return launch_h2<64, 192, 4, 2, 4, 0, false, false, 1, true>(a, s);
        case 156: return launch_h2<128, 128, 2, 4, 3, 0, false, false, 1, true>(a, s);    // 8 waves 64x32
        // round 4: FEWER waves with larger wave tiles (LDS fragment bytes per MFMA fall with the wave tile: a 32x32 wave tile reads 8 KB per
        // 12 MFMAs, a 64x48 one 14 KB per 36), several blocks per CU
        case 160: return launch_h2<64, 96, 1, 2, 3, 0, false, false, 2>(a, s);            // 2 waves 64x48, two blocks per CU
        case 161: return launch_h2<64, 96, 1, 2, 3, 0, true, false, 2>(a, s);
        case 162: return launch_h2<64, 64, 1, 2, 3, 0, false, false, 3>(a, s);            // 2 waves 64x32, three blocks per CU
        case 163: return launch_h2<64, 64, 2, 1, 3, 0, false, false, 3>(a, s);            // 2 waves 32x64
        case 164: return launch_h2<64, 64, 1, 1, 3, 0, true, false, 3>(a, s);             // 1 wave 64x64, pipelined
        case 165: return launch_h2<64, 64, 1, 1, 3, 0, false, false, 3>(a, s);
        case 166: return launch_h2<128, 96, 2, 2, 3, 0, false>(a, s);                     // 4 waves 64x48, one block per CU
        case 167: return launch_h2<128, 96, 2, 2, 3, 0, true>(a, s);
        case 168: return launch_h2<64, 192, 1, 4, 3, 0, false>(a, s);                     // 4 waves 64x48 side by side
        case 169: return launch_h2<64, 192, 1, 4, 3, 0, true>(a, s);
        case 171: return launch_h2<128, 256, 2, 4, 2, 0, false>(a, s);                    // 8 waves 64x64
        // round 5: one fat block per CU for grids of many tiles per CU (the K / V projections of all cross-attention layers): MFMA issue per K-tile
        // 3-4x the operand delivery (64 B/clk per CU)
        case 172: return launch_h2<256, 256, 2, 4, 2, 0, false>(a, s);                    // 8 waves 128x64
        case 173: return launch_h2<256, 192, 2, 4, 2, 0, false>(a, s);                    // 8 waves 128x48
        case 174: return launch_h2<256, 192, 4, 2, 2, 0, false>(a, s);                    // 8 waves 64x96
        case 175: return launch_h2<256, 128, 4, 2, 2, 0, false>(a, s);                    // 8 waves 64x64
        case 187: return launch_h2<64, 64, 2, 2, 3, 0, false, true, 2>(a, s);             // 120 with the residual fetched ahead of the K-loop
        // round 5 (VERDICT next #1b, "BK = 64"): TWO K-tiles per ring slot and barrier (KPB = 2), otherwise the shipped configurations
        case 180: return launch_h2<64, 192, 4, 2, 2, 0, false, false, 1, false, false, 2>(a, s);    // 100 with 128 KB of LDS: one block per CU
        case 181: return launch_h2<128, 192, 2, 4, 2, 0, false, false, 1, false, false, 2>(a, s);   // 170 with 160 KB: one block per CU instead of two
        case 183: return launch_h2<64, 64, 2, 2, 2, 0, false, false, 2, false, false, 2>(a, s);     // 120 with a ring of 2 x 32 KB: two blocks per CU
        case 184: return launch_h2<128, 128, 4, 2, 2, 0, false, false, 1, false, false, 2>(a, s);   // 113 with a ring of 2 x 64 KB
        case 185: return launch_h2<64, 128, 2, 4, 2, 0, false, false, 1, false, false, 2>(a, s);    // 8 waves 32x32, 96 KB
        case 186: return launch_h2<64, 64, 2, 2, 3, 0, false, false, 1, false, false, 2>(a, s);     // 120's ring of 3, 96 KB: one block per CU
        // instrumented twins (TRACE) of 101 / 103 / 105 / 141
        case 201: return launch_h2<64, 192, 4, 2, 3, 0, false, false, 1, false, true>(a, s);
        case 203: return launch_h2<64, 192, 2, 2, 3, 0, true, false, 1, false, true>(a, s);
        case 205: return launch_h2<64, 192, 2, 2, 3, 4, true, false, 1, false, true>(a, s);
        case 241: return launch_h2<64, 192, 4, 2, 3, 0, false, false, 1, true, true>(a, s);
#endif
        default: break;
    }
    return EMAGE_EINVAL;
}

}  // namespace

#ifdef EMAGE_TOOLS
extern "C" int emage_h2_set_trace(void* buf) { emage_dev::g_h2_trace = (unsigned long long*)buf; return 0; }
#endif

namespace emage_dev {

int run_pp_config(int cfg, GemmArgs& a, hipStream_t s);       // gemm_h2_pp.hip
int run_pp_group(int cfg, GemmArgs** a, int n, hipStream_t s);

// the tile configuration of one problem.  Measured on MI355X (tools/bench_gemm_h2.py, profiles/r03_h2_sweep*.txt).  Wide outputs: 8-wave
// 64x192 / 128x256 tiles; everything else: 64x64 tiles, three resident blocks per CU (their barriers de-synchronise, which hides each
// block's DMA / LDS phases behind the others' MFMAs better than one fat block per CU does at M = 4096).  < 0: unsupported argument
static int h2_config_for(const GemmArgs& a) {
    if (g_h2_force_config >= 0) return g_h2_force_config;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    const int v = g_h2_variant;
    if ((v & 1) && ncols < 1024 && ncols % 192 == 0 && a.M >= 2048 && !a.out_t) return 101;       // one 8-wave block per CU
    if ((v & 4) && ncols < 1024 && ncols % 192 == 0 && a.M >= 2048 && !a.out_t) return 124;       // 64x96, two 4-wave blocks per CU
    if ((v & 16) && ncols < 1024 && ncols % 96 == 0 && a.M >= 2048 && !a.out_t) return 160;      // 64x96, two 2-wave blocks per CU
    if ((v & 32) && ncols < 1024 && ncols % 96 == 0 && a.M >= 2048 && !a.out_t) return 161;
    if ((v & 64) && ncols < 1024 && !a.out_t) return 162;
    if ((v & 128) && ncols < 1024 && !a.out_t) return 163;
    if ((v & 256) && ncols < 1024 && !a.out_t) return 164;
    if (ncols >= 1024 && a.M >= 1024) {
        const long t128x256 = (long)((a.M + 127) / 128) * ((ncols + 255) / 256);
        // many tiles per CU (the K/V projections of all cross-attention layers, N = 12288): 64x48 wave tiles read 14 KB of fragments per 36
        // MFMAs (the 32x128 ones of 119: 20 KB per 48) and two 80 KB blocks fit a CU: 239 vs 270 us (profiles/r04_gemm_h2_sweep_fewer_waves_wide.txt)
        const long t128x192 = (long)((a.M + 127) / 128) * ((ncols + 191) / 192);
        const long cus = h2_cus();                     // the thresholds were measured on 256 CUs: 4 / 2 tiles per CU
        if (!(v & 512) && t128x192 >= 4 * cus && ncols % 192 == 0 && (!a.out_t || a.t_col0 % 192 == 0)) return 170;
        if (!(v & 2) && t128x256 >= 2 * cus && ncols % 256 == 0 && (!a.out_t || a.t_col0 % 256 == 0)) return 119;
        if (ncols % 192 == 0 && (!a.out_t || a.t_col0 % 192 == 0)) return 100;
        if (!a.out_t || a.t_col0 % 128 == 0) return 113;
    }
    // 768-wide outputs of a LONG contraction over many rows (the K / V projections of all cross-attention layers in the backward of a training
    // step: dW 12288 x 768 and 6144 x 768 with K = 3584, dX 3584 x 768 with K = 12288 / 6144): many 64 x 64 tiles per block slot, each re-streaming
    // its operands — the 8-wave tiles move half the bytes per MFMA (profiles/r05_gemm_h2_sweep_backward_two_pass.txt: 187-197 vs 234-266 us,
    // 104-116 vs 133-134 us).  Below ~20 M (rows x k) the 64 x 64 tile wins (the window shapes of inference: M = 4096, K <= 1536)
    if (!(v & 65536) && !a.out_t && ncols % 192 == 0 && ncols >= 768 && a.taps == 1) {
        const long mk = (long)a.M * a.K, per_cu = 1000L * 1000 * h2_cus() / 256;     // measured on 256 CUs: 40 M / 20 M (rows x k)
        if (mk >= 40L * per_cu) return 170;
        if (mk >= 20L * per_cu) return 100;
    }
    if (a.out_t && a.t_col0 % 64 != 0) return EMAGE_EINVAL;
#ifdef EMAGE_TOOLS
    // round 6 A/B (emage_set_tuning key 8 = g_h2_pp_cfg): 768-wide launches over many rows on an ANTIPHASE configuration (gemm_h2_pp.hip)
    if (g_h2_pp_cfg > 0 && !a.out_t && ncols % 96 == 0 && ncols >= 768 && a.M >= 2048) return g_h2_pp_cfg;
#endif
    return (v & 8) ? 130 : 120;
}

// a launch that folds a LayerNorm (operand statistics, a folded residual, statistics out) takes the LNF twin of its configuration — 100 / 120 only
static int h2_fold_config(const GemmArgs& a, int cfg) {
    if (!(a.ln_stats || a.rs_stats || a.st_out) || cfg < 0) return cfg;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    if (a.st_out) return (a.out_t || ncols >= 1024) ? EMAGE_EINVAL : 1120;      // row statistics come out of the 64 x 64 tile's 32 x 32 wave tiles only
    if (cfg == 120 || cfg == 100) return cfg + 1000;
    // batches whose plain launch would take another tile (119 / 170 / 113 beyond ~7 000 rows): the fold exists in two tiles only
    if (ncols >= 1024 && ncols % 192 == 0 && (!a.out_t || a.t_col0 % 192 == 0)) return 1100;
    return (a.out_t && a.t_col0 % 64 != 0) ? EMAGE_EINVAL : 1120;
}

// called by emage_gemm (gemm.hip) for dtype EMAGE_H2 after the common argument checks
int gemm_h2_dispatch(GemmArgs& a, hipStream_t s) {
    int cfg = h2_fold_config(a, h2_config_for(a));
#ifdef EMAGE_TOOLS
    // round 5, measured NEGATIVE (profiles/r05_small_grids_ring_of_8_ab.txt): grids of at most one 64 x 64 tile per CU (ONE clip: M = 64 rows; the
    // N = 256 heads / conv3 of the 64-clip batch) on config 188 — the same tile as a LONE block per CU with a ring of 8 K-tiles in flight instead
    // of 3 — are SLOWER: one clip 4.38-4.45 vs 4.33-4.34 ms, 28 s 29.2-29.4 vs 28.6-28.7, the 64-clip step 13.25-13.31 vs 13.14-13.19 (same bits).
    // Kept behind emage_set_tuning key 5 bit 524288 for A/B only
    if (cfg == 120 && g_h2_force_config < 0 && (g_h2_variant & 524288)) {
        const long t64 = (long)((a.M + 63) / 64) * ((((a.n_store > a.N ? a.n_store : a.N)) + 63) / 64);
        const bool bare = a.taps == 1 && !a.out && !a.out_t && !a.bias && !a.slope && a.out_f32 && a.K / 32 >= 64;
        if (t64 <= 256 && !bare) cfg = 188;
    }
    if (cfg == 120 && g_h2_force_config < 0 && ((g_h2_variant & 1048576) || g_h2_small_cfg > 0)) {       // tools A/B: lone-block grids on 8 waves / on any 64 x 64 configuration
        const long t64 = (long)((a.M + 63) / 64) * ((((a.n_store > a.N ? a.n_store : a.N)) + 63) / 64);
        const bool bare = a.taps == 1 && !a.out && !a.out_t && !a.bias && !a.slope && a.out_f32 && a.K / 32 >= 64;
        if (t64 <= 256 && !bare) cfg = g_h2_small_cfg > 0 ? g_h2_small_cfg : 189;
    }
#endif
    if (cfg == 120 || cfg == 1120) {                                    // few rows + a fix-up workspace: split-K inside the launch
        int per = 0;
        const int ksplit = h2_skf_slices(a, &per);
        if (ksplit > 1) return cfg == 120 ? launch_h2_skf<false>(a, ksplit, per, s) : launch_h2_skf<true>(a, ksplit, per, s);
    }
    if (cfg >= 300 && cfg < 400) return run_pp_config(cfg, a, s);       // antiphase tiles: gemm_h2_pp.hip
    return cfg < 0 ? cfg : run_config(cfg, a, s);
}

// called by emage_gemm_grouped: problems that take the same (groupable) configuration share launches of up to MAXG problems; the rest —
// other configurations, split-K contractions — are launched one by one.  Launch order follows the first problem of each launch.
// count_only: no launch, returns the number of launches the call would make (bench.py's per-launch accounting)
int gemm_h2_dispatch_group(GemmArgs* a, int n, hipStream_t s, bool count_only) {
    int launches = 0;
    int cfg[64];
    bool done[64];
    for (int i = 0; i < n; ++i) {
        cfg[i] = h2_fold_config(a[i], h2_config_for(a[i]));
        if (cfg[i] < 0) return cfg[i];
        done[i] = false;
    }
    auto groupable = [&](int i) {
        switch (cfg[i]) {
            case 100: case 1100: return !h2_wants_split_k<64, 192>(a[i]);
            case 1120: { int per = 0; return !h2_wants_split_k<64, 64>(a[i]) && h2_skf_slices(a[i], &per) < 2; }
            case 113: return !h2_wants_split_k<128, 128>(a[i]);
            case 119: return !h2_wants_split_k<128, 256>(a[i]);
            case 120: { int per = 0; return !h2_wants_split_k<64, 64>(a[i]) && h2_skf_slices(a[i], &per) < 2; }
            case 170: return !h2_wants_split_k<128, 192>(a[i]);
#ifdef EMAGE_TOOLS
            case 300: case 301: case 302: case 303: return true;
            case 160: case 161: return !h2_wants_split_k<64, 96>(a[i]);
            case 162: case 163: case 164: return !h2_wants_split_k<64, 64>(a[i]);
#endif
            default: return false;
        }
    };
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        GemmArgs* grp[MAXG];
        int m = 0;
        if (groupable(i))
            for (int j = i; j < n && m < MAXG; ++j)
                if (!done[j] && cfg[j] == cfg[i] && groupable(j)) { grp[m++] = &a[j]; done[j] = true; }
        int rc;
        ++launches;
        if (count_only) {
            done[i] = true;
            continue;
        }
        if (m >= 2 && cfg[i] >= 300 && cfg[i] < 400) {
            rc = run_pp_group(cfg[i], grp, m, s);
        } else if (m >= 2) {
            switch (cfg[i]) {
                case 100: rc = launch_h2_group<64, 192, 4, 2, 2, 0, false>(grp, m, s); break;
                case 1100: rc = launch_h2_group<64, 192, 4, 2, 2, 0, false, false, 1, false, true>(grp, m, s); break;
                case 1120: rc = launch_h2_group<64, 64, 2, 2, 3, 0, false, false, 2, false, true>(grp, m, s); break;
                case 113: rc = launch_h2_group<128, 128, 4, 2, 3, 0, false>(grp, m, s); break;
                case 119: rc = launch_h2_group<128, 256, 4, 2, 2, 0, false>(grp, m, s); break;
                case 170: rc = launch_h2_group<128, 192, 2, 4, 2, 0, false>(grp, m, s); break;
#ifdef EMAGE_TOOLS
                case 160: rc = launch_h2_group<64, 96, 1, 2, 3, 0, false, false, 2>(grp, m, s); break;
                case 161: rc = launch_h2_group<64, 96, 1, 2, 3, 0, true, false, 2>(grp, m, s); break;
                case 162: rc = launch_h2_group<64, 64, 1, 2, 3, 0, false, false, 3>(grp, m, s); break;
                case 163: rc = launch_h2_group<64, 64, 2, 1, 3, 0, false, false, 3>(grp, m, s); break;
                case 164: rc = launch_h2_group<64, 64, 1, 1, 3, 0, true, false, 3>(grp, m, s); break;
#endif
                default: rc = launch_h2_group<64, 64, 2, 2, 3, 0, false, false, 2>(grp, m, s); break;
            }
        } else {
            done[i] = true;
            int per = 0;
            const int ksplit = (cfg[i] == 120 || cfg[i] == 1120) ? h2_skf_slices(a[i], &per) : -1;
            if (ksplit > 1) rc = cfg[i] == 120 ? launch_h2_skf<false>(a[i], ksplit, per, s) : launch_h2_skf<true>(a[i], ksplit, per, s);
            else rc = run_config(cfg[i], a[i], s);
        }
        if (rc) return rc;
    }
    return count_only ? launches : 0;
}

}  // namespace emage_dev

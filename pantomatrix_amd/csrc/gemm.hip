// emage_gemm — Linear / Conv1d as implicit GEMM on MFMA, fused bias + LeakyReLU + residual epilogue.
// One kernel for both operand types: a lane's operand unit is a 16-byte chunk (8 bf16 -> one
// v_mfma_f32_16x16x32_bf16, or 4 fp32 -> four v_mfma_f32_16x16x4_f32), so the tile geometry in
// bytes is identical: K-tile = 8 chunks = 128 B per row, LDS rows XOR-swizzled by (row>>1)&7 so a
// ds_read_b128 lane group touches 16 distinct 16-B slots.
#include "common.h"
#include <utility>
#include <cstdlib>

#include "gemm_tile.h"
#include "h2.h"

namespace emage_dev {
#ifdef EMAGE_TOOLS
extern int g_lstm_layer_dbg;                        // csrc/lstmseq.hip
extern int g_h2_force_config, g_h2_variant, g_h2_small_cfg, g_h2_pp_cfg;         // csrc/gemm_h2.hip
extern int g_attn_variant;                          // csrc/attention.hip
#endif
int gemm_h2_dispatch(GemmArgs& a, hipStream_t s);   // csrc/gemm_h2.hip: the EMAGE_H2 (pre-split operands) tile kernels
int gemm_h2_dispatch_group(GemmArgs* a, int n, hipStream_t s, bool count_only);   // several independent problems, same-configuration ones in shared launches
}

namespace {

using namespace emage_dev;

#ifdef EMAGE_TOOLS
// tools build only (libemage_hip_tools.so, emage_set_tuning): process-global, not thread-safe.  The product library has no mutable state.
int g_force_config = -1;   // -1 = heuristic, else a fixed tile configuration id
int g_debug_skip = 0;      // tools/bench_gemm.py --ablate: 1 = no operand DMA, 2 = no LDS reads / MFMA, 4 = no epilogue
int g_variant = 0;         // heuristic variant for A/B runs (key 2): 0 = shipped; 1 = two K-tiles per ring slot in f16x3 (measured slower end to
                           // end: profiles/r02_bench_two_ktiles_per_slot_ab.json); 2 = epilogue operands prefetched ahead of the K-loop for 64x192 tiles
#else
constexpr int g_force_config = -1, g_debug_skip = 0, g_variant = 0;
#endif

#ifdef EMAGE_TOOLS
// epilogue shared by both kernels: lane holds rows (lane>>4)*4 + r, column lane&15 of each 16x16 fragment
template <typename T, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[FM][FN], int mw, int nw, int fr, int fg) {
    T* __restrict__ out = (T*)p.out;
    T* __restrict__ out_t = (T*)p.out_t;
    const int t_ncols = p.N - p.t_col0;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = nw + j * 16 + fr;
        const bool n_in = n < p.N;
        const float bv = (n_in && p.bias) ? p.bias[n] : 0.f;
        const float sv = (n_in && p.slope) ? p.slope[n] : 1.f;
        const bool to_t = out_t != nullptr && n >= p.t_col0;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + i * 16 + fg * 4 + r;
                if (m >= p.M) continue;
                if (n_in) {
                    float v = acc[i][j][r] + bv;
                    float rv = 0.f;
                    if (p.res) {
                        rv = p.res_is_f32 ? ((const float*)p.res)[(long)m * p.ldr + n]
                                          : Elem<T>::from(((const T*)p.res)[(long)m * p.ldr + n]);
                    }
                    if (p.res_first) v += rv;
                    v = leaky(v, sv);
                    if (!p.res_first) v += rv;
                    if (to_t) {
                        const int b = m / p.t_rows, l = m - b * p.t_rows;
                        out_t[((long)b * t_ncols + (n - p.t_col0)) * p.t_ld + l] = Elem<T>::to(v);
                    } else {
                        if (out) out[(long)m * p.ldo + n] = Elem<T>::to(v);
                        if (p.out_f32) p.out_f32[(long)m * p.ldf + n] = v;
                    }
                } else if (out && n < p.n_store) {
                    out[(long)m * p.ldo + n] = Elem<T>::to(0.f);
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = KCH * EPC;            // K elements per tile
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int A_PER = BM * KCH / NTHREADS;   // chunks per thread per tile
    constexpr int B_PER = BN * KCH / NTHREADS;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(A_PER >= 1 && B_PER >= 1, "tile too small");

    __shared__ uint4 sA[2][BM * KCH];
    __shared__ uint4 sB[2][BN * KCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles (neighbouring tiles share A rows / W columns in that XCD's L2).
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;

    // per-thread load slots: chunk c = tid + 256*i -> (row = c>>3, g = c&7)
    const int g = tid & 7;
    long a_off[A_PER]; int a_lpos[A_PER]; bool a_ok[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / p.Lout, l = mm - b * p.Lout;
        a_lpos[i] = l * p.stride - p.pad;
        a_off[i] = ((long)b * p.Lin + a_lpos[i]) * p.lda + g * EPC;
    }
    long b_off[B_PER]; bool b_ok[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        b_ok[i] = n < p.N;
        b_off[i] = (long)(b_ok[i] ? n : 0) * p.K + g * EPC;
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[A_PER], rb[B_PER];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int tap = k0 / p.Cp;
        const int c0 = k0 - tap * p.Cp;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const bool ok = a_ok[i] && (unsigned)(a_lpos[i] + tap) < (unsigned)p.Lin;
            ra[i] = ok ? *(const uint4*)(A + a_off[i] + (long)tap * p.lda + c0) : zero4;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            rb[i] = b_ok[i] ? *(const uint4*)(W + b_off[i] + k0) : zero4;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int row = (tid >> 3) + 32 * i;
            sA[buf][row * KCH + (g ^ ((row >> 1) & 7))] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int row = (tid >> 3) + 32 * i;
            sB[buf][row * KCH + (g ^ ((row >> 1) & 7))] = rb[i];
        }
    };

    const int nk = p.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            uint4 af[FM], bf[FN];
            const int gg = kg * 4 + fg;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * WTM + i * 16 + fr;
                af[i] = sA[cur][row * KCH + (gg ^ ((row >> 1) & 7))];
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WTN + j * 16 + fr;
                bf[j] = sB[cur][row * KCH + (gg ^ ((row >> 1) & 7))];
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = Elem<T>::mma(af[i], bf[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<T, FM, FN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, fr, fg);
}
#endif   // EMAGE_TOOLS: the register-staged reference kernel
template <typename T, int BM, int BN, int WM, int WN, int NS, int KC, bool X3, int KPS, bool FPRE>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? lds_blocks<BM, BN, NS, KC, KPS>() : 1)) void gemm_pipe_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[pipe_smem_bytes<T, BM, BN, NS, KC, KPS>()];
    // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so each XCD walks a contiguous run of
    // tiles (neighbouring tiles share A rows / W columns in that XCD's L2)
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    gemm_pipe_tile<T, BM, BN, WM, WN, NS, KC, FPRE, X3, EPI_LINEAR, KPS>(p, tile_m * BM, tile_n * BN, smem);
}

#ifdef EMAGE_TOOLS
template <typename T, int BM, int BN, int WM, int WN>
int launch(GemmArgs& a, hipStream_t s) {
    a.tiles_m = (a.M + BM - 1) / BM;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    a.tiles_n = (ncols + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN>), dim3(a.tiles_m * a.tiles_n), dim3(NTHREADS), 0, s, a);
    return launch_status();
}

#endif

template <typename T, bool X3, int BM, int BN, int WM, int WN, int NS, int KC = 8, int KPS = 1, bool FPRE = false>
int launch_pipe(GemmArgs& a, hipStream_t s) {
    a.tiles_m = (a.M + BM - 1) / BM;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    a.tiles_n = (ncols + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_pipe_kernel<T, BM, BN, WM, WN, NS, KC, X3, KPS, FPRE>), dim3(a.tiles_m * a.tiles_n), dim3(WM * WN * 64), 0, s, a);
    return launch_status();
}

// Tile configurations.  0 / 3: register-staged kernels (the independent implementation tools/bench_gemm.py validates
// the ring kernels against).  The rest are LDS-DMA ring kernels; 25, 32, 33, 34, 36 are what the heuristic selects,
// 18 / 27 / 37 are kept for tools/bench_gemm.py sweeps.
template <typename T, bool X3>
int run_config(int cfg, GemmArgs& a, hipStream_t s) {
    // the product library carries the five configurations the heuristic selects; everything else is the tools build's
    switch (cfg) {
        case 25: return launch_pipe<T, X3, 64, 64, 2, 2, 2>(a, s);
        case 32: return launch_pipe<T, X3, 64, 192, 4, 2, 3, 8>(a, s);    // 8 waves (two per SIMD): a lone block per CU hides its own latencies
        case 33: return launch_pipe<T, X3, 64, 192, 4, 2, 2, 8>(a, s);
        case 34: return launch_pipe<T, X3, 128, 128, 4, 2, 2, 8>(a, s);
        case 36: return launch_pipe<T, X3, 128, 64, 4, 2, 3, 8>(a, s);
        default: break;
    }
#ifdef EMAGE_TOOLS
    if constexpr (!X3) {
        if (cfg == 0) return launch<T, 128, 128, 2, 2>(a, s);
        if (cfg == 3) return launch<T, 64, 64, 2, 2>(a, s);
    }
    switch (cfg) {
        case 18: return launch_pipe<T, X3, 128, 128, 2, 2, 2>(a, s);
        case 27: return launch_pipe<T, X3, 64, 192, 2, 2, 2, 8>(a, s);    // one (clip, head) per block at T = 64, hd = 192
        case 37: return launch_pipe<T, X3, 128, 192, 4, 2, 2, 8>(a, s);
        case 38: return launch_pipe<T, X3, 128, 128, 4, 2, 3, 8>(a, s);   // deeper rings / fatter tiles: sweeps of the per-CU operand stream
        case 39: return launch_pipe<T, X3, 128, 128, 4, 2, 4, 8>(a, s);
        case 40: return launch_pipe<T, X3, 64, 192, 4, 2, 4, 8>(a, s);
        case 41: return launch_pipe<T, X3, 128, 256, 4, 2, 2, 8>(a, s);
        case 42: return launch_pipe<T, X3, 128, 192, 4, 2, 3, 8>(a, s);
        case 43: return launch_pipe<T, X3, 64, 64, 2, 2, 4>(a, s);         // deeper rings for small grids (measured: no gain,
        case 44: return launch_pipe<T, X3, 64, 64, 2, 2, 3>(a, s);         // profiles/r02_gemm_sweep_f16x3_ring_depth_small_grids.txt)
        case 49: return launch_pipe<T, X3, 64, 192, 4, 2, 2, 8, 1, true>(a, s);   // 33 with bias / slope / residual fetched ahead of the K-loop
        default: break;
    }
    if constexpr (sizeof(T) == 4) {      // fp32-storage modes: two K-tiles per ring slot / barrier (even K-tile count guaranteed)
        switch (cfg) {
            case 45: return launch_pipe<T, X3, 64, 192, 4, 2, 2, 8, 2>(a, s);
            case 46: return launch_pipe<T, X3, 128, 128, 4, 2, 2, 8, 2>(a, s);
            case 47: return launch_pipe<T, X3, 64, 64, 2, 2, 2, 8, 2>(a, s);
            case 48: return launch_pipe<T, X3, 128, 64, 4, 2, 2, 8, 2>(a, s);
            default: break;
        }
    }
#endif
    return EMAGE_EINVAL;
}

template <typename T, bool X3>
int dispatch(GemmArgs& a, hipStream_t s) {
    if (g_force_config >= 0) return run_config<T, X3>(g_force_config, a, s);
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    const long t128 = (long)((a.M + 127) / 128) * ((ncols + 127) / 128);
    // measured on MI355X (tools/bench_gemm.py, profiles/r01_gemm_sweep_8wave.txt): with M = 4096 the operand stream, not
    // MFMA, bounds these launches.  Many small resident blocks (64x64, 5 per CU) win on narrow outputs; everywhere else
    // 8-wave blocks (two waves per SIMD hide a block's own DMA / epilogue latency) beat the 4-wave tiles of the same shape
    if constexpr (X3) {
        // split-f16 mode (fp32 operand bytes; profiles/r02_gemm_sweep_f16x3.txt): the strided WavEncoder convs with their
        // stacked shortcut (N >= 128) and the QKV projection prefer 128x128 / 64x192 tiles with a 2-deep ring; 2-deep rings
        // (<= 64 KiB of LDS, ~100 VGPRs) also let kernels of two stream lanes share a CU
        const bool two = g_variant == 1;      // two K-tiles (64 k) per ring slot and barrier
        const int c33 = g_variant == 2 ? 49 : 33;
        if (a.taps >= 15 && a.stride > 1 && a.M > 8192) return run_config<T, X3>(two ? 46 : 34, a, s);
        if (a.taps >= 15 && a.M > 8192) return run_config<T, X3>(two ? 48 : 36, a, s);
        if (ncols > 64 && t128 >= 512) return run_config<T, X3>(ncols <= 2304 && ncols % 192 == 0 ? (two ? 45 : c33) : (two ? 46 : 34), a, s);
        if (ncols % 192 == 0 && (long)((a.M + 63) / 64) * (ncols / 192) == 512) return run_config<T, X3>(two ? 45 : c33, a, s);
        if (a.taps == 1 && ncols == 768 && a.M >= 2048) return run_config<T, X3>(two ? 45 : c33, a, s);
        return run_config<T, X3>(25, a, s);
    }
    if (a.taps >= 15 && a.M > 8192) return run_config<T, X3>(36, a, s);     // WavEncoder convs on long sequences: 128x64, 8 waves
    if (ncols > 64 && t128 >= 512) return run_config<T, X3>(34, a, s);      // wide outputs (QKV, batched K/V projections): 128x128, 8 waves
    if (ncols % 192 == 0 && (long)((a.M + 63) / 64) * (ncols / 192) == 512) return run_config<T, X3>(33, a, s);   // FFN up-projection: exactly 2 blocks per CU
    if (a.taps == 1 && ncols == 768 && a.M >= 2048) return run_config<T, X3>(32, a, s);   // 768-wide projections: one (clip, head) tile, 8 waves
    return run_config<T, X3>(25, a, s);
}

}  // namespace

namespace {
// argument checks of emage_gemm / emage_gemm_grouped -> the kernels' argument block
int make_args(GemmArgs& a, int dtype, const void* A, int lda, const void* W, const float* bias, const float* slope,
              const void* res, int ldr, int res_is_f32, int res_first,
              void* out, int ldo, int n_store, float* out_f32, int ldf,
              void* out_t, int t_col0, int t_rows, int t_ld,
              int M, int N, int Cp, int taps, int stride, int pad, int Lin, int Lout, float a_scale, float w_scale) {
    const int epc = dtype == EMAGE_BF16 ? 8 : 4;
    if (!A || !W || M <= 0 || N <= 0 || taps <= 0 || Cp <= 0 || Cp % 64 != 0) return EMAGE_EINVAL;
    if (dtype != EMAGE_BF16 && dtype != EMAGE_F32 && dtype != EMAGE_F16X3 && dtype != EMAGE_H2) return EMAGE_EINVAL;
    if (dtype == EMAGE_H2) {       // 32-byte groups of 8 logical columns: every row of every h2 operand starts on a group
        if (lda % 8 || (res && ldr % (res_is_f32 ? 4 : 8)) || (res && ((uintptr_t)res & 15))) return EMAGE_EINVAL;
        if (out && (ldo % 8 || ((uintptr_t)out & 15) || ldo < ((((out_t ? t_col0 : N) > n_store ? (out_t ? t_col0 : N) : n_store) + 7) & ~7))) return EMAGE_EINVAL;
        if ((bias && ((uintptr_t)bias & 15)) || (slope && ((uintptr_t)slope & 15))) return EMAGE_EINVAL;
    }
    if (lda % epc != 0 || lda < Cp) return EMAGE_EINVAL;                 // 16-byte aligned operand rows
    if (((uintptr_t)A | (uintptr_t)W) & 15) return EMAGE_EINVAL;
    if (Lout <= 0 || Lin <= 0 || M % Lout != 0 || stride <= 0) return EMAGE_EINVAL;
    if (!out && !out_f32 && !out_t) return EMAGE_EINVAL;
    if (out_t && (t_rows <= 0 || M % t_rows != 0 || t_ld < t_rows || t_col0 < 0 || t_col0 > N)) return EMAGE_EINVAL;
    if ((dtype == EMAGE_F16X3 || dtype == EMAGE_H2) && !(a_scale > 0.f && w_scale > 0.f)) return EMAGE_EINVAL;
    {   // operands are addressed through 32-bit buffer offsets: each must span less than 2 GiB (the host splits larger batches)
        const long es = dtype == EMAGE_BF16 ? 2 : 4;
        const long a_span = ((((long)(M / Lout)) * Lin - 1) * lda + Cp + (long)pad * lda) * es;
        if (a_span >= (1L << 31) || (long)N * taps * Cp * es >= (1L << 31)) return EMAGE_EINVAL;
    }
    a.A = A; a.W = W; a.bias = bias; a.slope = slope; a.res = res; a.out = out; a.out_f32 = out_f32; a.out_t = out_t;
    a.lda = lda; a.ldr = ldr; a.ldo = ldo; a.ldf = ldf; a.res_is_f32 = res_is_f32; a.res_first = res_first; a.n_store = out ? n_store : 0;
    a.t_col0 = out_t ? t_col0 : N; a.t_rows = t_rows > 0 ? t_rows : 1; a.t_ld = t_ld;
    a.tiles_m = a.tiles_n = 0;
    a.dbg = g_debug_skip;
    a.trace = nullptr; a.cstate = nullptr; a.ldc = 0; a.ksplit = 1; a.nk_split = 0; a.ws = nullptr; a.ws_plane = 0; a.ldws = 0; a.tile_order = 0;
    a.sk_ws = nullptr; a.sk_count = nullptr; a.sk_ws_bytes = 0; a.sk_tiles = 0;
    a.ln_stats = nullptr; a.ln_np = 0; a.ln_c = nullptr; a.rs_stats = nullptr; a.rs_np = 0; a.rs_gamma = nullptr; a.rs_beta = nullptr; a.st_out = nullptr; a.ln_eps = 0.f;
    a.M = M; a.N = N; a.K = taps * Cp; a.Cp = Cp; a.taps = taps; a.stride = stride; a.pad = pad; a.Lin = Lin; a.Lout = Lout;
    const bool split = dtype == EMAGE_F16X3 || dtype == EMAGE_H2;
    a.a_scale = split ? a_scale : 1.f;
    a.o_scale = split ? 1.f / (a_scale * w_scale) : 1.f;
    a.h2s = emage_dev::H2_SCALE; a.h2i = emage_dev::H2_INV;
    return 0;
}

// the LayerNorm-fold fields of an emage_gemm_problem (include/emage_hip.h) -> GemmArgs; EMAGE_H2 only
int apply_fold(int dtype, GemmArgs& a, const emage_gemm_problem& q) {
    if (!q.ln_stats && !q.rs_stats && !q.st_out) return 0;
    if (dtype != EMAGE_H2 || a.taps != 1) return EMAGE_EINVAL;
    if (q.ln_stats) {
        if (!q.ln_c || q.ln_np != 24 || q.ln_np * 32 != a.Cp || !(q.ln_eps > 0.f)) return EMAGE_EINVAL;     // the operand's rows ARE the normalised rows
        if (((uintptr_t)q.ln_stats & 15) || ((uintptr_t)q.ln_c & 15)) return EMAGE_EINVAL;
    }
    if (q.rs_stats) {
        if (!a.res || a.res_is_f32 || !q.rs_gamma || !q.rs_beta || q.rs_np != 24 || q.rs_np * 32 != a.N || !(q.ln_eps > 0.f)) return EMAGE_EINVAL;
        if (((uintptr_t)q.rs_stats & 15) || ((uintptr_t)q.rs_gamma & 15) || ((uintptr_t)q.rs_beta & 15)) return EMAGE_EINVAL;
    }
    if (q.st_out) {
        if (a.out_t || a.N % 64 != 0 || a.n_store > a.N || ((uintptr_t)q.st_out & 7)) return EMAGE_EINVAL;                    // whole 64 x 64 tiles of 32 x 32 wave tiles
    }
    a.ln_stats = q.ln_stats; a.ln_np = q.ln_np; a.ln_c = q.ln_c; a.rs_stats = q.rs_stats; a.rs_np = q.rs_np; a.rs_gamma = q.rs_gamma; a.rs_beta = q.rs_beta;
    a.st_out = q.st_out; a.ln_eps = q.ln_eps;
    return 0;
}

int dispatch_one(int dtype, GemmArgs& a, hipStream_t s) {
    if (dtype == EMAGE_H2) return gemm_h2_dispatch(a, s);
    if (dtype == EMAGE_F16X3) return dispatch<float, true>(a, s);
    return dtype == EMAGE_BF16 ? dispatch<bf16_t, false>(a, s) : dispatch<float, false>(a, s);
}
}  // namespace

extern "C" int emage_gemm(int dtype, const void* A, int lda, const void* W, const float* bias, const float* slope,
                          const void* res, int ldr, int res_is_f32, int res_first,
                          void* out, int ldo, int n_store, float* out_f32, int ldf,
                          void* out_t, int t_col0, int t_rows, int t_ld,
                          int M, int N, int Cp, int taps, int stride, int pad, int Lin, int Lout,
                          float a_scale, float w_scale, void* stream) {
    GemmArgs a;
    emage_dev::H2Scale hs;
    if (emage_dev::h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    const int rc = make_args(a, dtype, A, lda, W, bias, slope, res, ldr, res_is_f32, res_first, out, ldo, n_store, out_f32, ldf, out_t, t_col0, t_rows, t_ld,
                             M, N, Cp, taps, stride, pad, Lin, Lout, a_scale, w_scale);
    if (rc) return rc;
    a.h2s = hs.s; a.h2i = hs.inv;
    return dispatch_one(dtype, a, (hipStream_t)stream);
}

// emage_gemm with a caller-owned workspace: a split-K contraction (EMAGE_H2: bare weight-gradient shapes, see gemm_h2.hip) stores its
// K-slices' partial tiles as planes of the workspace and sums them in slice order with a second launch — no atomics, bit-reproducible.
extern "C" int emage_gemm_ws(int dtype, const void* A, int lda, const void* W, const float* bias, const float* slope,
                             const void* res, int ldr, int res_is_f32, int res_first,
                             void* out, int ldo, int n_store, float* out_f32, int ldf,
                             void* out_t, int t_col0, int t_rows, int t_ld,
                             int M, int N, int Cp, int taps, int stride, int pad, int Lin, int Lout,
                             float a_scale, float w_scale, void* workspace, size_t workspace_bytes, void* stream) {
    GemmArgs a;
    emage_dev::H2Scale hs;
    if (emage_dev::h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    const int rc = make_args(a, dtype, A, lda, W, bias, slope, res, ldr, res_is_f32, res_first, out, ldo, n_store, out_f32, ldf, out_t, t_col0, t_rows, t_ld,
                             M, N, Cp, taps, stride, pad, Lin, Lout, a_scale, w_scale);
    if (rc) return rc;
    a.h2s = hs.s; a.h2i = hs.inv;
    if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 16)) return EMAGE_EINVAL;
    if (workspace && dtype == EMAGE_H2) {
        a.ws = (float*)workspace;
        a.ws_plane = (long)(workspace_bytes / sizeof(float));       // the dispatch turns the capacity (in floats) into the plane stride it uses
    }
    return dispatch_one(dtype, a, (hipStream_t)stream);
}

namespace {
int grouped(int dtype, const emage_gemm_problem* problems, int n_problems, hipStream_t s, bool count_only) {
    constexpr int MAX_PROBLEMS = 64;
    if (!problems || n_problems <= 0 || n_problems > MAX_PROBLEMS) return EMAGE_EINVAL;
    emage_dev::H2Scale hs;
    if (emage_dev::h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    GemmArgs args[MAX_PROBLEMS];
    for (int i = 0; i < n_problems; ++i) {         // every problem is checked before the first launch
        const emage_gemm_problem& q = problems[i];
        const int rc = make_args(args[i], dtype, q.A, q.lda, q.W, q.bias, q.slope, q.res, q.ldr, q.res_is_f32, q.res_first, q.out, q.ldo, q.n_store,
                                 q.out_f32, q.ldf, q.out_t, q.t_col0, q.t_rows, q.t_ld, q.M, q.N, q.Cp, q.taps, q.stride, q.pad, q.Lin, q.Lout,
                                 q.a_scale, q.w_scale);
        if (rc) return rc;
        args[i].h2s = hs.s; args[i].h2i = hs.inv;
        const int rf = apply_fold(dtype, args[i], q);
        if (rf) return rf;
        if (q.sk_ws && q.sk_count && dtype == EMAGE_H2) {      // split-K fix-up workspace (optional: the dispatch decides)
            if (((uintptr_t)q.sk_ws & 15) || q.sk_ws_bytes <= 0 || q.sk_tiles <= 0) return EMAGE_EINVAL;
            args[i].sk_ws = (float*)q.sk_ws; args[i].sk_count = q.sk_count; args[i].sk_ws_bytes = q.sk_ws_bytes; args[i].sk_tiles = q.sk_tiles;
        }
    }
    if (dtype == EMAGE_H2) return gemm_h2_dispatch_group(args, n_problems, s, count_only);
    if (count_only) return n_problems;
    for (int i = 0; i < n_problems; ++i) {
        const int rc = dispatch_one(dtype, args[i], s);
        if (rc) return rc;
    }
    return 0;
}
}  // namespace

extern "C" int emage_gemm_grouped(int dtype, const emage_gemm_problem* problems, int n_problems, void* stream) {
    return grouped(dtype, problems, n_problems, (hipStream_t)stream, false);
}

extern "C" int emage_gemm_grouped_launches(int dtype, const emage_gemm_problem* problems, int n_problems) {
    return grouped(dtype, problems, n_problems, nullptr, true);
}

#ifdef EMAGE_TOOLS
extern "C" int emage_set_tuning(int key, int value) {
    if (key == 0) { g_force_config = value; return 0; }
    if (key == 1) { g_debug_skip = value; return 0; }
    if (key == 2) { g_variant = value; return 0; }
    if (key == 3) { emage_dev::g_lstm_layer_dbg = value; return 0; }     // csrc/lstmseq.hip: A/B and timing ablations
    if (key == 4) { emage_dev::g_h2_force_config = value; return 0; }    // csrc/gemm_h2.hip: fixed EMAGE_H2 tile configuration
    if (key == 5) { emage_dev::g_h2_variant = value; return 0; }         // csrc/gemm_h2.hip: dispatch-heuristic variant (A/B runs)
    if (key == 6) { emage_dev::g_attn_variant = value; return 0; }       // csrc/attention.hip: 1 = split-f16 attention without the LDS-staged K / V^T (A/B, bitwise test)
    if (key == 7) { emage_dev::g_h2_small_cfg = value; return 0; }       // csrc/gemm_h2.hip: EMAGE_H2 configuration for lone-block grids (A/B)
    if (key == 8) { emage_dev::g_h2_pp_cfg = value; return 0; }          // csrc/gemm_h2.hip: antiphase configuration (gemm_h2_pp.hip) for the 768-wide launches (A/B)
    return EMAGE_EINVAL;
}
#endif

// emage_gemm, EMAGE_H2 mode, ANTIPHASE tile configurations (h2_pp_tile.h; round 6): one 8-wave block per CU whose two wave groups run
// the K-loop half a step apart on ONE shared A / W panel.  Dispatched by gemm_h2.hip (configuration ids 300-399).
#include "common.h"
#include <utility>
#include "h2_pp_tile.h"

namespace emage_dev {
#ifdef EMAGE_TOOLS
extern unsigned long long* g_h2_trace;
#else
static constexpr unsigned long long* g_h2_trace = nullptr;
#endif
}

namespace {

using namespace emage_dev;

template <int BM, int BN, int WM, int WN, int NS, bool PRIO, bool TRACE, int MODE>
__global__ __launch_bounds__(512, 2) void gemm_h2_pp_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[h2_pp_smem_bytes<BM, BN, NS>()];
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = (int)blockIdx.x;
    if (p.tile_order == 0 && !EMAGE_DBG(p, 32)) {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    gemm_h2_pp_tile<BM, BN, WM, WN, NS, PRIO, TRACE, MODE>(p, tile_m * BM, tile_n * BN, smem);
}

template <int BM, int BN, int WM, int WN, int NS, bool PRIO = true, bool TRACE = false, int MODE = 0>
int launch_h2_pp(GemmArgs& a, hipStream_t s) {
    if (a.out_t && a.t_col0 % BN != 0) return EMAGE_EINVAL;      // a tile is either row-major or transposed
    if (a.ln_stats || a.rs_stats || a.st_out) return EMAGE_EINVAL; // no LayerNorm fold in these tiles
    a.tiles_m = (a.M + BM - 1) / BM;
    const int ncols = a.n_store > a.N ? a.n_store : a.N;
    a.tiles_n = (ncols + BN - 1) / BN;
    a.trace = TRACE ? g_h2_trace : nullptr;
    a.tile_order = 0;
    a.ksplit = 1;
    a.ws = nullptr; a.ws_plane = 0; a.ldws = 0;
    hipLaunchKernelGGL((gemm_h2_pp_kernel<BM, BN, WM, WN, NS, PRIO, TRACE, MODE>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
    return launch_status();
}

// grouped launch (gemm_h2.hip `launch_h2_group`): the problems' tiles concatenated, XCD-aware order over the whole grid
template <int BM, int BN, int WM, int WN, int NS, bool PRIO>
__global__ __launch_bounds__(512, 2) void gemm_h2_pp_group_kernel(GroupArgs g) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[h2_pp_smem_bytes<BM, BN, NS>()];
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
    gemm_h2_pp_tile<BM, BN, WM, WN, NS, PRIO, false>(p, tile_m * BM, tile_n * BN, smem);
}

template <int BM, int BN, int WM, int WN, int NS, bool PRIO = true>
int launch_h2_pp_group(GemmArgs** a, int n, hipStream_t s) {
    GroupArgs g;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs& q = *a[i];
        if (q.out_t && q.t_col0 % BN != 0) return EMAGE_EINVAL;
        q.tiles_m = (q.M + BM - 1) / BM;
        const int ncols = q.n_store > q.N ? q.n_store : q.N;
        q.tiles_n = (ncols + BN - 1) / BN;
        q.trace = nullptr;
        q.ksplit = 1;
        q.ws = nullptr; q.ws_plane = 0; q.ldws = 0;
        total += q.tiles_m * q.tiles_n;
        g.p[i] = q;
        g.tile_end[i] = total;
    }
    for (int i = n; i < MAXG; ++i) { g.p[i] = *a[0]; g.tile_end[i] = total; }
    g.n = n;
    g.total = total;
    hipLaunchKernelGGL((gemm_h2_pp_group_kernel<BM, BN, WM, WN, NS, PRIO>), dim3(total), dim3(512), 0, s, g);
    return launch_status();
}

}  // namespace

namespace emage_dev {

int run_pp_group(int cfg, GemmArgs** a, int n, hipStream_t s) {
    switch (cfg) {
#ifdef EMAGE_TOOLS
        case 300: return launch_h2_pp_group<128, 96, 4, 2, 4, true>(a, n, s);
        case 301: return launch_h2_pp_group<128, 96, 4, 2, 4, false>(a, n, s);
        case 302: return launch_h2_pp_group<128, 96, 4, 2, 3, true>(a, n, s);
        case 303: return launch_h2_pp_group<128, 96, 4, 2, 5, true>(a, n, s);
#endif
        default: break;
    }
    return EMAGE_EINVAL;
}

int run_pp_config(int cfg, GemmArgs& a, hipStream_t s) {
    switch (cfg) {
#ifdef EMAGE_TOOLS
        // round 6 (VERDICT next #1b): ANTIPHASE tiles — one 8-wave block per CU, two wave groups half a K-step apart (h2_pp_tile.h)
        //                          BM   BN  WM WN NS  PRIO
        case 300: return launch_h2_pp<128, 96, 4, 2, 4, true>(a, s);        // 768-wide: 256 tiles of 128 x 96 at M = 4096; waves 32 x 48
        case 301: return launch_h2_pp<128, 96, 4, 2, 4, false>(a, s);       // ... without s_setprio around the MFMA clusters
        case 302: return launch_h2_pp<128, 96, 4, 2, 3, true>(a, s);        // ring of 3
        case 303: return launch_h2_pp<128, 96, 4, 2, 5, true>(a, s);        // ring of 5 (140 KB)
        case 310: return launch_h2_pp<128, 192, 4, 2, 3, true>(a, s);       // 1536- / 2304-wide: waves 32 x 96
        case 311: return launch_h2_pp<128, 192, 2, 4, 3, true>(a, s);       // waves 64 x 48
        case 312: return launch_h2_pp<128, 192, 2, 4, 4, true>(a, s);       // ring of 4 (160 KB)
        case 320: return launch_h2_pp<64, 192, 2, 4, 4, true>(a, s);        // 64 x 192: waves 32 x 48
        case 321: return launch_h2_pp<64, 192, 2, 4, 5, true>(a, s);
        case 330: return launch_h2_pp<64, 64, 2, 4, 6, true>(a, s);         // the N = 256 heads: one 64 x 64 tile per CU, waves 32 x 16
        case 331: return launch_h2_pp<128, 64, 4, 2, 5, true>(a, s);        // waves 32 x 32
        case 350: return launch_h2_pp<128, 96, 4, 2, 4, true, true>(a, s);  // 300 with the phase tracer
        // v2: where the DMA pieces are issued (MODE 1: in front of the fragment reads; MODE 2: between the MFMAs of the C phase)
        case 340: return launch_h2_pp<128, 96, 4, 2, 4, true, false, 1>(a, s);
        case 341: return launch_h2_pp<128, 96, 4, 2, 4, true, false, 2>(a, s);
        case 342: return launch_h2_pp<128, 96, 4, 2, 5, true, false, 2>(a, s);
        case 343: return launch_h2_pp<128, 96, 4, 2, 5, false, false, 2>(a, s);
        case 344: return launch_h2_pp<64, 192, 2, 4, 5, true, false, 2>(a, s);
        case 345: return launch_h2_pp<128, 192, 2, 4, 4, true, false, 2>(a, s);
        case 346: return launch_h2_pp<128, 192, 4, 2, 4, true, false, 2>(a, s);
        case 351: return launch_h2_pp<128, 96, 4, 2, 4, true, true, 1>(a, s);     // tracers
        case 352: return launch_h2_pp<128, 96, 4, 2, 5, true, true, 2>(a, s);
#endif
        default: break;
    }
    return EMAGE_EINVAL;
}

}  // namespace emage_dev

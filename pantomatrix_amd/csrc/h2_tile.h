// Tile routine of the EMAGE_H2 contraction: BOTH operands arrive pre-split (h2.h), so the K-loop is LDS-DMA + ds_read_b128 +
// v_mfma_f32_16x16x32_f16 only — no VALU.  Shared by gemm_h2.hip (one tile per block) and fused kernels.
//
// Structure (all template parameters):
//   * BM x BN block tile, WM x WN COMPUTE waves (wave tile WTM x WTN = FM x FN fragments of 16 x 16); optionally NLW
//     dedicated LOADER waves that only issue the operand DMA (the compute waves then never stall on the texture path:
//     an LDS-DMA instruction costs its issuing wave 60-180 cycles, MI355X_MICROARCH.md);
//   * NS-deep LDS ring of K-tiles (32 k = 128 bytes per row, image [4 hi chunks | 4 lo chunks] per row, XOR-swizzled slots
//     as in gemm_tile.h), ONE raw s_barrier per K-tile, counted s_waitcnt vmcnt on the issuing waves;
//   * PIPE: the fragments of K-tile k+1 are read from LDS while the MFMAs of K-tile k run (two register sets, reads
//     interleaved between the MFMAs) — affordable here because no fp32 staging registers / split temporaries exist;
//   * swapped MFMA operands + permuted W rows: a lane ends with 8 consecutive output columns of one row (one h2 group or
//     two float4); an odd fragment count leaves one lone fragment with 4 consecutive columns;
//   * tiles inside the V^T column range (out_t) run the MFMAs UN-swapped: a lane then holds 4 consecutive rows of one
//     column and stores them straight into the transposed destination (no LDS staging pass).
#pragma once
#include "gemm_tile.h"
#include "h2.h"

namespace emage_dev {

template <int N> struct IC { static constexpr int value = N; };
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// grouped launches (emage_gemm_grouped): several independent problems of ONE tile configuration in one grid, the argument block by value
constexpr int MAXG = 8;
struct GroupArgs {
    GemmArgs p[MAXG];
    int tile_end[MAXG];          // running sum of tiles_m * tiles_n
    int n, total;
};

template <int BM, int BN, int NS, int KPB = 1> constexpr int h2_smem_bytes() { return NS * KPB * (BM + BN) * 128; }

// ---- LayerNorm fold (GemmArgs::ln_stats / rs_stats / st_out): per-row statistics travel as partials {mean, M2} over 32 columns each ----
// Chan's merge of partial b (32 values) into the running (n, mean, M2)
__device__ __forceinline__ void ln_merge32(float& n, float& mean, float& m2, const float mb, const float m2b) {
    const float nn = n + 32.f, d = mb - mean, w = 32.f / nn;
    mean += d * w;
    m2 += m2b + d * d * n * w;
    n = nn;
}
// (mu, rstd) of the R = WTM rows of a wave tile from their `np` partials each (np % 8 == 0; 24 = the 768-wide residual stream), loaded
// COALESCED: the R rows' partials are one contiguous block of R np 8 bytes, lane L takes bytes [L, L + 1) R np / 8 of it — 64 / R lanes per
// row, np R / 64 partials per lane (np = 24: 6 x 16 bytes per lane at R = 32, all in flight before the first merge) — merges its share (Chan),
// meets the other lanes of its row through lane exchanges (equal counts), and the lane (fr, fg) of the MFMA accumulator layout fetches the
// statistics of its rows 16 i + fr from the lanes that hold them.  (The first version had each lane read 16-byte pieces of ITS row: 64
// scattered pieces per load instruction, ~3 us per launch on the texture path: profiles/r06_ln_fold_per_launch_v1.txt.)
// Split in two so that the loads go out IN FRONT of the operand DMA of the first K-tiles (they return first; the merge then runs while the DMA is
// still landing) — `ln_wave_load` / `ln_wave_finish`.  np == 24 (the 768-wide residual stream; gemm.hip refuses anything else): every trip count
// is a compile-time constant, rows beyond M read the last row's partials (in bounds, never used).
constexpr int LN_NP = 24;
template <int R> struct LnPart { float4 q[LN_NP * R / 128]; };
template <int R>
__device__ __forceinline__ void ln_wave_load(const float* __restrict__ st, const int m_base, const int M, const int lane, LnPart<R>& w) {
    static_assert(R == 16 || R == 32 || R == 64, "rows of a wave tile");
    constexpr int LPR = 64 / R, PER = LN_NP / LPR, NV = PER / 2;    // lanes per row (4, 2, 1), partials / float4s per lane (6 / 3, 12 / 6, 24 / 12)
    const int row = lane / LPR, part = lane % LPR;
    const int m = m_base + row < M ? m_base + row : M - 1;
    const float4* s = (const float4*)(st + ((long)m * LN_NP + part * PER) * 2);
#pragma unroll
    for (int j = 0; j < NV; ++j) w.q[j] = s[j];
}
template <int R, int FM>
__device__ __forceinline__ void ln_wave_finish(const int fr, const float eps, const LnPart<R>& w, float (&mu)[FM], float (&rstd)[FM]) {
    constexpr int LPR = 64 / R, PER = LN_NP / LPR, NV = PER / 2;
    // PER partials of EQUAL count (32): mean = the mean of the means, M2 = sum M2_j + 32 sum (mean_j - mean)^2 — pairwise sums, a dependent
    // chain of ~10 operations (Chan's running merge was 6 per partial: ~0.4 us in front of the K-loop)
    float sm[NV], sq[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { sm[j] = w.q[j].x + w.q[j].z; sq[j] = w.q[j].y + w.q[j].w; }
#pragma unroll
    for (int stp = 1; stp < NV; stp *= 2)
#pragma unroll
        for (int j = 0; j + stp < NV; j += 2 * stp) { sm[j] += sm[j + stp]; sq[j] += sq[j + stp]; }
    float mean = sm[0] * (1.0f / (float)PER), m2 = sq[0];
    float dv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { const float d0 = w.q[j].x - mean, d1 = w.q[j].z - mean; dv[j] = d0 * d0 + d1 * d1; }
#pragma unroll
    for (int stp = 1; stp < NV; stp *= 2)
#pragma unroll
        for (int j = 0; j + stp < NV; j += 2 * stp) dv[j] += dv[j + stp];
    m2 += 32.f * dv[0];
    float cnt = 32.f * (float)PER;                     // values behind each lane's (mean, m2)
    if constexpr (LPR >= 2) {
        const float mo = __shfl_xor(mean, 1), qo = __shfl_xor(m2, 1), d = mo - mean;
        m2 = m2 + qo + d * d * (0.5f * cnt); mean = 0.5f * (mean + mo); cnt *= 2.f;
    }
    if constexpr (LPR >= 4) {
        const float mo = __shfl_xor(mean, 2), qo = __shfl_xor(m2, 2), d = mo - mean;
        m2 = m2 + qo + d * d * (0.5f * cnt); mean = 0.5f * (mean + mo); cnt *= 2.f;
    }
    const float rs = 1.0f / sqrtf(m2 / cnt + eps);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int src = (16 * i + fr) * LPR;
        mu[i] = __shfl(mean, src);
        rstd[i] = __shfl(rs, src);
    }
}

// ---- the epilogue of an EMAGE_H2 tile (gemm_h2_tile; kept separate for fused kernels that end in the same stores) ----
// acc: the wave's FM x FN accumulator fragments; (mw, nw): first row / column of the wave tile.  NAT = false: W rows were fed to the MFMAs in
// the pair-permuted order (a lane ends with 8 consecutive columns per fragment pair; an odd last fragment in natural order); NAT = true:
// every fragment in natural row order (a lane holds 4 consecutive columns per fragment).  Swapped MFMA operands (row-major tiles): lane
// (fr, fg) holds row mw + 16 i + fr; V^T tiles (un-swapped): lane holds rows mw + 16 i + 4 fg + r of one column.
// ln_mu / ln_rs / rs_mu / rs_rs: (mu, rstd) of the lane's FM rows for the folded LayerNorm of the operand / of the residual (row-major tiles;
// V^T tiles fetch theirs here); ignored unless p.ln_stats / p.rs_stats
template <int FM, int FN, bool NAT, bool PRE, int PM, int PP, bool LNF = false, bool SKF = false>
__device__ __forceinline__ void h2_tile_epilogue(const GemmArgs& p, f32x4 (&acc)[FM][FN], const int mw, const int nw, const int fr, const int fg,
                                                 const bool vt_tile, const float (&pre_r)[PM][PP][8], const int split = 0,
                                                 const float* ln_mu = nullptr, const float* ln_rs = nullptr, const float* rs_mu = nullptr, const float* rs_rs = nullptr) {
    constexpr int FP = NAT ? 0 : FN / 2;
    constexpr bool LONE = !NAT && (FN & 1) != 0;
    constexpr int NLONE = NAT ? FN : (LONE ? 1 : 0);          // trailing fragments handled 4 columns at a time
    const int ncol_n = p.out_t ? p.t_col0 : p.N;     // columns below this go to out / out_f32
    h2_t* __restrict__ out = (h2_t*)p.out;
    const float os = p.o_scale;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = acc[i][j] * os;

    if (vt_tile) {
        // un-swapped MFMAs: lane (fr, fg) holds rows m0 + wm*WTM + 16 i + 4 fg + r (r = 0..3) of column n(j, fr)
        float* __restrict__ out_t = (float*)p.out_t;
        const int t_ncols = p.N - p.t_col0;
        const bool tvec = (p.t_rows % 4 == 0) && (p.t_ld % 4 == 0) && (((uintptr_t)out_t & 15) == 0);
        // folded LayerNorm of the operand: ln_mu / ln_rs hold the statistics of row mw + 16 i + fr (the row-major layout's rows); this lane's
        // rows mw + 16 i + 4 fg + r live in lanes 4 fg + r
        float vmu[LNF ? FM : 1][4], vrs[LNF ? FM : 1][4];
        if constexpr (LNF) {
            if (p.ln_stats) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { vmu[i][r] = __shfl(ln_mu[i], 4 * fg + r); vrs[i][r] = __shfl(ln_rs[i], 4 * fg + r); }
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int wrow = (NAT || (LONE && j == FN - 1)) ? j * 16 + fr : (j >> 1) * 32 + (j & 1) * 4 + 8 * (fr >> 2) + (fr & 3);
            const int n = nw + wrow;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f, sv = p.slope ? p.slope[n] : 1.f;
            const float cv = (LNF && p.ln_stats) ? p.ln_c[n] : 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = mw + i * 16 + fg * 4;
                if (m >= p.M) continue;
                float v[4];
                if constexpr (LNF) {
                    if (p.ln_stats) {                    // folded LayerNorm of the operand
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = vrs[i][r] * (acc[i][j][r] - vmu[i][r] * cv);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = leaky(acc[i][j][r] + bv, sv);
                if (tvec && m + 3 < p.M) {
                    const int b = m / p.t_rows, l = m - b * p.t_rows;
                    *(float4*)(out_t + ((long)b * t_ncols + (n - p.t_col0)) * p.t_ld + l) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mm = m + r;
                        if (mm < p.M) {
                            const int b = mm / p.t_rows, l = mm - b * p.t_rows;
                            out_t[((long)b * t_ncols + (n - p.t_col0)) * p.t_ld + l] = v[r];
                        }
                    }
                }
            }
        }
        return;
    }

    // ---- row-major epilogue.  Lane (fr, fg): row m0 + wm*WTM + 16 i + fr; fragment pair jp -> 8 consecutive columns
    // n0 + wn*WTN + jp*32 + fg*8 + e (e < 4 from acc[i][2jp], e >= 4 from acc[i][2jp+1]); lone fragment -> 4 columns ----
    const int n_lim = ncol_n > p.n_store ? ncol_n : p.n_store;   // columns any store may touch (`out` zero-fills [N, n_store))
    // split-K with a workspace (emage_gemm_ws): this slice's partial tile goes to ITS plane with plain stores — block-uniform scalars
    const int ksplit = (SKF && p.sk_count) ? 1 : p.ksplit;        // a fix-up launch ends in ONE ordinary epilogue per tile
    const bool to_plane = ksplit > 1 && p.ws != nullptr;
    float* __restrict__ of32 = to_plane ? p.ws + (long)split * p.ws_plane : p.out_f32;
    const int ldf = to_plane ? p.ldws : p.ldf;
    const bool f32_vec = of32 && (ldf % 4 == 0) && (((uintptr_t)of32 & 15) == 0);
    auto finish = [&](auto wc, const int m, const int n, float (&x)[decltype(wc)::value], const float (&rpre)[decltype(wc)::value], const bool have_pre, const int i) {
        // x: accumulators (already scaled) of W consecutive columns n.. of row m (fragment row i of the lane) -> bias, residual, activation, stores
        constexpr int W = decltype(wc)::value;
        const bool full = n + W <= ncol_n;
        float bv[W], sv[W], rv[W];
#pragma unroll
        for (int e = 0; e < W; ++e) { bv[e] = 0.f; sv[e] = 1.f; rv[e] = 0.f; }
        if (LNF && p.ln_stats) {                        // folded LayerNorm of the operand: x <- rstd (x - mu c)
            const float mu = ln_mu[i], rs = ln_rs[i];
            float cv[W];
            if (EMAGE_DBG(p, 1024)) {
#pragma unroll
                for (int e = 0; e < W; ++e) cv[e] = 0.5f;
            } else if (full) { if constexpr (W == 8) load8<float>(p.ln_c + n, cv); else { const float4 q = *(const float4*)(p.ln_c + n); cv[0] = q.x; cv[1] = q.y; cv[2] = q.z; cv[3] = q.w; } }
            else {
#pragma unroll
                for (int e = 0; e < W; ++e) cv[e] = n + e < ncol_n ? p.ln_c[n + e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < W; ++e) x[e] = rs * (x[e] - mu * cv[e]);
        }
        if (full) {
            if (p.bias) { if constexpr (W == 8) load8<float>(p.bias + n, bv); else { const float4 t = *(const float4*)(p.bias + n); bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w; } }
            if (p.slope) { if constexpr (W == 8) load8<float>(p.slope + n, sv); else { const float4 t = *(const float4*)(p.slope + n); sv[0] = t.x; sv[1] = t.y; sv[2] = t.z; sv[3] = t.w; } }
            if (have_pre) {
#pragma unroll
                for (int e = 0; e < W; ++e) rv[e] = rpre[e];
            } else if (p.res) {
                if (p.res_is_f32) {
                    if constexpr (W == 8) load8<float>((const float*)p.res + (long)m * p.ldr + n, rv);
                    else { const float4 t = *(const float4*)((const float*)p.res + (long)m * p.ldr + n); rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w; }
                } else {
                    if constexpr (W == 8) h2_load8((const h2_t*)p.res + (long)m * p.ldr + n, rv, p.h2i);
                    else h2_load4((const h2_t*)p.res + (long)m * p.ldr + n, n, rv, p.h2i);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                if (n + e < ncol_n) {
                    if (p.bias) bv[e] = p.bias[n + e];
                    if (p.slope) sv[e] = p.slope[n + e];
                }
            }
            if (p.res && n < ncol_n) {               // the group exists in the residual's padded row
                float t[W];
                if (p.res_is_f32) {
#pragma unroll
                    for (int e = 0; e < W; ++e) t[e] = n + e < ncol_n ? ((const float*)p.res)[(long)m * p.ldr + n + e] : 0.f;
                } else {
                    if constexpr (W == 8) h2_load8((const h2_t*)p.res + (long)m * p.ldr + n, t, p.h2i);
                    else h2_load4((const h2_t*)p.res + (long)m * p.ldr + n, n, t, p.h2i);
                }
#pragma unroll
                for (int e = 0; e < W; ++e) rv[e] = n + e < ncol_n ? t[e] : 0.f;
            }
        }
        if (LNF && p.rs_stats && p.res) {               // the residual is a folded LayerNorm of the raw sum just read
            const float mu = rs_mu[i], rs = rs_rs[i];
            float gv[W], bt[W];
            if (EMAGE_DBG(p, 1024)) {
#pragma unroll
                for (int e = 0; e < W; ++e) { gv[e] = 1.f; bt[e] = 0.f; }
            } else if (full) {
                if constexpr (W == 8) { load8<float>(p.rs_gamma + n, gv); load8<float>(p.rs_beta + n, bt); }
                else { const float4 q = *(const float4*)(p.rs_gamma + n), r4 = *(const float4*)(p.rs_beta + n);
                       gv[0] = q.x; gv[1] = q.y; gv[2] = q.z; gv[3] = q.w; bt[0] = r4.x; bt[1] = r4.y; bt[2] = r4.z; bt[3] = r4.w; }
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e) { gv[e] = n + e < ncol_n ? p.rs_gamma[n + e] : 0.f; bt[e] = n + e < ncol_n ? p.rs_beta[n + e] : 0.f; }
            }
#pragma unroll
            for (int e = 0; e < W; ++e) rv[e] = n + e < ncol_n ? (rv[e] - mu) * rs * gv[e] + bt[e] : 0.f;
        }
        float v[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float y = x[e] + bv[e];
            if (p.res_first) y += rv[e];
            y = leaky(y, sv[e]);
            if (!p.res_first) y += rv[e];
            v[e] = (n + e < ncol_n) ? y : 0.f;
        }
        if (out && n < n_lim) {
            // out rows are padded to a multiple of 8 columns (host contract): the whole group is always addressable
            if constexpr (W == 8) h2_store8(out + (long)m * p.ldo + n, v, p.h2s);
            else h2_store4(out + (long)m * p.ldo + n, n, v, p.h2s);
        }
        if constexpr (LNF && W == 8 && FN == 2 && !NAT) {
            if (p.st_out) {                             // partial row statistics of the 32 columns this wave holds of row m: 8 per lane, 4 lanes (fg) per row
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += v[e];
                float mean = sm * 0.125f, m2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; m2 += d * d; }
                // Chan's merge of equal counts, 8 -> 16 -> 32 (lanes fr + 16 fg: the same row)
                { const float mo = __shfl_xor(mean, 16), qo = __shfl_xor(m2, 16), d = mo - mean; m2 = m2 + qo + d * d * 4.f; mean = 0.5f * (mean + mo); }
                { const float mo = __shfl_xor(mean, 32), qo = __shfl_xor(m2, 32), d = mo - mean; m2 = m2 + qo + d * d * 8.f; mean = 0.5f * (mean + mo); }
                if (fg == 0) *(float2*)(p.st_out + ((long)m * (p.N >> 5) + (n >> 5)) * 2) = make_float2(mean, m2);
            }
        }
        if (of32 && n < ncol_n) {
            float* dst = of32 + (long)m * ldf + n;
            if (ksplit > 1 && !to_plane) {            // partial sums of the K-slices meet in memory (the destination was cleared by the host call)
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (n + e < ncol_n) __hip_atomic_fetch_add(dst + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (full && f32_vec) {
                if constexpr (W == 8) store8<float>(dst, v);
                else *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (n + e < ncol_n) dst[e] = v[e];
            }
        }
    };
#pragma unroll
    for (int jp = 0; jp < FP; ++jp) {
        const int n = nw + jp * 32 + fg * 8;
        if (n >= n_lim) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mw + i * 16 + fr;
            if (m >= p.M) continue;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = e < 4 ? acc[i][2 * jp][e] : acc[i][2 * jp + 1][e - 4];
            finish(IC<8>{}, m, n, x, pre_r[i % PM][jp % PP], PRE && p.res && n + 8 <= ncol_n, i);
        }
    }
#pragma unroll
    for (int jl = FN - NLONE; jl < FN; ++jl) {
        const int n = nw + jl * 16 + fg * 4;
        if (n < n_lim) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = mw + i * 16 + fr;
                if (m >= p.M) continue;
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = acc[i][jl][e];
                const float none[4] = {0.f, 0.f, 0.f, 0.f};
                finish(IC<4>{}, m, n, x, none, false, i);
            }
        }
    }
}

// KPB (round 5, VERDICT next #1b "BK = 64"): K-tiles per ring slot and per s_barrier — a slot holds KPB consecutive 32-k sub-tiles (each in the
// unchanged [4 hi | 4 lo] row image), one rendezvous and one counted wait serve KPB x (3 FM FN) MFMAs per wave; the MFMA order per
// accumulator is that of KPB = 1, so the result is the same bits.  Plain K-loop only (not PIPE / DILV).
// LNF: the instantiation carries the LayerNorm-fold paths (GemmArgs::ln_stats / rs_stats / st_out); the plain one ignores those fields
// SKF: the instantiation carries the in-kernel split-K fix-up (GemmArgs::sk_ws / sk_count)
template <int BM, int BN, int WM, int WN, int NS, int NLW, bool PIPE, bool PRE, bool DILV = false, bool TRACE = false, int KPB = 1, bool LNF = false, bool SKF = false>
__device__ __forceinline__ void gemm_h2_tile(const GemmArgs& p, const int m0, const int n0, unsigned char* smem, const int split = 0) {
    constexpr int ES = 4, BK = 32, RB = 128, RPI = 8;
    constexpr int NCW = WM * WN, NL = NLW ? NLW : NCW;
    constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16, FP = FN / 2;
    constexpr bool LONE = (FN & 1) != 0;
    constexpr int GA = BM / RPI / NL, GB = BN / RPI / NL, G = GA + GB;
    static_assert((BM / RPI) % NL == 0 && (BN / RPI) % NL == 0, "every loading wave issues the same number of DMA instructions");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && NS >= 2 && NS <= 8, "tile shape");
    static_assert(NS <= 4 || !PIPE, "rings deeper than 4: the plain K-loop");
    static_assert((NS - 2) * (BM / 8 / (NLW ? NLW : WM * WN) + BN / 8 / (NLW ? NLW : WM * WN)) * KPB <= 63, "vmcnt is a 6-bit counter");
    static_assert(!PIPE || NS >= 3, "the register-pipelined K-loop reads one stage ahead: ring of >= 3");
    static_assert(KPB == 1 || (!PIPE && !DILV && KPB == 2), "several K-tiles per slot: the plain K-loop");
    constexpr int SUB = (BM + BN) * RB;               // one 32-k sub-tile of a slot
    constexpr int STAGE = KPB * SUB;
    constexpr int AH = PIPE ? 1 : 0;                  // stages the fragment reads run ahead of the MFMAs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = NLW ? wave >= NCW : true;
    const bool is_compute = NLW ? wave < NCW : true;
    const int lw = NLW ? wave - NCW : wave;
    const int wm = (wave % NCW) / WN, wn = (wave % NCW) % WN;
    // TRACE (tools): lane 0 of every wave of block 0 stamps s_memtime at the phase boundaries: trace[wave * 512 + k]
    int tr_n = 0;
    auto tr = [&]() {
        if constexpr (TRACE) {
            if (p.trace && blockIdx.x == 0 && tr_n < 511) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) p.trace[wave * 512 + 1 + tr_n] = t;
                ++tr_n;
            }
        }
    };
    tr();

    // ---- operand DMA (loading waves): lane-fixed byte offsets, K advance in the scalar offset ----
    const int nbatch = p.M / p.Lout;
    const unsigned a_shift = (unsigned)p.pad * (unsigned)(p.lda * ES);
    const unsigned a_bytes = (unsigned)((((long)nbatch * p.Lin - 1) * p.lda + p.Cp) * ES) + a_shift;
    const unsigned w_bytes = (unsigned)((long)p.N * p.K * ES);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A - a_shift), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);
    const int lrow = lane >> 3, lslot = lane & 7;
    const bool is_conv = p.taps > 1;
    // LDS slot s of row r holds logical chunk lc = s ^ swz(r): lc 0..3 = hi of k-group lc, 4..7 = lo of k-group lc - 4;
    // in memory (h2.h) k-group g is the 32 bytes [hi | lo] at g*32: source chunk = 2*(lc & 3) + (lc >> 2)
    unsigned a_voff[GA]; int a_lpos[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int row = (lw + NL * j) * RPI + lrow;
        const int m = m0 + row;
        const int lc = lslot ^ swz<8>(row);
        const unsigned chunk = (unsigned)((2 * (lc & 3) + (lc >> 2)) * 16);
        if (!is_conv) {
            a_lpos[j] = 0;
            a_voff[j] = m < p.M ? (unsigned)m * (unsigned)(p.lda * ES) + chunk : OOB;
        } else {
            const int mm = m < p.M ? m : 0;
            const int b = mm / p.Lout, l = mm - b * p.Lout;
            a_lpos[j] = m < p.M ? l * p.stride - p.pad : -0x40000000;
            a_voff[j] = (unsigned)(((long)b * p.Lin + l * p.stride) * p.lda * ES) + chunk;
        }
    }
    unsigned b_voff[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int row = (lw + NL * j) * RPI + lrow;
        const int n = n0 + row;
        const int lc = lslot ^ swzW<8>(row);
        b_voff[j] = n < p.N ? (unsigned)n * (unsigned)(p.K * ES) + (unsigned)((2 * (lc & 3) + (lc >> 2)) * 16) : OOB;
    }
    // split-K (p.ksplit > 1, Linear only): this block contracts K-tiles [kt0, kt0 + nk) and adds its partial tile into out_f32
    const int nk_all = p.K / BK;
    const int kt0 = p.ksplit > 1 ? split * p.nk_split : 0;
    // (a slice of a convolution starts inside tap (kt0 BK) / Cp: the scalar offset of A advances lda elements per tap)
    int is_tap = (kt0 * BK) / p.Cp, is_c0 = kt0 * BK - is_tap * p.Cp, is_slot = 0, is_sub = 0;
    unsigned soff_a = (unsigned)((is_tap * p.lda + is_c0) * ES), soff_w = (unsigned)(kt0 * BK * ES);
    const unsigned tap_step = (unsigned)(p.lda - p.Cp + BK) * ES;
    // DMA instruction J of a stage (J < GA: A rows, else W rows) and the bookkeeping that follows the last one
    auto issue_piece = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (EMAGE_DBG(p, 1)) return;                   // tools, timing only (emage_set_tuning key 1): no operand DMA
        unsigned char* base = smem + is_slot * STAGE + (KPB > 1 ? is_sub * SUB : 0);
        if constexpr (J < GA) {
            unsigned vo = a_voff[J];
            if (is_conv) vo = (unsigned)(a_lpos[J] + is_tap) < (unsigned)p.Lin ? vo : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + (lw + NL * J) * 1024),
                                                     16, (int)vo, (int)soff_a, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + BM * RB + (lw + NL * (J - GA)) * 1024),
                                                     16, (int)b_voff[J - GA], (int)soff_w, 0, 0);
        }
    };
    auto issue_advance = [&]() {
        soff_w += BK * ES;
        is_c0 += BK;
        if (is_c0 == p.Cp) { is_c0 = 0; ++is_tap; soff_a += tap_step; } else { soff_a += BK * ES; }
        if constexpr (KPB > 1) {
            if (++is_sub == KPB) { is_sub = 0; if (++is_slot == NS) is_slot = 0; }
        } else {
            if (++is_slot == NS) is_slot = 0;
        }
    };
    auto issue = [&]() {                              // one ring slot: KPB sub-tiles
        static_for<KPB>([&](auto) {
            static_for<G>([&](auto jc) { issue_piece(jc); });
            issue_advance();
        });
    };

    // ---- epilogue operands fetched ahead of the K-loop (compute waves; oldest entries of their memory queue) ----
    const int fr = lane & 15, fg = lane >> 4;
    const bool vt_tile = p.out_t != nullptr && n0 >= p.t_col0;       // block-uniform
    const int ncol_n = p.out_t ? p.t_col0 : p.N;     // columns below this go to out / out_f32
    h2_t* __restrict__ out = (h2_t*)p.out;
    constexpr int PM = PRE ? FM : 1, PP = PRE ? FP : 1;
    float pre_r[PM][PP][8];
    if constexpr (PRE) {
#pragma unroll
        for (int jp = 0; jp < PP; ++jp) {
            const int n = n0 + wn * WTN + jp * 32 + fg * 8;
#pragma unroll
            for (int i = 0; i < PM; ++i) {
                const int m = m0 + wm * WTM + i * 16 + fr;
#pragma unroll
                for (int e = 0; e < 8; ++e) pre_r[i][jp][e] = 0.f;
                if (is_compute && !vt_tile && p.res && m < p.M && n + 8 <= ncol_n) {
                    if (p.res_is_f32) load8<float>((const float*)p.res + (long)m * p.ldr + n, pre_r[i][jp]);
                    else h2_load8((const h2_t*)p.res + (long)m * p.ldr + n, pre_r[i][jp], p.h2i);
                }
            }
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.ksplit > 1 ? (nk_all - kt0 < p.nk_split ? nk_all - kt0 : p.nk_split) : nk_all) / KPB;      // ring slots to walk (K-tiles: even, gemm_h2.hip)
    if (is_loader) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nk) issue();
    }
    // folded LayerNorms (GemmArgs::ln_stats / rs_stats): the wave tile's partial row statistics are requested here, behind the first K-tiles'
    // DMA (in front of it, a register-reuse wait of the compiler stalled the DMA issue), ...
    LnPart<LNF ? WTM : 16> lnp, rsp;
    if constexpr (LNF) {
        if (is_compute && p.ln_stats && !EMAGE_DBG(p, 256)) ln_wave_load<WTM>(p.ln_stats, m0 + wm * WTM, p.M, lane, lnp);       // (tools, timing only: bit 256 = no statistics loads,
        if (is_compute && p.rs_stats && !EMAGE_DBG(p, 256)) ln_wave_load<WTM>(p.rs_stats, m0 + wm * WTM, p.M, lane, rsp);       //  512 = no merge, 1024 = no c / gamma / beta loads)
    }

    // ---- folded LayerNorms (GemmArgs::ln_stats / rs_stats): (mu, rstd) of the lane's FM rows ----
    float ln_mu[FM], ln_rs[FM], rs_mu[FM], rs_rs[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) { ln_mu[i] = 0.f; ln_rs[i] = 1.f; rs_mu[i] = 0.f; rs_rs[i] = 1.f; }

    // ---- fragment read addresses ----
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int arow = wm * WTM + fr;
    const int brow = wn * WTN + 8 * (fr >> 2) + (fr & 3);
    const int browL = wn * WTN + (FN - 1) * 16 + fr;             // lone fragment: natural row order
    const unsigned a_rd = lds0 + arow * RB + ((fg ^ swz<8>(arow)) << 4);
    const unsigned b_rd = lds0 + BM * RB + brow * RB + ((fg ^ swzW<8>(brow)) << 4);
    const unsigned b_rdL = lds0 + BM * RB + browL * RB + ((fg ^ swzW<8>(browL)) << 4);

    struct Frag { u32x4 ah[FM], al[FM], wh[FN], wl[FN]; };
    constexpr int NR = 2 * FM + 2 * FN;              // ds_read_b128 per K-tile
    // read R of the K-tile at ring byte offset sb; order: A hi, W lo (the operands of the first MFMA sweep), W hi, A lo
    auto read_one = [&](auto rc, Frag& f, const unsigned sb) {
        constexpr int R = decltype(rc)::value;
        if constexpr (R < FM) {
            f.ah[R] = lds_read128_off<R * 16 * RB>(a_rd + sb);
        } else if constexpr (R < FM + FN) {
            constexpr int J = R - FM;
            if constexpr (LONE && J == FN - 1) f.wl[J] = lds_read128_off<0>((b_rdL ^ 64u) + sb);
            else f.wl[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * RB>((b_rd ^ 64u) + sb);
        } else if constexpr (R < FM + 2 * FN) {
            constexpr int J = R - FM - FN;
            if constexpr (LONE && J == FN - 1) f.wh[J] = lds_read128_off<0>(b_rdL + sb);
            else f.wh[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * RB>(b_rd + sb);
        } else {
            constexpr int I = R - FM - 2 * FN;
            f.al[I] = lds_read128_off<I * 16 * RB>((a_rd ^ 64u) + sb);
        }
    };
    // the K-loop, instantiated for swapped (row-major tiles) and un-swapped (V^T tiles) MFMA operands
    auto kloop = [&](auto vtc) __attribute__((always_inline)) {
        constexpr bool VT = decltype(vtc)::value != 0;
        // MFMA Q of a K-tile: three sweeps (W lo x A hi, W hi x A lo, W hi x A hi: small terms first), fragment-major inside a
        // sweep so that back-to-back MFMAs never share an accumulator
        auto mma_one = [&](auto qc, const Frag& f) {
            constexpr int Q = decltype(qc)::value;
            constexpr int t = Q / (FM * FN), i = (Q % (FM * FN)) / FN, j = Q % FN;
            const f16x8 w = __builtin_bit_cast(f16x8, t == 0 ? f.wl[j] : f.wh[j]);
            const f16x8 a = __builtin_bit_cast(f16x8, t == 1 ? f.al[i] : f.ah[i]);
            if constexpr (VT) acc[i][j] = mma_f16(a, w, acc[i][j]);
            else acc[i][j] = mma_f16(w, a, acc[i][j]);
        };
        constexpr int NM = 3 * FM * FN;

        auto wait_landed = [&](const int st) {
            // the loading waves wait until stage st + AH has landed; stages issued after it may stay in flight
            const int last_issued = (st + NS - 2 < nk - 1) ? st + NS - 2 : nk - 1;
            const int infl = last_issued - (st + AH);      // 0 .. NS - 2 ring slots stay in flight behind the one needed now
            if constexpr (NS <= 4) {
                if (NS >= 4 && infl >= 2) wait_vmcnt<2 * G * KPB>();
                else if (NS >= 3 && infl >= 1) wait_vmcnt<G * KPB>();
                else wait_vmcnt<0>();
            } else {                                       // deep rings (round 5: lone blocks on small grids): one counted wait per depth
                static_for<NS - 1>([&](auto kc) {
                    constexpr int KQ = decltype(kc)::value;
                    if (infl == KQ) wait_vmcnt<KQ * G * KPB>();
                });
            }
        };

        // DILV: the DMA instructions of the stage issued in this iteration are spread between the MFMAs instead of in front of them
        // (a barrier per K-tile keeps every wave in the same phase: a burst of DMA issues right behind it stalls all waves on
        // the texture path while the matrix pipes idle; behind an MFMA, one wave's issue stall is covered by its SIMD partner)
        constexpr int DSTEP = NM / G > 0 ? NM / G : 1;
        if constexpr (!PIPE) {
            unsigned sb = 0;
            for (int st = 0; st < nk; ++st) {
                tr();
                if (is_loader) wait_landed(st);
                tr();
                __builtin_amdgcn_s_barrier();
                tr();
                const bool do_issue = is_loader && st + NS - 1 < nk;
                if (!(DILV && NLW == 0) && do_issue) issue();
                tr();
                if (is_compute) {
                  static_for<KPB>([&](auto subc) {
                    constexpr int SB2 = decltype(subc)::value * SUB;
                    Frag f;
                    if (!EMAGE_DBG(p, 2)) static_for<NR>([&](auto rc) { read_one(rc, f, sb + SB2); });        // (tools, timing only: bit 2 = no fragment reads)
                    else { static_for<FM>([&](auto ic) { f.ah[decltype(ic)::value] = f.al[decltype(ic)::value] = u32x4{0u, 0u, 0u, 0u}; });
                           static_for<FN>([&](auto jc) { f.wh[decltype(jc)::value] = f.wl[decltype(jc)::value] = u32x4{0u, 0u, 0u, 0u}; }); }
                    tr();
                    wait_lgkmcnt<FN + FM>();              // A hi and W lo are there: first sweep
                    __builtin_amdgcn_sched_barrier(0);
                    tr();
                    static_for<NM>([&](auto qc) {
                        constexpr int Q = decltype(qc)::value;
                        if constexpr (Q == FM * FN) {
                            __builtin_amdgcn_sched_barrier(0);
                            wait_lgkmcnt<0>();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (!EMAGE_DBG(p, 4)) mma_one(qc, f);                                                 // (bit 4 = no MFMAs)
                        if constexpr (DILV && NLW == 0 && Q % DSTEP == DSTEP - 1 && Q / DSTEP < G) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (do_issue) issue_piece(IC<Q / DSTEP>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    if constexpr (decltype(subc)::value + 1 < KPB) __builtin_amdgcn_sched_barrier(0);
                  });
                    if constexpr (DILV && NLW == 0) {
                        static_for<G>([&](auto jc) {      // pieces the interleave did not reach
                            if constexpr (decltype(jc)::value >= NM / DSTEP) { if (do_issue) issue_piece(jc); }
                        });
                        if (do_issue) issue_advance();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    tr();
                }
                sb += STAGE;
                if (sb == NS * STAGE) sb = 0;
            }
        } else {
            Frag f0, f1;
            // prologue: stage 0 landed -> first fragments
            if (is_loader) {
                const int last_issued = (NS - 2 < nk - 1) ? NS - 2 : nk - 1;
                if (NS >= 4 && last_issued >= 2) wait_vmcnt<2 * G>();
                else if (NS >= 3 && last_issued >= 1) wait_vmcnt<G>();
                else wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            if (is_compute) {
                static_for<NR>([&](auto rc) { read_one(rc, f0, 0u); });
                wait_lgkmcnt<0>();
            }
            unsigned sb = STAGE;                          // ring offset of stage st + 1
            // reads of the next K-tile are spread between this K-tile's MFMAs: one read after every RSTEP-th MFMA
            constexpr int RSTEP = NM / NR > 0 ? NM / NR : 1;
            auto step = [&](const int st, Frag& cur, Frag& nxt) {
                tr();
                if (is_loader) wait_landed(st);
                tr();
                __builtin_amdgcn_s_barrier();
                tr();
                const bool do_issue = is_loader && st + NS - 1 < nk;
                if (!(DILV && NLW == 0) && do_issue) issue();
                tr();
                if (is_compute) {
                    const bool more = st + 1 < nk;        // wave-uniform
                    static_for<NM>([&](auto qc) {
                        constexpr int Q = decltype(qc)::value;
                        mma_one(qc, cur);
                        if constexpr (Q % RSTEP == RSTEP - 1 && Q / RSTEP < NR) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) read_one(IC<Q / RSTEP>{}, nxt, sb);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (DILV && NLW == 0 && Q % DSTEP == DSTEP / 2 && Q / DSTEP < G) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (do_issue) issue_piece(IC<Q / DSTEP>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    static_for<NR>([&](auto rc) {         // reads the interleave did not reach
                        constexpr int R = decltype(rc)::value;
                        if constexpr (R >= NM / RSTEP) { if (more) read_one(rc, nxt, sb); }
                    });
                    if constexpr (DILV && NLW == 0) {
                        static_for<G>([&](auto jc) {
                            if constexpr (decltype(jc)::value >= (NM + DSTEP - 1 - DSTEP / 2) / DSTEP) { if (do_issue) issue_piece(jc); }
                        });
                        if (do_issue) issue_advance();
                    }
                    tr();
                    wait_lgkmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    tr();
                }
                sb += STAGE;
                if (sb == NS * STAGE) sb = 0;
            };
            for (int st = 0; st < nk; st += 2) {          // nk is even (Cp % 64 == 0)
                step(st, f0, f1);
                step(st + 1, f1, f0);
            }
        }
    };
    tr();
    if (vt_tile) kloop(IC<1>{}); else kloop(IC<0>{});
    tr();

    if (!is_compute) { if constexpr (TRACE) { if (p.trace && blockIdx.x == 0 && lane == 0) p.trace[wave * 512] = (unsigned long long)tr_n; } __syncthreads(); return; }
    if constexpr (PRE && NLW > 0) wait_vmcnt<0>();

    if constexpr (SKF) {
        if (p.sk_count) {
            // ---- split-K fix-up: this block's accumulators go to its slot of the workspace (write-through: other XCDs' L2s are not coherent);
            // the last block of the tile to arrive adds the slots in slice order (the same bits whoever is last) and goes on to the epilogue ----
            static_assert(NLW == 0, "every wave computes");
            constexpr int NF = FM * FN, NT = NCW * 64;
            const int tile_id = (m0 / BM) * p.tiles_n + n0 / BN;
            const unsigned slot_bytes = (unsigned)(NT * NF * 16);
            const __amdgpu_buffer_rsrc_t ws_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_ws, 0, (unsigned)(p.tiles_m * p.tiles_n * p.ksplit) * slot_bytes, 0x00020000);
            const unsigned mine = (unsigned)(tile_id * p.ksplit + split) * slot_bytes + (unsigned)tid * 16;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), ws_rsrc, (int)(mine + (unsigned)((i * FN + j) * NT * 16)), 0, 16);     // aux 16 = sc1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = (int*)smem;                    // the operand ring is idle behind the K-loop
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(p.sk_count + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == p.ksplit - 1;
                if (last) __hip_atomic_store(p.sk_count + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                *flag = last;
            }
            __syncthreads();
            if (*flag == 0) return;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < p.ksplit; ++sl) {
                const unsigned at = (unsigned)(tile_id * p.ksplit + sl) * slot_bytes + (unsigned)tid * 16;
                f32x4 part[FM][FN];
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        part[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws_rsrc, (int)(at + (unsigned)((i * FN + j) * NT * 16)), 0, 16));
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = acc[i][j] + part[i][j];
            }
        }
    }
    if constexpr (LNF) {
        // ... and merged HERE, behind the K-loop: the partials arrived long ago, and no wait for them sits in front of the loop (merged in the
        // prologue, the compiler's vmcnt(0) for these loads also drained the first K-tiles' DMA: +1 us per launch, profiles/r06_ln_fold_per_launch_v3.txt)
        if (p.ln_stats && !EMAGE_DBG(p, 256 | 512)) ln_wave_finish<WTM, FM>(fr, p.ln_eps, lnp, ln_mu, ln_rs);     // (wave-uniform conditions: the lane exchanges inside need every lane)
        if (p.rs_stats && !EMAGE_DBG(p, 256 | 512)) ln_wave_finish<WTM, FM>(fr, p.ln_eps, rsp, rs_mu, rs_rs);
    }
    h2_tile_epilogue<FM, FN, false, PRE, PM, PP, LNF, SKF>(p, acc, m0 + wm * WTM, n0 + wn * WTN, fr, fg, vt_tile, pre_r, split, ln_mu, ln_rs, rs_mu, rs_rs);
    if (vt_tile) { __syncthreads(); return; }
    tr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr();
    if constexpr (TRACE) { if (p.trace && blockIdx.x == 0 && lane == 0) p.trace[wave * 512] = (unsigned long long)tr_n; }
    __syncthreads();
}

}  // namespace emage_dev

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

// ---- the epilogue of an EMAGE_H2 tile (gemm_h2_tile; kept separate for fused kernels that end in the same stores) ----
// acc: the wave's FM x FN accumulator fragments; (mw, nw): first row / column of the wave tile.  NAT = false: W rows were fed to the MFMAs in
// the pair-permuted order (a lane ends with 8 consecutive columns per fragment pair; an odd last fragment in natural order); NAT = true:
// every fragment in natural row order (a lane holds 4 consecutive columns per fragment).  Swapped MFMA operands (row-major tiles): lane
// (fr, fg) holds row mw + 16 i + fr; V^T tiles (un-swapped): lane holds rows mw + 16 i + 4 fg + r of one column.
template <int FM, int FN, bool NAT, bool PRE, int PM, int PP>
__device__ __forceinline__ void h2_tile_epilogue(const GemmArgs& p, f32x4 (&acc)[FM][FN], const int mw, const int nw, const int fr, const int fg,
                                                 const bool vt_tile, const float (&pre_r)[PM][PP][8], const int split = 0) {
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
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int wrow = (NAT || (LONE && j == FN - 1)) ? j * 16 + fr : (j >> 1) * 32 + (j & 1) * 4 + 8 * (fr >> 2) + (fr & 3);
            const int n = nw + wrow;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f, sv = p.slope ? p.slope[n] : 1.f;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = mw + i * 16 + fg * 4;
                if (m >= p.M) continue;
                float v[4];
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
    const bool to_plane = p.ksplit > 1 && p.ws != nullptr;
    float* __restrict__ of32 = to_plane ? p.ws + (long)split * p.ws_plane : p.out_f32;
    const int ldf = to_plane ? p.ldws : p.ldf;
    const bool f32_vec = of32 && (ldf % 4 == 0) && (((uintptr_t)of32 & 15) == 0);
    auto finish = [&](auto wc, const int m, const int n, float (&x)[decltype(wc)::value], const float (&rpre)[decltype(wc)::value], const bool have_pre) {
        // x: accumulators (already scaled) of W consecutive columns n.. of row m -> bias, residual, activation, stores
        constexpr int W = decltype(wc)::value;
        const bool full = n + W <= ncol_n;
        float bv[W], sv[W], rv[W];
#pragma unroll
        for (int e = 0; e < W; ++e) { bv[e] = 0.f; sv[e] = 1.f; rv[e] = 0.f; }
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
                    if constexpr (W == 8) h2_load8((const h2_t*)p.res + (long)m * p.ldr + n, rv);
                    else h2_load4((const h2_t*)p.res + (long)m * p.ldr + n, n, rv);
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
                    if constexpr (W == 8) h2_load8((const h2_t*)p.res + (long)m * p.ldr + n, t);
                    else h2_load4((const h2_t*)p.res + (long)m * p.ldr + n, n, t);
                }
#pragma unroll
                for (int e = 0; e < W; ++e) rv[e] = n + e < ncol_n ? t[e] : 0.f;
            }
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
            if constexpr (W == 8) h2_store8(out + (long)m * p.ldo + n, v);
            else h2_store4(out + (long)m * p.ldo + n, n, v);
        }
        if (of32 && n < ncol_n) {
            float* dst = of32 + (long)m * ldf + n;
            if (p.ksplit > 1 && !to_plane) {          // partial sums of the K-slices meet in memory (the destination was cleared by the host call)
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
            finish(IC<8>{}, m, n, x, pre_r[i % PM][jp % PP], PRE && p.res && n + 8 <= ncol_n);
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
                finish(IC<4>{}, m, n, x, none, false);
            }
        }
    }
}

// KPB (round 5, VERDICT next #1b "BK = 64"): K-tiles per ring slot and per s_barrier — a slot holds KPB consecutive 32-k sub-tiles (each in the
// unchanged [4 hi | 4 lo] row image), one rendezvous and one counted wait serve KPB x (3 FM FN) MFMAs per wave; the MFMA order per
// accumulator is that of KPB = 1, so the result is the same bits.  Plain K-loop only (not PIPE / DILV).
template <int BM, int BN, int WM, int WN, int NS, int NLW, bool PIPE, bool PRE, bool DILV = false, bool TRACE = false, int KPB = 1>
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
    int is_tap = 0, is_c0 = kt0 * BK, is_slot = 0, is_sub = 0;
    unsigned soff_a = (unsigned)(kt0 * BK * ES), soff_w = (unsigned)(kt0 * BK * ES);
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
                    else h2_load8((const h2_t*)p.res + (long)m * p.ldr + n, pre_r[i][jp]);
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

    h2_tile_epilogue<FM, FN, false, PRE, PM, PP>(p, acc, m0 + wm * WTM, n0 + wn * WTN, fr, fg, vt_tile, pre_r, split);
    if (vt_tile) { __syncthreads(); return; }
    tr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr();
    if constexpr (TRACE) { if (p.trace && blockIdx.x == 0 && lane == 0) p.trace[wave * 512] = (unsigned long long)tr_n; }
    __syncthreads();
}

}  // namespace emage_dev

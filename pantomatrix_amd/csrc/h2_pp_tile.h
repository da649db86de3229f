// ANTIPHASE ("ping-pong") tile routine of the EMAGE_H2 contraction (round 6, VERDICT next #1b): ONE 8-wave block per CU whose two wave
// groups (4 waves each, one per SIMD) share ONE A / W panel in LDS and run the K-loop half a step apart:
//
//      interval      0      1      2      3      4    ...
//      group 0      L(0)   C(0)   L(1)   C(1)   L(2)          L(s): ds_read_b128 of K-tile s into registers + this group's share of the
//      group 1       -     L(0)   C(0)   L(1)   C(1)                operand DMA of K-tile s + NS - 1;   C(s): the 3 FM FN MFMAs of K-tile s
//
// with ONE raw s_barrier between consecutive intervals (group 1 enters through one extra barrier).  While a group issues MFMAs, its SIMD
// partner of the other group issues the LDS reads and the LDS-DMA: the two kinds of work that serialise inside the one-barrier-domain
// K-loop of h2_tile.h (DMA issue -> ds_read -> MFMA, profiles/r03_gemm_h2_phase_trace.txt) now overlap INSIDE the block, and a
// 128 x 96 block moves (128 + 96) x 128 B per K-tile for 12 288 outputs where three 64 x 64 blocks move 3 x 128 x 128 B.
//
// Ordering (every rule is "wait, then a barrier the reader has passed"; MI355X_MICROARCH.md, two waves per SIMD, item 7):
//   RAW  K-tile s is read by group 0 in interval 2s, by group 1 in interval 2s + 1.  BOTH groups retire their share of K-tile s with a
//        counted s_waitcnt vmcnt before the barrier that OPENS interval 2s: group 0 at the end of C(s - 1), group 1 at the end of
//        L(s - 1); K-tile 0 before the prologue barrier.  At that point a wave has issued through K-tile s + NS - 2, so NS - 2 stages
//        of its pieces stay in flight across the barrier (never vmcnt(0) in the loop).
//   WAR  K-tile s + NS - 1 lands in the slot of K-tile s - 1, whose last reader is group 1 in interval 2s - 1; the reads are retired
//        (lgkmcnt(0)) before the barrier that ends the reading interval, and the earliest restage is group 0's L(s) in interval 2s.
// The MFMA order per accumulator (three sweeps lo x hi, hi x lo, hi x hi per K-tile, K ascending) is h2_tile.h's: a result is the same
// bits as the 64 x 64 configuration's.  Group 0's waves load the A rows, group 1's the W rows (GA / GB pieces of 1 KiB per wave and K-tile).
#pragma once
#include "h2_tile.h"

namespace emage_dev {

template <int BM, int BN, int NS> constexpr int h2_pp_smem_bytes() { return NS * (BM + BN) * 128; }

// MODE: where a wave issues its DMA pieces — 0: in its L phase behind the fragment reads; 1: in its L phase in front of them; 2: in its C
// phase, spread between the MFMAs (no ds_read of the wave is outstanding there: an LDS-DMA instruction behind a burst of ds_read_b128
// stalls its wave ~180 clocks, profiles/r06_gemm_h2_pp_phase_trace_v1.txt)
template <int BM, int BN, int WM, int WN, int NS, bool PRIO, bool TRACE = false, int MODE = 0>
__device__ __forceinline__ void gemm_h2_pp_tile(const GemmArgs& p, const int m0, const int n0, unsigned char* smem) {
    constexpr int ES = 4, BK = 32, RB = 128, RPI = 8;
    static_assert(WM * WN == 8 && WM % 2 == 0, "two wave groups of four waves, split along M");
    constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
    constexpr bool LONE = (FN & 1) != 0;
    constexpr int GA = BM / RPI / 4, GB = BN / RPI / 4;      // DMA pieces per wave and K-tile: group 0 loads A, group 1 loads W
    static_assert((BM / RPI) % 4 == 0 && (BN / RPI) % 4 == 0, "every wave of a group issues the same number of DMA instructions");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && NS >= 3 && NS <= 6 && (MODE != 2 || NS >= 4), "tile shape / ring depth");
    static_assert((NS - 2) * (GA > GB ? GA : GB) <= 63, "vmcnt is a 6-bit counter");
    constexpr int STAGE = (BM + BN) * RB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;                // the hardware spreads waves 0-3 and 4-7 over the four SIMDs each
    const int wm = grp * (WM / 2) + w4 / WN, wn = w4 % WN;
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

    // ---- operand DMA: lane-fixed byte offsets, K advance in the scalar offset (h2_tile.h) ----
    const int nbatch = p.M / p.Lout;
    const unsigned a_shift = (unsigned)p.pad * (unsigned)(p.lda * ES);
    const unsigned a_bytes = (unsigned)((((long)nbatch * p.Lin - 1) * p.lda + p.Cp) * ES) + a_shift;
    const unsigned w_bytes = (unsigned)((long)p.N * p.K * ES);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A - a_shift), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);
    const int lrow = lane >> 3, lslot = lane & 7;
    const bool is_conv = p.taps > 1;
    unsigned a_voff[GA]; int a_lpos[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int row = (w4 + 4 * j) * RPI + lrow;
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
        const int row = (w4 + 4 * j) * RPI + lrow;
        const int n = n0 + row;
        const int lc = lslot ^ swzW<8>(row);
        b_voff[j] = n < p.N ? (unsigned)n * (unsigned)(p.K * ES) + (unsigned)((2 * (lc & 3) + (lc >> 2)) * 16) : OOB;
    }
    int is_tap = 0, is_c0 = 0, is_slot = 0;
    unsigned soff_a = 0, soff_w = 0;
    const unsigned tap_step = (unsigned)(p.lda - p.Cp + BK) * ES;
    constexpr int GMAX = GA > GB ? GA : GB;
    auto issue_piece = [&](auto jc) {                  // piece J of this wave's share of the next K-tile: an A row block (group 0) or a W row block (group 1)
        constexpr int J = decltype(jc)::value;
        if (EMAGE_DBG(p, 1)) return;                   // tools, timing only: no operand DMA
        unsigned char* base = smem + is_slot * STAGE;
        if (grp == 0) {
            if constexpr (J < GA) {
                unsigned vo = a_voff[J];
                if (is_conv) vo = (unsigned)(a_lpos[J] + is_tap) < (unsigned)p.Lin ? vo : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + (w4 + 4 * J) * 1024),
                                                         16, (int)vo, (int)soff_a, 0, 0);
            }
        } else {
            if constexpr (J < GB)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + BM * RB + (w4 + 4 * J) * 1024),
                                                         16, (int)b_voff[J], (int)soff_w, 0, 0);
        }
    };
    auto issue_advance = [&]() {
        soff_w += BK * ES;
        is_c0 += BK;
        if (is_c0 == p.Cp) { is_c0 = 0; ++is_tap; soff_a += tap_step; } else { soff_a += BK * ES; }
        if (++is_slot == NS) is_slot = 0;
    };
    auto issue = [&]() {
        static_for<GMAX>([&](auto jc) { issue_piece(jc); });
        issue_advance();
    };

    const int fr = lane & 15, fg = lane >> 4;
    const bool vt_tile = p.out_t != nullptr && n0 >= p.t_col0;       // block-uniform
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    // K-tile s has landed (this wave's share): the wave has issued through K-tile min(s + NS - 2, nk - 1); those behind s stay in flight
    // `behind`: K-tiles this wave's issue runs behind the L-phase schedule at the point of the wait (MODE 2: group 1 waits in L(st), one C phase
    // before it issues K-tile st + NS - 1)
    auto wait_stage = [&](const int s, const int behind = 0) {
        const int last = (s + NS - 2 - behind < nk - 1) ? s + NS - 2 - behind : nk - 1;
        const int infl = last - s;                      // 0 .. NS - 2
        static_for<NS - 1>([&](auto kc) {
            constexpr int KQ = decltype(kc)::value;
            if (infl == KQ) { if (grp == 0) wait_vmcnt<KQ * GA>(); else wait_vmcnt<KQ * GB>(); }
        });
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue();

    // ---- fragment read addresses (h2_tile.h) ----
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int arow = wm * WTM + fr;
    const int brow = wn * WTN + 8 * (fr >> 2) + (fr & 3);
    const int browL = wn * WTN + (FN - 1) * 16 + fr;
    const unsigned a_rd = lds0 + arow * RB + ((fg ^ swz<8>(arow)) << 4);
    const unsigned b_rd = lds0 + BM * RB + brow * RB + ((fg ^ swzW<8>(brow)) << 4);
    const unsigned b_rdL = lds0 + BM * RB + browL * RB + ((fg ^ swzW<8>(browL)) << 4);
    struct Frag { u32x4 ah[FM], al[FM], wh[FN], wl[FN]; };
    constexpr int NR = 2 * FM + 2 * FN;
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

    wait_stage(0);
    tr();
    __builtin_amdgcn_s_barrier();                       // prologue: K-tile 0 is in LDS for everybody
    if (grp == 1) __builtin_amdgcn_s_barrier();         // group 1 enters half a step late
    tr();

    auto kloop = [&](auto vtc) __attribute__((always_inline)) {
        constexpr bool VT = decltype(vtc)::value != 0;
        auto mma_one = [&](auto qc, const Frag& f) {
            constexpr int Q = decltype(qc)::value;
            constexpr int t = Q / (FM * FN), i = (Q % (FM * FN)) / FN, j = Q % FN;
            const f16x8 w = __builtin_bit_cast(f16x8, t == 0 ? f.wl[j] : f.wh[j]);
            const f16x8 a = __builtin_bit_cast(f16x8, t == 1 ? f.al[i] : f.ah[i]);
            if constexpr (VT) acc[i][j] = mma_f16(a, w, acc[i][j]);
            else acc[i][j] = mma_f16(w, a, acc[i][j]);
        };
        constexpr int NM = 3 * FM * FN;
        unsigned sb = 0;
        for (int st = 0; st < nk; ++st) {
            const bool more = st + 1 < nk;
            // ---- L(st): fragments of K-tile st, this group's share of K-tile st + NS - 1 ----
            Frag f;
            const bool do_issue = st + NS - 1 < nk;
            if constexpr (MODE == 1) { if (do_issue) issue(); __builtin_amdgcn_sched_barrier(0); tr(); }
            if (!EMAGE_DBG(p, 2)) static_for<NR>([&](auto rc) { read_one(rc, f, sb); });       // (tools, timing only: bit 2 = no fragment reads)
            else static_for<NR>([&](auto rc) { constexpr int R = decltype(rc)::value; if constexpr (R < FM) f.ah[R] = u32x4{0u, 0u, 0u, 0u}; else if constexpr (R < FM + FN) f.wl[R - FM] = u32x4{0u, 0u, 0u, 0u}; else if constexpr (R < FM + 2 * FN) f.wh[R - FM - FN] = u32x4{0u, 0u, 0u, 0u}; else f.al[R - FM - 2 * FN] = u32x4{0u, 0u, 0u, 0u}; });
            __builtin_amdgcn_sched_barrier(0);
            tr();
            if constexpr (MODE == 0) { if (do_issue) issue(); tr(); }
            wait_lgkmcnt<0>();                          // the reads are retired before the barrier: the slot may be restaged behind it
            if (grp == 1 && more) wait_stage(st + 1, MODE == 2 ? 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
            tr();
            if (grp == 0 || more) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            tr();
            // ---- C(st) ----
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            if constexpr (MODE == 2) {
                constexpr int DSTEP = NM / (GMAX + 1);
                static_for<NM>([&](auto qc) {
                    constexpr int Q = decltype(qc)::value;
                    mma_one(qc, f);
                    if constexpr (Q % DSTEP == DSTEP - 1 && Q / DSTEP < GMAX) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (do_issue) issue_piece(IC<Q / DSTEP>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if (do_issue) issue_advance();
            } else {
                if (!EMAGE_DBG(p, 4)) static_for<NM>([&](auto qc) { mma_one(qc, f); });       // (tools, timing only: bit 4 = no MFMAs)
                else asm volatile("" :: "v"(f.ah[0]), "v"(f.wl[0]), "v"(f.wh[0]), "v"(f.al[0]));
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            tr();
            if (grp == 0 && more) wait_stage(st + 1);
            if (more) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            tr();
            sb += STAGE;
            if (sb == NS * STAGE) sb = 0;
        }
    };
    if (vt_tile) kloop(IC<1>{}); else kloop(IC<0>{});
    tr();

    const float pre_r[1][1][8] = {{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}}};
    h2_tile_epilogue<FM, FN, false, false, 1, 1>(p, acc, m0 + wm * WTM, n0 + wn * WTN, fr, fg, vt_tile, pre_r, 0);
    tr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr();
    if constexpr (TRACE) { if (p.trace && blockIdx.x == 0 && lane == 0) p.trace[wave * 512] = (unsigned long long)tr_n; }
}

}  // namespace emage_dev

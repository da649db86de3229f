// EMAGE_H2 contraction with the WEIGHT operand taken out of LDS (round 4; DESIGN.md 4.1 "W from global").
//
// What round 3's phase trace said about gemm_h2_tile (profiles/r03_gemm_h2_phase_trace.txt): per 32-k K-tile of a 64 x 192 tile the LDS port
// moves 112 KB of fragment reads — 96 KB of them W fragments that every one of the 4 M-waves re-reads — plus 32 KB of DMA writes, behind ONE
// barrier that keeps DMA issue, fragment reads and MFMAs of all waves in the same phase: 1100-1300 cycles against 576 cycles of MFMA issue.
// Here a block is 4 waves side by side along N (1 x NW grid): wave w owns ALL BM = 64 rows and its own BN / NW columns, so
//   * no two waves need the same W rows: a wave loads its W fragments global -> VGPR directly, in MFMA operand layout, from a weight image
//     packed in FRAGMENT ORDER (`ops.split_f16_weights_h2w`: per 16-row block and K-tile 2 KB = [hi plane | lo plane], a plane = 64 lanes x
//     16 bytes) — every load instruction is one fully coalesced 1 KB run, prefetched two K-tiles ahead into a 3-deep register ring;
//   * only the 64-row A panel (8 KB per K-tile) goes through LDS — by way of REGISTERS (global -> VGPR three K-tiles ahead, ds_write one
//     iteration before its fragments are read), not by LDS-DMA, so that every operation in a wave's vector-memory queue is of one kind
//     (a VGPR-returning buffer load) and `s_waitcnt vmcnt(N)` can be counted by hand without assuming an order between kinds;
//     the panel's fragments for K-tile k+1 are read while the MFMAs of K-tile k run (two register sets), and the per-K-tile barrier guards
//     8 KB instead of 32;
//   * the memory operations of an iteration (2 A loads, 2 FN W loads, 2 FM fragment reads) are spread between its 3 FM FN MFMAs.
// Arithmetic, K order and per-accumulator MFMA order are those of gemm_h2_tile (three sweeps: W lo x A hi, W hi x A lo, W hi x A hi), so the
// result is BIT-IDENTICAL to the LDS-staged kernels (tests/test_kernels_gpu.py).  W fragments use the natural row order, so a lane ends the
// K-loop with 4 consecutive output columns per fragment; the epilogue swaps accumulator halves between neighbouring lane rows
// (v_permlane16_swap: h2_tile_epilogue<NAT = true>) and stores 8 consecutive columns per lane like the LDS-staged kernels (the 4-column
// form was measured 2.5x slower: 29 k instead of 11.6 k cycles of epilogue — store-issue bound).
//
// vmcnt bookkeeping: every iteration issues exactly G = GA + 2 FN buffer loads in a fixed order (A of K-tile st + 3, W of K-tile st + 2);
// K-tiles past the end are issued with out-of-range offsets (the buffer unit returns zeros: no branch, the counts stay static).  At the top
// of iteration st everything issued up to iteration st - 2 must have landed (A of K-tile st + 1, W of K-tile st): `s_waitcnt vmcnt(G)`
// leaves exactly iteration st - 1's G operations in flight.
#pragma once
#include "h2_tile.h"

namespace emage_dev {

template <int BM, int NS = 2> constexpr int h2w_smem_bytes() { return NS * BM * 128; }

// 16-byte buffer load into VGPRs the compiler does not track (its own vmcnt insertion would drain the LDS-DMA queue at the first use:
// cdna_hip_programming.md 5.7); callers count the queue by hand and pin the consumers behind `s_waitcnt` + sched_barrier
template <int IMM>
__device__ __forceinline__ void buf_load128(u32x4& dst, const u32x4& rsrc, const unsigned voff, const unsigned soff) {
    // "+v": the destination is tied to the variable's current register, so a set keeps ONE physical home across the loop (a copy of a
    // register with a load in flight would read stale data)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff), "i"(IMM) : "memory");
}

template <int BM, int BN, int NW, bool ILV, bool TRACE = false>
__device__ __forceinline__ void gemm_h2w_tile(const GemmArgs& p, const int m0, const int n0, unsigned char* smem) {
    constexpr int ES = 4, BK = 32, RB = 128, RPI = 8, NS = 2;
    constexpr int WTN = BN / NW, FM = BM / 16, FN = WTN / 16;
    constexpr int GA = BM / RPI / NW, GW = 2 * FN, G = GA + GW;
    static_assert((BM / RPI) % NW == 0 && WTN % 16 == 0 && BM % 16 == 0, "tile shape");
    constexpr int STAGE = BM * RB;
    constexpr int WBLK = BK * 16 * ES;                // bytes of one 16-row block x one K-tile in the fragment-order image: [hi 1 KB | lo 1 KB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    int tr_n = 0;
    auto tr = [&]() {
        if constexpr (TRACE) {
            if (p.trace && blockIdx.x == 0 && tr_n < 511) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) p.trace[wn * 512 + 1 + tr_n] = t;
                ++tr_n;
            }
        }
    };
    tr();

    // ---- A panel: LDS-DMA, addressing of gemm_h2_tile (lane-fixed byte offsets, K advance in the scalar offset, conv taps) ----
    const int nbatch = p.M / p.Lout;
    const unsigned a_shift = (unsigned)p.pad * (unsigned)(p.lda * ES);
    const unsigned a_bytes = (unsigned)((((long)nbatch * p.Lin - 1) * p.lda + p.Cp) * ES) + a_shift;
    u32x4 a_rsrc;
    {
        const unsigned long long aa = (unsigned long long)(uintptr_t)((const char*)p.A - a_shift);
        a_rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)aa);
        a_rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(aa >> 32) & 0xffffu);
        a_rsrc[2] = __builtin_amdgcn_readfirstlane(a_bytes);
        a_rsrc[3] = 0x00020000u;
    }
    const int lrow = lane >> 3, lslot = lane & 7;
    const bool is_conv = p.taps > 1;
    unsigned a_voff[GA]; int a_lpos[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int row = (wn + NW * j) * RPI + lrow;
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
    // ---- W fragments: fragment-order image, block (n / 16) at block * nk * WBLK, K-tile kt at + kt * WBLK, lane l at + 16 l (hi), + 1024 + 16 l (lo) ----
    const int nk = p.K / BK;
    const int nblk = (p.N + 15) >> 4;
    const unsigned w_bytes = (unsigned)((long)nblk * nk * WBLK);
    u32x4 w_rsrc;
    {
        const unsigned long long wa = (unsigned long long)(uintptr_t)p.W;
        w_rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
        w_rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
        w_rsrc[2] = __builtin_amdgcn_readfirstlane(w_bytes);
        w_rsrc[3] = 0x00020000u;
    }
    unsigned w_voff[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int blk = ((n0 + wn * WTN) >> 4) + j;
        w_voff[j] = blk < nblk ? (unsigned)blk * (unsigned)(nk * WBLK) + (unsigned)(lane * 16) : OOB;
    }

    // issue state: K-tile of the next A stage / of the next W set (both walk 0, 1, 2, ... ; past nk - 1 the offsets go out of range)
    int a_kt = 0, a_tap = 0, a_c0 = 0;
    unsigned soff_a = 0;
    const unsigned tap_step = (unsigned)(p.lda - p.Cp + BK) * ES;
    int w_kt = 0;
    unsigned soff_w = 0;
    struct APanel { u32x4 r[GA]; };                          // a wave's share of one K-tile of the A panel (8 rows x 128 B per piece) on its way to LDS
    auto issue_a_piece = [&](auto jc, APanel& ap) {
        constexpr int J = decltype(jc)::value;
        unsigned vo = a_voff[J];
        if (is_conv) vo = (unsigned)(a_lpos[J] + a_tap) < (unsigned)p.Lin ? vo : OOB;
        vo = a_kt < nk ? vo : OOB;
        buf_load128<0>(ap.r[J], a_rsrc, vo, soff_a);
    };
    auto advance_a = [&]() {
        ++a_kt;
        a_c0 += BK;
        if (a_c0 == p.Cp) { a_c0 = 0; ++a_tap; soff_a += tap_step; } else { soff_a += BK * ES; }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // the LDS image of a stage is the LDS-DMA kernels' (lane-linear per 1-KiB piece; the XOR swizzle sits in the SOURCE chunk a lane loads)
    auto write_a = [&](const APanel& ap, const int slot) {
        static_for<GA>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const unsigned addr = lds0 + (unsigned)(slot * STAGE + (wn + NW * J) * 1024 + lane * 16);
            const u32x4 v = ap.r[J];
            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
        });
    };
    struct WSet { u32x4 h[FN], l[FN]; };
    auto issue_w_piece = [&](auto jc, WSet& ws) {          // piece 2 j: hi plane of fragment j, 2 j + 1: lo plane
        constexpr int J = decltype(jc)::value;
        const unsigned vo = w_kt < nk ? w_voff[J >> 1] : OOB;
        if constexpr ((J & 1) == 0) buf_load128<0>(ws.h[J >> 1], w_rsrc, vo, soff_w);
        else buf_load128<1024>(ws.l[J >> 1], w_rsrc, vo, soff_w);
    };
    auto advance_w = [&]() { ++w_kt; soff_w += WBLK; };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool vt_tile = p.out_t != nullptr && n0 >= p.t_col0;       // block-uniform
    WSet w0 = {}, w1 = {}, w2 = {};
    APanel p0 = {}, p1 = {};
    struct ASet { u32x4 h[FM], l[FM]; };
    ASet a0, a1;

    // ---- prologue.  Queue order: A 0, A 1, W 0 | A 2, W 1 — the same (A of K-tile k + 3, W of K-tile k + 2) groups the iterations issue ----
    static_for<GA>([&](auto jc) { issue_a_piece(jc, p0); });
    advance_a();
    static_for<GA>([&](auto jc) { issue_a_piece(jc, p1); });
    advance_a();
    static_for<GW>([&](auto jc) { issue_w_piece(jc, w0); });
    advance_w();
    wait_vmcnt<GA + GW>();                 // A 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    write_a(p0, 0);
    static_for<GA>([&](auto jc) { issue_a_piece(jc, p0); });       // A 2 into the registers A 0 has just left
    advance_a();
    static_for<GW>([&](auto jc) { issue_w_piece(jc, w1); });
    advance_w();

    const unsigned a_rd = lds0 + fr * RB + ((fg ^ swz<8>(fr)) << 4);       // row fr + 16 i: swz is invariant under row += 16
    auto read_a = [&](auto rc, ASet& as, const unsigned sb) {              // reads 0 .. FM-1: hi planes, FM .. 2 FM-1: lo planes
        constexpr int R = decltype(rc)::value;
        if constexpr (R < FM) as.h[R] = lds_read128_off<R * 16 * RB>(a_rd + sb);
        else as.l[R - FM] = lds_read128_off<(R - FM) * 16 * RB>((a_rd ^ 64u) + sb);
    };
    wait_lgkmcnt<0>();
    __builtin_amdgcn_s_barrier();
    static_for<2 * FM>([&](auto rc) { read_a(rc, a0, 0u); });
    wait_lgkmcnt<0>();
    // Everything issued so far must have LANDED before the loop is entered: the compiler gives the register sets their loop homes here
    // (v_mov copies in the pre-header), and a copy of a register with a load still in flight carries stale data — the bug of the first
    // two versions of this kernel: right on a half-empty chip, wrong as soon as every CU had a block.  Inside the loop the sets keep one
    // home ("+v" ties; checked in the ISA: no instruction but MFMA / ds_write reads a load destination).  Cost: the prologue's latency, once
    wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    tr();

    constexpr int NM = 3 * FM * FN;
    constexpr int NMEM = G + 2 * FM;       // memory operations of an iteration: G issues + the next K-tile's A fragment reads
    constexpr int MSTEP = NM / NMEM > 0 ? NM / NMEM : 1;
    auto kloop = [&](auto vtc) __attribute__((always_inline)) {
        constexpr bool VT = decltype(vtc)::value != 0;
        // one iteration: K-tile st + 1 of the A panel (registers pnew, landed) -> LDS; MFMAs of K-tile st on (acur, wcur); A fragments of K-tile
        // st + 1 -> anxt; issues the loads of A st + 3 -> pnew (free again) and W st + 2 -> wnew
        auto step = [&](const int st, const ASet& acur, ASet& anxt, const WSet& wcur, WSet& wnew, APanel& pnew) {
            tr();
            wait_vmcnt<G>();
            __builtin_amdgcn_sched_barrier(0);
            write_a(pnew, (st + 1) & 1);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            tr();
            const unsigned sb = (unsigned)(((st + 1) & 1) * STAGE);
            auto mem_op = [&](auto oc) {               // memory operation O of the iteration
                constexpr int O = decltype(oc)::value;
                if constexpr (O < GA) issue_a_piece(IC<O>{}, pnew);
                else if constexpr (O < G) issue_w_piece(IC<O - GA>{}, wnew);
                else read_a(IC<O - G>{}, anxt, sb);          // past the last K-tile: a ghost stage (zeros), never used
            };
            if constexpr (!ILV) {
                static_for<NMEM>([&](auto oc) { mem_op(oc); });
                __builtin_amdgcn_sched_barrier(0);
            }
            static_for<NM>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                constexpr int t = Q / (FM * FN), i = (Q % (FM * FN)) / FN, j = Q % FN;
                const f16x8 w = __builtin_bit_cast(f16x8, t == 0 ? wcur.l[j] : wcur.h[j]);
                const f16x8 a = __builtin_bit_cast(f16x8, t == 1 ? acur.l[i] : acur.h[i]);
                if constexpr (VT) acc[i][j] = mma_f16(a, w, acc[i][j]);
                else acc[i][j] = mma_f16(w, a, acc[i][j]);
                if constexpr (ILV && Q % MSTEP == MSTEP - 1 && Q / MSTEP < NMEM) {
                    __builtin_amdgcn_sched_barrier(0);
                    mem_op(IC<Q / MSTEP>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr (ILV) {
                static_for<NMEM>([&](auto oc) {        // operations the interleave did not reach
                    if constexpr (decltype(oc)::value >= NM / MSTEP) mem_op(oc);
                });
            }
            advance_a();
            advance_w();
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            tr();
        };
        // register sets rotate with periods 2 (A) and 3 (W): the loop body covers 6 K-tiles, branch-free — when nk is not a multiple of 6 the
        // trailing iterations multiply ghost operands (zeros: the accumulators do not move); the host packs a weight for this kernel
        // only when its K-tile count is a multiple of 6 (768, 1536, 2304 ...)
        for (int st = 0; st < nk; st += 6) {
            step(st, a0, a1, w0, w2, p1);
            step(st + 1, a1, a0, w1, w0, p0);
            step(st + 2, a0, a1, w2, w1, p1);
            step(st + 3, a1, a0, w0, w2, p0);
            step(st + 4, a0, a1, w1, w0, p1);
            step(st + 5, a1, a0, w2, w1, p0);
        }
    };
    if (vt_tile) kloop(IC<1>{}); else kloop(IC<0>{});
    wait_vmcnt<0>();                       // the ghost issues of the last iterations
    __builtin_amdgcn_sched_barrier(0);
    tr();

    const float none[1][1][8] = {};
    h2_tile_epilogue<FM, FN, true, false, 1, 1>(p, acc, m0, n0 + wn * WTN, fr, fg, vt_tile, none);
    tr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr();
    if constexpr (TRACE) { if (p.trace && blockIdx.x == 0 && lane == 0) p.trace[wn * 512] = (unsigned long long)tr_n; }
    __syncthreads();
}

}  // namespace emage_dev

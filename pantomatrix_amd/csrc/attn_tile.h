// Device-side body of emage_attention (attention.hip): one query tile of one (batch, head) per wave64, and the staging of a
// workgroup's K / V^T in LDS for the split-f16 form.
#pragma once
#include "common.h"
#include "h2.h"
#include <math.h>

namespace emage_dev {

struct AttnArgs {
    const void* q; const void* k; const void* vt; void* out;
    int ldq, ldk, ldvt, vt_rows, ldo, B, H, Tq, Tk;
    float scale;
    const float* pmask;       // optional (B, H, Tq, Tk) fp32 factor on the probabilities: the attention dropout of a training forward
    float h2s, h2i;           // split-f16 forms: the power of two Q / K / V are scaled by before their fp16 split — and the scale of the EMAGE_H2 output image —
                              // and its inverse (csrc/h2.h: 16 unless the dtype code carries the model's activation shift)
};

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint4 bload128(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 bload64(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_uint2(v.x, v.y);
}
// finite inputs only (probabilities, attention outputs): plain round-to-nearest-even, no NaN path
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    unsigned a = __builtin_bit_cast(unsigned, lo), b = __builtin_bit_cast(unsigned, hi);
    a += 0x7fffu + ((a >> 16) & 1u);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xffff0000u);
}

// ---- split-f16 form (EMAGE_F16X3): fp32 operands, every 16x16x32 product as three fp16 MFMAs (hi*hi + hi*lo + lo*hi,
// fp32 accumulate) instead of eight v_mfma_f32_16x16x4_f32 — fp32-grade scores and outputs at bf16-like kernel time.
// Two consecutive fp32 chunks of a lane (8 values) form one MFMA operand; x*s = hi + lo with both planes fp16.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
struct SplitF16 { h16x8 hi, lo; };
__device__ __forceinline__ SplitF16 split_chunks(const uint4& c0, const uint4& c1, const float s) {
    const float x[8] = {__builtin_bit_cast(float, c0.x), __builtin_bit_cast(float, c0.y), __builtin_bit_cast(float, c0.z), __builtin_bit_cast(float, c0.w),
                        __builtin_bit_cast(float, c1.x), __builtin_bit_cast(float, c1.y), __builtin_bit_cast(float, c1.z), __builtin_bit_cast(float, c1.w)};
    SplitF16 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xs = x[e] * s;
        const _Float16 h = (_Float16)xs;
        r.hi[e] = h;
        r.lo[e] = (_Float16)(xs - (float)h);
    }
    return r;
}
__device__ __forceinline__ f32x4 mma_split(const SplitF16& a, const SplitF16& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, acc, 0, 0, 0);
}
// Q, K, V planes: scaled by AttnArgs::h2s before the split — 16 (|x| < 4094 stays finite in fp16) unless the dtype code carries the model's
// activation shift (include/emage_hip.h EMAGE_H2_SHIFT: 2^(4 - k)); powers of two, undone exactly behind the MFMAs
constexpr float ATT_P_SCALE = 1024.0f;     // probabilities are <= 1

constexpr int QW = 4;    // query tiles per workgroup: the waves of one (batch, head) share K / V^T through the CU's L1
constexpr int DS = 1;    // waves per query tile: with DS = 2 each wave recomputes the 16x64 scores and owns half of the d tiles
                         // of P V (two shorter waves per SIMD).  Measured: 16.0 us vs 13.3 us with DS = 1 — not a win.

// softmax(Q K^T * scale) V for query tile qt (16 queries) of (batch b, head h); one wave64 (wave `dpart` of the DSP that share that tile: each recomputes the scores and owns NDT / DSP of the output d tiles).
// H2OUT (split-f16 form only): the output is written as an EMAGE_H2 image (csrc/h2.h) — V^T rows are fetched in a permuted order
// so that a lane ends with 8 consecutive d of its query from each PAIR of d tiles (one 32-byte group).
// ---- K / V^T of one (batch, head) staged ONCE per workgroup in LDS as split fp16 planes (split-f16 form, Tk <= 64) ---------------
// Without it each of the 4 query-tile waves of a (batch, head) fetched all of K and V^T itself (4 x 96 KB through the CU's
// texture path) and split every chunk on its own VALU.  Staged layout = the MFMA operand order: one 32-byte cell [8 fp16 hi | 8 fp16
// lo] per (row, 32-wide contraction step, lane group fg) holding the 8 values lane (row, fg) feeds one MFMA with — columns
// {32 s + 4 fg + r} U {32 s + 16 + 4 fg + r} — so a fragment is two ds_read_b128 and the arithmetic (operands, order of the three
// MFMAs, order over the contraction) is EXACTLY that of the register path: results are bit-identical.  Rows are padded by 16 bytes:
// the 16 rows a quarter-wave reads land in 16 different 16-byte bank groups.
template <int HD, int NT> struct AttnLds {
    static constexpr int KS = HD / 32;                    // contraction steps of Q K^T
    static constexpr int KROW = KS * 4 * 32 + 16;         // bytes per staged key row
    static constexpr int NKEY = NT * 16;
    static constexpr int VS = NT / 2;                     // contraction steps of P V (32 keys each)
    static constexpr int VROW = VS * 4 * 32 + 16;         // bytes per staged V^T row (one d)
    static constexpr int K_BYTES = NKEY * KROW, V_BYTES = HD * VROW, BYTES = K_BYTES + V_BYTES;
};

// all 256 threads of the workgroup; H2OUT selects the permuted d order of attn_tile's V^T rows.  Every global load of the thread is
// issued before the first split (compile-time trip counts, fully unrolled): one memory latency per workgroup, not one per cell.
template <int HD, int NT, bool H2OUT>
__device__ __forceinline__ void attn_stage_kv(const AttnArgs& p, const int b, const int h, unsigned char* smem) {
    using L = AttnLds<HD, NT>;
    constexpr int NTHR = 64 * QW;
    constexpr int KU = L::NKEY * L::KS * 4 / NTHR, VU = HD * L::VS * 4 / NTHR;      // cells per thread
    static_assert(L::NKEY * L::KS * 4 % NTHR == 0 && HD * L::VS * 4 % NTHR == 0, "cells divide evenly over the workgroup");
    const float* __restrict__ K = (const float*)p.k;
    const float* __restrict__ V = (const float*)p.vt;
    const int tid = threadIdx.x;
    uint4 k0[KU], k1[KU], v0[VU], v1[VU];
#pragma unroll
    for (int i = 0; i < KU; ++i) {
        const int u = i * NTHR + tid;
        const int key = u / (L::KS * 4), cell = u - key * (L::KS * 4), s = cell >> 2, fg = cell & 3;
        const int krow = min(key, p.Tk - 1);
        const float* src = K + ((long)b * p.Tk + krow) * p.ldk + h * HD + 32 * s + 4 * fg;
        k0[i] = *(const uint4*)src;
        k1[i] = *(const uint4*)(src + 16);
    }
#pragma unroll
    for (int i = 0; i < VU; ++i) {
        const int u = i * NTHR + tid;
        const int j = u / (L::VS * 4), cell = u - j * (L::VS * 4), c = cell >> 2, fg = cell & 3;
        const int dt = j >> 4, fr = j & 15;               // staged row j = operand row fr of d tile dt
        const int drow = H2OUT ? 32 * (dt >> 1) + 4 * (dt & 1) + 8 * (fr >> 2) + (fr & 3) : j;
        const float* src = V + ((long)b * p.vt_rows + h * HD + drow) * p.ldvt + 32 * c + 4 * fg;
        v0[i] = *(const uint4*)src;
        v1[i] = *(const uint4*)(src + 16);
    }
    // keep EVERY load above the first split: the splits are pure arithmetic the instruction selector would otherwise hoist between the
    // loads (each with its s_waitcnt: ~6 loads in flight instead of all of them); the empty asm statements pin the loaded registers
    // behind the scheduling barrier
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < KU; ++i) {
        asm volatile("" : "+v"(k0[i].x), "+v"(k0[i].y), "+v"(k0[i].z), "+v"(k0[i].w), "+v"(k1[i].x), "+v"(k1[i].y), "+v"(k1[i].z), "+v"(k1[i].w));
    }
#pragma unroll
    for (int i = 0; i < VU; ++i) {
        asm volatile("" : "+v"(v0[i].x), "+v"(v0[i].y), "+v"(v0[i].z), "+v"(v0[i].w), "+v"(v1[i].x), "+v"(v1[i].y), "+v"(v1[i].z), "+v"(v1[i].w));
    }
#pragma unroll
    for (int i = 0; i < KU; ++i) {
        const int u = i * NTHR + tid;
        const int key = u / (L::KS * 4), cell = u - key * (L::KS * 4);
        const SplitF16 f = split_chunks(k0[i], k1[i], p.h2s);
        unsigned char* dst = smem + key * L::KROW + cell * 32;
        *(h16x8*)dst = f.hi;
        *(h16x8*)(dst + 16) = f.lo;
    }
#pragma unroll
    for (int i = 0; i < VU; ++i) {
        const int u = i * NTHR + tid;
        const int j = u / (L::VS * 4), cell = u - j * (L::VS * 4);
        const SplitF16 f = split_chunks(v0[i], v1[i], p.h2s);
        unsigned char* dst = smem + L::K_BYTES + j * L::VROW + cell * 32;
        *(h16x8*)dst = f.hi;
        *(h16x8*)(dst + 16) = f.lo;
    }
}

template <typename T, int HD, int NT, int DSP = DS, bool X3 = false, bool H2OUT = false, bool LDSKV = false>
__device__ __forceinline__ void attn_tile(const AttnArgs& p, const int b, const int h, const int qt, const int dpart, unsigned char* smem = nullptr) {
    static_assert(!LDSKV || (X3 && DSP == 1 && NT <= 4 && NT % 2 == 0), "staged K / V^T: split-f16 form, Tk <= 64");
    constexpr int EPC = Elem<T>::EPC;
    static_assert(!X3 || (EPC == 4 && NT % 2 == 0), "split-f16 attention: fp32 operands, key tiles pair up");
    constexpr int ES = 16 / EPC;
    constexpr int NSTEP = HD / (4 * EPC);     // contraction steps over head_dim (4 chunks per step)
    constexpr int NDT = HD / 16;              // output d tiles
    constexpr int NPC = (EPC == 8) ? (NT + 1) / 2 : NT;   // P chunks (A/B operand units along the key axis)
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fg = lane >> 4;
    const int q0 = qt * 16;

    // The kernel is a single wave per SIMD with ~50 MFMAs of work: instruction count and exposed latency are what
    // it costs.  Operands therefore come through buffer descriptors (one VGPR offset per row, the walk along the
    // head dimension / key axis / d tiles in scalar or immediate offsets), and ALL of Q, K and V^T are requested
    // before the first MFMA (this wave owns its SIMD's register file).
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)p.q, 0, (unsigned)((long)p.B * p.Tq * p.ldq * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)p.k, 0, (unsigned)((long)p.B * p.Tk * p.ldk * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)p.vt, 0, (unsigned)((long)p.B * p.vt_rows * p.ldvt * ES), 0x00020000);

    const int qrow = min(q0 + fr, p.Tq - 1);
    const int qoff = ((b * p.Tq + qrow) * p.ldq + h * HD + fg * EPC) * ES;
    uint4 qf[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) qf[s] = bload128(rq, qoff, s * 64);
    if constexpr (LDSKV) {
        // the workgroup's K / V^T go to LDS behind this wave's Q loads (already in flight); EVERY wave of the workgroup takes part,
        // also one whose query tile lies past Tq (it leaves behind the barrier)
        attn_stage_kv<HD, NT, H2OUT>(p, b, h, smem);
        __syncthreads();
        if (q0 >= p.Tq) return;
    }

    // Tk <= 64 (every inference window): everything in flight at once — when it fits the 256 registers a wave gets with
    // two waves per SIMD (fp32 operands are twice as wide: only up to Tk <= 32)
    constexpr bool PRE = !LDSKV && NT <= 4 && (DSP == 1 || EPC == 8 || NT <= 2);
    constexpr int KT = PRE ? NT : 1;
    uint4 kf[KT][NSTEP];
    // staged fragments (LDSKV): the cell of (row, step, fg) in the layout of attn_stage_kv
    auto lds_frag = [&](const unsigned char* base, int row_bytes, int row, int step) {
        const unsigned char* c = base + row * row_bytes + (step * 4 + fg) * 32;
        SplitF16 f;
        f.hi = *(const h16x8*)c;
        f.lo = *(const h16x8*)(c + 16);
        return f;
    };
    auto load_k = [&](int nt, uint4 (&dst)[NSTEP]) {
        const int krow = min(nt * 16 + fr, p.Tk - 1);
        const int koff = ((b * p.Tk + krow) * p.ldk + h * HD + fg * EPC) * ES;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) dst[s] = bload128(rk, koff, s * 64);
    };
    if constexpr (PRE) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_k(nt, kf[nt]);
    }
    // V^T chunk c of d-tile dt: keys {32c + 4g + r} U {32c + 16 + 4g + r} (bf16) / {16c + 4g + r} (fp32), r = 0..3
    static_assert(!H2OUT || (X3 && DSP == 1 && (HD / 16) % 2 == 0), "h2 output: split-f16 form, one wave per query tile, d tiles pair up");
    // d row of (d tile dt, operand row fr): natural 16 dt + fr; H2OUT: 32 (dt >> 1) + 8 (fr >> 2) + 4 (dt & 1) + (fr & 3)
    const int vrow0 = H2OUT ? 8 * (fr >> 2) + (fr & 3) : fr;
    const int voff = ((b * p.vt_rows + h * HD + vrow0) * p.ldvt + fg * 4) * ES;
    const int vstep = 16 * p.ldvt * ES;
    auto vsoff = [&](int dt) { return H2OUT ? (32 * (dt >> 1) + 4 * (dt & 1)) * p.ldvt * ES : dt * vstep; };
    static_assert(NDT % DSP == 0, "d tiles split evenly over the DSP waves");
    constexpr int NDW = NDT / DSP;            // d tiles per wave
    constexpr int VT_ = PRE ? NDW : 1;
    uint4 vf[VT_][NPC];
    auto load_v = [&](int dt, uint4 (&dst)[NPC]) {
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            if constexpr (EPC == 8) {
                const uint2 lo = bload64(rv, voff, vsoff(dt) + c * 64);
                const uint2 hi = bload64(rv, voff, vsoff(dt) + c * 64 + 32);
                dst[c] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                dst[c] = bload128(rv, voff, vsoff(dt) + c * 64);
            }
        }
    };
    const int dt0 = dpart * NDW;
    if constexpr (PRE) {
#pragma unroll
        for (int dt = 0; dt < NDW; ++dt) load_v(dt0 + dt, vf[dt]);
        __builtin_amdgcn_sched_barrier(0);
    }

    // scores S^T[key][q] = K Q^T: lane holds keys {16nt + 4g + r} of query column fr
    f32x4 sc[NT];
    if constexpr (X3) {
        SplitF16 qs[NSTEP / 2];                // Q is re-used by every key tile: split once
#pragma unroll
        for (int s = 0; s < NSTEP / 2; ++s) qs[s] = split_chunks(qf[2 * s], qf[2 * s + 1], p.h2s);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LDSKV) {
#pragma unroll
                for (int s = 0; s < NSTEP / 2; ++s) acc = mma_split(lds_frag(smem, AttnLds<HD, NT>::KROW, nt * 16 + fr, s), qs[s], acc);
            } else {
                if constexpr (!PRE) load_k(nt, kf[0]);
#pragma unroll
                for (int s = 0; s < NSTEP / 2; ++s)
                    acc = mma_split(split_chunks(kf[PRE ? nt : 0][2 * s], kf[PRE ? nt : 0][2 * s + 1], p.h2s), qs[s], acc);
            }
            sc[nt] = acc * (p.h2i * p.h2i);
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!PRE) load_k(nt, kf[0]);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) acc = Elem<T>::mma(kf[PRE ? nt : 0][s], qf[s], acc);
            sc[nt] = acc;
        }
    }

    // softmax over keys for query column fr
    float mx = -INFINITY;
    const bool ragged = p.Tk < NT * 16;       // wave-uniform: full windows (Tk = 64) skip the per-key mask
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = sc[nt][r] * p.scale;
            if (ragged && nt * 16 + fg * 4 + r >= p.Tk) v = -INFINITY;
            sc[nt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // bf16 mode rounds P to 8 bits of mantissa right after: the hardware exp2 path is ample there;
            // the fp32 parity mode keeps the correctly-rounded expf
            const float e = (EPC == 8) ? __expf(sc[nt][r] - mx) : expf(sc[nt][r] - mx);
            sc[nt][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;

    // P chunks (contraction = keys), normalised
    uint4 pc[NPC];
    if constexpr (EPC == 8) {
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            const f32x4 lo = sc[2 * c];
            f32x4 hi = {0.f, 0.f, 0.f, 0.f};
            if (2 * c + 1 < NT) hi = sc[2 * c + 1];
            pc[c].x = pack_bf16x2(lo[0] * inv, lo[1] * inv);
            pc[c].y = pack_bf16x2(lo[2] * inv, lo[3] * inv);
            pc[c].z = pack_bf16x2(hi[0] * inv, hi[1] * inv);
            pc[c].w = pack_bf16x2(hi[2] * inv, hi[3] * inv);
        }
    } else {
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            const f32x4 v = sc[c];
            f32x4 pv = {v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
            if (p.pmask) {                    // wave-uniform; train-mode forward only: dropout(softmax(.)) before P V, as torch does
                const float* pm = p.pmask + (((long)b * p.H + h) * p.Tq + min(q0 + fr, p.Tq - 1)) * p.Tk + c * 16 + fg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c * 16 + fg * 4 + r < p.Tk) pv[r] *= pm[r];
            }
            pc[c] = __builtin_bit_cast(uint4, pv);
        }
    }

    // O^T = V^T P^T (operands swapped): lane holds O[q = q0 + fr][d = 16dt + 4g + r], 4 consecutive d -> one
    // 8-byte (bf16) / 16-byte (fp32) store per d tile
    T* __restrict__ O = (T*)p.out;
    const int qq = q0 + fr;
    T* orow = O + ((long)b * p.Tq + qq) * p.ldo + h * HD + fg * 4;
    SplitF16 ps[X3 ? NPC / 2 : 1];             // P is re-used by every d tile: split once (pairs of 4-key chunks = 8 keys)
    if constexpr (X3) {
#pragma unroll
        for (int c = 0; c < NPC / 2; ++c) ps[c] = split_chunks(pc[2 * c], pc[2 * c + 1], ATT_P_SCALE);
    }
    f32x4 held = {0.f, 0.f, 0.f, 0.f};            // H2OUT: the even d tile of a pair
#pragma unroll
    for (int dw = 0; dw < NDW; ++dw) {
        const int dt = dt0 + dw;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!PRE && !LDSKV) load_v(dt, vf[0]);
        if constexpr (LDSKV) {
#pragma unroll
            for (int c = 0; c < NPC / 2; ++c)
                acc = mma_split(lds_frag(smem + AttnLds<HD, NT>::K_BYTES, AttnLds<HD, NT>::VROW, dt * 16 + fr, c), ps[c], acc);
            acc = acc * (p.h2i * (1.0f / ATT_P_SCALE));
        } else if constexpr (X3) {
#pragma unroll
            for (int c = 0; c < NPC / 2; ++c)
                acc = mma_split(split_chunks(vf[PRE ? dw : 0][2 * c], vf[PRE ? dw : 0][2 * c + 1], p.h2s), ps[c], acc);
            acc = acc * (p.h2i * (1.0f / ATT_P_SCALE));
        } else {
#pragma unroll
            for (int c = 0; c < NPC; ++c) acc = Elem<T>::mma(vf[PRE ? dw : 0][c], pc[c], acc);
        }
        if constexpr (H2OUT) {
            if ((dw & 1) == 0) { held = acc; continue; }
            if (qq < p.Tq) {
                const float v8[8] = {held[0], held[1], held[2], held[3], acc[0], acc[1], acc[2], acc[3]};
                h2_store8((h2_t*)p.out + ((long)b * p.Tq + qq) * p.ldo + h * HD + 32 * (dt >> 1) + 8 * fg, v8, p.h2s);
            }
            continue;
        }
        if (qq < p.Tq) {
            if constexpr (EPC == 8) {
                uint2 t;                      // convex combination of finite V rows: no NaN path needed
                t.x = pack_bf16x2(acc[0], acc[1]);
                t.y = pack_bf16x2(acc[2], acc[3]);
                *(uint2*)(orow + dt * 16) = t;
            } else {
                *(float4*)(orow + dt * 16) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
    }
}

}  // namespace emage_dev

// Device-side building blocks of emage_gemm (gemm.hip; lstm.hip's step kernels call the same tile routine with the LSTM-cell
// epilogue; h2_tile.h builds on the ring helpers): GemmArgs, the vector epilogue helpers and the LDS-DMA ring tile routine
// `gemm_pipe_tile`.  The diagnostic branches (p.dbg) exist only in the tools build (-DEMAGE_TOOLS, libemage_hip_tools.so).
#pragma once
#include "common.h"
#include <utility>

#ifdef EMAGE_TOOLS
#define EMAGE_DBG(p, mask) ((p).dbg & (mask))
#else
#define EMAGE_DBG(p, mask) false
#endif

namespace emage_dev {

struct GemmArgs {
    const void* A; const void* W; const float* bias; const float* slope;
    const void* res; void* out; float* out_f32; void* out_t;
    int lda, ldr, ldo, ldf, res_is_f32, res_first, n_store;
    int t_col0, t_rows, t_ld;
    int M, N, K, Cp, taps, stride, pad, Lin, Lout;
    int tiles_m, tiles_n;
    int dbg;
    float a_scale, o_scale;   // split-f16 mode: A is multiplied by a_scale before the hi/lo split, accumulators by o_scale after the K-loop
    float h2s, h2i;           // EMAGE_H2: the scale of the activation IMAGES this launch writes (`out`) and reads as its residual, and its inverse (csrc/h2.h: 16 unless the dtype code carries a shift)
    float* cstate; int ldc;   // EPI_LSTM: the (M, N/4) cell state, updated in place
    int ksplit, nk_split;        // EMAGE_H2 split-K (gemm_h2.hip): > 1 K-slices of nk_split K-tiles each; partial tiles atomically added into out_f32,
    float* ws; long ws_plane; int ldws;   // ... or (ws != NULL, emage_gemm_ws) stored as plane `slice` of the workspace — (M, ldws) fp32 each, ws_plane
                                 // elements apart — and summed in slice order by a second launch (deterministic)
    int tile_order;              // EMAGE_H2 single launches: 0 = XCD-aware runs (each XCD walks a contiguous run of tiles), 1 = dispatch order (gemm_h2.hip)
    // LayerNorm FOLDED into the contractions around it (round 6, EMAGE_H2 only; every pointer NULL = off).  A post-norm layer's
    // y = LN(s) W^T + b is computed on the RAW pre-norm sum s:  y = rstd (s W'^T - mu c) + b'  with W' = W gamma, c[n] = sum_k W'[n][k],
    // b' = W beta + b (packed by the host; b' arrives as `bias`), and the residual LN(s) of the next sub-layer is recomputed from s:
    const float* ln_stats; int ln_np;     // operand A = s: (M, ln_np) partial row statistics {mean, M2} over 32 columns each (ln_np * 32 = the normalised width)
    const float* ln_c;                    // c (N)
    const float* rs_stats; int rs_np;     // `res` = the raw sum s' of a folded LayerNorm: its partial row statistics, and ...
    const float* rs_gamma; const float* rs_beta;      // ... its affine parameters: res'[m][n] = (s'[m][n] - mu) rstd gamma[n] + beta[n]
    float* st_out;                        // (M, N / 32) {mean, M2}: partial row statistics of THIS launch's output rows, one per 32 columns (64 x 64 tiles of 32 x 32 wave tiles only)
    float ln_eps;
    // split-K with an IN-KERNEL fix-up (round 6, EMAGE_H2; few-row launches: ONE clip has M = 64): the tile's K range is cut into `ksplit` slices
    // (blockIdx = slice * tiles + tile, K-tiles [slice * nk_split, ...)); every block stores its accumulators write-through into sk_ws, counts
    // itself on sk_count[tile], and the LAST block to arrive adds the slices in slice order and runs the whole epilogue (csrc/h2_tile.h)
    float* sk_ws; int* sk_count;
    long sk_ws_bytes; int sk_tiles;      // capacities the caller provides (bytes of sk_ws, counters behind sk_count): the dispatch splits only within them
    unsigned long long* trace;   // tools builds: per-wave s_memtime stamps of one block (h2_tile.h TRACE), else NULL
};

// Epilogue kinds of gemm_pipe_tile.  EPI_LSTM (emage_lstm_step): the contraction is h_{t-1} W_hh^T with the gate rows of
// W_hh interleaved per hidden unit (column 4u + g, g = input / forget / cell / output), `res` holds the step's input
// projection x_t W_ih^T + b_ih + b_hh in the same layout, and the epilogue applies the LSTM cell (torch.nn.LSTM:
// c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c')) — c' to cstate[m][u], h' to out_f32[m][u].
constexpr int EPI_LINEAR = 0, EPI_LSTM = 1;
// The cell's non-linearities.  FAST = false (exact-fp32 mode): libm-grade expf / tanhf and IEEE divisions.  FAST = true (split-f16
// mode): hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32, 1 ulp each; saturate correctly at +-inf).  The cell sits on the SERIAL
// path of the persistent recurrence (lstmseq.hip): libm's tanhf + three IEEE divisions per (clip, unit) were ~2.8 us of a 9 us
// time step (the "cell + h store only" ablation, profiles/r03_lstm_layer_breakdown.json).  Absolute error ~1e-7 per gate; the step
// kernels and the persistent kernel share these functions, so they stay bit-identical to each other.
template <bool FAST> __device__ __forceinline__ float lstm_sigmoid(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    else return 1.0f / (1.0f + expf(-x));
}
template <bool FAST> __device__ __forceinline__ float lstm_tanh(float x) {
    if constexpr (FAST) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
    else return tanhf(x);
}

constexpr int NTHREADS = 256;
constexpr int KCH = 8;   // 16-byte chunks per K-tile row

// Row-contiguous store of a staged fp32 tile Cs[BM][CLD]: thread -> (row, 4 consecutive columns); 16-byte
// vector loads/stores when the leading dimensions allow, scalar otherwise.  Columns >= t_col0 (V^T region)
// are stored column-contiguous along the sequence axis instead.
template <typename T> struct Pack4;
template <> struct Pack4<float> {
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
};
template <> struct Pack4<bf16_t> {
    __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        uint2 t;
        t.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        t.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *(uint2*)p = t;
    }
    __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
        const uint2 t = *(const uint2*)p;
        v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
    }
};

// 8 consecutive elements: two 16-byte accesses (fp32) or one (bf16)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *(const uint4*)p;
    v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
    v[4] = __builtin_bit_cast(float, t.z << 16); v[5] = __builtin_bit_cast(float, t.z & 0xffff0000u);
    v[6] = __builtin_bit_cast(float, t.w << 16); v[7] = __builtin_bit_cast(float, t.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    t.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    t.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    t.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    *(uint4*)p = t;
}

template <typename T, int BM, int BN, int CLD, int NT = NTHREADS>
__device__ __forceinline__ void transposed_store(const GemmArgs& p, const float* Cs, int m0, int n0, int tid) {
    T* __restrict__ out_t = (T*)p.out_t;
    const int t_ncols = p.N - p.t_col0;
    constexpr int RG = BM / 4;                          // 4-row groups per column
    const bool tvec = (p.t_rows % 4 == 0) && (p.t_ld % 4 == 0) && (((uintptr_t)out_t & 15) == 0);
    for (int gid = tid; gid < BN * RG; gid += NT) {
        const int col = gid / RG, rg = gid - col * RG;
        const int n = n0 + col, m = m0 + rg * 4;
        if (n < p.t_col0 || n >= p.N || m >= p.M) continue;
        const float bv = p.bias ? p.bias[n] : 0.f, sv = p.slope ? p.slope[n] : 1.f;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = leaky(Cs[(rg * 4 + e) * CLD + col] + bv, sv);
        const int b = m / p.t_rows, l = m - b * p.t_rows;
        T* dst = out_t + ((long)b * t_ncols + (n - p.t_col0)) * p.t_ld + l;
        if (tvec && m + 3 < p.M) {
            Pack4<T>::store(dst, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mm = m + e;
                if (mm < p.M) {
                    const int bb = mm / p.t_rows, ll = mm - bb * p.t_rows;
                    out_t[((long)bb * t_ncols + (n - p.t_col0)) * p.t_ld + ll] = Elem<T>::to(v[e]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pipelined variant: global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), an NS-deep
// ring of K-tiles kept in flight across ONE raw s_barrier per K-tile with counted s_waitcnt vmcnt(N),
// fragment reads as inline-asm ds_read_b128 (hipcc would otherwise drain every in-flight LDS-DMA before a
// compiler-visible LDS read of the same array).  Same LDS image as the kernel above: row-major 128-B rows,
// 16-B slot s of row r holds K-chunk s ^ ((r>>1)&7); the DMA writes LDS linearly in lane order, so the
// swizzle is applied to each lane's SOURCE chunk.  Rows outside M / N / the conv's valid span read a
// 16-byte zero block instead.
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __attribute__((aligned(16))) const unsigned g_zero_chunk[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// ---- split-f16 operands (EMAGE_F16X3): fp32 activations are split in registers into two fp16 planes,
// x*s = hi + lo with |x*s - hi - lo| <= 2^-23 |x*s| (hi = rne(x*s), lo = rne(x*s - hi); the difference is exact in
// fp32), the weights arrive pre-split from the host in the same k order.  Three v_mfma_f32_16x16x32_f16
// (hi*hi + hi*lo + lo*hi, fp32 accumulate) then carry ~22 mantissa bits per product: fp32-grade results at a third
// of the fp16 MFMA rate instead of the 1/16 of v_mfma_f32_16x16x4_f32.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_f16(const u32x4& c0, const u32x4& c1, const float s, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xs = __builtin_bit_cast(float, e < 4 ? c0[e] : c1[e - 4]) * s;
        const _Float16 h = (_Float16)xs;
        hi[e] = h;
        lo[e] = (_Float16)(xs - (float)h);
    }
}
__device__ __forceinline__ f32x4 mma_f16(const f16x8& a, const f16x8& b, const f32x4& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N > 15 ? 15 : N) : "memory"); }   // 4-bit counter

// LDS image per ring slot: rows of KC 16-byte chunks (KC = 8: 128-B rows, two MFMA k-groups per tile;
// KC = 4: 64-B rows, one k-group).  16-B slot of chunk g in row r: g ^ swz(r), chosen so that each 16-lane
// service group of ds_read_b128 hits 16 distinct slots of the 256-B bank row:
//   KC = 8: swz = (r>>1)&7        KC = 4: swz = (-(r>>2))&3        (both invariant under r += 16)
template <int KC> __device__ __forceinline__ int swz(int row) { return KC == 8 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3); }
// The W tile is read with a PERMUTED row order (see the kernel): lane fr of fragment j fetches W row
// (j>>1)*32 + (j&1)*4 + 8*(fr>>2) + (fr&3), so a 16-lane group covers rows {0-3, 8-11, 16-19, 24-27}.  Its slot
// swizzle is chosen for that pattern (distinct (row parity, slot) per ds_read_b128 service group):
//   KC = 8: swzW = ((r>>1)&1) | (((r>>3)&3)<<1)      KC = 4: swzW = (-(r>>3))&3     (both invariant under r += 4, 32)
template <int KC> __device__ __forceinline__ int swzW(int row) {
    return KC == 8 ? (((row >> 1) & 1) | (((row >> 3) & 3) << 1)) : ((-(row >> 3)) & 3);
}

template <int OFF> __device__ __forceinline__ u32x4 lds_read128_off(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// The kernel is instruction-issue bound at these sizes (rocprof: MFMA is <20 % of active wave cycles in a
// naive loop), so the K-loop carries no per-tile vector address math: operands stream through
// buffer_load_dwordx4 ... lds with a per-lane byte offset fixed at kernel start and the K-tile advance in the
// scalar offset; rows outside M / N and conv halo rows get an out-of-range offset, for which the buffer unit
// returns zeros.  MFMA operands are SWAPPED (A-operand = W rows, B-operand = activation rows) so that a lane
// ends up with 4 consecutive output COLUMNS of one row: the epilogue is direct 16-byte (fp32) / 8-byte (bf16)
// vector loads/stores, no LDS staging.  Only V^T tiles (column-contiguous along the sequence) stage through LDS.
// On top of that the W rows are fed to MFMA in a permuted order, so that the two fragments of a pair give a lane
// EIGHT consecutive output columns: 16-byte bf16 stores, 64 contiguous bytes per row per store instruction.
constexpr unsigned OOB = 0x80000000u;    // > any num_records we create (buffers are < 2 GiB)

// blocks per CU the LDS footprint admits (capped at 3): used as the launch-bounds occupancy target so the register
// allocator does not cost a resident block
template <int BM, int BN, int NS, int KC, int KPS = 1>
constexpr int lds_blocks() {
    const int ring = NS * KPS * (BM + BN) * KC * 16;
    const int stage_full = BM * (BN + 4) * 4;
    const int stage = stage_full <= ring ? stage_full : stage_full / 2;
    const int bytes = ring > stage ? ring : stage;
    const int n = 160 * 1024 / bytes;
    return n >= 3 ? 3 : (n >= 2 ? 2 : 1);
}

// LDS bytes a tile of this shape needs: the operand ring, or the fp32 V^T staging tile when that is larger
template <typename T, int BM, int BN, int NS, int KC, int KPS = 1>
constexpr int pipe_smem_bytes() {
    constexpr int STAGE = KPS * (BM + BN) * KC * 16;
    constexpr int CLD = BN + 4;
    constexpr int EP = (BM * CLD * 4 <= NS * STAGE) ? 1 : 2;
    constexpr int CBYTES = (BM / EP) * CLD * 4;
    return NS * STAGE > CBYTES ? NS * STAGE : CBYTES;
}

// One BM x BN output tile at (m0, n0): operand ring, K-loop and epilogue.  Called by every thread of a 256-thread
// (or, with WM * WN = 8, 512-thread) block; `smem` is the block's LDS (pipe_smem_bytes, 128-byte aligned); ends with a __syncthreads() so the caller
// may start the next tile (or any other use of the LDS) right away.
// FPRE: fetch the epilogue operands (bias, slope, residual) ahead of the K-loop whatever the tile size (the fused
// layer kernel has the registers; a lone block per CU cannot hide their latency behind another block).
// X3: split-f16 MFMA on fp32 operands (T = float, KC = 8): the K-tile is 32 k; a lane's two chunks (k = 4g..4g+3 and
// 16+4g..16+4g+3 of the tile) form the 8 k-values of one 16x16x32 MFMA; the W tile row holds [4 hi chunks | 4 lo chunks]
// packed by the host in exactly that k order (pantomatrix_amd.modeling_emage_audio._Packed._split_f16).
// KPS: K-tiles per ring slot (1 or 2).  With 2, a slot holds two consecutive K-tiles ([A | W] [A | W]) and the block meets at
// ONE barrier / counted wait per TWO K-tiles: the waves of a lone block per CU spend a third of their cycles waiting at the
// per-K-tile rendezvous (profiles/r02_pmc_gemm_f16x3.json), and in the fp32-byte modes a K-tile is only 32 k deep.  Requires
// an even K-tile count (always true for fp32 storage: K % 64 == 0, BK = 32).
template <typename T, int BM, int BN, int WM, int WN, int NS, int KC, bool FPRE = false, bool X3 = false, int EPI = EPI_LINEAR, int KPS = 1>
__device__ __forceinline__ void gemm_pipe_tile(const GemmArgs& p, const int m0, const int n0, unsigned char* smem) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int ES = 16 / EPC;                     // element size in bytes
    constexpr int BK = KC * EPC;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int RB = KC * 16;                      // bytes per tile row
    constexpr int RPI = 1024 / RB;                   // rows filled by one 1-KiB DMA wave-instruction
    constexpr int NW = WM * WN;                     // waves per block (4, or 8 = two per SIMD)
    constexpr int GA = BM / (NW * RPI), GB = BN / (NW * RPI);   // DMA instructions per wave per tile
    constexpr int G = GA + GB;
    constexpr int STAGE = (BM + BN) * RB;            // bytes of one K-tile image: A rows then W rows
    constexpr int SLOT = KPS * STAGE;                // bytes per ring slot
    static_assert(KPS == 1 || (KPS == 2 && EPC == 4), "two K-tiles per slot: fp32-storage modes only (even K-tile count)");
    constexpr int NKG = KC / 4;                      // MFMA k-groups per tile
    static_assert((NW == 4 || NW == 8) && BM % (NW * RPI) == 0 && BN % (NW * RPI) == 0, "tile shape");
    static_assert(FN % 2 == 0, "fragments pair up along N");
    constexpr int FP = FN / 2;                       // fragment pairs = 8-column groups per lane per M fragment
    static_assert(NS >= 2 && NS <= 4 && (KC == 4 || KC == 8), "ring depth / row width");
    static_assert(!X3 || (EPC == 4 && KC == 8), "split-f16 mode: fp32 storage, 128-byte tile rows");

    constexpr int CLD = BN + 4;                      // fp32 row stride of the V^T staging tile
    constexpr int EP = (BM * CLD * 4 <= NS * SLOT) ? 1 : 2;    // staging passes (row halves) so it fits the ring
    constexpr int HB = BM / EP;
    constexpr int CBYTES = HB * CLD * 4;
    static_assert(HB % 16 == 0, "staging half must hold whole fragments");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // buffer descriptors (wave-uniform): raw buffers, byte offsets, out-of-range reads return 0
    const int nbatch = p.M / p.Lout;
    // A's descriptor starts `pad` rows BEFORE the tensor so that every per-lane offset is non-negative (the range
    // check is on the unsigned offset); rows in front of / behind a clip are masked explicitly per tap below
    const unsigned a_shift = (unsigned)p.pad * (unsigned)(p.lda * ES);
    const unsigned a_bytes = (unsigned)((((long)nbatch * p.Lin - 1) * p.lda + p.Cp) * ES) + a_shift;
    const unsigned w_bytes = (unsigned)((long)p.N * p.K * ES);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A - a_shift), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);

    // DMA slots of this lane: instruction j of this wave fills rows (wave + 4j)*RPI .. +RPI-1 of the A (or W)
    // tile; lane -> row offset lane / KC, LDS slot lane % KC, source chunk slot ^ swz(row)
    const int lrow = lane / KC, lslot = lane % KC;
    const bool is_conv = p.taps > 1;
    unsigned a_voff[GA]; int a_lpos[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int row = (wave + NW * j) * RPI + lrow;
        const int m = m0 + row;
        const unsigned chunk = (unsigned)((lslot ^ swz<KC>(row)) * 16);
        if (!is_conv) {
            a_lpos[j] = 0;
            a_voff[j] = m < p.M ? (unsigned)m * (unsigned)(p.lda * ES) + chunk : OOB;
        } else {
            const int mm = m < p.M ? m : 0;
            const int b = mm / p.Lout, l = mm - b * p.Lout;
            a_lpos[j] = m < p.M ? l * p.stride - p.pad : -0x40000000;      // invalid rows fail every tap's range test
            a_voff[j] = (unsigned)(((long)b * p.Lin + l * p.stride) * p.lda * ES) + chunk;   // relative to the shifted base
        }
    }
    unsigned b_voff[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int row = (wave + NW * j) * RPI + lrow;
        const int n = n0 + row;
        b_voff[j] = n < p.N ? (unsigned)n * (unsigned)(p.K * ES) + (unsigned)((lslot ^ swzW<KC>(row)) * 16) : OOB;
    }

    // running position of the next tile to issue: tap, channel offset, scalar byte offsets
    int is_tap = 0, is_c0 = 0, is_slot = 0, is_sub = 0;
    unsigned soff_a = 0, soff_w = 0;
    const unsigned tap_step = (unsigned)(p.lda - p.Cp + BK) * ES;            // soff_a jump when the tap advances
    auto issue = [&]() {
        if (EMAGE_DBG(p, 1)) return;
        unsigned char* base = smem + is_slot * SLOT + is_sub * STAGE;
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            unsigned vo = a_voff[j];
            if (is_conv) vo = (unsigned)(a_lpos[j] + is_tap) < (unsigned)p.Lin ? vo : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(base + (wave + NW * j) * 1024),
                                                     16, (int)vo, (int)soff_a, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < GB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(base + BM * RB + (wave + NW * j) * 1024),
                                                     16, (int)b_voff[j], (int)soff_w, 0, 0);
        soff_w += BK * ES;
        is_c0 += BK;
        if (is_c0 == p.Cp) { is_c0 = 0; ++is_tap; soff_a += tap_step; } else { soff_a += BK * ES; }
        if (++is_sub == KPS) { is_sub = 0; if (++is_slot == NS) is_slot = 0; }
    };

    // Epilogue operands (residual rows, bias, slope) are fetched NOW, ahead of the operand DMA: they are the
    // oldest entries of this wave's memory queue, so the first counted vmcnt wait of the K-loop retires them and
    // the epilogue does not start with an exposed HBM round trip.  (Vector path only; the scalar fallback for
    // odd widths loads in the epilogue.)
    const int fr = lane & 15, fg = lane >> 4;
    const int ncol_n = p.out_t ? p.t_col0 : p.N;     // columns below this go to out / out_f32
    T* __restrict__ out = (T*)p.out;
    auto vec_ok = [&]() {
        return (p.N % 8 == 0) && (ncol_n % 8 == 0) && (p.n_store % 8 == 0) &&
               (!out || (p.ldo % 8 == 0 && ((uintptr_t)out & 15) == 0)) &&
               (!p.out_f32 || (p.ldf % 4 == 0 && ((uintptr_t)p.out_f32 & 15) == 0)) &&
               (!p.res || (p.ldr % 8 == 0 && ((uintptr_t)p.res & 15) == 0)) &&
               (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.slope || ((uintptr_t)p.slope & 15) == 0);
    };
    constexpr bool PRE = FM * FN <= 4 || FPRE;       // small tiles only (unless forced): the prefetch costs 4 VGPRs per fragment
    constexpr int PM = PRE ? FM : 1, PP = PRE ? FP : 1;
    float pre_b[PP][8], pre_s[PP][8], pre_r[PM][PP][8];
    bool vec = false;
    if constexpr (PRE) vec = vec_ok();
#pragma unroll
    for (int jp = 0; jp < (PRE ? PP : 0); ++jp) {
        const int n = n0 + wn * WTN + jp * 32 + fg * 8;
        const bool on = PRE && vec && n + 7 < ncol_n;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pre_b[jp][e] = 0.f; pre_s[jp][e] = 1.f; }
        if (on && p.bias) load8<float>(p.bias + n, pre_b[jp]);
        if (on && p.slope) load8<float>(p.slope + n, pre_s[jp]);
#pragma unroll
        for (int i = 0; i < PM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
#pragma unroll
            for (int e = 0; e < 8; ++e) pre_r[i][jp][e] = 0.f;
            if (on && p.res && m < p.M) {
                if (p.res_is_f32) load8<float>((const float*)p.res + (long)m * p.ldr + n, pre_r[i][jp]);
                else load8<T>((const T*)p.res + (long)m * p.ldr + n, pre_r[i][jp]);
            }
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;                          // K-tiles
    const int nst = nk / KPS;                         // ring stages (KPS K-tiles each)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nst) {
#pragma unroll
            for (int u = 0; u < KPS; ++u) issue();
        }

    // fragment read addresses inside a K-tile image (bytes): row * RB + swizzled 16-B slot; fragment i adds the
    // immediate i*16*RB (the swizzle is invariant under +16 rows); k-group 1 flips slot bit 2
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int arow = wm * WTM + fr, brow = wn * WTN + 8 * (fr >> 2) + (fr & 3);
    const unsigned a_rd0 = lds0 + arow * RB + ((fg ^ swz<KC>(arow)) << 4);
    const unsigned b_rd0 = lds0 + BM * RB + brow * RB + ((fg ^ swzW<KC>(brow)) << 4);
    const unsigned a_rd1 = a_rd0 ^ 64u, b_rd1 = b_rd0 ^ 64u;   // lds0 is 128-B aligned, so the xor acts on the slot bit

    unsigned sb = 0;                                  // byte offset of the slot being consumed
    for (int st = 0; st < nst; ++st) {
        // stage st must have landed; stages issued after it (at most NS-2) may stay in flight
        const int newer = nst - 1 - st;
        if (NS >= 4 && newer >= 2) wait_vmcnt<2 * KPS * G>();
        else if (NS >= 3 && newer >= 1) wait_vmcnt<KPS * G>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (st + NS - 1 < nst) {
#pragma unroll
            for (int u = 0; u < KPS; ++u) issue();
        }
        if (EMAGE_DBG(p, 2)) { sb += SLOT; if (sb == NS * SLOT) sb = 0; continue; }
#pragma unroll
        for (int u = 0; u < KPS; ++u) {
        u32x4 af0[FM], bf0[FN];
        const unsigned a0 = a_rd0 + sb + u * STAGE, b0 = b_rd0 + sb + u * STAGE;
        [&]<int... I>(std::integer_sequence<int, I...>) { ((af0[I] = lds_read128_off<I * 16 * RB>(a0)), ...); }(std::make_integer_sequence<int, FM>{});
        if constexpr (X3) {
            // read order: both activation chunks first (they need the VALU split), then the W hi / lo chunks
            u32x4 af1[FM], bf1[FN];
            const unsigned a1 = a_rd1 + sb + u * STAGE, b1 = b_rd1 + sb + u * STAGE;
            [&]<int... I>(std::integer_sequence<int, I...>) { ((af1[I] = lds_read128_off<I * 16 * RB>(a1)), ...); }(std::make_integer_sequence<int, FM>{});
            [&]<int... J>(std::integer_sequence<int, J...>) { ((bf0[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * RB>(b0)), ...); }(std::make_integer_sequence<int, FN>{});
            [&]<int... J>(std::integer_sequence<int, J...>) { ((bf1[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * RB>(b1)), ...); }(std::make_integer_sequence<int, FN>{});
            wait_lgkmcnt<2 * FN>();                   // both activation chunks are there: split them while the W reads land
            __builtin_amdgcn_sched_barrier(0);
            f16x8 ah[FM], al[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) split_f16(af0[i], af1[i], p.a_scale, ah[i], al[i]);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            // three sweeps over the fragments (small terms first), so back-to-back MFMAs never share an accumulator
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = mma_f16(__builtin_bit_cast(f16x8, t == 0 ? bf1[j] : bf0[j]), t == 1 ? al[i] : ah[i], acc[i][j]);
        } else if constexpr (NKG == 2) {
            [&]<int... J>(std::integer_sequence<int, J...>) { ((bf0[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * RB>(b0)), ...); }(std::make_integer_sequence<int, FN>{});
            u32x4 af1[FM], bf1[FN];
            const unsigned a1 = a_rd1 + sb + u * STAGE, b1 = b_rd1 + sb + u * STAGE;
            [&]<int... I>(std::integer_sequence<int, I...>) { ((af1[I] = lds_read128_off<I * 16 * RB>(a1)), ...); }(std::make_integer_sequence<int, FM>{});
            [&]<int... J>(std::integer_sequence<int, J...>) { ((bf1[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * RB>(b1)), ...); }(std::make_integer_sequence<int, FN>{});
            wait_lgkmcnt<FM + FN>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = Elem<T>::mma(__builtin_bit_cast(uint4, bf0[j]), __builtin_bit_cast(uint4, af0[i]), acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = Elem<T>::mma(__builtin_bit_cast(uint4, bf1[j]), __builtin_bit_cast(uint4, af1[i]), acc[i][j]);
        } else {
            [&]<int... J>(std::integer_sequence<int, J...>) { ((bf0[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * RB>(b0)), ...); }(std::make_integer_sequence<int, FN>{});
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = Elem<T>::mma(__builtin_bit_cast(uint4, bf0[j]), __builtin_bit_cast(uint4, af0[i]), acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        }
        sb += SLOT;
        if (sb == NS * SLOT) sb = 0;
    }

    // ---- epilogue.  With the permuted W rows, lane (fr, fg) holds for M fragment i and fragment pair jp the 8
    // consecutive columns  out[m0 + wm*WTM + i*16 + fr][n0 + wn*WTN + jp*32 + fg*8 + e],
    // e = 0..3 from acc[i][2jp], e = 4..7 from acc[i][2jp+1]. ----
    if constexpr (X3) {
        const float os = p.o_scale;                   // a power of two: exact
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = acc[i][j] * os;
    }
    if (EMAGE_DBG(p, 4)) {                            // diagnostics: keep the accumulators live, store nothing
        if (acc[0][0][0] == 123.456f) ((float*)p.out_f32)[0] = 0.f;
        __syncthreads();
        return;
    }
    if (n0 < ncol_n || (out && n0 < p.n_store)) {
        if constexpr (!PRE) vec = vec_ok();
#pragma unroll
        for (int jp = 0; jp < FP; ++jp) {
            const int n = n0 + wn * WTN + jp * 32 + fg * 8;
            if (vec && n + 7 < ncol_n) {
                float bv[8], sv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { bv[e] = 0.f; sv[e] = 1.f; }
                if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { bv[e] = pre_b[jp % PP][e]; sv[e] = pre_s[jp % PP][e]; }
                } else {
                    if (p.bias) load8<float>(p.bias + n, bv);
                    if (p.slope) load8<float>(p.slope + n, sv);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int m = m0 + wm * WTM + i * 16 + fr;
                    if (m >= p.M) continue;
                    float rv[8], v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] = 0.f;
                    if constexpr (PRE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = pre_r[i % PM][jp % PP][e];
                    } else if (p.res) {
                        if (p.res_is_f32) load8<float>((const float*)p.res + (long)m * p.ldr + n, rv);
                        else load8<T>((const T*)p.res + (long)m * p.ldr + n, rv);
                    }
                    if constexpr (EPI == EPI_LSTM) {
                        // 8 consecutive columns = 2 hidden units x (i, f, g, o); rv = the input projection of this step
                        float2 cs = *(const float2*)(p.cstate + (long)m * p.ldc + (n >> 2));
                        float hv[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const f32x4 a4 = acc[i][2 * jp + u];
                            const float gi = lstm_sigmoid<X3>(a4[0] + rv[4 * u + 0]), gf = lstm_sigmoid<X3>(a4[1] + rv[4 * u + 1]);
                            const float gg = lstm_tanh<X3>(a4[2] + rv[4 * u + 2]), go = lstm_sigmoid<X3>(a4[3] + rv[4 * u + 3]);
                            const float cn = gf * (u == 0 ? cs.x : cs.y) + gi * gg;
                            (u == 0 ? cs.x : cs.y) = cn;
                            hv[u] = go * lstm_tanh<X3>(cn);
                        }
                        *(float2*)(p.cstate + (long)m * p.ldc + (n >> 2)) = cs;
                        *(float2*)(p.out_f32 + (long)m * p.ldf + (n >> 2)) = make_float2(hv[0], hv[1]);
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = (e < 4 ? acc[i][2 * jp][e] : acc[i][2 * jp + 1][e - 4]) + bv[e];
                        if (p.res_first) x += rv[e];
                        x = leaky(x, sv[e]);
                        if (!p.res_first) x += rv[e];
                        v[e] = x;
                    }
                    if (EMAGE_DBG(p, 24)) {               // experiment (emage_set_tuning key 1, bits 8 / 16): write-through (sc1) or
                        const int aux_sc1 = 16;       // non-temporal result stores instead of plain ones
                        if (out) {
                            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)m0 * p.ldo), 0, 0x7fffffff, 0x00020000);
                            const int vo = (int)(((long)(m - m0) * p.ldo + n) * ES);
                            if constexpr (EPC == 8) {
                                uint4 t;
                                t.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                                t.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                                t.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
                                t.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
                                const u32x4 tv = {t.x, t.y, t.z, t.w};
                                if (EMAGE_DBG(p, 8)) __builtin_amdgcn_raw_buffer_store_b128(tv, ro, vo, 0, aux_sc1);
                                else __builtin_amdgcn_raw_buffer_store_b128(tv, ro, vo, 0, 2);
                            } else {
                                const u32x4 t0 = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
                                const u32x4 t1 = {__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])};
                                if (EMAGE_DBG(p, 8)) { __builtin_amdgcn_raw_buffer_store_b128(t0, ro, vo, 0, aux_sc1); __builtin_amdgcn_raw_buffer_store_b128(t1, ro, vo + 16, 0, aux_sc1); }
                                else { __builtin_amdgcn_raw_buffer_store_b128(t0, ro, vo, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(t1, ro, vo + 16, 0, 2); }
                            }
                        }
                        if (p.out_f32) {
                            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_f32 + (long)m0 * p.ldf), 0, 0x7fffffff, 0x00020000);
                            const int vo = (int)(((long)(m - m0) * p.ldf + n) * 4);
                            const u32x4 t0 = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
                            const u32x4 t1 = {__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])};
                            if (EMAGE_DBG(p, 8)) { __builtin_amdgcn_raw_buffer_store_b128(t0, rf, vo, 0, aux_sc1); __builtin_amdgcn_raw_buffer_store_b128(t1, rf, vo + 16, 0, aux_sc1); }
                            else { __builtin_amdgcn_raw_buffer_store_b128(t0, rf, vo, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(t1, rf, vo + 16, 0, 2); }
                        }
                    } else {
                        if (out) store8<T>(out + (long)m * p.ldo + n, v);
                        if (p.out_f32) store8<float>(p.out_f32 + (long)m * p.ldf + n, v);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int m = m0 + wm * WTM + i * 16 + fr;
                    if (m >= p.M) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int nn = n + e;
                        if (nn < ncol_n) {
                            float x = (e < 4 ? acc[i][2 * jp][e] : acc[i][2 * jp + 1][e - 4]) + (p.bias ? p.bias[nn] : 0.f);
                            float r = 0.f;
                            if (p.res) r = p.res_is_f32 ? ((const float*)p.res)[(long)m * p.ldr + nn]
                                                        : Elem<T>::from(((const T*)p.res)[(long)m * p.ldr + nn]);
                            if (p.res_first) x += r;
                            x = leaky(x, p.slope ? p.slope[nn] : 1.f);
                            if (!p.res_first) x += r;
                            if (out) out[(long)m * p.ldo + nn] = Elem<T>::to(x);
                            if (p.out_f32) p.out_f32[(long)m * p.ldf + nn] = x;
                        } else if (out && nn >= p.N && nn < p.n_store) {
                            out[(long)m * p.ldo + nn] = Elem<T>::to(0.f);
                        }
                    }
                }
            }
        }
    }
    // V^T columns: stage the tile in LDS (row-major fp32) and store it column-contiguous along the sequence
    if (p.out_t && n0 + BN > p.t_col0) {
        float* Cs = (float*)smem;
#pragma unroll
        for (int h = 0; h < EP; ++h) {
            __syncthreads();                          // ring (or previous half) no longer being read
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int rb = wm * WTM + i * 16;
                if (rb / HB == h) {
#pragma unroll
                    for (int jp = 0; jp < FP; ++jp) {
                        float* dst = Cs + (rb - h * HB + fr) * CLD + wn * WTN + jp * 32 + fg * 8;
                        *(float4*)dst = make_float4(acc[i][2 * jp][0], acc[i][2 * jp][1], acc[i][2 * jp][2], acc[i][2 * jp][3]);
                        *(float4*)(dst + 4) = make_float4(acc[i][2 * jp + 1][0], acc[i][2 * jp + 1][1], acc[i][2 * jp + 1][2], acc[i][2 * jp + 1][3]);
                    }
                }
            }
            __syncthreads();
            transposed_store<T, HB, BN, CLD, NW * 64>(p, Cs, m0 + h * HB, n0, tid);
        }
    }
    __syncthreads();                              // the next tile re-uses the ring / staging LDS
}

}  // namespace emage_dev

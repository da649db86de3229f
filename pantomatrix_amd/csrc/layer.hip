// emage_transformer_layer — one nn.TransformerDecoderLayer / nn.TransformerEncoderLayer (post-norm, ReLU, no masks;
// d = 768, 4 heads x 192, FFN 1536, 64-frame windows) as ONE launch.
//
// Why: at 64 clips x 64 frames a layer is 11 launches of 6-33 us whose MFMA content is 1-3 us each:
// launch floor, first-touch latency and the lock-step prologue / DMA / epilogue phases of every launch add up.  Here
// FOUR workgroups own one clip's 64 rows for the whole layer: member j is head j of both attentions and the column
// slice [192j, 192j+192) of every projection (one 64x192 tile of gemm_pipe_tile — the same routine, hence the same
// bits, as emage_gemm), activations hand over through L2, and the kernel boundaries become group barriers:
//
//   QKV_j (3 tiles) . self-attention head j | out-proj slice j + residual | LayerNorm rows 16j..16j+15 |
//   Q_j . cross-attention head j | out-proj slice j + residual | LayerNorm | FFN-up slices 2j, 2j+1 (ReLU) |
//   FFN-down slice j + residual | LayerNorm (+ post_add)                    ('.' block-local, '|' group barrier)
//
// Group barrier: agent-scope release fence by every wave, one arrive (atomic add on the clip's counter) and a
// bounded spin by thread 0, agent-scope acquire fence (L1 invalidate) by every wave.  The 4 members of clip g are
// blocks 32*(g/8) + 8j + g%8: consecutive in dispatch order (all resident together) and, with the round-robin
// workgroup -> XCD assignment, on ONE XCD, so the hand-over lines stay in that XCD's L2.
// A clip group never waits on another group; a spin that exceeds its bound raises the error word instead of hanging.
#include "common.h"
#include "gemm_tile.h"
#include "attn_tile.h"
#include "ln_row.h"
#include <cstdlib>

namespace {

using namespace emage_dev;

constexpr int D = 768, NH = 4, HD = 192, FF = 1536, TW = 64;   // the only geometry EMAGE uses
constexpr int BM = 64, BN = 192;

struct LnArgs {
    const void* x; const float* g; const float* b; const void* add; void* y;
    int ldx, ldadd, ldy;
};

// A layer is a short program of steps run by every member: a projection tile at column n_base + j * n_mul, an
// attention head (after a block-local hand-over of the tile outputs), or this member's 16 LayerNorm rows.
enum { STEP_TILE = 0, STEP_ATTN = 1, STEP_NORM = 2 };
struct Step { int kind, idx, n_base, n_mul, sync_after; };
constexpr int MAX_STEPS = 16;

struct LayerArgs {
    GemmArgs gemm[6];      // qkv, sa_out, ca_q, ca_out, ff1, ff2
    AttnArgs attn[2];      // self, cross
    LnArgs ln[3];
    Step steps[MAX_STEPS];
    unsigned* sync;        // zeroed by the launcher: [B] arrive counters, [1] error word, [B] per-clip XCD masks
    int prefetch;          // tuning key 3: issue L2 prefetches of the next projection's W slice before each group barrier
    int B, n_steps, dbg;   // dbg (tools/bench_layer.py ablations): 1 no spin, 2 no fences, 4 no tiles, 8 no attention, 16 no LayerNorm, 32/64 agent-scope fences
    float eps;
};

constexpr unsigned SPIN_LIMIT = 1u << 22;     // x ~0.3 us per poll: about a second, then give up loudly

__device__ __forceinline__ void block_handoff() {
    // global stores of this block -> loads of this block (same CU, same L1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// L2-level atomic add returning the old value (no sc1: performed in this XCD's L2, where all members of a group live)
__device__ __forceinline__ unsigned l2_atomic_add(unsigned* p, unsigned v) {
    unsigned r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}

// device-scope OR returning the old value (performed at the memory side: valid whichever XCDs the callers are on)
__device__ __forceinline__ unsigned l2_atomic_or_agent(unsigned* p, unsigned v) {
    return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Group barrier between the 4 workgroups of a clip.  They run on one XCD, so their L2 is shared and coherent: the
// release side only has to drain this wave's stores to L2 (the vector L1 is write-through), the acquire side only has
// to drop the CU's L1 (buffer_inv sc0) — no L2 write-back / invalidate (an agent-scope fence pair costs ~20 us here:
// it empties the L2, and every weight byte then comes from the Infinity Cache again).
// dbg & 32: agent-scope acquire instead; dbg & 64: agent-scope release as well (the portable, slow form).
__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned target, unsigned* err, int dbg) {
    if (dbg & 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else if (!(dbg & 2)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0 && !(dbg & 1)) {
        unsigned seen = l2_atomic_add(ctr, 1u) + 1u;
        unsigned spins = 0;
        while (seen < target) {
            __builtin_amdgcn_s_sleep(1);
            seen = l2_atomic_add(ctr, 0u);
            if (++spins > SPIN_LIMIT) { l2_atomic_or_agent(err, 1u); break; }
        }
    }
    __syncthreads();
    if (dbg & 32) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (!(dbg & 2)) asm volatile("buffer_inv sc0" ::: "memory");
}

// NW = 4: one wave per SIMD (256 VGPR + AGPRs: everything of an attention head preloaded).  NW = 8: two waves per SIMD
// (<= 256 registers each) that hide each other's exposed latencies — wave tile 16x96, two waves per attention query
// tile (each owns half of the output d tiles), two LayerNorm rows per wave.  Same arithmetic per output element.
// Pull the W rows [n0, n0 + BN) of projection p towards this XCD's L2: one dword per 128-byte line, delivered by LDS-DMA
// into a scratch area (no VGPR is written, nothing waits on it).  W never depends on a barrier, so this is issued right
// before the block parks in one: the fetch from the Infinity Cache overlaps the wait, and the tile's ring then streams
// L2 hits.  The loads are older than anything the next tile issues, so its counted vmcnt waits retire them first.
template <int NT>
__device__ __forceinline__ void prefetch_w(const GemmArgs& p, int n0, unsigned char* scratch) {
    const unsigned row_bytes = (unsigned)p.K * 2u;
    const unsigned w_bytes = (unsigned)p.N * row_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);
    const unsigned base = (unsigned)n0 * row_bytes, lines = (unsigned)BN * row_bytes / 128u;
    for (unsigned l = threadIdx.x; l < lines; l += NT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)scratch, 4, (int)(base + l * 128u), 0, 0, 0);
}

// PF (W prefetch ahead of the group barriers) is written but NOT instantiated yet: it was added after the round's GPU
// budget was spent, so the shipped binary is exactly the validated one; flip EMAGE_LAYER_PREFETCH to measure it.
#ifndef EMAGE_LAYER_PREFETCH
#define EMAGE_LAYER_PREFETCH 0
#endif
template <int NS, int NW, bool PF = (EMAGE_LAYER_PREFETCH != 0)>
__global__ __launch_bounds__(NW * 64, 1) void transformer_layer_kernel(LayerArgs a) {
    typedef bf16_t T;
    constexpr int RING = pipe_smem_bytes<T, BM, BN, NS, 8>();
    __shared__ __attribute__((aligned(128))) unsigned char smem[RING + (PF ? 256 : 0)];   // + LDS-DMA scratch of prefetch_w
    const int bidx = blockIdx.x;
    const int g = (bidx >> 5) * 8 + (bidx & 7);          // clip
    const int j = (bidx >> 3) & 3;                       // member: head / column slice
    if (g >= a.B) return;                                // whole groups leave together
    unsigned* ctr = a.sync + g;
    unsigned* err = a.sync + a.B;
    unsigned arrived = 0;
    const int m0 = g * BM;
    // The light barrier below is only coherent inside one XCD: every member publishes the XCD it runs on (XCC_ID, hwreg 20)
    // and the first barrier checks that the clip's four agree; a mismatch raises bit 1 of the error word.
    unsigned* xmask = a.sync + a.B + 1 + g;
    if (threadIdx.x == 0) l2_atomic_or_agent(xmask, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u));
    bool placement_checked = false;
    const int wave = (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = 0; i < a.n_steps; ++i) {
        const Step st = a.steps[i];
        if (st.kind == STEP_TILE) {
            if (!(a.dbg & 4)) gemm_pipe_tile<T, BM, BN, NW / 2, 2, NS, 8, true>(a.gemm[st.idx], m0, st.n_base + j * st.n_mul, smem);
        } else if (st.kind == STEP_ATTN) {
            block_handoff();
            if (!(a.dbg & 8)) attn_tile<T, HD, 4, NW / 4>(a.attn[st.idx], g, j, wave / (NW / 4), wave % (NW / 4));
        } else {
            const LnArgs& n = a.ln[st.idx];
            if (!(a.dbg & 16)) {
                // this wave's rows: all loads in flight before the first reduction
                constexpr int R = 16 / NW;
                const long row0 = m0 + j * 16 + wave * R;
                float4 v[R][3];
#pragma unroll
                for (int r = 0; r < R; ++r) layernorm_row_load<T, 3>((const T*)n.x + (row0 + r) * n.ldx, D, lane, v[r]);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    layernorm_row_finish<T, 3>(v[r], n.g, n.b, a.eps, n.add ? (const T*)n.add + (row0 + r) * n.ldadd : nullptr,
                                               nullptr, (T*)n.y + (row0 + r) * n.ldy, D, lane);
            }
        }
        if (st.sync_after) {
            if constexpr (PF) if (a.prefetch) {            // the next projection's W slice(s): consecutive tiles of one GEMM
                int nx = i + 1;
                while (nx < a.n_steps && a.steps[nx].kind != STEP_TILE) ++nx;
                for (int q = nx; q < a.n_steps && a.steps[q].kind == STEP_TILE && a.steps[q].idx == a.steps[nx].idx; ++q)
                    prefetch_w<NW * 64>(a.gemm[a.steps[q].idx], a.steps[q].n_base + j * a.steps[q].n_mul, smem + RING);
            }
            arrived += NH;
            group_barrier(ctr, arrived, err, a.dbg);
            if (!placement_checked) {
                placement_checked = true;
                if (threadIdx.x == 0) {
                    const unsigned m = l2_atomic_or_agent(xmask, 0u);
                    if (m & (m - 1)) l2_atomic_or_agent(err, 2u);
                }
            }
        }
    }
}

int g_ring = 3;            // ring depth of the fused kernel
int g_dbg = 0;             // ablation mask (timing only: results are wrong when set)
int g_prefetch = 0;        // tuning key 3 (see LayerArgs::prefetch); not yet measured
int g_waves = 8;           // waves per workgroup of the fused kernel (4 or 8; 8 measured 154 vs 203 us per layer)

GemmArgs linear_args(const void* A, int lda, const void* W, const float* bias, const float* slope, const void* res, int ldr,
                     void* out, int ldo, void* out_t, int t_col0, int M, int N, int K) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.slope = slope; g.res = res; g.out = out; g.out_f32 = nullptr; g.out_t = out_t;
    g.lda = lda; g.ldr = ldr; g.ldo = ldo; g.ldf = 0; g.res_is_f32 = 0; g.res_first = 0; g.n_store = 0;
    g.t_col0 = out_t ? t_col0 : N; g.t_rows = TW; g.t_ld = TW;
    g.M = M; g.N = N; g.K = K; g.Cp = K; g.taps = 1; g.stride = 1; g.pad = 0; g.Lin = M; g.Lout = M;
    g.tiles_m = M / BM; g.tiles_n = N / BN; g.dbg = g_dbg >> 8;   // tile-level ablations: 256 no DMA, 512 no LDS reads / MFMA, 1024 no epilogue
    return g;
}


}  // namespace

extern "C" size_t emage_transformer_layer_workspace(int B) {
    if (B <= 0) return 0;
    const size_t M = (size_t)B * TW;
    // qk (M x 2D) | vt (B x D x TW) | att | s | x1 | x2 (M x D each) | f (M x FF), bf16; then 2B + 1 sync words
    return (M * (2 * D + D + 4 * D + FF)) * 2 + (2 * (size_t)B + 1) * 4 + 256;
}

extern "C" int emage_transformer_layer(int dtype, const void* x, int ldx,
                                       const void* const* weights, const float* const* biases,
                                       const float* const* ln_gamma, const float* const* ln_beta, float eps,
                                       const void* mem_k, int ldk, const void* mem_vt, int vt_rows, int ldvt, int Tk,
                                       const void* post_add, int ld_add, const float* relu_slope,
                                       void* workspace, size_t workspace_bytes, void* out, int ldo,
                                       int B, int T, int d_model, int n_head, int d_ffn, void* stream) {
    if (dtype != EMAGE_BF16 || T != TW || d_model != D || n_head != NH || d_ffn != FF) return EMAGE_EINVAL;
    if (!x || !weights || !biases || !ln_gamma || !ln_beta || !workspace || !out || !relu_slope || B <= 0) return EMAGE_EINVAL;
    if (workspace_bytes < emage_transformer_layer_workspace(B) || ((uintptr_t)workspace & 255)) return EMAGE_EINVAL;
    if (ldx % 8 || ldo % 8 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || (post_add && (ld_add % 8 || ((uintptr_t)post_add & 15)))) return EMAGE_EINVAL;
    const bool cross = mem_k != nullptr;
    if (cross && (!mem_vt || Tk <= 32 || Tk > TW || ldk % 8 || ldvt % 32 || ldvt < ((Tk + 31) / 32) * 32 || vt_rows < D ||
                  (((uintptr_t)mem_k | (uintptr_t)mem_vt) & 15))) return EMAGE_EINVAL;
    for (int i = 0; i < 6; ++i) {
        if (!cross && (i == 2 || i == 3)) continue;
        if (!weights[i] || !biases[i] || ((uintptr_t)weights[i] & 15) || ((uintptr_t)biases[i] & 15)) return EMAGE_EINVAL;
    }
    for (int i = 0; i < 3; ++i) {
        if (!cross && i == 1) continue;
        if (!ln_gamma[i] || !ln_beta[i] || (((uintptr_t)ln_gamma[i] | (uintptr_t)ln_beta[i]) & 15)) return EMAGE_EINVAL;
    }
    const int M = B * TW;
    if ((long)M * (cross ? ldk : 2 * D) * 2 >= (1L << 31)) return EMAGE_EINVAL;
    bf16_t* ws = (bf16_t*)workspace;
    bf16_t* qk = ws;                     ws += (size_t)M * 2 * D;
    bf16_t* vt = ws;                     ws += (size_t)M * D;
    bf16_t* att = ws;                    ws += (size_t)M * D;
    bf16_t* s = ws;                      ws += (size_t)M * D;
    bf16_t* x1 = ws;                     ws += (size_t)M * D;
    bf16_t* x2 = ws;                     ws += (size_t)M * D;
    bf16_t* f = ws;                      ws += (size_t)M * FF;
    unsigned* sync = (unsigned*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);

    LayerArgs a{};
    a.B = B; a.eps = eps; a.sync = sync; a.dbg = g_dbg; a.prefetch = g_prefetch;
    int ns = 0;
    auto step = [&](int kind, int idx, int n_base, int n_mul, int sync_after) { a.steps[ns++] = Step{kind, idx, n_base, n_mul, sync_after}; };
    // self-attention: [q | k] -> qk, V^T -> vt; out-proj + x -> s; LN -> x1
    a.gemm[0] = linear_args(x, ldx, weights[0], biases[0], nullptr, nullptr, 0, qk, 2 * D, vt, 2 * D, M, 3 * D, D);
    a.attn[0] = AttnArgs{qk, qk + D, vt, att, 2 * D, 2 * D, TW, D, D, B, NH, TW, TW, 1.0f / sqrtf((float)HD)};
    a.gemm[1] = linear_args(att, D, weights[1], biases[1], nullptr, x, ldx, s, D, nullptr, 0, M, D, D);
    a.ln[0] = LnArgs{s, ln_gamma[0], ln_beta[0], nullptr, x1, D, 0, D};
    step(STEP_TILE, 0, 0, HD, 0); step(STEP_TILE, 0, D, HD, 0); step(STEP_TILE, 0, 2 * D, HD, 0);   // q, k, V^T of head j
    step(STEP_ATTN, 0, 0, 0, 1);
    step(STEP_TILE, 1, 0, HD, 1);
    step(STEP_NORM, 0, 0, 0, 1);
    const bf16_t* xin = x1;
    if (cross) {
        a.gemm[2] = linear_args(x1, D, weights[2], biases[2], nullptr, nullptr, 0, qk, 2 * D, nullptr, 0, M, D, D);
        a.attn[1] = AttnArgs{qk, mem_k, mem_vt, att, 2 * D, ldk, ldvt, vt_rows, D, B, NH, TW, Tk, 1.0f / sqrtf((float)HD)};
        a.gemm[3] = linear_args(att, D, weights[3], biases[3], nullptr, x1, D, s, D, nullptr, 0, M, D, D);
        a.ln[1] = LnArgs{s, ln_gamma[1], ln_beta[1], nullptr, x2, D, 0, D};
        xin = x2;
        step(STEP_TILE, 2, 0, HD, 0);
        step(STEP_ATTN, 1, 0, 0, 1);
        step(STEP_TILE, 3, 0, HD, 1);
        step(STEP_NORM, 1, 0, 0, 1);
    }
    a.gemm[4] = linear_args(xin, D, weights[4], biases[4], relu_slope, nullptr, 0, f, FF, nullptr, 0, M, FF, D);
    a.gemm[5] = linear_args(f, FF, weights[5], biases[5], nullptr, xin, D, s, D, nullptr, 0, M, D, FF);
    a.ln[2] = LnArgs{s, ln_gamma[2], ln_beta[2], post_add, out, D, ld_add, ldo};
    step(STEP_TILE, 4, 0, 2 * HD, 0); step(STEP_TILE, 4, HD, 2 * HD, 1);
    step(STEP_TILE, 5, 0, HD, 1);
    step(STEP_NORM, 2, 0, 0, 0);
    a.n_steps = ns;

    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sync, 0, (2 * (size_t)B + 1) * 4, st) != hipSuccess) return (int)hipGetLastError();
    const int grid = ((B + 7) / 8) * 32;
    static const int env_waves = [] { const char* e = getenv("EMAGE_LAYER_WAVES"); return e ? atoi(e) : 0; }();
    const int waves = (env_waves == 4 || env_waves == 8) ? env_waves : g_waves;
    if (waves == 8) {
        if (g_ring == 2) hipLaunchKernelGGL((transformer_layer_kernel<2, 8>), dim3(grid), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((transformer_layer_kernel<3, 8>), dim3(grid), dim3(512), 0, st, a);
    } else {
        if (g_ring == 2) hipLaunchKernelGGL((transformer_layer_kernel<2, 4>), dim3(grid), dim3(256), 0, st, a);
        else if (g_ring == 4) hipLaunchKernelGGL((transformer_layer_kernel<4, 4>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((transformer_layer_kernel<3, 4>), dim3(grid), dim3(256), 0, st, a);
    }
    return launch_status();
}

// error word of the last layer run on this workspace (tests / diagnostics; synchronises): bit 0 = a group barrier gave up
// waiting, bit 1 = the four workgroups of some clip were not on one XCD (the light barrier is then not coherent)
extern "C" int emage_transformer_layer_status(const void* workspace, int B) {
    const size_t M = (size_t)B * TW;
    const uintptr_t end = (uintptr_t)workspace + (M * (2 * D + D + 4 * D + FF)) * 2;
    const unsigned* sync = (const unsigned*)((end + 255) & ~(uintptr_t)255);
    unsigned v = 0;
    if (hipMemcpy(&v, sync + B, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

extern "C" int emage_layer_set_tuning(int key, int value) {
    if (key == 0 && value >= 2 && value <= 4) { g_ring = value; return 0; }
    if (key == 1) { g_dbg = value; return 0; }
    if (key == 2 && (value == 4 || value == 8)) { g_waves = value; return 0; }
    if (key == 3 && (value == 0 || value == 1)) { g_prefetch = value; return 0; }
    return EMAGE_EINVAL;
}

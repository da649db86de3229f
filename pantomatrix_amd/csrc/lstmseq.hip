// emage_lstm_layer — the whole recurrence of ONE bidirectional nn.LSTM layer (all time steps, both directions) in one launch
// (models/disco_audio/modeling_disco_audio.py:190-195,252; models/camn_audio/modeling_camn_audio.py:204-218,261-268).
//
// The per-step form (lstm.hip) pays a launch and a cold W_hh stream from L2 for ~1.3 us of MFMA work per step.  Here the
// recurrent weights never move after the prologue:
//   * a block owns 16 hidden units (64 gate rows i,f,g,o interleaved per unit) of one direction and 64 clips of the batch;
//     its 4 waves hold their 16 x H slice of W_hh — both fp16 planes of the split-f16 form — in REGISTERS (128 VGPRs at
//     H = 512) for the whole sequence; the cell state of the block's (clip, unit) pairs lives in registers too;
//   * per step: wait until the H/16 blocks of the group (same direction, same clips) have published h_{t-1}, stage the
//     64 x H slice of h_{t-1} into LDS as fp16 hi / lo planes (split once per block, in the packed-weight chunk order, XOR-
//     swizzled rows: conflict-free writes and reads), 3 MFMAs per product as in emage_gemm's X3 form — same operand
//     roles, same order of the three terms, K ascending: BIT-IDENTICAL to the emage_lstm_step sequence — then the LSTM
//     cell in registers, h_t written into the (B, T, 2H) layer output, which is also the exchange buffer of the next step;
//   * hand-over of h_t inside a group (same direction, same clips; its H/16 blocks can sit on any XCD), round 3: THE DATA IS THE
//     FLAG.  The entry point fills the layer output with a sentinel word (0xFFFFFFFF, a NaN no LSTM cell produces); producers
//     store h_t write-through (`sc1`, relaxed agent-scope 4-byte stores) and move on; consumers read h_{t-1} with `sc1` buffer
//     loads and simply re-read while any word of their share still holds the sentinel (one v_max_u32 chain + a wave vote per
//     attempt).  No store drain, no arrival counter, no second poll: the serial path of a step is store -> L2 -> load.
//     Round 2's protocol — drain the stores (`s_waitcnt vmcnt(0)`), block barrier, relaxed agent-scope atomic add on a per-group
//     counter, poll it, then load — is kept as the LL = false instantiation for A/B in the tools build (emage_set_tuning key 3,
//     bit 32).  (Release / acquire FENCES at agent scope cost a whole-L2 write-back / invalidate per wave and step —
//     `buffer_wbl2` / `buffer_inv` — and made the very first version slower than one launch per step: 30-34 us per step against
//     10, profiles/r02_lstm_layer_breakdown.json.)
//     Blocks must be co-resident (one per CU: 128 KB of LDS), so
//     a launch covers at most n_CU / (2 * H/16) slices of 64 clips; larger batches are walked in sequential launches.
//     The waiting loop is bounded: a wave that waits longer than ~1 s raises the error word and every block leaves
//     (the host checks the word) — a lost block can never hang the device.
#include "common.h"
#include <math.h>
#include "gemm_tile.h"
#include "h2_tile.h"          // IC / static_for

namespace {

using namespace emage_dev;

struct SeqArgs {
    const float* gx; long ld_gx_b; int ld_gx_t;        // gates_x[b][t][dir * 4H + 4u + g]: the input projection incl. both biases
    float* hseq; long ld_h_b; int ld_h_t;              // layer output [b][t][dir * H + u]
    const unsigned char* w[2]; float os[2]; float a_scale;
    unsigned* sync;                                    // this launch's record: arrival counter of group g at word 32 g (one 128-B line each), error word at 32 * 16
    int B, T, slices;
    int dbg;                                           // emage_set_tuning key 3 (tools): timing-only
                                                       // ablations (wrong results): 2 = no MFMA phase, 4 = no h load / staging, 8 = no waiting for the group; same bits: 16 = two staging phases (counter protocol only), 32 = round 2's counter protocol, 64 = s_sleep 1 between re-reads
};

constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr unsigned H_SENTINEL = 0xFFFFFFFFu;           // "not written yet": what emage_lstm_layer fills the layer output with (memset 0xFF)
constexpr int MAX_GROUPS = 16;                         // 2 directions x at most 8 slices of 64 clips per launch
static_assert(EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH == 32 * (MAX_GROUPS + 1), "sync record layout");

__device__ __forceinline__ int swz4(int row) { return (-(row >> 2)) & 3; }

// LDS hand-over inside a step: the staging writes of this wave are done (lgkmcnt) and every wave has arrived — a raw s_barrier,
// NOT __syncthreads(): its workgroup-release fence would also drain the global loads still in flight for the next K half
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ROWS (round 5): clips per block — 64, or 32 so that a SMALL batch still fills the chip (DisCo's 128 clips are 2 slices of 64 = 128 blocks on 256
// CUs; as 4 slices of 32 every CU has a block, and a block's staging, MFMA and cell work per time step halve).  Same arithmetic per clip: same bits.
// OCC = 2 (ROWS = 32 only): the register budget of TWO resident blocks per CU (<= 256 VGPRs) — a batch of up to 256 clips then runs as 8 slices of 32
// clips, two independent blocks per CU whose waiting, staging and MFMA phases interleave.
template <int H, int HALVES, bool LL, int ROWS = 64, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void lstm_seq_kernel(SeqArgs p) {
    constexpr int KT = H / 32, WPG = H / 16;           // K-tiles; blocks per group
    constexpr int FB = ROWS / 16;                      // 16-clip fragments per block
    constexpr unsigned PLANE = KT * ROWS * 64;         // one fp16 plane of the ROWS x H slice: [kt][row][4 chunks of 16 B]
    static_assert(ROWS == 64 || ROWS == 32, "clips per block");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    int* s_flag = (int*)(smem + 2 * PLANE);

    const int groups = 2 * p.slices;
    const int g = blockIdx.x % groups, j = blockIdx.x / groups;
    const int dir = g / p.slices, slice = g - dir * p.slices;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int b_base = slice * ROWS;
    const int unit = j * 16 + wave * 4 + fg;           // the hidden unit whose 4 gates this lane ends up with
    unsigned* const cnt = p.sync + 32 * g;
    unsigned* const err = p.sync + 32 * MAX_GROUPS;

    // this wave's 16 gate rows of W_hh, both planes, all K: the first MFMA operand (row = lane & 15, k-chunk = lane >> 4)
    f16x8 wh[KT], wl[KT];
    {
        const unsigned char* wrow = p.w[dir] + (long)(j * 64 + wave * 16 + fr) * (H * 4) + fg * 16;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            wh[kt] = *(const f16x8*)(wrow + kt * 128);
            wl[kt] = *(const f16x8*)(wrow + kt * 128 + 64);
        }
    }
    const float os = p.os[dir], as = p.a_scale;
    float c[FB];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) c[fb] = 0.f;

    const int st_row = lane >> 2, st_g = lane & 3;

    // the input projection of a step is fetched one step ahead (it does not depend on the recurrence): its HBM latency hides
    // behind the previous step instead of sitting on the serial path
    auto load_gx = [&](int step, float4 (&dst)[FB]) {
        const int tt = dir ? p.T - 1 - step : step;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            const int b = b_base + fb * 16 + fr;
            dst[fb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < p.B) dst[fb] = *(const float4*)(p.gx + (long)b * p.ld_gx_b + (long)tt * p.ld_gx_t + dir * 4 * H + 4 * unit);
        }
    };
    float4 rv[FB];
    load_gx(0, rv);
    if constexpr (LL) {
        if (tid == 0) *s_flag = 0;
        __syncthreads();
    }

    for (int s = 0; s < p.T; ++s) {
        const int t = dir ? p.T - 1 - s : s;
        float4 rvn[FB];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) rvn[fb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s == 0 && p.T > 1) load_gx(1, rvn);
        f32x4 acc[FB];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) acc[fb] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (s > 0) {
            if constexpr (!LL) {
            if (p.dbg & 8) {
                __syncthreads();
            } else {
            if (tid == 0) {
                const unsigned target = (unsigned)WPG * (unsigned)s;
                unsigned it = 0;
                int bad = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    ++it;
                    if (it > SPIN_LIMIT || ((it & 63u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (bad) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_flag = bad;
            }
            __syncthreads();
            if (*s_flag) return;
            }
            }

            // stage h_{t-1}[b_base .. +64][dir * H .. +H] as split fp16 planes
            const int tp = dir ? t + 1 : t - 1;
            const float* hp = p.hseq + (long)b_base * p.ld_h_b + (long)tp * p.ld_h_t + dir * H;      // row 0 of the block's slice at step tp
            // rows past the batch end lie beyond num_records: the buffer load returns zeros for them, no branch per load
            const int rows_in = p.B - b_base < ROWS ? p.B - b_base : ROWS;
            const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)hp, 0, (int)((long)(rows_in - 1) * p.ld_h_b * 4) + H * 4, 0x00020000);
            // staging role: iteration `it` of this wave moves rows 16 * (it & 3) + (lane >> 2), chunk lane & 3 of K-tile
            // kt_of(it); with HALVES = 2 (A/B only: measured equal, 9.5 vs 9.4 us per step) iterations [0, KT/2) cover the first K half, the rest the
            // second, and the MFMAs of the first half run while the second half is still arriving
            constexpr int KH = KT / HALVES;                 // K-tiles per staging phase
            constexpr int IPP = KH / 4 * FB;                // staging iterations of a wave per phase: its KH / 4 K-tiles x FB row groups
            constexpr int NI = IPP * HALVES;
            static_assert(KH % 4 == 0, "a phase gives every wave whole K-tiles");
            auto kt_of = [&](int it) { return (it / IPP) * KH + wave * (KH / 4) + (it % IPP) / FB; };
            auto row_of = [&](int it) { return 16 * (it % FB) + st_row; };
            u32x4 c0[NI], c1[NI];
            auto load_part = [&](auto hc) {             // the loads of staging phase `hc` (iterations [hc * KH, (hc + 1) * KH))
                constexpr int HF = decltype(hc)::value;
#pragma unroll
                for (int i2 = 0; i2 < IPP; ++i2) {
                    const int it = HF * IPP + i2;
                    const int row = row_of(it), kt = kt_of(it);
                    const int off = (int)((long)row * p.ld_h_b * 4) + (kt * 32 + st_g * 4) * 4;         // < 2^31: checked by the host entry
                    c0[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off, 0, 16));      // aux 16 = sc1: agent-coherent reads
                    c1[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off + 64, 0, 16));
                }
            };
            auto part_ready = [&](auto hc) {            // wave-uniform: no word of the phase's loads still holds the sentinel
                constexpr int HF = decltype(hc)::value;
                unsigned mx = 0u;
#pragma unroll
                for (int i2 = 0; i2 < IPP; ++i2) {
                    const int it = HF * IPP + i2;
                    const unsigned a = max(max(c0[it].x, c0[it].y), max(c0[it].z, c0[it].w));
                    const unsigned b2 = max(max(c1[it].x, c1[it].y), max(c1[it].z, c1[it].w));
                    mx = max(mx, max(a, b2));
                }
                return __builtin_amdgcn_ballot_w64(mx == H_SENTINEL) == 0ull;
            };
            auto load_h = [&]() { emage_dev::static_for<HALVES>([&](auto hc) { load_part(hc); }); };
            unsigned spins = 0;
            int bad = 0;
            bool gave_up = false;
            // one polling loop of staging phase `hc`; first: the phase's loads have not been issued yet
            auto poll_part = [&](auto hc, const bool first) {
                bool issued = !first;
                for (;;) {
                    if (!issued) load_part(hc);
                    issued = false;
                    if (part_ready(hc)) break;
                    ++spins;
                    if (spins > SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { bad = 1; break; }
                }
            };
            if constexpr (LL && HALVES > 1) {
                // PIPELINED hand-over (round 5, VERDICT next #7): only phase 0 of h_{t-1} is waited for here; the loads of the later phases
                // are issued once and checked just before their staging — they land while phase 0 is staged and multiplied
                if (!(p.dbg & 4)) {                         // (tools ablation bit 4: no h loads / no staging — honoured here as in the single-phase branch)
                    poll_part(emage_dev::IC<0>{}, true);
                    emage_dev::static_for<HALVES - 1>([&](auto hc) { load_part(emage_dev::IC<decltype(hc)::value + 1>{}); });
                }
                if (bad) {
                    if (lane == 0) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *s_flag = 1;
                    }
                }
                lds_barrier();
                if (*s_flag) return;
            } else if constexpr (LL) {
                // the data is the flag: re-read this wave's share until no word of it holds the sentinel any more
                if (!(p.dbg & 4)) {
                    for (;;) {
                        load_h();
                        unsigned mx = 0u;
#pragma unroll
                        for (int it = 0; it < NI; ++it) {
                            const unsigned a = max(max(c0[it].x, c0[it].y), max(c0[it].z, c0[it].w));
                            const unsigned b2 = max(max(c1[it].x, c1[it].y), max(c1[it].z, c1[it].w));
                            mx = max(mx, max(a, b2));
                        }
                        if (__builtin_amdgcn_ballot_w64(mx == H_SENTINEL) == 0ull || (p.dbg & 8)) break;
                        ++spins;
                        if (spins > SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { bad = 1; break; }
                        if (p.dbg & 64) __builtin_amdgcn_s_sleep(1);          // re-read immediately (measured: 8.5 vs 8.9 us per step); tools A/B, bit 64: sleep 64 clocks first
                    }
                }
                if (bad) {                                  // wave-uniform
                    if (lane == 0) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *s_flag = 1;
                    }
                }
                // every wave is past its LDS reads of the previous step (staging below overwrites them) and has voted
                lds_barrier();
                if (*s_flag) return;
            } else {
                if (!(p.dbg & 4)) load_h();
            }
            if (s + 1 < p.T) load_gx(s + 1, rvn);          // behind the h loads in the queue, lands during this step
            emage_dev::static_for<HALVES>([&](auto halfc) {
                constexpr int half = decltype(halfc)::value;
                if constexpr (LL && HALVES > 1 && half > 0) {
                    if (!(p.dbg & 4)) poll_part(halfc, false);                 // normally true at the first look: the phase arrived behind the previous phase's MFMAs
                    if (bad && lane == 0) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *s_flag = 1;
                    }
                }
                if (!(p.dbg & 4)) {
#pragma unroll
                    for (int i2 = 0; i2 < IPP; ++i2) {
                        const int it = half * IPP + i2;
                        const int row = row_of(it), kt = kt_of(it);
                        f16x8 hi, lo;
                        split_f16(c0[it], c1[it], as, hi, lo);
                        const unsigned off = kt * (ROWS * 64) + row * 64 + ((st_g ^ swz4(row)) << 4);
                        *(f16x8*)(smem + off) = hi;
                        *(f16x8*)(smem + PLANE + off) = lo;
                    }
                }
                lds_barrier();
                if constexpr (LL && HALVES > 1 && half > 0) { if (*s_flag) { gave_up = true; return; } }
                if (gave_up) return;
                if (!(p.dbg & 2)) {
#pragma unroll
                    for (int k2 = 0; k2 < KH; ++k2) {
                        const int kt = half * KH + k2;
                        f16x8 ah[FB], al[FB];
#pragma unroll
                        for (int fb = 0; fb < FB; ++fb) {
                            const int row = fb * 16 + fr;
                            const unsigned off = kt * (ROWS * 64) + row * 64 + ((fg ^ swz4(row)) << 4);
                            ah[fb] = *(const f16x8*)(smem + off);
                            al[fb] = *(const f16x8*)(smem + PLANE + off);
                        }
                        // small terms first, the order of emage_gemm's X3 K-loop; consecutive MFMAs never share an accumulator
#pragma unroll
                        for (int fb = 0; fb < FB; ++fb) acc[fb] = mma_f16(wl[kt], ah[fb], acc[fb]);
#pragma unroll
                        for (int fb = 0; fb < FB; ++fb) acc[fb] = mma_f16(wh[kt], al[fb], acc[fb]);
#pragma unroll
                        for (int fb = 0; fb < FB; ++fb) acc[fb] = mma_f16(wh[kt], ah[fb], acc[fb]);
                    }
                }
            });
            if (gave_up) return;
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) acc[fb] = acc[fb] * os;
        }

        // the LSTM cell (gate order i, f, g, o), arithmetic of gemm_tile.h's EPI_LSTM
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            const int b = b_base + fb * 16 + fr;
            const float gi = lstm_sigmoid<true>(acc[fb][0] + rv[fb].x), gf = lstm_sigmoid<true>(acc[fb][1] + rv[fb].y);
            const float gg = lstm_tanh<true>(acc[fb][2] + rv[fb].z), go = lstm_sigmoid<true>(acc[fb][3] + rv[fb].w);
            const float cn = gf * c[fb] + gi * gg;
            c[fb] = cn;
            if (b < p.B) {
                float* dst = p.hseq + (long)b * p.ld_h_b + (long)t * p.ld_h_t + dir * H + unit;
                __hip_atomic_store(dst, go * lstm_tanh<true>(cn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // write-through (sc1)
            }
        }

#pragma unroll
        for (int fb = 0; fb < FB; ++fb) rv[fb] = rvn[fb];

        if constexpr (!LL) {
        if (s + 1 < p.T) {                              // publish h_t to the group
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's h stores have reached the agent coherence point
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        }
    }
}

int max_slices_for(int H);
}
#ifdef EMAGE_TOOLS
namespace emage_dev { int g_lstm_layer_dbg = 0; }   // tools build: emage_set_tuning key 3 (gemm.hip)
#else
namespace emage_dev { constexpr int g_lstm_layer_dbg = 0; }
#endif
namespace {


int max_slices_for(int H) {                            // co-resident blocks: one per CU, 2 * H/16 per slice of 64 clips
    int m = device_cus() / (2 * (H / 16));
    return m > MAX_GROUPS / 2 ? MAX_GROUPS / 2 : m;
}

// The launch geometry for a batch of B clips (round 5).  64 clips per block, one block per CU, is round 4's: a launch holds max_slices_for(H) slices.
//   * B <= 32 x max_slices: 32 clips per block, one block per CU — the whole batch in ONE launch with every CU busy (DisCo's 128 clips: 7.4-7.8 ->
//     6.2-6.4 us per time step, profiles/r05_lstm_layer_32_clip_slices.json);
//   * else 64 clips per block, the batch walked in launches of 64 x max_slices clips (CaMN's 256 clips: one launch, every CU busy).
//   Measured NEGATIVE, tools only (emage_set_tuning key 3 bit 1024): 32 clips per block with TWO blocks per CU (the OCC = 2 build: 256 VGPRs, no spills,
//   2 x 64 KB of LDS) for batches up to 256 clips — CaMN's time step 8.5 -> 11.1-11.3 us, the forward 72 -> 80 ms (profiles/r05_lstm_layer_two_blocks_per_cu.json):
//   two resident blocks double the hand-over traffic and polling on every CU and slow each other more than they fill each other's waits.
struct SeqGeometry { int rows, occ, max_slices; };
SeqGeometry geometry_for(int B, int H) {
    const int one = max_slices_for(H);
    if (one < 1) return SeqGeometry{64, 1, one};
    const int dbg = emage_dev::g_lstm_layer_dbg;
    if (!(dbg & 512) && B <= 32 * one) return SeqGeometry{32, 1, one};                      // tools A/B: 512 = always 64 clips per block
#ifdef EMAGE_TOOLS
    int two = 2 * device_cus() / (2 * (H / 16));
    if (two > MAX_GROUPS / 2) two = MAX_GROUPS / 2;
    if (!(dbg & 512) && (dbg & 1024) && B <= 32 * two) return SeqGeometry{32, 2, two};
#endif
    return SeqGeometry{64, 1, one};
}

template <int H, int HALVES, bool LL, int ROWS = 64, int OCC = 1>
int launch_seq(SeqArgs a, int B, int max_slices, unsigned* sync, hipStream_t s) {
    constexpr int KT = H / 32, WPG = H / 16;
    constexpr size_t LDS = 2 * (size_t)KT * ROWS * 64 + 128;
    static const hipError_t configured = hipFuncSetAttribute((const void*)lstm_seq_kernel<H, HALVES, LL, ROWS, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (configured != hipSuccess) return (int)configured;
    // the group barrier needs every block of a launch resident at once: one block per CU (registers), so the occupancy query must admit >= 1
    // block per CU for this kernel's register / LDS footprint (a plain launch has the residency of a cooperative one, without its check)
    static const int blocks_per_cu = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)lstm_seq_kernel<H, HALVES, LL, ROWS, OCC>, 256, LDS) != hipSuccess) return 0;
        return n;
    }();
    if (blocks_per_cu < 1 || 2 * max_slices * WPG > blocks_per_cu * device_cus()) return EMAGE_EINVAL;
    const float* gx = a.gx;
    float* hseq = a.hseq;
    int chunk = 0;
    for (int b0 = 0; b0 < B; b0 += ROWS * max_slices, ++chunk) {
        const int nb = B - b0 < ROWS * max_slices ? B - b0 : ROWS * max_slices;
        a.gx = gx + (long)b0 * a.ld_gx_b;
        a.hseq = hseq + (long)b0 * a.ld_h_b;
        a.B = nb;
        a.slices = (nb + ROWS - 1) / ROWS;
        a.sync = sync + chunk * EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH;
        hipLaunchKernelGGL((lstm_seq_kernel<H, HALVES, LL, ROWS, OCC>), dim3(2 * a.slices * WPG), dim3(256), LDS, s, a);
        const int rc = launch_status();
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

// counter[0] += number of launch records in `sync` whose error word is set (a block gave up waiting for its group): folded into the
// runners' in-graph health counter, so EVERY replay reports a lost block with the D2H read it does anyway
namespace {
__global__ void lstm_health_kernel(const unsigned* __restrict__ sync, int records, int* __restrict__ counter) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < records && sync[(long)r * EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH + EMAGE_LSTM_SYNC_ERROR_WORD] != 0u) atomicAdd(counter, 1);
}
}  // namespace

extern "C" int emage_lstm_layer_health(const unsigned* sync, int sync_words, int* counter, void* stream) {
    if (!sync || !counter || sync_words <= 0 || sync_words % EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH) return EMAGE_EINVAL;
    const int records = sync_words / EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH;
    hipLaunchKernelGGL(lstm_health_kernel, dim3((records + 63) / 64), dim3(64), 0, (hipStream_t)stream, sync, records, counter);
    return launch_status();
}

extern "C" int emage_lstm_layer_sync_words(int B, int H) {
    if (B <= 0 || (H != 256 && H != 512)) return EMAGE_EINVAL;
    const int max_slices = max_slices_for(H);
    if (max_slices < 1) return EMAGE_EINVAL;
    const SeqGeometry geo = geometry_for(B, H);
    const int launches = (B + geo.rows * geo.max_slices - 1) / (geo.rows * geo.max_slices);
    return launches * EMAGE_LSTM_SYNC_WORDS_PER_LAUNCH;
}

extern "C" int emage_lstm_layer(int dtype, const float* gates_x, long ld_gx_b, int ld_gx_t, const void* w_hh0, const void* w_hh1,
                                float w_scale0, float w_scale1, float a_scale, float* hseq, long ld_h_b, int ld_h_t,
                                int B, int T, int H, unsigned* sync, int sync_words, void* stream) {
    if (dtype != EMAGE_F16X3 || !gates_x || !w_hh0 || !w_hh1 || !hseq || !sync || B <= 0 || T <= 0) return EMAGE_EINVAL;
    if (H != 256 && H != 512) return EMAGE_EINVAL;
    if (!(a_scale > 0.f && w_scale0 > 0.f && w_scale1 > 0.f)) return EMAGE_EINVAL;
    if (ld_gx_t % 4 || ld_gx_t < 8 * H || ld_gx_b % 4 || ld_h_t % 4 || ld_h_t < 2 * H || ld_h_b % 4) return EMAGE_EINVAL;
    if (((uintptr_t)gates_x | (uintptr_t)w_hh0 | (uintptr_t)w_hh1 | (uintptr_t)hseq) & 15 || ((uintptr_t)sync & 3)) return EMAGE_EINVAL;
    if (64 * ld_h_b * 4 + 4096 >= (1L << 31)) return EMAGE_EINVAL;                    // a block's 64 clip rows are addressed with 32-bit buffer offsets
    const int need = emage_lstm_layer_sync_words(B, H);
    if (need < 0 || sync_words < need) return EMAGE_EINVAL;
    const int max_slices = max_slices_for(H);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sync, 0, (size_t)need * sizeof(unsigned), s);
    if (e != hipSuccess) return (int)e;
    const bool ll = !(emage_dev::g_lstm_layer_dbg & 32);
    if (ll) {
        // the hand-over protocol: every word of the layer output starts as the sentinel (0xFFFFFFFF), "not written yet"
        if (ld_h_t == 2 * H && ld_h_b == (long)T * ld_h_t) e = hipMemsetAsync(hseq, 0xFF, (size_t)B * T * 2 * H * sizeof(float), s);
        else if (ld_h_b == (long)T * ld_h_t) e = hipMemset2DAsync(hseq, (size_t)ld_h_t * sizeof(float), 0xFF, (size_t)2 * H * sizeof(float), (size_t)B * T, s);
        else
            for (int b = 0; b < B && e == hipSuccess; ++b)
                e = hipMemset2DAsync(hseq + (long)b * ld_h_b, (size_t)ld_h_t * sizeof(float), 0xFF, (size_t)2 * H * sizeof(float), (size_t)T, s);
        if (e != hipSuccess) return (int)e;
    }
    SeqArgs a{};
    a.gx = gates_x; a.ld_gx_b = ld_gx_b; a.ld_gx_t = ld_gx_t;
    a.hseq = hseq; a.ld_h_b = ld_h_b; a.ld_h_t = ld_h_t;
    a.w[0] = (const unsigned char*)w_hh0; a.w[1] = (const unsigned char*)w_hh1;
    a.os[0] = 1.f / (a_scale * w_scale0); a.os[1] = 1.f / (a_scale * w_scale1);
    a.a_scale = a_scale;
    a.T = T;
    a.dbg = emage_dev::g_lstm_layer_dbg;
#ifdef EMAGE_TOOLS
    if (ll && (a.dbg & 128)) return H == 512 ? launch_seq<512, 1, true>(a, B, max_slices, sync, s) : launch_seq<256, 1, true>(a, B, max_slices, sync, s);   // tools A/B: round 3's single hand-over phase
    if (ll && (a.dbg & 256) && H == 512) return launch_seq<512, 4, true>(a, B, max_slices, sync, s);                                                       // tools A/B: four phases (slower)
    if (!ll) {                                          // tools A/B: round 2's counter protocol (bit 16: two staging phases, measured equal)
        if (a.dbg & 16) return H == 512 ? launch_seq<512, 2, false>(a, B, max_slices, sync, s) : launch_seq<256, 2, false>(a, B, max_slices, sync, s);
        return H == 512 ? launch_seq<512, 1, false>(a, B, max_slices, sync, s) : launch_seq<256, 1, false>(a, B, max_slices, sync, s);
    }
#endif
    // round 5: TWO pipelined hand-over phases — only the first K half of h_{t-1} is waited for up front, the second half's loads land behind the
    // first half's staging + MFMAs: 8.11-8.13 -> 7.48-7.64 us per time step at DisCo's size, 8.4-9.0 -> 8.4-8.5 at CaMN's, same bits
    // (profiles/r05_lstm_layer_pipelined_handover.json); four phases are slower (8.7-8.8 / 9.4-9.5)
    const SeqGeometry geo = geometry_for(B, H);      // round 5 (profiles/r05_lstm_layer_32_clip_slices.json, r05_lstm_layer_two_blocks_per_cu.json)
#ifdef EMAGE_TOOLS
    if (geo.rows == 32 && geo.occ == 2)
        return H == 512 ? launch_seq<512, 2, true, 32, 2>(a, B, geo.max_slices, sync, s) : launch_seq<256, 2, true, 32, 2>(a, B, geo.max_slices, sync, s);
#endif
    if (geo.rows == 32)
        return H == 512 ? launch_seq<512, 2, true, 32>(a, B, geo.max_slices, sync, s) : launch_seq<256, 2, true, 32>(a, B, geo.max_slices, sync, s);
    return H == 512 ? launch_seq<512, 2, true>(a, B, max_slices, sync, s) : launch_seq<256, 2, true>(a, B, max_slices, sync, s);
}

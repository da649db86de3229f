// emage_conv_slab — the stride-1, k = 15 convolutions of the WavEncoder (BasicBlock conv2 of every block and conv1 of the
// stride-1 blocks, P:263-306) with the INPUT SLAB RESIDENT IN LDS, and WavEncoder block 0 fused end to end.
//
// The generic implicit GEMM (emage_gemm) re-fetches the A tile for every tap: 15 taps re-read the same input rows shifted
// by one, so the operand stream of these layers is 15x their input (3.7 GB per launch for block 0 at 128 sequences) and, in
// the split-f16 mode, every A fragment is re-split by every wave at every tap.  Here a block owns 128 output positions of
// ONE sequence and all N = C output channels:
//   * the 128 + 14 input rows it needs are brought into LDS once — in split-f16 mode already split into the fp16 hi / lo
//     planes, chunked exactly like the packed weights, so the K-loop has no VALU work at all;
//   * the K-loop walks (tap, channel group) with the A fragment row offset by the tap, streaming only the W tiles
//     (C rows x 128 B) through the LDS-DMA ring;
//   * SRC_WAVE (block 0, P:283-294): the slab IS conv1 — act(bn1(conv1(wav))) is computed from the raw waveform (Cin = 1,
//     15 taps, VALU) straight into LDS, and the epilogue evaluates the downsample shortcut (conv + bn, same geometry) for its
//     own outputs, so neither the (B*L, 4q) first-layer tensor nor its re-read ever touches HBM.
// Arithmetic (tap-major K order, fmaf chains of the first layer, epilogue association) is that of the unfused sequence
// emage_wav_conv_in + emage_gemm: results are bit-identical, which is what tests/test_kernels_gpu.py asserts.
#include "common.h"
#include <utility>
#include "gemm_tile.h"

namespace {

using namespace emage_dev;

constexpr int SRC_ROWS = 0, SRC_WAVE = 1;
constexpr int SBM = 128;          // output positions per block
constexpr int MAXT = 16;          // first-layer taps (15)

struct SlabArgs {
    const void* A; int lda;                       // SRC_ROWS: input rows (nseq*L, lda) in T
    const float* wav; long ldw, hop; int nclip, Lw, stride1, pad1, taps1;   // SRC_WAVE: waveform windows + first-layer geometry
    const float* w1; const float* b1; float slope1;                         //   conv1 (C, taps1) fp32 (BN folded), bias, LeakyReLU slope
    const float* wds; const float* bds;                                     //   shortcut conv (C, taps1), bias
    const void* W; const float* bias; const float* slope;                   // conv2: packed (C, taps*C) in T / split-f16, bias, slope vector
    const void* res; int ldr;                     // SRC_ROWS: shortcut rows (nseq*L, ldr) in T (added before the activation) or NULL
    void* out; int ldo;
    int nseq, L, taps, pad, tiles_l;
    float a_scale, o_scale;
};

// Slab slot swizzle: 16-byte slot of chunk c in row r is c ^ swzS(r).  A ds_read_b128 service group is 16 lanes = 16 distinct
// rows r0 + tap .. r0 + tap + 15, eight of them (fr in {0-3, 12-15}) reading chunk c0 and eight (fr in 4..11) chunk c0 + 1.
// (r & 7) << 1 leaves chunk bit 0 alone and gives each half of the group 8 distinct values on bits 1-3 for EVERY tap offset
// (r & 15, the first version, collided 2-way on odd taps: SQ_LDS_BANK_CONFLICT 20 %, profiles/r02_pmc_gemm_f16x3.json).
// 128-byte rows (bf16, C = 64) keep the two-rows-per-bank-line form of the GEMM ring.
template <int CH> __device__ __forceinline__ int swzS(int row) { return CH == 8 ? ((row >> 1) & 7) : ((row & 7) << 1); }

template <typename T, bool X3, int C, int SRC, int NS>
__global__ __launch_bounds__(512, 1) void conv_slab_kernel(SlabArgs p) {
    constexpr int EPC = Elem<T>::EPC, ES = 16 / EPC;
    constexpr int RBS = C * ES;                       // slab row bytes
    constexpr int CH = RBS / 16;                      // 16-byte chunks per slab row
    constexpr int KPT = RBS / 128;                    // K-tiles (128 B of a row) per tap
    constexpr int ROWS = SBM + MAXT;                  // slab rows (128 + taps - 1, rounded up)
    constexpr int BN = C, NW = 8, WM = 4, WN = 2;
    constexpr int WTN = BN / WN, FM = 2, FN = WTN / 16, FP = FN / 2;
    constexpr int KPS = 2;                            // K-tiles per W ring slot / barrier (the W tile of one K-tile is only C x 128 B)
    constexpr int WSTAGE = KPS * BN * 128;            // bytes of one W ring slot: KPS consecutive K-tiles, each BN rows x 128 B
    constexpr int GB = KPS * BN / (NW * 8);           // 1-KiB DMA instructions per wave per W stage
    static_assert((C == 64 || C == 128) && FN % 2 == 0 && GB >= 1, "channel count");
    static_assert(!X3 || EPC == 4, "split-f16 slab: fp32 storage");

    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* slab = smem;                                   // ROWS * RBS
    unsigned char* wring = smem + ROWS * RBS;                     // NS * WSTAGE
    float* s_x = (float*)(wring + NS * WSTAGE);                   // SRC_WAVE: the waveform span of this tile, then w1 | wds | b1 | bds

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fg = lane >> 4;
    const int seq = blockIdx.x / p.tiles_l, l0 = (blockIdx.x % p.tiles_l) * SBM;
    const int nk = p.taps * KPT;

    // ---- W ring (as in gemm_pipe_tile: lane-fixed byte offsets, K advance in the scalar offset, permuted-row swizzle) ----
    const unsigned w_bytes = (unsigned)((long)C * p.taps * C * ES);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);
    // DMA instruction j of this wave fills rows (wave + NW*j)*8 .. +7 of the slot image [k-tile 0: BN rows | k-tile 1: BN rows];
    // a K-tile past the end of K (odd tile count) reads beyond the buffer and lands as zeros
    unsigned b_voff[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int srow = (wave + NW * j) * 8 + lane / 8;          // row of the slot image
        const int sub = srow / BN, row = srow - sub * BN;
        b_voff[j] = (unsigned)row * (unsigned)(p.taps * C * ES) + (unsigned)(sub * 128) + (unsigned)(((lane % 8) ^ swzW<8>(row)) * 16);
    }
    const int nst = (nk + KPS - 1) / KPS;             // W stages
    unsigned soff_w = 0;
    int is_slot = 0, is_kt = 0;
    auto issue_w = [&]() {
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const int sub = ((wave + NW * j) * 8) / BN;           // wave-uniform: which K-tile of the slot this instruction fills
            // past the last K-tile the row offset would run into the NEXT weight row: force the out-of-range offset (zeros)
            const unsigned vo = is_kt + sub < nk ? b_voff[j] : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(wring + is_slot * WSTAGE + (wave + NW * j) * 1024),
                                                     16, (int)vo, (int)soff_w, 0, 0);
        }
        soff_w += KPS * 128;
        is_kt += KPS;
        if (++is_slot == NS) is_slot = 0;
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nst) issue_w();

    // ---- slab fill: rows r <-> input position l0 - pad + r of this sequence, zeros outside [0, L) ----
    const int pos0 = l0 - p.pad;
    if constexpr (SRC == SRC_WAVE) {
        // stage the waveform span and the two first-layer filters, then evaluate conv1 for every slab element
        const int t1 = p.taps1;
        const int span = (ROWS - 1) * p.stride1 + t1;
        const float* __restrict__ src = p.wav + (long)(seq % p.nclip) * p.ldw + (long)(seq / p.nclip) * p.hop;
        const int x0 = pos0 * p.stride1 - p.pad1;
        for (int i = tid; i < span; i += 512) {
            const int xi = x0 + i;
            s_x[i] = (xi >= 0 && xi < p.Lw) ? src[xi] : 0.f;
        }
        // the two first-layer filters (conv1, shortcut) and their biases: staged once per block with coalesced loads; the
        // register copies below then come from LDS (16-byte reads) instead of ~120 latency-bound global loads per thread
        float* s_w1 = s_x + ((span + 3) & ~3);
        float* s_wd = s_w1 + C * MAXT;
        float* s_b1 = s_wd + C * MAXT;
        float* s_bd = s_b1 + C;
        for (int i = tid; i < C * MAXT; i += 512) {
            const int c = i / MAXT, k = i - c * MAXT;
            s_w1[i] = k < t1 ? p.w1[c * t1 + k] : 0.f;
            s_wd[i] = k < t1 ? p.wds[c * t1 + k] : 0.f;
        }
        for (int i = tid; i < C; i += 512) { s_b1[i] = p.b1[i]; s_bd[i] = p.bds[i]; }
        __syncthreads();
        // A thread owns one 8-channel item column q (the two 4-channel groups of one K-chunk pair) and every RSTEP-th row:
        // the 15 filter taps of 4 channels stay in registers, each waveform sample is read from LDS once per row and phase
        constexpr int IPR = C / 8, RSTEP = 512 / IPR, RPT = (ROWS + RSTEP - 1) / RSTEP;
        const int q = tid % IPR, rb = tid / IPR;
        const int g32 = q >> 2, f4 = q & 3;
        // channels: split-f16 -> {32g + 4f + e, 32g + 16 + 4f + e}; else 8 consecutive
        const int cgrp[2] = {X3 ? 32 * g32 + 4 * f4 : 8 * q, X3 ? 32 * g32 + 4 * f4 + 16 : 8 * q + 4};
        float v[RPT][8];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float wr[4][MAXT], bb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bb[c] = s_b1[cgrp[ph] + c];
#pragma unroll
                for (int k4 = 0; k4 < MAXT / 4; ++k4) {
                    const float4 w4 = *(const float4*)&s_w1[(cgrp[ph] + c) * MAXT + 4 * k4];
                    wr[c][4 * k4] = w4.x; wr[c][4 * k4 + 1] = w4.y; wr[c][4 * k4 + 2] = w4.z; wr[c][4 * k4 + 3] = w4.w;
                }
            }
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int r = rb + j * RSTEP;
                const int pos = pos0 + r;
                const bool live = r < ROWS && pos >= 0 && pos < p.L && r < SBM + p.taps - 1;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                if (r < ROWS) {
#pragma unroll
                    for (int k = 0; k < MAXT; ++k) {
                        if (k < t1) {
                            const float xv = s_x[r * p.stride1 + k];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = fmaf(xv, wr[c][k], acc[c]);
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) v[j][4 * ph + c] = live ? leaky(acc[c] + bb[c], p.slope1) : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = rb + j * RSTEP;
            if (r >= ROWS) continue;
            unsigned char* rowp = slab + r * RBS;
            if constexpr (X3) {
                f16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xs = v[j][e] * p.a_scale;
                    const _Float16 h = (_Float16)xs;
                    hi[e] = h;
                    lo[e] = (_Float16)(xs - (float)h);
                }
                const int c0 = g32 * 8 + f4;
                *(f16x8*)(rowp + ((c0 ^ swzS<CH>(r)) << 4)) = hi;
                *(f16x8*)(rowp + (((c0 + 4) ^ swzS<CH>(r)) << 4)) = lo;
            } else if constexpr (EPC == 8) {
                uint4 t;
                t.x = (unsigned)f32_to_bf16(v[j][0]) | ((unsigned)f32_to_bf16(v[j][1]) << 16);
                t.y = (unsigned)f32_to_bf16(v[j][2]) | ((unsigned)f32_to_bf16(v[j][3]) << 16);
                t.z = (unsigned)f32_to_bf16(v[j][4]) | ((unsigned)f32_to_bf16(v[j][5]) << 16);
                t.w = (unsigned)f32_to_bf16(v[j][6]) | ((unsigned)f32_to_bf16(v[j][7]) << 16);
                *(uint4*)(rowp + ((q ^ swzS<CH>(r)) << 4)) = t;
            } else {
                *(float4*)(rowp + (((2 * q) ^ swzS<CH>(r)) << 4)) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                *(float4*)(rowp + (((2 * q + 1) ^ swzS<CH>(r)) << 4)) = make_float4(v[j][4], v[j][5], v[j][6], v[j][7]);
            }
        }
    } else {
        const T* __restrict__ A = (const T*)p.A;
        if constexpr (X3) {
            constexpr int IPR = C / 8;                // chunk pairs per row
            for (int it = tid; it < ROWS * IPR; it += 512) {
                const int r = it / IPR, q = it - r * IPR;
                const int pos = pos0 + r;
                const int g32 = q >> 2, f4 = q & 3;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                if (pos >= 0 && pos < p.L) {
                    const float* rp = (const float*)A + ((long)seq * p.L + pos) * p.lda + 32 * g32 + 4 * f4;
                    a = *(const float4*)rp;
                    b = *(const float4*)(rp + 16);
                }
                f16x8 hi, lo;
                const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xs = x[e] * p.a_scale;
                    const _Float16 h = (_Float16)xs;
                    hi[e] = h;
                    lo[e] = (_Float16)(xs - (float)h);
                }
                unsigned char* rowp = slab + r * RBS;
                const int c0 = g32 * 8 + f4;
                *(f16x8*)(rowp + ((c0 ^ swzS<CH>(r)) << 4)) = hi;
                *(f16x8*)(rowp + (((c0 + 4) ^ swzS<CH>(r)) << 4)) = lo;
            }
        } else {
            for (int it = tid; it < ROWS * CH; it += 512) {      // plain 16-byte chunks
                const int r = it / CH, c = it - r * CH;
                const int pos = pos0 + r;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (pos >= 0 && pos < p.L) v = *(const uint4*)((const unsigned char*)A + (((long)seq * p.L + pos) * p.lda) * ES + c * 16);
                *(uint4*)(slab + r * RBS + ((c ^ swzS<CH>(r)) << 4)) = v;
            }
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_slab = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)slab;
    const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wring;
    const int arow = wm * 32 + fr;                    // + tap: the slab row of this lane's first A fragment
    const int brow = wn * WTN + 8 * (fr >> 2) + (fr & 3);
    const unsigned b_rd0 = lds_w + brow * 128 + ((fg ^ swzW<8>(brow)) << 4);
    const unsigned b_rd1 = b_rd0 ^ 64u;

    unsigned sb = 0;
    int tap = 0, h = 0;
    for (int st = 0; st < nst; ++st) {
        const int newer = nst - 1 - st;
        if (NS >= 4 && newer >= 2) wait_vmcnt<2 * GB>();
        else if (NS >= 3 && newer >= 1) wait_vmcnt<GB>();
        else wait_vmcnt<0>();
        if (st == 0) wait_lgkmcnt<0>();               // this wave's slab writes are in LDS
        __builtin_amdgcn_s_barrier();                 // W stage st landed in every wave's view (and, at st = 0, the slab is complete)
        if (st + NS - 1 < nst) issue_w();
#pragma unroll
        for (int sub = 0; sub < KPS; ++sub) {
            // a K-tile past the end (odd tile count) multiplies zero weights with the (finite) rows behind the last tap
            const int row = arow + tap;
            const unsigned a0 = lds_slab + row * RBS + (((h * 8 + fg) ^ swzS<CH>(row)) << 4);
            const unsigned a1 = a0 ^ 64u;
            const unsigned wb = sb + sub * (BN * 128);
            u32x4 af0[FM], af1[FM], bf0[FN], bf1[FN];
            [&]<int... I>(std::integer_sequence<int, I...>) { ((af0[I] = lds_read128_off<I * 16 * RBS>(a0)), ...); }(std::make_integer_sequence<int, FM>{});
            [&]<int... I>(std::integer_sequence<int, I...>) { ((af1[I] = lds_read128_off<I * 16 * RBS>(a1)), ...); }(std::make_integer_sequence<int, FM>{});
            [&]<int... J>(std::integer_sequence<int, J...>) { ((bf0[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * 128>(b_rd0 + wb)), ...); }(std::make_integer_sequence<int, FN>{});
            [&]<int... J>(std::integer_sequence<int, J...>) { ((bf1[J] = lds_read128_off<((J >> 1) * 32 + (J & 1) * 4) * 128>(b_rd1 + wb)), ...); }(std::make_integer_sequence<int, FN>{});
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (X3) {
                // both operands arrive split: chunk (c) = hi plane, chunk (c + 4) = lo plane
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = mma_f16(__builtin_bit_cast(f16x8, t == 0 ? bf1[j] : bf0[j]), __builtin_bit_cast(f16x8, t == 1 ? af1[i] : af0[i]), acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = Elem<T>::mma(__builtin_bit_cast(uint4, bf0[j]), __builtin_bit_cast(uint4, af0[i]), acc[i][j]);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = Elem<T>::mma(__builtin_bit_cast(uint4, bf1[j]), __builtin_bit_cast(uint4, af1[i]), acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (++h == KPT) { h = 0; ++tap; }
        }
        sb += WSTAGE;
        if (sb == NS * WSTAGE) sb = 0;
    }

    // ---- epilogue: lane (fr, fg) holds rows l0 + wm*32 + i*16 + fr, columns wn*WTN + jp*32 + fg*8 + e ----
    T* __restrict__ out = (T*)p.out;
#pragma unroll
    for (int jp = 0; jp < FP; ++jp) {
        const int n = wn * WTN + jp * 32 + fg * 8;
        float bv[8], sv[8];
        load8<float>(p.bias + n, bv);
        load8<float>(p.slope + n, sv);
        float rv[FM][8];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[i][e] = 0.f;
        if constexpr (SRC == SRC_WAVE) {
            // the downsample shortcut of block 0 for this lane's outputs: conv(wav) + bias, no activation (P:288-290); the
            // filter taps of 4 columns at a time in registers (loaded once for both rows), the row's samples read from LDS
            const int t1 = p.taps1;
            const int span = (ROWS - 1) * p.stride1 + t1;
            const float* s_wd = s_x + ((span + 3) & ~3) + C * MAXT;
            const float* s_bd = s_wd + C * MAXT + C;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float wd[4][MAXT], bd[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bd[c] = s_bd[n + 4 * hf + c];
#pragma unroll
                    for (int k4 = 0; k4 < MAXT / 4; ++k4) {
                        const float4 w4 = *(const float4*)&s_wd[(n + 4 * hf + c) * MAXT + 4 * k4];
                        wd[c][4 * k4] = w4.x; wd[c][4 * k4 + 1] = w4.y; wd[c][4 * k4 + 2] = w4.z; wd[c][4 * k4 + 3] = w4.w;
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const float* xs = s_x + (wm * 32 + i * 16 + fr + p.pad) * p.stride1;   // slab row of output l is its tile row + pad
                    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < MAXT; ++k) {
                        if (k < t1) {
                            const float xv = xs[k];
#pragma unroll
                            for (int c = 0; c < 4; ++c) a4[c] = fmaf(xv, wd[c][k], a4[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) rv[i][4 * hf + c] = Elem<T>::from(Elem<T>::to(a4[c] + bd[c]));   // the unfused path stores the shortcut in T
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int lr = wm * 32 + i * 16 + fr, l = l0 + lr;
            if (l >= p.L) continue;
            const long m = (long)seq * p.L + l;
            if constexpr (SRC == SRC_ROWS) {
                if (p.res) load8<T>((const T*)p.res + m * p.ldr + n, rv[i]);
            }
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = (e < 4 ? acc[i][2 * jp][e] : acc[i][2 * jp + 1][e - 4]);
                if constexpr (X3) x *= p.o_scale;
                v[e] = leaky((x + bv[e]) + rv[i][e], sv[e]);
            }
            store8<T>(out + m * p.ldo + n, v);
        }
    }
}

template <typename T, bool X3, int C, int SRC>
int launch_slab(SlabArgs& a, hipStream_t s) {
    constexpr int ES = 16 / Elem<T>::EPC;
    constexpr int NS = 2;     // ring depth: with two K-tiles per slot and (C = 64) two blocks per CU, one slot in flight is enough,
                              // and C = 128 in fp32 bytes (74 KiB slab + 2 x 32 KiB ring) only fits the 160 KiB this way
    a.tiles_l = (a.L + SBM - 1) / SBM;
    size_t lds = (size_t)(SBM + MAXT) * C * ES + (size_t)NS * 2 * C * 128;
    if (SRC == SRC_WAVE) lds += ((size_t)((SBM + MAXT - 1) * a.stride1 + a.taps1 + 3) / 4 * 4 + 2 * (size_t)C * MAXT + 2 * C) * sizeof(float);
    if (lds > 160 * 1024) return EMAGE_EINVAL;
    auto kern = conv_slab_kernel<T, X3, C, SRC, NS>;
    // per instantiation, once (thread-safe static initialisation): allow more than 64 KiB of dynamic LDS
    static const hipError_t configured = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (configured != hipSuccess) return (int)configured;
    hipLaunchKernelGGL(kern, dim3(a.nseq * a.tiles_l), dim3(512), lds, s, a);
    return launch_status();
}

template <int SRC>
int dispatch_slab(int dtype, int C, SlabArgs& a, hipStream_t s) {
    if (C == 64) {
        if (dtype == EMAGE_F16X3) return launch_slab<float, true, 64, SRC>(a, s);
        if (dtype == EMAGE_F32) return launch_slab<float, false, 64, SRC>(a, s);
        return launch_slab<bf16_t, false, 64, SRC>(a, s);
    }
    if (C == 128) {
        if (dtype == EMAGE_F16X3) return launch_slab<float, true, 128, SRC>(a, s);
        if (dtype == EMAGE_F32) return launch_slab<float, false, 128, SRC>(a, s);
        return launch_slab<bf16_t, false, 128, SRC>(a, s);
    }
    return EMAGE_EINVAL;
}

}  // namespace

extern "C" int emage_conv_slab(int dtype, const void* A, int lda, const void* W, const float* bias, const float* slope,
                               const void* res, int ldr, void* out, int ldo,
                               int nseq, int L, int C, int taps, int pad, float a_scale, float w_scale, void* stream) {
    if (!A || !W || !bias || !slope || !out || nseq <= 0 || L <= 0 || taps <= 0 || taps > MAXT || pad < 0 || pad >= taps) return EMAGE_EINVAL;
    if (dtype != EMAGE_BF16 && dtype != EMAGE_F32 && dtype != EMAGE_F16X3) return EMAGE_EINVAL;
    const int epc = dtype == EMAGE_BF16 ? 8 : 4;
    if (lda % epc || lda < C || ldo % 8 || ldo < C || (res && (ldr % 8 || ldr < C))) return EMAGE_EINVAL;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)res | (uintptr_t)bias | (uintptr_t)slope) & 15) return EMAGE_EINVAL;
    if (dtype == EMAGE_F16X3 && !(a_scale > 0.f && w_scale > 0.f)) return EMAGE_EINVAL;
    SlabArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.bias = bias; a.slope = slope; a.res = res; a.ldr = ldr; a.out = out; a.ldo = ldo;
    a.nseq = nseq; a.L = L; a.taps = taps; a.pad = pad;
    a.a_scale = dtype == EMAGE_F16X3 ? a_scale : 1.f;
    a.o_scale = dtype == EMAGE_F16X3 ? 1.f / (a_scale * w_scale) : 1.f;
    return dispatch_slab<SRC_ROWS>(dtype, C, a, (hipStream_t)stream);
}

extern "C" int emage_wav_block0(int dtype, const float* wav, long ldw, int Lw, int nwin, long hop, int nclip,
                                const float* w1, const float* b1, float slope1, const float* wds, const float* bds,
                                int taps1, int stride1, int pad1,
                                const void* W2, const float* bias2, const float* slope2, int taps2, int pad2,
                                void* out, int ldo, int L, int C, float a_scale, float w_scale, void* stream) {
    if (!wav || !w1 || !b1 || !wds || !bds || !W2 || !bias2 || !slope2 || !out || nwin <= 0 || nclip <= 0 || L <= 0) return EMAGE_EINVAL;
    if (dtype != EMAGE_BF16 && dtype != EMAGE_F32 && dtype != EMAGE_F16X3) return EMAGE_EINVAL;
    if (taps1 <= 0 || taps1 > MAXT || taps2 <= 0 || taps2 > MAXT || pad2 < 0 || pad2 >= taps2 || stride1 <= 0 || stride1 > 8 || hop < 0 || Lw <= 0) return EMAGE_EINVAL;
    if (ldw < (long)(nwin - 1) * hop + Lw || ldo % 8 || ldo < C) return EMAGE_EINVAL;
    if (((uintptr_t)W2 | (uintptr_t)out | (uintptr_t)bias2 | (uintptr_t)slope2) & 15) return EMAGE_EINVAL;
    if (dtype == EMAGE_F16X3 && !(a_scale > 0.f && w_scale > 0.f)) return EMAGE_EINVAL;
    SlabArgs a{};
    a.wav = wav; a.ldw = ldw; a.hop = hop; a.nclip = nclip; a.Lw = Lw; a.stride1 = stride1; a.pad1 = pad1; a.taps1 = taps1;
    a.w1 = w1; a.b1 = b1; a.slope1 = slope1; a.wds = wds; a.bds = bds;
    a.W = W2; a.bias = bias2; a.slope = slope2; a.out = out; a.ldo = ldo;
    a.nseq = nwin * nclip; a.L = L; a.taps = taps2; a.pad = pad2;
    a.a_scale = dtype == EMAGE_F16X3 ? a_scale : 1.f;
    a.o_scale = dtype == EMAGE_F16X3 ? 1.f / (a_scale * w_scale) : 1.f;
    return dispatch_slab<SRC_WAVE>(dtype, C, a, (hipStream_t)stream);
}

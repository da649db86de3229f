// emage_wav_conv_in — first layer of WavEncoder: Conv1d(1 -> C, k=15, stride 5, pad 1600) on the raw 16 kHz
// waveform with eval BatchNorm folded, LeakyReLU per channel, for conv1 and the downsample shortcut of
// block 0 at once (and for both audio encoders when the host stacks their filters).  K = 15 is far too thin
// for MFMA; this is a VALU kernel whose cost is the (B*Lout, C) store, which it does as coalesced rows.
#include "common.h"

namespace {

// Output positions per block: 16 per thread — a thread keeps 4 channels x 16 taps of filters in registers, loaded once per block, so narrow
// encoders (DisCo / CaMN: C = 32 -> 8 channel groups, 32 row lanes) need MANY rows per block to amortise those 64 loads: with a fixed 64 rows
// (2 per thread) the kernel ran at 0.4-0.8 TB/s of its output stream (CaMN: 7.3 ms for 2.9 GB, profiles/r05_lstm_kernel_stats_camn.csv).
// C = 256 (EMAGE's two stacked encoders) keeps its 64 rows.  Per output the arithmetic is unchanged: the same bits.
static inline int rows_per_block(int C) { const int r = 16384 / C; return r < 64 ? 64 : (r > 1024 ? 1024 : r); }
constexpr int MAXTAPS = 16;

__device__ __forceinline__ void store4(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    t.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    *(uint2*)p = t;
}

template <typename T>
__global__ __launch_bounds__(256) void wav_conv_in_kernel(const float* __restrict__ wav, long ldw, int L, int nclip, long hop, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ slope,
                                                          T* __restrict__ out, int ldo, int Lout, int C, int taps, int stride, int pad, int ROWS) {
    extern __shared__ float s_x[];                       // ROWS*stride + taps samples
    const int b = blockIdx.y;                            // output sequence b = window*nclip + clip
    const float* __restrict__ src = wav + (long)(b % nclip) * ldw + (long)(b / nclip) * hop;
    const int l0 = blockIdx.x * ROWS;
    const int span = (ROWS - 1) * stride + taps;
    const int x0 = l0 * stride - pad;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const int xi = x0 + i;
        s_x[i] = (xi >= 0 && xi < L) ? src[xi] : 0.f;
    }
    __syncthreads();
    const int ngroups = C >> 2;                          // 4 channels per thread
    {   // channel group fixed per thread (256 % ngroups == 0, i.e. C in {8,16,...,1024} powers of two)
        const int cg = threadIdx.x % ngroups, rl = threadIdx.x / ngroups;
        float wr[4][MAXTAPS];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < MAXTAPS; ++k) wr[c][k] = k < taps ? w[(cg * 4 + c) * taps + k] : 0.f;
        float bv[4], sv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { bv[c] = bias ? bias[cg * 4 + c] : 0.f; sv[c] = slope ? slope[cg * 4 + c] : 1.f; }
        const int rstep = blockDim.x / ngroups;
        for (int r = rl; r < ROWS; r += rstep) {
            const int l = l0 + r;
            if (l >= Lout) break;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < MAXTAPS; ++k) {
                if (k < taps) {
                    const float xv = s_x[r * stride + k];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(xv, wr[c][k], acc[c]);
                }
            }
            T* op = out + ((long)b * Lout + l) * ldo + cg * 4;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = leaky(acc[c] + bv[c], sv[c]);
            store4(op, v);                               // one 8-byte (bf16) / 16-byte (fp32) store per thread
        }
    }
}

}  // namespace

extern "C" int emage_wav_conv_in(int dtype, const float* wav, long ldw, int L, int nwin, long hop,
                                 const float* w, const float* bias, const float* slope,
                                 void* out, int ldo, int B, int Lout, int C, int taps, int stride, int pad, void* stream) {
    if (!wav || !w || !out || B <= 0 || Lout <= 0 || C <= 0 || C % 8 || taps <= 0 || taps > MAXTAPS || stride <= 0 || ldo < C) return EMAGE_EINVAL;
    if (256 % (C / 4) != 0 || C / 4 > 256 || ldo % 4 != 0 || ((uintptr_t)out & 15)) return EMAGE_EINVAL;
    if (nwin <= 0 || hop < 0 || L <= 0 || ldw < (long)(nwin - 1) * hop + L || (long)nwin * B > 65535) return EMAGE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int ROWS = rows_per_block(C);
    while (ROWS > 64 && ((size_t)(ROWS - 1) * stride + taps) * sizeof(float) > 60 * 1024) ROWS >>= 1;       // the sample span stays inside 60 KB of LDS
    const dim3 grid((Lout + ROWS - 1) / ROWS, nwin * B), block(256);
    const size_t lds = ((size_t)(ROWS - 1) * stride + taps) * sizeof(float);
    if (dtype == EMAGE_BF16) hipLaunchKernelGGL((wav_conv_in_kernel<bf16_t>), grid, block, lds, s, wav, ldw, L, B, hop, w, bias, slope, (bf16_t*)out, ldo, Lout, C, taps, stride, pad, ROWS);
    else if (dtype == EMAGE_F32) hipLaunchKernelGGL((wav_conv_in_kernel<float>), grid, block, lds, s, wav, ldw, L, B, hop, w, bias, slope, (float*)out, ldo, Lout, C, taps, stride, pad, ROWS);
    else return EMAGE_EINVAL;
    return launch_status();
}

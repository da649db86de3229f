// emage_attention — softmax(Q K^T / sqrt(hd)) V for the EMAGE transformer layers: Tq, Tk <= 128,
// hd = 192, no mask.  One wave64 per (batch, head, 16-query tile); no LDS at all:
//   * S^T = K Q^T is computed "swapped" (A operand = K rows, B operand = Q rows), so a lane holds, for
//     its query column q = lane&15, the scores of keys {16*nt + 4*(lane>>4) + r}: the softmax reduction
//     is in-lane plus two cross-lane steps (xor 16, 32), and the probabilities are already laid out as
//     the A operand of the P V product (row = query, contraction = key);
//   * V arrives transposed (emitted by emage_gemm's out_t path), so its B-operand chunks are
//     contiguous; Q, K, V^T chunks are loaded straight from L2 — each (b,h) problem is 72 KB.
// The key order inside a P chunk is {16*nt+4g+r} (not 8 consecutive keys); V^T is gathered with the
// same mapping, and MFMA sums over the contraction index, so any consistent bijection is exact.
#include "common.h"
#include <math.h>
#include "attn_tile.h"

namespace emage_dev {
#ifdef EMAGE_TOOLS
int g_attn_variant = 0;            // tools build: emage_set_tuning key 6; 1 = the register path (every wave fetches K / V^T itself) for A/B
#else
constexpr int g_attn_variant = 0;
#endif
}

namespace {

using namespace emage_dev;

template <typename T, int HD, int NT, bool X3, bool H2OUT = false>
__global__ __launch_bounds__(64 * QW * DS, 1) void attn_kernel(AttnArgs p) {
    const int qtiles = (p.Tq + 15) >> 4;
    const int qgroups = (qtiles + QW - 1) / QW;
    int bid = blockIdx.x;
    const int wave = (int)(threadIdx.x >> 6);
    const int qt = (bid % qgroups) * QW + wave / DS; bid /= qgroups;
    const int h = bid % p.H;
    const int b = bid / p.H;
    if (qt >= qtiles) return;                 // no barriers below: surplus waves simply leave
    attn_tile<T, HD, NT, DS, X3, H2OUT>(p, b, h, qt, wave % DS);
}

// Split-f16 form, Tk <= 64: K and V^T of the workgroup's (batch, head) are staged once in LDS as split fp16 planes (attn_tile.h,
// attn_stage_kv); the four query-tile waves then run attn_tile on LDS fragments.  Bit-identical to attn_kernel<float, ..., true, ...>.
template <int HD, int NT, bool H2OUT>
__global__ __launch_bounds__(64 * QW, 1) void attn_x3_lds_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int qtiles = (p.Tq + 15) >> 4;
    const int qgroups = (qtiles + QW - 1) / QW;
    int bid = blockIdx.x;
    const int wave = (int)(threadIdx.x >> 6);
    const int qt = (bid % qgroups) * QW + wave; bid /= qgroups;
    const int h = bid % p.H;
    const int b = bid / p.H;
    attn_tile<float, HD, NT, 1, true, H2OUT, true>(p, b, h, qt, 0, smem);     // stages K / V^T with all four waves, then leaves if qt >= qtiles
}

template <int NT, bool H2OUT>
int launch_x3_lds(AttnArgs& a, int grid, hipStream_t s) {
    constexpr int LDS = AttnLds<192, NT>::BYTES;
    static const hipError_t configured = hipFuncSetAttribute((const void*)attn_x3_lds_kernel<192, NT, H2OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (configured != hipSuccess) return (int)configured;
    hipLaunchKernelGGL((attn_x3_lds_kernel<192, NT, H2OUT>), dim3(grid), dim3(64 * QW), LDS, s, a);
    return launch_status();
}

template <typename T, bool X3, bool H2OUT = false>
int dispatch(AttnArgs& a, int hd, hipStream_t s) {
    if (hd != 192) return EMAGE_EINVAL;
    const int qtiles = (a.Tq + 15) / 16;
    const int grid = a.B * a.H * ((qtiles + QW - 1) / QW);
    if constexpr (X3 && DS == 1) {
        if (a.Tk <= 64 && !(emage_dev::g_attn_variant & 1)) return a.Tk <= 32 ? launch_x3_lds<2, H2OUT>(a, grid, s) : launch_x3_lds<4, H2OUT>(a, grid, s);
    }
    if (a.Tk <= 32) hipLaunchKernelGGL((attn_kernel<T, 192, 2, X3, H2OUT>), dim3(grid), dim3(64 * QW * DS), 0, s, a);
    else if (a.Tk <= 64) hipLaunchKernelGGL((attn_kernel<T, 192, 4, X3, H2OUT>), dim3(grid), dim3(64 * QW * DS), 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<T, 192, 8, X3, H2OUT>), dim3(grid), dim3(64 * QW * DS), 0, s, a);
    return launch_status();
}

}  // namespace

static int attention_impl(int dtype, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int vt_rows,
                          void* out, int ldo, int B, int H, int Tq, int Tk, int hd, const float* pmask, void* stream) {
    if (!q || !k || !vt || !out || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || Tk > 128 || vt_rows < H * hd) return EMAGE_EINVAL;
    emage_dev::H2Scale hs;
    if (emage_dev::h2_dtype(dtype, hs)) return EMAGE_EINVAL;
    if (dtype != EMAGE_BF16 && dtype != EMAGE_F32 && dtype != EMAGE_F16X3 && dtype != EMAGE_H2) return EMAGE_EINVAL;
    if (pmask && (dtype == EMAGE_BF16 || dtype == EMAGE_H2)) return EMAGE_EINVAL;
    if (dtype == EMAGE_H2 && ldo % 8) return EMAGE_EINVAL;                  // out is an EMAGE_H2 image; q / k / vt are float32                 // the training forward runs in the fp32-storage modes
    const int epc = dtype == EMAGE_BF16 ? 8 : 4;
    if (ldq % epc || ldk % epc || ldo % 4 || ldvt % 32 || ldvt < ((Tk + 31) / 32) * 32) return EMAGE_EINVAL;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) return EMAGE_EINVAL;
    if ((long)B * Tq * ldq * 4 >= (1L << 31) || (long)B * Tk * ldk * 4 >= (1L << 31) || (long)B * vt_rows * ldvt * 4 >= (1L << 31)) return EMAGE_EINVAL;
    AttnArgs a{q, k, vt, out, ldq, ldk, ldvt, vt_rows, ldo, B, H, Tq, Tk, 1.0f / sqrtf((float)hd), pmask, hs.s, hs.inv};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EMAGE_F16X3) return dispatch<float, true>(a, hd, s);     // float32 tensors, split-f16 MFMA
    if (dtype == EMAGE_H2) return dispatch<float, true, true>(a, hd, s);  // the same arithmetic, output as an EMAGE_H2 image
    return dtype == EMAGE_BF16 ? dispatch<bf16_t, false>(a, hd, s) : dispatch<float, false>(a, hd, s);
}

extern "C" int emage_attention(int dtype, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int vt_rows,
                               void* out, int ldo, int B, int H, int Tq, int Tk, int hd, void* stream) {
    return attention_impl(dtype, q, ldq, k, ldk, vt, ldvt, vt_rows, out, ldo, B, H, Tq, Tk, hd, nullptr, stream);
}

extern "C" int emage_attention_dropout(int dtype, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int vt_rows,
                                       void* out, int ldo, int B, int H, int Tq, int Tk, int hd, const float* pmask, void* stream) {
    if (!pmask) return EMAGE_EINVAL;
    return attention_impl(dtype, q, ldq, k, ldk, vt, ldvt, vt_rows, out, ldo, B, H, Tq, Tk, hd, pmask, stream);
}

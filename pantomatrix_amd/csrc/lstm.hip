// Kernels of the two LSTM gesture generators next to EMAGE: DisCo
// (models/disco_audio/modeling_disco_audio.py, D:) and CaMN (models/camn_audio/modeling_camn_audio.py, C:).
//   emage_lstm_step      one time step of one direction of nn.LSTM for the whole batch: h_{t-1} W_hh^T on MFMA (the
//                        emage_gemm tile routine) with the LSTM cell fused into the epilogue
//   emage_softmax2_mix   DisCo's two-expert content blend (D:244-247)
//   emage_lstm_inputs    the [speaker | seed motion | is-seed flag] tail of the LSTM input rows (D:200-243, C:224-259)
//   emage_rot6d_scatter  rot-6D -> axis-angle scattered into the 55 SMPL-X joints (D:256-258, recover_from_mask_ts D:81-96)
// The input projection x_t W_ih^T of ALL time steps is one emage_gemm per layer (both directions stacked along N); only
// the recurrence is sequential: one launch per step and direction, captured in a hipGraph by the host.
#include "common.h"
#include <math.h>
#include "gemm_tile.h"
#include "rot_math.h"

namespace {

using namespace emage_dev;
using namespace emage_rot;

template <bool X3>
__global__ __launch_bounds__(256, 3) void lstm_step_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[pipe_smem_bytes<float, 64, 64, 2, 8>()];
    const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
    gemm_pipe_tile<float, 64, 64, 2, 2, 2, 8, false, X3, EPI_LSTM>(p, tile_m * 64, tile_n * 64, smem);
}

// Both directions of a bidirectional layer in ONE launch: blockIdx.y picks the direction's problem (its own time step, h_prev,
// W_hh, gate rows, cell state, output columns).  The two recurrences are independent, so this halves the launch count of the
// sequential part and makes the two problems share the CUs by construction instead of by stream scheduling.
template <bool X3>
__global__ __launch_bounds__(256, 3) void lstm_step_pair_kernel(GemmArgs p0, GemmArgs p1) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[pipe_smem_bytes<float, 64, 64, 2, 8>()];
    const GemmArgs& p = blockIdx.y == 0 ? p0 : p1;
    const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
    gemm_pipe_tile<float, 64, 64, 2, 2, 2, 8, false, X3, EPI_LSTM>(p, tile_m * 64, tile_n * 64, smem);
}

__global__ __launch_bounds__(256) void softmax2_mix_kernel(const float* __restrict__ sel, int lds, const float* __restrict__ c1, int ld1,
                                                           const float* __restrict__ c2, int ld2, float* __restrict__ out, int ldo, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        const float a = sel[(long)m * lds], b = sel[(long)m * lds + 1];
        const float mx = fmaxf(a, b);                      // torch.softmax: exp(x - max) / sum
        const float ea = expf(a - mx), eb = expf(b - mx);
        const float s = ea + eb;
        out[(long)m * ldo + c] = (ea / s) * c1[(long)m * ld1 + c] + (eb / s) * c2[(long)m * ld2 + c];
    }
}

// Row (b, t) of the LSTM input, columns [0, F + P + 1 + pad): [speaker_embedding[id_b] (F) | seed motion (P) | is-seed | 0...].
// The seed block is the reference's padded tensor (D:229-242): frame s < seed_frames carries seed_motion[b][s] and flag 1,
// every other frame zeros; `src_map` gives, for each output frame t, the frame of that padded tensor it shows (the
// length reconciliation between motion and audio frames), or -1 for an all-zero frame.
__global__ __launch_bounds__(256) void lstm_inputs_kernel(const float* __restrict__ spk_table, const int64_t* __restrict__ spk_id, int F,
                                                          const float* __restrict__ seed_motion, long ld_seed_b, int P, int seed_frames,
                                                          const int* __restrict__ src_map, float* __restrict__ out, int ldo, int n_store,
                                                          int B, int Tn) {
    const long total = (long)B * Tn * n_store;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n_store), c = (int)(i - (long)m * n_store);
        const int b = m / Tn, t = m - b * Tn;
        float v = 0.f;
        if (c < F) {
            v = spk_table[spk_id[b] * F + c];
        } else if (c <= F + P) {
            const int s = src_map[t];
            if (s >= 0 && s < seed_frames) v = (c == F + P) ? 1.f : (seed_motion ? seed_motion[(long)b * ld_seed_b + (long)s * P + (c - F)] : 0.f);
        }
        out[(long)m * ldo + c] = v;
    }
}

// one thread per (row, SMPL-X joint): selected joints convert their rot-6D, the others are zero
__global__ __launch_bounds__(256) void rot6d_scatter_kernel(const float* __restrict__ rot6d, int ldr, const int* __restrict__ slot_of_joint,
                                                            float* __restrict__ out, int M, int J) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * J) return;
    const int m = (int)(i / J), j = (int)(i - (long)m * J);
    const int slot = slot_of_joint[j];
    float aa[3] = {0.f, 0.f, 0.f};
    if (slot >= 0) {
        float d6[6];
        for (int c = 0; c < 6; ++c) d6[c] = rot6d[(long)m * ldr + slot * 6 + c];
        rot6d_to_aa(d6, aa);
    }
    for (int c = 0; c < 3; ++c) out[((long)m * J + j) * 3 + c] = aa[c];
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

static int lstm_args(GemmArgs& a, int dtype, const float* h_prev, int ld_hprev, const void* w_hh, float w_scale, float a_scale,
                     const float* gates_x, int ld_gx, float* cstate, int ldc, float* h_out, int ld_hout, int B, int H) {
    if (!h_prev || !w_hh || !gates_x || !cstate || !h_out || B <= 0 || H <= 0 || H % 64 != 0) return EMAGE_EINVAL;
    if (dtype != EMAGE_F32 && dtype != EMAGE_F16X3) return EMAGE_EINVAL;
    if (ld_hprev % 4 || ld_hprev < H || ld_gx % 8 || ld_gx < 4 * H || ldc % 2 || ldc < H || ld_hout % 4 || ld_hout < H) return EMAGE_EINVAL;
    if (((uintptr_t)h_prev | (uintptr_t)w_hh | (uintptr_t)gates_x | (uintptr_t)h_out) & 15 || ((uintptr_t)cstate & 7)) return EMAGE_EINVAL;
    if (dtype == EMAGE_F16X3 && !(a_scale > 0.f && w_scale > 0.f)) return EMAGE_EINVAL;
    a = GemmArgs{};
    a.A = h_prev; a.W = w_hh; a.res = gates_x; a.out_f32 = h_out; a.cstate = cstate;
    a.lda = ld_hprev; a.ldr = ld_gx; a.ldf = ld_hout; a.ldc = ldc; a.res_is_f32 = 1;
    a.M = B; a.N = 4 * H; a.K = H; a.Cp = H; a.taps = 1; a.stride = 1; a.pad = 0; a.Lin = B; a.Lout = B;
    a.t_col0 = a.N; a.t_rows = 1;
    a.a_scale = dtype == EMAGE_F16X3 ? a_scale : 1.f;
    a.o_scale = dtype == EMAGE_F16X3 ? 1.f / (a_scale * w_scale) : 1.f;
    a.tiles_m = (B + 63) / 64; a.tiles_n = (4 * H) / 64;
    return 0;
}

extern "C" int emage_lstm_step_pair(int dtype, const float* h_prev0, const float* h_prev1, int ld_hprev0, int ld_hprev1,
                                    const void* w_hh0, const void* w_hh1, float w_scale0, float w_scale1, float a_scale,
                                    const float* gates_x0, const float* gates_x1, int ld_gx,
                                    float* cstate0, float* cstate1, int ldc, float* h_out0, float* h_out1, int ld_hout,
                                    int B, int H, void* stream) {
    GemmArgs a0, a1;
    int rc = lstm_args(a0, dtype, h_prev0, ld_hprev0, w_hh0, w_scale0, a_scale, gates_x0, ld_gx, cstate0, ldc, h_out0, ld_hout, B, H);
    if (rc) return rc;
    rc = lstm_args(a1, dtype, h_prev1, ld_hprev1, w_hh1, w_scale1, a_scale, gates_x1, ld_gx, cstate1, ldc, h_out1, ld_hout, B, H);
    if (rc) return rc;
    const dim3 grid(a0.tiles_m * a0.tiles_n, 2), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EMAGE_F16X3) hipLaunchKernelGGL((lstm_step_pair_kernel<true>), grid, block, 0, s, a0, a1);
    else hipLaunchKernelGGL((lstm_step_pair_kernel<false>), grid, block, 0, s, a0, a1);
    return launch_status();
}

extern "C" int emage_lstm_step(int dtype, const float* h_prev, int ld_hprev, const void* w_hh, float w_scale, float a_scale,
                               const float* gates_x, int ld_gx, float* cstate, int ldc, float* h_out, int ld_hout,
                               int B, int H, void* stream) {
    {
        GemmArgs chk;
        const int rc = lstm_args(chk, dtype, h_prev, ld_hprev, w_hh, w_scale, a_scale, gates_x, ld_gx, cstate, ldc, h_out, ld_hout, B, H);
        if (rc) return rc;
    }
    GemmArgs a{};
    a.A = h_prev; a.W = w_hh; a.res = gates_x; a.out_f32 = h_out; a.cstate = cstate;
    a.lda = ld_hprev; a.ldr = ld_gx; a.ldf = ld_hout; a.ldc = ldc; a.res_is_f32 = 1;
    a.M = B; a.N = 4 * H; a.K = H; a.Cp = H; a.taps = 1; a.stride = 1; a.pad = 0; a.Lin = B; a.Lout = B;
    a.t_col0 = a.N; a.t_rows = 1;
    a.a_scale = dtype == EMAGE_F16X3 ? a_scale : 1.f;
    a.o_scale = dtype == EMAGE_F16X3 ? 1.f / (a_scale * w_scale) : 1.f;
    a.tiles_m = (B + 63) / 64; a.tiles_n = (4 * H) / 64;
    const dim3 grid(a.tiles_m * a.tiles_n), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == EMAGE_F16X3) hipLaunchKernelGGL((lstm_step_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((lstm_step_kernel<false>), grid, block, 0, s, a);
    return launch_status();
}

extern "C" int emage_softmax2_mix(const float* sel, int ld_sel, const float* c1, int ld1, const float* c2, int ld2,
                                  float* out, int ldo, int M, int C, void* stream) {
    if (!sel || !c1 || !c2 || !out || M <= 0 || C <= 0 || ld_sel < 2 || ld1 < C || ld2 < C || ldo < C) return EMAGE_EINVAL;
    hipLaunchKernelGGL(softmax2_mix_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, sel, ld_sel, c1, ld1, c2, ld2, out, ldo, M, C);
    return launch_status();
}

extern "C" int emage_lstm_inputs(const float* speaker_table, const int64_t* speaker_id, int speaker_f,
                                 const float* seed_motion, long ld_seed_b, int pose_dims, int seed_frames, const int* src_map,
                                 float* out, int ldo, int n_store, int B, int T, void* stream) {
    if (!out || !src_map || B <= 0 || T <= 0 || pose_dims <= 0 || speaker_f < 0 || seed_frames < 0) return EMAGE_EINVAL;
    if ((speaker_f > 0 && (!speaker_table || !speaker_id)) || n_store < speaker_f + pose_dims + 1 || ldo < n_store) return EMAGE_EINVAL;
    hipLaunchKernelGGL(lstm_inputs_kernel, dim3(grid_for((long)B * T * n_store)), dim3(256), 0, (hipStream_t)stream, speaker_table, speaker_id, speaker_f,
                       seed_motion, ld_seed_b, pose_dims, seed_frames, src_map, out, ldo, n_store, B, T);
    return launch_status();
}

extern "C" int emage_rot6d_scatter(const float* rot6d, int ld, const int* slot_of_joint, float* axis_angle, int M, int n_joints, void* stream) {
    if (!rot6d || !slot_of_joint || !axis_angle || M <= 0 || n_joints <= 0) return EMAGE_EINVAL;
    const long total = (long)M * n_joints;
    hipLaunchKernelGGL(rot6d_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rot6d, ld, slot_of_joint, axis_angle, M, n_joints);
    return launch_status();
}

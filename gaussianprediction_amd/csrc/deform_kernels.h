// deform_kernels.h -- device-side argument structs + kernel declarations (definitions in deform_kernels.hip)
#pragma once
#include "gp_common.h"

struct MlpDev {
    long rows;
    int in_dim, in_pad, out_dim;
    int feature_dim, xyz_freq, time_freq;
    const float* w[5];
    const float* b[5];
    const float* feature;
    const float* xyz;
    const float* t;
    const float* pk;        // fragment-ordered copy of w[0..3] for the small-row kernels (gp_mlp_pack), or NULL
};

// Layout of the fragment-ordered weight copy (float4 units).  A wave's operand load of the 16-row kernels is
// (16 features) x (4 lane groups x one float4 of k): from the row-major matrices that is 16 separate 64-byte pieces per
// instruction, and the rate at which ONE CU takes those in bounds a 250-row pass (deform_mlp_small.hip).  Here the float4 of
// (tile t, k-step q, lane) sits at ((t * nq + q) * 64 + lane): one contiguous kilobyte per instruction.
//   forward  layer l:  F_l[t][q][lane] = W_l[16 t + (lane & 15)][16 q + 4 (lane >> 4) + 0..3]            (out^T = W . cur^T)
//   backward layer l:  B_l[t][q][lane] = W_l[16 q + 4 (lane >> 4) + 0..3][16 t + (lane & 15)]            (out^T = W^T . cur^T)
struct MlpPackLayout {
    int q0;                 // k-steps of layer 0's forward: ceil(in_dim / 16)
    int tb0;                // tiles of layer 0's backward: 2 ceil(in_dim / 32)
    long off_f[4], off_b[4], total;      // float4 offsets
};
static inline __host__ __device__ MlpPackLayout mlp_pack_layout(int in_dim) {
    MlpPackLayout L;
    L.q0 = (in_dim + 15) / 16;
    L.tb0 = 2 * ((in_dim + 31) / 32);
    long o = 0;
    L.off_f[0] = o; o += 16L * L.q0 * 64;
    for (int l = 1; l < 4; ++l) { L.off_f[l] = o; o += 16L * 16 * 64; }
    L.off_b[0] = o; o += (long)L.tb0 * 16 * 64;
    for (int l = 1; l < 4; ++l) { L.off_b[l] = o; o += 16L * 16 * 64; }
    L.total = o;
    return L;
}

struct MlpWeightJobs {
    const float* dZ[5];
    const float* H[5];
    float* dW[5];
    float* db[5];
    int n_out[5], n_in[5], ldh[5];
};

struct BlendDev {
    long N, K;
    int nn, out_dim, norm_rotation;
    const float* delta;
    const float* raw_w;
    const int64_t* knn;
    const float* xyz;
    const float* rot;
    const uint16_t* knn16;      // the neighbour indices as 16-bit words, or NULL
};

__global__ __launch_bounds__(256) void gp_mlp_pack_kernel(MlpDev p, float4* __restrict__ out);
__global__ __launch_bounds__(512) void gp_mlp_fwd_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x,
                                                         float* __restrict__ saved_h);
__global__ __launch_bounds__(512) void gp_mlp_bwd_data_kernel(MlpDev p, const float* __restrict__ saved_h,
                                                              const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                              float* __restrict__ dfeature, float* __restrict__ dxyz);
__global__ __launch_bounds__(512) void gp_mlp_bwd_weight_kernel(const float* __restrict__ dZ, int n_out,
                                                                const float* __restrict__ H, int ldh, int n_in, long rows,
                                                                long rows_per_block, float* __restrict__ dW, int lddw,
                                                                float* __restrict__ db);
__global__ __launch_bounds__(256) void gp_blend_fwd_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t);
__global__ __launch_bounds__(256) void gp_blend_fwd6_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t);
__global__ __launch_bounds__(256) void gp_blend_fwd8_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t);
__global__ __launch_bounds__(256) void gp_blend_fwd6_i16_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t);
__global__ __launch_bounds__(256) void gp_blend_fwd8_i16_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t);
__global__ __launch_bounds__(256) void gp_blend_bwd_kernel(BlendDev a, const float* __restrict__ g_xyz_t,
                                                           const float* __restrict__ g_q_t, float* __restrict__ g_delta,
                                                           float* __restrict__ g_raw_w, float* __restrict__ g_xyz,
                                                           float* __restrict__ g_rot, float* __restrict__ partial);
__global__ __launch_bounds__(256) void gp_blend_bwd6_kernel(BlendDev a, const float* __restrict__ g_xyz_t,
                                                           const float* __restrict__ g_q_t, float* __restrict__ g_delta,
                                                           float* __restrict__ g_raw_w, float* __restrict__ g_xyz,
                                                           float* __restrict__ g_rot, float* __restrict__ partial);
__global__ __launch_bounds__(256) void gp_blend_bwd8_kernel(BlendDev a, const float* __restrict__ g_xyz_t,
                                                           const float* __restrict__ g_q_t, float* __restrict__ g_delta,
                                                           float* __restrict__ g_raw_w, float* __restrict__ g_xyz,
                                                           float* __restrict__ g_rot, float* __restrict__ partial);
__global__ __launch_bounds__(256) void gp_blend_bwd6_i16_kernel(BlendDev a, const float* __restrict__ g_xyz_t,
                                                           const float* __restrict__ g_q_t, float* __restrict__ g_delta,
                                                           float* __restrict__ g_raw_w, float* __restrict__ g_xyz,
                                                           float* __restrict__ g_rot, float* __restrict__ partial);
__global__ __launch_bounds__(256) void gp_blend_bwd8_i16_kernel(BlendDev a, const float* __restrict__ g_xyz_t,
                                                           const float* __restrict__ g_q_t, float* __restrict__ g_delta,
                                                           float* __restrict__ g_raw_w, float* __restrict__ g_xyz,
                                                           float* __restrict__ g_rot, float* __restrict__ partial);
__global__ __launch_bounds__(1024) void gp_blend_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int KA,
                                                                  int od, float* __restrict__ g_delta);
__global__ __launch_bounds__(256) void gp_act_fwd_kernel(long n, const float* __restrict__ scaling_raw,
                                                         const float* __restrict__ opacity_raw,
                                                         const float* __restrict__ delta_o, int stride, float beta,
                                                         float* __restrict__ scale, float* __restrict__ opacity);
__global__ __launch_bounds__(256) void gp_act_bwd_kernel(long n, const float* __restrict__ scaling_raw,
                                                         const float* __restrict__ opacity_raw,
                                                         const float* __restrict__ delta_o, int stride, float beta,
                                                         const float* __restrict__ g_scale,
                                                         const float* __restrict__ g_opacity,
                                                         float* __restrict__ g_scaling_raw, float* __restrict__ g_opacity_raw,
                                                         float* __restrict__ g_delta_o);

__global__ __launch_bounds__(512) void gp_mlp_bwd_weight5_kernel(MlpWeightJobs t, long rows);
__global__ __launch_bounds__(512) void gp_mlp_fwd_small_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x,
                                                               float* __restrict__ saved_h);
__global__ __launch_bounds__(512) void gp_mlp_bwd_data_small_kernel(MlpDev p, const float* __restrict__ saved_h,
                                                                    const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                                    float* __restrict__ dfeature, float* __restrict__ dxyz);

__global__ __launch_bounds__(256) void gp_mlp_fwd_split_small_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x, float* hx,
                                                                     uint32_t* flags, uint32_t* err);
__global__ __launch_bounds__(256) void gp_mlp_fwd_split_small_agent_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x,
                                                                           float* hx, uint32_t* flags, uint32_t* err);
__global__ __launch_bounds__(256) void gp_mlp_bwd_data_split_small_kernel(MlpDev p, const float* __restrict__ saved_h, const float* __restrict__ dL_dout,
                                                                          float* dz, float* __restrict__ dfeature, float* __restrict__ dxyz, float* gx,
                                                                          uint32_t* flags, uint32_t* err, int form);
struct AdamTable;      // loss_adam_kernels.h
__global__ __launch_bounds__(512) void gp_mlp_bwd_data_small_adam_kernel(MlpDev p, const float* __restrict__ saved_h,
                                                                         const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                                         float* __restrict__ dfeature, float* __restrict__ dxyz,
                                                                         unsigned n_mlp, AdamTable t, float b1, float b2, float eps,
                                                                         int zero_grad, const uint32_t* __restrict__ skip_flag);

__global__ __launch_bounds__(256) void gp_mlp_bwd_weight64_kernel(const float* __restrict__ dZ, int n_out, const float* __restrict__ H,
                                                                 int ldh, int n_in, long rows, long rows_per_block,
                                                                 unsigned n_row_blocks, float* __restrict__ dW, int lddw,
                                                                 float* __restrict__ db);
__global__ __launch_bounds__(512) void gp_mlp_fwd2_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x,
                                                          float* __restrict__ saved_h, uint32_t* __restrict__ masks);
__global__ __launch_bounds__(512) void gp_mlp_bwd_data2_kernel(MlpDev p, const uint32_t* __restrict__ masks,
                                                               const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                               float* __restrict__ dfeature, float* __restrict__ dxyz);

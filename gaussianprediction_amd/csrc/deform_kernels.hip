// deform_kernels.hip -- the per-frame deformation step of GaussianModel.forward on gfx950:
//   * fused positional-encoding + Deformable_Field MLP, forward and backward, on the exact-fp32
//     matrix cores (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, so RGB parity at 1e-4 holds)
//   * keypoint blend (two softmaxes + SPARSE nn-neighbour gather instead of the reference's dense
//     [N,K] scatter + matmul), quaternion compose / normalise, forward and backward
//   * exp / sigmoid activations with the optional lifecycle-opacity factor
// Reference: scene/gaussian_model.py:180-189, 214-229, 231-304, 314-315; scene/deformable_field.py:63-72,
// 102-127; utils/camera_utils.py:158-170.
//
// MLP data layout.  A workgroup (8 waves) owns 32 rows.  Activations live TRANSPOSED in LDS,
// act[feature][row], XOR-swizzled (row ^ (feature & 31)) so both the coalesced staging writes and the
// MFMA operand reads are bank-conflict free.  Each layer computes H_out^T = W . H_in^T: the weight
// matrix is the MFMA A operand (read straight from L2, nn.Linear's [out,in] row-major layout gives
// each lane one float4 = 4 consecutive k), the activations are the B operand (LDS), and wave w
// produces output features [32w, 32w+32).  No activation ever goes through HBM in inference mode.
#include "gp_common.h"
#include "deform_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_ROWS 32
#define MLP_THREADS 512
#define MLP_W 256

__device__ __forceinline__ int act_idx(int f, int j) { return f * MLP_ROWS + (j ^ (f & 31)); }
// C/D layout of v_mfma_f32_32x32x2_f32: lane l, reg r -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
__device__ __forceinline__ int cd_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Builds X^T (input features) for the 32 rows of this workgroup into `buf` (swizzled), zero-padded to
// in_pad features.  Row layout [feature(fd) | PE(xyz) 6*xf | PE(t) 2*tf]
// [REF scene/gaussian_model.py:180-184; scene/deformable_field.py:63-72: (sin,cos) interleaved per
// (coordinate, frequency), coordinate-major, frequencies 2^j, no pi factor].
__device__ __forceinline__ void build_input(float* buf, const MlpDev& p, long row0, int tid) {
    const int fd = p.feature_dim, xf = p.xyz_freq, tf = p.time_freq;
    for (int e = tid; e < fd * MLP_ROWS; e += MLP_THREADS) {
        const int jj = e / fd, f = e - jj * fd;
        const long row = row0 + jj;
        buf[act_idx(f, jj)] = row < p.rows ? p.feature[row * fd + f] : 0.f;
    }
    for (int e = tid; e < 3 * xf * MLP_ROWS; e += MLP_THREADS) {
        const int jj = e % MLP_ROWS, cf = e / MLP_ROWS;  // cf = c*xf + fr
        const int c = cf / xf, fr = cf - c * xf;
        const long row = row0 + jj;
        float sv = 0.f, cv = 0.f;
        if (row < p.rows) {
            const float a = p.xyz[row * 3 + c] * (float)(1u << fr);
            sincosf(a, &sv, &cv);
        }
        const int f = fd + 2 * cf;
        buf[act_idx(f, jj)] = sv;
        buf[act_idx(f + 1, jj)] = cv;
    }
    const float tv = tf > 0 ? p.t[0] : 0.f;
    for (int e = tid; e < tf * MLP_ROWS; e += MLP_THREADS) {
        const int jj = e % MLP_ROWS, fr = e / MLP_ROWS;
        float sv, cv;
        sincosf(tv * (float)(1u << fr), &sv, &cv);
        const bool ok = row0 + jj < p.rows;
        const int f = fd + 6 * xf + 2 * fr;
        buf[act_idx(f, jj)] = ok ? sv : 0.f;
        buf[act_idx(f + 1, jj)] = ok ? cv : 0.f;
    }
    for (int e = tid; e < (p.in_pad - p.in_dim) * MLP_ROWS; e += MLP_THREADS) {
        const int jj = e % MLP_ROWS, f = p.in_dim + e / MLP_ROWS;
        buf[act_idx(f, jj)] = 0.f;
    }
}

// coalesced copy LDS act^T[0:nf][32] -> global dst[(row0+jj)*ld + f]
__device__ __forceinline__ void store_rows(const float* buf, float* dst, int nf, int ld, long row0, long rows, int tid) {
    for (int e = tid; e < nf * MLP_ROWS; e += MLP_THREADS) {
        const int jj = e / nf, f = e - jj * nf;
        if (row0 + jj < rows) dst[(row0 + jj) * (long)ld + f] = buf[act_idx(f, jj)];
    }
}

// one dense layer on the matrix cores: out^T[32*wave .. +32][32 rows] = W[., 0:K] . cur^T + bias
__device__ __forceinline__ f32x16 layer_mfma(const float* __restrict__ W, int ldw, int K, int out_rows,
                                             const float* __restrict__ bias, const float* cur, int wave, int lane) {
    const int half = lane >> 5, j = lane & 31;
    const int i_row = 32 * wave + j;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 32 * wave + cd_row(r, half);
        acc[r] = (bias && f < out_rows) ? bias[f] : 0.f;
    }
    const bool row_ok = i_row < out_rows;
    const float* wrow = W + (size_t)i_row * ldw;
    const int nkq = K / 8;
    // 4 k-groups per trip: all weight loads are issued before the 16 dependent MFMAs (hides L2 latency)
    for (int kq0 = 0; kq0 < nkq; kq0 += 4) {
        float4 wv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kbase = 8 * (kq0 + g) + 4 * half;
            wv[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok && kq0 + g < nkq && kbase < ldw) wv[g] = *(const float4*)(wrow + kbase);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (kq0 + g < nkq) {
                const int kbase = 8 * (kq0 + g) + 4 * half;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[g].x, cur[act_idx(kbase + 0, j)], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[g].y, cur[act_idx(kbase + 1, j)], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[g].z, cur[act_idx(kbase + 2, j)], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[g].w, cur[act_idx(kbase + 3, j)], acc, 0, 0, 0);
            }
        }
    }
    return acc;
}

// transposed layer for backward: out^T[i][row] = sum_k W[k][i] * cur^T[k][row]  (W is [Kout, ldw])
__device__ __forceinline__ f32x16 layer_mfma_T(const float* __restrict__ W, int ldw, int K, int k_valid, int out_rows,
                                               const float* cur, int wave, int lane) {
    const int half = lane >> 5, j = lane & 31;
    const int i_row = 32 * wave + j;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool row_ok = i_row < out_rows;
    const int nkq = K / 8;
    for (int kq0 = 0; kq0 < nkq; kq0 += 2) {
        float a[8];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 8 * (kq0 + g) + 4 * half + u;
                a[4 * g + u] = (row_ok && kq0 + g < nkq && k < k_valid) ? W[(size_t)k * ldw + i_row] : 0.f;
            }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (kq0 + g < nkq) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 8 * (kq0 + g) + 4 * half + u;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + u], cur[act_idx(k, j)], acc, 0, 0, 0);
                }
            }
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------
// MLP forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MLP_THREADS) void gp_mlp_fwd_kernel(MlpDev p, float* __restrict__ out,
                                                                 float* __restrict__ saved_x, float* __restrict__ saved_h) {
    __shared__ float smem[2][MLP_W * MLP_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const long row0 = (long)blockIdx.x * MLP_ROWS;
    float* cur = smem[0];
    float* nxt = smem[1];
    build_input(cur, p, row0, tid);
    __syncthreads();
    if (saved_x) store_rows(cur, saved_x, p.in_pad, p.in_pad, row0, p.rows, tid);
    for (int l = 0; l < 4; ++l) {
        const int K = l == 0 ? p.in_pad : MLP_W, ldw = l == 0 ? p.in_dim : MLP_W;
        f32x16 acc = layer_mfma(p.w[l], ldw, K, MLP_W, p.b[l], cur, wave, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) nxt[act_idx(32 * wave + cd_row(r, half), j)] = fmaxf(acc[r], 0.f);
        __syncthreads();
        if (saved_h) store_rows(nxt, saved_h + (size_t)l * p.rows * MLP_W, MLP_W, MLP_W, row0, p.rows, tid);
        float* t = cur; cur = nxt; nxt = t;
    }
    // output layer: out_dim <= 8 rows of W4; split K over the 8 waves, reduce through LDS
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool row_ok = j < p.out_dim;
        const float* wrow = p.w[4] + (size_t)j * MLP_W;
        for (int kq = 4 * wave; kq < 4 * wave + 4; ++kq) {
            const int kbase = 8 * kq + 4 * half;
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok) wv = *(const float4*)(wrow + kbase);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, cur[act_idx(kbase + 0, j)], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, cur[act_idx(kbase + 1, j)], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, cur[act_idx(kbase + 2, j)], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, cur[act_idx(kbase + 3, j)], acc, 0, 0, 0);
        }
        // features 0..3 sit in regs 0..3 of half 0, features 4..7 in regs 0..3 of half 1
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[(wave * 8 + r + 4 * half) * MLP_ROWS + j] = acc[r];
        __syncthreads();
        if (tid < 8 * MLP_ROWS) {
            const int jj = tid / 8, f = tid % 8;
            if (f < p.out_dim && row0 + jj < p.rows) {
                float v = p.b[4][f];
#pragma unroll
                for (int w = 0; w < 8; ++w) v += nxt[(w * 8 + f) * MLP_ROWS + jj];
                out[(row0 + jj) * p.out_dim + f] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MLP backward, data chain: dZ_l for l = 4..1 (written to dz[l-1]), dX -> d feature, d xyz
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MLP_THREADS) void gp_mlp_bwd_data_kernel(MlpDev p, const float* __restrict__ saved_h,
                                                                      const float* __restrict__ dL_dout,
                                                                      float* __restrict__ dz, float* __restrict__ dfeature,
                                                                      float* __restrict__ dxyz) {
    __shared__ float smem[2][MLP_W * MLP_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const long row0 = (long)blockIdx.x * MLP_ROWS;
    float* cur = smem[0];
    float* nxt = smem[1];
    // dZ5^T [8][32] (zero-padded)
    for (int e = tid; e < 8 * MLP_ROWS; e += MLP_THREADS) {
        const int jj = e / 8, f = e % 8;
        const long row = row0 + jj;
        cur[act_idx(f, jj)] = (f < p.out_dim && row < p.rows) ? dL_dout[row * p.out_dim + f] : 0.f;
    }
    __syncthreads();
    for (int l = 4; l >= 1; --l) {
        // dH_l^T = W_l^T dZ_{l+1}^T ; W_l = p.w[l] is [Kout, 256]
        const int K = l == 4 ? 8 : MLP_W, kv = l == 4 ? p.out_dim : MLP_W;
        f32x16 acc = layer_mfma_T(p.w[l], MLP_W, K, kv, MLP_W, cur, wave, lane);
        const float* h = saved_h + (size_t)(l - 1) * p.rows * MLP_W;
        const long row = row0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * wave + cd_row(r, half);
            const float hv = row < p.rows ? h[row * MLP_W + f] : 0.f;
            nxt[act_idx(f, j)] = hv > 0.f ? acc[r] : 0.f;
        }
        __syncthreads();
        store_rows(nxt, dz + (size_t)(l - 1) * p.rows * MLP_W, MLP_W, MLP_W, row0, p.rows, tid);
        float* t = cur; cur = nxt; nxt = t;
    }
    // dX^T [in_pad][32] = W_0^T dZ_1^T ; W_0 is [256, in_dim]
    if (dfeature || dxyz) {
        if (wave * 32 < p.in_pad) {
            f32x16 acc = layer_mfma_T(p.w[0], p.in_dim, MLP_W, MLP_W, p.in_dim, cur, wave, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) nxt[act_idx(32 * wave + cd_row(r, half), j)] = acc[r];
        }
        __syncthreads();
        if (dfeature) store_rows(nxt, dfeature, p.feature_dim, p.feature_dim, row0, p.rows, tid);
        if (dxyz) {
            // d/dx sin(x 2^f) = 2^f cos, d/dx cos(x 2^f) = -2^f sin
            if (tid < 3 * MLP_ROWS) {
                const int jj = tid / 3, c = tid % 3;
                const long row = row0 + jj;
                if (row < p.rows) {
                    const float x = p.xyz[row * 3 + c];
                    float g = 0.f;
                    for (int fr = 0; fr < p.xyz_freq; ++fr) {
                        const float sc = (float)(1u << fr);
                        float sv, cv;
                        sincosf(x * sc, &sv, &cv);
                        const int f = p.feature_dim + 2 * (c * p.xyz_freq + fr);
                        g += sc * (cv * nxt[act_idx(f, jj)] - sv * nxt[act_idx(f + 1, jj)]);
                    }
                    dxyz[row * 3 + c] = g;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Large row counts: TWO 32-row tiles per workgroup.  Every workgroup streams all five weight matrices from L2 (0.9 MB);
// at 32 rows per workgroup that is 8 KB per row and layer and the kernels are L2-bound.  With two tiles a weight fragment
// is fetched once and used for both (half the L2 traffic); the activations of both tiles are updated IN PLACE
// (64 KB of LDS, the static limit), which costs one more barrier per layer.
// ------------------------------------------------------------------------------------------------
#define MLP_TILE (MLP_W * MLP_ROWS)

__device__ __forceinline__ void layer_mfma2(const float* __restrict__ W, int ldw, int K, int out_rows, const float* __restrict__ bias,
                                            const float* cur, int wave, int lane, f32x16& acc0, f32x16& acc1) {
    const int half = lane >> 5, j = lane & 31;
    const int i_row = 32 * wave + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 32 * wave + cd_row(r, half);
        acc0[r] = (bias && f < out_rows) ? bias[f] : 0.f;
        acc1[r] = acc0[r];
    }
    const bool row_ok = i_row < out_rows;
    const float* wrow = W + (size_t)i_row * ldw;
    const int nkq = K / 8;
    for (int kq0 = 0; kq0 < nkq; kq0 += 4) {
        float4 wv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kbase = 8 * (kq0 + g) + 4 * half;
            wv[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok && kq0 + g < nkq && kbase < ldw) wv[g] = *(const float4*)(wrow + kbase);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (kq0 + g < nkq) {
                const int kbase = 8 * (kq0 + g) + 4 * half;
                const float wk[4] = {wv[g].x, wv[g].y, wv[g].z, wv[g].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[u], cur[act_idx(kbase + u, j)], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[u], cur[MLP_TILE + act_idx(kbase + u, j)], acc1, 0, 0, 0);
                }
            }
        }
    }
}

__device__ __forceinline__ void layer_mfma_T2(const float* __restrict__ W, int ldw, int K, int k_valid, int out_rows, const float* cur,
                                              int wave, int lane, f32x16& acc0, f32x16& acc1) {
    const int half = lane >> 5, j = lane & 31;
    const int i_row = 32 * wave + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const bool row_ok = i_row < out_rows;
    const int nkq = K / 8;
    for (int kq0 = 0; kq0 < nkq; kq0 += 2) {
        float a[8];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 8 * (kq0 + g) + 4 * half + u;
                a[4 * g + u] = (row_ok && kq0 + g < nkq && k < k_valid) ? W[(size_t)k * ldw + i_row] : 0.f;
            }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (kq0 + g < nkq) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 8 * (kq0 + g) + 4 * half + u;
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + u], cur[act_idx(k, j)], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + u], cur[MLP_TILE + act_idx(k, j)], acc1, 0, 0, 0);
                }
            }
        }
    }
}

__global__ __launch_bounds__(MLP_THREADS) void gp_mlp_fwd2_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x,
                                                                  float* __restrict__ saved_h, uint32_t* __restrict__ masks) {
    __shared__ float cur[2 * MLP_TILE];          // 64 KB: two row tiles, in place
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const long row0 = (long)blockIdx.x * (2 * MLP_ROWS);
    build_input(cur, p, row0, tid);
    build_input(cur + MLP_TILE, p, row0 + MLP_ROWS, tid);
    __syncthreads();
    if (saved_x) {
        store_rows(cur, saved_x, p.in_pad, p.in_pad, row0, p.rows, tid);
        store_rows(cur + MLP_TILE, saved_x, p.in_pad, p.in_pad, row0 + MLP_ROWS, p.rows, tid);
    }
    for (int l = 0; l < 4; ++l) {
        const int K = l == 0 ? p.in_pad : MLP_W, ldw = l == 0 ? p.in_dim : MLP_W;
        f32x16 acc0, acc1;
        layer_mfma2(p.w[l], ldw, K, MLP_W, p.b[l], cur, wave, lane, acc0, acc1);
        __syncthreads();                 // every wave has read its operands
        uint32_t m0 = 0, m1 = 0;         // ReLU sign bits of this lane's 16 features (bit = feature within the wave's tile)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = act_idx(32 * wave + cd_row(r, half), j);
            cur[idx] = fmaxf(acc0[r], 0.f);
            cur[MLP_TILE + idx] = fmaxf(acc1[r], 0.f);
            m0 |= (acc0[r] > 0.f ? 1u : 0u) << cd_row(r, half);
            m1 |= (acc1[r] > 0.f ? 1u : 0u) << cd_row(r, half);
        }
        if (masks) {   // [4][rows][8] words: the backward reads 4 bytes per (row, tile) instead of 128 bytes of activations
            m0 |= (uint32_t)__shfl_xor((int)m0, 32);
            m1 |= (uint32_t)__shfl_xor((int)m1, 32);
            if (half == 0) {
                if (row0 + j < p.rows) masks[((size_t)l * p.rows + row0 + j) * 8 + wave] = m0;
                if (row0 + MLP_ROWS + j < p.rows) masks[((size_t)l * p.rows + row0 + MLP_ROWS + j) * 8 + wave] = m1;
            }
        }
        __syncthreads();
        if (saved_h) {
            store_rows(cur, saved_h + (size_t)l * p.rows * MLP_W, MLP_W, MLP_W, row0, p.rows, tid);
            store_rows(cur + MLP_TILE, saved_h + (size_t)l * p.rows * MLP_W, MLP_W, MLP_W, row0 + MLP_ROWS, p.rows, tid);
        }
    }
    // output layer: out_dim <= 8 rows of W4; K split over the 8 waves; partials leave through the (now free) tile buffer
    {
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        const bool row_ok = j < p.out_dim;
        const float* wrow = p.w[4] + (size_t)j * MLP_W;
        for (int kq = 4 * wave; kq < 4 * wave + 4; ++kq) {
            const int kbase = 8 * kq + 4 * half;
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok) wv = *(const float4*)(wrow + kbase);
            const float wk[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[u], cur[act_idx(kbase + u, j)], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[u], cur[MLP_TILE + act_idx(kbase + u, j)], acc1, 0, 0, 0);
            }
        }
        __syncthreads();                 // all reads of the activations are done: reuse the buffer for the partials
        // features 0..3 sit in regs 0..3 of half 0, features 4..7 in regs 0..3 of half 1
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cur[(wave * 8 + r + 4 * half) * MLP_ROWS + j] = acc0[r];
            cur[MLP_TILE + (wave * 8 + r + 4 * half) * MLP_ROWS + j] = acc1[r];
        }
        __syncthreads();
        if (tid < 2 * 8 * MLP_ROWS) {
            const int t = tid / (8 * MLP_ROWS), e = tid % (8 * MLP_ROWS);
            const int jj = e / 8, f = e % 8;
            const long row = row0 + t * MLP_ROWS + jj;
            if (f < p.out_dim && row < p.rows) {
                float v = p.b[4][f];
#pragma unroll
                for (int w = 0; w < 8; ++w) v += cur[t * MLP_TILE + (w * 8 + f) * MLP_ROWS + jj];
                out[row * p.out_dim + f] = v;
            }
        }
    }
}

__global__ __launch_bounds__(MLP_THREADS) void gp_mlp_bwd_data2_kernel(MlpDev p, const uint32_t* __restrict__ masks,
                                                                       const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                                       float* __restrict__ dfeature, float* __restrict__ dxyz) {
    __shared__ float cur[2 * MLP_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const long row0 = (long)blockIdx.x * (2 * MLP_ROWS);
    // dZ5^T [8][32] per tile (zero-padded)
    for (int e = tid; e < 2 * 8 * MLP_ROWS; e += MLP_THREADS) {
        const int t = e / (8 * MLP_ROWS), ee = e % (8 * MLP_ROWS);
        const int jj = ee / 8, f = ee % 8;
        const long row = row0 + t * MLP_ROWS + jj;
        cur[t * MLP_TILE + act_idx(f, jj)] = (f < p.out_dim && row < p.rows) ? dL_dout[row * p.out_dim + f] : 0.f;
    }
    __syncthreads();
    for (int l = 4; l >= 1; --l) {
        // dH_l^T = W_l^T dZ_{l+1}^T ; W_l = p.w[l] is [Kout, 256]
        const int K = l == 4 ? 8 : MLP_W, kv = l == 4 ? p.out_dim : MLP_W;
        f32x16 acc0, acc1;
        layer_mfma_T2(p.w[l], MLP_W, K, kv, MLP_W, cur, wave, lane, acc0, acc1);
        const uint32_t* mk = masks + (size_t)(l - 1) * p.rows * 8;
        const long ra = row0 + j, rb = row0 + MLP_ROWS + j;
        const uint32_t ma = ra < p.rows ? mk[ra * 8 + wave] : 0u, mb = rb < p.rows ? mk[rb * 8 + wave] : 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * wave + cd_row(r, half);
            cur[act_idx(f, j)] = ((ma >> cd_row(r, half)) & 1u) ? acc0[r] : 0.f;
            cur[MLP_TILE + act_idx(f, j)] = ((mb >> cd_row(r, half)) & 1u) ? acc1[r] : 0.f;
        }
        __syncthreads();
        store_rows(cur, dz + (size_t)(l - 1) * p.rows * MLP_W, MLP_W, MLP_W, row0, p.rows, tid);
        store_rows(cur + MLP_TILE, dz + (size_t)(l - 1) * p.rows * MLP_W, MLP_W, MLP_W, row0 + MLP_ROWS, p.rows, tid);
    }
    // dX^T [in_pad][32] = W_0^T dZ_1^T ; W_0 is [256, in_dim]
    if (dfeature || dxyz) {
        f32x16 acc0, acc1;
        const bool mine = wave * 32 < p.in_pad;
        if (mine) layer_mfma_T2(p.w[0], p.in_dim, MLP_W, MLP_W, p.in_dim, cur, wave, lane, acc0, acc1);
        __syncthreads();
        if (mine) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                cur[act_idx(32 * wave + cd_row(r, half), j)] = acc0[r];
                cur[MLP_TILE + act_idx(32 * wave + cd_row(r, half), j)] = acc1[r];
            }
        }
        __syncthreads();
        for (int t = 0; t < 2; ++t) {
            const float* buf = cur + t * MLP_TILE;
            const long trow0 = row0 + t * MLP_ROWS;
            if (dfeature) store_rows(buf, dfeature, p.feature_dim, p.feature_dim, trow0, p.rows, tid);
            if (dxyz) {
                // d/dx sin(x 2^f) = 2^f cos, d/dx cos(x 2^f) = -2^f sin
                if (tid < 3 * MLP_ROWS) {
                    const int jj = tid / 3, c = tid % 3;
                    const long row = trow0 + jj;
                    if (row < p.rows) {
                        const float x = p.xyz[row * 3 + c];
                        float g = 0.f;
                        for (int fr = 0; fr < p.xyz_freq; ++fr) {
                            const float sc = (float)(1u << fr);
                            float sv, cv;
                            sincosf(x * sc, &sv, &cv);
                            const int f = p.feature_dim + 2 * (c * p.xyz_freq + fr);
                            g += sc * (cv * buf[act_idx(f, jj)] - sv * buf[act_idx(f + 1, jj)]);
                        }
                        dxyz[row * 3 + c] = g;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MLP backward, weight grads: dW[o][i] += sum_rows dZ[row][o] * H[row][i]   (A = dZ^T, B = H)
// grid = (row blocks, i-tiles, o-tiles).  The 8 waves of a workgroup split the block's rows, each
// accumulates a 32x32 tile on the matrix cores (two rows per MFMA, operands straight from global:
// each 32-lane half reads 128 contiguous bytes), the 8 partial tiles are summed through LDS and leave
// as ONE update per element -- a plain read-modify-write when a single row block owns the tile
// (small K: no atomics at all), an atomic only across row blocks.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mlp_bwd_weight_body(const float* __restrict__ dZ, int n_out, const float* __restrict__ H, int ldh,
                                                    int n_in, long rows, long rows_per_block, float* __restrict__ dW, int lddw,
                                                    float* __restrict__ db, int row_block, int n_row_blocks, int bz, int by) {
    __shared__ float s_red[8][16][64];
    __shared__ float s_b[8][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int o = 32 * bz + j;  // A row (output feature)
    const int i = 32 * by + j;  // B col (input feature)
    const long b_begin = (long)row_block * rows_per_block;
    long b_end = b_begin + rows_per_block;
    if (b_end > rows) b_end = rows;
    long per_wave = ((b_end - b_begin + 7) / 8 + 1) & ~1L;  // even, so (rb, rb+1) pairs never straddle slices
    const long r_begin = b_begin + wave * per_wave;
    long r_end = r_begin + per_wave;
    if (r_end > b_end) r_end = b_end;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    const bool o_ok = o < n_out, i_ok = i < n_in;
    for (long rb = r_begin; rb < r_end; rb += 8) {  // uniform trip count per wave: MFMA needs the whole wave
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = rb + 2 * u + half;
            a[u] = (r < r_end && o_ok) ? dZ[r * n_out + o] : 0.f;
            b[u] = (r < r_end && i_ok) ? H[r * (long)ldh + i] : 0.f;
            bsum += a[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[r];
    bsum += __shfl_xor(bsum, 32);
    if (half == 0) s_b[wave][j] = bsum;
    __syncthreads();
    const bool single = n_row_blocks == 1;
    for (int e = tid; e < 16 * 64; e += MLP_THREADS) {
        const int r = e >> 6, l = e & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_red[w][r][l];
        const int oo = 32 * bz + cd_row(r, l >> 5), ii = 32 * by + (l & 31);
        if (oo < n_out && ii < n_in) {
            float* dst = &dW[(size_t)oo * lddw + ii];
            if (single) *dst += v; else atomicAdd(dst, v);
        }
    }
    if (db && by == 0 && tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_b[w][tid];
        const int oo = 32 * bz + tid;
        if (oo < n_out) { if (single) db[oo] += v; else atomicAdd(&db[oo], v); }
    }
}

__global__ __launch_bounds__(MLP_THREADS) void gp_mlp_bwd_weight_kernel(const float* __restrict__ dZ, int n_out,
                                                                        const float* __restrict__ H, int ldh, int n_in,
                                                                        long rows, long rows_per_block,
                                                                        float* __restrict__ dW, int lddw,
                                                                        float* __restrict__ db) {
    mlp_bwd_weight_body(dZ, n_out, H, ldh, n_in, rows, rows_per_block, dW, lddw, db, blockIdx.x, gridDim.x, blockIdx.z, blockIdx.y);
}
// all five layers in ONE launch (small row counts: one row block per tile, no atomics): blockIdx.x = layer
__global__ __launch_bounds__(MLP_THREADS) void gp_mlp_bwd_weight5_kernel(MlpWeightJobs t, long rows) {
    const int l = blockIdx.x;
    if ((int)blockIdx.y * 32 >= t.n_in[l] || (int)blockIdx.z * 32 >= t.n_out[l]) return;
    mlp_bwd_weight_body(t.dZ[l], t.n_out[l], t.H[l], t.ldh[l], t.n_in[l], rows, rows, t.dW[l], t.n_in[l], t.db[l], 0, 1, blockIdx.z,
                        blockIdx.y);
}

// Large row counts: 64 x 64 output block per workgroup (4 waves, each a 2 x 2 arrangement of 32 x 32 tiles over its quarter
// of the block's rows, summed through LDS).  With 32 x 32 blocks every dZ / H element is read by 8 workgroups and the
// kernel is L2-bandwidth bound; 64 x 64 halves that twice.  1-D grid over (row slab, tile pair), see the mapping below.
__global__ __launch_bounds__(256) void gp_mlp_bwd_weight64_kernel(const float* __restrict__ dZ, int n_out, const float* __restrict__ H,
                                                                 int ldh, int n_in, long rows, long rows_per_block,
                                                                 unsigned n_row_blocks, float* __restrict__ dW, int lddw,
                                                                 float* __restrict__ db) {
    __shared__ float s_red[4][4][16][64];   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    // XCD-aware mapping: workgroups go to the 8 XCDs round-robin by flat id, and each XCD has its own L2.  All tile pairs
    // of one row slab get ids that are congruent mod 8, so a slab is fetched into ONE L2 instead of eight.
    const int n_ib = (n_in + 63) / 64;
    const int ntp = n_ib * ((n_out + 63) / 64);
    const unsigned flat = blockIdx.x, xcd = flat & 7u, seq = flat >> 3;
    const unsigned slab = (seq / ntp) * 8u + xcd, tp = seq % ntp;
    if (slab >= n_row_blocks) return;
    const int o0 = 64 * (tp / n_ib), i0 = 64 * (tp % n_ib);
    const long b_begin = (long)slab * rows_per_block;
    long b_end = b_begin + rows_per_block;
    if (b_end > rows) b_end = rows;
    const long per_wave = ((b_end - b_begin + 3) / 4 + 7) & ~7L;
    const long r_begin = b_begin + wave * per_wave;
    long r_end = r_begin + per_wave;
    if (r_end > b_end) r_end = b_end;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[2] = {0.f, 0.f};
    const int oa = o0 + j, ob = o0 + 32 + j, ia = i0 + j, ib = i0 + 32 + j;
    const bool oka = oa < n_out, okb = ob < n_out, ika = ia < n_in, ikb = ib < n_in;
    for (long rb = r_begin; rb < r_end; rb += 8) {   // uniform trip count per wave
        float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = rb + 2 * u + half;
            const bool rk = r < r_end;
            a0[u] = (rk && oka) ? dZ[r * n_out + oa] : 0.f;
            a1[u] = (rk && okb) ? dZ[r * n_out + ob] : 0.f;
            b0[u] = (rk && ika) ? H[r * (long)ldh + ia] : 0.f;
            b1[u] = (rk && ikb) ? H[r * (long)ldh + ib] : 0.f;
            bsum[0] += a0[u]; bsum[1] += a1[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[wave][2 * a + b][r][lane] = acc[a][b][r];
    __syncthreads();
    const bool single = n_row_blocks == 1;
    for (int e = tid; e < 4 * 16 * 64; e += 256) {
        const int tl = e >> 10, r = (e >> 6) & 15, l = e & 63;
        const float v = (s_red[0][tl][r][l] + s_red[1][tl][r][l]) + (s_red[2][tl][r][l] + s_red[3][tl][r][l]);
        const int oo = o0 + 32 * (tl >> 1) + cd_row(r, l >> 5), ii = i0 + 32 * (tl & 1) + (l & 31);
        if (oo < n_out && ii < n_in) {
            float* dst = &dW[(size_t)oo * lddw + ii];
            if (single) *dst += v; else atomicAdd(dst, v);
        }
    }
    if (db && i0 == 0) {
        __syncthreads();
        float* sb = &s_red[0][0][0][0];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float v = bsum[u] + __shfl_xor(bsum[u], 32);
            if (half == 0) sb[(wave * 2 + u) * 32 + j] = v;
        }
        __syncthreads();
        if (tid < 64) {
            const int u = tid >> 5, jj = tid & 31;
            const float v = (sb[(0 * 2 + u) * 32 + jj] + sb[(1 * 2 + u) * 32 + jj]) + (sb[(2 * 2 + u) * 32 + jj] + sb[(3 * 2 + u) * 32 + jj]);
            const int oo = o0 + 32 * u + jj;
            if (oo < n_out) { if (single) db[oo] += v; else atomicAdd(&db[oo], v); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// keypoint blend + pose composition
// ------------------------------------------------------------------------------------------------
#define GP_MAX_NN 16

__device__ __forceinline__ void quat_mul(const float* q, const float* r, float* p) {  // p = q (x) r, (w,x,y,z)
    p[0] = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    p[1] = r[1] * q[0] + r[0] * q[1] + r[3] * q[2] - r[2] * q[3];
    p[2] = r[2] * q[0] - r[3] * q[1] + r[0] * q[2] + r[1] * q[3];
    p[3] = r[3] * q[0] + r[2] * q[1] - r[1] * q[2] + r[0] * q[3];
}
__device__ __forceinline__ float norm4(const float* v) {
    return fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]), 1e-12f);  // F.normalize eps
}
__device__ __forceinline__ void softmax_n(const float* __restrict__ raw, int n, float* w) {
    float m = raw[0];
#pragma unroll
    for (int k = 1; k < n; ++k) m = fmaxf(m, raw[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < n; ++k) { w[k] = expf(raw[k] - m); s += w[k]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int k = 0; k < n; ++k) w[k] *= inv;
}

// one Gaussian's 2*NN raw weights and NN neighbour ids with 16-byte loads (rows are 48 / 64 B: aligned whenever the tensors are)
template <int NN, bool I16>
__device__ __forceinline__ void load_row_nn(const BlendDev& a, long i, float (&raw)[2 * NN], int (&kps)[NN]) {
    const float4* r4 = (const float4*)(a.raw_w + i * 2 * NN);
#pragma unroll
    for (int u = 0; u < 2 * NN / 4; ++u) { const float4 t = r4[u]; raw[4 * u] = t.x; raw[4 * u + 1] = t.y; raw[4 * u + 2] = t.z; raw[4 * u + 3] = t.w; }
    if (I16) {                // packed copy: NN / 2 dwords per row instead of NN quadwords (36 of ~170 B per Gaussian at nn = 6).  A
        const uint32_t* k1 = (const uint32_t*)(a.knn16 + i * NN);      // kernel VARIANT, not a branch: behind `if (a.knn16)` the row's
#pragma unroll                                                          // loads were waited for one group at a time
        for (int u = 0; u < NN / 2; ++u) { const uint32_t t = k1[u]; kps[2 * u] = (int)(t & 0xFFFFu); kps[2 * u + 1] = (int)(t >> 16); }
    } else {
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        const ll2* k2 = (const ll2*)(a.knn + i * NN);
#pragma unroll
        for (int u = 0; u < NN / 2; ++u) { const ll2 t = k2[u]; kps[2 * u] = (int)t.x; kps[2 * u + 1] = (int)t.y; }
    }
}

// NN > 0: compile-time neighbour count (loops unroll, the nn gathers are issued together);
// NN == 0: run-time a.nn (including the stage-1 case a.nn == 0)
template <int NN, bool I16 = false>
__device__ __forceinline__ void blend_fwd_body(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    float dxyz[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    const int od = a.out_dim;
    const int nn = NN > 0 ? NN : a.nn;
    const float x0[3] = {a.xyz[3 * i], a.xyz[3 * i + 1], a.xyz[3 * i + 2]};          // (issued with the row loads, used at the end)
    const float r[4] = {a.rot[4 * i], a.rot[4 * i + 1], a.rot[4 * i + 2], a.rot[4 * i + 3]};
    if (nn > 0) {
        float wx[NN > 0 ? NN : GP_MAX_NN], wr[NN > 0 ? NN : GP_MAX_NN];
        int kpv[NN > 0 ? NN : 1];
        if constexpr (NN > 0) {
            float raw[2 * (NN > 0 ? NN : 1)];
            load_row_nn<NN, I16>(a, i, raw, kpv);
            softmax_n(raw, NN, wx);
            softmax_n(raw + NN, NN, wr);
        } else {
            softmax_n(a.raw_w + i * 2 * nn, nn, wx);
            softmax_n(a.raw_w + i * 2 * nn + nn, nn, wr);
        }
        if constexpr (NN > 0) {
            // the NN keypoint rows gathered first, all in flight (inside the blend loop every gather was waited for on its own:
            // NN dependent trips to L2 per wave), then the blend in the same order
            float dlr[NN > 0 ? NN : 1][7];
#pragma unroll
            for (int k = 0; k < NN; ++k) {
                const float* dl = a.delta + (long)kpv[k] * od;
#pragma unroll
                for (int c = 0; c < 7; ++c) dlr[k][c] = dl[c];
            }
#pragma unroll
            for (int k = 0; k < NN; ++k) {
                dxyz[0] = fmaf(wx[k], dlr[k][0], dxyz[0]);
                dxyz[1] = fmaf(wx[k], dlr[k][1], dxyz[1]);
                dxyz[2] = fmaf(wx[k], dlr[k][2], dxyz[2]);
                float v[4] = {dlr[k][3], dlr[k][4], dlr[k][5], dlr[k][6]};
                if (a.norm_rotation) { const float inv = 1.f / norm4(v); v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv; }
                dq[0] = fmaf(wr[k], v[0], dq[0]); dq[1] = fmaf(wr[k], v[1], dq[1]);
                dq[2] = fmaf(wr[k], v[2], dq[2]); dq[3] = fmaf(wr[k], v[3], dq[3]);
            }
        } else {
            for (int k = 0; k < nn; ++k) {
                const long kp = a.knn16 ? (long)a.knn16[i * nn + k] : a.knn[i * nn + k];
                const float* dl = a.delta + kp * od;
                dxyz[0] = fmaf(wx[k], dl[0], dxyz[0]);
                dxyz[1] = fmaf(wx[k], dl[1], dxyz[1]);
                dxyz[2] = fmaf(wx[k], dl[2], dxyz[2]);
                float v[4] = {dl[3], dl[4], dl[5], dl[6]};
                if (a.norm_rotation) { const float inv = 1.f / norm4(v); v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv; }
                dq[0] = fmaf(wr[k], v[0], dq[0]); dq[1] = fmaf(wr[k], v[1], dq[1]);
                dq[2] = fmaf(wr[k], v[2], dq[2]); dq[3] = fmaf(wr[k], v[3], dq[3]);
            }
        }
    } else {
        const float* dl = a.delta + i * od;
        dxyz[0] = dl[0]; dxyz[1] = dl[1]; dxyz[2] = dl[2];
        dq[0] = dl[3]; dq[1] = dl[4]; dq[2] = dl[5]; dq[3] = dl[6];
        if (a.norm_rotation) { const float inv = 1.f / norm4(dq); dq[0] *= inv; dq[1] *= inv; dq[2] *= inv; dq[3] *= inv; }
    }
    xyz_t[3 * i] = x0[0] + dxyz[0];
    xyz_t[3 * i + 1] = x0[1] + dxyz[1];
    xyz_t[3 * i + 2] = x0[2] + dxyz[2];
    const float invq = 1.f / norm4(dq);
    const float q[4] = {dq[0] * invq, dq[1] * invq, dq[2] * invq, dq[3] * invq};
    float pq[4];
    quat_mul(q, r, pq);
    const float invp = 1.f / norm4(pq);
    q_t[4 * i] = pq[0] * invp; q_t[4 * i + 1] = pq[1] * invp; q_t[4 * i + 2] = pq[2] * invp; q_t[4 * i + 3] = pq[3] * invp;
}
__global__ __launch_bounds__(256) void gp_blend_fwd_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t) { blend_fwd_body<0>(a, xyz_t, q_t); }
__global__ __launch_bounds__(256) void gp_blend_fwd6_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t) { blend_fwd_body<6>(a, xyz_t, q_t); }
__global__ __launch_bounds__(256) void gp_blend_fwd8_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t) { blend_fwd_body<8>(a, xyz_t, q_t); }
__global__ __launch_bounds__(256) void gp_blend_fwd6_i16_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t) { blend_fwd_body<6, true>(a, xyz_t, q_t); }
__global__ __launch_bounds__(256) void gp_blend_fwd8_i16_kernel(BlendDev a, float* __restrict__ xyz_t, float* __restrict__ q_t) { blend_fwd_body<8, true>(a, xyz_t, q_t); }

// gradient through y = v / max(|v|, eps):  dv = (g - y (y.g)) / |v|
__device__ __forceinline__ void normalize_bwd(const float* v, const float* g, float* dv) {
    const float n = norm4(v), inv = 1.f / n;
    const float y[4] = {v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
    const float dot = y[0] * g[0] + y[1] * g[1] + y[2] * g[2] + y[3] * g[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) dv[k] = (g[k] - y[k] * dot) * inv;
}

// Keypoint gradients  dDelta[kp] = sum_i w_i,kp g_i  are a sparse-transpose product.  LDS float atomics run at
// about one lane per clock, so instead of 7 atomics per (Gaussian, neighbour) each 256-Gaussian chunk is
// COUNTING-SORTED by keypoint in LDS (integer atomics only, for the ranks) and every keypoint's owner thread sums
// its own contiguous list into a plain LDS accumulator: no floating-point atomics, deterministic per workgroup.
// The normalisation Jacobian of a keypoint's quaternion is linear in the incoming gradient, so it is applied once
// per (workgroup, keypoint) after the sum.  Workgroup partials go to `partial`; gp_blend_bwd_reduce_kernel adds them.
//
// dynamic LDS: acc[K*7] | delta[K*od] | cnt[K] | base[K+1] | g[7*256] | inv[K] | w[256*2*nn] | sorted u16 [256*nn]
#define BB_LONG 20          // a keypoint's list beyond this length is summed by a wave (mean length = nn)
#define BB_LONG_CAP 320     // >= 256 * GP_MAX_NN / (BB_LONG + 1) lists can be that long
template <int NN, bool I16 = false>
__device__ __forceinline__ void blend_bwd_body(BlendDev a, const float* __restrict__ g_xyz_t,
                                               const float* __restrict__ g_q_t, float* __restrict__ g_delta,
                                               float* __restrict__ g_raw_w, float* __restrict__ g_xyz,
                                               float* __restrict__ g_rot, float* __restrict__ partial) {
    extern __shared__ float s_dyn[];
    const int od = a.out_dim;
    const int nn = NN > 0 ? NN : a.nn;
    const int K = (int)a.K;
    const int tid = threadIdx.x;
    float* s_acc = s_dyn;                               // [K*7]
    float* s_delta = s_acc + (nn > 0 ? K * 7 : 0);      // [K*od]
    int* s_cnt = (int*)(s_delta + (nn > 0 ? K * od : 0));   // [K]
    int* s_base = s_cnt + (nn > 0 ? K : 0);             // [K+1]
    float* s_g = (float*)(s_base + (nn > 0 ? K + 1 : 0));   // [7][256]   component-major: a keypoint's owner gathers RANDOM Gaussians of the
    float* s_inv = s_g + 256 * 7;                       // [K] 1 / |keypoint quaternion| (norm_rotation)
    float* s_w = s_inv + (nn > 0 ? K : 0);              // [2*nn][256] chunk; [256][8] / [256][12] put 64 lanes on 4 / 8 banks (measured: 71 % of the LDS cycles were conflicts)
    unsigned short* s_sorted = (unsigned short*)(s_w + 256 * 2 * nn);   // [256*nn]
    __shared__ int s_wsum[4];
    __shared__ int s_nlong, s_long[BB_LONG_CAP];
    if (nn > 0) {
        for (int e = tid; e < K * 7; e += 256) s_acc[e] = 0.f;
        for (int e = tid; e < K * od; e += 256) s_delta[e] = a.delta[e];
        if (a.norm_rotation) {
            // the keypoints' quaternions are normalised ONCE per workgroup, in place (y = v / |v|, 1 / |v| beside it): per (Gaussian,
            // neighbour) that was a square root and a division, twice -- a third of the per-Gaussian arithmetic.  Same expression
            // per keypoint as before, so the values are the same.
            __syncthreads();
            for (int kp = tid; kp < K; kp += 256) {
                float* dl = s_delta + kp * od;
                const float v[4] = {dl[3], dl[4], dl[5], dl[6]};
                const float inv = 1.f / norm4(v);
                s_inv[kp] = inv;
                dl[3] = v[0] * inv; dl[4] = v[1] * inv; dl[5] = v[2] * inv; dl[6] = v[3] * inv;
            }
        }
    }
    const long chunks = (a.N + 255) / 256;
    for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const long i = c * 256 + tid;
        const bool live = i < a.N;
        if (nn > 0) {
            for (int e = tid; e < K; e += 256) s_cnt[e] = 0;
            if (tid == 0) s_nlong = 0;
            __syncthreads();
        }
        int kps[NN > 0 ? NN : GP_MAX_NN], rk[NN > 0 ? NN : GP_MAX_NN];
        if (live) {
            const float gx[3] = {g_xyz_t[3 * i], g_xyz_t[3 * i + 1], g_xyz_t[3 * i + 2]};
            const float gq[4] = {g_q_t[4 * i], g_q_t[4 * i + 1], g_q_t[4 * i + 2], g_q_t[4 * i + 3]};
            const float r[4] = {a.rot[4 * i], a.rot[4 * i + 1], a.rot[4 * i + 2], a.rot[4 * i + 3]};
            // (every global load of the chunk is issued before the first global store: a store through g_xyz may alias the
            // rows read through `a` as far as the compiler knows, and used to split the loads into two dependent round trips)
            // recompute forward
            float wx[NN > 0 ? NN : GP_MAX_NN], wr[NN > 0 ? NN : GP_MAX_NN];
            float dq[4] = {0.f, 0.f, 0.f, 0.f};
            if (nn > 0) {
                if constexpr (NN > 0) {
                    float raw[2 * (NN > 0 ? NN : 1)];
                    load_row_nn<NN, I16>(a, i, raw, kps);
                    softmax_n(raw, NN, wx);
                    softmax_n(raw + NN, NN, wr);
                } else {
                    softmax_n(a.raw_w + i * 2 * nn, nn, wx);
                    softmax_n(a.raw_w + i * 2 * nn + nn, nn, wr);
                }
#pragma unroll
                for (int k = 0; k < nn; ++k) {
                    if (NN == 0) kps[k] = a.knn16 ? (int)a.knn16[i * nn + k] : (int)a.knn[i * nn + k];
                    const float* dl = s_delta + kps[k] * od;
                    const float v[4] = {dl[3], dl[4], dl[5], dl[6]};          // (normalised in LDS when norm_rotation)
                    dq[0] = fmaf(wr[k], v[0], dq[0]); dq[1] = fmaf(wr[k], v[1], dq[1]);
                    dq[2] = fmaf(wr[k], v[2], dq[2]); dq[3] = fmaf(wr[k], v[3], dq[3]);
                }
            } else {
                const float* dl = a.delta + i * od;
                dq[0] = dl[3]; dq[1] = dl[4]; dq[2] = dl[5]; dq[3] = dl[6];
                if (a.norm_rotation) { const float inv = 1.f / norm4(dq); dq[0] *= inv; dq[1] *= inv; dq[2] *= inv; dq[3] *= inv; }
            }
            const float invq = 1.f / norm4(dq);
            const float q[4] = {dq[0] * invq, dq[1] * invq, dq[2] * invq, dq[3] * invq};
            g_xyz[3 * i] = gx[0]; g_xyz[3 * i + 1] = gx[1]; g_xyz[3 * i + 2] = gx[2];
            float pq[4];
            quat_mul(q, r, pq);
            float gp[4];
            normalize_bwd(pq, gq, gp);
            // p = q (x) r : bilinear
            const float gqn[4] = {gp[0] * r[0] + gp[1] * r[1] + gp[2] * r[2] + gp[3] * r[3],
                                  -gp[0] * r[1] + gp[1] * r[0] - gp[2] * r[3] + gp[3] * r[2],
                                  -gp[0] * r[2] + gp[1] * r[3] + gp[2] * r[0] - gp[3] * r[1],
                                  -gp[0] * r[3] - gp[1] * r[2] + gp[2] * r[1] + gp[3] * r[0]};
            g_rot[4 * i + 0] = gp[0] * q[0] + gp[1] * q[1] + gp[2] * q[2] + gp[3] * q[3];
            g_rot[4 * i + 1] = -gp[0] * q[1] + gp[1] * q[0] + gp[2] * q[3] - gp[3] * q[2];
            g_rot[4 * i + 2] = -gp[0] * q[2] - gp[1] * q[3] + gp[2] * q[0] + gp[3] * q[1];
            g_rot[4 * i + 3] = -gp[0] * q[3] + gp[1] * q[2] - gp[2] * q[1] + gp[3] * q[0];
            float gdq[4];
            normalize_bwd(dq, gqn, gdq);  // grad wrt blended (or per-Gaussian normalised) dq
            if (nn > 0) {
                float gwx[NN > 0 ? NN : GP_MAX_NN], gwr[NN > 0 ? NN : GP_MAX_NN];
                float sx = 0.f, sr = 0.f;
#pragma unroll
                for (int k = 0; k < nn; ++k) {
                    const float* dl = s_delta + kps[k] * od;
                    gwx[k] = dl[0] * gx[0] + dl[1] * gx[1] + dl[2] * gx[2];
                    const float vn[4] = {dl[3], dl[4], dl[5], dl[6]};
                    gwr[k] = vn[0] * gdq[0] + vn[1] * gdq[1] + vn[2] * gdq[2] + vn[3] * gdq[3];
                    sx += wx[k] * gwx[k];
                    sr += wr[k] * gwr[k];
                    s_w[k * 256 + tid] = wx[k];
                    s_w[(nn + k) * 256 + tid] = wr[k];
                    rk[k] = atomicAdd(&s_cnt[kps[k]], 1);          // rank of this entry within its keypoint
                }
                if (g_raw_w) {        // (NULL: the weights are inputs without a gradient, 8 nn bytes per Gaussian not written)
                    if constexpr (NN > 0) {
                        float go[2 * (NN > 0 ? NN : 1)];
#pragma unroll
                        for (int k = 0; k < NN; ++k) { go[k] = wx[k] * (gwx[k] - sx); go[NN + k] = wr[k] * (gwr[k] - sr); }
                        float4* o4 = (float4*)(g_raw_w + i * 2 * NN);
#pragma unroll
                        for (int u = 0; u < 2 * NN / 4; ++u) o4[u] = make_float4(go[4 * u], go[4 * u + 1], go[4 * u + 2], go[4 * u + 3]);
                    } else {
                        for (int k = 0; k < nn; ++k) {
                            g_raw_w[i * 2 * nn + k] = wx[k] * (gwx[k] - sx);
                            g_raw_w[i * 2 * nn + nn + k] = wr[k] * (gwr[k] - sr);
                        }
                    }
                }
                s_g[0 * 256 + tid] = gx[0]; s_g[1 * 256 + tid] = gx[1]; s_g[2 * 256 + tid] = gx[2];
                s_g[3 * 256 + tid] = gdq[0]; s_g[4 * 256 + tid] = gdq[1]; s_g[5 * 256 + tid] = gdq[2]; s_g[6 * 256 + tid] = gdq[3];
            } else {
                const float* dl = a.delta + i * od;
                float gv[4] = {gdq[0], gdq[1], gdq[2], gdq[3]};
                // here dq is already the normalised per-Gaussian delta when norm_rotation: chain once more
                if (a.norm_rotation) { const float v[4] = {dl[3], dl[4], dl[5], dl[6]}; normalize_bwd(v, gdq, gv); }
                float* gd = g_delta + i * od;
                gd[0] = gx[0]; gd[1] = gx[1]; gd[2] = gx[2];
                gd[3] = gv[0]; gd[4] = gv[1]; gd[5] = gv[2]; gd[6] = gv[3];
                for (int k = 7; k < od; ++k) gd[k] = 0.f;
            }
        }
        if (nn > 0) {
            __syncthreads();
            {   // exclusive scan of cnt[0..K) -> base[0..K]: each thread owns ceil(K/256) consecutive bins
                const int per = (K + 255) / 256;
                const int b0 = tid * per;
                int loc = 0;
                for (int e = 0; e < per; ++e) if (b0 + e < K) loc += s_cnt[b0 + e];
                int inc = loc;
                inc = gp_wave_scan_add(inc);
                if ((tid & 63) == 63) s_wsum[tid >> 6] = inc;
                __syncthreads();
                int woff = 0;
                for (int w = 0; w < (tid >> 6); ++w) woff += s_wsum[w];
                int run = woff + inc - loc;
                for (int e = 0; e < per; ++e) if (b0 + e < K) { s_base[b0 + e] = run; run += s_cnt[b0 + e]; }
                if (tid == 255) s_base[K] = run;
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int k = 0; k < nn; ++k) s_sorted[s_base[kps[k]] + rk[k]] = (unsigned short)((k << 8) | tid);
            }
            __syncthreads();
            for (int kp = tid; kp < K; kp += 256) {
                float sacc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const int pe = s_base[kp + 1];
                if (pe - s_base[kp] > BB_LONG) { s_long[atomicAdd(&s_nlong, 1)] = kp; continue; }   // summed by a whole wave below
                for (int pp = s_base[kp]; pp < pe; pp += 2) {          // two entries per round: their gathers overlap (summed in list order)
                    const int e0 = s_sorted[pp], e1 = pp + 1 < pe ? s_sorted[pp + 1] : -1;
                    const int t0 = e0 & 255, k0 = e0 >> 8, t1 = e1 & 255, k1 = (e1 >> 8) & 15;
                    const float wx0 = s_w[k0 * 256 + t0], wr0 = s_w[(nn + k0) * 256 + t0];
                    const float wx1 = e1 >= 0 ? s_w[k1 * 256 + t1] : 0.f, wr1 = e1 >= 0 ? s_w[(nn + k1) * 256 + t1] : 0.f;
                    float g0[7], g1[7];
#pragma unroll
                    for (int cc = 0; cc < 7; ++cc) { g0[cc] = s_g[cc * 256 + t0]; g1[cc] = s_g[cc * 256 + t1]; }
                    sacc[0] = fmaf(wx0, g0[0], sacc[0]); sacc[1] = fmaf(wx0, g0[1], sacc[1]); sacc[2] = fmaf(wx0, g0[2], sacc[2]);
                    sacc[3] = fmaf(wr0, g0[3], sacc[3]); sacc[4] = fmaf(wr0, g0[4], sacc[4]);
                    sacc[5] = fmaf(wr0, g0[5], sacc[5]); sacc[6] = fmaf(wr0, g0[6], sacc[6]);
                    if (e1 >= 0) {
                        sacc[0] = fmaf(wx1, g1[0], sacc[0]); sacc[1] = fmaf(wx1, g1[1], sacc[1]); sacc[2] = fmaf(wx1, g1[2], sacc[2]);
                        sacc[3] = fmaf(wr1, g1[3], sacc[3]); sacc[4] = fmaf(wr1, g1[4], sacc[4]);
                        sacc[5] = fmaf(wr1, g1[5], sacc[5]); sacc[6] = fmaf(wr1, g1[6], sacc[6]);
                    }
                }
                float* acc = s_acc + kp * 7;
#pragma unroll
                for (int cc = 0; cc < 7; ++cc) acc[cc] += sacc[cc];
            }
            __syncthreads();
            // Long lists (spatially coherent storage order -- a densified or sorted cloud -- puts most of a chunk's 256 x nn entries
            // on a dozen keypoints: their owner threads would walk hundreds of entries while the rest of the workgroup idles;
            // the bench scene stored along a Morton curve: 0.09 -> 0.21 ms).  A wave strides over such a list and reduces.
            const int nlong = s_nlong;
            for (int q = tid >> 6; q < nlong; q += 4) {
                const int kp = s_long[q], pe = s_base[kp + 1];
                float sacc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int pp = s_base[kp] + (tid & 63); pp < pe; pp += 64) {
                    const int e = s_sorted[pp], t = e & 255, k = e >> 8;
                    const float wxk = s_w[k * 256 + t], wrk = s_w[(nn + k) * 256 + t];
                    sacc[0] = fmaf(wxk, s_g[0 * 256 + t], sacc[0]); sacc[1] = fmaf(wxk, s_g[1 * 256 + t], sacc[1]);
                    sacc[2] = fmaf(wxk, s_g[2 * 256 + t], sacc[2]); sacc[3] = fmaf(wrk, s_g[3 * 256 + t], sacc[3]);
                    sacc[4] = fmaf(wrk, s_g[4 * 256 + t], sacc[4]); sacc[5] = fmaf(wrk, s_g[5 * 256 + t], sacc[5]);
                    sacc[6] = fmaf(wrk, s_g[6 * 256 + t], sacc[6]);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                    for (int cc = 0; cc < 7; ++cc) sacc[cc] += __shfl_xor(sacc[cc], off);
                if ((tid & 63) == 0) {
                    float* acc = s_acc + kp * 7;
#pragma unroll
                    for (int cc = 0; cc < 7; ++cc) acc[cc] += sacc[cc];
                }
            }
            if (nlong > 0) __syncthreads();                  // (uniform) the next chunk re-uses s_sorted / s_w / s_g
        }
    }
    if (nn > 0) {
        // stage 1 of the cross-workgroup reduction: plain coalesced stores of this workgroup's partials
        const int KA = K * 7;
        for (int kp = tid; kp < K; kp += 256) {
            float* acc = s_acc + kp * 7;
            if (a.norm_rotation) {      // normalize_bwd with y and 1 / |v| as stored at the top
                const float* y = s_delta + kp * od + 3;
                const float inv = s_inv[kp];
                const float cq[4] = {acc[3], acc[4], acc[5], acc[6]};
                const float dot = y[0] * cq[0] + y[1] * cq[1] + y[2] * cq[2] + y[3] * cq[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[3 + k] = (cq[k] - y[k] * dot) * inv;
            }
        }
        __syncthreads();
        for (int e = tid; e < KA; e += 256) partial[(size_t)blockIdx.x * KA + e] = s_acc[e];
    }
}

#define BB_ARGS BlendDev a, const float* __restrict__ g_xyz_t, const float* __restrict__ g_q_t, float* __restrict__ g_delta, \
    float* __restrict__ g_raw_w, float* __restrict__ g_xyz, float* __restrict__ g_rot, float* __restrict__ partial
__global__ __launch_bounds__(256) void gp_blend_bwd_kernel(BB_ARGS) { blend_bwd_body<0>(a, g_xyz_t, g_q_t, g_delta, g_raw_w, g_xyz, g_rot, partial); }
__global__ __launch_bounds__(256) void gp_blend_bwd6_kernel(BB_ARGS) { blend_bwd_body<6>(a, g_xyz_t, g_q_t, g_delta, g_raw_w, g_xyz, g_rot, partial); }
__global__ __launch_bounds__(256) void gp_blend_bwd8_kernel(BB_ARGS) { blend_bwd_body<8>(a, g_xyz_t, g_q_t, g_delta, g_raw_w, g_xyz, g_rot, partial); }
__global__ __launch_bounds__(256) void gp_blend_bwd6_i16_kernel(BB_ARGS) { blend_bwd_body<6, true>(a, g_xyz_t, g_q_t, g_delta, g_raw_w, g_xyz, g_rot, partial); }
__global__ __launch_bounds__(256) void gp_blend_bwd8_i16_kernel(BB_ARGS) { blend_bwd_body<8, true>(a, g_xyz_t, g_q_t, g_delta, g_raw_w, g_xyz, g_rot, partial); }

// stage 2: g_delta[k, c] = sum over workgroups (deterministic, no atomics).  64 elements per workgroup, the
// workgroup's 16 waves split the partials (four independent sums each) and meet in LDS.
__global__ __launch_bounds__(1024) void gp_blend_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int KA,
                                                                  int od, float* __restrict__ g_delta) {
    __shared__ float s_r[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (e < KA) {
        int b = wave;
        for (; b + 240 < nblocks; b += 256) {         // sixteen partials per trip to memory, added in the order of the loop below
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = partial[(size_t)(b + 16 * u) * KA + e];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { v0 += t[u]; v1 += t[u + 1]; v2 += t[u + 2]; v3 += t[u + 3]; }
        }
        for (; b + 48 < nblocks; b += 64) {
            v0 += partial[(size_t)b * KA + e]; v1 += partial[(size_t)(b + 16) * KA + e];
            v2 += partial[(size_t)(b + 32) * KA + e]; v3 += partial[(size_t)(b + 48) * KA + e];
        }
        for (; b < nblocks; b += 16) v0 += partial[(size_t)b * KA + e];
    }
    s_r[wave][lane] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (wave == 0 && e < KA) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_r[w][lane];
        g_delta[(size_t)(e / 7) * od + (e % 7)] = t;
        if (e % 7 == 6) for (int c = 7; c < od; ++c) g_delta[(size_t)(e / 7) * od + c] = 0.f;   // (the whole row is written: no zero-fill by the caller)
    }
}

// ------------------------------------------------------------------------------------------------
// activations
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gp_act_fwd_kernel(long n, const float* __restrict__ scaling_raw,
                                                        const float* __restrict__ opacity_raw,
                                                        const float* __restrict__ delta_o, int stride, float beta,
                                                        float* __restrict__ scale, float* __restrict__ opacity) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s0 = scaling_raw[3 * i], s1 = scaling_raw[3 * i + 1], s2 = scaling_raw[3 * i + 2];   // (all loads first)
    const float op_raw = opacity_raw[i];
    const float dlt = delta_o ? delta_o[i * stride] : 0.f;
    scale[3 * i] = expf(s0);
    scale[3 * i + 1] = expf(s1);
    scale[3 * i + 2] = expf(s2);
    float o = sigmoidf(op_raw);
    if (delta_o) o *= 1.f / (1.f + expf(-dlt / beta));  // sharp_sigmoid [REF gaussian_model.py:51]
    opacity[i] = o;
}
__global__ __launch_bounds__(256) void gp_act_bwd_kernel(long n, const float* __restrict__ scaling_raw,
                                                        const float* __restrict__ opacity_raw,
                                                        const float* __restrict__ delta_o, int stride, float beta,
                                                        const float* __restrict__ g_scale, const float* __restrict__ g_opacity,
                                                        float* __restrict__ g_scaling_raw, float* __restrict__ g_opacity_raw,
                                                        float* __restrict__ g_delta_o) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // every load first (interleaved with the stores they were three dependent round trips: load -> store -> load ...)
    float gs[3] = {0.f, 0.f, 0.f}, sr[3] = {0.f, 0.f, 0.f};
    if (g_scaling_raw && g_scale) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { gs[k] = g_scale[3 * i + k]; sr[k] = scaling_raw[3 * i + k]; }
    }
    const float op_raw = opacity_raw[i];
    const float go = g_opacity ? g_opacity[i] : 0.f;
    const float dlt = delta_o ? delta_o[i * stride] : 0.f;
    if (g_scaling_raw) {
#pragma unroll
        for (int k = 0; k < 3; ++k) g_scaling_raw[3 * i + k] = g_scale ? gs[k] * expf(sr[k]) : 0.f;
    }
    const float so = sigmoidf(op_raw);
    float life = 1.f;
    if (delta_o) life = 1.f / (1.f + expf(-dlt / beta));
    if (g_opacity_raw) g_opacity_raw[i] = go * life * so * (1.f - so);
    if (delta_o && g_delta_o) g_delta_o[i * stride] = go * so * life * (1.f - life) / beta;
}

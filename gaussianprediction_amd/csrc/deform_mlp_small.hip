// deform_mlp_small.hip -- the Deformable_Field MLP for SMALL row counts (stage 2/3: the rows are the K <= 512
// keypoints [REF scene/gaussian_model.py:262-270]).  The 32-row kernels of deform_kernels.hip would occupy only
// K/32 = 8..16 of the 256 CUs, and each of those CUs is matrix-core-bound on its 32 x 256 x 256 layer.  Here a
// workgroup owns 16 rows and uses v_mfma_f32_16x16x4_f32 (still exact fp32), so twice as many CUs share the work
// and each does half of it.  Same fusion as the large kernels: positional encoding + concat + 5 layers, activations
// transposed in LDS, weights straight from L2 as the A operand, saved activations in the same row-major layout
// (the weight-gradient kernel is shared).
//
// LDS layout: act[pi(f)][16 rows] with pi(16 q + 4 a + b) = 16 q + 4 b + a.  MFMA step (q, u) needs k = 16 q + 4 kg + u
// from lane group kg (so that a lane's four A values are one float4 of a weight row); those four k sit in the four
// CONSECUTIVE LDS rows 16 q + 4 u + kg -- 64 consecutive floats per instruction, bank-conflict free -- and the C/D
// layout (lane group kg' holds features 4 kg' + r) writes back as 64 consecutive floats per register r as well.
#include "gp_common.h"
#include "deform_kernels.h"
#include "loss_adam_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SR 16
#define ST 512
#define SW 256

__device__ __forceinline__ int pi16(int f) { return (f & ~15) | ((f & 3) << 2) | ((f >> 2) & 3); }
__device__ __forceinline__ int sidx(int f, int n) { return pi16(f) * SR + n; }

template <int NT = 512>
__device__ __forceinline__ void build_input_s(float* buf, const MlpDev& p, long row0, int tid, int k16) {
    const int fd = p.feature_dim, xf = p.xyz_freq, tf = p.time_freq;
    for (int e = tid; e < fd * SR; e += NT) {
        const int jj = e / fd, f = e - jj * fd;
        const long row = row0 + jj;
        buf[sidx(f, jj)] = row < p.rows ? p.feature[row * fd + f] : 0.f;
    }
    for (int e = tid; e < 3 * xf * SR; e += NT) {
        const int jj = e % SR, cf = e / SR;  // cf = c*xf + fr
        const int c = cf / xf, fr = cf - c * xf;
        const long row = row0 + jj;
        float sv = 0.f, cv = 0.f;
        if (row < p.rows) sincosf(p.xyz[row * 3 + c] * (float)(1u << fr), &sv, &cv);
        const int f = fd + 2 * cf;
        buf[sidx(f, jj)] = sv;
        buf[sidx(f + 1, jj)] = cv;
    }
    const float tv = tf > 0 ? p.t[0] : 0.f;
    for (int e = tid; e < tf * SR; e += NT) {
        const int jj = e % SR, fr = e / SR;
        float sv, cv;
        sincosf(tv * (float)(1u << fr), &sv, &cv);
        const bool ok = row0 + jj < p.rows;
        const int f = fd + 6 * xf + 2 * fr;
        buf[sidx(f, jj)] = ok ? sv : 0.f;
        buf[sidx(f + 1, jj)] = ok ? cv : 0.f;
    }
    for (int e = tid; e < (k16 - p.in_dim) * SR; e += NT) buf[sidx(p.in_dim + e / SR, e % SR)] = 0.f;
}

// coalesced copy LDS act^T[0:nf][16] -> global dst[(row0+jj)*ld + f]
template <int NT = 512>
__device__ __forceinline__ void store_rows_s(const float* buf, float* dst, int nf, int nf_valid, int ld, long row0, long rows, int tid) {
    for (int e = tid; e < nf * SR; e += NT) {
        const int jj = e / nf, f = e - jj * nf;
        if (row0 + jj < rows) dst[(row0 + jj) * (long)ld + f] = f < nf_valid ? buf[sidx(f, jj)] : 0.f;
    }
}

// two 16-feature tiles (t0, t0+1) of  out^T = W[., 0:k_valid] . cur^T.  K16 = k range walked (multiple of 16; LDS rows
// beyond k_valid must hold finite values, the A operand is zero there).
__device__ __forceinline__ void tiles_mfma(const float* __restrict__ W, int ldw, int k_valid, int K16, int out_rows, int t0,
                                           const float* cur, int lane, f32x4& acc0, f32x4& acc1) {
    const int i = lane & 15, kg = lane >> 4;
    const int r0 = 16 * t0 + i, r1 = r0 + 16;
    const bool ok0 = r0 < out_rows, ok1 = r1 < out_rows;
    const float* w0 = W + (size_t)r0 * ldw;
    const float* w1 = W + (size_t)r1 * ldw;
    const bool vec = (ldw & 3) == 0;
    const int nq = K16 / 16;
    for (int q0 = 0; q0 < nq; q0 += 2) {
        float4 a0[2], a1[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int kb = 16 * (q0 + g) + 4 * kg;
            a0[g] = make_float4(0.f, 0.f, 0.f, 0.f); a1[g] = a0[g];
            if (q0 + g < nq) {
                if (vec && kb + 3 < k_valid) {
                    if (ok0) a0[g] = *(const float4*)(w0 + kb);
                    if (ok1) a1[g] = *(const float4*)(w1 + kb);
                } else {
                    float* p0 = (float*)&a0[g]; float* p1 = (float*)&a1[g];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ok0 && kb + u < k_valid) p0[u] = w0[kb + u];
                        if (ok1 && kb + u < k_valid) p1[u] = w1[kb + u];
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (q0 + g < nq) {
                const float* b = cur + (16 * (q0 + g) + kg) * SR + i;   // row pi(16 q + 4 kg + u) = 16 q + 4 u + kg
                const float b0 = b[0], b1 = b[4 * SR], b2 = b[8 * SR], b3 = b[12 * SR];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].x, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].x, b0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].y, b1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].y, b1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].z, b2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].z, b2, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].w, b3, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].w, b3, acc1, 0, 0, 0);
            }
        }
    }
}

// transposed: out^T[i][row] = sum_k W[k][i] cur^T[k][row], W is [k_valid, ldw]
__device__ __forceinline__ void tiles_mfma_T(const float* __restrict__ W, int ldw, int k_valid, int K16, int out_rows, int t0,
                                             const float* cur, int lane, f32x4& acc0, f32x4& acc1) {
    const int i = lane & 15, kg = lane >> 4;
    const int r0 = 16 * t0 + i, r1 = r0 + 16;
    const bool ok0 = r0 < out_rows, ok1 = r1 < out_rows;
    const int nq = K16 / 16;
    for (int q = 0; q < nq; ++q) {
        float a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 16 * q + 4 * kg + u;
            a0[u] = (ok0 && k < k_valid) ? W[(size_t)k * ldw + r0] : 0.f;
            a1[u] = (ok1 && k < k_valid) ? W[(size_t)k * ldw + r1] : 0.f;
        }
        const float* b = cur + (16 * q + kg) * SR + i;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float bv = b[4 * u * SR];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bv, acc1, 0, 0, 0);
        }
    }
}

// K = 256 variants: with so few workgroups per CU nothing hides L2 latency, so a wave fetches its whole slice of the
// weight matrix (2 tiles x 16 k-steps x float4 = 128 VGPRs) in one burst before the 128 dependent MFMAs.
__device__ __forceinline__ void tiles_mfma_256(const float* __restrict__ W, int out_rows, int t0, const float* cur, int lane,
                                               f32x4& acc0, f32x4& acc1) {
    const int i = lane & 15, kg = lane >> 4;
    const int r0 = 16 * t0 + i, r1 = r0 + 16;
    const bool ok0 = r0 < out_rows, ok1 = r1 < out_rows;
    // Rows beyond out_rows are CLAMPED to a valid row instead of selected to zero after the load (their products land in
    // accumulator rows nobody stores): a select on every loaded register made the compiler wait for ALL 32 loads
    // (s_waitcnt vmcnt(0)) before the first MFMA -- 1.7 us of L2 streaming per layer in front of 3.4 us of matrix work that
    // could have started after the first return.  With plain loads the waits are counted (vmcnt(30), (28), ...).
    const float4* w0 = (const float4*)(W + (size_t)(ok0 ? r0 : 0) * SW) + kg;
    const float4* w1 = (const float4*)(W + (size_t)(ok1 ? r1 : 0) * SW) + kg;
    // Four groups of four k-steps, fetched TWO groups ahead of their MFMAs (96 registers of weights in flight instead of the 128 of
    // one burst; measured equal in the forward, 0.045 -> 0.043 ms in the backward).  What bounds a 256 x 256 layer here is the rate
    // at which ONE CU takes the weights in: 8.7 us per layer (s_memtime stamps) against 4.5 us with the loads ablated and 3.4 us of
    // matrix work -- 256 KB per layer and CU at ~25 B/clk, every load instruction being 16 rows x 64 B = 16 separate requests.
    // Neither warming the L2s from the idle CUs nor the order of the requests changes it (both built and measured).
    float4 a0[16], a1[16];
    auto fetch = [&](int g) {
#pragma unroll
        for (int q = 4 * g; q < 4 * g + 4; ++q) { a0[q] = w0[4 * q]; a1[q] = w1[4 * q]; }
    };
    fetch(0);
    fetch(1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g + 2 < 4) fetch(g + 2);
        __builtin_amdgcn_sched_barrier(0);  // the next-but-one group's loads stay in front of this group's MFMAs (the scheduler would
                                            // otherwise sink them to their first use, exposing every L2 latency)
#pragma unroll
        for (int q = 4 * g; q < 4 * g + 4; ++q) {
            const float* b = cur + (16 * q + kg) * SR + i;
            const float b0 = b[0], b1 = b[4 * SR], b2 = b[8 * SR], b3 = b[12 * SR];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].x, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].x, b0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].y, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].y, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].z, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].z, b2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].w, b3, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].w, b3, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void tiles_mfma_T_256(const float* __restrict__ W, int ldw, int out_rows, int t0, const float* cur, int lane,
                                                 f32x4& acc0, f32x4& acc1) {
    const int i = lane & 15, kg = lane >> 4;
    const int r0 = 16 * t0 + i, r1 = r0 + 16;
    const bool ok0 = r0 < out_rows, ok1 = r1 < out_rows;
    const int c0 = ok0 ? r0 : 0, c1 = ok1 ? r1 : 0;        // (clamped, not selected: see tiles_mfma_256)
    float a0[64], a1[64];
    auto fetch = [&](int g) {               // (groups of four k-steps, two groups ahead: see tiles_mfma_256)
#pragma unroll
        for (int q = 4 * g; q < 4 * g + 4; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 16 * q + 4 * kg + u;
                a0[4 * q + u] = W[(size_t)k * ldw + c0];
                a1[4 * q + u] = W[(size_t)k * ldw + c1];
            }
    };
    fetch(0);
    fetch(1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g + 2 < 4) fetch(g + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 4 * g; q < 4 * g + 4; ++q) {
            const float* b = cur + (16 * q + kg) * SR + i;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float bv = b[4 * u * SR];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * q + u], bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * q + u], bv, acc1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- fragment-ordered weights (MlpPackLayout, deform_kernels.h): every operand load of a wave is one contiguous kilobyte ----------
// generic k range (layer 0's forward: nq = ceil(in_dim / 16) k-steps, zero-padded in the copy)
__device__ __forceinline__ void tiles_mfma_pk(const float4* __restrict__ P, int nq, int t0, const float* cur, int lane,
                                              f32x4& acc0, f32x4& acc1) {
    const int i = lane & 15, kg = lane >> 4;
    const float4* w0 = P + (size_t)t0 * nq * 64 + lane;
    const float4* w1 = w0 + (size_t)nq * 64;
    for (int q0 = 0; q0 < nq; q0 += 2) {
        float4 a0[2], a1[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int q = q0 + g < nq ? q0 + g : nq - 1;          // (clamped: a repeated step is not used)
            a0[g] = w0[64 * q]; a1[g] = w1[64 * q];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (q0 + g < nq) {
                const float* b = cur + (16 * (q0 + g) + kg) * SR + i;
                const float b0 = b[0], b1 = b[4 * SR], b2 = b[8 * SR], b3 = b[12 * SR];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].x, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].x, b0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].y, b1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].y, b1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].z, b2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].z, b2, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g].w, b3, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g].w, b3, acc1, 0, 0, 0);
            }
        }
    }
}
// K = 256 (16 k-steps), forward AND backward: the two copies differ in what they hold, not in how they are read -- the backward's
// four scalar loads per k-step (W[k][r]: four rows) are one float4 of the B copy.  Same two-groups-ahead prefetch as tiles_mfma_256.
__device__ __forceinline__ void tiles_mfma_256_pk(const float4* __restrict__ P, int t0, const float* cur, int lane,
                                                  f32x4& acc0, f32x4& acc1) {
    const int i = lane & 15, kg = lane >> 4;
    const float4* w0 = P + (size_t)t0 * 16 * 64 + lane;
    const float4* w1 = w0 + 16 * 64;
    float4 a0[16], a1[16];
    auto fetch = [&](int g) {
#pragma unroll
        for (int q = 4 * g; q < 4 * g + 4; ++q) { a0[q] = w0[64 * q]; a1[q] = w1[64 * q]; }
    };
    fetch(0);
    fetch(1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g + 2 < 4) fetch(g + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 4 * g; q < 4 * g + 4; ++q) {
            const float* b = cur + (16 * q + kg) * SR + i;
            const float b0 = b[0], b1 = b[4 * SR], b2 = b[8 * SR], b3 = b[12 * SR];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].x, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].x, b0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].y, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].y, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].z, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].z, b2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q].w, b3, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q].w, b3, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one thread per float4 of the copy (0.9 MB: a few microseconds, once per weight state)
__global__ __launch_bounds__(256) void gp_mlp_pack_kernel(MlpDev p, float4* __restrict__ out) {
    const MlpPackLayout L = mlp_pack_layout(p.in_dim);
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= L.total) return;
    int l = 0;
    bool bwd = false;
    long base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e >= L.off_f[k]) { l = k; bwd = false; base = L.off_f[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e >= L.off_b[k]) { l = k; bwd = true; base = L.off_b[k]; }
    }
    const int ldw = l == 0 ? p.in_dim : SW;                    // W_l is [256, ldw]
    const float* W = p.w[l];
    const long r = e - base;
    const int nq = (!bwd && l == 0) ? L.q0 : 16;
    const int lane = (int)(r & 63), q = (int)((r >> 6) % nq), t = (int)((r >> 6) / nq);
    const int i = lane & 15, k0 = 16 * q + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!bwd) { const int row = 16 * t + i, col = k0 + u; v[u] = col < ldw ? W[(size_t)row * ldw + col] : 0.f; }
        else { const int row = k0 + u, col = 16 * t + i; v[u] = col < ldw ? W[(size_t)row * ldw + col] : 0.f; }
    }
    out[e] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(ST) void gp_mlp_fwd_small_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x,
                                                              float* __restrict__ saved_h) {
    __shared__ float smem[2][SW * SR];
    __shared__ float s_out[8][16][SR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4, n = lane & 15;
    const long row0 = (long)blockIdx.x * SR;
    float* cur = smem[0];
    float* nxt = smem[1];
    const int k16 = (p.in_dim + 15) & ~15;
    // this lane's biases of all four hidden layers, requested with the input tile's loads (fetched at the top of each layer they were
    // a trip to L2 in front of every layer's product: the accumulators start from them)
    float4 bia[4][2];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        bia[l][0] = *(const float4*)(p.b[l] + 32 * wave + 4 * kg);
        bia[l][1] = *(const float4*)(p.b[l] + 32 * wave + 16 + 4 * kg);
    }
    build_input_s(cur, p, row0, tid, k16);
    __syncthreads();
    if (saved_x) store_rows_s(cur, saved_x, p.in_pad, p.in_dim, p.in_pad, row0, p.rows, tid);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int kv = l == 0 ? p.in_dim : SW, K16 = l == 0 ? k16 : SW, ldw = l == 0 ? p.in_dim : SW;
        f32x4 acc0 = {bia[l][0].x, bia[l][0].y, bia[l][0].z, bia[l][0].w}, acc1 = {bia[l][1].x, bia[l][1].y, bia[l][1].z, bia[l][1].w};
        if (p.pk) {
            const MlpPackLayout L = mlp_pack_layout(p.in_dim);
            const float4* P = (const float4*)p.pk + L.off_f[l];
            if (l == 0) tiles_mfma_pk(P, L.q0, 2 * wave, cur, lane, acc0, acc1);
            else tiles_mfma_256_pk(P, 2 * wave, cur, lane, acc0, acc1);
        } else if (l == 0) tiles_mfma(p.w[l], ldw, kv, K16, SW, 2 * wave, cur, lane, acc0, acc1);
        else tiles_mfma_256(p.w[l], SW, 2 * wave, cur, lane, acc0, acc1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            nxt[sidx(32 * wave + 4 * kg + r, n)] = fmaxf(acc0[r], 0.f);
            nxt[sidx(32 * wave + 16 + 4 * kg + r, n)] = fmaxf(acc1[r], 0.f);
        }
        __syncthreads();
        if (saved_h) store_rows_s(nxt, saved_h + (size_t)l * p.rows * SW, SW, SW, SW, row0, p.rows, tid);
        float* t = cur; cur = nxt; nxt = t;
    }
    // output layer: out_dim <= 8 rows of W4 (one 16-feature tile); K split over the 8 waves, reduced through LDS
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dummy = {0.f, 0.f, 0.f, 0.f};
        // wave w walks k in [32 w, 32 w + 32): shift the weight / activation windows accordingly
        tiles_mfma(p.w[4] + 32 * wave, SW, 32, 32, p.out_dim, 0, cur + 32 * wave * SR, lane, acc, dummy);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_out[wave][4 * kg + r][n] = acc[r];
        __syncthreads();
        if (tid < 8 * SR) {
            const int jj = tid / 8, f = tid % 8;
            if (f < p.out_dim && row0 + jj < p.rows) {
                float v = p.b[4][f];
#pragma unroll
                for (int w = 0; w < 8; ++w) v += s_out[w][f][jj];
                out[(row0 + jj) * p.out_dim + f] = v;
            }
        }
    }
}

__device__ __forceinline__ void mlp_bwd_data_small_body(const MlpDev& p, const float* __restrict__ saved_h,
                                                        const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                        float* __restrict__ dfeature, float* __restrict__ dxyz, unsigned block) {
    __shared__ float smem[2][SW * SR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4, n = lane & 15;
    const long row0 = (long)block * SR;
    float* cur = smem[0];
    float* nxt = smem[1];
    // dZ5^T [16][16] (zero-padded beyond out_dim)
    for (int e = tid; e < 16 * SR; e += ST) {
        const int jj = e / 16, f = e % 16;
        const long row = row0 + jj;
        cur[sidx(f, jj)] = (f < p.out_dim && row < p.rows) ? dL_dout[row * p.out_dim + f] : 0.f;
    }
    __syncthreads();
    for (int l = 4; l >= 1; --l) {
        // dH_l^T = W_l^T dZ_{l+1}^T ; W_l = p.w[l] is [Kout, 256]
        const int K16 = l == 4 ? 16 : SW, kv = l == 4 ? p.out_dim : SW;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        // the saved activations that gate this layer's gradient (ReLU) are requested BEFORE the product (clamped row, selected
        // afterwards): loaded where they are used they were a dependent trip to memory behind every layer's matrix work
        const float* h = saved_h + (size_t)(l - 1) * p.rows * SW;
        const long row = row0 + n, rowc = row < p.rows ? row : p.rows - 1;
        float hv0[4], hv1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hv0[r] = h[rowc * SW + 32 * wave + 4 * kg + r];
            hv1[r] = h[rowc * SW + 32 * wave + 16 + 4 * kg + r];
        }
        if (l == 4) tiles_mfma_T(p.w[l], SW, kv, K16, SW, 2 * wave, cur, lane, acc0, acc1);
        else if (p.pk) tiles_mfma_256_pk((const float4*)p.pk + mlp_pack_layout(p.in_dim).off_b[l], 2 * wave, cur, lane, acc0, acc1);
        else tiles_mfma_T_256(p.w[l], SW, SW, 2 * wave, cur, lane, acc0, acc1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = 32 * wave + 4 * kg + r, f1 = f0 + 16;
            nxt[sidx(f0, n)] = (row < p.rows && hv0[r] > 0.f) ? acc0[r] : 0.f;
            nxt[sidx(f1, n)] = (row < p.rows && hv1[r] > 0.f) ? acc1[r] : 0.f;
        }
        __syncthreads();
        store_rows_s(nxt, dz + (size_t)(l - 1) * p.rows * SW, SW, SW, SW, row0, p.rows, tid);
        float* t = cur; cur = nxt; nxt = t;
    }
    // dX^T [in_dim][16] = W_0^T dZ_1^T ; W_0 is [256, in_dim]
    if (dfeature || dxyz) {
        if (32 * wave < p.in_dim) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            if (p.pk) tiles_mfma_256_pk((const float4*)p.pk + mlp_pack_layout(p.in_dim).off_b[0], 2 * wave, cur, lane, acc0, acc1);
            else tiles_mfma_T_256(p.w[0], p.in_dim, p.in_dim, 2 * wave, cur, lane, acc0, acc1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nxt[sidx(32 * wave + 4 * kg + r, n)] = acc0[r];
                nxt[sidx(32 * wave + 16 + 4 * kg + r, n)] = acc1[r];
            }
        }
        __syncthreads();
        if (dfeature) store_rows_s(nxt, dfeature, p.feature_dim, p.feature_dim, p.feature_dim, row0, p.rows, tid);
        if (dxyz) {
            // d/dx sin(x 2^f) = 2^f cos, d/dx cos(x 2^f) = -2^f sin
            if (tid < 3 * SR) {
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
                        g += sc * (cv * nxt[sidx(f, jj)] - sv * nxt[sidx(f + 1, jj)]);
                    }
                    dxyz[row * 3 + c] = g;
                }
            }
        }
    }
}

__global__ __launch_bounds__(ST) void gp_mlp_bwd_data_small_kernel(MlpDev p, const float* __restrict__ saved_h,
                                                                   const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                                   float* __restrict__ dfeature, float* __restrict__ dxyz) {
    mlp_bwd_data_small_body(p, saved_h, dL_dout, dz, dfeature, dxyz, blockIdx.x);
}

// The same launch with a RIDER (loss_adam_kernels.h): workgroups [0, n_mlp) are the data backward above -- a handful of workgroups, each
// bound for tens of microseconds by the rate at which ONE CU takes the layer weights in -- and workgroups [n_mlp, ...) are the chunks of an
// optimizer launch that needs nothing this backward produces (the per-Gaussian tensors' Adam: HBM-bound, every CU).  The low block ids
// are dispatched first, so the long-running MLP workgroups start at once and the stream of Adam chunks fills the rest of the part around
// them.  (Round 5 ran the two as separate launches on two streams: the fork / join events cost more than the overlap brought,
// profiles/r05_early_adam_ab.txt.)  The chunk body is element-wise: the update is bit-identical to gp_adam_multi_kernel's.
// Measured (profiles/r06_adam_rider_ab.txt): 0.039 + 0.058 ms as two launches, 0.080 ms as one -- not the 0.058 of the longer half:
// the MLP body's 196 registers leave ONE 512-thread workgroup per CU, and at that occupancy the update alone takes 0.078 ms whatever
// its shape (one chunk per workgroup in two trips of 8 loads per lane: 0.078; the whole chunk in one trip of 32 loads: 0.086; half a chunk
// per workgroup: 0.080; eight chunks per workgroup, plain or software-pipelined: 0.13 - 0.14 -- on this part a wave's loads queue behind
// its own earlier STORES, one counter in order, so every further trip of a workgroup waits for the write acknowledgements of the trip before).
__global__ __launch_bounds__(ST) void gp_mlp_bwd_data_small_adam_kernel(MlpDev p, const float* __restrict__ saved_h,
                                                                        const float* __restrict__ dL_dout, float* __restrict__ dz,
                                                                        float* __restrict__ dfeature, float* __restrict__ dxyz,
                                                                        unsigned n_mlp, AdamTable t, float b1, float b2, float eps,
                                                                        int zero_grad, const uint32_t* __restrict__ skip_flag) {
    if (blockIdx.x < n_mlp) mlp_bwd_data_small_body(p, saved_h, dL_dout, dz, dfeature, dxyz, blockIdx.x);
    else adam_chunk_body<ST>(t, blockIdx.x - n_mlp, threadIdx.x, b1, b2, eps, zero_grad, skip_flag);
}

// ------------------------------------------------------------------------------------------------
// FEATURE-SPLIT forward for <= 512 rows (round 6).  What bounds the 16-row kernels above is the rate at which ONE CU takes the layer
// weights in (~25 B/clk: 8.7 us per 256 x 256 layer, 4 layers), and every workgroup needs ALL the weights however few rows it owns: more
// workgroups along the rows buy nothing.  Here the 16 row tiles are split along the FEATURES as well: workgroup (row tile rt, feature
// tile ft) computes 16 rows x 16 features of every hidden layer -- 16 KB of weights per layer instead of 256 KB, requested for all
// layers before the first product -- and the 16 workgroups of a row tile exchange the layer's activations through memory: the saved
// activations of the training pass ARE the exchange buffer (the scratch's when nothing is saved).  Per layer: wait for the 16 arrivals of
// the layer before (one counter per (row tile, layer) in the caller's scratch; agent-scope release / acquire around it), 4 x 16 B per lane
// straight from memory as the MFMA's B operand, K split over the 4 waves (16 MFMAs each), the four partial tiles added in a fixed order
// through LDS, bias + ReLU, 64 B per row to the exchange buffer, arrive.  The 16 workgroups of a row tile are placed on ONE XCD (blockIdx
// & 7 = XCD: the exchange stays in that XCD's L2; correctness does not depend on it).  Feature tile 0 also writes the input tile and runs
// the output layer; having passed the last counter it knows every partner has passed all of its own and zeroes the row tile's counters:
// the scratch returns to its initial state at the end of every launch.  A counter that does not fill (partners not resident: the host
// checks the occupancy before choosing this kernel) ends the wait after ~1 s and raises bit 0 of the scratch's error word instead of
// hanging.  The XCD-local form of the exchange (ns_arrive / ns_wait below) NEEDS the one-XCD placement: every workgroup publishes its
// XCC_ID, a row tile seen on two XCDs raises bit 1 of the error word, and the host validates the first launch on a device before it
// trusts the form (gp_capi_deform.hip; otherwise the agent-scope form or the 16-row kernels).
// Sums: K in four quarters, each over two accumulators, added as (w0 + w1) + (w2 + w3) + bias -- another order than the 16-row kernels'
// single chain (agreement ~1e-7 relative; the golden-vector tests hold both).
// ------------------------------------------------------------------------------------------------
#define NS_T 256
// LOCAL: the 16 workgroups of a row tile share an XCD, whose L2 is then the point of coherence -- stores are complete there once the
// wave's counter has drained (the L1 writes through), the counter is an L2 atomic, and nothing has to be written back or invalidated
// (the agent-scope form's L2 write-backs / invalidations serialise per XCD: 93 us at 256 workgroups, 177 us at 512).
template <bool LOCAL>
__device__ __forceinline__ void ns_arrive(uint32_t* flag) {          // every thread of the workgroup calls it, behind its stores
    if constexpr (LOCAL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <bool LOCAL>
__device__ __forceinline__ void ns_wait(uint32_t* flag, uint32_t target, uint32_t* err) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        // (an agent-scope load: never answered by the CU's L1 -- a workgroup-scope read-modify-write of 0 was turned into a plain load
        // and spun on the L1's copy until the limit below)
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 23)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        if constexpr (!LOCAL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
// the 4 x 4 MFMAs of a wave's K quarter: A = 4 float4 of weights, B = 4 float4 of activations (lane (i, kg): k = 16 q + 4 kg + 0..3)
__device__ __forceinline__ f32x4 ns_product(const float4 (&a)[4], const float4 (&b)[4]) {
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].x, b[g].x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].y, b[g].y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].z, b[g].z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].w, b[g].w, c1, 0, 0, 0);
    }
    return c0 + c1;
}

template <bool LOCAL>
__device__ __forceinline__ void mlp_fwd_split_small_body(const MlpDev& p, float* __restrict__ out, float* __restrict__ saved_x,
                                                         float* hx /* [4][rows][256]: read AND written, by other CUs too */,
                                                         uint32_t* flags, uint32_t* err) {
    __shared__ float s_in[SW * SR];
    __shared__ __attribute__((aligned(16))) float s_part[4][SR * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kg = lane >> 4;
    const unsigned slot = blockIdx.x >> 3;
    const int rt = (int)((slot >> 4) * 8 + (blockIdx.x & 7)), ft = (int)(slot & 15);
    const long row0 = (long)rt * SR;
    if (row0 >= p.rows) return;
    const int f0 = 16 * ft;
    uint32_t* fl = flags + 32 * rt;         // a row tile's counters in a 128-byte line of their own (all tiles' in one line: every
                                            // arrival and every poll of 256 workgroups met at the same address -- 22 us instead of 17 at 250 rows)
    uint32_t* xm = fl + 4;                  // which XCDs the row tile's workgroups run on (LOCAL: it must be one)
    if (LOCAL && tid == 0)
        __hip_atomic_fetch_or(xm, 1u << (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // HW_REG_XCC_ID[3:0]
    const size_t lstride = (size_t)p.rows * SW;
    // every weight this workgroup will need, requested now
    float4 wr[3][4], w4[4];
#pragma unroll
    for (int l = 1; l < 4; ++l)
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[l - 1][g] = *(const float4*)(p.w[l] + (size_t)(f0 + i) * SW + 16 * (4 * wave + g) + 4 * kg);
    if (ft == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) w4[g] = *(const float4*)(p.w[4] + (size_t)(i < p.out_dim ? i : 0) * SW + 16 * (4 * wave + g) + 4 * kg);
    }
    float bias[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) bias[l] = p.b[l][f0 + (tid & 15)];
    const int k16 = (p.in_dim + 15) & ~15, nq0 = k16 / 16;
    float4 w0[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {           // layer 0: k-steps wave, wave + 4, ... (in_dim <= 256: at most four per wave)
        const int k = 16 * (wave + 4 * g) + 4 * kg;
        w0[g] = k < p.in_dim ? *(const float4*)(p.w[0] + (size_t)(f0 + i) * p.in_dim + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    build_input_s<NS_T>(s_in, p, row0, tid, k16);
    __syncthreads();
    if (ft == 0 && saved_x) store_rows_s<NS_T>(s_in, saved_x, p.in_pad, p.in_dim, p.in_pad, row0, p.rows, tid);
    auto reduce = [&](const f32x4& acc) {       // the tile's element (row tid >> 4, feature tid & 15) without bias
        *(float4*)&s_part[wave][i * 16 + 4 * kg] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        return (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
    };
    const long my_row = row0 + (tid >> 4);
    {   // layer 0: B from the LDS input tile (k >= in_dim: zero rows of the tile against zero weights)
        float4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = wave + 4 * g;
            if (q < nq0) {
                const float* bp = s_in + (16 * q + kg) * SR + i;        // row pi(16 q + 4 kg + u) = 16 q + 4 u + kg
                b[g] = make_float4(bp[0], bp[4 * SR], bp[8 * SR], bp[12 * SR]);
            } else b[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float v = fmaxf(reduce(ns_product(w0, b)) + bias[0], 0.f);
        if (my_row < p.rows) hx[my_row * SW + f0 + (tid & 15)] = v;
        ns_arrive<LOCAL>(fl + 0);
    }
    const long rowc = row0 + i < p.rows ? row0 + i : p.rows - 1;       // (rows beyond the input: a valid row's values, never stored)
#pragma unroll
    for (int l = 1; l < 4; ++l) {
        ns_wait<LOCAL>(fl + l - 1, 16u, err);
        const float* src = hx + (size_t)(l - 1) * lstride + rowc * SW + 64 * wave + 4 * kg;
        float4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = *(const float4*)(src + 16 * g);
        const float v = fmaxf(reduce(ns_product(wr[l - 1], b)) + bias[l], 0.f);
        if (my_row < p.rows) hx[(size_t)l * lstride + my_row * SW + f0 + (tid & 15)] = v;
        ns_arrive<LOCAL>(fl + l);
    }
    if (ft != 0) return;
    ns_wait<LOCAL>(fl + 3, 16u, err);
    if (tid < 4) fl[tid] = 0u;              // every partner has passed all of its waits: the counters return to zero for the next launch
    if (LOCAL && tid == 0) {                // (every partner's bit is in: each set its own before its first arrival)
        const uint32_t m = __hip_atomic_load(xm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m & (m - 1u)) __hip_atomic_fetch_or(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *xm = 0u;
    }
    {
        const float* src = hx + (size_t)3 * lstride + rowc * SW + 64 * wave + 4 * kg;
        float4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = *(const float4*)(src + 16 * g);
        const float v = reduce(ns_product(w4, b));
        const int f = tid & 15;
        if (f < p.out_dim && my_row < p.rows) out[my_row * p.out_dim + f] = v + p.b[4][f];
    }
}
__global__ __launch_bounds__(NS_T) void gp_mlp_fwd_split_small_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x, float* hx,
                                                                      uint32_t* flags, uint32_t* err) {
    mlp_fwd_split_small_body<true>(p, out, saved_x, hx, flags, err);
}
__global__ __launch_bounds__(NS_T) void gp_mlp_fwd_split_small_agent_kernel(MlpDev p, float* __restrict__ out, float* __restrict__ saved_x,
                                                                            float* hx, uint32_t* flags, uint32_t* err) {
    mlp_fwd_split_small_body<false>(p, out, saved_x, hx, flags, err);
}

// ------------------------------------------------------------------------------------------------
// FEATURE-SPLIT data backward for <= 512 rows (round 6): the forward's scheme run backwards.  Workgroup (row tile, feature tile ft) owns
// 16 of the 256 columns of every dZ buffer (which the weight-gradient kernel needs in memory anyway: they are the exchange buffers):
//   step 0: dZ_4 = gate(W_4^T dL/dout)            (K = out_dim: one wave)        -> arrive 0
//   step s = 1..3: wait s - 1; dZ_{4-s} = gate(W_{4-s}^T dZ_{5-s}), K over the 4 waves -> arrive s
//   feature tiles below ceil(in_dim / 16): wait 3; dX = W_0^T dZ_1; d feature written by its owners; the encoding's part of dX goes
//   through the scratch's exchange region (free in a training pass: the forward exchanged through the saved activations) -> arrive 4;
//   feature tile 0: wait 4, counters back to zero, d xyz through the encoding's derivative.
// Counters: words 8..12 of the row tile's line in gp_mlp_params.scratch, XCC mask in word 13.  Same sums as the forward's remark: K in
// four quarters over two accumulators each -- another order than the 16-row kernel's.
// ------------------------------------------------------------------------------------------------
template <bool LOCAL>
__device__ __forceinline__ void mlp_bwd_data_split_small_body(const MlpDev& p, const float* __restrict__ saved_h, const float* __restrict__ dL_dout,
                                                              float* dz /* [4][rows][256]: exchange */, float* __restrict__ dfeature,
                                                              float* __restrict__ dxyz, float* gx /* [rows][in_pad]: exchange */,
                                                              uint32_t* flags, uint32_t* err, unsigned block, bool add_dfeature) {
    __shared__ __attribute__((aligned(16))) float s_part[4][SR * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kg = lane >> 4;
    const unsigned slot = block >> 3;
    const int rt = (int)((slot >> 4) * 8 + (block & 7)), ft = (int)(slot & 15);
    const long row0 = (long)rt * SR;
    if (row0 >= p.rows) return;
    const int f0 = 16 * ft, nft0 = (p.in_dim + 15) / 16;
    uint32_t* fl = flags + 32 * rt + 8;
    uint32_t* xm = fl + 5;
    if (LOCAL && tid == 0)
        __hip_atomic_fetch_or(xm, 1u << (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // HW_REG_XCC_ID[3:0]
    const size_t lstride = (size_t)p.rows * SW;
    const bool want_dx = (dfeature || dxyz) && ft < nft0;
    // every operand that does not come out of the exchange, requested now
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4, wt[3][4], a0[4];
    const long rowc = row0 + i < p.rows ? row0 + i : p.rows - 1;
    if (wave == 0) {
        float* av = (float*)&a4; float* bv = (float*)&b4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * kg + u;
            av[u] = k < p.out_dim ? p.w[4][(size_t)k * SW + f0 + i] : 0.f;
            bv[u] = k < p.out_dim ? dL_dout[rowc * p.out_dim + k] : 0.f;
        }
    }
#pragma unroll
    for (int l = 3; l >= 1; --l)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* w = p.w[l] + (size_t)(16 * (4 * wave + g) + 4 * kg) * SW + f0 + i;
            wt[l - 1][g] = make_float4(w[0], w[SW], w[2 * SW], w[3 * SW]);
        }
    if (want_dx) {
        const int c = f0 + i;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* w = p.w[0] + (size_t)(16 * (4 * wave + g) + 4 * kg) * p.in_dim + (c < p.in_dim ? c : 0);
            a0[g] = c < p.in_dim ? make_float4(w[0], w[p.in_dim], w[2 * p.in_dim], w[3 * p.in_dim]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const long my_row = row0 + (tid >> 4), my_rowc = my_row < p.rows ? my_row : p.rows - 1;
    float hv[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) hv[l] = saved_h[(size_t)l * lstride + my_rowc * SW + f0 + (tid & 15)];
    auto reduce = [&](const f32x4& acc) {       // the tile's element (row tid >> 4, feature tid & 15)
        *(float4*)&s_part[wave][i * 16 + 4 * kg] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        return (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
    };
    {   // step 0
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (wave == 0) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        const float v = reduce(acc);
        if (my_row < p.rows) dz[(size_t)3 * lstride + my_row * SW + f0 + (tid & 15)] = hv[3] > 0.f ? v : 0.f;
        ns_arrive<LOCAL>(fl + 0);
    }
#pragma unroll
    for (int s = 1; s < 4; ++s) {
        const int l = 4 - s;            // reads dZ buffer l, writes buffer l - 1 through W_l^T
        ns_wait<LOCAL>(fl + s - 1, 16u, err);
        const float* src = dz + (size_t)l * lstride + rowc * SW + 64 * wave + 4 * kg;
        float4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = *(const float4*)(src + 16 * g);
        const float v = reduce(ns_product(wt[l - 1], b));
        if (my_row < p.rows) dz[(size_t)(l - 1) * lstride + my_row * SW + f0 + (tid & 15)] = hv[l - 1] > 0.f ? v : 0.f;
        ns_arrive<LOCAL>(fl + s);
    }
    if (ft >= nft0) return;             // (feature tile 0 is below nft0: in_dim >= 1)
    ns_wait<LOCAL>(fl + 3, 16u, err);
    if (dfeature || dxyz) {
        const float* src = dz + rowc * SW + 64 * wave + 4 * kg;
        float4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = *(const float4*)(src + 16 * g);
        const float v = reduce(ns_product(a0, b));
        const int c = f0 + (tid & 15);
        if (my_row < p.rows && c < p.in_dim) {
            if (dfeature && c < p.feature_dim) {
                float* d = dfeature + my_row * p.feature_dim + c;
                *d = add_dfeature ? *d + v : v;
            }
            if (dxyz) gx[my_row * p.in_pad + c] = v;
        }
    }
    ns_arrive<LOCAL>(fl + 4);
    if (ft != 0) return;
    ns_wait<LOCAL>(fl + 4, (uint32_t)nft0, err);
    if (tid < 5) fl[tid] = 0u;          // every partner has passed all of its waits: the counters return to zero for the next launch
    if (LOCAL && tid == 0) {
        const uint32_t m = __hip_atomic_load(xm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m & (m - 1u)) __hip_atomic_fetch_or(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *xm = 0u;
    }
    if (dxyz && tid < 3 * SR) {         // d/dx sin(x 2^f) = 2^f cos, d/dx cos(x 2^f) = -2^f sin
        const int jj = tid / 3, c = tid % 3;
        const long row = row0 + jj;
        if (row < p.rows) {
            const float x = p.xyz[row * 3 + c];
            const float* grow = gx + row * p.in_pad;
            float g = 0.f;
            for (int fr = 0; fr < p.xyz_freq; ++fr) {
                const float sc = (float)(1u << fr);
                float sv, cv;
                sincosf(x * sc, &sv, &cv);
                const int f = p.feature_dim + 2 * (c * p.xyz_freq + fr);
                g += sc * (cv * grow[f] - sv * grow[f + 1]);
            }
            dxyz[row * 3 + c] = g;
        }
    }
}
__global__ __launch_bounds__(NS_T) void gp_mlp_bwd_data_split_small_kernel(MlpDev p, const float* __restrict__ saved_h, const float* __restrict__ dL_dout,
                                                                           float* dz, float* __restrict__ dfeature, float* __restrict__ dxyz, float* gx,
                                                                           uint32_t* flags, uint32_t* err, int form /* bit 0: agent-scope exchange, bit 1: dfeature += */) {
    if (form & 1) mlp_bwd_data_split_small_body<false>(p, saved_h, dL_dout, dz, dfeature, dxyz, gx, flags, err, blockIdx.x, (form & 2) != 0);
    else mlp_bwd_data_split_small_body<true>(p, saved_h, dL_dout, dz, dfeature, dxyz, gx, flags, err, blockIdx.x, (form & 2) != 0);
}
// (Carrying the optimizer's rider in THIS launch, as gp_mlp_bwd_data_small_adam_kernel does for the 16-row form, was built and measured:
// 0.092 ms for the fused launch against 0.019 + 0.054 ms for the two -- under the Adam chunks' HBM stream every one of the exchange's
// dependent trips to memory takes several times as long, and the chain of five is the kernel's critical path.  The rider stays with
// the 16-row form: profiles/r06_adam_rider_ab.txt.)

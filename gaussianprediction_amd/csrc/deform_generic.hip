// deform_generic.hip -- Deformable_Field for the shapes the fused kernels do not cover.
// The reference builds the network from --d / --w (options/gaussian_option.py:54-55, default 4 / 256) and has two dormant
// variants (use_softmax, split_xyz) [REF scene/deformable_field.py:74-127].  The fused kernels (deform_kernels.hip,
// deform_mlp_small.hip, deform_mlp16.hip) implement the operating point d = 4, w = 256; every other depth / width runs layer by
// layer on the kernels below: a positional-encoding + concat kernel pair, one tiled fp32 GEMM (v_mfma_f32_32x32x2_f32, exact fp32
// products and accumulation) used in its three orientations -- forward, data gradient, weight gradient -- a bias-gradient column sum
// and a row softmax.  Correct for any d >= 1, any w >= 1; not tuned beyond coalesced staging: it is the completeness path, the
// shipped configurations never take it.
#include "gp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GG_BM 64
#define GG_BN 64
#define GG_BK 16

struct GenGemm {
    const float* A; long sa_i, sa_k;     // A(i, k) = A[i sa_i + k sa_k]
    const float* mask;                   // optional, indexed like A: the operand is A where mask > 0, else 0 (ReLU'(z) = [y > 0])
    const float* B; long sb_k, sb_j;     // B(k, j)
    float* C; long sc_i;                 // C[i sc_i + j]
    long I, J, K;
    const float* bias;                   // optional [J]
    int relu;                            // C = max(., 0)
    int atomic;                          // 1: C += (atomicAdd; the K range is split over blockIdx.z, k_chunk each)
    long k_chunk;
    int a_kfast, b_jfast;                // which index of the operand is contiguous in memory (staging order)
};

// C[I, J] (+)= A[I, K] . B[K, J]: a 64 x 64 block per workgroup, one 32 x 32 MFMA tile per wave, K walked in steps of 16 through LDS.
__global__ __launch_bounds__(256) void gp_gen_gemm_kernel(GenGemm g) {
    __shared__ float As[GG_BK][GG_BM + 1];
    __shared__ float Bs[GG_BK][GG_BN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const long i0 = (long)blockIdx.x * GG_BM, j0 = (long)blockIdx.y * GG_BN;
    const long kz0 = (long)blockIdx.z * g.k_chunk;
    const long kz1 = kz0 + g.k_chunk < g.K ? kz0 + g.k_chunk : g.K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (long k0 = kz0; k0 < kz1; k0 += GG_BK) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            int m, kk;
            if (g.a_kfast) { m = e >> 4; kk = e & 15; } else { kk = e >> 6; m = e & 63; }
            const long gi = i0 + m, gk = k0 + kk;
            float v = 0.f;
            if (gi < g.I && gk < kz1) {
                const long off = gi * g.sa_i + gk * g.sa_k;
                v = g.A[off];
                if (g.mask && !(g.mask[off] > 0.f)) v = 0.f;
            }
            As[kk][m] = v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            int n, kk;
            if (g.b_jfast) { kk = e >> 6; n = e & 63; } else { n = e >> 4; kk = e & 15; }
            const long gj = j0 + n, gk = k0 + kk;
            Bs[kk][n] = (gj < g.J && gk < kz1) ? g.B[gk * g.sb_k + gj * g.sb_j] : 0.f;
        }
        __syncthreads();
        // v_mfma_f32_32x32x2_f32: lane l supplies A(m = l % 32, k = l / 32) and B(k = l / 32, n = l % 32)
#pragma unroll
        for (int kk = 0; kk < GG_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + (lane >> 5)][wm + (lane & 31)], Bs[kk + (lane >> 5)][wn + (lane & 31)], acc, 0, 0, 0);
        __syncthreads();
    }
    // accumulator register r of lane l: row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
    const long j = j0 + wn + (lane & 31);
    if (j >= g.J) return;
    const float bj = (g.bias && blockIdx.z == 0) ? g.bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long i = i0 + wm + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (i >= g.I) continue;
        float v = acc[r] + bj;
        if (g.relu) v = fmaxf(v, 0.f);
        float* c = g.C + i * g.sc_i + j;
        if (g.atomic) atomicAdd(c, v); else *c = v;
    }
}

// out[j] += sum_i dy[i, j] [y[i, j] > 0]   (the bias gradient)
__global__ __launch_bounds__(256) void gp_gen_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ y, long rows, int cols,
                                                           float* __restrict__ out) {
    const long r0 = (long)blockIdx.x * 256;
    const long r1 = r0 + 256 < rows ? r0 + 256 : rows;
    for (int j = threadIdx.x; j < cols; j += 256) {
        float s = 0.f;
        for (long i = r0; i < r1; ++i) {
            const float v = dy[i * cols + j];
            s += (!y || y[i * cols + j] > 0.f) ? v : 0.f;
        }
        atomicAdd(out + j, s);
    }
}

// row i of the network input: [ feature[i, :] | (sin, cos)(xyz[i, c] 2^f) for c, f | (sin, cos)(t 2^f) for f ]
// [REF scene/gaussian_model.py:180-189, scene/deformable_field.py:63-72 (ori = False)] -- the fused kernels' encoding (sincosf)
__global__ __launch_bounds__(256) void gp_gen_input_fwd_kernel(long rows, int fd, int xf, int tf, const float* __restrict__ feature,
                                                              const float* __restrict__ xyz, const float* __restrict__ t, float* __restrict__ out) {
    const int in_dim = fd + 6 * xf + 2 * tf;
    const int per_row = fd + 3 * xf + tf;                   // one thread per feature element / (sin, cos) pair
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * per_row) return;
    const long i = e / per_row;
    const int q = (int)(e - i * per_row);
    float* o = out + i * in_dim;
    if (q < fd) { o[q] = feature[i * fd + q]; return; }
    float sv, cv;
    if (q < fd + 3 * xf) {
        const int cf = q - fd, c = cf / xf, f = cf - c * xf;
        sincosf(xyz[i * 3 + c] * (float)(1u << f), &sv, &cv);
        o[fd + 2 * cf] = sv; o[fd + 2 * cf + 1] = cv;
    } else {
        const int f = q - fd - 3 * xf;
        sincosf(t[0] * (float)(1u << f), &sv, &cv);
        o[fd + 6 * xf + 2 * f] = sv; o[fd + 6 * xf + 2 * f + 1] = cv;
    }
}
// d feature = the first fd columns; d xyz[c] = sum_f 2^f (cos . d sin - sin . d cos)
__global__ __launch_bounds__(256) void gp_gen_input_bwd_kernel(long rows, int fd, int xf, int tf, const float* __restrict__ xyz,
                                                              const float* __restrict__ d_in, float* __restrict__ d_feature,
                                                              float* __restrict__ d_xyz) {
    const int in_dim = fd + 6 * xf + 2 * tf;
    const int per_row = fd + 3;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * per_row) return;
    const long i = e / per_row;
    const int q = (int)(e - i * per_row);
    const float* d = d_in + i * in_dim;
    if (q < fd) { if (d_feature) d_feature[i * fd + q] = d[q]; return; }
    if (!d_xyz) return;
    const int c = q - fd;
    const float x = xyz[i * 3 + c];
    float s = 0.f;
    for (int f = 0; f < xf; ++f) {
        const float sc = (float)(1u << f);
        float sv, cv;
        sincosf(x * sc, &sv, &cv);
        const int k = fd + 2 * (c * xf + f);
        s += sc * (cv * d[k] - sv * d[k + 1]);
    }
    d_xyz[i * 3 + c] = s;
}

// nn.Softmax(dim=-1) over a handful of columns: one thread per row
__global__ __launch_bounds__(256) void gp_gen_softmax_fwd_kernel(const float* __restrict__ x, long rows, int dim, float* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    float mx = -INFINITY;
    for (int j = 0; j < dim; ++j) mx = fmaxf(mx, x[i * dim + j]);
    float s = 0.f;
    for (int j = 0; j < dim; ++j) s += expf(x[i * dim + j] - mx);
    for (int j = 0; j < dim; ++j) y[i * dim + j] = expf(x[i * dim + j] - mx) / s;
}
__global__ __launch_bounds__(256) void gp_gen_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long rows, int dim,
                                                                float* __restrict__ dx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    float dot = 0.f;
    for (int j = 0; j < dim; ++j) dot += y[i * dim + j] * dy[i * dim + j];
    for (int j = 0; j < dim; ++j) dx[i * dim + j] = y[i * dim + j] * (dy[i * dim + j] - dot);
}

// ------------------------------------------------------------------------------------------------------------------------------------
static int gen_gemm(GenGemm g, hipStream_t s) {
    if (g.I <= 0 || g.J <= 0) return 0;
    long splits = 1;
    if (g.atomic) {         // weight gradient: the reduction runs over the rows
        g.k_chunk = 2048;
        splits = (g.K + g.k_chunk - 1) / g.k_chunk;
        if (splits < 1) splits = 1;
        if (splits > 65535) { g.k_chunk = (g.K + 65534) / 65535; g.k_chunk = (g.k_chunk + GG_BK - 1) / GG_BK * GG_BK; splits = (g.K + g.k_chunk - 1) / g.k_chunk; }
    } else {
        g.k_chunk = g.K > 0 ? g.K : 1;
    }
    const long gx = (g.I + GG_BM - 1) / GG_BM, gy = (g.J + GG_BN - 1) / GG_BN;
    if (gx > 0x7FFFFFFFL || gy > 65535) GP_FAIL("generic layer: %ld x %ld output blocks exceed the launch limits", gx, gy);
    hipLaunchKernelGGL(gp_gen_gemm_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)splits), dim3(256), 0, s, g);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_linear_forward(const float* x, int64_t rows, int32_t in_dim, const float* w, const float* b, int32_t out_dim, int32_t relu,
                                 float* y, gp_stream_t stream) {
    if (rows < 0 || in_dim <= 0 || out_dim <= 0) GP_FAIL("gp_linear_forward: bad shape (%lld x %d -> %d)", (long long)rows, in_dim, out_dim);
    if (rows == 0) return 0;
    if (!x || !w || !y) GP_FAIL("gp_linear_forward: null pointer");
    GenGemm g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.sa_i = in_dim; g.sa_k = 1; g.a_kfast = 1;
    g.B = w; g.sb_k = 1; g.sb_j = in_dim; g.b_jfast = 0;           // B(k, j) = w[j, k]
    g.C = y; g.sc_i = out_dim; g.I = rows; g.J = out_dim; g.K = in_dim; g.bias = b; g.relu = relu ? 1 : 0;
    GpProfScope _p("linear_fwd", (hipStream_t)stream);
    return gen_gemm(g, (hipStream_t)stream);
}

extern "C" int gp_linear_backward(const float* x, const float* y, const float* dy, int64_t rows, int32_t in_dim, const float* w, int32_t out_dim,
                                  int32_t relu, float* dx, float* dw, float* db, gp_stream_t stream) {
    if (rows < 0 || in_dim <= 0 || out_dim <= 0) GP_FAIL("gp_linear_backward: bad shape (%lld x %d -> %d)", (long long)rows, in_dim, out_dim);
    if (rows == 0) return 0;
    if (!dy || (relu && !y)) GP_FAIL("gp_linear_backward: null pointer (the forward's output is needed behind a ReLU)");
    hipStream_t s = (hipStream_t)stream;
    const float* mask = relu ? y : nullptr;
    GpProfScope _p("linear_bwd", s);
    if (dx) {               // dx[i, k] = sum_j dz[i, j] w[j, k]
        if (!w) GP_FAIL("gp_linear_backward: null weights");
        GenGemm g;
        memset(&g, 0, sizeof(g));
        g.A = dy; g.mask = mask; g.sa_i = out_dim; g.sa_k = 1; g.a_kfast = 1;
        g.B = w; g.sb_k = in_dim; g.sb_j = 1; g.b_jfast = 1;
        g.C = dx; g.sc_i = in_dim; g.I = rows; g.J = in_dim; g.K = out_dim;
        if (gen_gemm(g, s)) return 1;
    }
    if (dw) {               // dw[j, k] += sum_i dz[i, j] x[i, k]
        if (!x) GP_FAIL("gp_linear_backward: null input");
        GenGemm g;
        memset(&g, 0, sizeof(g));
        g.A = dy; g.mask = mask; g.sa_i = 1; g.sa_k = out_dim; g.a_kfast = 0;       // A(j, i) = dz[i, j]
        g.B = x; g.sb_k = in_dim; g.sb_j = 1; g.b_jfast = 1;
        g.C = dw; g.sc_i = in_dim; g.I = out_dim; g.J = in_dim; g.K = rows; g.atomic = 1;
        if (gen_gemm(g, s)) return 1;
    }
    if (db) {
        hipLaunchKernelGGL(gp_gen_colsum_kernel, dim3(gp_blocks((size_t)rows, 256)), dim3(256), 0, s, dy, mask, (long)rows, (int)out_dim, db);
        GP_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gp_mlp_input_forward(const gp_mlp_input* x, float* out, gp_stream_t stream) {
    if (!x || !out) GP_FAIL("gp_mlp_input_forward: null argument");
    if (x->rows <= 0) return 0;
    if (x->feature_dim < 0 || x->xyz_freq < 0 || x->time_freq < 0 || x->xyz_freq > 24 || x->time_freq > 24) GP_FAIL("gp_mlp_input_forward: bad encoding sizes");
    if ((x->feature_dim > 0 && !x->feature) || (x->xyz_freq > 0 && !x->xyz) || (x->time_freq > 0 && !x->t)) GP_FAIL("gp_mlp_input_forward: null input");
    const long per_row = x->feature_dim + 3 * x->xyz_freq + x->time_freq;
    if (per_row == 0) return 0;
    hipLaunchKernelGGL(gp_gen_input_fwd_kernel, dim3(gp_blocks((size_t)(x->rows * per_row), 256)), dim3(256), 0, (hipStream_t)stream, (long)x->rows,
                       x->feature_dim, x->xyz_freq, x->time_freq, x->feature, x->xyz, x->t, out);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_mlp_input_backward(const gp_mlp_input* x, const float* dL_din, float* dL_dfeature, float* dL_dxyz, gp_stream_t stream) {
    if (!x || !dL_din) GP_FAIL("gp_mlp_input_backward: null argument");
    if (x->rows <= 0 || (!dL_dfeature && !dL_dxyz)) return 0;
    if (dL_dxyz && x->xyz_freq > 0 && !x->xyz) GP_FAIL("gp_mlp_input_backward: null xyz");
    if (dL_dxyz && x->xyz_freq == 0) GP_HIP_CHECK(hipMemsetAsync(dL_dxyz, 0, (size_t)x->rows * 3 * sizeof(float), (hipStream_t)stream));
    const long per_row = x->feature_dim + 3;
    hipLaunchKernelGGL(gp_gen_input_bwd_kernel, dim3(gp_blocks((size_t)(x->rows * per_row), 256)), dim3(256), 0, (hipStream_t)stream, (long)x->rows,
                       x->feature_dim, x->xyz_freq, x->time_freq, x->xyz, dL_din, dL_dfeature, x->xyz_freq > 0 ? dL_dxyz : nullptr);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_softmax_forward(const float* x, int64_t rows, int32_t dim, float* y, gp_stream_t stream) {
    if (rows <= 0) return 0;
    if (!x || !y || dim <= 0) GP_FAIL("gp_softmax_forward: bad argument");
    hipLaunchKernelGGL(gp_gen_softmax_fwd_kernel, dim3(gp_blocks((size_t)rows, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)rows, (int)dim, y);
    GP_LAUNCH_CHECK();
    return 0;
}
extern "C" int gp_softmax_backward(const float* y, const float* dy, int64_t rows, int32_t dim, float* dx, gp_stream_t stream) {
    if (rows <= 0) return 0;
    if (!y || !dy || !dx || dim <= 0) GP_FAIL("gp_softmax_backward: bad argument");
    hipLaunchKernelGGL(gp_gen_softmax_bwd_kernel, dim3(gp_blocks((size_t)rows, 256)), dim3(256), 0, (hipStream_t)stream, y, dy, (long)rows, (int)dim, dx);
    GP_LAUNCH_CHECK();
    return 0;
}

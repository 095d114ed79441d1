// dist_kernels.hip -- device side of the view-parallel exchange's FACTORISED SH form.
//
// Per view the gradient of the SH coefficients is rank one: dL/dSH[k][ch] = Y_k(dir) dL/dRGB[ch] (the clamp of the colour is already
// in dL/dRGB) [REF utils/sh_utils.py:57-112 is the forward it differentiates].  Ranks of a view-parallel step can therefore exchange
// (dL/dRGB, dir) = 24 B per Gaussian and view by ONE all-gather instead of reducing 192 B of gradient per Gaussian -- and then all need
// the SUM over the views, in the same order on every rank so that the replicas stay bit-identical.  This kernel forms it:
//     g_dc[i][ch]      = sum_v Y_0        f[v][i][0:3][ch]
//     g_rest[i][k-1][ch] = sum_v Y_k(f[v][i][3:6]) f[v][i][0:3][ch]      k = 1 .. (degree + 1)^2 - 1, zero beyond
// One thread per (Gaussian, coefficient): consecutive threads write consecutive 12-byte pieces (coalesced); the 16 threads of a
// Gaussian re-read its factors through L1.  Views are summed in rank order with one fused multiply-add per view and channel.
#include "gp_common.h"

__device__ __forceinline__ float sh_basis_k(int k, float x, float y, float z) {
    const float xx = x * x, yy = y * y, zz = z * z;
    switch (k) {
    case 0: return 0.28209479177387814f;
    case 1: return -0.4886025119029199f * y;
    case 2: return 0.4886025119029199f * z;
    case 3: return -0.4886025119029199f * x;
    case 4: return 1.0925484305920792f * (x * y);
    case 5: return -1.0925484305920792f * (y * z);
    case 6: return 0.31539156525252005f * (2.f * zz - xx - yy);
    case 7: return -1.0925484305920792f * (x * z);
    case 8: return 0.5462742152960396f * (xx - yy);
    case 9: return -0.5900435899266435f * y * (3.f * xx - yy);
    case 10: return 2.890611442640554f * (x * y) * z;
    case 11: return -0.4570457994644658f * y * (4.f * zz - xx - yy);
    case 12: return 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
    case 13: return -0.4570457994644658f * x * (4.f * zz - xx - yy);
    case 14: return 1.445305721320277f * z * (xx - yy);
    default: return -0.5900435899266435f * x * (xx - 3.f * yy);
    }
}

__global__ __launch_bounds__(256) void gp_sh_factor_gradient_kernel(long n, int world, const float* __restrict__ factors, int n_coef,
                                                                   float* __restrict__ g_dc, float* __restrict__ g_rest) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = t >> 4;
    const int k = (int)(t & 15);
    if (i >= n) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (k < n_coef) {
        for (int v = 0; v < world; ++v) {
            const float* f = factors + ((size_t)v * n + i) * 6;
            const float y = sh_basis_k(k, f[3], f[4], f[5]);
            a0 = fmaf(y, f[0], a0); a1 = fmaf(y, f[1], a1); a2 = fmaf(y, f[2], a2);
        }
    }
    float* dst = k == 0 ? g_dc + 3 * i : g_rest + (size_t)i * 45 + 3 * (k - 1);
    dst[0] = a0; dst[1] = a1; dst[2] = a2;
}

extern "C" int gp_sh_factor_gradient(int64_t n, int32_t world, const float* factors, int32_t sh_degree, float* g_dc, float* g_rest,
                                     gp_stream_t stream_) {
    if (n < 0 || world < 1 || sh_degree < 0 || sh_degree > 3) GP_FAIL("gp_sh_factor_gradient: n >= 0, world >= 1, sh_degree 0..3");
    if (n == 0) return 0;
    if (!factors || !g_dc || !g_rest) GP_FAIL("gp_sh_factor_gradient: null pointer");
    hipLaunchKernelGGL(gp_sh_factor_gradient_kernel, dim3(gp_blocks((size_t)n * 16, 256)), dim3(256), 0, (hipStream_t)stream_, (long)n, (int)world, factors,
                       (sh_degree + 1) * (sh_degree + 1), g_dc, g_rest);
    GP_LAUNCH_CHECK();
    return 0;
}

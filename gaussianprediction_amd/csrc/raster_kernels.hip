// raster_kernels.hip -- the rasterizer kernels for gfx950 (wave64), hand-written:
//   preprocess fwd/bwd  (per Gaussian: projection, EWA cov2D, conic, radius, SH->RGB)
//   binning helpers     (depth keys, per-tile duplication, tile ranges, heavy-first launch order)
//   composite forward   (per 16x16 tile; 4 waves = 4 quadrants of 8x8, one pixel per lane; LDS-staged splat
//                        batches; exact per-quadrant culling, compacted slot lists; front-to-back alpha blend +
//                        depth + tidx; saves the per-instance quadrant masks)
//   composite backward  (SPLAT-parallel per (tile, quadrant): lanes own splats, pixels are walked uniformly;
//                        transmittance and suffix sums come from wave scans; gradients flushed as whole lines)
// Arithmetic mirrors oracle/gp_oracle.c expression-for-expression (explicit fmaf, built with
// -ffp-contract=off) so the discrete results (radii, tile rects, depth keys, per-tile order) are
// bit-identical to the float32 oracle.  Replaces the CUDA kernels of the reference's absent
// submodule `diff-gaussian-rasterization-w-depth` [/root/reference/.gitmodules:4-6]; call sites
// gaussian_renderer/__init__.py:98-106.
#include "gp_common.h"
#include "raster_kernels.h"

// SH constants [REF utils/sh_utils.py:26-44]
#define GP_LOG2E 1.4426950408889634f      // 0x3fb8aa3b
__device__ static const float SH_C0 = 0.28209479177387814f;
__device__ static const float SH_C1 = 0.4886025119029199f;
__device__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

// ------------------------------------------------------------------------------------------------
// shared per-Gaussian math
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float3 xform4x3(const float* __restrict__ m, float x, float y, float z) {
    float3 o;
    o.x = fmaf(m[0], x, fmaf(m[4], y, fmaf(m[8], z, m[12])));
    o.y = fmaf(m[1], x, fmaf(m[5], y, fmaf(m[9], z, m[13])));
    o.z = fmaf(m[2], x, fmaf(m[6], y, fmaf(m[10], z, m[14])));
    return o;
}
__device__ __forceinline__ float4 xform4x4(const float* __restrict__ m, float x, float y, float z) {
    float4 o;
    o.x = fmaf(m[0], x, fmaf(m[4], y, fmaf(m[8], z, m[12])));
    o.y = fmaf(m[1], x, fmaf(m[5], y, fmaf(m[9], z, m[13])));
    o.z = fmaf(m[2], x, fmaf(m[6], y, fmaf(m[10], z, m[14])));
    o.w = fmaf(m[3], x, fmaf(m[7], y, fmaf(m[11], z, m[15])));
    return o;
}
__device__ __forceinline__ void quat_to_R(float r, float x, float y, float z, float* Rm) {
    Rm[0] = 1.f - 2.f * (y * y + z * z);
    Rm[1] = 2.f * (x * y - r * z);
    Rm[2] = 2.f * (x * z + r * y);
    Rm[3] = 2.f * (x * y + r * z);
    Rm[4] = 1.f - 2.f * (x * x + z * z);
    Rm[5] = 2.f * (y * z - r * x);
    Rm[6] = 2.f * (x * z - r * y);
    Rm[7] = 2.f * (y * z + r * x);
    Rm[8] = 1.f - 2.f * (x * x + y * y);
}
// cov3D = (R S)(R S)^T  [REF scene/gaussian_model.py:35-39, utils/general_utils.py:101-110]
__device__ __forceinline__ void compute_cov3D(const float* __restrict__ scale, float mod, const float* __restrict__ q,
                                              float* c6) {
    float Rm[9];
    quat_to_R(q[0], q[1], q[2], q[3], Rm);
    const float s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    float L[9];
    L[0] = Rm[0] * s0; L[1] = Rm[1] * s1; L[2] = Rm[2] * s2;
    L[3] = Rm[3] * s0; L[4] = Rm[4] * s1; L[5] = Rm[5] * s2;
    L[6] = Rm[6] * s0; L[7] = Rm[7] * s1; L[8] = Rm[8] * s2;
    c6[0] = fmaf(L[0], L[0], fmaf(L[1], L[1], L[2] * L[2]));
    c6[1] = fmaf(L[0], L[3], fmaf(L[1], L[4], L[2] * L[5]));
    c6[2] = fmaf(L[0], L[6], fmaf(L[1], L[7], L[2] * L[8]));
    c6[3] = fmaf(L[3], L[3], fmaf(L[4], L[4], L[5] * L[5]));
    c6[4] = fmaf(L[3], L[6], fmaf(L[4], L[7], L[5] * L[8]));
    c6[5] = fmaf(L[6], L[6], fmaf(L[7], L[7], L[8] * L[8]));
}

struct ProjCtx {
    float T0[3], T1[3];
    float tx, ty, tz, gx, gy;
};
__device__ __forceinline__ void compute_cov2D(float3 pv, float fx, float fy, float tanfovx, float tanfovy,
                                              const float* c6, const float* __restrict__ view, float* abc,
                                              ProjCtx* ctx) {
    const float tz = pv.z;
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = pv.x / tz, tytz = pv.y / tz;
    const float cx = fminf(limx, fmaxf(-limx, txtz));
    const float cy = fminf(limy, fmaxf(-limy, tytz));
    const float tx = cx * tz, ty = cy * tz;
    const float itz = 1.f / tz;
    const float itz2 = itz * itz;
    const float J00 = fx * itz;
    const float J02 = -(fx * tx) * itz2;
    const float J11 = fy * itz;
    const float J12 = -(fy * ty) * itz2;
    const float W0[3] = {view[0], view[4], view[8]};
    const float W1[3] = {view[1], view[5], view[9]};
    const float W2[3] = {view[2], view[6], view[10]};
    float T0[3], T1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T0[k] = fmaf(J00, W0[k], J02 * W2[k]);
        T1[k] = fmaf(J11, W1[k], J12 * W2[k]);
    }
    const float S0[3] = {c6[0], c6[1], c6[2]};
    const float S1[3] = {c6[1], c6[3], c6[4]};
    const float S2[3] = {c6[2], c6[4], c6[5]};
    float u0[3], u1[3];
    u0[0] = fmaf(S0[0], T0[0], fmaf(S0[1], T0[1], S0[2] * T0[2]));
    u0[1] = fmaf(S1[0], T0[0], fmaf(S1[1], T0[1], S1[2] * T0[2]));
    u0[2] = fmaf(S2[0], T0[0], fmaf(S2[1], T0[1], S2[2] * T0[2]));
    u1[0] = fmaf(S0[0], T1[0], fmaf(S0[1], T1[1], S0[2] * T1[2]));
    u1[1] = fmaf(S1[0], T1[0], fmaf(S1[1], T1[1], S1[2] * T1[2]));
    u1[2] = fmaf(S2[0], T1[0], fmaf(S2[1], T1[1], S2[2] * T1[2]));
    abc[0] = fmaf(T0[0], u0[0], fmaf(T0[1], u0[1], T0[2] * u0[2]));
    abc[1] = fmaf(T0[0], u1[0], fmaf(T0[1], u1[1], T0[2] * u1[2]));
    abc[2] = fmaf(T1[0], u1[0], fmaf(T1[1], u1[1], T1[2] * u1[2]));
    if (ctx) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { ctx->T0[k] = T0[k]; ctx->T1[k] = T1[k]; }
        ctx->tx = tx; ctx->ty = ty; ctx->tz = tz;
        ctx->gx = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        ctx->gy = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int f2i_sat(float v) { return (int)fminf(1e9f, fmaxf(-1e9f, v)); }

__device__ __forceinline__ void tile_rect(float pix, float piy, float rad_f, int gx, int gy, int& minx, int& miny,
                                          int& maxx, int& maxy) {
    minx = clampi(f2i_sat((pix - rad_f) / (float)GP_TILE), 0, gx);
    miny = clampi(f2i_sat((piy - rad_f) / (float)GP_TILE), 0, gy);
    maxx = clampi(f2i_sat((pix + rad_f + (float)(GP_TILE - 1)) / (float)GP_TILE), 0, gx);
    maxy = clampi(f2i_sat((piy + rad_f + (float)(GP_TILE - 1)) / (float)GP_TILE), 0, gy);
}

// [REF utils/sh_utils.py:57-112], shs layout [M][3]
__device__ __forceinline__ void sh_to_rgb(int deg, const float* __restrict__ sh, float x, float y, float z, float* out3) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float res = SH_C0 * sh[0 * 3 + ch];
        if (deg > 0) {
            res = res - SH_C1 * y * sh[1 * 3 + ch] + SH_C1 * z * sh[2 * 3 + ch] - SH_C1 * x * sh[3 * 3 + ch];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * sh[4 * 3 + ch] + SH_C2[1] * yz * sh[5 * 3 + ch] +
                      SH_C2[2] * (2.f * zz - xx - yy) * sh[6 * 3 + ch] + SH_C2[3] * xz * sh[7 * 3 + ch] +
                      SH_C2[4] * (xx - yy) * sh[8 * 3 + ch];
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.f * xx - yy) * sh[9 * 3 + ch] + SH_C3[1] * xy * z * sh[10 * 3 + ch] +
                          SH_C3[2] * y * (4.f * zz - xx - yy) * sh[11 * 3 + ch] +
                          SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[12 * 3 + ch] +
                          SH_C3[4] * x * (4.f * zz - xx - yy) * sh[13 * 3 + ch] + SH_C3[5] * z * (xx - yy) * sh[14 * 3 + ch] +
                          SH_C3[6] * x * (xx - 3.f * yy) * sh[15 * 3 + ch];
                }
            }
        }
        out3[ch] = res;
    }
}

// ------------------------------------------------------------------------------------------------
// preprocess forward.  SH coefficients ([N,16,3] = 192 B per Gaussian, or the model's two tensors
// features_dc [N,1,3] + features_rest [N,15,3] passed separately so the per-frame torch.cat of
// [REF scene/gaussian_model.py:155-159] disappears) are staged through LDS: the 256 Gaussians of a
// workgroup own one contiguous span of global memory, read with coalesced float4 loads, then every
// thread picks its own coefficients at an odd LDS stride (conflict-free).
// SH_MODE 0: generic (direct global loads)  1: one tensor, M = 16  2: dc + rest, M = 16
// ------------------------------------------------------------------------------------------------
template <int CNT>
__device__ __forceinline__ void stage_sh(float* s_sh, const float* __restrict__ src, int nblk, int tid) {
    constexpr int STRIDE = CNT | 1;
    const int total = nblk * CNT;
    int e = tid * 4;
    for (; e + 3 * 1024 + 3 < total; e += 4 * 1024) {      // four 16-byte loads in flight per thread (see adam_unstage_sh)
        float4 f[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) f[w] = *(const float4*)(src + e + w * 1024);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* fv = (const float*)&f[w];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = e + w * 1024 + u;
                const int g = idx / CNT, k = idx - g * CNT;
                s_sh[g * STRIDE + k] = fv[u];
            }
        }
    }
    for (; e < total; e += 1024) {
        float v[4];
        if (e + 3 < total) {
            const float4 f = *(const float4*)(src + e);
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (e + u < total) ? src[e + u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = e + u;
            if (idx < total) {
                const int g = idx / CNT, k = idx - g * CNT;
                s_sh[g * STRIDE + k] = v[u];
            }
        }
    }
}
// A full workgroup's span (256 Gaussians) with EVERY load in flight at once: one round trip to memory instead of three (the rolled
// form above waits for four loads per trip).  12 float4 per thread; the last trip of a span that is not a multiple of 4 KB is clamped
// to a valid address and dropped at the LDS write.
template <int CNT>
__device__ __forceinline__ void stage_sh_full(float* s_sh, const float* __restrict__ src, int tid) {
    constexpr int STRIDE = CNT | 1, TOTAL = 256 * CNT, NL = (TOTAL + 1023) / 1024;
    static_assert(TOTAL % 4 == 0, "float4 trips");
    float4 f[NL];
#pragma unroll
    for (int w = 0; w < NL; ++w) {
        const int e = tid * 4 + w * 1024;
        f[w] = *(const float4*)(src + (e + 3 < TOTAL ? e : TOTAL - 4));
    }
#pragma unroll
    for (int w = 0; w < NL; ++w) {
        const int e = tid * 4 + w * 1024;
        if (e + 3 < TOTAL) {
            const float* fv = (const float*)&f[w];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = e + u;
                const int g = idx / CNT, k = idx - g * CNT;
                s_sh[g * STRIDE + k] = fv[u];
            }
        }
    }
}
template <int CNT>
__device__ __forceinline__ void unstage_sh(const float* s_sh, float* __restrict__ dst, int nblk, int tid, bool accumulate) {
    constexpr int STRIDE = CNT | 1;
    const int total = nblk * CNT;
    for (int e = tid * 4; e < total; e += 1024) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = e + u;
            const int g = idx / CNT, k = idx - g * CNT;
            v[u] = (idx < total) ? s_sh[g * STRIDE + k] : 0.f;
        }
        if (e + 3 < total) {
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (accumulate) { const float4 c = *(const float4*)(dst + e); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
            *(float4*)(dst + e) = o;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u < total) dst[e + u] = accumulate ? dst[e + u] + v[u] : v[u];
        }
    }
}

// view-dependent colour of Gaussian i from its SH coefficients [REF gaussian_renderer/__init__.py:86-91, utils/sh_utils.py:57-112]:
// clamp_min(eval_sh(...) + 0.5, 0); returns the per-channel clamp flags.  One statement of the arithmetic, used by the fused
// preprocess kernel and by the late colour kernel below (bit-identical results).
template <int SH_MODE>
__device__ __forceinline__ uint8_t sh_color(const RasterDims& d, int i, int tid, float px, float py, float pz,
                                            const float* __restrict__ campos, const float* __restrict__ shs, const float* s_sh,
                                            float* col) {
    const float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
    const float len = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
    const float inv = 1.f / len;
    float raw[3];
    if (SH_MODE == 0) {
        sh_to_rgb(d.D, shs + (size_t)i * d.M * 3, dx * inv, dy * inv, dz * inv, raw);
    } else {
        float shl[48];
        if (SH_MODE == 1) {
            if (d.D > 0) {
#pragma unroll
                for (int k = 0; k < 48; ++k) shl[k] = s_sh[tid * 49 + k];
            } else {
                shl[0] = shs[(size_t)i * 48]; shl[1] = shs[(size_t)i * 48 + 1]; shl[2] = shs[(size_t)i * 48 + 2];
            }
        } else {
            shl[0] = shs[3 * (size_t)i]; shl[1] = shs[3 * (size_t)i + 1]; shl[2] = shs[3 * (size_t)i + 2];
            if (d.D > 0) {
#pragma unroll
                for (int k = 0; k < 45; ++k) shl[3 + k] = s_sh[tid * 45 + k];
            }
        }
        sh_to_rgb(d.D, shl, dx * inv, dy * inv, dz * inv, raw);
    }
    uint8_t cl = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = raw[k] + 0.5f;
        if (v < 0.f) cl |= (uint8_t)(1u << k);
        col[k] = fmaxf(v, 0.f);
    }
    return cl;
}

// The per-Gaussian half of the composite forward's sub-block test (gp_sb_mask below, which states the geometry): (dyr, inv_cx) from the
// record's own floats with the hardware rcp / sqrt -- once per Gaussian here instead of once per tile-splat instance in the staging lanes
// (R / N = 4 instances per Gaussian; three rcp and one sqrt each).  BUILT AND MEASURED, SHIPS OFF (GP_SB_HOIST = 0, raster_kernels.h):
// profiles/r06_sb_hoist_ab.txt -- through the record's spare word the composite forward does not move (0.1148 against 0.1150 ms), through
// a side array (a fourth gathered line per instance) it is 10 % slower: the staging phase waits for its gathers, not for its arithmetic.  inv_cx = NaN: do not cull (degenerate conic, or an extent beyond 1e8;
// an opacity below 1/255 gives NaN as well and is caught in front by the mask's own test of tau).
__device__ __forceinline__ float2 gp_sb_side(const float4 q0, const float4 q1) {
    const float cx = -2.f * q0.z, cy = -q0.w, cz = -2.f * q1.x;
    const float det = cx * cz - cy * cy;
    const float tau = q1.y + 7.994353436858858f;
    const float tt = 2.f * tau * 1.004f + 0.03f;
    const float ex = __builtin_amdgcn_sqrtf(tt * cz * __builtin_amdgcn_rcpf(det));
    const float inv_cx = 0.25f * __builtin_amdgcn_rcpf(cx);
    const float dyr = -(cy * __builtin_amdgcn_rcpf(cz)) * ex;
    const bool cull = det > 0.f && cx > 0.f && cz > 0.f && ex <= 1e8f;
    return make_float2(dyr, cull ? inv_cx : __uint_as_float(0x7fc00000u));
}

template <int SH_MODE>
__device__ __forceinline__ void preprocess_fwd_body(RasterDims d, const float* __restrict__ means3D,
                                                    const float* __restrict__ scales, const float* __restrict__ rotations,
                                                    const float* __restrict__ opacities, const float* __restrict__ shs,
                                                    const float* __restrict__ shs_rest, const float* __restrict__ colors_precomp,
                                                    const float* __restrict__ cov3D_precomp, const float* __restrict__ view,
                                                    const float* __restrict__ proj, const float* __restrict__ campos,
                                                    int32_t* __restrict__ radii, float4* __restrict__ rec,
                                                    uint32_t* __restrict__ depth_key, uint2* __restrict__ tiles_touched,
                                                    uint8_t* __restrict__ clamped) {
    __shared__ float s_sh[SH_MODE == 0 ? 1 : 256 * 49];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * 256;
    const int i = base + tid;
    const int nblk = min(256, d.N - base);
    const bool use_sh = !colors_precomp && !d.late_color;
    // the thread's own inputs, issued together IN FRONT of the SH staging (they ride on its first round trip) and pinned behind
    // it, before the culling branches (placed where they are used, the compiler sinks them behind each early return: position ->
    // wait -> rotation, scale -> wait -> opacity -> wait)
    const int ii = i < d.N ? i : d.N - 1;
    const float px = means3D[3 * ii], py = means3D[3 * ii + 1], pz = means3D[3 * ii + 2];
    float sc3[3] = {0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    if (!cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sc3[k] = scales[3 * ii + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] = rotations[4 * ii + k];
    }
    float opac = opacities[ii];
    if (SH_MODE == 1 && use_sh && d.D > 0) {
        if (nblk == 256) stage_sh_full<48>(s_sh, shs + (size_t)base * 48, tid);
        else stage_sh<48>(s_sh, shs + (size_t)base * 48, nblk, tid);
    }
    if (SH_MODE == 2 && use_sh && d.D > 0) {
        if (nblk == 256) stage_sh_full<45>(s_sh, shs_rest + (size_t)base * 45, tid);
        else stage_sh<45>(s_sh, shs_rest + (size_t)base * 45, nblk, tid);
    }
    if (SH_MODE != 0) __syncthreads();
    asm volatile("" ::"v"(px), "v"(py), "v"(pz), "v"(sc3[0]), "v"(sc3[1]), "v"(sc3[2]), "v"(q4[0]), "v"(q4[1]), "v"(q4[2]), "v"(q4[3]), "v"(opac));
    if (d.raw_opacity) {            // (gp_raster_settings.raw_activations: gp_act_fwd_kernel's expressions)
#pragma unroll
        for (int k = 0; k < 3; ++k) sc3[k] = expf(sc3[k]);
        opac = 1.f / (1.f + expf(-opac));
    }
    if (i >= d.N) return;
    if (i < d.n_zero) d.zero_words[i] = 0u;              // (only handed over when N >= n_zero)
    radii[i] = 0;
    if (d.visible) d.visible[i] = 0;
    tiles_touched[i] = make_uint2(0u, 0u);
    depth_key[i] = d.key_culled;
    clamped[i] = 0;
    const float3 pv = xform4x3(view, px, py, pz);
    if (!(pv.z > 0.2f)) return;
    const float4 ph = xform4x4(proj, px, py, pz);
    const float pw = 1.f / (ph.w + 0.0000001f);
    const float ndcx = ph.x * pw, ndcy = ph.y * pw;
    float c6[6];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
    } else {
        compute_cov3D(sc3, d.scale_mod, q4, c6);
    }
    float abc[3];
    compute_cov2D(pv, d.fx, d.fy, d.tanfovx, d.tanfovy, c6, view, abc, nullptr);
    const float a = abc[0] + 0.3f, b = abc[1], c = abc[2] + 0.3f;
    const float det = a * c - b * b;
    if (det == 0.f) return;
    const float det_inv = 1.f / det;
    const float conx = c * det_inv, cony = -b * det_inv, conz = a * det_inv;
    const float mid = 0.5f * (a + c);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float l1 = mid + sq, l2 = mid - sq;
    const float rad_f = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
    const float pix = ((ndcx + 1.f) * (float)d.W - 1.f) * 0.5f;
    const float piy = ((ndcy + 1.f) * (float)d.H - 1.f) * 0.5f;
    int minx, miny, maxx, maxy;
    tile_rect(pix, piy, rad_f, d.gx, d.gy, minx, miny, maxx, maxy);
    if ((maxx - minx) * (maxy - miny) == 0) return;
    float col[3] = {0.f, 0.f, 0.f};
    uint8_t cl = 0;
    if (colors_precomp) {
        col[0] = colors_precomp[3 * i]; col[1] = colors_precomp[3 * i + 1]; col[2] = colors_precomp[3 * i + 2];
    } else if (!d.late_color) {
        cl = sh_color<SH_MODE>(d, i, tid, px, py, pz, campos, shs, s_sh, col);
    }
    const int rad_i = f2i_sat(rad_f);
    radii[i] = rad_i;
    if (d.visible) d.visible[i] = rad_i > 0;
    clamped[i] = cl;
    {
        const uint32_t key = __float_as_uint(pv.z) - d.key_base;      // (key_base = 0 without a promise)
        depth_key[i] = key;
        // the caller's promise about the keys' range (gp_raster_settings.depth_key_bits), checked for every visible Gaussian:
        // a plain store of THIS call's tag (every writer writes the same value) into the status block's scratch word; the kernel that
        // writes {R, overflow} later in the call turns `scratch == tag` into the overflow flag -- no word has to be cleared between
        // frames, a stale tag of an earlier call never matches
        if (key & d.key_hi) *d.key_flag = d.key_tag;      // (key_hi = 0 without a promise; a key below the base wraps around)
    }
    // tile rectangle (first tile | extent, 16 bits each): the binning stage expands it without touching `rec` again
    tiles_touched[i] = make_uint2((uint32_t)minx | ((uint32_t)miny << 16), (uint32_t)(maxx - minx) | ((uint32_t)(maxy - miny) << 16));
    // The composite evaluates alpha = min(0.99, opacity exp(power)) as exp2(power' + log2 opacity) with power' = log2(e) power:
    // the record carries the quadratic form pre-scaled by log2(e) and log2(opacity) beside the opacity itself (GP_LOG2E; the
    // oracle restates the same products), which takes two multiplies out of every (pixel, splat) evaluation of both passes.
    rec[3 * (size_t)i + 0] = make_float4(pix, piy, (-0.5f * conx) * GP_LOG2E, (-cony) * GP_LOG2E);
    {
        [[maybe_unused]] const float4 r0 = make_float4(pix, piy, (-0.5f * conx) * GP_LOG2E, (-cony) * GP_LOG2E);
        float4 r1 = make_float4((-0.5f * conz) * GP_LOG2E, log2f(opac), pv.z, __int_as_float(i));
#if GP_SB_HOIST
        const float2 side = gp_sb_side(r0, r1);
#endif
#if GP_SB_HOIST == 2
        r1.w = side.y != side.y ? side.y : side.x;      // the word that held the Gaussian's id (read by nobody): dyr, or NaN for "do not cull"
#elif GP_SB_HOIST == 1
        if (d.sb_side) d.sb_side[i] = side;
#endif
        rec[3 * (size_t)i + 1] = r1;
    }
    rec[3 * (size_t)i + 2] = make_float4(col[0], col[1], col[2], opac);
}

#define PF_ARGS RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales, \
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs, \
    const float* __restrict__ shs_rest, const float* __restrict__ colors_precomp, const float* __restrict__ cov3D_precomp, \
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos, \
    int32_t* __restrict__ radii, float4* __restrict__ rec, uint32_t* __restrict__ depth_key, \
    uint2* __restrict__ tiles_touched, uint8_t* __restrict__ clamped
#define PF_PASS d, means3D, scales, rotations, opacities, shs, shs_rest, colors_precomp, cov3D_precomp, view, proj, campos, \
    radii, rec, depth_key, tiles_touched, clamped
__global__ __launch_bounds__(256) void gp_preprocess_fwd_kernel(PF_ARGS) { preprocess_fwd_body<0>(PF_PASS); }
__global__ __launch_bounds__(256) void gp_preprocess_fwd_sh16_kernel(PF_ARGS) { preprocess_fwd_body<1>(PF_PASS); }
__global__ __launch_bounds__(256) void gp_preprocess_fwd_split_kernel(PF_ARGS) { preprocess_fwd_body<2>(PF_PASS); }

// SH -> RGB as a kernel of its own (gp_raster_settings.sh_ready_event): the view-parallel harness updates and all-gathers the
// SH coefficients (3/4 of all parameter bytes) asynchronously; projection, both sorts and the binning do not read them, so
// the forward waits for them only here, right in front of the composite -- ~0.3 ms of cover for the exchange.  Same arithmetic
// as the fused kernel (sh_color), visible Gaussians only; writes the colour third of the 48-byte record and the clamp flags.
template <int SH_MODE>
__device__ __forceinline__ void sh_color_body(RasterDims d, const float* __restrict__ means3D, const float* __restrict__ shs,
                                              const float* __restrict__ shs_rest, const float* __restrict__ campos,
                                              const int32_t* __restrict__ radii, float4* __restrict__ rec, uint8_t* __restrict__ clamped) {
    __shared__ float s_sh[SH_MODE == 0 ? 1 : 256 * 49];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * 256;
    const int i = base + tid;
    const int nblk = min(256, d.N - base);
    if (SH_MODE == 1 && d.D > 0) stage_sh<48>(s_sh, shs + (size_t)base * 48, nblk, tid);
    if (SH_MODE == 2 && d.D > 0) stage_sh<45>(s_sh, shs_rest + (size_t)base * 45, nblk, tid);
    if (SH_MODE != 0) __syncthreads();
    if (i >= d.N || radii[i] <= 0) return;
    float col[3];
    const uint8_t cl = sh_color<SH_MODE>(d, i, tid, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], campos, shs, s_sh, col);
    clamped[i] = cl;
    float* c3 = (float*)(rec + 3 * (size_t)i + 2);       // (the fourth word is the opacity, written by the projection kernel)
    c3[0] = col[0]; c3[1] = col[1]; c3[2] = col[2];
}
#define SC_ARGS RasterDims d, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest, \
    const float* __restrict__ campos, const int32_t* __restrict__ radii, float4* __restrict__ rec, uint8_t* __restrict__ clamped
__global__ __launch_bounds__(256) void gp_sh_color_kernel(SC_ARGS) { sh_color_body<0>(d, means3D, shs, shs_rest, campos, radii, rec, clamped); }
__global__ __launch_bounds__(256) void gp_sh_color_sh16_kernel(SC_ARGS) { sh_color_body<1>(d, means3D, shs, shs_rest, campos, radii, rec, clamped); }
__global__ __launch_bounds__(256) void gp_sh_color_split_kernel(SC_ARGS) { sh_color_body<2>(d, means3D, shs, shs_rest, campos, radii, rec, clamped); }

// {min, max} of the visible Gaussians' depth keys (gp_raster_settings.depth_key_range; out2 = {0xFFFFFFFF, 0} on entry): one pair
// of atomics per workgroup of 1024 keys -- a set-up-step kernel, not part of the timed path
__global__ __launch_bounds__(256) void gp_key_range_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ radii, int n,
                                                          uint32_t base, uint32_t* __restrict__ out2) {
    __shared__ uint32_t s_mn[4], s_mx[4];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    const int first = blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = first + 256 * u;
        if (i < n && radii[i] > 0) { const uint32_t k = keys[i] + base; mn = min(mn, k); mx = max(mx, k); }
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, dd)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, dd)); }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
        mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
        if (mn <= mx) { atomicMin(out2, mn); atomicMax(out2 + 1, mx); }
    }
}

__global__ __launch_bounds__(256) void gp_mark_visible_kernel(int n, const float* __restrict__ means3D,
                                                             const float* __restrict__ view, uint8_t* __restrict__ present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float3 pv = xform4x3(view, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    present[i] = pv.z > 0.2f ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// binning helpers
// ------------------------------------------------------------------------------------------------
// Expand every visible Gaussian into its (tile key, id) instances at offset[i] .. (exclusive prefix sum of the tile counts over
// the depth order).  The prefix arrives in two pieces -- `offsets` scanned inside blocks of GP_SCAN_TILE entries and the blocks'
// totals (gp_scan_blocks_u32) -- and every workgroup adds up the totals in front of its block itself (a few hundred values, one
// reduction): one scan launch instead of three.  `total[0]` = R.
// Wave-cooperative: the 64 Gaussians of a wave own one CONTIGUOUS run of instances, so the wave walks that run 64 entries
// at a time -- each lane finds its entry's owner by a binary search over the wave's relative offsets (LDS) -- and every
// store is one coalesced 256-byte line; the tile rectangles arrive in depth order from the depth sort's last pass
// (GpSortEpilogue), so nothing is gathered here.  (One thread per Gaussian re-deriving its rectangle from `rec[id]` and looping over its own tiles: three
// random gathers per Gaussian, 64-way scattered stores, and one large footprint serialising its whole wave.)
__global__ __launch_bounds__(256) void gp_duplicate_kernel(RasterDims d, const uint32_t* __restrict__ sorted_ids,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ total_R,
                                                          const uint2* __restrict__ rect_sorted,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t capacity,
                                                          uint32_t* __restrict__ status, uint32_t n_dup_blocks, uint32_t key_tag) {
    __shared__ int4 s_own[4][64];        // (relative offset, first tile x, first tile y, tiles per row)
    __shared__ uint32_t s_id[4][64];
    __shared__ uint32_t s_part[4];
    if (blockIdx.x >= n_dup_blocks) {
        // capacity mode, the blocks behind the expansion: status = {R, R > capacity} and sentinel keys behind the R real
        // instances (their values are never read: no tile range covers them) -- two tiny kernels folded into this launch
        const uint32_t R = gp_total_of(total_R);
        if (blockIdx.x == n_dup_blocks && threadIdx.x == 0) { status[0] = R; status[1] = (R > capacity || (key_tag && status[2] == key_tag)) ? 1u : 0u; }
        const uint32_t b0 = (blockIdx.x - n_dup_blocks) * 4096u;
        for (uint32_t i = b0 + threadIdx.x; i < b0 + 4096u && i < capacity; i += 256u)
            if (i >= R) keys[i] = 0xFFFFFFFFu;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t before = 0;                                     // instances of the scan blocks in front of this workgroup's
    {
        const int nsb = (blockIdx.x * 256) / GP_SCAN_TILE;   // (256 consecutive Gaussians never straddle a scan block)
        uint32_t acc = 0;
        for (int b = threadIdx.x; b < nsb; b += 256) acc += block_sums[b];
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) acc += __shfl_xor(acc, dd);
        if (lane == 0) s_part[wave] = acc;
        __syncthreads();
        before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
    const int i0 = blockIdx.x * 256 + wave * 64;             // uniform per wave
    if (i0 >= d.N) return;
    const int i = i0 + lane;
    uint32_t id = 0;
    int cnt = 0, minx = 0, miny = 0, w = 1;
    if (i < d.N) {                                           // everything is read in depth order: coalesced, no gathers
        id = sorted_ids[i];
        const uint2 r = rect_sorted[i];
        minx = (int)(r.x & 0xFFFFu); miny = (int)(r.x >> 16);
        w = (int)(r.y & 0xFFFFu);
        cnt = w * (int)(r.y >> 16);
        if (w < 1) w = 1;
    }
    const uint32_t base = before + offsets[i0];
    const int incl = gp_wave_scan_add(cnt);                  // inclusive scan of the counts over the wave
    const int total = __shfl(incl, 63);
    s_own[wave][lane] = make_int4(incl - cnt, minx, miny, w);
    s_id[wave][lane] = id;
    __builtin_amdgcn_wave_barrier();
    for (int b = 0; b < total; b += 64) {
        const int j = b + lane;
        if (j < total) {
            int lo = 0;                                      // the last lane whose run starts at or before j
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1)
                if (s_own[wave][lo + step].x <= j) lo += step;   // lo + step <= 63 always
            const int4 o = s_own[wave][lo];
            const int k = j - o.x;
            const int yy = k / o.w, xx = k - yy * o.w;
            const uint32_t off = base + (uint32_t)j;
            if (off < capacity) {       // capacity-mode overflow: the list is cut at the capacity (flagged in binning_status);
                keys[off] = (uint32_t)((o.z + yy) * d.gx + o.y + xx);   // every slot below it is still written: no stale id
                vals[off] = s_id[wave][lo];
            }
        }
    }
}
__global__ __launch_bounds__(256) void gp_tile_ranges_kernel(const uint32_t* __restrict__ keys, uint32_t R, uint32_t n_tiles,
                                                            int2* __restrict__ ranges) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= R) return;
    const uint32_t t = keys[k], tp = keys[k > 0 ? k - 1 : 0], tn = keys[k + 1 < R ? k + 1 : R - 1];   // (three loads, one round trip)
    if (t >= n_tiles) return;                            // sentinel padding of the capacity mode (sorted behind every tile)
    if (k == 0 || tp != t) ranges[t].x = (int)k;
    if (k == R - 1 || tn != t) ranges[t].y = (int)(k + 1);
}
// exact mode with a status word requested: status = {R, 0}  (capacity mode writes it from the duplicate launch)
__global__ void gp_binning_status_kernel(const uint32_t* __restrict__ total, uint32_t capacity, uint32_t* __restrict__ status, uint32_t key_tag) {
    const uint32_t R = gp_total_of(total);
    status[0] = R;
    status[1] = (R > capacity || (key_tag && status[2] == key_tag)) ? 1u : 0u;      // (key_tag: gp_raster_settings.depth_key_bits)
}

// Heavy-first launch order.  Per-tile work varies by >10x; the hardware dispatches workgroups in blockIdx order, so
// with natural order the last wave of workgroups contains a few very long tiles and the chip idles behind them.
// One workgroup buckets the tiles by a 2-mantissa-bit logarithm of their work estimate (counting sort, descending);
// the order within a bucket only affects scheduling, never results.
// work = ranges length, or min(length, work_hint) when the forward's per-tile "last contributor" is known.
// Round 4: every tile's range / work hint is loaded up front (as two `for (t = tid; ...) atomicAdd(.. ranges[t] ..)` loops the
// kernel was twelve dependent round trips to L2: 8 us for 5 440 tiles, twice per step) and the 128 bucket bases come from a
// parallel prefix instead of one thread's loop.  (A STABLE variant -- tiles of a bucket in row-major order, ranks from match-any
// ballots -- was built too: neighbouring tiles then run together and share Gaussians in L2; it measured 10.8 us against 8.4 for
// no change in the composite kernels, and went.)
#define TO_MAX 8                    // tiles per thread held in registers: T <= 8192 on the fast path
__device__ __forceinline__ void gp_tile_order_body(const int2* __restrict__ ranges, const int32_t* __restrict__ work_hint, int T,
                                                   uint32_t* __restrict__ order) {
    __shared__ uint32_t s_cnt[128], s_base[128];
    const int tid = threadIdx.x;
    auto bucket_of_w = [](int w) { return gp_tile_bucket(w); };
    if (tid < 128) s_cnt[tid] = 0;
    int bk[TO_MAX];
    const bool fast = T <= TO_MAX * 1024;
    if (fast) {
        int2 rg[TO_MAX];
        int wh[TO_MAX];
#pragma unroll
        for (int r = 0; r < TO_MAX; ++r) {
            const int t = r * 1024 + tid;
            rg[r] = ranges[t < T ? t : T - 1];
            wh[r] = work_hint ? work_hint[t < T ? t : T - 1] : 0x7fffffff;
        }
#pragma unroll
        for (int r = 0; r < TO_MAX; ++r) bk[r] = bucket_of_w(min(rg[r].y - rg[r].x, wh[r]));
    }
    auto bucket_of = [&](int t) { const int2 r = ranges[t]; int w = r.y - r.x; if (work_hint) w = min(w, work_hint[t]); return bucket_of_w(w); };
    __syncthreads();
    if (fast) {
#pragma unroll
        for (int r = 0; r < TO_MAX; ++r) if (r * 1024 + tid < T) atomicAdd(&s_cnt[bk[r]], 1u);
    } else {
        for (int t = tid; t < T; t += 1024) atomicAdd(&s_cnt[bucket_of(t)], 1u);
    }
    __syncthreads();
    if (tid < 128) {                // (every thread sums the counts in front of its bucket: 128 independent LDS reads, no serial chain)
        uint32_t run = 0;
#pragma unroll 16
        for (int b = 0; b < 128; ++b) { const uint32_t c = s_cnt[b]; run += b < tid ? c : 0u; }
        s_base[tid] = run;
    }
    __syncthreads();
    if (fast) {
#pragma unroll
        for (int r = 0; r < TO_MAX; ++r) if (r * 1024 + tid < T) order[atomicAdd(&s_base[bk[r]], 1u)] = (uint32_t)(r * 1024 + tid);
    } else {
        for (int t = tid; t < T; t += 1024) order[atomicAdd(&s_base[bucket_of(t)], 1u)] = (uint32_t)t;
    }
}
__global__ __launch_bounds__(1024) void gp_tile_order_kernel(const int2* __restrict__ ranges, const int32_t* __restrict__ work_hint,
                                                             int T, uint32_t* __restrict__ order) {
    gp_tile_order_body(ranges, work_hint, T, order);
}
// The backward's prologue in one launch: workgroup 0 orders the tiles (by the forward's per-tile last contributor) while the
// others clear the per-Gaussian gradient accumulators the composite backward adds into -- a 9 us single-workgroup kernel and a
// 40 MB fill that used to run one after the other.
__global__ __launch_bounds__(1024) void gp_bwd_prologue_kernel(const int2* __restrict__ ranges, const int32_t* __restrict__ work_hint, int T,
                                                               uint32_t* __restrict__ order, float* __restrict__ acc, size_t acc_floats) {
    if (blockIdx.x == 0) { gp_tile_order_body(ranges, work_hint, T, order); return; }
    const size_t n4 = acc_floats >> 2, stride = (size_t)(gridDim.x - 1) * 1024;
    float4* a4 = reinterpret_cast<float4*>(acc);
    for (size_t i = (size_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < n4; i += stride) a4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 1 && threadIdx.x < (acc_floats & 3)) acc[(n4 << 2) + threadIdx.x] = 0.f;
}

#define CF_THREADS 256
typedef float v2f __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// composite forward, sub-block lists.  One 256-thread workgroup (4 waves) per 16x16 tile; splat records (48 B) are gathered
// once per tile into LDS in batches of 256.  (Round 1's kernel -- wave = 8x8 quadrant, one list per quadrant -- was kept as an
// A/B baseline through round 3 and is gone: round 4 changed the alpha expression, see the record format in
// preprocess_fwd_body.)
//
// Measured on MI355X that quadrant kernel was bound by the vector ALU (41 issue slots per wave-visit x 4 cycles
// x 3.2 M wave-visits = its whole 0.22 ms), with only one lane in three contributing: a splat that touches an 8x8
// quadrant reaches alpha >= 1/255 on a third of its pixels.  This kernel culls at 4x4 granularity instead:
//   * a tile is 4 x 4 sub-blocks (SB) of 4 x 4 pixels; wave w still owns quadrant w, but its four 16-lane groups are
//     the quadrant's four SBs, and EACH GROUP WALKS ITS OWN LIST of the batch's splats (lists in LDS as byte offsets
//     of the staged records, so a visit is one ds_read_u16 + three record reads at up to four distinct addresses);
//   * the staging lane derives the exact 16-bit SB mask of its splat from four row strips: over the strip
//     dy in [d0, d0 + 3] the alpha >= 1/255 ellipse spans x in [c(dl) - h(dl), c(dr) + h(dr)], where dr / dl are the
//     strip's points nearest to the ellipse's rightmost / leftmost point -- two sqrt per strip, about the cost of the
//     four rectangle tests it replaces;
//   * 'done' and 'list exhausted' are ONE per-lane limit (i < lim), so a visit carries one compare for both.
// Wave-steps per tile drop from 4 x 146 to 4 x 96 (lock-step over four lists costs 14 % against ideal 4x4 culling) and
// a visit from 41 to 29 VALU slots (round 2), 19 - 20 with the exponent folded into the record (round 4):
//     e = dx (A' dx + B' dy) + ((C' dy) dy + lop),   A' B' C' = log2(e) x the quadratic form, lop = log2(opacity)
//     skip if e > lop (i.e. power > 0);  alpha = min(0.99, exp2(e));  skip if alpha < 1/255;  w = alpha T;  stop if T - w < 1e-4
// -- the published algorithm's min(0.99, opacity exp(power)) with the two multiplies moved into the per-Gaussian record.
// ------------------------------------------------------------------------------------------------
#define CF2_REC 48            // bytes per staged record: (x, y, A', B') (C', log2 opacity, r, g) (b, depth, -, -)
__device__ __forceinline__ uint32_t gp_sb_mask(const float4 q0, const float4 q1, const float2 side, float X0, float Y0) {
    // alpha >= 1/255  <=>  q'(d) := cx dx^2 + 2 cy dx dy + cz dy^2 <= 2 (lop + log2 255) =: 2 tau, everything in the record's
    // log2 units (cx = -2 A' ..., the geometry is homogeneous in the scale).
    // hardware rcp / sqrt (about 1 ulp) instead of the correctly rounded sequences (10 - 15 instructions each): the test
    // carries 0.4 % + 0.03 of slack on q' and 0.01 px on the spans, orders of magnitude above their error
    const float mx = q0.x - X0, my = q0.y - Y0;                  // centre in tile-local pixel coordinates
    const float cx = -2.f * q0.z, cy = -q0.w, cz = -2.f * q1.x;
    const float det = cx * cz - cy * cy;
    const float tau = q1.y + 7.994353436858858f;                 // log2(255 opacity)
    if (!(tau > 0.f)) return 0u;                                 // opacity < 1/255: alpha never reaches the threshold
    // per Gaussian, from the projection kernel (gp_sb_side): ex = sqrt(tt cz / det) = half extent in x, the rightmost point at
    // dy = dyr = -(cy / cz) ex, inv_cx = 1 / (4 cx); NaN = degenerate conic (det, cx or cz not positive) or ex > 1e8: do not cull
#if GP_SB_HOIST == 1
    const float dyr = side.x, inv_cx = side.y;
    if (inv_cx != inv_cx) return 0xFFFFu;
    const float tt = 2.f * tau * 1.004f + 0.03f;                 // q' <= tt, with slack for the rounding of the exponent / exp2
#elif GP_SB_HOIST == 2   // dyr alone, in the record's spare word (q1.w): no second gather; NaN = do not cull
    const float dyr = q1.w;
    if (dyr != dyr) return 0xFFFFu;
    const float tt = 2.f * tau * 1.004f + 0.03f;
    const float inv_cx = 0.25f * __builtin_amdgcn_rcpf(cx);
#else       // (-DGP_SB_HOIST=0: the same values computed per instance, as through round 5 -- the A/B of profiles/r06_sb_hoist_ab.txt)
    if (!(det > 0.f && cx > 0.f && cz > 0.f)) return 0xFFFFu;
    const float tt = 2.f * tau * 1.004f + 0.03f;
    const float ex = __builtin_amdgcn_sqrtf(tt * cz * __builtin_amdgcn_rcpf(det));
    if (!(ex <= 1e8f)) return 0xFFFFu;
    const float inv_cx = 0.25f * __builtin_amdgcn_rcpf(cx);
    const float dyr = -(cy * __builtin_amdgcn_rcpf(cz)) * ex;
#endif
    // everything below in SB columns (quarter pixels): column k covers pixel centres 4k .. 4k + 3
    const float rxy = -cy * inv_cx;                              // centre line of the row spans: c(dy) = rxy dy
    const float ctt = cx * tt;
    const float mxr = fmaf(mx + 0.01f, 0.25f, 1.f);              // (+ 1: floor(x) + 1 = one past the last column)
    const float mxl = (mx - 3.01f) * 0.25f;                      // (xl - 3) / 4
    const float ndet = -det;
    uint32_t mask = 0u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float d0 = (float)(4 * s) - my, d1 = d0 + 3.f;
        const float dr = __builtin_amdgcn_fmed3f(dyr, d0, d1), dl = __builtin_amdgcn_fmed3f(-dyr, d0, d1);
        const float Dr = fmaf(ndet * dr, dr, ctt), Dl = fmaf(ndet * dl, dl, ctt);
        // A strip that misses the ellipse's y range has Dr < 0: its square root is a NaN, which travels through floor and the
        // subtraction into the width and is turned into ZERO by the clamp (v_med3_f32 with a NaN operand returns the smallest of
        // the other two): no compare, no select.  (Dl < 0 with Dr >= 0 -- rounding at the very edge of the y range -- clamps to
        // column 0: a superset.)
        const float kh1 = floorf(fmaf(__builtin_amdgcn_sqrtf(Dr), inv_cx, fmaf(rxy, dr, mxr)));        // last column + 1 (unclamped)
        const float klf = ceilf(fmaf(-__builtin_amdgcn_sqrtf(Dl), inv_cx, fmaf(rxy, dl, mxl)));        // first column (unclamped)
        const float kl = __builtin_amdgcn_fmed3f(klf, 0.f, 4.f);
        const float w = __builtin_amdgcn_fmed3f(kh1 - kl, 0.f, 4.f);                                   // columns kl .. kl + w - 1
        uint32_t bits;
        asm("v_bfm_b32 %0, %1, %2" : "=v"(bits) : "v"((uint32_t)w), "v"((uint32_t)kl));               // ((1 << w) - 1) << kl
        mask |= (bits & 0xFu) << (4 * s);
    }
    return mask;
}

// (MODE 2) pair counters of the composite forward: [0] (pixel, splat) pairs that CONTRIBUTE (alpha >= 1/255, pixel not saturated),
// [1] pairs the sub-block lists make the kernel evaluate.  Read and cleared by gp_debug_counters(); bench.py's roofline.contributing_pairs.
__device__ unsigned long long g_pair_counters[4];

template <int MODE>   // 0: compiler-scheduled visit loop (the readable statement of the algorithm)  1: hand-scheduled (shipped)
                      // 2: as 0, and counts evaluated / contributing pairs into g_pair_counters (diagnostics)
__device__ __forceinline__ void gp_composite_fwd_sb_body(RasterDims d, const int2* __restrict__ ranges,
                                                                         const uint32_t* __restrict__ point_list,
                                                                         const float4* __restrict__ rec,
                                                                         const float* __restrict__ bg,
                                                                         float* __restrict__ out_color,
                                                                         float* __restrict__ out_depth,
                                                                         int32_t* __restrict__ out_tidx,
                                                                         float* __restrict__ final_T,
                                                                         int32_t* __restrict__ n_contrib,
                                                                         const uint32_t* __restrict__ order,
                                                                         int32_t* __restrict__ tile_work,
                                                                         uint16_t* __restrict__ smask) {
    __shared__ __attribute__((aligned(16))) unsigned char s_rec[CF_THREADS * CF2_REC];
    __shared__ unsigned short s_list[16][CF_THREADS];    // per SB: byte offsets of the batch's records that touch it, in depth order
    __shared__ int s_cnt[4][16];                         // [staging wave][SB]
    __shared__ int s_tot[16];
    __shared__ int s_done[4];
    __shared__ int s_last[4];
    const int tile = order ? (int)order[blockIdx.x] : (int)blockIdx.x;
    const int tx = tile % d.gx, ty = tile / d.gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    // wave = quadrant, 16-lane group = 4x4 sub-block of the quadrant
    const int grp = lane >> 4;
    const int sbx = 2 * (wave & 1) + (grp & 1), sby = 2 * (wave >> 1) + (grp >> 1);
    const int mysb = sby * 4 + sbx;
    const int px = tx * GP_TILE + sbx * 4 + (lane & 3);
    const int py = ty * GP_TILE + sby * 4 + ((lane >> 2) & 3);
    const float pxf = (float)px, pyf = (float)py;
    const int2 range = ranges[tile];
    float T = 1.f, best = 0.f;
    int best_pos = -1, last = -1;                        // list positions x CF2_REC (tile-relative), -1 = none
    const bool inside = px < d.W && py < d.H;
    int lim = inside ? 0 : -1;                           // < 0: this pixel is finished (or outside the image)
    if (tid < 4) s_done[tid] = 0;
    {   // every list entry is a valid record offset at all times (lanes past their list's end read them, masked)
        uint4* z = (uint4*)&s_list[0][0];
        z[tid] = make_uint4(0u, 0u, 0u, 0u);
        z[tid + CF_THREADS] = make_uint4(0u, 0u, 0u, 0u);
    }
    const float X0 = (float)(tx * GP_TILE), Y0 = (float)(ty * GP_TILE);
    const unsigned short* lst = s_list[mysb];
    v2f C01 = {0.f, 0.f}, C2D = {0.f, 0.f};              // (C0, C1), (C2, depth): the packed accumulators of the asm path
    unsigned n_eval = 0, n_contr = 0;                    // (MODE 2 only)

    // the batch's Gaussian ids are fetched one batch ahead (in flight during the previous batch's visits): a batch then waits for ONE
    // trip to memory (the record gather), not two dependent ones
    uint32_t id_next = range.x + tid < range.y ? point_list[range.x + tid] : 0u;
    for (int base = range.x; base < range.y; base += CF_THREADS) {
        __syncthreads();
        if (s_done[0] && s_done[1] && s_done[2] && s_done[3]) break;
        const int k = base + tid;
        uint32_t rel = 0u;
        const uint32_t id = id_next;
        if (k < range.y) {
            const float4 q0 = rec[3 * (size_t)id], q1 = rec[3 * (size_t)id + 1], q2 = rec[3 * (size_t)id + 2];
#if GP_SB_HOIST == 1
            const float2 side = d.sb_side[id];
#else
            const float2 side = make_float2(0.f, 0.f);
#endif
            float4* dst = (float4*)(s_rec + tid * CF2_REC);
            dst[0] = q0;
            dst[1] = make_float4(q1.x, q1.y, q2.x, q2.y);
            *(float2*)(dst + 2) = make_float2(q2.z, q1.z);
            rel = gp_sb_mask(q0, q1, side, X0, Y0);
            smask[k] = (uint16_t)rel;           // saved for the backward: which 4x4 sub-blocks of its tile the instance can touch
        }
        unsigned long long bal[16];
        int mycnt = 0;
        {   // ballot of bit sb for all 16 SBs: shifting the mask out through the carry flag costs ONE VALU instruction per ballot
            uint32_t sh = rel << 16;
#define CF2_BAL(SB)                                                                                         \
    asm("v_add_co_u32 %0, %1, %0, %0" : "+v"(sh), "=s"(bal[SB]));                                           \
    asm("v_writelane_b32 %0, %1, " #SB : "+v"(mycnt) : "s"((int)__popcll(bal[SB])));
            CF2_BAL(15) CF2_BAL(14) CF2_BAL(13) CF2_BAL(12) CF2_BAL(11) CF2_BAL(10) CF2_BAL(9) CF2_BAL(8)
            CF2_BAL(7) CF2_BAL(6) CF2_BAL(5) CF2_BAL(4) CF2_BAL(3) CF2_BAL(2) CF2_BAL(1) CF2_BAL(0)
#undef CF2_BAL
        }
        if (lane < 16) s_cnt[wave][lane] = mycnt;
        __syncthreads();
        int offv = 0;                                    // lanes 0 .. 15: exclusive prefix of SB `lane` over the staging waves
        if (lane < 16) {                                 // (three loads in flight, selected by the wave-uniform wave number)
            const int c0 = s_cnt[0][lane], c1 = s_cnt[1][lane], c2 = s_cnt[2][lane];
            offv = (wave_s > 0 ? c0 : 0) + (wave_s > 1 ? c1 : 0) + (wave_s > 2 ? c2 : 0);
            if (wave_s == 3) s_tot[lane] = offv + mycnt;
        }
        {   // list entries.  The lanes that write SB sb's list ARE its ballot (exec <- bal[sb]); the slot is the wave's base for the
            // SB (a readlane of the prefix above: no LDS round trip per SB) + the lane's rank in the ballot.  4 VALU per SB.
            const uint32_t val = (uint32_t)(tid * CF2_REC);
            uint32_t t;
            unsigned long long sv;
#define CF2_PUT(SB)                                                                                         \
    "s_mov_b32 exec_lo, %[l" #SB "]\n\t"                                                                    \
    "s_mov_b32 exec_hi, %[h" #SB "]\n\t"                                                                    \
    "v_mbcnt_lo_u32_b32 %[t], %[l" #SB "], 0\n\t"                                                           \
    "v_mbcnt_hi_u32_b32 %[t], %[h" #SB "], %[t]\n\t"                                                        \
    "v_lshl_add_u32 %[t], %[t], 1, %[b" #SB "]\n\t"                                                         \
    "ds_write_b16 %[t], %[val]\n\t"
#define CF2_ADDR(SB) ((uint32_t)(uintptr_t)&s_list[SB][0] + 2u * (uint32_t)__builtin_amdgcn_readlane(offv, SB))
#define CF2_PUT4(A, B, C_, D)                                                                                               \
    asm volatile("s_mov_b64 %[sv], exec\n\t" CF2_PUT(A) CF2_PUT(B) CF2_PUT(C_) CF2_PUT(D) "s_mov_b64 exec, %[sv]\n\t"      \
                 : [t] "=&v"(t), [sv] "=&s"(sv)                                                                             \
                 : [val] "v"(val), [l##A] "s"((uint32_t)bal[A]), [h##A] "s"((uint32_t)(bal[A] >> 32)), [b##A] "s"(CF2_ADDR(A)), \
                   [l##B] "s"((uint32_t)bal[B]), [h##B] "s"((uint32_t)(bal[B] >> 32)), [b##B] "s"(CF2_ADDR(B)),               \
                   [l##C_] "s"((uint32_t)bal[C_]), [h##C_] "s"((uint32_t)(bal[C_] >> 32)), [b##C_] "s"(CF2_ADDR(C_)),          \
                   [l##D] "s"((uint32_t)bal[D]), [h##D] "s"((uint32_t)(bal[D] >> 32)), [b##D] "s"(CF2_ADDR(D))                \
                 : "memory");
            CF2_PUT4(0, 1, 2, 3) CF2_PUT4(4, 5, 6, 7) CF2_PUT4(8, 9, 10, 11) CF2_PUT4(12, 13, 14, 15)
#undef CF2_PUT4
#undef CF2_ADDR
#undef CF2_PUT
        }
        __syncthreads();
        if (k + CF_THREADS < range.y) id_next = point_list[k + CF_THREADS];     // (issued behind the last barrier of the batch: nothing waits for it before the visits are done)
        if (__any(lim >= 0)) {
            const int tot = s_tot[mysb];
            lim = lim < 0 ? -1 : tot;
            const int nmax = max(max(__builtin_amdgcn_readlane(tot, 0), __builtin_amdgcn_readlane(tot, 16)),
                                 max(__builtin_amdgcn_readlane(tot, 32), __builtin_amdgcn_readlane(tot, 48)));
            const int sbase = (base - range.x) * CF2_REC;
            if (MODE == 1) {
                // Four visits per block, hand-scheduled: the compiler's version of this loop carries 37 VALU slots per visit
                // (flag bytes, moves, duplicated compares); this one carries 20 (19 on the fast path).  Temporaries and the two
                // record buffers are fixed registers v24 .. v53 (clobbered); LDS returns in order, so `s_waitcnt lgkmcnt(3)` =
                // "the older record buffer is complete" while the younger one is still in flight.  exec is restored before leaving.
                // FAST blocks (all four visits lie inside every group's list: i0 + 3 < the shortest of the wave's four lists) run
                // under the mask of the pixels still alive, kept in an SGPR pair and shrunk when a pixel saturates, instead of
                // comparing the list position with `lim` in front of every visit.
                uint32_t lp = (uint32_t)(uintptr_t)lst;
                const int nmin = min(min(__builtin_amdgcn_readlane(tot, 0), __builtin_amdgcn_readlane(tot, 16)),
                                     min(__builtin_amdgcn_readlane(tot, 32), __builtin_amdgcn_readlane(tot, 48)));
                unsigned long long alive = __builtin_amdgcn_ballot_w64(lim >= 0);
#define CF2_HEAD_SLOW(K)                                                                                            \
    "s_add_i32 %[sk], %[i0], " #K "\n\t"                                                                            \
    "v_cmp_lt_i32 vcc, %[sk], %[lim]\n\t"                                                                           \
    "s_and_b64 exec, %[sv], vcc\n\t"                                                                                \
    "s_cbranch_execz 1" #K "f\n\t"
#define CF2_VISIT(K, HEAD, TERM, RESTORE, X, Y, A_, B_, C_, OP, RG, BD, OFF)                                        \
    HEAD                                                                                                            \
    "v_sub_f32 v48, " X ", %[pxf]\n\t"                                                                              \
    "v_sub_f32 v49, " Y ", %[pyf]\n\t"                                                                              \
    "v_mul_f32 v50, " B_ ", v49\n\t"                                                                                \
    "v_fmac_f32 v50, " A_ ", v48\n\t"                                                                               \
    "v_mul_f32 v51, " C_ ", v49\n\t"                                                                                \
    "v_fma_f32 v51, v51, v49, " OP "\n\t"            /* (C' dy) dy + lop */                                          \
    "v_fmac_f32 v51, v48, v50\n\t"                                                                                  \
    "v_cmp_ngt_f32 vcc, v51, " OP "\n\t"             /* not (e > lop)  <=>  not (power > 0) */                       \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                 \
    "v_exp_f32 v51, v51\n\t"                                                                                        \
    "s_nop 0\n\t"                                                                                                   \
    "v_min_f32 v51, 0x3f7d70a4, v51\n\t"                                                                            \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v51\n\t"                                                                        \
    "s_and_b64 exec, exec, vcc\n\t"                                                                                 \
    "s_cbranch_execz 1" #K "f\n\t"                                                                                  \
    "v_mul_f32 v52, v51, %[T]\n\t"                  /* w = alpha T */                                               \
    "v_sub_f32 v50, %[T], v52\n\t"                  /* T - w: the transmittance behind this splat */                \
    "v_cmp_ngt_f32 vcc, 0x38d1b717, v50\n\t"                                                                        \
    "s_andn2_b64 %[sd], exec, vcc\n\t"              /* lanes whose transmittance would drop below 1e-4: finished */ \
    "s_cbranch_scc0 2" #K "f\n\t"                                                                                   \
    "s_mov_b64 exec, %[sd]\n\t"                                                                                     \
    "v_mov_b32 %[lim], -1\n\t"                                                                                      \
    TERM                                                                                                            \
    "2" #K ":\n\t"                                                                                                  \
    "s_mov_b64 exec, vcc\n\t"                       /* (v_cmp leaves vcc a subset of the lanes it ran on) */        \
    "v_add_u32 %[last], %[sbase], " OFF "\n\t"                                                                      \
    "v_cmp_gt_f32 vcc, v52, %[best]\n\t"                                                                            \
    "v_pk_fma_f32 %[C01], " RG ", v[52:53], %[C01] op_sel_hi:[1,0,1]\n\t"                                           \
    "v_pk_fma_f32 %[C2D], " BD ", v[52:53], %[C2D] op_sel_hi:[1,0,1]\n\t"                                           \
    "v_mov_b32 %[T], v50\n\t"                                                                                       \
    "s_cbranch_vccz 1" #K "f\n\t"                   /* nobody's blend weight is a new maximum */                    \
    "v_cndmask_b32 %[best], %[best], v52, vcc\n\t"                                                                  \
    "v_cndmask_b32 %[bpos], %[bpos], %[last], vcc\n\t"                                                              \
    "1" #K ":\n\t"                                                                                                  \
    "s_mov_b64 exec, " RESTORE "\n\t"
#define CF2_BLOCK(ENTER, VA, VB)                                                                                    \
                        "s_mov_b64 %[sv], exec\n\t"                                                                 \
                        ENTER                                                                                       \
                        "ds_read_u16 v44, %[lp]\n\t"                                                                \
                        "ds_read_u16 v45, %[lp] offset:2\n\t"                                                       \
                        "ds_read_u16 v46, %[lp] offset:4\n\t"                                                       \
                        "ds_read_u16 v47, %[lp] offset:6\n\t"                                                       \
                        "s_waitcnt lgkmcnt(2)\n\t"                                                                  \
                        "ds_read_b128 v[24:27], v44\n\t"                                                            \
                        "ds_read_b128 v[28:31], v44 offset:16\n\t"                                                  \
                        "ds_read_b64 v[32:33], v44 offset:32\n\t"                                                   \
                        "ds_read_b128 v[34:37], v45\n\t"                                                            \
                        "ds_read_b128 v[38:41], v45 offset:16\n\t"                                                  \
                        "ds_read_b64 v[42:43], v45 offset:32\n\t"                                                   \
                        "s_waitcnt lgkmcnt(3)\n\t"                                                                  \
                        VA(0, "v44")                                                                                \
                        "ds_read_b128 v[24:27], v46\n\t"                                                            \
                        "ds_read_b128 v[28:31], v46 offset:16\n\t"                                                  \
                        "ds_read_b64 v[32:33], v46 offset:32\n\t"                                                   \
                        "s_waitcnt lgkmcnt(3)\n\t"                                                                  \
                        VB(1, "v45")                                                                                \
                        "ds_read_b128 v[34:37], v47\n\t"                                                            \
                        "ds_read_b128 v[38:41], v47 offset:16\n\t"                                                  \
                        "ds_read_b64 v[42:43], v47 offset:32\n\t"                                                   \
                        "s_waitcnt lgkmcnt(3)\n\t"                                                                  \
                        VA(2, "v46")                                                                                \
                        "s_waitcnt lgkmcnt(0)\n\t"                                                                  \
                        VB(3, "v47")                                                                                \
                        "s_add_i32 %[sk], %[i0], 4\n\t"                                                             \
                        "v_cmp_lt_i32 vcc, %[sk], %[lim]\n\t"                                                       \
                        "s_mov_b64 %[am], vcc\n\t"                                                                  \
                        "s_mov_b64 exec, %[sv]\n\t"
#define CF2_SLOW_A(K, OFF) CF2_VISIT(K, CF2_HEAD_SLOW(K), "", "%[sv]", "v24", "v25", "v26", "v27", "v28", "v29", "v[30:31]", "v[32:33]", OFF)
#define CF2_SLOW_B(K, OFF) CF2_VISIT(K, CF2_HEAD_SLOW(K), "", "%[sv]", "v34", "v35", "v36", "v37", "v38", "v39", "v[40:41]", "v[42:43]", OFF)
#define CF2_FAST_TERM "s_andn2_b64 %[al], %[al], %[sd]\n\t"
#define CF2_FAST_A(K, OFF) CF2_VISIT(K, "", CF2_FAST_TERM, "%[al]", "v24", "v25", "v26", "v27", "v28", "v29", "v[30:31]", "v[32:33]", OFF)
#define CF2_FAST_B(K, OFF) CF2_VISIT(K, "", CF2_FAST_TERM, "%[al]", "v34", "v35", "v36", "v37", "v38", "v39", "v[40:41]", "v[42:43]", OFF)
#define CF2_CLOBBERS "vcc", "scc", "memory", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",      \
                          "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", \
                          "v52", "v53"
                const int nmin_s = __builtin_amdgcn_readfirstlane(nmin), nmax_s = __builtin_amdgcn_readfirstlane(nmax);
                const int sbase_s = __builtin_amdgcn_readfirstlane(sbase);
                int i0 = 0;                          // (kept in an SGPR by the statement below, which also advances it and lp)
#pragma unroll 1
                for (;;) {
                    unsigned long long sv, am, sd;
                    int sk;
                    asm volatile(       // (one statement for both block forms: as two, the compiler shuffled 14 registers between them per block)
                        "s_add_i32 %[sk], %[i0], 3\n\t"
                        "s_cmp_lt_i32 %[sk], %[nmin]\n\t"
                        "s_cbranch_scc0 90f\n\t"
                        CF2_BLOCK("s_mov_b64 exec, %[al]\n\t", CF2_FAST_A, CF2_FAST_B)
                        "s_branch 91f\n\t"
                        "90:\n\t"
                        CF2_BLOCK("", CF2_SLOW_A, CF2_SLOW_B)
                        "91:\n\t"
                        "s_add_i32 %[i0], %[i0], 4\n\t"
                        "v_add_u32 %[lp], 8, %[lp]\n\t"
                        : [T] "+v"(T), [C01] "+v"(C01), [C2D] "+v"(C2D), [best] "+v"(best), [bpos] "+v"(best_pos), [last] "+v"(last),
                          [lim] "+v"(lim), [lp] "+v"(lp), [i0] "+s"(i0), [al] "+s"(alive), [sv] "=&s"(sv), [am] "=&s"(am), [sk] "=&s"(sk), [sd] "=&s"(sd)
                        : [pxf] "v"(pxf), [pyf] "v"(pyf), [sbase] "s"(sbase_s), [nmin] "s"(nmin_s)
                        : CF2_CLOBBERS);
                    if (am == 0ull || i0 >= nmax_s) break;
                }
#undef CF2_HEAD_SLOW
#undef CF2_VISIT
#undef CF2_BLOCK
#undef CF2_SLOW_A
#undef CF2_SLOW_B
#undef CF2_FAST_TERM
#undef CF2_FAST_A
#undef CF2_FAST_B
#undef CF2_CLOBBERS
            } else {
#pragma unroll 1
                for (int i = 0; i < nmax; ++i) {
                    const bool act = i < lim;
                    if (!__any(act)) break;
                    if (act) {
                        if (MODE == 2) ++n_eval;
                        const int off = (int)lst[i];
                        const float4 q0 = *(const float4*)(s_rec + off);
                        const float4 q1 = *(const float4*)(s_rec + off + 16);
                        const float2 q2 = *(const float2*)(s_rec + off + 32);
                        const float dx = q0.x - pxf, dy = q0.y - pyf;
                        const float e = fmaf(dx, fmaf(q0.z, dx, q0.w * dy), fmaf(q1.x * dy, dy, q1.y));     // q1.y = log2(opacity)
                        if (!(e > q1.y)) {
                            const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(e));
                            if (!(alpha < 1.f / 255.f)) {
                                const float w = alpha * T;
                                const float test_T = T - w;          // (the published T (1 - alpha), one operation shorter)
                                if (test_T < 0.0001f) {
                                    lim = -1;
                                } else {
                                    const int posv = off + sbase;
                                    C01.x = fmaf(q1.z, w, C01.x);
                                    C01.y = fmaf(q1.w, w, C01.y);
                                    C2D.x = fmaf(q2.x, w, C2D.x);
                                    C2D.y = fmaf(q2.y, w, C2D.y);
                                    if (w > best) { best = w; best_pos = posv; }
                                    T = test_T;
                                    last = posv;
                                    if (MODE == 2) ++n_contr;
                                }
                            }
                        }
                    }
                }
            }
            if (!__any(lim >= 0) && lane == 0) s_done[wave] = 1;
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) { n_eval += __shfl_xor(n_eval, dd); n_contr += __shfl_xor(n_contr, dd); }
        if (lane == 0) { atomicAdd(&g_pair_counters[0], (unsigned long long)n_contr); atomicAdd(&g_pair_counters[1], (unsigned long long)n_eval); }
    }
    const float C0 = C01.x, C1 = C01.y, C2 = C2D.x, Dp = C2D.y;
    const size_t HW = (size_t)d.H * d.W;
    const int last_n = last < 0 ? 0 : last / CF2_REC + 1;          // 1-based position of the last contributor in the tile list
    if (inside) {
        const size_t pix = (size_t)py * d.W + px;
        out_color[pix] = fmaf(T, bg[0], C0);
        out_color[HW + pix] = fmaf(T, bg[1], C1);
        out_color[2 * HW + pix] = fmaf(T, bg[2], C2);
        out_depth[pix] = Dp;
        out_tidx[pix] = best_pos < 0 ? -1 : (int32_t)point_list[range.x + best_pos / CF2_REC];
        final_T[pix] = T;
        n_contrib[pix] = last_n;
    }
    if (tile_work) {   // largest list position any pixel of the tile consumed: the backward's work estimate
        int mx = inside ? last_n : 0;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) mx = max(mx, __shfl_xor(mx, dd));
        if (lane == 0) s_last[wave] = mx;
        __syncthreads();
        if (tid == 0) tile_work[tile] = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    }
}
#define CF2_ARGS RasterDims d, const int2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, \
    const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ out_depth, int32_t* __restrict__ out_tidx, \
    float* __restrict__ final_T, int32_t* __restrict__ n_contrib, const uint32_t* __restrict__ order, int32_t* __restrict__ tile_work, \
    uint16_t* __restrict__ smask
__global__ __launch_bounds__(CF_THREADS) void gp_composite_fwd_sb_kernel(CF2_ARGS) {
    gp_composite_fwd_sb_body<1>(d, ranges, point_list, rec, bg, out_color, out_depth, out_tidx, final_T, n_contrib, order, tile_work, smask);
}
__global__ __launch_bounds__(CF_THREADS) void gp_composite_fwd_sbc_kernel(CF2_ARGS) {   // compiler-scheduled inner loop (A/B reference)
    gp_composite_fwd_sb_body<0>(d, ranges, point_list, rec, bg, out_color, out_depth, out_tidx, final_T, n_contrib, order, tile_work, smask);
}
__global__ __launch_bounds__(CF_THREADS) void gp_composite_fwd_count_kernel(CF2_ARGS) {   // ... + pair counters (gp_debug_option(0, 3))
    gp_composite_fwd_sb_body<2>(d, ranges, point_list, rec, bg, out_color, out_depth, out_tidx, final_T, n_contrib, order, tile_work, smask);
}
int gp_pair_counters_read(unsigned long long* out4) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_pair_counters), 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[4] = {0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pair_counters), z, sizeof(z)) == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// composite backward, SPLAT-parallel.  One wave per (tile, 8x8 quadrant).  Lanes own the 64 splats of the
// current batch (record in registers) and the wave walks the quadrant's pixels uniformly, two at a time with
// v_pk_*_f32.  For pixel p and the batch's splats j (front to back):
//     T_j(p)      = T_in(p) * prod_{k<j} (1 - alpha_k)                 (wave PRODUCT scan, v_mul_f32_dpp ladder)
//     suffix_j(p) = rem_in(p) - sum_{k<=j} s_k,   s_k = alpha_k T_k (c_k . dL_dpix + z_k dL_ddepth)
//                                                                      (wave SUM scan, v_add_f32_dpp ladder)
//     dL/dalpha_j = T_j (c_j . dLp + z_j dLd) - (suffix_j + T_final bg . dLp) / (1 - alpha_j)
// with rem_in(p) initialised from the forward outputs, (C(p) - T_final bg) . dLp + D(p) dLd, and (T_in, rem_in)
// carried from batch to batch in LDS.  Each lane accumulates its splat's sums in registers:
//   * colour / depth / opacity directly;
//   * the geometric gradients as MOMENTS of h = dL/dG * G (sum h, sum h dx, sum h dx^2 per step; dy is constant
//     along a pixel row, so the dy moments are folded in once per row) -- 5 packed ops per step instead of 14.
// Batches are built from the forward's per-instance quadrant masks (qmask), so only splats whose alpha >= 1/255
// footprint meets THIS quadrant are fetched, and lanes are always full.
// Flush: scattered float atomics are the slowest thing this kernel could do (one L2 transaction per lane); the
// wave transposes its 64 x 10 sums through LDS so that 16 consecutive lanes add to the 16 consecutive floats of ONE
// Gaussian's 64-byte accumulator line -- an atomic instruction then touches 4 cache lines instead of 64.
// ------------------------------------------------------------------------------------------------
// inclusive wave SUM scan of two values: row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast 15 / 31 across
// rows; the two chains are interleaved so each covers the other's DPP wait states.
__device__ __forceinline__ void dpp_scan2_add(float& a, float& b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}
// inclusive wave PRODUCT scan of two values (same DPP ladder; lanes without a source keep their value)
__device__ __forceinline__ void dpp_scan2_mul(float& a, float& b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}



// Per-pixel-pair constants of the backward: per (tile, quadrant) GP_BWD_PAIRS pairs x 4 float4
//   o[0] dLr0 dLr1 dLg0 dLg1   o[1] dLb0 dLb1 tb0 tb1   o[2] nc0 nc1 rem0 rem1   o[3] dLd0 dLd1 - -
// (tb = T_final * bg . dL_dpix, nc = n_contrib, rem = initial remaining suffix)
// The per-pixel-pair constants of the backward, from the forward's outputs and the incoming gradient: o[0..3] as listed above.
// Round 2 computed them in a launch of their own (gp_bwd_pixprep_kernel: 88 MB of traffic + a launch, 0.032 ms at c3); every
// pair is consumed by exactly one wave of the composite backward, whose prologue now computes its 32 pairs itself.
__device__ __forceinline__ void gp_pixpair(const RasterDims& d, int px0, int py, float bg0, float bg1, float bg2,
                                           const float* __restrict__ out_color, const float* __restrict__ out_depth,
                                           const float* __restrict__ final_T, const int32_t* __restrict__ n_contrib,
                                           const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpixdepth, float4* o) {
    const size_t HW = (size_t)d.H * d.W;
    float dl[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float tb[2] = {0.f, 0.f}, rem[2] = {0.f, 0.f};
    int nc[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int px = px0 + u;
        if (px < d.W && py < d.H) {
            const size_t pix = (size_t)py * d.W + px;
            dl[u][0] = dL_dpix[pix]; dl[u][1] = dL_dpix[HW + pix]; dl[u][2] = dL_dpix[2 * HW + pix];
            dl[u][3] = dL_dpixdepth ? dL_dpixdepth[pix] : 0.f;
            tb[u] = final_T[pix] * (bg0 * dl[u][0] + bg1 * dl[u][1] + bg2 * dl[u][2]);
            rem[u] = out_color[pix] * dl[u][0] + out_color[HW + pix] * dl[u][1] + out_color[2 * HW + pix] * dl[u][2] - tb[u];
            if (dL_dpixdepth) rem[u] += out_depth[pix] * dl[u][3];
            nc[u] = n_contrib[pix];
        }
    }
    o[0] = make_float4(dl[0][0], dl[1][0], dl[0][1], dl[1][1]);
    o[1] = make_float4(dl[0][2], dl[1][2], tb[0], tb[1]);
    o[2] = make_float4(__int_as_float(nc[0]), __int_as_float(nc[1]), rem[0], rem[1]);
    o[3] = make_float4(dl[0][3], dl[1][3], 0.f, 0.f);
}

// (diagnostics, gp_debug_option(1, bits): 1 = the flush computes but does not issue its atomics, 2 = no pixel walk: what the
// accumulator traffic and the arithmetic each cost, tools/probe/bwd_ablate.sh)
__device__ int g_bwd_ablate;
int gp_bwd_set_ablate(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_ablate), &v, sizeof(int)) == hipSuccess ? 0 : 1; }

template <bool HAS_DEPTH, int ROWS, int COLS>
__device__ __forceinline__ void gp_composite_bwd_body(RasterDims d, const int2* __restrict__ ranges,
                                                       const uint32_t* __restrict__ point_list, const uint16_t* __restrict__ smask,
                                                       const float4* __restrict__ rec, const float* __restrict__ bg,
                                                       const float* __restrict__ out_color, const float* __restrict__ out_depth,
                                                       const float* __restrict__ final_T, const int32_t* __restrict__ n_contrib,
                                                       const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpixdepth,
                                                       float* __restrict__ g_mean2D,
                                                       float* __restrict__ g_conic, float* __restrict__ g_opacity,
                                                       float* __restrict__ g_color, float* __restrict__ g_depth,
                                                       const uint32_t* __restrict__ order) {
    constexpr int PAIRS = ROWS * COLS / 2;
    static_assert(PAIRS <= 64, "one lane per pixel pair in the prologue");
    // per pixel pair, 64 B: [0] n_contrib0 n_contrib1 - -   [1] dLr0 dLr1 dLg0 dLg1   [2] dLb0 dLb1 tb0 tb1 (tb = T_final * bg . dL_dpix)
    //                       [3] Tin0 Tin1 rem0 rem1 (carried between batches).  One array: a step reaches all four with immediate offsets.
    __shared__ float4 s_pp[PAIRS][4];
    __shared__ float2 s_dd[HAS_DEPTH ? PAIRS : 1];   // dLd0 dLd1
    constexpr int parts_x = GP_TILE / COLS, parts = (GP_TILE / ROWS) * parts_x;
    // XCD-aware mapping: workgroups go round-robin over the 8 XCDs (blockIdx & 7), each with an L2 of its own.  The four parts
    // of one tile read the same pixel rows (a 64-byte line holds 16 pixels: two parts' halves) and add into the same Gaussians'
    // accumulator lines, so they are given to ONE XCD, back to back: work item (j, xcd) -> tile slot 8 (j / parts) + xcd, part
    // j % parts.  (blockIdx / parts, blockIdx % parts spread every tile over four L2s: 510 MB of fabric traffic against 443.)
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int slot = (jw / parts) * 8 + xcd;
    const int part = jw % parts;
    if (slot >= d.gx * d.gy) return;
    const int tile = __builtin_amdgcn_readfirstlane(order ? (int)order[slot] : slot);
    const int tx = tile % d.gx, ty = tile / d.gx;
    const int lane = threadIdx.x;
    const int2 range = ranges[tile];
    int max_nc;
    {
        max_nc = 0;
        if (lane < PAIRS) {
            float4 me[4];
            gp_pixpair(d, tx * GP_TILE + (part % parts_x) * COLS + 2 * (lane % (COLS / 2)), ty * GP_TILE + (part / parts_x) * ROWS + lane / (COLS / 2),
                       bg[0], bg[1], bg[2], out_color, out_depth, final_T, n_contrib, dL_dpix, HAS_DEPTH ? dL_dpixdepth : nullptr, me);
            const float4 t0 = me[0], t1 = me[1], t = me[2];
            s_pp[lane][1] = t0; s_pp[lane][2] = t1;
            s_pp[lane][0] = make_float4(t.x, t.y, 0.f, 0.f);
            if (HAS_DEPTH) { const float4 t3 = me[3]; s_dd[lane] = make_float2(t3.x, t3.y); }
            s_pp[lane][3] = make_float4(1.f, 1.f, t.z, t.w);
            max_nc = max(__float_as_int(t.x), __float_as_int(t.y));
        }
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) max_nc = max(max_nc, __shfl_xor(max_nc, dd));
    __builtin_amdgcn_wave_barrier();
    const float halfW = 0.5f * (float)d.W, halfH = 0.5f * (float)d.H;
    const int ablate = g_bwd_ablate;
    const int count = min(range.y - range.x, max_nc);
    const float px_base = (float)(tx * GP_TILE + (part % parts_x) * COLS), py_base = (float)(ty * GP_TILE + (part / parts_x) * ROWS);
    // ---- compaction.  The forward saved, per tile-splat instance, which quadrants its footprint touches
    // (qmask): a refill reads 256 mask bytes + ids with two coalesced loads, keeps the instances of THIS part
    // (stable, so depth order is kept) and appends (list position, id) to a queue.  Records are gathered only
    // for queued instances, one batch ahead of the pixel walk.
    static_assert(ROWS == 8 && COLS == 8, "parts are the 8x8 quadrants: four sub-block bits of the saved 16-bit mask each");
    const uint32_t quad_bits = 0x33u << (8 * (part >> 1) + 2 * (part & 1));
    constexpr int QCAP = 448;
    __shared__ int2 s_q[QCAP];            // (list position, gaussian id)
    __shared__ float4 s_fl[3][64];        // flush staging
    __shared__ uint32_t s_flid[64];
    int qn = 0, src = 0;
    auto refill = [&]() {   // candidates src + 64 e + lane, e = 0..3: coalesced byte / dword loads
        bool r[4];
        uint32_t ids[4], qm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = src + 64 * e + lane;
            const int kk = range.x + (k < count ? k : count - 1);       // clamped (count > 0 here)
            qm[e] = smask[kk];
            ids[e] = point_list[kk];
        }
        // pinned: the eight loads leave together and are waited for once (as conditional loads each (mask, id) pair was waited for
        // before the next was issued: four dependent trips to L2 per refill)
        asm volatile("" ::"v"(qm[0]), "v"(qm[1]), "v"(qm[2]), "v"(qm[3]), "v"(ids[0]), "v"(ids[1]), "v"(ids[2]), "v"(ids[3]));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = src + 64 * e + lane;
            r[e] = k < count && (qm[e] & quad_bits) != 0u;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long bal = __ballot(r[e]);
            if (r[e]) s_q[qn + (int)gp_mbcnt(bal)] = make_int2(src + 64 * e + lane, (int)ids[e]);
            qn += (int)__popcll(bal);
        }
        src += 256;
    };
    while (qn < 128 && src < count) refill();
    __builtin_amdgcn_wave_barrier();
    // records of the first batch
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    int2 ne = make_int2(0x7fffffff, 0);
    if (lane < qn) {
        ne = s_q[lane];
        n0 = rec[3 * (size_t)ne.y]; n1 = rec[3 * (size_t)ne.y + 1]; n2 = rec[3 * (size_t)ne.y + 2];
    }
    while (qn > 0) {
        const int nb = min(64, qn);
        const bool have = lane < nb;
        const int b0 = __builtin_amdgcn_readfirstlane(ne.x);        // smallest list position in this batch (uniform)
        const int pos = have ? ne.x : 0x7fffffff;
        const uint32_t id = (uint32_t)ne.y;
        const float4 q0 = n0, q1 = n1, q2 = n2;
        {   // pop the batch, top the queue up, and start fetching the next batch's records
            int2 mv[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) mv[e] = (64 + e * 64 + lane < qn) ? s_q[64 + e * 64 + lane] : make_int2(0, 0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 6; ++e) if (64 + e * 64 + lane < qn) s_q[e * 64 + lane] = mv[e];
            qn -= nb;
            __builtin_amdgcn_wave_barrier();
            while (qn < 128 && src < count) refill();
            __builtin_amdgcn_wave_barrier();
            n0 = make_float4(0.f, 0.f, 0.f, 0.f); n1 = n0; n2 = n0;
            ne = make_int2(0x7fffffff, 0);
            if (lane < qn) {
                ne = s_q[lane];
                n0 = rec[3 * (size_t)ne.y]; n1 = rec[3 * (size_t)ne.y + 1]; n2 = rec[3 * (size_t)ne.y + 2];
            }
        }
        // alpha is recomputed with the FORWARD's expression tree, bit for bit (dx = x - px, dy = y - py from the same operands, the same
        // fma nesting, exp2 of the same exponent): the backward's discrete decisions (e > lop, alpha < 1/255) are then the forward's own.
        // The record holds A' B' C' = log2(e) x the quadratic form and lop = log2(opacity) (preprocess_fwd_body); E := exp2(e) =
        // opacity G is the unclamped alpha, so  h = dL/dG G = E dL/dalpha  needs no opacity factor and  dL/dopacity = sum(E dL/dalpha) /
        // opacity  is divided once per (batch, splat) at the flush.
        const float As = q0.z, Bs = q0.w, Cs = q1.x;
        const float lop = have ? q1.y : -INFINITY, zdep = q1.z;         // (an idle lane: exponent -inf, E = 0)
        const float opac = q2.w;
        const float LN2 = 0.6931471805599453f;                         // conic = ln 2 x the record's primed form
        const float cxx = -2.f * LN2 * q0.z, cxy = -LN2 * q0.w, cyy = -2.f * LN2 * q1.x;
        const v2f cr = {q2.x, q2.x}, cg = {q2.y, q2.y}, cb = {q2.z, q2.z};
        // h := dL/dG * G per (splat, pixel).  The geometric gradients are moments of h:
        //   S_x = sum h dx, S_y = sum h dy, S_xx = sum h dx^2, S_xy = sum h dx dy, S_yy = sum h dy^2
        // dy is constant along a row, so only sum h, sum h dx and S_xx are accumulated per step; the row totals
        // are folded in at the end of each row.
        v2f a_op = {0.f, 0.f}, a_r = {0.f, 0.f}, a_g = {0.f, 0.f}, a_b = {0.f, 0.f}, a_d = {0.f, 0.f}, s_xx = {0.f, 0.f};
        float S_x = 0.f, S_y = 0.f, S_xy = 0.f, S_yy = 0.f;
        float any_m = 0.f;
        auto fma2 = [](v2f a, v2f b, v2f c) { return (v2f){fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; };   // -> v_pk_fma_f32
        v2f dxs[COLS / 2];                     // dx of the two pixels of each pair: the same in every row
#pragma unroll
        for (int cp = 0; cp < COLS / 2; ++cp) dxs[cp] = (v2f){q0.x - (px_base + (float)(2 * cp)), q0.x - (px_base + (float)(2 * cp + 1))};
#pragma unroll 1
        for (int row = 0; row < ((ablate & 2) ? 0 : ROWS); ++row) {
            const float dy = q0.y - (py_base + (float)row);
            const float tB = Bs * dy, uC = fmaf(Cs * dy, dy, lop);
            v2f r_h = {0.f, 0.f}, r_hx = {0.f, 0.f};
            if (!HAS_DEPTH) {
                // The four pixel-pair steps of a row, hand-scheduled (the compiler's version of the step below carries ~95 issue
                // slots: phi copies at the merge points, address moves, duplicated selects; this one 80).  Temporaries v64..v97 are
                // fixed registers (clobbered); exec is restored before leaving.  Same arithmetic as the C++ step (HAS_DEPTH path).
                const v2f As2 = {As, As}, tB2 = {tB, tB}, uC2 = {uC, uC};
                const v2f dx0 = dxs[0], dx1 = dxs[1], dx2 = dxs[2], dx3 = dxs[3];
                const uint32_t vpp = (uint32_t)(uintptr_t)&s_pp[row * (COLS / 2)][0];
                unsigned long long sv, c0, c1, t0;
// (-DGP_CB_LADDER_NOPS=0: the two scans WITHOUT their wait states -- wrong sums, timing only: the most that any re-pairing of the ladders
// (two pixel pairs through one ladder, so that independent DPP operations take the place of the nops) could ever gain;
// tools/probe/bwd_ladder_nops.sh, profiles/r06_bwd_ladder_nops.txt)
#ifndef GP_CB_LADDER_NOPS
#define GP_CB_LADDER_NOPS 1
#endif
#if GP_CB_LADDER_NOPS
#define CB_NOP0 "s_nop 0\n\t"
#define CB_NOP1 "s_nop 1\n\t"
#else
#define CB_NOP0 ""
#define CB_NOP1 ""
#endif
#define CB_SCAN2(OPC, R0, R1)                                                                   \
    CB_NOP1                                                                                     \
    OPC " " R0 ", " R0 ", " R0 " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                       \
    OPC " " R1 ", " R1 ", " R1 " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                       \
    CB_NOP0                                                                                     \
    OPC " " R0 ", " R0 ", " R0 " row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                       \
    OPC " " R1 ", " R1 ", " R1 " row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                       \
    CB_NOP0                                                                                     \
    OPC " " R0 ", " R0 ", " R0 " row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                       \
    OPC " " R1 ", " R1 ", " R1 " row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                       \
    CB_NOP0                                                                                     \
    OPC " " R0 ", " R0 ", " R0 " row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                       \
    OPC " " R1 ", " R1 ", " R1 " row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                       \
    CB_NOP0                                                                                     \
    OPC " " R0 ", " R0 ", " R0 " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
    OPC " " R1 ", " R1 ", " R1 " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
    CB_NOP0                                                                                     \
    OPC " " R0 ", " R0 ", " R0 " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                    \
    OPC " " R1 ", " R1 ", " R1 " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                    \
    CB_NOP1
#define CB_STEP(CP, O0, O1, O2, O3, DX)                                                                               \
    "ds_read_b64 v[64:65], %[vpp] offset:" #O0 "\n\t"                                                                \
    "ds_read_b128 v[66:69], %[vpp] offset:" #O1 "\n\t"                                                               \
    "ds_read_b128 v[70:73], %[vpp] offset:" #O2 "\n\t"                                                               \
    "ds_read_b128 v[74:77], %[vpp] offset:" #O3 "\n\t"                                                               \
    "v_pk_fma_f32 v[78:79], %[As2], " DX ", %[tB2] op_sel_hi:[0,1,0]\n\t"                                             \
    "s_nop 0\n\t"                                                                                                   \
    "v_pk_fma_f32 v[78:79], " DX ", v[78:79], %[uC2] op_sel_hi:[1,1,0]\n\t"                                           \
    "s_waitcnt lgkmcnt(3)\n\t"                                                                                       \
    "v_max_i32 v80, v64, v65\n\t"                                                                                    \
    "v_cmp_lt_i32 vcc, %[b0], v80\n\t"                                                                               \
    "s_cbranch_vccz 9" #CP "f\n\t"                                                                                   \
    "v_exp_f32 v80, v78\n\t"                         /* E = exp2(e): the unclamped alpha (opacity x G) */           \
    "v_exp_f32 v81, v79\n\t"                                                                                         \
    "v_cmp_lt_i32 %[c0], %[pos], v64\n\t"                                                                            \
    "v_cmp_lt_i32 %[c1], %[pos], v65\n\t"                                                                            \
    "v_cmp_ngt_f32 %[t0], v78, %[lop]\n\t"           /* not (e > lop)  <=>  not (power > 0), as the forward */      \
    "s_and_b64 %[c0], %[c0], %[t0]\n\t"                                                                              \
    "v_cmp_ngt_f32 %[t0], v79, %[lop]\n\t"                                                                           \
    "s_and_b64 %[c1], %[c1], %[t0]\n\t"                                                                              \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v80\n\t"         /* alpha < 1/255 <=> E < 1/255 (a literal needs VOPC: vcc) */   \
    "s_and_b64 %[c0], %[c0], vcc\n\t"                                                                                \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v81\n\t"                                                                         \
    "s_and_b64 %[c1], %[c1], vcc\n\t"                                                                              \
    "s_or_b64 %[t0], %[c0], %[c1]\n\t"                                                                               \
    "s_cbranch_scc0 9" #CP "f\n\t"                                                                                   \
    "v_cndmask_b32 v80, 0, v80, %[c0]\n\t"           /* a lane that does not contribute takes part with E = alpha = 0 */ \
    "v_cndmask_b32 v81, 0, v81, %[c1]\n\t"                                                                           \
    "v_min_f32 v82, 0x3f7d70a4, v80\n\t"                                                                             \
    "v_min_f32 v83, 0x3f7d70a4, v81\n\t"                                                                             \
    "v_max3_f32 %[any], %[any], v82, v83\n\t"                                                                        \
    "v_pk_add_f32 v[84:85], v[82:83], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                       \
    "v_pk_mul_f32 v[86:87], %[cb], v[70:71]\n\t"                                                                     \
    "v_rcp_f32 v88, v84\n\t"                                                                                         \
    "v_rcp_f32 v89, v85\n\t"                                                                                         \
    "v_pk_fma_f32 v[86:87], %[cg], v[68:69], v[86:87]\n\t"                                                           \
    CB_SCAN2("v_mul_f32_dpp", "v84", "v85")                                                                           \
    "v_pk_fma_f32 v[86:87], %[cr], v[66:67], v[86:87]\n\t"                                                           \
    "v_pk_mul_f32 v[90:91], v[74:75], v[84:85]\n\t"                                                                  \
    "s_nop 0\n\t"                                                                                                   \
    "v_pk_mul_f32 v[84:85], v[90:91], v[88:89]\n\t"                                                                  \
    "s_nop 0\n\t"                                                                                                   \
    "v_pk_mul_f32 v[94:95], v[82:83], v[84:85]\n\t"                                                                  \
    "s_nop 0\n\t"                                                                                                   \
    "v_pk_mul_f32 v[96:97], v[94:95], v[86:87]\n\t"                                                                  \
    CB_SCAN2("v_add_f32_dpp", "v96", "v97")                                                                           \
    "v_pk_add_f32 v[92:93], v[76:77], v[96:97] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                         \
    "s_bfm_b64 exec, 1, 63\n\t"                                                                                      \
    "ds_write_b128 %[vpp], v[90:93] offset:" #O3 "\n\t"                                                              \
    "s_mov_b64 exec, %[sv]\n\t"                                                                                      \
    "v_pk_add_f32 v[96:97], v[92:93], v[72:73]\n\t"                                                                  \
    "v_pk_fma_f32 %[a_r], v[94:95], v[66:67], %[a_r]\n\t"                                                            \
    "v_pk_mul_f32 v[96:97], v[96:97], v[88:89]\n\t"                                                                  \
    "v_pk_fma_f32 %[a_g], v[94:95], v[68:69], %[a_g]\n\t"                                                            \
    "v_pk_fma_f32 v[96:97], v[84:85], v[86:87], v[96:97] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"                           \
    "v_pk_fma_f32 %[a_b], v[94:95], v[70:71], %[a_b]\n\t"                                                            \
    "v_pk_mul_f32 v[96:97], v[80:81], v[96:97]\n\t"                                                                  \
    "s_nop 0\n\t"                                                                                                   \
    "v_pk_add_f32 %[a_op], %[a_op], v[96:97]\n\t"      /* sum E dL/dalpha = opacity dL/dopacity = sum h */           \
    "v_pk_add_f32 %[r_h], %[r_h], v[96:97]\n\t"                                                                      \
    "v_pk_mul_f32 v[96:97], v[96:97], " DX "\n\t"                                                                    \
    "s_nop 0\n\t"                                                                                                   \
    "v_pk_add_f32 %[r_hx], %[r_hx], v[96:97]\n\t"                                                                    \
    "v_pk_fma_f32 %[s_xx], v[96:97], " DX ", %[s_xx]\n\t"                                                            \
    "9" #CP ":\n\t"
                asm volatile(
                    "s_mov_b64 %[sv], exec\n\t"
                    CB_STEP(0, 0, 16, 32, 48, "%[dx0]")
                    CB_STEP(1, 64, 80, 96, 112, "%[dx1]")
                    CB_STEP(2, 128, 144, 160, 176, "%[dx2]")
                    CB_STEP(3, 192, 208, 224, 240, "%[dx3]")
                    : [a_op] "+v"(a_op), [a_r] "+v"(a_r), [a_g] "+v"(a_g), [a_b] "+v"(a_b), [s_xx] "+v"(s_xx), [r_h] "+v"(r_h), [r_hx] "+v"(r_hx),
                      [any] "+v"(any_m), [sv] "=&s"(sv), [c0] "=&s"(c0), [c1] "=&s"(c1), [t0] "=&s"(t0)
                    : [As2] "v"(As2), [tB2] "v"(tB2), [uC2] "v"(uC2), [lop] "v"(lop), [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cb), [dx0] "v"(dx0),
                      [dx1] "v"(dx1), [dx2] "v"(dx2), [dx3] "v"(dx3), [pos] "v"(pos), [vpp] "v"(vpp), [b0] "s"(b0)
                    : "vcc", "scc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
                      "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94",
                      "v95", "v96", "v97");
#undef CB_STEP
#undef CB_SCAN2
            } else
#pragma unroll
            for (int cp = 0; cp < COLS / 2; ++cp) {
                const int pr = row * (COLS / 2) + cp;
                // all four broadcast reads are issued up front; the alpha math below covers their latency
                const float4 ncf = s_pp[pr][0];
                const int2 nc = make_int2(__float_as_int(ncf.x), __float_as_int(ncf.y));
                const float4 v0 = s_pp[pr][1], v1 = s_pp[pr][2], cy = s_pp[pr][3];
                float2 v3 = make_float2(0.f, 0.f);
                if (HAS_DEPTH) v3 = s_dd[pr];
                const v2f dx = dxs[cp];
                const v2f pw = {fmaf(dx.x, fmaf(As, dx.x, tB), uC), fmaf(dx.y, fmaf(As, dx.y, tB), uC)};
                const v2f E = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};          // pw = the exponent e (uC carries lop)
                const int ncx = nc.x, ncy = nc.y;
                if (max(ncx, ncy) > b0) {   // uniform: otherwise both pixels finished before this batch
                    const bool c0 = (pos < ncx) && !(pw.x > lop) && !(E.x < 1.f / 255.f);
                    const bool c1 = (pos < ncy) && !(pw.y > lop) && !(E.y < 1.f / 255.f);
                    if (__builtin_amdgcn_ballot_w64(c0 || c1) != 0ull) {   // otherwise nobody in the wave touches either pixel
                        // a lane that does not contribute to a pixel takes part with G = 0 (alpha = 0, factor 1 in the product scan,
                        // 0 in the sum scan, zero gradient): one select per pixel instead of mask multiplications
                        const v2f Em = {c0 ? E.x : 0.f, c1 ? E.y : 0.f};
                        const v2f am = {fminf(0.99f, Em.x), fminf(0.99f, Em.y)};
                        any_m = fmaxf(any_m, fmaxf(am.x, am.y));
                        const v2f om = 1.f - am;
                        v2f cdot = cb * (v2f){v1.x, v1.y};
                        cdot = fma2(cg, (v2f){v0.z, v0.w}, cdot);
                        cdot = fma2(cr, (v2f){v0.x, v0.y}, cdot);
                        v2f dLd = {0.f, 0.f};
                        if (HAS_DEPTH) { dLd.x = v3.x; dLd.y = v3.y; cdot = fma2((v2f){zdep, zdep}, dLd, cdot); }
                        // T_j = T_in * prod_{k<j} (1 - alpha_k): inclusive product scan, then divide the own factor out
                        const v2f rom = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                        float il0 = om.x, il1 = om.y;
                        dpp_scan2_mul(il0, il1);
                        const v2f Tnext = (v2f){cy.x, cy.y} * (v2f){il0, il1};    // transmittance AFTER splat j
                        const v2f Tj = Tnext * rom;
                        const v2f w = am * Tj;
                        const v2f sv = w * cdot;
                        float is0 = sv.x, is1 = sv.y;
                        dpp_scan2_add(is0, is1);
                        const v2f rem = {cy.z, cy.w};
                        const v2f tbv = {v1.z, v1.w};
                        const v2f suffix = rem - (v2f){is0, is1};
                        const v2f dL_dalpha = fma2(Tj, cdot, -((suffix + tbv) * rom));
                        a_r = fma2(w, (v2f){v0.x, v0.y}, a_r);
                        a_g = fma2(w, (v2f){v0.z, v0.w}, a_g);
                        a_b = fma2(w, (v2f){v1.x, v1.y}, a_b);
                        if (HAS_DEPTH) a_d = fma2(w, dLd, a_d);
                        const v2f h = Em * dL_dalpha;
                        a_op += h;
                        const v2f hx = h * dx;
                        r_h += h;
                        r_hx += hx;
                        s_xx = fma2(hx, dx, s_xx);
                        // carry to the next batch
                        if (lane == 63) s_pp[pr][3] = make_float4(Tnext.x, Tnext.y, cy.z - is0, cy.w - is1);
                    }
                }
            }
            {   // fold the row:  sum h dy = dy sum h, sum h dx dy = dy sum h dx, sum h dy^2 = dy^2 sum h
                const float rh = r_h.x + r_h.y, rhx = r_hx.x + r_hx.y;
                S_x += rhx;
                S_y = fmaf(dy, rh, S_y);
                S_xy = fmaf(dy, rhx, S_xy);
                S_yy = fmaf(dy * dy, rh, S_yy);
            }
        }
        // flush: transpose through LDS so that 16 consecutive lanes add to the 16 consecutive floats of ONE
        // Gaussian's accumulator line -- an atomic instruction then touches 4 cache lines instead of 64.
        {
            const bool mine = have && any_m > 0.f;
            float* fl = (float*)&s_fl[0][0];        // [64 splats][4]: comps 0..3
            float* fm = (float*)&s_fl[1][0];        //                 comps 4..7
            float* fh = (float*)&s_fl[2][0];        //                 comps 8..11
            // dG/dmean = -G (conic d),  dG/d(conic) = -0.5 G (dx^2, 2 dx dy, dy^2)
            const float S_xx = s_xx.x + s_xx.y;
            const float g_mx = -(cxx * S_x + cxy * S_y), g_my = -(cyy * S_y + cxy * S_x);
            ((float4*)fl)[lane] = make_float4(g_mx * halfW, g_my * halfH, -0.5f * S_xx, -S_xy);
            ((float4*)fm)[lane] = make_float4(-0.5f * S_yy, mine ? (a_op.x + a_op.y) / opac : 0.f, a_r.x + a_r.y, a_g.x + a_g.y);
            ((float4*)fh)[lane] = make_float4(a_b.x + a_b.y, HAS_DEPTH ? a_d.x + a_d.y : 0.f, 0.f, 0.f);
            s_flid[lane] = mine ? id : 0xffffffffu;
            __builtin_amdgcn_wave_barrier();
            const int comp = lane & 15, sub = lane >> 4;
            const float* src_c = comp < 4 ? fl : (comp < 8 ? fm : fh);
            if (comp < (HAS_DEPTH ? 10 : 9)) {
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int sp = r * 4 + sub;
                    const uint32_t gid = s_flid[sp];
                    if (gid != 0xffffffffu && !(ablate & 1)) atomicAdd(&g_mean2D[GP_ACC_STRIDE * (size_t)gid + comp], src_c[sp * 4 + (comp & 3)]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
#define CB_ARGS RasterDims d, const int2* __restrict__ ranges, const uint32_t* __restrict__ point_list, \
    const uint16_t* __restrict__ smask, const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ out_color, \
    const float* __restrict__ out_depth, const float* __restrict__ final_T, const int32_t* __restrict__ n_contrib, \
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpixdepth, float* __restrict__ g_mean2D, \
    float* __restrict__ g_conic, float* __restrict__ g_opacity, float* __restrict__ g_color, float* __restrict__ g_depth, \
    const uint32_t* __restrict__ order
__global__ __launch_bounds__(64) void gp_composite_bwd_kernel(CB_ARGS) {
    gp_composite_bwd_body<false, GP_BWD_ROWS, GP_BWD_COLS>(d, ranges, point_list, smask, rec, bg, out_color, out_depth, final_T, n_contrib, dL_dpix, dL_dpixdepth, g_mean2D, g_conic, g_opacity, g_color, g_depth, order);
}
__global__ __launch_bounds__(64) void gp_composite_bwd_depth_kernel(CB_ARGS) {
    gp_composite_bwd_body<true, GP_BWD_ROWS, GP_BWD_COLS>(d, ranges, point_list, smask, rec, bg, out_color, out_depth, final_T, n_contrib, dL_dpix, dL_dpixdepth, g_mean2D, g_conic, g_opacity, g_color, g_depth, order);
}

// ------------------------------------------------------------------------------------------------
// composite backward, SUB-BLOCK groups (round 5).  AN EXPERIMENT, NOT SHIPPED: gp_debug_option(7, 3) selects it, the quadrant kernel
// above stays the default.  Same wave geometry -- one wave per (tile, 8x8 quadrant), the same queue of the quadrant's instances, the
// same per-pixel state in LDS -- but the walk is culled at 4x4:
//
//   the forward saves the 16-bit sub-block mask of every instance (which 4x4 blocks of its tile its alpha >= 1/255 footprint can
//   reach; the quadrant kernel keeps an instance if it reaches any of the quadrant's four blocks, and then walks ALL 64 pixels with it:
//   3.0 evaluated (pixel, splat) pairs per contributing one at configs[2], because an instance that reaches a quadrant reaches on
//   average two of its four blocks).  Here a batch of 64 queued instances is split into FOUR lists, one per sub-block, and the wave's
//   four DPP rows (16 lanes each) walk one sub-block each: row r takes the next 16 entries of list r into its lanes and steps through
//   the 8 pixel pairs of ITS 4x4 block, with the scans running inside the row (row_shr 1 / 2 / 4 / 8: four DPP stages instead of
//   six).  A batch costs max_r ceil(n_r / 16) rounds of 8 steps instead of 32 steps.  Records are staged once per batch in LDS (the
//   lanes of a round pick theirs by list entry).  The queue entry carries the instance's four sub-block bits above the Gaussian id
//   (N < 2^28: GP_BWD_SB_MAX_N, checked by the host).
//
// What it measured (configs[2], profiles/r05_bwd_subblock_ab.txt; gradients equal to the quadrant kernel's to 7e-7 rel-L2):
//   * evaluated / contributing pairs 3.0 -> 1.51 (counting variant, g_pair_counters[2], [3]): the culling works;
//   * vector instructions 122.6 M -> 107.2 M only: the lists are SHORT (a quadrant's batch of 64 gives ~33 entries per sub-block =
//     3 rounds of 16 where 2.05 would do, and the wave runs the longest of its four lists): 24 steps per batch instead of 32, plus
//     the per-round staging;
//   * the sums of a (splat, sub-block) have to meet the splat's other sub-blocks' before they leave, or leave on their own:
//       - gathered in LDS with ds_add_f32: an LDS float atomic costs ~190 cycles per wave instruction here -- 1 700 LDS cycles per
//         round against 2 400 of arithmetic: 0.48 ms;
//       - flushed per round (the form below): 3 x the global atomic instructions of the quadrant kernel, and the texture-address
//         path retires roughly ONE atomic lane per cycle and CU -- 2.4 M wave-atomics = 0.25 ms of that unit: 0.40 ms with them,
//         0.219 ms with the atomics ablated (quadrant kernel: 0.243 / 0.231).
//   With a non-atomic four-phase read-modify-write in LDS and a hand-scheduled step the estimate is ~0.21 ms against 0.243: not
//   enough for a second hand-written walk in the tree.  Kept selectable so the numbers above can be reproduced.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dpp_rowscan2_add(float& a, float& b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void dpp_rowscan2_mul(float& a, float& b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}

// (MODE 1 of the kernel below) pairs the backward evaluates / that contribute, as the forward's counters: g_pair_counters[2], [3]
template <bool HAS_DEPTH, int MODE>     // MODE 0: shipped  1: + pair counters (diagnostics)
__device__ __forceinline__ void gp_composite_bwd_sb_body(RasterDims d, const int2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ point_list, const uint16_t* __restrict__ smask,
                                                          const float4* __restrict__ rec, const float* __restrict__ bg,
                                                          const float* __restrict__ out_color, const float* __restrict__ out_depth,
                                                          const float* __restrict__ final_T, const int32_t* __restrict__ n_contrib,
                                                          const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpixdepth,
                                                          float* __restrict__ g_mean2D, const uint32_t* __restrict__ order) {
    constexpr int PAIRS = 32;                            // pair index = sub-block (0..3) * 8 + pixel row (0..3) * 2 + column pair (0..1)
    constexpr int QCAP = 192;                            // (<= 63 queued + one refill of 128 candidates)
    // per pixel pair, 64 B, as the quadrant kernel: [0] n_contrib0 n_contrib1 - -  [1] dLr0 dLr1 dLg0 dLg1  [2] dLb0 dLb1 tb0 tb1
    //                                               [3] Tin0 Tin1 rem0 rem1 (carried from round to round)
    // The four rows read FOUR different pairs in one instruction: a row's eight pairs are 33 float4 apart (528 B), which puts the rows'
    // 16-byte reads on four different bank groups (at 512 B -- the natural stride -- every read was a 4-way bank conflict: the first
    // version of this kernel spent 2.3 x the quadrant kernel's time waiting for LDS).
    constexpr int RS = 33;                               // float4 per row of pairs
    __shared__ float4 s_ppf[4 * RS];
    __shared__ float2 s_dd[HAS_DEPTH ? 4 * 9 : 1];       // (row stride 9 float2 = 72 B)
#define SPP(pr, j) s_ppf[((pr) >> 3) * RS + ((pr) & 7) * 4 + (j)]
#define SDD(pr) s_dd[((pr) >> 3) * 9 + ((pr) & 7)]
    __shared__ int2 s_q[QCAP];                           // queue of the quadrant's instances: (list position, id | sub-block bits << 28)
    __shared__ float4 s_rec[64][3];                      // the batch's records: (x y A' B') (C' lop r g) (b depth pos -)
    __shared__ unsigned char s_rl[4][64];                // per sub-block: the batch lanes whose instance reaches it, in depth order
    __shared__ uint32_t s_gid[64];                       // the batch's Gaussian ids
    __shared__ float s_fl[10][64];                       // flush staging (per ROUND: a first version gathered a splat's sums from its
    __shared__ uint32_t s_flid[64];                      // up to four rounds in LDS with ds_add_f32 -- ~190 cycles per wave instruction,
                                                         // 1 700 LDS cycles per round against 2 400 of arithmetic: the kernel ran at 0.48 ms)
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int slot_t = (jw >> 2) * 8 + xcd;
    const int part = jw & 3;
    if (slot_t >= d.gx * d.gy) return;
    const int tile = __builtin_amdgcn_readfirstlane(order ? (int)order[slot_t] : slot_t);
    const int tx = tile % d.gx, ty = tile / d.gx;
    const int lane = threadIdx.x;
    const int row = lane >> 4, li = lane & 15;
    const int2 range = ranges[tile];
    const int qx = tx * GP_TILE + (part & 1) * 8, qy = ty * GP_TILE + (part >> 1) * 8;
    int max_nc = 0;
    if (lane < PAIRS) {
        const int r = lane >> 3, y = (lane >> 1) & 3, cp = lane & 1;
        float4 me[4];
        gp_pixpair(d, qx + (r & 1) * 4 + 2 * cp, qy + (r >> 1) * 4 + y, bg[0], bg[1], bg[2], out_color, out_depth, final_T, n_contrib, dL_dpix,
                   HAS_DEPTH ? dL_dpixdepth : nullptr, me);
        const float4 t0 = me[0], t1 = me[1], t = me[2];
        SPP(lane, 1) = t0; SPP(lane, 2) = t1;
        SPP(lane, 0) = make_float4(t.x, t.y, 0.f, 0.f);
        if (HAS_DEPTH) { const float4 t3 = me[3]; SDD(lane) = make_float2(t3.x, t3.y); }
        SPP(lane, 3) = make_float4(1.f, 1.f, t.z, t.w);
        max_nc = max(__float_as_int(t.x), __float_as_int(t.y));
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) max_nc = max(max_nc, __shfl_xor(max_nc, dd));
    __builtin_amdgcn_wave_barrier();
    const float halfW = 0.5f * (float)d.W, halfH = 0.5f * (float)d.H;
    const int ablate = g_bwd_ablate;
    const int count = min(range.y - range.x, max_nc);
    const int shift = 8 * (part >> 1) + 2 * (part & 1);          // bit of the quadrant's first sub-block in the 16-bit mask
    const float pxs = (float)(qx + (row & 1) * 4), pys = (float)(qy + (row >> 1) * 4);   // this ROW's sub-block origin
    int qn = 0, src = 0;
    auto refill = [&]() {   // candidates src + 64 e + lane, e = 0, 1: coalesced loads of the masks and ids
        bool r[2];
        uint32_t ids[2], qm[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = src + 64 * e + lane;
            const int kk = range.x + (k < count ? k : count - 1);
            qm[e] = smask[kk];
            ids[e] = point_list[kk];
        }
        asm volatile("" ::"v"(qm[0]), "v"(qm[1]), "v"(ids[0]), "v"(ids[1]));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = src + 64 * e + lane;
            const uint32_t t = qm[e] >> shift;                   // bits 0, 1: upper two sub-blocks; bits 4, 5: lower two
            qm[e] = (t & 3u) | ((t >> 2) & 12u);
            r[e] = k < count && qm[e] != 0u;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned long long bal = __ballot(r[e]);
            if (r[e]) s_q[qn + (int)gp_mbcnt(bal)] = make_int2(src + 64 * e + lane, (int)(ids[e] | (qm[e] << 28)));
            qn += (int)__popcll(bal);
        }
        src += 128;
    };
    while (qn < 64 && src < count) refill();
    __builtin_amdgcn_wave_barrier();
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    int2 ne = make_int2(0x7fffffff, 0);
    if (lane < qn) {
        ne = s_q[lane];
        const size_t gid = (size_t)((uint32_t)ne.y & 0x0FFFFFFFu);
        n0 = rec[3 * gid]; n1 = rec[3 * gid + 1]; n2 = rec[3 * gid + 2];
    }
    auto fma2 = [](v2f a, v2f b, v2f c) { return (v2f){fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; };
    unsigned n_eval = 0, n_contr = 0;
    while (qn > 0) {
        const int nb = min(64, qn);
        const bool have = lane < nb;
        const uint32_t idm = (uint32_t)ne.y;
        const uint32_t id = idm & 0x0FFFFFFFu;
        const uint32_t m4 = have ? idm >> 28 : 0u;
        const float4 q0 = n0, q1 = n1, q2 = n2;
        // ---- stage the batch: records, zeroed accumulators, the four sub-block lists
        s_rec[lane][0] = q0;
        s_rec[lane][1] = make_float4(q1.x, q1.y, q2.x, q2.y);
        s_rec[lane][2] = make_float4(q2.z, q1.z, __int_as_float(have ? ne.x : 0x7fffffff), q2.w);
        s_gid[lane] = id;
        int nr0, nr1, nr2, nr3;
        {
            const unsigned long long b0 = __ballot((m4 & 1u) != 0u), b1 = __ballot((m4 & 2u) != 0u), b2 = __ballot((m4 & 4u) != 0u),
                                     b3 = __ballot((m4 & 8u) != 0u);
            if (m4 & 1u) s_rl[0][gp_mbcnt(b0)] = (unsigned char)lane;
            if (m4 & 2u) s_rl[1][gp_mbcnt(b1)] = (unsigned char)lane;
            if (m4 & 4u) s_rl[2][gp_mbcnt(b2)] = (unsigned char)lane;
            if (m4 & 8u) s_rl[3][gp_mbcnt(b3)] = (unsigned char)lane;
            nr0 = (int)__popcll(b0); nr1 = (int)__popcll(b1); nr2 = (int)__popcll(b2); nr3 = (int)__popcll(b3);
        }
        {   // pop the batch, top the queue up, start fetching the next batch's records
            int2 mv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) mv[e] = (64 + e * 64 + lane < qn) ? s_q[64 + e * 64 + lane] : make_int2(0, 0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 2; ++e) if (64 + e * 64 + lane < qn) s_q[e * 64 + lane] = mv[e];
            qn -= nb;
            __builtin_amdgcn_wave_barrier();
            while (qn < 64 && src < count) refill();
            __builtin_amdgcn_wave_barrier();
            n0 = make_float4(0.f, 0.f, 0.f, 0.f); n1 = n0; n2 = n0;
            ne = make_int2(0x7fffffff, 0);
            if (lane < qn) {
                ne = s_q[lane];
                const size_t gid = (size_t)((uint32_t)ne.y & 0x0FFFFFFFu);
                n0 = rec[3 * gid]; n1 = rec[3 * gid + 1]; n2 = rec[3 * gid + 2];
            }
        }
        const int n_mine = row == 0 ? nr0 : (row == 1 ? nr1 : (row == 2 ? nr2 : nr3));
        const int rounds = (ablate & 2) ? 0 : (max(max(nr0, nr1), max(nr2, nr3)) + 15) >> 4;
#pragma unroll 1
        for (int k = 0; k < rounds; ++k) {
            const int sl = 16 * k + li;
            const bool act = sl < n_mine;
            const int idx = act ? (int)s_rl[row][sl] : 0;
            const float4 r0 = s_rec[idx][0], r1 = s_rec[idx][1], r2 = s_rec[idx][2];
            // alpha is recomputed with the FORWARD's expression tree, bit for bit (see the quadrant kernel above)
            const float As = r0.z, Bs = r0.w, Cs = r1.x;
            const float lop = act ? r1.y : -INFINITY, zdep = r2.y;
            const int pos = act ? __float_as_int(r2.z) : 0x7fffffff;
            const v2f cr = {r1.z, r1.z}, cg = {r1.w, r1.w}, cb = {r2.x, r2.x};
            v2f a_op = {0.f, 0.f}, a_r = {0.f, 0.f}, a_g = {0.f, 0.f}, a_b = {0.f, 0.f}, a_d = {0.f, 0.f}, s_xx = {0.f, 0.f};
            float S_x = 0.f, S_y = 0.f, S_xy = 0.f, S_yy = 0.f, any_m = 0.f;
            const v2f dxs[2] = {(v2f){r0.x - pxs, r0.x - (pxs + 1.f)}, (v2f){r0.x - (pxs + 2.f), r0.x - (pxs + 3.f)}};
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const float dy = r0.y - (pys + (float)y);
                const float tB = Bs * dy, uC = fmaf(Cs * dy, dy, lop);
                v2f r_h = {0.f, 0.f}, r_hx = {0.f, 0.f};
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    const int pr = row * 8 + y * 2 + cp;
                    const float4 ncf = SPP(pr, 0);
                    const int ncx = __float_as_int(ncf.x), ncy = __float_as_int(ncf.y);
                    const float4 v0 = SPP(pr, 1), v1 = SPP(pr, 2), cy = SPP(pr, 3);
                    float2 v3 = make_float2(0.f, 0.f);
                    if (HAS_DEPTH) v3 = SDD(pr);
                    const v2f dx = dxs[cp];
                    const v2f pw = {fmaf(dx.x, fmaf(As, dx.x, tB), uC), fmaf(dx.y, fmaf(As, dx.y, tB), uC)};
                    const v2f E = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
                    if (MODE == 1 && act) n_eval += (pos < ncx) + (pos < ncy);
                    const bool c0 = (pos < ncx) && !(pw.x > lop) && !(E.x < 1.f / 255.f);
                    const bool c1 = (pos < ncy) && !(pw.y > lop) && !(E.y < 1.f / 255.f);
                    if (__builtin_amdgcn_ballot_w64(c0 || c1) != 0ull) {      // otherwise no row has anything to do at its pixel pair
                        if (MODE == 1) n_contr += (unsigned)c0 + (unsigned)c1;
                        const v2f Em = {c0 ? E.x : 0.f, c1 ? E.y : 0.f};
                        const v2f am = {fminf(0.99f, Em.x), fminf(0.99f, Em.y)};
                        any_m = fmaxf(any_m, fmaxf(am.x, am.y));
                        const v2f om = 1.f - am;
                        v2f cdot = cb * (v2f){v1.x, v1.y};
                        cdot = fma2(cg, (v2f){v0.z, v0.w}, cdot);
                        cdot = fma2(cr, (v2f){v0.x, v0.y}, cdot);
                        v2f dLd = {0.f, 0.f};
                        if (HAS_DEPTH) { dLd.x = v3.x; dLd.y = v3.y; cdot = fma2((v2f){zdep, zdep}, dLd, cdot); }
                        const v2f rom = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                        float il0 = om.x, il1 = om.y;
                        dpp_rowscan2_mul(il0, il1);                             // product over the ROW's splats up to and including this one
                        const v2f Tnext = (v2f){cy.x, cy.y} * (v2f){il0, il1};
                        const v2f Tj = Tnext * rom;
                        const v2f w = am * Tj;
                        const v2f sv = w * cdot;
                        float is0 = sv.x, is1 = sv.y;
                        dpp_rowscan2_add(is0, is1);
                        const v2f rem = {cy.z, cy.w};
                        const v2f tbv = {v1.z, v1.w};
                        const v2f suffix = rem - (v2f){is0, is1};
                        const v2f dL_dalpha = fma2(Tj, cdot, -((suffix + tbv) * rom));
                        a_r = fma2(w, (v2f){v0.x, v0.y}, a_r);
                        a_g = fma2(w, (v2f){v0.z, v0.w}, a_g);
                        a_b = fma2(w, (v2f){v1.x, v1.y}, a_b);
                        if (HAS_DEPTH) a_d = fma2(w, dLd, a_d);
                        const v2f h = Em * dL_dalpha;
                        a_op += h;
                        const v2f hx = h * dx;
                        r_h += h;
                        r_hx += hx;
                        s_xx = fma2(hx, dx, s_xx);
                        if (li == 15) SPP(pr, 3) = make_float4(Tnext.x, Tnext.y, cy.z - is0, cy.w - is1);     // the row's carry
                    }
                }
                const float rh = r_h.x + r_h.y, rhx = r_hx.x + r_hx.y;
                S_x += rhx;
                S_y = fmaf(dy, rh, S_y);
                S_xy = fmaf(dy, rhx, S_xy);
                S_yy = fmaf(dy * dy, rh, S_yy);
            }
            // ---- flush of the round: 16 consecutive lanes add to the 16 consecutive floats of ONE Gaussian's accumulator line (as the
            // quadrant kernel; a splat that reaches several sub-blocks is flushed once per sub-block)
            {
                const bool mine = act && any_m > 0.f;
                const float LN2 = 0.6931471805599453f;
                const float cxx = -2.f * LN2 * As, cxy = -LN2 * Bs, cyy = -2.f * LN2 * Cs;
                const float S_xx = s_xx.x + s_xx.y;
                s_fl[0][lane] = -(cxx * S_x + cxy * S_y) * halfW; s_fl[1][lane] = -(cyy * S_y + cxy * S_x) * halfH;
                s_fl[2][lane] = -0.5f * S_xx; s_fl[3][lane] = -S_xy; s_fl[4][lane] = -0.5f * S_yy;
                s_fl[5][lane] = mine ? (a_op.x + a_op.y) / r2.w : 0.f;
                s_fl[6][lane] = a_r.x + a_r.y; s_fl[7][lane] = a_g.x + a_g.y; s_fl[8][lane] = a_b.x + a_b.y;
                if (HAS_DEPTH) s_fl[9][lane] = a_d.x + a_d.y;
                s_flid[lane] = mine ? s_gid[idx] : 0xffffffffu;
                __builtin_amdgcn_wave_barrier();
                const int comp = lane & 15, sub = lane >> 4;
                if (comp < (HAS_DEPTH ? 10 : 9)) {
                    const float* src_c = &s_fl[comp][0];
#pragma unroll 4
                    for (int r = 0; r < 16; ++r) {
                        const int sp = r * 4 + sub;
                        const uint32_t gid = s_flid[sp];
                        if (gid != 0xffffffffu && !(ablate & 1)) atomicAdd(&g_mean2D[GP_ACC_STRIDE * (size_t)gid + comp], src_c[sp]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) { n_eval += __shfl_xor(n_eval, dd); n_contr += __shfl_xor(n_contr, dd); }
        if (lane == 0) { atomicAdd(&g_pair_counters[2], (unsigned long long)n_contr); atomicAdd(&g_pair_counters[3], (unsigned long long)n_eval); }
    }
#undef SPP
#undef SDD
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void gp_composite_bwd_sb_kernel(CB_ARGS) {
    gp_composite_bwd_sb_body<false, 0>(d, ranges, point_list, smask, rec, bg, out_color, out_depth, final_T, n_contrib, dL_dpix, dL_dpixdepth, g_mean2D, order);
}
__global__ __launch_bounds__(64) void gp_composite_bwd_sb_depth_kernel(CB_ARGS) {
    gp_composite_bwd_sb_body<true, 0>(d, ranges, point_list, smask, rec, bg, out_color, out_depth, final_T, n_contrib, dL_dpix, dL_dpixdepth, g_mean2D, order);
}
__global__ __launch_bounds__(64) void gp_composite_bwd_sb_count_kernel(CB_ARGS) {     // + pair counters (gp_debug_option(7, 2))
    gp_composite_bwd_sb_body<false, 1>(d, ranges, point_list, smask, rec, bg, out_color, out_depth, final_T, n_contrib, dL_dpix, dL_dpixdepth, g_mean2D, order);
}

// Adam on the workgroup's span of SH-rest coefficients, gradients taken from LDS (instead of unstage_sh + a later pass of the
// optimizer kernel over the same 180 B per Gaussian: the gradient is neither written nor read back, the coefficients are still in L2)
template <int CNT>
__device__ __forceinline__ void adam_unstage_sh(const float* s_sh, const AdamFuseDev& af, size_t off, int nblk, int tid) {
    constexpr int STRIDE = CNT | 1;
    const int total = nblk * CNT;
    float* __restrict__ p = af.p_rest + off; float* __restrict__ m = af.m_rest + off; float* __restrict__ v = af.v_rest + off;
    // The stream that bounds the kernel (28 B per coefficient, 1.35 GB per launch at 1 M Gaussians).  One 1024-float chunk per
    // iteration put three 16-byte loads in flight per thread, twelve waves per CU: ~36 KB per CU, i.e. latency x bandwidth for
    // ~5 TB/s and no more.  AU chunks per iteration: 3 AU loads in flight before the first update.
    constexpr int AU = 3;
    int e = tid * 4;
    for (; e + (AU - 1) * 1024 + 3 < total; e += AU * 1024) {
        float4 pv[AU], mv[AU], vv[AU];
#pragma unroll
        for (int c = 0; c < AU; ++c) {
            pv[c] = *(const float4*)(p + e + 1024 * c); mv[c] = *(const float4*)(m + e + 1024 * c); vv[c] = *(const float4*)(v + e + 1024 * c);
        }
#pragma unroll
        for (int c = 0; c < AU; ++c) {
            float* pp = (float*)&pv[c]; float* mm = (float*)&mv[c]; float* vq = (float*)&vv[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = e + 1024 * c + u;
                const int gi = idx / CNT, k = idx - gi * CNT;
                gp_adam_update(pp[u], s_sh[gi * STRIDE + k], mm[u], vq[u], af.b1, af.b2, af.eps, af.step_rest, af.bc2_sqrt);
            }
        }
#pragma unroll
        for (int c = 0; c < AU; ++c) {
            *(float4*)(p + e + 1024 * c) = pv[c]; *(float4*)(m + e + 1024 * c) = mv[c]; *(float4*)(v + e + 1024 * c) = vv[c];
        }
    }
    {   // the rest -- at most AU - 1 whole chunks and one ragged 16 bytes -- fetched together as well (one chunk per iteration,
        // these were two more dependent round trips per thread: as many as the whole unrolled part above)
        float4 pv[AU - 1], mv[AU - 1], vv[AU - 1];
        bool ok[AU - 1];
#pragma unroll
        for (int c = 0; c < AU - 1; ++c) {
            const int ec = e + 1024 * c;
            ok[c] = ec + 3 < total;
            const int el = ok[c] ? ec : 0;
            pv[c] = *(const float4*)(p + el); mv[c] = *(const float4*)(m + el); vv[c] = *(const float4*)(v + el);
        }
#pragma unroll
        for (int c = 0; c < AU - 1; ++c) {
            if (!ok[c]) continue;
            float* pp = (float*)&pv[c]; float* mm = (float*)&mv[c]; float* vq = (float*)&vv[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = e + 1024 * c + u;
                const int gi = idx / CNT, k = idx - gi * CNT;
                gp_adam_update(pp[u], s_sh[gi * STRIDE + k], mm[u], vq[u], af.b1, af.b2, af.eps, af.step_rest, af.bc2_sqrt);
            }
            *(float4*)(p + e + 1024 * c) = pv[c]; *(float4*)(m + e + 1024 * c) = mv[c]; *(float4*)(v + e + 1024 * c) = vv[c];
        }
#pragma unroll
        for (int c = 0; c < AU - 1; ++c) {              // a ragged last vector (partial workgroups only)
            const int ec = e + 1024 * c;
            if (!ok[c] && ec < total) {
                for (int u = 0; u < 4; ++u) {
                    const int idx = ec + u;
                    if (idx < total) {
                        const int gi = idx / CNT, k = idx - gi * CNT;
                        gp_adam_update(p[idx], s_sh[gi * STRIDE + k], m[idx], v[idx], af.b1, af.b2, af.eps, af.step_rest, af.bc2_sqrt);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// preprocess backward (per Gaussian) -- mirrors gpo_preprocess_bwd of the oracle
// ------------------------------------------------------------------------------------------------
template <int SH_MODE>
__device__ __forceinline__ void preprocess_bwd_body(
    RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ shs, const float* __restrict__ shs_rest, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
    const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped, const float* __restrict__ g_mean2D,
    const float* __restrict__ g_conic, const float* __restrict__ g_opacity, const float* __restrict__ g_color,
    const float* __restrict__ g_depth, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_dshs, float* __restrict__ dL_dshs_rest, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales, float* __restrict__ dL_drots,
    float* __restrict__ dL_dcov3D, int accumulate_shs, AdamFuseDev af) {
    __shared__ float s_sh[SH_MODE == 0 ? 1 : 256 * 49];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * 256;
    const int i = base + tid;
    const int nblk = min(256, d.N - base);
    const bool use_sh = dL_dcolors == nullptr;
    // SH_MODE 1/2: coefficients are staged in LDS and their gradients REPLACE them in place (one channel's 16
    // coefficients are in registers at a time), then leave as one coalesced span per workgroup.
    constexpr int SROW = SH_MODE == 1 ? 49 : 45;        // LDS row stride (odd: conflict-free)
    constexpr int SOFF = SH_MODE == 1 ? 0 : 3;          // first coefficient float held in the row
    float dsh_dc[3] = {0.f, 0.f, 0.f};                  // SH_MODE 2: gradient of the dc coefficient (separate tensor)
    // Everything the thread reads from global memory, issued TOGETHER up front, in front of the SH staging (they ride on its first round trip) (clamped index, unconditional): as loads placed
    // where the values are used -- behind the visibility branch, inside the covariance / colour / rotation sections -- they were
    // some ten dependent round trips to memory per wave.  The five accumulator pointers are offsets into ONE 64-byte line per
    // Gaussian (gp_capi_raster.hip: mean2D 0..1 | conic 2..4 | opacity 5 | colour 6..8 | depth 9): three 16-byte loads.
    const int ii = i < d.N ? i : d.N - 1;
    const int rad_i = radii[ii];
    const float4* accl = (const float4*)(g_mean2D + GP_ACC_STRIDE * (size_t)ii);
    const float4 A0 = accl[0], A1 = accl[1], A2 = accl[2];
    const float px = means3D[3 * ii], py = means3D[3 * ii + 1], pz = means3D[3 * ii + 2];
    float sc3[3] = {0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    if (!cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sc3[k] = scales[3 * ii + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] = rotations[4 * ii + k];
    }
    const float raw_o = d.raw_opacity ? d.raw_opacity[ii] : 0.f;
    const uint8_t cl = dL_dcolors ? (uint8_t)0 : clamped[ii];
    const float acc_mean2D[2] = {A0.x, A0.y}, acc_conic[3] = {A0.z, A0.w, A1.x}, acc_opacity = A1.y,
                acc_color[3] = {A1.z, A1.w, A2.x}, acc_depth = A2.y;
    if (SH_MODE != 0) {
        if (use_sh && d.D > 0) {
            if (SH_MODE == 1) stage_sh<48>(s_sh, shs + (size_t)base * 48, nblk, tid);
            if (SH_MODE == 2) stage_sh<45>(s_sh, shs_rest + (size_t)base * 45, nblk, tid);
        }
        __syncthreads();
    }
    do {
    if (i >= d.N) break;
    const bool vis = rad_i > 0;
    dL_dmeans2D[3 * i] = vis ? acc_mean2D[0] : 0.f;
    dL_dmeans2D[3 * i + 1] = vis ? acc_mean2D[1] : 0.f;
    dL_dmeans2D[3 * i + 2] = 0.f;
    {
        const float go = vis ? acc_opacity : 0.f;
        float out = go;
        if (d.raw_opacity) {        // gp_act_bwd_kernel's expression (life = 1)
            const float so = 1.f / (1.f + expf(-raw_o));
            out = go * 1.f * so * (1.f - so);
        }
        dL_dopacities[i] = out;
    }
    // A culled Gaussian (radii = 0: 0.4 % of the bench scene) runs the same arithmetic on whatever its position gives -- its
    // accumulator line is zero, so every colour / SH term is an exact zero -- and `vis` SELECTS zeros where a division by a
    // non-positive depth could have produced NaN.  (As an early-out branch it let the compiler sink the loads of scales, rotations
    // and the clamp flags behind it: a second and third dependent round trip for every wave.)
    const float3 pv = xform4x3(view, px, py, pz);
    float c6[6];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
    } else {
        if (d.raw_opacity) {        // (log-scales: gp_act_fwd_kernel's exp; applied here, behind the loads' and the SH staging's round trips)
#pragma unroll
            for (int k = 0; k < 3; ++k) sc3[k] = expf(sc3[k]);
        }
        compute_cov3D(sc3, d.scale_mod, q4, c6);
    }
    float abc[3];
    ProjCtx cx;
    compute_cov2D(pv, d.fx, d.fy, d.tanfovx, d.tanfovy, c6, view, abc, &cx);
    const float a = abc[0] + 0.3f, b = abc[1], c = abc[2] + 0.3f;
    const float det = a * c - b * b;
    const float gA = acc_conic[0], gB = acc_conic[1], gC = acc_conic[2];
    float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f;
    float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float W0[3] = {view[0], view[4], view[8]};
    const float W1[3] = {view[1], view[5], view[9]};
    const float W2[3] = {view[2], view[6], view[10]};
    if (det != 0.f) {
        const float d2 = 1.f / (det * det);
        const float dL_da = d2 * (-c * c * gA + b * c * gB - b * b * gC);
        const float dL_db = d2 * (2.f * b * c * gA - (a * c + b * b) * gB + 2.f * a * b * gC);
        const float dL_dc = d2 * (-b * b * gA + a * b * gB - a * a * gC);
        const float* T0 = cx.T0;
        const float* T1 = cx.T1;
        g6[0] = dL_da * T0[0] * T0[0] + dL_db * T0[0] * T1[0] + dL_dc * T1[0] * T1[0];
        g6[3] = dL_da * T0[1] * T0[1] + dL_db * T0[1] * T1[1] + dL_dc * T1[1] * T1[1];
        g6[5] = dL_da * T0[2] * T0[2] + dL_db * T0[2] * T1[2] + dL_dc * T1[2] * T1[2];
        g6[1] = 2.f * dL_da * T0[0] * T0[1] + dL_db * (T0[0] * T1[1] + T0[1] * T1[0]) + 2.f * dL_dc * T1[0] * T1[1];
        g6[2] = 2.f * dL_da * T0[0] * T0[2] + dL_db * (T0[0] * T1[2] + T0[2] * T1[0]) + 2.f * dL_dc * T1[0] * T1[2];
        g6[4] = 2.f * dL_da * T0[1] * T0[2] + dL_db * (T0[1] * T1[2] + T0[2] * T1[1]) + 2.f * dL_dc * T1[1] * T1[2];
        const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        float dT0[3], dT1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float st0 = S[3 * r] * T0[0] + S[3 * r + 1] * T0[1] + S[3 * r + 2] * T0[2];
            const float st1 = S[3 * r] * T1[0] + S[3 * r + 1] * T1[1] + S[3 * r + 2] * T1[2];
            dT0[r] = 2.f * dL_da * st0 + dL_db * st1;
            dT1[r] = 2.f * dL_dc * st1 + dL_db * st0;
        }
        const float dJ00 = dT0[0] * W0[0] + dT0[1] * W0[1] + dT0[2] * W0[2];
        const float dJ02 = dT0[0] * W2[0] + dT0[1] * W2[1] + dT0[2] * W2[2];
        const float dJ11 = dT1[0] * W1[0] + dT1[1] * W1[1] + dT1[2] * W1[2];
        const float dJ12 = dT1[0] * W2[0] + dT1[1] * W2[1] + dT1[2] * W2[2];
        const float itz = 1.f / cx.tz, itz2 = itz * itz, itz3 = itz2 * itz;
        const float dtx = cx.gx * (-d.fx * itz2) * dJ02;
        const float dty = cx.gy * (-d.fy * itz2) * dJ12;
        const float dtz = -d.fx * itz2 * dJ00 - d.fy * itz2 * dJ11 + (2.f * d.fx * cx.tx) * itz3 * dJ02 +
                          (2.f * d.fy * cx.ty) * itz3 * dJ12;
        gm0 += W0[0] * dtx + W1[0] * dty + W2[0] * dtz;
        gm1 += W0[1] * dtx + W1[1] * dty + W2[1] * dtz;
        gm2 += W0[2] * dtx + W1[2] * dty + W2[2] * dtz;
    }
    {   // depth
        const float gd = acc_depth;
        gm0 += view[2] * gd; gm1 += view[6] * gd; gm2 += view[10] * gd;
    }
    {   // mean2D (NDC) -> mean3D
        const float4 ph = xform4x4(proj, px, py, pz);
        const float mw = 1.f / (ph.w + 0.0000001f);
        const float mul1 = ph.x * mw * mw, mul2 = ph.y * mw * mw;
        const float g2x = acc_mean2D[0], g2y = acc_mean2D[1];
        gm0 += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
        gm1 += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
        gm2 += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
    }
    if (dL_dcolors) {
        dL_dcolors[3 * i] = acc_color[0]; dL_dcolors[3 * i + 1] = acc_color[1]; dL_dcolors[3 * i + 2] = acc_color[2];
    } else {
        const float ddx0 = px - campos[0], ddy0 = py - campos[1], ddz0 = pz - campos[2];
        const float len = sqrtf(fmaf(ddx0, ddx0, fmaf(ddy0, ddy0, ddz0 * ddz0)));
        const float inv = 1.f / len;
        const float x = ddx0 * inv, y = ddy0 * inv, z = ddz0 * inv;
        float ddir0 = 0.f, ddir1 = 0.f, ddir2 = 0.f;
        const int D = d.D;
        const int used = (D + 1) * (D + 1);
        const float* sh_g = shs + (size_t)i * d.M * 3;          // SH_MODE 0
        float* dsh_g = dL_dshs + (size_t)i * d.M * 3;           // SH_MODE 0
        float* row = s_sh + tid * SROW - SOFF;                  // SH_MODE 1/2: row[3 k + ch], k >= SOFF / 3
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float g = ((cl >> ch) & 1) ? 0.f : acc_color[ch];
            // this channel's coefficients first (their slots are about to be overwritten by the gradients)
            float sh[16];
#pragma unroll
            for (int k = 1; k < 16; ++k) sh[k] = (k < used) ? (SH_MODE == 0 ? sh_g[3 * k + ch] : row[3 * k + ch]) : 0.f;
            float dsh[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) dsh[k] = 0.f;
            dsh[0] = SH_C0 * g;
            if (D > 0) {
                dsh[1] = -SH_C1 * y * g;
                dsh[2] = SH_C1 * z * g;
                dsh[3] = -SH_C1 * x * g;
                float ddx = -SH_C1 * sh[3];
                float ddy = -SH_C1 * sh[1];
                float ddz = SH_C1 * sh[2];
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    dsh[4] = SH_C2[0] * xy * g;
                    dsh[5] = SH_C2[1] * yz * g;
                    dsh[6] = SH_C2[2] * (2.f * zz - xx - yy) * g;
                    dsh[7] = SH_C2[3] * xz * g;
                    dsh[8] = SH_C2[4] * (xx - yy) * g;
                    ddx += SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] +
                           SH_C2[4] * 2.f * x * sh[8];
                    ddy += SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] +
                           SH_C2[4] * 2.f * -y * sh[8];
                    ddz += SH_C2[1] * y * sh[5] + SH_C2[2] * 4.f * z * sh[6] + SH_C2[3] * x * sh[7];
                    if (D > 2) {
                        dsh[9] = SH_C3[0] * y * (3.f * xx - yy) * g;
                        dsh[10] = SH_C3[1] * xy * z * g;
                        dsh[11] = SH_C3[2] * y * (4.f * zz - xx - yy) * g;
                        dsh[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                        dsh[13] = SH_C3[4] * x * (4.f * zz - xx - yy) * g;
                        dsh[14] = SH_C3[5] * z * (xx - yy) * g;
                        dsh[15] = SH_C3[6] * x * (xx - 3.f * yy) * g;
                        ddx += SH_C3[0] * sh[9] * 6.f * xy + SH_C3[1] * sh[10] * yz +
                               SH_C3[2] * sh[11] * -2.f * xy + SH_C3[3] * sh[12] * -6.f * xz +
                               SH_C3[4] * sh[13] * (4.f * zz - 3.f * xx - yy) + SH_C3[5] * sh[14] * 2.f * xz +
                               SH_C3[6] * sh[15] * 3.f * (xx - yy);
                        ddy += SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz +
                               SH_C3[2] * sh[11] * (4.f * zz - xx - 3.f * yy) + SH_C3[3] * sh[12] * -6.f * yz +
                               SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz +
                               SH_C3[6] * sh[15] * -6.f * xy;
                        ddz += SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 8.f * yz +
                               SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[13] * 8.f * xz +
                               SH_C3[5] * sh[14] * (xx - yy);
                    }
                }
                ddir0 += ddx * g; ddir1 += ddy * g; ddir2 += ddz * g;
            }
            if (SH_MODE == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) if (k < d.M) dsh_g[3 * k + ch] = vis ? dsh[k] : 0.f;
                for (int k = 16; k < d.M; ++k) dsh_g[3 * k + ch] = 0.f;
            } else {
                if (SH_MODE == 2) dsh_dc[ch] = vis ? dsh[0] : 0.f; else row[ch] = vis ? dsh[0] : 0.f;
#pragma unroll
                for (int k = 1; k < 16; ++k) row[3 * k + ch] = vis ? dsh[k] : 0.f;
            }
        }
        const float dot = x * ddir0 + y * ddir1 + z * ddir2;
        gm0 += (ddir0 - x * dot) * inv;
        gm1 += (ddir1 - y * dot) * inv;
        gm2 += (ddir2 - z * dot) * inv;
    }
    dL_dmeans3D[3 * i] = vis ? gm0 : 0.f; dL_dmeans3D[3 * i + 1] = vis ? gm1 : 0.f; dL_dmeans3D[3 * i + 2] = vis ? gm2 : 0.f;
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = vis ? g6[k] : 0.f;
    } else {
        const float Gs[9] = {g6[0], 0.5f * g6[1], 0.5f * g6[2], 0.5f * g6[1], g6[3], 0.5f * g6[4], 0.5f * g6[2], 0.5f * g6[4], g6[5]};
        const float* q = q4;
        float Rm[9];
        quat_to_R(q[0], q[1], q[2], q[3], Rm);
        const float s[3] = {d.scale_mod * sc3[0], d.scale_mod * sc3[1], d.scale_mod * sc3[2]};
        float L[9], dLm[9], dR[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) L[3 * r + k] = Rm[3 * r + k] * s[k];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dLm[3 * r + k] = 2.f * (Gs[3 * r] * L[k] + Gs[3 * r + 1] * L[3 + k] + Gs[3 * r + 2] * L[6 + k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) { acc += dLm[3 * r + k] * Rm[3 * r + k]; dR[3 * r + k] = dLm[3 * r + k] * s[k]; }
            const float gs = vis ? acc * d.scale_mod : 0.f;
            dL_dscales[3 * i + k] = d.raw_opacity ? gs * sc3[k] : gs;        // (raw: gp_act_bwd_kernel's g exp(s))
        }
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        dL_drots[4 * i + 0] = vis ? 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]) : 0.f;
        dL_drots[4 * i + 1] = vis ? 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]) : 0.f;
        dL_drots[4 * i + 2] = vis ? 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]) : 0.f;
        dL_drots[4 * i + 3] = vis ? 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]) : 0.f;
    }
    } while (0);
    if (SH_MODE != 0 && use_sh) {
        // gradients of the SH coefficients leave through LDS as one coalesced span per workgroup
        __syncthreads();
        if (SH_MODE == 1) {
            unstage_sh<48>(s_sh, dL_dshs + (size_t)base * 48, nblk, tid, accumulate_shs != 0);
        } else if (af.on) {
            // the optimizer step of the SH coefficients, here: no gradient leaves the kernel (skip: an invalid frame, nothing is touched)
            if (!(af.skip && *af.skip != 0)) {
                if (i < d.N) {      // (all nine values first: through the references the three updates were nine dependent round trips)
                    float pp[3], mm[3], vq[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { pp[k] = af.p_dc[3 * (size_t)i + k]; mm[k] = af.m_dc[3 * (size_t)i + k]; vq[k] = af.v_dc[3 * (size_t)i + k]; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) gp_adam_update(pp[k], dsh_dc[k], mm[k], vq[k], af.b1, af.b2, af.eps, af.step_dc, af.bc2_sqrt);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { af.p_dc[3 * (size_t)i + k] = pp[k]; af.m_dc[3 * (size_t)i + k] = mm[k]; af.v_dc[3 * (size_t)i + k] = vq[k]; }
                }
                adam_unstage_sh<45>(s_sh, af, (size_t)base * 45, nblk, tid);
            }
        } else {
            if (i < d.N) {
                if (accumulate_shs) { dL_dshs[3 * (size_t)i] += dsh_dc[0]; dL_dshs[3 * (size_t)i + 1] += dsh_dc[1]; dL_dshs[3 * (size_t)i + 2] += dsh_dc[2]; }
                else { dL_dshs[3 * (size_t)i] = dsh_dc[0]; dL_dshs[3 * (size_t)i + 1] = dsh_dc[1]; dL_dshs[3 * (size_t)i + 2] = dsh_dc[2]; }
            }
            unstage_sh<45>(s_sh, dL_dshs_rest + (size_t)base * 45, nblk, tid, accumulate_shs != 0);
        }
    }
}

#define PB_ARGS RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales, \
    const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ shs_rest, \
    const float* __restrict__ cov3D_precomp, const float* __restrict__ view, const float* __restrict__ proj, \
    const float* __restrict__ campos, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped, \
    const float* __restrict__ g_mean2D, const float* __restrict__ g_conic, const float* __restrict__ g_opacity, \
    const float* __restrict__ g_color, const float* __restrict__ g_depth, float* __restrict__ dL_dmeans3D, \
    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dshs, float* __restrict__ dL_dshs_rest, \
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales, \
    float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D, int accumulate_shs, AdamFuseDev af
#define PB_PASS d, means3D, scales, rotations, shs, shs_rest, cov3D_precomp, view, proj, campos, radii, clamped, g_mean2D, \
    g_conic, g_opacity, g_color, g_depth, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dshs_rest, dL_dcolors, dL_dopacities, \
    dL_dscales, dL_drots, dL_dcov3D, accumulate_shs, af
__global__ __launch_bounds__(256) void gp_preprocess_bwd_kernel(PB_ARGS) { preprocess_bwd_body<0>(PB_PASS); }
__global__ __launch_bounds__(256) void gp_preprocess_bwd_sh16_kernel(PB_ARGS) { preprocess_bwd_body<1>(PB_PASS); }
__global__ __launch_bounds__(256) void gp_preprocess_bwd_split_kernel(PB_ARGS) { preprocess_bwd_body<2>(PB_PASS); }

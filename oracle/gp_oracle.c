/*
 * gp_oracle.c -- CPU ORACLE for the dynamic-Gaussian rasterizer hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (gaussianprediction_amd/)
 * never imports, links or executes anything under oracle/.
 *
 * PARITY STATUS: **parity unpinned** for the rasterizer arithmetic.
 *   The reference rasterizer (`diff_gaussian_rasterization`, submodule
 *   submodules/diff-gaussian-rasterization-w-depth, url in /root/reference/.gitmodules:4-6) is an
 *   EMPTY, un-vendored, un-pinned git submodule: no source, no wheel, no tests, no golden images
 *   exist in /root/reference.  This file therefore restates the *published* tile-rasterizer
 *   algorithm of 3D Gaussian Splatting (Kerbl et al. 2023, "diff-gaussian-rasterization", plus the
 *   depth / per-pixel-index outputs the fork adds), anchored on the reference's own call sites:
 *     - argument list, 4-tuple return:      gaussian_renderer/__init__.py:37-52, 98-106
 *     - radii>0 <=> visible:                gaussian_renderer/__init__.py:112
 *     - means2D.grad[:, :2] is screen grad: scene/gaussian_model.py:757
 *     - tidx is [H,W], -1 = nothing:        eval.py:39-45
 *     - matrices are row-vector/transposed: scene/cameras.py:59-62
 *   and on the reference's own Python restatements of two sub-steps, which ARE pinned by golden
 *   vectors (tests/golden/):
 *     - cov3D = R S S^T R^T, packed [xx,xy,xz,yy,yz,zz]:  utils/general_utils.py:64-110,
 *                                                        scene/gaussian_model.py:35-39
 *     - SH -> RGB (+0.5, clamp >= 0):                     utils/sh_utils.py:57-112,
 *                                                        gaussian_renderer/__init__.py:86-91
 *   Declared (not derivable from the reference) semantics: near plane 0.2, 16x16 tiles, 3-sigma
 *   radius, +0.3 px^2 low-pass, alpha = min(0.99, o*G) (evaluated as exp2 of one folded exponent, see
 *   gauss_exponent), skip alpha < 1/255, stop when T < 1e-4,
 *   depth image = sum_i z_i alpha_i T_i, tidx = index of the Gaussian with the largest blending
 *   weight alpha_i T_i at that pixel (first wins on ties), -1 if none.
 *
 * Build:  float32  -> libgp_oracle_f32.so   (same expression trees as the HIP kernels)
 *         float64  -> libgp_oracle_f64.so   (-DGP_F64: shadow mode, gradient ground truth)
 * Both are compiled with -ffp-contract=off; every fused multiply-add is written explicitly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef GP_F64
typedef double real;
#define FMA(a, b, c) fma((a), (b), (c))
#define SQRT(x) sqrt(x)
#define EXPR(x) exp(x)
#define EXP2R(x) exp2(x)
#define LOG2R(x) log2(x)
#define CEILR(x) ceil(x)
#define FMINR(a, b) fmin((a), (b))
#define FMAXR(a, b) fmax((a), (b))
#define FABSR(a) fabs(a)
#else
typedef float real;
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(x) sqrtf(x)
#define EXPR(x) expf(x)
#define EXP2R(x) exp2f(x)
#define LOG2R(x) log2f(x)
#define CEILR(x) ceilf(x)
#define FMINR(a, b) fminf((a), (b))
#define FMAXR(a, b) fmaxf((a), (b))
#define FABSR(a) fabsf(a)
#endif

#define R(x) ((real)(x))
#define TILE 16

int gpo_real_bytes(void) { return (int)sizeof(real); }

/* SH constants: utils/sh_utils.py:26-44 */
static const real SH_C0 = R(0.28209479177387814);
static const real SH_C1 = R(0.4886025119029199);
static const real SH_C2[5] = {R(1.0925484305920792), R(-1.0925484305920792), R(0.31539156525252005),
                              R(-1.0925484305920792), R(0.5462742152960396)};
static const real SH_C3[7] = {R(-0.5900435899266435), R(2.890611442640554), R(-0.4570457994644658),
                              R(0.3731763325901154), R(-0.4570457994644658), R(1.445305721320277),
                              R(-0.5900435899266435)};

/* ------------------------------------------------------------------------------------------- */
/* small helpers                                                                               */
/* ------------------------------------------------------------------------------------------- */

/* p_view = V p (V column-major in m: reference passes world_view_transform = W2C^T row-major,
 * scene/cameras.py:59, whose memory image is column-major W2C). */
static inline void xform4x3(const real* m, real x, real y, real z, real* o) {
    o[0] = FMA(m[0], x, FMA(m[4], y, FMA(m[8], z, m[12])));
    o[1] = FMA(m[1], x, FMA(m[5], y, FMA(m[9], z, m[13])));
    o[2] = FMA(m[2], x, FMA(m[6], y, FMA(m[10], z, m[14])));
}
static inline void xform4x4(const real* m, real x, real y, real z, real* o) {
    o[0] = FMA(m[0], x, FMA(m[4], y, FMA(m[8], z, m[12])));
    o[1] = FMA(m[1], x, FMA(m[5], y, FMA(m[9], z, m[13])));
    o[2] = FMA(m[2], x, FMA(m[6], y, FMA(m[10], z, m[14])));
    o[3] = FMA(m[3], x, FMA(m[7], y, FMA(m[11], z, m[15])));
}

/* rotation matrix (row-major, standard) from UN-normalised quaternion (r,x,y,z):
 * utils/general_utils.py:78-99 minus the normalisation (the rasterizer receives the already
 * normalised q from GaussianModel.get_rotation_, scene/gaussian_model.py:314-315). */
static inline void quat_to_R(const real* q, real* Rm) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
    Rm[0] = R(1) - R(2) * (y * y + z * z);
    Rm[1] = R(2) * (x * y - r * z);
    Rm[2] = R(2) * (x * z + r * y);
    Rm[3] = R(2) * (x * y + r * z);
    Rm[4] = R(1) - R(2) * (x * x + z * z);
    Rm[5] = R(2) * (y * z - r * x);
    Rm[6] = R(2) * (x * z - r * y);
    Rm[7] = R(2) * (y * z + r * x);
    Rm[8] = R(1) - R(2) * (x * x + y * y);
}

/* cov3D = (R S)(R S)^T packed [xx,xy,xz,yy,yz,zz]: scene/gaussian_model.py:35-39,
 * utils/general_utils.py:64-73,101-110 */
static inline void compute_cov3D(const real* scale, real mod, const real* q, real* c6) {
    real Rm[9];
    quat_to_R(q, Rm);
    real s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    real L[9];
    L[0] = Rm[0] * s0; L[1] = Rm[1] * s1; L[2] = Rm[2] * s2;
    L[3] = Rm[3] * s0; L[4] = Rm[4] * s1; L[5] = Rm[5] * s2;
    L[6] = Rm[6] * s0; L[7] = Rm[7] * s1; L[8] = Rm[8] * s2;
    c6[0] = FMA(L[0], L[0], FMA(L[1], L[1], L[2] * L[2]));
    c6[1] = FMA(L[0], L[3], FMA(L[1], L[4], L[2] * L[5]));
    c6[2] = FMA(L[0], L[6], FMA(L[1], L[7], L[2] * L[8]));
    c6[3] = FMA(L[3], L[3], FMA(L[4], L[4], L[5] * L[5]));
    c6[4] = FMA(L[3], L[6], FMA(L[4], L[7], L[5] * L[8]));
    c6[5] = FMA(L[6], L[6], FMA(L[7], L[7], L[8] * L[8]));
}

/* EWA projection: T = J W (2x3), cov2D = T Sigma T^T. Returns a,b,c WITHOUT the 0.3 dilation
 * and the T rows, the clamped t, and the clamp gates (for backward). */
typedef struct {
    real T0[3], T1[3];
    real tx, ty, tz;      /* tx,ty after the 1.3*tanfov clamp */
    real gx, gy;          /* gradient gates: 1 if unclamped */
    real fx, fy;
} proj_ctx;

static inline void compute_cov2D(const real* pv, real fx, real fy, real tanfovx, real tanfovy,
                                 const real* c6, const real* view, real* abc, proj_ctx* ctx) {
    real tz = pv[2];
    real limx = R(1.3) * tanfovx, limy = R(1.3) * tanfovy;
    real txtz = pv[0] / tz, tytz = pv[1] / tz;
    real cx = FMINR(limx, FMAXR(-limx, txtz));
    real cy = FMINR(limy, FMAXR(-limy, tytz));
    real tx = cx * tz, ty = cy * tz;
    real itz = R(1) / tz;
    real itz2 = itz * itz;
    real J00 = fx * itz;
    real J02 = -(fx * tx) * itz2;
    real J11 = fy * itz;
    real J12 = -(fy * ty) * itz2;
    /* W rows (standard view rotation): W_ij = view[j*4+i] */
    real W0[3] = {view[0], view[4], view[8]};
    real W1[3] = {view[1], view[5], view[9]};
    real W2[3] = {view[2], view[6], view[10]};
    real T0[3], T1[3];
    for (int k = 0; k < 3; ++k) {
        T0[k] = FMA(J00, W0[k], J02 * W2[k]);
        T1[k] = FMA(J11, W1[k], J12 * W2[k]);
    }
    /* v = Sigma T^T rows */
    real S0[3] = {c6[0], c6[1], c6[2]};
    real S1[3] = {c6[1], c6[3], c6[4]};
    real S2[3] = {c6[2], c6[4], c6[5]};
    real u0[3], u1[3];
    u0[0] = FMA(S0[0], T0[0], FMA(S0[1], T0[1], S0[2] * T0[2]));
    u0[1] = FMA(S1[0], T0[0], FMA(S1[1], T0[1], S1[2] * T0[2]));
    u0[2] = FMA(S2[0], T0[0], FMA(S2[1], T0[1], S2[2] * T0[2]));
    u1[0] = FMA(S0[0], T1[0], FMA(S0[1], T1[1], S0[2] * T1[2]));
    u1[1] = FMA(S1[0], T1[0], FMA(S1[1], T1[1], S1[2] * T1[2]));
    u1[2] = FMA(S2[0], T1[0], FMA(S2[1], T1[1], S2[2] * T1[2]));
    abc[0] = FMA(T0[0], u0[0], FMA(T0[1], u0[1], T0[2] * u0[2]));
    abc[1] = FMA(T0[0], u1[0], FMA(T0[1], u1[1], T0[2] * u1[2]));
    abc[2] = FMA(T1[0], u1[0], FMA(T1[1], u1[1], T1[2] * u1[2]));
    if (ctx) {
        for (int k = 0; k < 3; ++k) { ctx->T0[k] = T0[k]; ctx->T1[k] = T1[k]; }
        ctx->tx = tx; ctx->ty = ty; ctx->tz = tz;
        ctx->gx = (txtz < -limx || txtz > limx) ? R(0) : R(1);
        ctx->gy = (tytz < -limy || tytz > limy) ? R(0) : R(1);
        ctx->fx = fx; ctx->fy = fy;
    }
}

/* SH basis evaluation, utils/sh_utils.py:57-112 (deg 0..3), layout shs[k][ch]. */
static inline void sh_to_rgb(int deg, const real* sh /*[M][3]*/, const real* dir, real* out3) {
    real x = dir[0], y = dir[1], z = dir[2];
    for (int ch = 0; ch < 3; ++ch) {
        real res = SH_C0 * sh[0 * 3 + ch];
        if (deg > 0) {
            res = res - SH_C1 * y * sh[1 * 3 + ch] + SH_C1 * z * sh[2 * 3 + ch] - SH_C1 * x * sh[3 * 3 + ch];
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * sh[4 * 3 + ch] + SH_C2[1] * yz * sh[5 * 3 + ch] +
                      SH_C2[2] * (R(2) * zz - xx - yy) * sh[6 * 3 + ch] + SH_C2[3] * xz * sh[7 * 3 + ch] +
                      SH_C2[4] * (xx - yy) * sh[8 * 3 + ch];
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (R(3) * xx - yy) * sh[9 * 3 + ch] +
                          SH_C3[1] * xy * z * sh[10 * 3 + ch] +
                          SH_C3[2] * y * (R(4) * zz - xx - yy) * sh[11 * 3 + ch] +
                          SH_C3[3] * z * (R(2) * zz - R(3) * xx - R(3) * yy) * sh[12 * 3 + ch] +
                          SH_C3[4] * x * (R(4) * zz - xx - yy) * sh[13 * 3 + ch] +
                          SH_C3[5] * z * (xx - yy) * sh[14 * 3 + ch] +
                          SH_C3[6] * x * (xx - R(3) * yy) * sh[15 * 3 + ch];
                }
            }
        }
        out3[ch] = res;
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int f2i_sat(real v) {
    v = FMINR(R(1e9), FMAXR(R(-1e9), v));
    return (int)v; /* truncation toward zero */
}

/* ------------------------------------------------------------------------------------------- */
/* 1. preprocess forward (per Gaussian)                                                        */
/* ------------------------------------------------------------------------------------------- */
int gpo_preprocess_fwd(int N, int D, int M, const real* means3D, const real* scales, real scale_modifier,
                       const real* rotations, const real* opacities, const real* shs,
                       const real* colors_precomp, const real* cov3D_precomp, const real* view,
                       const real* proj, const real* campos, int W, int H, real tanfovx, real tanfovy,
                       int32_t* radii, real* xy, real* depths, real* cov3D, real* rgb, real* conic_opacity,
                       int32_t* rect, uint32_t* tiles_touched, uint8_t* clamped) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const real fx = (real)W / (R(2) * tanfovx), fy = (real)H / (R(2) * tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        depths[i] = 0;
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = 0;
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 4; ++k) { conic_opacity[4 * i + k] = 0; rect[4 * i + k] = 0; }
        real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        real pv[3];
        xform4x3(view, px, py, pz, pv);
        if (!(pv[2] > R(0.2))) continue; /* near-plane cull */
        real ph[4];
        xform4x4(proj, px, py, pz, ph);
        real pw = R(1) / (ph[3] + R(0.0000001));
        real ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        real c6[6];
        if (cov3D_precomp) {
            for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
        } else {
            compute_cov3D(scales + 3 * i, scale_modifier, rotations + 4 * i, c6);
        }
        real abc[3];
        compute_cov2D(pv, fx, fy, tanfovx, tanfovy, c6, view, abc, NULL);
        real a = abc[0] + R(0.3), b = abc[1], c = abc[2] + R(0.3);
        real det = a * c - b * b;
        if (det == R(0)) continue;
        real det_inv = R(1) / det;
        real conx = c * det_inv, cony = -b * det_inv, conz = a * det_inv;
        real mid = R(0.5) * (a + c);
        real sq = SQRT(FMAXR(R(0.1), mid * mid - det));
        real l1 = mid + sq, l2 = mid - sq;
        real rad_f = CEILR(R(3) * SQRT(FMAXR(l1, l2)));
        int rad = f2i_sat(rad_f);
        real pix = ((ndcx + R(1)) * (real)W - R(1)) * R(0.5);
        real piy = ((ndcy + R(1)) * (real)H - R(1)) * R(0.5);
        int minx = clampi(f2i_sat((pix - rad_f) / R(TILE)), 0, gx);
        int miny = clampi(f2i_sat((piy - rad_f) / R(TILE)), 0, gy);
        int maxx = clampi(f2i_sat((pix + rad_f + R(TILE - 1)) / R(TILE)), 0, gx);
        int maxy = clampi(f2i_sat((piy + rad_f + R(TILE - 1)) / R(TILE)), 0, gy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;
        if (colors_precomp) {
            for (int k = 0; k < 3; ++k) rgb[3 * i + k] = colors_precomp[3 * i + k];
        } else {
            real dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
            real len = SQRT(FMA(dx, dx, FMA(dy, dy, dz * dz)));
            real inv = R(1) / len;
            real dir[3] = {dx * inv, dy * inv, dz * inv};
            real col[3];
            sh_to_rgb(D, shs + (size_t)i * M * 3, dir, col);
            for (int k = 0; k < 3; ++k) {
                real v = col[k] + R(0.5);
                clamped[3 * i + k] = (v < R(0));
                rgb[3 * i + k] = FMAXR(v, R(0));
            }
        }
        depths[i] = pv[2];
        radii[i] = rad;
        xy[2 * i] = pix; xy[2 * i + 1] = piy;
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = c6[k];
        conic_opacity[4 * i] = conx; conic_opacity[4 * i + 1] = cony; conic_opacity[4 * i + 2] = conz;
        conic_opacity[4 * i + 3] = opacities[i];
        rect[4 * i] = minx; rect[4 * i + 1] = miny; rect[4 * i + 2] = maxx; rect[4 * i + 3] = maxy;
        tiles_touched[i] = (uint32_t)((maxx - minx) * (maxy - miny));
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* 2. binning: duplicate per touched tile, stable sort by (tile, depth), per-tile ranges        */
/* ------------------------------------------------------------------------------------------- */
/* (tile, depth, id) order = what one stable sort of the instances by (tile << 32 | depth bits) gives when the instances
 * are generated in increasing Gaussian id.  Done as the GPU path does it, in two levels: a counting sort by tile (stable:
 * ids stay ascending inside a tile), then every tile's segment is sorted by (depth, id) on its own -- independent
 * segments, one OpenMP task each.  (A single merge sort over all R instances, as this file first had, copies the whole
 * array once per pass and leaves most cores idle: as a CPU baseline it was a straw man.) */
typedef struct { real depth; uint32_t id; } bin_item;

static int bin_cmp(const void* pa, const void* pb) {
    const bin_item *a = (const bin_item*)pa, *b = (const bin_item*)pb;
    if (a->depth < b->depth) return -1;
    if (a->depth > b->depth) return 1;
    return (a->id > b->id) - (a->id < b->id);
}

/* returns R = number of tile-splat instances; call with point_list==NULL to get R only. */
long gpo_bin(int N, int W, int H, const uint32_t* tiles_touched, const int32_t* rect, const real* depths,
             uint32_t* point_list /*R*/, uint32_t* point_tile /*R or NULL*/, int32_t* ranges /*T*2*/) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
    long Rn = 0;
    for (int i = 0; i < N; ++i) Rn += tiles_touched[i];
    if (!point_list) return Rn;
    long* start = (long*)calloc((size_t)T + 1, sizeof(long));
    for (int i = 0; i < N; ++i) {
        if (!tiles_touched[i]) continue;
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) ++start[y * gx + x + 1];
    }
    for (int t = 0; t < T; ++t) start[t + 1] += start[t];
    long* cursor = (long*)malloc((size_t)T * sizeof(long));
    memcpy(cursor, start, (size_t)T * sizeof(long));
    bin_item* items = (bin_item*)malloc((size_t)(Rn > 0 ? Rn : 1) * sizeof(bin_item));
    for (int i = 0; i < N; ++i) {
        if (!tiles_touched[i]) continue;
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) {
                bin_item* it = &items[cursor[y * gx + x]++];
                it->depth = depths[i];
                it->id = (uint32_t)i;
            }
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < T; ++t) {
        const long lo = start[t], hi = start[t + 1];
        if (hi - lo > 1) qsort(items + lo, (size_t)(hi - lo), sizeof(bin_item), bin_cmp);
        ranges[2 * t] = hi > lo ? (int32_t)lo : 0;
        ranges[2 * t + 1] = hi > lo ? (int32_t)hi : 0;
        for (long k = lo; k < hi; ++k) {
            point_list[k] = items[k].id;
            if (point_tile) point_tile[k] = (uint32_t)t;
        }
    }
    free(items); free(start); free(cursor);
    return Rn;
}

/* ------------------------------------------------------------------------------------------- */
/* 3. composite forward (per tile, per pixel, front to back)                                   */
/* ------------------------------------------------------------------------------------------- */
/* The composite's alpha = min(0.99, opacity * exp(power)), power = -0.5 (cx dx^2 + cz dy^2) - cy dx dy, in the form the HIP
 * kernels evaluate it since round 4 (csrc/raster_kernels.hip, preprocess_fwd_body / CF2_VISIT): the quadratic form is
 * pre-scaled by log2(e), the opacity enters as its base-2 logarithm, and alpha = min(0.99, exp2(e)) with
 *     e = dx (A' dx + B' dy) + ((C' dy) dy + lop),  A' = (-0.5 cx) log2e, B' = (-cy) log2e, C' = (-0.5 cz) log2e, lop = log2(opacity)
 * "power > 0" (never true for a positive-definite conic except by rounding) is the test e > lop.  Same expression tree,
 * operand for operand, in float32; exp2 / log2 are the only operations whose last bits differ (libm vs v_exp_f32 / ocml). */
#define GP_LOG2E R(1.4426950408889634)
static inline real gauss_exponent(const real* co /* conic x, y, z, opacity */, real dx, real dy, real* lop_out) {
    real A = (R(-0.5) * co[0]) * GP_LOG2E, B = (-co[1]) * GP_LOG2E, C = (R(-0.5) * co[2]) * GP_LOG2E;
    real lop = LOG2R(co[3]);
    *lop_out = lop;
    return FMA(dx, FMA(A, dx, B * dy), FMA(C * dy, dy, lop));
}

int gpo_composite_fwd(int W, int H, const int32_t* ranges, const uint32_t* point_list, const real* xy,
                      const real* rgb, const real* depths, const real* conic_opacity, const real* bg,
                      real* out_color /*3,H,W*/, real* out_depth /*H,W*/, int32_t* out_tidx /*H,W*/,
                      real* final_T /*H,W*/, int32_t* n_contrib /*H,W*/, uint8_t* ambiguous /*H,W or NULL*/) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
        int lo = ranges[2 * tile], hi = ranges[2 * tile + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; ++py)
            for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
                real T = R(1), C0 = 0, C1 = 0, C2 = 0, Dp = 0, best = 0, second = 0;
                int32_t best_id = -1;
                int contributor = 0, last = 0;
                uint8_t amb = 0;
                real pxf = (real)px, pyf = (real)py;
                for (int k = lo; k < hi; ++k) {
                    ++contributor;
                    uint32_t id = point_list[k];
                    real dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                    const real* co = conic_opacity + 4 * id;
                    real lop, e = gauss_exponent(co, dx, dy, &lop);
                    if (FABSR(e - lop) < R(2e-6)) amb |= 1;
                    if (e > lop) continue;
                    real alpha = FMINR(R(0.99), EXP2R(e));
                    if (FABSR(alpha * R(255) - R(1)) < R(2e-5)) amb |= 1;
                    if (alpha < R(1) / R(255)) continue;
                    real w = alpha * T;
                    real test_T = T - w;                       /* the published T * (1 - alpha), written as the kernel computes it */
                    if (FABSR(test_T * R(1e4) - R(1)) < R(1e-4)) amb |= 1;
                    if (test_T < R(0.0001)) break;
                    C0 = FMA(rgb[3 * id], w, C0);
                    C1 = FMA(rgb[3 * id + 1], w, C1);
                    C2 = FMA(rgb[3 * id + 2], w, C2);
                    Dp = FMA(depths[id], w, Dp);
                    if (w > best) { second = best; best = w; best_id = (int32_t)id; }
                    else if (w > second) second = w;
                    T = test_T;
                    last = contributor;
                }
                if (best > 0 && second > best * R(0.9999)) amb |= 2;
                size_t pix = (size_t)py * W + px;
                out_color[0 * (size_t)H * W + pix] = FMA(T, bg[0], C0);
                out_color[1 * (size_t)H * W + pix] = FMA(T, bg[1], C1);
                out_color[2 * (size_t)H * W + pix] = FMA(T, bg[2], C2);
                out_depth[pix] = Dp;
                out_tidx[pix] = best_id;
                final_T[pix] = T;
                n_contrib[pix] = last;
                if (ambiguous) ambiguous[pix] = amb;
            }
    }
    return 0;
}

/* 3b. The same loop in the PUBLISHED form -- power = -0.5 (cx dx^2 + cz dy^2) - cy dx dy, skip power > 0,
 *     alpha = min(0.99, opacity * exp(power)), skip alpha < 1/255, test_T = T * (1 - alpha), stop below 1e-4 -- with none of the
 *     kernels' algebra (no log2 folding, no T - alpha T).  It exists to keep gauss_exponent() honest: the folded expression was
 *     adopted so that the float32 oracle and the kernels make bit-identical discrete decisions, which makes the oracle follow
 *     the kernels' algebra; this entry point is the independent statement the folded one is compared with at full size
 *     (tests/test_oracle_published_form.py): same n_contrib and tidx on every pixel outside the ambiguity bands, same image.
 *     `ambiguous` uses the same relative bands around the same three thresholds (+ the tidx tie). */
int gpo_composite_fwd_published(int W, int H, const int32_t* ranges, const uint32_t* point_list, const real* xy,
                                const real* rgb, const real* depths, const real* conic_opacity, const real* bg,
                                real* out_color /*3,H,W*/, int32_t* out_tidx /*H,W*/, real* final_T /*H,W*/,
                                int32_t* n_contrib /*H,W*/, uint8_t* ambiguous /*H,W or NULL*/, double band_scale) {
    /* band_scale widens the ambiguity bands: 1 when both sides compute in one precision; the float64 run that is compared with a
     * float32 one uses 20 (a pixel coordinate near 1000 carries 6e-5 px of float32 rounding, i.e. ~1e-4 relative on alpha) */
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const real bs = (real)band_scale;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
        int lo = ranges[2 * tile], hi = ranges[2 * tile + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; ++py)
            for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
                real T = R(1), C[3] = {0, 0, 0}, best = 0, second = 0;
                int32_t best_id = -1;
                int contributor = 0, last = 0;
                uint8_t amb = 0;
                for (int k = lo; k < hi; ++k) {
                    ++contributor;
                    uint32_t id = point_list[k];
                    real dx = xy[2 * id] - (real)px, dy = xy[2 * id + 1] - (real)py;
                    const real* co = conic_opacity + 4 * id;
                    real power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (FABSR(power) < R(1.4e-6) * bs) amb |= 1;
                    if (power > R(0)) continue;
                    real alpha = FMINR(R(0.99), co[3] * EXPR(power));
                    if (FABSR(alpha * R(255) - R(1)) < R(2e-5) * bs) amb |= 1;
                    if (alpha < R(1) / R(255)) continue;
                    real test_T = T * (R(1) - alpha);
                    if (FABSR(test_T * R(1e4) - R(1)) < R(1e-4) * bs) amb |= 1;
                    if (test_T < R(0.0001)) break;
                    real w = alpha * T;
                    for (int c = 0; c < 3; ++c) C[c] += rgb[3 * id + c] * w;
                    if (w > best) { second = best; best = w; best_id = (int32_t)id; }
                    else if (w > second) second = w;
                    T = test_T;
                    last = contributor;
                }
                if (best > 0 && second > best * (R(1) - R(1e-4) * bs)) amb |= 2;
                size_t pix = (size_t)py * W + px;
                for (int c = 0; c < 3; ++c) out_color[c * (size_t)H * W + pix] = C[c] + T * bg[c];
                out_tidx[pix] = best_id;
                final_T[pix] = T;
                n_contrib[pix] = last;
                if (ambiguous) ambiguous[pix] = amb;
            }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* 4. composite backward (per pixel, BACK to front replay, upstream-style recursion)            */
/*    outputs (double accumulators, deterministic order inside a tile, atomics across tiles):   */
/*      dL_dmean2D [N,2]  in NDC units (x0.5W, x0.5H), as means2D.grad expects                  */
/*      dL_dconic  [N,3]  TRUE partials wrt conic (A=xx, B=xy, C=yy of the inverse covariance)  */
/*      dL_dopacity[N], dL_dcolor[N,3], dL_ddepth[N]                                            */
/* ------------------------------------------------------------------------------------------- */
int gpo_composite_bwd(int W, int H, int N, const int32_t* ranges, const uint32_t* point_list, const real* xy,
                      const real* rgb, const real* depths, const real* conic_opacity, const real* bg,
                      const real* final_T, const int32_t* n_contrib, const real* dL_dpix /*3,H,W*/,
                      const real* dL_dpixdepth /*H,W or NULL*/, double* dL_dmean2D, double* dL_dconic,
                      double* dL_dopacity, double* dL_dcolor, double* dL_ddepth) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    memset(dL_dmean2D, 0, sizeof(double) * 2 * (size_t)N);
    memset(dL_dconic, 0, sizeof(double) * 3 * (size_t)N);
    memset(dL_dopacity, 0, sizeof(double) * (size_t)N);
    memset(dL_dcolor, 0, sizeof(double) * 3 * (size_t)N);
    memset(dL_ddepth, 0, sizeof(double) * (size_t)N);
    const real ddelx_dx = R(0.5) * (real)W, ddely_dy = R(0.5) * (real)H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
        int lo = ranges[2 * tile];
        for (int py = ty0; py < ty0 + TILE && py < H; ++py)
            for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
                size_t pix = (size_t)py * W + px;
                real pxf = (real)px, pyf = (real)py;
                real T_final = final_T[pix];
                real T = T_final;
                int last = n_contrib[pix];
                real dLp[3] = {dL_dpix[pix], dL_dpix[(size_t)H * W + pix], dL_dpix[2 * (size_t)H * W + pix]};
                real dLd = dL_dpixdepth ? dL_dpixdepth[pix] : R(0);
                real accum[3] = {0, 0, 0}, accum_d = 0;
                real last_alpha = 0, last_col[3] = {0, 0, 0}, last_depth = 0;
                for (int k = lo + last - 1; k >= lo; --k) {
                    uint32_t id = point_list[k];
                    real dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                    const real* co = conic_opacity + 4 * id;
                    real lop, e = gauss_exponent(co, dx, dy, &lop);
                    if (e > lop) continue;
                    real E = EXP2R(e);                          /* = opacity * G, the unclamped alpha */
                    real G = E / co[3];
                    real alpha = FMINR(R(0.99), E);
                    if (alpha < R(1) / R(255)) continue;
                    T = T / (R(1) - alpha);
                    real w = alpha * T;
                    real dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ++ch) {
                        real c = rgb[3 * id + ch];
                        accum[ch] = last_alpha * last_col[ch] + (R(1) - last_alpha) * accum[ch];
                        last_col[ch] = c;
                        dL_dalpha += (c - accum[ch]) * dLp[ch];
#pragma omp atomic
                        dL_dcolor[3 * id + ch] += (double)(w * dLp[ch]);
                    }
                    {
                        real d = depths[id];
                        accum_d = last_alpha * last_depth + (R(1) - last_alpha) * accum_d;
                        last_depth = d;
                        dL_dalpha += (d - accum_d) * dLd;
#pragma omp atomic
                        dL_ddepth[id] += (double)(w * dLd);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    real bg_dot = bg[0] * dLp[0] + bg[1] * dLp[1] + bg[2] * dLp[2];
                    dL_dalpha += (-T_final / (R(1) - alpha)) * bg_dot;
                    real dL_dG = co[3] * dL_dalpha;
                    real gdx = G * dx, gdy = G * dy;
                    real dG_ddelx = -gdx * co[0] - gdy * co[1];
                    real dG_ddely = -gdy * co[2] - gdx * co[1];
#pragma omp atomic
                    dL_dmean2D[2 * id] += (double)(dL_dG * dG_ddelx * ddelx_dx);
#pragma omp atomic
                    dL_dmean2D[2 * id + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
#pragma omp atomic
                    dL_dconic[3 * id] += (double)(R(-0.5) * gdx * dx * dL_dG);
#pragma omp atomic
                    dL_dconic[3 * id + 1] += (double)(-gdx * dy * dL_dG);
#pragma omp atomic
                    dL_dconic[3 * id + 2] += (double)(R(-0.5) * gdy * dy * dL_dG);
#pragma omp atomic
                    dL_dopacity[id] += (double)(G * dL_dalpha);
                }
            }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* 5. preprocess backward (per Gaussian)                                                        */
/* ------------------------------------------------------------------------------------------- */
int gpo_preprocess_bwd(int N, int D, int M, const real* means3D, const real* scales, real scale_modifier,
                       const real* rotations, const real* shs, int have_colors_precomp, const real* cov3D_precomp,
                       const real* view, const real* proj, const real* campos, int W, int H, real tanfovx,
                       real tanfovy, const int32_t* radii, const real* cov3D, const uint8_t* clamped,
                       const double* dL_dmean2D, const double* dL_dconic, const double* dL_dcolor,
                       const double* dL_ddepth,
                       real* dL_dmeans3D /*N,3*/, real* dL_dshs /*N,M,3*/, real* dL_dcolors_precomp /*N,3*/,
                       real* dL_dscales /*N,3*/, real* dL_drots /*N,4*/, real* dL_dcov3D /*N,6*/) {
    const real fx = (real)W / (R(2) * tanfovx), fy = (real)H / (R(2) * tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) { dL_dmeans3D[3 * i + k] = 0; dL_dscales[3 * i + k] = 0; dL_dcolors_precomp[3 * i + k] = 0; }
        for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0;
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0;
        for (int k = 0; k < M * 3; ++k) dL_dshs[(size_t)i * M * 3 + k] = 0;
        if (!(radii[i] > 0)) continue;
        real px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        real pv[3];
        xform4x3(view, px, py, pz, pv);
        const real* c6 = cov3D + 6 * i;
        real abc[3];
        proj_ctx cx;
        compute_cov2D(pv, fx, fy, tanfovx, tanfovy, c6, view, abc, &cx);
        real a = abc[0] + R(0.3), b = abc[1], c = abc[2] + R(0.3);
        real det = a * c - b * b;
        real gA = (real)dL_dconic[3 * i], gB = (real)dL_dconic[3 * i + 1], gC = (real)dL_dconic[3 * i + 2];
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real gm[3] = {0, 0, 0}; /* dL/dmean3D accumulation */
        if (det != R(0)) {
            real d2 = R(1) / (det * det);
            dL_da = d2 * (-c * c * gA + b * c * gB - b * b * gC);
            dL_db = d2 * (R(2) * b * c * gA - (a * c + b * b) * gB + R(2) * a * b * gC);
            dL_dc = d2 * (-b * b * gA + a * b * gB - a * a * gC);
            const real* T0 = cx.T0; const real* T1 = cx.T1;
            /* packed cov3D gradient: diag = dL_da T0i^2 + dL_db T0i T1i + dL_dc T1i^2,
             * offdiag(i<j) = 2 dL_da T0i T0j + dL_db (T0i T1j + T0j T1i) + 2 dL_dc T1i T1j */
            real g6[6];
            g6[0] = dL_da * T0[0] * T0[0] + dL_db * T0[0] * T1[0] + dL_dc * T1[0] * T1[0];
            g6[3] = dL_da * T0[1] * T0[1] + dL_db * T0[1] * T1[1] + dL_dc * T1[1] * T1[1];
            g6[5] = dL_da * T0[2] * T0[2] + dL_db * T0[2] * T1[2] + dL_dc * T1[2] * T1[2];
            g6[1] = R(2) * dL_da * T0[0] * T0[1] + dL_db * (T0[0] * T1[1] + T0[1] * T1[0]) + R(2) * dL_dc * T1[0] * T1[1];
            g6[2] = R(2) * dL_da * T0[0] * T0[2] + dL_db * (T0[0] * T1[2] + T0[2] * T1[0]) + R(2) * dL_dc * T1[0] * T1[2];
            g6[4] = R(2) * dL_da * T0[1] * T0[2] + dL_db * (T0[1] * T1[2] + T0[2] * T1[1]) + R(2) * dL_dc * T1[1] * T1[2];
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = g6[k];
            /* dL/dT rows */
            real S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
            real ST0[3], ST1[3];
            for (int r = 0; r < 3; ++r) {
                ST0[r] = S[3 * r] * T0[0] + S[3 * r + 1] * T0[1] + S[3 * r + 2] * T0[2];
                ST1[r] = S[3 * r] * T1[0] + S[3 * r + 1] * T1[1] + S[3 * r + 2] * T1[2];
            }
            real dT0[3], dT1[3];
            for (int r = 0; r < 3; ++r) {
                dT0[r] = R(2) * dL_da * ST0[r] + dL_db * ST1[r];
                dT1[r] = R(2) * dL_dc * ST1[r] + dL_db * ST0[r];
            }
            real W0[3] = {view[0], view[4], view[8]};
            real W1[3] = {view[1], view[5], view[9]};
            real W2[3] = {view[2], view[6], view[10]};
            real dJ00 = dT0[0] * W0[0] + dT0[1] * W0[1] + dT0[2] * W0[2];
            real dJ02 = dT0[0] * W2[0] + dT0[1] * W2[1] + dT0[2] * W2[2];
            real dJ11 = dT1[0] * W1[0] + dT1[1] * W1[1] + dT1[2] * W1[2];
            real dJ12 = dT1[0] * W2[0] + dT1[1] * W2[1] + dT1[2] * W2[2];
            real tz = cx.tz, itz = R(1) / tz, itz2 = itz * itz, itz3 = itz2 * itz;
            real dtx = cx.gx * (-fx * itz2) * dJ02;
            real dty = cx.gy * (-fy * itz2) * dJ12;
            real dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + (R(2) * fx * cx.tx) * itz3 * dJ02 +
                       (R(2) * fy * cx.ty) * itz3 * dJ12;
            /* dL/dmean = W^T dt */
            gm[0] += W0[0] * dtx + W1[0] * dty + W2[0] * dtz;
            gm[1] += W0[1] * dtx + W1[1] * dty + W2[1] * dtz;
            gm[2] += W0[2] * dtx + W1[2] * dty + W2[2] * dtz;
        }
        /* depth: z_view = row 2 of V */
        {
            real gd = (real)dL_ddepth[i];
            gm[0] += view[2] * gd; gm[1] += view[6] * gd; gm[2] += view[10] * gd;
        }
        /* mean2D (NDC) -> mean3D */
        {
            real ph[4];
            xform4x4(proj, px, py, pz, ph);
            real mw = R(1) / (ph[3] + R(0.0000001));
            real mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
            real g2x = (real)dL_dmean2D[2 * i], g2y = (real)dL_dmean2D[2 * i + 1];
            gm[0] += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
            gm[1] += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
            gm[2] += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
        }
        /* colour */
        if (have_colors_precomp) {
            for (int k = 0; k < 3; ++k) dL_dcolors_precomp[3 * i + k] = (real)dL_dcolor[3 * i + k];
        } else {
            real dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
            real len = SQRT(FMA(dx, dx, FMA(dy, dy, dz * dz)));
            real inv = R(1) / len;
            real x = dx * inv, y = dy * inv, z = dz * inv;
            const real* sh = shs + (size_t)i * M * 3;
            real* dsh = dL_dshs + (size_t)i * M * 3;
            real dRGB[3];
            for (int ch = 0; ch < 3; ++ch) dRGB[ch] = clamped[3 * i + ch] ? R(0) : (real)dL_dcolor[3 * i + ch];
            real ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch) {
                real g = dRGB[ch];
                dsh[0 * 3 + ch] = SH_C0 * g;
                if (D > 0) {
                    dsh[1 * 3 + ch] = -SH_C1 * y * g;
                    dsh[2 * 3 + ch] = SH_C1 * z * g;
                    dsh[3 * 3 + ch] = -SH_C1 * x * g;
                    real ddx = -SH_C1 * sh[3 * 3 + ch];
                    real ddy = -SH_C1 * sh[1 * 3 + ch];
                    real ddz = SH_C1 * sh[2 * 3 + ch];
                    if (D > 1) {
                        real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        dsh[4 * 3 + ch] = SH_C2[0] * xy * g;
                        dsh[5 * 3 + ch] = SH_C2[1] * yz * g;
                        dsh[6 * 3 + ch] = SH_C2[2] * (R(2) * zz - xx - yy) * g;
                        dsh[7 * 3 + ch] = SH_C2[3] * xz * g;
                        dsh[8 * 3 + ch] = SH_C2[4] * (xx - yy) * g;
                        ddx += SH_C2[0] * y * sh[4 * 3 + ch] + SH_C2[2] * R(2) * -x * sh[6 * 3 + ch] +
                               SH_C2[3] * z * sh[7 * 3 + ch] + SH_C2[4] * R(2) * x * sh[8 * 3 + ch];
                        ddy += SH_C2[0] * x * sh[4 * 3 + ch] + SH_C2[1] * z * sh[5 * 3 + ch] +
                               SH_C2[2] * R(2) * -y * sh[6 * 3 + ch] + SH_C2[4] * R(2) * -y * sh[8 * 3 + ch];
                        ddz += SH_C2[1] * y * sh[5 * 3 + ch] + SH_C2[2] * R(4) * z * sh[6 * 3 + ch] +
                               SH_C2[3] * x * sh[7 * 3 + ch];
                        if (D > 2) {
                            dsh[9 * 3 + ch] = SH_C3[0] * y * (R(3) * xx - yy) * g;
                            dsh[10 * 3 + ch] = SH_C3[1] * xy * z * g;
                            dsh[11 * 3 + ch] = SH_C3[2] * y * (R(4) * zz - xx - yy) * g;
                            dsh[12 * 3 + ch] = SH_C3[3] * z * (R(2) * zz - R(3) * xx - R(3) * yy) * g;
                            dsh[13 * 3 + ch] = SH_C3[4] * x * (R(4) * zz - xx - yy) * g;
                            dsh[14 * 3 + ch] = SH_C3[5] * z * (xx - yy) * g;
                            dsh[15 * 3 + ch] = SH_C3[6] * x * (xx - R(3) * yy) * g;
                            ddx += SH_C3[0] * sh[9 * 3 + ch] * R(6) * xy + SH_C3[1] * sh[10 * 3 + ch] * yz +
                                   SH_C3[2] * sh[11 * 3 + ch] * R(-2) * xy + SH_C3[3] * sh[12 * 3 + ch] * R(-6) * xz +
                                   SH_C3[4] * sh[13 * 3 + ch] * (R(4) * zz - R(3) * xx - yy) +
                                   SH_C3[5] * sh[14 * 3 + ch] * R(2) * xz + SH_C3[6] * sh[15 * 3 + ch] * R(3) * (xx - yy);
                            ddy += SH_C3[0] * sh[9 * 3 + ch] * R(3) * (xx - yy) + SH_C3[1] * sh[10 * 3 + ch] * xz +
                                   SH_C3[2] * sh[11 * 3 + ch] * (R(4) * zz - xx - R(3) * yy) +
                                   SH_C3[3] * sh[12 * 3 + ch] * R(-6) * yz + SH_C3[4] * sh[13 * 3 + ch] * R(-2) * xy +
                                   SH_C3[5] * sh[14 * 3 + ch] * R(-2) * yz + SH_C3[6] * sh[15 * 3 + ch] * R(-6) * xy;
                            ddz += SH_C3[1] * sh[10 * 3 + ch] * xy + SH_C3[2] * sh[11 * 3 + ch] * R(8) * yz +
                                   SH_C3[3] * sh[12 * 3 + ch] * R(3) * (R(2) * zz - xx - yy) +
                                   SH_C3[4] * sh[13 * 3 + ch] * R(8) * xz + SH_C3[5] * sh[14 * 3 + ch] * (xx - yy);
                        }
                    }
                    ddir[0] += ddx * g; ddir[1] += ddy * g; ddir[2] += ddz * g;
                }
            }
            /* through the normalisation dir = v/|v| */
            real dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
            gm[0] += (ddir[0] - x * dot) * inv;
            gm[1] += (ddir[1] - y * dot) * inv;
            gm[2] += (ddir[2] - z * dot) * inv;
        }
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = gm[k];
        /* cov3D -> scale, rotation */
        if (!cov3D_precomp) {
            const real* g6 = dL_dcov3D + 6 * i;
            real Gs[9] = {g6[0], R(0.5) * g6[1], R(0.5) * g6[2], R(0.5) * g6[1], g6[3], R(0.5) * g6[4],
                          R(0.5) * g6[2], R(0.5) * g6[4], g6[5]};
            real Rm[9];
            const real* q = rotations + 4 * i;
            quat_to_R(q, Rm);
            real s[3] = {scale_modifier * scales[3 * i], scale_modifier * scales[3 * i + 1],
                         scale_modifier * scales[3 * i + 2]};
            real L[9], dL[9];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) L[3 * r + k] = Rm[3 * r + k] * s[k];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k)
                    dL[3 * r + k] = R(2) * (Gs[3 * r] * L[k] + Gs[3 * r + 1] * L[3 + k] + Gs[3 * r + 2] * L[6 + k]);
            real dR[9];
            for (int k = 0; k < 3; ++k) {
                real acc = 0;
                for (int r = 0; r < 3; ++r) { acc += dL[3 * r + k] * Rm[3 * r + k]; dR[3 * r + k] = dL[3 * r + k] * s[k]; }
                dL_dscales[3 * i + k] = acc * scale_modifier;
            }
            real r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drots[4 * i + 0] = R(2) * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            dL_drots[4 * i + 1] = R(2) * (y * dR[1] + z * dR[2] + y * dR[3] - R(2) * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - R(2) * x * dR[8]);
            dL_drots[4 * i + 2] = R(2) * (-R(2) * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - R(2) * y * dR[8]);
            dL_drots[4 * i + 3] = R(2) * (-R(2) * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - R(2) * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        }
    }
    return 0;
}

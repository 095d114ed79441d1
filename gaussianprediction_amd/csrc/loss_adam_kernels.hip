// loss_adam_kernels.hip -- the two steps that follow the render in every training iteration
// (SURVEY.md section 8f rows 2 and 3), fused for gfx950:
//   * L1 + SSIM(11x11 Gaussian window, sigma 1.5, zero padding) loss, forward and backward
//     [REF utils/loss_utils.py:54-100, train.py:105-108].  One workgroup = one 32x32 tile of one
//     channel: halo staged in LDS once, separable 11-tap horizontal then vertical pass, all five
//     (forward) / three (backward) filtered maps produced from that one staging -- instead of five
//     depthwise conv2d launches each round-tripping [3,H,W] through HBM.
//   * Adam step [REF scene/gaussian_model.py:472 (eps=1e-15), train.py:196-197] that also zeroes the
//     gradient it consumed (the flat gradient bucket is reused by the next step).
#include "gp_common.h"
#include "loss_adam_kernels.h"

#define LT 32              // output tile edge
#define LH 5               // window half width
#define LE (LT + 2 * LH)   // staged tile edge (42)
#define LP (LE + 1)        // padded LDS row
#define LHP (LT + 1)       // padded row of the horizontally filtered maps

struct Win11 { float w[11]; };

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// forward: sums[0] += sum |a-b| ; sums[1] += sum ssim_map ; optional derivative maps
// dmap[0] = d ssim / d conv(a), dmap[1] = d ssim / d conv(a^2), dmap[2] = d ssim / d conv(a b)
__global__ __launch_bounds__(256) void gp_l1_ssim_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                            int H, int W, Win11 win, double* __restrict__ sums,
                                                            float* __restrict__ dmap) {
    __shared__ float s_a[LE][LP], s_b[LE][LP];
    __shared__ float s_h[4][LE][LHP];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT, ch = blockIdx.z;
    const size_t HW = (size_t)H * W;
    const float* a_img = img + ch * HW;
    const float* b_img = gt + ch * HW;
    {   // halo staging: ALL of a thread's loads issued before the first LDS store (clamped addresses, selected afterwards).  As a
        // rolled loop with conditional loads this was load -> wait -> store, 14 dependent round trips to memory per workgroup:
        // the kernel's whole duration (0.067 ms; the filter arithmetic is 0.02 ms)
        constexpr int NST = (LE * LE + 255) / 256;
        float va[NST], vb[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u, y = i / LE, x = i - y * LE;
            const int gy = ty0 - LH + y, gx = tx0 - LH + x;
            const bool in = i < LE * LE && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = in ? (size_t)gy * W + gx : 0;
            va[u] = a_img[o]; vb[u] = b_img[o];
        }
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u, y = i / LE, x = i - y * LE;
            const int gy = ty0 - LH + y, gx = tx0 - LH + x;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (i < LE * LE) { s_a[y][x] = in ? va[u] : 0.f; s_b[y][x] = in ? vb[u] : 0.f; }
        }
    }
    __syncthreads();
    // horizontal pass: a thread owns 8 consecutive outputs of one staged row -- 18 + 18 LDS reads feed 8 x 5 outputs
    // (a sliding window in registers instead of 22 reads per output).  Row strides 43 / 33 keep every access conflict-free.
    if (tid < LE * 4) {
        const int y = tid >> 2, x0 = (tid & 3) * 8;
        // FOUR filtered maps, not five (round 6): SSIM reads sigma_a^2 + sigma_b^2 only as a sum [REF utils/loss_utils.py:94-98], and the
        // window is linear -- conv(a^2) + conv(b^2) = conv(a^2 + b^2) up to rounding (~1e-7 of the sum; the loss golden vectors hold it)
        float a[18], b[18], s2[18], ab[18];
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            a[j] = s_a[y][x0 + j]; b[j] = s_b[y][x0 + j];
            s2[j] = a[j] * a[j] + b[j] * b[j]; ab[j] = a[j] * b[j];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float m1 = 0.f, m2 = 0.f, ss2 = 0.f, sab = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = win.w[k];
                m1 = fmaf(w, a[e + k], m1); m2 = fmaf(w, b[e + k], m2);
                ss2 = fmaf(w, s2[e + k], ss2); sab = fmaf(w, ab[e + k], sab);
            }
            s_h[0][y][x0 + e] = m1; s_h[1][y][x0 + e] = m2; s_h[2][y][x0 + e] = ss2; s_h[3][y][x0 + e] = sab;
        }
    }
    __syncthreads();
    // vertical pass: a thread owns 4 consecutive outputs of one column (14 reads per map feed 4 outputs)
    float l1 = 0.f, ss = 0.f;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    {
        const int x = tid & 31, y0 = (tid >> 5) * 4;
        float o[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[14];
#pragma unroll
            for (int j = 0; j < 14; ++j) v[j] = s_h[q][y0 + j][x];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc = fmaf(win.w[k], v[e + k], acc);
                o[q][e] = acc;
            }
        }
        const int gx = tx0 + x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int y = y0 + e, gy = ty0 + y;
            if (gy >= H || gx >= W) continue;
            const float mu1 = o[0][e], mu2 = o[1][e], s2f = o[2][e], ab = o[3][e];
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float s12 = ab - mu12;
            const float N1 = 2.f * mu12 + C1, N2 = 2.f * s12 + C2, D1 = mu1s + mu2s + C1, D2 = ((s2f - mu1s) - mu2s) + C2;
            const float inv = 1.f / (D1 * D2);
            const float ssim = N1 * N2 * inv;
            ss += ssim;
            l1 += fabsf(s_a[y + LH][x + LH] - s_b[y + LH][x + LH]);
            if (dmap) {
                const size_t oo = ch * HW + (size_t)gy * W + gx;
                // d/dmu1 (total, through s11 = aa - mu1^2 and s12 = ab - mu1 mu2)
                const float dmu1 = (2.f * mu2 * N2 - 2.f * mu2 * N1) * inv - ssim * (2.f * mu1 * D2 - 2.f * mu1 * D1) * inv;
                dmap[oo] = dmu1;
                dmap[3 * HW + oo] = -ssim / D2;        // d/d conv(a^2)
                dmap[6 * HW + oo] = 2.f * N1 * inv;    // d/d conv(a b)
            }
        }
    }
    const float l1_tot = block_sum_256(l1, s_red);
    __syncthreads();
    const float ss_tot = block_sum_256(ss, s_red);
    if (tid == 0) {      // one slot per workgroup, plain stores: nothing to zero beforehand, and the finalize sums in a fixed order
        const unsigned slot = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        sums[2 * slot] = (double)l1_tot;
        sums[2 * slot + 1] = (double)ss_tot;
    }
}

// backward: dimg = g * [ (1-lam)/n sign(a-b) - lam/n ( conv(dA) + 2 a conv(dB) + b conv(dC) ) ]
__global__ __launch_bounds__(256) void gp_l1_ssim_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                            const float* __restrict__ dmap, int H, int W, Win11 win,
                                                            float lambda, const float* __restrict__ upstream,
                                                            float* __restrict__ dimg, const float* __restrict__ reg_x, long reg_n,
                                                            float reg_scale_over_n, float* __restrict__ reg_g) {
    __shared__ float s_m[3][LE][LP];
    __shared__ float s_h[3][LE][LHP];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT, ch = blockIdx.z;
    if (reg_g && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        // gradient of the regulariser scale * mean|x| riding along (a few thousand elements: not worth a launch of its own)
        const float c = (upstream ? upstream[0] : 1.f) * reg_scale_over_n;
        for (long i = tid; i < reg_n; i += 256) { const float v = reg_x[i]; reg_g[i] = v > 0.f ? c : (v < 0.f ? -c : 0.f); }
    }
    const size_t HW = (size_t)H * W;
    // the output pixels' own (img, gt) values: fetched up front (clamped addresses), consumed after the two filter passes --
    // as conditional loads inside the output loop they were four more dependent round trips per workgroup
    float pa[4], pb[4];
    {
        const int x = tid & 31, y0 = (tid >> 5) * 4, gx = min(tx0 + x, W - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t oo = ch * HW + (size_t)min(ty0 + y0 + e, H - 1) * W + gx;
            pa[e] = img[oo]; pb[e] = gt[oo];
        }
    }
    {   // halo staging with every load in flight before the first LDS store (see the forward kernel)
        constexpr int NST = (LE * LE + 255) / 256;
        float v0[NST], v1[NST], v2[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u, y = i / LE, x = i - y * LE;
            const int gy = ty0 - LH + y, gx = tx0 - LH + x;
            const bool in = i < LE * LE && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = ch * HW + (in ? (size_t)gy * W + gx : 0);
            v0[u] = dmap[o]; v1[u] = dmap[3 * HW + o]; v2[u] = dmap[6 * HW + o];
        }
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u, y = i / LE, x = i - y * LE;
            const int gy = ty0 - LH + y, gx = tx0 - LH + x;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (i < LE * LE) { s_m[0][y][x] = in ? v0[u] : 0.f; s_m[1][y][x] = in ? v1[u] : 0.f; s_m[2][y][x] = in ? v2[u] : 0.f; }
        }
    }
    __syncthreads();
    if (tid < LE * 4) {   // horizontal pass, 8 outputs per thread (see the forward kernel)
        const int y = tid >> 2, x0 = (tid & 3) * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v[18];
#pragma unroll
            for (int j = 0; j < 18; ++j) v[j] = s_m[q][y][x0 + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc = fmaf(win.w[k], v[e + k], acc);
                s_h[q][y][x0 + e] = acc;
            }
        }
    }
    __syncthreads();
    const float g = upstream ? upstream[0] : 1.f;
    const float inv_n = 1.f / (3.f * (float)HW);
    {   // vertical pass, 4 outputs per thread
        const int x = tid & 31, y0 = (tid >> 5) * 4;
        float o[3][4];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v[14];
#pragma unroll
            for (int j = 0; j < 14; ++j) v[j] = s_h[q][y0 + j][x];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc = fmaf(win.w[k], v[e + k], acc);
                o[q][e] = acc;
            }
        }
        const int gx = tx0 + x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gy = ty0 + y0 + e;
            if (gy >= H || gx >= W) continue;
            const size_t oo = ch * HW + (size_t)gy * W + gx;
            const float a = pa[e], b = pb[e];
            const float d = a - b;
            const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            dimg[oo] = g * inv_n * ((1.f - lambda) * sgn - lambda * (o[0][e] + 2.f * a * o[1][e] + b * o[2][e]));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Loss forward AND image gradient in one kernel (round 6; the fused train step's form -- the two kernels above stay the autograd
// path's).  d loss / d img does not depend on the loss value, only on the SSIM derivative maps within 5 pixels of the output
// pixel: a workgroup owning 32 x 32 outputs therefore needs the maps on 42 x 42, i.e. the statistics on 42 x 42, i.e. the two
// images on 52 x 52 -- staged ONCE; the three derivative maps never leave the CU (the pair of kernels wrote 9 floats per pixel and
// read them back with a halo: ~370 MB of traffic at 1352 x 1014 for 16 MB of images).  Same filter chains in the same order: the
// gradient and the per-tile sums are bit-identical to gp_l1_ssim_fwd_kernel + gp_l1_ssim_bwd_kernel.
// LDS: the image halos (22 KB; the derivative maps take their place once the last horizontal pass has read them) + two horizontally
// filtered planes at a time (18 KB; later the backward's three) = 40 KB, three to four workgroups per CU.
// ------------------------------------------------------------------------------------------------
#define LF (LT + 4 * LH)    // staged image edge (52)
#define LFP (LF + 1)
__global__ __launch_bounds__(256) void gp_l1_ssim_fused_kernel(const float* __restrict__ img, const float* __restrict__ gt, int H, int W,
                                                              Win11 win, float lambda, const float* __restrict__ upstream,
                                                              double* __restrict__ sums, float* __restrict__ dimg,
                                                              const float* __restrict__ reg_x, long reg_n, float reg_scale_over_n,
                                                              float* __restrict__ reg_g) {
    // The two maps of a filter round live INTERLEAVED in LDS (one 8-byte read = the operand pair of a packed operation; as two planes
    // every pair cost two reads and two register moves: a quarter of the kernel's instructions).
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ f2 s_ab[LF][LFP];                 // (img, gt) halos, 22 KB; later the derivative maps: pairs [42][43] + single [42][43]
    __shared__ f2 s_h[LF][LP];                   // the horizontally filtered pair, 18 KB; later the backward's pair [42][33] + single [42][33]
    __shared__ float s_red[4];
    f2 (*s_m01)[LP] = (f2 (*)[LP]) & s_ab[0][0];
    float (*s_m2)[LP] = (float (*)[LP])((float*)&s_ab[0][0] + 2 * LE * LP);
    f2 (*s_g01)[LHP] = (f2 (*)[LHP]) & s_h[0][0];
    float (*s_g2)[LHP] = (float (*)[LHP])((float*)&s_h[0][0] + 2 * LE * LHP);
    static_assert(3 * LE * LP <= 2 * LF * LFP && 3 * LE * LHP <= 2 * LF * LP, "aliased planes must fit");
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT, ch = blockIdx.z;
    if (reg_g && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {   // the regulariser's gradient rides along (see the backward kernel)
        const float c = (upstream ? upstream[0] : 1.f) * reg_scale_over_n;
        for (long i = tid; i < reg_n; i += 256) { const float v = reg_x[i]; reg_g[i] = v > 0.f ? c : (v < 0.f ? -c : 0.f); }
    }
    const size_t HW = (size_t)H * W;
    const float* a_img = img + ch * HW;
    const float* b_img = gt + ch * HW;
    {   // halo staging, every load in flight before the first LDS store
        constexpr int NST = (LF * LF + 255) / 256;
        float va[NST], vb[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u, y = i / LF, x = i - y * LF;
            const int gy = ty0 - 2 * LH + y, gx = tx0 - 2 * LH + x;
            const bool in = i < LF * LF && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = in ? (size_t)gy * W + gx : 0;
            va[u] = a_img[o]; vb[u] = b_img[o];
        }
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u, y = i / LF, x = i - y * LF;
            const int gy = ty0 - 2 * LH + y, gx = tx0 - 2 * LH + x;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (i < LF * LF) s_ab[y][x] = in ? f2{va[u], vb[u]} : f2{0.f, 0.f};
        }
    }
    __syncthreads();
    // this thread's four output pixels (column ox, rows oy0 .. oy0 + 3): their own values, kept for the L1 term and the last step
    const int ox = tid & 31, oy0 = (tid >> 5) * 4;
    f2 pab[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pab[e] = s_ab[oy0 + e + 2 * LH][ox + 2 * LH];
    // ---- statistics on 42 x 42: FOUR maps in two rounds -- (a, b), then (a^2 + b^2, a b), each pair the halves of packed fp32 operations
    // (v_pk_fma_f32: one instruction per tap and output for BOTH maps).  Round 6: SSIM reads sigma_a^2 + sigma_b^2 only as a sum [REF
    // utils/loss_utils.py:94-98] and the window is linear, so conv(a^2 + b^2) replaces conv(a^2) + conv(b^2) (rounding aside) -- a
    // fifth of the statistics' filter work; gp_l1_ssim_fwd_kernel does the same, bit for bit.  Horizontal: an item = 11 consecutive
    // outputs of one staged row (52 rows x 4 groups = 208 items: one pass over the threads); vertical: a thread = 7 consecutive rows of
    // one column (42 columns x 6 groups = 252 threads).
    const int vx = tid % LE, vy0 = (tid / LE) * 7;
    const bool vert = tid < LE * 6;
    float st[4][7];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (tid < LF * 4) {
            const int y = tid >> 2, x0 = (tid & 3) * 11;
            f2 v[21];           // (columns beyond the staged 52 feed outputs beyond 42 only: discarded)
#pragma unroll
            for (int j = 0; j < 21; ++j) v[j] = s_ab[y][x0 + j];
            if (round == 1) {
#pragma unroll
                for (int j = 0; j < 21; ++j) {
                    const f2 sq = v[j] * v[j];
                    v[j] = f2{sq[0] + sq[1], v[j][0] * v[j][1]};
                }
            }
#pragma unroll
            for (int e = 0; e < 11; ++e) {
                f2 m = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 11; ++k) m = __builtin_elementwise_fma(f2{win.w[k], win.w[k]}, v[e + k], m);
                if (x0 + e < LE) s_h[y][x0 + e] = m;
            }
        }
        __syncthreads();
        if (vert) {
            f2 v[17];
#pragma unroll
            for (int j = 0; j < 17; ++j) v[j] = s_h[vy0 + j][vx];
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                f2 acc = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 11; ++k) acc = __builtin_elementwise_fma(f2{win.w[k], win.w[k]}, v[e + k], acc);
                st[2 * round][e] = acc[0];
                st[2 * round + 1][e] = acc[1];
            }
        }
        if (round < 1) __syncthreads();         // (after the last round nobody reads the image planes again: the maps may overwrite them)
    }
    // ---- SSIM and its three derivative maps on 42 x 42 (zero outside the image, as the backward's staging had them)
    float l1 = 0.f, ss = 0.f;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    if (vert) {
        const int gx = tx0 - LH + vx;
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            const int y = vy0 + e, gy = ty0 - LH + y;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const float mu1 = st[0][e], mu2 = st[1][e], s2f = st[2][e], ab = st[3][e];
                const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
                const float s12 = ab - mu12;
                const float N1 = 2.f * mu12 + C1, N2 = 2.f * s12 + C2, D1 = mu1s + mu2s + C1, D2 = ((s2f - mu1s) - mu2s) + C2;
                const float inv = 1.f / (D1 * D2);
                const float ssim = N1 * N2 * inv;
                if (vx >= LH && vx < LH + LT && y >= LH && y < LH + LT) ss += ssim;        // this workgroup's own 32 x 32
                d0 = (2.f * mu2 * N2 - 2.f * mu2 * N1) * inv - ssim * (2.f * mu1 * D2 - 2.f * mu1 * D1) * inv;
                d1 = -ssim / D2;
                d2 = 2.f * N1 * inv;
            }
            s_m01[y][vx] = f2{d0, d1};
            s_m2[y][vx] = d2;
        }
    }
    __syncthreads();
    // ---- the gradient's two filter passes (as gp_l1_ssim_bwd_kernel; maps 0 and 1 as a packed pair)
    if (tid < LE * 4) {
        const int y = tid >> 2, x0 = (tid & 3) * 8;
        f2 v[18];
        float v2[18];
#pragma unroll
        for (int j = 0; j < 18; ++j) { v[j] = s_m01[y][x0 + j]; v2[j] = s_m2[y][x0 + j]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f2 acc = {0.f, 0.f};
            float acc2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) { acc = __builtin_elementwise_fma(f2{win.w[k], win.w[k]}, v[e + k], acc); acc2 = fmaf(win.w[k], v2[e + k], acc2); }
            s_g01[y][x0 + e] = acc;
            s_g2[y][x0 + e] = acc2;
        }
    }
    __syncthreads();
    const float g = upstream ? upstream[0] : 1.f;
    const float inv_n = 1.f / (3.f * (float)HW);
    {
        float o[3][4];
        {
            f2 v[14];
            float v2[14];
#pragma unroll
            for (int j = 0; j < 14; ++j) { v[j] = s_g01[oy0 + j][ox]; v2[j] = s_g2[oy0 + j][ox]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f2 acc = {0.f, 0.f};
                float acc2 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) { acc = __builtin_elementwise_fma(f2{win.w[k], win.w[k]}, v[e + k], acc); acc2 = fmaf(win.w[k], v2[e + k], acc2); }
                o[0][e] = acc[0]; o[1][e] = acc[1]; o[2][e] = acc2;
            }
        }
        const int gx = tx0 + ox;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gy = ty0 + oy0 + e;
            if (gy >= H || gx >= W) continue;
            const float a = pab[e][0], b = pab[e][1];
            const float d = a - b;
            l1 += fabsf(d);
            const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            dimg[ch * HW + (size_t)gy * W + gx] = g * inv_n * ((1.f - lambda) * sgn - lambda * (o[0][e] + 2.f * a * o[1][e] + b * o[2][e]));
        }
    }
    const float l1_tot = block_sum_256(l1, s_red);
    __syncthreads();
    const float ss_tot = block_sum_256(ss, s_red);
    if (tid == 0) {
        const unsigned slot = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        sums[2 * slot] = (double)l1_tot;
        sums[2 * slot + 1] = (double)ss_tot;
    }
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, amsgrad=False, weight_decay=0) + gradient zeroing
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                     float bc1, float bc2_sqrt, int zero_grad) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 pv = *(float4*)(p + i), gv = *(float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
        float* pp = (float*)&pv; float* gg = (float*)&gv; float* mm = (float*)&mv; float* vq = (float*)&vv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
            vq[k] = b2 * vq[k] + (1.f - b2) * gg[k] * gg[k];
            const float denom = sqrtf(vq[k]) / bc2_sqrt + eps;
            pp[k] -= (lr / bc1) * (mm[k] / denom);
        }
        *(float4*)(p + i) = pv; *(float4*)(m + i) = mv; *(float4*)(v + i) = vv;
        if (zero_grad) *(float4*)(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (; i < n; ++i) {
            const float gk = g[i];
            const float mk = b1 * m[i] + (1.f - b1) * gk;
            const float vk = b2 * v[i] + (1.f - b2) * gk * gk;
            const float denom = sqrtf(vk) / bc2_sqrt + eps;
            p[i] -= (lr / bc1) * (mk / denom);
            m[i] = mk; v[i] = vk;
            if (zero_grad) g[i] = 0.f;
        }
    }
}

// multi-tensor form: ONE launch walks every parameter tensor (chunk table built by the host; table and chunk body: loss_adam_kernels.h)
__global__ __launch_bounds__(256) void gp_adam_multi_kernel(AdamTable t, float b1, float b2, float eps, int zero_grad,
                                                           const uint32_t* __restrict__ skip_flag) {
    adam_chunk_body<256>(t, blockIdx.x, threadIdx.x, b1, b2, eps, zero_grad, skip_flag);
}

// The slot totals in a fixed order: thread k walks slots k, k + 256, ..., xor-butterfly inside the wave, the four wave sums
// through LDS.  (Round 2 let ONE thread walk the slots: 512 dependent double loads, 19 us for a scalar.)
__device__ __forceinline__ void loss_slot_totals(const double* __restrict__ sums, int nslots, double* s_red /*[8]*/, double& s0, double& s1) {
    const int tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    // eight slot pairs per trip to memory, added in the same order as one by one (as a rolled loop every pair was a dependent
    // round trip: 16 + 8 of them made this scalar's kernel 9.5 us between the loss forward and its backward)
    const double2* s2 = reinterpret_cast<const double2*>(sums);
    for (int k0 = tid; k0 < nslots; k0 += 256 * 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + 256 * u; v[u] = s2[k < nslots ? k : nslots - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + 256 * u < nslots) { a += v[u].x; b += v[u].y; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d);
        b += __shfl_xor(b, d);
    }
    if ((tid & 63) == 0) { s_red[2 * (tid >> 6)] = a; s_red[2 * (tid >> 6) + 1] = b; }
    __syncthreads();
    s0 = (s_red[0] + s_red[2]) + (s_red[4] + s_red[6]);
    s1 = (s_red[1] + s_red[3]) + (s_red[5] + s_red[7]);
}

// loss = (1-lam) * sums[0]/n + lam * (1 - sums[1]/n)   (keeps the scalar on the device)
__global__ __launch_bounds__(256) void gp_loss_finalize_kernel(const double* __restrict__ sums, int nslots, double n, float lambda, float* __restrict__ loss) {
    __shared__ double s_tot[8];
    double s0, s1;
    loss_slot_totals(sums, nslots, s_tot, s0, s1);
    if (threadIdx.x == 0) loss[0] = (float)((1.0 - (double)lambda) * s0 / n + (double)lambda * (1.0 - s1 / n));
}

// the same + scale/n * sum|x| (one workgroup; fixed summation order)
__global__ __launch_bounds__(256) void gp_loss_finalize_reg_kernel(const double* __restrict__ sums, int nslots, double n, float lambda,
                                                                  const float* __restrict__ x, long nx, float scale_over_n,
                                                                  float* __restrict__ loss) {
    __shared__ float s_red[4];
    __shared__ double s_tot[8];
    float acc = 0.f;
    if ((nx & 3) == 0 && (((uintptr_t)x) & 15) == 0) {          // 16-byte loads, four independent per thread in flight
        const float4* x4 = (const float4*)x;
        const long n4 = nx >> 2;
        for (long i0 = threadIdx.x; i0 < n4; i0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const long i = i0 + 256 * u; v[u] = x4[i < n4 ? i : n4 - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 256 * u < n4) acc += (fabsf(v[u].x) + fabsf(v[u].y)) + (fabsf(v[u].z) + fabsf(v[u].w));
        }
    } else {
        for (long i = threadIdx.x; i < nx; i += 256) acc += fabsf(x[i]);
    }
    const float tot = block_sum_256(acc, s_red);
    double s0, s1;
    loss_slot_totals(sums, nslots, s_tot, s0, s1);
    if (threadIdx.x == 0)
        loss[0] = (float)((1.0 - (double)lambda) * s0 / n + (double)lambda * (1.0 - s1 / n)) + tot * scale_over_n;
}

// out[0] = base[0] + scale * mean|x|   [REF scene/gaussian_model.py:174-178: 1e-5 * mean(|motion feature|)]
// (multi-block: out is initialised by block 0's thread 0 through the host-side memset + base add)
__global__ __launch_bounds__(256) void gp_l1_mean_fwd_kernel(const float* __restrict__ x, long n, float scale_over_n,
                                                            const float* __restrict__ base, float* __restrict__ out) {
    __shared__ float s_red[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += fabsf(x[i]);
    const float tot = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) atomicAdd(out, tot * scale_over_n + (blockIdx.x == 0 ? base[0] : 0.f));
}
// g[i] = upstream[0] * scale/n * sign(x[i])
__global__ __launch_bounds__(256) void gp_l1_mean_bwd_kernel(const float* __restrict__ x, long n, float scale_over_n,
                                                            const float* __restrict__ upstream, float* __restrict__ g) {
    const float c = upstream[0] * scale_over_n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        g[i] = v > 0.f ? c : (v < 0.f ? -c : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static Win11 make_window() {
    // [REF utils/loss_utils.py:60-62]: exp(-(x - 5)^2 / (2 * 1.5^2)), normalised
    Win11 w;
    double s = 0.0, t[11];
    for (int x = 0; x < 11; ++x) { t[x] = exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); s += (float)t[x]; }
    for (int x = 0; x < 11; ++x) w.w[x] = (float)((float)t[x] / (float)s);
    return w;
}

extern "C" int gp_loss_l1_ssim_forward(const float* img, const float* gt, int32_t channels, int32_t H, int32_t W, double* sums,
                                       float* dmaps, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (!img || !gt || !sums) GP_FAIL("null argument");
    if (channels != 3 || H <= 0 || W <= 0) GP_FAIL("expects a [3,H,W] image (got C=%d H=%d W=%d)", channels, H, W);
    GpProfScope _p("l1_ssim_fwd", s);
    hipLaunchKernelGGL(gp_l1_ssim_fwd_kernel, dim3((W + LT - 1) / LT, (H + LT - 1) / LT, 3), dim3(256), 0, s, img, gt, H, W,
                       make_window(), sums, dmaps);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_loss_l1_ssim_finalize(const double* sums, int32_t channels, int32_t H, int32_t W, float lambda_dssim, float* loss,
                                        gp_stream_t stream_) {
    if (!sums || !loss) GP_FAIL("null argument");
    if (((uintptr_t)sums & 15) != 0) GP_FAIL("sums must be 16-byte aligned");
    hipLaunchKernelGGL(gp_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, sums, (int)GP_LOSS_SUM_SLOTS(H, W), (double)channels * H * W, lambda_dssim, loss);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_loss_l1_ssim_backward(const float* img, const float* gt, const float* dmaps, int32_t channels, int32_t H,
                                        int32_t W, float lambda_dssim, const float* upstream, float* dimg, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (!img || !gt || !dmaps || !dimg) GP_FAIL("null argument");
    if (channels != 3 || H <= 0 || W <= 0) GP_FAIL("expects a [3,H,W] image");
    GpProfScope _p("l1_ssim_bwd", s);
    hipLaunchKernelGGL(gp_l1_ssim_bwd_kernel, dim3((W + LT - 1) / LT, (H + LT - 1) / LT, 3), dim3(256), 0, s, img, gt, dmaps, H, W,
                       make_window(), lambda_dssim, upstream, dimg, (const float*)nullptr, 0L, 0.f, (float*)nullptr);
    GP_LAUNCH_CHECK();
    return 0;
}

// forward sums + image gradient in one launch (gp_l1_ssim_fused_kernel); the loss value still comes from a finalize call on `sums`
extern "C" int gp_loss_l1_ssim_fused(const float* img, const float* gt, int32_t channels, int32_t H, int32_t W, float lambda_dssim,
                                     const float* upstream, double* sums, float* dimg, const float* x, int64_t n, float scale, float* gx,
                                     gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (!img || !gt || !sums || !dimg) GP_FAIL("null argument");
    if (channels != 3 || H <= 0 || W <= 0) GP_FAIL("expects a [3,H,W] image (got C=%d H=%d W=%d)", channels, H, W);
    if ((x != nullptr) != (gx != nullptr)) GP_FAIL("regulariser input and gradient must be given together");
    if (x && (n <= 0 || n > 65536)) GP_FAIL("regulariser input must have 1..65536 elements");
    GpProfScope _p("l1_ssim_fused", s);
    hipLaunchKernelGGL(gp_l1_ssim_fused_kernel, dim3((W + LT - 1) / LT, (H + LT - 1) / LT, 3), dim3(256), 0, s, img, gt, H, W, make_window(),
                       lambda_dssim, upstream, sums, dimg, x, x ? (long)n : 0L, x ? scale / (float)n : 0.f, gx);
    GP_LAUNCH_CHECK();
    return 0;
}

// the two calls above with the regulariser  scale * mean|x|  [REF scene/gaussian_model.py:174-178] folded in (x small: the
// keypoint features of stage 2/3) -- saves the regulariser's own forward and backward launches
#define GP_LOSS_REG_MAX 65536
extern "C" int gp_loss_l1_ssim_finalize_reg(const double* sums, int32_t channels, int32_t H, int32_t W, float lambda_dssim,
                                            const float* x, int64_t n, float scale, float* loss, gp_stream_t stream_) {
    if (!sums || !loss || !x) GP_FAIL("null argument");
    if (((uintptr_t)sums & 15) != 0) GP_FAIL("sums must be 16-byte aligned");
    if (n <= 0 || n > GP_LOSS_REG_MAX) GP_FAIL("regulariser input must have 1..%d elements (use gp_l1_mean_forward beyond)", GP_LOSS_REG_MAX);
    hipLaunchKernelGGL(gp_loss_finalize_reg_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, sums, (int)GP_LOSS_SUM_SLOTS(H, W), (double)channels * H * W, lambda_dssim,
                       x, (long)n, scale / (float)n, loss);
    GP_LAUNCH_CHECK();
    return 0;
}
extern "C" int gp_loss_l1_ssim_backward_reg(const float* img, const float* gt, const float* dmaps, int32_t channels, int32_t H, int32_t W,
                                            float lambda_dssim, const float* upstream, float* dimg, const float* x, int64_t n,
                                            float scale, float* gx, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (!img || !gt || !dmaps || !dimg || !x || !gx) GP_FAIL("null argument");
    if (channels != 3 || H <= 0 || W <= 0) GP_FAIL("expects a [3,H,W] image");
    if (n <= 0 || n > GP_LOSS_REG_MAX) GP_FAIL("regulariser input must have 1..%d elements", GP_LOSS_REG_MAX);
    GpProfScope _p("l1_ssim_bwd", s);
    hipLaunchKernelGGL(gp_l1_ssim_bwd_kernel, dim3((W + LT - 1) / LT, (H + LT - 1) / LT, 3), dim3(256), 0, s, img, gt, dmaps, H, W,
                       make_window(), lambda_dssim, upstream, dimg, x, (long)n, scale / (float)n, gx);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                            float beta2, float eps, int64_t step, int32_t zero_grad, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (n < 0 || step < 1) GP_FAIL("bad n/step");
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) GP_FAIL("null argument");
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) GP_FAIL("adam: pointers must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    GpProfScope _p("adam", s);
    hipLaunchKernelGGL(gp_adam_kernel, dim3(gp_blocks(((size_t)n + 3) / 4, 256)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq,
                       (size_t)n, lr, beta1, beta2, eps, bc1, bc2_sqrt, zero_grad);
    GP_LAUNCH_CHECK();
    return 0;
}

// torch.optim.Adam counts steps PER PARAMETER: a tensor whose .grad is None when step() runs is skipped and its count stays behind
// (in the reference: the per-Gaussian tensors on every densify / prune iteration, train.py:164-197).  `steps` = the 1-based count of
// each tensor's update; the bias corrections are formed per tensor on the host.
// the chunk table of a launch and its number of chunks (0: nothing to update)
static int adam_build_table(AdamTable& t, long& n_chunks, int32_t count, float* const* params, float* const* grads, float* const* exp_avgs,
                            float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, const int64_t* steps, float beta1,
                            float beta2, uint32_t keep_grad_mask) {
    n_chunks = 0;
    if (count < 0 || count > ADAM_MAX_TENSORS) GP_FAIL("adam: at most %d tensors per call (got %d)", ADAM_MAX_TENSORS, count);
    if (count == 0) return 0;
    if (!params || !grads || !exp_avgs || !exp_avg_sqs || !numels || !lrs || !steps) GP_FAIL("null argument");
    t.count = 0;
    t.keep_grad_mask = 0;
    unsigned chunks = 0;
    for (int k = 0; k < count; ++k) {
        if (numels[k] <= 0) continue;
        if (steps[k] < 1) GP_FAIL("adam: bad step %lld (tensor %d)", (long long)steps[k], k);
        if ((((uintptr_t)params[k] | (uintptr_t)grads[k] | (uintptr_t)exp_avgs[k] | (uintptr_t)exp_avg_sqs[k]) & 15) != 0)
            GP_FAIL("adam: pointers must be 16-byte aligned (tensor %d)", k);
        const int j = t.count++;
        if ((keep_grad_mask >> k) & 1u) t.keep_grad_mask |= 1u << j;
        t.p[j] = params[k]; t.g[j] = grads[k]; t.m[j] = exp_avgs[k]; t.v[j] = exp_avg_sqs[k];
        t.n[j] = (unsigned long long)numels[k];
        t.step_size[j] = lrs[k] / (1.f - powf(beta1, (float)steps[k]));
        t.bc2_sqrt[j] = sqrtf(1.f - powf(beta2, (float)steps[k]));
        t.chunk_begin[j] = chunks;
        chunks += (unsigned)((numels[k] + ADAM_CHUNK - 1) / ADAM_CHUNK);
    }
    t.chunk_begin[t.count] = chunks;
    n_chunks = (long)chunks;
    return 0;
}

extern "C" int gp_adam_step_multi_steps(int32_t count, float* const* params, float* const* grads, float* const* exp_avgs,
                                        float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, const int64_t* steps,
                                        float beta1, float beta2, float eps, int32_t zero_grad, uint32_t keep_grad_mask,
                                        const uint32_t* skip_flag, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    AdamTable t;
    long chunks = 0;
    if (adam_build_table(t, chunks, count, params, grads, exp_avgs, exp_avg_sqs, numels, lrs, steps, beta1, beta2, keep_grad_mask)) return 1;
    if (chunks == 0) return 0;
    GpProfScope _p("adam", s);
    hipLaunchKernelGGL(gp_adam_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, s, t, beta1, beta2, eps, zero_grad, skip_flag);
    GP_LAUNCH_CHECK();
    return 0;
}

// ---- the rider (loss_adam_kernels.h)
GpAdamRider* gp_adam_rider_slot() {
    static thread_local GpAdamRider slot = {};
    return &slot;
}
int gp_adam_rider_arm(int count, float* const* params, float* const* grads, float* const* exp_avgs, float* const* exp_avg_sqs,
                      const int64_t* numels, const float* lrs, const int64_t* steps, float beta1, float beta2, float eps, int zero_grad,
                      uint32_t keep_grad_mask, const uint32_t* skip_flag) {
    GpAdamRider* r = gp_adam_rider_slot();
    r->armed = false;
    long chunks = 0;
    if (adam_build_table(r->t, chunks, count, params, grads, exp_avgs, exp_avg_sqs, numels, lrs, steps, beta1, beta2, keep_grad_mask)) return 1;
    r->b1 = beta1; r->b2 = beta2; r->eps = eps; r->zero_grad = zero_grad; r->skip_flag = skip_flag; r->chunks = (unsigned)chunks;
    r->armed = chunks > 0;
    return 0;
}
int gp_adam_rider_flush(hipStream_t s) {
    GpAdamRider* r = gp_adam_rider_slot();
    if (!r->armed) return 0;
    r->armed = false;
    GpProfScope _p("adam", s);
    hipLaunchKernelGGL(gp_adam_multi_kernel, dim3(r->chunks), dim3(256), 0, s, r->t, r->b1, r->b2, r->eps, r->zero_grad, r->skip_flag);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_adam_step_multi(int32_t count, float* const* params, float* const* grads, float* const* exp_avgs,
                                  float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, float beta1, float beta2,
                                  float eps, int64_t step, int32_t zero_grad, uint32_t keep_grad_mask, const uint32_t* skip_flag,
                                  gp_stream_t stream_) {
    if (count < 0 || count > ADAM_MAX_TENSORS) GP_FAIL("adam: at most %d tensors per call (got %d)", ADAM_MAX_TENSORS, count);
    if (step < 1) GP_FAIL("bad step");
    int64_t steps[ADAM_MAX_TENSORS];
    for (int k = 0; k < count; ++k) steps[k] = step;
    return gp_adam_step_multi_steps(count, params, grads, exp_avgs, exp_avg_sqs, numels, lrs, steps, beta1, beta2, eps, zero_grad,
                                    keep_grad_mask, skip_flag, stream_);
}

extern "C" int gp_l1_mean_forward(const float* x, int64_t n, float scale, const float* base, float* out, gp_stream_t stream_) {
    if (n <= 0) GP_FAIL("l1 mean: empty input");
    if (!x || !base || !out) GP_FAIL("null argument");
    hipStream_t s = (hipStream_t)stream_;
    GP_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float), s));
    unsigned blocks = gp_blocks((size_t)n, 4096);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(gp_l1_mean_fwd_kernel, dim3(blocks), dim3(256), 0, s, x, (long)n, scale / (float)n, base, out);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_l1_mean_backward(const float* x, int64_t n, float scale, const float* upstream, float* g, gp_stream_t stream_) {
    if (n <= 0) GP_FAIL("l1 mean: empty input");
    if (!x || !upstream || !g) GP_FAIL("null argument");
    unsigned blocks = gp_blocks((size_t)n, 1024);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gp_l1_mean_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, (long)n, scale / (float)n, upstream, g);
    GP_LAUNCH_CHECK();
    return 0;
}

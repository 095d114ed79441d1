// weights_kernels.hip -- the two per-frame stage-2/3 steps that produce the hot path's "keypoint weights"
// (SURVEY.md section 8f rank 1), hand-written for gfx950:
//   * multiresolution hash-grid encoding, forward and backward, of the reference's
//     tcnn.NetworkWithInputEncoding [REF scene/gaussian_model.py:370-392, called at :257]
//     (the 64-wide, bias-free MLP behind it is three plain GEMMs and stays on the library path);
//   * k nearest keypoints of every Gaussian [REF scene/gaussian_model.py:110-125, frnn_grid_points].
// Both dependencies (tinycudann float32 fork, frnn) are absent from /root/reference: the arithmetic follows the
// published algorithms as restated in oracle/weights_oracle.py -- parity unpinned.
#include "gp_common.h"

#define HG_MAX_LEVELS 16

struct HashGridDev {
    float scale[HG_MAX_LEVELS];
    uint32_t res[HG_MAX_LEVELS];
    uint32_t size[HG_MAX_LEVELS];
    uint32_t off[HG_MAX_LEVELS];
    int dense[HG_MAX_LEVELS];
    int L;
};

static int make_grid(const gp_hashgrid_config* c, HashGridDev& g, uint64_t* total) {
    if (!c) GP_FAIL("null hash-grid config");
    if (c->n_levels < 1 || c->n_levels > HG_MAX_LEVELS) GP_FAIL("hash grid: n_levels must be 1..%d", HG_MAX_LEVELS);
    if (c->n_features_per_level != 4) GP_FAIL("hash grid: only n_features_per_level = 4 is implemented");
    if (c->log2_hashmap_size < 4 || c->log2_hashmap_size > 28 || c->base_resolution < 1 || !(c->per_level_scale >= 1.f))
        GP_FAIL("hash grid: bad configuration");
    const double log2_b = log2((double)c->per_level_scale);   // in double: the host libm's float variants differ in the last bit
    uint64_t off = 0;
    g.L = c->n_levels;
    for (int l = 0; l < c->n_levels; ++l) {
        const float scale = (float)(exp2((double)l * log2_b) * (double)c->base_resolution - 1.0);
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint64_t full = (uint64_t)res * res * res;
        uint64_t size = (full + 7) / 8 * 8;
        const uint64_t cap = 1ull << c->log2_hashmap_size;
        if (size > cap) size = cap;
        g.scale[l] = scale; g.res[l] = res; g.size[l] = (uint32_t)size; g.off[l] = (uint32_t)off;
        g.dense[l] = full <= size;
        off += size;
        if (off > 0xFFFFFFFFull) GP_FAIL("hash grid: table too large");
    }
    if (total) *total = off;
    return 0;
}

extern "C" int64_t gp_hashgrid_table_entries(const gp_hashgrid_config* c) {
    HashGridDev g;
    uint64_t total = 0;
    if (make_grid(c, g, &total)) return -1;
    return (int64_t)total;
}

__device__ __forceinline__ uint32_t hg_index(const HashGridDev& g, int l, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t idx;
    if (g.dense[l]) idx = x + y * g.res[l] + z * g.res[l] * g.res[l];
    else idx = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return idx % g.size[l];
}

__device__ __forceinline__ void hg_cell(const HashGridDev& g, int l, const float* __restrict__ xyz, long i, float w[3], uint32_t c[3]) {
    const float s = g.scale[l];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p = fmaf(s, xyz[3 * i + d], 0.5f);
        const float f = floorf(p);
        w[d] = p - f;
        c[d] = (uint32_t)(int)f;
    }
}

// Forward.  A workgroup owns 64 points; `perm` (optional) lists the points in a spatially coherent order (Morton), so the
// 64 lanes of a wave -- which all work on the SAME level at a time -- read the same or neighbouring table entries.
// wave w encodes levels w, w+4, w+8, ...; the [64 points][L*4] tile leaves through LDS as whole 256-byte rows.
__global__ __launch_bounds__(256) void gp_hashgrid_fwd_kernel(HashGridDev g, long n, const float* __restrict__ xyz,
                                                             const int32_t* __restrict__ perm, const float4* __restrict__ table,
                                                             float4* __restrict__ out) {
    __shared__ float4 s_o[64][HG_MAX_LEVELS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long slot = (long)blockIdx.x * 64 + lane;
    const bool live = slot < n;
    const long i = live ? (perm ? (long)perm[slot] : slot) : 0;
    for (int l = wave; l < g.L; l += 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            float w[3];
            uint32_t c[3];
            hg_cell(g, l, xyz, i, w, c);
            const float4* tl = table + g.off[l];
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
                float cw = (bx ? w[0] : 1.f - w[0]);
                cw = cw * (by ? w[1] : 1.f - w[1]);
                cw = cw * (bz ? w[2] : 1.f - w[2]);
                const float4 v = tl[hg_index(g, l, c[0] + bx, c[1] + by, c[2] + bz)];
                acc.x = acc.x + cw * v.x; acc.y = acc.y + cw * v.y; acc.z = acc.z + cw * v.z; acc.w = acc.w + cw * v.w;
            }
        }
        s_o[lane][l] = acc;
    }
    __syncthreads();
    for (int e = tid; e < 64 * g.L; e += 256) {
        const int p = e / g.L, l = e - p * g.L;
        const long sp = (long)blockIdx.x * 64 + p;
        if (sp < n) out[(perm ? (long)perm[sp] : sp) * g.L + l] = s_o[p][l];
    }
}

// Backward w.r.t. the table: same ownership (64 points per workgroup, one level per wave pass), gradients staged through
// LDS; then FOUR lanes per (point, level), one per feature component, issue the atomics, so the four adds of a corner
// go to one 16-byte entry from adjacent lanes, and -- with `perm` -- adjacent points go to the same or nearby entries.
__global__ __launch_bounds__(256) void gp_hashgrid_bwd_kernel(HashGridDev g, long n, const float* __restrict__ xyz,
                                                             const int32_t* __restrict__ perm, const float* __restrict__ dL_dout,
                                                             float* __restrict__ dtable) {
    __shared__ float s_g[64][HG_MAX_LEVELS * 4 + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * g.L * 4; e += 256) {
        const int p = e / (g.L * 4), f = e - p * g.L * 4;
        const long sp = (long)blockIdx.x * 64 + p;
        s_g[p][f] = sp < n ? dL_dout[(perm ? (long)perm[sp] : sp) * g.L * 4 + f] : 0.f;
    }
    __syncthreads();
    const int comp = tid & 3, pl = tid >> 2;      // 64 (point) slots x 4 components per pass over the levels
    const long slot = (long)blockIdx.x * 64 + pl;
    if (slot >= n) return;
    const long i = perm ? (long)perm[slot] : slot;
    for (int l = 0; l < g.L; ++l) {
        const float go = s_g[pl][4 * l + comp];
        float w[3];
        uint32_t c[3];
        hg_cell(g, l, xyz, i, w, c);
        float* tl = dtable + 4 * (size_t)g.off[l];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
            float cw = (bx ? w[0] : 1.f - w[0]);
            cw = cw * (by ? w[1] : 1.f - w[1]);
            cw = cw * (bz ? w[2] : 1.f - w[2]);
            const float v = cw * go;
            if (v != 0.f) atomicAdd(tl + 4 * (size_t)hg_index(g, l, c[0] + bx, c[1] + by, c[2] + bz) + comp, v);
        }
    }
}

extern "C" int gp_hashgrid_forward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm, const float* table,
                                   float* out, gp_stream_t stream_) {
    HashGridDev g;
    if (make_grid(cfg, g, nullptr)) return 1;
    if (n < 0) GP_FAIL("negative n");
    if (n == 0) return 0;
    if (!xyz || !table || !out) GP_FAIL("null argument");
    if ((((uintptr_t)table | (uintptr_t)out) & 15) != 0) GP_FAIL("hash grid: table and output must be 16-byte aligned");
    GpProfScope _p("hashgrid_fwd", (hipStream_t)stream_);
    hipLaunchKernelGGL(gp_hashgrid_fwd_kernel, dim3(gp_blocks((size_t)n, 64)), dim3(256), 0, (hipStream_t)stream_, g, (long)n,
                       xyz, perm, (const float4*)table, (float4*)out);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_hashgrid_backward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm,
                                    const float* dL_dout, float* dtable, gp_stream_t stream_) {
    HashGridDev g;
    if (make_grid(cfg, g, nullptr)) return 1;
    if (n < 0) GP_FAIL("negative n");
    if (n == 0) return 0;
    if (!xyz || !dL_dout || !dtable) GP_FAIL("null argument");
    GpProfScope _p("hashgrid_bwd", (hipStream_t)stream_);
    hipLaunchKernelGGL(gp_hashgrid_bwd_kernel, dim3(gp_blocks((size_t)n, 64)), dim3(256), 0, (hipStream_t)stream_, g,
                       (long)n, xyz, perm, dL_dout, dtable);
    GP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// k nearest keypoints.  Keypoints (K x D, D = 3 or 3 + feature_dim) are staged TRANSPOSED in LDS so that every lane
// reads the same word (broadcast); a thread owns one Gaussian, keeps its D coordinates in registers and a sorted
// list of the best NN candidates.  Squared differences are summed in dimension order (as the oracle does).
// ------------------------------------------------------------------------------------------------
#define KNN_MAX_NN 16
#define KNN_MAX_D 35

#define KNN_TILE 256   // keypoints staged per pass
typedef float kv2f __attribute__((ext_vector_type(2)));

// NN is a template parameter: a run-time `best_d[nn - 1]` would put the candidate list in scratch memory.
// Two keypoints (j, j+1) are evaluated per step with packed fp32 math; each one's squared differences are still
// summed in dimension order, so the distances are bit-identical to the sequential form.
//
// `order` (optional permutation of the points, e.g. along a Morton curve) only changes which points share a wavefront.  A
// partial sum of squared differences is a lower bound of the full one (fp32 addition of non-negative terms never decreases),
// so a keypoint whose partial sum already reaches every lane's current NN-th best cannot enter any list: the wavefront drops it
// after 3 (and again after 19) of the 35 dimensions.  With spatially coherent wavefronts ~4 of 5 keypoints go that way;
// distances that are completed are summed exactly as before, so the result does not depend on `order`.
template <int D, int NN>
__global__ __launch_bounds__(256) void gp_knn_kernel(long n, const float* __restrict__ xyz, const float* __restrict__ feat,
                                                    float amplify, int K, const float* __restrict__ kp_xyz,
                                                    const float* __restrict__ kp_feat, const int32_t* __restrict__ order,
                                                    int64_t* __restrict__ idx_out, float* __restrict__ d2_out,
                                                    uint16_t* __restrict__ idx16_out) {
    __shared__ float s_kp[D][KNN_TILE];
    const int tid = threadIdx.x;
    const long nchunks = (n + 255) / 256;
    // K <= KNN_TILE (the reference's keypoint counts): the keypoints are staged ONCE and the workgroup walks over chunks of 256
    // points; more keypoints: one chunk per workgroup, tiles restaged (the launch sizes the grid accordingly)
    for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const long slot = chunk * 256 + tid;
        const bool live = slot < n;
        long i = order ? (long)order[live ? slot : n - 1] : (live ? slot : n - 1);         // clamped: the loads below are unconditional
        if ((unsigned long)i >= (unsigned long)n) i = n - 1;                              // (a malformed `order` must not read out of bounds)
        float x[D];
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = xyz[3 * i + d];
        if (D > 3) {                                                                   // rows of 32 floats, 16-byte aligned (checked at launch)
            const float4* row = (const float4*)(feat + (size_t)i * (D - 3));
#pragma unroll
            for (int q = 0; q < (D - 3) / 4; ++q) {
                const float4 v = row[q];
                x[3 + 4 * q + 0] = amplify * v.x; x[3 + 4 * q + 1] = amplify * v.y; x[3 + 4 * q + 2] = amplify * v.z; x[3 + 4 * q + 3] = amplify * v.w;
            }
        }
        float best_d[NN];
        int best_i[NN];
#pragma unroll
        for (int k = 0; k < NN; ++k) { best_d[k] = 3.4e38f; best_i[k] = -1; }
        auto offer = [&](float cd, int ci) {
            if (cd < best_d[NN - 1] || best_i[NN - 1] < 0) {
                // insert, keeping ascending order; equal distances keep the earlier (lower) index first.  Once the new entry is
                // placed everything behind it moves down one place (also entries that TIE with the one being carried)
                bool take = false;
#pragma unroll
                for (int k = 0; k < NN; ++k) {
                    take = take || best_i[k] < 0 || cd < best_d[k];
                    const float td = best_d[k];
                    const int ti = best_i[k];
                    if (take) { best_d[k] = cd; best_i[k] = ci; cd = td; ci = ti; }
                }
            }
        };
        for (int k0 = 0; k0 < K; k0 += KNN_TILE) {
            const int kt = min(KNN_TILE, K - k0);
            if (K > KNN_TILE || chunk == (long)blockIdx.x) {                            // uniform
                __syncthreads();
                for (int e = tid; e < KNN_TILE * 4; e += 256) {                         // positions (+ one padding lane of the /4 split)
                    const int j = e >> 2, d = e & 3;
                    if (d < 3) s_kp[d][j] = j < kt ? kp_xyz[3 * (size_t)(k0 + j) + d] : 3.0e18f;   // padding keypoints: never among the nearest
                }
                if (D > 3) {
                    for (int e = tid; e < KNN_TILE * (D - 3); e += 256) {               // coalesced feature rows
                        const int j = e / (D - 3), d = e - j * (D - 3);
                        s_kp[3 + d][j] = j < kt ? amplify * kp_feat[(size_t)(k0 + j) * (D - 3) + d] : 3.0e18f;
                    }
                }
                __syncthreads();
            }
            for (int j = 0; j < kt; j += 2) {
                kv2f d2 = {0.f, 0.f};
                constexpr int CUT0 = D > 3 ? 3 : D, CUT1 = D > 19 ? 19 : D;
#pragma unroll
                for (int d = 0; d < CUT0; ++d) {
                    const kv2f kp2 = *(const kv2f*)&s_kp[d][j];
                    const kv2f df = (kv2f){x[d], x[d]} - kp2;
                    d2 = d2 + df * df;
                }
                if (D > CUT0 && !__any(best_i[NN - 1] < 0 || d2.x < best_d[NN - 1] || d2.y < best_d[NN - 1])) continue;
#pragma unroll
                for (int d = CUT0; d < CUT1; ++d) {
                    const kv2f kp2 = *(const kv2f*)&s_kp[d][j];
                    const kv2f df = (kv2f){x[d], x[d]} - kp2;
                    d2 = d2 + df * df;
                }
                if (D > CUT1 && !__any(best_i[NN - 1] < 0 || d2.x < best_d[NN - 1] || d2.y < best_d[NN - 1])) continue;
#pragma unroll
                for (int d = CUT1; d < D; ++d) {
                    const kv2f kp2 = *(const kv2f*)&s_kp[d][j];
                    const kv2f df = (kv2f){x[d], x[d]} - kp2;
                    d2 = d2 + df * df;
                }
                offer(d2.x, k0 + j);
                if (j + 1 < kt) offer(d2.y, k0 + j + 1);
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < NN; ++k) {
                idx_out[i * NN + k] = best_i[k];
                if (d2_out) d2_out[i * NN + k] = best_d[k];
                if (idx16_out) idx16_out[i * NN + k] = (uint16_t)best_i[k];       // (the blend kernels' packed copy: K < 65536)
            }
        }
    }
}

template <int D>
static void launch_knn(int nn, dim3 grid, hipStream_t s, long n, const float* xyz, const float* feat, float amplify, int K,
                       const float* kp_xyz, const float* kp_feat, const int32_t* order, int64_t* idx_out, float* d2_out,
                       uint16_t* idx16_out) {
#define KNN_CASE(NNV) case NNV: hipLaunchKernelGGL((gp_knn_kernel<D, NNV>), grid, dim3(256), 0, s, n, xyz, feat, amplify, K, kp_xyz, \
                                                   kp_feat, order, idx_out, d2_out, idx16_out); break;
    switch (nn) {
        KNN_CASE(1) KNN_CASE(2) KNN_CASE(3) KNN_CASE(4) KNN_CASE(5) KNN_CASE(6) KNN_CASE(7) KNN_CASE(8)
        KNN_CASE(9) KNN_CASE(10) KNN_CASE(11) KNN_CASE(12) KNN_CASE(13) KNN_CASE(14) KNN_CASE(15) KNN_CASE(16)
    }
#undef KNN_CASE
}

extern "C" int gp_knn_keypoints(int64_t n, const float* xyz, const float* feat, int32_t feat_dim, float amplify, int64_t K,
                                const float* kp_xyz, const float* kp_feat, int32_t nn, const int32_t* order, int64_t* idx_out,
                                float* d2_out, uint16_t* idx16_out, gp_stream_t stream_) {
    if (n < 0 || K < 0) GP_FAIL("negative size");
    if (n == 0) return 0;
    if (nn < 1 || nn > KNN_MAX_NN) GP_FAIL("knn: nearest_num %d unsupported (1..%d)", nn, KNN_MAX_NN);
    if (K < nn) GP_FAIL("knn: fewer keypoints (%ld) than nearest_num (%d)", (long)K, nn);
    if (!xyz || !kp_xyz || !idx_out) GP_FAIL("null argument");
    if (idx16_out && K > 65535) GP_FAIL("knn: idx16_out needs fewer than 65536 keypoints");
    if (feat_dim != 0 && feat_dim != 32) GP_FAIL("knn: feature_dim must be 0 (knn_type 3D) or 32 (knn_type hybird), got %d", feat_dim);
    if (feat_dim && (!feat || !kp_feat)) GP_FAIL("knn: null feature pointers");
    hipStream_t s = (hipStream_t)stream_;
    GpProfScope _p("knn", s);
    if (feat_dim && (((uintptr_t)feat & 15) != 0)) GP_FAIL("knn: feat must be 16-byte aligned");
    const unsigned nchunks = gp_blocks((size_t)n, 256);
    const unsigned resident = 256u * (feat_dim ? 4u : 8u);                             // workgroups the chip holds (registers / LDS of the two forms)
    const dim3 grid(K <= KNN_TILE && nchunks > resident ? resident : nchunks);
    if (feat_dim == 0) launch_knn<3>(nn, grid, s, (long)n, xyz, feat, amplify, (int)K, kp_xyz, kp_feat, order, idx_out, d2_out, idx16_out);
    else launch_knn<35>(nn, grid, s, (long)n, xyz, feat, amplify, (int)K, kp_xyz, kp_feat, order, idx_out, d2_out, idx16_out);
    GP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Mean squared distance of every point to its three nearest OTHER points -- simple_knn's distCUDA2, which sizes the initial
// Gaussians in create_from_pcd [REF scene/gaussian_model.py:340-341].  simple_knn is an un-vendored CUDA dependency of the
// reference (published algorithm: exact 3-NN over Morton-ordered boxes, self excluded BY INDEX -- a coincident point counts
// with distance 0 -- result (d0 + d1 + d2) / 3 of the SQUARED distances): parity unpinned, oracle/weights_oracle.py states it.
// Initialisation-time, run once: exact brute force, every workgroup streams all points through LDS in tiles of 256 (the
// inner product form is NOT used: (x - y)^2 summed per axis keeps coincident points at exactly 0).  1e5 points: ~3 ms.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_knn3_mean_dist2_kernel(long n, const float* __restrict__ xyz, float* __restrict__ out) {
    __shared__ float4 s_p[256];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (i < n) { qx = xyz[3 * i]; qy = xyz[3 * i + 1]; qz = xyz[3 * i + 2]; }
    float b0 = 3.402823466e38f, b1 = b0, b2 = b0;          // ascending; FLT_MAX where fewer than three other points exist
    for (long t0 = 0; t0 < n; t0 += 256) {
        const long j = t0 + threadIdx.x;
        __syncthreads();
        s_p[threadIdx.x] = j < n ? make_float4(xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const int cnt = (int)((n - t0) < 256 ? (n - t0) : 256);
        for (int k = 0; k < cnt; ++k) {
            const float4 p = s_p[k];                       // (broadcast read)
            const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
            float d = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
            if (t0 + k == i) d = 3.402823466e38f;           // self, by index
            if (d < b2) {
                if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; }
                else b2 = d;
            }
        }
    }
    if (i < n) out[i] = (b0 + b1 + b2) / 3.f;
}

extern "C" int gp_knn3_mean_dist2(int64_t n, const float* xyz, float* out, gp_stream_t stream_) {
    if (n < 0) GP_FAIL("negative size");
    if (n == 0) return 0;
    if (!xyz || !out) GP_FAIL("null argument");
    hipLaunchKernelGGL(gp_knn3_mean_dist2_kernel, dim3(gp_blocks((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream_, (long)n, xyz, out);
    GP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Fused weights model: hash-grid encoding + the 64-wide bias-free MLP (64 -> 64 -> 64 -> 16, ReLU) on the exact-fp32
// matrix cores (v_mfma_f32_16x16x4_f32), persistent workgroups.  A workgroup walks blocks of 64 points (in `perm` order); per
// block the activations live transposed in LDS, T[f][p] with a row pitch of 65 words.
//   forward : out[i][0..n_out) and (training) the encoded features, saved slot-major ([slot][64], slot = position in perm)
//   backward: recomputes the two hidden layers from the saved features, back-propagates, accumulates the three weight
//             gradients in MFMA accumulators across all of the workgroup's blocks (flushed once) and writes dL/d(features)
//             slot-major for the table-gradient kernel.
// Row-strip tiling: wave w of a workgroup owns the 16 feature rows 16 w .. 16 w + 15 of every 64-row product and all 64 points of
// the block (four 16-point column tiles = four independent accumulator chains).  A strip of a 64 x 64 matrix is 16 registers
// (a[s] = A[16 w + i][s + 16 kg], lane = (i, kg)), so the four weight operands of the backward (W1, W2, W1^T, W2^T) cost 64
// registers instead of the 128 a 32 x 32 tiling needs; the kernels fit 256 registers and TWO workgroups share a CU: one's
// matrix-core phases cover the other's gathers, LDS epilogues and barriers (32 x 32 tiles, one wave per SIMD: backward 1.23 ms
// with the matrix pipe 29 % busy, forward 1.16 ms; now 0.58 / 0.90 ms at 1 M points).
// K index of MFMA step s in lane group kg: s + 16 kg.  With the pitch of 65 every operand read is conflict-free (rows s / s + 16
// land 16 banks apart, a column of 16 rows walks 16 banks) and -- unlike an XOR swizzle -- an address is lane base + compile-time
// offset, so the reads are immediates off ONE register per access pattern (with XOR the compiler hoists some 150 loop-invariant
// addresses out of the block loop and spills).
// ------------------------------------------------------------------------------------------------
#define WM_THREADS 256
typedef float wf32x4 __attribute__((ext_vector_type(4)));
#define WM4_PITCH 65
__device__ __forceinline__ int wm4_ti(int f, int p) { return f * WM4_PITCH + p; }

__device__ __forceinline__ void wm_encode_block(const HashGridDev& g, long n, long slot0, const float* __restrict__ xyz,
                                                const int32_t* __restrict__ perm, const float4* __restrict__ table, float* T,
                                                int tid) {
    const int p = tid & 63;
    const long slot = slot0 + p;
    const bool live = slot < n;
    const long i = live ? (perm ? (long)perm[slot] : slot) : 0;
    const float x3[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    // a wave encodes levels w, w + 4, w + 8, w + 12: TWO levels' sixteen corner fetches are in flight at a time (one level at a
    // time, each level's eight gathers were waited for before the next level's addresses were even computed)
#pragma unroll
    for (int lp = 0; lp < 2; ++lp) {
        float4 v[2][8];
        float w[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int l = (tid >> 6) + 8 * lp + 4 * h;
            uint32_t c[3];
            const float s = g.scale[l];
#pragma unroll
            for (int d = 0; d < 3; ++d) {                       // (hg_cell on the registers)
                const float pp = fmaf(s, x3[d], 0.5f);
                const float f = floorf(pp);
                w[h][d] = pp - f;
                c[d] = (uint32_t)(int)f;
            }
            const float4* tl = table + g.off[l];
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
                v[h][corner] = tl[hg_index(g, l, c[0] + bx, c[1] + by, c[2] + bz)];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int l = (tid >> 6) + 8 * lp + 4 * h;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
                float cw = (bx ? w[h][0] : 1.f - w[h][0]);
                cw = cw * (by ? w[h][1] : 1.f - w[h][1]);
                cw = cw * (bz ? w[h][2] : 1.f - w[h][2]);
                const float4 q = v[h][corner];
                acc.x = acc.x + cw * q.x; acc.y = acc.y + cw * q.y; acc.z = acc.z + cw * q.z; acc.w = acc.w + cw * q.w;
            }
            if (!(live && l < g.L)) acc = make_float4(0.f, 0.f, 0.f, 0.f);
            T[wm4_ti(4 * l + 0, p)] = acc.x; T[wm4_ti(4 * l + 1, p)] = acc.y; T[wm4_ti(4 * l + 2, p)] = acc.z; T[wm4_ti(4 * l + 3, p)] = acc.w;
        }
    }
}

// acc[c][r] = sum_k A[16 w + 4 kg + r][k] T[k][16 c + n]   (a[s] = A-strip operand, K = 64)
__device__ __forceinline__ void wm4_layer(wf32x4 (&acc)[4], const float (&a)[16], const float* T, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (wf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = s + 16 * kg;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], T[wm4_ti(k, 16 * c + n)], acc[c], 0, 0, 0);
    }
}
// g[t][r] += sum over the block's 64 points of TA[ra + 4 kg + r][p] * TB[16 t + n][p]
__device__ __forceinline__ void wm4_outer(wf32x4 (&g)[4], const float* TA, int ra, const float* TB, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int p = s + 16 * kg;
        const float a = TA[wm4_ti(ra + n, p)];
#pragma unroll
        for (int t = 0; t < 4; ++t) g[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, TB[wm4_ti(16 * t + n, p)], g[t], 0, 0, 0);
    }
}
// write a strip's product back: T[16 w + 4 kg + r][16 c + n] = f(acc[c][r])
template <class F>
__device__ __forceinline__ void wm4_store(float* T, const wf32x4 (&acc)[4], int w, int lane, F f) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = wm4_ti(16 * w + 4 * kg + r, 16 * c + n);
            T[idx] = f(acc[c][r], idx);
        }
}

__global__ __launch_bounds__(WM_THREADS, 2) void gp_wm_fwd_kernel(HashGridDev g, long n, const float* __restrict__ xyz,
                                                                 const int32_t* __restrict__ perm, const float* __restrict__ params,
                                                                 int n_out, float* __restrict__ out, float* __restrict__ saved_feat) {
    __shared__ float sA[64 * WM4_PITCH], sB[64 * WM4_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kg = lane >> 4;
    const float* W1 = params;
    const float* W2 = params + 4096;
    const float* W3 = params + 8192;
    const float4* table = (const float4*)(params + 9216);
    float w1[16], w2[16], w3[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = s + 16 * kg;
        w1[s] = W1[(16 * w + i) * 64 + k];
        w2[s] = W2[(16 * w + i) * 64 + k];
        w3[s] = W3[i * 64 + k];                                       // the output layer's 16 (padded) rows, the same strip in every wave
    }
    const long nblocks = (n + 63) / 64;
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long slot0 = b * 64;
        __syncthreads();
        wm_encode_block(g, n, slot0, xyz, perm, table, sA, tid);
        __syncthreads();
        if (saved_feat) {   // slot-major rows of 64 floats
            for (int e = tid; e < 64 * 64; e += WM_THREADS) {
                const int p = e >> 6, f = e & 63;
                if (slot0 + p < n) saved_feat[(slot0 + p) * 64 + f] = sA[wm4_ti(f, p)];
            }
        }
        wf32x4 acc[4];
        wm4_layer(acc, w1, sA, lane);
        wm4_store(sB, acc, w, lane, [](float v, int) { return fmaxf(v, 0.f); });
        __syncthreads();
        wm4_layer(acc, w2, sB, lane);
        __syncthreads();                        // everyone is done reading sA (features) before it is overwritten
        wm4_store(sA, acc, w, lane, [](float v, int) { return fmaxf(v, 0.f); });
        __syncthreads();
        {                                       // output layer: 16 rows x the wave's 16 points (column tile w)
            wf32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[s], sA[wm4_ti(s + 16 * kg, 16 * w + i)], o, 0, 0, 0);
            const long slot = slot0 + 16 * w + i;
            if (slot < n) {
                const long pt = perm ? (long)perm[slot] : slot;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * kg + r < n_out) out[pt * n_out + 4 * kg + r] = o[r];
            }
        }
    }
}

// One block's inputs of the backward kernel into registers: 64 points x 64 saved features (float4 per thread x 4) and
// the upstream gradient rows (gathered through the Morton permutation), issued a whole block ahead of their use.
__device__ __forceinline__ void wm_bwd_fetch(float4 (&px)[4], float (&pd)[4], long b, long nblocks, long n, const int32_t* __restrict__ perm,
                                             int n_out, const float* __restrict__ saved_feat, const float* __restrict__ dL_dout, int tid) {
    if (b >= nblocks) return;
    const long slot0 = b * 64;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e4 = tid + WM_THREADS * u, p = e4 >> 4, f = 4 * (e4 & 15);
        px[u] = slot0 + p < n ? *(const float4*)(saved_feat + (slot0 + p) * 64 + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + WM_THREADS * u, p = e >> 4, f = e & 15;
        float v = 0.f;
        if (f < n_out && slot0 + p < n) v = dL_dout[(perm ? (long)perm[slot0 + p] : slot0 + p) * n_out + f];
        pd[u] = v;
    }
}

__global__ __launch_bounds__(WM_THREADS, 2) void gp_wm_bwd_kernel(long n, const int32_t* __restrict__ perm,
                                                                 const float* __restrict__ params, int n_out,
                                                                 const float* __restrict__ saved_feat, const float* __restrict__ dL_dout,
                                                                 float* __restrict__ dparams, float* __restrict__ dfeat) {
    __shared__ float sA[64 * WM4_PITCH], sB[64 * WM4_PITCH], sC[64 * WM4_PITCH], sD[16 * WM4_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kg = lane >> 4;
    const float* W1 = params;
    const float* W2 = params + 4096;
    const float* W3 = params + 8192;
    float w1[16], w2[16], w1t[16], w2t[16], w3t[4];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = s + 16 * kg, row = 16 * w + i;
        w1[s] = W1[row * 64 + k];
        w2[s] = W2[row * 64 + k];
        w1t[s] = W1[k * 64 + row];                                    // (W1^T)[row][k]
        w2t[s] = W2[k * 64 + row];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) w3t[s] = W3[(s + 4 * kg) * 64 + 16 * w + i];   // (W3^T)[row][k], k = s + 4 kg < 16
    wf32x4 g1[4], g2[4], g3;   // rows 16 w.. of dW1 / dW2 (all 64 columns); columns 16 w.. of dW3 (its 16 rows)
#pragma unroll
    for (int t = 0; t < 4; ++t) { g1[t] = (wf32x4){0.f, 0.f, 0.f, 0.f}; g2[t] = g1[t]; }
    g3 = (wf32x4){0.f, 0.f, 0.f, 0.f};
    const long nblocks = (n + 63) / 64;
    float4 px[4];
    float pd[4];
    wm_bwd_fetch(px, pd, blockIdx.x, nblocks, n, perm, n_out, saved_feat, dL_dout, tid);
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long slot0 = b * 64;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                // X^T (prefetched one block ahead)
            const int e4 = tid + WM_THREADS * u, p = e4 >> 4, f = 4 * (e4 & 15);
            sA[wm4_ti(f + 0, p)] = px[u].x; sA[wm4_ti(f + 1, p)] = px[u].y; sA[wm4_ti(f + 2, p)] = px[u].z; sA[wm4_ti(f + 3, p)] = px[u].w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                // dZ3^T (16 rows, those >= n_out zero)
            const int e = tid + WM_THREADS * u;
            sD[wm4_ti(e & 15, e >> 4)] = pd[u];
        }
        wm_bwd_fetch(px, pd, b + gridDim.x, nblocks, n, perm, n_out, saved_feat, dL_dout, tid);   // in flight during this block's GEMM phases
        __syncthreads();
        wf32x4 acc[4];
        wm4_layer(acc, w1, sA, lane);                                // H1^T -> sB
        wm4_store(sB, acc, w, lane, [](float v, int) { return fmaxf(v, 0.f); });
        __syncthreads();
        wm4_layer(acc, w2, sB, lane);                                // H2^T -> sC
        wm4_store(sC, acc, w, lane, [](float v, int) { return fmaxf(v, 0.f); });
        __syncthreads();
        {   // dW3[o][16 w + n] += dZ3^T[o][p] H2^T[16 w + n][p]      (one 16 x 16 tile per wave)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int p = s + 16 * kg;
                g3 = __builtin_amdgcn_mfma_f32_16x16x4f32(sD[wm4_ti(i, p)], sC[wm4_ti(16 * w + i, p)], g3, 0, 0, 0);
            }
        }
        {   // dH2^T = W3^T dZ3^T (K = 16), masked by H2 > 0, in place over sC
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = (wf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3t[s], sD[wm4_ti(s + 4 * kg, 16 * c + i)], acc[c], 0, 0, 0);
            __syncthreads();                                         // every wave's dW3 products have read sC
            wm4_store(sC, acc, w, lane, [&](float v, int idx) { return sC[idx] > 0.f ? v : 0.f; });
        }
        __syncthreads();
        wm4_outer(g2, sC, 16 * w, sB, lane);                         // dW2[o][i] += dZ2^T[o][p] H1^T[i][p]
        wm4_layer(acc, w2t, sC, lane);                               // dH1^T = W2^T dZ2^T
        __syncthreads();                                             // every wave's dW2 products have read sB
        wm4_store(sB, acc, w, lane, [&](float v, int idx) { return sB[idx] > 0.f ? v : 0.f; });   // dZ1^T in place over H1^T
        __syncthreads();
        wm4_outer(g1, sB, 16 * w, sA, lane);                         // dW1[o][i] += dZ1^T[o][p] X^T[i][p]
        wm4_layer(acc, w1t, sB, lane);                               // dX^T = W1^T dZ1^T -> sC (dZ2 is dead: all waves passed the barrier above)
        wm4_store(sC, acc, w, lane, [](float v, int) { return v; });
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += WM_THREADS) {            // slot-major rows for the table-gradient kernel
            const int p = e >> 6, f = e & 63;
            if (slot0 + p < n) dfeat[(slot0 + p) * 64 + f] = sC[wm4_ti(f, p)];
        }
    }
    // flush the weight gradients: for a fixed register the 16 lanes of a group write 64 contiguous bytes
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + 4 * kg + r;
            atomicAdd(&dparams[row * 64 + 16 * t + i], g1[t][r]);
            atomicAdd(&dparams[4096 + row * 64 + 16 * t + i], g2[t][r]);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&dparams[8192 + (4 * kg + r) * 64 + 16 * w + i], g3[r]);
}

// table gradient from slot-major feature gradients (the fused backward's output).
// Measured cost model on this part: an atomic instruction costs ~29 CU-cycles per DISTINCT cache line it touches.
// EIGHT lanes per (point, level): (x-corner bit, component).  The two x-neighbours of a corner pair are adjacent table
// entries on dense levels and -- because the x prime of the spatial hash is 1 -- within the same aligned group of four
// entries (one 64-byte line) three times out of four on hashed levels, so the eight lanes usually share ONE line:
// ~5 line-operations per (point, level) instead of 8.
#define HG_BOX_CAP 1024   // LDS box entries (x 4 floats) for the dense levels
__global__ __launch_bounds__(256) void gp_hashgrid_bwd_slots_kernel(HashGridDev g, long n, const float* __restrict__ xyz,
                                                                   const int32_t* __restrict__ perm, const float* __restrict__ dfeat,
                                                                   float* __restrict__ dtable) {
    __shared__ float s_box[HG_BOX_CAP * 4];
    __shared__ int s_lo[3], s_hi[3];
    const int tid = threadIdx.x;
    const int comp = tid & 3, bx = (tid >> 2) & 1, pl = tid >> 3;
    const long slot = (long)blockIdx.x * 32 + pl;
    const bool live = slot < n;
    const long i = live ? (perm ? (long)perm[slot] : slot) : 0;
    for (int l = 0; l < g.L; ++l) {
        const float go = live ? dfeat[slot * 64 + 4 * l + comp] : 0.f;
        float w[3];
        uint32_t c[3];
        hg_cell(g, l, xyz, i, w, c);
        float* tl = dtable + 4 * (size_t)g.off[l];
        const float wx = bx ? w[0] : 1.f - w[0];
        bool boxed = false;
        int ex = 0, ey = 0;
        if (g.dense[l]) {
            // Dense (coarse) level: the workgroup's 32 spatially adjacent points fall into a handful of cells, i.e. they
            // would all hammer the same few entries.  Accumulate in an LDS box spanning their cells and flush each entry
            // once, as whole cache lines (consecutive x entries x 4 components from consecutive lanes).
            if (tid < 3) { s_lo[tid] = 0x7fffffff; s_hi[tid] = -0x7fffffff; }
            __syncthreads();
            if (live && bx == 0 && comp < 3) { atomicMin(&s_lo[comp], (int)c[comp]); atomicMax(&s_hi[comp], (int)c[comp]); }
            __syncthreads();
            const int lo0 = s_lo[0], lo1 = s_lo[1], lo2 = s_lo[2];
            ex = s_hi[0] - lo0 + 2; ey = s_hi[1] - lo1 + 2;
            const int ez = s_hi[2] - lo2 + 2;
            const long vol = (long)ex * ey * ez;
            boxed = s_hi[0] >= lo0 && vol <= HG_BOX_CAP;            // uniform
            if (boxed) {
                for (int e = tid; e < (int)vol * 4; e += 256) s_box[e] = 0.f;
                __syncthreads();
                if (live) {
#pragma unroll
                    for (int cyz = 0; cyz < 4; ++cyz) {
                        const int by = cyz & 1, bz = cyz >> 1;
                        float cw = wx * (by ? w[1] : 1.f - w[1]);
                        cw = cw * (bz ? w[2] : 1.f - w[2]);
                        const float v = cw * go;
                        const int bi = (((int)c[2] - lo2 + bz) * ey + ((int)c[1] - lo1 + by)) * ex + ((int)c[0] - lo0 + bx);
                        if (v != 0.f) atomicAdd(&s_box[4 * bi + comp], v);
                    }
                }
                __syncthreads();
                for (int e = tid; e < (int)vol * 4; e += 256) {
                    const float v = s_box[e];
                    if (v != 0.f) {
                        const int bi = e >> 2, x = bi % ex, y = (bi / ex) % ey, z = bi / (ex * ey);
                        atomicAdd(tl + 4 * (size_t)hg_index(g, l, (uint32_t)(lo0 + x), (uint32_t)(lo1 + y), (uint32_t)(lo2 + z)) + (e & 3), v);
                    }
                }
                __syncthreads();
            }
        }
        if (!boxed && live) {
#pragma unroll
            for (int cyz = 0; cyz < 4; ++cyz) {
                const int by = cyz & 1, bz = cyz >> 1;
                float cw = wx * (by ? w[1] : 1.f - w[1]);
                cw = cw * (bz ? w[2] : 1.f - w[2]);
                const float v = cw * go;
                if (v != 0.f) atomicAdd(tl + 4 * (size_t)hg_index(g, l, c[0] + bx, c[1] + by, c[2] + bz) + comp, v);
            }
        }
    }
}

static int wm_check(const gp_hashgrid_config* cfg, HashGridDev& g, int64_t n, int32_t n_out) {
    if (make_grid(cfg, g, nullptr)) return 1;
    if (g.L != 16) GP_FAIL("fused weights model: n_levels must be 16 (64 encoded features)");
    if (n < 0) GP_FAIL("negative n");
    if (n_out < 1 || n_out > 16) GP_FAIL("fused weights model: n_out must be 1..16");
    return 0;
}

extern "C" int gp_weights_forward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm, const float* params,
                                  int32_t n_out, float* out, float* saved_feat, gp_stream_t stream_) {
    HashGridDev g;
    if (wm_check(cfg, g, n, n_out)) return 1;
    if (n == 0) return 0;
    if (!xyz || !params || !out) GP_FAIL("null argument");
    if (((uintptr_t)params & 15) != 0) GP_FAIL("weights model: params must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream_;
    GpProfScope _p("weights_fwd", s);
    const unsigned nb = (unsigned)((n + 63) / 64);
    const unsigned cap = gp_debug_get(4) > 0 ? (unsigned)gp_debug_get(4) : 512u;   // two resident workgroups per CU (254 registers), persistent
    hipLaunchKernelGGL(gp_wm_fwd_kernel, dim3(nb < cap ? nb : cap), dim3(WM_THREADS), 0, s, g, (long)n, xyz, perm, params, n_out, out,
                       saved_feat);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_weights_backward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm, const float* params,
                                   int32_t n_out, const float* saved_feat, const float* dL_dout, float* dparams, gp_alloc_fn alloc,
                                   void* alloc_ctx, gp_stream_t stream_) {
    HashGridDev g;
    if (wm_check(cfg, g, n, n_out)) return 1;
    if (n == 0) return 0;
    if (!xyz || !params || !saved_feat || !dL_dout || !dparams || !alloc) GP_FAIL("null argument");
    hipStream_t s = (hipStream_t)stream_;
    float* dfeat = (float*)alloc(alloc_ctx, GP_BUF_TEMP, gp_align_up((size_t)n * 64 * sizeof(float), 256));
    if (!dfeat) GP_FAIL("allocator returned NULL for TEMP");
    const unsigned nb = (unsigned)((n + 63) / 64);
    {
        GpProfScope _p("weights_bwd_mlp", s);
        hipLaunchKernelGGL(gp_wm_bwd_kernel, dim3(nb < 512u ? nb : 512u), dim3(WM_THREADS), 0, s, (long)n, perm, params, n_out, saved_feat,
                           dL_dout, dparams, dfeat);
        GP_LAUNCH_CHECK();
    }
    GpProfScope _p("weights_bwd_table", s);
    hipLaunchKernelGGL(gp_hashgrid_bwd_slots_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, g, (long)n, xyz, perm,
                       (const float*)dfeat, dparams + 9216);
    GP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Furthest-point sampling (keypoint growth: GaussianModel.get_new_kpts [REF scene/gaussian_model.py:196-212]; the
// reference calls pointops' CUDA kernel through utils/fps.py:71-88).  Contract: idx[0] = 0; idx[j] = the point with the
// largest distance to the already selected set (first maximum on ties).  The selection is inherently sequential in j, so
// ONE workgroup of 1024 threads walks it: per round every thread refreshes min-distance for its strided share of the points
// against the point chosen last, the (distance, index) maximum goes through a wave64 DPP-free shuffle tree and one LDS
// exchange between the 16 waves.  n = 300 k candidates, m = 300 samples: ~1 ms, every few hundred iterations.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gp_fps_kernel(long n, const float* __restrict__ xyz, long m, int32_t* __restrict__ idx,
                                                       float* __restrict__ dist) {
    __shared__ float s_far_d2[16];
    __shared__ int s_far_id[16];
    __shared__ int s_sel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (long k = tid; k < n; k += 1024) dist[k] = 1e10f;
    if (tid == 0) { idx[0] = 0; s_sel = 0; }
    __syncthreads();
    for (long j = 1; j < m; ++j) {
        const int cur = s_sel;
        const float cx = xyz[3 * (size_t)cur], cy = xyz[3 * (size_t)cur + 1], cz = xyz[3 * (size_t)cur + 2];
        float far_d2 = -1.f;
        int far_id = 0x7fffffff;
        for (long k = tid; k < n; k += 1024) {
            const float dx = xyz[3 * k] - cx, dy = xyz[3 * k + 1] - cy, dz = xyz[3 * k + 2] - cz;
            const float d = fminf(dx * dx + dy * dy + dz * dz, dist[k]);
            dist[k] = d;
            if (d > far_d2) { far_d2 = d; far_id = (int)k; }          // strictly greater: the first maximum of this thread's share
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const float o_d2 = __shfl_xor(far_d2, dd);
            const int o_id = __shfl_xor(far_id, dd);
            if (o_d2 > far_d2 || (o_d2 == far_d2 && o_id < far_id)) { far_d2 = o_d2; far_id = o_id; }
        }
        __syncthreads();                                         // (s_sel was read by everyone)
        if (lane == 0) { s_far_d2[wave] = far_d2; s_far_id[wave] = far_id; }
        __syncthreads();
        if (tid == 0) {
            float b = s_far_d2[0];
            int bi = s_far_id[0];
            for (int w = 1; w < 16; ++w)
                if (s_far_d2[w] > b || (s_far_d2[w] == b && s_far_id[w] < bi)) { b = s_far_d2[w]; bi = s_far_id[w]; }
            idx[j] = bi;
            s_sel = bi;
        }
        __syncthreads();
    }
}

extern "C" int gp_furthest_point_sampling(int64_t n, const float* xyz, int64_t m, int32_t* idx_out, float* tmp_dist, gp_stream_t stream_) {
    if (n < 0 || n > 0x7FFFFFF0LL || m < 0 || m > n) GP_FAIL("gp_furthest_point_sampling: need 0 <= m <= n < 2^31");
    if (m == 0) return 0;
    if (!xyz || !idx_out || !tmp_dist) GP_FAIL("null argument");
    hipLaunchKernelGGL(gp_fps_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream_, (long)n, xyz, (long)m, idx_out, tmp_dist);
    GP_LAUNCH_CHECK();
    return 0;
}

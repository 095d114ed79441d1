// gp_capi_deform.hip -- extern "C" entry points of the deformation path (see include/gp_hip.h).
#include "gp_common.h"
#include "deform_kernels.h"
#include "loss_adam_kernels.h"
#include <stdlib.h>

static int make_mlp(const gp_mlp_params* p, const gp_mlp_input* x, MlpDev& m) {
    if (!p || !x) GP_FAIL("null mlp params/input");
    if (p->width != 256 || p->depth != 4) GP_FAIL("Deformable_Field: only d=4, w=256 is implemented (got d=%d w=%d)", p->depth, p->width);
    if (p->out_dim < 7 || p->out_dim > 8) GP_FAIL("out_dim must be 7 or 8 (got %d)", p->out_dim);
    if (x->feature_dim <= 0 || x->feature_dim % 4 || x->xyz_freq < 0 || x->xyz_freq > 16 || x->time_freq < 0 || x->time_freq > 16)
        GP_FAIL("bad input encoding dims");
    const int in_dim = x->feature_dim + 6 * x->xyz_freq + 2 * x->time_freq;
    if (in_dim != p->in_dim) GP_FAIL("in_dim %d != feature_dim + 6*xyz_freq + 2*time_freq = %d", p->in_dim, in_dim);
    if (in_dim % 4 || in_dim > 256) GP_FAIL("in_dim %d must be a multiple of 4 and <= 256", in_dim);
    if (x->rows < 0) GP_FAIL("negative rows");
    for (int l = 0; l < 5; ++l)
        if (!p->w[l] || !p->b[l]) GP_FAIL("null weight pointer (layer %d)", l);
    if (x->rows > 0 && (!x->feature || (x->xyz_freq > 0 && !x->xyz) || (x->time_freq > 0 && !x->t))) GP_FAIL("null input pointer");
    m.rows = x->rows; m.in_dim = in_dim; m.in_pad = (in_dim + 7) / 8 * 8; m.out_dim = p->out_dim;
    m.feature_dim = x->feature_dim; m.xyz_freq = x->xyz_freq; m.time_freq = x->time_freq;
    for (int l = 0; l < 5; ++l) { m.w[l] = p->w[l]; m.b[l] = p->b[l]; }
    m.feature = x->feature; m.xyz = x->xyz; m.t = x->t;
    m.pk = p->packed;
    if (m.pk && ((uintptr_t)m.pk & 15) != 0) GP_FAIL("gp_mlp_params.packed must be 16-byte aligned");
    if (p->scratch && ((uintptr_t)p->scratch & 15) != 0) GP_FAIL("gp_mlp_params.scratch must be 16-byte aligned");
    return 0;
}

extern "C" int64_t gp_mlp_packed_floats(int32_t in_dim) {
    if (in_dim <= 0 || in_dim > 256) return -1;
    return 4 * (int64_t)mlp_pack_layout(in_dim).total;
}

extern "C" int gp_mlp_pack(const gp_mlp_params* p, float* packed, gp_stream_t stream_) {
    if (!p || !packed) GP_FAIL("null argument");
    if (p->width != 256 || p->depth != 4) GP_FAIL("Deformable_Field: only d=4, w=256 is implemented");
    if (p->in_dim <= 0 || p->in_dim % 4 || p->in_dim > 256) GP_FAIL("in_dim %d must be a multiple of 4 and <= 256", p->in_dim);
    if (((uintptr_t)packed & 15) != 0) GP_FAIL("packed must be 16-byte aligned");
    MlpDev m;
    memset(&m, 0, sizeof(m));
    m.in_dim = p->in_dim;
    for (int l = 0; l < 4; ++l) { if (!p->w[l]) GP_FAIL("null weight pointer (layer %d)", l); m.w[l] = p->w[l]; }
    const long total = mlp_pack_layout(p->in_dim).total;
    GpProfScope _p("mlp_pack", (hipStream_t)stream_);
    hipLaunchKernelGGL(gp_mlp_pack_kernel, dim3(gp_blocks((size_t)total, 256)), dim3(256), 0, (hipStream_t)stream_, m, (float4*)packed);
    GP_LAUNCH_CHECK();
    return 0;
}

// layout of the activation record: [X: rows x in_pad][H1..H4: 4 x rows x 256]
static inline size_t acts_x_floats(const MlpDev& m) { return (size_t)m.rows * m.in_pad; }

#define GP_MLP_SMALL_ROWS 2048
#define GP_MLP_LARGE_ROWS 32768
#define GP_MLP_SPLIT_ROWS 512       // the feature-split kernel's range (deform_mlp_small.hip): <= 32 row tiles x 16 feature tiles
#define GP_MLP_SCRATCH_FLAG_BYTES 8192       // 32 row tiles x 128 B of counters, the error word at byte 4096
#define GP_MLP_SCRATCH_ERR_WORD 1024

extern "C" int64_t gp_mlp_scratch_bytes(int64_t rows) {
    if (rows <= 0 || rows > GP_MLP_SPLIT_ROWS) return 0;
    return GP_MLP_SCRATCH_FLAG_BYTES + (int64_t)4 * rows * 256 * (int64_t)sizeof(float);
}

// the feature-split forward's grid, or 0 where it cannot run: its workgroups wait for each other, so all of them must be resident.
// mode: 1 = XCD-local exchange, validated on this device; 2 = XCD-local, FIRST launch (the caller validates it); 3 = agent-scope fences
static unsigned mlp_split_grid(const MlpDev& m, const void* scratch, int& mode, int*& state_out) {
    mode = 0; state_out = nullptr;
    if (!scratch || m.rows > GP_MLP_SPLIT_ROWS || gp_debug_get(13) == 1) return 0;
    const unsigned rt = (unsigned)((m.rows + 15) / 16), grid = 8u * 16u * ((rt + 7u) / 8u);
    // per device: how many workgroups of the kernel the part holds at once (0: not asked yet), and what the first launch showed
    // (0 not run, 1 XCD-local placement holds, -1 it does not: the 16-row kernels from then on)
    static int resident[32] = {0}, state[32] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 0;
    if (__atomic_load_n(&resident[dev], __ATOMIC_RELAXED) == 0) {
        int per_cu = 0, r = -1;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gp_mlp_fwd_split_small_kernel, 256, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, dev) == hipSuccess && per_cu * prop.multiProcessorCount > 0)
            r = per_cu * prop.multiProcessorCount;
        else (void)hipGetLastError();
        __atomic_store_n(&resident[dev], r, __ATOMIC_RELAXED);
    }
    if (__atomic_load_n(&resident[dev], __ATOMIC_RELAXED) < (int)grid) return 0;
    if (gp_debug_get(13) == 2) { mode = 3; return grid; }
    const int st = __atomic_load_n(&state[dev], __ATOMIC_ACQUIRE);
    if (st < 0) return 0;
    mode = st > 0 ? 1 : 2;
    state_out = &state[dev];
    return grid;
}

extern "C" int gp_mlp_forward(const gp_mlp_params* p, const gp_mlp_input* x, float* out, float* acts, gp_stream_t stream_) {
    MlpDev m;
    if (make_mlp(p, x, m)) return 1;
    if (m.rows == 0) return 0;
    if (!out) GP_FAIL("null output");
    float* sx = acts;
    float* sh = acts ? acts + gp_align_up(acts_x_floats(m), 64) : nullptr;
    { GpProfScope _p("mlp_fwd", (hipStream_t)stream_);
        // few rows (stage 2/3: the keypoints): 16-row workgroups on v_mfma_f32_16x16x4_f32 spread the work over twice the CUs
        int mode = 0;
        int* state = nullptr;
        const unsigned split = mlp_split_grid(m, p->scratch, mode, state);
        if (split) {    // <= 512 rows with a scratch: 16 feature tiles per row tile (the saved activations are the exchange buffer)
            float* hx = sh ? sh : (float*)((char*)p->scratch + GP_MLP_SCRATCH_FLAG_BYTES);
            uint32_t* flags = (uint32_t*)p->scratch;
            hipLaunchKernelGGL(mode == 3 ? gp_mlp_fwd_split_small_agent_kernel : gp_mlp_fwd_split_small_kernel, dim3(split), dim3(256), 0,
                               (hipStream_t)stream_, m, out, sx, hx, flags, flags + GP_MLP_SCRATCH_ERR_WORD);
            if (mode == 2) {
                // The FIRST launch of the XCD-local form on this device is validated before the form is trusted: wait for it, read the
                // scratch's error word (a counter that never filled; a row tile's workgroups on two XCDs).  A bad word: the word is
                // cleared, the device is marked, and this call's result comes from the 16-row kernel -- as every later one's will.
                uint32_t e = 1;
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize((hipStream_t)stream_) != hipSuccess ||
                    hipMemcpy(&e, flags + GP_MLP_SCRATCH_ERR_WORD, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) e = 1;
                if (e != 0) {
                    (void)hipGetLastError();
                    __atomic_store_n(state, -1, __ATOMIC_RELEASE);
                    GP_HIP_CHECK(hipMemsetAsync(flags, 0, GP_MLP_SCRATCH_FLAG_BYTES, (hipStream_t)stream_));
                    hipLaunchKernelGGL(gp_mlp_fwd_small_kernel, dim3(gp_blocks((size_t)m.rows, 16)), dim3(512), 0, (hipStream_t)stream_, m, out, sx, sh);
                } else __atomic_store_n(state, 1, __ATOMIC_RELEASE);
            }
        }
        else if (m.rows <= GP_MLP_SMALL_ROWS)
            hipLaunchKernelGGL(gp_mlp_fwd_small_kernel, dim3(gp_blocks((size_t)m.rows, 16)), dim3(512), 0, (hipStream_t)stream_, m, out, sx, sh);
        else if (m.rows >= GP_MLP_LARGE_ROWS)   // two row tiles per workgroup: half the weight traffic from L2
            hipLaunchKernelGGL(gp_mlp_fwd2_kernel, dim3(gp_blocks((size_t)m.rows, 64)), dim3(512), 0, (hipStream_t)stream_, m, out, sx, sh,
                               acts ? (uint32_t*)(sh + (size_t)4 * m.rows * 256) : (uint32_t*)nullptr);
        else
            hipLaunchKernelGGL(gp_mlp_fwd_kernel, dim3(gp_blocks((size_t)m.rows, 32)), dim3(512), 0, (hipStream_t)stream_, m, out, sx, sh);
    GP_LAUNCH_CHECK(); }
    return 0;
}

// Internal (gp_train_step_run): would gp_mlp_backward take the feature-split data kernel for these arguments?  Only that kernel honours
// a request to ADD the input-feature gradient into dL_dfeature (gp_mlp_backward_accumulate_dfeature_once: the keypoint features also take
// the regulariser's gradient, which the loss kernel has already written there -- one launch less per step than "=" into a temporary + add).
static thread_local int g_dfeature_accumulate = 0;
bool gp_mlp_backward_splits(const gp_mlp_params* p, int64_t rows) {
    if (!p) return false;
    MlpDev m;
    memset(&m, 0, sizeof(m));
    m.rows = rows;
    int mode = 0;
    int* st = nullptr;
    return mlp_split_grid(m, p->scratch, mode, st) != 0 && (mode == 1 || mode == 3);
}
void gp_mlp_backward_accumulate_dfeature_once() { g_dfeature_accumulate = 1; }

extern "C" int gp_mlp_backward(const gp_mlp_params* p, const gp_mlp_input* x, const float* acts, const float* dL_dout,
                               gp_mlp_grads* g, float* dL_dfeature, float* dL_dxyz, gp_alloc_fn alloc, void* alloc_ctx,
                               gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    struct ClearAcc { ~ClearAcc() { g_dfeature_accumulate = 0; } } clear_acc;      // (a "+=" request is for THIS call, however it returns)
    MlpDev m;
    if (make_mlp(p, x, m)) return 1;
    if (m.rows == 0) return 0;
    if (!acts || !dL_dout || !g || !alloc) GP_FAIL("null argument");
    for (int l = 0; l < 5; ++l)
        if (!g->dw[l] || !g->db[l]) GP_FAIL("null weight-grad pointer (layer %d)", l);
    const float* sx = acts;
    const float* sh = acts + gp_align_up(acts_x_floats(m), 64);
    const size_t dz_bytes = gp_align_up((size_t)4 * m.rows * 256 * sizeof(float), 256);
    float* dz = (float*)alloc(alloc_ctx, GP_BUF_TEMP, dz_bytes);
    if (!dz) GP_FAIL("allocator returned NULL for TEMP (%zu B)", dz_bytes);
    GpAdamRider* rider = gp_adam_rider_slot();
    int split_mode = 0;
    int* split_state = nullptr;
    // the feature-split form (deform_mlp_small.hip) once a forward has validated its XCD-local exchange on this device (mode 1), or in
    // the agent-scope form (mode 3)
    const unsigned split = mlp_split_grid(m, p->scratch, split_mode, split_state);
    if (split && (split_mode == 1 || split_mode == 3)) {
        float* gx = (float*)((char*)p->scratch + GP_MLP_SCRATCH_FLAG_BYTES);
        uint32_t* flags = (uint32_t*)p->scratch;
        // (an armed rider stays armed: beside this latency-chained kernel the optimizer's stream costs more than it hides -- it is
        // launched behind the MLP backward by its owner, gp_adam_rider_flush)
        GpProfScope _p("mlp_bwd_data", s);
        const int acc = g_dfeature_accumulate;
        g_dfeature_accumulate = 0;
        hipLaunchKernelGGL(gp_mlp_bwd_data_split_small_kernel, dim3(split), dim3(256), 0, s, m, sh, dL_dout, dz, dL_dfeature, dL_dxyz, gx, flags,
                           flags + GP_MLP_SCRATCH_ERR_WORD, (split_mode == 3 ? 1 : 0) | (acc ? 2 : 0));
        GP_LAUNCH_CHECK();
    } else if (m.rows <= GP_MLP_SMALL_ROWS && rider->armed) {
        // gp_train_step_run left an optimizer launch that needs nothing of this backward: its chunks ride in the data kernel's launch
        // (deform_mlp_small.hip).  The scope carries the optimizer's name: its bytes are what the launch moves.
        rider->armed = false;
        const unsigned n_mlp = gp_blocks((size_t)m.rows, 16);
        GpProfScope _p("adam", s);
        hipLaunchKernelGGL(gp_mlp_bwd_data_small_adam_kernel, dim3(n_mlp + rider->chunks), dim3(512), 0, s, m, sh, dL_dout, dz, dL_dfeature,
                           dL_dxyz, n_mlp, rider->t, rider->b1, rider->b2, rider->eps, rider->zero_grad, rider->skip_flag);
        GP_LAUNCH_CHECK();
    } else { GpProfScope _p("mlp_bwd_data", s);
        if (m.rows <= GP_MLP_SMALL_ROWS)
            hipLaunchKernelGGL(gp_mlp_bwd_data_small_kernel, dim3(gp_blocks((size_t)m.rows, 16)), dim3(512), 0, s, m, sh, dL_dout, dz,
                               dL_dfeature, dL_dxyz);
        else if (m.rows >= GP_MLP_LARGE_ROWS)
            hipLaunchKernelGGL(gp_mlp_bwd_data2_kernel, dim3(gp_blocks((size_t)m.rows, 64)), dim3(512), 0, s, m,
                               (const uint32_t*)(sh + (size_t)4 * m.rows * 256), dL_dout, dz,
                               dL_dfeature, dL_dxyz);
        else
            hipLaunchKernelGGL(gp_mlp_bwd_data_kernel, dim3(gp_blocks((size_t)m.rows, 32)), dim3(512), 0, s, m, sh, dL_dout, dz,
                       dL_dfeature, dL_dxyz);
    GP_LAUNCH_CHECK(); }
    // weight grads: dW_l = dZ_{l+1}^T H_l,  H_0 = X
    // row blocks: one for small K (no atomics), up to 16 for large row counts
    long nrb_l = (m.rows + 4095) / 4096;
    if (nrb_l > 16) nrb_l = 16;
    if (nrb_l < 1) nrb_l = 1;
    long rpb = ((m.rows + nrb_l - 1) / nrb_l + 15) & ~15L;
    const unsigned nrb = (unsigned)((m.rows + rpb - 1) / rpb);
    GpProfScope _pw("mlp_bwd_weight", s);
    if (nrb == 1) {   // one launch for all five layers
        MlpWeightJobs jobs;
        for (int l = 0; l < 5; ++l) {
            jobs.dZ[l] = l < 4 ? dz + (size_t)l * m.rows * 256 : dL_dout;
            jobs.n_out[l] = l < 4 ? 256 : m.out_dim;
            jobs.H[l] = l == 0 ? sx : sh + (size_t)(l - 1) * m.rows * 256;
            jobs.ldh[l] = l == 0 ? m.in_pad : 256;
            jobs.n_in[l] = l == 0 ? m.in_dim : 256;
            jobs.dW[l] = g->dw[l]; jobs.db[l] = g->db[l];
        }
        hipLaunchKernelGGL(gp_mlp_bwd_weight5_kernel, dim3(5, 8, 8), dim3(512), 0, s, jobs, m.rows);
        GP_LAUNCH_CHECK();
        return 0;
    }
    if (m.rows >= 16384) {   // large pass: 64 x 64 blocks, 4096-row slabs
        long nb = (m.rows + 4095) / 4096;                 // 4096-row slabs, but at least 64 row blocks (>= 1024 workgroups)
        if (nb < 64) nb = 64;
        const long brpb = ((m.rows + nb - 1) / nb + 7) & ~7L;
        const unsigned bnrb = (unsigned)((m.rows + brpb - 1) / brpb);
        for (int l = 0; l < 5; ++l) {
            const float* dZl = l < 4 ? dz + (size_t)l * m.rows * 256 : dL_dout;
            const int n_out = l < 4 ? 256 : m.out_dim;
            const float* H = l == 0 ? sx : sh + (size_t)(l - 1) * m.rows * 256;
            const int ldh = l == 0 ? m.in_pad : 256;
            const int n_in = l == 0 ? m.in_dim : 256;
            const unsigned ntp = (unsigned)((n_in + 63) / 64) * (unsigned)((n_out + 63) / 64);
            hipLaunchKernelGGL(gp_mlp_bwd_weight64_kernel, dim3(ntp * ((bnrb + 7u) & ~7u)), dim3(256), 0, s, dZl, n_out, H, ldh, n_in,
                               m.rows, brpb, bnrb, g->dw[l], n_in, g->db[l]);
            GP_LAUNCH_CHECK();
        }
        return 0;
    }
    for (int l = 0; l < 5; ++l) {
        const float* dZl = l < 4 ? dz + (size_t)l * m.rows * 256 : dL_dout;
        const int n_out = l < 4 ? 256 : m.out_dim;
        const float* H = l == 0 ? sx : sh + (size_t)(l - 1) * m.rows * 256;
        const int ldh = l == 0 ? m.in_pad : 256;
        const int n_in = l == 0 ? m.in_dim : 256;
        {
        hipLaunchKernelGGL(gp_mlp_bwd_weight_kernel, dim3(nrb, (unsigned)((n_in + 31) / 32), (unsigned)((n_out + 31) / 32)),
                           dim3(512), 0, s, dZl, n_out, H, ldh, n_in, m.rows, rpb, g->dw[l], n_in, g->db[l]);
        GP_LAUNCH_CHECK(); }
    }
    return 0;
}

static int make_blend(const gp_blend_args* a, BlendDev& b) {
    if (!a) GP_FAIL("null blend args");
    if (a->num_gaussians < 0) GP_FAIL("negative num_gaussians");
    if (a->out_dim < 7 || a->out_dim > 8) GP_FAIL("out_dim must be 7 or 8");
    if (a->nearest_num < 0 || a->nearest_num > 16) GP_FAIL("nearest_num %d unsupported (0..16)", a->nearest_num);
    if (a->nearest_num > 0 && (a->num_keypoints <= 0 || !a->raw_w || !a->knn_idx)) GP_FAIL("stage-2 blend needs keypoints, raw_w and knn_idx");
    if (a->nearest_num > 0 && a->num_keypoints * 7 * sizeof(float) > 60000) GP_FAIL("too many keypoints (%ld) for the LDS accumulator", (long)a->num_keypoints);
    if (a->num_gaussians > 0 && (!a->delta || !a->xyz || !a->rot)) GP_FAIL("null blend input");
    if (a->knn_idx16 && (a->num_keypoints > 65535 || ((uintptr_t)a->knn_idx16 & 3) != 0)) GP_FAIL("knn_idx16 needs K < 65536 and 4-byte alignment");
    b.N = a->num_gaussians; b.K = a->num_keypoints; b.nn = a->nearest_num; b.out_dim = a->out_dim;
    b.norm_rotation = a->norm_rotation; b.delta = a->delta; b.raw_w = a->raw_w; b.knn = a->knn_idx; b.xyz = a->xyz; b.rot = a->rot; b.knn16 = a->knn_idx16;
    return 0;
}

extern "C" int gp_blend_forward(const gp_blend_args* a, float* xyz_t, float* q_t, gp_stream_t stream_) {
    BlendDev b;
    if (make_blend(a, b)) return 1;
    if (b.N == 0) return 0;
    if (!xyz_t || !q_t) GP_FAIL("null output");
    { GpProfScope _p("blend_fwd", (hipStream_t)stream_);
        const bool al = (((uintptr_t)b.raw_w | (uintptr_t)b.knn) & 15) == 0;      // the fixed-nn kernels use 16-byte row loads
        const bool i16 = b.knn16 != nullptr;
        hipLaunchKernelGGL(b.nn == 6 && al ? (i16 ? gp_blend_fwd6_i16_kernel : gp_blend_fwd6_kernel)
                           : b.nn == 8 && al ? (i16 ? gp_blend_fwd8_i16_kernel : gp_blend_fwd8_kernel) : gp_blend_fwd_kernel,
                       dim3(gp_blocks((size_t)b.N, 256)), dim3(256), 0, (hipStream_t)stream_, b, xyz_t, q_t);
    GP_LAUNCH_CHECK(); }
    return 0;
}

extern "C" int gp_blend_backward(const gp_blend_args* a, const float* dL_dxyz_t, const float* dL_dq_t, float* dL_ddelta,
                                 float* dL_draw_w, float* dL_dxyz, float* dL_drot, gp_alloc_fn alloc, void* alloc_ctx,
                                 gp_stream_t stream_) {
    BlendDev b;
    if (make_blend(a, b)) return 1;
    if (b.N == 0) return 0;
    if (!dL_dxyz_t || !dL_dq_t || !dL_ddelta || !dL_dxyz || !dL_drot) GP_FAIL("null argument");     // dL_draw_w may be NULL
    unsigned blocks = gp_blocks((size_t)b.N, 256);
    if (b.nn > 0 && blocks > 1024) blocks = 1024;   // persistent: four workgroups per CU (LDS), their partials are summed by the reduce kernel
    // acc[K*7] | delta[K*od] | cnt[K] | base[K+1] | g[7*256] | inv[K] | w[256*2*nn] | sorted u16 [256*nn]
    const size_t lds = b.nn > 0 ? ((size_t)b.K * (7 + b.out_dim + 3) + 1 + 256 * 7 + 256 * 2 * (size_t)b.nn) * 4 + 256 * (size_t)b.nn * 2 + 16 : 256 * 8 * 4;
    if (lds > 64 * 1024) GP_FAIL("keypoint blend backward: K = %ld, nn = %d needs %zu B of LDS (> 64 KiB)", (long)b.K, b.nn, lds);
    float* partial = nullptr;
    const int KA = (int)b.K * 7;
    if (b.nn > 0) {
        if (!alloc) GP_FAIL("null allocator");
        partial = (float*)alloc(alloc_ctx, GP_BUF_TEMP, gp_align_up((size_t)blocks * KA * sizeof(float), 256));
        if (!partial) GP_FAIL("allocator returned NULL for TEMP");
    }
    { GpProfScope _p("blend_bwd", (hipStream_t)stream_);
    const bool al = (((uintptr_t)b.raw_w | (uintptr_t)b.knn | (uintptr_t)dL_draw_w) & 15) == 0;
    const bool i16 = b.knn16 != nullptr;
    hipLaunchKernelGGL(b.nn == 6 && al ? (i16 ? gp_blend_bwd6_i16_kernel : gp_blend_bwd6_kernel)
                       : b.nn == 8 && al ? (i16 ? gp_blend_bwd8_i16_kernel : gp_blend_bwd8_kernel) : gp_blend_bwd_kernel,
                       dim3(blocks), dim3(256), lds, (hipStream_t)stream_, b, dL_dxyz_t, dL_dq_t, dL_ddelta,
                       dL_draw_w, dL_dxyz, dL_drot, partial);
    GP_LAUNCH_CHECK();
    if (b.nn > 0) {
        hipLaunchKernelGGL(gp_blend_bwd_reduce_kernel, dim3(gp_blocks((size_t)KA, 64)), dim3(1024), 0, (hipStream_t)stream_, partial,
                           (int)blocks, KA, b.out_dim, dL_ddelta);
        GP_LAUNCH_CHECK();
    } }
    return 0;
}

extern "C" int gp_activations_forward(int64_t n, const float* scaling_raw, const float* opacity_raw, const float* delta_o,
                                      int32_t stride, float beta, float* scale, float* opacity, gp_stream_t stream_) {
    if (n < 0) GP_FAIL("negative n");
    if (n == 0) return 0;
    if (!scaling_raw || !opacity_raw || !scale || !opacity) GP_FAIL("null argument");
    if (delta_o && (stride <= 0 || !(beta > 0.f))) GP_FAIL("bad delta_o stride/beta");
    hipLaunchKernelGGL(gp_act_fwd_kernel, dim3(gp_blocks((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream_, (long)n, scaling_raw,
                       opacity_raw, delta_o, stride, beta, scale, opacity);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_activations_backward(int64_t n, const float* scaling_raw, const float* opacity_raw, const float* delta_o,
                                       int32_t stride, float beta, const float* dL_dscale, const float* dL_dopacity,
                                       float* dL_dscaling_raw, float* dL_dopacity_raw, float* dL_ddelta_o, gp_stream_t stream_) {
    if (n < 0) GP_FAIL("negative n");
    if (n == 0) return 0;
    if (!scaling_raw || !opacity_raw) GP_FAIL("null argument");
    hipLaunchKernelGGL(gp_act_bwd_kernel, dim3(gp_blocks((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream_, (long)n, scaling_raw,
                       opacity_raw, delta_o, stride, beta, dL_dscale, dL_dopacity, dL_dscaling_raw, dL_dopacity_raw, dL_ddelta_o);
    GP_LAUNCH_CHECK();
    return 0;
}

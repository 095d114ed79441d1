// gp_train_step.hip -- gp_train_step_run (include/gp_hip.h): one training iteration of the hot path enqueued by one call.
// It CALLS the library's own entry points in the order the autograd graph of gaussianprediction_amd/train_step.py runs them:
// nothing is re-implemented here, so the fused step and the drop-in surfaces cannot drift apart.
// [REF train.py:101-133, 196-197; scene/gaussian_model.py:251-273]
#include "gp_common.h"
#include "loss_adam_kernels.h"

bool gp_mlp_backward_splits(const gp_mlp_params* p, int64_t rows);     // gp_capi_deform.hip (internal)
void gp_mlp_backward_accumulate_dfeature_once();

// dst[i] += src[i]: the keypoint features take a gradient from the regulariser AND from the MLP's input (both "=" producers)
__global__ __launch_bounds__(256) void gp_step_accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// gp_adam_step_multi[_steps] over the tensors of `u`'s tables selected by `sel` -- launched on `stream`, or (ride) left in the rider
// slot for the keypoint MLP's data backward to carry (loss_adam_kernels.h)
static int step_adam(const gp_step_update* u, uint32_t sel, gp_stream_t stream, bool ride = false) {
    float *P[32], *G[32], *M[32], *V[32];
    int64_t NUM[32], ST[32];
    float LR[32];
    int n = 0;
    uint32_t keep = 0;
    for (int k = 0; k < u->adam_count && k < 32; ++k) {
        if (!((sel >> k) & 1u)) continue;
        P[n] = u->adam_params[k]; G[n] = u->adam_grads[k]; M[n] = u->adam_exp_avgs[k]; V[n] = u->adam_exp_avg_sqs[k];
        NUM[n] = u->adam_numels[k]; LR[n] = u->adam_lrs[k];
        ST[n] = u->adam_steps ? u->adam_steps[k] : u->step;
        if ((u->keep_grad_mask >> k) & 1u) keep |= 1u << n;
        ++n;
    }
    if (n == 0) return 0;
    if (ride) return gp_adam_rider_arm(n, P, G, M, V, NUM, LR, ST, u->beta1, u->beta2, u->eps, 1, keep, u->skip_flag);
    return gp_adam_step_multi_steps(n, P, G, M, V, NUM, LR, ST, u->beta1, u->beta2, u->eps, 1, keep, u->skip_flag, stream);
}

extern "C" int gp_train_step_run(const gp_step_plan* p, const gp_step_view* v, const gp_step_update* u, gp_alloc_fn alloc,
                                 void* alloc_ctx, gp_stream_t stream) {
    if (!p || !v || !u || !alloc) GP_FAIL("gp_train_step_run: null argument");
    const int64_t N = p->num_gaussians, K = p->num_keypoints;
    if (N <= 0 || K <= 0 || p->nearest_num <= 0) GP_FAIL("gp_train_step_run: needs Gaussians, keypoints and neighbours (the stage-3 form)");
    if (u->binning_capacity <= 0 || !u->binning_status) GP_FAIL("gp_train_step_run: capacity mode only (binning_capacity > 0 with a status word)");
    if (!p->xyz || !p->scaling || !p->rotation || !p->opacity || !p->features_dc || !p->features_rest || !p->keypoints ||
        !p->keypoint_features || !p->raw_w || !p->knn_idx)
        GP_FAIL("gp_train_step_run: null parameter pointer");
    if (!p->g_xyz || !p->g_scaling || !p->g_rotation || !p->g_opacity || !p->g_keypoints || !p->g_keypoint_features)
        GP_FAIL("gp_train_step_run: null gradient pointer");
    if (!u->adam_shs && (!p->g_features_dc || !p->g_features_rest)) GP_FAIL("gp_train_step_run: SH gradient buffers or adam_shs needed");
    if (!p->delta || !p->acts || !p->xyz_t || !p->q_t || !p->scale || !p->opacity_t || !p->loss_sums || !p->loss || !p->g_xyz_t || !p->g_q_t || !p->g_scale || !p->g_opacity_t || !p->g_means2D || !p->g_delta || !p->g_feature_tmp)
        GP_FAIL("gp_train_step_run: null intermediate buffer");
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->campos || !v->gt_image || !v->time) GP_FAIL("gp_train_step_run: null view pointer");
    // the optimizer's table is checked BEFORE anything is enqueued: a refusal behind the backward would leave the gradients written and the
    // SH coefficients (updated inside the rasterizer backward) already stepped
    if (u->adam_count < 0 || u->adam_count > 32) GP_FAIL("gp_train_step_run: 0..32 optimizer tensors (got %d)", u->adam_count);
    if (u->adam_count > 0 && (!u->adam_params || !u->adam_grads || !u->adam_exp_avgs || !u->adam_exp_avg_sqs || !u->adam_numels || !u->adam_lrs))
        GP_FAIL("gp_train_step_run: null optimizer table");
    const int H = p->image_height, W = p->image_width, od = p->mlp.out_dim;
    const bool reg = p->reg_scale != 0.f;
    const int64_t nfeat = K * (int64_t)p->feature_dim;
    if (reg && nfeat > 65536) GP_FAIL("gp_train_step_run: the folded regulariser takes at most 65536 feature elements");

    // ---- GaussianModel.forward, stage 3 [REF scene/gaussian_model.py:251-273, 285-286]
    gp_mlp_input mi;
    mi.rows = K; mi.feature_dim = p->feature_dim; mi.xyz_freq = p->xyz_freq; mi.time_freq = p->time_freq;
    mi.feature = p->keypoint_features; mi.xyz = p->keypoints; mi.t = v->time;
    if (gp_mlp_forward(&p->mlp, &mi, p->delta, p->acts, stream)) return 1;
    gp_blend_args ba;
    ba.num_gaussians = N; ba.num_keypoints = K; ba.nearest_num = p->nearest_num; ba.out_dim = od; ba.norm_rotation = p->norm_rotation;
    ba.delta = p->delta; ba.raw_w = p->raw_w; ba.knn_idx = p->knn_idx; ba.xyz = p->xyz; ba.rot = p->rotation; ba.knn_idx16 = p->knn_idx16;
    if (gp_blend_forward(&ba, p->xyz_t, p->q_t, stream)) return 1;
    // (exp / sigmoid of the raw scales and opacities [REF scene/gaussian_model.py get_scaling, get_opacity] run INSIDE the projection
    // kernel and its backward -- gp_raster_settings.raw_activations, the expressions of gp_activations_forward / _backward bit for bit:
    // two launches and 64 B per Gaussian less per step; gp_debug_option(14, 1): the launches of their own, for A/B)
    const bool raw_act = gp_debug_get(14) == 0;
    if (!raw_act && gp_activations_forward(N, p->scaling, p->opacity, nullptr, 0, 1.f, p->scale, p->opacity_t, stream)) return 1;

    // ---- GaussianRasterizer [REF gaussian_renderer/__init__.py:37-52, 98-106]
    gp_raster_settings st;
    memset(&st, 0, sizeof(st));
    st.image_height = H; st.image_width = W; st.tanfovx = v->tanfovx; st.tanfovy = v->tanfovy; st.scale_modifier = 1.f;
    st.sh_degree = p->sh_degree; st.sh_coeffs = 16;
    st.bg = v->bg; st.viewmatrix = v->viewmatrix; st.projmatrix = v->projmatrix; st.campos = v->campos;
    st.binning_capacity = u->binning_capacity; st.binning_status = u->binning_status; st.sh_ready_event = u->sh_ready_event;
    st.depth_key_bits = u->depth_key_bits; st.depth_key_base = u->depth_key_base;
    st.raw_activations = raw_act ? 1 : 0;
    gp_raster_inputs in;
    memset(&in, 0, sizeof(in));
    in.num_gaussians = N; in.means3D = p->xyz_t; in.shs = p->features_dc; in.shs_rest = p->features_rest;
    in.opacities = raw_act ? p->opacity : p->opacity_t;
    in.scales = raw_act ? p->scaling : p->scale; in.rotations = p->q_t;
    gp_raster_outputs out = p->out;
    gp_raster_saved saved;
    memset(&saved, 0, sizeof(saved));
    if (gp_raster_forward(&st, &in, &out, &saved, alloc, alloc_ctx, stream)) return 1;
    alloc(alloc_ctx, GP_BUF_TEMP_DONE, 0);
    if (u->hook) u->hook(u->hook_ctx, GP_STEP_AFTER_FORWARD);

    // ---- loss [REF train.py:105-109, utils/loss_utils.py:54-100] and its image gradient: one kernel (gp_loss_l1_ssim_fused: the SSIM
    // derivative maps stay in LDS).  The pair-of-kernels form keeps its plan field: there the maps live from the loss forward to the loss
    // backward only -- with NULL in the plan they are a TEMP buffer the allocator carves out of memory the rasterizer's sort just used.
    // The image gradient is read by the composite backward WHILE its accumulators are written: it stays a buffer of its own.
    float* dimg = p->dL_dimage;
    if (!dimg) GP_FAIL("gp_train_step_run: null intermediate buffer");
    if (gp_debug_get(11) == 0) {
        // one launch: the sums of the loss and its image gradient (round 6; gp_debug_option(11, 1): the pair of kernels, for A/B)
        if (gp_loss_l1_ssim_fused(out.color, v->gt_image, 3, H, W, p->lambda_dssim, nullptr, p->loss_sums, dimg, reg ? p->keypoint_features : nullptr,
                                  nfeat, p->reg_scale, reg ? p->g_keypoint_features : nullptr, stream)) return 1;
        if (reg) { if (gp_loss_l1_ssim_finalize_reg(p->loss_sums, 3, H, W, p->lambda_dssim, p->keypoint_features, nfeat, p->reg_scale, p->loss, stream)) return 1; }
        else if (gp_loss_l1_ssim_finalize(p->loss_sums, 3, H, W, p->lambda_dssim, p->loss, stream)) return 1;
    } else {
        float* dmaps = p->dmaps;
        if (!dmaps) {
            dmaps = (float*)alloc(alloc_ctx, GP_BUF_TEMP, gp_align_up((size_t)9 * H * W * 4, 256));
            if (!dmaps) GP_FAIL("gp_train_step_run: allocator returned NULL for the SSIM derivative maps");
        }
        if (gp_loss_l1_ssim_forward(out.color, v->gt_image, 3, H, W, p->loss_sums, dmaps, stream)) return 1;
        if (reg) {
            if (gp_loss_l1_ssim_finalize_reg(p->loss_sums, 3, H, W, p->lambda_dssim, p->keypoint_features, nfeat, p->reg_scale, p->loss, stream)) return 1;
            if (gp_loss_l1_ssim_backward_reg(out.color, v->gt_image, dmaps, 3, H, W, p->lambda_dssim, nullptr, dimg,
                                             p->keypoint_features, nfeat, p->reg_scale, p->g_keypoint_features, stream)) return 1;
        } else {
            if (gp_loss_l1_ssim_finalize(p->loss_sums, 3, H, W, p->lambda_dssim, p->loss, stream)) return 1;
            if (gp_loss_l1_ssim_backward(out.color, v->gt_image, dmaps, 3, H, W, p->lambda_dssim, nullptr, dimg, stream)) return 1;
        }
        if (!p->dmaps) alloc(alloc_ctx, GP_BUF_TEMP_DONE, 0);
    }

    // ---- backward, in the order autograd runs it
    gp_raster_grads g;
    memset(&g, 0, sizeof(g));
    g.dL_dmeans3D = p->g_xyz_t; g.dL_dmeans2D = p->g_means2D; g.dL_dshs = p->g_features_dc; g.dL_dshs_rest = p->g_features_rest;
    g.dL_dopacities = raw_act ? p->g_opacity : p->g_opacity_t; g.dL_dscales = raw_act ? p->g_scaling : p->g_scale; g.dL_drotations = p->g_q_t; g.accumulate_shs = 0; g.adam_shs = u->adam_shs;
    if (gp_raster_backward(&st, &in, &out, &saved, dimg, nullptr, &g, alloc, alloc_ctx, stream)) return 1;
    alloc(alloc_ctx, GP_BUF_TEMP_DONE, 0);
    if (u->hook) u->hook(u->hook_ctx, GP_STEP_AFTER_RASTER_BACKWARD);
    if (!raw_act && gp_activations_backward(N, p->scaling, p->opacity, nullptr, 0, 1.f, p->g_scale, p->g_opacity_t, p->g_scaling, p->g_opacity, nullptr, stream))
        return 1;
    if (gp_blend_backward(&ba, p->g_xyz_t, p->g_q_t, p->g_delta, nullptr, p->g_xyz, p->g_rotation, alloc, alloc_ctx, stream)) return 1;
    alloc(alloc_ctx, GP_BUF_TEMP_DONE, 0);
    // ---- optimizer, first part [REF train.py:196-197]: the per-Gaussian tensors' gradients are final here.  Their update (HBM-bound,
    // every CU) needs nothing the keypoint MLP's backward (latency-bound, 16 CUs) produces and touches none of its tensors: the tensors of
    // adam_early_mask RIDE in the launch of that backward's data kernel (the rider of loss_adam_kernels.h; a row count the small-row
    // kernels do not serve leaves the rider unconsumed: it is then launched right behind the MLP backward).
    const uint32_t all = u->adam_count >= 32 ? 0xFFFFFFFFu : ((1u << u->adam_count) - 1u);
    const uint32_t early = (u->hook || u->adam_count <= 0) ? 0u : (u->adam_early_mask & all);
    if (early && step_adam(u, early, stream, true)) return 1;
    gp_mlp_grads mg = p->g_mlp;
    // (the keypoint features take a gradient from the regulariser -- written by the loss kernel -- AND from the MLP's input: the
    // feature-split data backward adds its part in place; the 16-row form writes a temporary that a launch of its own adds)
    const bool add_in_place = reg && gp_mlp_backward_splits(&p->mlp, K);
    if (add_in_place) gp_mlp_backward_accumulate_dfeature_once();
    const int rc_mlp = gp_mlp_backward(&p->mlp, &mi, p->acts, p->g_delta, &mg, (reg && !add_in_place) ? p->g_feature_tmp : p->g_keypoint_features,
                                       p->g_keypoints, alloc, alloc_ctx, stream);
    if (rc_mlp) { gp_adam_rider_slot()->armed = false; return 1; }
    if (gp_adam_rider_flush((hipStream_t)stream)) return 1;
    if (reg && !add_in_place) {
        hipLaunchKernelGGL(gp_step_accumulate_kernel, dim3(gp_blocks((size_t)nfeat, 256)), dim3(256), 0, (hipStream_t)stream,
                           p->g_keypoint_features, (const float*)p->g_feature_tmp, nfeat);
        GP_LAUNCH_CHECK();
    }
    if (u->hook) u->hook(u->hook_ctx, GP_STEP_AFTER_BACKWARD);

    // ---- optimizer [REF train.py:196-197, scene/gaussian_model.py:472]
    if (u->adam_count > 0 && step_adam(u, all & ~early, stream)) return 1;
    return 0;
}

/*
 * gp_hip.h -- C ABI of libgp_hip.so, the MI355X (gfx950) implementation of the dynamic-Gaussian
 * render hot path of BoMingZhao/GaussianPrediction.
 *
 * Drop-in boundary.  The reference binds this path through the pybind module
 * `diff_gaussian_rasterization._C` of an (absent, un-vendored) CUDA submodule
 * [/root/reference/.gitmodules:4-6]; its Python-visible contract is the only thing in the tree:
 *     GaussianRasterizationSettings(...) kwargs      gaussian_renderer/__init__.py:37-50
 *     GaussianRasterizer(...)(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
 *                             cov3D_precomp) -> (image, radii, depth, tidx)
 *                                                    gaussian_renderer/__init__.py:98-106
 *     GaussianModel.forward(t, it) -> xyz,q,s,o      scene/gaussian_model.py:231-304
 *     Deformable_Field.forward(x)                    scene/deformable_field.py:112-127
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions: plain device pointers + sizes, no torch types.  Every function enqueues on the
 * hipStream_t it is given (the caller's current stream) and returns 0 on success; on failure it
 * returns non-zero, nothing is thrown across the ABI, and gp_last_error() holds the message.
 * All float tensors are fp32, dense, row-major.  Matrices are the reference's row-vector
 * (transposed) 4x4s [scene/cameras.py:59-61], i.e. column-major in memory.
 */
#ifndef GP_HIP_H
#define GP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gp_stream_t; /* hipStream_t */

/* scratch classes handed to the allocator callback */
enum { GP_BUF_GEOM = 0, GP_BUF_BINNING = 1, GP_BUF_IMAGE = 2, GP_BUF_TEMP = 3,
       GP_BUF_TEMP_DONE = 4 /* not a request (bytes = 0, the return value is ignored): gp_train_step_run tells the allocator that the last
                             * kernel touching the TEMP buffers handed out so far has been enqueued -- an allocator that recycles TEMP
                             * memory in stream order may hand the same memory out again (the step's working set stays what the
                             * separate entry points' is) */ };

/* Allocator callback: return a device pointer to `bytes` bytes (256-B aligned) that stays valid
 * until the caller frees it; `which` is one of GP_BUF_*.  The Python host backs this with torch's
 * caching allocator and keeps GEOM/BINNING/IMAGE alive for backward (ctx.save_for_backward). */
typedef void* (*gp_alloc_fn)(void* ctx, int which, size_t bytes);

/* mirrors GaussianRasterizationSettings [REF gaussian_renderer/__init__.py:37-50] */
typedef struct gp_raster_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;   /* active degree 0..3 */
    int32_t sh_coeffs;   /* coefficients per channel present in `shs` (16 for max degree 3) */
    int32_t prefiltered; /* accepted for API parity; unused (as in the public rasterizer) */
    int32_t debug;
    const float* bg;         /* device [3] */
    const float* viewmatrix; /* device [16] */
    const float* projmatrix; /* device [16] */
    const float* campos;     /* device [3] */
    /* Binning capacity.  0 = exact mode: gp_raster_forward reads R (the number of tile-splat instances) back to size the
     * binning buffers -- the call's one host synchronisation.  > 0 = the caller promises R <= binning_capacity (e.g. a
     * high-water mark of earlier frames): buffers are sized by the capacity, the sort is padded with sentinel keys and the
     * host never waits.  `binning_status` (device, 2 words, required when capacity > 0) receives {R, overflow}: on overflow
     * (R > capacity) the instance lists are truncated -- every access stays in bounds, the image is NOT valid -- and the
     * caller must discard the frame (gp_adam_step_multi takes the same word as its skip flag) and retry with more room. */
    int64_t binning_capacity;
    uint32_t* binning_status;
    /* Optional hipEvent_t.  NULL: `shs` / `shs_rest` are read by the first kernel of the call (fused projection + SH -> RGB).
     * Non-NULL: another stream may still be WRITING the SH tensors when gp_raster_forward is called (the asynchronous all-gather
     * of the updated coefficients in view-parallel training -- 3/4 of all parameter bytes); projection, both sorts and the
     * binning do not read them, so the call makes `stream` wait for this event only in front of a separate SH -> RGB kernel
     * placed right before the composite.  Identical results (one shared device function). */
    void* sh_ready_event;
    /* Depth-key speculation (round 5).  The depth sort orders the 32-bit patterns of the view-space depths (positive floats) in four
     * 8-bit passes.  The visible Gaussians of a frame usually span less than a factor of four in depth, i.e. their keys (two octaves
     * of floats = 2^24 consecutive patterns) fit 24 bits above the smallest one -- which the library cannot know without reading the
     * keys back.  depth_key_bits = B (8 .. 31): the CALLER promises that every visible Gaussian's key k satisfies
     * 0 <= k - depth_key_base < 2^B; the sort then orders k - depth_key_base on its low B bits (24 bits: three passes instead of four;
     * the order of the visible Gaussians is the full sort's, bit for bit -- the subtraction is monotonic).  The projection kernel checks the
     * promise for every visible Gaussian and raises binning_status[1] when it is broken: the frame is invalid exactly as after a binning
     * overflow.  binning_status is required then and has THREE words: [2] is scratch of the library (it receives the number of the call
     * whose promise broke; nothing needs clearing between frames).  0 (or 32) = sort all 32 bits.
     * depth_key_range (optional, device, 2 words): receives {min, max} of the visible Gaussians' keys ({0xFFFFFFFF, 0} if none) --
     * what a caller sizes B and the prefix from (TrainStep: during its exact-mode set-up steps, with a margin). */
    int32_t depth_key_bits;
    uint32_t depth_key_base;
    uint32_t* depth_key_range;
    /* != 0 (round 6): `scales` are the model's LOG-scales and `opacities` its LOGITS [REF scene/gaussian_model.py: get_scaling = exp(_scaling),
     * get_opacity = sigmoid(_opacity)] -- the projection kernel applies exp / sigmoid itself, and gp_raster_backward returns dL_dscales /
     * dL_dopacities with respect to the RAW values (g exp(s); g sigma (1 - sigma)): the same expressions as gp_activations_forward /
     * _backward without a lifecycle term, bit for bit, minus two launches and 64 B per Gaussian of traffic.  Needs scales + rotations
     * (not cov3D_precomp). */
    int32_t raw_activations;
    int32_t reserved0;
} gp_raster_settings;

/* inputs of GaussianRasterizer.forward [REF gaussian_renderer/__init__.py:98-106] */
typedef struct gp_raster_inputs {
    int64_t num_gaussians;
    const float* means3D;        /* [N,3] */
    const float* shs;            /* [N,sh_coeffs,3] or NULL; with shs_rest: features_dc [N,1,3] */
    const float* shs_rest;       /* NULL, or features_rest [N,sh_coeffs-1,3]: the model's two SH tensors
                                    [REF scene/gaussian_model.py:155-159] passed without the per-frame cat */
    const float* colors_precomp; /* [N,3] or NULL (exactly one of shs/colors_precomp) */
    const float* opacities;      /* [N,1] */
    const float* scales;         /* [N,3] or NULL */
    const float* rotations;      /* [N,4] or NULL */
    const float* cov3D_precomp;  /* [N,6] or NULL (exactly one of (scales,rotations)/cov3D_precomp) */
} gp_raster_inputs;

typedef struct gp_raster_outputs {
    float* color;   /* [3,H,W] */
    int32_t* radii; /* [N] */
    float* depth;   /* [1,H,W]  sum_i z_i alpha_i T_i */
    int32_t* tidx;  /* [H,W]    id of the Gaussian with the largest blend weight, -1 if none */
    uint8_t* visible; /* optional [N]: radii > 0, what render() returns as visibility_filter [REF gaussian_renderer/__init__.py:113]
                       * (written by the projection kernel: saves the caller a compare launch per frame); NULL = not wanted */
} gp_raster_outputs;

/* opaque state saved between forward and backward */
typedef struct gp_raster_saved {
    void* geom;    size_t geom_bytes;
    void* binning; size_t binning_bytes;
    void* image;   size_t image_bytes;
    int64_t num_rendered; /* R = sum of tiles touched (exact mode); the binning capacity in capacity mode */
} gp_raster_saved;

/* Optional: torch.optim.Adam's step of the SH coefficients applied INSIDE gp_raster_backward (extension; the reference runs
 * optimizer.step() afterwards, train.py:196).  The preprocess backward holds a workgroup's SH gradients in LDS and the
 * coefficients are still in L2: updating them there saves writing the gradient (192 B per Gaussian) and reading gradient +
 * coefficients back in the optimizer kernel.  Same arithmetic as gp_adam_step_multi, bit for bit (one shared device function).
 * in->shs / in->shs_rest are then UPDATED IN PLACE (they must be the parameters themselves), dL_dshs / dL_dshs_rest are not
 * written and may be NULL.  skip_flag: optional device word, non-zero = leave parameters and moments untouched. */
typedef struct gp_adam_fuse {
    float* exp_avg_dc;       /* [N,1,3]  moments of shs */
    float* exp_avg_sq_dc;
    float* exp_avg_rest;     /* [N,15,3] moments of shs_rest */
    float* exp_avg_sq_rest;
    float lr_dc, lr_rest, beta1, beta2, eps;
    int64_t step;            /* 1-based step number of this update */
    const uint32_t* skip_flag;
} gp_adam_fuse;

typedef struct gp_raster_grads {
    float* dL_dmeans3D;        /* [N,3] */
    float* dL_dmeans2D;        /* [N,3] (x,y in NDC units, z = 0): the screenspace_points grad sink
                                  [REF gaussian_renderer/__init__.py:27-31, scene/gaussian_model.py:757] */
    float* dL_dshs;            /* [N,sh_coeffs,3] ([N,1,3] with shs_rest) or NULL */
    float* dL_dshs_rest;       /* [N,sh_coeffs-1,3] or NULL */
    float* dL_dcolors_precomp; /* [N,3] or NULL */
    float* dL_dopacities;      /* [N,1] */
    float* dL_dscales;         /* [N,3] or NULL */
    float* dL_drotations;      /* [N,4] or NULL */
    float* dL_dcov3D_precomp;  /* [N,6] or NULL */
    int32_t accumulate_shs;    /* 1: dL_dshs / dL_dshs_rest are "+=" targets (the parameters' own .grad buffers:
                                  no temporary, no separate accumulate pass over 192 B/Gaussian); 0: "=" */
    const gp_adam_fuse* adam_shs; /* NULL, or: apply the optimizer step of (shs, shs_rest) in this call (see gp_adam_fuse) */
} gp_raster_grads;

/* replaces _C.rasterize_gaussians (the forward of GaussianRasterizer). One host sync (reads R). */
int gp_raster_forward(const gp_raster_settings* st, const gp_raster_inputs* in, gp_raster_outputs* out,
                      gp_raster_saved* saved, gp_alloc_fn alloc, void* alloc_ctx, gp_stream_t stream);

/* replaces _C.rasterize_gaussians_backward.  dL_ddepth may be NULL.  No host sync. */
int gp_raster_backward(const gp_raster_settings* st, const gp_raster_inputs* in, const gp_raster_outputs* fwd_out,
                       const gp_raster_saved* saved, const float* dL_dcolor /*[3,H,W]*/,
                       const float* dL_ddepth /*[1,H,W] or NULL*/, gp_raster_grads* grads, gp_alloc_fn alloc,
                       void* alloc_ctx, gp_stream_t stream);

/* replaces _C.mark_visible: present[i] = 1 iff Gaussian i passes the near-plane test. */
int gp_raster_mark_visible(int64_t n, const float* means3D, const float* viewmatrix, uint8_t* present,
                           gp_stream_t stream);

/* test/diagnostic accessor: copies the binning result of a forward call to host-visible device
 * buffers: point_list[R] (Gaussian ids in composite order) and ranges[T*2]. */
int gp_raster_debug_binning(const gp_raster_settings* st, const gp_raster_saved* saved, uint32_t* point_list,
                            int32_t* ranges, gp_stream_t stream);

/* ---- deformation path ----------------------------------------------------------------------- */

/* Parameters of Deformable_Field(d=4, w=256) [REF scene/deformable_field.py:102-110]: weights are
 * nn.Linear layout [out,in] row-major: mlp.{0,2,4,6}.{weight,bias}, feature_to_deformation.0.* */
typedef struct gp_mlp_params {
    int32_t in_dim;     /* feature_dim + 6*xyz_freq + 2*time_freq */
    int32_t width;      /* 256 */
    int32_t depth;      /* 4 hidden layers */
    int32_t out_dim;    /* 7 or 8 */
    const float* w[5];  /* w[0]:[width,in_dim]  w[1..3]:[width,width]  w[4]:[out_dim,width] */
    const float* b[5];
    const float* packed; /* optional (NULL = read w[] as they are): gp_mlp_pack's fragment-ordered copy of w[0..3]; used by the
                          * passes over <= 2048 rows (stage 2/3: the rows are the keypoints), which are bound by the rate at which a
                          * single CU takes the weights in.  Must describe the CURRENT values of w[0..3]. */
    void* scratch;       /* optional (NULL = the 16-row kernels): gp_mlp_scratch_bytes(rows) bytes of device memory, ZEROED ONCE by the caller
                          * and then left to the library -- for passes over <= 512 rows gp_mlp_forward then splits every row tile along
                          * the features over 16 workgroups that exchange the activations through memory (counters in the scratch, back
                          * at zero when the call's kernel ends; the hidden activations too when `acts` is NULL): the weights each CU has to
                          * take in drop 16-fold.  ONE scratch per stream: calls that may run at the same time need their own.  The 32-bit word
                          * at byte 4096 is raised if a counter never filled (bit 0) or a row tile's workgroups ran on two XCDs (bit 1):
                          * the result is then undefined (the library validates the first launch on a device and falls back by itself). */
} gp_mlp_params;
/* 0 where the scratch is not used (rows > 512) */
int64_t gp_mlp_scratch_bytes(int64_t rows);

/* fragment-ordered copy of w[0..3] (both the forward's and the backward's operand order; gp_mlp_packed_floats(in_dim) floats,
 * 16-byte aligned): one contiguous kilobyte per wavefront operand load instead of sixteen 64-byte pieces.  Repack whenever the
 * weights change (the Python host keys it on the parameters' version counters). */
int64_t gp_mlp_packed_floats(int32_t in_dim);
int gp_mlp_pack(const gp_mlp_params* p, float* packed, gp_stream_t stream);

typedef struct gp_mlp_grads {
    float* dw[5]; /* accumulated INTO (+=), caller zeroes */
    float* db[5];
} gp_mlp_grads;

/* input row i = [ feature[i, 0:feature_dim] | PE(xyz[i], xyz_freq) | PE(t, time_freq) ]
 * [REF scene/gaussian_model.py:180-189, scene/deformable_field.py:63-72] */
typedef struct gp_mlp_input {
    int64_t rows;
    int32_t feature_dim; /* 32 */
    int32_t xyz_freq;    /* 10 */
    int32_t time_freq;   /* 6 / 8 / 10 */
    const float* feature; /* [rows, feature_dim] */
    const float* xyz;     /* [rows, 3] */
    const float* t;       /* device [1] */
} gp_mlp_input;

/* replaces Deformable_Field.forward on the concatenated input (get_motion_delta).
 * `acts` (optional, training): ceil64(rows * in_pad) + depth * rows * width + depth * rows * 8 floats = the kernel input
 * [rows, in_pad] (in_pad = in_dim rounded up to 8), the [depth, rows, width] post-ReLU activations, and [depth, rows, 8]
 * 32-bit words of ReLU sign bits (written by the large-row kernels). */
int gp_mlp_forward(const gp_mlp_params* p, const gp_mlp_input* x, float* out /*[rows,out_dim]*/,
                   float* acts, gp_stream_t stream);

/* backward of gp_mlp_forward: weight/bias grads (+=), d feature [rows,feature_dim] (=, may be NULL),
 * d xyz [rows,3] through the positional encoding (=, may be NULL). */
int gp_mlp_backward(const gp_mlp_params* p, const gp_mlp_input* x, const float* acts, const float* dL_dout,
                    gp_mlp_grads* g, float* dL_dfeature, float* dL_dxyz, gp_alloc_fn alloc, void* alloc_ctx,
                    gp_stream_t stream);

/* ---- Deformable_Field for shapes other than d = 4, w = 256 [REF options/gaussian_option.py:54-55; scene/deformable_field.py:74-127:
 * any depth / width, and the dormant use_softmax / split_xyz variants].  The network runs layer by layer: exact-fp32 matrix cores,
 * any sizes >= 1, row-major tensors.  (round 5; the fused entries above stay the path of every shipped configuration) */
/* row i of the network input: [ feature[i, :] | PE(xyz[i], xyz_freq) | PE(t, time_freq) ] -> out [rows, feature_dim + 6 xyz_freq + 2 time_freq] */
int gp_mlp_input_forward(const gp_mlp_input* x, float* out, gp_stream_t stream);
/* dL_dfeature [rows, feature_dim] (=, may be NULL), dL_dxyz [rows, 3] through the encoding (=, may be NULL) */
int gp_mlp_input_backward(const gp_mlp_input* x, const float* dL_din, float* dL_dfeature, float* dL_dxyz, gp_stream_t stream);
/* nn.Linear (+ nn.ReLU when relu != 0): y [rows, out_dim] = act(x [rows, in_dim] . w [out_dim, in_dim]^T + b [out_dim] (b may be NULL)) */
int gp_linear_forward(const float* x, int64_t rows, int32_t in_dim, const float* w, const float* b, int32_t out_dim, int32_t relu,
                      float* y, gp_stream_t stream);
/* its backward.  y = the forward's output (read only behind a ReLU: dz = dy where y > 0).  dx (=, may be NULL), dw [out_dim, in_dim] and
 * db [out_dim] (+=, atomics: the caller zeroes or accumulates; each may be NULL) */
int gp_linear_backward(const float* x, const float* y, const float* dy, int64_t rows, int32_t in_dim, const float* w, int32_t out_dim,
                       int32_t relu, float* dx, float* dw, float* db, gp_stream_t stream);
/* nn.Softmax(dim=-1) over [rows, dim] and its backward (dx = y (dy - sum_j y_j dy_j)) */
int gp_softmax_forward(const float* x, int64_t rows, int32_t dim, float* y, gp_stream_t stream);
int gp_softmax_backward(const float* y, const float* dy, int64_t rows, int32_t dim, float* dx, gp_stream_t stream);

/* ---- 16-bit-operand variant (fp16 or bf16 inputs, fp32 accumulate; BASELINE config 5).  Opt-in: results
 * differ from the fp32 path at the 1e-3 (fp16) / 1e-2 (bf16) relative level.  Weights are passed as 16-bit
 * copies prepared by the host (zero-padded): forward  w16[0]:[256,in_pad16] w16[1..3]:[256,256] w16[4]:[32,256];
 * backward takes the TRANSPOSED copies  w16[0]:[in_pad32,256] w16[1..3]:[256,256]^T w16[4]:[256,16]
 * (in_pad32 = in_dim rounded up to 32), each matrix [F][K] stored FRAGMENT-PACKED so that one wavefront load is 1 KB
 * contiguous: element (f, k) at (((k / 16) * (F / 32) + f / 32) * 64 + ((k / 8) & 1) * 32 + f % 32) * 8 + k % 8.
 * GP_DTYPE_F16_SPLIT: fp32-grade results at the 16-bit matrix-core rate.  Every operand is carried as two fp16 numbers
 * (hi = fp16(x), zero below 2^-14; lo' = fp16((x - hi) * 2^11)) and a product sum as three MFMA chains
 * (hi.hi + 2^-11 (hi.lo' + lo'.hi), fp32 accumulate: ~22 significant bits per operand).  Each w16[l] then holds the hi
 * copy followed by the lo' copy (same shape each), the saved tensors hold 2 nf features per row block ([hi | lo']), i.e.
 * twice the bytes, and the positional encoding uses sincosf like gp_mlp_forward.  |activations| must stay below 65504. */
enum { GP_DTYPE_F16 = 1, GP_DTYPE_BF16 = 2, GP_DTYPE_F16_SPLIT = 3 };
typedef struct gp_mlp16_params {
    int32_t dtype;
    int32_t in_dim, width, depth, out_dim;
    const void* w16[5];
    const float* b[5];
    uint32_t* range_flag;   /* optional device word (GP_DTYPE_F16_SPLIT forward): OR-ed with 1 when a hidden activation reached 2^15 --
                             * half the fp16 range the (hi, lo') form saturates at; the caller then repeats the pass with gp_mlp_forward */
} gp_mlp16_params;
/* saved (training, all optional together), 16-bit, BLOCKED by 16 rows, rows zero-padded to a multiple of 64, every tensor's EXTENT
 * padded to a multiple of 128 rows (the 128-row workgroups store whole tiles; what lies beyond the 64-row padding is never read):
 * xT [ceil(rows/128)*8][in_pad16][16] and hT [4][ceil(rows/128)*8][256][16]
 * (element (f, row) at ((row >> 4) * nf + f) * 16 + (row & 15)); GP_DTYPE_F16_SPLIT: twice the features ([hi | lo'] per row block);
 * masks: 8 rows u32 per layer, an opaque hand-over from the forward to the backward (ReLU sign bits: [4][4 feature groups][rows][2],
 * one word per lane of the kernels that write and read it; round 6). */
int gp_mlp16_forward(const gp_mlp16_params* p, const gp_mlp_input* x, float* out, void* saved_xT, void* saved_hT,
                     uint32_t* masks, gp_stream_t stream);
int gp_mlp16_backward(const gp_mlp16_params* p_transposed, const gp_mlp_input* x, const void* saved_xT, const void* saved_hT,
                      const uint32_t* masks, const float* dL_dout, gp_mlp_grads* g, float* dL_dfeature, float* dL_dxyz,
                      gp_alloc_fn alloc, void* alloc_ctx, gp_stream_t stream);
/* The host's part of the above as one launch: the 16-bit copies w16[0..4] (zero-padded, fragment-packed; GP_DTYPE_F16_SPLIT: hi copy
 * followed by lo' copy) of the nn.Linear weights p->w[0..4], in the forward's (transposed = 0) or the backward's (transposed != 0)
 * operand order.  out[l] must hold gp_mlp16_packed_elems(l, in_dim, transposed) 16-bit elements (twice that in split mode).  (round 6:
 * the torch expression of the same was ~140 launches per training step) */
int64_t gp_mlp16_packed_elems(int32_t layer, int32_t in_dim, int32_t transposed);
int gp_mlp16_pack(const gp_mlp_params* p, int32_t dtype, int32_t transposed, void* const* out, gp_stream_t stream);

/* keypoint blend + pose composition  [REF scene/gaussian_model.py:214-229,266-273,285-286,314-315;
 * utils/camera_utils.py:158-170]:
 *   stage 1 (nn == 0): delta is per Gaussian [N,out_dim]
 *   stage 2/3 (nn > 0): delta is per keypoint [K,out_dim]; w = softmax(raw_w[:, :nn]) and
 *   softmax(raw_w[:, nn:2nn]) gathered at knn_idx [N,nn]
 *   xyz_t = xyz + blend(delta[:, 0:3]);  q_t = normalize(quat_mul(normalize(blend(dq)), rot))
 *   where dq = normalize(delta[:, 3:7]) if norm_rotation else delta[:, 3:7]. */
typedef struct gp_blend_args {
    int64_t num_gaussians;
    int64_t num_keypoints;   /* 0 in stage 1 */
    int32_t nearest_num;     /* 0 in stage 1 */
    int32_t out_dim;         /* 7 or 8 */
    int32_t norm_rotation;
    const float* delta;      /* [N or K, out_dim] */
    const float* raw_w;      /* [N, 2*nn] or NULL */
    const int64_t* knn_idx;  /* [N, nn] or NULL */
    const float* xyz;        /* [N,3] */
    const float* rot;        /* [N,4] raw (un-normalised) _rotation */
    const uint16_t* knn_idx16; /* optional: the same indices as 16-bit words (K < 65536; 4-byte aligned).  The reference's
                                * tensor is int64 [REF scene/gaussian_model.py:110-125]: 8 B per neighbour for values below 512;
                                * with this copy the kernels read 2 (gp_knn_keypoints can emit it; NULL = read knn_idx) */
} gp_blend_args;

int gp_blend_forward(const gp_blend_args* a, float* xyz_t /*[N,3]*/, float* q_t /*[N,4]*/, gp_stream_t stream);

/* grads (all "="): d delta, d raw_w (may be NULL: weights without a gradient), d xyz, d rot.  With nn>0 the keypoint gradient
 * is reduced in two deterministic stages (per-workgroup LDS partials, then a sum over workgroups); every element of
 * dL_ddelta is written (columns >= 7 with zeros). */
int gp_blend_backward(const gp_blend_args* a, const float* dL_dxyz_t, const float* dL_dq_t, float* dL_ddelta,
                      float* dL_draw_w, float* dL_dxyz, float* dL_drot, gp_alloc_fn alloc, void* alloc_ctx,
                      gp_stream_t stream);

/* activations [REF scene/gaussian_model.py:41-51,138-162,291-298]:
 *   scale = exp(_scaling); opacity = sigmoid(_opacity) [* sigmoid(delta_o / beta) if delta_o] */
int gp_activations_forward(int64_t n, const float* scaling_raw /*[N,3]*/, const float* opacity_raw /*[N,1]*/,
                           const float* delta_o /*[N, stride] or NULL*/, int32_t delta_o_stride, float beta,
                           float* scale /*[N,3]*/, float* opacity /*[N,1]*/, gp_stream_t stream);
int gp_activations_backward(int64_t n, const float* scaling_raw, const float* opacity_raw, const float* delta_o,
                            int32_t delta_o_stride, float beta, const float* dL_dscale, const float* dL_dopacity,
                            float* dL_dscaling_raw, float* dL_dopacity_raw, float* dL_ddelta_o /*[N,stride] col 0*/,
                            gp_stream_t stream);

/* ---- loss + optimizer (the steps right after the render in every training iteration) ---------- */

/* sum |img - gt| and the sum of the SSIM map (11x11 Gaussian window, sigma 1.5, zero padding) over a [3,H,W]
 * image pair [REF utils/loss_utils.py:54-100].  `sums` (16-byte aligned) has 2 * GP_LOSS_SUM_SLOTS(H, W) doubles, one (sum |a-b|, sum ssim) pair
 * per workgroup (a 32 x 32 tile of one channel), written with plain stores: it need not be initialised, and the totals the
 * finalize calls form are summed in a fixed order (bit-reproducible).  gp_loss_l1_ssim_finalize forms
 * (1-l) * S0/n + l * (1 - S1/n) [REF train.py:108].  dmaps (optional, [3,3,H,W]) receives the SSIM
 * partial-derivative maps the backward needs. */
#define GP_LOSS_SUM_SLOTS(H, W) (3 * (((W) + 31) / 32) * (((H) + 31) / 32))
int gp_loss_l1_ssim_forward(const float* img, const float* gt, int32_t channels, int32_t H, int32_t W, double* sums,
                            float* dmaps, gp_stream_t stream);
/* loss[0] = (1-l) * S0/n + l * (1 - S1/n), n = channels*H*W, S = slot totals: keeps the scalar on the device. */
int gp_loss_l1_ssim_finalize(const double* sums, int32_t channels, int32_t H, int32_t W, float lambda_dssim, float* loss,
                             gp_stream_t stream);
/* dimg = upstream[0] * d loss / d img  (upstream: device scalar, NULL = 1). */
int gp_loss_l1_ssim_backward(const float* img, const float* gt, const float* dmaps, int32_t channels, int32_t H, int32_t W,
                             float lambda_dssim, const float* upstream, float* dimg, gp_stream_t stream);

/* gp_loss_l1_ssim_forward's sums AND gp_loss_l1_ssim_backward[_reg]'s gradient in ONE launch (round 6: the fused train step's form;
 * the derivative maps never leave the CU).  Bit-identical sums and gradient.  x / gx: the regulariser's input and gradient as in
 * gp_loss_l1_ssim_backward_reg (both NULL: none).  The loss value: a finalize call on `sums` as before. */
int gp_loss_l1_ssim_fused(const float* img, const float* gt, int32_t channels, int32_t H, int32_t W, float lambda_dssim,
                          const float* upstream, double* sums, float* dimg, const float* x, int64_t n, float scale, float* gx,
                          gp_stream_t stream);

/* finalize / backward with the regulariser  scale * mean(|x|)  folded in (n <= 65536: the keypoint features of stage 2/3;
 * [REF scene/gaussian_model.py:174-178, train.py:108-109]):  loss[0] = the finalize value + scale * mean|x|;
 * gx = upstream[0] * scale/n * sign(x) written by the backward launch. */
int gp_loss_l1_ssim_finalize_reg(const double* sums, int32_t channels, int32_t H, int32_t W, float lambda_dssim, const float* x,
                                 int64_t n, float scale, float* loss, gp_stream_t stream);
int gp_loss_l1_ssim_backward_reg(const float* img, const float* gt, const float* dmaps, int32_t channels, int32_t H, int32_t W,
                                 float lambda_dssim, const float* upstream, float* dimg, const float* x, int64_t n, float scale,
                                 float* gx, gp_stream_t stream);

/* out[0] = base[0] + scale * mean(|x|): the motion-feature regulariser added to the loss
 * [REF scene/gaussian_model.py:174-178, train.py:108-109]; g = upstream[0] * scale/n * sign(x). */
int gp_l1_mean_forward(const float* x, int64_t n, float scale, const float* base, float* out, gp_stream_t stream);
int gp_l1_mean_backward(const float* x, int64_t n, float scale, const float* upstream, float* g, gp_stream_t stream);

/* torch.optim.Adam step (amsgrad off, no weight decay) on one flat tensor; optionally zeroes `grad`
 * [REF scene/gaussian_model.py:472, train.py:196-197].  `step` is the 1-based step count. */
int gp_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                 float eps, int64_t step, int32_t zero_grad, gp_stream_t stream);

/* the same for up to 32 tensors in ONE launch (host arrays of device pointers / sizes / learning rates).
 * keep_grad_mask: bit k set = do NOT zero tensor k's gradient even when zero_grad != 0 (its next producer overwrites it).
 * skip_flag: NULL, or a device word read by the kernel: non-zero = the gradients come from an invalid frame (binning
 * overflow of the rasterizer's capacity mode, gp_raster_settings.binning_status + 1): parameters and moments stay as they
 * are, the gradients are still zeroed. */
int gp_adam_step_multi(int32_t count, float* const* params, float* const* grads, float* const* exp_avgs,
                       float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, float beta1, float beta2, float eps,
                       int64_t step, int32_t zero_grad, uint32_t keep_grad_mask, const uint32_t* skip_flag, gp_stream_t stream);

/* the same with a step count PER TENSOR (host array, 1-based), as torch.optim.Adam keeps it: a parameter whose .grad is None when
 * step() runs is skipped and its count stays behind -- in the reference the per-Gaussian tensors on every densify / prune
 * iteration, which replace them before optimizer.step() [REF train.py:164-197, scene/gaussian_model.py:547-599]. */
int gp_adam_step_multi_steps(int32_t count, float* const* params, float* const* grads, float* const* exp_avgs,
                             float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, const int64_t* steps, float beta1,
                             float beta2, float eps, int32_t zero_grad, uint32_t keep_grad_mask, const uint32_t* skip_flag,
                             gp_stream_t stream);

/* ---- one training iteration of the hot path in ONE call (round 5) -------------------------------------------------------------
 * What the reference's loop does per view [REF train.py:101-133, 196-197] --
 *     render(cam, gaussians, ..., time, it)  ->  0.8 L1 + 0.2 (1 - SSIM) + 1e-5 mean|keypoint feature|  ->  backward  ->  Adam
 * -- for the stage-3 form of GaussianModel.forward (keypoint MLP + sparse blend, [REF scene/gaussian_model.py:251-273]; keypoint
 * weights and neighbour indices supplied, as BASELINE.json's north_star words it), enqueued by ONE call of the library instead of a
 * dozen calls out of a Python autograd graph: gp_mlp_forward -> gp_blend_forward -> gp_activations_forward -> gp_raster_forward ->
 * gp_loss_l1_ssim_forward / _finalize[_reg] / _backward[_reg] -> gp_raster_backward -> gp_activations_backward -> gp_blend_backward ->
 * gp_mlp_backward -> gp_adam_step_multi[_steps].  Same kernels, same order, same arithmetic as the separate entry points (it CALLS
 * them); the host's work per step drops from ~0.6 ms of Python to the launches themselves, which is what a view-parallel step
 * needs (its Python-side exchange does not hide behind 1.2 ms of kernels any more).  GaussianRasterizer / render() stay the drop-in
 * autograd surface; this is the harness's fast path (gaussianprediction_amd/train_step.py: TrainStep(fused=...)).
 *
 * Every buffer is the caller's: parameters, gradient buffers, moments, and the intermediates the plan struct lists (persistent
 * across steps: their addresses never change, so the plan is filled once).  Scratch of the rasterizer / backward kernels still comes
 * from the allocator callback.  Gradient buffers: "=" -- written whole by one producer; "+=" -- accumulated into, zero on entry
 * (the optimizer launch re-zeroes them: keep_grad_mask must leave their bits clear). */
typedef struct gp_step_plan {
    int64_t num_gaussians, num_keypoints;
    int32_t nearest_num, norm_rotation, sh_degree, image_height, image_width;
    float lambda_dssim;          /* 0.2 [REF arguments/__init__.py:84] */
    float reg_scale;             /* 1e-5: + reg_scale * mean|keypoint_features| [REF scene/gaussian_model.py:174-178]; 0 = none */
    /* parameters */
    const float *xyz, *scaling, *rotation, *opacity;      /* [N,3] [N,3] [N,4] [N,1] (raw, pre-activation) */
    float *features_dc, *features_rest;                   /* [N,1,3] [N,15,3]; updated in place when gp_step_update.adam_shs is set */
    const float *keypoints, *keypoint_features;           /* [K,3] [K,feature_dim] */
    gp_mlp_params mlp;                                    /* Deformable_Field weights (packed = NULL in training) */
    int32_t feature_dim, xyz_freq, time_freq, reserved0;
    const float* raw_w;          /* [N, 2 nn] keypoint weights (no gradient) */
    const int64_t* knn_idx;      /* [N, nn] */
    const uint16_t* knn_idx16;   /* optional, see gp_blend_args */
    /* gradients */
    float *g_xyz, *g_scaling, *g_rotation, *g_opacity;    /* "=" */
    float *g_features_dc, *g_features_rest;               /* "="; may be NULL when adam_shs is set */
    float *g_keypoints;                                   /* "=" */
    float *g_keypoint_features;                           /* "=" (regulariser + MLP input gradient) */
    gp_mlp_grads g_mlp;                                   /* "+=" */
    /* intermediates (caller-owned, persistent) */
    float* delta;                /* [K, out_dim] */
    float* acts;                 /* gp_mlp_forward's activation record for K rows */
    float *xyz_t, *q_t, *scale, *opacity_t;               /* [N,3] [N,4] [N,3] [N,1] */
    gp_raster_outputs out;       /* color [3,H,W], radii [N], depth [1,H,W], tidx [H,W], visible [N] (optional) */
    double* loss_sums;           /* 2 * GP_LOSS_SUM_SLOTS(H, W) */
    float* dmaps;                /* [3,3,H,W], or NULL: taken from the allocator as a TEMP buffer (it is dead before the backward's own TEMP) */
    float* loss;                 /* [1] */
    float* dL_dimage;            /* [3,H,W] */
    float *g_xyz_t, *g_q_t, *g_scale, *g_opacity_t;       /* [N,3] [N,4] [N,3] [N,1] */
    float* g_means2D;            /* [N,3]: what viewspace_points.grad holds [REF scene/gaussian_model.py:757] */
    float* g_delta;              /* [K, out_dim] */
    float* g_feature_tmp;        /* [K, feature_dim] */
} gp_step_plan;

typedef struct gp_step_view {   /* the camera of this step [REF gaussian_renderer/__init__.py:34-46] and its target */
    float tanfovx, tanfovy;
    const float *bg, *viewmatrix, *projmatrix, *campos;   /* device [3] [16] [16] [3] */
    const float* gt_image;       /* [3,H,W] */
    const float* time;           /* device [1] */
} gp_step_view;

typedef void (*gp_step_hook_fn)(void* ctx, int32_t point);
enum { GP_STEP_AFTER_RASTER_BACKWARD = 0,   /* the SH gradients (and g_means2D) are final: a view-parallel caller starts their exchange */
       GP_STEP_AFTER_BACKWARD = 1,          /* every gradient is final (called in front of the optimizer launch, if there is one) */
       GP_STEP_AFTER_FORWARD = 2 };         /* the rasterizer forward is enqueued: binning_status is final once the stream gets there */

typedef struct gp_step_update {
    int64_t binning_capacity;    /* > 0: capacity mode (required: the call never synchronises) */
    uint32_t* binning_status;    /* device {R, overflow} */
    int32_t depth_key_bits;      /* see gp_raster_settings */
    uint32_t depth_key_base;
    void* sh_ready_event;        /* see gp_raster_settings */
    const gp_adam_fuse* adam_shs;/* NULL, or the SH pair's update inside the rasterizer backward */
    /* the optimizer launch behind the backward: gp_adam_step_multi (steps == NULL) / gp_adam_step_multi_steps; count 0 = none */
    int32_t adam_count;
    /* bit k: tensor k's gradient is FINAL once the blend backward has run (the per-Gaussian tensors: nothing behind that point writes
     * them) and the keypoint MLP's backward neither reads nor writes the tensor.  Those updates travel in the SAME LAUNCH as that
     * backward's data kernel (a handful of workgroups, each bound for tens of microseconds by the rate at which one CU takes the
     * layer weights in: the optimizer's HBM-bound chunks fill the rest of the part around them; round 6 -- round 5 used a second stream,
     * whose fork / join events cost more than the overlap brought, profiles/r05_early_adam_ab.txt); the other tensors follow the MLP
     * backward as before.  Element-wise arithmetic: the result is bit-identical to one launch.  0 = one launch behind the backward.
     * Ignored (one launch) when `hook` is set: a view-parallel caller reduces the gradients at GP_STEP_AFTER_BACKWARD first. */
    uint32_t adam_early_mask;
    float* const* adam_params; float* const* adam_grads; float* const* adam_exp_avgs; float* const* adam_exp_avg_sqs;
    const int64_t* adam_numels; const float* adam_lrs; const int64_t* adam_steps;
    float beta1, beta2, eps;
    int64_t step;
    uint32_t keep_grad_mask;
    const uint32_t* skip_flag;
    gp_step_hook_fn hook;        /* optional: called on the host, between enqueues, at the GP_STEP_* points */
    void* hook_ctx;
} gp_step_update;

int gp_train_step_run(const gp_step_plan* plan, const gp_step_view* view, const gp_step_update* upd, gp_alloc_fn alloc,
                      void* alloc_ctx, gp_stream_t stream);

/* ---- keypoint weights (parity unpinned: tinycudann / frnn are absent from the reference tree) --- */

/* the Grid/Hash encoding of weights_model = tcnn.NetworkWithInputEncoding(...) [REF scene/gaussian_model.py:370-392] */
typedef struct gp_hashgrid_config {
    int32_t n_levels;               /* 16 */
    int32_t n_features_per_level;   /* 4 (only value implemented) */
    int32_t log2_hashmap_size;      /* 19 */
    int32_t base_resolution;        /* 16 */
    float per_level_scale;          /* exp(ln(2048/16)/15) */
} gp_hashgrid_config;

/* number of table entries (n_features_per_level floats each) over all levels; -1 on a bad configuration */
int64_t gp_hashgrid_table_entries(const gp_hashgrid_config* cfg);
/* out[n, n_levels*4] = trilinear multiresolution hash encoding of xyz[n,3] (used as given: the reference does not
 * normalise positions) -- the encoding half of weights_model(self.get_xyz.detach()) [REF scene/gaussian_model.py:257].
 * perm (optional, int32[n]): a permutation of the points in a spatially coherent order (e.g. Morton); it only changes
 * which points share a wavefront -- neighbouring lanes then read neighbouring table entries -- never the result. */
int gp_hashgrid_forward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm, const float* table,
                        float* out, gp_stream_t stream);
/* dtable += d out / d table applied to dL_dout[n, n_levels*4] (xyz is detached in the reference: no position gradient) */
int gp_hashgrid_backward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm, const float* dL_dout,
                         float* dtable, gp_stream_t stream);
/* the whole weights model fused: hash-grid encoding + the 64-wide bias-free MLP (64 -> 64 -> 64 -> 16 padded, ReLU) on
 * the fp32 matrix cores.  params = [W1 64x64 | W2 64x64 | W3 16x64 | table entries x 4] (one flat tensor, as tcnn exposes
 * it); out[n, n_out] = first n_out columns.  saved_feat (optional, [n,64], slot-major: row k belongs to point perm[k]) is
 * what the backward needs.  n_levels must be 16. */
int gp_weights_forward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm, const float* params,
                       int32_t n_out, float* out, float* saved_feat, gp_stream_t stream);
/* dparams (same layout as params) += gradient of sum(out * dL_dout) */
int gp_weights_backward(const gp_hashgrid_config* cfg, int64_t n, const float* xyz, const int32_t* perm, const float* params,
                        int32_t n_out, const float* saved_feat, const float* dL_dout, float* dparams, gp_alloc_fn alloc,
                        void* alloc_ctx, gp_stream_t stream);
/* idx_out[n, nn] (int64, ascending squared distance; ties to the lower index) = the nn nearest of the K keypoints, in 3-D
 * (feat_dim = 0: knn_type "3D") or in [xyz | amplify * feature] (feat_dim = 32: "hybird")
 * [REF scene/gaussian_model.py:110-125, frnn.frnn_grid_points].  d2_out (optional) receives the squared distances, idx16_out
 * (optional, K < 65536) the indices again as 16-bit words (gp_blend_args.knn_idx16).
 * order (optional, int32[n]: a permutation of the points, e.g. along a Morton curve) only decides which points share a
 * wavefront -- spatially coherent wavefronts drop most keypoints after 3 of the 35 dimensions; the result does not depend on it. */
int gp_knn_keypoints(int64_t n, const float* xyz, const float* feat, int32_t feat_dim, float amplify, int64_t K,
                     const float* kp_xyz, const float* kp_feat, int32_t nn, const int32_t* order, int64_t* idx_out, float* d2_out,
                     uint16_t* idx16_out, gp_stream_t stream);

/* out[n] = mean of the squared distances from point i to its three nearest OTHER points (self excluded by index): replaces
 * simple_knn's distCUDA2, which sizes the initial Gaussians [REF scene/gaussian_model.py:340-341 create_from_pcd].  Exact brute
 * force, initialisation-time.  Fewer than three other points: the missing ones count as FLT_MAX (as the published kernel). */
int gp_knn3_mean_dist2(int64_t n, const float* xyz /*[n,3]*/, float* out /*[n]*/, gp_stream_t stream);

/* Furthest-point sampling of xyz[n,3], starting at point 0: idx_out[m] (int32); idx_out[j] is the point farthest from
 * {idx_out[0..j)} (first maximum on ties).  tmp_dist: n floats of scratch.  Replaces pointops' furthestsampling_cuda for one
 * batch [REF utils/fps.py:71-88, scene/gaussian_model.py:196-212 get_new_kpts]. */
int gp_furthest_point_sampling(int64_t n, const float* xyz, int64_t m, int32_t* idx_out, float* tmp_dist, gp_stream_t stream);

/* ---- measurement ----------------------------------------------------------------------------- */
/* gp_profile_enable(level): 0 = off, 1 = bracket only the roofline kernel (composite forward), 2 = every
 * kernel.  When enabled, the library brackets kernels with hipEvent pairs recorded on the launch stream.
 * gp_profile_collect() waits for the recorded events and returns per-kernel launch counts and summed
 * durations since the previous collect (used by bench.py for the roofline figures). */
typedef struct gp_profile_entry {
    char name[48];
    int32_t launches;
    float total_ms;
} gp_profile_entry;
int gp_profile_enable(int on);
int gp_profile_collect(gp_profile_entry* out, int max_entries, int* n_out);

/* Peak microbenchmarks (the measured stream-copy and MFMA peaks are reported beside the vendor
 * numbers).  Each call enqueues ONE kernel on `stream`, bracketed (profile level >= 1) under the names "mb_copy",
 * "mb_read", "mb_mfma_f32|f16|bf16"; the caller divides bytes / flop by the collected time.
 *   copy: dst[0..bytes) = src[0..bytes), one 16-byte vector per thread, one-shot grid (the shape that reaches the part's copy
 *         peak: profiles/r03_copy_peak_sweep.jsonl; bytes <= 64 GiB); moves 2*bytes over HBM.
 *   read: reads src[0..bytes) and discards it (sink is never written for a zero-filled source).
 *   mfma: 4 independent accumulator chains of `iters` 32x32 MFMAs per wave, 4 waves x 8 workgroups per CU; dtype
 *         0 = f32 (32x32x2), 1 = f16, 2 = bf16 (32x32x16); *flop_out = total floating-point operations enqueued. */
int gp_microbench_copy(void* dst, const void* src, size_t bytes, gp_stream_t stream);
int gp_microbench_read(const void* src, size_t bytes, float* sink, gp_stream_t stream);
int gp_microbench_mfma(int dtype, int iters, float* sink, double* flop_out, gp_stream_t stream);
/* Instruction-rate microbenchmarks: 8 independent chains per lane of ONE instruction kind, 4 waves x 8 workgroups per CU
 * (the occupancy of the composite kernels); *instr_out = wave-instructions of that kind enqueued.  kind: 0 v_fma_f32,
 * 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_cmp_f32 + v_cndmask_b32 (two instructions), 4 v_rcp_f32, 5 v_add_f32_dpp, 6 broadcast
 * ds_read_b128, 7 v_min_f32, 8 v_pk_mul_f32, 9 v_sqrt_f32.  Bracketed as "mb_valu_*" / "mb_lds_read_b128". */
int gp_microbench_valu(int kind, int iters, float* sink, double* instr_out, gp_stream_t stream);
/* Random gather (the record fetch of the composite kernels): thread i reads rec_bytes at src + idx[i] * stride_bytes.
 * Bracketed as "mb_gather"; run under `rocprofv3 --pmc FETCH_SIZE` it calibrates the counter for this access shape. */
int gp_microbench_gather(const void* src, int rec_bytes, int stride_bytes, const uint32_t* idx, size_t n_idx, float* sink,
                         gp_stream_t stream);
/* Diagnostics: select a kernel variant for A/B profiling (key 0: composite forward, key 1: composite backward; value 0 =
 * the shipped default; key 2: access shape of gp_microbench_copy, tools/copy_peak_sweep.py; key 3: 1 = round 2's 2-D grid of the
 * split-mode weight-gradient kernel; key 4: workgroup cap of the fused weights-model forward;
 * key 5: 1 = tile binning by duplicate + radix sort instead of by counting, csrc/bin_kernels.hip; key 6: ablation bits of the
 * binning scatter kernel; key 8: 2 = depth sort in three 11-bit counting passes instead of four 8-bit radix passes -- measured slower, kept for the A/B;
 * key 9: ablation bits of the 16-bit MLP forward, tools/probe/mlp16_ablate.py; key 10: 16-row blocks per slab of the 16-bit weight gradient;
 * key 11: 1 = the pair of loss kernels in gp_train_step_run; key 12: P > 1 = gp_profile_enable(1) brackets every P-th launch only; key 13: 1 = the 16-row kernels where the feature-split small-row MLP would run,
 * 2 = its agent-scope form of the exchange; key 14: 1 = activation launches of their own in gp_train_step_run instead of raw_activations).
 * Never needed by a caller of the render path. */
int gp_debug_option(int key, int value);
/* Diagnostics: with gp_debug_option(0, 3) the composite forward counts, over all launches since the last call, out4[0] = the
 * (pixel, splat) pairs that contribute (alpha >= 1/255, pixel not yet saturated) and out4[1] = the pairs its sub-block lists make
 * it evaluate; this call synchronises the device, returns and clears them (bench.py: roofline.contributing_pairs). */
int gp_debug_counters(uint64_t* out4);

const char* gp_last_error(void);
const char* gp_version(void);

/* View-parallel training, factorised SH exchange (round 6; gaussianprediction_amd/dist.py): per view the SH gradient is
 * rank one, dL/dSH[k][ch] = Y_k(dir) dL/dRGB[ch] [REF utils/sh_utils.py:57-112], so the ranks all-gather factors [world][n][6] =
 * (dL/dRGB[3] | unit view direction[3]) -- 24 B per Gaussian and view instead of 192 B of gradient -- and this call forms the SUM over the
 * views in rank order (identical on every rank): g_dc [n,3] and g_rest [n,15,3] are WRITTEN (coefficients beyond sh_degree: zero). */
int gp_sh_factor_gradient(int64_t n, int32_t world, const float* factors, int32_t sh_degree, float* g_dc, float* g_rest, gp_stream_t stream);

/* ABI number of this header.  It changes whenever a struct gains / loses a field, an entry point's signature changes or a
 * buffer-size macro (GP_LOSS_SUM_SLOTS) changes; a binding built against another number must refuse to run (the Python loader
 * does: gaussianprediction_amd/_lib.py).  History: 1 = rounds 1-2; 2 = round 3 (gp_raster_settings.sh_ready_event / visible,
 * gp_knn_keypoints' `order`, GP_LOSS_SUM_SLOTS per image size); 3 = round 4 (gp_abi_version itself); 4 = round 4 (gp_mlp_params.packed, gp_blend_args.knn_idx16, gp_knn_keypoints' signature,
 * gp_adam_step_multi_steps); 5 = round 5 (gp_train_step_run and its three structs);
 * 6 = round 6 (gp_mlp16_pack / gp_mlp16_packed_elems, gp_loss_l1_ssim_fused, gp_sh_factor_gradient; the ReLU words gp_mlp16_forward hands to gp_mlp16_backward changed layout);
 * 7 = round 6, last session (gp_mlp_params.scratch + gp_mlp_scratch_bytes, gp_raster_settings.raw_activations, the saved 16-bit tensors' extent padded to 128 rows). */
#define GP_ABI_VERSION 7
int gp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GP_HIP_H */

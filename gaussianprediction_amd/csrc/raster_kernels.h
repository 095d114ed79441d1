// raster_kernels.h -- declarations of the rasterizer kernels (definitions in raster_kernels.hip)
#pragma once
#include "gp_common.h"

// backward accumulators: one 64-byte line per Gaussian  [0,1] mean2D  [2..4] conic  [5] opacity  [6..8] colour  [9] depth
#define GP_ACC_STRIDE 16
// the backward walks the image in parts of GP_BWD_ROWS x GP_BWD_COLS pixels, one wave each
#ifndef GP_BWD_ROWS
#define GP_BWD_ROWS 8
#define GP_BWD_COLS 8
#endif
#define GP_BWD_PARTS ((16 / GP_BWD_ROWS) * (16 / GP_BWD_COLS))
#define GP_BWD_PAIRS (GP_BWD_ROWS * GP_BWD_COLS / 2)

// 0 (shipped): the composite forward's staging lanes compute the whole sub-block test per instance; 1: its per-Gaussian half comes from a side
// array the projection kernel writes; 2: from the record's spare word (raster_kernels.hip gp_sb_side; tools/probe/sb_hoist_ab.sh)
#ifndef GP_SB_HOIST
#define GP_SB_HOIST 0
#endif
struct RasterDims {
    int N, M, D;       // gaussians, sh coeffs per channel, active sh degree
    int W, H, gx, gy;  // image and tile grid
    float tanfovx, tanfovy, fx, fy, scale_mod;
    int late_color;    // 1: the preprocess kernel leaves the colour to gp_sh_color_*_kernel (gp_raster_settings.sh_ready_event)
    // riding on the preprocess launch instead of launches of their own (each ~4 us on the step's critical path):
    uint8_t* visible;      // optional [N]: radii > 0 (the renderer's visibility_filter)
    uint32_t* zero_words;  // optional: n_zero words the binning stage expects zeroed (tile ranges + instance-counter slots)
    int n_zero;
    // depth-key speculation (gp_raster_settings.depth_key_bits): key_hi = the bits of (key - key_base) the caller promised to be zero (0 = no promise),
    // key_culled = the key of a Gaussian without tiles (sorts behind every visible one), key_flag = the word raised on a broken promise
    uint32_t key_hi, key_base, key_culled, key_tag;
    uint32_t* key_flag;     // binning_status + 2: receives key_tag (this call's number) when a visible key breaks the promise
    // per-Gaussian half of the composite forward's sub-block test (gp_sb_mask), written by the projection kernel: (dyr, inv_cx), inv_cx = NaN
    // for "do not cull" (degenerate conic).  The staging lanes computed these -- three rcp and a sqrt -- per tile-splat INSTANCE (R = 4 N)
    float2* sb_side;
    // gp_raster_settings.raw_activations: != NULL = the `opacities` pointer, holding LOGITS (and `scales` log-scales): the projection
    // applies sigmoid / exp, the backward chains through them
    const float* raw_opacity;
};
__global__ __launch_bounds__(256) void gp_key_range_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ radii, int n,
                                                          uint32_t base, uint32_t* __restrict__ out2);

__global__ __launch_bounds__(256) void gp_preprocess_fwd_kernel(RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales,      const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,      const float* __restrict__ shs_rest, const float* __restrict__ colors_precomp, const float* __restrict__ cov3D_precomp,      const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,      int32_t* __restrict__ radii, float4* __restrict__ rec, uint32_t* __restrict__ depth_key,      uint2* __restrict__ tiles_touched, uint8_t* __restrict__ clamped);

__global__ __launch_bounds__(256) void gp_preprocess_fwd_sh16_kernel(RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales,      const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,      const float* __restrict__ shs_rest, const float* __restrict__ colors_precomp, const float* __restrict__ cov3D_precomp,      const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,      int32_t* __restrict__ radii, float4* __restrict__ rec, uint32_t* __restrict__ depth_key,      uint2* __restrict__ tiles_touched, uint8_t* __restrict__ clamped);

__global__ __launch_bounds__(256) void gp_preprocess_fwd_split_kernel(RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales,      const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,      const float* __restrict__ shs_rest, const float* __restrict__ colors_precomp, const float* __restrict__ cov3D_precomp,      const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,      int32_t* __restrict__ radii, float4* __restrict__ rec, uint32_t* __restrict__ depth_key,      uint2* __restrict__ tiles_touched, uint8_t* __restrict__ clamped);

#define GP_SC_ARGS RasterDims d, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest, \
    const float* __restrict__ campos, const int32_t* __restrict__ radii, float4* __restrict__ rec, uint8_t* __restrict__ clamped
__global__ __launch_bounds__(256) void gp_sh_color_kernel(GP_SC_ARGS);
__global__ __launch_bounds__(256) void gp_sh_color_sh16_kernel(GP_SC_ARGS);
__global__ __launch_bounds__(256) void gp_sh_color_split_kernel(GP_SC_ARGS);

__global__ __launch_bounds__(256) void gp_mark_visible_kernel(int n, const float* __restrict__ means3D,
                                                             const float* __restrict__ view, uint8_t* __restrict__ present);


__global__ __launch_bounds__(256) void gp_duplicate_kernel(RasterDims d, const uint32_t* __restrict__ sorted_ids,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ total,
                                                          const uint2* __restrict__ rect_sorted,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t capacity,
                                                          uint32_t* __restrict__ status, uint32_t n_dup_blocks, uint32_t key_tag);

__global__ __launch_bounds__(256) void gp_tile_ranges_kernel(const uint32_t* __restrict__ keys, uint32_t R, uint32_t n_tiles,
                                                            int2* __restrict__ ranges);
__global__ void gp_binning_status_kernel(const uint32_t* __restrict__ total, uint32_t capacity, uint32_t* __restrict__ status, uint32_t key_tag);

__global__ __launch_bounds__(1024) void gp_tile_order_kernel(const int2* __restrict__ ranges, const int32_t* __restrict__ work_hint, int T, uint32_t* __restrict__ order);
__global__ __launch_bounds__(1024) void gp_bwd_prologue_kernel(const int2* __restrict__ ranges, const int32_t* __restrict__ work_hint, int T,
                                                               uint32_t* __restrict__ order, float* __restrict__ acc, size_t acc_floats);



__global__ __launch_bounds__(256) void gp_preprocess_bwd_kernel(RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales,      const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ shs_rest,      const float* __restrict__ cov3D_precomp, const float* __restrict__ view, const float* __restrict__ proj,      const float* __restrict__ campos, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,      const float* __restrict__ g_mean2D, const float* __restrict__ g_conic, const float* __restrict__ g_opacity,      const float* __restrict__ g_color, const float* __restrict__ g_depth, float* __restrict__ dL_dmeans3D,      float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dshs, float* __restrict__ dL_dshs_rest,      float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales,      float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D, int accumulate_shs, AdamFuseDev af);

__global__ __launch_bounds__(256) void gp_preprocess_bwd_sh16_kernel(RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales,      const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ shs_rest,      const float* __restrict__ cov3D_precomp, const float* __restrict__ view, const float* __restrict__ proj,      const float* __restrict__ campos, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,      const float* __restrict__ g_mean2D, const float* __restrict__ g_conic, const float* __restrict__ g_opacity,      const float* __restrict__ g_color, const float* __restrict__ g_depth, float* __restrict__ dL_dmeans3D,      float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dshs, float* __restrict__ dL_dshs_rest,      float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales,      float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D, int accumulate_shs, AdamFuseDev af);

__global__ __launch_bounds__(256) void gp_preprocess_bwd_split_kernel(RasterDims d, const float* __restrict__ means3D, const float* __restrict__ scales,      const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ shs_rest,      const float* __restrict__ cov3D_precomp, const float* __restrict__ view, const float* __restrict__ proj,      const float* __restrict__ campos, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,      const float* __restrict__ g_mean2D, const float* __restrict__ g_conic, const float* __restrict__ g_opacity,      const float* __restrict__ g_color, const float* __restrict__ g_depth, float* __restrict__ dL_dmeans3D,      float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dshs, float* __restrict__ dL_dshs_rest,      float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales,      float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D, int accumulate_shs, AdamFuseDev af);


__global__ __launch_bounds__(256) void gp_composite_fwd_sb_kernel(RasterDims d, const int2* __restrict__ ranges,
                                                                  const uint32_t* __restrict__ point_list,
                                                                  const float4* __restrict__ rec,
                                                                  const float* __restrict__ bg,
                                                                  float* __restrict__ out_color,
                                                                  float* __restrict__ out_depth,
                                                                  int32_t* __restrict__ out_tidx,
                                                                  float* __restrict__ final_T,
                                                                  int32_t* __restrict__ n_contrib, const uint32_t* __restrict__ order, int32_t* __restrict__ tile_work, uint16_t* __restrict__ smask);

// heavy-first launch order: bucket of a tile by a 2-mantissa-bit logarithm of its work (0 = heaviest ... 127 = no work)
__device__ __forceinline__ int gp_tile_bucket(int w) {
    if (w <= 0) return 127;
    const int lg = 31 - __clz(w);
    const int sub = lg >= 2 ? ((w >> (lg - 2)) & 3) : 0;
    const int v = lg * 4 + sub;
    return 126 - (v < 126 ? v : 126);
}
int gp_pair_counters_read(unsigned long long* out4);
int gp_bwd_set_ablate(int v);
__global__ __launch_bounds__(256) void gp_composite_fwd_count_kernel(RasterDims d, const int2* __restrict__ ranges,
                                                                  const uint32_t* __restrict__ point_list,
                                                                  const float4* __restrict__ rec,
                                                                  const float* __restrict__ bg,
                                                                  float* __restrict__ out_color,
                                                                  float* __restrict__ out_depth,
                                                                  int32_t* __restrict__ out_tidx,
                                                                  float* __restrict__ final_T,
                                                                  int32_t* __restrict__ n_contrib, const uint32_t* __restrict__ order, int32_t* __restrict__ tile_work, uint16_t* __restrict__ smask);

__global__ __launch_bounds__(256) void gp_composite_fwd_sbc_kernel(RasterDims d, const int2* __restrict__ ranges,
                                                                  const uint32_t* __restrict__ point_list,
                                                                  const float4* __restrict__ rec,
                                                                  const float* __restrict__ bg,
                                                                  float* __restrict__ out_color,
                                                                  float* __restrict__ out_depth,
                                                                  int32_t* __restrict__ out_tidx,
                                                                  float* __restrict__ final_T,
                                                                  int32_t* __restrict__ n_contrib, const uint32_t* __restrict__ order, int32_t* __restrict__ tile_work, uint16_t* __restrict__ smask);

#define GP_CB_ARGS RasterDims d, const int2* __restrict__ ranges, const uint32_t* __restrict__ point_list, \
    const uint16_t* __restrict__ smask, const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ out_color, \
    const float* __restrict__ out_depth, const float* __restrict__ final_T, const int32_t* __restrict__ n_contrib, \
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpixdepth, float* __restrict__ g_mean2D, float* __restrict__ g_conic, \
    float* __restrict__ g_opacity, float* __restrict__ g_color, float* __restrict__ g_depth, const uint32_t* __restrict__ order
__global__ __launch_bounds__(64) void gp_composite_bwd_kernel(GP_CB_ARGS);
__global__ __launch_bounds__(64) void gp_composite_bwd_depth_kernel(GP_CB_ARGS);
// round 5: 16-lane groups walk the quadrant's four 4x4 sub-blocks (an A/B variant, gp_debug_option(7, 3): measured slower than the two above)
__global__ __launch_bounds__(64) void gp_composite_bwd_sb_kernel(GP_CB_ARGS);
__global__ __launch_bounds__(64) void gp_composite_bwd_sb_depth_kernel(GP_CB_ARGS);
__global__ __launch_bounds__(64) void gp_composite_bwd_sb_count_kernel(GP_CB_ARGS);
#define GP_BWD_SB_MAX_N (1 << 28)      // the sub-block kernel packs its 4 sub-block bits above the Gaussian id

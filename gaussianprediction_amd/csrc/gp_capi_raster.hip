// gp_capi_raster.hip -- extern "C" entry points of the rasterizer (see include/gp_hip.h).
// Host orchestration only: sizes buffers, asks the caller's allocator for scratch, enqueues the
// kernels of raster_kernels.hip / sort_scan.hip on the caller's stream.
#include "gp_common.h"
#include "raster_kernels.h"
#include <stdlib.h>
#include <atomic>

thread_local char gp_err_buf[512] = "";
static std::atomic<uint32_t> g_key_tag{0};       // numbers the forward calls that check a depth-key promise (never 0)

extern "C" const char* gp_last_error(void) { return gp_err_buf; }
extern "C" const char* gp_version(void) { return "gaussianprediction_amd 0.4 (gfx950)"; }
extern "C" int gp_abi_version(void) { return GP_ABI_VERSION; }

static int tile_bits_for(int T) {
    int b = 1;
    while ((1 << b) < T) ++b;
    return b;
}

static int make_dims(const gp_raster_settings* st, const gp_raster_inputs* in, RasterDims& d) {
    if (!st || !in) GP_FAIL("null settings/inputs");
    if (st->image_width <= 0 || st->image_height <= 0) GP_FAIL("bad image size %dx%d", st->image_width, st->image_height);
    if (in->num_gaussians < 0 || in->num_gaussians > 0x7FFFFFF0LL) GP_FAIL("num_gaussians out of range");
    if (st->sh_degree < 0 || st->sh_degree > 3) GP_FAIL("sh_degree %d unsupported (0..3)", st->sh_degree);
    const bool has_sr = in->scales != nullptr || in->rotations != nullptr;
    if (in->num_gaussians > 0) {  // (an empty tensor has a NULL data pointer: nothing to check then)
        if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) GP_FAIL("Please provide exactly one of either SHs or precomputed colors!");
        if (has_sr == (in->cov3D_precomp != nullptr) || (has_sr && (!in->scales || !in->rotations)))
            GP_FAIL("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    }
    if (in->shs && st->sh_coeffs < (st->sh_degree + 1) * (st->sh_degree + 1)) GP_FAIL("shs has %d coeffs, degree %d needs more", st->sh_coeffs, st->sh_degree);
    if (in->shs && st->sh_coeffs > 16) GP_FAIL("sh_coeffs %d > 16 unsupported", st->sh_coeffs);
    if (in->shs_rest && (!in->shs || st->sh_coeffs != 16)) GP_FAIL("shs_rest needs shs (= features_dc) and sh_coeffs == 16");
    if (!st->bg || !st->viewmatrix || !st->projmatrix || !st->campos) GP_FAIL("null camera pointers");
    if (in->num_gaussians > 0 && (!in->means3D || !in->opacities)) GP_FAIL("null means3D/opacities");
    d.N = (int)in->num_gaussians;
    d.M = in->shs ? st->sh_coeffs : 0;
    d.D = st->sh_degree;
    d.W = st->image_width;
    d.H = st->image_height;
    d.gx = (d.W + GP_TILE - 1) / GP_TILE;
    d.gy = (d.H + GP_TILE - 1) / GP_TILE;
    d.tanfovx = st->tanfovx;
    d.tanfovy = st->tanfovy;
    d.fx = (float)d.W / (2.f * st->tanfovx);
    d.fy = (float)d.H / (2.f * st->tanfovy);
    d.scale_mod = st->scale_modifier;
    d.late_color = (st->sh_ready_event && in->shs) ? 1 : 0;
    d.visible = nullptr; d.zero_words = nullptr; d.n_zero = 0;
    d.key_hi = 0u; d.key_base = 0u; d.key_culled = 0xFFFFFFFFu; d.key_flag = nullptr; d.key_tag = 0u;
    d.sb_side = nullptr;
    d.raw_opacity = nullptr;
    if (st->raw_activations) {
        if (in->cov3D_precomp || !in->scales) GP_FAIL("raw_activations needs scales + rotations (not cov3D_precomp)");
        d.raw_opacity = in->opacities;
    }
    if (st->depth_key_bits != 0 && st->depth_key_bits != 32) {
        if (st->depth_key_bits < 8 || st->depth_key_bits > 31) GP_FAIL("depth_key_bits must be 0, 32 or 8 .. 31 (got %d)", st->depth_key_bits);
        if (!st->binning_status) GP_FAIL("depth_key_bits needs binning_status (the word a broken promise raises)");
        const uint32_t low = (1u << st->depth_key_bits) - 1u;
        d.key_hi = ~low;
        d.key_base = st->depth_key_base;
        d.key_culled = low;                         // behind (or level with) every visible key of the promised range
        d.key_flag = st->binning_status + 2;        // (the status block's scratch word: 3 words are required then)
        do { d.key_tag = ++g_key_tag; } while (d.key_tag == 0u);
    }
    return 0;
}

// saved-state layouts (recomputed identically in forward and backward)
struct GeomLayout {
    float4* rec;
    uint8_t* clamped;
    float2* sb_side;        // (RasterDims.sb_side)
    size_t bytes;
    GeomLayout(void* base, size_t N) {
        GpCarver c(base);
        rec = c.take<float4>(3 * N + 1);
        clamped = c.take<uint8_t>(N + 1);
        sb_side = c.take<float2>(GP_SB_HOIST == 1 ? N + 1 : 0);
        bytes = c.bytes();
    }
};
struct ImageLayout {
    int2* ranges;
    float* final_T;
    int32_t* n_contrib;
    int32_t* tile_work;    // per tile: largest list position consumed by the forward (backward work estimate)
    uint32_t* order;       // heavy-first launch order of the forward
    size_t bytes;
    uint32_t* total;       // R, written by the scan of the per-Gaussian tile counts (zeroed with `ranges`: it sits behind them)
    ImageLayout(void* base, size_t T, size_t P) {
        GpCarver c(base);
        ranges = c.take<int2>(T + GP_TOTAL_SLOTS / 2);
        total = (uint32_t*)(ranges + T);
        final_T = c.take<float>(P);
        n_contrib = c.take<int32_t>(P);
        tile_work = c.take<int32_t>(T);
        order = c.take<uint32_t>(T);
        bytes = c.bytes();
    }
};

extern "C" int gp_raster_forward(const gp_raster_settings* st, const gp_raster_inputs* in, gp_raster_outputs* out,
                                 gp_raster_saved* saved, gp_alloc_fn alloc, void* alloc_ctx, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    RasterDims d;
    if (make_dims(st, in, d)) return 1;
    if (!out || !out->color || !out->depth || !out->tidx || (d.N > 0 && !out->radii)) GP_FAIL("null output pointers");
    if (!saved || !alloc) GP_FAIL("null saved/alloc");
    const size_t N = (size_t)d.N, T = (size_t)d.gx * d.gy, P = (size_t)d.W * d.H;

    ImageLayout il0(nullptr, T, P);
    void* img = alloc(alloc_ctx, GP_BUF_IMAGE, il0.bytes);
    if (!img) GP_FAIL("allocator returned NULL for IMAGE (%zu B)", il0.bytes);
    ImageLayout il(img, T, P);
    GeomLayout gl0(nullptr, N);
    void* geom = alloc(alloc_ctx, GP_BUF_GEOM, gl0.bytes);
    if (!geom) GP_FAIL("allocator returned NULL for GEOM (%zu B)", gl0.bytes);
    GeomLayout gl(geom, N);
    d.sb_side = gl.sb_side;
    saved->geom = geom; saved->geom_bytes = gl.bytes;
    saved->image = img; saved->image_bytes = il.bytes;
    saved->binning = nullptr; saved->binning_bytes = 0; saved->num_rendered = 0;

    {   // tile ranges + the instance counter's slots behind them start from zero: by the preprocess launch when it has enough
        // threads, else by a memset of its own
        const size_t words = (T + GP_TOTAL_SLOTS / 2) * 2;
        if ((size_t)N >= words) { d.zero_words = (uint32_t*)il.ranges; d.n_zero = (int)words; }
        else GP_HIP_CHECK(hipMemsetAsync(il.ranges, 0, words * sizeof(uint32_t), s));
    }
    d.visible = out->visible;
    uint32_t R = 0;
    uint32_t* point_list = nullptr;
    bool order_done = false;            // (the counting path's scatter launch also writes the forward's tile order)
    if (N > 0) {
        // ---- per-Gaussian temporaries -----------------------------------------------------------
        GpCarver tc0(nullptr);
        const size_t hist_elems = gp_sort_hist_elems(N);
        const size_t scan_elems = gp_scan_tmp_elems(256 * ((N + 4095) / 4096) > N + 1 ? 256 * ((N + 4095) / 4096) : N + 1) + (N + GP_SCAN_TILE - 1) / GP_SCAN_TILE + 64;
        // binning by counting (bin_kernels.hip) whenever the per-tile histogram fits a workgroup's LDS; else duplicate + radix sort
        const bool counting = gp_bin_supported(N, T);
        GpBinPlan bp = {0, 0, 0};
        if (counting) bp = gp_bin_plan(N, T);
        const bool dsort3 = gp_dsort_supported(N);          // depth sort in three 11-bit counting passes (else four 8-bit radix passes)
        auto carve_tmp = [&](GpCarver& c, uint32_t*& k0, uint32_t*& k1, uint32_t*& v0, uint32_t*& v1, uint2*& tiles,
                             uint2*& rects, uint32_t*& tt, uint32_t*& hist, uint32_t*& scan_tmp, uint32_t*& binhist, uint32_t*& dshist) {
            k0 = c.take<uint32_t>(N); k1 = c.take<uint32_t>(N); v0 = c.take<uint32_t>(N); v1 = c.take<uint32_t>(N);
            tiles = c.take<uint2>(N); rects = c.take<uint2>(N); tt = c.take<uint32_t>(N + 1);
            hist = c.take<uint32_t>(hist_elems); scan_tmp = c.take<uint32_t>(scan_elems);
            binhist = c.take<uint32_t>(bp.hist_elems);
            dshist = c.take<uint32_t>(dsort3 ? gp_dsort_hist_elems(N) : 0);
        };
        uint32_t *k0, *k1, *v0, *v1, *tt, *hist, *scan_tmp, *binhist, *dshist;
        uint2 *tiles, *rects;      // per-Gaussian tile rectangle by id, and the same in depth order
        carve_tmp(tc0, k0, k1, v0, v1, tiles, rects, tt, hist, scan_tmp, binhist, dshist);
        void* tmp = alloc(alloc_ctx, GP_BUF_TEMP, tc0.bytes());
        if (!tmp) GP_FAIL("allocator returned NULL for TEMP (%zu B)", tc0.bytes());
        GpCarver tc(tmp);
        carve_tmp(tc, k0, k1, v0, v1, tiles, rects, tt, hist, scan_tmp, binhist, dshist);

        {
            GpProfScope _p("preprocess_fwd", s);
            const bool al16 = (((uintptr_t)in->shs | (uintptr_t)in->shs_rest) & 15) == 0;
            auto kern = gp_preprocess_fwd_kernel;
            if (in->shs && in->shs_rest) {
                if (!al16) GP_FAIL("shs/shs_rest must be 16-byte aligned");
                kern = gp_preprocess_fwd_split_kernel;
            } else if (in->shs && d.M == 16 && al16) {
                kern = gp_preprocess_fwd_sh16_kernel;
            }
            hipLaunchKernelGGL(kern, dim3(gp_blocks(N, 256)), dim3(256), 0, s, d, in->means3D, in->scales, in->rotations,
                               in->opacities, in->shs, in->shs_rest, in->colors_precomp, in->cov3D_precomp, st->viewmatrix,
                               st->projmatrix, st->campos, out->radii, gl.rec, k0, tiles, gl.clamped);
            GP_LAUNCH_CHECK();
        }
        GpSortBufs sb;
        sb.k[0] = k0; sb.k[1] = k1; sb.v[0] = v0; sb.v[1] = v1; sb.hist = hist; sb.scan_tmp = scan_tmp; sb.scan_tmp_elems = scan_elems;
        int r1;
        // (values = 0 .. N-1, generated by the first pass)
        // the last pass also carries every Gaussian's tile rectangle + tile count into depth order (was a launch of its own)
        const GpSortEpilogue ep = {tiles, rects, tt};
        const int key_bits = d.key_hi ? st->depth_key_bits : 32;       // (a promised key range: fewer passes, same order)
        if (st->depth_key_range) {      // {min, max} key of the visible Gaussians, for a caller that sizes depth_key_bits from it
            GP_HIP_CHECK(hipMemsetAsync(st->depth_key_range, 0xFF, 4, s));
            GP_HIP_CHECK(hipMemsetAsync(st->depth_key_range + 1, 0, 4, s));
            hipLaunchKernelGGL(gp_key_range_kernel, dim3(gp_blocks(N, 1024)), dim3(256), 0, s, (const uint32_t*)k0, (const int32_t*)out->radii, (int)N,
                               d.key_base, st->depth_key_range);
            GP_LAUNCH_CHECK();
        }
        { GpProfScope _p("depth_sort", s); r1 = (dsort3 && key_bits == 32) ? gp_depth_sort3(sb, N, dshist, s, &ep) : gp_radix_sort_pairs(sb, N, key_bits, s, true, &ep); }
        if (r1 < 0) return 1;
        const uint32_t* sorted_ids = sb.v[r1];
        // tile counts -> instance offsets: scanned inside blocks here, finished by the duplicate kernel (one launch, not three)
        uint32_t* block_sums = scan_tmp;
        if (counting) {         // per-(block, tile) instance counts; the blocks' totals go into the R slots
            GpProfScope _p("bin_count", s);
            if (gp_bin_count(bp, N, d.gx, T, rects, binhist, il.total, s)) return 1;
        } else if (gp_scan_blocks_u32(tt, N, block_sums, il.total, s)) return 1;
        const bool capacity_mode = st->binning_capacity > 0;
        if (capacity_mode) {    // no host synchronisation: everything below is sized by the caller's capacity
            if (!st->binning_status) GP_FAIL("binning_capacity needs binning_status (device, 2 words)");
            if (st->binning_capacity > 0x7FFFFF00ll) GP_FAIL("binning_capacity too large");
            R = (uint32_t)st->binning_capacity;          // (the status word and the sentinel keys are written by the duplicate launch)
        } else {
            uint32_t slots[GP_TOTAL_SLOTS];
            GP_HIP_CHECK(hipMemcpyAsync(slots, il.total, sizeof(slots), hipMemcpyDeviceToHost, s));
            GP_HIP_CHECK(hipStreamSynchronize(s));
            uint64_t Rsum = 0;
            for (int k = 0; k < GP_TOTAL_SLOTS; ++k) Rsum += slots[k];
            if (Rsum > 0x7FFFFF00ull) GP_FAIL("too many tile-splat instances (%llu)", (unsigned long long)Rsum);
            R = (uint32_t)Rsum;
            if (R > 0x7FFFFF00u) GP_FAIL("too many tile-splat instances (%u)", R);
            if (st->binning_status && (!counting || R == 0)) {   // exact mode reports R too (a caller sizing its capacity reads it from here; the counting path's scatter writes it)
                hipLaunchKernelGGL(gp_binning_status_kernel, dim3(1), dim3(1), 0, s, il.total, 0xFFFFFFFFu, st->binning_status, d.key_tag);
                GP_LAUNCH_CHECK();
            }
        }

        if (R > 0 && counting) {
            const size_t bin_bytes = gp_align_up((size_t)R * 4, 256) + gp_align_up(2 * (size_t)R + 8, 256);
            void* bin = alloc(alloc_ctx, GP_BUF_BINNING, bin_bytes);
            if (!bin) GP_FAIL("allocator returned NULL for BINNING");
            point_list = (uint32_t*)bin;
            saved->binning = bin; saved->binning_bytes = bin_bytes;
            if (gp_bin_scatter(bp, N, d.gx, T, sorted_ids, rects, binhist, point_list, R, il.ranges, st->binning_status, il.order, d.key_tag, s)) return 1;
            order_done = true;
        } else if (R > 0) {
            // capacity mode pads the keys with 0xFFFFFFFF: its low `tbits` bits must sort behind every real tile id
            const int tbits = tile_bits_for((int)T + (capacity_mode ? 1 : 0));
            const int passes = (tbits + 7) / 8;
            const int res = passes & 1;  // buffer pair holding the sorted result
            // BINNING = point_list[R] (u32) followed by smask[R] (u16: which 4x4 sub-blocks of its tile an instance can touch)
            const size_t bin_bytes = gp_align_up((size_t)R * 4, 256) + gp_align_up(2 * (size_t)R + 8, 256);
            void* bin = alloc(alloc_ctx, GP_BUF_BINNING, bin_bytes);
            if (!bin) GP_FAIL("allocator returned NULL for BINNING");
            point_list = (uint32_t*)bin;
            saved->binning = bin; saved->binning_bytes = bin_bytes;
            const size_t bh = gp_sort_hist_elems(R), bs = gp_scan_tmp_elems(256 * (((size_t)R + 4095) / 4096));
            auto carve_bin = [&](GpCarver& c, uint32_t*& bk0, uint32_t*& bk1, uint32_t*& bvo, uint32_t*& bhist, uint32_t*& bscan) {
                bk0 = c.take<uint32_t>(R); bk1 = c.take<uint32_t>(R); bvo = c.take<uint32_t>(R);
                bhist = c.take<uint32_t>(bh); bscan = c.take<uint32_t>(bs);
            };
            uint32_t *bk0, *bk1, *bvo, *bhist, *bscan;
            GpCarver bc0(nullptr);
            carve_bin(bc0, bk0, bk1, bvo, bhist, bscan);
            void* btmp = alloc(alloc_ctx, GP_BUF_TEMP, bc0.bytes());
            if (!btmp) GP_FAIL("allocator returned NULL for TEMP (%zu B)", bc0.bytes());
            GpCarver bc(btmp);
            carve_bin(bc, bk0, bk1, bvo, bhist, bscan);
            GpSortBufs tb;
            tb.k[0] = bk0; tb.k[1] = bk1;
            tb.v[res] = point_list; tb.v[res ^ 1] = bvo;
            tb.hist = bhist; tb.scan_tmp = bscan; tb.scan_tmp_elems = bs;
            { GpProfScope _p("duplicate", s);
            const unsigned ndup = gp_blocks(N, 256);
            hipLaunchKernelGGL(gp_duplicate_kernel, dim3(ndup + (capacity_mode ? gp_blocks(R, 4096) : 0u)), dim3(256), 0, s, d, sorted_ids, tt,
                               (const uint32_t*)block_sums, (const uint32_t*)il.total, rects, tb.k[0], tb.v[0], R, st->binning_status, ndup, d.key_tag);
            GP_LAUNCH_CHECK(); }
            int r2;
            { GpProfScope _p("tile_sort", s); r2 = gp_radix_sort_pairs(tb, R, tbits, s); }
            if (r2 < 0) return 1;
            if (r2 != res) GP_FAIL("internal: sort parity mismatch");
            { GpProfScope _p("tile_ranges", s);
        hipLaunchKernelGGL(gp_tile_ranges_kernel, dim3(gp_blocks(R, 256)), dim3(256), 0, s, tb.k[r2], R, (uint32_t)T, il.ranges);
            GP_LAUNCH_CHECK(); }
        }
    }
    if (N == 0 && st->binning_status)      // nothing to bin: {R, overflow} = {0, 0} (a stale overflow word would make Adam skip the step)
        GP_HIP_CHECK(hipMemsetAsync(st->binning_status, 0, 2 * sizeof(uint32_t), s));
    saved->num_rendered = (int64_t)R;
    if (!order_done) {
        hipLaunchKernelGGL(gp_tile_order_kernel, dim3(1), dim3(1024), 0, s, il.ranges, (const int32_t*)nullptr, (int)T, il.order);
        GP_LAUNCH_CHECK();
    }
    if (d.late_color && N > 0) {    // the SH coefficients may still be in flight (parameter all-gather): wait here, not at the top
        GP_HIP_CHECK(hipStreamWaitEvent(s, (hipEvent_t)st->sh_ready_event, 0));
        GpProfScope _p("sh_color", s);
        const bool al16 = (((uintptr_t)in->shs | (uintptr_t)in->shs_rest) & 15) == 0;
        auto kern = gp_sh_color_kernel;
        if (in->shs_rest) kern = gp_sh_color_split_kernel;                 // (alignment was checked for the preprocess launch)
        else if (d.M == 16 && al16) kern = gp_sh_color_sh16_kernel;
        GeomLayout glc(saved->geom, N);
        hipLaunchKernelGGL(kern, dim3(gp_blocks(N, 256)), dim3(256), 0, s, d, in->means3D, in->shs, in->shs_rest, st->campos,
                           (const int32_t*)out->radii, glc.rec, glc.clamped);
        GP_LAUNCH_CHECK();
    }
    { GpProfScope _p("composite_fwd", s, 1);
        hipLaunchKernelGGL((gp_debug_get(0) == 3 ? gp_composite_fwd_count_kernel : gp_debug_get(0) == 2 ? gp_composite_fwd_sbc_kernel : gp_composite_fwd_sb_kernel), dim3((unsigned)T), dim3(256), 0, s, d, il.ranges, point_list, gl.rec, st->bg,
                       out->color, out->depth, out->tidx, il.final_T, il.n_contrib, il.order, il.tile_work,
                       point_list ? (uint16_t*)((uint8_t*)point_list + gp_align_up((size_t)R * 4, 256)) : (uint16_t*)nullptr);
    GP_LAUNCH_CHECK(); }
    return 0;
}

extern "C" int gp_raster_backward(const gp_raster_settings* st, const gp_raster_inputs* in, const gp_raster_outputs* fwd,
                                  const gp_raster_saved* saved, const float* dL_dcolor, const float* dL_ddepth,
                                  gp_raster_grads* g, gp_alloc_fn alloc, void* alloc_ctx, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    RasterDims d;
    if (make_dims(st, in, d)) return 1;
    if (!fwd || !saved || !g || !alloc || !dL_dcolor) GP_FAIL("null argument");
    const size_t N = (size_t)d.N, T = (size_t)d.gx * d.gy, P = (size_t)d.W * d.H;
    if (N == 0) return 0;
    if (!g->dL_dmeans3D || !g->dL_dmeans2D || !g->dL_dopacities) GP_FAIL("null gradient outputs");
    if (in->shs ? (!g->dL_dshs && !g->adam_shs) : !g->dL_dcolors_precomp) GP_FAIL("missing colour gradient output");
    if (in->cov3D_precomp ? !g->dL_dcov3D_precomp : (!g->dL_dscales || !g->dL_drotations)) GP_FAIL("missing covariance gradient output");
    if (!saved->geom || !saved->image) GP_FAIL("saved state missing");
    AdamFuseDev af;
    memset(&af, 0, sizeof(af));
    if (g->adam_shs) {      // the optimizer step of (shs, shs_rest) inside this kernel: validated before anything is written
        const gp_adam_fuse* a = g->adam_shs;
        if (!(in->shs && in->shs_rest) || d.M != 16) GP_FAIL("adam_shs needs the split SH layout (shs [N,1,3] + shs_rest [N,15,3])");
        if (!a->exp_avg_dc || !a->exp_avg_sq_dc || !a->exp_avg_rest || !a->exp_avg_sq_rest) GP_FAIL("adam_shs: null moment pointer");
        if ((((uintptr_t)in->shs_rest | (uintptr_t)a->exp_avg_rest | (uintptr_t)a->exp_avg_sq_rest) & 15) != 0)
            GP_FAIL("adam_shs: shs_rest and its moments must be 16-byte aligned");
        if (a->step < 1) GP_FAIL("adam_shs: bad step");
        const float bc1 = 1.f - powf(a->beta1, (float)a->step);      // as gp_adam_step_multi
        af.p_dc = (float*)in->shs; af.m_dc = a->exp_avg_dc; af.v_dc = a->exp_avg_sq_dc;
        af.p_rest = (float*)in->shs_rest; af.m_rest = a->exp_avg_rest; af.v_rest = a->exp_avg_sq_rest;
        af.step_dc = a->lr_dc / bc1; af.step_rest = a->lr_rest / bc1;
        af.b1 = a->beta1; af.b2 = a->beta2; af.eps = a->eps;
        af.bc2_sqrt = sqrtf(1.f - powf(a->beta2, (float)a->step));
        af.skip = a->skip_flag; af.on = 1;
    }
    GeomLayout gl(saved->geom, N);
    ImageLayout il(saved->image, T, P);
    const uint32_t R = (uint32_t)saved->num_rendered;
    const uint32_t* point_list = (const uint32_t*)saved->binning;
    if (R > 0 && !point_list) GP_FAIL("saved binning state missing");

    const size_t acc_floats = (size_t)GP_ACC_STRIDE * N;
    float* acc = (float*)alloc(alloc_ctx, GP_BUF_TEMP, gp_align_up(acc_floats * 4, 256) + gp_align_up(T * 4, 256));
    if (!acc) GP_FAIL("allocator returned NULL for TEMP");
    uint32_t* order_bwd = (uint32_t*)((char*)acc + gp_align_up(acc_floats * 4, 256));
    if (R > 0) {        // (tile order for the composite backward + the accumulators' fill, one launch)
        const unsigned zb = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (acc_floats / 4 + 1023) / 1024));
        hipLaunchKernelGGL(gp_bwd_prologue_kernel, dim3(1 + zb), dim3(1024), 0, s, il.ranges, il.tile_work, (int)T, order_bwd, acc, acc_floats);
        GP_LAUNCH_CHECK();
    } else {
        GP_HIP_CHECK(hipMemsetAsync(acc, 0, acc_floats * 4, s));
    }
    float* g_mean2D = acc;   // AoS, stride GP_ACC_STRIDE
    float* g_conic = acc + 2;
    float* g_opacity = acc + 5;
    float* g_color = acc + 6;
    float* g_depth = acc + 9;
    if (R > 0) {
        static thread_local int ablate_set = 0;
        if (gp_debug_get(1) != ablate_set) { ablate_set = gp_debug_get(1); if (gp_bwd_set_ablate(ablate_set)) GP_FAIL("bwd ablate flag"); }
        GpProfScope _p("composite_bwd", s);
        // gp_debug_option(7, v): 0 = the quadrant kernel (shipped); 3 = the sub-block kernel of round 5 (4x4 culling: 1.5 evaluated pairs
        // per contributing one instead of 3.0, and slower -- raster_kernels.hip), 2 = its counting variant
        // (g_pair_counters[2], [3]).  The sub-block kernel packs four bits above the Gaussian id: N < GP_BWD_SB_MAX_N.
        const int variant = gp_debug_get(7);
        auto kern = dL_ddepth ? gp_composite_bwd_depth_kernel : gp_composite_bwd_kernel;
        if (variant == 3 && N < (size_t)GP_BWD_SB_MAX_N) kern = dL_ddepth ? gp_composite_bwd_sb_depth_kernel : gp_composite_bwd_sb_kernel;
        else if (variant == 2 && !dL_ddepth && N < (size_t)GP_BWD_SB_MAX_N) kern = gp_composite_bwd_sb_count_kernel;
        hipLaunchKernelGGL(kern, dim3((unsigned)((T + 7) / 8 * 8) * GP_BWD_PARTS),
                           dim3(64), 0, s, d, il.ranges, point_list, (const uint16_t*)((const uint8_t*)point_list + gp_align_up((size_t)R * 4, 256)), gl.rec,
                           st->bg, (const float*)fwd->color, (const float*)fwd->depth, (const float*)il.final_T,
                           (const int32_t*)il.n_contrib, dL_dcolor, dL_ddepth, g_mean2D, g_conic, g_opacity, g_color, g_depth, order_bwd);
        GP_LAUNCH_CHECK();
    }
    {
        GpProfScope _p("preprocess_bwd", s);
        const bool al16 = (((uintptr_t)in->shs | (uintptr_t)in->shs_rest | (uintptr_t)g->dL_dshs | (uintptr_t)g->dL_dshs_rest) & 15) == 0;
        auto kern = gp_preprocess_bwd_kernel;
        if (in->shs && in->shs_rest) {
            if (!al16 || (!g->dL_dshs_rest && !af.on)) GP_FAIL("split SH mode needs 16-byte aligned tensors and dL_dshs_rest");
            kern = gp_preprocess_bwd_split_kernel;
        } else if (in->shs && d.M == 16 && al16) {
            kern = gp_preprocess_bwd_sh16_kernel;
        }
        if (g->accumulate_shs && kern == gp_preprocess_bwd_kernel)
            GP_FAIL("accumulate_shs needs the staged SH kernels (16 coeffs, 16-byte aligned)");      // (before anything is written)
        hipLaunchKernelGGL(kern, dim3(gp_blocks(N, 256)), dim3(256), 0, s, d, in->means3D, in->scales, in->rotations, in->shs,
                           in->shs_rest, in->cov3D_precomp, st->viewmatrix, st->projmatrix, st->campos, fwd->radii, gl.clamped,
                           g_mean2D, g_conic, g_opacity, g_color, g_depth, g->dL_dmeans3D, g->dL_dmeans2D, g->dL_dshs,
                           g->dL_dshs_rest, in->shs ? nullptr : g->dL_dcolors_precomp, g->dL_dopacities, g->dL_dscales,
                           g->dL_drotations, g->dL_dcov3D_precomp, (kern != gp_preprocess_bwd_kernel) ? g->accumulate_shs : 0, af);
        GP_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gp_debug_counters(uint64_t* out4) {
    if (!out4) GP_FAIL("null argument");
    unsigned long long v[4];
    if (gp_pair_counters_read(v)) GP_FAIL("gp_debug_counters: device read failed");
    for (int k = 0; k < 4; ++k) out4[k] = v[k];
    return 0;
}

extern "C" int gp_raster_mark_visible(int64_t n, const float* means3D, const float* viewmatrix, uint8_t* present,
                                      gp_stream_t stream_) {
    if (n < 0 || n > 0x7FFFFFF0LL) GP_FAIL("n out of range");
    if (n == 0) return 0;
    if (!means3D || !viewmatrix || !present) GP_FAIL("null argument");
    hipLaunchKernelGGL(gp_mark_visible_kernel, dim3(gp_blocks((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream_, (int)n,
                       means3D, viewmatrix, present);
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_raster_debug_binning(const gp_raster_settings* st, const gp_raster_saved* saved, uint32_t* point_list,
                                       int32_t* ranges, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (!st || !saved || !ranges) GP_FAIL("null argument");
    const int gx = (st->image_width + GP_TILE - 1) / GP_TILE, gy = (st->image_height + GP_TILE - 1) / GP_TILE;
    const size_t T = (size_t)gx * gy, P = (size_t)st->image_width * st->image_height;
    ImageLayout il(saved->image, T, P);
    GP_HIP_CHECK(hipMemcpyAsync(ranges, il.ranges, T * sizeof(int2), hipMemcpyDeviceToDevice, s));
    if (saved->num_rendered > 0) {
        if (!point_list) GP_FAIL("null point_list");
        GP_HIP_CHECK(hipMemcpyAsync(point_list, saved->binning, (size_t)saved->num_rendered * 4, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

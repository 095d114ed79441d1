// gp_common.h -- internal helpers shared by the HIP translation units of libgp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gp_hip.h"

#define GP_TILE 16
#define GP_WAVE 64

extern thread_local char gp_err_buf[512];

#define GP_FAIL(...)                                              \
    do {                                                          \
        snprintf(gp_err_buf, sizeof(gp_err_buf), __VA_ARGS__);    \
        return 1;                                                 \
    } while (0)

#define GP_HIP_CHECK(expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) GP_FAIL("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define GP_LAUNCH_CHECK() GP_HIP_CHECK(hipGetLastError())

static inline size_t gp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// carve typed arrays out of one allocation, 256-B aligned
struct GpCarver {
    char* base;
    size_t off;
    explicit GpCarver(void* p) : base((char*)p), off(0) {}
    template <typename T>
    T* take(size_t n) {
        off = gp_align_up(off, 256);
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
    size_t bytes() const { return gp_align_up(off, 256); }
};

// ---- launch geometry -------------------------------------------------------------------------
static inline unsigned gp_blocks(size_t n, unsigned per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// ---- device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ unsigned gp_mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ unsigned long long gp_readfirstlane64(unsigned long long v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// exp(x) for x <= 0 via the hardware exp2 (v_exp_f32, ~1 ulp)
__device__ __forceinline__ float gp_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// ---- optional kernel timing (gp_profile.hip) ----------------------------------------------------
bool gp_prof_on();
void* gp_prof_begin(const char* name, hipStream_t s, int level);
void gp_prof_end(void* h, hipStream_t s);
struct GpProfScope {
    void* h; hipStream_t s;
    GpProfScope(const char* name, hipStream_t st, int level = 2) : h(gp_prof_begin(name, st, level)), s(st) {}
    ~GpProfScope() { gp_prof_end(h, s); }
};

int gp_debug_get(int key);

// Inclusive wave-wide prefix sum on the DPP network (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast 15 / 31 across the
// rows): six VALU instructions.  `__shfl_up` is a ds_bpermute -- a scan built from it is six DEPENDENT LDS round trips.
__device__ __forceinline__ int gp_wave_scan_add(int x) {
#define GP_DPP_(v, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rmask), 0xf, false)
    x += GP_DPP_(x, 0x111, 0xf); x += GP_DPP_(x, 0x112, 0xf); x += GP_DPP_(x, 0x114, 0xf); x += GP_DPP_(x, 0x118, 0xf);
    x += GP_DPP_(x, 0x142, 0xa); x += GP_DPP_(x, 0x143, 0xc);
#undef GP_DPP_
    return x;
}

// torch.optim.Adam's update of one element (eps added after the bias-corrected sqrt) -- the ONE statement of the arithmetic,
// shared by gp_adam_multi_kernel and the update fused into the rasterizer backward, so the two are bit-identical.
__device__ __forceinline__ void gp_adam_update(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step_size,
                                               float bc2_sqrt) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
}
// Adam fused into the preprocess backward (gp_adam_fuse of include/gp_hip.h, resolved by the host)
struct AdamFuseDev {
    float* p_dc; float* m_dc; float* v_dc;
    float* p_rest; float* m_rest; float* v_rest;
    float step_dc, step_rest, b1, b2, eps, bc2_sqrt;     // step_* = lr / (1 - b1^t)
    const uint32_t* skip;
    int on;
};
   // diagnostics knobs (gp_profile.hip)

// ---- sub-module entry points (host side, defined in the .hip files) -----------------------------
size_t gp_scan_tmp_elems(size_t n);

struct GpSortBufs {
    uint32_t *k[2], *v[2];
    uint32_t* hist;      // 256 * nblocks
    uint32_t* scan_tmp;  // gp_scan_tmp_elems(256*nblocks)
    size_t scan_tmp_elems;
};
size_t gp_sort_hist_elems(size_t n);
// stable LSD radix sort of (key,val) pairs on key bits [0, nbits). Input in k[0]/v[0]; returns the
// index (0/1) of the buffer pair that holds the result, or -1 on error.
// iota_vals: the input values are 0 .. n-1 (generated by the first pass; v[0] need not be filled).
// epilogue of the LAST pass (optional): for every sorted value v at output position pos,  sorted_out[pos] = by_value[v]  and
// count_out[pos] = (that .y & 0xFFFF) * (.y >> 16)  -- the depth sort carries each Gaussian's tile rectangle and its tile count
// into depth order itself (a launch of its own in round 2: gp_gather_tiles_kernel, 14 us)
struct GpSortEpilogue {
    const uint2* by_value;
    uint2* sorted_out;
    uint32_t* count_out;
};
int gp_radix_sort_pairs(GpSortBufs& b, size_t n, int nbits, hipStream_t s, bool iota_vals = false, const GpSortEpilogue* epilogue = nullptr);
int gp_scan_blocks_u32(uint32_t* data, size_t n, uint32_t* block_sums, uint32_t* total, hipStream_t s);

// depth sort in three 11-bit counting passes (bin_kernels.hip): keys in b.k[0], values = indices; returns the buffer pair (0 / 1)
// holding the result, -1 on error.  `hist`: gp_dsort_hist_elems(n) words of scratch.
bool gp_dsort_supported(size_t n);
size_t gp_dsort_hist_elems(size_t n);
int gp_depth_sort3(GpSortBufs& b, size_t n, uint32_t* hist, hipStream_t s, const GpSortEpilogue* epilogue);

// tile binning by counting (bin_kernels.hip): per-tile lists in depth order from the depth-ordered (id, tile rectangle) arrays,
// three launches, no instance keys.  Needs the whole per-tile histogram in a workgroup's LDS: T <= GP_BIN_MAX_TILES.
#define GP_BIN_MAX_TILES 8192
struct GpBinPlan { int G, NB; size_t hist_elems; };      // Gaussians per block, blocks, words of histogram scratch
bool gp_bin_supported(size_t N, size_t T);
GpBinPlan gp_bin_plan(size_t N, size_t T);
int gp_bin_count(const GpBinPlan& p, size_t N, int gx, size_t T, const uint2* rect_sorted, uint32_t* hist, uint32_t* total_slots, hipStream_t s);
int gp_bin_scatter(const GpBinPlan& p, size_t N, int gx, size_t T, const uint32_t* sorted_ids, const uint2* rect_sorted, uint32_t* hist,
                   uint32_t* point_list, uint32_t capacity, int2* ranges, uint32_t* status, uint32_t* order, uint32_t key_tag, hipStream_t s);
// (`order`, optional: the composite forward's heavy-first tile order, computed by one extra workgroup of the scatter launch)
#define GP_SCAN_TILE 2048      // elements per block of gp_scan_blocks_u32
#define GP_TOTAL_SLOTS 16       // gp_scan_blocks_u32 spreads its grand total over this many words (block b adds into word b % 16:
                               // 500 atomics on ONE address cost 5 us); readers add the words up
__device__ __forceinline__ uint32_t gp_total_of(const uint32_t* __restrict__ slots) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < GP_TOTAL_SLOTS; ++k) r += slots[k];
    return r;
}

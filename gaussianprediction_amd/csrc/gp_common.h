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

int gp_debug_get(int key);   // diagnostics knobs (gp_profile.hip)

// ---- sub-module entry points (host side, defined in the .hip files) -----------------------------
int gp_scan_exclusive_u32(uint32_t* data, size_t n, uint32_t* tmp, size_t tmp_elems, hipStream_t s);
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
int gp_radix_sort_pairs(GpSortBufs& b, size_t n, int nbits, hipStream_t s);

// gp_profile.hip -- optional per-kernel timing with hipEvent pairs recorded on the launch stream
// (torch.cuda.Event only sees torch's current stream; these see exactly the kernels they bracket).
#include <mutex>
#include <string>
#include <vector>

#include "gp_common.h"

struct ProfRec { const char* name; hipEvent_t a, b; };
static std::mutex g_mu;
static int g_level = 0;   // 0 off, 1 = only the kernels tagged level 1 (roofline kernel), 2 = every kernel
static unsigned g_seen = 0;   // level-1 launches since gp_profile_enable (the first one is always bracketed)
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

bool gp_prof_on() { return g_level > 0; }

void* gp_prof_begin(const char* name, hipStream_t s, int level) {
    if (g_level < level) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    // level 1 (one tagged kernel inside a timed region): every P-th launch only, P = gp_debug_option(12, P) -- an event pair is a
    // ~10 us bubble on the stream, and the region being timed should not pay it at every step for an average that 1 / P of the
    // launches give as well (the count of bracketed launches comes back in gp_profile_entry.launches)
    if (g_level == 1) {
        const int period = gp_debug_get(12);
        if (period > 1 && (g_seen++ % (unsigned)period) != 0) return nullptr;
    }
    ProfRec r{name, take_event(), take_event()};
    if (!r.a || !r.b) return nullptr;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (void*)(uintptr_t)g_recs.size();  // 1-based index
}
void gp_prof_end(void* h, hipStream_t s) {
    if (!h) return;
    std::lock_guard<std::mutex> lk(g_mu);
    size_t i = (size_t)(uintptr_t)h - 1;
    if (i < g_recs.size()) (void)hipEventRecord(g_recs[i].b, s);
}

extern "C" int gp_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_level = on < 0 ? 0 : on;
    g_seen = 0;
    return 0;
}

extern "C" int gp_profile_collect(gp_profile_entry* out, int max_entries, int* n_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!n_out) GP_FAIL("null n_out");
    int n = 0;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            int k = 0;
            for (; k < n; ++k)
                if (strncmp(out[k].name, r.name, sizeof(out[k].name)) == 0) break;
            if (k == n && n < max_entries && out) {
                memset(&out[n], 0, sizeof(out[n]));
                strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
                ++n;
            }
            if (k < n) { out[k].launches += 1; out[k].total_ms += ms; }
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    *n_out = n;
    return 0;
}

// ---- peak microbenchmarks ("verify on box with a stream-copy kernel and use the measured
// copy peak as the denominator alongside the vendor number") ------------------------------------------------------
// Timed by the caller through gp_profile ("mb_copy", "mb_read", "mb_mfma_*").

typedef float mb_f4 __attribute__((ext_vector_type(4)));
typedef float mb_f16v __attribute__((ext_vector_type(16)));
typedef _Float16 mb_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 mb_b8 __attribute__((ext_vector_type(8)));

#define MB_THREADS 256
#define MB_UNROLL 4

__global__ __launch_bounds__(MB_THREADS) void gp_mb_copy_kernel(mb_f4* __restrict__ dst, const mb_f4* __restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * MB_THREADS;
    size_t i = (size_t)blockIdx.x * MB_THREADS + threadIdx.x;
    for (; i + (MB_UNROLL - 1) * stride < n16; i += MB_UNROLL * stride) {
        mb_f4 v[MB_UNROLL];
#pragma unroll
        for (int u = 0; u < MB_UNROLL; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
        for (int u = 0; u < MB_UNROLL; ++u) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
    }
    for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}

// Variants of the copy (gp_debug_option(2, v); tools/copy_peak_sweep.py): which access shape reaches the part's copy peak is
// measured, not assumed (profiles/r03_copy_peak_sweep.jsonl, 1 / 4 GiB buffers): the plain float4 copy with ONE vector per
// thread and a one-shot grid reaches 6.29 / 6.31 TB/s (the guide's figure); every persistent grid-stride form stays at
// 4.5 - 5.8 TB/s (round 2's default, non-temporal grid-stride: 4.9).  Default (v = 0) is therefore the one-shot copy;
// 1: temporal loads and stores, grid-stride; 2: the one-shot copy; 3: each workgroup owns one CONTIGUOUS chunk; 4: non-temporal
// loads + temporal stores, grid-stride; 5: round 2's kernel (non-temporal both ways, grid-stride).
template <int VARIANT>
__global__ __launch_bounds__(MB_THREADS) void gp_mb_copy_var_kernel(mb_f4* __restrict__ dst, const mb_f4* __restrict__ src, size_t n16) {
    if constexpr (VARIANT == 2) {
        const size_t i = (size_t)blockIdx.x * MB_THREADS + threadIdx.x;
        if (i < n16) dst[i] = src[i];
    } else if constexpr (VARIANT == 3) {
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
        size_t i = lo + threadIdx.x;
        for (; i + (MB_UNROLL - 1) * MB_THREADS < hi; i += MB_UNROLL * MB_THREADS) {
            mb_f4 v[MB_UNROLL];
#pragma unroll
            for (int u = 0; u < MB_UNROLL; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * MB_THREADS]);
#pragma unroll
            for (int u = 0; u < MB_UNROLL; ++u) __builtin_nontemporal_store(v[u], &dst[i + u * MB_THREADS]);
        }
        for (; i < hi; i += MB_THREADS) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
    } else {
        const size_t stride = (size_t)gridDim.x * MB_THREADS;
        size_t i = (size_t)blockIdx.x * MB_THREADS + threadIdx.x;
        for (; i + (MB_UNROLL - 1) * stride < n16; i += MB_UNROLL * stride) {
            mb_f4 v[MB_UNROLL];
#pragma unroll
            for (int u = 0; u < MB_UNROLL; ++u) v[u] = VARIANT == 1 ? src[i + u * stride] : __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
            for (int u = 0; u < MB_UNROLL; ++u) dst[i + u * stride] = v[u];
        }
        for (; i < n16; i += stride) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(MB_THREADS) void gp_mb_read_kernel(const mb_f4* __restrict__ src, size_t n16, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * MB_THREADS;
    size_t i = (size_t)blockIdx.x * MB_THREADS + threadIdx.x;
    mb_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (MB_UNROLL - 1) * stride < n16; i += MB_UNROLL * stride) {
        mb_f4 v[MB_UNROLL];
#pragma unroll
        for (int u = 0; u < MB_UNROLL; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
        for (int u = 0; u < MB_UNROLL; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(&src[i]);
    const float t = acc.x + acc.y + acc.z + acc.w;
    if (t == 123456.789f) sink[0] = t;      // never true for the zero-filled source; keeps the loads alive
}

// Four independent accumulator chains of back-to-back MFMAs per wave, 4 waves per workgroup, 8 workgroups per CU.
template <int DTYPE>
__global__ __launch_bounds__(256) void gp_mb_mfma_kernel(int iters, float* __restrict__ sink) {
    mb_f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const float seed = (float)(threadIdx.x & 7) * 0.125f;
    if constexpr (DTYPE == 0) {
        float a = seed, b = 1.f - seed;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
    } else if constexpr (DTYPE == 1) {
        mb_h8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)seed; b[k] = (_Float16)(1.f - seed); }
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
    } else {
        mb_b8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)seed; b[k] = (__bf16)(1.f - seed); }
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    }
    mb_f16v c = c0 + c1 + c2 + c3;
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += c[k];
    if (t == 123456.789f) sink[0] = t;
}

static int mb_grid() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus * 8;
}

extern "C" int gp_microbench_copy(void* dst, const void* src, size_t bytes, void* stream) {
    if (!src || !dst || (bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) GP_FAIL("gp_microbench_copy: 16-byte aligned buffers and size required");
    hipStream_t s = (hipStream_t)stream;
    {
        GpProfScope _p("mb_copy", s, 1);
        const int var = gp_debug_get(2) & 7, gmul = gp_debug_get(2) >> 3;      // (bits 3..: grid = CUs x 8 x 2^gmul)
        const unsigned grid = (unsigned)mb_grid() << gmul;
        const size_t n16 = bytes / 16;
        if (var == 5) hipLaunchKernelGGL(gp_mb_copy_kernel, dim3(grid), dim3(MB_THREADS), 0, s, (mb_f4*)dst, (const mb_f4*)src, n16);
        else if (var == 1) hipLaunchKernelGGL(gp_mb_copy_var_kernel<1>, dim3(grid), dim3(MB_THREADS), 0, s, (mb_f4*)dst, (const mb_f4*)src, n16);
        else if (var == 3) hipLaunchKernelGGL(gp_mb_copy_var_kernel<3>, dim3(grid), dim3(MB_THREADS), 0, s, (mb_f4*)dst, (const mb_f4*)src, n16);
        else if (var == 4) hipLaunchKernelGGL(gp_mb_copy_var_kernel<4>, dim3(grid), dim3(MB_THREADS), 0, s, (mb_f4*)dst, (const mb_f4*)src, n16);
        else hipLaunchKernelGGL(gp_mb_copy_var_kernel<2>, dim3((unsigned)((n16 + MB_THREADS - 1) / MB_THREADS)), dim3(MB_THREADS), 0, s, (mb_f4*)dst, (const mb_f4*)src, n16);
    }
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_microbench_read(const void* src, size_t bytes, float* sink, void* stream) {
    if (!src || !sink || (bytes & 15) || ((uintptr_t)src & 15)) GP_FAIL("gp_microbench_read: 16-byte aligned buffer and size, and a sink, required");
    hipStream_t s = (hipStream_t)stream;
    {
        GpProfScope _p("mb_read", s, 1);
        hipLaunchKernelGGL(gp_mb_read_kernel, dim3(mb_grid()), dim3(MB_THREADS), 0, s, (const mb_f4*)src, bytes / 16, sink);
    }
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_microbench_mfma(int dtype, int iters, float* sink, double* flop_out, void* stream) {
    if (!sink || iters <= 0 || dtype < 0 || dtype > 2) GP_FAIL("gp_microbench_mfma: dtype 0 (f32) / 1 (f16) / 2 (bf16), iters > 0, sink required");
    hipStream_t s = (hipStream_t)stream;
    const int grid = mb_grid();
    const double flop_per_mfma = dtype == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    if (flop_out) *flop_out = flop_per_mfma * 4.0 * (double)iters * 4.0 /*waves*/ * (double)grid;
    {
        GpProfScope _p(dtype == 0 ? "mb_mfma_f32" : dtype == 1 ? "mb_mfma_f16" : "mb_mfma_bf16", s, 1);
        if (dtype == 0) hipLaunchKernelGGL(gp_mb_mfma_kernel<0>, dim3(grid), dim3(256), 0, s, iters, sink);
        else if (dtype == 1) hipLaunchKernelGGL(gp_mb_mfma_kernel<1>, dim3(grid), dim3(256), 0, s, iters, sink);
        else hipLaunchKernelGGL(gp_mb_mfma_kernel<2>, dim3(grid), dim3(256), 0, s, iters, sink);
    }
    GP_LAUNCH_CHECK();
    return 0;
}

// ---- instruction-rate microbenchmarks ---------------------------------------------------------------------------
// The composite kernels are bound by the vector ALU, not by HBM; what a wave64 VALU instruction costs on this part
// (plain vs packed fp32, transcendental, DPP) decides every design choice there, so it is measured, not assumed.
// Each kind runs 8 independent dependency chains per lane, `iters` x 8 instructions of the kind under test per chain
// group, 4 waves per workgroup, 8 workgroups per CU (the same occupancy as the composite kernels).
typedef float mb_f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void gp_mb_valu_kernel(int iters, float* __restrict__ sink) {
    const float seed = 1.f + (float)(threadIdx.x & 15) * 1e-3f;
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = seed + (float)k * 1e-4f;
    mb_f2 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = (mb_f2){r[k], r[k] * 0.5f};
    const float m = 0.999f, a = 1e-6f;
    __shared__ mb_f4 s_lds[64];
    if (KIND == 6) { if (threadIdx.x < 64) s_lds[threadIdx.x] = (mb_f4){seed, seed, seed, seed}; __syncthreads(); }
    const uint32_t lp = (uint32_t)(uintptr_t)&s_lds[(blockIdx.x + iters) & 63];    // uniform LDS address: a broadcast read
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(m), "v"(a));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"((mb_f2){m, m}), "v"((mb_f2){a, a}));
            if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(r[k]));
            if (KIND == 3) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[k]) : "v"(m), "v"(a) : "vcc");
            if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k]));
            if (KIND == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[k]));
            if (KIND == 6) { mb_f4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lp)); asm volatile("s_waitcnt lgkmcnt(8)"); r[k] += 0.f * t.x; }
            if (KIND == 7) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[k]) : "v"(m));
            if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"((mb_f2){m, m}));
            if (KIND == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[k]));
        }
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += r[k] + p[k].x + p[k].y;
    if (t == 123456.789f) sink[0] = t;
}

extern "C" int gp_microbench_valu(int kind, int iters, float* sink, double* instr_out, void* stream) {
    if (!sink || iters <= 0 || kind < 0 || kind > 9) GP_FAIL("gp_microbench_valu: kind 0..9, iters > 0, sink required");
    hipStream_t s = (hipStream_t)stream;
    const int grid = mb_grid();
    // wave-instructions of the kind under test (kind 3 issues two VALU instructions per slot)
    if (instr_out) *instr_out = 8.0 * (double)iters * 4.0 /*waves*/ * (double)grid * (kind == 3 ? 2.0 : 1.0);
    static const char* names[10] = {"mb_valu_fma", "mb_valu_pk_fma", "mb_valu_exp", "mb_valu_cmp_cndmask", "mb_valu_rcp",
                                    "mb_valu_dpp_add", "mb_lds_read_b128", "mb_valu_min", "mb_valu_pk_mul", "mb_valu_sqrt"};
    {
        GpProfScope _p(names[kind], s, 1);
        switch (kind) {
#define MBV(K) case K: hipLaunchKernelGGL(gp_mb_valu_kernel<K>, dim3(grid), dim3(256), 0, s, iters, sink); break;
            MBV(0) MBV(1) MBV(2) MBV(3) MBV(4) MBV(5) MBV(6) MBV(7) MBV(8) MBV(9)
#undef MBV
        }
    }
    GP_LAUNCH_CHECK();
    return 0;
}

// Random gather: every thread reads rec_bytes (16 / 48 / 64) at record idx[i] of `src` -- the access shape of the
// composite kernels' record fetch.  Run under `rocprofv3 --pmc` it calibrates the FETCH_SIZE correction for gathers
// (the x2 figure was calibrated on a streaming kernel); bracketed as "mb_gather".
__global__ __launch_bounds__(256) void gp_mb_gather_kernel(const mb_f4* __restrict__ src, const uint32_t* __restrict__ idx, size_t n_idx,
                                                           int vecs, int stride_vecs, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    mb_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_idx; i += stride) {
        const mb_f4* p = src + (size_t)idx[i] * stride_vecs;
        for (int v = 0; v < vecs; ++v) acc += p[v];
    }
    const float t = acc.x + acc.y + acc.z + acc.w;
    if (t == 123456.789f) sink[0] = t;
}
extern "C" int gp_microbench_gather(const void* src, int rec_bytes, int stride_bytes, const uint32_t* idx, size_t n_idx, float* sink,
                                    void* stream) {
    if (!src || !idx || !sink || (rec_bytes & 15) || rec_bytes <= 0 || (stride_bytes & 15) || stride_bytes < rec_bytes || ((uintptr_t)src & 15))
        GP_FAIL("gp_microbench_gather: 16-byte multiples and aligned source required");
    hipStream_t s = (hipStream_t)stream;
    {
        GpProfScope _p("mb_gather", s, 1);
        hipLaunchKernelGGL(gp_mb_gather_kernel, dim3(mb_grid()), dim3(256), 0, s, (const mb_f4*)src, idx, n_idx, rec_bytes / 16,
                           stride_bytes / 16, sink);
    }
    GP_LAUNCH_CHECK();
    return 0;
}

// ---- diagnostics knobs (A/B selection of kernel variants while profiling; 0 = the shipped default everywhere) -----
static int g_debug_opt[16] = {0};
int gp_debug_get(int key) { return (key >= 0 && key < 16) ? g_debug_opt[key] : 0; }
extern "C" int gp_debug_option(int key, int value) {
    if (key < 0 || key >= 16) GP_FAIL("gp_debug_option: key 0..15");
    g_debug_opt[key] = value;
    return 0;
}

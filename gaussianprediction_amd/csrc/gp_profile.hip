// gp_profile.hip -- optional per-kernel timing with hipEvent pairs recorded on the launch stream
// (torch.cuda.Event only sees torch's current stream; these see exactly the kernels they bracket).
#include <mutex>
#include <string>
#include <vector>

#include "gp_common.h"

struct ProfRec { const char* name; hipEvent_t a, b; };
static std::mutex g_mu;
static int g_level = 0;   // 0 off, 1 = only the kernels tagged level 1 (roofline kernel), 2 = every kernel
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
        hipLaunchKernelGGL(gp_mb_copy_kernel, dim3(mb_grid()), dim3(MB_THREADS), 0, s, (mb_f4*)dst, (const mb_f4*)src, bytes / 16);
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

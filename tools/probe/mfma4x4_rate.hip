// Probe: issue cost of v_mfma_f32_4x4x1_16b_f32 on gfx950, dependent chain vs independent accumulators, with and without
// VALU work interleaved (hipcc --offload-arch=gfx950 -O3 -o mfma4x4_rate ...).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float* sink) {
    const float a = 1.f + threadIdx.x * 1e-3f, b = 0.5f;
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // dependent chain of 6
            for (int q = 0; q < 6; ++q) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {   // 4 independent accumulators, 6 each (24 MFMA)
            for (int q = 0; q < 6; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
            }
        } else if (MODE == 2) {   // 72 VALU fma only
            for (int q = 0; q < 9; ++q)
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
        } else {                  // 6 dependent MFMA + 72 VALU
            for (int q = 0; q < 6; ++q) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            for (int q = 0; q < 9; ++q)
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
        }
    }
    float t = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 8; ++i) t += v[i];
    if (t == 123456.789f) sink[0] = t;
}
template <int MODE> double run(int iters, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, iters, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, iters, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 64);
    const int iters = 2048;
    const double waves_per_simd = 2048.0 * 4 / 1024;    // 8 waves per SIMD in total, 2 rounds of 4? (256 CUs x 8 blocks resident)
    double m0 = run<0>(iters, d), m1 = run<1>(iters, d), m2 = run<2>(iters, d), m3 = run<3>(iters, d);
    auto cyc = [&](double ms, double n_per_iter) { return ms * 1e-3 * 2.4e9 / (iters * n_per_iter * waves_per_simd); };
    printf("dependent chain:   %.3f ms  -> %.1f cycles per MFMA per SIMD\n", m0, cyc(m0, 6));
    printf("4 independent:     %.3f ms  -> %.1f cycles per MFMA per SIMD\n", m1, cyc(m1, 24));
    printf("72 v_fma only:     %.3f ms  -> %.1f cycles per VALU per SIMD\n", m2, cyc(m2, 72));
    printf("6 MFMA + 72 v_fma: %.3f ms  (sum of the two alone %.3f ms)\n", m3, m0 + m2);
    return 0;
}

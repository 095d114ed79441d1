// Probe: lane / register layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (hipcc --offload-arch=gfx950 -o mfma4x4_layout ...).
// A = 100 * lane, B = lane: D[r] of lane l shows which A-lane and B-lane meet there.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(1000.f * (float)(lane + 1), (float)(lane + 1), c, 0, 0, 0);
    // second accumulate with another pattern (checks chaining): += 1 * 0.001*(lane+1)
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, 0.001f * (float)(lane + 1), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            // D = 1000*(la+1)*(lb+1) + 0.001*(lb2+1)
            double v = h[l * 4 + r];
            printf("  r%d=%.3f", r, v);
        }
        printf("\n");
    }
    return 0;
}

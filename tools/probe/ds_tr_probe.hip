// What ds_read_b64_tr_b16 returns (gfx950): LDS holds lds[i] = i (16-bit); every lane supplies a byte address; print lane -> 4 elements.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/ds_tr_probe.hip -o build/ds_tr_probe && build/ds_tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* addr, uint32_t* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = addr[threadIdx.x] + (uint32_t)(uintptr_t)lds;
    u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[2 * threadIdx.x] = r[0]; out[2 * threadIdx.x + 1] = r[1];
}
int main() {
    uint32_t h_addr[64], h_out[128], *d_addr, *d_out;
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l)
            h_addr[l] = pat == 0 ? l * 8 : pat == 1 ? (63 - l) * 8 : ((l * 37) % 64) * 64 + (l & 3) * 8;   // linear, reversed, scattered
        hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (lane: byte address -> elements as 16-bit indices; source lane of element = which lane's address range holds it)\n", pat);
        for (int l = 0; l < 64; ++l) {
            int e[4] = {(int)(h_out[2 * l] & 0xffff), (int)(h_out[2 * l] >> 16), (int)(h_out[2 * l + 1] & 0xffff), (int)(h_out[2 * l + 1] >> 16)};
            printf("  lane %2d addr %5u ->", l, h_addr[l]);
            for (int j = 0; j < 4; ++j) {
                int src = -1, pos = -1;
                for (int s = 0; s < 64; ++s)
                    if (e[j] * 2 >= (int)h_addr[s] && e[j] * 2 < (int)h_addr[s] + 8) { src = s; pos = (e[j] * 2 - h_addr[s]) / 2; }
                printf(" %4d(lane %2d el %d)", e[j], src, pos);
            }
            printf("\n");
        }
    }
    return 0;
}

// Microbenchmark (GPU box): how many cycles does one wave64 integer VALU instruction hold a SIMD for, at 1..8 waves/SIMD?
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a0 = a0 * 5 + a1; a1 = a1 ^ (a2 >> 3); a2 = a2 + a3; a3 = a3 ^ a4; a4 = a4 + a5; a5 = a5 ^ (a6 << 1); a6 = a6 + a7; a7 = a7 ^ a0;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 2048 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD: blocks of 256 threads = 1 wave per SIMD per block
        const int blocks = 256 * wps, iters = 20000;
        k<<<blocks, 256>>>(d, 10);
        hipEventRecord(e0); k<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per loop body: 16 x (mul-add(2?) ...) count ~ 16*9 = 144 VALU-ish; report cycles per body per wave at 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("waves/SIMD %d: %.3f ms, %.1f cycles per 16x8-op body per SIMD-wave-slot, i.e. %.2f cycles per body per wave\n", wps, ms, cyc / iters, cyc / iters / wps);
    }
    return 0;
}

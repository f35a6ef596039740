// Microbenchmark (GPU box): issue cost of v_mul_lo_u32 vs v_mul_u32_u24 / v_mad_u32_u24 / v_add / v_xor on a wave64.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mul_rate.hip -o /tmp/mul_rate && /tmp/mul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned* out, int iters, unsigned m) {
    unsigned a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == 0) a[j] = (a[j] * m) ^ a[(j + 1) & 7];                                           // v_mul_lo_u32
                if (OP == 1) a[j] = __umul24(a[j], m) ^ a[(j + 1) & 7];                // v_mul_u32_u24
                if (OP == 2) a[j] = (a[j] + m) ^ a[(j + 1) & 7];                                          // v_add_u32
                if (OP == 3) a[j] = (a[j] ^ m) + (a[j] >> 7);                            // xor + shift-add (2-3 ops)
                if (OP == 4) a[j] = __umul24(a[j], m) + a[(j + 1) & 7];  // v_mad_u32_u24
            }
    }
    unsigned x = 0;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int OP> void run(const char* name, unsigned* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4, iters = 20000;       // 4 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 10, 0x9E3779B1u);
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, 0x9E3779B1u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %.3f ms: %.2f clocks (2.4 GHz) per statement per wave on a SIMD shared by 4 waves\n", name, ms, ms * 1e-3 * 2.4e9 / iters / 64 / 4);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 2048 * 8 * 4);
    run<0>("mul_lo_u32 + xor", d); run<1>("mul_u32_u24 + xor", d); run<2>("add + xor", d); run<3>("(a^m)+(a>>7)", d); run<4>("mad_u32_u24", d);
    return 0;
}

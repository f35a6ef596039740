// Microbenchmark (GPU box, under rocprofv3 --pmc FETCH_SIZE): what FETCH_SIZE reports for the two access shapes of k_stream --
// (A) a streaming read, 16 bytes per lane, non-temporal; (B) one random 64-byte line per lane read as 4 x 16 bytes (an EC-table
// lookup).  Known byte counts: (A) n * 16 B per pass, (B) n * 64 B per pass.  MI355X_MICROARCH.md says (A) reports half.
// hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_calib_stream(const u32x4* p, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_calib_lines(const uint4* table, size_t lines, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        const uint4* q = table + (z % lines) * 4;
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = 8ull << 30, n16 = bytes / 16, lines = (1ull << 30) / 64, nacc = 100ull << 20;
    void *a, *t; unsigned* o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&t, lines * 64) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    if (hipMemset(a, 1, bytes) != hipSuccess || hipMemset(t, 1, lines * 64) != hipSuccess) return 1;
    k_calib_stream<<<4096, 256>>>((const u32x4*)a, n16, o);
    k_calib_lines<<<4096, 256>>>((const uint4*)t, lines, nacc, o);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("k_calib_stream: %zu bytes read; k_calib_lines: %zu accesses of one 64-byte line = %zu bytes (1 GiB table)\n", bytes, nacc, nacc * 64);
    return 0;
}

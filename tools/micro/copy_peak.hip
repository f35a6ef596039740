// Measurement helper (not the product): what a plain streaming copy gets out of this card's HBM, with the access shape k_stream
// uses -- 16 bytes per lane, non-temporal loads -- so that bench.py can quote the stream kernel against a measured peak of
// the same kind (MI355X_MICROARCH.md: ~6.3 TB/s for a float4 copy; torch.Tensor.copy_ reaches ~4.9).
// Built by alntools_amd/build.py into tools/micro/bin/libcopy_peak.so; loaded with ctypes by bench.py only.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every workgroup walks its own contiguous stretch, 4 loads in flight per lane
__global__ __launch_bounds__(256) void k_copy16(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16, uint64_t per_block) {
    const uint64_t b0 = (uint64_t)blockIdx.x * per_block, b1 = b0 + per_block < n16 ? b0 + per_block : n16;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += 4 * 256) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < b1) v[k] = __builtin_nontemporal_load(src + i + k * 256);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < b1) __builtin_nontemporal_store(v[k], dst + i + k * 256);
    }
}
// read-only variant: the sum keeps the loads alive; one store per workgroup
__global__ __launch_bounds__(256) void k_read16(const u32x4* __restrict__ src, uint32_t* __restrict__ out, uint64_t n16, uint64_t per_block) {
    const uint64_t b0 = (uint64_t)blockIdx.x * per_block, b1 = b0 + per_block < n16 ? b0 + per_block : n16;
    uint32_t acc = 0;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += 4 * 256) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i + k * 256 < b1 ? __builtin_nontemporal_load(src + i + k * 256) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;      // (practically never: the loads must not be optimised away)
}

extern "C" {
// copies `bytes` (a multiple of 16) from d_src to d_dst `reps` times on the null stream; *gbps = (read + written bytes) per second / 1e9.
int copy_peak_run(const void* d_src, void* d_dst, uint64_t bytes, int reps, double* gbps) {
    if (!d_src || !d_dst || !gbps || bytes < 16 || reps < 1) return -1;
    const uint64_t n16 = bytes / 16;
    const unsigned blocks = 256 * 16;                   // 16 workgroups per CU's worth of stretches
    const uint64_t per_block = (n16 + blocks - 1) / blocks;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -2;
    k_copy16<<<blocks, 256>>>((const u32x4*)d_src, (u32x4*)d_dst, n16, per_block);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k_copy16<<<blocks, 256>>>((const u32x4*)d_src, (u32x4*)d_dst, n16, per_block);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) return -3;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *gbps = 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9;
    return 0;
}
// reads `bytes` `reps` times; *gbps = bytes read per second / 1e9 (d_out: at least 4096 uint32)
int read_peak_run(const void* d_src, void* d_out, uint64_t bytes, int reps, double* gbps) {
    if (!d_src || !d_out || !gbps || bytes < 16 || reps < 1) return -1;
    const uint64_t n16 = bytes / 16;
    const unsigned blocks = 256 * 16;
    const uint64_t per_block = (n16 + blocks - 1) / blocks;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -2;
    k_read16<<<blocks, 256>>>((const u32x4*)d_src, (uint32_t*)d_out, n16, per_block);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k_read16<<<blocks, 256>>>((const u32x4*)d_src, (uint32_t*)d_out, n16, per_block);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) return -3;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *gbps = (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9;
    return 0;
}
}

// Measurement helper (not the product): the READ side of k_stream alone, with its shape as parameters -- persistent waves that claim slices
// of the streams from a counter and walk them in steps of `unit` bytes per stream (k_stream: 2 KB = a tile's 512 records x 4 B), `nstreams`
// streams read at the same offsets (k_stream: 3), 16 bytes per lane per load, non-temporal.  What does the rate depend on: the unit, the number
// of streams, the slice length -- and where the streams sit in HBM?  (k_stream moves by up to 14 % with the placement of
// its tuples, a plain sweep by 1 %.)  Built by hand: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/bin/libstream3.so tools/micro/stream3.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNITS>   // loads of 1 KB (64 lanes x 16 B) per stream and step
__global__ __launch_bounds__(256) void k_stream3(const char* s0, const char* s1, const char* s2, int nstreams, uint64_t bytes, uint64_t slice,
                                                 unsigned long long* next, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    for (;;) {
        unsigned long long w = 0;
        if (lane == 0) w = atomicAdd(next, 1ull);
        w = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
        const uint64_t b0 = w * slice;
        if (b0 >= bytes) break;
        const uint64_t b1 = b0 + slice < bytes ? b0 + slice : bytes;
        for (uint64_t at = b0; at + (uint64_t)UNITS * 1024 <= b1; at += (uint64_t)UNITS * 1024) {
            u32x4 v[3][UNITS];
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                v[0][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(s0 + at + u * 1024 + lane * 16));
                if (nstreams > 1) v[1][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(s1 + at + u * 1024 + lane * 16));
                if (nstreams > 2) v[2][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(s2 + at + u * 1024 + lane * 16));
            }
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                acc += v[0][u].x ^ v[0][u].w;
                if (nstreams > 1) acc += v[1][u].y;
                if (nstreams > 2) acc += v[2][u].z;
            }
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

extern "C" int stream3_run(const void* s0, const void* s1, const void* s2, int nstreams, uint64_t bytes, int units, uint64_t slice, int waves_per_simd,
                           void* d_counter, void* d_out, int reps, double* gbps) {
    if (!s0 || !gbps || units < 1 || reps < 1) return -1;
    const unsigned blocks = 256u * (unsigned)waves_per_simd;      // 4 waves per workgroup: waves_per_simd workgroups per CU
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -2;
    auto launch = [&]() {
        hipMemsetAsync(d_counter, 0, 8, 0);
        const char *a = (const char*)s0, *b = (const char*)s1, *c = (const char*)s2;
        switch (units) {
            case 1: k_stream3<1><<<blocks, 256>>>(a, b, c, nstreams, bytes, slice, (unsigned long long*)d_counter, (uint32_t*)d_out); break;
            case 2: k_stream3<2><<<blocks, 256>>>(a, b, c, nstreams, bytes, slice, (unsigned long long*)d_counter, (uint32_t*)d_out); break;
            case 3: k_stream3<3><<<blocks, 256>>>(a, b, c, nstreams, bytes, slice, (unsigned long long*)d_counter, (uint32_t*)d_out); break;
            case 4: k_stream3<4><<<blocks, 256>>>(a, b, c, nstreams, bytes, slice, (unsigned long long*)d_counter, (uint32_t*)d_out); break;
            case 6: k_stream3<6><<<blocks, 256>>>(a, b, c, nstreams, bytes, slice, (unsigned long long*)d_counter, (uint32_t*)d_out); break;
            default: k_stream3<8><<<blocks, 256>>>(a, b, c, nstreams, bytes, slice, (unsigned long long*)d_counter, (uint32_t*)d_out); break;
        }
    };
    launch();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) return -3;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *gbps = (double)bytes * nstreams * reps / (ms * 1e-3) / 1e9;
    return 0;
}

// Microbenchmark (GPU box): what one wave64 instruction of each kind costs a SIMD / a CU on gfx950, at 1..8 waves per SIMD.
// The instructions are written as inline asm so that the compiler neither folds nor reorders them; eight independent
// registers per kind, 64 instructions per loop trip.  Cycles are s_memtime ticks of the waves themselves (the shader clock),
// so DVFS does not enter: reported = wave cycles / (instructions x waves per SIMD) = cycles of SIMD time per instruction.
// hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rate.hip -o tools/micro/bin/issue_rate && tools/micro/bin/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { OP_ADD, OP_AND, OP_LSHL, OP_CNDMASK, OP_CMP, OP_MULLO, OP_FMA, OP_DPP, OP_READLANE, OP_WRITELANE, OP_LSHLOR, OP_ALIGNBIT, OP_BFE,
       OP_SALU, OP_MIX_VS, OP_DS_CAS, OP_DS_OR, OP_DS_READ, OP_BALLOT_MBCNT, OP_N };
static const char* NAMES[OP_N] = {"v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_cndmask_b32", "v_cmp_lt_u32", "v_mul_lo_u32", "v_fma_f32",
                                  "v_mov_b32 dpp row_shr:1", "v_readlane_b32", "v_writelane_b32", "v_lshl_or_b32", "v_alignbit_b32", "v_bfe_u32",
                                  "s_add_u32 (SALU)", "1 v_add + 1 s_add (pairs)", "ds_cmpst_rtn_b32 (own slot)", "ds_or_b32 (own slot)", "ds_read_b32",
                                  "v_cmp+mbcnt_lo+mbcnt_hi (3 ops)"};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* cyc, int iters) {
    __shared__ unsigned lds[4096];
    unsigned a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 7u + j;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    for (int j = threadIdx.x; j < 4096; j += 256) lds[j] = 0;
    __syncthreads();
    const unsigned addr = (threadIdx.x * 4u) + ((threadIdx.x >> 6) * 1024u * 4u) * 0u;   // own dword, conflict-free
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == OP_ADD) {
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if (OP == OP_AND) {
#define X(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if (OP == OP_LSHL) {
#define X(j) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[j]));
            REP64(X)
#undef X
        } else if (OP == OP_CNDMASK) {
#define X(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(a[(j + 1) & 7]) : "vcc");
            REP64(X)
#undef X
        } else if (OP == OP_CMP) {
#define X(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[j]), "v"(a[(j + 1) & 7]) : "vcc");
            REP64(X)
#undef X
        } else if (OP == OP_MULLO) {
#define X(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if (OP == OP_FMA) {
#define X(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if (OP == OP_DPP) {
#define X(j) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if (OP == OP_READLANE) {
#define X(j) asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s##j) : "v"(a[j]));
            REP64(X)
#undef X
        } else if (OP == OP_WRITELANE) {
#define X(j) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[j]) : "s"(s##j));
            REP64(X)
#undef X
        } else if (OP == OP_LSHLOR) {
#define X(j) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if (OP == OP_ALIGNBIT) {
#define X(j) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if (OP == OP_BFE) {
#define X(j) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[j]));
            REP64(X)
#undef X
        } else if (OP == OP_SALU) {
#define X(j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s##j) : "s"(s0) : "scc");
            REP64(X)
#undef X
        } else if (OP == OP_MIX_VS) {
#define X(j) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(a[j]), "+s"(s##j) : "v"(a[(j + 1) & 7]) : "scc");
            REP64(X)
#undef X
        } else if (OP == OP_DS_CAS) {
#define X(j) asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(addr), "v"(a[(j + 1) & 7]) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == OP_DS_OR) {
#define X(j) asm volatile("ds_or_b32 %0, %1" : : "v"(addr), "v"(a[j]) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == OP_DS_READ) {
#define X(j) asm volatile("ds_read_b32 %0, %1" : "=v"(a[j]) : "v"(addr) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == OP_BALLOT_MBCNT) {
#define X(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_mbcnt_lo_u32_b32 %0, vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %0, vcc_hi, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]) : "vcc");
            REP64(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned x = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(unsigned* d, unsigned long long* dc) {
    const int iters = 2000;
    const int per_trip = (OP == OP_MIX_VS) ? 128 : (OP == OP_BALLOT_MBCNT ? 192 : 64);
    printf("%-32s", NAMES[OP]);
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * wps;                 // 256-thread blocks: one wave per SIMD each
        k<OP><<<blocks, 256>>>(d, dc, 10);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(d, dc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 4);
        hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : c) sum += (double)v;
        const double wave_cyc = sum / c.size();
        // s_memtime ticks at 100 MHz on this part?  report both: ticks per instruction-slot, and wall-derived cycles at 2.4 GHz
        printf("  w%d: %6.2f tick %6.2f cyc@2.4", wps, wave_cyc / ((double)iters * per_trip * wps), ms * 1e-3 * 2.4e9 / ((double)iters * per_trip * wps));
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
}

int main() {
    unsigned* d; unsigned long long* dc;
    hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dc, 256 * 8 * 4 * 8);
    printf("per instruction, SIMD time (wave time / waves per SIMD); 'tick' = readcyclecounter units, 'cyc@2.4' = wall time x 2.4 GHz\n");
    run<OP_ADD>(d, dc); run<OP_AND>(d, dc); run<OP_LSHL>(d, dc); run<OP_CNDMASK>(d, dc); run<OP_CMP>(d, dc); run<OP_MULLO>(d, dc); run<OP_FMA>(d, dc);
    run<OP_DPP>(d, dc); run<OP_READLANE>(d, dc); run<OP_WRITELANE>(d, dc); run<OP_LSHLOR>(d, dc); run<OP_ALIGNBIT>(d, dc); run<OP_BFE>(d, dc);
    run<OP_SALU>(d, dc); run<OP_MIX_VS>(d, dc); run<OP_DS_CAS>(d, dc); run<OP_DS_OR>(d, dc); run<OP_DS_READ>(d, dc); run<OP_BALLOT_MBCNT>(d, dc);
    return 0;
}

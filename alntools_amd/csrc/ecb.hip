// libecb -- equivalence-class builder for alntools' bam2ec / bam2emase hot path on MI355X (gfx950).
//
// What the reference does per alignment in Python (alntools/bam_utils.py:258-344), per merge (:680-724) and per EC
// (:788-847) is done here by these groups of kernels (DESIGN.md section 4 has the measurements behind the shapes):
//
//   k_stream   one pass over the record tuples (12 B/record, 16-byte loads).  Wave-autonomous: each wave owns a
//              contiguous slice of the stream and walks it in tiles of 512 records with wave-private LDS and no
//              workgroup barriers.  Per tile: (a) record filter (bam_utils.py:264-270) and read heads from the host's run
//              counter (:289-320); (b) one {locus -> haplotype mask} open-addressing table per read in LDS -- ds_cmpst on the
//              locus, ds_or of the haplotype bit: the duplicate collapse of :322-325 and the per-(EC, target, haplotype) bit
//              test of :800-819 in two LDS atomics; each table entry is hashed once and summed per read (a commutative set
//              hash, the stand-in for the sorted string key of :307); (c) one lane per read looks the 126-bit key up in the
//              global EC table or inserts it, and records the slot of the read.  The lane that creates an EC copies its
//              (locus, mask) pairs from LDS to the key arena.
//   k_slow     the same for single reads that do not fit a tile or hit a full table (one workgroup per read).
//   k_count    reads per EC and first read per EC (:309-312, 688-698) from the per-read slots, without global atomics:
//              partition by slot range, count in LDS.
//   finalize   rank ECs by first appearance (bitmap + scan: :682-698), scan of row lengths, sort each row by locus and
//              emit CSR A / N (:835-847, bin_utils.py:208-211).
//   k_merge    multi-GPU: re-insert another rank's serialised table (the ordered merge of :680-724).
//   k_ms_*, k_cv_*   multisample (EC, cell, file) triples; CSR(bitmask) <-> per-haplotype CSC (bin_utils.py:979-1028).
//
// Integer / indexing work only: no MFMA.  The bound is HBM bandwidth; today the stream kernel is VALU-issue bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>   // radix sort of the (EC, cell, file) keys of the multisample path only

#include "../../include/ecb.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int TPB = 256;             // threads per workgroup (4 waves of 64)
constexpr u32 MAX_PROBE = 256;       // EC-table probes before a read is deferred to k_slow
constexpr u32 PENDING = 0xFFFFFFFFu;
constexpr u32 ARENA_CHUNK = 512;     // pairs a wave reserves from the key arena per global atomic
constexpr u32 ARENA_REGIONS = 64;    // the key arena has this many allocation cursors (see arena_alloc)
constexpr u32 MAX_LOCI = 1u << 27;   // (locus << 5 | hap) + 1 must fit 32 bits

constexpr u32 ERR_CONTRACT = 1u;     // device error bits (Counters::err)
constexpr u32 ERR_RANGE = 2u;
constexpr u32 ERR_ARENA = 4u;
constexpr u32 ERR_QUEUE = 8u;

struct Slot {                        // 32 bytes, one EC
    u64 lo, hi;                      // 126-bit set hash, both non-zero once claimed
    u32 count;                       // reads in this EC
    u32 first_inv;                   // ~(smallest read index)  (atomicMax on zero-initialised memory)
    u32 off, n;                      // key = arena[off .. off+n): (locus, haplotype mask) pairs
};

struct Counters {
    u64 all, valid;                  // records offered / passing the filter
    u64 arena_top;                   // pairs placed by ecb_table_adopt_device (dense from 0; an adopting handle allocates nothing else)
    u64 n_queue;                     // reads deferred to k_slow
    u64 n_ecs;                       // ECs created by k_slow / k_merge (k_stream's are counted by k_collect_new)
    u32 err;
    u32 full;                        // set when a read found no EC-table slot: workgroups park, host grows the table
    u64 arena_reg[ARENA_REGIONS];    // next free pair of every arena region
    u64 next_slice;                  // k_stream: next unclaimed slice of the batch (zeroed per launch)
};

// The key arena is cut into equal regions with a cursor each; a wave allocates from the region its index picks and moves
// on when that one is full.  One cursor for the whole arena is one address for every reservation of every wave:
// same-address atomics complete ~18 ns apart, and the 66 k chunk reservations of a C3 pass added 1.4 ms to k_stream.
// (Regions of at least 64 k pairs; small arenas have fewer.)  Keys are found through Slot::off, nothing needs the arena dense.
__host__ __device__ __forceinline__ u32 arena_regions(u64 arena_cap) { return (u32)(arena_cap >> 16 >= ARENA_REGIONS ? ARENA_REGIONS : (arena_cap >> 16 ? arena_cap >> 16 : 1)); }
__device__ __forceinline__ u64 arena_alloc(Counters* ctr, u64 arena_cap, u32 n, u32 hint) {
    const u32 R = arena_regions(arena_cap);
    const u64 per = arena_cap / R;
    for (u32 t = 0; t < R; ++t) {
        const u32 r = (hint + t) % R;
        const u64 at = atomicAdd(&ctr->arena_reg[r], (u64)n);     // (a failed try leaves the cursor past the end: that region is full)
        if (at + n <= (u64)(r + 1) * per) return at;
    }
    return ~0ull;
}

// ---------------------------------------------------------------------------------------------
// hashing: EC identity = the SET of (locus, haplotype) targets of a read (bam_utils.py:307 builds a
// sorted string for the same purpose).  The set is held as {locus -> haplotype mask}; its hash is the sum
// over loci of four 32-bit mixes of (locus, mask), finalised to 2 x 63 bits.  The sum commutes, so records
// need no sorting, and OR-ing haplotype bits makes duplicate (read, target) records vanish by itself.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ u32 fmix32(u32 h) {           // murmur3 finaliser: a bijection with full avalanche
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
// four 32-bit hashes of one (locus, haplotype mask) pair of a read's target set
__device__ __forceinline__ void pair_hash(u32 locus, u32 mask, u32* h) {
    const u32 x = fmix32(locus ^ 0x9E3779B9u), y = fmix32(mask * 0x9E3779B1u + 0x7F4A7C15u);
    h[0] = fmix32(x + y);
    h[1] = fmix32((x ^ 0x85EBCA77u) - (y << 7 | y >> 25));
    h[2] = fmix32((x << 13 | x >> 19) ^ (y + 0xC2B2AE3Du));
    h[3] = fmix32(~x + (y << 19 | y >> 13) * 0x27D4EB2Fu);
}
// the sums of fmix32 outputs are already uniformly spread: the table key is the pair of sums itself
__device__ __forceinline__ void finish_hash(u64 s0, u64 s1, u32 n, u64& lo, u64& hi) {
    lo = (s0 + ((u64)n << 32) + n) | 1ull;
    hi = s1 | 1ull;
}
__device__ __forceinline__ void pair_hash64(u32 locus, u32 mask, u64& a, u64& b) {
    u32 h[4];
    pair_hash(locus, mask, h);
    a = ((u64)h[1] << 32) | h[0]; b = ((u64)h[3] << 32) | h[2];
}

// record filter, bam_utils.py:264-270 (host bits 12/13 carry the two non-flag terms)
__device__ __forceinline__ bool rec_valid(u32 hf) {
    if (hf & 0x4u) return false;
    if (hf & 0x1u) {
        if ((hf & 0x80u) || !(hf & 0x2u) || (hf & (ECB_FLAG_MATE_OTHER_REF | ECB_FLAG_NEXT_POS_NEG))) return false;
    }
    return true;
}

// EC-table lookup / insert.  Returns the slot index, or ~0 if no slot within MAX_PROBE (table too full).
// *created is set for the one caller that claimed the slot.  Keys only ever go 0 -> value, so a plain
// load that shows another key (or ours) can be trusted; a plain load that shows "empty" is re-checked by
// the CAS.  Read counts and first appearances are NOT maintained here: per-read atomics on a skewed EC
// distribution run at ~5 G/s chip-wide (measured), so k_stream only records the slot of every read and
// k_count reduces them afterwards without global atomics.
__device__ __forceinline__ u64 table_find_or_insert(Slot* table, u64 cap_mask, u64 lo, u64 hi, bool* created) {
    u64 j = lo & cap_mask;
    *created = false;
    for (u32 probe = 0; probe < MAX_PROBE; ++probe, j = (j + 1) & cap_mask) {
        Slot* s = table + j;
        u64 clo = s->lo, chi = s->hi;
        if (clo == lo && chi == hi) return j;
        if (clo != 0ull && (clo != lo || (chi != 0ull && chi != hi))) continue;
        clo = atomicCAS(&s->lo, 0ull, lo);
        if (clo != 0ull && clo != lo) continue;
        chi = atomicCAS(&s->hi, 0ull, hi);
        if (chi == 0ull) *created = true;
        else if (chi != hi) continue;
        return j;
    }
    return ~0ull;
}

// ---------------------------------------------------------------------------------------------
// k_stream -- wave-autonomous: every wave owns a contiguous slice of the record stream and walks it in
// tiles of WT records with wave-private LDS and no workgroup barriers, so the 16 waves of a CU overlap
// each other's HBM and EC-table latency.  Per tile and lane: 8 records (2 x 16-byte loads per stream).
// ---------------------------------------------------------------------------------------------
#ifndef ECB_RPL
#define ECB_RPL 8
#endif
constexpr int RPL = ECB_RPL;         // records per lane per tile: groups of 4 consecutive records (one 16-byte load per stream)
static_assert(RPL == 8, "only 8 records per lane is validated (16 was measured: 233 VGPRs, 2 waves/SIMD, 11-25 % slower; it also needs a 32-bit ent[])");
constexpr int NG = RPL / 4;          // groups; group g of lane l holds records 256 g + 4 l .. + 3 of the tile
constexpr int WT = 64 * RPL;         // records per wave tile
constexpr int WMAXR = 64;            // reads finished per wave tile (one lane each in phase (c))
constexpr int NWAVE = TPB / 64;
#ifndef ECB_SLOT_SHIFT
#define ECB_SLOT_SHIFT 1
#endif
#ifndef ECB_WAVES_PER_SIMD
#define ECB_WAVES_PER_SIMD 4
#endif
constexpr int SLOT_SHIFT = ECB_SLOT_SHIFT;           // table slots per record = 1 + 2^-SLOT_SHIFT (1: 1.5, 2: 1.25)
constexpr int TSLOTS = WT + (WT >> SLOT_SHIFT);      // LDS table slots per wave tile
constexpr u32 SBITS = (TSLOTS <= 1024) ? 10 : 11;   // bits of a table-slot index
constexpr u32 SMASK = (1u << SBITS) - 1u;
struct alignas(16) WaveLds {
    u32 tkey[TSLOTS];                // per-read {locus -> mask} tables, 1.5 slots per record of the read; key = locus + 1
    u32 tmask[TSLOTS];               // (contiguous with tkey: cleared together with 16-byte stores)
    u64 acc[WMAXR][2];               // per read: set-hash sums (2 x 64 bits)
    unsigned short ent[WT];          // table entries created in this tile: slot | read << 10 (SBITS = 10 at 8 records per lane)
    unsigned short npair[WMAXR];     // per read: number of (locus, mask) pairs
    u32 seg[WMAXR + 3];              // seg[rl + 1]: first table slot of read rl | end slot << 16; repacked for finished reads
};
__device__ __forceinline__ u32 tslot(u32 rec) { return rec + (rec >> SLOT_SHIFT); }   // first table slot of a read starting at `rec`
__device__ __forceinline__ u32 unslot(u32 t) {                                          // inverse, t < 1024
    if (SLOT_SHIFT == 1) { const u32 m = (t * 683u) >> 11; return 2u * m + (t - 3u * m); }
    const u32 m = (t * 205u) >> 10; return 4u * m + (t - 5u * m);
}

struct StreamArgs {
    const u32* rid; const u32* loc; const u32* hf;
    u64 n, chunk;
    u32 prev_rid;                    // read_id of the record before this batch (0xFFFFFFFF at stream start)
    u32 n_loci, n_haps;
    Slot* table; u64 cap_mask;
    uint2* arena; u64 arena_cap;
    Counters* ctr;
    u32* read_slot;                  // slot of every read (indexed by read_id)
    u64* queue; u64 queue_cap;       // head record index of deferred reads
    u64* resume;                     // per wave {next record to process, records counted up to}
    u32* wave_counts;                // per wave {records offered, records valid, ECs created}: summed by k_sum_counts
    u64* wave_arena;                 // per wave {start, pairs left} of its arena reservation, kept from launch to launch
    u32 ablate;                      // profiling only (env ECB_ABLATE): 1 = stop after (a), 2 = after (b), 4 = no EC table
    u64* timing;                     // profiling only (-DECB_TIMING): clocks per phase, summed over waves
};

__device__ __forceinline__ void wave_sync() {   // orders this wave's LDS traffic (lanes run in lockstep)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Wave-wide sums and scans on the DPP cross-lane paths of the VALU (row_shr within rows of 16 lanes, then row_bcast:15 / :31
// across rows): six dependent VALU ops instead of six dependent trips through the LDS crossbar (ds_bpermute, which is
// what __shfl* compile to) -- the latter were ~700 clocks of latency per use.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_add(u32 v) { return v + (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);   // row_shr:1,2,4,8
    v = dpp_add<0x142, 0xa>(v);            // row_bcast:15 -> rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);            // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) { return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63); }
__device__ __forceinline__ u32 lane_above(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }   // wave_shr:1 (lane 0 gets 0)
__device__ __forceinline__ int clamp04(int x, int, int) { return min(max(x, 0), 4); }   // v_med3_i32
// bits [lo, hi) of a 4-record group, lo/hi given relative to the group's first record
__device__ __forceinline__ u32 group_mask(int lo, int hi) {
    lo = max(lo, 0); hi = min(hi, 4);
    return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}

struct TileRegs { u32 rr[RPL], ll[RPL], hh[RPL]; };

// -DECB_TIMING: profiling build.  Every wave adds up the shader clocks it spends in each phase of a tile (stalls are
// charged to the phase whose s_waitcnt sits them out); k_stream adds them into StreamArgs::timing[8].
#ifdef ECB_TIMING
#define TICK(i) do { const u64 t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define TICK(i) do { } while (0)
#endif

__device__ __forceinline__ void load_tile(const StreamArgs& A, u64 tb, u64 te, u32 lane, TileRegs& R) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const u64 i0 = tb + (u64)g * 256 + 4u * lane;
        if (i0 + 4 <= te) {
            // (non-temporal: the stream is read once; without the hint it pushes the EC table's lines out of L2 -- measured 3 %)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A.rid + i0));
            R.rr[4 * g] = v.x; R.rr[4 * g + 1] = v.y; R.rr[4 * g + 2] = v.z; R.rr[4 * g + 3] = v.w;
            v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A.loc + i0));
            R.ll[4 * g] = v.x; R.ll[4 * g + 1] = v.y; R.ll[4 * g + 2] = v.z; R.ll[4 * g + 3] = v.w;
            v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A.hf + i0));
            R.hh[4 * g] = v.x; R.hh[4 * g + 1] = v.y; R.hh[4 * g + 2] = v.z; R.hh[4 * g + 3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = i0 + j < te;
                R.rr[4 * g + j] = in ? A.rid[i0 + j] : 0u;
                R.ll[4 * g + j] = in ? A.loc[i0 + j] : 0u;
                R.hh[4 * g + j] = in ? A.hf[i0 + j] : 0x4u;
            }
        }
    }
}

// VERIFY = false: the hot kernel.  VERIFY = true: the exactness pass (same tiling, compares instead of inserting).
template <bool VERIFY>
__global__ __launch_bounds__(TPB, ECB_WAVES_PER_SIMD) void k_stream(StreamArgs A) {
    __shared__ WaveLds wl[NWAVE];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    WaveLds& L = wl[w];
    // Persistent waves: the launch holds only as many waves as are resident at once, and every wave takes the next
    // unclaimed slice of the stream when it has finished one.  (Slices bound to workgroups at launch left a SIMD slot
    // idle until the slowest of a workgroup's four waves was done, and the last round ragged: a fifth of the kernel.)
    const u64 pw = (u64)blockIdx.x * NWAVE + w;   // this wave
    u32 my_all = 0, my_valid = 0, my_new = 0;     // per lane; a wave sees far fewer than 2^32 records
    u32 bad = 0;
    // this wave's current reservation in the key arena: what the last launch left of it is used first (a stream pushed in
    // many small batches would otherwise leave the tail of a 512-pair chunk behind per wave and launch)
    u64 chunk_at = VERIFY ? 0ull : A.wave_arena[2 * pw];
    u32 chunk_left = VERIFY ? 0u : (u32)A.wave_arena[2 * pw + 1];
#ifdef ECB_TIMING
    u64 tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  for (;;) {
    u64 wid = 0;                                  // the slice
    if (lane == 0) wid = atomicAdd(&A.ctr->next_slice, 1ull);
    wid = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(wid >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)wid);
    const u64 c0 = wid * A.chunk;
    if (c0 >= A.n) break;
    const u64 c1 = min(c0 + A.chunk, A.n);
    u64 p = A.resume[2 * wid], counted = A.resume[2 * wid + 1];
    if (p >= c1) continue;                        // finished before a relaunch

    u32 base = (p == 0 ? A.prev_rid : A.rid[p - 1]) + 1u;     // read index of the first head >= p
    TileRegs R;
    load_tile(A, p & ~(u64)3, min((p & ~(u64)3) + (u64)WT, A.n), lane, R);
    u32 parked = __hip_atomic_load(&A.ctr->full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(parked));          // (settled before the loop: otherwise the loop header carries an s_waitcnt vmcnt(0) that every tile pays)

    while (p < c1) {
        if (parked) break;            // the EC table filled up somewhere: the host grows it and relaunches
        p = ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(p >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((u32)p);   // (the builtin returns int: no sign extension)
        base = (u32)__builtin_amdgcn_readfirstlane(base);
        const u64 tb = p & ~(u64)3;
        const u64 te = min(tb + (u64)WT, A.n);
        // tile-relative bounds (all below 2^31)
        const int p_rel = (int)(p - tb), te_rel = (int)(te - tb);
        const int c1_rel = (int)min(c1 - tb, (u64)WT), cnt_lo = (int)(max(counted, tb) - tb), cnt_hi = min(te_rel, c1_rel);
        // clear this wave's per-tile LDS state
        {
            uint4* z = reinterpret_cast<uint4*>(L.tkey);      // tkey and tmask are contiguous
#pragma unroll
            for (int t = 0; t < (2 * TSLOTS) / (4 * 64); ++t) z[t * 64 + lane] = make_uint4(0, 0, 0, 0);
            L.acc[lane][0] = 0; L.acc[lane][1] = 0; L.npair[lane] = 0;
        }
        // ---- (a) filter, heads -------------------------------------------------------------------
        // Written with integer bit arithmetic throughout: every instruction costs a 4-cycle issue slot, and
        // compare -> mask -> select chains were a third of this kernel's instruction count.
        u32 r_key[RPL], r_bit[RPL], r_rl[RPL];    // locus + 1, haplotype bit, read index within the tile
        u32 m_ok = 0, m_head = 0, m_own = 0;       // bit k: valid & in range / head / head of a read we own
        {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int i0 = g * 256 + 4 * (int)lane;
                // the record before this group's first: the previous lane's last record of the group, or (lane 0) the
                // previous group's very last record
                const u32 up = lane_above(R.rr[4 * g + 3]);
                const u32 wrap = g == 0 ? base - 1u : (u32)__builtin_amdgcn_readlane((int)R.rr[4 * (g > 0 ? g - 1 : 0) + 3], 63);
                const u32 lo_in = (1u << clamp04(p_rel - i0, 0, 4)) - 1u;        // records before p
                const u32 in4 = ((1u << clamp04(te_rel - i0, 0, 4)) - 1u) & ~lo_in;
                const u32 own4 = ((1u << clamp04(c1_rel - i0, 0, 4)) - 1u) & ~lo_in;
                const u32 cnt4 = ((1u << clamp04(cnt_hi - i0, 0, 4)) - 1u) &
                                 ~((1u << clamp04(cnt_lo - i0, 0, 4)) - 1u);
                u32 prev = lane == 0 ? wrap : up;
                u32 ok4 = 0, head4 = 0, big4 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * g + j;
                    const u32 f = R.hh[k], nf = ~f;
                    // record filter (bam_utils.py:264-270): not unmapped, and if paired: read1, proper, mate on the same
                    // reference, next_pos >= 0  <=>  ((f ^ 2) & 0x3082) == 0
                    const u32 pair_ok = ((((f ^ 0x2u) & 0x3082u) - 1u) >> 31);
                    const u32 ok = (nf >> 2) & (nf | pair_ok) & 1u;
                    const u32 step = R.rr[k] - prev;
                    prev = R.rr[k];
                    r_rl[k] = R.rr[k] - base;
                    r_key[k] = R.ll[k] + 1u;
                    r_bit[k] = 1u << ((f >> ECB_HAP_SHIFT) & 31u);
                    ok4 |= ok << j;
                    head4 |= (step & 1u) << j;
                    big4 |= min(step >> 1, 1u) << j;                 // the run counter may only step by 0 or 1
                }
                ok4 &= in4; head4 &= in4;
                bad |= ((big4 & in4) | (head4 & ~ok4)) ? ERR_CONTRACT : 0u;   // ... and only on a valid record
                my_all += __popc(cnt4); my_valid += __popc(cnt4 & ok4);
                m_ok |= ok4 << (4 * g); m_head |= head4 << (4 * g); m_own |= (head4 & own4) << (4 * g);
            }
        }
        TICK(0);
        if (__ballot(bad != 0u)) break;            // never index LDS with a broken run counter
        const u32 sums = wave_sum(__popc(m_head) | (__popc(m_own) << 16));
        const u32 nr = sums & 0xFFFFu, nown = sums >> 16;         // heads in [p, te) / in [p, c1): ours
        {   // a head at record x starts its read and ends the one before it; stored as table-slot offsets
            unsigned short* sh = reinterpret_cast<unsigned short*>(L.seg);
#pragma unroll
            for (int k = 0; k < RPL; ++k)
                if ((m_head >> k & 1u) && r_rl[k] <= (u32)WMAXR) {
                    // tslot(4*lane + c) = 6*lane + tslot(c) for even-multiple-of-4 offsets: one add per record
                    const unsigned short x = (unsigned short)((4u + (4u >> SLOT_SHIFT)) * lane + tslot((k & 3) + (k >> 2) * 256));
                    sh[2 * r_rl[k] + 2] = x;                           // start of read rl   (seg is indexed rl + 1)
                    sh[2 * r_rl[k] + 1] = x;                           // end of read rl - 1 (lands in the unused seg[0] for rl = 0)
                }
            if (lane == 0 && nr <= (u32)WMAXR) {
                const unsigned short x = (unsigned short)tslot((u32)te_rel);
                sh[2 * nr + 1] = x;
                sh[2 * nr + 2] = x;
            }
        }
        wave_sync();
        const bool last_complete = (te == A.n);                  // batches end on a read boundary
        const u32 nrc = last_complete ? nr : (nr ? nr - 1u : 0u);
        const u32 nproc = min(min(nrc, nown), (u32)WMAXR);
        // Reads finished in this tile get their table geometry packed once: first slot (SBITS bits) | end slot (SBITS) | mask of
        // the largest power of two within the range (10 bits).  Probing starts at first + (locus & mask): the loci of a read are
        // mostly consecutive target ids, which low bits never collide on -- cheaper than a multiplicative hash and the
        // collision path below becomes rare.
        if (lane < nproc) {
            const u32 sg = L.seg[lane + 1], s2 = sg & 0xFFFFu, e2 = sg >> 16;
            L.seg[lane + 1] = s2 | (e2 << SBITS) | (min((1u << (31 - __clz((int)(e2 - s2)))) - 1u, 0x3FFu) << (2 * SBITS));
        }
        wave_sync();
        const bool done = (te >= c1 && nown <= nproc);           // every read that starts in our slice
        u64 p_next = te;
        u32 base_next = base + nr;
        bool giant = false;
        if (!done && nproc < nr) {
            // (readfirstlane: the value is wave-uniform; saying so keeps the tile addressing in scalar registers)
            const u64 h = tb + unslot((u32)__builtin_amdgcn_readfirstlane(L.seg[nproc + 1]) & 0xFFFFu);   // first read not finished here
            if (h == p) giant = true;                            // one read fills the whole tile: k_slow
            else { p_next = h; base_next = base + nproc; }
        }
        if (done) p_next = c1;
        TICK(1);
        // ---- prefetch the next tile while this one is hashed and looked up --------------------------
        TileRegs N;
        u32 parked_next = 0;
        if (p_next < c1) {
            load_tile(A, p_next & ~(u64)3, min((p_next & ~(u64)3) + (u64)WT, A.n), lane, N);
            parked_next = __hip_atomic_load(&A.ctr->full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (giant && lane == 0) {
            const u64 qi = atomicAdd(&A.ctr->n_queue, 1ull);
            if (qi < A.queue_cap) A.queue[qi] = p; else atomicOr(&A.ctr->err, ERR_QUEUE);
        }

        // ---- (b) per-read {locus -> haplotype mask} tables in LDS ------------------------------------
        // Staged so that the 8 records' LDS round trips overlap: segment reads, CAS on the locus, OR of the bit.
        // A lane whose CAS created an entry queues it; every entry is hashed once when all masks are final.
        TICK(2);
        u32 n_ent = 0;
        if (!(A.ablate & 1u)) {
            u32 q[RPL], old[RPL], act = 0, coll = 0;
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const u32 on = (m_ok >> k) & (u32)(r_rl[k] < nproc) & 1u;
                act |= on << k;
                const u32 sg = L.seg[on ? r_rl[k] + 1u : 0u];
                q[k] = (sg & SMASK) + (r_key[k] & (sg >> (2 * SBITS)));
            }
#pragma unroll
            for (int k = 0; k < RPL; ++k)
                if (act >> k & 1u) old[k] = atomicCAS(&L.tkey[q[k]], 0u, r_key[k]);
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const bool on = act >> k & 1u;
                const bool made = on && old[k] == 0u;                // this lane created the (read, locus) entry
                const bool hit = made || (on && old[k] == r_key[k]);
                coll |= (u32)(on && !hit) << k;
                if (hit) atomicOr(&L.tmask[q[k]], r_bit[k]);         // duplicate (read, target) records vanish here: bam_utils.py:322-325
                const u64 mm = __ballot(made);
                if (made) L.ent[n_ent + __builtin_amdgcn_mbcnt_hi((u32)(mm >> 32), __builtin_amdgcn_mbcnt_lo((u32)mm, 0u))] =
                    (unsigned short)(q[k] | (r_rl[k] << SBITS));
                n_ent += (u32)__popcll(mm);
            }
            if (__ballot(coll != 0u)) {                              // slot taken by another locus of the read: probe on (rare)
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    bool made = false;
                    if (coll >> k & 1u) {
                        const u32 sg = L.seg[r_rl[k] + 1u];
                        const u32 s2 = sg & SMASK, e2 = (sg >> SBITS) & SMASK;
                        u32 o;
                        bool lapped = false, stuck = false;  // (a read's range always has a free slot: the second lap only keeps a
                        do {                                 //  corrupted geometry from spinning a wave for ever)
                            if (++q[k] == e2) { q[k] = s2; stuck = lapped; lapped = true; }
                            o = atomicCAS(&L.tkey[q[k]], 0u, r_key[k]);
                        } while (o != 0u && o != r_key[k] && !stuck);
                        if (stuck) bad |= ERR_CONTRACT;
                        made = (o == 0u);
                        atomicOr(&L.tmask[q[k]], r_bit[k]);
                    }
                    const u64 mm = __ballot(made);
                    if (made) L.ent[n_ent + __builtin_amdgcn_mbcnt_hi((u32)(mm >> 32), __builtin_amdgcn_mbcnt_lo((u32)mm, 0u))] =
                        (unsigned short)(q[k] | (r_rl[k] << SBITS));
                    n_ent += (u32)__popcll(mm);
                }
            }
        }
        wave_sync();
        TICK(3);
        for (u32 e = lane; e < n_ent; e += 64) {
            const u32 en = L.ent[e], qq = en & SMASK, rl = en >> SBITS;
            u64 a, b;
            pair_hash64(L.tkey[qq] - 1u, L.tmask[qq], a, b);
            atomicAdd(&L.acc[rl][0], a); atomicAdd(&L.acc[rl][1], b);
            atomicAdd(reinterpret_cast<u32*>(&L.npair[rl & ~1u]), (rl & 1u) ? 0x10000u : 1u);   // two 16-bit counters per word
        }
        wave_sync();

        TICK(4);
        // ---- (c) one lane per read: EC lookup; a new EC gets its key from the read's LDS table ---------
        u64 slot = ~0ull;
        bool created = false;
        u32 np = 0;
        const bool on = lane < nproc && !(A.ablate & 3u);
        const u32 rd = base + lane;
        if (on && VERIFY) {                                         // exactness pass: set of this read == key of its EC ?
            const Slot s = A.table[A.read_slot[rd]];
            bool same = s.n == L.npair[lane];
            const u32 b2 = L.seg[lane + 1] & SMASK, f2 = (L.seg[lane + 1] >> SBITS) & SMASK;
            for (u32 t = b2; t < f2 && same; ++t) {
                const u32 kk = L.tkey[t];
                if (!kk) continue;
                bool found = false;
                for (u32 i = 0; i < s.n; ++i) {
                    const uint2 pr = A.arena[s.off + i];
                    found |= (pr.x == kk - 1u) && (pr.y == L.tmask[t]);
                }
                same = found;
            }
            if (!same) my_new += 1;                                 // (counted as "mismatches" in verify mode)
        }
        else if (on) {
            u64 lo, hi;
            np = L.npair[lane];
            finish_hash(L.acc[lane][0], L.acc[lane][1], np, lo, hi);
            if (A.ablate & 4u) slot = lo & A.cap_mask;
            else slot = table_find_or_insert(A.table, A.cap_mask, lo, hi, &created);
        }
        TICK(5);
        // Take over the prefetched tile HERE, right behind the lookup's own wait and before this tile issues any store.
        // vmcnt counts in order: wherever the compiler first touches these registers it waits for everything issued before
        // that point, and at the end of the tile (where the moves sink to if left alone) or at the top of the next one
        // (the parked flag) that meant sitting out the round trips of the founders' stores -- ~3000 clocks per tile on C3.
        R = N; parked = parked_next;
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            asm volatile("" : "+v"(R.rr[k])); asm volatile("" : "+v"(R.ll[k])); asm volatile("" : "+v"(R.hh[k]));
        }
        asm volatile("" : "+v"(parked));
        if (on && !VERIFY) {
            if (slot == ~0ull) {                                    // table too full here: defer the read, park
                atomicExch(&A.ctr->full, 1u);
                const u64 qi = atomicAdd(&A.ctr->n_queue, 1ull);
                if (qi < A.queue_cap) A.queue[qi] = tb + unslot(L.seg[lane + 1] & SMASK); else atomicOr(&A.ctr->err, ERR_QUEUE);
            } else {
                A.read_slot[rd] = (u32)slot;
            }
        }
        {
            const u64 cmask = __ballot(created);
            if (cmask) {                                            // some read of this tile founded an EC
                const u32 want = created ? np : 0u;
                const u32 incl = wave_incl_scan(want);
                const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
                if (total > chunk_left) {                           // reserve another stretch of the key arena
                    const u32 take = max(total, ARENA_CHUNK);
                    u64 at = 0;
                    if (lane == 0) at = arena_alloc(A.ctr, A.arena_cap, take, (u32)pw);
                    chunk_at = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(at >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)at);
                    chunk_left = take;
                    if (chunk_at == ~0ull) { bad |= ERR_ARENA; chunk_left = 0; }
                }
                if (!(bad & ERR_ARENA)) {
                    const u32 off = (u32)(chunk_at + (incl - want));   // the arena holds < 2^32 pairs (Slot::off)
                    if (created) { A.table[slot].off = off; A.table[slot].n = np; }
                    // The key is copied by the whole wave from the tile's entry queue (every (read, locus) entry once, any
                    // order: rows are sorted when they are emitted), not by the founding lane walking its read's table:
                    // that serial walk with 1-3 lanes alive was ~400 issue slots per tile, a quarter of the kernel.
                    for (u32 e0 = 0; e0 < n_ent; e0 += 64) {
                        const u32 e = e0 + lane;
                        const u32 en = e < n_ent ? L.ent[e] : 0u, qq = en & SMASK, rl = en >> SBITS;
                        const u32 o = __shfl(off, rl);                                  // (all lanes: the loop bound is uniform)
                        if (e < n_ent && (cmask >> rl & 1ull)) {
                            const u32 was = atomicSub(reinterpret_cast<u32*>(&L.npair[rl & ~1u]), (rl & 1u) ? 0x10000u : 1u);
                            const u32 pos = ((rl & 1u) ? was >> 16 : was & 0xFFFFu) - 1u;   // a place of its own among the read's pairs
                            A.arena[(u64)o + pos] = make_uint2(L.tkey[qq] - 1u, L.tmask[qq]);
                        }
                    }
                    chunk_at += total; chunk_left -= total;
                }
            }
            if (!VERIFY) my_new += (u32)__popcll(__ballot(created));
        }
        wave_sync();

        TICK(6);
        counted = max(counted, tb + (u64)cnt_hi);
        p = p_next; base = base_next;
    }
    const bool stop = __ballot(bad != 0u) || parked;
    if (__ballot(bad != 0u)) { if (bad) atomicOr(&A.ctr->err, bad); p = c1; }
    if (lane == 0) { A.resume[2 * wid] = p; A.resume[2 * wid + 1] = counted; }
    if (stop) break;
  }
    // records offered / valid: one atomic pair per wave
    const u32 wa = wave_sum(my_all), wv = wave_sum(my_valid);
    // per-wave totals go to their own words: thousands of waves adding to three shared counters serialise (~50 ns each)
    const u32 wn = VERIFY ? wave_sum(my_new) : my_new;
#ifdef ECB_TIMING
    if (lane == 0 && A.timing) for (int i = 0; i < 8; ++i) atomicAdd(A.timing + i, tacc[i]);
#endif
    if (lane == 0) { A.wave_counts[3 * pw] = wa; A.wave_counts[3 * pw + 1] = wv; A.wave_counts[3 * pw + 2] = wn; }
    if (!VERIFY && lane == 0) { A.wave_arena[2 * pw] = chunk_at; A.wave_arena[2 * pw + 1] = chunk_left; }
}

// resume points of a fresh batch: slice b starts (and has counted its records up to) record b * chunk
__global__ void k_init_resume(u64* resume, u64 slices, u64 chunk) {
    const u64 b = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (b < slices) { resume[2 * b] = b * chunk; resume[2 * b + 1] = b * chunk; }
}

__global__ __launch_bounds__(1024) void k_sum_counts(const u32* wave_counts, u64 waves, Counters* ctr) {
    __shared__ u64 s[3][16];
    u64 a = 0, v = 0, e = 0;
    for (u64 i = threadIdx.x; i < waves; i += 1024) { a += wave_counts[3 * i]; v += wave_counts[3 * i + 1]; e += wave_counts[3 * i + 2]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d); v += __shfl_xor(v, d); e += __shfl_xor(e, d); }
    if ((threadIdx.x & 63u) == 0) { s[0][threadIdx.x >> 6] = a; s[1][threadIdx.x >> 6] = v; s[2][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = v = e = 0;
        for (int k = 0; k < 16; ++k) { a += s[0][k]; v += s[1][k]; e += s[2][k]; }
        ctr->all += a; ctr->valid += v; ctr->n_ecs += e;
    }
}

// reference_start ranges per target (bam_utils.py:282-286): a separate pass, only with ECB_F_RANGES
__global__ __launch_bounds__(TPB) void k_ranges(const u32* loc, const u32* hf, const int* pos, u64 n, u32 n_loci, u32 n_haps,
                                                int* rng_min, int* rng_max) {
    for (u64 i = (u64)blockIdx.x * TPB + threadIdx.x; i < n; i += (u64)gridDim.x * TPB) {
        const u32 f = hf[i];
        if (!rec_valid(f)) continue;
        const u32 lc = loc[i], hap = (f >> ECB_HAP_SHIFT) & 0xFFu;
        if (lc >= n_loci || hap >= n_haps) continue;          // k_stream reports these
        const u64 sl = (u64)lc * n_haps + hap;
        const int ps = pos[i];
        if (ps < rng_min[sl]) atomicMin(rng_min + sl, ps);
        if (ps > rng_max[sl]) atomicMax(rng_max + sl, ps);
    }
}

// ---------------------------------------------------------------------------------------------
// k_count: reads per EC and first appearance per EC (bam_utils.py:309-312, 688-698), reduced from the
// per-read slot ids without global atomics: partition the (slot, read) pairs by slot range (LDS
// histogram, scan, scatter), then one workgroup per range counts in LDS and owns its slots' counters.
// ---------------------------------------------------------------------------------------------
constexpr u32 BIN_BITS = 13;                   // slots per range = LDS bins of one k_count_bins workgroup
constexpr u32 N_BINS = 1u << BIN_BITS;
constexpr u32 MAX_BUCKETS = 8192;              // LDS histogram of the partition passes

// Partition passes: few, fat workgroups.  Every workgroup keeps one open line per range in flight (2 048 of them at C3); with
// 512 workgroups of 1 024 threads those lines stay in L2 until they are full far more often than with 1 024 x 256
// (measured: -0.25 ms of 1.4 at C3; non-temporal stores, which skip that write-combining, cost +3.5 ms).
constexpr int TPB_PART = 1024;
constexpr u32 PART_G = 512;
__global__ __launch_bounds__(TPB_PART) void k_part_hist(const u32* read_slot, u64 n_reads, u32 n_buckets, u32* hist) {
    extern __shared__ u32 sh[];
    const u64 G = gridDim.x, g = blockIdx.x, per = (n_reads + G - 1) / G;
    const u64 r0 = g * per, r1 = min(r0 + per, n_reads);
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB_PART) sh[b] = 0;
    __syncthreads();
    for (u64 r = r0 + threadIdx.x; r < r1; r += TPB_PART) {
        const u32 s = read_slot[r];
        if (s != PENDING) atomicAdd(&sh[s >> BIN_BITS], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB_PART) hist[(u64)b * G + g] = sh[b];
}

__global__ __launch_bounds__(TPB_PART) void k_part_scatter(const u32* read_slot, u64 n_reads, u32 n_buckets, const u32* offs,
                                                      uint2* pairs) {
    extern __shared__ u32 sh[];
    const u64 G = gridDim.x, g = blockIdx.x, per = (n_reads + G - 1) / G;
    const u64 r0 = g * per, r1 = min(r0 + per, n_reads);
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB_PART) sh[b] = offs[(u64)b * G + g];
    __syncthreads();
    // four reads per thread and trip: their LDS cursor bumps are independent, so the round trips overlap
    for (u64 rb = r0; rb < r1; rb += 4 * TPB_PART) {
        u32 s[4], pos[4];
        u64 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = rb + (u64)k * TPB_PART + threadIdx.x;
            s[k] = r[k] < r1 ? read_slot[r[k]] : PENDING;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (s[k] != PENDING) pos[k] = atomicAdd(&sh[s[k] >> BIN_BITS], 1u);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (s[k] != PENDING) pairs[pos[k]] = make_uint2(s[k], (u32)r[k]);
    }
}

// The scatter with its elements sorted by range in LDS first: a workgroup takes STAGE reads at a time, ranks them within their
// range (LDS counters), lays them out range by range in LDS and writes them from there -- elements of one range leave in
// runs of consecutive addresses (STAGE / n_buckets of them on average) instead of one 8-byte store per lane and range.
constexpr u32 STAGE = 8 * TPB_PART;            // 8 192 elements = 64 KB of LDS
constexpr u32 STAGE_MAX_BUCKETS = 4096;        // 3 x 16 KB of counters / starts / cursors beside the stage
__global__ __launch_bounds__(TPB_PART) void k_part_scatter_staged(const u32* read_slot, u64 n_reads, u32 n_buckets, const u32* offs,
                                                             uint2* pairs) {
    extern __shared__ u32 sh[];                   // cnt[nb] | start[nb] | gcur[nb] | stage[STAGE] (uint2)
    u32 *cnt = sh, *start = sh + n_buckets, *gcur = sh + 2 * n_buckets;
    uint2* stage = reinterpret_cast<uint2*>(sh + ((3 * n_buckets + 1) & ~1u));      // (8-byte aligned)
    __shared__ u32 s_wave[TPB_PART / 64];
    const u64 G = gridDim.x, g = blockIdx.x, per = (n_reads + G - 1) / G;
    const u64 r0 = g * per, r1 = min(r0 + per, n_reads);
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 bpt = (n_buckets + TPB_PART - 1) / TPB_PART;            // ranges per thread in the scan (1 .. 4)
    for (u32 b = tid; b < n_buckets; b += TPB_PART) gcur[b] = offs[(u64)b * G + g];
    for (u64 rb = r0; rb < r1; rb += STAGE) {
        for (u32 b = tid; b < n_buckets; b += TPB_PART) cnt[b] = 0;
        __syncthreads();
        u32 s[8], lr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u64 r = rb + (u64)k * TPB_PART + tid;
            s[k] = r < r1 ? read_slot[r] : PENDING;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (s[k] != PENDING) lr[k] = atomicAdd(&cnt[s[k] >> BIN_BITS], 1u);
        __syncthreads();
        // exclusive scan of cnt[] -> start[]: bpt consecutive ranges per thread, DPP scan per wave, wave totals through LDS
        u32 mine = 0;
        for (u32 j = 0; j < bpt; ++j) { const u32 b = tid * bpt + j; mine += b < n_buckets ? cnt[b] : 0u; }
        const u32 incl = wave_incl_scan(mine);
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        u32 run = incl - mine;
        for (u32 k = 0; k < w; ++k) run += s_wave[k];
        for (u32 j = 0; j < bpt; ++j) { const u32 b = tid * bpt + j; if (b < n_buckets) { start[b] = run; run += cnt[b]; } }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s[k] != PENDING) stage[start[s[k] >> BIN_BITS] + lr[k]] = make_uint2(s[k], (u32)(rb + (u64)k * TPB_PART + tid));
        __syncthreads();
        u32 n_here = 0;
        for (u32 k = 0; k < TPB_PART / 64; ++k) n_here += s_wave[k];
        for (u32 i = tid; i < n_here; i += TPB_PART) {
            const uint2 e = stage[i];
            const u32 b = e.x >> BIN_BITS;
            pairs[gcur[b] + (i - start[b])] = e;
        }
        __syncthreads();
        for (u32 b = tid; b < n_buckets; b += TPB_PART) gcur[b] += cnt[b];
    }
}

constexpr int TPB_COUNT = 1024;
// A hot EC makes its range the tail of the kernel (C2: one range of 512 held 4 % of the reads): the host cuts ranges
// far above the average into pieces.  One piece = one workgroup; the pieces of a cut range add into the slots with
// atomics, whole ranges are the only writer of their slots and just store.
struct CountWork { u32 bucket, start, end, shared; };
__global__ void k_bucket_starts(const u32* offs, u32 G, u32 n_buckets, u32 total, u32* starts) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_buckets) starts[b] = offs[(u64)b * G];
    if (b == n_buckets) starts[b] = total;
}
__global__ __launch_bounds__(TPB_COUNT) void k_count_bins(const uint2* pairs, const CountWork* work, Slot* table) {
    __shared__ u32 cnt[N_BINS], fst[N_BINS];
    const CountWork wk = work[blockIdx.x];
    const u32 b = wk.bucket, lane = threadIdx.x & 63u;
    const u32 start = wk.start, end = wk.end;
    for (u32 q = threadIdx.x; q < N_BINS; q += TPB_COUNT) { cnt[q] = 0; fst[q] = 0xFFFFFFFFu; }
    __syncthreads();
    for (u32 i0 = start; i0 < end; i0 += 4 * TPB_COUNT) {
        uint2 pr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                       // four independent loads in flight per thread
            const u32 i = i0 + k * TPB_COUNT + threadIdx.x;
            pr[k] = i < end ? pairs[i] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool have = pr[k].x != 0xFFFFFFFFu;
            const u32 bin = pr[k].x & (N_BINS - 1);
            const u64 hm = __ballot(have);
            if (!hm) continue;
            // a hot EC fills most lanes of a wave: add it once per wave, the rest go one by one
            const u32 v = __shfl(bin, __ffsll((long long)hm) - 1);
            const bool same = have && bin == v;
            const u64 m = __ballot(same);
            if (__popcll(m) >= 8) {
                u32 mn = same ? pr[k].y : 0xFFFFFFFFu;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) mn = min(mn, (u32)__shfl_xor(mn, d));
                if (lane == (u32)(__ffsll((long long)m) - 1)) { atomicAdd(&cnt[v], (u32)__popcll(m)); atomicMin(&fst[v], mn); }
                if (have && !same) { atomicAdd(&cnt[bin], 1u); atomicMin(&fst[bin], pr[k].y); }
            } else if (have) {
                atomicAdd(&cnt[bin], 1u); atomicMin(&fst[bin], pr[k].y);
            }
        }
    }
    __syncthreads();
    for (u32 q = threadIdx.x; q < N_BINS; q += TPB_COUNT) {
        const u32 c = cnt[q];
        if (c) {
            Slot* s = table + (((u64)b << BIN_BITS) | q);
            if (wk.shared) { atomicAdd(&s->count, c); atomicMax(&s->first_inv, ~fst[q]); }
            else { s->count += c; s->first_inv = max(s->first_inv, ~fst[q]); }      // the only writer of its slots
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_slow: one workgroup per deferred read (longer than a tile, or bounced off a full table).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_slow_len(const u32* rid, u64 n, const u64* queue, u64 nq, u64* len) {
    // length in records of each queued read (its head index .. the next change of read_id)
    const u64 q = blockIdx.x;
    if (q >= nq) return;
    __shared__ u64 s_end;
    const u64 h = queue[q];
    const u32 r0 = rid[h];
    if (threadIdx.x == 0) s_end = n;
    __syncthreads();
    for (u64 b = h; b < n; b += TPB) {
        const u64 i = b + threadIdx.x;
        if (i < n && rid[i] != r0) atomicMin(&s_end, i);
        __syncthreads();
        const bool found = (s_end != n);
        __syncthreads();
        if (found) break;
    }
    __syncthreads();
    if (threadIdx.x == 0) len[q] = s_end - h;
}

struct SlowArgs {
    const u32* rid; const u32* loc; const u32* hf;
    const u64* queue; const u64* len; const u64* scr_off;   // per queued read
    u32* scr_key; u32* scr_mask;                            // global scratch tables (zeroed)
    u32 n_loci, n_haps;
    Slot* table; u64 cap_mask;
    uint2* arena; u64 arena_cap;
    Counters* ctr;
    u32* read_slot;
    u64* requeue; u64* n_requeue;                           // reads that still found no slot
};

__global__ __launch_bounds__(TPB) void k_slow(SlowArgs A) {
    const u64 q = blockIdx.x;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u64 h = A.queue[q], L = A.len[q];
    const u64 cap2 = 2 * L;
    u32* key = A.scr_key + A.scr_off[q];
    u32* msk = A.scr_mask + A.scr_off[q];
    __shared__ u64 s_acc[TPB / 64][2];
    __shared__ u32 s_n[TPB / 64];
    __shared__ u64 s_off;
    __shared__ u32 s_created, s_cnt, s_fits;

    for (u64 i = tid; i < L; i += TPB) {
        const u32 f = A.hf[h + i];
        if (!rec_valid(f)) continue;
        const u32 lc = A.loc[h + i], hap = (f >> ECB_HAP_SHIFT) & 0xFFu;
        if (lc >= A.n_loci || hap >= A.n_haps) { atomicOr(&A.ctr->err, ERR_RANGE); continue; }
        u64 p = __umul64hi((u64)(lc * 0x9E3779B1u) << 32, cap2);
        for (;;) {
            const u32 old = atomicCAS(&key[p], 0u, lc + 1u);
            if (old == 0u || old == lc + 1u) { atomicOr(&msk[p], 1u << hap); break; }
            if (++p == cap2) p = 0;
        }
    }
    __threadfence();
    __syncthreads();
    u64 a0 = 0, a1 = 0;
    u32 np = 0;
    for (u64 p = tid; p < cap2; p += TPB) {
        const u32 k = __hip_atomic_load(&key[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k) {
            const u32 m = __hip_atomic_load(&msk[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++np;
            u64 x, y; pair_hash64(k - 1u, m, x, y);         // same set hash as k_stream
            a0 += x; a1 += y;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a0 += __shfl_xor(a0, d); a1 += __shfl_xor(a1, d); }
    np = wave_sum(np);
    if (lane == 0) { s_acc[tid >> 6][0] = a0; s_acc[tid >> 6][1] = a1; s_n[tid >> 6] = np; }
    __syncthreads();
    if (tid == 0) {
        a0 = a1 = 0; np = 0;
        for (int w = 0; w < TPB / 64; ++w) { a0 += s_acc[w][0]; a1 += s_acc[w][1]; np += s_n[w]; }
        u64 lo, hi; finish_hash(a0, a1, np, lo, hi);
        bool created = false;
        const u32 r0 = A.rid[h];
        const u64 slot = table_find_or_insert(A.table, A.cap_mask, lo, hi, &created);
        s_created = 0; s_cnt = 0; s_fits = 0; s_off = 0;
        if (slot == ~0ull) {
            A.requeue[atomicAdd(A.n_requeue, 1ull)] = h;
        } else {
            A.read_slot[r0] = (u32)slot;
            if (created) {
                const u64 off = arena_alloc(A.ctr, A.arena_cap, np, blockIdx.x);
                atomicAdd(&A.ctr->n_ecs, 1ull);
                s_created = 1; s_off = off; s_fits = (off != ~0ull);
                if (s_fits) { A.table[slot].off = (u32)off; A.table[slot].n = np; }
                else atomicOr(&A.ctr->err, ERR_ARENA);
            }
        }
    }
    __syncthreads();
    for (u64 p = tid; p < cap2; p += TPB) {
        const u32 k = __hip_atomic_load(&key[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k) {
            if (s_created && s_fits) {
                const u32 m = __hip_atomic_load(&msk[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                A.arena[s_off + atomicAdd(&s_cnt, 1u)] = make_uint2(k - 1u, m);
            }
            key[p] = 0; msk[p] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// table maintenance: grow (rehash), compact, merge
// ---------------------------------------------------------------------------------------------
__global__ void k_rehash(const Slot* old_t, u64 old_cap, Slot* new_t, u64 new_mask) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < old_cap; i += (u64)gridDim.x * blockDim.x) {
        const Slot s = old_t[i];
        if (!s.hi) continue;
        u64 j = s.lo & new_mask;
        for (;; j = (j + 1) & new_mask) {                // new table is larger and keys are distinct
            if (atomicCAS(&new_t[j].lo, 0ull, s.lo) == 0ull) {
                new_t[j].hi = s.hi; new_t[j].count = s.count; new_t[j].first_inv = s.first_inv;
                new_t[j].off = s.off; new_t[j].n = s.n;
                break;
            }
        }
    }
}

__global__ void k_remap_read_slot(u32* read_slot, u64 n_reads, const Slot* old_t, const Slot* new_t, u64 new_mask) {
    for (u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x; r < n_reads; r += (u64)gridDim.x * blockDim.x) {
        const u32 os = read_slot[r];
        if (os == 0xFFFFFFFFu) continue;
        const u64 lo = old_t[os].lo, hi = old_t[os].hi;
        u64 j = lo & new_mask;
        while (!(new_t[j].lo == lo && new_t[j].hi == hi)) j = (j + 1) & new_mask;
        read_slot[r] = (u32)j;
    }
}

// occupied slots -> dense list (order irrelevant: ranks come from `first`); one global atomic per 4096 slots
constexpr int TPB_COMPACT = 1024;
// (finalize passes bitmap / list_fn: the slot being scanned already holds the EC's first read and key length, so the
//  first-appearance bitmap is marked and both are written next to the list here -- coalesced -- instead of being gathered
//  from the table again by separate kernels)
__global__ __launch_bounds__(TPB_COMPACT) void k_compact(const Slot* table, u64 cap, u32* list, u64 max_list, u64* n_list,
                                                         u32* bitmap, u64 n_bits, uint2* list_fn) {
    __shared__ u32 s_w[TPB_COMPACT / 64];
    __shared__ u64 s_base;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    for (u64 b = (u64)blockIdx.x * 4 * TPB_COMPACT; b < cap; b += (u64)gridDim.x * 4 * TPB_COMPACT) {
        u32 occ = 0, off[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u64 i = b + (u64)k * TPB_COMPACT + tid;
            const bool o = i < cap && table[i].hi != 0ull;
            const u64 m = __ballot(o);
            off[k] = tot + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            tot += (u32)__popcll(m);
            occ |= (u32)o << k;
        }
        if (lane == 0) s_w[w] = tot;
        __syncthreads();
        if (tid == 0) {
            u32 run = 0;
            for (int k = 0; k < TPB_COMPACT / 64; ++k) { const u32 c = s_w[k]; s_w[k] = run; run += c; }
            s_base = run ? atomicAdd(n_list, (u64)run) : 0ull;
        }
        __syncthreads();
        const u64 base = s_base + s_w[w];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((occ >> k & 1u) && base + off[k] < max_list) {
                const u64 i = b + (u64)k * TPB_COMPACT + tid;
                list[base + off[k]] = (u32)i;
                if (list_fn) {
                    const u32 f = ~table[i].first_inv;
                    list_fn[base + off[k]] = make_uint2(f, table[i].n);
                    if (f < n_bits) atomicOr(&bitmap[f >> 5], 1u << (f & 31u));
                }
            }
        __syncthreads();
    }
}

// Partitioned export (multi-GPU merge by key range): part of an EC = a few high bits of its key, so every rank sends
// part p of its table to rank p, rank p merges what it gets, and no EC is ever merged on two ranks.
constexpr u32 MAX_PARTS = 64;
__device__ __forceinline__ u32 part_of(u64 lo, u32 n_parts) { return (u32)((lo >> 40) % n_parts); }   // (the low bits pick the table slot)

constexpr u32 PARTS_PER_BLOCK = 16 * TPB;     // entries per workgroup: same-address global atomics cost ~50 ns each, so few of them
__global__ __launch_bounds__(TPB) void k_parts_count(const Slot* table, const u32* list, u64 n, u32 n_parts, u64* cnt) {
    __shared__ u32 ce[MAX_PARTS], cp[MAX_PARTS];
    if (threadIdx.x < MAX_PARTS) { ce[threadIdx.x] = 0; cp[threadIdx.x] = 0; }
    __syncthreads();
    const u64 e0 = blockIdx.x * (u64)PARTS_PER_BLOCK, e1 = min(e0 + PARTS_PER_BLOCK, n);
    for (u64 eb = e0; eb < e1; eb += TPB) {
        const u64 e = eb + threadIdx.x;
        u32 q = MAX_PARTS, np = 0;
        if (e < e1) { const Slot& s = table[list[e]]; q = part_of(s.lo, n_parts); np = s.n; }
        for (u32 t = 0; t < n_parts; ++t) {          // one LDS atomic per wave and part (few parts = few, hot counters)
            const u64 m = __ballot(q == t);
            if (!m) continue;
            const u32 tot = wave_sum(q == t ? np : 0u);
            if ((threadIdx.x & 63u) == 0) { atomicAdd(&ce[t], (u32)__popcll(m)); atomicAdd(&cp[t], tot); }
        }
    }
    __syncthreads();
    if (threadIdx.x < n_parts) {
        if (ce[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (u64)ce[threadIdx.x]);
        if (cp[threadIdx.x]) atomicAdd(&cnt[n_parts + threadIdx.x], (u64)cp[threadIdx.x]);
    }
}
// cur[0..n_parts) / cur[n_parts..2 n_parts): next free entry / pair of every part (start = the part's offset);
// pair_base[q] = first pair of part q: Slot::off is written relative to it.  Two walks over the workgroup's entries:
// count per part, reserve (one global atomic per part), then place.
__global__ __launch_bounds__(TPB) void k_parts_export(const Slot* table, const u32* list, u64 n, const uint2* arena, u32 n_parts,
                                                      u64* cur, const u64* pair_base, Slot* out_e, uint2* out_p, u32 read_base) {
    __shared__ u32 ce[MAX_PARTS], cp[MAX_PARTS];
    __shared__ u64 be[MAX_PARTS], bp[MAX_PARTS];
    const u32 lane = threadIdx.x & 63u;
    const u64 e0 = blockIdx.x * (u64)PARTS_PER_BLOCK, e1 = min(e0 + PARTS_PER_BLOCK, n);
    for (int pass = 0; pass < 2; ++pass) {
        if (threadIdx.x < MAX_PARTS) { ce[threadIdx.x] = 0; cp[threadIdx.x] = 0; }
        __syncthreads();
        for (u64 eb = e0; eb < e1; eb += TPB) {
            const u64 e = eb + threadIdx.x;
            Slot s{};
            u32 q = MAX_PARTS, re = 0, rp = 0;
            if (e < e1) { s = table[list[e]]; q = part_of(s.lo, n_parts); }
            for (u32 t = 0; t < n_parts; ++t) {      // rank within the workgroup: wave prefix + one LDS atomic per wave and part
                const bool mine = q == t;
                const u64 m = __ballot(mine);
                if (!m) continue;
                const u32 v = mine ? s.n : 0u, incl = wave_incl_scan(v);
                const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);   // (read here, with every lane alive: inside the branch below
                u32 b_e = 0, b_p = 0;                                            //  the compiler sinks the scan's last add under exec = lane 0)
                if (lane == 0) { b_e = atomicAdd(&ce[t], (u32)__popcll(m)); b_p = atomicAdd(&cp[t], tot); }
                b_e = (u32)__builtin_amdgcn_readfirstlane((int)b_e); b_p = (u32)__builtin_amdgcn_readfirstlane((int)b_p);
                if (mine) { re = b_e + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); rp = b_p + incl - v; }
            }
            if (pass == 1 && e < e1) {
                const u64 po = bp[q] + rp;
                for (u32 t = 0; t < s.n; ++t) out_p[po + t] = arena[s.off + t];
                s.first_inv = ~(~s.first_inv + read_base);
                s.off = (u32)(po - pair_base[q]);
                out_e[be[q] + re] = s;
            }
        }
        __syncthreads();
        if (pass == 0 && threadIdx.x < n_parts) {
            be[threadIdx.x] = ce[threadIdx.x] ? atomicAdd(&cur[threadIdx.x], (u64)ce[threadIdx.x]) : 0ull;
            bp[threadIdx.x] = cp[threadIdx.x] ? atomicAdd(&cur[n_parts + threadIdx.x], (u64)cp[threadIdx.x]) : 0ull;
        }
        __syncthreads();
    }
}
// adopt: entries known to be distinct ECs go to consecutive slots of an empty table, keys to the arena, no hashing
__global__ void k_adopt(const Slot* ent, u64 n, Slot* table, u64 slot_base, u32 arena_base) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    Slot s = ent[e];
    s.off += arena_base;
    table[slot_base + e] = s;
}

constexpr u32 MERGE_PER_BLOCK = 16 * TPB;     // entries per workgroup (one EC-count atomic per workgroup, not per wave: ~18 ns each on one address)
__global__ __launch_bounds__(TPB) void k_merge(const Slot* ent, u64 n, const uint2* pairs, u64 n_pairs, Slot* table, u64 cap_mask,
                                               uint2* arena, u64 arena_cap, Counters* ctr) {
    __shared__ u32 s_new;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const u32 lane = threadIdx.x & 63u;
    const u64 e0 = blockIdx.x * (u64)MERGE_PER_BLOCK, e1 = min(e0 + MERGE_PER_BLOCK, n);
    u32 my_new = 0;                                // (wave-uniform)
    for (u64 eb = e0; eb < e1; eb += TPB) {
        const u64 e = eb + threadIdx.x;
        const bool on = e < e1;
        Slot s{};
        if (on) s = ent[e];
        const bool ok = on && (u64)s.off + s.n <= n_pairs;
        if (on && !ok) atomicOr(&ctr->err, ERR_CONTRACT);
        bool created = false;
        u64 j = ~0ull;
        if (ok) {
            j = table_find_or_insert(table, cap_mask, s.lo, s.hi, &created);
            if (j == ~0ull) atomicAdd(&ctr->n_queue, 1ull);        // host sizes the table so this cannot happen
            else {
                atomicAdd(&table[j].count, s.count);                 // one pair of atomics per merged EC, not per read
                atomicMax(&table[j].first_inv, s.first_inv);
            }
        }
        // key arena: one reservation per wave, not per created EC
        const u64 cm = __ballot(created);
        if (!cm) continue;
        const u32 want = created ? s.n : 0u, incl = wave_incl_scan(want);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        my_new += (u32)__popcll(cm);
        u64 at = 0;
        if (lane == 0) at = arena_alloc(ctr, arena_cap, total, blockIdx.x * (TPB / 64) + (threadIdx.x >> 6));
        at = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(at >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)at);
        if (at == ~0ull) { if (lane == 0) atomicOr(&ctr->err, ERR_ARENA); continue; }
        if (created) {
            const u64 off = at + (incl - want);
            for (u32 t = 0; t < s.n; ++t) arena[off + t] = pairs[s.off + t];
            table[j].off = (u32)off; table[j].n = s.n;
        }
    }
    if (lane == 0 && my_new) atomicAdd(&s_new, my_new);
    __syncthreads();
    if (threadIdx.x == 0 && s_new) atomicAdd(&ctr->n_ecs, (u64)s_new);
}

// ---------------------------------------------------------------------------------------------
// finalize: rank by first appearance, CSR emit
// ---------------------------------------------------------------------------------------------
__global__ void k_popc(const u32* in, u64 n, u32* out) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = __popc(in[i]);
}

// exclusive scan of u32 (three launches): per-block sums, scan of the sums, per-block scan + offset
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = TPB * SCAN_ITEMS;
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* total) {   // over TPB threads
    __shared__ u32 s_w[TPB / 64];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(incl, d); if (lane >= (u32)d) incl += t; }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u32 add = 0, tot = 0;
    for (u32 k = 0; k < TPB / 64; ++k) { if (k < w) add += s_w[k]; tot += s_w[k]; }
    __syncthreads();
    *total = tot;
    return add + incl - v;
}
__global__ __launch_bounds__(TPB) void k_scan_sums(const u32* in, u64 n, u32* sums) {
    const u64 b0 = (u64)blockIdx.x * SCAN_BLOCK;
    u32 s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) { const u64 i = b0 + (u64)threadIdx.x * SCAN_ITEMS + k; if (i < n) s += in[i]; }
    u32 tot; block_excl_scan(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(TPB) void k_scan_top(u32* sums, u64 nb, u32* grand) {   // one block
    u32 carry = 0;
    for (u64 b0 = 0; b0 < nb; b0 += TPB) {
        const u64 i = b0 + threadIdx.x;
        const u32 v = i < nb ? sums[i] : 0u;
        u32 tot; const u32 ex = block_excl_scan(v, &tot);
        if (i < nb) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand = carry;
}
__global__ __launch_bounds__(TPB) void k_scan_apply(const u32* in, u64 n, const u32* sums, u32* out) {
    const u64 b0 = (u64)blockIdx.x * SCAN_BLOCK;
    u32 v[SCAN_ITEMS], s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) { const u64 i = b0 + (u64)threadIdx.x * SCAN_ITEMS + k; v[k] = i < n ? in[i] : 0u; s += v[k]; }
    u32 tot; u32 ex = block_excl_scan(s, &tot) + sums[blockIdx.x];
    for (int k = 0; k < SCAN_ITEMS; ++k) { const u64 i = b0 + (u64)threadIdx.x * SCAN_ITEMS + k; if (i < n) out[i] = ex; ex += v[k]; }
}

__global__ void k_rank(const u32* list, const uint2* list_fn, u64 n, u64 n_bits, const u32* bitmap, const u32* wprefix,
                       u32* order, u32* rowlen, u32* rank_of_slot) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u32 si = list[e];
    const uint2 fn = list_fn[e];
    const u32 f = fn.x;
    if (f >= n_bits) return;                       // (an EC without a first read: finalize's count check reports it)
    const u32 r = wprefix[f >> 5] + __popc(bitmap[f >> 5] & ((1u << (f & 31u)) - 1u));
    if (r >= n) return;
    order[r] = si;
    rowlen[r] = fn.y;
    rank_of_slot[si] = r;
}

// CSR rows: every (locus, mask) pair of an EC is ranked by locus and written in place (columns ascending, as scipy's
// csc -> csr leaves them: bin_utils.py:211).  Short rows: one thread each.  Long rows: queued, one wave each.
// The indices the host supplied are validated here, once per EC instead of once per record.
constexpr u32 EMIT_SMALL = 16;
__global__ __launch_bounds__(TPB) void k_emit_small(const Slot* table, const u32* order, u64 n, const uint2* arena,
                                                     const u32* indptr, int* indices, int* data, int* counts,
                                                     u32 n_loci, u32 n_haps, u32* big, u32* n_big, Counters* ctr) {
    const u64 e = (u64)blockIdx.x * TPB + threadIdx.x;
    if (e >= n) return;
    const Slot s = table[order[e]];
    counts[e] = (int)s.count;
    if (s.n > EMIT_SMALL) { big[atomicAdd(n_big, 1u)] = (u32)e; return; }
    const uint2* src = arena + s.off;
    const u32 dst = indptr[e];
    bool bad = false;
    for (u32 i = 0; i < s.n; ++i) {
        const uint2 pi = src[i];
        u32 r = 0;
        for (u32 j = 0; j < s.n; ++j) r += src[j].x < pi.x;   // loci within a key are distinct
        indices[dst + r] = (int)pi.x;
        data[dst + r] = (int)pi.y;
        bad |= pi.x >= n_loci || (pi.y >> n_haps) != 0u;
    }
    if (bad) atomicOr(&ctr->err, ERR_RANGE);
}
__global__ __launch_bounds__(TPB) void k_emit_big(const Slot* table, const u32* order, const u32* big, u32 n_big,
                                                   const uint2* arena, const u32* indptr, int* indices, int* data,
                                                   u32 n_loci, u32 n_haps, Counters* ctr) {
    const u32 b = (blockIdx.x * TPB + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (b >= n_big) return;
    const u32 e = big[b];
    const Slot s = table[order[e]];
    const uint2* src = arena + s.off;
    const u32 dst = indptr[e];
    bool bad = false;
    for (u32 i = lane; i < s.n; i += 64) {
        const uint2 pi = src[i];
        u32 r = 0;
        for (u32 j = 0; j < s.n; ++j) r += src[j].x < pi.x;
        indices[dst + r] = (int)pi.x;
        data[dst + r] = (int)pi.y;
        bad |= pi.x >= n_loci || (pi.y >> n_haps) != 0u;
    }
    if (bad) atomicOr(&ctr->err, ERR_RANGE);
}

// multisample: key = EC rank << 32 | meta (cell, file) of every read
__global__ void k_ms_keys(const u32* read_slot, const u32* rank_of_slot, const u32* meta, u64 n, u64* keys, u32* vals) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = ((u64)rank_of_slot[read_slot[i]] << 32) | meta[i]; vals[i] = (u32)i; }
}
__global__ void k_ms_heads(const u64* keys, u64 n, u32* flag) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_ms_emit(const u64* keys, const u32* vals, const u32* flag, const u32* pos, u64 n, u32 n_out,
                          u64* okey, u32* ofirst, u32* ostart) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n && flag[i]) { okey[pos[i]] = keys[i]; ofirst[pos[i]] = vals[i]; ostart[pos[i]] = (u32)i; }
    if (i == 0) ostart[n_out] = (u32)n;
}
__global__ void k_ms_split(const u64* okey, const u32* ostart, const u32* ocount, u64 n, u32* ec, u32* meta, u32* count) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { ec[i] = (u32)(okey[i] >> 32); meta[i] = (u32)okey[i]; count[i] = ocount ? ocount[i] : ostart[i + 1] - ostart[i]; }
}
// multisample across GPUs: the keys of the final ECs in rank order; their ranks looked up in a shard's own table; the
// shards' (EC, cell, file) triples combined on the root
__global__ void k_export_keys(const Slot* table, const u32* order, u64 n, uint4* out) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const Slot& s = table[order[e]];
    out[e] = make_uint4((u32)s.lo, (u32)(s.lo >> 32), (u32)s.hi, (u32)(s.hi >> 32));
}
__global__ void k_set_global_rank(const uint4* keys, u64 n, const Slot* table, u64 cap_mask, u32* grank) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint4 k = keys[e];
    const u64 lo = ((u64)k.y << 32) | k.x, hi = ((u64)k.w << 32) | k.z;
    u64 j = lo & cap_mask;
    for (u64 probe = 0; probe <= cap_mask; ++probe, j = (j + 1) & cap_mask) {
        const u64 clo = table[j].lo;
        if (clo == 0ull) return;                                  // this shard never saw that EC
        if (clo == lo && table[j].hi == hi) { grank[j] = (u32)e; return; }
    }
}
__global__ void k_ms_out(const u64* okey, const u32* ofirst, const u32* ostart, u64 n, u32 read_base, u64* key, u32* count, u32* first) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { key[i] = okey[i]; count[i] = ostart[i + 1] - ostart[i]; first[i] = ofirst[i] + read_base; }
}
__global__ void k_ms_combine(const u64* keys, const u32* idx, const u32* flag, const u32* pos, u64 n, const u32* cnt_in, const u32* first_in,
                             u64* okey, u32* ocount, u32* ofirst) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 o = pos[i] - (flag[i] ? 0u : 1u);                  // pos = exclusive scan of the head flags
    if (flag[i]) okey[o] = keys[i];
    atomicAdd(&ocount[o], cnt_in[idx[i]]);
    atomicMin(&ofirst[o], first_in[idx[i]]);
}


// ---------------------------------------------------------------------------------------------
// f-2: CSR(bitmask) <-> per-haplotype CSC (bin_utils.py:979-1028).  Expand to (key, value) pairs, stable radix sort
// (rocprim), locate rows / columns by binary search on the sorted keys.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 lower_bound_u32(const u32* a, u64 n, u32 v) {
    u64 lo = 0, hi = n;
    while (lo < hi) { const u64 m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo;
}
__device__ __forceinline__ u64 lower_bound_u64(const u64* a, u64 n, u64 v) {
    u64 lo = 0, hi = n;
    while (lo < hi) { const u64 m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo;
}
__global__ void k_cv_popc(const int* data, u64 nnz, u32* cnt) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < nnz) cnt[i] = __popc((u32)data[i]);
}
__global__ void k_cv_expand(const int* indptr, u32 n_ecs, const int* indices, const int* data, u64 nnz, const u32* pos,
                            u32 n_loci, u32* keys, u32* vals) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    u32 lo = 0, hi = n_ecs;                                  // row of entry i: last e with indptr[e] <= i
    while (lo < hi) { const u32 m = (lo + hi + 1) >> 1; if ((u64)indptr[m] <= i) lo = m; else hi = m - 1; }
    u32 m = (u32)data[i], at = pos[i];
    while (m) { const u32 h = __ffs(m) - 1; m &= m - 1; keys[at] = h * n_loci + (u32)indices[i]; vals[at] = lo; ++at; }
}
__global__ void k_cv_cscptr(const u32* keys, u64 total, u32 n_loci, u32 n_haps, int* cscptr) {
    const u64 c = blockIdx.x * (u64)blockDim.x + threadIdx.x;          // c = h * (T + 1) + t
    if (c >= (u64)n_haps * (n_loci + 1)) return;
    const u32 h = (u32)(c / (n_loci + 1)), t = (u32)(c % (n_loci + 1));
    cscptr[c] = (int)(lower_bound_u32(keys, total, h * n_loci + t) - lower_bound_u32(keys, total, h * n_loci));
}
__global__ void k_cv_back_expand(const int* cscptr, const int* cscidx, u64 total, u32 n_loci, u32 n_haps, const u64* hap_start,
                                 u64* keys, u32* vals) {
    const u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (g >= total) return;
    u32 h = 0;
    while (h + 1 < n_haps && hap_start[h + 1] <= g) ++h;
    const int* ptr = cscptr + (u64)h * (n_loci + 1);
    const u64 j = g - hap_start[h];
    u32 lo = 0, hi = n_loci;                                 // column of entry j: last t with ptr[t] <= j
    while (lo < hi) { const u32 m = (lo + hi + 1) >> 1; if ((u64)ptr[m] <= j) lo = m; else hi = m - 1; }
    keys[g] = (u64)(u32)cscidx[g] * n_loci + lo;
    vals[g] = 1u << h;
}
__global__ void k_cv_back_emit(const u64* keys, const u32* vals, const u32* flag, const u32* pos, u64 total, u32 n_loci,
                               int* indices, int* data) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= total || !flag[i]) return;
    u32 m = 0;
    for (u64 j = i; j < total && keys[j] == keys[i]; ++j) m |= vals[j];
    indices[pos[i]] = (int)(keys[i] % n_loci);
    data[pos[i]] = (int)m;
}
__global__ void k_cv_back_rowptr(const u64* keys, const u32* pos, u64 total, u32 nnz, u32 n_ecs, u32 n_loci, int* indptr) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e > n_ecs) return;
    const u64 at = lower_bound_u64(keys, total, (u64)e * n_loci);
    indptr[e] = at < total ? (int)pos[at] : (int)nnz;       // pos[] = index of the run that starts at or after `at`
}

__global__ void k_iota(int* out, u64 n) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)i;
}
__global__ void k_read_ec(const u32* read_slot, u64 n, const u32* rank_of_slot, int* out) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = read_slot[i] == 0xFFFFFFFFu ? -1 : (int)rank_of_slot[read_slot[i]];
}
__global__ void k_range_len(const int* mn, const int* mx, u64 n, long long* out) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = mx[i] == INT_MIN ? 0ll : (long long)mx[i] - (long long)mn[i] + 1ll;
}
__global__ void k_fill_i32(int* p, u64 n, int v) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

// =============================================================================================
// host side
// =============================================================================================
struct ecb_handle {
    ecb_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool finalized = false;
    bool counted = false;             // Slot::count / first_inv hold the reads pushed so far (k_count ran)
    u64* wave_arena = nullptr; u64 wave_arena_n = 0;   // see StreamArgs::wave_arena
    bool scatter_attr_set = false;
    u64 resident_blocks = 0, rounds = 32;     // k_stream's launch shape (queried once)
    bool adopted = false;             // the table holds adopted entries in consecutive slots (no hashing): finalize / export only

    Slot* table = nullptr; u64 cap = 0;
    uint2* arena = nullptr; u64 arena_cap = 0;
    Counters* ctr = nullptr;
    Counters hctr{};                  // last read-back
    u32* read_slot = nullptr; u64 read_slot_cap = 0;
    u32* meta = nullptr; u64 meta_cap = 0, meta_hi = 0;   // multisample: cell | file << 22 per read
    u64 n_triples = 0; u64* ms_okey = nullptr; u32 *ms_ofirst = nullptr, *ms_ostart = nullptr, *ms_ocount = nullptr;
    bool ms_adopted = false;          // the triples came from ecb_ms_adopt_triples_device (multi-GPU)
    int *rng_min = nullptr, *rng_max = nullptr;
    u64* queue = nullptr; u64 queue_cap = 0;
    u64 n_ecs() const { return hctr.n_ecs; }
    u32 prev_rid = 0xFFFFFFFFu;       // read_id of the last record pushed so far
    u64 n_reads = 0;
    u64 reads_hi = 0;                 // read_slot entries [0, reads_hi) may be set (n_reads, or more mid-batch)
    u64 extra_all = 0, extra_valid = 0, extra_reads = 0;   // counters merged in from other ranks

    // host-pointer staging
    u32 *st_rid = nullptr, *st_loc = nullptr, *st_hf = nullptr; int* st_pos = nullptr; u64 st_cap = 0;
    std::vector<u32> c_rid, c_loc, c_hf; std::vector<int> c_pos;   // open read carried between pushes

    // results
    u32* list = nullptr; u64 n_list = 0;
    u32 *order = nullptr, *rank_of_slot = nullptr, *indptr = nullptr;
    int *indices = nullptr, *data = nullptr, *counts = nullptr;
    ecb_sizes sizes{};

    // device scratch reused across calls (grown on demand, freed at destroy)
    enum { P_RESUME, P_SUMS, P_HIST, P_OFFS, P_PAIRS, P_CNT, P_PARTS, P_STARTS, P_WORK, P_LIST, P_BITMAP, P_WPOP, P_WPREFIX, P_ROWLEN, P_ORDER,
           P_WCOUNTS, P_RANK, P_INDPTR, P_COUNTS, P_INDICES, P_DATA, P_MS_KEYS, P_MS_KEYS2, P_MS_VALS, P_MS_VALS2, P_MS_TMP,
           P_MS_FLAG, P_MS_POS, P_MS_OKEY, P_MS_OFIRST, P_MS_OSTART, P_MS_X, P_MS_OCOUNT, P_MS_GRANK, P_MS_CIN, P_MS_FIN, P_LISTFN, P_N };
    void* pool[P_N] = {}; u64 pool_bytes[P_N] = {};

    // profiling
    bool prof = false; double prof_ms = 0; u64 prof_launches = 0, prof_records = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

std::string g_create_err;

int fail(ecb_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}
#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail(h, ECB_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

u64 next_pow2(u64 x) { u64 p = 1; while (p < x) p <<= 1; return p; }
inline unsigned nblk(u64 n, unsigned per) { return (unsigned)((n + per - 1) / per); }

template <class T>
int pool_get(ecb_handle* h, int id, u64 count, T** out) {
    const u64 need = std::max<u64>(count, 1) * sizeof(T);
    if (h->pool_bytes[id] < need) {
        if (h->pool[id]) hipFree(h->pool[id]);
        h->pool[id] = nullptr; h->pool_bytes[id] = 0;
        const u64 take = need + need / 4;
        HIPCHK(h, hipMalloc(&h->pool[id], take));
        h->pool_bytes[id] = take;
    }
    *out = reinterpret_cast<T*>(h->pool[id]);
    return ECB_OK;
}
#define POOL(h, id, ptr, count) do { int rc_ = pool_get(h, ecb_handle::id, count, &(ptr)); if (rc_ != ECB_OK) return rc_; } while (0)

// zero the device counters and point every arena region's cursor at its first pair
int clear_counters(ecb_handle* h) {
    Counters c{};
    const u32 R = arena_regions(h->arena_cap);
    const u64 per = h->arena_cap / R;
    for (u32 r = 0; r < ARENA_REGIONS; ++r) c.arena_reg[r] = r < R ? r * per : h->arena_cap;
    h->hctr = c;
    HIPCHK(h, hipMemcpyAsync(h->ctr, &h->hctr, sizeof(Counters), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}
// pairs of the key arena handed out so far (an upper bound of the pairs in use: the tail of a wave's last chunk is idle)
u64 arena_used(const ecb_handle* h) {
    const u32 R = arena_regions(h->arena_cap);
    const u64 per = h->arena_cap / R;
    u64 used = h->hctr.arena_top;
    for (u32 r = 0; r < R; ++r) used += std::min<u64>(h->hctr.arena_reg[r], (u64)(r + 1) * per) - r * per;
    return used;
}

int sync_counters(ecb_handle* h) {
    HIPCHK(h, hipMemcpyAsync(&h->hctr, h->ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->hctr.err & ERR_CONTRACT) return fail(h, ECB_ERR_CONTRACT, "read_id run counter violates the tuple contract (see ecb.h)");
    if (h->hctr.err & ERR_RANGE) return fail(h, ECB_ERR_CONTRACT, "locus or haplotype index out of range in a valid record");
    if (h->hctr.err & ERR_ARENA) return fail(h, ECB_ERR_TABLE_FULL, "EC key arena exhausted (%llu pairs): raise arena_capacity", (unsigned long long)h->arena_cap);
    if (h->hctr.err & ERR_QUEUE) return fail(h, ECB_ERR_TABLE_FULL, "deferred-read queue exhausted");
    return ECB_OK;
}

int grow_table(ecb_handle* h, u64 new_cap) {
    Slot* nt = nullptr;
    HIPCHK(h, hipMalloc(&nt, new_cap * sizeof(Slot)));
    HIPCHK(h, hipMemsetAsync(nt, 0, new_cap * sizeof(Slot), h->stream));
    k_rehash<<<2048, TPB, 0, h->stream>>>(h->table, h->cap, nt, new_cap - 1);
    if (h->reads_hi)
        k_remap_read_slot<<<2048, TPB, 0, h->stream>>>(h->read_slot, h->reads_hi, h->table, nt, new_cap - 1);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(h->table));
    h->table = nt; h->cap = new_cap;
    return ECB_OK;
}

int ensure_read_slot(ecb_handle* h, u64 need) {
    if (need <= h->read_slot_cap) return ECB_OK;
    u64 nc = std::max<u64>(need, h->read_slot_cap * 2);
    nc = std::max<u64>(nc, 1024);
    u32* p = nullptr;
    HIPCHK(h, hipMalloc(&p, nc * sizeof(u32)));
    HIPCHK(h, hipMemsetAsync(p, 0xFF, nc * sizeof(u32), h->stream));
    if (h->read_slot) {
        HIPCHK(h, hipMemcpyAsync(p, h->read_slot, h->n_reads * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(h->read_slot));
    }
    h->read_slot = p; h->read_slot_cap = nc;
    return ECB_OK;
}

// deferred reads: measure, scratch, k_slow; grow the table and repeat while reads bounce
int run_slow(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, u64 n, u64* d_q, u64 nq) {
    u64* d_requeue = nullptr;
    int rc = ECB_OK;
    while (nq) {
        u64 *d_len = nullptr, *d_off = nullptr, *d_nre = nullptr;
        HIPCHK(h, hipMalloc(&d_len, nq * sizeof(u64)));
        HIPCHK(h, hipMalloc(&d_off, nq * sizeof(u64)));
        HIPCHK(h, hipMalloc(&d_nre, sizeof(u64)));
        u64* nre_buf = nullptr;
        HIPCHK(h, hipMalloc(&nre_buf, nq * sizeof(u64)));
        HIPCHK(h, hipMemsetAsync(d_nre, 0, sizeof(u64), h->stream));
        k_slow_len<<<(unsigned)nq, TPB, 0, h->stream>>>(d_rid, n, d_q, nq, d_len);
        std::vector<u64> len(nq), off(nq);
        HIPCHK(h, hipMemcpyAsync(len.data(), d_len, nq * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        u64 tot = 0;
        for (u64 i = 0; i < nq; ++i) { off[i] = tot; tot += 2 * len[i]; }
        u32 *sk = nullptr, *sm = nullptr;
        HIPCHK(h, hipMalloc(&sk, std::max<u64>(tot, 1) * sizeof(u32)));
        HIPCHK(h, hipMalloc(&sm, std::max<u64>(tot, 1) * sizeof(u32)));
        HIPCHK(h, hipMemsetAsync(sk, 0, tot * sizeof(u32), h->stream));
        HIPCHK(h, hipMemsetAsync(sm, 0, tot * sizeof(u32), h->stream));
        HIPCHK(h, hipMemcpyAsync(d_off, off.data(), nq * sizeof(u64), hipMemcpyHostToDevice, h->stream));
        SlowArgs a{d_rid, d_loc, d_hf, d_q, d_len, d_off, sk, sm, h->cfg.n_loci, h->cfg.n_haplotypes,
                   h->table, h->cap - 1, h->arena, h->arena_cap, h->ctr, h->read_slot, nre_buf, d_nre};
        k_slow<<<(unsigned)nq, TPB, 0, h->stream>>>(a);
        u64 nre = 0;
        HIPCHK(h, hipMemcpyAsync(&nre, d_nre, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
        rc = sync_counters(h);
        hipFree(d_len); hipFree(d_off); hipFree(d_nre); hipFree(sk); hipFree(sm);
        if (d_requeue) hipFree(d_requeue);
        d_requeue = nre_buf; d_q = nre_buf; nq = nre;
        if (rc != ECB_OK) break;
        if (nq) { rc = grow_table(h, h->cap * 4); if (rc != ECB_OK) break; }
    }
    if (d_requeue) hipFree(d_requeue);
    return rc;
}

int excl_scan(ecb_handle* h, const u32* in, u64 n, u32* out, u32* total);

// one batch of whole reads, device-resident
int process_batch(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, const int* d_pos, u64 n) {
    if (n == 0) return ECB_OK;
    u32 last_rid = 0;
    HIPCHK(h, hipMemcpyAsync(&last_rid, d_rid + (n - 1), sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const u64 reads_after = (u64)(u32)(last_rid + 1u);
    if (reads_after < h->n_reads) return fail(h, ECB_ERR_CONTRACT, "read_id went backwards across pushes");
    int rc = ensure_read_slot(h, reads_after);
    if (rc != ECB_OK) return rc;
    h->reads_hi = reads_after;
    // keep the table at most half full before a batch (it grows again, via k_slow, if a batch overfills it)
    while (h->n_ecs() * 2 > h->cap) { rc = grow_table(h, h->cap * 4); if (rc != ECB_OK) return rc; }
    // The stream is cut into ECB_ROUNDS x as many slices as waves are resident at once; the launch holds the resident
    // waves only, which claim slice after slice (`waves` below counts slices).
    if (!h->resident_blocks) {                         // (asked once per handle: two runtime queries per batch add up on a streamed BAM)
        int cus = 256, bpc = 4;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_stream<false>, TPB, 0);
        h->resident_blocks = (u64)std::max(cus, 1) * std::max(bpc, 1);
        h->rounds = getenv("ECB_ROUNDS") ? std::max(1, atoi(getenv("ECB_ROUNDS"))) : 32;
    }
    const u64 rounds = h->rounds, resident_blocks = h->resident_blocks;
    u64 waves = resident_blocks * NWAVE * rounds;
    waves = std::min<u64>(waves, (n + 2 * WT - 1) / (2 * WT));
    waves = std::max<u64>(waves, 1);
    u64 chunk = (n + waves - 1) / waves;
    chunk = (chunk + 3) & ~(u64)3;
    waves = (n + chunk - 1) / chunk;
    const u64 blocks = std::min<u64>((waves + NWAVE - 1) / NWAVE, resident_blocks);
    const u64 pwaves = blocks * NWAVE;           // waves of the launch
    // a parked launch defers at most the reads of the tiles in flight
    const u64 need_q = waves * (u64)(WMAXR + 1) + 16;
    if (h->queue_cap < need_q) {
        if (h->queue) hipFree(h->queue);
        h->queue_cap = need_q;
        HIPCHK(h, hipMalloc(&h->queue, h->queue_cap * sizeof(u64)));
    }
    u64* d_resume = nullptr;
    POOL(h, P_RESUME, d_resume, 2 * waves);
    k_init_resume<<<nblk(waves, TPB), TPB, 0, h->stream>>>(d_resume, waves, chunk);
    if (h->rng_min)
        k_ranges<<<(unsigned)std::min<u64>(4096, (n + TPB - 1) / TPB), TPB, 0, h->stream>>>(
            d_loc, d_hf, d_pos, n, h->cfg.n_loci, h->cfg.n_haplotypes, h->rng_min, h->rng_max);
    u32* d_wcounts = nullptr;
    POOL(h, P_WCOUNTS, d_wcounts, 3 * pwaves);
    if (h->wave_arena_n < pwaves) {                 // (only ever grows to the resident wave count; zero = nothing reserved)
        u64* wa = nullptr;
        HIPCHK(h, hipMalloc(&wa, 2 * pwaves * sizeof(u64)));
        HIPCHK(h, hipMemsetAsync(wa, 0, 2 * pwaves * sizeof(u64), h->stream));
        if (h->wave_arena) {
            HIPCHK(h, hipMemcpyAsync(wa, h->wave_arena, 2 * h->wave_arena_n * sizeof(u64), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            hipFree(h->wave_arena);
        }
        h->wave_arena = wa; h->wave_arena_n = pwaves;
    }
    StreamArgs a{d_rid, d_loc, d_hf, n, chunk, h->prev_rid, h->cfg.n_loci, h->cfg.n_haplotypes,
                 h->table, h->cap - 1, h->arena, h->arena_cap, h->ctr, h->read_slot, h->queue, h->queue_cap, d_resume, d_wcounts, h->wave_arena,
                 getenv("ECB_ABLATE") ? (u32)atoi(getenv("ECB_ABLATE")) : 0u, nullptr};
#ifdef ECB_TIMING
    HIPCHK(h, hipMalloc(&a.timing, 8 * sizeof(u64)));
    HIPCHK(h, hipMemset(a.timing, 0, 8 * sizeof(u64)));
#endif
    for (;;) {
        HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, sizeof(u64), h->stream));   // per launch
        HIPCHK(h, hipMemsetAsync(&h->ctr->full, 0, sizeof(u32), h->stream));
        HIPCHK(h, hipMemsetAsync(&h->ctr->next_slice, 0, sizeof(u64), h->stream));
        a.table = h->table; a.cap_mask = h->cap - 1;
        HIPCHK(h, hipMemsetAsync(d_wcounts, 0, 3 * pwaves * sizeof(u32), h->stream));
        if (h->prof) hipEventRecord(h->ev0, h->stream);
        k_stream<false><<<(unsigned)blocks, TPB, 0, h->stream>>>(a);
        if (h->prof) hipEventRecord(h->ev1, h->stream);
        k_sum_counts<<<1, 1024, 0, h->stream>>>(d_wcounts, pwaves, h->ctr);
        HIPCHK(h, hipGetLastError());
        rc = sync_counters(h);
        if (h->prof) {
            float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1);
            h->prof_ms += ms; h->prof_launches += 1;
        }
        if (rc != ECB_OK) break;
        const bool parked = h->hctr.full != 0;
        if (h->hctr.n_queue) {
            rc = run_slow(h, d_rid, d_loc, d_hf, n, h->queue, std::min<u64>(h->hctr.n_queue, h->queue_cap));
            if (rc != ECB_OK) break;
        }
        if (!parked) break;
        rc = grow_table(h, h->cap * 4);                 // some workgroups stopped early: more room, then resume
        if (rc != ECB_OK) break;
    }
#ifdef ECB_TIMING
    {
        u64 t[8];
        hipMemcpy(t, a.timing, sizeof(t), hipMemcpyDeviceToHost); hipFree(a.timing);
        static const char* nm[8] = {"clear+filter+heads", "seg/geometry", "prefetch issue", "(b) LDS tables", "hash entries", "(c) lookup", "claim", "-"};
        u64 tot = 0; for (int i = 0; i < 7; ++i) tot += t[i];
        fprintf(stderr, "[ecb timing] %llu waves, clocks per wave:", (unsigned long long)waves);
        for (int i = 0; i < 7; ++i) fprintf(stderr, "  %s %.0f (%.1f%%)", nm[i], (double)t[i] / waves, 100.0 * t[i] / std::max<u64>(tot, 1));
        fprintf(stderr, "\n");
    }
#endif
    if (rc != ECB_OK) return rc;
    if (h->prof) h->prof_records += n;
    h->prev_rid = last_rid;
    h->n_reads = reads_after;
    return ECB_OK;
}

int ensure_staging(ecb_handle* h, u64 need) {
    if (need <= h->st_cap) return ECB_OK;
    if (h->st_rid) { hipFree(h->st_rid); hipFree(h->st_loc); hipFree(h->st_hf); if (h->st_pos) hipFree(h->st_pos); }
    h->st_cap = std::max<u64>(need, h->cfg.max_batch_records);
    HIPCHK(h, hipMalloc(&h->st_rid, h->st_cap * sizeof(u32)));
    HIPCHK(h, hipMalloc(&h->st_loc, h->st_cap * sizeof(u32)));
    HIPCHK(h, hipMalloc(&h->st_hf, h->st_cap * sizeof(u32)));
    if (h->cfg.flags & ECB_F_RANGES) HIPCHK(h, hipMalloc(&h->st_pos, h->st_cap * sizeof(int)));
    return ECB_OK;
}

// send carry[0..nc) ++ src[0..m) as one batch
int stage_and_process(ecb_handle* h, const u32* rid, const u32* loc, const u32* hf, const int* pos, u64 m) {
    const u64 nc = h->c_rid.size();
    const u64 n = nc + m;
    if (!n) return ECB_OK;
    int rc = ensure_staging(h, n);
    if (rc != ECB_OK) return rc;
    const bool rg = (h->cfg.flags & ECB_F_RANGES) != 0;
    if (nc) {
        HIPCHK(h, hipMemcpyAsync(h->st_rid, h->c_rid.data(), nc * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->st_loc, h->c_loc.data(), nc * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->st_hf, h->c_hf.data(), nc * 4, hipMemcpyHostToDevice, h->stream));
        if (rg) HIPCHK(h, hipMemcpyAsync(h->st_pos, h->c_pos.data(), nc * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (m) {
        HIPCHK(h, hipMemcpyAsync(h->st_rid + nc, rid, m * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->st_loc + nc, loc, m * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->st_hf + nc, hf, m * 4, hipMemcpyHostToDevice, h->stream));
        if (rg) HIPCHK(h, hipMemcpyAsync(h->st_pos + nc, pos, m * 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));   // the carry vectors may be rewritten by the caller next
    h->c_rid.clear(); h->c_loc.clear(); h->c_hf.clear(); h->c_pos.clear();
    return process_batch(h, h->st_rid, h->st_loc, h->st_hf, rg ? h->st_pos : nullptr, n);
}

void free_results(ecb_handle* h) {   // result buffers live in the pool: nothing to free, just forget them
    h->list = h->order = h->rank_of_slot = h->indptr = nullptr;
    h->indices = h->data = h->counts = nullptr;
}

int excl_scan(ecb_handle* h, const u32* in, u64 n, u32* out, u32* total) {
    const u64 nb = std::max<u64>(1, (n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    u32* sums = nullptr;
    POOL(h, P_SUMS, sums, nb + 1);
    k_scan_sums<<<(unsigned)nb, TPB, 0, h->stream>>>(in, n, sums);
    k_scan_top<<<1, TPB, 0, h->stream>>>(sums, nb, sums + nb);
    k_scan_apply<<<(unsigned)nb, TPB, 0, h->stream>>>(in, n, sums, out);
    HIPCHK(h, hipMemcpyAsync(total, sums + nb, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

// reads per EC / first appearance, from read_slot[0, n_reads) (once, when the stream is closed)
int ensure_counts(ecb_handle* h) {
    if (h->counted) return ECB_OK;
    const u64 R = h->n_reads;
    if (R) {
        const u32 nb = (u32)std::max<u64>(1, h->cap >> BIN_BITS);
        if (nb > MAX_BUCKETS) return fail(h, ECB_ERR_LIMIT, "EC table larger than 2^26 slots is not supported yet");
        const u32 G = (u32)std::min<u64>(PART_G, (R + 4095) / 4096);
        u32 *hist = nullptr, *offs = nullptr;
        uint2* pairs = nullptr;
        POOL(h, P_HIST, hist, (u64)nb * G); POOL(h, P_OFFS, offs, (u64)nb * G);
        POOL(h, P_PAIRS, pairs, R);
        k_part_hist<<<G, TPB_PART, nb * 4, h->stream>>>(h->read_slot, R, nb, hist);
        u32 total = 0;
        int rc = excl_scan(h, hist, (u64)nb * G, offs, &total);
        if (rc == ECB_OK) {
            if (nb <= STAGE_MAX_BUCKETS) {
                if (!h->scatter_attr_set) {             // (per handle = per device: more than 64 KB of dynamic LDS has to be asked for)
                    HIPCHK(h, hipFuncSetAttribute((const void*)k_part_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  3 * STAGE_MAX_BUCKETS * 4 + 8 + STAGE * 8));
                    h->scatter_attr_set = true;
                }
                k_part_scatter_staged<<<G, TPB_PART, 3 * nb * 4 + 8 + STAGE * 8, h->stream>>>(h->read_slot, R, nb, offs, pairs);
            } else                                     // (tables beyond 2^25 slots: the counters would crowd the stage out of LDS)
                k_part_scatter<<<G, TPB_PART, nb * 4, h->stream>>>(h->read_slot, R, nb, offs, pairs);
            // work list of k_count_bins: ranges far above the average are cut into pieces (see CountWork)
            u32* d_starts = nullptr;
            POOL(h, P_STARTS, d_starts, (u64)nb + 1);
            k_bucket_starts<<<nblk((u64)nb + 1, TPB), TPB, 0, h->stream>>>(offs, G, nb, total, d_starts);
            std::vector<u32> starts(nb + 1);
            HIPCHK(h, hipMemcpyAsync(starts.data(), d_starts, ((u64)nb + 1) * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            const u32 piece = std::max<u32>(32768u, 2u * (u32)((total + nb - 1) / nb));
            std::vector<CountWork> work;
            work.reserve(nb + total / piece + 1);
            for (u32 b = 0; b < nb; ++b) {
                const u32 s0 = starts[b], s1 = starts[b + 1];
                if (s1 == s0) continue;
                if (s1 - s0 <= piece + piece / 2) { work.push_back(CountWork{b, s0, s1, 0u}); continue; }
                for (u32 a = s0; a < s1; a += piece) work.push_back(CountWork{b, a, std::min(a + piece, s1), 1u});
            }
            CountWork* d_work = nullptr;
            POOL(h, P_WORK, d_work, work.size());
            HIPCHK(h, hipMemcpyAsync(d_work, work.data(), work.size() * sizeof(CountWork), hipMemcpyHostToDevice, h->stream));
            if (!work.empty()) k_count_bins<<<(unsigned)work.size(), TPB_COUNT, 0, h->stream>>>(pairs, d_work, h->table);
            hipError_t e = hipStreamSynchronize(h->stream);               // (also: `work` lives on this stack frame)
            if (e != hipSuccess) rc = fail(h, ECB_ERR_HIP, "k_count: %s", hipGetErrorString(e));
        }
        if (rc != ECB_OK) return rc;
    }
    h->counted = true;
    return ECB_OK;
}

int compact_table(ecb_handle* h, u32* bitmap = nullptr, u64 n_bits = 0, uint2* list_fn = nullptr) {
    u64* d_n = nullptr;
    POOL(h, P_CNT, d_n, 1);
    HIPCHK(h, hipMemsetAsync(d_n, 0, sizeof(u64), h->stream));
    POOL(h, P_LIST, h->list, h->n_ecs());
    k_compact<<<(unsigned)std::min<u64>(2048, (h->cap + 4 * TPB_COMPACT - 1) / (4 * TPB_COMPACT)), TPB_COMPACT, 0, h->stream>>>(h->table, h->cap, h->list, std::max<u64>(h->n_ecs(), 1), d_n, bitmap, n_bits, list_fn);
    HIPCHK(h, hipMemcpyAsync(&h->n_list, d_n, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->n_list != h->n_ecs()) return fail(h, ECB_ERR_HIP, "internal: %llu occupied slots but %llu ECs created",
                                             (unsigned long long)h->n_list, (unsigned long long)h->n_ecs());
    return ECB_OK;
}

// distinct (EC, cell, file) triples of this handle's reads: sort the per-read keys (EC id << 32 | meta), run-length encode.
// ec_of_slot maps a table slot to the EC id to use (the handle's own ranks, or global ranks in a multi-GPU run).
int ms_reduce(ecb_handle* h, const u32* ec_of_slot) {
    const u64 R = h->n_reads;
    if (h->meta_hi < R) return fail(h, ECB_ERR_STATE, "ecb_push_cells covered %llu of %llu reads",
                                    (unsigned long long)h->meta_hi, (unsigned long long)R);
    u64 *keys = nullptr, *keys2 = nullptr; u32 *vals = nullptr, *vals2 = nullptr, *flag = nullptr, *pos = nullptr;
    POOL(h, P_MS_KEYS, keys, R); POOL(h, P_MS_KEYS2, keys2, R); POOL(h, P_MS_VALS, vals, R); POOL(h, P_MS_VALS2, vals2, R);
    POOL(h, P_MS_FLAG, flag, R); POOL(h, P_MS_POS, pos, R);
    k_ms_keys<<<nblk(R, TPB), TPB, 0, h->stream>>>(h->read_slot, ec_of_slot, h->meta, R, keys, vals);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, R, 0, 64, h->stream) != hipSuccess)
        return fail(h, ECB_ERR_HIP, "rocprim::radix_sort_pairs (size query)");
    char* tmp = nullptr;
    POOL(h, P_MS_TMP, tmp, tmp_bytes);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, R, 0, 64, h->stream) != hipSuccess)
        return fail(h, ECB_ERR_HIP, "rocprim::radix_sort_pairs");
    k_ms_heads<<<nblk(R, TPB), TPB, 0, h->stream>>>(keys2, R, flag);
    u32 nt = 0;
    int rc = excl_scan(h, flag, R, pos, &nt);
    if (rc != ECB_OK) return rc;
    POOL(h, P_MS_OKEY, h->ms_okey, nt); POOL(h, P_MS_OFIRST, h->ms_ofirst, nt); POOL(h, P_MS_OSTART, h->ms_ostart, (u64)nt + 1);
    k_ms_emit<<<nblk(R, TPB), TPB, 0, h->stream>>>(keys2, vals2, flag, pos, R, nt, h->ms_okey, h->ms_ofirst, h->ms_ostart);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_triples = nt; h->ms_ocount = nullptr;
    return ECB_OK;
}

}  // namespace

extern "C" {

int ecb_abi_version(void) { return ECB_ABI_VERSION; }

int ecb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ecb_last_error(const ecb_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int ecb_create(const ecb_config* cfg, ecb_handle** out) {
    if (!cfg || !out || cfg->struct_size != sizeof(ecb_config)) return fail(nullptr, ECB_ERR_ARG, "bad ecb_config (struct_size)");
    if (cfg->n_loci == 0 || cfg->n_loci >= MAX_LOCI) return fail(nullptr, ECB_ERR_ARG, "n_loci out of range (1 .. 2^27-1)");
    if (cfg->n_haplotypes == 0 || cfg->n_haplotypes > 31) return fail(nullptr, ECB_ERR_ARG, "n_haplotypes must be 1..31 (A stores a bitmask in int32)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, ECB_ERR_NO_DEVICE, "no HIP device: libecb has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, ECB_ERR_ARG, "device %d out of range (%d present)", cfg->device, ndev);
    ecb_handle* h = new ecb_handle();
    h->cfg = *cfg;
    h->device = cfg->device;
    if (!h->cfg.ec_capacity) h->cfg.ec_capacity = 1ull << 22;
    if (!h->cfg.arena_capacity) h->cfg.arena_capacity = 1ull << 26;
    if (!h->cfg.max_batch_records) h->cfg.max_batch_records = 1ull << 24;
    h->cap = std::max<u64>(next_pow2(h->cfg.ec_capacity), 1024);
    h->arena_cap = std::min<u64>(h->cfg.arena_capacity, 1ull << 32);
    auto bail = [&](int code, const char* what, hipError_t e) {
        fail(nullptr, code, "%s: %s", what, hipGetErrorString(e));
        ecb_destroy(h);
        return code;
    };
    hipError_t e;
    if ((e = hipSetDevice(h->device)) != hipSuccess) return bail(ECB_ERR_HIP, "hipSetDevice", e);
    if ((e = hipStreamCreate(&h->stream)) != hipSuccess) return bail(ECB_ERR_HIP, "hipStreamCreate", e);
    if ((e = hipMalloc(&h->table, h->cap * sizeof(Slot))) != hipSuccess) return bail(ECB_ERR_HIP, "hipMalloc(table)", e);
    if ((e = hipMalloc(&h->arena, h->arena_cap * sizeof(uint2))) != hipSuccess) return bail(ECB_ERR_HIP, "hipMalloc(arena)", e);
    if ((e = hipMalloc(&h->ctr, sizeof(Counters))) != hipSuccess) return bail(ECB_ERR_HIP, "hipMalloc(counters)", e);
    hipMemsetAsync(h->table, 0, h->cap * sizeof(Slot), h->stream);
    clear_counters(h);
    if (cfg->flags & ECB_F_RANGES) {
        const u64 ns = (u64)cfg->n_loci * cfg->n_haplotypes;
        if ((e = hipMalloc(&h->rng_min, ns * sizeof(int))) != hipSuccess) return bail(ECB_ERR_HIP, "hipMalloc(ranges)", e);
        if ((e = hipMalloc(&h->rng_max, ns * sizeof(int))) != hipSuccess) return bail(ECB_ERR_HIP, "hipMalloc(ranges)", e);
        k_fill_i32<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng_min, ns, INT_MAX);
        k_fill_i32<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng_max, ns, INT_MIN);
    }
    hipEventCreate(&h->ev0); hipEventCreate(&h->ev1);
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail(ECB_ERR_HIP, "init", e);
    *out = h;
    return ECB_OK;
}

void ecb_destroy(ecb_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    free_results(h);
    hipFree(h->table); hipFree(h->arena); hipFree(h->ctr); hipFree(h->read_slot); hipFree(h->meta);
    hipFree(h->rng_min); hipFree(h->rng_max); hipFree(h->queue); hipFree(h->wave_arena);
    for (int i = 0; i < ecb_handle::P_N; ++i) hipFree(h->pool[i]);
    hipFree(h->st_rid); hipFree(h->st_loc); hipFree(h->st_hf); hipFree(h->st_pos);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int ecb_verify_device(ecb_handle* h, const void* d_read_id, const void* d_locus, const void* d_hapflag, size_t n,
                      uint64_t* n_mismatch, uint64_t* n_skipped) {
    if (!h || !n_mismatch || !n_skipped) return ECB_ERR_ARG;
    if (!n || !d_read_id || !d_locus || !d_hapflag) return fail(h, ECB_ERR_ARG, "null tuple stream");
    if (((uintptr_t)d_read_id | (uintptr_t)d_locus | (uintptr_t)d_hapflag) & 15) return fail(h, ECB_ERR_ARG, "device streams must be 16-byte aligned");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    const Counters before = h->hctr;
    int cus = 256, bpc = 4;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_stream<false>, TPB, 0);
    u64 waves = std::min<u64>((u64)cus * std::max(bpc, 1) * NWAVE * 4, (n + 2 * WT - 1) / (2 * WT));
    waves = std::max<u64>(waves, 1);
    u64 chunk = ((n + waves - 1) / waves + 3) & ~(u64)3;
    waves = (n + chunk - 1) / chunk;
    const u64 blocks = (waves + NWAVE - 1) / NWAVE;
    u64* d_resume = nullptr; u32* d_wcounts = nullptr;
    POOL(h, P_RESUME, d_resume, 2 * waves); POOL(h, P_WCOUNTS, d_wcounts, 3 * blocks * NWAVE);
    k_init_resume<<<nblk(waves, TPB), TPB, 0, h->stream>>>(d_resume, waves, chunk);
    HIPCHK(h, hipMemsetAsync(d_wcounts, 0, 3 * blocks * NWAVE * sizeof(u32), h->stream));
    HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, sizeof(u64), h->stream));
    const u64 need_q = waves * (u64)(WMAXR + 1) + 16;
    if (h->queue_cap < need_q) {
        if (h->queue) hipFree(h->queue);
        h->queue_cap = need_q;
        HIPCHK(h, hipMalloc(&h->queue, h->queue_cap * sizeof(u64)));
    }
    StreamArgs a{(const u32*)d_read_id, (const u32*)d_locus, (const u32*)d_hapflag, n, chunk, 0xFFFFFFFFu,
                 h->cfg.n_loci, h->cfg.n_haplotypes, h->table, h->cap - 1, h->arena, h->arena_cap, h->ctr,
                 h->read_slot, h->queue, h->queue_cap, d_resume, d_wcounts, nullptr, 0u};
    HIPCHK(h, hipMemsetAsync(&h->ctr->next_slice, 0, sizeof(u64), h->stream));
    k_stream<true><<<(unsigned)blocks, TPB, 0, h->stream>>>(a);
    k_sum_counts<<<1, 1024, 0, h->stream>>>(d_wcounts, blocks * NWAVE, h->ctr);
    rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    *n_mismatch = h->hctr.n_ecs - before.n_ecs;
    *n_skipped = h->hctr.n_queue;
    // the pass re-counted the records: put the stream's own counters back
    Counters fix = h->hctr;
    fix.all = before.all; fix.valid = before.valid; fix.n_ecs = before.n_ecs; fix.n_queue = 0;
    h->hctr = fix;
    HIPCHK(h, hipMemcpyAsync(h->ctr, &h->hctr, sizeof(Counters), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_reset(ecb_handle* h) {
    if (!h) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    free_results(h);
#if defined(ECB_TIMING) || defined(ECB_EXPERIMENTS)     // experiment builds only (tools/exp_hits.py: a pass over a table that holds every EC already)
    if (!getenv("ECB_KEEP_TABLE"))
#endif
    HIPCHK(h, hipMemsetAsync(h->table, 0, h->cap * sizeof(Slot), h->stream));
    { int rc_ = clear_counters(h); if (rc_ != ECB_OK) return rc_; }
    if (h->wave_arena) HIPCHK(h, hipMemsetAsync(h->wave_arena, 0, 2 * h->wave_arena_n * sizeof(u64), h->stream));
    if (h->read_slot && h->reads_hi) HIPCHK(h, hipMemsetAsync(h->read_slot, 0xFF, h->reads_hi * sizeof(u32), h->stream));
    if (h->rng_min) {
        const u64 ns = (u64)h->cfg.n_loci * h->cfg.n_haplotypes;
        k_fill_i32<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng_min, ns, INT_MAX);
        k_fill_i32<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng_max, ns, INT_MIN);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->prev_rid = 0xFFFFFFFFu; h->n_reads = 0; h->reads_hi = 0; h->meta_hi = 0; h->n_triples = 0; h->ms_ocount = nullptr; h->ms_adopted = false;
    h->extra_all = h->extra_valid = h->extra_reads = 0;
    h->c_rid.clear(); h->c_loc.clear(); h->c_hf.clear(); h->c_pos.clear();
    h->finalized = false; h->counted = false; h->adopted = false; h->sizes = ecb_sizes{}; h->n_list = 0;
    return ECB_OK;
}

int ecb_push_device(ecb_handle* h, const void* d_read_id, const void* d_locus, const void* d_hapflag,
                    const void* d_pos, size_t n) {
    if (!h) return ECB_ERR_ARG;
    if (h->finalized || h->counted) return fail(h, ECB_ERR_STATE, "push after finalize / table export");
    if (n && (!d_read_id || !d_locus || !d_hapflag)) return fail(h, ECB_ERR_ARG, "null tuple stream");
    if ((h->cfg.flags & ECB_F_RANGES) && n && !d_pos) return fail(h, ECB_ERR_ARG, "ECB_F_RANGES needs pos");
    if (((uintptr_t)d_read_id | (uintptr_t)d_locus | (uintptr_t)d_hapflag) & 15) return fail(h, ECB_ERR_ARG, "device streams must be 16-byte aligned");
    if (!h->c_rid.empty()) return fail(h, ECB_ERR_STATE, "ecb_push_device while a host push has an open read");
    HIPCHK(h, hipSetDevice(h->device));
    return process_batch(h, (const u32*)d_read_id, (const u32*)d_locus, (const u32*)d_hapflag, (const int*)d_pos, n);
}

int ecb_push(ecb_handle* h, const uint32_t* rid, const uint32_t* loc, const uint32_t* hf, const int32_t* pos, size_t n) {
    if (!h) return ECB_ERR_ARG;
    if (h->finalized || h->counted) return fail(h, ECB_ERR_STATE, "push after finalize / table export");
    if (n && (!rid || !loc || !hf)) return fail(h, ECB_ERR_ARG, "null tuple stream");
    const bool rg = (h->cfg.flags & ECB_F_RANGES) != 0;
    if (rg && n && !pos) return fail(h, ECB_ERR_ARG, "ECB_F_RANGES needs pos");
    HIPCHK(h, hipSetDevice(h->device));
    u64 done = 0;
    while (done < n) {
        const u64 m = std::min<u64>(h->cfg.max_batch_records, n - done);
        const u32 *r = rid + done, *l = loc + done, *f = hf + done;
        const int* ps = rg ? pos + done : nullptr;
        // records [cut, m) share the window's last read_id: that read may continue in the next push
        const u32 last = r[m - 1];
        u64 cut = m;
        while (cut > 0 && r[cut - 1] == last) --cut;
        const bool carry_continues = !h->c_rid.empty() && h->c_rid.back() == last;
        if (cut == 0 && (carry_continues || h->c_rid.empty())) {
            h->c_rid.insert(h->c_rid.end(), r, r + m);
            h->c_loc.insert(h->c_loc.end(), l, l + m);
            h->c_hf.insert(h->c_hf.end(), f, f + m);
            if (rg) h->c_pos.insert(h->c_pos.end(), ps, ps + m);
        } else {
            int rc = stage_and_process(h, r, l, f, ps, cut);      // carry ++ window[0, cut): whole reads
            if (rc != ECB_OK) return rc;
            h->c_rid.assign(r + cut, r + m);
            h->c_loc.assign(l + cut, l + m);
            h->c_hf.assign(f + cut, f + m);
            if (rg) h->c_pos.assign(ps + cut, ps + m);
        }
        done += m;
    }
    return ECB_OK;
}

int ecb_push_cells(ecb_handle* h, const uint32_t* meta, uint64_t first_read, size_t n) {
    if (!h) return ECB_ERR_ARG;
    if (!(h->cfg.flags & ECB_F_MULTISAMPLE)) return fail(h, ECB_ERR_STATE, "handle was created without ECB_F_MULTISAMPLE");
    if (h->finalized) return fail(h, ECB_ERR_STATE, "push after finalize");
    if (!n) return ECB_OK;
    if (!meta) return fail(h, ECB_ERR_ARG, "null meta");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 need = first_read + n;
    if (need > h->meta_cap) {
        const u64 nc = std::max<u64>(need, h->meta_cap * 2);
        u32* p = nullptr;
        HIPCHK(h, hipMalloc(&p, nc * sizeof(u32)));
        if (h->meta) {
            HIPCHK(h, hipMemcpyAsync(p, h->meta, h->meta_hi * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipFree(h->meta));
        }
        h->meta = p; h->meta_cap = nc;
    }
    HIPCHK(h, hipMemcpyAsync(h->meta + first_read, meta, n * sizeof(u32), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->meta_hi = std::max<u64>(h->meta_hi, need);
    return ECB_OK;
}

int ecb_finalize(ecb_handle* h, ecb_sizes* out) {
    if (!h || !out) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->finalized) {
        if (!h->c_rid.empty()) {                     // the stream ends here: the carried read is complete
            int rc = stage_and_process(h, nullptr, nullptr, nullptr, nullptr, 0);
            if (rc != ECB_OK) return rc;
        }
        int rc = sync_counters(h);
        if (rc != ECB_OK) return rc;
        rc = ensure_counts(h);
        if (rc != ECB_OK) return rc;
    }
    const u64 E = h->n_ecs();
    const u64 valid = h->hctr.valid + h->extra_valid;
    if (E == 0 || valid == 0) return fail(h, ECB_ERR_EMPTY, "no valid alignments: nothing to build (the reference fails here too)");
    if (E >= (1ull << 31) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^31-2 equivalence classes");
    free_results(h);
    // rank by first appearance: bitmap over read indices (marked while the table is compacted), popcount prefix
    const u64 total_reads = h->n_reads + h->extra_reads;
    const u64 words = (total_reads + 31) / 32 + 1;
    u32 *bitmap = nullptr, *wpop = nullptr, *wprefix = nullptr, *rowlen = nullptr;
    uint2* list_fn = nullptr;
    POOL(h, P_BITMAP, bitmap, words); POOL(h, P_WPOP, wpop, words); POOL(h, P_WPREFIX, wprefix, words);
    POOL(h, P_ROWLEN, rowlen, E); POOL(h, P_LISTFN, list_fn, E);
    POOL(h, P_ORDER, h->order, E); POOL(h, P_RANK, h->rank_of_slot, h->cap);
    POOL(h, P_INDPTR, h->indptr, E + 1); POOL(h, P_COUNTS, h->counts, E);
    HIPCHK(h, hipMemsetAsync(bitmap, 0, words * 4, h->stream));
    int rc = compact_table(h, bitmap, total_reads, list_fn);
    if (rc != ECB_OK) return rc;
    k_popc<<<nblk(words, TPB), TPB, 0, h->stream>>>(bitmap, words, wpop);
    u32 tot = 0;
    rc = excl_scan(h, wpop, words, wprefix, &tot);
    if (rc != ECB_OK) return rc;
    if (tot != E) return fail(h, ECB_ERR_HIP, "internal: %u distinct first-appearance indices for %llu ECs", tot, (unsigned long long)E);
    k_rank<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->list, list_fn, E, total_reads, bitmap, wprefix, h->order, rowlen, h->rank_of_slot);
    u32 nnz = 0;
    {   // 64-bit check of the row-length total before trusting a 32-bit scan
        rc = excl_scan(h, rowlen, E, h->indptr, &nnz);
        if (rc != ECB_OK) return rc;
        if (arena_used(h) >= (1ull << 31)) return fail(h, ECB_ERR_LIMIT, "A has more than 2^31-1 non-zeros");
        HIPCHK(h, hipMemcpyAsync(h->indptr + E, &nnz, 4, hipMemcpyHostToDevice, h->stream));
    }
    POOL(h, P_INDICES, h->indices, nnz); POOL(h, P_DATA, h->data, nnz);
    {
        u32 *big = nullptr, *d_nbig = nullptr, n_big = 0;
        POOL(h, P_MS_X, big, E + 1);
        d_nbig = big + E;
        HIPCHK(h, hipMemsetAsync(d_nbig, 0, 4, h->stream));
        k_emit_small<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->order, E, h->arena, h->indptr, h->indices, h->data,
                                                           h->counts, h->cfg.n_loci, h->cfg.n_haplotypes, big, d_nbig, h->ctr);
        HIPCHK(h, hipMemcpyAsync(&n_big, d_nbig, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (n_big)
            k_emit_big<<<nblk((u64)n_big * 64, TPB), TPB, 0, h->stream>>>(h->table, h->order, big, n_big, h->arena, h->indptr,
                                                                         h->indices, h->data, h->cfg.n_loci,
                                                                         h->cfg.n_haplotypes, h->ctr);
        rc = sync_counters(h);
        if (rc != ECB_OK) return rc;
    }
    h->sizes.n_ecs = E; h->sizes.nnz_a = nnz; h->sizes.n_samples = 1; h->sizes.nnz_n = E;
    if (h->cfg.flags & ECB_F_MULTISAMPLE) {
        if (h->adopted) {                            // multi-GPU: the triples arrive through ecb_ms_adopt_triples_device
            h->n_triples = 0; h->sizes.n_samples = 0; h->sizes.nnz_n = 0;
        } else {
            if (h->extra_reads) return fail(h, ECB_ERR_STATE, "multisample across GPUs: adopt the merged ECs (ecb_table_adopt_device), then ecb_ms_adopt_triples_device");
            rc = ms_reduce(h, h->rank_of_slot);
            if (rc != ECB_OK) return rc;
            h->sizes.n_samples = 0; h->sizes.nnz_n = h->n_triples;
        }
    }
    h->sizes.all_alignments = h->hctr.all + h->extra_all;
    h->sizes.valid_alignments = valid;
    h->sizes.n_reads = total_reads;
    h->finalized = true;
    *out = h->sizes;
    return ECB_OK;
}

int ecb_export_device(ecb_handle* h, void* ia, void* ja, void* da, void* in_, void* jn, void* dn) {
    if (!h) return ECB_ERR_ARG;
    if (!h->finalized) return fail(h, ECB_ERR_STATE, "export before finalize");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 E = h->sizes.n_ecs, nnz = h->sizes.nnz_a;
    if (ia) HIPCHK(h, hipMemcpyAsync(ia, h->indptr, (E + 1) * 4, hipMemcpyDeviceToDevice, h->stream));
    if (ja) HIPCHK(h, hipMemcpyAsync(ja, h->indices, nnz * 4, hipMemcpyDeviceToDevice, h->stream));
    if (da) HIPCHK(h, hipMemcpyAsync(da, h->data, nnz * 4, hipMemcpyDeviceToDevice, h->stream));
    if (in_) { const int v[2] = {0, (int)E}; HIPCHK(h, hipMemcpyAsync(in_, v, 8, hipMemcpyHostToDevice, h->stream)); }
    if (jn) k_iota<<<nblk(E, TPB), TPB, 0, h->stream>>>((int*)jn, E);
    if (dn) HIPCHK(h, hipMemcpyAsync(dn, h->counts, E * 4, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_export(ecb_handle* h, int32_t* ia, int32_t* ja, int32_t* da, int32_t* in_, int32_t* jn, int32_t* dn) {
    if (!h) return ECB_ERR_ARG;
    if (!h->finalized) return fail(h, ECB_ERR_STATE, "export before finalize");
    if ((h->cfg.flags & ECB_F_MULTISAMPLE) && (in_ || jn || dn))
        return fail(h, ECB_ERR_STATE, "multisample: N comes from ecb_export_pairs");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 E = h->sizes.n_ecs, nnz = h->sizes.nnz_a;
    if (ia) HIPCHK(h, hipMemcpyAsync(ia, h->indptr, (E + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    if (ja) HIPCHK(h, hipMemcpyAsync(ja, h->indices, nnz * 4, hipMemcpyDeviceToHost, h->stream));
    if (da) HIPCHK(h, hipMemcpyAsync(da, h->data, nnz * 4, hipMemcpyDeviceToHost, h->stream));
    if (dn) HIPCHK(h, hipMemcpyAsync(dn, h->counts, E * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (in_) { in_[0] = 0; in_[1] = (int32_t)E; }
    if (jn) for (u64 i = 0; i < E; ++i) jn[i] = (int32_t)i;
    return ECB_OK;
}

int ecb_export_ranges(ecb_handle* h, int64_t* out) {
    if (!h || !out) return ECB_ERR_ARG;
    if (!(h->cfg.flags & ECB_F_RANGES)) return fail(h, ECB_ERR_STATE, "handle was created without ECB_F_RANGES");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 ns = (u64)h->cfg.n_loci * h->cfg.n_haplotypes;
    long long* d = nullptr;
    HIPCHK(h, hipMalloc(&d, ns * 8));
    k_range_len<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng_min, h->rng_max, ns, d);
    HIPCHK(h, hipMemcpyAsync(out, d, ns * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(d));
    return ECB_OK;
}

int ecb_export_range_minmax(ecb_handle* h, int32_t* mn, int32_t* mx) {
    if (!h || !mn || !mx) return ECB_ERR_ARG;
    if (!(h->cfg.flags & ECB_F_RANGES)) return fail(h, ECB_ERR_STATE, "handle was created without ECB_F_RANGES");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 ns = (u64)h->cfg.n_loci * h->cfg.n_haplotypes;
    HIPCHK(h, hipMemcpyAsync(mn, h->rng_min, ns * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(mx, h->rng_max, ns * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_export_pairs(ecb_handle* h, uint32_t* ec, uint32_t* meta, uint32_t* count, uint32_t* first_read) {
    if (!h || !ec || !meta || !count || !first_read) return ECB_ERR_ARG;
    if (!h->finalized || !(h->cfg.flags & ECB_F_MULTISAMPLE)) return fail(h, ECB_ERR_STATE, "no multisample result");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 nt = h->n_triples;
    u32* x = nullptr;
    POOL(h, P_MS_X, x, 3 * nt);
    if (h->adopted && !h->ms_adopted) return fail(h, ECB_ERR_STATE, "multisample across GPUs: no triples adopted yet (ecb_ms_adopt_triples_device)");
    k_ms_split<<<nblk(nt, TPB), TPB, 0, h->stream>>>(h->ms_okey, h->ms_ostart, h->ms_ocount, nt, x, x + nt, x + 2 * nt);
    HIPCHK(h, hipMemcpyAsync(ec, x, nt * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(meta, x + nt, nt * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(count, x + 2 * nt, nt * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(first_read, h->ms_ofirst, nt * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_export_read_ec(ecb_handle* h, int32_t* out) {
    if (!h || !out) return ECB_ERR_ARG;
    if (!h->finalized) return fail(h, ECB_ERR_STATE, "export before finalize");
    if (h->extra_reads) return fail(h, ECB_ERR_STATE, "per-read EC ids are not kept across a multi-GPU merge");
    HIPCHK(h, hipSetDevice(h->device));
    int* d = nullptr;
    HIPCHK(h, hipMalloc(&d, std::max<u64>(h->n_reads, 1) * 4));
    k_read_ec<<<nblk(h->n_reads, TPB), TPB, 0, h->stream>>>(h->read_slot, h->n_reads, h->rank_of_slot, d);
    HIPCHK(h, hipMemcpyAsync(out, d, h->n_reads * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(d));
    return ECB_OK;
}

int ecb_table_sizes(ecb_handle* h, uint64_t* n_entries, uint64_t* n_pairs, uint64_t* n_reads) {
    if (!h) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->c_rid.empty()) {
        int rc = stage_and_process(h, nullptr, nullptr, nullptr, nullptr, 0);
        if (rc != ECB_OK) return rc;
    }
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    rc = ensure_counts(h);
    if (rc != ECB_OK) return rc;
    if (n_entries) *n_entries = h->n_ecs();
    if (n_pairs) *n_pairs = arena_used(h);
    if (n_reads) *n_reads = h->n_reads;
    return ECB_OK;
}

int ecb_table_export_device(ecb_handle* h, void* d_entries, void* d_pairs, uint64_t read_base) {
    uint64_t eo[2], po[2];                             // one part: entries in table order, their keys packed behind each other
    return ecb_table_export_parts_device(h, d_entries, d_pairs, read_base, 1, eo, po);
}

int ecb_table_merge_batch_device(ecb_handle* h, uint32_t n_tables, const void* const* d_entries, const uint64_t* n_entries,
                                const void* const* d_pairs, const uint64_t* n_pairs) {
    if (!h) return ECB_ERR_ARG;
    if (h->finalized) return fail(h, ECB_ERR_STATE, "merge after finalize");
    if (h->adopted) return fail(h, ECB_ERR_STATE, "merge into a table that adopted entries");
    if (n_tables && (!d_entries || !n_entries || !d_pairs || !n_pairs)) return fail(h, ECB_ERR_ARG, "null table lists");
    u64 total = 0;
    for (u32 t = 0; t < n_tables; ++t) {
        if (n_entries[t] && (!d_entries[t] || (n_pairs[t] && !d_pairs[t]))) return fail(h, ECB_ERR_ARG, "null table buffers");
        total += n_entries[t];
    }
    if (!total) return ECB_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    rc = ensure_counts(h);                       // own reads first: merged counts are added on top
    if (rc != ECB_OK) return rc;
    // room for every entry being new: one growth up front, then the merges queue up behind each other with one sync at the end
    while ((h->n_ecs() + total) * 2 > h->cap) { rc = grow_table(h, h->cap * 4); if (rc != ECB_OK) return rc; }
    HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, sizeof(u64), h->stream));
    for (u32 t = 0; t < n_tables; ++t)
        if (n_entries[t])
            k_merge<<<nblk(n_entries[t], MERGE_PER_BLOCK), TPB, 0, h->stream>>>((const Slot*)d_entries[t], n_entries[t], (const uint2*)d_pairs[t],
                                                                              n_pairs[t], h->table, h->cap - 1, h->arena, h->arena_cap, h->ctr);
    rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    if (h->hctr.n_queue) return fail(h, ECB_ERR_TABLE_FULL, "internal: merge found no slot in a half-empty table");
    return ECB_OK;
}

int ecb_table_merge_device(ecb_handle* h, const void* d_entries, uint64_t n_entries, const void* d_pairs, uint64_t n_pairs) {
    return ecb_table_merge_batch_device(h, 1, &d_entries, &n_entries, &d_pairs, &n_pairs);
}

int ecb_table_export_parts_device(ecb_handle* h, void* d_entries, void* d_pairs, uint64_t read_base, uint32_t n_parts,
                                  uint64_t* entry_offsets, uint64_t* pair_offsets) {
    if (!h || !d_entries || !d_pairs || !entry_offsets || !pair_offsets) return ECB_ERR_ARG;
    if (n_parts == 0 || n_parts > MAX_PARTS) return fail(h, ECB_ERR_LIMIT, "1 .. %u parts", MAX_PARTS);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    if (read_base + h->n_reads >= (1ull << 32) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^32-2 reads in total");
    rc = ensure_counts(h);
    if (rc != ECB_OK) return rc;
    rc = compact_table(h);
    if (rc != ECB_OK) return rc;
    const u64 E = h->n_ecs();
    u64* d_cnt = nullptr;                              // [0, 2P): counts, then cursors; [2P, 3P): first pair of every part
    POOL(h, P_PARTS, d_cnt, 3 * (u64)n_parts);
    HIPCHK(h, hipMemsetAsync(d_cnt, 0, 2 * n_parts * sizeof(u64), h->stream));
    if (E) k_parts_count<<<nblk(E, PARTS_PER_BLOCK), TPB, 0, h->stream>>>(h->table, h->list, E, n_parts, d_cnt);
    std::vector<u64> cnt(2 * n_parts), cur(3 * n_parts);
    HIPCHK(h, hipMemcpyAsync(cnt.data(), d_cnt, 2 * n_parts * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    entry_offsets[0] = pair_offsets[0] = 0;
    for (u32 q = 0; q < n_parts; ++q) {
        entry_offsets[q + 1] = entry_offsets[q] + cnt[q];
        pair_offsets[q + 1] = pair_offsets[q] + cnt[n_parts + q];
        cur[q] = entry_offsets[q]; cur[n_parts + q] = pair_offsets[q]; cur[2 * n_parts + q] = pair_offsets[q];
    }
    HIPCHK(h, hipMemcpyAsync(d_cnt, cur.data(), 3 * n_parts * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    if (E) k_parts_export<<<nblk(E, PARTS_PER_BLOCK), TPB, 0, h->stream>>>(h->table, h->list, E, h->arena, n_parts, d_cnt, d_cnt + 2 * n_parts,
                                                               (Slot*)d_entries, (uint2*)d_pairs, (u32)read_base);
    HIPCHK(h, hipStreamSynchronize(h->stream));      // (cur lives on this stack frame)
    return ECB_OK;
}

int ecb_table_adopt_batch_device(ecb_handle* h, uint32_t n_tables, const void* const* d_entries, const uint64_t* n_entries,
                                const void* const* d_pairs, const uint64_t* n_pairs) {
    if (!h) return ECB_ERR_ARG;
    if (h->finalized) return fail(h, ECB_ERR_STATE, "adopt after finalize");
    if (n_tables && (!d_entries || !n_entries || !d_pairs || !n_pairs)) return fail(h, ECB_ERR_ARG, "null table lists");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    if (!h->adopted && (h->n_ecs() || h->n_reads || !h->c_rid.empty()))
        return fail(h, ECB_ERR_STATE, "adopt needs an empty handle (use ecb_table_merge_device to add to a built table)");
    u64 add_e = 0, add_p = 0;
    for (u32 t = 0; t < n_tables; ++t) {
        if (n_entries[t] && (!d_entries[t] || (n_pairs[t] && !d_pairs[t]))) return fail(h, ECB_ERR_ARG, "null table buffers");
        if (n_entries[t]) { add_e += n_entries[t]; add_p += n_pairs[t]; }
    }
    h->adopted = true; h->counted = true;
    if (!add_e) return ECB_OK;
    u64 have = h->n_ecs(), top = h->hctr.arena_top;
    if (top + add_p > h->arena_cap || top + add_p >= (1ull << 32))
        return fail(h, ECB_ERR_TABLE_FULL, "EC key arena exhausted (%llu pairs): raise arena_capacity", (unsigned long long)h->arena_cap);
    if (have + add_e > h->cap) {                        // consecutive slots: a bigger array and a copy, no rehash
        u64 nc = h->cap;
        while (nc < have + add_e) nc *= 2;
        Slot* nt = nullptr;
        HIPCHK(h, hipMalloc(&nt, nc * sizeof(Slot)));
        HIPCHK(h, hipMemsetAsync(nt, 0, nc * sizeof(Slot), h->stream));
        HIPCHK(h, hipMemcpyAsync(nt, h->table, have * sizeof(Slot), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(h->table));
        h->table = nt; h->cap = nc;
    }
    for (u32 t = 0; t < n_tables; ++t) {
        if (!n_entries[t]) continue;
        k_adopt<<<nblk(n_entries[t], TPB), TPB, 0, h->stream>>>((const Slot*)d_entries[t], n_entries[t], h->table, have, (u32)top);
        if (n_pairs[t]) HIPCHK(h, hipMemcpyAsync(h->arena + top, d_pairs[t], n_pairs[t] * sizeof(uint2), hipMemcpyDeviceToDevice, h->stream));
        have += n_entries[t]; top += n_pairs[t];
    }
    h->hctr.n_ecs = have; h->hctr.arena_top = top;
    HIPCHK(h, hipMemcpyAsync(&h->ctr->n_ecs, &h->hctr.n_ecs, sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(&h->ctr->arena_top, &h->hctr.arena_top, sizeof(u64), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_table_adopt_device(ecb_handle* h, const void* d_entries, uint64_t n_entries, const void* d_pairs, uint64_t n_pairs) {
    return ecb_table_adopt_batch_device(h, 1, &d_entries, &n_entries, &d_pairs, &n_pairs);
}

int ecb_export_ec_keys_device(ecb_handle* h, void* d_keys) {
    if (!h || !d_keys) return ECB_ERR_ARG;
    if (!h->finalized) return fail(h, ECB_ERR_STATE, "export before finalize");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 E = h->sizes.n_ecs;
    k_export_keys<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->order, E, (uint4*)d_keys);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_ms_local_triples_device(ecb_handle* h, const void* d_keys, uint64_t n_ecs, uint64_t read_base,
                                void* d_key, void* d_count, void* d_first, uint64_t* n_triples) {
    if (!h || !n_triples) return ECB_ERR_ARG;
    if (!(h->cfg.flags & ECB_F_MULTISAMPLE)) return fail(h, ECB_ERR_STATE, "handle was created without ECB_F_MULTISAMPLE");
    if (h->finalized || h->adopted) return fail(h, ECB_ERR_STATE, "a shard's triples come from the handle its reads were pushed into");
    *n_triples = 0;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->c_rid.empty()) {                         // the stream ends here: the carried read is complete
        int rc0 = stage_and_process(h, nullptr, nullptr, nullptr, nullptr, 0);
        if (rc0 != ECB_OK) return rc0;
    }
    if (!h->n_reads) return ECB_OK;
    if (!d_keys || !d_key || !d_count || !d_first) return fail(h, ECB_ERR_ARG, "null buffers");
    if (read_base + h->n_reads >= (1ull << 32) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^32-2 reads in total");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    u32* grank = nullptr;
    POOL(h, P_MS_GRANK, grank, h->cap);
    HIPCHK(h, hipMemsetAsync(grank, 0xFF, h->cap * sizeof(u32), h->stream));
    k_set_global_rank<<<nblk(n_ecs, TPB), TPB, 0, h->stream>>>((const uint4*)d_keys, n_ecs, h->table, h->cap - 1, grank);
    rc = ms_reduce(h, grank);
    if (rc != ECB_OK) return rc;
    const u64 nt = h->n_triples;
    u64 last = 0;                                     // keys are sorted: an EC id of 0xFFFFFFFF would be the last one
    HIPCHK(h, hipMemcpyAsync(&last, h->ms_okey + (nt - 1), sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((last >> 32) == 0xFFFFFFFFull) return fail(h, ECB_ERR_CONTRACT, "a read's EC is missing from the merged EC list");
    k_ms_out<<<nblk(nt, TPB), TPB, 0, h->stream>>>(h->ms_okey, h->ms_ofirst, h->ms_ostart, nt, (u32)read_base, (u64*)d_key, (u32*)d_count, (u32*)d_first);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n_triples = nt;
    return ECB_OK;
}

int ecb_ms_adopt_triples_device(ecb_handle* h, uint32_t n_tables, const void* const* d_key, const void* const* d_count,
                                const void* const* d_first, const uint64_t* n, uint64_t* n_triples) {
    if (!h) return ECB_ERR_ARG;
    if (!(h->cfg.flags & ECB_F_MULTISAMPLE)) return fail(h, ECB_ERR_STATE, "handle was created without ECB_F_MULTISAMPLE");
    if (!h->finalized || !h->adopted) return fail(h, ECB_ERR_STATE, "triples are adopted by the finalized handle that adopted the merged ECs");
    if (n_tables && (!d_key || !d_count || !d_first || !n)) return fail(h, ECB_ERR_ARG, "null lists");
    HIPCHK(h, hipSetDevice(h->device));
    u64 tot = 0;
    for (u32 t = 0; t < n_tables; ++t) { if (n[t] && (!d_key[t] || !d_count[t] || !d_first[t])) return fail(h, ECB_ERR_ARG, "null buffers"); tot += n[t]; }
    if (tot >= (1ull << 32)) return fail(h, ECB_ERR_LIMIT, "more than 2^32-1 triples");
    u64 *keys = nullptr, *keys2 = nullptr; u32 *vals = nullptr, *vals2 = nullptr, *flag = nullptr, *pos = nullptr, *cin = nullptr, *fin = nullptr;
    POOL(h, P_MS_KEYS, keys, tot); POOL(h, P_MS_KEYS2, keys2, tot); POOL(h, P_MS_VALS, vals, tot); POOL(h, P_MS_VALS2, vals2, tot);
    POOL(h, P_MS_FLAG, flag, tot); POOL(h, P_MS_POS, pos, tot); POOL(h, P_MS_CIN, cin, tot); POOL(h, P_MS_FIN, fin, tot);
    u64 at = 0;
    for (u32 t = 0; t < n_tables; ++t) {
        if (!n[t]) continue;
        HIPCHK(h, hipMemcpyAsync(keys + at, d_key[t], n[t] * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(cin + at, d_count[t], n[t] * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(fin + at, d_first[t], n[t] * 4, hipMemcpyDeviceToDevice, h->stream));
        at += n[t];
    }
    u32 nt = 0;
    if (tot) {
        k_iota<<<nblk(tot, TPB), TPB, 0, h->stream>>>((int*)vals, tot);
        size_t tmp_bytes = 0;
        if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, tot, 0, 64, h->stream) != hipSuccess)
            return fail(h, ECB_ERR_HIP, "rocprim::radix_sort_pairs (size query)");
        char* tmp = nullptr;
        POOL(h, P_MS_TMP, tmp, tmp_bytes);
        if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, tot, 0, 64, h->stream) != hipSuccess)
            return fail(h, ECB_ERR_HIP, "rocprim::radix_sort_pairs");
        k_ms_heads<<<nblk(tot, TPB), TPB, 0, h->stream>>>(keys2, tot, flag);
        int rc = excl_scan(h, flag, tot, pos, &nt);
        if (rc != ECB_OK) return rc;
        POOL(h, P_MS_OKEY, h->ms_okey, nt); POOL(h, P_MS_OFIRST, h->ms_ofirst, nt); POOL(h, P_MS_OCOUNT, h->ms_ocount, nt);
        HIPCHK(h, hipMemsetAsync(h->ms_ocount, 0, (u64)nt * 4, h->stream));
        HIPCHK(h, hipMemsetAsync(h->ms_ofirst, 0xFF, (u64)nt * 4, h->stream));
        k_ms_combine<<<nblk(tot, TPB), TPB, 0, h->stream>>>(keys2, vals2, flag, pos, tot, cin, fin, h->ms_okey, h->ms_ocount, h->ms_ofirst);
        u64 last = 0;
        HIPCHK(h, hipMemcpyAsync(&last, keys2 + (tot - 1), sizeof(u64), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if ((last >> 32) >= h->sizes.n_ecs) return fail(h, ECB_ERR_CONTRACT, "triple with an EC id beyond the merged ECs");
    }
    h->n_triples = nt; h->ms_adopted = true;
    h->sizes.n_samples = 0; h->sizes.nnz_n = nt;
    if (n_triples) *n_triples = nt;
    return ECB_OK;
}

int ecb_counters(ecb_handle* h, uint64_t* all_alignments, uint64_t* valid_alignments, uint64_t* n_reads) {
    if (!h) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->finalized && !h->c_rid.empty()) {
        int rc = stage_and_process(h, nullptr, nullptr, nullptr, nullptr, 0);
        if (rc != ECB_OK) return rc;
    }
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    if (all_alignments) *all_alignments = h->hctr.all + h->extra_all;
    if (valid_alignments) *valid_alignments = h->hctr.valid + h->extra_valid;
    if (n_reads) *n_reads = h->n_reads + h->extra_reads;
    return ECB_OK;
}

int ecb_add_counters(ecb_handle* h, uint64_t all_alignments, uint64_t valid_alignments, uint64_t n_reads) {
    if (!h) return ECB_ERR_ARG;
    h->extra_all += all_alignments; h->extra_valid += valid_alignments; h->extra_reads += n_reads;
    return ECB_OK;
}

int ecb_profile(ecb_handle* h, int enable) {
    if (!h) return ECB_ERR_ARG;
    h->prof = enable != 0; h->prof_ms = 0; h->prof_launches = 0; h->prof_records = 0;
    return ECB_OK;
}

int ecb_profile_read(ecb_handle* h, double* ms, uint64_t* launches, uint64_t* records) {
    if (!h) return ECB_ERR_ARG;
    if (ms) *ms = h->prof_ms;
    if (launches) *launches = h->prof_launches;
    if (records) *records = h->prof_records;
    return ECB_OK;
}

}  // extern "C"

// ---- f-2 conversions (stateless; scratch is allocated per call: this is not the hot path) ------------------------
namespace {
struct Scratch {                       // frees what it allocated
    std::vector<void*> p;
    template <class T> T* get(u64 n) { void* q = nullptr; if (hipMalloc(&q, std::max<u64>(n, 1) * sizeof(T)) != hipSuccess) return nullptr; p.push_back(q); return (T*)q; }
    ~Scratch() { for (void* q : p) hipFree(q); }
};
int cv_scan(hipStream_t st, const u32* in, u64 n, u32* out, u32* total, Scratch& sc) {
    const u64 nb = std::max<u64>(1, (n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    u32* sums = sc.get<u32>(nb + 1);
    if (!sums) return ECB_ERR_HIP;
    k_scan_sums<<<(unsigned)nb, TPB, 0, st>>>(in, n, sums);
    k_scan_top<<<1, TPB, 0, st>>>(sums, nb, sums + nb);
    k_scan_apply<<<(unsigned)nb, TPB, 0, st>>>(in, n, sums, out);
    if (hipMemcpyAsync(total, sums + nb, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return ECB_ERR_HIP;
    return hipStreamSynchronize(st) == hipSuccess ? ECB_OK : ECB_ERR_HIP;
}
}  // namespace

extern "C" int ecb_csr_to_hapcsc_device(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const void* d_indptr,
                                        const void* d_indices, const void* d_data, void* d_cscptr, void* d_cscidx,
                                        uint64_t* total) {
    if (!d_indptr || !d_indices || !d_data || !total || !n_ecs || !n_loci || !n_haps || n_haps > 31) return fail(nullptr, ECB_ERR_ARG, "bad argument");
    if ((u64)n_haps * n_loci >= (1ull << 32)) return fail(nullptr, ECB_ERR_LIMIT, "haplotypes x loci does not fit 32 bits");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, ECB_ERR_NO_DEVICE, "no such device");
    hipStream_t st = nullptr;
    int nnz_i = 0;
    if (hipMemcpy(&nnz_i, (const int*)d_indptr + n_ecs, 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "read nnz");
    const u64 nnz = (u64)nnz_i;
    Scratch sc;
    u32 *cnt = sc.get<u32>(nnz), *pos = sc.get<u32>(nnz);
    if (!cnt || !pos) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    k_cv_popc<<<nblk(nnz, TPB), TPB, 0, st>>>((const int*)d_data, nnz, cnt);
    u32 tot = 0;
    if (cv_scan(st, cnt, nnz, pos, &tot, sc) != ECB_OK) return fail(nullptr, ECB_ERR_HIP, "scan");
    *total = tot;
    if (!d_cscidx || !d_cscptr) return ECB_OK;
    u32 *keys = sc.get<u32>(tot), *vals = sc.get<u32>(tot), *keys2 = sc.get<u32>(tot);
    if (!keys || !vals || !keys2) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    k_cv_expand<<<nblk(nnz, TPB), TPB, 0, st>>>((const int*)d_indptr, n_ecs, (const int*)d_indices, (const int*)d_data, nnz, pos,
                                               n_loci, keys, vals);
    size_t tb = 0;
    unsigned bits = 1; while ((1ull << bits) < (u64)n_haps * n_loci) ++bits;
    rocprim::radix_sort_pairs(nullptr, tb, keys, keys2, vals, (u32*)d_cscidx, tot, 0, bits, st);
    char* tmp = sc.get<char>(tb);
    if (!tmp) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    if (rocprim::radix_sort_pairs(tmp, tb, keys, keys2, vals, (u32*)d_cscidx, tot, 0, bits, st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "sort");
    const u64 nc = (u64)n_haps * (n_loci + 1);
    k_cv_cscptr<<<nblk(nc, TPB), TPB, 0, st>>>(keys2, tot, n_loci, n_haps, (int*)d_cscptr);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csr -> csc");
    return ECB_OK;
}

extern "C" int ecb_hapcsc_to_csr_device(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const void* d_cscptr,
                                        const void* d_cscidx, uint64_t total, void* d_indptr, void* d_indices, void* d_data,
                                        uint64_t* nnz_out) {
    if (!d_cscptr || !d_cscidx || !d_indptr || !d_indices || !d_data || !nnz_out || !n_ecs || !n_loci || !n_haps || n_haps > 31 || !total)
        return fail(nullptr, ECB_ERR_ARG, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, ECB_ERR_NO_DEVICE, "no such device");
    hipStream_t st = nullptr;
    // start of every haplotype's block = running sum of its last column pointer
    std::vector<int> last(n_haps);
    for (u32 h = 0; h < n_haps; ++h)
        if (hipMemcpy(&last[h], (const int*)d_cscptr + (u64)h * (n_loci + 1) + n_loci, 4, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(nullptr, ECB_ERR_HIP, "read csc pointers");
    std::vector<u64> hs(n_haps + 1, 0);
    for (u32 h = 0; h < n_haps; ++h) hs[h + 1] = hs[h] + (u64)last[h];
    if (hs[n_haps] != total) return fail(nullptr, ECB_ERR_ARG, "total does not match the column pointers");
    Scratch sc;
    u64 *d_hs = sc.get<u64>(n_haps + 1), *keys = sc.get<u64>(total), *keys2 = sc.get<u64>(total);
    u32 *vals = sc.get<u32>(total), *vals2 = sc.get<u32>(total), *flag = sc.get<u32>(total), *pos = sc.get<u32>(total);
    if (!d_hs || !keys || !keys2 || !vals || !vals2 || !flag || !pos) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    hipMemcpy(d_hs, hs.data(), (n_haps + 1) * 8, hipMemcpyHostToDevice);
    k_cv_back_expand<<<nblk(total, TPB), TPB, 0, st>>>((const int*)d_cscptr, (const int*)d_cscidx, total, n_loci, n_haps, d_hs, keys, vals);
    size_t tb = 0;
    rocprim::radix_sort_pairs(nullptr, tb, keys, keys2, vals, vals2, total, 0, 64, st);
    char* tmp = sc.get<char>(tb);
    if (!tmp) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    if (rocprim::radix_sort_pairs(tmp, tb, keys, keys2, vals, vals2, total, 0, 64, st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "sort");
    k_ms_heads<<<nblk(total, TPB), TPB, 0, st>>>(keys2, total, flag);
    u32 nnz = 0;
    if (cv_scan(st, flag, total, pos, &nnz, sc) != ECB_OK) return fail(nullptr, ECB_ERR_HIP, "scan");
    k_cv_back_emit<<<nblk(total, TPB), TPB, 0, st>>>(keys2, vals2, flag, pos, total, n_loci, (int*)d_indices, (int*)d_data);
    k_cv_back_rowptr<<<nblk((u64)n_ecs + 1, TPB), TPB, 0, st>>>(keys2, pos, total, nnz, n_ecs, n_loci, (int*)d_indptr);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csc -> csr");
    *nnz_out = nnz;
    return ECB_OK;
}

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
//              test of :800-819 in two LDS atomics.  A pass of that table spans as many tiles as fit; at its end (a "flush") each
//              table entry is hashed once and summed per read (a commutative set hash that only places the EC: identity is the
//              key itself, the stand-in for the sorted string key of :307) and (c) one lane per read looks its set up in the
//              global EC table -- hash, then an exact compare of the stored key -- or inserts it, and records the slot of the
//              read.  Founders' (locus, mask) pairs are copied from LDS into the slot / the key arena by the whole wave.
//              (csrc/k_stream.inc, compiled twice: passes of up to 64 reads at five waves per SIMD; passes of up to 128 reads with one-byte
//              masks at four, for batches of short reads -- chosen per batch when ecb_hint_reads has bounded the stream's reads.)
//   k_slow     the same for single reads that do not fit a tile or hit a full table (one workgroup per read).
//   k_count    reads per EC and first read per EC (:309-312, 688-698) from the per-read slots, without global atomics:
//              partition by slot range, count in LDS.
//   finalize   rank ECs by first appearance (bitmap + scan: :682-698), scan of row lengths, sort each row by locus and
//              emit CSR A / N (:835-847, bin_utils.py:208-211).
//   k_merge    multi-GPU: re-insert another rank's serialised table (the ordered merge of :680-724).
//   k_ms_*, k_msf*   multisample: (EC, cell, file) triples of the reads (one radix sort); cell order, minimum-count filter, EC re-rank
//              and CSC N in linear passes over them (bam_utils_multisample.py:503-636, 737-791).
//   k_cv_*     CSR(bitmask) -> per-haplotype CSC as a transposition of the non-zeros; k_cvu_* / k_cvb_*: back, as a union per column
//              in LDS hash tables and the same transposition (bin_utils.py:979-1028).
//
// Integer / indexing work only: no MFMA.  The bound is HBM bandwidth: the stream part of k_stream runs at the chip's streaming rate, what is
// left above it is the EC table's random lines sharing the memory system with the stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ecb.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;

constexpr int TPB = 256;             // threads per workgroup (4 waves of 64)
constexpr u32 MAX_PROBE = 256;       // EC-table probes before a read is deferred to k_slow
constexpr u32 PENDING = 0xFFFFFFFFu;
constexpr u32 ARENA_CHUNK = 512;     // pairs a wave reserves from the key arena per global atomic
constexpr u32 QSTRIPES = 64;         // the deferred-read queue is filled in this many stripes, each with a counter of its own (queue_defer)
constexpr u32 ARENA_REGIONS = 64;    // the key arena has this many allocation cursors (see arena_alloc)
constexpr u32 MAX_LOCI = (1u << 26) - 2u;         // k_stream's LDS keys hold locus + 1 below the read's index within the pass: 26 bits (k_stream.inc) ...
constexpr u32 MAX_LOCI_SHORT = (1u << 25) - 2u;   // ... 25 in the kernel for short reads (seven bits of read index)
constexpr u32 INL = 5;               // (locus, mask) pairs of an EC's key held in its table slot; longer keys continue in the arena
constexpr u32 DEAD_KEY = 0xFFFFFFFFu;   // Slot::n1 of a slot whose key could not be stored (arena exhausted: the run fails)
constexpr u32 SPIN_MAX = 1u << 16;   // polls of a claimed slot's n1 before giving up (ERR_INTERNAL: the launch winds down; never seen)

constexpr u32 ERR_CONTRACT = 1u;     // device error bits (Counters::err)
constexpr u32 ERR_RANGE = 2u;
constexpr u32 ERR_ARENA = 4u;
constexpr u32 ERR_QUEUE = 8u;
constexpr u32 ERR_INTERNAL = 16u;

// One EC = one 64-byte line.  EC identity is EXACT: `lo` is only the hash that picks the slot; a lookup that finds its hash
// compares the read's {locus -> haplotype mask} set with the key stored in the slot, pair by pair, before it calls the slot
// its own (bam_utils.py:307-312 compares the sorted tid strings).  Two different target sets with the same 64-bit hash
// simply live in two slots.  Keys of up to INL loci sit in the line the lookup fetches anyway, so the compare costs no
// memory traffic; longer keys continue in the key arena.
// Life of a slot: lo 0 -> hash (atomicCAS: claimed); the claimant stores the pairs and the word {n1 = n + 1, off}, each
// with one 8-byte write-through store, in any order and without waiting for any of them.  A reader needs no ordering
// either: the haplotype mask of a key pair is never zero and slot and arena start out zeroed, so a key is complete exactly
// when n1 is set and its n masks are non-zero -- anything less is "not published yet" and is polled again.  Nothing ever
// changes lo, n1, off or a pair once written, until the table is cleared.
struct alignas(64) Slot {
    u64 lo;                          // 64-bit set hash, non-zero once claimed
    u32 n1;                          // 0 until the key is complete, then number of pairs + 1
    u32 off;                         // key pairs INL.. : arena[off .. off + n - INL)
    u32 count;                       // reads in this EC                       (k_count)
    u32 first_inv;                   // ~(smallest read index)                 (k_count; atomicMax on zero-initialised memory)
    uint2 pair[INL];                 // key pairs 0 .. min(n, INL): (locus, haplotype mask), in no particular order
};
static_assert(sizeof(Slot) == 64, "one EC per 64-byte line");
static_assert(offsetof(Slot, n1) == 8 && offsetof(Slot, off) == 12 && offsetof(Slot, count) == 16 && offsetof(Slot, first_inv) == 20,
              "k_count_bins reads {n1, off} and {count, first_inv} as 8-byte words");
// what ranks exchange (ecb_table_export_* / merge / adopt): 32 bytes per EC, its key pairs packed in a list of their own
struct Entry {
    u64 lo, reserved;
    u32 count, first_inv;
    u32 off, n;                      // key = pairs[off .. off + n), sorted by locus
};
static_assert(sizeof(Entry) == 32, "exchange format");

struct Counters {
    u64 all, valid;                  // records offered / passing the filter
    u64 arena_top;                   // pairs placed by ecb_table_adopt_device (dense from 0; an adopting handle allocates nothing else)
    u64 n_queue;                     // reads deferred to k_slow (k_sum_counts: the sum over the stripes below)
    u64 n_queue_s[QSTRIPES];         // ... counted per stripe of the queue (queue_defer)
    u64 n_ecs;                       // ECs created by k_slow / k_merge (k_stream's are counted by k_collect_new)
    u32 err;
    u32 full;                        // set when a read found no EC-table slot: workgroups park, host grows the table
    u64 arena_reg[ARENA_REGIONS];    // next free pair of every arena region
    u64 next_slice;                  // k_stream: next unclaimed slice of the batch (zeroed per launch)
    u64 n_mismatch;                  // exactness pass (ecb_verify_device): reads whose set differs from their EC's key
    u32 last_rid, pad_;              // read_id of the last record of the batch k_sum_counts closed (ecb_hint_reads: the host learns it here)
    u64 n_probe_tiles;               // tile visits on k_stream's probe-on path (records that found their first LDS slot taken), over the batches so far
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
// hashing: an EC is the SET of (locus, haplotype) targets of a read (bam_utils.py:307 builds a sorted string for the
// same purpose).  The set is held as {locus -> haplotype mask}; its hash is the sum over loci of a 64-bit mix of
// (locus, mask).  The sum commutes, so records need no sorting, and OR-ing haplotype bits makes duplicate
// (read, target) records vanish by itself.  The hash only places the EC in the table: equality is decided on the key.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 fmix32(u32 h) {           // murmur3 finaliser: a bijection with full avalanche
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ u64 pair_hash64(u32 locus, u32 mask) {
    // two finalisers instead of four: the hash only places an EC in the table (identity is the key), and what the sum over a
    // read's pairs needs is that the low word spreads over the slots and the high word differs where the low one collides
    const u32 h0 = fmix32(locus * 0x9E3779B1u + mask * 0x85EBCA77u + 0x7F4A7C15u);
    const u32 h1 = fmix32((h0 ^ locus) * 0xC2B2AE3Du + mask);
    return ((u64)h1 << 32) | h0;
}
// the sums of fmix32 outputs are already uniformly spread: the table hash is the sum itself (made non-zero)
__device__ __forceinline__ u64 finish_hash(u64 s0, u32 n) { return (s0 + ((u64)n << 32) + n) | 1ull; }

// record filter, bam_utils.py:264-270 (host bits 12/13 carry the two non-flag terms)
__device__ __forceinline__ bool rec_valid(u32 hf) {
    if (hf & 0x4u) return false;
    if (hf & 0x1u) {
        if ((hf & 0x80u) || !(hf & 0x2u) || (hf & (ECB_FLAG_MATE_OTHER_REF | ECB_FLAG_NEXT_POS_NEG))) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// EC-table lookup / insert with exact keys.
//   table_lookup: probes from j.  Plain loads: a cached line can only be an OLDER state of the slot (empty -> claimed ->
//     published, never back), so what a stale line can cost is an atomic or a trip through table_settle, never a wrong
//     answer: "empty" is re-checked by the CAS, "not published" and "hash equal, key different" are both settled on fresh
//     reads, and a match is a match (the pairs of a slot only ever go from zero to their final value).
//   table_settle: the slot carries our hash but its key was not (visibly) complete, or did not compare equal.  Polls n1
//     and re-reads the key with read-modify-write atomics, which execute at the memory side, and decides for good.
//   Callers that create a slot publish it (store off + pairs write-through, s_waitcnt vmcnt(0), then n1) BEFORE any lane
//   of their wave calls table_settle: the creator a lane waits for may sit in its own wave.
// Read counts and first appearances are NOT maintained here: per-read atomics on a skewed EC distribution run at
// ~5 G/s chip-wide (measured), so k_stream only records the slot of every read and k_count reduces them afterwards.
// ---------------------------------------------------------------------------------------------
enum { ST_NONE = 0, ST_LOOK, ST_HIT, ST_CREATED, ST_PENDING, ST_FULL, ST_OTHER, ST_STUCK };
enum { CMP_EQUAL = 0, CMP_DIFFERENT, CMP_INCOMPLETE, CMP_UNSURE };
struct SlotView { u32 n, off; uint2 p[INL]; };

__device__ __forceinline__ u64 fresh64(u64* p) { return atomicOr(p, 0ull); }      // a read that cannot be served from a stale cache line
__device__ __forceinline__ void store_wt64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // write-through
__device__ __forceinline__ u64 pack2(uint2 v) { return ((u64)v.y << 32) | v.x; }
__device__ __forceinline__ uint2 unpack2(u64 v) { return make_uint2((u32)v, (u32)(v >> 32)); }
__device__ __forceinline__ void publish_key(Slot* s, u32 n1, u32 off) { store_wt64(reinterpret_cast<u64*>(&s->n1), ((u64)off << 32) | n1); }   // n1 and off: one word
// pair i of the key of a PUBLISHED slot, outside the kernels that publish (finalize, export, rehash): plain loads
__device__ __forceinline__ uint2 key_pair(const Slot& s, const uint2* arena, u32 i) { return i < INL ? s.pair[i] : arena[(u64)s.off + (i - INL)]; }
// ... and of a slot that may be in the middle of being published: a read that no cache can serve
__device__ __forceinline__ uint2 key_pair_fresh(Slot* s, uint2* arena, u32 off, u32 i) {
    return unpack2(fresh64(reinterpret_cast<u64*>(i < INL ? &s->pair[i] : arena + ((u64)off + (i - INL)))));
}

// ---------------------------------------------------------------------------------------------
// EC-table lookup / insert with exact keys.  A comparator has two forms:
//   quick(view)  on the line a lookup has just loaded: CMP_EQUAL (same size, every stored pair in my set), CMP_DIFFERENT (a
//                stored pair -- complete, since its mask is non-zero -- is not in my set, or the sizes differ), CMP_INCOMPLETE
//                (a mask still zero) or CMP_UNSURE (it would take more work to tell);
//   full(slot, n, off)   reads the key itself, fresh (read-modify-write atomics): the first three, after whatever work it takes.
//   table_lookup: probes from j, the whole line in one round trip.  Plain loads: a cached line can only be an OLDER state
//     of the slot (zeros where values will be), so what a stale line can cost is an atomic or a trip through table_settle,
//     never a wrong answer: "empty" is re-checked by the CAS, "not complete" is polled, and "equal" / "different" are
//     decided on pairs that are final.
//   table_settle: the slot carries our hash but its key was not (visibly) complete, or the quick look could not tell.
//     Polls with read-modify-write atomics, which execute at the memory side, until the key is whole, and decides.
//   Callers that create a slot publish it BEFORE any lane of their wave calls table_settle: the creator a lane waits for
//   may sit in its own wave.
// Read counts and first appearances are NOT maintained here: per-read atomics on a skewed EC distribution run at
// ~5 G/s chip-wide (measured), so k_stream only records the slot of every read and k_count reduces them afterwards.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ SlotView make_view(uint4 a, uint4 b, uint4 c, uint4 d) {
    SlotView v;
    v.n = a.z - 1u; v.off = a.w;
    v.p[0] = make_uint2(b.z, b.w); v.p[1] = make_uint2(c.x, c.y); v.p[2] = make_uint2(c.z, c.w);
    v.p[3] = make_uint2(d.x, d.y); v.p[4] = make_uint2(d.z, d.w);
    return v;
}
// The pairs of a key of at most RANKED_MAX pairs in ascending order of their loci: emit(rank, locus, mask) once per pair.  The key is held in
// registers and ranked with every index a compile-time constant: pairs picked by a run-time index (`key_pair(s, arena, j)` on a copy of the
// slot) put the copy in scratch memory and every step of the ranking behind a round trip to it -- k_emit_small 0.20 ms at C3 and 0.60 on the
// paralog stream.  A pair that is not there holds locus 2^32 - 1: it ranks behind everything and is not emitted.
constexpr u32 RANKED_MAX = 16;
template <class F>
__device__ __forceinline__ void ranked_pairs(const SlotView& v, const uint2* arena, F&& emit) {
    const u32 sn = v.n;
    u32 X[RANKED_MAX], Y[RANKED_MAX];
#pragma unroll
    for (u32 i = 0; i < INL; ++i) { X[i] = i < sn ? v.p[i].x : 0xFFFFFFFFu; Y[i] = v.p[i].y; }
    if (sn <= INL) {
#pragma unroll
        for (u32 i = 0; i < INL; ++i) {
            u32 r = 0;
#pragma unroll
            for (u32 j = 0; j < INL; ++j) r += X[j] < X[i] ? 1u : 0u;    // loci within a key are distinct
            if (i < sn) emit(r, X[i], Y[i]);
        }
    } else {                                         // ... the rest of a longer key from the arena, all loads in flight together
#pragma unroll
        for (u32 k = 0; k < RANKED_MAX - INL; ++k) {
            const uint2 t = INL + k < sn ? arena[(u64)v.off + k] : make_uint2(0xFFFFFFFFu, 0u);
            X[INL + k] = t.x; Y[INL + k] = t.y;
        }
#pragma unroll
        for (u32 i = 0; i < RANKED_MAX; ++i) {
            u32 r = 0;
#pragma unroll
            for (u32 j = 0; j < RANKED_MAX; ++j) r += X[j] < X[i] ? 1u : 0u;
            if (i < sn) emit(r, X[i], Y[i]);
        }
    }
}
// ... and of a longer key, by one wave: emit(rank, locus, mask) once per pair.  The loci are ranked out of LDS -- the wave's copy of them (`sx`: BIG_LDS
// words), read four at a time, every lane the same address -- where a rank used to be a walk over the key in global memory per pair: 700 x 700
// loads for a read of 700 loci (k_emit_big 6 ms for the 39 k long ECs of tools/slow_path.py; profiles/r04_long_reads.txt).  Keys of more than
// BIG_LDS pairs walk the key in memory as before.
constexpr u32 BIG_LDS = 2048;                   // 8 KB per wave
template <class F>
__device__ __forceinline__ void ranked_pairs_wave(const Slot& s, const uint2* arena, u32 sn, u32* sx, u32 lane, F&& emit) {
    const bool in_lds = sn <= BIG_LDS;
    if (in_lds) {
        for (u32 i = lane; i < sn; i += 64) sx[i] = key_pair(s, arena, i).x;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (u32 i = lane; i < sn; i += 64) {
        const uint2 pi = key_pair(s, arena, i);
        u32 r = 0;
        if (in_lds) {
            u32 j = 0;
            for (; j + 4u <= sn; j += 4u) {
                const uint4 v = *reinterpret_cast<const uint4*>(&sx[j]);
                r += (v.x < pi.x ? 1u : 0u) + (v.y < pi.x ? 1u : 0u) + (v.z < pi.x ? 1u : 0u) + (v.w < pi.x ? 1u : 0u);
            }
            for (; j < sn; ++j) r += sx[j] < pi.x ? 1u : 0u;
        } else {
            for (u32 j = 0; j < sn; ++j) r += key_pair(s, arena, j).x < pi.x;        // loci within a key are distinct
        }
        emit(r, pi.x, pi.y);
    }
    if (in_lds) {                                  // (the wave's next key overwrites the copy)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
template <class Cmp>
__device__ __forceinline__ int table_lookup(Slot* table, u64 cap_mask, u64 lo, u64& j, u32& probes, const Cmp& cmp, u32 abl = 0u, u32* first_inv_seen = nullptr) {
    for (; probes < MAX_PROBE; ++probes, j = (j + 1) & cap_mask) {
        Slot* s = table + j;
        const uint4* q = reinterpret_cast<const uint4*>(s);
        if (first_inv_seen) *first_inv_seen = 0u;
        uint4 a = q[0], b = a, c = a, d = a;                      // lo, n1, off | count, first_inv, pair 0 | pairs 1, 2 | pairs 3, 4
        if (!(abl & 16u)) { b = q[1]; c = q[2]; d = q[3]; }       // (profiling only: 16 = first 16 bytes only, 8 = no key compare)
        u64 clo = ((u64)a.y << 32) | a.x;
        if (clo == 0ull) {
            clo = atomicCAS(&s->lo, 0ull, lo);
            if (clo == 0ull) return ST_CREATED;
            if (clo == lo) return ST_PENDING;                     // claimed with our hash a moment ago: its key decides
            continue;
        }
        if (clo != lo) continue;
        if (a.z == 0u) return ST_PENDING;
        if (a.z == DEAD_KEY) continue;
        if (first_inv_seen) *first_inv_seen = b.y;                 // (what this line -- possibly an old copy -- shows of the EC's first read)
        if (abl & 8u) return ST_HIT;
        const int r = cmp.quick(make_view(a, b, c, d));
        if (r == CMP_EQUAL) return ST_HIT;
        if (r != CMP_DIFFERENT) return ST_PENDING;
    }                                                             // (same hash, another key: a true collision takes the next slot)
    return ST_FULL;
}
template <class Cmp>
__device__ __forceinline__ int table_settle(Slot* s, const Cmp& cmp) {
    for (u32 spin = 0; spin < SPIN_MAX; ++spin) {
        const u64 w = fresh64(reinterpret_cast<u64*>(&s->n1));
        const u32 n1 = (u32)w;
        if (n1 == DEAD_KEY) return ST_OTHER;
        if (n1) {
            const int r = cmp.full(s, n1 - 1u, (u32)(w >> 32));
            if (r == CMP_EQUAL) return ST_HIT;
            if (r == CMP_DIFFERENT) return ST_OTHER;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    return ST_STUCK;
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
constexpr int NWAVE = TPB / 64;
// What k_stream needs once in a while sits behind ONE pointer (scalar registers are what this kernel runs out of: every
// kernel argument occupies a pair for the whole launch, and what does not fit is shuffled through vector lanes).
struct StreamCold {
    uint2* arena; u64 arena_cap;
    u64* queue; u64 queue_cap;       // head record index of deferred reads
    u64* resume;                     // per slice {where a relaunch takes it up, records counted up to}
    u32* wave_counts;                // per wave {records offered, records valid, ECs created}: summed by k_sum_counts
    u64* wave_arena;                 // per wave {start, pairs left} of its arena reservation, kept from launch to launch
    u64* timing;                     // profiling only (-DECB_TIMING): clocks per phase, summed over waves
    u64 chunk;                       // records per slice (a multiple of WT)
    u32 prev_rid;                    // read_id of the record before this batch (0xFFFFFFFF at stream start)
    // ECB_F_RANGES (k_stream<false, true>): reference_start of every record and the per-(locus, haplotype) extremes
    const int* pos; int2* rng; u32 n_loci, n_haps;     // rng[locus * n_haps + hap] = {min, max}: one 8-byte load per record
};
// A read k_stream hands to k_slow (longer than a pass carries, or bounced off a full table): its head goes into one of QSTRIPES stretches of the
// queue, picked by a hash of the head, each with its own counter.  (One counter for all: a stream with 1 % of reads of 700 loci made 200 k
// atomics on one address -- ~75 ns apiece at the memory side, in order: 15 ms on a 1 ms kernel, sat out by whatever the waves waited for next.
// profiles/r04_long_reads.txt)  k_queue_compact closes the stretches up for k_slow.
__device__ __forceinline__ void queue_defer(Counters* ctr, const StreamCold* C, u64 head) {
    const u32 st = (u32)((head * 0x9E3779B97F4A7C15ull) >> 58);
    static_assert(QSTRIPES == 64, "six bits of the hash pick the stripe");
    const u64 seg = C->queue_cap / QSTRIPES;
    const u64 qi = atomicAdd(&ctr->n_queue_s[st], 1ull);
    if (qi < seg) C->queue[(u64)st * seg + qi] = head; else atomicOr(&ctr->err, ERR_QUEUE);
}
// Phases of k_stream can be switched off at run time in a profiling build (libecb_ablate.so: tools/pmc_ladder.sh, tools_ablate.sh);
// the product build has the tests compiled out -- they cost a scalar register and a handful of branches per tile.
#ifdef ECB_ABLATE_RT
#define ABL(A, bits) ((A).ablate & (bits))
#else
#define ABL(A, bits) 0u
#endif
struct StreamArgs {
    const u32* rid; const u32* loc; const u32* hf;
    u64 n;
    Slot* table; u64 cap_mask;
    Counters* ctr;
    u32* read_slot;                  // slot of every read (indexed by read_id)
    u64 reads_hi;                    // read_slot holds [0, reads_hi): a read index beyond it is a broken run counter
    const StreamCold* cold;
    u32 ablate;                      // profiling builds only (-DECB_ABLATE_RT, env ECB_ABLATE): 1 = stop after (a), 2 = after (b), 4 = no EC table, 8 / 16 / 32 see table_lookup
    // ECB_F_RANGES (k_stream<false, true>) reads these two HERE and not behind `cold`: a pointer loaded from memory has no address space the compiler
    // knows, and the position stream and the range table were FLAT loads and atomics (57 of them in that compilation); the other compilations never
    // touch the two words, which then cost them nothing
    const int* pos; int2* rng;
};
// Words from one tile of a stream to the next: 512 = three arrays (ecb_push_device); 1536 = whole tiles, a tile's 512 read ids, 512 loci and 512
// haplotype/flag words side by side (ecb_push_device_tiled: rid = base, loc = base + 512, hf = base + 1024).  Read off the pointers -- three arrays
// that sit like this hold at most one tile, for which the two readings are the same -- rather than passed: a kernel argument is a scalar register
// pair for the whole launch, and the stream kernel spills them.
__device__ __forceinline__ u32 stream_tw(const u32* rid, const u32* loc, const u32* hf) { return (loc == rid + 512 && hf == rid + 1024) ? 1536u : 512u; }
#define A_TW(A) stream_tw((A).rid, (A).loc, (A).hf)
// record i of a stream whose tiles are tw words apart
__host__ __device__ __forceinline__ u64 rec_at(u64 i, u32 tw) { return (i >> 9) * (u64)tw + (i & 511ull); }

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

struct TileRegs { u32 rr[RPL], ll[RPL], hh[RPL]; };


// -DECB_TIMING: profiling build.  Every wave adds up the shader clocks it spends in each phase of a tile (stalls are
// charged to the phase whose s_waitcnt sits them out); k_stream adds them into StreamArgs::timing[8].
#ifdef ECB_TIMING
#define TICK(i) do { const u64 t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define TICK(i) do { } while (0)
#endif

// First record of read `rd` among records [b, e) -- the run counter never decreases, and it steps ON a read's first record
// (the rare paths that hand a read to k_slow ask; nothing on the way of an ordinary tile keeps head positions)
__device__ __forceinline__ u64 tile_head(const u32* rid, u64 b, u64 e, u32 rd, u32 tw) {
    while (b < e) { const u64 m = b + ((e - b) >> 1); if ((int)(rid[rec_at(m, tw)] - rd) < 0) b = m + 1; else e = m; }
    return b;
}
__device__ __forceinline__ void load_tile(const StreamArgs& A, u64 tb, u64 te, u32 lane, TileRegs& R) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    tb = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tb >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)tb);   // (wave-uniform: say so)
    if (tb + (u64)WT <= te) {
        // a whole tile: wave-uniform base (scalar registers) + 16 bytes per lane, the second group 4 KB on -- the address costs
        // one shift per tile, not six 64-bit additions
        // (non-temporal: the stream is read once; without the hint it pushes the EC table's lines out of L2 -- measured 3 %)
        const u64 tbw = (tb >> 9) * (u64)A_TW(A);              // (tb is a tile boundary)
        const char* pr = reinterpret_cast<const char*>(A.rid + tbw);
        const char* pl = reinterpret_cast<const char*>(A.loc + tbw);
        const char* ph = reinterpret_cast<const char*>(A.hf + tbw);
        const u32 off = lane * 16u;                      // (32 bits: the loads take it as an offset to the scalar base)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pr + (off + (u32)g * 1024u)));
            R.rr[4 * g] = v.x; R.rr[4 * g + 1] = v.y; R.rr[4 * g + 2] = v.z; R.rr[4 * g + 3] = v.w;
            v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pl + (off + (u32)g * 1024u)));
            R.ll[4 * g] = v.x; R.ll[4 * g + 1] = v.y; R.ll[4 * g + 2] = v.z; R.ll[4 * g + 3] = v.w;
            v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ph + (off + (u32)g * 1024u)));
            R.hh[4 * g] = v.x; R.hh[4 * g + 1] = v.y; R.hh[4 * g + 2] = v.z; R.hh[4 * g + 3] = v.w;
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {                      // the stream's last tile: record by record
        const u64 i0 = tb + (u64)g * 256 + 4u * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = i0 + j < te;
            R.rr[4 * g + j] = A.rid[rec_at(in ? i0 + j : te - 1, A_TW(A))];          // (past the end: the last read id again -- no step, no head)
            R.ll[4 * g + j] = in ? A.loc[rec_at(i0 + j, A_TW(A))] : 0u;
            R.hh[4 * g + j] = in ? A.hf[rec_at(i0 + j, A_TW(A))] : 0x4u;
        }
    }
}

// ECB_F_RANGES: the fourth stream of a tile, same shape as load_tile's three
__device__ __forceinline__ void load_pos(const int* pos, u64 tb, u64 te, u32 lane, int* P) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    tb = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tb >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)tb);
    if (tb + (u64)WT <= te) {
        const char* pp = reinterpret_cast<const char*>(pos + tb);
        const u32 off = lane * 16u;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const i32x4 v = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(pp + (off + (u32)g * 1024u)));
            P[4 * g] = v.x; P[4 * g + 1] = v.y; P[4 * g + 2] = v.z; P[4 * g + 3] = v.w;
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const u64 i0 = tb + (u64)g * 256 + 4u * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) P[4 * g + j] = i0 + j < te ? pos[i0 + j] : 0;
    }
}

// k_stream itself, twice (see k_stream.inc): the geometry of a pass is a compile-time matter (LDS per wave, waves per SIMD, key bits)
#define ECB_KS_SHORT 0
#define ECB_KS_PAR 0
namespace ks_std {
#include "k_stream.inc"
}
#undef ECB_KS_SHORT
#define ECB_KS_SHORT 1
namespace ks_short {
#include "k_stream.inc"
}
#undef ECB_KS_SHORT
#undef ECB_KS_PAR
#define ECB_KS_SHORT 0
#define ECB_KS_PAR 1
namespace ks_par {
#include "k_stream.inc"
}
#undef ECB_KS_PAR
#undef ECB_KS_SHORT

// resume points of a fresh batch: slice b starts (and has counted its records up to) record b * chunk
// (ctr: the per-launch words of the batch's first launch are zeroed here too -- three memsets fewer in front of k_stream)
__global__ void k_init_resume(u64* resume, u64 slices, u64 chunk, Counters* ctr = nullptr) {
    const u64 b = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (b < slices) { resume[2 * b] = b * chunk; resume[2 * b + 1] = b * chunk; }
    if (b == 0 && ctr) { ctr->n_queue = 0; ctr->full = 0; ctr->next_slice = 0; }
    if (b < QSTRIPES && ctr) ctr->n_queue_s[b] = 0;
}

__global__ __launch_bounds__(1024) void k_sum_counts(const u32* wave_counts, u64 waves, Counters* ctr, u32 verify, u64 offered, u64 queue_cap, const u32* d_last = nullptr) {
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
        if (verify) ctr->n_mismatch += e;                  // the exactness pass counts differing reads, and recounts nothing
        else { ctr->all += offered; ctr->valid += v; ctr->n_ecs += e; ctr->n_probe_tiles += a; }
        if (d_last) ctr->last_rid = *d_last;
        u64 nq = 0;                                        // deferred reads: what the stripes of the queue hold (a stripe that ran over has said so: ERR_QUEUE)
        for (u32 st = 0; st < QSTRIPES; ++st) nq += min(ctr->n_queue_s[st], queue_cap / QSTRIPES);
        ctr->n_queue = nq;
    }
}
// the stripes of the deferred-read queue, closed up: one workgroup per stripe
__global__ void k_queue_compact(const u64* queue, u64 queue_cap, const Counters* ctr, u64* out) {
    const u64 seg = queue_cap / QSTRIPES;
    u64 at = 0;
    for (u32 st = 0; st < blockIdx.x; ++st) at += min(ctr->n_queue_s[st], seg);
    const u64 n = min(ctr->n_queue_s[blockIdx.x], seg);
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) out[at + i] = queue[(u64)blockIdx.x * seg + i];
}

// ---------------------------------------------------------------------------------------------
// k_count: reads per EC and first appearance per EC (bam_utils.py:309-312, 688-698), reduced from the
// per-read slot ids without global atomics: partition the (slot, read) pairs by slot range (LDS
// histogram, scan, scatter), then one workgroup per range counts in LDS and owns its slots' counters.
// ---------------------------------------------------------------------------------------------
constexpr u32 BIN_BITS = 14;                   // slots per range = LDS bins of one k_count_bins workgroup: 2^14 (64 KB of LDS, two workgroups per CU;
                                               // 2^13 made the partition's runs half as long: +0.08 ms at C3) up to a table of 2^27 slots,
constexpr u32 MAX_BIN_BITS = 15;               // 2^15 (128 KB) beyond -- a run-time argument `bb`, chosen per stream (ensure_counts)
constexpr u32 MIN_BIN_BITS = 11;
constexpr u32 MAX_BUCKETS = 8192;
static_assert(MAX_BIN_BITS <= 16, "a slot's index within its range is kept in 16 bits (k_part_scatter*, k_count_bins)");              // LDS histogram of the partition passes  (=> at most 2^28 slots)

// Partition passes: few, fat workgroups.  Every workgroup keeps one open line per range in flight (2 048 of them at C3); with
// 512 workgroups of 1 024 threads those lines stay in L2 until they are full far more often than with 1 024 x 256
// (measured: -0.25 ms of 1.4 at C3; non-temporal stores, which skip that write-combining, cost +3.5 ms).
constexpr int TPB_PART = 1024;
constexpr u32 PART_G = 512;
// reads per workgroup of the partition passes (the same cut in all of them): a multiple of 4, so that a workgroup's first read sits on a 16-byte boundary
__device__ __forceinline__ u64 part_per(u64 n_reads, u64 G) { return (((n_reads + G - 1) / G) + 3ull) & ~3ull; }
__global__ __launch_bounds__(TPB_PART) void k_part_hist(const u32* read_slot, u64 n_reads, u32 n_buckets, u32 bb, u32* hist) {
    extern __shared__ u32 sh[];
    const u64 G = gridDim.x, g = blockIdx.x, per = part_per(n_reads, G);
    const u64 r0 = min(g * per, n_reads), r1 = min(r0 + per, n_reads);
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB_PART) sh[b] = 0;
    __syncthreads();
    // 16 bytes per lane where the range allows (which element a thread counts does not matter); the ragged ends one by one
    const u64 a0 = min((r0 + 3) & ~3ull, r1), a1 = max(r1 & ~3ull, a0);
    for (u64 r = r0 + threadIdx.x; r < a0; r += TPB_PART) { const u32 s = read_slot[r]; if (s != PENDING) atomicAdd(&sh[s >> bb], 1u); }
    for (u64 r = a0 + 4ull * threadIdx.x; r < a1; r += 4ull * TPB_PART) {
        const uint4 v = *reinterpret_cast<const uint4*>(read_slot + r);
        if (v.x != PENDING) atomicAdd(&sh[v.x >> bb], 1u);
        if (v.y != PENDING) atomicAdd(&sh[v.y >> bb], 1u);
        if (v.z != PENDING) atomicAdd(&sh[v.z >> bb], 1u);
        if (v.w != PENDING) atomicAdd(&sh[v.w >> bb], 1u);
    }
    for (u64 r = a1 + threadIdx.x; r < r1; r += TPB_PART) { const u32 s = read_slot[r]; if (s != PENDING) atomicAdd(&sh[s >> bb], 1u); }
    __syncthreads();
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB_PART) hist[(u64)b * G + g] = sh[b];
}

__global__ __launch_bounds__(TPB_PART) void k_part_scatter(const u32* read_slot, u64 n_reads, u32 n_buckets, u32 bb, const u32* offs,
                                                      u16* pairs) {
    extern __shared__ u32 sh[];
    const u64 G = gridDim.x, g = blockIdx.x, per = part_per(n_reads, G);
    const u64 r0 = min(g * per, n_reads), r1 = min(r0 + per, n_reads);
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
        for (int k = 0; k < 4; ++k) if (s[k] != PENDING) pos[k] = atomicAdd(&sh[s[k] >> bb], 1u);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (s[k] != PENDING) pairs[pos[k]] = (u16)(s[k] & ((1u << bb) - 1u));
    }
}

// (Elements are slot ids alone: the first read of an EC is kept current by k_stream itself -- the lookup has the slot's line in
//  hand and adds an atomicMax only when its read is earlier than what the line shows -- so nothing but counts is left here.
//  And within its range a slot id is its low `bb` <= 15 bits: the partitioned elements are 2 bytes each, half the bytes to
//  write here and to read in k_count_bins.)
// The scatter with its elements sorted by range in LDS first: a workgroup takes STAGE reads at a time, ranks them within their
// range (LDS counters), lays them out range by range in LDS and writes them from there -- elements of one range leave in
// runs of consecutive addresses (STAGE / n_buckets of them on average) instead of one 8-byte store per lane and range.
constexpr u32 STAGE = 8 * TPB_PART;            // 8 192 elements = 32 KB of LDS
constexpr u32 STAGE_MAX_BUCKETS = 4096;        // 3 x 16 KB of counters / starts / cursors beside the stage
__global__ __launch_bounds__(TPB_PART) void k_part_scatter_staged(const u32* read_slot, u64 n_reads, u32 n_buckets, u32 bb, const u32* offs,
                                                             u16* pairs) {
    extern __shared__ u32 sh[];                   // cnt[nb] | start[nb] | gcur[nb] | stage[STAGE]
    u32 *cnt = sh, *start = sh + n_buckets, *gcur = sh + 2 * n_buckets;
    u32* stage = sh + 3 * n_buckets;
    __shared__ u32 s_wave[TPB_PART / 64];
    const u64 G = gridDim.x, g = blockIdx.x, per = part_per(n_reads, G);
    const u64 r0 = min(g * per, n_reads), r1 = min(r0 + per, n_reads);
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 bpt = (n_buckets + TPB_PART - 1) / TPB_PART;            // ranges per thread in the scan (1 .. 4)
    for (u32 b = tid; b < n_buckets; b += TPB_PART) gcur[b] = offs[(u64)b * G + g];
    for (u64 rb = r0; rb < r1; rb += STAGE) {
        for (u32 b = tid; b < n_buckets; b += TPB_PART) cnt[b] = 0;
        __syncthreads();
        u32 s[8], lr[8];
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4) {                  // 16 bytes per lane and load (which element a thread takes does not matter)
            const u64 r = rb + ((u64)k4 * TPB_PART + tid) * 4u;
            if (r + 4 <= r1) {
                const uint4 v = *reinterpret_cast<const uint4*>(read_slot + r);
                s[4 * k4] = v.x; s[4 * k4 + 1] = v.y; s[4 * k4 + 2] = v.z; s[4 * k4 + 3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[4 * k4 + j] = r + j < r1 ? read_slot[r + j] : PENDING;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (s[k] != PENDING) lr[k] = atomicAdd(&cnt[s[k] >> bb], 1u);
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
            if (s[k] != PENDING) stage[start[s[k] >> bb] + lr[k]] = s[k];
        __syncthreads();
        u32 n_here = 0;
        for (u32 k = 0; k < TPB_PART / 64; ++k) n_here += s_wave[k];
        for (u32 i = tid; i < n_here; i += TPB_PART) {
            const u32 e = stage[i];
            const u32 b = e >> bb;
            pairs[gcur[b] + (i - start[b])] = (u16)(e & ((1u << bb) - 1u));
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
// The work list is built on the device (one workgroup: a few thousand ranges), so that nothing between the stream kernel and
// the CSR waits for the host.  work[0 .. *n_work): whole ranges, or pieces of `piece` elements of a range far above the average.
__global__ __launch_bounds__(1024) void k_build_work(const u32* offs, u32 G, const u64* total, u32 n_buckets, u32 piece, CountWork* work, u32 max_work, u32* n_work) {
    __shared__ u32 s_w[16];
    __shared__ u32 s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u32 b0 = 0; b0 < n_buckets; b0 += 1024) {
        const u32 b = b0 + tid;
        u32 s0 = 0, s1 = 0, np = 0;
        if (b < n_buckets) { s0 = offs[(u64)b * G]; s1 = b + 1 < n_buckets ? offs[(u64)(b + 1) * G] : (u32)*total; }      // (a range starts where its first workgroup's elements do)
        const u32 len = s1 - s0;
        const bool cut = len > piece + piece / 2;
        if (len) np = cut ? (len + piece - 1) / piece : 1u;
        const u32 incl = wave_incl_scan(np);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        u32 at = s_carry + incl - np, tot = 0;
        for (u32 k = 0; k < 16; ++k) { if (k < w) at += s_w[k]; tot += s_w[k]; }
        for (u32 k = 0; k < np; ++k)
            if (at + k < max_work) work[at + k] = cut ? CountWork{b, s0 + k * piece, min(s0 + (k + 1) * piece, s1), 1u} : CountWork{b, s0, s1, 0u};
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0) *n_work = min(s_carry, max_work);
}
// sink (optional): what finalize's k_compact would otherwise find by scanning the whole table -- see the end of the kernel
struct CompactSink { u32* list; u64 max_list; u64* n_list; u32* bitmap; u64 n_bits; uint2* list_fn; };
__global__ __launch_bounds__(TPB_COUNT) void k_count_bins(const u16* pairs, const CountWork* work, const u32* n_work, u32 bb, Slot* table, CompactSink sink) {
    extern __shared__ u32 cnt[];                           // 2^bb counters
    const u32 N_BINS = 1u << bb;
    if (blockIdx.x >= *n_work) return;                     // (the grid is the list's upper bound: its length never visits the host)
    const CountWork wk = work[blockIdx.x];
    const u32 b = wk.bucket, lane = threadIdx.x & 63u;
    const u32 start = wk.start, end = wk.end;
    for (u32 q = threadIdx.x; q < N_BINS; q += TPB_COUNT) cnt[q] = 0;
    __syncthreads();
    // one element for every lane of the wave (wave-uniform call: the ballots).  A hot EC fills most lanes of a wave: it is added
    // once per wave, the rest go one by one.
    auto add = [&](u32 bin, bool have) {
        const u64 hm = __ballot(have);
        if (!hm) return;
        const u32 v = (u32)__builtin_amdgcn_readlane((int)bin, __ffsll((long long)hm) - 1);      // (the lane index is wave-uniform: a VALU readlane, not __shfl's trip through the LDS crossbar)
        const bool same = have && bin == v;
        const u64 m = __ballot(same);
        if (__popcll(m) >= 8) {
            if (lane == (u32)(__ffsll((long long)m) - 1)) atomicAdd(&cnt[v], (u32)__popcll(m));
            if (have && !same) atomicAdd(&cnt[bin], 1u);
        } else if (have) {
            atomicAdd(&cnt[bin], 1u);
        }
    };
    // 16 bytes = 8 elements per lane and load where the piece allows, four loads in flight; its ragged ends one element per thread
    const u32 a0 = min((start + 7u) & ~7u, end), a1 = max(end & ~7u, a0);
    { const u32 i = start + threadIdx.x; const bool have = i < a0; add(have ? (u32)pairs[i] : 0u, have); }
    { const u32 i = a1 + threadIdx.x; const bool have = i < end; add(have ? (u32)pairs[i] : 0u, have); }
    for (u32 i0 = a0; i0 < a1; i0 += 32 * TPB_COUNT) {
        uint4 pv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 i = i0 + (k * TPB_COUNT + threadIdx.x) * 8u;
            pv[k] = i < a1 ? *reinterpret_cast<const uint4*>(pairs + i) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool have = i0 + (k * TPB_COUNT + threadIdx.x) * 8u < a1;
            const u32 w[4] = {pv[k].x, pv[k].y, pv[k].z, pv[k].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) add((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu, have);
        }
    }
    __syncthreads();
    // Every EC of a handle that has only seen its own reads has at least one read: the slots this workgroup adds a count to are
    // exactly the occupied slots of its range, and the first to add to a slot (counts start at zero) puts it on the list of
    // occupied slots, with its first read and key length, and marks that read in the first-appearance bitmap -- the slot's line
    // is in hand here.  finalize then needs no pass over the table (1 GB at 2^24 slots) to find 3.7 M entries.
    // A whole range with a sink (the usual case) visits every occupied slot's line ONCE: how many there are is known from the LDS
    // counters alone, so the list positions are handed out first, and then one look at the slot's line (n1, off, count,
    // first_inv) serves the count update, the list entry and the bitmap -- eight slots' loads in flight per thread.
    const bool once = sink.list && !wk.shared;
    u32 mine = 0;
    for (u32 q = threadIdx.x; q < N_BINS; q += TPB_COUNT) {
        const u32 c = cnt[q];
        u32 first = 0;
        if (once) first = c != 0u;
        else if (c) {
            Slot* s = table + (((u64)b << bb) | q);
            if (wk.shared) first = atomicAdd(&s->count, c) == 0u;
            else { s->count += c; first = 1u; }             // the only writer of its slots
        }
        if (sink.list && !once) cnt[q] = first;
        mine += first;
    }
    if (!sink.list) return;
    __shared__ u32 s_wtot[TPB_COUNT / 64];
    __shared__ u64 s_base;
    const u32 incl = wave_incl_scan(mine);
    if (lane == 63) s_wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int k = 0; k < TPB_COUNT / 64; ++k) { const u32 c = s_wtot[k]; s_wtot[k] = run; run += c; }
        s_base = run ? atomicAdd(sink.n_list, (u64)run) : 0ull;
    }
    __syncthreads();
    u64 at = s_base + s_wtot[threadIdx.x >> 6] + (incl - mine);
    if (once) {
        for (u32 q0 = threadIdx.x; q0 < N_BINS; q0 += 8 * TPB_COUNT) {
            u32 c[8];
            uint2 vk[8], vc[8];                              // {n1, off} and {count, first_inv}: two 8-byte words of the slot's line
#pragma unroll
            for (int k = 0; k < 8; ++k) { const u32 q = q0 + k * TPB_COUNT; c[k] = q < N_BINS ? cnt[q] : 0u; }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c[k]) {
                    const Slot* sl = table + (((u64)b << bb) | (q0 + k * TPB_COUNT));
                    vk[k] = *reinterpret_cast<const uint2*>(&sl->n1); vc[k] = *reinterpret_cast<const uint2*>(&sl->count);
                }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (!c[k]) continue;
                const u64 i = ((u64)b << bb) | (q0 + k * TPB_COUNT);
                table[i].count = vc[k].x + c[k];
                if (at < sink.max_list) {
                    sink.list[at] = (u32)i;
                    if (sink.list_fn) {
                        const u32 f = ~vc[k].y;
                        sink.list_fn[at] = make_uint2(f, vk[k].x - 1u);
                        if (f < sink.n_bits) atomicOr(&sink.bitmap[f >> 5], 1u << (f & 31u));
                    }
                }
                ++at;
            }
        }
        return;
    }
    for (u32 q = threadIdx.x; q < N_BINS; q += TPB_COUNT) {
        if (!cnt[q]) continue;
        const u64 i = ((u64)b << bb) | q;
        if (at < sink.max_list) {
            const Slot* s = table + i;
            sink.list[at] = (u32)i;
            if (sink.list_fn) {
                const u32 f = ~s->first_inv;
                sink.list_fn[at] = make_uint2(f, s->n1 - 1u);
                if (f < sink.n_bits) atomicOr(&sink.bitmap[f >> 5], 1u << (f & 31u));
            }
        }
        ++at;
    }
}

// ---------------------------------------------------------------------------------------------
// k_slow: one workgroup per deferred read (longer than a tile, or bounced off a full table).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_slow_len(const u32* rid, u64 n, const u64* queue, u64 nq, u64* len, u32 tw) {
    // length in records of each queued read (its head index .. the next change of read_id)
    const u64 q = blockIdx.x;
    if (q >= nq) return;
    __shared__ u64 s_end;
    const u64 h = queue[q];
    const u32 r0 = rid[rec_at(h, tw)];
    if (threadIdx.x == 0) s_end = n;
    __syncthreads();
    for (u64 b = h; b < n; b += TPB) {
        const u64 i = b + threadIdx.x;
        if (i < n && rid[rec_at(i, tw)] != r0) atomicMin(&s_end, i);
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
    u32* read_slot; u64 reads_hi;
    u64* requeue; u64* n_requeue;                           // reads that still found no slot
    u32 verify;                                             // exactness pass: compare with the read's EC instead of inserting
};

__device__ __forceinline__ u64 slow_probe_start(u32 lc, u64 cap2) { return __umul64hi((u64)(lc * 0x9E3779B1u) << 32, cap2); }

// The read's {locus -> mask} table (2 slots per record) sits in LDS when it fits SLOW_LDS slots (reads of up to 2 048 records:
// all but the pathological ones), in the global scratch otherwise: its compare-and-swaps are what a long read costs.
constexpr u32 SLOW_LDS = 4096;
// (IN_LDS is a template argument, not a pointer picked at run time: with one pointer for both homes every access to the read's table was a FLAT
//  instruction with agent-scope ordering -- compare-and-swaps, ORs and "atomic" loads on what is, for all but pathological reads, this workgroup's
//  own LDS.  profiles/r04_long_reads.txt)
template <bool IN_LDS> __device__ __forceinline__ u32 slow_ld(const u32* p) {
    if constexpr (IN_LDS) return *p; else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool IN_LDS>
__device__ __forceinline__ void slow_body(const SlowArgs& A, u32* key, u32* msk, const u64 q, const u64 h, const u64 L) {
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u64 cap2 = 2 * L;
    if (IN_LDS) {
        for (u32 p = tid; p < (u32)cap2; p += TPB) { key[p] = 0; msk[p] = 0; }
        __syncthreads();
    }
    __shared__ u64 s_acc[TPB / 64];
    __shared__ u32 s_n[TPB / 64];
    __shared__ u64 s_lo, s_j;
    __shared__ u32 s_np, s_st, s_n1, s_off, s_cnt, s_diff, s_inc, s_probes;

    for (u64 i = tid; i < L; i += TPB) {
        const u32 f = A.hf[rec_at(h + i, A_TW(A))];
        if (!rec_valid(f)) continue;
        const u32 lc = A.loc[rec_at(h + i, A_TW(A))], hap = (f >> ECB_HAP_SHIFT) & 0xFFu;
        if (lc >= A.n_loci || hap >= A.n_haps) { atomicOr(&A.ctr->err, ERR_RANGE); continue; }
        u64 p = slow_probe_start(lc, cap2);
        for (;;) {
            const u32 old = atomicCAS(&key[p], 0u, lc + 1u);
            if (old == 0u || old == lc + 1u) { atomicOr(&msk[p], 1u << hap); break; }
            if (++p == cap2) p = 0;
        }
    }
    if (!IN_LDS) __threadfence();
    __syncthreads();
    u64 a0 = 0;
    u32 np = 0;
    for (u64 p = tid; p < cap2; p += TPB) {
        const u32 k = slow_ld<IN_LDS>(&key[p]);
        if (k) {
            const u32 m = slow_ld<IN_LDS>(&msk[p]);
            ++np;
            a0 += pair_hash64(k - 1u, m);                   // same set hash as k_stream
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) a0 += __shfl_xor(a0, d);
    np = wave_sum(np);
    if (lane == 0) { s_acc[tid >> 6] = a0; s_n[tid >> 6] = np; }
    __syncthreads();
    const u32 r0 = A.rid[rec_at(h, A_TW(A))];
    if (tid == 0) {
        a0 = 0; np = 0;
        for (int w = 0; w < TPB / 64; ++w) { a0 += s_acc[w]; np += s_n[w]; }
        s_np = np; s_lo = finish_hash(a0, np); s_j = s_lo & A.cap_mask; s_probes = 0; s_cnt = 0;
        if ((u64)r0 >= A.reads_hi) { atomicOr(&A.ctr->err, ERR_CONTRACT); s_st = ST_NONE; }
        else s_st = ST_PENDING;
    }
    __syncthreads();
    np = s_np;
    // Is the key of slot s_j the set in the scratch table?  Thread 0 polls the slot's {n1, off} word; every thread takes
    // pairs tid, tid + TPB, ... with fresh reads (the slot may be in the middle of being published by another CU: a pair
    // whose mask is still zero sends everybody round again).  CMP_EQUAL / CMP_DIFFERENT, or CMP_INCOMPLETE = gave up.
    auto same_key = [&]() -> int {
        for (u32 spin = 0; spin < SPIN_MAX; ++spin) {
            __syncthreads();
            if (tid == 0) {
                const u64 w = fresh64(reinterpret_cast<u64*>(&A.table[s_j].n1));
                s_n1 = (u32)w; s_off = (u32)(w >> 32); s_diff = 0; s_inc = 0;
            }
            __syncthreads();
            const u32 n1 = s_n1;
            if (n1 == DEAD_KEY) return CMP_DIFFERENT;
            if (n1 == 0u) { __builtin_amdgcn_s_sleep(8); continue; }
            const u32 n = n1 - 1u;
            if (n != np) return CMP_DIFFERENT;
            Slot* sl = A.table + s_j;
            bool diff = false, inc = false;
            for (u32 i = tid; i < n && !diff && !inc; i += TPB) {
                u64* src = reinterpret_cast<u64*>(i < INL ? &sl->pair[i] : A.arena + ((u64)s_off + (i - INL)));
                // (a pair only ever goes from zero to its final value: a plain load that shows a mask shows the pair; only a zero is
                //  asked again where no cache can answer -- a key of 700 pairs was 700 read-modify-writes per compare)
                uint2 pr = unpack2(*src);
                if (pr.y == 0u) pr = unpack2(fresh64(src));
                if (pr.y == 0u) { inc = true; break; }
                diff = true;
                if (pr.x < A.n_loci) {
                    u64 p = slow_probe_start(pr.x, cap2);
                    for (u64 t = 0; t < cap2; ++t) {
                        const u32 k = slow_ld<IN_LDS>(&key[p]);
                        if (k == pr.x + 1u) { diff = slow_ld<IN_LDS>(&msk[p]) != pr.y; break; }
                        if (k == 0u) break;
                        if (++p == cap2) p = 0;
                    }
                }
            }
            if (diff) atomicOr(&s_diff, 1u);
            if (inc) atomicOr(&s_inc, 1u);
            __syncthreads();
            if (s_diff) return CMP_DIFFERENT;                     // (a complete pair that is not mine settles it, whatever is still missing)
            if (!s_inc) return CMP_EQUAL;
            __builtin_amdgcn_s_sleep(8);
        }
        return CMP_INCOMPLETE;
    };
    if (A.verify) {
        if (s_st != ST_NONE) {
            if (tid == 0) s_j = A.read_slot[r0];
            __syncthreads();
            const bool same = s_j != (u64)PENDING && same_key() == CMP_EQUAL;
            if (tid == 0 && !same) atomicAdd(&A.ctr->n_mismatch, 1ull);
        }
    } else {
        // find or insert, one probe sequence driven by thread 0, key compares by the whole workgroup
        for (u32 round = 0; round < 4 * MAX_PROBE && s_st == ST_PENDING; ++round) {
            __syncthreads();
            if (tid == 0) {
                u64 j = s_j; u32 probes = s_probes;
                struct Defer { __device__ int quick(const SlotView&) const { return CMP_UNSURE; } };   // ST_PENDING = "slot j carries my hash": compared below
                const int st = table_lookup(A.table, A.cap_mask, s_lo, j, probes, Defer());
                s_j = j; s_probes = probes; s_st = (u32)st; s_off = 0; s_n1 = 0;
                if (st == ST_CREATED) {
                    u64 off = 0;
                    if (np > INL) off = arena_alloc(A.ctr, A.arena_cap, np - INL, blockIdx.x);
                    atomicAdd(&A.ctr->n_ecs, 1ull);
                    s_off = (u32)off;
                    if (off == ~0ull) { atomicOr(&A.ctr->err, ERR_ARENA); s_n1 = DEAD_KEY; }
                } else if (st == ST_FULL) {
                    A.requeue[atomicAdd(A.n_requeue, 1ull)] = h;
                }
            }
            __syncthreads();
            if (s_st == ST_PENDING) {
                const int r = same_key();
                __syncthreads();
                if (tid == 0) {
                    if (r == CMP_EQUAL) { s_st = ST_HIT; A.read_slot[r0] = (u32)s_j; atomicMax(&A.table[s_j].first_inv, ~r0); }
                    else if (r == CMP_INCOMPLETE) { s_st = ST_NONE; atomicOr(&A.ctr->err, ERR_INTERNAL); }
                    else { s_j = (s_j + 1) & A.cap_mask; s_probes += 1; if (s_probes >= MAX_PROBE) { s_st = ST_FULL; A.requeue[atomicAdd(A.n_requeue, 1ull)] = h; } }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        if (s_st == ST_CREATED) {                           // publish: pairs (write-through) by everybody, {n1, off} by one; no waiting
            Slot* sl = A.table + s_j;
            const bool dead = s_n1 == DEAD_KEY;
            if (!dead) for (u64 p = tid; p < cap2; p += TPB) {
                const u32 k = slow_ld<IN_LDS>(&key[p]);
                if (k) {
                    const u32 m = slow_ld<IN_LDS>(&msk[p]);
                    const u32 pos = atomicAdd(&s_cnt, 1u);
                    uint2* dst = pos < INL ? &sl->pair[pos] : A.arena + ((u64)s_off + (pos - INL));
                    store_wt64(reinterpret_cast<u64*>(dst), pack2(make_uint2(k - 1u, m)));
                }
            }
            if (tid == 0) {
                publish_key(sl, dead ? DEAD_KEY : np + 1u, s_off);
                A.read_slot[r0] = (u32)s_j;
                atomicMax(&sl->first_inv, ~r0);
            }
        }
    }
    __syncthreads();
    if (!IN_LDS) for (u64 p = tid; p < cap2; p += TPB) {    // leave the global scratch zeroed for the next round
        if (slow_ld<IN_LDS>(&key[p])) { key[p] = 0; msk[p] = 0; }
    }
}
__global__ __launch_bounds__(TPB) void k_slow(SlowArgs A) {
    extern __shared__ u32 sh_scr[];                // SLOW_LDS keys, SLOW_LDS masks
    const u64 q = blockIdx.x;
    const u64 h = A.queue[q], L = A.len[q];
    if (2 * L <= (u64)SLOW_LDS) slow_body<true>(A, sh_scr, sh_scr + SLOW_LDS, q, h, L);
    else slow_body<false>(A, A.scr_key + A.scr_off[q], A.scr_mask + A.scr_off[q], q, h, L);
}

// ---------------------------------------------------------------------------------------------
// table maintenance: grow (rehash), compact, merge
// ---------------------------------------------------------------------------------------------
// new_of_old[i] = where slot i went: the reads' slot ids are re-mapped through it (no second lookup, no hashing)
__global__ void k_rehash(const Slot* old_t, u64 old_cap, Slot* new_t, u64 new_mask, u32* new_of_old) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < old_cap; i += (u64)gridDim.x * blockDim.x) {
        const Slot s = old_t[i];
        if (!s.n1) continue;
        u64 j = s.lo & new_mask;
        for (;; j = (j + 1) & new_mask) {                // the new table is larger; two ECs with one hash take two slots
            if (atomicCAS(&new_t[j].lo, 0ull, s.lo) == 0ull) {
                Slot t = s;
                uint4* d = reinterpret_cast<uint4*>(new_t + j);
                const uint4* v = reinterpret_cast<const uint4*>(&t);
                d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
                new_t[j].off = s.off; new_t[j].n1 = s.n1;
                new_of_old[i] = (u32)j;
                break;
            }
        }
    }
}

__global__ void k_remap_read_slot(u32* read_slot, u64 n_reads, const u32* new_of_old) {
    for (u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x; r < n_reads; r += (u64)gridDim.x * blockDim.x) {
        const u32 os = read_slot[r];
        if (os != PENDING) read_slot[r] = new_of_old[os];
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
            const bool o = i < cap && table[i].n1 != 0u;
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
                    list_fn[base + off[k]] = make_uint2(f, table[i].n1 - 1u);
                    if (f < n_bits) atomicOr(&bitmap[f >> 5], 1u << (f & 31u));
                }
            }
        __syncthreads();
    }
}

// ... and with the list already there (a merged table that kept it): what k_compact writes next to the list, from the list
__global__ void k_list_fn(const Slot* table, const u32* list, u64 max_list, const u64* n_list, u64* n_out, u32* bitmap, u64 n_bits, uint2* list_fn) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    const u64 n = *n_list;
    if (e == 0) *n_out = n;
    if (e >= n || e >= max_list) return;
    const Slot* s = table + list[e];
    const u32 f = ~s->first_inv;
    list_fn[e] = make_uint2(f, s->n1 - 1u);
    if (f < n_bits) atomicOr(&bitmap[f >> 5], 1u << (f & 31u));
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
        if (e < e1) { const Slot& s = table[list[e]]; q = part_of(s.lo, n_parts); np = s.n1 - 1u; }
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
// pair_base[q] = first pair of part q: Entry::off is written relative to it.  Two walks over the workgroup's entries:
// count per part, reserve (one global atomic per part), then place.  Keys leave SORTED by locus (ranked like the CSR rows
// of k_emit_*): two exported keys are then equal iff they are equal element by element, which is what k_merge tests first.
// Keys of more than EXPORT_SMALL pairs are queued for k_parts_sort_big (one wave each).
constexpr u32 EXPORT_SMALL = RANKED_MAX;
__global__ __launch_bounds__(TPB) void k_parts_export(const Slot* table, const u32* list, u64 n, const uint2* arena, u32 n_parts,
                                                      u64* cur, const u64* pair_base, Entry* out_e, uint2* out_p, u32 read_base,
                                                      u64* big, u32* n_big) {
    __shared__ u32 ce[MAX_PARTS], cp[MAX_PARTS];
    __shared__ u64 be[MAX_PARTS], bp[MAX_PARTS];
    const u32 lane = threadIdx.x & 63u;
    const u64 e0 = blockIdx.x * (u64)PARTS_PER_BLOCK, e1 = min(e0 + PARTS_PER_BLOCK, n);
    for (int pass = 0; pass < 2; ++pass) {
        if (threadIdx.x < MAX_PARTS) { ce[threadIdx.x] = 0; cp[threadIdx.x] = 0; }
        __syncthreads();
        for (u64 eb = e0; eb < e1; eb += TPB) {
            const u64 e = eb + threadIdx.x;
            uint4 sa = make_uint4(0u, 0u, 1u, 0u), sb = sa, sc = sa, sd = sa;      // the slot's line (n1 = 1: an empty key)
            u32 q = MAX_PARTS, re = 0, rp = 0, sn = 0;
            if (e < e1) {
                const uint4* sl = reinterpret_cast<const uint4*>(table + list[e]);
                sa = sl[0]; sb = sl[1]; sc = sl[2]; sd = sl[3];
                q = part_of(((u64)sa.y << 32) | sa.x, n_parts); sn = sa.z - 1u;
            }
            for (u32 t = 0; t < n_parts; ++t) {      // rank within the workgroup: wave prefix + one LDS atomic per wave and part
                const bool mine = q == t;
                const u64 m = __ballot(mine);
                if (!m) continue;
                const u32 v = mine ? sn : 0u, incl = wave_incl_scan(v);
                const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);   // (read here, with every lane alive: inside the branch below
                u32 b_e = 0, b_p = 0;                                            //  the compiler sinks the scan's last add under exec = lane 0)
                if (lane == 0) { b_e = atomicAdd(&ce[t], (u32)__popcll(m)); b_p = atomicAdd(&cp[t], tot); }
                b_e = (u32)__builtin_amdgcn_readfirstlane((int)b_e); b_p = (u32)__builtin_amdgcn_readfirstlane((int)b_p);
                if (mine) { re = b_e + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); rp = b_p + incl - v; }
            }
            if (pass == 1 && e < e1) {
                const u64 po = bp[q] + rp;
                if (sn <= EXPORT_SMALL) {
                    ranked_pairs(make_view(sa, sb, sc, sd), arena, [&](u32 r, u32 x, u32 y) { out_p[po + r] = make_uint2(x, y); });
                } else {
                    const u32 bi = atomicAdd(n_big, 1u);
                    big[2 * (u64)bi] = po; big[2 * (u64)bi + 1] = ((u64)list[e] << 32) | sn;
                }
                Entry en;
                en.lo = ((u64)sa.y << 32) | sa.x; en.reserved = 0; en.count = sb.x;
                en.first_inv = ~(~sb.y + read_base);
                en.off = (u32)(po - pair_base[q]); en.n = sn;
                out_e[be[q] + re] = en;
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
// long keys of an export: one wave per key, rank = number of smaller loci
__global__ __launch_bounds__(TPB) void k_parts_sort_big(const Slot* table, const uint2* arena, const u64* big, u32 n_big, uint2* out_p) {
    __shared__ __attribute__((aligned(16))) u32 sx[TPB / 64][BIG_LDS];
    const u32 b = (blockIdx.x * TPB + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (b >= n_big) return;
    const u64 po = big[2 * (u64)b];
    const u32 n = (u32)big[2 * (u64)b + 1];
    const Slot& s = table[big[2 * (u64)b + 1] >> 32];
    ranked_pairs_wave(s, arena, n, sx[threadIdx.x >> 6], lane, [&](u32 r, u32 x, u32 y) { out_p[po + r] = make_uint2(x, y); });
}
// adopt: entries known to be distinct ECs go to consecutive slots of an empty table, no hashing.  The pair list was copied to
// the arena at arena_base as it is: the slot takes its first INL pairs, the rest stay where they are.
__global__ void k_adopt(const Entry* ent, u64 n, const uint2* pairs, Slot* table, u64 slot_base, u32 arena_base) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const Entry en = ent[e];
    Slot s{};
    s.lo = en.lo; s.n1 = en.n + 1u; s.off = arena_base + en.off + INL; s.count = en.count; s.first_inv = en.first_inv;
    for (u32 i = 0; i < INL && i < en.n; ++i) s.pair[i] = pairs[(u64)en.off + i];
    table[slot_base + e] = s;
}

// The ordered merge of bam_utils.py:680-724, one thread per incoming EC: find its key in the table (exact compare) or
// insert it; counts are added, first appearances minimised.
struct ListCmp {                                   // incoming key = a sorted pair list; stored key = slot + arena, read fresh
    const uint2* inc; u32 n; uint2* arena;
    __device__ __forceinline__ int quick(const SlotView& v) const { return v.n != n ? CMP_DIFFERENT : CMP_UNSURE; }   // (keys are compared in full(): not a hot path)
    __device__ __noinline__ int full(Slot* s, u32 sn, u32 off) const {
        if (sn != n) return CMP_DIFFERENT;
        bool same = true, inc_ = false;            // both sorted (keys that came from an export): element by element
        for (u32 i = 0; i < n && same; ++i) {
            const uint2 a = key_pair_fresh(s, arena, off, i), b = inc[i];
            inc_ |= a.y == 0u;
            same = a.x == b.x && a.y == b.y;
        }
        if (same) return CMP_EQUAL;
        if (inc_) return CMP_INCOMPLETE;
        for (u32 i = 0; i < n; ++i) {              // the stored key may be one k_stream wrote, in no order: compare as sets
            const uint2 a = key_pair_fresh(s, arena, off, i);
            if (a.y == 0u) return CMP_INCOMPLETE;
            bool found = false;
            for (u32 k = 0; k < n && !found; ++k) found = inc[k].x == a.x && inc[k].y == a.y;
            if (!found) return CMP_DIFFERENT;
        }
        return CMP_EQUAL;
    }
};
constexpr u32 MERGE_PER_BLOCK = 2 * TPB;      // entries per workgroup: a piece of one key range is a few hundred thousand entries, and with 16 x TPB
                                              // each it occupied 71 of the 256 CUs (one EC-count atomic per workgroup, fire and forget)
// All tables of a call in ONE launch (blockIdx.y = table): counts add and first reads minimise in any order, and a key that
// two tables bring at the same moment is settled by the publication protocol like two waves of k_stream founding one EC --
// eight pieces of a key range one after the other were eight launches of 0.04 ms that each left most of the chip idle.
struct MergeDesc { const Entry* ent; const uint2* pairs; u64 n, n_pairs; };
// (list / n_list, optional: every slot this launch founds is appended -- a table that started empty, or whose list of occupied
//  slots was current, keeps a current list, and neither an export nor finalize has to scan it for its few occupied slots)
__global__ __launch_bounds__(TPB) void k_merge(const MergeDesc* D, Slot* table, u64 cap_mask, uint2* arena, u64 arena_cap, Counters* ctr,
                                               u32* list, u64 list_cap, u64* n_list) {
    const MergeDesc d = D[blockIdx.y];
    const Entry* const ent = d.ent; const uint2* const pairs = d.pairs;
    const u64 n = d.n, n_pairs = d.n_pairs;
    if (blockIdx.x * (u64)MERGE_PER_BLOCK >= n) return;       // (the grid is as wide as the longest table)
    __shared__ u32 s_new;
    __shared__ u32 s_found[MERGE_PER_BLOCK];                  // the slots this workgroup founds (for `list`: one reservation per workgroup)
    __shared__ u32 s_nfound;
    __shared__ u64 s_lbase;
    if (threadIdx.x == 0) { s_new = 0; s_nfound = 0; }
    __syncthreads();
    const u32 lane = threadIdx.x & 63u;
    const u64 e0 = blockIdx.x * (u64)MERGE_PER_BLOCK, e1 = min(e0 + MERGE_PER_BLOCK, n);
    u32 my_new = 0;                                // (wave-uniform)
    for (u64 eb = e0; eb < e1; eb += TPB) {
        const u64 e = eb + threadIdx.x;
        const bool on = e < e1;
        Entry s{};
        if (on) s = ent[e];
        const bool ok = on && (u64)s.off + s.n <= n_pairs && s.lo != 0ull;
        if (on && !ok) atomicOr(&ctr->err, ERR_CONTRACT);
        ListCmp cmp{pairs + s.off, s.n, arena};
        u64 j = s.lo & cap_mask;
        u32 probes = 0;
        int st = ST_NONE;
        if (ok) st = table_lookup(table, cap_mask, s.lo, j, probes, cmp);
        bool created = false;
        for (u32 round = 0;; ++round) {            // publish, then settle: as in k_stream (the creator a lane waits for may be its neighbour)
            const u64 cm = __ballot(st == ST_CREATED);
            if (cm) {
                const bool cr = st == ST_CREATED;
                const u32 want = cr && s.n > INL ? s.n - INL : 0u;
                u64 at = 0;
                if (__ballot(want != 0u)) {                                 // key arena: one reservation per wave, not per created EC
                    const u32 incl = wave_incl_scan(want);
                    const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
                    if (lane == 0) at = arena_alloc(ctr, arena_cap, total, (blockIdx.y * gridDim.x + blockIdx.x) * (TPB / 64) + (threadIdx.x >> 6));
                    at = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(at >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)at);
                    if (at == ~0ull) { if (lane == 0) atomicOr(&ctr->err, ERR_ARENA); }
                    else at += incl - want;
                }
                my_new += (u32)__popcll(cm);
                if (list) {                                                 // (uniform; every entry founds at most one slot)
                    const int first = __ffsll((long long)cm) - 1;
                    u32 lb = 0;
                    if ((int)lane == first) lb = atomicAdd(&s_nfound, (u32)__popcll(cm));
                    lb = (u32)__builtin_amdgcn_readlane((int)lb, first);
                    if (cr) s_found[lb + __builtin_amdgcn_mbcnt_hi((u32)(cm >> 32), __builtin_amdgcn_mbcnt_lo((u32)cm, 0u))] = (u32)j;
                }
                if (cr) {
                    const bool dead = at == ~0ull;
                    if (!dead)
                        for (u32 t = 0; t < s.n; ++t) {
                            uint2* dst = t < INL ? &table[j].pair[t] : arena + (at + (t - INL));
                            store_wt64(reinterpret_cast<u64*>(dst), pack2(pairs[(u64)s.off + t]));
                        }
                    publish_key(table + j, dead ? DEAD_KEY : s.n + 1u, (u32)at);
                    st = ST_HIT; created = true;
                }
            }
            if (!__ballot(st == ST_PENDING)) break;
            if (round >= 64u) { if (st == ST_PENDING) { atomicOr(&ctr->err, ERR_INTERNAL); st = ST_NONE; } break; }
            if (st == ST_PENDING) {
                const int r = table_settle(table + j, cmp);
                if (r == ST_HIT) st = ST_HIT;
                else if (r == ST_STUCK) { atomicOr(&ctr->err, ERR_INTERNAL); st = ST_NONE; }
                else { j = (j + 1) & cap_mask; ++probes; st = table_lookup(table, cap_mask, s.lo, j, probes, cmp); }
            }
        }
        (void)created;
        if (st == ST_FULL) atomicAdd(&ctr->n_queue, 1ull);             // host sizes the table so this cannot happen
        else if (st == ST_HIT) {
            atomicAdd(&table[j].count, s.count);                      // one pair of atomics per merged EC, not per read
            atomicMax(&table[j].first_inv, s.first_inv);
        }
    }
    if (lane == 0 && my_new) atomicAdd(&s_new, my_new);
    __syncthreads();
    if (threadIdx.x == 0 && s_new) atomicAdd(&ctr->n_ecs, (u64)s_new);
    if (list) {
        if (threadIdx.x == 0) s_lbase = s_nfound ? atomicAdd(n_list, (u64)s_nfound) : 0ull;
        __syncthreads();
        for (u32 i = threadIdx.x; i < s_nfound; i += TPB) if (s_lbase + i < list_cap) list[s_lbase + i] = s_found[i];
    }
}

// ---------------------------------------------------------------------------------------------
// finalize: rank by first appearance, CSR emit
// ---------------------------------------------------------------------------------------------
// First-appearance ranks: rank of read f among the marked reads = marked reads before f's 64-byte line of the bitmap (an exclusive scan of the
// lines' popcounts: 1/16 of the bitmap's words, a table that stays in L2) + the marked bits before f within its line.  (A prefix per 32-bit word
// made every rank two reads of lines that are nowhere near each other -- the bitmap's and the prefix array's -- where this is one.)
constexpr u32 BM_LINE = 16;                     // bitmap words per line; the bitmap is allocated in whole lines (bitmap_words)
inline u64 bitmap_words(u64 n_bits) { return ((n_bits + 31) / 32 + BM_LINE) / BM_LINE * BM_LINE; }
__global__ void k_popc(const u32* in, u64 n_lines, u32* out) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    const uint4* q = reinterpret_cast<const uint4*>(in + i * BM_LINE);
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < (int)BM_LINE / 4; ++k) { const uint4 v = q[k]; c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
    out[i] = c;
}
__device__ __forceinline__ u32 bit_rank(const u32* bitmap, const u32* lprefix, u32 f) {
    const uint4* q = reinterpret_cast<const uint4*>(bitmap + ((f >> 5) & ~(BM_LINE - 1u)));
    const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
    const u32 w[BM_LINE] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    const u32 k = (f >> 5) & (BM_LINE - 1u), below = (1u << (f & 31u)) - 1u;
    u32 r = lprefix[f >> 9];
    static_assert(BM_LINE == 16, "f >> 9: 512 bits per line");
#pragma unroll
    for (u32 i = 0; i < BM_LINE; ++i) r += __popc(w[i] & (i < k ? 0xFFFFFFFFu : i == k ? below : 0u));
    return r;
}

// Exclusive scan of u32, ONE pass over the data: a workgroup takes the next stretch of 16 384 values (a ticket: stretches start in
// order), publishes its sum, and one of its waves walks back over the stretches before it, 64 status words per step, adding their
// sums up to the nearest one whose inclusive prefix is known -- decoupled look-back, as in the radix sort below.  A status word is
// 2 bits of state and a 62-bit value, written and read with relaxed agent-scope atomics; the words and the ticket are zeroed by the
// launcher (scan_launch).  (Rounds 1-3a: per-block sums, a scan of the sums, per-block scan + offset -- three launches, the input read
// twice, and 4-byte loads at a 32-byte lane stride; 18 such scans were 1.4 ms of the config-4 path.)
// (stride: the input may be one member of an array of small structs -- element i is in[i * stride])
constexpr int SCB_TPB = 1024, SCB_ITEMS = 16, SCB = SCB_TPB * SCB_ITEMS;
constexpr u64 SC_AGG = 1ull << 62, SC_PFX = 2ull << 62, SC_VAL = (1ull << 62) - 1ull;
inline u64 scan_blocks(u64 n) { return std::max<u64>(1, (n + SCB - 1) / SCB); }
inline u64 scan_words(u64 n) { return 2 * (scan_blocks(n) + 2); }             // u32 words of scratch a scan of n values needs
__global__ __launch_bounds__(SCB_TPB) void k_scan_lb(const u32* in, u64 n, u32* out, u64* st, u64 nb, u64* d_total, u32 stride, u32* last32) {
    __shared__ u64 s_b, s_excl;
    __shared__ u32 s_w[SCB_TPB / 64];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid == 0) s_b = atomicAdd(reinterpret_cast<unsigned long long*>(&st[nb]), 1ull);
    __syncthreads();
    const u64 b = s_b;
    if (b >= nb) return;
    const u64 base = b * SCB + (u64)tid * SCB_ITEMS;
    u32 v[SCB_ITEMS];
    const bool whole = base + SCB_ITEMS <= n;
    if (stride == 1u && whole && (reinterpret_cast<uintptr_t>(in) & 15u) == 0u) {
#pragma unroll
        for (int k = 0; k < SCB_ITEMS / 4; ++k) {
            const uint4 q = reinterpret_cast<const uint4*>(in + base)[k];
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCB_ITEMS; ++k) v[k] = base + k < n ? in[(base + k) * stride] : 0u;
    }
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < SCB_ITEMS; ++k) s += v[k];
    const u32 incl = wave_incl_scan(s);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u32 before = 0, tot = 0;
#pragma unroll
    for (u32 k = 0; k < SCB_TPB / 64; ++k) { const u32 c = s_w[k]; if (k < w) before += c; tot += c; }
    if (w == 0) {
        u64 excl = 0;
        if (lane == 0) __hip_atomic_store(&st[b], (b == 0 ? SC_PFX : SC_AGG) | (u64)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b > 0) {
            long long t = (long long)b - 1;                  // the nearest stretch not yet added
            for (u32 spins = 0; spins < (1u << 24);) {
                const long long idx = t - (long long)lane;
                const u64 x = idx >= 0 ? __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SC_PFX;      // (before the first: prefix 0)
                const u64 ready = __ballot(x != 0ull), pfx = __ballot(x >= SC_PFX);
                u64 take = 0;                                // lanes whose value goes in
                if (pfx) {
                    const int p = __ffsll((long long)pfx) - 1;
                    const u64 need = p == 63 ? ~0ull : (1ull << (p + 1)) - 1ull;
                    if ((ready & need) == need) take = need;
                } else if (ready == ~0ull) take = ~0ull;
                if (!take) { ++spins; __builtin_amdgcn_s_sleep(1); continue; }
                u64 val = (take >> lane & 1ull) ? (x & SC_VAL) : 0ull;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) val += __shfl_xor(val, d);
                excl += val;
                if (pfx) break;
                t -= 64;
            }
            if (lane == 0) __hip_atomic_store(&st[b], SC_PFX | ((excl + tot) & SC_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = excl;
            if (b + 1 == nb && d_total) *d_total = excl + tot;
            if (b + 1 == nb && last32) *last32 = (u32)(excl + tot);      // (a row-pointer array's last element: the total, where its scan ends)
        }
    }
    __syncthreads();
    u32 run = (u32)s_excl + before + incl - s;
    if (whole && (reinterpret_cast<uintptr_t>(out) & 15u) == 0u) {
#pragma unroll
        for (int k = 0; k < SCB_ITEMS / 4; ++k) {
            uint4 q;
            q.x = run; run += v[4 * k]; q.y = run; run += v[4 * k + 1]; q.z = run; run += v[4 * k + 2]; q.w = run; run += v[4 * k + 3];
            reinterpret_cast<uint4*>(out + base)[k] = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCB_ITEMS; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    }
}
// queue a scan on `st`: scratch = scan_words(n) u32 words (8-byte aligned); the total (64 bits) lands in *d_total (may be null)
inline hipError_t scan_launch(hipStream_t st, const u32* in, u64 n, u32* out, u32* scratch, u64* d_total, u32 stride = 1, u32* last32 = nullptr) {
    const u64 nb = scan_blocks(n);
    u64* words = reinterpret_cast<u64*>(scratch);
    hipError_t e = hipMemsetAsync(words, 0, ((nb + 1) * 8 + 15) & ~15ull, st);      // (nb + 1 words; in whole 16 bytes -- scan_words has the room --: the runtime fills a ragged end with a kernel of its own)
    if (e != hipSuccess) return e;
    k_scan_lb<<<(unsigned)nb, SCB_TPB, 0, st>>>(in, n, out, words, nb, d_total, stride, last32);
    return hipGetLastError();
}

// (slot and row length of rank r go out as one 8-byte word: the scatter is what this kernel costs, and two 4-byte stores to
//  two arrays were twice the partial lines; k_emit_small, which walks the ranks in order, writes `order` for the later readers)
__global__ void k_rank(const u32* list, const uint2* list_fn, u64 n, u64 n_bits, const u32* bitmap, const u32* wprefix,
                       uint2* ord2) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u32 si = list[e];
    const uint2 fn = list_fn[e];
    const u32 f = fn.x;
    if (f >= n_bits) return;                       // (an EC without a first read: finalize's count check reports it)
    const u32 r = bit_rank(bitmap, wprefix, f);
    if (r >= n) return;
    ord2[r] = make_uint2(si, fn.y);
}
// EC rank of every occupied slot: only the per-read EC ids need it (multisample, ecb_export_read_ec) -- a scatter over the
// whole table's index space that the single-sample result never reads
__global__ void k_slot_ranks(const u32* order, u64 n, u32* rank_of_slot) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (r < n) rank_of_slot[order[r]] = (u32)r;
}

// CSR rows: every (locus, mask) pair of an EC is ranked by locus and written in place (columns ascending, as scipy's
// csc -> csr leaves them: bin_utils.py:211).  Short rows: one thread each.  Long rows: queued, one wave each.
// The indices the host supplied are validated here, once per EC instead of once per record.
constexpr u32 EMIT_SMALL = RANKED_MAX;
// The rows of a wave's 64 ECs are one stretch of `indices` / `data`: the ranked pairs are laid out in LDS and leave in whole lines (a lane
// storing its own row pair by pair wrote 4 bytes at a time, some twenty bytes apart from its neighbour's: three times the bytes in write
// transactions).  A wave with a long row among its 64 (k_emit_big's) stores directly.
__global__ __launch_bounds__(TPB) void k_emit_small(const Slot* table, const uint2* ord2, u32* order, u64 n, const uint2* arena,
                                                     const u32* indptr, int* indices, int* data, int* counts,
                                                     u32 n_loci, u32 n_haps, u32* big, u32* n_big, Counters* ctr) {
    __shared__ u32 sx[TPB / 64][64 * EMIT_SMALL], sy[TPB / 64][64 * EMIT_SMALL];
    const u64 e = (u64)blockIdx.x * TPB + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool have = e < n;
    const u64 hm = __ballot(have);
    if (!hm) return;
    uint4 a = make_uint4(0u, 0u, 1u, 0u), b = a, c = a, d = a;      // (n1 = 1: an empty key)
    u32 dst = 0;
    if (have) {
        const u32 si = ord2[e].x;
        order[e] = si;
        const uint4* q = reinterpret_cast<const uint4*>(table + si);
        a = q[0]; b = q[1]; c = q[2]; d = q[3];
        counts[e] = (int)b.x;                       // Slot::count
        dst = indptr[e];
    }
    const SlotView v = make_view(a, b, c, d);
    const bool is_big = have && v.n > EMIT_SMALL;
    if (is_big) big[atomicAdd(n_big, 1u)] = (u32)e;
    const u32 base = (u32)__builtin_amdgcn_readfirstlane((int)dst);                                     // (lane 0 has an EC whenever any lane has)
    const u32 end = (u32)__builtin_amdgcn_readlane((int)(dst + v.n), 63 - __builtin_clzll(hm));        // ... and the last one that has ends the stretch
    const bool staged = __ballot(is_big) == 0ull && end - base <= 64u * EMIT_SMALL;
    bool bad = false;
    if (!is_big)
        ranked_pairs(v, arena, [&](u32 r, u32 x, u32 y) {
            if (staged) { const u32 at = min(dst - base + r, 64u * EMIT_SMALL - 1u); sx[w][at] = x; sy[w][at] = y; }     // (the bound: a broken row pointer must not reach beyond the stage)
            else { indices[dst + r] = (int)x; data[dst + r] = (int)y; }
            bad |= x >= n_loci || (y >> n_haps) != 0u;
        });
    if (staged) {
        wave_sync();
        for (u32 i = lane; i < end - base; i += 64u) { indices[base + i] = (int)sx[w][i]; data[base + i] = (int)sy[w][i]; }
    }
    if (bad) atomicOr(&ctr->err, ERR_RANGE);
}
__global__ __launch_bounds__(TPB) void k_emit_big(const Slot* table, const u32* order, const u32* big, const u32* n_big,
                                                   const uint2* arena, const u32* indptr, int* indices, int* data,
                                                   u32 n_loci, u32 n_haps, Counters* ctr) {
    __shared__ __attribute__((aligned(16))) u32 sx[TPB / 64][BIG_LDS];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6, nb = *n_big;
    bool bad = false;
    for (u32 b = (blockIdx.x * TPB + threadIdx.x) >> 6; b < nb; b += (gridDim.x * TPB) >> 6) {
        const u32 e = big[b];
        const Slot& s = table[order[e]];
        const u32 dst = indptr[e];
        ranked_pairs_wave(s, arena, s.n1 - 1u, sx[w], lane, [&](u32 r, u32 x, u32 y) {
            indices[dst + r] = (int)x;
            data[dst + r] = (int)y;
            bad |= x >= n_loci || (y >> n_haps) != 0u;
        });
    }
    if (bad) atomicOr(&ctr->err, ERR_RANGE);
}

// multisample: sort key = EC rank << 32 | (cell, file) of every read, the cell ABOVE the file: the triples come out in the order
// (EC, cell, file), in which the files of one (EC, cell) pair -- one entry of N -- follow each other
constexpr u32 MS_FILE_BITS = 32 - ECB_CELL_BITS;
__host__ __device__ __forceinline__ u32 ms_swz(u32 meta) { return ((meta & ((1u << ECB_CELL_BITS) - 1u)) << MS_FILE_BITS) | (meta >> ECB_CELL_BITS); }
__host__ __device__ __forceinline__ u32 ms_unswz(u32 s) { return (s >> MS_FILE_BITS) | (s << ECB_CELL_BITS); }
__host__ __device__ __forceinline__ u64 ms_swz_key(u64 key) { return (key & 0xFFFFFFFF00000000ull) | ms_swz((u32)key); }
__global__ void k_ms_keys(const u32* read_slot, const u32* rank_of_slot, const u32* meta, u64 n, u64* keys, u32* vals) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = ((u64)rank_of_slot[read_slot[i]] << 32) | ms_swz(meta[i]); vals[i] = (u32)i; }
}
__global__ void k_ms_swz_keys(u64* keys, u64 n) {           // EC << 32 | meta  ->  the sort key above, in place
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ms_swz_key(keys[i]);
}
__global__ void k_ms_heads(const u64* keys, u64 n, u32* flag) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_ms_emit(const u64* keys, const u32* vals, const u32* flag, const u32* pos, u64 n, u32 n_out,
                          u64* okey, u32* ofirst, u32* ostart) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n && flag[i]) { okey[pos[i]] = (keys[i] & 0xFFFFFFFF00000000ull) | ms_unswz((u32)keys[i]); ofirst[pos[i]] = vals[i]; ostart[pos[i]] = (u32)i; }
    if (i == 0) ostart[n_out] = (u32)n;
}
__global__ void k_ms_split(const u64* okey, const u32* ostart, const u32* ocount, u64 n, u32* ec, u32* meta, u32* count) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { ec[i] = (u32)(okey[i] >> 32); meta[i] = (u32)okey[i]; count[i] = ocount ? ocount[i] : ostart[i + 1] - ostart[i]; }
}
// multisample across GPUs: the keys of the final ECs in rank order; their ranks looked up in a shard's own table; the
// shards' (EC, cell, file) triples combined on the root
__global__ void k_export_keys(const Slot* table, const u32* order, u64 n, u64* out) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e < n) out[e] = table[order[e]].lo;
}
// ... and where no table stands behind the result (it was assembled from per-range pieces): the hash of an EC is a function of its
// key alone -- the sum of its pairs' hashes, as the stream kernel forms it -- so it is taken from the finished CSR row
__global__ void k_row_keys(const u32* indptr, const int* indices, const int* data, u64 n, u64* out) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u32 a = indptr[e], z = indptr[e + 1];
    u64 acc = 0;
    for (u32 i = a; i < z; ++i) acc += pair_hash64((u32)indices[i], (u32)data[i]);
    out[e] = finish_hash(acc, z - a);
}
// EC e of the merged result = (hash keys[e], CSR row e with ascending loci).  Find it in this shard's table, by its hash and
// then by its key, pair for pair: grank[slot] = e.
__global__ void k_set_global_rank(const u64* keys, const int* indptr, const int* indices, const int* data, u64 n,
                                  const Slot* table, u64 cap_mask, const uint2* arena, u32* grank) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u64 lo = keys[e];
    const u32 r0 = (u32)indptr[e], rn = (u32)indptr[e + 1] - r0;
    u64 j = lo & cap_mask;
    for (u64 probe = 0; probe <= cap_mask; ++probe, j = (j + 1) & cap_mask) {
        const Slot& s = table[j];
        if (s.lo == 0ull) return;                                 // this shard never saw that EC
        if (s.lo != lo || s.n1 != rn + 1u) continue;
        bool same = true;
        for (u32 i = 0; i < rn && same; ++i) {
            const uint2 pr = key_pair(s, arena, i);
            u32 a = 0, b = rn;                                    // the row is sorted by locus
            while (a < b) { const u32 m = (a + b) >> 1; if ((u32)indices[r0 + m] < pr.x) a = m + 1; else b = m; }
            same = a < rn && (u32)indices[r0 + a] == pr.x && (u32)data[r0 + a] == pr.y;
        }
        if (same) { grank[j] = (u32)e; return; }
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
    if (flag[i]) okey[o] = (keys[i] & 0xFFFFFFFF00000000ull) | ms_unswz((u32)keys[i]);
    atomicAdd(&ocount[o], cnt_in[idx[i]]);
    atomicMin(&ofirst[o], first_in[idx[i]]);
}


// ---------------------------------------------------------------------------------------------
// f-2: CSR(bitmask) <-> per-haplotype CSC (bin_utils.py:979-1028).  CSR -> CSC: a transposition of the non-zeros (k_cv_keys ...
// k_cv_ptr below).  CSC -> CSR: the set bits expanded to (row * T + column) keys, the stable radix sort (the LSD sort below),
// runs of equal keys OR-ed, row pointers by binary search on the sorted keys.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 lower_bound_u64(const u64* a, u64 n, u64 v) {
    u64 lo = 0, hi = n;
    while (lo < hi) { const u64 m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo;
}
// ---- per-haplotype CSC -> CSR(bitmask): a union per column, then a transposition ----------------------------------------------
//   k_cvu_colsum    S[l] = sum over haplotypes of their column pointers (S[l + 1] - S[l] row indices name column l); the pointers
//                   are checked here
//   k_cvu_pieces    a column is one piece of work, or -- above CVU_PIECE row indices -- several: equal ranges of EC ids
//   k_cvu_items / k_cvu_bounds   the list of pieces; where every haplotype's list of the column enters a piece's EC range
//                   (binary searches: lists of a column that is cut must be ascending, as scipy and the forward conversion write them)
//   k_cvu_union     one workgroup per piece: the piece's row indices from every haplotype's list go into an LDS hash
//                   {EC -> haplotype mask}; within a piece no order is needed and an EC listed twice ORs into the same bit.
//                   First launch: distinct ECs per column; second, behind a scan of those: the entries go out column by column as
//                   (EC << 32 | locus, mask) -- within a column in any order --
//   then a stable sort on the EC digits alone makes rows with ascending loci, and k_cvb_out / k_cvb_rowptr write the CSR.
// A row index outside its piece's EC range (a cut column whose lists are not ascending) is noticed: the caller then takes the
// general, sort-everything path below.  A piece with more row indices than the table takes (EC ids bunched in one range) goes
// entry by entry: the first haplotype's copy of an EC gathers the mask by binary searches in the other haplotypes' lists.
constexpr u32 CVU_PIECE = 1536;     // row indices a piece aims at
constexpr u32 CVU_MAX = 3072;       // row indices the LDS table takes
constexpr u32 CVU_TSZ = 4096;       // table slots (a power of two; small pieces use a power-of-two part of it): 32 KB, four workgroups per CU
constexpr u32 CVU_TPB = 512;
constexpr u32 CVU_K = CVU_MAX / CVU_TPB;      // row indices per thread, all loaded before the first goes into the table
constexpr u32 CVB_ERR_PTR = 1u, CVB_ERR_EC = 2u, CVB_ERR_ORDER = 4u;
__global__ void k_cvb_last(const int* cscptr, u32 n_loci, u32 n_haps, int* last) {
    const u32 h = threadIdx.x;
    if (h < n_haps) last[h] = cscptr[(u64)h * (n_loci + 1) + n_loci];
}
__global__ void k_cvu_colsum(const int* cscptr, u32 n_loci, u32 n_haps, u32* S, u32* err) {
    const u64 l = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (l > n_loci) return;
    u32 sum = 0, bad = 0;
    for (u32 h = 0; h < n_haps; ++h) {
        const int* ptr = cscptr + (u64)h * (n_loci + 1);
        const int v = ptr[l];
        if (v < 0 || (l == 0 && v != 0) || (l > 0 && ptr[l - 1] > v)) bad = CVB_ERR_PTR;
        sum += (u32)v;
    }
    S[l] = sum;
    if (bad) atomicOr(err, bad);
}
__global__ void k_cvu_pieces(const u32* S, u32 n_loci, u32* pieces) {
    const u64 l = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (l >= n_loci) return;
    const u32 n = S[l + 1] - S[l];
    pieces[l] = n <= CVU_PIECE + CVU_PIECE / 2 ? (n ? 1u : 0u) : (n + CVU_PIECE - 1u) / CVU_PIECE;
}
__global__ void k_cvu_items(const u32* pieces, const u32* pbase, u32 n_loci, u32* item_col) {
    const u64 l = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (l >= n_loci) return;
    const u32 n = pieces[l], at = pbase[l];
    for (u32 r = 0; r < n; ++r) item_col[at + r] = (u32)l;
}
// EC range of piece r of a column cut into np pieces: [r * w, (r + 1) * w), w = ceil(n_ecs / np)
__device__ __forceinline__ u32 cvu_width(u32 n_ecs, u32 np) { return (u32)(((u64)n_ecs + np - 1u) / np); }
__global__ void k_cvu_bounds(const int* cscptr, const int* cscidx, const u64* hs, const u32* pieces, const u32* pbase, const u32* item_col,
                             u32 n_items, u32 n_loci, u32 n_haps, u32 n_ecs, u32* bnd) {
    const u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t >= (u64)n_items * n_haps) return;
    const u32 i = (u32)(t / n_haps), h = (u32)(t % n_haps);
    const u32 l = item_col[i], np = pieces[l], r = i - pbase[l];
    const int* ptr = cscptr + (u64)h * (n_loci + 1);
    const u64 nh = hs[h + 1] - hs[h];
    u64 a = min((u64)(u32)ptr[l], nh), b = min((u64)(u32)ptr[l + 1], nh);
    if (r) {                                                 // first row index of the list at or above the piece's first EC
        const u64 lo = (u64)r * cvu_width(n_ecs, np);
        const int* base = cscidx + hs[h];
        while (a < b) { const u64 m = a + ((b - a) >> 1); if ((u64)(u32)base[m] < lo) a = m + 1; else b = m; }
    }
    bnd[t] = (u32)a;
}
// is EC e in haplotype hh's list of column l?  (lists ascending; a list that is not is reported and the answer discarded)
__device__ __forceinline__ bool cvb_contains(const int* cscptr, const int* cscidx, const u64* hs, u32 n_loci, u32 hh, u32 l, u32 e) {
    const int* p = cscptr + (u64)hh * (n_loci + 1);
    const u64 nh = hs[hh + 1] - hs[hh];
    u64 a = min((u64)(u32)p[l], nh), b = min((u64)(u32)p[l + 1], nh);
    const int* base = cscidx + hs[hh];
    while (a < b) { const u64 m = a + ((b - a) >> 1); const u32 v = (u32)base[m]; if (v < e) a = m + 1; else if (v > e) b = m; else return true; }
    return false;
}
template <bool EMIT>
__global__ __launch_bounds__(CVU_TPB) void k_cvu_union(const int* cscptr, const int* cscidx, const u64* hs, const u32* pieces, const u32* pbase,
                                                       const u32* item_col, const u32* bnd, u32 n_items, u32 n_loci, u32 n_haps, u32 n_ecs,
                                                       u32* colcnt, const u32* colbase, u32* colcur, u64 nnz, u64* keys, u32* vals, u32* err) {
    __shared__ u32 tkey[CVU_TSZ];
    __shared__ u32 tmask[CVU_TSZ];
    __shared__ u32 seg[64];                                  // per haplotype: first and one-past-last row index of the piece (within the haplotype)
    __shared__ u32 segp[33];                                 // ... and how many row indices the haplotypes before it bring
    __shared__ u64 shs[32];
    __shared__ u32 s_n, s_base;
    const u32 i = blockIdx.x;
    const u32 l = item_col[i], np = pieces[l], r = i - pbase[l];
    const u32 w = cvu_width(n_ecs, np);
    const u32 e_lo = (u32)min((u64)r * w, (u64)n_ecs), e_hi = r + 1u == np ? n_ecs : (u32)min((u64)(r + 1u) * w, (u64)n_ecs);
    if (threadIdx.x < n_haps) {
        const u32 h = threadIdx.x;
        const u64 h0 = hs[h], nh = hs[h + 1] - h0;
        shs[h] = h0;
        seg[2 * h] = bnd[(u64)i * n_haps + h];
        seg[2 * h + 1] = r + 1u < np ? bnd[(u64)(i + 1u) * n_haps + h] : (u32)min((u64)(u32)cscptr[(u64)h * (n_loci + 1) + l + 1], nh);
    }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (threadIdx.x == 0) { u32 run = 0; for (u32 h = 0; h < n_haps; ++h) { segp[h] = run; run += seg[2 * h + 1] - seg[2 * h]; } segp[n_haps] = run; }
    __syncthreads();
    const u32 tot = segp[n_haps];
    if (tot == 0) return;
    if (tot > CVU_MAX) {
        // entry by entry: the first haplotype's copy of an EC speaks for it
        for (u32 h = 0; h < n_haps; ++h) {
            const int* base = cscidx + hs[h];
            for (u32 j = seg[2 * h] + threadIdx.x; j < seg[2 * h + 1]; j += CVU_TPB) {
                const u32 e = (u32)base[j];
                u32 bad = e >= n_ecs ? CVB_ERR_EC : 0u;
                if (e < e_lo || e >= e_hi || (j > seg[2 * h] && (u32)base[j - 1] >= e)) bad |= CVB_ERR_ORDER;
                if (bad) { atomicOr(err, bad); continue; }
                bool first = true;
                for (u32 hh = h; hh-- > 0 && first;) first = !cvb_contains(cscptr, cscidx, hs, n_loci, hh, l, e);
                if (!first) continue;
                if (!EMIT) { atomicAdd(&colcnt[l], 1u); continue; }
                u32 mask = 1u << h;
                for (u32 hh = h + 1; hh < n_haps; ++hh) if (cvb_contains(cscptr, cscidx, hs, n_loci, hh, l, e)) mask |= 1u << hh;
                const u64 pos = (u64)colbase[l] + atomicAdd(&colcur[l], 1u);
                if (pos < nnz) { keys[pos] = ((u64)e << 32) | l; vals[pos] = mask; }
            }
        }
        return;
    }
    u32 tsz = 64;                                            // slots in use: a power of two, at least 4/3 of the row indices
    while (tsz < tot + tot / 3u + 1u) tsz <<= 1;
    // all of a thread's row indices are fetched before the first is inserted (their loads overlap), haplotype after haplotype in one index space
    u32 ev[CVU_K], hv[CVU_K];
#pragma unroll
    for (u32 k = 0; k < CVU_K; ++k) {
        const u32 u = threadIdx.x + k * CVU_TPB;
        hv[k] = 0xFFFFFFFFu; ev[k] = 0;
        if (u < tot) {
            u32 h = 0;
            while (segp[h + 1] <= u) ++h;
            hv[k] = h;
            ev[k] = (u32)cscidx[shs[h] + seg[2 * h] + (u - segp[h])];
        }
    }
    for (u32 q = threadIdx.x; q < tsz; q += CVU_TPB) { tkey[q] = 0xFFFFFFFFu; tmask[q] = 0u; }      // (while the loads are under way)
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < CVU_K; ++k) {
        if (hv[k] == 0xFFFFFFFFu) continue;
        const u32 e = ev[k];                                 // (never 0xFFFFFFFF once it is below n_ecs)
        if (e >= n_ecs) { atomicOr(err, CVB_ERR_EC); continue; }
        if (e < e_lo || e >= e_hi) { atomicOr(err, CVB_ERR_ORDER); continue; }
        u32 q = (e * 0x9E3779B1u) >> 7 & (tsz - 1u);
        for (u32 it = 0; it < tsz; ++it) {                   // (fewer keys than slots: a free one is always found)
            const u32 old = atomicCAS(&tkey[q], 0xFFFFFFFFu, e);
            if (old == 0xFFFFFFFFu || old == e) { atomicOr(&tmask[q], 1u << hv[k]); break; }
            q = (q + 1u) & (tsz - 1u);
        }
    }
    __syncthreads();
    // the distinct ECs of the piece: counted, then (EMIT) given consecutive places in the column's stretch of the output
    u32 mine = 0;
    for (u32 q = threadIdx.x; q < tsz; q += CVU_TPB) mine += tkey[q] != 0xFFFFFFFFu ? 1u : 0u;
    const u32 at = mine ? atomicAdd(&s_n, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && s_n) { if (EMIT) s_base = colbase[l] + atomicAdd(&colcur[l], s_n); else atomicAdd(&colcnt[l], s_n); }
    if (!EMIT) return;
    __syncthreads();
    u32 rnk = at;
    for (u32 q = threadIdx.x; q < tsz; q += CVU_TPB) {
        const u32 e = tkey[q];
        if (e == 0xFFFFFFFFu) continue;
        const u64 pos = (u64)s_base + rnk++;
        if (pos < nnz) { keys[pos] = ((u64)e << 32) | l; vals[pos] = tmask[q]; }
    }
}
__global__ void k_cvb_out(const u64* keys, const u32* vals, u64 nnz, int* indices, int* data) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < nnz) { indices[i] = (int)(u32)keys[i]; data[i] = (int)vals[i]; }
}
__global__ void k_cvb_rowptr(const u64* keys, u64 nnz, u32 n_ecs, int* indptr) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (r <= n_ecs) indptr[r] = (int)lower_bound_u64(keys, nnz, r << 32);
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

// ---- CSR(bitmask) -> per-haplotype CSC as a transposition of the NON-ZEROS (bin_utils.py:979-995, Sparse3DMatrix.py:189-193) ----
// The non-zeros (locus, EC, mask) are sorted by locus -- stable, so ECs stay ascending within a column -- which is the whole
// transposition: 12 M elements and the digits of the locus alone (three 8-bit passes at 80 k loci), not the 79 M set bits and the
// digits of (haplotype, locus).  A haplotype's row indices are then the ECs of the sorted non-zeros that carry its bit, in
// that order: one counting pass per block of CVB non-zeros, one scan over (haplotype, block), one pass that writes.
constexpr int CVB = 1024;
__global__ __launch_bounds__(TPB) void k_cv_bits(const int* data, u64 nnz, u64* total) {
    u64 v = 0;
    for (u64 i = blockIdx.x * (u64)TPB + threadIdx.x; i < nnz; i += (u64)gridDim.x * TPB) v += __popc((u32)data[i]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(total, v);
}
// one thread per row: key = locus << 32 | EC, value = mask; what a .bin of unknown origin may hold is checked here
__global__ void k_cv_keys(const int* indptr, u32 n_ecs, const int* indices, const int* data, u64 nnz, u32 n_loci, u32 n_haps,
                          u64* keys, u32* vals, u32* err) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n_ecs) return;
    const long long a = indptr[e], b = indptr[e + 1];
    if (a < 0 || b < a || (u64)b > nnz || (e == 0 && a != 0)) { atomicOr(err, 1u); return; }
    bool bad = false;
    for (long long i = a; i < b; ++i) {
        const u32 lc = (u32)indices[i], m = (u32)data[i];
        bad |= lc >= n_loci || m == 0u || (m >> n_haps) != 0u;
        keys[i] = ((u64)lc << 32) | (u32)e;
        vals[i] = m;
    }
    if (bad) atomicOr(err, 2u);
}
__global__ __launch_bounds__(TPB) void k_cv_cnt(const u32* masks, u64 nnz, u32 n_haps, u32 nb, u32* blk) {
    __shared__ u32 cnt[32];
    if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
    __syncthreads();
    const u64 i0 = (u64)blockIdx.x * CVB;
#pragma unroll
    for (int r = 0; r < CVB / TPB; ++r) {
        const u64 i = i0 + (u64)r * TPB + threadIdx.x;
        const u32 m = i < nnz ? masks[i] : 0u;
        for (u32 h = 0; h < n_haps; ++h) {
            const u64 bm = __ballot((m >> h) & 1u);
            if ((threadIdx.x & 63u) == 0 && bm) atomicAdd(&cnt[h], (u32)__popcll(bm));
        }
    }
    __syncthreads();
    if (threadIdx.x < n_haps) blk[(u64)threadIdx.x * nb + blockIdx.x] = cnt[threadIdx.x];
}
// scan[h * nb + b]: set bits of haplotype h before block b, counted on from the haplotypes before it -- the place of the block's
// first such bit in the row-index array.  headval[h * n_loci + t]: where column t of haplotype h starts within h's own block.
__global__ __launch_bounds__(TPB) void k_cv_emit(const u64* keys, const u32* masks, u64 nnz, u32 n_haps, u32 n_loci, u32 nb,
                                                 const u32* scan, int* cscidx, u32* headval) {
    __shared__ u32 run[32], wcnt[TPB / 64][32];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid < 32) run[tid] = tid < n_haps ? scan[(u64)tid * nb + blockIdx.x] : 0u;
    const u64 lt = (1ull << lane) - 1ull;
    const u64 i0 = (u64)blockIdx.x * CVB;
    for (int r = 0; r < CVB / TPB; ++r) {
        const u64 i = i0 + (u64)r * TPB + tid;
        const bool have = i < nnz;
        const u64 key = have ? keys[i] : 0ull;
        const u32 m = have ? masks[i] : 0u;
        const u32 lc = (u32)(key >> 32);
        const bool head = have && (i == 0 || (u32)(keys[i - 1] >> 32) != lc);
        for (u32 h = 0; h < n_haps; ++h) {
            const u64 bm = __ballot((m >> h) & 1u);
            if (lane == 0) wcnt[w][h] = (u32)__popcll(bm);
        }
        __syncthreads();
        for (u32 h = 0; h < n_haps; ++h) {
            const u64 bm = __ballot((m >> h) & 1u);
            u32 at = run[h] + (u32)__popcll(bm & lt);
            for (u32 k = 0; k < w; ++k) at += wcnt[k][h];
            if ((m >> h) & 1u) cscidx[at] = (int)(u32)key;
            if (head) headval[(u64)h * n_loci + lc] = at - scan[(u64)h * nb];
        }
        __syncthreads();
        if (tid < n_haps) { u32 c = 0; for (u32 k = 0; k < TPB / 64; ++k) c += wcnt[k][tid]; run[tid] += c; }
        __syncthreads();
    }
}
// column pointers: column t of haplotype h starts where the first non-empty column at or after t does
__global__ void k_cv_ptr(const u64* keys, u64 nnz, u32 n_loci, u32 n_haps, u32 nb, const u32* scan, const u64* grand, const u32* headval, int* cscptr) {
    const u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t > n_loci) return;
    const u64 at = lower_bound_u64(keys, nnz, t << 32);
    const u32 t2 = at < nnz ? (u32)(keys[at] >> 32) : 0u;
    for (u32 h = 0; h < n_haps; ++h) {
        const u32 tot = (h + 1 < n_haps ? scan[(u64)(h + 1) * nb] : (u32)*grand) - scan[(u64)h * nb];
        cscptr[(u64)h * (n_loci + 1) + t] = (int)(at < nnz ? headval[(u64)h * n_loci + t2] : tot);
    }
}

__global__ void k_iota(int* out, u64 n) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)i;
}
__global__ void k_read_ec(const u32* read_slot, u64 n, const u32* rank_of_slot, int* out) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = read_slot[i] == 0xFFFFFFFFu ? -1 : (int)rank_of_slot[read_slot[i]];
}
__global__ void k_range_len(const int2* rng, u64 n, long long* out) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { const int2 r = rng[i]; out[i] = r.y == INT_MIN ? 0ll : (long long)r.y - (long long)r.x + 1ll; }
}
__global__ void k_fill_minmax(int2* p, u64 n) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_int2(INT_MAX, INT_MIN);
}
__global__ void k_split_minmax(const int2* rng, u64 n, int* mn, int* mx) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { const int2 r = rng[i]; mn[i] = r.x; mx[i] = r.y; }
}
// the used stretches of the key arena, back to zero (ecb_reset): blockIdx.y = stretch
struct ArenaClear { u64 start[ARENA_REGIONS + 1], len[ARENA_REGIONS + 1]; };
__global__ void k_clear_arena(uint2* arena, ArenaClear A) {
    const u64 s = A.start[blockIdx.y], n = A.len[blockIdx.y];
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) arena[s + i] = make_uint2(0u, 0u);
}
__global__ void k_clear_slots(Slot* table, const u32* list, u64 n) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4* s = reinterpret_cast<uint4*>(table + list[i]);
    const uint4 z = make_uint4(0, 0, 0, 0);
    s[0] = z; s[1] = z; s[2] = z; s[3] = z;
}

// ---------------------------------------------------------------------------------------------
// LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass, stable: the sort behind the multisample triples
// (bam_utils_multisample.py:503-576, 737-791) and the sparse-format conversions (bin_utils.py:979-1028).
// A pass = k_rs_hist (LDS histogram per 4 096-key tile) -> exclusive scan over (digit, tile) -> k_rs_scatter.  The scatter
// ranks keys WITHOUT reordering equal digits: every wave owns a contiguous quarter of the tile and walks it 64 keys at a
// time; lanes with the same digit find each other with eight ballots (one per digit bit), the lowest of them bumps the
// wave's LDS counter of that digit by the group's size, and a key's rank is that counter value plus the number of group
// members in lower lanes.  Counters of the four waves are added up in wave order afterwards.
// ---------------------------------------------------------------------------------------------
#ifndef ECB_RS_TPB
#define ECB_RS_TPB 1024
#endif
constexpr int RS_TPB = ECB_RS_TPB, RS_TILE = 4096, RS_ITEMS = RS_TILE / RS_TPB;
static_assert(RS_TPB >= 256 && RS_TPB % 64 == 0 && RS_TILE % RS_TPB == 0, "one thread per digit among the first 256");
constexpr int RS_MAX_PASSES = 8;
// Round 3: a pass is ONE kernel.  The digit histograms of all passes are taken in one sweep up front (they do not depend on the
// order of the keys), which gives every pass the first place of each digit; where a TILE's keys of a digit go within that is
// found while the pass runs, by decoupled look-back: a workgroup takes the next tile (a counter: tiles start in order), publishes
// its per-digit counts (AGG | count), walks back over the tiles before it adding up their counts until it meets one whose inclusive
// prefix is known (PFX | prefix), and publishes its own.  One 4-byte word per (tile, digit) carries status and value, written and
// read with relaxed agent-scope atomics (write-through / L2-bypassing on gfx950: one granule, nothing else to order).  A tile only
// ever waits for tiles that were started before it, so the walk ends; the polls are bounded all the same.
// (Before: a histogram kernel, three scan kernels over (digit, workgroup) counters and the scatter per pass -- a third of a pass.)
constexpr u32 RS_AGG = 1u << 30, RS_PFX = 2u << 30, RS_VAL = (1u << 30) - 1u;
constexpr u32 RS_SPIN_MAX = 1u << 24;
#ifndef ECB_RS_LOOK
#define ECB_RS_LOOK 4
#endif
constexpr int RS_LOOK = ECB_RS_LOOK;          // tiles looked back at per round trip
struct RsShifts { u32 s[RS_MAX_PASSES]; u32 n; };
__global__ __launch_bounds__(RS_TPB) void k_rs_hist_all(const u64* keys, u64 n, RsShifts sh, u32* ghist) {
    __shared__ u32 h[RS_MAX_PASSES][256];
    for (u32 q = threadIdx.x; q < RS_MAX_PASSES * 256; q += RS_TPB) (&h[0][0])[q] = 0;
    __syncthreads();
    for (u64 i = blockIdx.x * (u64)RS_TPB + threadIdx.x; i < n; i += (u64)gridDim.x * RS_TPB) {
        const u64 k = keys[i];
        for (u32 p = 0; p < sh.n; ++p) atomicAdd(&h[p][(u32)(k >> sh.s[p]) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256u) for (u32 p = 0; p < sh.n; ++p) { const u32 c = h[p][threadIdx.x]; if (c) atomicAdd(&ghist[p * 256 + threadIdx.x], c); }
}
__global__ __launch_bounds__(256) void k_rs_bases(const u32* ghist, u32 n_passes, u32* base) {      // exclusive scan of every pass's 256 counts
    __shared__ u32 s_w[4];
    for (u32 p = 0; p < n_passes; ++p) {
        const u32 v = ghist[p * 256 + threadIdx.x];
        const u32 incl = wave_incl_scan(v);
        if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        u32 before = 0;
        for (u32 k = 0; k < (threadIdx.x >> 6); ++k) before += s_w[k];
        base[p * 256 + threadIdx.x] = before + incl - v;
        __syncthreads();
    }
}
__global__ __launch_bounds__(RS_TPB) void k_rs_pass(const u64* kin, const u32* vin, u64 n, u32 shift, const u32* base, u32* desc,
                                                    u32* ticket, u32* err, u64* kout, u32* vout) {
    // A tile is laid out digit by digit in LDS first and leaves from there: consecutive lanes then write consecutive
    // addresses of one digit's run (16 elements on average) instead of 64 lanes writing to 64 different runs.
    __shared__ u32 cnt[RS_TPB / 64][256];          // per wave: keys of each digit; then: keys of that digit in the waves before
    __shared__ u32 dstart[256];                    // first place of digit d within the tile
    __shared__ u32 gbase[256];                     // where the tile's first key of digit d goes
    __shared__ u32 s_wsum[RS_TPB / 64];
    __shared__ u32 s_tile;
    __shared__ u64 skey[RS_TILE];
    __shared__ u32 sval[RS_TILE];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (u32 q = tid; q < (RS_TPB / 64) * 256; q += RS_TPB) (&cnt[0][0])[q] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    const u64 t0 = (u64)tile * RS_TILE;
    if (t0 >= n) return;
    const u64 w0 = t0 + (u64)w * (RS_TILE / (RS_TPB / 64));
    const u64 lt = (1ull << lane) - 1ull;
    u64 key[RS_ITEMS];
    u32 val[RS_ITEMS], rk[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u64 i = w0 + (u64)r * 64 + lane;
        const bool have = i < n;
        key[r] = have ? kin[i] : 0ull;
        val[r] = have ? vin[i] : 0u;
        const u32 d = (u32)(key[r] >> shift) & 255u;
        u64 peers = __ballot(have);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const u64 m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const u32 below = (u32)__popcll(peers & lt);
        u32 bs = 0;
        if (have && below == 0u) bs = atomicAdd(&cnt[w][d], (u32)__popcll(peers));
        bs = __shfl(bs, have ? __ffsll((long long)peers) - 1 : (int)lane);
        rk[r] = bs + below;
    }
    __syncthreads();
    u32 tot = 0;                                   // digit tid: its keys in the tile, and -- in place of the per-wave counts -- the keys of the waves before
    const bool dig = tid < 256u;                   // (threads beyond the 256 digits only rank and move keys)
    const u32 dt = dig ? tid : 0u;
    if (dig) {
#pragma unroll
        for (int k = 0; k < RS_TPB / 64; ++k) { const u32 c = cnt[k][tid]; cnt[k][tid] = tot; tot += c; }
    }
    // publish the tile's count of digit tid, look back for the keys of that digit in the tiles before, publish the prefix
    u32* const mine = desc + (u64)tile * 256 + dt;
    u32 excl = 0;
    if (!dig) {
    } else if (tile == 0) {
        __hip_atomic_store(mine, RS_PFX | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(mine, RS_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32 spins = 0;
        bool done = false;
        for (long long t = (long long)tile - 1; t >= 0 && !done;) {
            // four tiles back per round trip (their words are independent loads); consumed in order, up to the first that has nothing yet
            u32 v[RS_LOOK];
#pragma unroll
            for (int k = 0; k < RS_LOOK; ++k)
                v[k] = t - k >= 0 ? __hip_atomic_load(desc + (u64)(t - k) * 256 + dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            int used = 0;
#pragma unroll
            for (int k = 0; k < RS_LOOK; ++k) {
                if (done || used != k || t - k < 0) continue;
                if (v[k] < RS_AGG) continue;         // nothing published there yet: this one again in the next round
                excl += v[k] & RS_VAL;
                used = k + 1;
                if (v[k] >= RS_PFX) done = true;
            }
            t -= used;
            if (!used) {
                if (++spins > RS_SPIN_MAX) { atomicOr(err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __hip_atomic_store(mine, RS_PFX | ((excl + tot) & RS_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dig) gbase[tid] = base[tid] + excl;
    {
        const u32 incl = wave_incl_scan(tot);
        if (lane == 63) s_wsum[w] = incl;          // (waves beyond the fourth: zeros)
        __syncthreads();
        u32 before = 0;
        for (u32 k = 0; k < w; ++k) before += s_wsum[k];
        if (dig) dstart[tid] = before + incl - tot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u64 i = w0 + (u64)r * 64 + lane;
        if (i < n) {
            const u32 d = (u32)(key[r] >> shift) & 255u;
            const u32 lp = dstart[d] + cnt[w][d] + rk[r];
            skey[lp] = key[r]; sval[lp] = val[r];
        }
    }
    __syncthreads();
    const u32 n_tile = (u32)min((u64)RS_TILE, n - t0);
    for (u32 i = tid; i < n_tile; i += RS_TPB) {
        const u64 k = skey[i];
        const u32 d = (u32)(k >> shift) & 255u;
        const u64 pos = (u64)gbase[d] + (i - dstart[d]);
        if (pos < n) { kout[pos] = k; vout[pos] = sval[i]; }      // (always, unless a look-back gave up: reported through *err)
    }
}
__global__ __launch_bounds__(TPB) void k_or_reduce(const u64* keys, u64 n, u64* out) {
    u64 v = 0;
    for (u64 i = blockIdx.x * (u64)TPB + threadIdx.x; i < n; i += (u64)gridDim.x * TPB) v |= keys[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d);
    if ((threadIdx.x & 63u) == 0 && v) atomicOr(out, v);
}
// ---------------------------------------------------------------------------------------------
// Multisample K4' / K5' (bam_utils_multisample.py:503-636, 737-791) from the (EC, cell, file) triples: cell order = insertion
// order of the reference's cr_totals (files in order; within a file ECs by first appearance there; within an EC cells by
// first appearance), per-cell totals, minimum-count filter, EC re-rank, CSC N, rows of A that survive.
// ---------------------------------------------------------------------------------------------
// Per cell: its reads, and the first file it has reads in (the reference walks the files in order: that file decides where the
// cell enters cr_totals).  Counters and minima are kept in LDS per workgroup and flushed once (n_cells <= MSF_LDS_CELLS); more
// cells than that go to memory directly.
constexpr u32 MSF_LDS_CELLS = 8192;
constexpr int TPB_MSF = 1024;
__global__ __launch_bounds__(TPB_MSF) void k_msf2_cells(const u32* meta, const u32* cnt, u64 n, u32 n_cells, u64* total, u32* firstfile, u32* err) {
    extern __shared__ u32 sh32[];                            // lfile[n_cells] | lcnt[n_cells]
    const bool lds = n_cells <= MSF_LDS_CELLS;
    u32 *lfile = sh32, *lcnt = sh32 + (lds ? n_cells : 0);
    if (lds) {
        for (u32 c = threadIdx.x; c < n_cells; c += TPB_MSF) { lfile[c] = 0xFFFFFFFFu; lcnt[c] = 0; }
        __syncthreads();
    }
    const u64 per = (n + gridDim.x - 1) / gridDim.x, t0 = blockIdx.x * per, t1 = min(t0 + per, n);
    for (u64 t = t0 + threadIdx.x; t < t1; t += TPB_MSF) {
        const u32 m = meta[t], c = m & ((1u << ECB_CELL_BITS) - 1u), f = m >> ECB_CELL_BITS;
        if (c >= n_cells) { atomicOr(err, 2u); continue; }   // (cells are the ids the host handed out, 0 .. n_cells - 1)
        if (lds) { atomicAdd(&lcnt[c], cnt[t]); atomicMin(&lfile[c], f); }
        else { atomicAdd(&total[c], (u64)cnt[t]); atomicMin(&firstfile[c], f); }
    }
    if (!lds) return;
    __syncthreads();
    for (u32 c = threadIdx.x; c < n_cells; c += TPB_MSF)
        if (lfile[c] != 0xFFFFFFFFu) { atomicAdd(&total[c], (u64)lcnt[c]); atomicMin(&firstfile[c], lfile[c]); }
}
// seg[e] = first triple of EC e (the triples are sorted by EC; an EC without triples gets an empty stretch), seg[n_ecs] = n
__global__ void k_msf2_seg(const u32* ec, u64 n, u32 n_ecs, u32* seg, u32* err) {
    const u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 e = ec[t];
    if (e >= n_ecs) { atomicOr(err, 1u); return; }
    const long long ep = t ? (long long)ec[t - 1] : -1ll;
    for (long long x = ep + 1; x <= (long long)e; ++x) seg[x] = (u32)t;
    if (t == n - 1) for (u64 x = (u64)e + 1; x <= n_ecs; ++x) seg[x] = (u32)n;
}
// Per EC: (i) the EC's first appearance in every file it has reads in -- the smallest first read among its triples of that
// file; (ii) every triple whose file is its cell's first file offers the cell (first appearance of the EC in that file, first
// read of the triple) -- the smallest offer is where the cell enters the reference's cr_totals (bam_utils_multisample.py:
// 513-546); (iii) whether any of the EC's cells survives the minimum count.
// One THREAD per EC of up to MSF_SMALL triples (the run of the mill: two dozen triples; the few triples that make an offer look
// their file's first appearance up by walking the EC's stretch again -- neighbours' stretches follow each other in memory), one
// WORKGROUP per larger EC, queued by the first kernel, with the per-file table in LDS.  (One wave per EC with three dependent
// sweeps was latency: 14 us per EC.)
constexpr u32 MSF_SMALL = 256;
constexpr u32 MSF_FILES = 1u << MS_FILE_BITS;
constexpr u32 MSF_GIANT = 1u << 15;           // triples above which an EC is shared out over the whole grid (k_msf2_giant_*), not given to one workgroup
__global__ void k_msf2_ecs_small(const u32* meta, const u32* first, const u32* seg, u32 n_ecs, const u64* total, const u32* firstfile,
                                 u64 min_count, u64* cellkey, u32* keep_ec, u32* big, u32* n_big, u32* giant, u32* n_giant) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n_ecs) return;
    const u32 a = seg[e], b = seg[e + 1];
    if (b - a > MSF_GIANT) { giant[atomicAdd(n_giant, 1u)] = (u32)e; return; }
    if (b - a > MSF_SMALL) { big[atomicAdd(n_big, 1u)] = (u32)e; return; }
    bool keep = false;
    u32 last_f = 0xFFFFFFFFu, last_fec = 0;                                  // (the EC's first appearance in the file asked for last: cells' first files repeat)
    for (u32 t = a; t < b; ++t) {
        const u32 m = meta[t], c = m & ((1u << ECB_CELL_BITS) - 1u), f = m >> ECB_CELL_BITS;
        keep |= total[c] >= min_count;
        if (firstfile[c] != f) continue;
        if (f != last_f) {
            u32 fec = first[t];
            for (u32 x = a; x < b; ++x) if ((meta[x] >> ECB_CELL_BITS) == f) fec = min(fec, first[x]);
            last_f = f; last_fec = fec;
        }
        const u32 fec = last_fec;
        const u64 offer = ((u64)fec << 32) | first[t];
        if (offer < cellkey[c]) atomicMin(&cellkey[c], offer);               // (the plain read may be stale: then one atomic too many)
    }
    if (keep) keep_ec[e] = 1u;
}
__global__ __launch_bounds__(TPB) void k_msf2_ecs_big(const u32* meta, const u32* first, const u32* seg, const u32* big, const u32* n_big,
                                                      const u64* total, const u32* firstfile, u64 min_count, u64* cellkey, u32* keep_ec) {
    __shared__ u32 fec[MSF_FILES];
    __shared__ u32 s_keep;
    for (u32 q = blockIdx.x; q < *n_big; q += gridDim.x) {
        const u32 e = big[q], a = seg[e], b = seg[e + 1];
        for (u32 f = threadIdx.x; f < MSF_FILES; f += TPB) fec[f] = 0xFFFFFFFFu;
        if (threadIdx.x == 0) s_keep = 0;
        __syncthreads();
        bool keep = false;
        for (u32 t = a + threadIdx.x; t < b; t += TPB) {
            const u32 m = meta[t];
            atomicMin(&fec[m >> ECB_CELL_BITS], first[t]);
            keep |= total[m & ((1u << ECB_CELL_BITS) - 1u)] >= min_count;
        }
        if (keep) s_keep = 1u;
        __syncthreads();
        for (u32 t = a + threadIdx.x; t < b; t += TPB) {
            const u32 m = meta[t], c = m & ((1u << ECB_CELL_BITS) - 1u), f = m >> ECB_CELL_BITS;
            if (firstfile[c] == f) {
                const u64 offer = ((u64)fec[f] << 32) | first[t];
                if (offer < cellkey[c]) atomicMin(&cellkey[c], offer);
            }
        }
        if (threadIdx.x == 0 && s_keep) keep_ec[e] = 1u;
        __syncthreads();
    }
}
// The few ECs with tens of thousands of triples and more (config 4's most popular EC has millions): one workgroup each was the tail of the
// whole filter.  Every workgroup of the grid takes a strided share of each such EC: first the EC's first appearance per file (LDS, then one
// global atomic per file and workgroup), then -- a launch later -- the offers.
__global__ __launch_bounds__(TPB) void k_msf2_giant_fec(const u32* meta, const u32* first, const u32* seg, const u32* giant, const u32* n_giant,
                                                        const u64* total, u64 min_count, u32* gfec, u32* keep_ec) {
    __shared__ u32 fec[MSF_FILES];
    __shared__ u32 s_keep;
    for (u32 g = 0; g < *n_giant; ++g) {
        const u32 e = giant[g], a = seg[e], b = seg[e + 1];
        for (u32 f = threadIdx.x; f < MSF_FILES; f += TPB) fec[f] = 0xFFFFFFFFu;
        if (threadIdx.x == 0) s_keep = 0;
        __syncthreads();
        bool keep = false;
        for (u64 t = (u64)a + (u64)blockIdx.x * TPB + threadIdx.x; t < b; t += (u64)gridDim.x * TPB) {
            const u32 m = meta[t];
            atomicMin(&fec[m >> ECB_CELL_BITS], first[t]);
            keep |= total[m & ((1u << ECB_CELL_BITS) - 1u)] >= min_count;
        }
        if (keep) s_keep = 1u;
        __syncthreads();
        for (u32 f = threadIdx.x; f < MSF_FILES; f += TPB) if (fec[f] != 0xFFFFFFFFu) atomicMin(&gfec[(u64)g * MSF_FILES + f], fec[f]);
        if (threadIdx.x == 0 && s_keep) keep_ec[e] = 1u;
        __syncthreads();
    }
}
__global__ __launch_bounds__(TPB) void k_msf2_giant_offer(const u32* meta, const u32* first, const u32* seg, const u32* giant, const u32* n_giant,
                                                          const u32* firstfile, const u32* gfec, u64* cellkey) {
    for (u32 g = 0; g < *n_giant; ++g) {
        const u32 e = giant[g], a = seg[e], b = seg[e + 1];
        for (u64 t = (u64)a + (u64)blockIdx.x * TPB + threadIdx.x; t < b; t += (u64)gridDim.x * TPB) {
            const u32 m = meta[t], c = m & ((1u << ECB_CELL_BITS) - 1u), f = m >> ECB_CELL_BITS;
            if (firstfile[c] == f) {
                const u64 offer = ((u64)gfec[(u64)g * MSF_FILES + f] << 32) | first[t];
                if (offer < cellkey[c]) atomicMin(&cellkey[c], offer);
            }
        }
    }
}
// cells that have reads, as the lists the ordering below works on
__global__ void k_msf2_cellflag(const u64* total, u32 n_cells, u32* flag) {
    const u64 c = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (c < n_cells) flag[c] = total[c] ? 1u : 0u;
}
__global__ void k_msf2_celllist(const u64* total, const u32* firstfile, const u64* cellkey, const u32* flag, const u32* pos, u32 n_cells,
                                u32* cell_id, u64* ctotal, u64* bhi, u64* blo) {
    const u64 c = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (c >= n_cells || !flag[c]) return;
    const u32 o = pos[c];
    cell_id[o] = (u32)c; ctotal[o] = total[c]; bhi[o] = firstfile[c]; blo[o] = cellkey[c];
}
// (EC, cell) pairs that survive: a pair = the triples of one EC and one cell, which follow each other (one per file)
__global__ void k_msf2_pairflag(const u32* ec, const u32* meta, u64 n, const u32* new_cell, u32* flag) {
    const u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 c = meta[t] & ((1u << ECB_CELL_BITS) - 1u);
    const bool head = t == 0 || ec[t - 1] != ec[t] || (meta[t - 1] & ((1u << ECB_CELL_BITS) - 1u)) != c;
    flag[t] = (head && new_cell[c] != 0xFFFFFFFFu) ? 1u : 0u;
}
__global__ void k_msf2_pairs(const u32* ec, const u32* meta, const u32* cnt, const u32* flag, const u32* pos, u64 n, const u32* new_cell,
                             const u32* new_rank, u64* key, u32* val) {
    const u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t >= n || !flag[t]) return;
    const u32 e = ec[t], c = meta[t] & ((1u << ECB_CELL_BITS) - 1u);
    u32 sum = 0;
    for (u64 x = t; x < n && ec[x] == e && (meta[x] & ((1u << ECB_CELL_BITS) - 1u)) == c; ++x) sum += cnt[x];   // (the same cell's reads of one EC in several files add up, :737-791)
    key[pos[t]] = ((u64)new_cell[c] << 32) | new_rank[e];
    val[pos[t]] = sum;
}
__global__ void k_msf2_nout(const u64* key, const u32* val, u64 n, int* indices, int* data) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { indices[i] = (int)(u32)key[i]; data[i] = (int)val[i]; }
}
__global__ void k_msf2_nptr(const u64* key, u64 n, u32 n_cells, int* indptr) {
    const u64 c = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (c <= n_cells) indptr[c] = (int)lower_bound_u64(key, n, c << 32);
}
__global__ void k_msf_iota(u32* v, u64 n) { const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; if (i < n) v[i] = (u32)i; }
__global__ void k_msf_gather64(const u64* src, const u32* idx, u64 n, u64* dst) { const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; if (i < n) dst[i] = src[idx[i]]; }
__global__ void k_msf_keepflag(const u32* order, const u64* total, u64 n, u64 min_count, u32* flag) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) flag[i] = total[order[i]] >= min_count ? 1u : 0u;
}
__global__ void k_msf_newcell(const u32* order, const u32* flag, const u32* pos, const u32* cell_id, u64 n, u32* new_cell, u32* kept_cells) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n && flag[i]) { const u32 c = cell_id[order[i]]; new_cell[c] = pos[i]; kept_cells[pos[i]] = c; }
}
__global__ void k_msf_rowlen(const u32* indptr, const u32* keep_ec, const u32* new_rank, u64 n_ecs, u32* rowlen2) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e < n_ecs && keep_ec[e]) rowlen2[new_rank[e]] = indptr[e + 1] - indptr[e];
}
__global__ void k_msf_rows(const u32* indptr, const int* indices, const int* data, const u32* keep_ec, const u32* new_rank,
                           const u32* indptr2, u64 n_ecs, int* indices2, int* data2) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n_ecs || !keep_ec[e]) return;
    const u32 a = indptr[e], b = indptr[e + 1], o = indptr2[new_rank[e]];
    for (u32 i = a; i < b; ++i) { indices2[o + (i - a)] = indices[i]; data2[o + (i - a)] = data[i]; }
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
    bool scatter_attr_set = false, count_attr_set = false;
    bool list_from_counts = false;             // finalize: k_count_bins listed the occupied slots (no k_compact pass)
    bool assembled = false;                    // the result was put together from per-range results (ecb_assemble_ranges_device)
    bool list_counted = false;                 // ... and so it did for a table export (the list alone; its length at *d_list_n)
    u64* d_list_n = nullptr;
    u64 resident_blocks = 0, resident_blocks_rg = 0, resident_blocks_sh = 0, rounds = 24, min_tiles = 32;     // k_stream's launch shape (queried once)
    bool short_reads = false;         // this batch goes through ks_short::k_stream (set per batch by process_batch)
    bool par_stream = false;          // ... through ks_par::k_stream: the batch before met loci that displace each other in the LDS table (sticky per handle)
    u64 resident_blocks_par = 0;
    u64 records_pushed = 0;           // records of the batches so far (with n_reads: how many records a read brings)
    bool ctr_synced = false;          // hctr is what the device holds (no kernel that counts has been queued since the last read-back)
    bool adopted = false;             // the table holds adopted entries in consecutive slots (no hashing): finalize / export only

    Slot* table = nullptr; u64 cap = 0;
    uint2* arena = nullptr; u64 arena_cap = 0;
    Counters* ctr = nullptr;
    Counters hctr{};                  // last read-back
    Counters* pin_ctr = nullptr;      // pinned staging for clear_counters
    struct PinOut { Counters c; u64 tot[8]; };
    PinOut* pin_out = nullptr;        // ... and where the device's counters and finalize's totals land (a copy into pageable memory is staged by the runtime, behind a wait of its own)
    StreamCold *d_cold = nullptr, *pin_cold = nullptr;   // k_stream's rarely used arguments (device copy, pinned staging)
    u32* read_slot = nullptr; u64 read_slot_cap = 0;
    u32* meta = nullptr; u64 meta_cap = 0, meta_hi = 0;   // multisample: cell | file << 22 per read
    u64 n_triples = 0; u64* ms_okey = nullptr; u32 *ms_ofirst = nullptr, *ms_ostart = nullptr, *ms_ocount = nullptr;
    bool ms_adopted = false;          // the triples came from ecb_ms_adopt_triples_device (multi-GPU)
    // ecb_ms_filter's results (device): kept cells in sample order, rows of A of the kept ECs, CSC N
    bool ms_filtered = false; ecb_ms_sizes msf{};
    u32* f_cells = nullptr; int *f_ipa = nullptr, *f_ixa = nullptr, *f_daa = nullptr, *f_ipn = nullptr, *f_ixn = nullptr, *f_dan = nullptr;
    int2* rng = nullptr;                       // ECB_F_RANGES: {min, max} of reference_start per (locus, haplotype)
    u64* queue = nullptr; u64 queue_cap = 0;
    u64 n_ecs() const { return hctr.n_ecs; }
    u32 prev_rid = 0xFFFFFFFFu;       // read_id of the last record pushed so far
    u64 n_reads = 0;
    u64 reads_hi = 0;                 // read_slot entries [0, reads_hi) may be set (n_reads, or more mid-batch)
    u64 reads_hint = 0;               // ecb_hint_reads: the stream holds at most this many reads (0 = not said)
    u64 extra_all = 0, extra_valid = 0, extra_reads = 0;   // counters merged in from other ranks
    u64 n_mismatch = 0;               // ECB_F_VERIFY: reads the exactness pass found in a wrong EC, over all pushes

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
           P_MS_FLAG, P_MS_POS, P_MS_OKEY, P_MS_OFIRST, P_MS_OSTART, P_MS_X, P_MS_OCOUNT, P_MS_GRANK, P_MS_CIN, P_MS_FIN, P_LISTFN,
           P_REMAP, P_TOTALS, P_SLOW_LEN, P_SLOW_OFF, P_SLOW_NRE, P_SLOW_REQ, P_SLOW_REQ2, P_QCOMPACT, P_SLOW_KEY, P_SLOW_MASK, P_BIG, P_RS_HIST, P_RS_OFFS, P_RS_SUMS,
           P_F_CELLS, P_F_IPA, P_F_IXA, P_F_DAA, P_F_IPN, P_F_IXN, P_F_DAN, P_EXPORT, P_N };
    void* pool[P_N] = {}; u64 pool_bytes[P_N] = {};

    // profiling
    bool prof = false; double prof_ms = 0; u64 prof_launches = 0, prof_records = 0;
    const char* last_kernel = "";     // which compilation of the stream kernel the last batch launched (ecb_profile_kernel)
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
struct Scratch {                       // frees what it allocated
    std::vector<void*> p;
    template <class T> T* get(u64 n) { void* q = nullptr; if (hipMalloc(&q, std::max<u64>(n, 1) * sizeof(T)) != hipSuccess) return nullptr; p.push_back(q); return (T*)q; }
    ~Scratch() { for (void* q : p) hipFree(q); }
};
#define POOL(h, id, ptr, count) do { int rc_ = pool_get(h, ecb_handle::id, count, &(ptr)); if (rc_ != ECB_OK) return rc_; } while (0)

// zero the device counters and point every arena region's cursor at its first pair
int clear_counters(ecb_handle* h) {
    Counters c{};
    const u32 R = arena_regions(h->arena_cap);
    const u64 per = h->arena_cap / R;
    for (u32 r = 0; r < ARENA_REGIONS; ++r) c.arena_reg[r] = r < R ? r * per : h->arena_cap;
    h->hctr = c;
    if (h->pin_ctr) {                               // pinned staging: queued behind the stream's work, no wait (rewritten only by the next reset,
        *h->pin_ctr = c;                            //  which comes after the waits of a push and a finalize)
        HIPCHK(h, hipMemcpyAsync(h->ctr, h->pin_ctr, sizeof(Counters), hipMemcpyHostToDevice, h->stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(h->ctr, &h->hctr, sizeof(Counters), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->ctr_synced = true;
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
    HIPCHK(h, hipMemcpyAsync(&h->pin_out->c, h->ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->hctr = h->pin_out->c;
    h->ctr_synced = true;
    if (h->hctr.err & ERR_CONTRACT) return fail(h, ECB_ERR_CONTRACT, "read_id run counter violates the tuple contract (see ecb.h)");
    if (h->hctr.err & ERR_RANGE) return fail(h, ECB_ERR_CONTRACT, "locus or haplotype index out of range in a valid record");
    if (h->hctr.err & ERR_ARENA) return fail(h, ECB_ERR_TABLE_FULL, "EC key arena exhausted (%llu pairs): raise arena_capacity", (unsigned long long)h->arena_cap);
    if (h->hctr.err & ERR_QUEUE) return fail(h, ECB_ERR_TABLE_FULL, "deferred-read queue exhausted");
    if (h->hctr.err & ERR_INTERNAL) return fail(h, ECB_ERR_HIP, "internal: an EC-table slot was claimed but its key never published");
    return ECB_OK;
}

int grow_table(ecb_handle* h, u64 new_cap) {
    Slot* nt = nullptr;
    u32* remap = nullptr;
    if (new_cap > (1ull << 32)) return fail(h, ECB_ERR_LIMIT, "EC table beyond 2^32 slots");
    POOL(h, P_REMAP, remap, h->cap);
    HIPCHK(h, hipMalloc(&nt, new_cap * sizeof(Slot)));
    hipError_t e = hipMemsetAsync(nt, 0, new_cap * sizeof(Slot), h->stream);
    // (slots the old table did not hold map to PENDING: read_slot may hold slot ids of an EARLIER stream beyond the reads processed so far --
    //  ecb_reset leaves them, ecb_hint_reads makes reads_hi run ahead of the stream -- and what such an entry maps to must not be pool garbage)
    if (e == hipSuccess) e = hipMemsetAsync(remap, 0xFF, h->cap * sizeof(u32), h->stream);
    if (e == hipSuccess) {
        k_rehash<<<2048, TPB, 0, h->stream>>>(h->table, h->cap, nt, new_cap - 1, remap);
        if (h->reads_hi)
            k_remap_read_slot<<<2048, TPB, 0, h->stream>>>(h->read_slot, h->reads_hi, remap);
        e = hipStreamSynchronize(h->stream);
    }
    if (e != hipSuccess) { hipFree(nt); return fail(h, ECB_ERR_HIP, "grow_table: %s", hipGetErrorString(e)); }
    HIPCHK(h, hipFree(h->table));
    h->table = nt; h->cap = new_cap; h->list_counted = false;     // (slot ids changed)
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

// deferred reads: measure, scratch, k_slow; grow the table and repeat while reads bounce.  All scratch lives in the
// handle's pool (a stream of long reads pays no hipMalloc per round, and no early return can leak it).
// verify: the exactness pass over the same reads (compare with the EC each was given; no insert, one round).
int run_slow(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, u64 n, u64* d_q, u64 nq, bool verify = false, u32 tw = 512u) {
    int rc = ECB_OK;
    int flip = 0;
    while (nq) {
        u64 *d_len = nullptr, *d_off = nullptr, *d_nre = nullptr, *nre_buf = nullptr;
        POOL(h, P_SLOW_LEN, d_len, nq); POOL(h, P_SLOW_OFF, d_off, nq); POOL(h, P_SLOW_NRE, d_nre, 1);
        if (flip) POOL(h, P_SLOW_REQ2, nre_buf, nq); else POOL(h, P_SLOW_REQ, nre_buf, nq);      // (d_q may be the other one)
        HIPCHK(h, hipMemsetAsync(d_nre, 0, sizeof(u64), h->stream));
        k_slow_len<<<(unsigned)nq, TPB, 0, h->stream>>>(d_rid, n, d_q, nq, d_len, tw);
        std::vector<u64> len(nq), off(nq);
        HIPCHK(h, hipMemcpyAsync(len.data(), d_len, nq * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        u64 tot = 0;
        for (u64 i = 0; i < nq; ++i) { off[i] = tot; tot += 2 * len[i]; }
        u32 *sk = nullptr, *sm = nullptr;
        const u64 had_k = h->pool_bytes[ecb_handle::P_SLOW_KEY], had_m = h->pool_bytes[ecb_handle::P_SLOW_MASK];
        POOL(h, P_SLOW_KEY, sk, tot); POOL(h, P_SLOW_MASK, sm, tot);
        // k_slow leaves its scratch zeroed: only a fresh (re)allocation needs clearing
        if (h->pool_bytes[ecb_handle::P_SLOW_KEY] != had_k) HIPCHK(h, hipMemsetAsync(sk, 0, h->pool_bytes[ecb_handle::P_SLOW_KEY], h->stream));
        if (h->pool_bytes[ecb_handle::P_SLOW_MASK] != had_m) HIPCHK(h, hipMemsetAsync(sm, 0, h->pool_bytes[ecb_handle::P_SLOW_MASK], h->stream));
        HIPCHK(h, hipMemcpyAsync(d_off, off.data(), nq * sizeof(u64), hipMemcpyHostToDevice, h->stream));
        SlowArgs a{d_rid, d_loc, d_hf, d_q, d_len, d_off, sk, sm, h->cfg.n_loci, h->cfg.n_haplotypes,
                   h->table, h->cap - 1, h->arena, h->arena_cap, h->ctr, h->read_slot, h->reads_hi, nre_buf, d_nre, verify ? 1u : 0u};
        k_slow<<<(unsigned)nq, TPB, 2 * SLOW_LDS * sizeof(u32), h->stream>>>(a);
        u64 nre = 0;
        HIPCHK(h, hipMemcpyAsync(&nre, d_nre, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
        rc = sync_counters(h);                          // (waits: `off` lives on this stack frame)
        if (rc != ECB_OK || verify) break;
        d_q = nre_buf; nq = nre; flip ^= 1;
        if (nq) { rc = grow_table(h, h->cap * 4); if (rc != ECB_OK) break; }
    }
    return rc;
}

int excl_scan(ecb_handle* h, const u32* in, u64 n, u32* out, u64* total);

// Launch shape of k_stream over n records: the stream is cut into ECB_ROUNDS x as many slices as waves are resident at
// once; the launch holds the resident waves only, which claim slice after slice.
struct StreamPlan { u64 slices, chunk, blocks, pwaves; };
int plan_stream(ecb_handle* h, u64 n, StreamPlan* P, bool ranges = false, bool short_reads = false, bool par = false) {
    if (!h->resident_blocks) {                         // (asked once per handle: two runtime queries per batch add up on a streamed BAM)
        int cus = 256, bpc = 4;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, ks_std::k_stream<false>, TPB, 0);
        h->resident_blocks = (u64)std::max(cus, 1) * std::max(bpc, 1);
        int bpr = 4;                                   // (the variant with the range update has fewer waves resident)
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpr, ks_std::k_stream<false, true>, TPB, 0);
        h->resident_blocks_rg = (u64)std::max(cus, 1) * std::max(bpr, 1);
        int bps = 4;                                   // (the variant for short reads: 9.9 KB of LDS per wave, four workgroups per CU)
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&bps, ks_short::k_stream<false>, TPB, 0);
        h->resident_blocks_sh = (u64)std::max(cus, 1) * std::max(bps, 1);
        int bpp = 5;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpp, ks_par::k_stream<false>, TPB, 0);
        h->resident_blocks_par = (u64)std::max(cus, 1) * std::max(bpp, 1);
        h->rounds = getenv("ECB_ROUNDS") ? std::max(1, atoi(getenv("ECB_ROUNDS"))) : 24;   // (16 .. 32 measure alike on C3; fewer slices = fewer slice tails read twice)
        h->min_tiles = getenv("ECB_MIN_TILES") ? std::max(2, atoi(getenv("ECB_MIN_TILES"))) : 32;
    }
    const u64 rounds = h->rounds, resident_blocks = short_reads ? h->resident_blocks_sh : (ranges ? h->resident_blocks_rg : (par ? h->resident_blocks_par : h->resident_blocks));
    // slices: `rounds` per resident wave for balance, but not shorter than MIN_TILES tiles while every resident wave still gets
    // one -- a wave runs on past its slice's end to finish its open read, so every slice costs about one tile read twice
    // (at 5 tiles per slice, an eighth of config 3 on one of 8 GPUs, that was a fifth of the kernel)
    const u64 MIN_TILES = h->min_tiles;
    const u64 res_waves = resident_blocks * NWAVE;
    u64 waves = std::min<u64>(res_waves * rounds, std::max<u64>(n / (MIN_TILES * WT), res_waves));
    waves = std::min<u64>(waves, (n + 2 * WT - 1) / (2 * WT));
    waves = std::max<u64>(waves, 1);
    u64 chunk = (n + waves - 1) / waves;
    chunk = (chunk + WT - 1) / WT * WT;          // whole tiles: a tile never straddles two slices
    waves = (n + chunk - 1) / chunk;
    P->slices = waves; P->chunk = chunk;
    P->blocks = std::min<u64>((waves + NWAVE - 1) / NWAVE, resident_blocks);
    P->pwaves = P->blocks * NWAVE;               // waves of the launch
    // deferred reads: a parked launch defers at most the reads of the tiles in flight (one tile per resident wave); a read
    // with more than CMAX loci takes more than CMAX records
    const u64 need_q = P->pwaves * (u64)(WT + 1) + n / 64 + 16;      // (64: the smaller of the two kernels' carry limits, k_stream.inc)
    if (h->queue_cap < need_q) {
        if (h->queue) hipFree(h->queue);
        h->queue = nullptr; h->queue_cap = 0;
        HIPCHK(h, hipMalloc(&h->queue, need_q * sizeof(u64)));
        h->queue_cap = need_q;
    }
    return ECB_OK;
}

// the reads a k_stream launch deferred (h->hctr.n_queue of them, in the stripes of h->queue), closed up and handed to k_slow
int run_deferred(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, u64 n, bool verify, u32 tw, u64* n_done = nullptr) {
    const u64 nq = h->hctr.n_queue;
    if (n_done) *n_done = nq;
    if (!nq) return ECB_OK;
    u64* d_q = nullptr;
    POOL(h, P_QCOMPACT, d_q, nq);
    k_queue_compact<<<QSTRIPES, TPB, 0, h->stream>>>(h->queue, h->queue_cap, h->ctr, d_q);
    HIPCHK(h, hipGetLastError());
    return run_slow(h, d_rid, d_loc, d_hf, n, d_q, nq, verify, tw);
}

// The exactness pass over one device-resident batch (whole reads, read ids continuing from prev_rid): every read's target
// set is derived again from its records and compared, pair by pair, with the key of the EC the read was given.
// Reads longer than a tile go through k_slow's compare.  *n_mismatch = reads in a wrong EC (0 = exact); *n_long = how many
// took the long path.
int verify_batch(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, u64 n, u32 prev_rid, u64* n_mismatch, u64* n_long, u32 tw = 512u) {
    StreamPlan P;
    int rc = plan_stream(h, n, &P);
    if (rc != ECB_OK) return rc;
    u64* d_resume = nullptr; u32* d_wcounts = nullptr;
    POOL(h, P_RESUME, d_resume, 2 * P.slices); POOL(h, P_WCOUNTS, d_wcounts, 3 * P.pwaves);
    k_init_resume<<<nblk(P.slices, TPB), TPB, 0, h->stream>>>(d_resume, P.slices, P.chunk);
    HIPCHK(h, hipMemsetAsync(d_wcounts, 0, 3 * P.pwaves * sizeof(u32), h->stream));
    HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, (1 + QSTRIPES) * sizeof(u64), h->stream));      // (the total and the stripes' counters behind it)
    HIPCHK(h, hipMemsetAsync(&h->ctr->full, 0, sizeof(u32), h->stream));
    HIPCHK(h, hipMemsetAsync(&h->ctr->next_slice, 0, sizeof(u64), h->stream));
    HIPCHK(h, hipMemsetAsync(&h->ctr->n_mismatch, 0, sizeof(u64), h->stream));
    *h->pin_cold = StreamCold{h->arena, h->arena_cap, h->queue, h->queue_cap, d_resume, d_wcounts, nullptr, nullptr, P.chunk, prev_rid,
                              nullptr, nullptr, 0u, 0u};
    HIPCHK(h, hipMemcpyAsync(h->d_cold, h->pin_cold, sizeof(StreamCold), hipMemcpyHostToDevice, h->stream));
    StreamArgs a{d_rid, d_loc, d_hf, n, h->table, h->cap - 1, h->ctr, h->read_slot, h->reads_hi, h->d_cold, 0u};
    ks_std::k_stream<true><<<(unsigned)P.blocks, TPB, 0, h->stream>>>(a);
    k_sum_counts<<<1, 1024, 0, h->stream>>>(d_wcounts, P.pwaves, h->ctr, 1u, 0ull, h->queue_cap);
    HIPCHK(h, hipGetLastError());
    rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    u64 nq = 0;
    rc = run_deferred(h, d_rid, d_loc, d_hf, n, true, tw, &nq);
    if (rc != ECB_OK) return rc;
    *n_mismatch = h->hctr.n_mismatch;
    *n_long = nq;
    HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, (1 + QSTRIPES) * sizeof(u64), h->stream));
    h->hctr.n_queue = 0;
    return ECB_OK;
}

// one batch of whole reads, device-resident
int process_batch(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, const int* d_pos, u64 n, u32 tw = 512u) {
    if (n == 0) return ECB_OK;
    // How many reads the stream holds after this batch: the read id of its last record, fetched before anything is launched -- or,
    // when the caller has said how many reads the whole stream holds at most (ecb_hint_reads), that bound now and the
    // exact number with the counters, at the batch's one wait.
    const bool hinted = h->reads_hint != 0;
    u32 last_rid = 0;
    u64 reads_after = std::max<u64>(h->reads_hint, h->n_reads);
    if (!hinted) {
        HIPCHK(h, hipMemcpyAsync(&last_rid, d_rid + rec_at(n - 1, tw), sizeof(u32), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        reads_after = (u64)(u32)(last_rid + 1u);
        if (reads_after < h->n_reads) return fail(h, ECB_ERR_CONTRACT, "read_id went backwards across pushes");
    }
    int rc = ensure_read_slot(h, reads_after);
    if (rc != ECB_OK) return rc;
    h->reads_hi = reads_after;
    // keep the table at most half full before a batch (it grows again, via k_slow, if a batch overfills it)
    while (h->n_ecs() * 2 > h->cap) { rc = grow_table(h, h->cap * 4); if (rc != ECB_OK) return rc; }
    // Short reads (a 512-record tile holds more reads than a pass of 64 takes): the kernel with passes of 128 reads.  Known only when the
    // caller has said how many reads the stream holds (ecb_hint_reads) -- the records-per-read of this batch is then n / (its share of them).
    // (records per read: of the batches before this one where there are any; a first batch is judged as if it were the whole stream -- it must
    //  then hold at least one record per announced read -- so that a long-read stream pushed in many batches is not taken for a short-read one)
    const bool few = h->n_reads ? h->records_pushed < 7 * h->n_reads : (n >= h->reads_hint && n < 7 * h->reads_hint);
    h->short_reads = hinted && !h->rng && h->cfg.n_loci < MAX_LOCI_SHORT && h->cfg.n_haplotypes <= 8 && !getenv("ECB_NO_SHORT") &&
                     (getenv("ECB_FORCE_SHORT") || few);
    // Loci that displace each other in the LDS table (paralogs: target ids anywhere): the compilation whose key compare settles displaced pairs
    // in registers, once a batch has shown that more than a quarter of its tiles took the probe-on path (and back below a sixteenth).
    const bool par = !h->short_reads && !h->rng && !getenv("ECB_NO_PAR") && (getenv("ECB_FORCE_PAR") || h->par_stream);
    StreamPlan P;
    rc = plan_stream(h, n, &P, h->rng != nullptr, h->short_reads, par);
    if (rc != ECB_OK) return rc;
    const u64 waves = P.slices, chunk = P.chunk, blocks = P.blocks, pwaves = P.pwaves;
    u64* d_resume = nullptr;
    POOL(h, P_RESUME, d_resume, 2 * waves);
    k_init_resume<<<nblk(waves, TPB), TPB, 0, h->stream>>>(d_resume, waves, chunk, h->ctr);
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
    StreamCold cold{h->arena, h->arena_cap, h->queue, h->queue_cap, d_resume, d_wcounts, h->wave_arena, nullptr, chunk, h->prev_rid,
                    d_pos, h->rng, h->cfg.n_loci, h->cfg.n_haplotypes};
#ifdef ECB_TIMING
    HIPCHK(h, hipMalloc(&cold.timing, 8 * sizeof(u64)));
    HIPCHK(h, hipMemset(cold.timing, 0, 8 * sizeof(u64)));
#endif
    *h->pin_cold = cold;      // (pinned: rewritten by the next batch, which starts after this one's host wait)
    HIPCHK(h, hipMemcpyAsync(h->d_cold, h->pin_cold, sizeof(StreamCold), hipMemcpyHostToDevice, h->stream));
    StreamArgs a{d_rid, d_loc, d_hf, n, h->table, h->cap - 1, h->ctr, h->read_slot, h->reads_hi, h->d_cold,
                 getenv("ECB_ABLATE") ? (u32)atoi(getenv("ECB_ABLATE")) : 0u, d_pos, h->rng};
    h->ctr_synced = false;
    const u64 probe_before = h->hctr.n_probe_tiles;
    u64 offered = n;                                    // records offered to the filter (bam_utils.py:261): all of the batch
    for (u32 launch = 0;; ++launch) {
        if (launch) {                                   // per launch (the first one's: k_init_resume)
            HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, (1 + QSTRIPES) * sizeof(u64), h->stream));
            HIPCHK(h, hipMemsetAsync(&h->ctr->full, 0, sizeof(u32), h->stream));
            HIPCHK(h, hipMemsetAsync(&h->ctr->next_slice, 0, sizeof(u64), h->stream));
        }
        a.table = h->table; a.cap_mask = h->cap - 1;     // (d_wcounts: every wave of the launch stores its three words when it ends)
        if (h->prof) hipEventRecord(h->ev0, h->stream);
        if (h->rng) { ks_std::k_stream<false, true><<<(unsigned)blocks, TPB, 0, h->stream>>>(a); h->last_kernel = "ks_std::k_stream<false, true>"; }     // ... with the range update fused in
        else if (h->short_reads) { ks_short::k_stream<false><<<(unsigned)blocks, TPB, 0, h->stream>>>(a); h->last_kernel = "ks_short::k_stream<false, false>"; }
        else if (par) { ks_par::k_stream<false><<<(unsigned)blocks, TPB, 0, h->stream>>>(a); h->last_kernel = "ks_par::k_stream<false, false>"; }
        else { ks_std::k_stream<false><<<(unsigned)blocks, TPB, 0, h->stream>>>(a); h->last_kernel = "ks_std::k_stream<false, false>"; }
        if (h->prof) hipEventRecord(h->ev1, h->stream);
        k_sum_counts<<<1, 1024, 0, h->stream>>>(d_wcounts, pwaves, h->ctr, 0u, offered, h->queue_cap, d_rid + rec_at(n - 1, tw));
        offered = 0;                                    // (a relaunch after a park continues the same batch)
        HIPCHK(h, hipGetLastError());
        rc = sync_counters(h);
        if (h->prof) {
            float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1);
            h->prof_ms += ms; h->prof_launches += 1;
        }
        if (rc != ECB_OK) break;
        const bool parked = h->hctr.full != 0;
        rc = run_deferred(h, d_rid, d_loc, d_hf, n, false, tw);
        if (rc != ECB_OK) break;
        if (!parked) break;
        rc = grow_table(h, h->cap * 4);                 // some workgroups stopped early: more room, then resume
        if (rc != ECB_OK) break;
    }
#ifdef ECB_TIMING
    {
        u64 t[8];
        hipMemcpy(t, cold.timing, sizeof(t), hipMemcpyDeviceToHost); hipFree(cold.timing);
        static const char* nm[8] = {"(a) filter+heads", "tile decisions", "prefetch issue+clear+geometry", "(b) LDS tables", "hash entries", "(c) lookup", "publish/settle/slot stores", "-"};
        u64 tot = 0; for (int i = 0; i < 7; ++i) tot += t[i];
        fprintf(stderr, "[ecb timing] %llu waves, clocks per wave:", (unsigned long long)waves);
        for (int i = 0; i < 7; ++i) fprintf(stderr, "  %s %.0f (%.1f%%)", nm[i], (double)t[i] / waves, 100.0 * t[i] / std::max<u64>(tot, 1));
        fprintf(stderr, "  tile visits on the probe-on path %llu of %llu tiles", (unsigned long long)t[7], (unsigned long long)((n + WT - 1) / WT));
        fprintf(stderr, "\n");
    }
#endif
    if (rc != ECB_OK) return rc;
    {
        const u64 tiles = (n + WT - 1) / WT, on_path = h->hctr.n_probe_tiles - probe_before;
        if (on_path * 4 > tiles) h->par_stream = true; else if (on_path * 16 < tiles) h->par_stream = false;
    }
    if (h->prof) h->prof_records += n;
    if (h->cfg.flags & ECB_F_VERIFY) {                  // belt and braces: the grouping is exact by construction (Slot), this re-derives it
        u64 bad = 0, nl = 0;
        rc = verify_batch(h, d_rid, d_loc, d_hf, n, h->prev_rid, &bad, &nl, tw);
        if (rc != ECB_OK) return rc;
        h->n_mismatch += bad;
        if (bad) return fail(h, ECB_ERR_VERIFY, "exactness pass: %llu read(s) of this batch sit in an EC whose key is not their target set", (unsigned long long)bad);
    }
    if (hinted) {
        last_rid = h->hctr.last_rid;
        reads_after = (u64)(u32)(last_rid + 1u);
        if (reads_after < h->n_reads) return fail(h, ECB_ERR_CONTRACT, "read_id went backwards across pushes");
    }
    h->prev_rid = last_rid;
    h->n_reads = reads_after;
    h->records_pushed += n;
    return ECB_OK;
}

int ensure_staging(ecb_handle* h, u64 need) {
    if (need <= h->st_cap) return ECB_OK;
    // one allocation for the three (four) staging streams, each on a 2 MiB boundary within it.  (Staging in whole tiles -- what
    // ecb_push_device_tiled takes -- was built and measured: the pitched host-to-device copies cost the PCIe-inclusive rate 8 %, and the
    // tile layout did not take the placement dependence out of the kernel after all: DESIGN.md section 6.)
    if (h->st_rid) hipFree(h->st_rid);
    h->st_rid = h->st_loc = h->st_hf = nullptr; h->st_pos = nullptr;
    h->st_cap = std::max<u64>(need, h->cfg.max_batch_records);
    const u64 stride = (h->st_cap * sizeof(u32) + (2ull << 20) - 1) / (2ull << 20) * (2ull << 20) / sizeof(u32);
    const bool rg = (h->cfg.flags & ECB_F_RANGES) != 0;
    HIPCHK(h, hipMalloc(&h->st_rid, (rg ? 4 : 3) * stride * sizeof(u32)));
    h->st_loc = h->st_rid + stride; h->st_hf = h->st_rid + 2 * stride;
    if (rg) h->st_pos = reinterpret_cast<int*>(h->st_rid + 3 * stride);
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

// exclusive scan of u32 values, queued on the handle's stream; the sum (64 bits) is left in *d_total on the device
// (a sum of 2^32 or more: the caller's limit check, `out` wrapped)
int excl_scan_dev(ecb_handle* h, const u32* in, u64 n, u32* out, u64* d_total, u32 stride = 1, u32* last32 = nullptr) {
    u32* sums = nullptr;
    POOL(h, P_SUMS, sums, scan_words(n));
    HIPCHK(h, scan_launch(h->stream, in, n, out, sums, d_total, stride, last32));
    return ECB_OK;
}
// ... and with the sum brought to the host (one wait)
int excl_scan(ecb_handle* h, const u32* in, u64 n, u32* out, u64* total) {
    u64* d_tot = nullptr;
    POOL(h, P_TOTALS, d_tot, 8);
    int rc = excl_scan_dev(h, in, n, out, d_tot + 7);
    if (rc != ECB_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(total, d_tot + 7, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

// Stable LSD radix sort of n (key, value) pairs on `st`: only the digits some key has a bit in are sorted on (one OR-reduction
// and one host wait up front).  k[0] / v[0] hold the input; the result lands in k[*where] / v[*where].  scratch: u32 hist and
// offs of 256 * tiles each, u32 sums of tiles / 8 + 8, and 8 bytes at d_word.
// scratch of a sort of n pairs: `hist` = the (tile, digit) words of the look-back, rs_words(n) of them; `offs` = RS_AUX_WORDS words
// (histograms and first places of all passes, the tile counter, the error word); `sums` unused; d_word = 8 bytes
struct SortScratch { u32 *hist, *offs, *sums; u64* d_word; };
inline u64 rs_tiles(u64 n) { return std::max<u64>(1, (n + RS_TILE - 1) / RS_TILE); }
constexpr u64 RS_AUX_WORDS = 2 * RS_MAX_PASSES * 256 + 64;
inline u64 rs_words(u64 n) { return std::max<u64>(256 * rs_tiles(n), RS_AUX_WORDS); }
inline u64 rs_scan_blocks(u64) { return 1; }
hipError_t radix_sort_pairs64(hipStream_t st, u64* k[2], u32* v[2], u64 n, const SortScratch& sc, int* where, u64 bit_mask = ~0ull) {
    *where = 0;
    if (n < 2) return hipSuccess;
    if (n >= (1ull << 30)) return hipErrorInvalidValue;          // (a look-back word carries a 30-bit count)
    hipError_t e = hipSuccess;
    u64 ormask = bit_mask;                          // (a caller that names the bits to sort on has no use for the reduction and its host wait)
    if (bit_mask == ~0ull) {
        if ((e = hipMemsetAsync(sc.d_word, 0, 8, st)) != hipSuccess) return e;
        k_or_reduce<<<(unsigned)std::min<u64>(1024, (n + TPB - 1) / TPB), TPB, 0, st>>>(k[0], n, sc.d_word);
        if ((e = hipMemcpyAsync(&ormask, sc.d_word, 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    }
    RsShifts sh{};
    for (u32 shift = 0; shift < 64; shift += 8)
        if ((ormask >> shift) & 255ull) sh.s[sh.n++] = shift;   // (every key has zero in the other digits: already in order)
    if (!sh.n) return hipSuccess;
    const u32 nt = (u32)rs_tiles(n);
    u32 *ghist = sc.offs, *base = sc.offs + RS_MAX_PASSES * 256, *ticket = sc.offs + 2 * RS_MAX_PASSES * 256, *err = ticket + 1;
    if ((e = hipMemsetAsync(sc.offs, 0, RS_AUX_WORDS * 4, st)) != hipSuccess) return e;
    k_rs_hist_all<<<(unsigned)std::min<u64>(2048, (n + 16 * RS_TPB - 1) / (16 * RS_TPB)), RS_TPB, 0, st>>>(k[0], n, sh, ghist);
    k_rs_bases<<<1, 256, 0, st>>>(ghist, sh.n, base);
    int cur = 0;
    for (u32 p = 0; p < sh.n; ++p) {
        if ((e = hipMemsetAsync(sc.hist, 0, (u64)nt * 256 * 4, st)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(ticket, 0, 4, st)) != hipSuccess) return e;
        k_rs_pass<<<nt, RS_TPB, 0, st>>>(k[cur], v[cur], n, sh.s[p], base + p * 256, sc.hist, ticket, err, k[cur ^ 1], v[cur ^ 1]);
        cur ^= 1;
    }
    *where = cur;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    u32 gave_up = 0;                                // (a look-back that ran out of polls leaves the order undefined: never silently)
    if ((e = hipMemcpyAsync(&gave_up, err, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    return gave_up ? hipErrorUnknown : hipSuccess;
}
int handle_sort(ecb_handle* h, u64* k[2], u32* v[2], u64 n, int* where) {
    SortScratch sc{};
    u64* tot = nullptr;
    POOL(h, P_RS_HIST, sc.hist, rs_words(n)); POOL(h, P_RS_OFFS, sc.offs, RS_AUX_WORDS);
    POOL(h, P_RS_SUMS, sc.sums, rs_scan_blocks(n) + 8); POOL(h, P_TOTALS, tot, 8);
    sc.d_word = tot + 6;
    const hipError_t e = radix_sort_pairs64(h->stream, k, v, n, sc, where);
    if (e != hipSuccess) return fail(h, ECB_ERR_HIP, "radix sort: %s", hipGetErrorString(e));
    return ECB_OK;
}

int ensure_slot_ranks(ecb_handle* h, u64 E) {
    if (h->rank_of_slot) return ECB_OK;
    POOL(h, P_RANK, h->rank_of_slot, h->cap);
    k_slot_ranks<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->order, E, h->rank_of_slot);
    HIPCHK(h, hipGetLastError());
    return ECB_OK;
}

// reads per EC / first appearance, from read_slot[0, n_reads) (once, when the stream is closed).  Only queues kernels:
// the totals and the work list of k_count_bins stay on the device.
int ensure_counts(ecb_handle* h, const CompactSink* sink = nullptr) {
    if (h->counted) return ECB_OK;
    const u64 R = h->n_reads;
    CompactSink own{nullptr, 0, nullptr, nullptr, 0, nullptr};
    if (R && !sink && !h->adopted && h->n_ecs()) {         // (a table export follows: it wants the list of occupied slots, see k_count_bins)
        u64* d_n = nullptr;
        POOL(h, P_LIST, h->list, h->n_ecs());
        POOL(h, P_CNT, d_n, 1);
        HIPCHK(h, hipMemsetAsync(d_n, 0, sizeof(u64), h->stream));
        own = CompactSink{h->list, h->n_ecs(), d_n, nullptr, 0, nullptr};
        sink = &own;
    }
    if (R) {
        // Slots per range: 512 ranges -- one k_count_bins workgroup for each of the chip's 512 places, all at once -- unless that takes more than 2^14 slots
        // each (64 KB of LDS counters: two workgroups per CU; 2^15 leaves room for one).  C3: 1 024 ranges of 2^14; C2: 512 of 2^13; the diploid stream
        // 512 of 2^13.  Fewer, fuller ranges leave CUs without a workgroup (C2 with 256 ranges: k_count_bins 0.19 ms against 0.13); more ranges make the
        // partition's runs shorter (k_part_scatter_staged at C3 with 2 048: 0.38 ms against 0.26; on the diploid stream with 2 048: 0.71 against 0.49 with 1 024)
        // -- profiles/r04_finalize_kernels.txt.
        u32 bb = BIN_BITS;
        {
            u32 lc = 0;
            while ((1ull << (lc + 1)) <= h->cap) ++lc;   // the table holds 2^lc slots
            bb = std::min<u32>(std::max<u32>(lc, 9u + MIN_BIN_BITS) - 9u, BIN_BITS);
        }
        if (const char* e = getenv("ECB_BIN_BITS")) bb = (u32)std::min<long>(std::max<long>(atol(e), MIN_BIN_BITS), MAX_BIN_BITS);   // (measurement knob: tools/tools_env.sh)
        while (bb < MAX_BIN_BITS && (h->cap >> bb) > MAX_BUCKETS) ++bb;
        const u32 nb = (u32)std::max<u64>(1, h->cap >> bb);
        if (nb > MAX_BUCKETS) return fail(h, ECB_ERR_LIMIT, "EC table larger than 2^28 slots is not supported");
        if (bb > 14 && !h->count_attr_set) {        // (more than 64 KB of dynamic LDS has to be asked for)
            HIPCHK(h, hipFuncSetAttribute((const void*)k_count_bins, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << MAX_BIN_BITS));
            h->count_attr_set = true;
        }
        const u32 G = (u32)std::min<u64>(PART_G, (R + 4095) / 4096);
        u32 *hist = nullptr, *offs = nullptr;
        u16* pairs = nullptr;
        u64* d_tot = nullptr;
        POOL(h, P_HIST, hist, (u64)nb * G); POOL(h, P_OFFS, offs, (u64)nb * G);
        POOL(h, P_PAIRS, pairs, R);
        POOL(h, P_TOTALS, d_tot, 8);
        k_part_hist<<<G, TPB_PART, nb * 4, h->stream>>>(h->read_slot, R, nb, bb, hist);
        d_tot += 6;                                 // ([0..3] are finalize's, which may run this with its own totals already zeroed; [7] is excl_scan's)
        int rc = excl_scan_dev(h, hist, (u64)nb * G, offs, d_tot);
        if (rc != ECB_OK) return rc;
        if (nb <= STAGE_MAX_BUCKETS) {
            if (!h->scatter_attr_set) {             // (per handle = per device: more than 64 KB of dynamic LDS has to be asked for)
                HIPCHK(h, hipFuncSetAttribute((const void*)k_part_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              3 * STAGE_MAX_BUCKETS * 4 + STAGE * 4));
                h->scatter_attr_set = true;
            }
            k_part_scatter_staged<<<G, TPB_PART, 3 * nb * 4 + STAGE * 4, h->stream>>>(h->read_slot, R, nb, bb, offs, pairs);
        } else                                     // (tables beyond 2^25 slots: the counters would crowd the stage out of LDS)
            k_part_scatter<<<G, TPB_PART, nb * 4, h->stream>>>(h->read_slot, R, nb, bb, offs, pairs);
        // work list of k_count_bins: ranges far above the average are cut into pieces (see CountWork)
        const u32 piece = std::max<u32>(32768u, 2u * (u32)((R + nb - 1) / nb));
        const u32 max_work = nb + (u32)(R / piece) + 1;
        CountWork* d_work = nullptr;
        POOL(h, P_WORK, d_work, (u64)max_work + 1);          // (+ 1: its length sits behind the list)
        u32* d_nwork = reinterpret_cast<u32*>(d_work + max_work);
        k_build_work<<<1, 1024, 0, h->stream>>>(offs, G, d_tot, nb, piece, d_work, max_work, d_nwork);
        k_count_bins<<<max_work, TPB_COUNT, 4u << bb, h->stream>>>(pairs, d_work, d_nwork, bb, h->table, sink ? *sink : CompactSink{nullptr, 0, nullptr, nullptr, 0, nullptr});
        if (sink == &own) { h->list_counted = true; h->d_list_n = own.n_list; }
        else if (sink) h->list_from_counts = true;
        HIPCHK(h, hipGetLastError());
    }
    h->counted = true;
    return ECB_OK;
}

// occupied slots -> h->list (+ first-appearance bitmap and (first, key length) per entry for finalize); the number found
// is left in *d_n on the device
int compact_table_dev(ecb_handle* h, u64* d_n, u32* bitmap = nullptr, u64 n_bits = 0, uint2* list_fn = nullptr) {
    HIPCHK(h, hipMemsetAsync(d_n, 0, sizeof(u64), h->stream));
    POOL(h, P_LIST, h->list, h->n_ecs());
    k_compact<<<(unsigned)std::min<u64>(2048, (h->cap + 4 * TPB_COMPACT - 1) / (4 * TPB_COMPACT)), TPB_COMPACT, 0, h->stream>>>(h->table, h->cap, h->list, std::max<u64>(h->n_ecs(), 1), d_n, bitmap, n_bits, list_fn);
    return ECB_OK;
}
int compact_table(ecb_handle* h) {
    u64* d_n = nullptr;
    int rc = ECB_OK;
    if (h->list_counted) {                               // the counting pass listed them: no scan of the table
        POOL(h, P_LIST, h->list, h->n_ecs());            // (the pool's buffer, contents and all)
        d_n = h->d_list_n;
    } else {
        POOL(h, P_CNT, d_n, 1);
        rc = compact_table_dev(h, d_n);
        if (rc != ECB_OK) return rc;
    }
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
    {
        u64* kk[2] = {keys, keys2}; u32* vv[2] = {vals, vals2};
        int where = 0;
        const int rc0 = handle_sort(h, kk, vv, R, &where);
        if (rc0 != ECB_OK) return rc0;
        keys2 = kk[where]; vals2 = vv[where];
    }
    k_ms_heads<<<nblk(R, TPB), TPB, 0, h->stream>>>(keys2, R, flag);
    u64 nt = 0;
    int rc = excl_scan(h, flag, R, pos, &nt);
    if (rc != ECB_OK) return rc;
    POOL(h, P_MS_OKEY, h->ms_okey, nt); POOL(h, P_MS_OFIRST, h->ms_ofirst, nt); POOL(h, P_MS_OSTART, h->ms_ostart, (u64)nt + 1);
    k_ms_emit<<<nblk(R, TPB), TPB, 0, h->stream>>>(keys2, vals2, flag, pos, R, nt, h->ms_okey, h->ms_ofirst, h->ms_ostart);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_triples = nt; h->ms_ocount = nullptr;
    return ECB_OK;
}

// exported entries number their first reads from the shard's own 0 (or from the read_base given at export): move them on by `base`
__global__ void k_rebase(Entry* e, u64 n, u32 base, u32 limit, Counters* ctr) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 first = ~e[i].first_inv;
    if (first >= limit) { atomicOr(&ctr->err, ERR_CONTRACT); return; }     // (would run past 2^32 - 2 reads)
    e[i].first_inv = ~(first + base);
}

// ---- finalize per key range (multi-GPU): every rank ranks and emits the ECs of its own key range, the root only puts the pieces
// in the order of first appearance.  The rank of an EC = the number of ECs whose first read comes before its own
// (bam_utils.py:682-698): a bitmap over the reads, marked from every piece's first reads, and its prefix popcount.
__global__ void k_export_firsts(const Slot* table, const u32* order, u64 n, u32* firsts) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e < n) firsts[e] = ~table[order[e]].first_inv;
}
struct PieceDesc { const u32* firsts; const int *indptr, *indices, *data, *counts; u64 n, nnz, at; };   // one finalized key range (at: its first EC among all pieces')
// (all pieces in one launch: blockIdx.y = piece)
__global__ void k_mark_bits(const PieceDesc* P, u64 n_bits, u32* bitmap, Counters* ctr) {
    const PieceDesc d = P[blockIdx.y];
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= d.n) return;
    const u32 f = d.firsts[e];
    if (f >= n_bits) { atomicOr(&ctr->err, ERR_CONTRACT); return; }
    atomicOr(&bitmap[f >> 5], 1u << (f & 31u));
}
// (what is known of rank r goes out as ONE 16-byte word {EC within its piece + 1, row length, count, piece}: the scatter is this kernel's cost)
__global__ void k_piece_place(const PieceDesc* P, u64 n_total, u64 n_bits, const u32* bitmap, const u32* wprefix, uint4* place, Counters* ctr) {
    const PieceDesc d = P[blockIdx.y];
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= d.n) return;
    const u32 f = d.firsts[e];
    if (f >= n_bits) return;                             // (reported by k_mark_bits)
    const long long s0 = d.indptr[e], s1 = d.indptr[e + 1];
    if (s0 < 0 || s1 < s0 || (u64)s1 > d.nnz) { atomicOr(&ctr->err, ERR_CONTRACT); return; }
    const u32 r = bit_rank(bitmap, wprefix, f);
    if (r >= n_total) return;
    place[r] = make_uint4((u32)e + 1u, (u32)(s1 - s0), (u32)d.counts[e], blockIdx.y);    // (two pieces claiming one first read: either, whole; the caller reports it)
}
// One thread per row of the result: neighbours write neighbouring rows, and read rows that follow each other within their piece
// (a piece is in first-read order itself).
__global__ void k_piece_rows(const PieceDesc* P, u32 n_pieces, const uint4* place, u64 n_total, const u32* indptr, int* indices, int* data, int* counts) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (r >= n_total) return;
    const uint4 pl = place[r];
    counts[r] = (int)pl.z;
    if (pl.x == 0u || pl.w >= n_pieces) return;         // (no piece claimed this rank: the caller reports it)
    const u32 q = pl.w;
    const u64 e = pl.x - 1u;
    const int* ip = P[q].indptr;
    const int s0 = ip[e], s1 = ip[e + 1];
    const u32 d0 = indptr[r];
    if (indptr[r + 1] - d0 != (u32)(s1 - s0) || pl.y != (u32)(s1 - s0)) return;   // (nothing is ever written past a row)
    const int *sj = P[q].indices, *sd = P[q].data;
    for (int i = s0; i < s1; ++i) { indices[d0 + (u32)(i - s0)] = sj[i]; data[d0 + (u32)(i - s0)] = sd[i]; }
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
    if (cfg->n_loci == 0 || cfg->n_loci >= MAX_LOCI) return fail(nullptr, ECB_ERR_ARG, "n_loci out of range (1 .. 2^26-3)");
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
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->pin_ctr), sizeof(Counters), hipHostMallocDefault)) != hipSuccess) return bail(ECB_ERR_HIP, "hipHostMalloc", e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->pin_cold), sizeof(StreamCold), hipHostMallocDefault)) != hipSuccess) return bail(ECB_ERR_HIP, "hipHostMalloc", e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), sizeof(ecb_handle::PinOut), hipHostMallocDefault)) != hipSuccess) return bail(ECB_ERR_HIP, "hipHostMalloc", e);
    if ((e = hipMalloc(&h->d_cold, sizeof(StreamCold))) != hipSuccess) return bail(ECB_ERR_HIP, "hipMalloc(args)", e);
    hipMemsetAsync(h->table, 0, h->cap * sizeof(Slot), h->stream);
    hipMemsetAsync(h->arena, 0, h->arena_cap * sizeof(uint2), h->stream);     // stale arena bytes must never look like a key (see ecb_reset)
    clear_counters(h);
    if (cfg->flags & ECB_F_RANGES) {
        const u64 ns = (u64)cfg->n_loci * cfg->n_haplotypes;
        if ((e = hipMalloc(&h->rng, ns * sizeof(int2))) != hipSuccess) return bail(ECB_ERR_HIP, "hipMalloc(ranges)", e);
        k_fill_minmax<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng, ns);
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
    hipFree(h->rng); hipFree(h->queue); hipFree(h->wave_arena);
    if (h->pin_ctr) hipHostFree(h->pin_ctr);
    if (h->pin_out) hipHostFree(h->pin_out);
    if (h->pin_cold) hipHostFree(h->pin_cold);
    hipFree(h->d_cold);
    for (int i = 0; i < ecb_handle::P_N; ++i) hipFree(h->pool[i]);
    hipFree(h->st_rid);                                  // (one allocation holds all staging streams: ensure_staging)
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int ecb_verify_device(ecb_handle* h, const void* d_read_id, const void* d_locus, const void* d_hapflag, size_t n,
                      uint64_t* n_mismatch, uint64_t* n_long) {
    if (!h || !n_mismatch || !n_long) return ECB_ERR_ARG;
    if (!n || !d_read_id || !d_locus || !d_hapflag) return fail(h, ECB_ERR_ARG, "null tuple stream");
    if (((uintptr_t)d_read_id | (uintptr_t)d_locus | (uintptr_t)d_hapflag) & 15) return fail(h, ECB_ERR_ARG, "device streams must be 16-byte aligned");
    HIPCHK(h, hipSetDevice(h->device));
    u64 bad = 0, nl = 0;
    const int rc = verify_batch(h, (const u32*)d_read_id, (const u32*)d_locus, (const u32*)d_hapflag, n, 0xFFFFFFFFu, &bad, &nl);
    *n_mismatch = bad; *n_long = nl;
    return rc;
}

int ecb_reset(ecb_handle* h) {
    if (!h) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const u32* occupied = h->finalized ? h->list : nullptr;     // (a finalized handle knows its occupied slots: every one of them, checked)
    const u64 n_occupied = h->n_list;
    free_results(h);
    // The used stretches of the key arena go back to zero: a key pair is only ever compared against bytes that are either
    // zero (no haplotype mask: never equal to a pair) or final, whatever a cache still holds of them.
    if (!h->ctr_synced) sync_counters(h);               // (the cursors; an error the run already reported is not this call's)
    {   // (one launch for all of them: a fill per region was 64 dispatches of ~5 us on a stream with long keys -- the paralog stream's reset 0.3 ms)
        const u32 R = arena_regions(h->arena_cap);
        const u64 per = h->arena_cap / R;
        ArenaClear ac{};
        u32 n = 0;
        u64 longest = 0;
        for (u32 r = 0; r < R; ++r) {
            const u64 used = std::min<u64>(h->hctr.arena_reg[r], (u64)(r + 1) * per) - r * per;
            if (used) { ac.start[n] = r * per; ac.len[n] = used; ++n; longest = std::max(longest, used); }
        }
        if (h->hctr.arena_top) { ac.start[n] = 0; ac.len[n] = std::min<u64>(h->hctr.arena_top, h->arena_cap); longest = std::max(longest, ac.len[n]); ++n; }
        if (n) {
            const dim3 grid((unsigned)std::min<u64>(nblk(longest, TPB), 2048), n);
            k_clear_arena<<<grid, TPB, 0, h->stream>>>(h->arena, ac);
            HIPCHK(h, hipGetLastError());
        }
    }
#if defined(ECB_TIMING) || defined(ECB_EXPERIMENTS)     // experiment builds only (tools/exp_hits.py: a pass over a table that holds every EC already)
    if (!getenv("ECB_KEEP_TABLE"))
#endif
    {
        // a sparsely filled table is cleared slot by slot from that list (config 3: 3.7 M of 16.8 M slots, 0.24 of 1 GB)
        if (h->assembled) {
            // (a result assembled from per-range pieces never touched this handle's table)
        } else if (occupied && n_occupied && n_occupied == h->n_ecs() && !h->hctr.err && n_occupied * 3 < h->cap)
            k_clear_slots<<<nblk(n_occupied, TPB), TPB, 0, h->stream>>>(h->table, occupied, n_occupied);
        else
            HIPCHK(h, hipMemsetAsync(h->table, 0, h->cap * sizeof(Slot), h->stream));
    }
    { int rc_ = clear_counters(h); if (rc_ != ECB_OK) return rc_; }
    if (h->wave_arena) HIPCHK(h, hipMemsetAsync(h->wave_arena, 0, 2 * h->wave_arena_n * sizeof(u64), h->stream));
    // (read_slot keeps the last run's slot ids: every read of the next stream has its entry written by k_stream or k_slow
    //  before anything reads it -- k_count, the exports and the exactness pass look at reads [0, n_reads) of a stream that was
    //  looked up without an error -- and entries of a fresh allocation are PENDING.  384 MB of stores per run at config 3.
    //  A handle that checks itself (ECB_F_VERIFY) pays for the fill: a read the stream kernel skipped then shows as PENDING, which the
    //  exactness pass counts as a read in a wrong EC.)
    if ((h->cfg.flags & ECB_F_VERIFY) && h->read_slot) HIPCHK(h, hipMemsetAsync(h->read_slot, 0xFF, h->read_slot_cap * sizeof(u32), h->stream));
    if (h->rng) {
        const u64 ns = (u64)h->cfg.n_loci * h->cfg.n_haplotypes;
        k_fill_minmax<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng, ns);
    }
    // (no wait: everything above is ordered on the handle's stream, where all later work goes too)
    h->prev_rid = 0xFFFFFFFFu; h->n_reads = 0; h->records_pushed = 0; h->reads_hi = 0; h->meta_hi = 0; h->n_triples = 0; h->ms_ocount = nullptr; h->ms_adopted = false;
    h->extra_all = h->extra_valid = h->extra_reads = 0;
    h->c_rid.clear(); h->c_loc.clear(); h->c_hf.clear(); h->c_pos.clear();
    h->finalized = false; h->counted = false; h->adopted = false; h->sizes = ecb_sizes{}; h->n_list = 0;
    h->list_counted = false; h->list_from_counts = false; h->assembled = false;
    h->n_mismatch = 0; h->ms_filtered = false;
    return ECB_OK;
}

int ecb_hint_reads(ecb_handle* h, uint64_t max_reads) {
    if (!h) return ECB_ERR_ARG;
    if (max_reads >= (1ull << 32)) return fail(h, ECB_ERR_LIMIT, "more than 2^32 - 1 reads");
    h->reads_hint = max_reads;
    return ECB_OK;
}

int ecb_push_device(ecb_handle* h, const void* d_read_id, const void* d_locus, const void* d_hapflag,
                    const void* d_pos, size_t n) {
    if (!h) return ECB_ERR_ARG;
    if (h->finalized || h->counted) return fail(h, ECB_ERR_STATE, "push after finalize / table export");
    if (n && (!d_read_id || !d_locus || !d_hapflag)) return fail(h, ECB_ERR_ARG, "null tuple stream");
    if ((h->cfg.flags & ECB_F_RANGES) && n && !d_pos) return fail(h, ECB_ERR_ARG, "ECB_F_RANGES needs pos");
    if (((uintptr_t)d_read_id | (uintptr_t)d_locus | (uintptr_t)d_hapflag | (uintptr_t)d_pos) & 15) return fail(h, ECB_ERR_ARG, "device streams must be 16-byte aligned");
    if (!h->c_rid.empty()) return fail(h, ECB_ERR_STATE, "ecb_push_device while a host push has an open read");
    HIPCHK(h, hipSetDevice(h->device));
    return process_batch(h, (const u32*)d_read_id, (const u32*)d_locus, (const u32*)d_hapflag, (const int*)d_pos, n);
}

// The same from ONE buffer of whole tiles: tile t = words [1536 t, 1536 t + 1536) = 512 read ids | 512 loci | 512 haplotype/flag words of records
// [512 t, 512 t + 512) (the last tile padded: the buffer holds ceil(n / 512) tiles).  What a tile of the stream kernel reads is then 6 KB in one
// place instead of 2 KB in each of three.  (Built to take the kernel's dependence on where the streams sit in HBM out -- a micro-benchmark of the
// reads alone said three streams side by side were the sensitive part -- and measured: it narrows the spread, 7.7 - 8.7 ms against 7.7 - 9.3 on
// config 3, but does not remove it: DESIGN.md section 6.  Kept as a second way in for callers whose tuples sit in one buffer.)
int ecb_push_device_tiled(ecb_handle* h, const void* d_tiles, size_t n) {
    if (!h) return ECB_ERR_ARG;
    if (h->finalized || h->counted) return fail(h, ECB_ERR_STATE, "push after finalize / table export");
    if (n && !d_tiles) return fail(h, ECB_ERR_ARG, "null tuple stream");
    if (h->cfg.flags & ECB_F_RANGES) return fail(h, ECB_ERR_STATE, "ECB_F_RANGES takes the four streams of ecb_push_device");
    if ((uintptr_t)d_tiles & 15) return fail(h, ECB_ERR_ARG, "device streams must be 16-byte aligned");
    if (!h->c_rid.empty()) return fail(h, ECB_ERR_STATE, "ecb_push_device_tiled while a host push has an open read");
    HIPCHK(h, hipSetDevice(h->device));
    const u32* d = (const u32*)d_tiles;
    return process_batch(h, d, d + 512, d + 1024, nullptr, n, 1536u);
}

int ecb_verify_device_tiled(ecb_handle* h, const void* d_tiles, size_t n, uint64_t* n_mismatch, uint64_t* n_long_reads) {
    if (!h || !n_mismatch) return ECB_ERR_ARG;
    if (!n || !d_tiles) return fail(h, ECB_ERR_ARG, "null tuple stream");
    if ((uintptr_t)d_tiles & 15) return fail(h, ECB_ERR_ARG, "device streams must be 16-byte aligned");
    HIPCHK(h, hipSetDevice(h->device));
    u64 bad = 0, nl = 0;
    const u32* d = (const u32*)d_tiles;
    const int rc = verify_batch(h, d, d + 512, d + 1024, n, 0xFFFFFFFFu, &bad, &nl, 1536u);
    if (rc != ECB_OK) return rc;
    *n_mismatch = bad;
    if (n_long_reads) *n_long_reads = nl;
    return ECB_OK;
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

int ecb_push_cells_device(ecb_handle* h, const void* d_meta, uint64_t first_read, size_t n) {
    if (!h) return ECB_ERR_ARG;
    if (!(h->cfg.flags & ECB_F_MULTISAMPLE)) return fail(h, ECB_ERR_STATE, "handle was created without ECB_F_MULTISAMPLE");
    if (h->finalized) return fail(h, ECB_ERR_STATE, "push after finalize");
    if (!n) return ECB_OK;
    if (!d_meta) return fail(h, ECB_ERR_ARG, "null meta");
    HIPCHK(h, hipSetDevice(h->device));
    if (first_read >= (1ull << 32) - 1 || (u64)n >= (1ull << 32) - 1 - first_read) return fail(h, ECB_ERR_LIMIT, "more than 2^32-2 reads");
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
    // (the copy is ordered on the handle's own stream, behind nothing of the caller's: d_meta must be COMPLETE when this is called -- see ecb.h)
    HIPCHK(h, hipMemcpyAsync(h->meta + first_read, d_meta, n * sizeof(u32), hipMemcpyDeviceToDevice, h->stream));
    h->meta_hi = std::max<u64>(h->meta_hi, need);
    return ECB_OK;
}

int ecb_finalize(ecb_handle* h, ecb_sizes* out) {
    if (!h || !out) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->assembled) { *out = h->sizes; return ECB_OK; }   // (the result of ecb_assemble_ranges_device: nothing left to rank)
    if (!h->finalized) {
        if (!h->c_rid.empty()) {                     // the stream ends here: the carried read is complete
            int rc = stage_and_process(h, nullptr, nullptr, nullptr, nullptr, 0);
            if (rc != ECB_OK) return rc;
        }
        int rc = h->ctr_synced ? ECB_OK : sync_counters(h);
        if (rc != ECB_OK) return rc;
    }
    const u64 E = h->n_ecs();
    const u64 valid = h->hctr.valid + h->extra_valid;
    if (E == 0 || valid == 0) return fail(h, ECB_ERR_EMPTY, "no valid alignments: nothing to build (the reference fails here too)");
    if (E >= (1ull << 31) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^31-2 equivalence classes");
    free_results(h);
    // Everything below is queued on the stream; the host waits once, at the end, and checks what the device counted.
    // rank by first appearance: bitmap over read indices (marked while the table is compacted), popcount prefix
    const u64 total_reads = h->n_reads + h->extra_reads;
    const u64 words = bitmap_words(total_reads), lines = words / BM_LINE;
    const u64 nnz_max = std::min<u64>(E * INL + arena_used(h), (1ull << 32) - 1);     // every key pair there can be
    u32 *bitmap = nullptr, *wpop = nullptr, *wprefix = nullptr, *ord2_raw = nullptr;
    uint2* list_fn = nullptr;
    u64* d_tot = nullptr;                            // [0] occupied slots, [1] distinct first reads, [2] nnz, [3] long rows (u32)
    POOL(h, P_BITMAP, bitmap, words); POOL(h, P_WPOP, wpop, lines); POOL(h, P_WPREFIX, wprefix, lines);
    POOL(h, P_ROWLEN, ord2_raw, 2 * E); POOL(h, P_LISTFN, list_fn, E);
    uint2* ord2 = reinterpret_cast<uint2*>(ord2_raw);           // (slot, row length) by rank
    POOL(h, P_ORDER, h->order, E);
    h->rank_of_slot = nullptr;                       // (made on demand: ensure_slot_ranks)
    POOL(h, P_INDPTR, h->indptr, E + 1); POOL(h, P_COUNTS, h->counts, E);
    POOL(h, P_INDICES, h->indices, nnz_max); POOL(h, P_DATA, h->data, nnz_max);
    POOL(h, P_TOTALS, d_tot, 8);
    u32* big = nullptr;
    POOL(h, P_MS_X, big, E);
    u32* d_nbig = reinterpret_cast<u32*>(d_tot + 3);
    HIPCHK(h, hipMemsetAsync(bitmap, 0, words * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(d_tot, 0, 8 * sizeof(u64), h->stream));
    int rc = ECB_OK;
    h->list_from_counts = false;
    if (!h->finalized && !h->counted && h->n_reads) {  // the usual case: the counting pass lists the occupied slots as it goes
        POOL(h, P_LIST, h->list, E);
        const CompactSink sink{h->list, E, d_tot, bitmap, total_reads, list_fn};
        rc = ensure_counts(h, &sink);
        if (rc != ECB_OK) return rc;
    } else if (!h->finalized) {
        rc = ensure_counts(h);
        if (rc != ECB_OK) return rc;
    }
    if (!h->list_from_counts) {                        // (counts were made earlier -- a table export, a merge -- or there were none to make)
        if (!h->finalized && h->list_counted) {        // ... and the list with them (a merged key range): no scan of the table
            POOL(h, P_LIST, h->list, E);               // (the pool's buffer, contents and all)
            k_list_fn<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->list, E, h->d_list_n, d_tot, bitmap, total_reads, list_fn);
        } else {
            rc = compact_table_dev(h, d_tot, bitmap, total_reads, list_fn);
            if (rc != ECB_OK) return rc;
        }
    }
    k_popc<<<nblk(lines, TPB), TPB, 0, h->stream>>>(bitmap, lines, wpop);
    rc = excl_scan_dev(h, wpop, lines, wprefix, d_tot + 1);
    if (rc != ECB_OK) return rc;
    k_rank<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->list, list_fn, E, total_reads, bitmap, wprefix, ord2);
    rc = excl_scan_dev(h, ord2_raw + 1, E, h->indptr, d_tot + 2, 2, h->indptr + E);
    if (rc != ECB_OK) return rc;
    k_emit_small<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, ord2, h->order, E, h->arena, h->indptr, h->indices, h->data,
                                                       h->counts, h->cfg.n_loci, h->cfg.n_haplotypes, big, d_nbig, h->ctr);
    // long rows, one wave each: a fixed launch that walks the queue (its length stays on the device)
    k_emit_big<<<(unsigned)std::min<u64>(nblk(E * 64, TPB), 2048), TPB, 0, h->stream>>>(h->table, h->order, big, d_nbig, h->arena, h->indptr,
                                                                                       h->indices, h->data, h->cfg.n_loci, h->cfg.n_haplotypes, h->ctr);
    u64* const tot = h->pin_out->tot;
    HIPCHK(h, hipMemcpyAsync(tot, d_tot, 4 * sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    rc = sync_counters(h);                           // the one wait
    if (rc != ECB_OK) return rc;
    h->n_list = tot[0];
    if (tot[0] != E) return fail(h, ECB_ERR_HIP, "internal: %llu occupied slots but %llu ECs created", (unsigned long long)tot[0], (unsigned long long)E);
    if (tot[1] != E) return fail(h, ECB_ERR_HIP, "internal: %llu distinct first-appearance indices for %llu ECs", (unsigned long long)tot[1], (unsigned long long)E);
    if (tot[2] >= (1ull << 31)) return fail(h, ECB_ERR_LIMIT, "A has more than 2^31-1 non-zeros");
    const u64 nnz = tot[2];
    h->sizes.n_ecs = E; h->sizes.nnz_a = nnz; h->sizes.n_samples = 1; h->sizes.nnz_n = E;
    if (h->cfg.flags & ECB_F_MULTISAMPLE) {
        if (h->adopted) {                            // multi-GPU: the triples arrive through ecb_ms_adopt_triples_device
            h->n_triples = 0; h->sizes.n_samples = 0; h->sizes.nnz_n = 0;
        } else {
            if (h->extra_reads) return fail(h, ECB_ERR_STATE, "multisample across GPUs: adopt the merged ECs (ecb_table_adopt_device), then ecb_ms_adopt_triples_device");
            rc = ensure_slot_ranks(h, E);
            if (rc != ECB_OK) return rc;
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

int ecb_export_firsts_device(ecb_handle* h, void* d_firsts) {
    if (!h || !d_firsts) return ECB_ERR_ARG;
    if (!h->finalized || h->assembled) return fail(h, ECB_ERR_STATE, "first reads are exported from a finalized table");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 E = h->sizes.n_ecs;
    k_export_firsts<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->order, E, (u32*)d_firsts);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_assemble_ranges_device(ecb_handle* h, uint32_t n_pieces, const void* const* d_indptr, const void* const* d_indices,
                               const void* const* d_data, const void* const* d_counts, const void* const* d_firsts,
                               const uint64_t* n_ecs, const uint64_t* nnz, uint64_t total_reads, uint64_t all_alignments,
                               uint64_t valid_alignments, ecb_sizes* out) {
    if (!h || !out) return ECB_ERR_ARG;
    if (h->finalized || h->adopted || h->n_reads || !h->c_rid.empty() || h->n_ecs()) return fail(h, ECB_ERR_STATE, "assembling needs an empty handle");
    if (n_pieces && (!d_indptr || !d_indices || !d_data || !d_counts || !d_firsts || !n_ecs || !nnz)) return fail(h, ECB_ERR_ARG, "null piece lists");
    if (n_pieces > 65535u) return fail(h, ECB_ERR_LIMIT, "at most 65535 pieces");
    u64 E = 0, NNZ = 0;
    for (u32 q = 0; q < n_pieces; ++q) {
        if (n_ecs[q] && (!d_indptr[q] || !d_counts[q] || !d_firsts[q] || (nnz[q] && (!d_indices[q] || !d_data[q])))) return fail(h, ECB_ERR_ARG, "null piece buffers");
        E += n_ecs[q]; NNZ += nnz[q];
    }
    if (E == 0 || valid_alignments == 0) return fail(h, ECB_ERR_EMPTY, "no valid alignments: nothing to build (the reference fails here too)");
    if (E >= (1ull << 31) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^31-2 equivalence classes");
    if (NNZ >= (1ull << 31)) return fail(h, ECB_ERR_LIMIT, "A has more than 2^31-1 non-zeros");
    if (total_reads >= (1ull << 32) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^32-2 reads in total");
    HIPCHK(h, hipSetDevice(h->device));
    free_results(h);
    const u64 words = bitmap_words(total_reads), lines = words / BM_LINE;
    u32 *bitmap = nullptr, *wpop = nullptr, *wprefix = nullptr, *place_raw = nullptr;
    u64* d_tot = nullptr;
    POOL(h, P_BITMAP, bitmap, words); POOL(h, P_WPOP, wpop, lines); POOL(h, P_WPREFIX, wprefix, lines);
    POOL(h, P_ROWLEN, place_raw, 4 * E);
    uint4* place = reinterpret_cast<uint4*>(place_raw);         // {EC within its piece + 1, row length, count, piece} by rank
    POOL(h, P_INDPTR, h->indptr, E + 1); POOL(h, P_COUNTS, h->counts, E);
    POOL(h, P_INDICES, h->indices, std::max<u64>(NNZ, 1)); POOL(h, P_DATA, h->data, std::max<u64>(NNZ, 1));
    POOL(h, P_TOTALS, d_tot, 8);
    HIPCHK(h, hipMemsetAsync(bitmap, 0, words * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(d_tot, 0, 8 * sizeof(u64), h->stream));
    HIPCHK(h, hipMemsetAsync(place_raw, 0, E * 16, h->stream));
    int rc = clear_counters(h);
    if (rc != ECB_OK) return rc;
    std::vector<PieceDesc> desc;
    u64 most = 0;
    for (u32 q = 0, at = 0; q < n_pieces; ++q) {
        if (!n_ecs[q]) continue;
        desc.push_back(PieceDesc{(const u32*)d_firsts[q], (const int*)d_indptr[q], (const int*)d_indices[q], (const int*)d_data[q], (const int*)d_counts[q],
                                 n_ecs[q], nnz[q], at});
        at += n_ecs[q]; most = std::max<u64>(most, n_ecs[q]);
    }
    u64* d_desc_raw = nullptr;
    POOL(h, P_PARTS, d_desc_raw, desc.size() * sizeof(PieceDesc) / sizeof(u64));
    const PieceDesc* d_desc = reinterpret_cast<const PieceDesc*>(d_desc_raw);
    HIPCHK(h, hipMemcpyAsync(d_desc_raw, desc.data(), desc.size() * sizeof(PieceDesc), hipMemcpyHostToDevice, h->stream));
    const dim3 grid((unsigned)nblk(most, TPB), (unsigned)desc.size());
    k_mark_bits<<<grid, TPB, 0, h->stream>>>(d_desc, total_reads, bitmap, h->ctr);
    k_popc<<<nblk(lines, TPB), TPB, 0, h->stream>>>(bitmap, lines, wpop);
    rc = excl_scan_dev(h, wpop, lines, wprefix, d_tot + 1);
    if (rc != ECB_OK) return rc;
    k_piece_place<<<grid, TPB, 0, h->stream>>>(d_desc, E, total_reads, bitmap, wprefix, place, h->ctr);
    rc = excl_scan_dev(h, place_raw + 1, E, h->indptr, d_tot + 2, 4, h->indptr + E);
    if (rc != ECB_OK) return rc;
    k_piece_rows<<<nblk(E, TPB), TPB, 0, h->stream>>>(d_desc, (u32)desc.size(), place, E, h->indptr, h->indices, h->data, h->counts);
    u64 tot[8];
    HIPCHK(h, hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, h->stream));
    h->ctr_synced = false;
    rc = sync_counters(h);                                   // (waits)
    if (rc == ECB_ERR_CONTRACT) return fail(h, rc, "a piece is malformed: a first read beyond the run's reads, or row offsets that are not its own");
    if (rc != ECB_OK) return rc;
    if (tot[1] != E) return fail(h, ECB_ERR_CONTRACT, "the pieces hold %llu ECs but %llu distinct first reads: ranges overlap or a first read is missing",
                                 (unsigned long long)E, (unsigned long long)tot[1]);
    if (tot[2] != NNZ) return fail(h, ECB_ERR_HIP, "internal: %llu non-zeros placed, %llu received", (unsigned long long)tot[2], (unsigned long long)NNZ);
    h->sizes = ecb_sizes{};
    h->sizes.n_ecs = E; h->sizes.nnz_a = NNZ; h->sizes.n_samples = 1; h->sizes.nnz_n = E;
    if (h->cfg.flags & ECB_F_MULTISAMPLE) { h->sizes.n_samples = 0; h->sizes.nnz_n = 0; h->n_triples = 0; h->ms_adopted = false; }   // (N comes with the shards' triples: ecb_ms_adopt_triples_device)
    h->sizes.all_alignments = all_alignments; h->sizes.valid_alignments = valid_alignments; h->sizes.n_reads = total_reads;
    h->finalized = true; h->assembled = true; h->counted = true;
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
    long long* d = nullptr;                            // (the handle's pool: nothing to free on an early return)
    POOL(h, P_EXPORT, d, ns);
    k_range_len<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng, ns, d);
    HIPCHK(h, hipMemcpyAsync(out, d, ns * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_export_range_minmax(ecb_handle* h, int32_t* mn, int32_t* mx) {
    if (!h || !mn || !mx) return ECB_ERR_ARG;
    if (!(h->cfg.flags & ECB_F_RANGES)) return fail(h, ECB_ERR_STATE, "handle was created without ECB_F_RANGES");
    HIPCHK(h, hipSetDevice(h->device));
    const u64 ns = (u64)h->cfg.n_loci * h->cfg.n_haplotypes;
    int* d = nullptr;
    POOL(h, P_EXPORT, d, 2 * ns);
    k_split_minmax<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng, ns, d, d + ns);
    HIPCHK(h, hipMemcpyAsync(mn, d, ns * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(mx, d + ns, ns * 4, hipMemcpyDeviceToHost, h->stream));
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
    if ((h->adopted || h->assembled) && !h->ms_adopted) return fail(h, ECB_ERR_STATE, "multisample across GPUs: no triples adopted yet (ecb_ms_adopt_triples_device)");
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
    if (h->extra_reads || h->assembled) return fail(h, ECB_ERR_STATE, "per-read EC ids are not kept across a multi-GPU merge");
    HIPCHK(h, hipSetDevice(h->device));
    int* d = nullptr;
    POOL(h, P_EXPORT, d, h->n_reads);
    { const int rc_ = ensure_slot_ranks(h, h->sizes.n_ecs); if (rc_ != ECB_OK) return rc_; }
    k_read_ec<<<nblk(h->n_reads, TPB), TPB, 0, h->stream>>>(h->read_slot, h->n_reads, h->rank_of_slot, d);
    HIPCHK(h, hipMemcpyAsync(out, d, h->n_reads * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
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
    if (n_pairs) *n_pairs = h->n_ecs() * INL + arena_used(h);      // (an upper bound: up to INL pairs per EC sit in its slot)
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
    // New ECs may join.  The list of occupied slots is carried along when it can be: the table is empty (the handle that merges one
    // key range), or its list is current and has room (a buffer that had to grow would lose what it holds).
    u32* mlist = nullptr;
    const u64 list_need = h->n_ecs() + total;
    if (h->n_ecs() == 0 || (h->list_counted && h->pool_bytes[ecb_handle::P_LIST] >= list_need * sizeof(u32))) {
        const bool fresh = h->n_ecs() == 0;
        POOL(h, P_LIST, h->list, list_need);
        if (fresh) { POOL(h, P_CNT, h->d_list_n, 1); HIPCHK(h, hipMemsetAsync(h->d_list_n, 0, sizeof(u64), h->stream)); }
        mlist = h->list;
    }
    h->list_counted = mlist != nullptr;
    std::vector<MergeDesc> desc;
    u64 most = 0;
    for (u32 t = 0; t < n_tables; ++t)
        if (n_entries[t]) { desc.push_back(MergeDesc{(const Entry*)d_entries[t], (const uint2*)d_pairs[t], n_entries[t], n_pairs[t]}); most = std::max<u64>(most, n_entries[t]); }
    if (desc.size() > 65535u) return fail(h, ECB_ERR_LIMIT, "at most 65535 tables per call");
    u64* d_desc = nullptr;
    POOL(h, P_PARTS, d_desc, desc.size() * sizeof(MergeDesc) / sizeof(u64));
    HIPCHK(h, hipMemcpyAsync(d_desc, desc.data(), desc.size() * sizeof(MergeDesc), hipMemcpyHostToDevice, h->stream));
    k_merge<<<dim3((unsigned)nblk(most, MERGE_PER_BLOCK), (unsigned)desc.size()), TPB, 0, h->stream>>>(reinterpret_cast<const MergeDesc*>(d_desc), h->table, h->cap - 1,
                                                                                                        h->arena, h->arena_cap, h->ctr,
                                                                                                        mlist, list_need, mlist ? h->d_list_n : nullptr);
    rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    if (h->hctr.n_queue) return fail(h, ECB_ERR_TABLE_FULL, "internal: merge found no slot in a half-empty table");
    return ECB_OK;
}

int ecb_table_rebase_device(ecb_handle* h, void* d_entries, uint64_t n_entries, uint64_t read_base) {
    if (!h || (n_entries && !d_entries)) return ECB_ERR_ARG;
    if (read_base >= (1ull << 32) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^32-2 reads in total");
    if (!n_entries || !read_base) return ECB_OK;
    HIPCHK(h, hipSetDevice(h->device));
    // (no wait: ordered on the handle's stream, ahead of a merge or adopt on this handle; an entry that would pass the limit is
    //  reported by the next call that reads the device counters, as ECB_ERR_CONTRACT)
    k_rebase<<<nblk(n_entries, TPB), TPB, 0, h->stream>>>((Entry*)d_entries, n_entries, (u32)read_base, (u32)((1ull << 32) - 1 - read_base), h->ctr);
    h->ctr_synced = false;
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
    if (E) {
        u64* big = nullptr;                            // long keys: (first pair, slot << 32 | length) + their number behind the list
        POOL(h, P_BIG, big, 2 * E + 1);
        u32* d_nbig = reinterpret_cast<u32*>(big + 2 * E);
        HIPCHK(h, hipMemsetAsync(d_nbig, 0, sizeof(u64), h->stream));
        k_parts_export<<<nblk(E, PARTS_PER_BLOCK), TPB, 0, h->stream>>>(h->table, h->list, E, h->arena, n_parts, d_cnt, d_cnt + 2 * n_parts,
                                                                   (Entry*)d_entries, (uint2*)d_pairs, (u32)read_base, big, d_nbig);
        u32 n_big = 0;
        HIPCHK(h, hipMemcpyAsync(&n_big, d_nbig, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));      // (cur lives on this stack frame)
        if (n_big) k_parts_sort_big<<<nblk((u64)n_big * 64, TPB), TPB, 0, h->stream>>>(h->table, h->arena, big, n_big, (uint2*)d_pairs);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
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
    h->adopted = true; h->counted = true; h->list_counted = false;
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
        k_adopt<<<nblk(n_entries[t], TPB), TPB, 0, h->stream>>>((const Entry*)d_entries[t], n_entries[t], (const uint2*)d_pairs[t], h->table, have, (u32)top);
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
    if (h->assembled) k_row_keys<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->indptr, h->indices, h->data, E, (u64*)d_keys);     // (no table behind an assembled result)
    else k_export_keys<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->order, E, (u64*)d_keys);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_ms_local_triples_device(ecb_handle* h, const void* d_keys, const void* d_indptr_a, const void* d_indices_a, const void* d_data_a,
                                uint64_t n_ecs, uint64_t read_base, void* d_key, void* d_count, void* d_first, uint64_t* n_triples) {
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
    if (!d_keys || !d_indptr_a || !d_indices_a || !d_data_a || !d_key || !d_count || !d_first) return fail(h, ECB_ERR_ARG, "null buffers");
    if (read_base + h->n_reads >= (1ull << 32) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^32-2 reads in total");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    u32* grank = nullptr;
    POOL(h, P_MS_GRANK, grank, h->cap);
    HIPCHK(h, hipMemsetAsync(grank, 0xFF, h->cap * sizeof(u32), h->stream));
    k_set_global_rank<<<nblk(n_ecs, TPB), TPB, 0, h->stream>>>((const u64*)d_keys, (const int*)d_indptr_a, (const int*)d_indices_a, (const int*)d_data_a,
                                                              n_ecs, h->table, h->cap - 1, h->arena, grank);
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
    if (!h->finalized || !(h->adopted || h->assembled)) return fail(h, ECB_ERR_STATE, "triples are adopted by the finalized handle that adopted the merged ECs (or assembled their ranges)");
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
    u64 nt = 0;
    if (tot) {
        k_ms_swz_keys<<<nblk(tot, TPB), TPB, 0, h->stream>>>(keys, tot);      // (sorted as (EC, cell, file), like a handle's own triples)
        k_iota<<<nblk(tot, TPB), TPB, 0, h->stream>>>((int*)vals, tot);
        {
            u64* kk[2] = {keys, keys2}; u32* vv[2] = {vals, vals2};
            int where = 0;
            const int rc0 = handle_sort(h, kk, vv, tot, &where);
            if (rc0 != ECB_OK) return rc0;
            keys2 = kk[where]; vals2 = vv[where];
        }
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

int ecb_ms_filter(ecb_handle* h, uint32_t n_cells, int64_t minimum_count, ecb_ms_sizes* out) {
    if (!h || !out) return ECB_ERR_ARG;
    if (!h->finalized || !(h->cfg.flags & ECB_F_MULTISAMPLE)) return fail(h, ECB_ERR_STATE, "no multisample result");
    if ((h->adopted || h->assembled) && !h->ms_adopted) return fail(h, ECB_ERR_STATE, "multisample across GPUs: no triples adopted yet (ecb_ms_adopt_triples_device)");
    if (!n_cells || n_cells > (1u << ECB_CELL_BITS)) return fail(h, ECB_ERR_ARG, "n_cells out of range");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const u64 T = h->n_triples, E = h->sizes.n_ecs;
    const u64 min_count = minimum_count <= 0 ? 1ull : (u64)minimum_count;          // bam_utils_multisample.py:596-597
    if (!T) return fail(h, ECB_ERR_EMPTY, "no (EC, cell) counts: nothing to filter");
    // The triples are sorted by (EC, cell, file) (ms_reduce / ecb_ms_adopt_triples_device).  Everything read-sized below is a
    // linear pass over them; the only sort left is the one that turns the surviving (EC, cell) pairs -- in EC order -- into
    // the columns of N, on the digits of the cell alone.  The triple-sized buffers are the ones ms_reduce sorted in (its pool).
    u32 *x = nullptr, *flag = nullptr, *pos = nullptr, *v0 = nullptr, *v1 = nullptr;
    u64 *k0 = nullptr, *k1 = nullptr;
    POOL(h, P_MS_TMP, x, 3 * T);
    POOL(h, P_MS_FLAG, flag, std::max<u64>(T, n_cells)); POOL(h, P_MS_POS, pos, std::max<u64>(T, n_cells));
    POOL(h, P_MS_KEYS, k0, T); POOL(h, P_MS_KEYS2, k1, T); POOL(h, P_MS_VALS, v0, T); POOL(h, P_MS_VALS2, v1, T);
    u32 *ec = x, *meta = x + T, *cnt = x + 2 * T;
    k_ms_split<<<nblk(T, TPB), TPB, 0, st>>>(h->ms_okey, h->ms_ostart, h->ms_ocount, T, ec, meta, cnt);
    const u32* first = h->ms_ofirst;
    Scratch sc;                                              // (cell- and EC-sized scratch: small)
    u64 *total = sc.get<u64>(n_cells), *cellkey = sc.get<u64>(n_cells);
    u32 *firstfile = sc.get<u32>(n_cells), *seg = sc.get<u32>(E + 1), *keep_ec = sc.get<u32>(E), *new_rank = sc.get<u32>(E), *new_cell = sc.get<u32>(n_cells);
    u32* d_err = sc.get<u32>(1);
    if (!total || !cellkey || !firstfile || !seg || !keep_ec || !new_rank || !new_cell || !d_err) return fail(h, ECB_ERR_HIP, "out of device memory");
    HIPCHK(h, hipMemsetAsync(total, 0, (u64)n_cells * 8, st));
    HIPCHK(h, hipMemsetAsync(cellkey, 0xFF, (u64)n_cells * 8, st));
    HIPCHK(h, hipMemsetAsync(firstfile, 0xFF, (u64)n_cells * 4, st));
    HIPCHK(h, hipMemsetAsync(keep_ec, 0, E * 4, st));
    HIPCHK(h, hipMemsetAsync(d_err, 0, 4, st));
    // 1. per cell: reads and first file
    {
        const bool lds = n_cells <= MSF_LDS_CELLS;
        const unsigned grid = (unsigned)std::min<u64>(lds ? 512 : 2048, nblk(T, TPB_MSF));
        k_msf2_cells<<<grid, TPB_MSF, lds ? (size_t)n_cells * 8 : 0, st>>>(meta, cnt, T, n_cells, total, firstfile, d_err);
    }
    // 2. per EC: its first appearance in every file; where every cell enters the cell order; whether the EC keeps a cell
    k_msf2_seg<<<nblk(T, TPB), TPB, 0, st>>>(ec, T, (u32)E, seg, d_err);
    {
        u32* big = sc.get<u32>(E + 1);                       // ECs of more than MSF_SMALL triples, their number behind the list
        const u64 max_giant = T / MSF_GIANT + 1;             // ... and of more than MSF_GIANT, with their per-file first appearances
        u32 *giant = sc.get<u32>(max_giant + 1), *gfec = sc.get<u32>(max_giant * MSF_FILES);
        if (!big || !giant || !gfec) return fail(h, ECB_ERR_HIP, "out of device memory");
        HIPCHK(h, hipMemsetAsync(big + E, 0, 4, st));
        HIPCHK(h, hipMemsetAsync(giant + max_giant, 0, 4, st));
        HIPCHK(h, hipMemsetAsync(gfec, 0xFF, max_giant * MSF_FILES * 4, st));
        k_msf2_ecs_small<<<nblk(E, TPB), TPB, 0, st>>>(meta, first, seg, (u32)E, total, firstfile, min_count, cellkey, keep_ec, big, big + E, giant, giant + max_giant);
        k_msf2_ecs_big<<<(unsigned)std::min<u64>(std::max<u64>(E / MSF_SMALL, 1), 4096), TPB, 0, st>>>(meta, first, seg, big, big + E, total, firstfile, min_count, cellkey, keep_ec);
        k_msf2_giant_fec<<<1024, TPB, 0, st>>>(meta, first, seg, giant, giant + max_giant, total, min_count, gfec, keep_ec);
        k_msf2_giant_offer<<<1024, TPB, 0, st>>>(meta, first, seg, giant, giant + max_giant, firstfile, gfec, cellkey);
    }
    // 3. cell order: by (first appearance of the EC in the cell's first file, first read), then -- stable -- by that file
    int rc, where = 0;
    k_msf2_cellflag<<<nblk(n_cells, TPB), TPB, 0, st>>>(total, n_cells, flag);
    u64 C = 0;
    rc = excl_scan(h, flag, n_cells, pos, &C); if (rc != ECB_OK) return rc;
    {
        u32 err = 0;
        HIPCHK(h, hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (err & 1u) return fail(h, ECB_ERR_CONTRACT, "triple with an EC id beyond the finalized ECs");
        if (err & 2u) return fail(h, ECB_ERR_CONTRACT, "a read's cell id is not below n_cells");
    }
    if (!C) return fail(h, ECB_ERR_EMPTY, "no (EC, cell) counts: nothing to filter");
    u32 *cell_id = sc.get<u32>(C), *corder = sc.get<u32>(C), *corder2 = sc.get<u32>(C), *cflag = sc.get<u32>(C), *cpos = sc.get<u32>(C);
    u64 *ctotal = sc.get<u64>(C), *bhi = sc.get<u64>(C), *blo = sc.get<u64>(C), *ck0 = sc.get<u64>(C), *ck1 = sc.get<u64>(C);
    if (!cell_id || !corder || !corder2 || !cflag || !cpos || !ctotal || !bhi || !blo || !ck0 || !ck1) return fail(h, ECB_ERR_HIP, "out of device memory");
    k_msf2_celllist<<<nblk(n_cells, TPB), TPB, 0, st>>>(total, firstfile, cellkey, flag, pos, n_cells, cell_id, ctotal, bhi, blo);
    k_msf_iota<<<nblk(C, TPB), TPB, 0, st>>>(corder, C);
    HIPCHK(h, hipMemcpyAsync(ck0, blo, C * 8, hipMemcpyDeviceToDevice, st));
    u32* ord = nullptr;
    { u64* kk[2] = {ck0, ck1}; u32* vv[2] = {corder, corder2}; rc = handle_sort(h, kk, vv, C, &where); if (rc != ECB_OK) return rc;
      u32* o1 = vv[where]; u32* o2 = vv[where ^ 1]; u64* ka = kk[where ^ 1]; u64* kb = kk[where];
      k_msf_gather64<<<nblk(C, TPB), TPB, 0, st>>>(bhi, o1, C, ka);
      u64* kk2[2] = {ka, kb}; u32* vv2[2] = {o1, o2}; rc = handle_sort(h, kk2, vv2, C, &where); if (rc != ECB_OK) return rc;
      ord = vv2[where]; }
    k_msf_keepflag<<<nblk(C, TPB), TPB, 0, st>>>(ord, ctotal, C, min_count, cflag);
    u64 S = 0;
    rc = excl_scan(h, cflag, C, cpos, &S); if (rc != ECB_OK) return rc;
    if (!S) return fail(h, ECB_ERR_EMPTY, "no cell reaches the minimum count");
    POOL(h, P_F_CELLS, h->f_cells, S);
    HIPCHK(h, hipMemsetAsync(new_cell, 0xFF, (u64)n_cells * 4, st));
    k_msf_newcell<<<nblk(C, TPB), TPB, 0, st>>>(ord, cflag, cpos, cell_id, C, new_cell, h->f_cells);
    // 4. ECs that keep a cell, re-ranked (bam_utils_multisample.py:611-636)
    u64 E2 = 0, K = 0;
    rc = excl_scan(h, keep_ec, E, new_rank, &E2); if (rc != ECB_OK) return rc;
    // 5. N as CSC over (kept EC, kept cell): the surviving (EC, cell) pairs, the reads of their files added up (:737-791), come
    //    in EC order; a stable sort on the cell's digits makes them the columns
    k_msf2_pairflag<<<nblk(T, TPB), TPB, 0, st>>>(ec, meta, T, new_cell, flag);
    rc = excl_scan(h, flag, T, pos, &K); if (rc != ECB_OK) return rc;
    if (K >= (1ull << 31)) return fail(h, ECB_ERR_LIMIT, "N has more than 2^31-1 non-zeros");
    k_msf2_pairs<<<nblk(T, TPB), TPB, 0, st>>>(ec, meta, cnt, flag, pos, T, new_cell, new_rank, k0, v0);
    const u64 nnz_n = K;
    {
        u32 sbits = 1;
        while (sbits < 32 && (1ull << sbits) < S) ++sbits;
        SortScratch ss{};
        u64* tot = nullptr;
        POOL(h, P_RS_HIST, ss.hist, rs_words(K)); POOL(h, P_RS_OFFS, ss.offs, RS_AUX_WORDS);
        POOL(h, P_RS_SUMS, ss.sums, rs_scan_blocks(K) + 8); POOL(h, P_TOTALS, tot, 8);
        ss.d_word = tot + 6;
        u64* kk[2] = {k0, k1}; u32* vv[2] = {v0, v1};
        const hipError_t e = radix_sort_pairs64(st, kk, vv, K, ss, &where, ((1ull << sbits) - 1ull) << 32);
        if (e != hipSuccess) return fail(h, ECB_ERR_HIP, "radix sort: %s", hipGetErrorString(e));
        POOL(h, P_F_IPN, h->f_ipn, S + 1); POOL(h, P_F_IXN, h->f_ixn, nnz_n); POOL(h, P_F_DAN, h->f_dan, nnz_n);
        k_msf2_nout<<<nblk(K, TPB), TPB, 0, st>>>(kk[where], vv[where], K, h->f_ixn, h->f_dan);
        k_msf2_nptr<<<nblk(S + 1, TPB), TPB, 0, st>>>(kk[where], K, (u32)S, h->f_ipn);
    }
    // 6. the rows of A of the ECs that are left
    u32* rowlen2 = sc.get<u32>(E2 + 1);
    if (!rowlen2) return fail(h, ECB_ERR_HIP, "out of device memory");
    POOL(h, P_F_IPA, h->f_ipa, E2 + 1);
    k_msf_rowlen<<<nblk(E, TPB), TPB, 0, st>>>(h->indptr, keep_ec, new_rank, E, rowlen2);
    u64 nnz_a = 0;
    rc = excl_scan(h, rowlen2, E2, reinterpret_cast<u32*>(h->f_ipa), &nnz_a); if (rc != ECB_OK) return rc;
    { const u32 last = (u32)nnz_a; HIPCHK(h, hipMemcpyAsync(h->f_ipa + E2, &last, 4, hipMemcpyHostToDevice, st)); HIPCHK(h, hipStreamSynchronize(st)); }
    POOL(h, P_F_IXA, h->f_ixa, nnz_a); POOL(h, P_F_DAA, h->f_daa, nnz_a);
    k_msf_rows<<<nblk(E, TPB), TPB, 0, st>>>(h->indptr, h->indices, h->data, keep_ec, new_rank, reinterpret_cast<const u32*>(h->f_ipa), E, h->f_ixa, h->f_daa);
    HIPCHK(h, hipStreamSynchronize(st));
    h->msf.n_cells_seen = C; h->msf.n_cells_kept = S; h->msf.n_ecs_kept = E2; h->msf.nnz_a = nnz_a; h->msf.nnz_n = nnz_n;
    h->ms_filtered = true;
    *out = h->msf;
    return ECB_OK;
}

int ecb_ms_export(ecb_handle* h, uint32_t* kept_cells, int32_t* ia, int32_t* ja, int32_t* da, int32_t* in_, int32_t* jn, int32_t* dn) {
    if (!h) return ECB_ERR_ARG;
    if (!h->ms_filtered) return fail(h, ECB_ERR_STATE, "ecb_ms_export before ecb_ms_filter");
    HIPCHK(h, hipSetDevice(h->device));
    const ecb_ms_sizes& m = h->msf;
    if (kept_cells) HIPCHK(h, hipMemcpyAsync(kept_cells, h->f_cells, m.n_cells_kept * 4, hipMemcpyDeviceToHost, h->stream));
    if (ia) HIPCHK(h, hipMemcpyAsync(ia, h->f_ipa, (m.n_ecs_kept + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    if (ja) HIPCHK(h, hipMemcpyAsync(ja, h->f_ixa, m.nnz_a * 4, hipMemcpyDeviceToHost, h->stream));
    if (da) HIPCHK(h, hipMemcpyAsync(da, h->f_daa, m.nnz_a * 4, hipMemcpyDeviceToHost, h->stream));
    if (in_) HIPCHK(h, hipMemcpyAsync(in_, h->f_ipn, (m.n_cells_kept + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    if (jn) HIPCHK(h, hipMemcpyAsync(jn, h->f_ixn, m.nnz_n * 4, hipMemcpyDeviceToHost, h->stream));
    if (dn) HIPCHK(h, hipMemcpyAsync(dn, h->f_dan, m.nnz_n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
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

// the stream kernel the last batch launched, as rocprofv3 names it (k_stream.inc is compiled more than once: which compilation a
// batch takes is decided per batch, in process_batch)
const char* ecb_profile_kernel(const ecb_handle* h) { return h ? h->last_kernel : ""; }

}  // extern "C"

// ---- f-2 conversions (stateless; scratch is allocated per call: this is not the hot path) ------------------------
namespace {
int cv_scan(hipStream_t st, const u32* in, u64 n, u32* out, u64* total, Scratch& sc) {
    u32* sums = sc.get<u32>(scan_words(n) + 4);
    if (!sums) return ECB_ERR_HIP;
    u64* grand = reinterpret_cast<u64*>(sums + scan_words(n));
    if (scan_launch(st, in, n, out, sums, grand) != hipSuccess) return ECB_ERR_HIP;
    if (hipMemcpyAsync(total, grand, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return ECB_ERR_HIP;
    return hipStreamSynchronize(st) == hipSuccess ? ECB_OK : ECB_ERR_HIP;
}
}  // namespace

// Scratch of the stateless conversions, kept per device between calls (grown on demand; ecb_release_scratch frees it): a
// config-3-sized conversion needs ~0.4 GB in a dozen buffers, and a dozen hipMallocs cost more than its kernels do.
namespace {
struct CvScratch {
    enum { KEYS0, KEYS1, VALS0, VALS1, HIST, OFFS, SUMS, SUMS2, BLK, SCAN, HEAD, WORDS, X0, X1, X2, X3, N };
    void* p[N] = {}; u64 bytes[N] = {};
    template <class T> T* get(int id, u64 count) {
        const u64 need = std::max<u64>(count, 1) * sizeof(T);
        if (bytes[id] < need) {
            if (p[id]) hipFree(p[id]);
            p[id] = nullptr; bytes[id] = 0;
            if (hipMalloc(&p[id], need + need / 8) != hipSuccess) return nullptr;
            bytes[id] = need + need / 8;
        }
        return reinterpret_cast<T*>(p[id]);
    }
    void release() { for (int i = 0; i < N; ++i) { if (p[i]) hipFree(p[i]); p[i] = nullptr; bytes[i] = 0; } }
};
constexpr int CV_MAX_DEV = 64;
CvScratch g_cv[CV_MAX_DEV];
std::mutex g_cv_lock;
// exclusive scan on `st`, nothing waits: sums = scan_words(n) words of scratch, the total (64 bits) lands in *d_grand
void cv_scan_queue(hipStream_t st, const u32* in, u64 n, u32* out, u32* sums, u64* d_grand) { (void)scan_launch(st, in, n, out, sums, d_grand); }
}  // namespace

extern "C" int ecb_release_scratch(int device) {
    if (device < 0 || device >= CV_MAX_DEV) return ECB_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, ECB_ERR_NO_DEVICE, "no such device");
    std::lock_guard<std::mutex> g(g_cv_lock);
    g_cv[device].release();
    return ECB_OK;
}

extern "C" int ecb_csr_to_hapcsc_device(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const void* d_indptr,
                                        const void* d_indices, const void* d_data, void* d_cscptr, void* d_cscidx,
                                        uint64_t* total) {
    if (!d_indptr || !total || !n_ecs || !n_loci || !n_haps || n_haps > 31) return fail(nullptr, ECB_ERR_ARG, "bad argument");
    if ((u64)n_haps * n_loci >= (1ull << 32)) return fail(nullptr, ECB_ERR_LIMIT, "haplotypes x loci does not fit 32 bits");
    if (device < 0 || device >= CV_MAX_DEV || hipSetDevice(device) != hipSuccess) return fail(nullptr, ECB_ERR_NO_DEVICE, "no such device");
    hipStream_t st = nullptr;
    int nnz_i = 0;
    if (hipMemcpy(&nnz_i, (const int*)d_indptr + n_ecs, 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "read nnz");
    if (nnz_i < 0) return fail(nullptr, ECB_ERR_CONTRACT, "malformed CSR: negative row pointer");
    const u64 nnz = (u64)nnz_i;
    if (nnz && (!d_indices || !d_data)) return fail(nullptr, ECB_ERR_ARG, "bad argument");       // (a matrix without non-zeros has no arrays to point at)
    std::lock_guard<std::mutex> guard(g_cv_lock);
    CvScratch& S = g_cv[device];
    u64* words = S.get<u64>(CvScratch::WORDS, 4);                  // [0] set bits, [1] the scan's total, [2] error bits
    if (!words) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    if (!d_cscidx || !d_cscptr) {                                  // the first call: how many row indices there will be
        u64 tot = 0;
        if (hipMemsetAsync(words, 0, 8, st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "memset");
        if (nnz) k_cv_bits<<<(unsigned)std::min<u64>(2048, nblk(nnz, TPB)), TPB, 0, st>>>((const int*)d_data, nnz, words);
        if (hipMemcpy(&tot, words, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "count");
        if (tot >= (1ull << 32)) return fail(nullptr, ECB_ERR_LIMIT, "more than 2^32-1 set haplotype bits");
        *total = tot;
        return ECB_OK;
    }
    const u64 nc = (u64)n_haps * (n_loci + 1);
    if (!nnz) {
        if (hipMemset(d_cscptr, 0, nc * 4) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "memset");
        *total = 0;
        return ECB_OK;
    }
    const u32 nb = (u32)nblk(nnz, CVB);
    u64 *k0 = S.get<u64>(CvScratch::KEYS0, nnz), *k1 = S.get<u64>(CvScratch::KEYS1, nnz);
    u32 *v0 = S.get<u32>(CvScratch::VALS0, nnz), *v1 = S.get<u32>(CvScratch::VALS1, nnz);
    SortScratch ss{S.get<u32>(CvScratch::HIST, rs_words(nnz)), S.get<u32>(CvScratch::OFFS, RS_AUX_WORDS),
                   S.get<u32>(CvScratch::SUMS, rs_scan_blocks(nnz) + 8), words + 3};
    u32 *blk = S.get<u32>(CvScratch::BLK, (u64)n_haps * nb), *scan = S.get<u32>(CvScratch::SCAN, (u64)n_haps * nb);
    u32 *sums2 = S.get<u32>(CvScratch::SUMS2, scan_words((u64)n_haps * nb) + 2), *headval = S.get<u32>(CvScratch::HEAD, (u64)n_haps * n_loci);
    if (!k0 || !k1 || !v0 || !v1 || !ss.hist || !ss.offs || !ss.sums || !blk || !scan || !sums2 || !headval) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    if (hipMemsetAsync(words, 0, 24, st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "memset");
    k_cv_keys<<<nblk(n_ecs, TPB), TPB, 0, st>>>((const int*)d_indptr, n_ecs, (const int*)d_indices, (const int*)d_data, nnz, n_loci, n_haps,
                                               k0, v0, reinterpret_cast<u32*>(words + 2));
    u32 lbits = 1;
    while (lbits < 32 && ((u64)1 << lbits) < n_loci) ++lbits;
    u64* kk[2] = {k0, k1}; u32* vv[2] = {v0, v1};
    int where = 0;
    // stable sort on the locus alone: ECs stay ascending within a column, as scipy's tocsc() leaves them
    if (radix_sort_pairs64(st, kk, vv, nnz, ss, &where, ((1ull << lbits) - 1ull) << 32) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "sort");
    const u64* keys = kk[where]; const u32* masks = vv[where];
    k_cv_cnt<<<nb, TPB, 0, st>>>(masks, nnz, n_haps, nb, blk);
    cv_scan_queue(st, blk, (u64)n_haps * nb, scan, sums2, words + 1);
    u64 back[3] = {0, 0, 0};
    if (hipMemcpy(back, words, 24, hipMemcpyDeviceToHost) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csr -> csc (count)");
    if (back[2]) return fail(nullptr, ECB_ERR_CONTRACT, "malformed CSR: row pointers out of order, a locus beyond n_loci, or a mask that is zero or beyond n_haplotypes");
    if (back[1] >= (1ull << 32)) return fail(nullptr, ECB_ERR_LIMIT, "more than 2^32-1 set haplotype bits");
    *total = back[1];
    k_cv_emit<<<nb, TPB, 0, st>>>(keys, masks, nnz, n_haps, n_loci, nb, scan, (int*)d_cscidx, headval);
    k_cv_ptr<<<nblk((u64)n_loci + 1, TPB), TPB, 0, st>>>(keys, nnz, n_loci, n_haps, nb, scan, words + 1, headval, (int*)d_cscptr);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csr -> csc");
    return ECB_OK;
}

namespace {
// the general case: lists in any order, an EC more than once in a list -- every row index becomes a 64-bit key, all of them are sorted
int hapcsc_to_csr_general(hipStream_t st, u32 n_ecs, u32 n_loci, u32 n_haps, const void* d_cscptr, const void* d_cscidx, u64 total,
                          const std::vector<u64>& hs, void* d_indptr, void* d_indices, void* d_data, uint64_t* nnz_out) {
    Scratch sc;
    u64 *d_hs = sc.get<u64>(n_haps + 1), *keys = sc.get<u64>(total), *keys2 = sc.get<u64>(total);
    u32 *vals = sc.get<u32>(total), *vals2 = sc.get<u32>(total), *flag = sc.get<u32>(total), *pos = sc.get<u32>(total);
    if (!d_hs || !keys || !keys2 || !vals || !vals2 || !flag || !pos) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    hipMemcpy(d_hs, hs.data(), (n_haps + 1) * 8, hipMemcpyHostToDevice);
    k_cv_back_expand<<<nblk(total, TPB), TPB, 0, st>>>((const int*)d_cscptr, (const int*)d_cscidx, total, n_loci, n_haps, d_hs, keys, vals);
    {
        SortScratch ss{sc.get<u32>(rs_words(total)), sc.get<u32>(RS_AUX_WORDS), sc.get<u32>(rs_scan_blocks(total) + 8), sc.get<u64>(1)};
        if (!ss.hist || !ss.offs || !ss.sums || !ss.d_word) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
        u64* kk[2] = {keys, keys2}; u32* vv[2] = {vals, vals2};
        int where = 0;
        if (radix_sort_pairs64(st, kk, vv, total, ss, &where) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "sort");
        keys2 = kk[where]; vals2 = vv[where];
    }
    k_ms_heads<<<nblk(total, TPB), TPB, 0, st>>>(keys2, total, flag);
    u64 nnz = 0;
    if (cv_scan(st, flag, total, pos, &nnz, sc) != ECB_OK) return fail(nullptr, ECB_ERR_HIP, "scan");
    k_cv_back_emit<<<nblk(total, TPB), TPB, 0, st>>>(keys2, vals2, flag, pos, total, n_loci, (int*)d_indices, (int*)d_data);
    k_cv_back_rowptr<<<nblk((u64)n_ecs + 1, TPB), TPB, 0, st>>>(keys2, pos, total, nnz, n_ecs, n_loci, (int*)d_indptr);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csc -> csr");
    *nnz_out = nnz;
    return ECB_OK;
}
}  // namespace

extern "C" int ecb_hapcsc_to_csr_device(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const void* d_cscptr,
                                        const void* d_cscidx, uint64_t total, void* d_indptr, void* d_indices, void* d_data,
                                        uint64_t* nnz_out) {
    if (!d_cscptr || !d_cscidx || !d_indptr || !d_indices || !d_data || !nnz_out || !n_ecs || !n_loci || !n_haps || n_haps > 31 || !total)
        return fail(nullptr, ECB_ERR_ARG, "bad argument");
    if (total >= (1ull << 32)) return fail(nullptr, ECB_ERR_LIMIT, "more than 2^32-1 row indices");
    if (device < 0 || device >= CV_MAX_DEV || hipSetDevice(device) != hipSuccess) return fail(nullptr, ECB_ERR_NO_DEVICE, "no such device");
    hipStream_t st = nullptr;
    std::lock_guard<std::mutex> guard(g_cv_lock);
    CvScratch& S = g_cv[device];
    // start of every haplotype's block = running sum of its last column pointer
    u64* words = S.get<u64>(CvScratch::WORDS, 40);                 // [0] error bits  [1] the scan's total  [3] the sort's  [4 ..) last pointers, then block starts
    if (!words) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    int* d_last = reinterpret_cast<int*>(words + 4);
    std::vector<int> last(n_haps);
    k_cvb_last<<<1, 64, 0, st>>>((const int*)d_cscptr, n_loci, n_haps, d_last);
    if (hipMemcpy(last.data(), d_last, n_haps * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "read csc pointers");
    std::vector<u64> hs(n_haps + 1, 0);
    for (u32 h = 0; h < n_haps; ++h) {
        if (last[h] < 0) return fail(nullptr, ECB_ERR_CONTRACT, "malformed CSC: negative column pointer");
        hs[h + 1] = hs[h] + (u64)last[h];
    }
    if (hs[n_haps] != total) return fail(nullptr, ECB_ERR_ARG, "total does not match the column pointers");
    u32 ebits = 0;
    while (ebits < 32 && ((u64)1 << ebits) < n_ecs) ++ebits;
    u64* d_hs = S.get<u64>(CvScratch::X0, n_haps + 1);
    u32 *Ssum = S.get<u32>(CvScratch::X1, (u64)n_loci + 1);
    u32 *colcnt = S.get<u32>(CvScratch::HEAD, 5ull * n_loci + 8), *sums = S.get<u32>(CvScratch::SUMS2, scan_words(n_loci) + 2);
    if (!d_hs || !Ssum || !colcnt || !sums) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    u32 *colbase = colcnt + n_loci, *colcur = colbase + n_loci, *pieces = colcur + n_loci, *pbase = pieces + n_loci;
    u32* d_err = reinterpret_cast<u32*>(words);
    if (hipMemcpyAsync(d_hs, hs.data(), (n_haps + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(words, 0, 24, st) != hipSuccess || hipMemsetAsync(colcnt, 0, 3ull * n_loci * 4, st) != hipSuccess)
        return fail(nullptr, ECB_ERR_HIP, "csc -> csr (set-up)");
    k_cvu_colsum<<<nblk((u64)n_loci + 1, TPB), TPB, 0, st>>>((const int*)d_cscptr, n_loci, n_haps, Ssum, d_err);
    k_cvu_pieces<<<nblk(n_loci, TPB), TPB, 0, st>>>(Ssum, n_loci, pieces);
    cv_scan_queue(st, pieces, n_loci, pbase, sums, words + 2);
    u64 back[3] = {0, 0, 0};
    if (hipMemcpy(back, words, 24, hipMemcpyDeviceToHost) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csc -> csr (pointers)");
    if ((u32)back[0]) return fail(nullptr, ECB_ERR_CONTRACT, "malformed CSC: column pointers do not start at zero or go backwards");
    const u64 n_items = back[2];
    if (n_items == 0 || n_items >= (1ull << 31)) return fail(nullptr, ECB_ERR_LIMIT, "csc -> csr: pieces of work");
    u32 *item_col = S.get<u32>(CvScratch::X2, n_items + 1), *bnd = S.get<u32>(CvScratch::X3, (n_items + 1) * n_haps);
    if (!item_col || !bnd) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    k_cvu_items<<<nblk(n_loci, TPB), TPB, 0, st>>>(pieces, pbase, n_loci, item_col);
    k_cvu_bounds<<<nblk(n_items * n_haps, TPB), TPB, 0, st>>>((const int*)d_cscptr, (const int*)d_cscidx, d_hs, pieces, pbase, item_col, (u32)n_items,
                                                             n_loci, n_haps, n_ecs, bnd);
    k_cvu_union<false><<<(unsigned)n_items, CVU_TPB, 0, st>>>((const int*)d_cscptr, (const int*)d_cscidx, d_hs, pieces, pbase, item_col, bnd, (u32)n_items,
                                                              n_loci, n_haps, n_ecs, colcnt, colbase, colcur, 0, nullptr, nullptr, d_err);
    cv_scan_queue(st, colcnt, n_loci, colbase, sums, words + 1);
    if (hipMemcpy(back, words, 16, hipMemcpyDeviceToHost) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csc -> csr (union)");
    if ((u32)back[0] & CVB_ERR_EC) return fail(nullptr, ECB_ERR_CONTRACT, "malformed CSC: a row index beyond the number of ECs");
    if ((u32)back[0] & CVB_ERR_ORDER)                              // a long column whose lists are not ascending: sort everything
        return hapcsc_to_csr_general(st, n_ecs, n_loci, n_haps, d_cscptr, d_cscidx, total, hs, d_indptr, d_indices, d_data, nnz_out);
    const u64 nnz = back[1];
    u64 *k0 = S.get<u64>(CvScratch::KEYS0, nnz), *k1 = S.get<u64>(CvScratch::KEYS1, nnz);
    u32 *v0 = S.get<u32>(CvScratch::VALS0, nnz), *v1 = S.get<u32>(CvScratch::VALS1, nnz);
    SortScratch ss{S.get<u32>(CvScratch::HIST, rs_words(nnz)), S.get<u32>(CvScratch::OFFS, RS_AUX_WORDS),
                   S.get<u32>(CvScratch::SUMS, rs_scan_blocks(nnz) + 8), words + 3};
    if (!k0 || !k1 || !v0 || !v1 || !ss.hist || !ss.offs || !ss.sums) return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    k_cvu_union<true><<<(unsigned)n_items, CVU_TPB, 0, st>>>((const int*)d_cscptr, (const int*)d_cscidx, d_hs, pieces, pbase, item_col, bnd, (u32)n_items,
                                                             n_loci, n_haps, n_ecs, colcnt, colbase, colcur, nnz, k0, v0, d_err);
    u64* kk[2] = {k0, k1}; u32* vv[2] = {v0, v1};
    int where = 0;
    // stable, on the EC alone: what left column by column arrives row by row with its loci ascending
    if (radix_sort_pairs64(st, kk, vv, nnz, ss, &where, (ebits >= 32 ? 0xFFFFFFFFull : (1ull << ebits) - 1ull) << 32) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "sort");
    k_cvb_out<<<nblk(nnz, TPB), TPB, 0, st>>>(kk[where], vv[where], nnz, (int*)d_indices, (int*)d_data);
    k_cvb_rowptr<<<nblk((u64)n_ecs + 1, TPB), TPB, 0, st>>>(kk[where], nnz, n_ecs, (int*)d_indptr);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(nullptr, ECB_ERR_HIP, "csc -> csr");
    *nnz_out = nnz;
    return ECB_OK;
}

// ---- the same two conversions from and to HOST arrays: the library allocates, fills and frees its own device buffers, so a host
// that only converts files (alntools ec2emase / emase2ec, the .h5 writer of bam2emase: bin_utils.py:979-1028) needs no device
// allocator of its own -- and the Python drop-in no PyTorch on that path.
namespace {
struct DevBuf {                        // a device buffer that goes away with its scope
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int take(u64 bytes) { return hipMalloc(&p, std::max<u64>(bytes, 4)) == hipSuccess ? ECB_OK : ECB_ERR_HIP; }
};
}  // namespace

extern "C" int ecb_csr_to_hapcsc(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const int32_t* indptr, const int32_t* indices,
                                 const int32_t* data, int32_t* csc_indptr, int32_t* csc_indices, uint64_t capacity, uint64_t* total) {
    if (!indptr || !total || !n_ecs || !n_loci || !n_haps || n_haps > 31) return fail(nullptr, ECB_ERR_ARG, "bad argument");
    if (device < 0 || device >= CV_MAX_DEV || hipSetDevice(device) != hipSuccess) return fail(nullptr, ECB_ERR_NO_DEVICE, "no such device");
    if (indptr[n_ecs] < 0) return fail(nullptr, ECB_ERR_CONTRACT, "malformed CSR: negative row pointer");
    const u64 nnz = (u64)indptr[n_ecs];
    if (nnz && (!indices || !data)) return fail(nullptr, ECB_ERR_ARG, "bad argument");
    if (!csc_indices || !csc_indptr) {                 // the count alone: the set bits of the masks, on the host (one pass over nnz words)
        u64 tot = 0;
        for (u64 i = 0; i < nnz; ++i) tot += (u64)__builtin_popcount((unsigned)data[i]);
        if (tot >= (1ull << 32)) return fail(nullptr, ECB_ERR_LIMIT, "more than 2^32-1 set haplotype bits");
        *total = tot;
        return ECB_OK;
    }
    DevBuf ip, ix, da, cp, ci;
    const u64 nc = (u64)n_haps * (n_loci + 1);
    if (ip.take(((u64)n_ecs + 1) * 4) || ix.take(nnz * 4) || da.take(nnz * 4) || cp.take(nc * 4) || ci.take(capacity * 4))
        return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    if (hipMemcpy(ip.p, indptr, ((u64)n_ecs + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (nnz && (hipMemcpy(ix.p, indices, nnz * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(da.p, data, nnz * 4, hipMemcpyHostToDevice) != hipSuccess)))
        return fail(nullptr, ECB_ERR_HIP, "copy to the device");
    // (the device entry point writes at most one row index per set bit: ask it for the count first when the caller's buffer might be short)
    uint64_t need = 0;
    int rc = ecb_csr_to_hapcsc_device(device, n_ecs, n_loci, n_haps, ip.p, ix.p, da.p, nullptr, nullptr, &need);
    if (rc != ECB_OK) return rc;
    *total = need;
    if (need > capacity) return fail(nullptr, ECB_ERR_ARG, "csc_indices holds %llu entries, the matrix has %llu set bits", (unsigned long long)capacity, (unsigned long long)need);
    rc = ecb_csr_to_hapcsc_device(device, n_ecs, n_loci, n_haps, ip.p, ix.p, da.p, cp.p, ci.p, total);
    if (rc != ECB_OK) return rc;
    if (hipMemcpy(csc_indptr, cp.p, nc * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (*total && hipMemcpy(csc_indices, ci.p, *total * 4, hipMemcpyDeviceToHost) != hipSuccess))
        return fail(nullptr, ECB_ERR_HIP, "copy from the device");
    return ECB_OK;
}

extern "C" int ecb_hapcsc_to_csr(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const int32_t* csc_indptr, const int32_t* csc_indices,
                                 uint64_t total, int32_t* indptr, int32_t* indices, int32_t* data, uint64_t* nnz) {
    if (!csc_indptr || !csc_indices || !indptr || !indices || !data || !nnz || !n_ecs || !n_loci || !n_haps || n_haps > 31 || !total)
        return fail(nullptr, ECB_ERR_ARG, "bad argument");
    if (total >= (1ull << 32)) return fail(nullptr, ECB_ERR_LIMIT, "more than 2^32-1 row indices");
    if (device < 0 || device >= CV_MAX_DEV || hipSetDevice(device) != hipSuccess) return fail(nullptr, ECB_ERR_NO_DEVICE, "no such device");
    DevBuf cp, ci, ip, ix, da;
    const u64 nc = (u64)n_haps * (n_loci + 1);
    if (cp.take(nc * 4) || ci.take(total * 4) || ip.take(((u64)n_ecs + 1) * 4) || ix.take(total * 4) || da.take(total * 4))
        return fail(nullptr, ECB_ERR_HIP, "out of device memory");
    if (hipMemcpy(cp.p, csc_indptr, nc * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(ci.p, csc_indices, total * 4, hipMemcpyHostToDevice) != hipSuccess)
        return fail(nullptr, ECB_ERR_HIP, "copy to the device");
    const int rc = ecb_hapcsc_to_csr_device(device, n_ecs, n_loci, n_haps, cp.p, ci.p, total, ip.p, ix.p, da.p, nnz);
    if (rc != ECB_OK) return rc;
    if (hipMemcpy(indptr, ip.p, ((u64)n_ecs + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (*nnz && (hipMemcpy(indices, ix.p, *nnz * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(data, da.p, *nnz * 4, hipMemcpyDeviceToHost) != hipSuccess)))
        return fail(nullptr, ECB_ERR_HIP, "copy from the device");
    return ECB_OK;
}

// ---- ecb_merge: the multi-GPU merge for ONE process that drives several GPUs ----------------------------------------------
// shards[r] holds the reads of contiguous read shard r on its own device; `root` is an empty handle (any device).  The same protocol as
// alntools_amd/dist.py runs over RCCL with one process per GPU, here with peer copies (hipMemcpyPeerAsync: xGMI between the GPUs of a
// node) -- cut every shard's table into n key ranges, range q of every shard to device q, merged there in shard order on a handle of its own,
// finalized there (ranked against the whole run's read numbering, rows emitted), the finished pieces placed by first read on the root.
// (bam_utils.py:646-724: contiguous chunks per worker, the workers' dicts merged in order.)  Everything is built from the entry points above;
// the devices work one after the other here -- a host that wants them side by side runs one thread or one process per GPU over the same calls.
namespace {
struct PeerBuf {                       // device memory on a given device, freed with its scope
    void* p = nullptr; int dev = 0;
    PeerBuf() {}
    PeerBuf(const PeerBuf&) = delete; PeerBuf& operator=(const PeerBuf&) = delete;
    PeerBuf(PeerBuf&& o) noexcept : p(o.p), dev(o.dev) { o.p = nullptr; }
    ~PeerBuf() { if (p) { hipSetDevice(dev); hipFree(p); } }
    int take(int device, u64 bytes) {
        dev = device;
        if (hipSetDevice(device) != hipSuccess) return ECB_ERR_NO_DEVICE;
        return hipMalloc(&p, std::max<u64>(bytes, 16)) == hipSuccess ? ECB_OK : ECB_ERR_HIP;
    }
};
struct HandleGuard { std::vector<ecb_handle*> h; ~HandleGuard() { for (ecb_handle* x : h) ecb_destroy(x); } };
}  // namespace

extern "C" int ecb_merge(ecb_handle* const* shards, uint32_t n, ecb_handle* root, ecb_sizes* out) {
    if (!shards || !n || !root || !out) return fail(root, ECB_ERR_ARG, "ecb_merge: null argument");
    if (n > MAX_PARTS) return fail(root, ECB_ERR_LIMIT, "ecb_merge: at most %u shards", MAX_PARTS);
    for (u32 r = 0; r < n; ++r) {
        if (!shards[r] || shards[r] == root) return fail(root, ECB_ERR_ARG, "ecb_merge: shard %u is null or the root itself", r);
        if (shards[r]->cfg.n_loci != root->cfg.n_loci || shards[r]->cfg.n_haplotypes != root->cfg.n_haplotypes)
            return fail(root, ECB_ERR_ARG, "ecb_merge: shard %u was created for other targets than the root", r);
        if ((shards[r]->cfg.flags | root->cfg.flags) & ECB_F_MULTISAMPLE) return fail(root, ECB_ERR_STATE, "ecb_merge: single-sample handles (the multisample merge has a second exchange: alntools_amd/dist.py)");
    }
    // 1. every shard: sizes, counters, its table cut into n key ranges (first reads counted from the shard's own 0)
    std::vector<u64> ne(n), np(n), nr(n), all(n), valid(n), base(n + 1, 0);
    std::vector<PeerBuf> ent(n), prs(n);
    std::vector<std::vector<uint64_t>> eoff(n, std::vector<uint64_t>(n + 1, 0)), poff(n, std::vector<uint64_t>(n + 1, 0));
    for (u32 r = 0; r < n; ++r) {
        uint64_t a = 0, b = 0, c = 0;
        int rc = ecb_table_sizes(shards[r], &a, &b, &c);
        if (rc != ECB_OK) return fail(root, rc, "ecb_merge: shard %u: %s", r, ecb_last_error(shards[r]));
        ne[r] = a; np[r] = b; nr[r] = c;
        uint64_t ca = 0, cv = 0, cr = 0;
        rc = ecb_counters(shards[r], &ca, &cv, &cr);
        if (rc != ECB_OK) return fail(root, rc, "ecb_merge: shard %u: %s", r, ecb_last_error(shards[r]));
        all[r] = ca; valid[r] = cv;
        base[r + 1] = base[r] + nr[r];
        if (ent[r].take(shards[r]->device, std::max<u64>(ne[r], 1) * sizeof(Entry)) || prs[r].take(shards[r]->device, std::max<u64>(np[r], 1) * sizeof(uint2)))
            return fail(root, ECB_ERR_HIP, "ecb_merge: out of device memory on device %d", shards[r]->device);
        if (ne[r]) {
            rc = ecb_table_export_parts_device(shards[r], ent[r].p, prs[r].p, 0, n, eoff[r].data(), poff[r].data());
            if (rc != ECB_OK) return fail(root, rc, "ecb_merge: shard %u: %s", r, ecb_last_error(shards[r]));
        }
    }
    u64 t_all = 0, t_valid = 0;
    for (u32 r = 0; r < n; ++r) { t_all += all[r]; t_valid += valid[r]; }
    const u64 t_reads = base[n];
    // 2. range q: its pieces to device q (the device of shard q), merged in shard order on a handle of its own, finalized there
    HandleGuard parts;
    struct Piece { PeerBuf ip, ix, da, cn, fi; u64 n_ecs = 0, nnz = 0; };
    std::vector<Piece> piece(n);
    for (u32 q = 0; q < n; ++q) {
        const int dq = shards[q]->device;
        ecb_config cfg = shards[q]->cfg;
        cfg.flags = 0; cfg.device = dq;
        u64 arriving = 0;
        for (u32 r = 0; r < n; ++r) arriving += eoff[r][q + 1] - eoff[r][q];
        if (!arriving) continue;
        cfg.ec_capacity = std::max<u64>(next_pow2(arriving * 2 + 1024), 1 << 12);
        ecb_handle* part = nullptr;
        int rc = ecb_create(&cfg, &part);
        if (rc != ECB_OK) return fail(root, rc, "ecb_merge: range %u: %s", q, ecb_last_error(nullptr));
        parts.h.push_back(part);
        std::vector<PeerBuf> pe(n), pp(n);
        std::vector<const void*> le, lp; std::vector<uint64_t> lne, lnp;
        for (u32 r = 0; r < n; ++r) {
            const u64 e_n = eoff[r][q + 1] - eoff[r][q], p_n = poff[r][q + 1] - poff[r][q];
            if (!e_n) continue;
            if (pe[r].take(dq, e_n * sizeof(Entry)) || pp[r].take(dq, std::max<u64>(p_n, 1) * sizeof(uint2))) return fail(root, ECB_ERR_HIP, "ecb_merge: out of device memory on device %d", dq);
            if (hipMemcpyPeer(pe[r].p, dq, (const char*)ent[r].p + eoff[r][q] * sizeof(Entry), shards[r]->device, e_n * sizeof(Entry)) != hipSuccess ||
                (p_n && hipMemcpyPeer(pp[r].p, dq, (const char*)prs[r].p + poff[r][q] * sizeof(uint2), shards[r]->device, p_n * sizeof(uint2)) != hipSuccess))
                return fail(root, ECB_ERR_HIP, "ecb_merge: peer copy from device %d to device %d failed", shards[r]->device, dq);
            if (base[r]) { rc = ecb_table_rebase_device(part, pe[r].p, e_n, base[r]); if (rc != ECB_OK) return fail(root, rc, "ecb_merge: %s", ecb_last_error(part)); }
            le.push_back(pe[r].p); lp.push_back(pp[r].p); lne.push_back(e_n); lnp.push_back(p_n);
        }
        rc = ecb_table_merge_batch_device(part, (uint32_t)le.size(), le.data(), lne.data(), lp.data(), lnp.data());
        if (rc == ECB_OK) rc = ecb_add_counters(part, t_all, t_valid, t_reads);
        ecb_sizes s{};
        if (rc == ECB_OK) rc = ecb_finalize(part, &s);
        if (rc != ECB_OK) return fail(root, rc, "ecb_merge: range %u: %s", q, ecb_last_error(part));
        Piece& P = piece[q];
        P.n_ecs = s.n_ecs; P.nnz = s.nnz_a;
        if (P.ip.take(dq, (s.n_ecs + 1) * 4) || P.ix.take(dq, s.nnz_a * 4) || P.da.take(dq, s.nnz_a * 4) || P.cn.take(dq, s.n_ecs * 4) || P.fi.take(dq, s.n_ecs * 4))
            return fail(root, ECB_ERR_HIP, "ecb_merge: out of device memory on device %d", dq);
        rc = ecb_export_device(part, P.ip.p, P.ix.p, P.da.p, nullptr, nullptr, P.cn.p);
        if (rc == ECB_OK) rc = ecb_export_firsts_device(part, P.fi.p);
        if (rc != ECB_OK) return fail(root, rc, "ecb_merge: range %u: %s", q, ecb_last_error(part));
    }
    // 3. the finished pieces to the root's device, placed by first read
    const int d0 = root->device;
    std::vector<Piece> at_root(n);
    std::vector<const void*> lip, lix, lda, lcn, lfi; std::vector<uint64_t> lne2, lnz;
    for (u32 q = 0; q < n; ++q) {
        Piece& S = piece[q];
        if (!S.n_ecs) continue;
        Piece* use = &S;
        if (S.ip.dev != d0) {
            Piece& D = at_root[q];
            if (D.ip.take(d0, (S.n_ecs + 1) * 4) || D.ix.take(d0, S.nnz * 4) || D.da.take(d0, S.nnz * 4) || D.cn.take(d0, S.n_ecs * 4) || D.fi.take(d0, S.n_ecs * 4))
                return fail(root, ECB_ERR_HIP, "ecb_merge: out of device memory on device %d", d0);
            if (hipMemcpyPeer(D.ip.p, d0, S.ip.p, S.ip.dev, (S.n_ecs + 1) * 4) != hipSuccess || hipMemcpyPeer(D.ix.p, d0, S.ix.p, S.ip.dev, S.nnz * 4) != hipSuccess ||
                hipMemcpyPeer(D.da.p, d0, S.da.p, S.ip.dev, S.nnz * 4) != hipSuccess || hipMemcpyPeer(D.cn.p, d0, S.cn.p, S.ip.dev, S.n_ecs * 4) != hipSuccess ||
                hipMemcpyPeer(D.fi.p, d0, S.fi.p, S.ip.dev, S.n_ecs * 4) != hipSuccess)
                return fail(root, ECB_ERR_HIP, "ecb_merge: peer copy from device %d to device %d failed", S.ip.dev, d0);
            D.n_ecs = S.n_ecs; D.nnz = S.nnz;
            use = &D;
        }
        lip.push_back(use->ip.p); lix.push_back(use->ix.p); lda.push_back(use->da.p); lcn.push_back(use->cn.p); lfi.push_back(use->fi.p);
        lne2.push_back(use->n_ecs); lnz.push_back(use->nnz);
    }
    return ecb_assemble_ranges_device(root, (uint32_t)lip.size(), lip.data(), lix.data(), lda.data(), lcn.data(), lfi.data(), lne2.data(), lnz.data(),
                                      t_reads, t_all, t_valid, out);
}

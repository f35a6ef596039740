// libecb -- equivalence-class builder for alntools' bam2ec / bam2emase hot path on MI355X (gfx950).
//
// What the reference does per alignment in Python (alntools/bam_utils.py:258-344), per merge
// (:680-724) and per EC (:788-847) is done here by three groups of kernels:
//
//   k_stream   one pass over the record tuples (12 B/record, coalesced 16-B loads).  A workgroup
//              walks a contiguous slice of the stream tile by tile (2048 records).  Per tile it
//              (a) applies the record filter (bam_utils.py:264-270) and finds read heads from the
//                  host's run counter (bam_utils.py:289-320);
//              (b) inserts every valid record into an LDS open-addressing table that is cut into one
//                  private range per read (2 slots per record of the read): key = locus, value = OR of
//                  haplotype bits -- this is the duplicate collapse of bam_utils.py:322-325 and the
//                  per-(EC,target,haplotype) bit test of bam_utils.py:800-819 in one LDS atomic;
//              (c) one lane per read walks its range, sums a 2x64-bit mix over the distinct
//                  (locus, mask) pairs -- an order-independent 126-bit set hash, the stand-in for the
//                  sorted string key of bam_utils.py:307 -- and upserts the global EC table
//                  (count += 1, first = min(read index): bam_utils.py:309-312, 688-698).  The lane that
//                  creates an EC copies its pairs into the key arena.
//   k_slow     the same for single reads that do not fit a tile or hit a full table (one workgroup per
//              read, global scratch table).
//   finalize   rank ECs by first appearance (bitmap + scan: bam_utils.py:682-698), exclusive scan of
//              row lengths, sort each row by locus and emit CSR A / N (bam_utils.py:835-847,
//              bin_utils.py:208-211).
//
// Integer / indexing work only: no MFMA.  The bound is HBM bandwidth.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/ecb.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int TPB = 256;             // threads per workgroup (4 waves of 64)
constexpr int RPL = 8;               // records per lane per tile (2 x 16-byte loads per stream)
constexpr int TILE = TPB * RPL;      // records per tile
constexpr int MAXR = 512;            // reads finished per tile (more heads than this: the next tile starts there)
constexpr u32 MAX_PROBE = 256;       // EC-table probes before a read is deferred to k_slow
constexpr u32 PENDING = 0xFFFFFFFFu;
constexpr u32 KEY_PENDING = 0x80000000u;   // Slot::n bit: key not extracted yet, (off, n) hold the head record index
constexpr u32 MAX_LOCI = 1u << 27;   // (locus << 5 | hap) + 1 must fit 32 bits

constexpr u32 ERR_CONTRACT = 1u;     // device error bits (Counters::err)
constexpr u32 ERR_RANGE = 2u;
constexpr u32 ERR_ARENA = 4u;
constexpr u32 ERR_QUEUE = 8u;

struct Slot {                        // 32 bytes, one EC
    u64 lo, hi;                      // 126-bit set hash, both non-zero once claimed
    u32 count;                       // reads in this EC
    u32 first_inv;                   // ~(smallest read index)  (atomicMax on zero-initialised memory)
    u32 off, n;                      // key = arena[off .. off+n); or, while n & KEY_PENDING, the creating read's head index
};

struct Counters {
    u64 all, valid;                  // records offered / passing the filter
    u64 arena_top;                   // pairs used in the key arena
    u64 n_queue;                     // reads deferred to k_slow
    u64 n_ecs;                       // ECs created by k_slow / k_merge (k_stream's are counted by k_collect_new)
    u32 err;
    u32 full;                        // set when a read found no EC-table slot: workgroups park, host grows the table
};

// ---------------------------------------------------------------------------------------------
// hashing: EC identity = the SET of (locus, haplotype) targets of a read (bam_utils.py:307 builds a
// sorted string for the same purpose).  Set hash = sum over distinct targets of a 2 x 64-bit mix,
// finalised; commutative, so records need no sorting and duplicates are dropped before summing.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ u32 target_key(u32 locus, u32 hap) { return (locus << 5) | hap; }
__device__ __forceinline__ void target_hash(u32 tkey, u64& a, u64& b) {
    const u64 x = tkey;
    a = mix64(x + 0x9E3779B97F4A7C15ull);
    b = mix64((x ^ 0xD6E8FEB86659FD93ull) * 0xFF51AFD7ED558CCDull + 0xC4CEB9FE1A85EC53ull);
}
__device__ __forceinline__ void finish_hash(u64 s1, u64 s2, u64& lo, u64& hi) {
    lo = mix64(s1) | 1ull;
    hi = mix64(s2 ^ 0xA0761D6478BD642Full) | 1ull;
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
// tiles of WT records with wave-private LDS and no workgroup barriers, so the 16-20 waves of a CU
// overlap each other's HBM and EC-table latency.  The workgroup only shares the hot-EC cache.
// ---------------------------------------------------------------------------------------------
constexpr int WT = 512;              // records per wave tile (8 per lane: 2 x 16-byte loads per stream)
constexpr int WMAXR = 64;            // reads finished per wave tile (one lane each in phase (c))
constexpr int NWAVE = TPB / 64;

struct WaveLds {
    unsigned short seg[WT + 4];      // tile-relative start of every read in the tile (+ end sentinel)
    u32 tkey[2 * WT];                // per-read sets of targets: 2 slots per record of the read, 0 = empty
    u64 acc_a[WMAXR], acc_b[WMAXR];  // per-read set-hash accumulators
};

struct StreamArgs {
    const u32* rid; const u32* loc; const u32* hf; const int* pos;
    u64 n, chunk;
    u32 prev_rid;                    // read_id of the record before this batch (0xFFFFFFFF at stream start)
    u32 n_loci, n_haps;
    Slot* table; u64 cap_mask;
    Counters* ctr;
    u32* read_slot;                  // slot of every read (indexed by read_id)
    int* rng_min; int* rng_max;      // per (locus*H + hap), or null
    u64* queue; u64 queue_cap;       // head record index of deferred reads
    u64* resume;                     // per wave {next record to process, records counted up to}
    u32 ablate;                      // profiling only (env ECB_ABLATE): 1 = stop after (a), 2 = after (b), 4 = no EC table
};

__device__ __forceinline__ void wave_sync() {   // orders this wave's LDS traffic (lanes run in lockstep)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

struct TileRegs { u32 rr[8], ll[8], hh[8]; };

__device__ __forceinline__ void load_tile(const StreamArgs& A, u64 tb, u64 te, u32 lane, TileRegs& R) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const u64 i0 = tb + (u64)g * (WT / 2) + 4u * lane;
        if (i0 + 4 <= te) {
            uint4 v = *reinterpret_cast<const uint4*>(A.rid + i0);
            R.rr[4 * g] = v.x; R.rr[4 * g + 1] = v.y; R.rr[4 * g + 2] = v.z; R.rr[4 * g + 3] = v.w;
            v = *reinterpret_cast<const uint4*>(A.loc + i0);
            R.ll[4 * g] = v.x; R.ll[4 * g + 1] = v.y; R.ll[4 * g + 2] = v.z; R.ll[4 * g + 3] = v.w;
            v = *reinterpret_cast<const uint4*>(A.hf + i0);
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

__global__ __launch_bounds__(TPB) void k_stream(StreamArgs A) {
    __shared__ WaveLds wl[NWAVE];

    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    WaveLds& L = wl[w];
    const u64 wid = (u64)blockIdx.x * NWAVE + w;
    const u64 c0 = wid * A.chunk;
    const u64 c1 = min(c0 + A.chunk, A.n);
    const bool have = c0 < A.n;


    u64 my_all = 0, my_valid = 0;
    u64 p = 0, counted = 0;
    if (have) {
        p = A.resume[2 * wid]; counted = A.resume[2 * wid + 1];
    }
    if (have && p < c1) {
        u32 base = (p == 0 ? A.prev_rid : A.rid[p - 1]) + 1u;     // read index of the first head >= p
        TileRegs R;
        load_tile(A, p & ~(u64)3, min((p & ~(u64)3) + (u64)WT, A.n), lane, R);
        u32 parked = __hip_atomic_load(&A.ctr->full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        while (p < c1) {
            if (parked) break;            // the EC table filled up somewhere: the host grows it and relaunches
            const u64 tb = p & ~(u64)3;
            const u64 te = min(tb + (u64)WT, A.n);
            const u64 cnt_hi = min(te, c1);
            // clear this wave's per-tile LDS state
            {
                uint4* z = reinterpret_cast<uint4*>(L.tkey);
#pragma unroll
                for (int t = 0; t < (2 * WT) / (4 * 64); ++t) z[t * 64 + lane] = make_uint4(0, 0, 0, 0);
                L.acc_a[lane] = 0; L.acc_b[lane] = 0;
            }
            // ---- (a) filter, heads ---------------------------------------------------------------
            u32 r_key[8], r_rl[8];
            u32 m_ok = 0, m_head = 0, m_own = 0;   // bit k: valid & in range / head / head of a read we own
            u32 bad = 0;
            {
                const u32 up0 = __shfl_up(R.rr[3], 1), up1 = __shfl_up(R.rr[7], 1), last0 = __shfl(R.rr[3], 63);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const u64 i0 = tb + (u64)g * (WT / 2) + 4u * lane;
                    u32 prev = g == 0 ? (lane == 0 ? base - 1u : up0) : (lane == 0 ? last0 : up1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * g + j;
                        const u64 i = i0 + j;
                        const bool in = (i >= p) && (i < te);
                        const u32 f = R.hh[k];
                        const bool ok = rec_valid(f);
                        const u32 hap = (f >> ECB_HAP_SHIFT) & 0xFFu;
                        const u32 step = R.rr[k] - prev;
                        r_rl[k] = R.rr[k] - base;
                        r_key[k] = target_key(R.ll[k], hap);
                        if (in) {
                            if (step > 1u || (step == 1u && !ok)) bad |= ERR_CONTRACT;
                            if (step == 1u) { m_head |= 1u << k; if (i < c1) m_own |= 1u << k; }
                            if (ok) {
                                if (R.ll[k] >= A.n_loci || hap >= A.n_haps) bad |= ERR_RANGE;
                                else m_ok |= 1u << k;
                            }
                            if (i >= counted && i < cnt_hi) {
                                my_all += 1;
                                if (ok) {
                                    my_valid += 1;
                                    if (A.rng_min && R.ll[k] < A.n_loci && hap < A.n_haps) {
                                        const u64 sl = (u64)R.ll[k] * A.n_haps + hap;
                                        const int ps = A.pos[i];
                                        atomicMin(A.rng_min + sl, ps);
                                        atomicMax(A.rng_max + sl, ps);
                                    }
                                }
                            }
                        }
                        prev = R.rr[k];
                    }
                }
            }
            if (__ballot(bad != 0u)) {               // never index LDS with a broken run counter
                if (bad) atomicOr(&A.ctr->err, bad);
                p = c1;
                break;
            }
            const u32 nr = wave_sum(__popc(m_head));                 // heads in [p, te)
            const u32 nown = wave_sum(__popc(m_own));                // heads in [p, c1): ours
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (m_head >> k & 1u) L.seg[r_rl[k]] = (unsigned short)(4u * lane + (k & 3) + (k >> 2) * (WT / 2));
            if (lane == 0) L.seg[nr] = (unsigned short)(te - tb);    // end sentinel
            wave_sync();
            const bool last_complete = (te == A.n);                  // batches end on a read boundary
            const u32 nrc = last_complete ? nr : (nr ? nr - 1u : 0u);
            const u32 nproc = min(min(nrc, nown), (u32)WMAXR);
            const bool done = (te >= c1 && nown <= nproc);           // every read that starts in our slice
            u64 p_next = te;
            u32 base_next = base + nr;
            bool giant = false;
            if (!done && nproc < nr) {
                const u64 h = tb + L.seg[nproc];                     // first read not finished here
                if (h == p) giant = true;                            // one read fills the whole tile: k_slow
                else { p_next = h; base_next = base + nproc; }
            }
            if (done) p_next = c1;
            // ---- prefetch the next tile while this one is hashed and looked up ----------------------
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

            // ---- (b) per-read target sets in LDS; first occurrences feed the read's set hash ------
            if (!(A.ablate & 1u)) {
                u64 ca = 0, cb = 0;
                u32 crl = PENDING;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((m_ok >> k & 1u) && r_rl[k] < nproc) {
                        const u32 rl = r_rl[k];
                        const u32 s2 = 2u * L.seg[rl], len2 = 2u * L.seg[rl + 1] - s2;
                        const u32 key = r_key[k] + 1u;
                        u32 q = s2 + __umulhi(r_key[k] * 0x9E3779B1u, len2);
                        bool fresh;
                        for (;;) {
                            const u32 old = atomicCAS(&L.tkey[q], 0u, key);
                            if (old == 0u) { fresh = true; break; }
                            if (old == key) { fresh = false; break; }   // duplicate (read, target): bam_utils.py:322-325
                            if (++q == s2 + len2) q = s2;
                        }
                        if (fresh) {
                            if (rl != crl) {
                                if (crl != PENDING) { atomicAdd(&L.acc_a[crl], ca); atomicAdd(&L.acc_b[crl], cb); }
                                crl = rl; ca = 0; cb = 0;
                            }
                            u64 a, b; target_hash(r_key[k], a, b); ca += a; cb += b;
                        }
                    }
                }
                if (crl != PENDING) { atomicAdd(&L.acc_a[crl], ca); atomicAdd(&L.acc_b[crl], cb); }
            }
            wave_sync();

            // ---- (c) one lane per read: EC lookup; the read's slot is all that is recorded ----------
            if (lane < nproc && !(A.ablate & 3u)) {
                u64 lo, hi;
                finish_hash(L.acc_a[lane], L.acc_b[lane], lo, hi);
                const u32 rd = base + lane;
                if (A.ablate & 4u) {
                    A.read_slot[rd] = (u32)(lo & A.cap_mask);
                } else {
                    bool created = false;
                    const u64 j = table_find_or_insert(A.table, A.cap_mask, lo, hi, &created);
                    const u64 h = tb + L.seg[lane];
                    if (j == ~0ull) {                               // table too full here: defer the read, park
                        atomicExch(&A.ctr->full, 1u);
                        const u64 qi = atomicAdd(&A.ctr->n_queue, 1ull);
                        if (qi < A.queue_cap) A.queue[qi] = h; else atomicOr(&A.ctr->err, ERR_QUEUE);
                    } else {
                        if (created) {                              // key extraction is deferred to k_keys
                            A.table[j].off = (u32)h;
                            A.table[j].n = KEY_PENDING | (u32)(h >> 32);
                        }
                        A.read_slot[rd] = (u32)j;
                    }
                }
            }
            wave_sync();

            counted = max(counted, cnt_hi);
            p = p_next; base = base_next; R = N; parked = parked_next;
        }
        if (lane == 0) { A.resume[2 * wid] = p; A.resume[2 * wid + 1] = counted; }
        // records offered / valid: one atomic pair per wave
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { my_all += __shfl_xor(my_all, d); my_valid += __shfl_xor(my_valid, d); }
        if (lane == 0) { atomicAdd(&A.ctr->all, my_all); atomicAdd(&A.ctr->valid, my_valid); }
    }

}

// ---------------------------------------------------------------------------------------------
// k_keys: extract the (locus, mask) key of every EC that k_stream created, from the records of the
// read that created it.  Pass 1 counts pairs per EC, an exclusive scan places them, pass 2 writes.
// One wave per EC; reads longer than KEYS_SMALL records go to the workgroup-wide variant.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_collect_new(const Slot* table, u64 cap, u32* list, u64 max_list, u64* n_list) {
    __shared__ u32 s_cnt;
    __shared__ u64 s_base;
    for (u64 b = (u64)blockIdx.x * TPB; b < cap; b += (u64)gridDim.x * TPB) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const u64 i = b + threadIdx.x;
        const bool pend = i < cap && table[i].hi != 0ull && (table[i].n & KEY_PENDING);
        u32 my = 0;
        if (pend) my = atomicAdd(&s_cnt, 1u);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(n_list, (u64)s_cnt);
        __syncthreads();
        if (pend && s_base + my < max_list) list[s_base + my] = (u32)i;
        __syncthreads();
    }
}

constexpr int KEYS_WT = 1024;        // entries of one wave's LDS table in k_keys
constexpr u32 KEYS_MAXL = KEYS_WT / 2;   // longer reads take the k_slow route (key-only mode)

template <bool WRITE>
__global__ __launch_bounds__(TPB) void k_keys(const u32* rid, const u32* loc, const u32* hf, u64 n,
                                              Slot* table, const u32* list, u64 n_list,
                                              u32* nlen, const u32* noff, u64 arena_base, uint2* arena,
                                              u64* slowq, u32* slowslot, u64* n_slow) {
    __shared__ u32 wk[TPB / 64][KEYS_WT], wm[TPB / 64][KEYS_WT];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const u64 e = (u64)blockIdx.x * (TPB / 64) + w;
    const bool act = e < n_list;
    u32* K = wk[w];
    u32* M = wm[w];
    for (u32 q = lane; q < KEYS_WT; q += 64) { K[q] = 0; M[q] = 0; }
    Slot* s = act ? table + list[e] : table;
    u64 head = 0;
    u32 L = 0;
    if (act) {
        head = (u64)s->off | ((u64)(s->n & ~KEY_PENDING) << 32);
        const u32 r0 = rid[head];
        for (u64 i0 = head;; i0 += 64) {                 // length of the creating read, in records
            const u64 i = i0 + lane;
            const u64 m = __ballot(i < n && rid[i] == r0);
            L += __popcll(m);
            if (m != ~0ull) break;
        }
    }
    const bool small = act && L <= KEYS_MAXL;
    __syncthreads();
    if (small) {
        for (u32 t = lane; t < L; t += 64) {
            const u32 f = hf[head + t];
            if (!rec_valid(f)) continue;
            const u32 lc = loc[head + t], key = lc + 1u, bit = 1u << ((f >> ECB_HAP_SHIFT) & 0xFFu);
            u32 q = __umulhi(lc * 0x9E3779B1u, (u32)KEYS_WT);
            for (;;) {
                const u32 old = atomicCAS(&K[q], 0u, key);
                if (old == 0u || old == key) { atomicOr(&M[q], bit); break; }
                if (++q == KEYS_WT) q = 0;
            }
        }
    }
    __syncthreads();
    if (!act) return;
    if (!small) {
        if (!WRITE && lane == 0) {
            const u64 qi = atomicAdd(n_slow, 1ull);
            slowq[qi] = head; slowslot[qi] = list[e];
            nlen[e] = 0;
        }
        return;
    }
    u32 cnt = 0;
    const u64 dst = WRITE ? arena_base + noff[e] : 0;
    for (u32 q0 = 0; q0 < KEYS_WT; q0 += 64) {
        const u32 q = q0 + lane;
        const bool occ = K[q] != 0u;
        const u64 m = __ballot(occ);
        if (WRITE && occ) arena[dst + cnt + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(K[q] - 1u, M[q]);
        cnt += __popcll(m);
    }
    if (lane == 0) {
        if (WRITE) { s->off = (u32)dst; s->n = cnt; }
        else nlen[e] = cnt;
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

__global__ __launch_bounds__(TPB) void k_part_hist(const u32* read_slot, u64 n_reads, u32 n_buckets, u32* hist) {
    extern __shared__ u32 sh[];
    const u64 G = gridDim.x, g = blockIdx.x, per = (n_reads + G - 1) / G;
    const u64 r0 = g * per, r1 = min(r0 + per, n_reads);
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB) sh[b] = 0;
    __syncthreads();
    for (u64 r = r0 + threadIdx.x; r < r1; r += TPB) {
        const u32 s = read_slot[r];
        if (s != PENDING) atomicAdd(&sh[s >> BIN_BITS], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB) hist[(u64)b * G + g] = sh[b];
}

__global__ __launch_bounds__(TPB) void k_part_scatter(const u32* read_slot, u64 n_reads, u32 n_buckets, const u32* offs,
                                                      uint2* pairs) {
    extern __shared__ u32 sh[];
    const u64 G = gridDim.x, g = blockIdx.x, per = (n_reads + G - 1) / G;
    const u64 r0 = g * per, r1 = min(r0 + per, n_reads);
    for (u32 b = threadIdx.x; b < n_buckets; b += TPB) sh[b] = offs[(u64)b * G + g];
    __syncthreads();
    for (u64 r = r0 + threadIdx.x; r < r1; r += TPB) {
        const u32 s = read_slot[r];
        if (s != PENDING) pairs[atomicAdd(&sh[s >> BIN_BITS], 1u)] = make_uint2(s, (u32)r);
    }
}

__global__ __launch_bounds__(TPB) void k_count_bins(const uint2* pairs, const u32* offs, u32 G, u32 n_buckets, u32 total,
                                                    Slot* table) {
    __shared__ u32 cnt[N_BINS], fst[N_BINS];
    const u32 b = blockIdx.x, lane = threadIdx.x & 63u;
    const u32 start = offs[(u64)b * G], end = (b + 1 < n_buckets) ? offs[(u64)(b + 1) * G] : total;
    for (u32 q = threadIdx.x; q < N_BINS; q += TPB) { cnt[q] = 0; fst[q] = 0xFFFFFFFFu; }
    __syncthreads();
    for (u32 i0 = start; i0 < end; i0 += TPB) {
        const u32 i = i0 + threadIdx.x;
        const bool have = i < end;
        uint2 pr = have ? pairs[i] : make_uint2(0, 0xFFFFFFFFu);
        const u32 bin = pr.x & (N_BINS - 1);
        // a hot EC fills most lanes of a wave: add it once per wave, the rest go one by one
        const u32 v = __shfl(bin, __ffsll((long long)__ballot(have)) - 1);
        const bool same = have && bin == v;
        const u64 m = __ballot(same);
        if (__popcll(m) >= 8) {
            u32 mn = same ? pr.y : 0xFFFFFFFFu;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mn = min(mn, (u32)__shfl_xor(mn, d));
            if (lane == (u32)(__ffsll((long long)m) - 1)) { atomicAdd(&cnt[v], (u32)__popcll(m)); atomicMin(&fst[v], mn); }
            if (have && !same) { atomicAdd(&cnt[bin], 1u); atomicMin(&fst[bin], pr.y); }
        } else if (have) {
            atomicAdd(&cnt[bin], 1u); atomicMin(&fst[bin], pr.y);
        }
    }
    __syncthreads();
    for (u32 q = threadIdx.x; q < N_BINS; q += TPB) {
        const u32 c = cnt[q];
        if (c) {                                           // this workgroup is the only writer of its slots
            Slot* s = table + (((u64)b << BIN_BITS) | q);
            s->count += c;
            s->first_inv = max(s->first_inv, ~fst[q]);
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
    const u32* key_slot;                                    // non-null: the EC exists (slot given); only extract its key
};

__global__ __launch_bounds__(TPB) void k_slow(SlowArgs A) {
    const u64 q = blockIdx.x;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u64 h = A.queue[q], L = A.len[q];
    const u64 cap2 = 2 * L;
    u32* key = A.scr_key + A.scr_off[q];
    u32* msk = A.scr_mask + A.scr_off[q];
    __shared__ u64 s_a[TPB / 64], s_b[TPB / 64];
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
    u64 a = 0, b = 0; u32 np = 0;
    for (u64 p = tid; p < cap2; p += TPB) {
        const u32 k = __hip_atomic_load(&key[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k) {
            u32 m = __hip_atomic_load(&msk[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++np;
            while (m) {                                     // same set hash as k_stream: sum over distinct targets
                const u32 hp = __ffs(m) - 1; m &= m - 1;
                u64 x, y; target_hash(target_key(k - 1u, hp), x, y); a += x; b += y;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_down(a, d); b += __shfl_down(b, d); np += __shfl_down(np, d); }
    if (lane == 0) { s_a[tid >> 6] = a; s_b[tid >> 6] = b; s_n[tid >> 6] = np; }
    __syncthreads();
    if (tid == 0) {
        a = b = 0; np = 0;
        for (int w = 0; w < TPB / 64; ++w) { a += s_a[w]; b += s_b[w]; np += s_n[w]; }
        u64 lo, hi; finish_hash(a, b, lo, hi);
        bool created = false;
        const u32 r0 = A.rid[h];
        u64 slot;
        if (A.key_slot) { slot = A.key_slot[q]; created = true; }
        else slot = table_find_or_insert(A.table, A.cap_mask, lo, hi, &created);
        s_created = 0; s_cnt = 0; s_fits = 0; s_off = 0;
        if (slot == ~0ull) {
            A.requeue[atomicAdd(A.n_requeue, 1ull)] = h;
        } else {
            if (!A.key_slot) A.read_slot[r0] = (u32)slot;
            if (created) {
                const u64 off = atomicAdd(&A.ctr->arena_top, (u64)np);
                if (!A.key_slot) atomicAdd(&A.ctr->n_ecs, 1ull);
                s_created = 1; s_off = off; s_fits = (off + np <= A.arena_cap);
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

// occupied slots -> dense list (order irrelevant: ranks come from `first`)
__global__ __launch_bounds__(TPB) void k_compact(const Slot* table, u64 cap, u32* list, u64 max_list, u64* n_list) {
    __shared__ u32 s_cnt;
    __shared__ u64 s_base;
    for (u64 b = (u64)blockIdx.x * TPB; b < cap; b += (u64)gridDim.x * TPB) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const u64 i = b + threadIdx.x;
        const bool occ = i < cap && table[i].hi != 0ull;
        u32 my = 0;
        if (occ) my = atomicAdd(&s_cnt, 1u);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(n_list, (u64)s_cnt);
        __syncthreads();
        if (occ && s_base + my < max_list) list[s_base + my] = (u32)i;
        __syncthreads();
    }
}

// serialise: entry e = slot list[e] with first rebased to the global read numbering
__global__ void k_export_entries(const Slot* table, const u32* list, u64 n, Slot* out, u32 read_base) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    Slot s = table[list[e]];
    s.first_inv = ~(~s.first_inv + read_base);
    out[e] = s;
}

__global__ void k_merge(const Slot* ent, u64 n, const uint2* pairs, u64 n_pairs, Slot* table, u64 cap_mask,
                        uint2* arena, u64 arena_cap, Counters* ctr) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const Slot s = ent[e];
    if ((u64)s.off + s.n > n_pairs) { atomicOr(&ctr->err, ERR_CONTRACT); return; }
    bool created = false;
    const u64 j = table_find_or_insert(table, cap_mask, s.lo, s.hi, &created);
    if (j == ~0ull) { atomicAdd(&ctr->n_queue, 1ull); return; }   // host sizes the table so this cannot happen
    atomicAdd(&table[j].count, s.count);                           // one pair of atomics per merged EC, not per read
    atomicMax(&table[j].first_inv, s.first_inv);
    if (created) {
        const u64 off = atomicAdd(&ctr->arena_top, (u64)s.n);
        atomicAdd(&ctr->n_ecs, 1ull);
        if (off + s.n > arena_cap) { atomicOr(&ctr->err, ERR_ARENA); return; }
        for (u32 t = 0; t < s.n; ++t) arena[off + t] = pairs[s.off + t];
        table[j].off = (u32)off; table[j].n = s.n;
    }
}

// ---------------------------------------------------------------------------------------------
// finalize: rank by first appearance, CSR emit
// ---------------------------------------------------------------------------------------------
__global__ void k_mark_first(const Slot* table, const u32* list, u64 n, u32* bitmap) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u32 f = ~table[list[e]].first_inv;
    atomicOr(&bitmap[f >> 5], 1u << (f & 31u));
}
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

__global__ void k_rank(const Slot* table, const u32* list, u64 n, const u32* bitmap, const u32* wprefix,
                       u32* order, u32* rowlen, u32* rank_of_slot) {
    const u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u32 si = list[e];
    const u32 f = ~table[si].first_inv;
    const u32 r = wprefix[f >> 5] + __popc(bitmap[f >> 5] & ((1u << (f & 31u)) - 1u));
    order[r] = si;
    rowlen[r] = table[si].n;
    rank_of_slot[si] = r;
}

// one wave per EC row: rank every (locus, mask) pair by locus and write it in place
__global__ __launch_bounds__(TPB) void k_emit(const Slot* table, const u32* order, u64 n, const uint2* arena,
                                               const u32* indptr, int* indices, int* data, int* counts) {
    const u64 e = ((u64)blockIdx.x * TPB + threadIdx.x) >> 6;
    const u32 lane = threadIdx.x & 63u;
    if (e >= n) return;
    const Slot s = table[order[e]];
    const uint2* src = arena + s.off;
    const u32 dst = indptr[e];
    for (u32 i = lane; i < s.n; i += 64) {
        const uint2 pi = src[i];
        u32 r = 0;
        for (u32 j = 0; j < s.n; ++j) r += src[j].x < pi.x;   // loci within a key are distinct
        indices[dst + r] = (int)pi.x;
        data[dst + r] = (int)pi.y;
    }
    if (lane == 0) counts[e] = (int)s.count;
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

    Slot* table = nullptr; u64 cap = 0;
    uint2* arena = nullptr; u64 arena_cap = 0;
    Counters* ctr = nullptr;
    Counters hctr{};                  // last read-back
    u32* read_slot = nullptr; u64 read_slot_cap = 0;
    int *rng_min = nullptr, *rng_max = nullptr;
    u64* queue = nullptr; u64 queue_cap = 0;
    u32* newlist = nullptr; u64 newlist_cap = 0;   // slots whose key is pending (k_collect_new)
    u64 n_ecs_stream = 0;             // ECs created by k_stream launches (counted when their keys are extracted)
    u64 n_ecs() const { return hctr.n_ecs + n_ecs_stream; }
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
int run_slow(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, u64 n,
             u64* d_q, u64 nq, const u32* d_key_slot) {
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
                   h->table, h->cap - 1, h->arena, h->arena_cap, h->ctr, h->read_slot, nre_buf, d_nre, d_key_slot};
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

// keys of the ECs the last k_stream launch created: collect, count, scan, write (k_keys)
int extract_keys(ecb_handle* h, const u32* d_rid, const u32* d_loc, const u32* d_hf, u64 n) {
    u64* d_cnt = nullptr;                       // [0] = new ECs, [1] = ECs routed to k_slow
    HIPCHK(h, hipMalloc(&d_cnt, 2 * sizeof(u64)));
    HIPCHK(h, hipMemsetAsync(d_cnt, 0, 2 * sizeof(u64), h->stream));
    if (h->newlist_cap < h->cap) {
        if (h->newlist) hipFree(h->newlist);
        h->newlist_cap = h->cap;
        HIPCHK(h, hipMalloc(&h->newlist, h->newlist_cap * sizeof(u32)));
    }
    k_collect_new<<<(unsigned)std::min<u64>(4096, (h->cap + TPB - 1) / TPB), TPB, 0, h->stream>>>(
        h->table, h->cap, h->newlist, h->newlist_cap, d_cnt);
    u64 n_new = 0;
    HIPCHK(h, hipMemcpyAsync(&n_new, d_cnt, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int rc = ECB_OK;
    if (n_new) {
        u32 *nlen = nullptr, *noff = nullptr, *slowslot = nullptr;
        u64* slowq = nullptr;
        HIPCHK(h, hipMalloc(&nlen, n_new * 4)); HIPCHK(h, hipMalloc(&noff, n_new * 4));
        HIPCHK(h, hipMalloc(&slowslot, n_new * 4)); HIPCHK(h, hipMalloc(&slowq, n_new * 8));
        const unsigned gb = nblk(n_new, TPB / 64);
        k_keys<false><<<gb, TPB, 0, h->stream>>>(d_rid, d_loc, d_hf, n, h->table, h->newlist, n_new, nlen, noff, 0,
                                                 h->arena, slowq, slowslot, d_cnt + 1);
        u32 total = 0;
        rc = excl_scan(h, nlen, n_new, noff, &total);
        if (rc == ECB_OK) rc = sync_counters(h);
        if (rc == ECB_OK) {
            const u64 base = h->hctr.arena_top;
            if (base + total > h->arena_cap)
                rc = fail(h, ECB_ERR_TABLE_FULL, "EC key arena exhausted (%llu pairs): raise arena_capacity",
                          (unsigned long long)h->arena_cap);
            else {
                const u64 top = base + total;
                hipMemcpyAsync(&h->ctr->arena_top, &top, sizeof(u64), hipMemcpyHostToDevice, h->stream);
                k_keys<true><<<gb, TPB, 0, h->stream>>>(d_rid, d_loc, d_hf, n, h->table, h->newlist, n_new, nlen, noff,
                                                        base, h->arena, slowq, slowslot, d_cnt + 1);
                u64 n_slowk = 0;
                hipMemcpyAsync(&n_slowk, d_cnt + 1, sizeof(u64), hipMemcpyDeviceToHost, h->stream);
                rc = sync_counters(h);
                if (rc == ECB_OK && n_slowk) rc = run_slow(h, d_rid, d_loc, d_hf, n, slowq, n_slowk, slowslot);
                if (rc == ECB_OK) h->n_ecs_stream += n_new;
            }
        }
        hipFree(nlen); hipFree(noff); hipFree(slowslot); hipFree(slowq);
    }
    hipFree(d_cnt);
    return rc;
}

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
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    // one contiguous slice per wave; enough waves to fill every CU a few times over
    u64 waves = std::min<u64>((u64)cus * 5 * NWAVE, (n + 2 * WT - 1) / (2 * WT));
    waves = std::max<u64>(waves, 1);
    u64 chunk = (n + waves - 1) / waves;
    chunk = (chunk + 3) & ~(u64)3;
    waves = (n + chunk - 1) / chunk;
    const u64 blocks = (waves + NWAVE - 1) / NWAVE;
    // a parked launch defers at most the reads of the tiles in flight
    const u64 need_q = waves * (u64)(WMAXR + 1) + 16;
    if (h->queue_cap < need_q) {
        if (h->queue) hipFree(h->queue);
        h->queue_cap = need_q;
        HIPCHK(h, hipMalloc(&h->queue, h->queue_cap * sizeof(u64)));
    }
    u64* d_resume = nullptr;
    HIPCHK(h, hipMalloc(&d_resume, 2 * waves * sizeof(u64)));
    {
        std::vector<u64> r0(2 * waves);
        for (u64 b = 0; b < waves; ++b) r0[2 * b] = r0[2 * b + 1] = b * chunk;
        HIPCHK(h, hipMemcpyAsync(d_resume, r0.data(), 2 * waves * sizeof(u64), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    StreamArgs a{d_rid, d_loc, d_hf, d_pos, n, chunk, h->prev_rid, h->cfg.n_loci, h->cfg.n_haplotypes,
                 h->table, h->cap - 1, h->ctr, h->read_slot,
                 h->rng_min, h->rng_max, h->queue, h->queue_cap, d_resume,
                 getenv("ECB_ABLATE") ? (u32)atoi(getenv("ECB_ABLATE")) : 0u};
    for (;;) {
        HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, sizeof(u64), h->stream));   // per launch
        HIPCHK(h, hipMemsetAsync(&h->ctr->full, 0, sizeof(u32), h->stream));
        a.table = h->table; a.cap_mask = h->cap - 1;
        if (h->prof) hipEventRecord(h->ev0, h->stream);
        k_stream<<<(unsigned)blocks, TPB, 0, h->stream>>>(a);
        if (h->prof) hipEventRecord(h->ev1, h->stream);
        HIPCHK(h, hipGetLastError());
        rc = sync_counters(h);
        if (h->prof) {
            float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1);
            h->prof_ms += ms; h->prof_launches += 1;
        }
        if (rc != ECB_OK) break;
        const bool parked = h->hctr.full != 0;
        if (h->hctr.n_queue) {
            rc = run_slow(h, d_rid, d_loc, d_hf, n, h->queue, std::min<u64>(h->hctr.n_queue, h->queue_cap), nullptr);
            if (rc != ECB_OK) break;
        }
        rc = extract_keys(h, d_rid, d_loc, d_hf, n);
        if (rc != ECB_OK) break;
        if (!parked) break;
        rc = grow_table(h, h->cap * 4);                 // some workgroups stopped early: more room, then resume
        if (rc != ECB_OK) break;
    }
    hipFree(d_resume);
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

void free_results(ecb_handle* h) {
    hipFree(h->list); hipFree(h->order); hipFree(h->rank_of_slot); hipFree(h->indptr);
    hipFree(h->indices); hipFree(h->data); hipFree(h->counts);
    h->list = h->order = h->rank_of_slot = h->indptr = nullptr;
    h->indices = h->data = h->counts = nullptr;
}

int excl_scan(ecb_handle* h, const u32* in, u64 n, u32* out, u32* total) {
    const u64 nb = std::max<u64>(1, (n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    u32* sums = nullptr;
    HIPCHK(h, hipMalloc(&sums, (nb + 1) * sizeof(u32)));
    k_scan_sums<<<(unsigned)nb, TPB, 0, h->stream>>>(in, n, sums);
    k_scan_top<<<1, TPB, 0, h->stream>>>(sums, nb, sums + nb);
    k_scan_apply<<<(unsigned)nb, TPB, 0, h->stream>>>(in, n, sums, out);
    HIPCHK(h, hipMemcpyAsync(total, sums + nb, sizeof(u32), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(sums));
    return ECB_OK;
}

// reads per EC / first appearance, from read_slot[0, n_reads) (once, when the stream is closed)
int ensure_counts(ecb_handle* h) {
    if (h->counted) return ECB_OK;
    const u64 R = h->n_reads;
    if (R) {
        const u32 nb = (u32)std::max<u64>(1, h->cap >> BIN_BITS);
        if (nb > MAX_BUCKETS) return fail(h, ECB_ERR_LIMIT, "EC table larger than 2^26 slots is not supported yet");
        const u32 G = (u32)std::min<u64>(1024, (R + 4095) / 4096);
        u32 *hist = nullptr, *offs = nullptr;
        uint2* pairs = nullptr;
        HIPCHK(h, hipMalloc(&hist, (u64)nb * G * 4)); HIPCHK(h, hipMalloc(&offs, (u64)nb * G * 4));
        HIPCHK(h, hipMalloc(&pairs, R * sizeof(uint2)));
        k_part_hist<<<G, TPB, nb * 4, h->stream>>>(h->read_slot, R, nb, hist);
        u32 total = 0;
        int rc = excl_scan(h, hist, (u64)nb * G, offs, &total);
        if (rc == ECB_OK) {
            k_part_scatter<<<G, TPB, nb * 4, h->stream>>>(h->read_slot, R, nb, offs, pairs);
            k_count_bins<<<nb, TPB, 0, h->stream>>>(pairs, offs, G, nb, total, h->table);
            hipError_t e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) rc = fail(h, ECB_ERR_HIP, "k_count: %s", hipGetErrorString(e));
        }
        hipFree(hist); hipFree(offs); hipFree(pairs);
        if (rc != ECB_OK) return rc;
    }
    h->counted = true;
    return ECB_OK;
}

int compact_table(ecb_handle* h) {
    if (h->list) { hipFree(h->list); h->list = nullptr; }
    u64* d_n = nullptr;
    HIPCHK(h, hipMalloc(&d_n, sizeof(u64)));
    HIPCHK(h, hipMemsetAsync(d_n, 0, sizeof(u64), h->stream));
    HIPCHK(h, hipMalloc(&h->list, std::max<u64>(h->n_ecs(), 1) * sizeof(u32)));
    k_compact<<<(unsigned)std::min<u64>(4096, (h->cap + TPB - 1) / TPB), TPB, 0, h->stream>>>(h->table, h->cap, h->list, std::max<u64>(h->n_ecs(), 1), d_n);
    HIPCHK(h, hipMemcpyAsync(&h->n_list, d_n, sizeof(u64), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(d_n));
    if (h->n_list != h->n_ecs()) return fail(h, ECB_ERR_HIP, "internal: %llu occupied slots but %llu ECs created",
                                             (unsigned long long)h->n_list, (unsigned long long)h->n_ecs());
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
    hipMemsetAsync(h->ctr, 0, sizeof(Counters), h->stream);
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
    hipFree(h->table); hipFree(h->arena); hipFree(h->ctr); hipFree(h->read_slot);
    hipFree(h->rng_min); hipFree(h->rng_max); hipFree(h->queue); hipFree(h->newlist);
    hipFree(h->st_rid); hipFree(h->st_loc); hipFree(h->st_hf); hipFree(h->st_pos);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int ecb_reset(ecb_handle* h) {
    if (!h) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    free_results(h);
    HIPCHK(h, hipMemsetAsync(h->table, 0, h->cap * sizeof(Slot), h->stream));
    HIPCHK(h, hipMemsetAsync(h->ctr, 0, sizeof(Counters), h->stream));
    if (h->read_slot && h->reads_hi) HIPCHK(h, hipMemsetAsync(h->read_slot, 0xFF, h->reads_hi * sizeof(u32), h->stream));
    if (h->rng_min) {
        const u64 ns = (u64)h->cfg.n_loci * h->cfg.n_haplotypes;
        k_fill_i32<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng_min, ns, INT_MAX);
        k_fill_i32<<<nblk(ns, TPB), TPB, 0, h->stream>>>(h->rng_max, ns, INT_MIN);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->hctr = Counters{};
    h->n_ecs_stream = 0;
    h->prev_rid = 0xFFFFFFFFu; h->n_reads = 0; h->reads_hi = 0;
    h->extra_all = h->extra_valid = h->extra_reads = 0;
    h->c_rid.clear(); h->c_loc.clear(); h->c_hf.clear(); h->c_pos.clear();
    h->finalized = false; h->counted = false; h->sizes = ecb_sizes{}; h->n_list = 0;
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

int ecb_push_cells(ecb_handle* h, const uint32_t*, uint64_t, size_t) {
    if (!h) return ECB_ERR_ARG;
    return fail(h, ECB_ERR_STATE, "multisample is not built into this libecb yet");
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
    int rc = compact_table(h);
    if (rc != ECB_OK) return rc;
    // rank by first appearance: bitmap over read indices, popcount prefix
    const u64 total_reads = h->n_reads + h->extra_reads;
    const u64 words = (total_reads + 31) / 32 + 1;
    u32 *bitmap = nullptr, *wpop = nullptr, *wprefix = nullptr, *rowlen = nullptr;
    HIPCHK(h, hipMalloc(&bitmap, words * 4)); HIPCHK(h, hipMalloc(&wpop, words * 4)); HIPCHK(h, hipMalloc(&wprefix, words * 4));
    HIPCHK(h, hipMalloc(&rowlen, E * 4));
    HIPCHK(h, hipMalloc(&h->order, E * 4)); HIPCHK(h, hipMalloc(&h->rank_of_slot, h->cap * 4));
    HIPCHK(h, hipMalloc(&h->indptr, (E + 1) * 4)); HIPCHK(h, hipMalloc(&h->counts, E * 4));
    HIPCHK(h, hipMemsetAsync(bitmap, 0, words * 4, h->stream));
    k_mark_first<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->list, E, bitmap);
    k_popc<<<nblk(words, TPB), TPB, 0, h->stream>>>(bitmap, words, wpop);
    u32 tot = 0;
    rc = excl_scan(h, wpop, words, wprefix, &tot);
    if (rc != ECB_OK) return rc;
    if (tot != E) return fail(h, ECB_ERR_HIP, "internal: %u distinct first-appearance indices for %llu ECs", tot, (unsigned long long)E);
    k_rank<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->list, E, bitmap, wprefix, h->order, rowlen, h->rank_of_slot);
    u32 nnz = 0;
    {   // 64-bit check of the row-length total before trusting a 32-bit scan
        rc = excl_scan(h, rowlen, E, h->indptr, &nnz);
        if (rc != ECB_OK) return rc;
        if (h->hctr.arena_top >= (1ull << 31)) return fail(h, ECB_ERR_LIMIT, "A has more than 2^31-1 non-zeros");
        if ((u64)nnz != h->hctr.arena_top) return fail(h, ECB_ERR_HIP, "internal: nnz %u != arena %llu", nnz, (unsigned long long)h->hctr.arena_top);
        HIPCHK(h, hipMemcpyAsync(h->indptr + E, &nnz, 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipMalloc(&h->indices, std::max<u64>(nnz, 1) * 4)); HIPCHK(h, hipMalloc(&h->data, std::max<u64>(nnz, 1) * 4));
    k_emit<<<nblk(E * 64, TPB), TPB, 0, h->stream>>>(h->table, h->order, E, h->arena, h->indptr, h->indices, h->data, h->counts);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipFree(bitmap); hipFree(wpop); hipFree(wprefix); hipFree(rowlen);
    h->sizes.n_ecs = E; h->sizes.nnz_a = nnz; h->sizes.n_samples = 1; h->sizes.nnz_n = E;
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
    if (n_pairs) *n_pairs = h->hctr.arena_top;
    if (n_reads) *n_reads = h->n_reads;
    return ECB_OK;
}

int ecb_table_export_device(ecb_handle* h, void* d_entries, void* d_pairs, uint64_t read_base) {
    if (!h || !d_entries || !d_pairs) return ECB_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    if (read_base + h->n_reads >= (1ull << 32) - 1) return fail(h, ECB_ERR_LIMIT, "more than 2^32-2 reads in total");
    rc = ensure_counts(h);
    if (rc != ECB_OK) return rc;
    rc = compact_table(h);
    if (rc != ECB_OK) return rc;
    const u64 E = h->n_ecs();
    if (E) k_export_entries<<<nblk(E, TPB), TPB, 0, h->stream>>>(h->table, h->list, E, (Slot*)d_entries, (u32)read_base);
    HIPCHK(h, hipMemcpyAsync(d_pairs, h->arena, h->hctr.arena_top * sizeof(uint2), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ECB_OK;
}

int ecb_table_merge_device(ecb_handle* h, const void* d_entries, uint64_t n_entries, const void* d_pairs, uint64_t n_pairs) {
    if (!h) return ECB_ERR_ARG;
    if (h->finalized) return fail(h, ECB_ERR_STATE, "merge after finalize");
    if (!n_entries) return ECB_OK;
    if (!d_entries || (n_pairs && !d_pairs)) return fail(h, ECB_ERR_ARG, "null table buffers");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    rc = ensure_counts(h);                       // own reads first: merged counts are added on top
    if (rc != ECB_OK) return rc;
    while ((h->n_ecs() + n_entries) * 2 > h->cap) { rc = grow_table(h, h->cap * 4); if (rc != ECB_OK) return rc; }
    HIPCHK(h, hipMemsetAsync(&h->ctr->n_queue, 0, sizeof(u64), h->stream));
    k_merge<<<nblk(n_entries, TPB), TPB, 0, h->stream>>>((const Slot*)d_entries, n_entries, (const uint2*)d_pairs, n_pairs,
                                                          h->table, h->cap - 1, h->arena, h->arena_cap, h->ctr);
    rc = sync_counters(h);
    if (rc != ECB_OK) return rc;
    if (h->hctr.n_queue) return fail(h, ECB_ERR_TABLE_FULL, "internal: merge found no slot in a half-empty table");
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

/*
 * ecb.h -- C ABI of libecb.so, the MI355X (gfx950) equivalence-class builder behind
 * alntools' bam2ec / bam2emase hot path.
 *
 * The reference (churchill-lab/alntools v0.1.1, pure Python) has no FFI; its seam is the
 * call boundary between bam_utils.convert() and its per-chunk worker:
 *
 *     process_convert_bam(cp: ConvertParams) -> ConvertResults      alntools/bam_utils.py:198-363
 *         in : chunk descriptors of one BAM file                    alntools/bam_utils.py:37-47
 *         out: {valid_alignments, all_alignments, ec: OrderedDict[key -> count],
 *               unique_reads, tid_ranges}                           alntools/bam_utils.py:53-62,356-363
 *     merge of the workers' results, EC rank = first appearance     alntools/bam_utils.py:680-724
 *     A (incidence) / N (count) construction                        alntools/bam_utils.py:768-847
 *     (multisample variant: alntools/bam_utils_multisample.py:175-321, 503-793)
 *
 * The functions below replace exactly that: the host keeps decoding BAM (pysam) and streams
 * the decoded per-record tuples in; the library filters, groups records into reads, builds
 * each read's target set, reduces reads into equivalence classes in first-appearance order
 * and hands back the CSR A matrix (value = haplotype bitmask, bin_utils.py:208-211) and the
 * N matrix that alntools' writers (bin_utils.ecsave2, APM.save) consume.
 *
 * Conventions: plain C types only; caller owns every buffer it passes; the library owns
 * the device memory behind the opaque handle; return 0 = OK, < 0 = error code and
 * ecb_last_error() explains.  A handle is bound to one GPU and is not thread-safe.
 * There is no CPU fallback: without a usable HIP device ecb_create fails.
 *
 * Record tuple (struct-of-arrays, 12 bytes per BAM record):
 *   read_id  u32  run counter of reads.  A read is a run of equal (space-trimmed) query
 *                 names among the records that PASS the filter (bam_utils.py:289-320); the
 *                 host numbers those runs 0,1,2,... and every record -- passing or not --
 *                 carries the number of the latest run started at or before it
 *                 (0xFFFFFFFF before the first).  So read_id never decreases, steps by at
 *                 most 1, and steps only on a passing record; the library verifies this.
 *   locus    u32  index of the record's main target (bam_utils.py:596-598 order)
 *   hapflag  u32  bits 0-11 BAM flag; bit 12 = (refID != next_refID); bit 13 = (next_pos < 0)
 *                 (the two fields of the filter that are not in the flag, bam_utils.py:269);
 *                 bits 16-23 haplotype index (bam_utils.py:602 order); other bits 0
 *   pos      i32  reference_start, only with ECB_F_RANGES (bam_utils.py:282-286)
 */
#ifndef ECB_H
#define ECB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECB_ABI_VERSION 4

#define ECB_OK               0
#define ECB_ERR_ARG         -1   /* bad argument / configuration */
#define ECB_ERR_HIP         -2   /* HIP runtime error (message has the HIP error string) */
#define ECB_ERR_NO_DEVICE   -3   /* no usable gfx950 device */
#define ECB_ERR_TABLE_FULL  -4   /* EC table or key arena exhausted: raise ec_capacity / arena_capacity */
#define ECB_ERR_CONTRACT    -5   /* input violates the tuple contract (read_id steps, locus/hap range) */
#define ECB_ERR_STATE       -6   /* call out of order (push after finalize, export before finalize) */
#define ECB_ERR_EMPTY       -7   /* no valid alignment at all (the reference fails too: bam_utils.py:336-339) */
#define ECB_ERR_LIMIT       -8   /* result exceeds the .bin format's int32 limits (bin_utils.py:214-232) */
#define ECB_ERR_VERIFY      -9   /* ECB_F_VERIFY: the exactness pass found a read in an EC whose key is not its target set */

#define ECB_F_RANGES        1u   /* track min/max reference_start per (locus, haplotype) */
#define ECB_F_MULTISAMPLE   2u   /* per-read cell ids; N becomes EC x cell (bam_utils_multisample.py) */
#define ECB_F_VERIFY        4u   /* run the exactness pass (see ecb_verify_device) over every batch before the push returns */

#define ECB_HAP_SHIFT 16
#define ECB_FLAG_MATE_OTHER_REF 0x1000u
#define ECB_FLAG_NEXT_POS_NEG   0x2000u

typedef struct ecb_handle ecb_handle;

typedef struct ecb_config {
    uint32_t struct_size;        /* = sizeof(ecb_config) */
    int32_t  device;             /* HIP device ordinal */
    uint32_t n_loci;             /* T: number of main targets, 1 .. 2^26 - 3 (the stream kernel's LDS keys hold locus + 1 in 26 bits) */
    uint32_t n_haplotypes;       /* H: 1..31 */
    uint32_t flags;              /* ECB_F_* */
    uint32_t reserved;
    uint64_t ec_capacity;        /* EC hash-table slots (rounded up to a power of two; grows x4 at load 1/2, at most 2^28); 0 = default */
    uint64_t arena_capacity;     /* (locus, mask) pairs of EC key storage; 0 = default */
    uint64_t max_batch_records;  /* device staging size for ecb_push (host pointers); 0 = default */
} ecb_config;

typedef struct ecb_sizes {
    uint64_t n_ecs;              /* E */
    uint64_t nnz_a;              /* non-zeros of A */
    uint64_t n_samples;          /* S (1 unless multisample) */
    uint64_t nnz_n;              /* non-zeros of N */
    uint64_t all_alignments;     /* records offered (bam_utils.py:261) */
    uint64_t valid_alignments;   /* records that passed the filter (bam_utils.py:272) */
    uint64_t n_reads;            /* reads = name runs among valid records */
} ecb_sizes;

/* Lifetime. */
int  ecb_abi_version(void);
int  ecb_device_count(void);
int  ecb_create(const ecb_config* cfg, ecb_handle** out);
void ecb_destroy(ecb_handle* h);
int  ecb_reset(ecb_handle* h);                      /* forget all input and results, keep the allocations */
const char* ecb_last_error(const ecb_handle* h);   /* h may be NULL: error of the last failed ecb_create */

/* Streaming input -- replaces the per-alignment loop of process_convert_bam (bam_utils.py:258-344).
 * ecb_push: host pointers; batches may cut a read anywhere (the library carries the open read over).
 * ecb_push_device: device pointers (HBM-resident tuples); every call must hold whole reads.  The streams must be 16-byte aligned and
 *   complete when the call is made (the library works on a stream of its own).  Advice, measured: keep the three (four) streams in ONE
 *   device allocation -- where separate multi-GB allocations land moves the stream kernel by 8 - 10 % on MI355X
 *   (profiles/r04_stream_placement*.txt); the library's own staging for ecb_push is laid out that way.
 * pos may be NULL unless ECB_F_RANGES. */
int ecb_push(ecb_handle* h, const uint32_t* read_id, const uint32_t* locus, const uint32_t* hapflag,
             const int32_t* pos, size_t n);
int ecb_push_device(ecb_handle* h, const void* d_read_id, const void* d_locus, const void* d_hapflag,
                    const void* d_pos, size_t n);
/* The same from ONE device buffer of whole tiles (ABI 4): tile t = uint32 words [1536 t, 1536 t + 1536) = the 512 read ids | 512 loci | 512
 * haplotype/flag words of records [512 t, 512 t + 512); the buffer holds ceil(n / 512) tiles (the last one padded with anything), 16-byte
 * aligned; whole reads per call as for ecb_push_device; not with ECB_F_RANGES.  A tile of the stream kernel then reads 6 KB in one place
 * instead of 2 KB in each of three multi-GB arrays: on config 3 that narrows how far the kernel's time moves with where the tuples sit in
 * HBM (7.7 - 8.7 ms against 7.7 - 9.3 ms), without removing it.  Batches of the two kinds may be mixed on one handle. */
int ecb_push_device_tiled(ecb_handle* h, const void* d_tiles, size_t n);
/* Optional: the whole stream holds at most max_reads reads (reads with a record that passes the filter).  With the bound
 * known, a push of device-resident tuples sizes its per-read state from it and no longer asks the device for the batch's
 * last read id before it launches anything: one host wait per push instead of two.  A stream that runs past the bound is
 * ECB_ERR_CONTRACT.  Kept across ecb_reset; 0 takes it back.  With the bound known the library also sees how many records a read
 * brings on average, and gives batches of short reads (fewer than seven records per read, n_loci < 2^25 - 2, at most 8 haplotypes) to a stream kernel laid out for them. */
int ecb_hint_reads(ecb_handle* h, uint64_t max_reads);

/* Multisample only (ECB_F_MULTISAMPLE): per read, in read order, for reads [first_read, first_read + n):
 *   meta = cell id (bits 0-21; dictionary-encoded by the host from the read name, bam_utils_multisample.py:270-280)
 *        | input file index << 22 (bits 22-31; the reference scans one file per worker, :473-480).
 * The host leaves out the last read of every file (the reference never counts it, :306-321). */
int ecb_push_cells(ecb_handle* h, const uint32_t* meta, uint64_t first_read, size_t n);
/* The same from device memory (the cell stream of a resident workload: +4 bytes per read).  The copy is queued on the
 * handle's own stream and waits for nothing of the caller's: whatever produced d_meta must have COMPLETED before the call (as for
 * the streams of ecb_push_device), and the caller keeps d_meta alive and unchanged until the next call that waits for the handle
 * (ecb_finalize).  first_read + n beyond 2^32 - 2 reads: ECB_ERR_LIMIT. */
int ecb_push_cells_device(ecb_handle* h, const void* d_meta, uint64_t first_read, size_t n);
#define ECB_CELL_BITS 22

/* Close the stream: rank ECs by first appearance (bam_utils.py:682-698), build CSR A and N. */
int ecb_finalize(ecb_handle* h, ecb_sizes* out);

/* Results, into caller buffers of the sizes ecb_finalize reported (int32, as the .bin stores them).
 * A: CSR over (EC, locus), columns ascending, value = OR of 1 << haplotype.
 * N: CSC over (EC, sample); single-sample: indptr = {0, E}, indices = 0..E-1, data = counts. */
int ecb_export(ecb_handle* h, int32_t* indptr_a, int32_t* indices_a, int32_t* data_a,
               int32_t* indptr_n, int32_t* indices_n, int32_t* data_n);
int ecb_export_device(ecb_handle* h, void* d_indptr_a, void* d_indices_a, void* d_data_a,
                      void* d_indptr_n, void* d_indices_n, void* d_data_n);
/* ECB_F_RANGES: per (locus, haplotype) max - min + 1 of reference_start over valid alignments, 0 if
 * none (the numbers of the reference's range file, bam_utils.py:756-763); n_loci * n_haplotypes values. */
int ecb_export_ranges(ecb_handle* h, int64_t* range_len);
/* The raw extremes behind ecb_export_ranges (min = INT32_MAX and max = INT32_MIN where nothing aligned). */
int ecb_export_range_minmax(ecb_handle* h, int32_t* range_min, int32_t* range_max);
/* Multisample, after ecb_finalize: the distinct (EC, cell, file) triples -- sizes.nnz_n of them, sorted by
 * (EC, cell, file) -- with the number of reads and the first read index of each: what the reference keeps as
 * ec[key][cell] per worker (bam_utils_multisample.py:288-290, 503-576).  Cell order, the minimum-count filter
 * and the CSC N matrix (:596-636, 737-791) are metadata-sized work done by the host from these. */
int ecb_export_pairs(ecb_handle* h, uint32_t* ec, uint32_t* meta, uint32_t* count, uint32_t* first_read);
/* Multisample, after ecb_finalize: what the reference does with its merged ec[key][cell] dicts (bam_utils_multisample.py:
 * 596-636, 737-791), on the device, from those triples: cells in the insertion order of the reference's cr_totals (files in
 * order; within a file ECs by first appearance there; within an EC cells by first appearance), cells with fewer than
 * minimum_count reads (<= 0 means 1) dropped, ECs left without a cell dropped and the rest re-ranked, N as CSC over
 * (kept EC, kept cell), and the rows of A of the kept ECs.  n_cells = number of cell ids the host handed out.
 * ecb_ms_export fills caller buffers of the sizes ecb_ms_filter reported: kept_cells[n_cells_kept] = cell ids in sample
 * order; A as CSR (n_ecs_kept + 1, nnz_a, nnz_a); N as CSC (n_cells_kept + 1, nnz_n, nnz_n).  Any pointer may be NULL. */
typedef struct ecb_ms_sizes {
    uint64_t n_cells_seen;       /* cells with at least one read */
    uint64_t n_cells_kept;       /* S */
    uint64_t n_ecs_kept;         /* E after the filter */
    uint64_t nnz_a;
    uint64_t nnz_n;
} ecb_ms_sizes;
int ecb_ms_filter(ecb_handle* h, uint32_t n_cells, int64_t minimum_count, ecb_ms_sizes* out);
int ecb_ms_export(ecb_handle* h, uint32_t* kept_cells, int32_t* indptr_a, int32_t* indices_a, int32_t* data_a,
                  int32_t* indptr_n, int32_t* indices_n, int32_t* data_n);
/* EC index of every read, in read order (n_reads values). */
int ecb_export_read_ec(ecb_handle* h, int32_t* ec_of_read);

/* Exactness.  EC identity is exact by construction: a read joins an EC only after its {locus -> haplotype mask} set has
 * been compared, pair by pair, with the key stored for that EC (the reference compares the sorted tid strings,
 * bam_utils.py:307-312); the 64-bit set hash only picks the table slot, and two target sets with one hash get two slots.
 * The same holds for the multi-GPU merge (ecb_table_merge_*) and the multisample rank lookup.
 * ecb_verify_device is the independent re-check used by the tests and by ECB_F_VERIFY: a second pass re-derives the set
 * of every read from the records and compares it with the key of the EC the read was assigned to.  The records must be
 * the device-resident stream that was pushed (one ecb_push_device call covering the whole stream).  *n_mismatch counts
 * reads whose set differs from their EC's key (0 = the grouping is exact); *n_long counts reads longer than a tile,
 * which are re-checked too, on the long-read path. */
int ecb_verify_device(ecb_handle* h, const void* d_read_id, const void* d_locus, const void* d_hapflag, size_t n,
                      uint64_t* n_mismatch, uint64_t* n_long);
int ecb_verify_device_tiled(ecb_handle* h, const void* d_tiles, size_t n, uint64_t* n_mismatch, uint64_t* n_long_reads);

/* Multi-GPU: one handle per GPU over contiguous read shards (the reference's contiguous chunk
 * ranges per process, bam_utils.py:646-658).  A rank serialises its EC table (device buffers the
 * caller allocates: n_entries * 32 bytes and n_pairs * 8 bytes), the caller moves it (RCCL), and the
 * receiving rank merges it -- the reference's ordered merge, bam_utils.py:680-724.
 * read_base = number of reads on all lower ranks (makes "first appearance" global).
 * (There is no merge call that takes an array of handles and runs the collective inside the library: libecb links neither RCCL nor MPI -- it hands out and
 *  takes device buffers, and whoever owns the process group moves them.  alntools_amd/dist.py is that owner for torch.distributed (backend
 *  "nccl" = RCCL); INTEGRATION.md section 3 lists the message pattern for a host that is not Python.)
 * ecb_table_sizes: *n_pairs is an upper bound of the key pairs in use (buffer size); the export writes the keys of the
 * entries, each sorted by locus, packed behind each other; an entry is 32 bytes {u64 hash, u64 reserved, u32 count,
 * u32 ~first_read, u32 off, u32 n} with its key at pairs[off .. off + n) of its part.  The merge compares keys, not hashes. */
int ecb_table_sizes(ecb_handle* h, uint64_t* n_entries, uint64_t* n_pairs, uint64_t* n_reads);
int ecb_table_export_device(ecb_handle* h, void* d_entries, void* d_pairs, uint64_t read_base);
int ecb_table_merge_device(ecb_handle* h, const void* d_entries, uint64_t n_entries,
                           const void* d_pairs, uint64_t n_pairs);
/* The same merge spread over the ranks (no serial root): ecb_table_export_parts_device writes the table grouped into
 * n_parts (<= 64) key ranges -- part q = entries [entry_offsets[q], entry_offsets[q+1]) and pairs [pair_offsets[q],
 * pair_offsets[q+1]), offsets returned in host arrays of n_parts + 1; buffers sized by ecb_table_sizes as above -- so
 * that rank q receives part q of every rank and merges them (ecb_table_merge_device, in rank order).  The merged parts
 * hold disjoint ECs: the root loads them with ecb_table_adopt_device (consecutive slots of an EMPTY handle, no hashing;
 * afterwards that handle only takes more adopts, ecb_add_counters, ecb_finalize and the exports). */
int ecb_table_export_parts_device(ecb_handle* h, void* d_entries, void* d_pairs, uint64_t read_base, uint32_t n_parts,
                                  uint64_t* entry_offsets, uint64_t* pair_offsets);
int ecb_table_adopt_device(ecb_handle* h, const void* d_entries, uint64_t n_entries,
                           const void* d_pairs, uint64_t n_pairs);
/* Exported entries carry first-read indices counted from read_base of their export; when the shard's place in the run is only
 * known later (after the one exchange of sizes), the receiver moves them on: first += read_base for n_entries entries, in
 * place, ordered on h's stream (no wait) ahead of ecb_table_merge_device / ecb_table_adopt_device on the same handle. */
int ecb_table_rebase_device(ecb_handle* h, void* d_entries, uint64_t n_entries, uint64_t read_base);
/* Several tables in one call, in the order given (arrays of n_tables device pointers / sizes): the kernels queue up
 * behind each other and the host waits once, not once per table. */
int ecb_table_merge_batch_device(ecb_handle* h, uint32_t n_tables, const void* const* d_entries, const uint64_t* n_entries,
                                 const void* const* d_pairs, const uint64_t* n_pairs);
int ecb_table_adopt_batch_device(ecb_handle* h, uint32_t n_tables, const void* const* d_entries, const uint64_t* n_entries,
                                 const void* const* d_pairs, const uint64_t* n_pairs);
/* Finalize per key range (single-sample runs): instead of sending its merged range's table to the root, rank q sets the
 * totals of the whole run on the handle that merged range q (ecb_add_counters: read indices are global there) and calls
 * ecb_finalize on it -- the ranking and the CSR emit of 1/N of the ECs -- then exports CSR A / counts (ecb_export_device) and
 * the first read of every EC, in the same order (ecb_export_firsts_device: n_ecs x uint32).  The root puts the N pieces
 * together on an EMPTY handle with ecb_assemble_ranges_device: the global rank of an EC = the number of ECs with an earlier
 * first read (a bitmap over the reads, marked from all pieces, and its prefix popcount), rows copied to their places.
 * Afterwards that handle behaves as finalized (ecb_finalize returns the sizes again; ecb_export, ecb_export_device), but it
 * holds no table: the per-read and hash exports refuse.  ECB_ERR_CONTRACT if two pieces name the same first read.
 * (bam_utils.py:680-724: the ordered merge of the workers' dicts; the order of the result is the same.) */
int ecb_export_firsts_device(ecb_handle* h, void* d_firsts);
int ecb_assemble_ranges_device(ecb_handle* h, uint32_t n_pieces, const void* const* d_indptr, const void* const* d_indices,
                               const void* const* d_data, const void* const* d_counts, const void* const* d_firsts,
                               const uint64_t* n_ecs, const uint64_t* nnz, uint64_t total_reads, uint64_t all_alignments,
                               uint64_t valid_alignments, ecb_sizes* out);
/* The whole multi-GPU merge in one call, for ONE process that drives several GPUs (ABI 4; SURVEY.md 8b's ecb_merge): shards[r] holds
 * contiguous read shard r of the run on its own device (pushed, not finalized), root is an empty handle on any device.  Runs the steps
 * above -- key ranges cut, range q of every shard copied to shard q's device (hipMemcpyPeer: xGMI between the GPUs of a node), merged
 * there in shard order, finalized there, the pieces assembled on the root -- and leaves root finalized (ecb_export / ecb_export_device);
 * the shards are spent (counted: they take no further pushes).  Single-sample handles only.  The devices work one after the other;
 * one process per GPU over RCCL (alntools_amd/dist.py) runs the same steps side by side.  (bam_utils.py:646-724.) */
int ecb_merge(ecb_handle* const* shards, uint32_t n_shards, ecb_handle* root, ecb_sizes* out);
/* Multisample across GPUs (the shards' handles and the root's adopting handle all carry ECB_F_MULTISAMPLE).  After the ECs
 * were merged and the root finalized: ecb_export_ec_keys_device writes the 8-byte set hash of every EC in rank order
 * (n_ecs * 8 bytes; broadcast it together with the root's CSR A from ecb_export_device).  A shard finds its own ECs in
 * that list -- by hash, then by comparing its stored key with the CSR row -- and reduces its reads to distinct
 * (EC, cell, file) triples with GLOBAL EC ids -- ecb_ms_local_triples_device: key = EC << 32 | meta (in the order (EC, cell, file)), count,
 * first read (read_base added); buffers of n_reads elements, *n_triples written.  The root combines the shards' triples
 * (a cell whose reads straddle two shards: counts added, first = min) with ecb_ms_adopt_triples_device, after which
 * ecb_export_pairs works as on one GPU.  (bam_utils_multisample.py:503-576: the merge of the workers' ec[key][cell].) */
int ecb_export_ec_keys_device(ecb_handle* h, void* d_keys);
int ecb_ms_local_triples_device(ecb_handle* h, const void* d_keys, const void* d_indptr_a, const void* d_indices_a,
                                const void* d_data_a, uint64_t n_ecs, uint64_t read_base,
                                void* d_key, void* d_count, void* d_first, uint64_t* n_triples);
int ecb_ms_adopt_triples_device(ecb_handle* h, uint32_t n_tables, const void* const* d_key, const void* const* d_count,
                                const void* const* d_first, const uint64_t* n, uint64_t* n_triples);
int ecb_counters(ecb_handle* h, uint64_t* all_alignments, uint64_t* valid_alignments, uint64_t* n_reads);
int ecb_add_counters(ecb_handle* h, uint64_t all_alignments, uint64_t valid_alignments, uint64_t n_reads);

/* f-2: the sparse-format conversions behind ec2emase / emase2ec (bin_utils.py:979-1028), on device arrays.
 * .bin holds A as one CSR over (EC, locus) whose value is the haplotype bitmask (bin_utils.py:208-211); EMASE holds one
 * CSC matrix (E x T, row indices ascending) per haplotype (Sparse3DMatrix.py:189-193, 325-342).
 * ecb_csr_to_hapcsc_device: call with d_csc_indices = NULL to learn *total (= sum of popcounts), then again to fill
 *   d_csc_indptr (int32, H x (T+1), each haplotype's pointers starting at 0) and d_csc_indices (int32 rows; haplotype h's
 *   block starts at the sum of the earlier haplotypes' nnz).
 * ecb_hapcsc_to_csr_device: the inverse (A = sum_h 2^h M_h, columns ascending); d_indices/d_data hold at most `total`
 *   entries, *nnz receives the count. */
int ecb_csr_to_hapcsc_device(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const void* d_indptr_a,
                             const void* d_indices_a, const void* d_data_a, void* d_csc_indptr, void* d_csc_indices,
                             uint64_t* total);
/* The conversions keep their device scratch between calls (per device); this gives it back. */
int ecb_release_scratch(int device);
int ecb_hapcsc_to_csr_device(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const void* d_csc_indptr,
                             const void* d_csc_indices, uint64_t total, void* d_indptr_a, void* d_indices_a,
                             void* d_data_a, uint64_t* nnz);

/* The same two conversions from and to HOST arrays (ABI 4): the library allocates, fills and frees its own device buffers, so a
 * host that only converts files needs no device allocator -- what the reference's ec2emase / emase2ec and the .h5 writer of
 * bam2emase do on the CPU (bin_utils.py:979-1028, Sparse3DMatrix.py:189-193).
 * ecb_csr_to_hapcsc: csc_indices == NULL -> *total = number of row indices (set bits of the masks) and nothing else; otherwise
 *   csc_indptr holds H x (T+1) int32 and csc_indices `capacity` int32 (ECB_ERR_ARG when that is too few; *total says how many).
 * ecb_hapcsc_to_csr: indices / data hold `total` int32 each (an upper bound of the non-zeros), indptr n_ecs + 1; *nnz = non-zeros. */
int ecb_csr_to_hapcsc(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const int32_t* indptr_a, const int32_t* indices_a,
                      const int32_t* data_a, int32_t* csc_indptr, int32_t* csc_indices, uint64_t capacity, uint64_t* total);
int ecb_hapcsc_to_csr(int device, uint32_t n_ecs, uint32_t n_loci, uint32_t n_haps, const int32_t* csc_indptr, const int32_t* csc_indices,
                      uint64_t total, int32_t* indptr_a, int32_t* indices_a, int32_t* data_a, uint64_t* nnz);

/* Measurement: HIP-event time of the record-stream kernel on the handle's own stream; ecb_profile_kernel (ABI 4): the name of the
 * kernel the last batch launched, as rocprofv3 prints it (the stream kernel is compiled more than once; the library picks per batch). */
int ecb_profile(ecb_handle* h, int enable);
int ecb_profile_read(ecb_handle* h, double* kernel_ms, uint64_t* launches, uint64_t* records);
const char* ecb_profile_kernel(const ecb_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* ECB_H */

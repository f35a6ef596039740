/* bamdec -- the host side's BAM decoder: what iterating a pysam.AlignmentFile gives the reference's scan loop
 * (alntools/bam_utils.py:253-320), as column arrays, without a Python loop over records.
 *
 * Per record it yields the raw fields the tuple encoder needs (flag, refID, pos, next_refID, next_pos), whether the record
 * passes the reference's filter (bam_utils.py:264-270) and whether it starts a new read: reads are RUNS of equal names among
 * the records that pass the filter, a name being cut at its first space if that space is not its first character
 * (bam_utils.py:289-320).  Names are compared as bytes here (the reference compares the decoded strings: the same thing for
 * valid UTF-8).  BGZF blocks are inflated by a small pool of threads (blocks are independent; ISIZE gives every block's
 * place in the output before it is inflated); records are parsed by the caller's thread.
 *
 * Plain C + zlib; bound from Python with ctypes (alntools_amd/bamdec.py).  Not part of libecb: the device library never
 * touches files.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "../../include/bamdec.h"


#define BLOCKS_PER_FILL 512            /* up to 32 MB of inflated data per refill */
#define MAX_THREADS 32

typedef struct {
    const uint8_t* src; uint32_t src_len;       /* raw deflate stream of one block */
    uint8_t* dst; uint32_t dst_len;             /* where it goes, and ISIZE */
    uint32_t crc;
} bd_block;

typedef struct bd_handle {
    FILE* f; uint64_t file_size;
    uint8_t* raw; size_t raw_cap;               /* compressed bytes of the blocks being inflated */
    uint8_t* buf; size_t cap, len, pos;         /* inflated stream: [pos, len) is unread */
    bd_block blocks[BLOCKS_PER_FILL];
    int n_threads;
    int eof;
    uint64_t fill_cstart, fill_clen, fill_ulen; /* the last refill: where its compressed bytes start in the file, how many, and what they inflated to */
    /* header */
    char* text; int32_t l_text; int32_t n_ref; char** names; int32_t* lens;
    char* ref_blob; size_t ref_blob_len;
    void* scratch; size_t scratch_cap;          /* bd_read_tuples: the decoded fields of one chunk of records */       /* every name with its NUL, back to back (bd_references; made on demand) */
    /* the (trimmed) name of the latest record that passed the filter */
    uint8_t* last; size_t last_cap; int32_t last_len; int has_last;
    /* multisample (bd_read_ms): the cell ids of the runs the last call started, back to back, and where each begins */
    uint8_t* cells; size_t cells_cap, cells_len; uint32_t* cell_off; size_t cell_off_cap, n_cells;
    char err[256];
} bd_handle;

static int fail(bd_handle* h, int code, const char* msg) {
    if (h) { strncpy(h->err, msg, sizeof(h->err) - 1); h->err[sizeof(h->err) - 1] = 0; }
    return code;
}

/* ---- inflate pool ---------------------------------------------------------------------------------------------- */
typedef struct { bd_block* blocks; int n, first, step; int rc; } bd_job;

static void* inflate_worker(void* arg) {
    bd_job* j = (bd_job*)arg;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) { j->rc = BD_ERR_MEM; return NULL; }
    for (int i = j->first; i < j->n; i += j->step) {
        bd_block* b = &j->blocks[i];
        if (b->dst_len == 0) continue;
        inflateReset(&zs);
        zs.next_in = (Bytef*)b->src; zs.avail_in = b->src_len;
        zs.next_out = b->dst; zs.avail_out = b->dst_len;
        const int r = inflate(&zs, Z_FINISH);
        if (r != Z_STREAM_END || zs.avail_out != 0 || crc32(crc32(0L, Z_NULL, 0), b->dst, b->dst_len) != b->crc) { j->rc = BD_ERR_FORMAT; break; }
    }
    inflateEnd(&zs);
    return NULL;
}

/* read up to BLOCKS_PER_FILL BGZF blocks and append their inflated bytes to the stream buffer */
static int refill(bd_handle* h) {
    if (h->eof) return BD_OK;
    /* keep the unread tail at the front */
    if (h->pos) { memmove(h->buf, h->buf + h->pos, h->len - h->pos); h->len -= h->pos; h->pos = 0; }
    size_t raw_used = 0, out_total = 0;
    int nb = 0;
    const long c_at = ftell(h->f);
    while (nb < BLOCKS_PER_FILL) {
        uint8_t hd[18];
        const size_t got = fread(hd, 1, 18, h->f);
        if (got == 0) { h->eof = 1; break; }
        if (got != 18 || hd[0] != 0x1f || hd[1] != 0x8b || hd[2] != 8 || !(hd[3] & 4)) return fail(h, BD_ERR_FORMAT, "not a BGZF block");
        const uint32_t xlen = hd[10] | (hd[11] << 8);
        /* the BC subfield is the first one in every BGZF writer's output; walk the extra field to be safe */
        uint32_t bsize = 0;
        uint8_t extra[65536];
        memcpy(extra, hd + 12, 6);
        if (xlen > 6 && fread(extra + 6, 1, xlen - 6, h->f) != xlen - 6) return fail(h, BD_ERR_IO, "truncated BGZF header");
        for (uint32_t o = 0; o + 4 <= xlen;) {
            const uint32_t sl = extra[o + 2] | (extra[o + 3] << 8);
            if (extra[o] == 'B' && extra[o + 1] == 'C' && sl == 2 && o + 6 <= xlen) { bsize = (extra[o + 4] | (extra[o + 5] << 8)) + 1u; break; }
            o += 4 + sl;
        }
        if (bsize < 12 + xlen + 8) return fail(h, BD_ERR_FORMAT, "BGZF block without a BC field");
        const uint32_t clen = bsize - 12 - xlen;             /* deflate data + CRC32 + ISIZE */
        if (raw_used + clen > h->raw_cap) {
            const size_t nc = (raw_used + clen) * 2 + (1 << 20);
            uint8_t* nr = (uint8_t*)realloc(h->raw, nc);
            if (!nr) return fail(h, BD_ERR_MEM, "out of memory");
            /* pointers into the old buffer */
            for (int i = 0; i < nb; ++i) h->blocks[i].src = nr + (h->blocks[i].src - h->raw);
            h->raw = nr; h->raw_cap = nc;
        }
        if (fread(h->raw + raw_used, 1, clen, h->f) != clen) return fail(h, BD_ERR_IO, "truncated BGZF block");
        const uint8_t* tail = h->raw + raw_used + clen - 8;
        bd_block* b = &h->blocks[nb++];
        b->src = h->raw + raw_used; b->src_len = clen - 8;
        b->crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
        b->dst_len = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        b->dst = (uint8_t*)(uintptr_t)out_total;             /* offset for now */
        if (b->dst_len > 65536) return fail(h, BD_ERR_FORMAT, "BGZF block larger than 64 KiB");
        raw_used += clen; out_total += b->dst_len;
    }
    if (nb == 0) return BD_OK;
    if (h->len + out_total > h->cap) {
        const size_t nc = (h->len + out_total) * 2 + (1 << 20);
        uint8_t* nbuf = (uint8_t*)realloc(h->buf, nc);
        if (!nbuf) return fail(h, BD_ERR_MEM, "out of memory");
        h->buf = nbuf; h->cap = nc;
    }
    for (int i = 0; i < nb; ++i) h->blocks[i].dst = h->buf + h->len + (size_t)(uintptr_t)h->blocks[i].dst;
    int nt = h->n_threads < nb ? h->n_threads : nb;
    if (nt < 1) nt = 1;
    bd_job jobs[MAX_THREADS];
    pthread_t th[MAX_THREADS];
    for (int t = 0; t < nt; ++t) { jobs[t].blocks = h->blocks; jobs[t].n = nb; jobs[t].first = t; jobs[t].step = nt; jobs[t].rc = BD_OK; }
    int started = 0;
    for (int t = 1; t < nt; ++t) { if (pthread_create(&th[t], NULL, inflate_worker, &jobs[t]) != 0) break; started = t; }
    if (started != nt - 1) {                                  /* could not start them all: the caller's thread does every block */
        for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
        jobs[0].first = 0; jobs[0].step = 1;
        inflate_worker(&jobs[0]);
        if (jobs[0].rc != BD_OK) return fail(h, jobs[0].rc, "corrupt BGZF block");
    } else {
        inflate_worker(&jobs[0]);
        for (int t = 1; t < nt; ++t) pthread_join(th[t], NULL);
        for (int t = 0; t < nt; ++t) if (jobs[t].rc != BD_OK) return fail(h, jobs[t].rc, "corrupt BGZF block");
    }
    h->len += out_total;
    { const long c_end = ftell(h->f); h->fill_cstart = c_at < 0 ? 0 : (uint64_t)c_at; h->fill_clen = c_end > c_at ? (uint64_t)(c_end - c_at) : 0; h->fill_ulen = out_total; }
    return BD_OK;
}

/* make sure n unread bytes are there; 0 = yes, 1 = the stream ended before, < 0 = error */
static int need(bd_handle* h, size_t n) {
    while (h->len - h->pos < n) {
        if (h->eof) return 1;
        const int rc = refill(h);
        if (rc != BD_OK) return rc;
    }
    return 0;
}

static inline int32_t rd_i32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }   /* (little-endian hosts only, like the rest of the repo) */

/* ---- API ------------------------------------------------------------------------------------------------------- */
int bd_abi_version(void) { return 4; }

/* How far the records handed out so far reach into the file, in compressed bytes, and how long the file is: a monotone measure of
 * progress.  What has been read ahead and inflated but not parsed yet is NOT counted (up to 512 blocks -- a whole small file --
 * sit in the buffer at a time): the unread part of the last refill is taken off in proportion to its inflated size.  (ABI 3 counted
 * the read-ahead, which made a multi-GPU deal-out of a small file hand nearly everything to the last rank.) */
int bd_progress(bd_handle* h, uint64_t* consumed, uint64_t* total) {
    if (!h || !h->f || !consumed || !total) return BD_ERR_ARG;
    uint64_t unread = (uint64_t)(h->len - h->pos);
    if (unread > h->fill_ulen) unread = h->fill_ulen;
    const uint64_t back = h->fill_ulen ? (uint64_t)((double)h->fill_clen * (double)unread / (double)h->fill_ulen) : 0;
    uint64_t at = h->fill_cstart + h->fill_clen - back;
    if (h->eof && unread == 0) at = h->file_size;
    *consumed = at > h->file_size ? h->file_size : at; *total = h->file_size;
    return BD_OK;
}

void bd_close(bd_handle* h) {
    if (!h) return;
    if (h->f) fclose(h->f);
    free(h->raw); free(h->buf); free(h->text); free(h->last); free(h->cells); free(h->cell_off);
    if (h->names) { for (int32_t i = 0; i < h->n_ref; ++i) free(h->names[i]); free(h->names); }
    free(h->lens); free(h->ref_blob); free(h->scratch);
    free(h);
}

const char* bd_last_error(const bd_handle* h) { return h ? h->err : "null handle"; }

int bd_open(const char* path, int n_threads, bd_handle** out) {
    if (!path || !out) return BD_ERR_ARG;
    *out = NULL;
    bd_handle* h = (bd_handle*)calloc(1, sizeof(bd_handle));
    if (!h) return BD_ERR_MEM;
    h->n_threads = n_threads < 1 ? 1 : (n_threads > MAX_THREADS ? MAX_THREADS : n_threads);
    h->f = fopen(path, "rb");
    if (!h->f) { free(h); return BD_ERR_IO; }
    setvbuf(h->f, NULL, _IOFBF, 1 << 22);
    if (fseek(h->f, 0, SEEK_END) == 0) { const long sz = ftell(h->f); h->file_size = sz > 0 ? (uint64_t)sz : 0; }
    if (fseek(h->f, 0, SEEK_SET) != 0) { fclose(h->f); free(h); return BD_ERR_IO; }
    int rc = need(h, 12);
    if (rc != 0 || memcmp(h->buf + h->pos, "BAM\1", 4) != 0) { bd_close(h); return BD_ERR_FORMAT; }
    h->l_text = rd_i32(h->buf + h->pos + 4);
    if (h->l_text < 0 || need(h, 12 + (size_t)h->l_text) != 0) { bd_close(h); return BD_ERR_FORMAT; }
    h->text = (char*)malloc((size_t)h->l_text + 1);
    if (!h->text) { bd_close(h); return BD_ERR_MEM; }
    memcpy(h->text, h->buf + h->pos + 8, (size_t)h->l_text); h->text[h->l_text] = 0;
    h->n_ref = rd_i32(h->buf + h->pos + 8 + h->l_text);
    h->pos += 12 + (size_t)h->l_text;
    if (h->n_ref < 0) { bd_close(h); return BD_ERR_FORMAT; }
    h->names = (char**)calloc((size_t)h->n_ref + 1, sizeof(char*));
    h->lens = (int32_t*)calloc((size_t)h->n_ref + 1, sizeof(int32_t));
    if (!h->names || !h->lens) { bd_close(h); return BD_ERR_MEM; }
    for (int32_t i = 0; i < h->n_ref; ++i) {
        if (need(h, 4) != 0) { bd_close(h); return BD_ERR_FORMAT; }
        const int32_t ln = rd_i32(h->buf + h->pos);
        if (ln < 1 || need(h, 8 + (size_t)ln) != 0) { bd_close(h); return BD_ERR_FORMAT; }
        h->names[i] = (char*)malloc((size_t)ln);
        if (!h->names[i]) { bd_close(h); return BD_ERR_MEM; }
        memcpy(h->names[i], h->buf + h->pos + 4, (size_t)ln); h->names[i][ln - 1] = 0;
        h->lens[i] = rd_i32(h->buf + h->pos + 4 + ln);
        h->pos += 8 + (size_t)ln;
    }
    *out = h;
    return BD_OK;
}

int32_t bd_n_references(const bd_handle* h) { return h ? h->n_ref : 0; }
const char* bd_reference_name(const bd_handle* h, int32_t i) { return (h && i >= 0 && i < h->n_ref) ? h->names[i] : NULL; }
int32_t bd_reference_length(const bd_handle* h, int32_t i) { return (h && i >= 0 && i < h->n_ref) ? h->lens[i] : 0; }
const char* bd_header_text(const bd_handle* h) { return h ? h->text : NULL; }

int bd_references(bd_handle* h, const char** blob, size_t* blob_len, const int32_t** lens) {
    if (!h || !blob || !blob_len || !lens) return BD_ERR_ARG;
    if (!h->ref_blob) {
        size_t total = 1;
        for (int32_t i = 0; i < h->n_ref; ++i) total += strlen(h->names[i]) + 1;
        char* b = (char*)malloc(total);
        if (!b) return fail(h, BD_ERR_MEM, "out of memory");
        size_t at = 0;
        for (int32_t i = 0; i < h->n_ref; ++i) { const size_t l = strlen(h->names[i]) + 1; memcpy(b + at, h->names[i], l); at += l; }
        h->ref_blob = b; h->ref_blob_len = at;
    }
    *blob = h->ref_blob; *blob_len = h->ref_blob_len; *lens = h->lens;
    return BD_OK;
}

/* Up to max_records records.  Arrays of max_records elements each; *n_out = records written (0 = end of file).
 * valid[i] = the record passes bam_utils.py:264-270; head[i] = it is valid and its (trimmed, if trim != 0) name differs from
 * the previous valid record's -- across calls too. */
int bd_read(bd_handle* h, size_t max_records, int trim, uint16_t* flag, int32_t* tid, int32_t* pos, int32_t* next_tid, int32_t* next_pos,
            uint8_t* valid, uint8_t* head, size_t* n_out) {
    if (!h || !flag || !tid || !pos || !next_tid || !next_pos || !valid || !head || !n_out) return BD_ERR_ARG;
    size_t n = 0;
    while (n < max_records) {
        /* fast path: the whole record is in the buffer (all but one record per 32 MB refill) */
        size_t avail = h->len - h->pos;
        int32_t bs = avail >= 4 ? rd_i32(h->buf + h->pos) : 0;
        if (avail < 4 || avail < 4 + (size_t)(bs < 0 ? 0 : bs)) {
            int rc = need(h, 4);
            if (rc == 1) { if (h->len != h->pos) return fail(h, BD_ERR_FORMAT, "truncated BAM"); break; }
            if (rc < 0) return rc;
            bs = rd_i32(h->buf + h->pos);
            if (bs < 32) return fail(h, BD_ERR_FORMAT, "BAM record shorter than its fixed part");
            rc = need(h, 4 + (size_t)bs);
            if (rc != 0) return rc < 0 ? rc : fail(h, BD_ERR_FORMAT, "truncated BAM record");
        }
        if (bs < 32) return fail(h, BD_ERR_FORMAT, "BAM record shorter than its fixed part");
        const uint8_t* p = h->buf + h->pos + 4;
        const int32_t ref = rd_i32(p), ps = rd_i32(p + 4);
        const uint32_t l_name = p[8];
        const uint16_t fl = (uint16_t)(p[14] | (p[15] << 8));
        const int32_t nref = rd_i32(p + 20), npos = rd_i32(p + 24);
        if (l_name < 1 || 32 + l_name > (uint32_t)bs) return fail(h, BD_ERR_FORMAT, "BAM record: name runs past the record");
        flag[n] = fl; tid[n] = ref; pos[n] = ps; next_tid[n] = nref; next_pos[n] = npos;
        int ok = !(fl & 0x4);
        if (ok && (fl & 0x1)) ok = !((fl & 0x80) || !(fl & 0x2) || ref != nref || npos < 0);
        uint8_t hd = 0;
        if (ok) {
            const uint8_t* name = p + 32;
            int32_t len = (int32_t)l_name - 1;                     /* without the NUL */
            if (trim) {
                const uint8_t* sp = (const uint8_t*)memchr(name, ' ', (size_t)len);
                if (sp && sp > name) len = (int32_t)(sp - name);
            }
            if (!h->has_last || len != h->last_len || memcmp(name, h->last, (size_t)len) != 0) {
                hd = 1;
                if ((size_t)len > h->last_cap) {
                    uint8_t* nl = (uint8_t*)realloc(h->last, (size_t)len + 64);
                    if (!nl) return fail(h, BD_ERR_MEM, "out of memory");
                    h->last = nl; h->last_cap = (size_t)len + 64;
                }
                memcpy(h->last, name, (size_t)len); h->last_len = len; h->has_last = 1;
            }
        }
        valid[n] = (uint8_t)ok; head[n] = hd;
        h->pos += 4 + (size_t)bs;
        ++n;
    }
    *n_out = n;
    return BD_OK;
}

/* The record tuples of include/ecb.h straight from the file (what tuples.TupleEncoder.encode_decoded makes of bd_read's arrays):
 * read_id = running count of read heads (*cur carries it across calls; 0xFFFFFFFF = no read yet, which is also what records
 * before the first head get), locus / haplotype looked up from the reference id of VALID records (others: reference 0),
 * hapflag = the 12 BAM flag bits | ECB_FLAG_MATE_OTHER_REF | ECB_FLAG_NEXT_POS_NEG | haplotype << 16, pos as it is. */
#define TUPLE_CHUNK 65536
int bd_read_tuples(bd_handle* h, size_t max_records, int trim, const uint32_t* tid2locus, const uint32_t* tid2hap, int32_t n_ref,
                   uint32_t* cur, uint32_t* read_id, uint32_t* locus, uint32_t* hapflag, int32_t* pos, size_t* n_out, size_t* n_valid) {
    if (!h || !cur || !read_id || !locus || !hapflag || !pos || !n_out || !n_valid || n_ref < 0 || (n_ref && (!tid2locus || !tid2hap))) return BD_ERR_ARG;
    const size_t per = sizeof(uint16_t) + 3 * sizeof(int32_t) + 2;           /* flag, tid, next_tid, next_pos, valid, head */
    if (h->scratch_cap < TUPLE_CHUNK * per) {
        void* ns = realloc(h->scratch, TUPLE_CHUNK * per);
        if (!ns) return fail(h, BD_ERR_MEM, "out of memory");
        h->scratch = ns; h->scratch_cap = TUPLE_CHUNK * per;
    }
    int32_t* tid = (int32_t*)h->scratch;
    int32_t* ntid = tid + TUPLE_CHUNK;
    int32_t* npos = ntid + TUPLE_CHUNK;
    uint16_t* flag = (uint16_t*)(npos + TUPLE_CHUNK);
    uint8_t* valid = (uint8_t*)(flag + TUPLE_CHUNK);
    uint8_t* head = valid + TUPLE_CHUNK;
    size_t n = 0, nv = 0;
    uint32_t c = *cur;
    while (n < max_records) {
        const size_t want = max_records - n < TUPLE_CHUNK ? max_records - n : TUPLE_CHUNK;
        size_t got = 0;
        const int rc = bd_read(h, want, trim, flag, tid, pos + n, ntid, npos, valid, head, &got);
        if (rc != BD_OK) return rc;
        if (got == 0) break;
        for (size_t i = 0; i < got; ++i) {
            c += head[i];                                                    /* (0xFFFFFFFF + 1 = 0: the first read) */
            uint32_t l = 0, hp = 0;
            if (n_ref) {
                const int32_t t = valid[i] ? tid[i] : 0;
                if (t < 0 || t >= n_ref) return fail(h, BD_ERR_FORMAT, "a mapped record names a reference the header does not have");
                l = tid2locus[t]; hp = tid2hap[t];
            } else if (valid[i]) return fail(h, BD_ERR_FORMAT, "a mapped record in a file without references");
            read_id[n + i] = c; locus[n + i] = l;
            hapflag[n + i] = (flag[i] & 0xFFFu) | (tid[i] != ntid[i] ? 0x1000u : 0u) | (npos[i] < 0 ? 0x2000u : 0u) | (hp << 16);
            nv += valid[i];
        }
        n += got;
        if (got < want) break;
    }
    *cur = c; *n_out = n; *n_valid = nv;
    return BD_OK;
}

/* ---- multisample scan (alntools/bam_utils_multisample.py:257-300) -------------------------------------------------- */
/* field 14 of the name split at every "|||" (the cell barcode, :270-280); -1 if there are fewer than 15 fields */
static int32_t cell_field(const uint8_t* s, int32_t n, int32_t* start) {
    int32_t pos = 0, field = 0, f0 = 0;
    while (field < 14) {
        int32_t k = pos;
        for (; k + 3 <= n; ++k) if (s[k] == '|' && s[k + 1] == '|' && s[k + 2] == '|') break;
        if (k + 3 > n) return -1;
        pos = k + 3; ++field; f0 = pos;
    }
    int32_t k = pos;
    for (; k + 3 <= n; ++k) if (s[k] == '|' && s[k + 1] == '|' && s[k + 2] == '|') break;
    *start = f0;
    return (k + 3 <= n ? k : n) - f0;
}

/* As bd_read, with the multisample path's run rule instead of `head`: newrun[i] = 1 if the record is valid and starts a run.
 * The tracked name starts as the first valid record's name cut at its first space (if that is not its first character); a
 * record whose name -- cut the same way -- differs from the tracked name starts a new run, and the tracked name becomes its
 * WHOLE name (:288-292: from then on a name with a space in it never equals what follows it).  The cell id of every run
 * started is field 14 of the tracked name split at "|||"; bd_ms_cells returns them.  BD_ERR_FORMAT if a run's name has no such
 * field (the reference raises there). */
int bd_read_ms(bd_handle* h, size_t max_records, uint16_t* flag, int32_t* tid, int32_t* pos, int32_t* next_tid, int32_t* next_pos,
               uint8_t* valid, uint8_t* newrun, size_t* n_out) {
    if (!h || !flag || !tid || !pos || !next_tid || !next_pos || !valid || !newrun || !n_out) return BD_ERR_ARG;
    size_t n = 0;
    h->cells_len = 0; h->n_cells = 0;
    while (n < max_records) {
        int rc = need(h, 4);
        if (rc == 1) { if (h->len != h->pos) return fail(h, BD_ERR_FORMAT, "truncated BAM"); break; }
        if (rc < 0) return rc;
        const int32_t bs = rd_i32(h->buf + h->pos);
        if (bs < 32) return fail(h, BD_ERR_FORMAT, "BAM record shorter than its fixed part");
        rc = need(h, 4 + (size_t)bs);
        if (rc != 0) return rc < 0 ? rc : fail(h, BD_ERR_FORMAT, "truncated BAM record");
        const uint8_t* p = h->buf + h->pos + 4;
        const int32_t ref = rd_i32(p), ps = rd_i32(p + 4);
        const uint32_t l_name = p[8];
        const uint16_t fl = (uint16_t)(p[14] | (p[15] << 8));
        const int32_t nref = rd_i32(p + 20), npos = rd_i32(p + 24);
        if (l_name < 1 || 32 + l_name > (uint32_t)bs) return fail(h, BD_ERR_FORMAT, "BAM record: name runs past the record");
        flag[n] = fl; tid[n] = ref; pos[n] = ps; next_tid[n] = nref; next_pos[n] = npos;
        int ok = !(fl & 0x4);
        if (ok && (fl & 0x1)) ok = !((fl & 0x80) || !(fl & 0x2) || ref != nref || npos < 0);
        uint8_t nr = 0;
        if (ok) {
            const uint8_t* name = p + 32;
            const int32_t full = (int32_t)l_name - 1;
            int32_t cut = full;
            const uint8_t* sp = (const uint8_t*)memchr(name, ' ', (size_t)full);
            if (sp && sp > name) cut = (int32_t)(sp - name);
            int32_t keep = -1;                                         /* length of the name to track from here on, if it changes */
            if (!h->has_last) { keep = cut; nr = 1; }                  /* :257-262: the first tracked name is the cut one */
            else if (cut != h->last_len || memcmp(name, h->last, (size_t)cut) != 0) { keep = full; nr = 1; }   /* :288-292: ... later ones are not */
            if (nr) {
                if ((size_t)keep > h->last_cap) {
                    uint8_t* nl = (uint8_t*)realloc(h->last, (size_t)keep + 64);
                    if (!nl) return fail(h, BD_ERR_MEM, "out of memory");
                    h->last = nl; h->last_cap = (size_t)keep + 64;
                }
                memcpy(h->last, name, (size_t)keep); h->last_len = keep; h->has_last = 1;
                int32_t c0 = 0;
                const int32_t cl = cell_field(h->last, h->last_len, &c0);
                if (cl < 0) return fail(h, BD_ERR_FORMAT, "a read name has no cell id in '|||' field 14 (bam_utils_multisample.py:270-280)");
                if (h->cells_len + (size_t)cl > h->cells_cap) {
                    const size_t nc = (h->cells_len + (size_t)cl) * 2 + 4096;
                    uint8_t* nb = (uint8_t*)realloc(h->cells, nc);
                    if (!nb) return fail(h, BD_ERR_MEM, "out of memory");
                    h->cells = nb; h->cells_cap = nc;
                }
                if (h->n_cells + 2 > h->cell_off_cap) {
                    const size_t nc = (h->n_cells + 2) * 2 + 1024;
                    uint32_t* no = (uint32_t*)realloc(h->cell_off, nc * sizeof(uint32_t));
                    if (!no) return fail(h, BD_ERR_MEM, "out of memory");
                    h->cell_off = no; h->cell_off_cap = nc;
                }
                h->cell_off[h->n_cells++] = (uint32_t)h->cells_len;
                memcpy(h->cells + h->cells_len, h->last + c0, (size_t)cl);
                h->cells_len += (size_t)cl;
                h->cell_off[h->n_cells] = (uint32_t)h->cells_len;
            }
        }
        valid[n] = (uint8_t)ok; newrun[n] = nr;
        h->pos += 4 + (size_t)bs;
        ++n;
    }
    *n_out = n;
    return BD_OK;
}

/* the cell ids of the runs the last bd_read_ms started: n of them, cell k = bytes [off[k], off[k + 1]) of *bytes (valid until the next call) */
int bd_ms_cells(const bd_handle* h, const uint8_t** bytes, const uint32_t** off, size_t* n) {
    if (!h || !bytes || !off || !n) return BD_ERR_ARG;
    *bytes = h->cells; *off = h->cell_off; *n = h->n_cells;
    return BD_OK;
}

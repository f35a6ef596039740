/*
 * ORACLE -- test infrastructure, not product code.
 *
 * Plain-C restatement of alntools' bam2ec hot path on the decoded tuple stream
 * (include/ecb.h layout), used (a) as the checker for GPU parity tests at sizes the
 * Python restatement (oracle/ec_oracle.py) is too slow for and (b) as bench.py's
 * cpu_baseline ("port").  It is validated against the Python restatement, which is
 * itself pinned byte-for-byte to fixtures produced by the reference
 * (tests/test_oracle_golden.py, tests/test_c_oracle.py).
 *
 * Follows, in the reference (/root/reference/alntools):
 *   record filter                         bam_utils.py:264-270
 *   read = run of equal names (valid)     bam_utils.py:289-320   (host run counter read_id)
 *   per-read distinct targets, canonical  bam_utils.py:307,322-325  (sorted list as the key)
 *   ordered EC dict, count += 1           bam_utils.py:217,309-312,341-344
 *   worker pool over contiguous shards    bam_utils.py:646-680   (threads here)
 *   ordered merge, rank = first seen      bam_utils.py:680-698
 *   incidence rows, haplotype bitmask     bam_utils.py:788-837, bin_utils.py:208-211
 *
 * Build: make -C oracle   ->  oracle/_build/libec_oracle.so
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t *keys;      /* key arena: concatenated sorted slot lists */
    uint64_t keys_len, keys_cap;
    uint64_t *off;       /* per EC: offset into keys */
    uint32_t *len;       /* per EC: key length */
    uint64_t *count;     /* per EC */
    uint64_t n, cap;     /* ECs, in first-seen order */
    uint64_t *tab;       /* open addressing: EC index + 1, 0 = empty */
    uint64_t tab_cap;    /* power of two */
    uint64_t n_all, n_valid, n_reads;
} ecmap;

static uint64_t hash_key(const uint32_t *k, uint32_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
    for (uint32_t i = 0; i < n; ++i) {
        h ^= k[i];
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
    }
    return h;
}

static void map_init(ecmap *m) {
    memset(m, 0, sizeof *m);
    m->tab_cap = 1 << 12;
    m->tab = calloc(m->tab_cap, sizeof(uint64_t));
    m->cap = 1 << 10;
    m->off = malloc(m->cap * sizeof(uint64_t));
    m->len = malloc(m->cap * sizeof(uint32_t));
    m->count = malloc(m->cap * sizeof(uint64_t));
    m->keys_cap = 1 << 14;
    m->keys = malloc(m->keys_cap * sizeof(uint32_t));
}

static void map_free(ecmap *m) {
    free(m->tab); free(m->off); free(m->len); free(m->count); free(m->keys);
}

static void map_rehash(ecmap *m) {
    uint64_t nc = m->tab_cap * 2;
    uint64_t *t = calloc(nc, sizeof(uint64_t));
    for (uint64_t e = 0; e < m->n; ++e) {
        uint64_t j = hash_key(m->keys + m->off[e], m->len[e]) & (nc - 1);
        while (t[j]) j = (j + 1) & (nc - 1);
        t[j] = e + 1;
    }
    free(m->tab);
    m->tab = t;
    m->tab_cap = nc;
}

/* ec[key] += add, appending the key if new (OrderedDict semantics) */
static void map_add(ecmap *m, const uint32_t *k, uint32_t n, uint64_t add) {
    uint64_t j = hash_key(k, n) & (m->tab_cap - 1);
    for (;; j = (j + 1) & (m->tab_cap - 1)) {
        uint64_t e = m->tab[j];
        if (!e) break;
        --e;
        if (m->len[e] == n && !memcmp(m->keys + m->off[e], k, n * sizeof(uint32_t))) {
            m->count[e] += add;
            return;
        }
    }
    if (m->n == m->cap) {
        m->cap *= 2;
        m->off = realloc(m->off, m->cap * sizeof(uint64_t));
        m->len = realloc(m->len, m->cap * sizeof(uint32_t));
        m->count = realloc(m->count, m->cap * sizeof(uint64_t));
    }
    while (m->keys_len + n > m->keys_cap) {
        m->keys_cap *= 2;
        m->keys = realloc(m->keys, m->keys_cap * sizeof(uint32_t));
    }
    memcpy(m->keys + m->keys_len, k, n * sizeof(uint32_t));
    m->off[m->n] = m->keys_len;
    m->len[m->n] = n;
    m->count[m->n] = add;
    m->keys_len += n;
    m->tab[j] = ++m->n;
    if (m->n * 2 > m->tab_cap) map_rehash(m);
}

static int rec_valid(uint32_t hf) {
    if (hf & 0x4u) return 0;
    if (hf & 0x1u)
        if ((hf & 0x80u) || !(hf & 0x2u) || (hf & 0x3000u)) return 0;
    return 1;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

typedef struct {
    const uint32_t *rid, *loc, *hf;
    uint64_t lo, hi;     /* record range; lo is the start of a read run, hi its end */
    uint32_t n_haps;
    ecmap map;
} shard;

/* the per-alignment loop of process_convert_bam over one contiguous shard */
static void *scan_shard(void *arg) {
    shard *s = arg;
    ecmap *m = &s->map;
    map_init(m);
    uint32_t *cur = malloc(64 * sizeof(uint32_t));
    uint32_t cap = 64, n = 0, cur_rid = 0;
    int open = 0;
    for (uint64_t i = s->lo; i < s->hi; ++i) {
        m->n_all++;
        if (!rec_valid(s->hf[i])) continue;
        m->n_valid++;
        uint32_t slot = s->loc[i] * s->n_haps + ((s->hf[i] >> 16) & 0xFFu);
        if (!open || s->rid[i] != cur_rid) {
            if (open) {
                qsort(cur, n, sizeof(uint32_t), cmp_u32);
                map_add(m, cur, n, 1);
            }
            open = 1; cur_rid = s->rid[i]; n = 0; m->n_reads++;
        }
        int dup = 0;                                    /* "if reference_id not in reference_ids" */
        for (uint32_t k = 0; k < n; ++k) if (cur[k] == slot) { dup = 1; break; }
        if (!dup) {
            if (n == cap) { cap *= 2; cur = realloc(cur, cap * sizeof(uint32_t)); }
            cur[n++] = slot;
        }
    }
    if (open) {
        qsort(cur, n, sizeof(uint32_t), cmp_u32);
        map_add(m, cur, n, 1);
    }
    free(cur);
    return 0;
}

typedef struct {
    uint64_t n_ecs, nnz, n_all, n_valid, n_reads;
    int32_t *indptr, *indices, *data, *count;
} ec_result;

/*
 * threads contiguous shards (cut at read boundaries) -> ordered merge -> CSR.
 * Returns 0, or -1 if there is no valid alignment.  Free with ec_oracle_free.
 */
int ec_oracle_run(const uint32_t *rid, const uint32_t *loc, const uint32_t *hf, uint64_t n,
                  uint32_t n_haps, int threads, ec_result *out) {
    if (threads < 1) threads = 1;
    shard *sh = calloc(threads, sizeof(shard));
    pthread_t *th = calloc(threads, sizeof(pthread_t));
    uint64_t prev = 0;
    for (int t = 0; t < threads; ++t) {
        uint64_t cut = (t == threads - 1) ? n : (n / threads) * (uint64_t)(t + 1);
        if (cut < prev) cut = prev;
        while (cut < n && cut > 0 && rid[cut] == rid[cut - 1]) ++cut;   /* never split a read */
        sh[t].rid = rid; sh[t].loc = loc; sh[t].hf = hf; sh[t].n_haps = n_haps;
        sh[t].lo = prev; sh[t].hi = cut;
        prev = cut;
    }
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], 0, scan_shard, &sh[t]);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], 0);
    /* ordered merge: bam_utils.py:680-698 */
    ecmap *fin = &sh[0].map;
    for (int t = 1; t < threads; ++t) {
        ecmap *m = &sh[t].map;
        for (uint64_t e = 0; e < m->n; ++e) map_add(fin, m->keys + m->off[e], m->len[e], m->count[e]);
        fin->n_all += m->n_all; fin->n_valid += m->n_valid; fin->n_reads += m->n_reads;
        map_free(m);
    }
    memset(out, 0, sizeof *out);
    out->n_all = fin->n_all; out->n_valid = fin->n_valid; out->n_reads = fin->n_reads;
    if (fin->n == 0) { map_free(fin); free(sh); free(th); return -1; }
    /* rows of A: group each (sorted) key by locus, OR the haplotype bits */
    uint64_t nnz = 0;
    for (uint64_t e = 0; e < fin->n; ++e) {
        const uint32_t *k = fin->keys + fin->off[e];
        for (uint32_t i = 0; i < fin->len[e]; ++i)
            if (i == 0 || k[i] / n_haps != k[i - 1] / n_haps) ++nnz;
    }
    out->n_ecs = fin->n; out->nnz = nnz;
    out->indptr = malloc((fin->n + 1) * sizeof(int32_t));
    out->indices = malloc((nnz ? nnz : 1) * sizeof(int32_t));
    out->data = malloc((nnz ? nnz : 1) * sizeof(int32_t));
    out->count = malloc(fin->n * sizeof(int32_t));
    uint64_t w = 0;
    for (uint64_t e = 0; e < fin->n; ++e) {
        const uint32_t *k = fin->keys + fin->off[e];
        out->indptr[e] = (int32_t)w;
        for (uint32_t i = 0; i < fin->len[e]; ++i) {
            uint32_t l = k[i] / n_haps, b = 1u << (k[i] % n_haps);
            if (i && l == k[i - 1] / n_haps) out->data[w - 1] |= (int32_t)b;
            else { out->indices[w] = (int32_t)l; out->data[w] = (int32_t)b; ++w; }
        }
        out->count[e] = (int32_t)fin->count[e];
    }
    out->indptr[fin->n] = (int32_t)w;
    map_free(fin);
    free(sh); free(th);
    return 0;
}

void ec_oracle_free(ec_result *r) {
    free(r->indptr); free(r->indices); free(r->data); free(r->count);
    memset(r, 0, sizeof *r);
}

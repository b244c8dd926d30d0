/*
 * kdf_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the canonical k-mer count / filter / probe path that
 * the reference (jlanej/kmer_denovo_filter) runs through
 *   samtools fasta | jellyfish count -C [--if] ; jellyfish dump -c -L ; jellyfish query
 * and through the Python sliding-window loop.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped package
 * (kmer_denovo_filter_amd/) never imports it.
 *
 * Parity status: PINNED.  Checked in tests/test_oracle_golden.py against
 *   - the real Jellyfish `binary/sorted` file tests/data/giab/mini_ref.fa.k31.jf
 *     (45 275 records, bit-equal (key,count) multiset),
 *   - the discovery goldens 51125 -> 6679 -> 630 and 195/11 informative reads
 *     (tests/example_output_discovery/giab_discovery.metrics.json:3-7),
 *   - the canonicalize KATs of tests/test_kmer_utils.py:32-44.
 *
 * Rules restated (reference file:line, relative to /root/reference):
 *   canonical form      src/kmer_denovo_filter/kmer_utils.py:15,35-38
 *                       (lexicographic min of k-mer and reverse complement ==
 *                       numeric min in the 2-bit code A=0 C=1 G=2 T=3, leftmost
 *                       base most significant -- the Jellyfish key encoding,
 *                       SURVEY.md section 0.5)
 *   window rule         src/kmer_denovo_filter/kmer_utils.py:91-121 (upper-case,
 *                       windows containing N skipped) and Jellyfish `count`
 *                       (any non-ACGT byte breaks the window; windows never
 *                       span records): core/jellyfish_wrappers.py:167-176,
 *                       discovery/pipeline.py:114-122
 *   count --if          core/jellyfish_wrappers.py:167-176, discovery/pipeline.py:377-386
 *                       (only k-mers present in the filter are counted; a filter
 *                       k-mer never seen reports 0 on query)
 *   dump -c -L n        discovery/pipeline.py:207-211 (count >= n, inclusive)
 *   query               discovery/pipeline.py:286-304 (one count per input
 *                       k-mer, input order, 0 if absent)
 *   scan                core/bam_scanner.py:433-443 (per read: hit positions and
 *                       number of DISTINCT canonical k-mers hit)
 *
 * Keys are 2k-bit integers held in unsigned __int128 (k <= 64); the API moves
 * them as (lo, hi) uint64 pairs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ map -- */

typedef struct {
    u128 *keys;
    uint32_t *vals;
    uint8_t *used;
    uint64_t cap;   /* power of two */
    uint64_t n;
} kmap;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33; return x;
}
static inline uint64_t hkey(u128 k) {
    return mix64((uint64_t)k ^ mix64((uint64_t)(k >> 64) + 0x9e3779b97f4a7c15ULL));
}

static int kmap_init(kmap *m, uint64_t cap_hint) {
    uint64_t cap = 1024;
    while (cap < cap_hint * 2) cap <<= 1;
    m->cap = cap; m->n = 0;
    m->keys = (u128 *)malloc(cap * sizeof(u128));
    m->vals = (uint32_t *)calloc(cap, sizeof(uint32_t));
    m->used = (uint8_t *)calloc(cap, 1);
    return (m->keys && m->vals && m->used) ? 0 : -1;
}
static void kmap_free(kmap *m) { free(m->keys); free(m->vals); free(m->used); }

static int kmap_grow(kmap *m);

/* returns slot index; inserts key with value 0 when absent and insert != 0;
 * returns UINT64_MAX when absent and insert == 0 */
static inline uint64_t kmap_find(kmap *m, u128 key, int insert) {
    if (insert && (m->n + 1) * 10 > m->cap * 7) kmap_grow(m);
    uint64_t mask = m->cap - 1, h = hkey(key) & mask;
    for (;;) {
        if (!m->used[h]) {
            if (!insert) return UINT64_MAX;
            m->used[h] = 1; m->keys[h] = key; m->vals[h] = 0; m->n++;
            return h;
        }
        if (m->keys[h] == key) return h;
        h = (h + 1) & mask;
    }
}
static int kmap_grow(kmap *m) {
    kmap b;
    if (kmap_init(&b, m->cap) != 0) return -1;   /* cap*2 */
    for (uint64_t i = 0; i < m->cap; i++)
        if (m->used[i]) { uint64_t s = kmap_find(&b, m->keys[i], 1); b.vals[s] = m->vals[i]; }
    kmap_free(m); *m = b; return 0;
}

/* ------------------------------------------------------------- encoding -- */

/* A=0 C=1 G=2 T=3 (case-insensitive); anything else = 4 (breaks the window). */
static const uint8_t *code_table(void) {
    static uint8_t t[256]; static int init = 0;
    if (!init) {
        memset(t, 4, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
        init = 1;
    }
    return t;
}

static inline u128 kmask(int k) { return (k >= 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }

/* Calls cb(canonical_key, window_start, ctx) for every valid window of seq.
 * kmer_utils.py:112-118: window i is seq[i:i+k]; skipped when it holds a
 * non-ACGT byte; canonical = min(fwd, revcomp). */
typedef void (*win_cb)(u128 canon, int64_t pos, void *ctx);

static void for_each_window(const char *seq, int64_t len, int k, win_cb cb, void *ctx) {
    const uint8_t *T = code_table();
    u128 mask = kmask(k), fwd = 0, rc = 0;
    int valid = 0, sh = 2 * (k - 1);
    for (int64_t i = 0; i < len; i++) {
        uint8_t c = T[(uint8_t)seq[i]];
        if (c > 3) { valid = 0; fwd = 0; rc = 0; continue; }
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((u128)(3 - c) << sh);
        if (++valid >= k) cb(fwd < rc ? fwd : rc, i - k + 1, ctx);
    }
}

/* ------------------------------------------------------------ the API ---- */

typedef struct { kmap m; int k; } kdfo_table;

void *kdfo_create(int k, uint64_t cap_hint) {
    if (k < 1 || k > 64) return NULL;
    kdfo_table *t = (kdfo_table *)malloc(sizeof *t);
    if (!t) return NULL;
    t->k = k;
    if (kmap_init(&t->m, cap_hint) != 0) { free(t); return NULL; }
    return t;
}
void kdfo_destroy(void *h) { if (h) { kmap_free(&((kdfo_table *)h)->m); free(h); } }
uint64_t kdfo_size(void *h) { return ((kdfo_table *)h)->m.n; }

/* canonical key of an ASCII k-mer; returns -1 when it holds a non-ACGT byte */
int kdfo_canonical_ascii(const char *kmer, int k, uint64_t *lo, uint64_t *hi) {
    const uint8_t *T = code_table();
    u128 fwd = 0, rc = 0;
    for (int i = 0; i < k; i++) {
        uint8_t c = T[(uint8_t)kmer[i]];
        if (c > 3) return -1;
        fwd = (fwd << 2) | c;
        rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));
    }
    u128 c = fwd < rc ? fwd : rc;
    *lo = (uint64_t)c; *hi = (uint64_t)(c >> 64);
    return 0;
}

static void cb_insert(u128 c, int64_t pos, void *ctx) {
    (void)pos; kmap *m = (kmap *)ctx;
    uint64_t s = kmap_find(m, c, 1);
    if (m->vals[s] != UINT32_MAX) m->vals[s]++;      /* 4-byte saturating counter */
}
static void cb_count_if(u128 c, int64_t pos, void *ctx) {
    (void)pos; kmap *m = (kmap *)ctx;
    uint64_t s = kmap_find(m, c, 0);
    if (s != UINT64_MAX && m->vals[s] != UINT32_MAX) m->vals[s]++;
}

/* jellyfish count -C : every valid window of every record is counted.
 * buf = concatenated ASCII records, offs[n+1] = record boundaries. */
int kdfo_count_reads(void *h, const char *buf, const int64_t *offs, int64_t n_reads) {
    kdfo_table *t = (kdfo_table *)h;
    for (int64_t r = 0; r < n_reads; r++)
        for_each_window(buf + offs[r], offs[r + 1] - offs[r], t->k, cb_insert, &t->m);
    return 0;
}

/* load the --if filter: keys present with count 0 */
int kdfo_load_filter(void *h, const uint64_t *lo, const uint64_t *hi, int64_t n) {
    kdfo_table *t = (kdfo_table *)h;
    for (int64_t i = 0; i < n; i++) {
        u128 key = ((u128)(hi ? hi[i] : 0) << 64) | lo[i];
        kmap_find(&t->m, key, 1);
    }
    return 0;
}

/* jellyfish count -C --if : only windows whose canonical k-mer is in the table count */
int kdfo_count_reads_filtered(void *h, const char *buf, const int64_t *offs, int64_t n_reads) {
    kdfo_table *t = (kdfo_table *)h;
    for (int64_t r = 0; r < n_reads; r++)
        for_each_window(buf + offs[r], offs[r + 1] - offs[r], t->k, cb_count_if, &t->m);
    return 0;
}

/* jellyfish query: counts in input order, 0 if absent */
int kdfo_query(void *h, const uint64_t *lo, const uint64_t *hi, int64_t n, uint32_t *out) {
    kdfo_table *t = (kdfo_table *)h;
    for (int64_t i = 0; i < n; i++) {
        u128 key = ((u128)(hi ? hi[i] : 0) << 64) | lo[i];
        uint64_t s = kmap_find(&t->m, key, 0);
        out[i] = (s == UINT64_MAX) ? 0 : t->m.vals[s];
    }
    return 0;
}

typedef struct { u128 k; uint32_t v; } kv;
static int kv_cmp(const void *a, const void *b) {
    u128 x = ((const kv *)a)->k, y = ((const kv *)b)->k;
    return (x > y) - (x < y);
}

/* jellyfish dump -c -L min_count, emitted in ascending key order (the
 * reference does not rely on Jellyfish's hash order).  Two-call protocol:
 * pass lo == NULL to get the number of records. */
int64_t kdfo_export_ge(void *h, uint32_t min_count, uint64_t *lo, uint64_t *hi, uint32_t *counts) {
    kdfo_table *t = (kdfo_table *)h;
    int64_t n = 0;
    for (uint64_t i = 0; i < t->m.cap; i++)
        if (t->m.used[i] && t->m.vals[i] >= min_count) n++;
    if (!lo) return n;
    kv *a = (kv *)malloc((n ? n : 1) * sizeof(kv));
    if (!a) return -1;
    int64_t j = 0;
    for (uint64_t i = 0; i < t->m.cap; i++)
        if (t->m.used[i] && t->m.vals[i] >= min_count) { a[j].k = t->m.keys[i]; a[j].v = t->m.vals[i]; j++; }
    qsort(a, n, sizeof(kv), kv_cmp);
    for (j = 0; j < n; j++) {
        lo[j] = (uint64_t)a[j].k; if (hi) hi[j] = (uint64_t)(a[j].k >> 64);
        if (counts) counts[j] = a[j].v;
    }
    free(a);
    return n;
}

/* Module-3 scan (bam_scanner.py:433-443): for every read, hit_bits gets one
 * bit per window start (bit (offs[r]+i) of a bitmap over the concatenated
 * buffer) when the canonical k-mer is in the table with count > 0, and
 * distinct[r] = number of DISTINCT canonical k-mers hit in that read. */
typedef struct { kmap *m; uint8_t *bits; int64_t base; kmap seen; uint32_t distinct; } scan_ctx;
static void cb_scan(u128 c, int64_t pos, void *ctx) {
    scan_ctx *s = (scan_ctx *)ctx;
    uint64_t slot = kmap_find(s->m, c, 0);
    if (slot == UINT64_MAX || s->m->vals[slot] == 0) return;
    int64_t b = s->base + pos;
    s->bits[b >> 3] |= (uint8_t)(1u << (b & 7));
    uint64_t n0 = s->seen.n;
    kmap_find(&s->seen, c, 1);
    if (s->seen.n != n0) s->distinct++;
}
int kdfo_scan_reads(void *h, const char *buf, const int64_t *offs, int64_t n_reads,
                    uint8_t *hit_bits, uint32_t *distinct) {
    kdfo_table *t = (kdfo_table *)h;
    for (int64_t r = 0; r < n_reads; r++) {
        scan_ctx s; s.m = &t->m; s.bits = hit_bits; s.base = offs[r]; s.distinct = 0;
        if (kmap_init(&s.seen, 64) != 0) return -1;
        for_each_window(buf + offs[r], offs[r + 1] - offs[r], t->k, cb_scan, &s);
        distinct[r] = s.distinct;
        kmap_free(&s.seen);
    }
    return 0;
}

/* number of valid windows (the bench's unit of work) */
static void cb_n(u128 c, int64_t pos, void *ctx) { (void)c; (void)pos; (*(int64_t *)ctx)++; }
int64_t kdfo_count_windows(const char *buf, const int64_t *offs, int64_t n_reads, int k) {
    int64_t n = 0;
    for (int64_t r = 0; r < n_reads; r++)
        for_each_window(buf + offs[r], offs[r + 1] - offs[r], k, cb_n, &n);
    return n;
}

/* ---------------------------------------------------- threaded counting -- */
/* CPU baseline leg for bench.py: T threads, each owns the keys whose hash
 * falls in its partition (hash % T), so no locks; every thread walks the whole
 * input (the rolling window is cheap next to the hash-map miss).  Same
 * semantics as kdfo_count_reads; result merged into the handle. */
typedef struct { const char *buf; const int64_t *offs; int64_t n; int k; int tid, T; kmap m; } part_job;
typedef struct { kmap *m; int tid, T; } part_ctx;
static void cb_part(u128 c, int64_t pos, void *ctx) {
    (void)pos; part_ctx *p = (part_ctx *)ctx;
    if ((int)((hkey(c) >> 40) % (uint64_t)p->T) != p->tid) return;
    uint64_t s = kmap_find(p->m, c, 1);
    if (p->m->vals[s] != UINT32_MAX) p->m->vals[s]++;
}
static void *part_run(void *arg) {
    part_job *j = (part_job *)arg;
    part_ctx p = { &j->m, j->tid, j->T };
    for (int64_t r = 0; r < j->n; r++)
        for_each_window(j->buf + j->offs[r], j->offs[r + 1] - j->offs[r], j->k, cb_part, &p);
    return NULL;
}
int kdfo_count_reads_mt(void *h, const char *buf, const int64_t *offs, int64_t n_reads, int threads) {
    kdfo_table *t = (kdfo_table *)h;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    part_job *jobs = (part_job *)calloc(threads, sizeof(part_job));
    pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
    for (int i = 0; i < threads; i++) {
        jobs[i].buf = buf; jobs[i].offs = offs; jobs[i].n = n_reads; jobs[i].k = t->k;
        jobs[i].tid = i; jobs[i].T = threads;
        kmap_init(&jobs[i].m, 1 << 16);
        pthread_create(&th[i], NULL, part_run, &jobs[i]);
    }
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    for (int i = 0; i < threads; i++) {
        kmap *m = &jobs[i].m;
        for (uint64_t s = 0; s < m->cap; s++)
            if (m->used[s]) {
                uint64_t d = kmap_find(&t->m, m->keys[s], 1);
                uint64_t v = (uint64_t)t->m.vals[d] + m->vals[s];
                t->m.vals[d] = v > UINT32_MAX ? UINT32_MAX : (uint32_t)v;
            }
        kmap_free(m);
    }
    free(jobs); free(th);
    return 0;
}

/* CPU baseline leg proper (bench.py): the same count as kdfo_count_reads followed by the
 * `dump -L min_count` tally, organised the way a multi-core counter would be: phase 1, every
 * thread takes a contiguous share of the reads and deals its canonical k-mers into one buffer
 * per key partition (hash % T); phase 2, thread p counts partition p in a map of its own.  No
 * locks, every window is extracted once, nothing is merged: out[0] = distinct k-mers, out[1] =
 * sum of the counts (= valid windows), out[2] = k-mers with count >= min_count.
 * tests/test_oracle_golden.py holds it to kdfo_count_reads + kdfo_export_ge. */
typedef struct { u128 *v; int64_t n, cap; } kbuf;
typedef struct {
    const char *buf; const int64_t *offs; int64_t r0, r1; int k, tid, T;
    kbuf *out;                 /* phase 1: T buffers of this thread */
    kbuf **all;                /* phase 2: all[t] = thread t's buffers */
    uint32_t min_count; uint64_t res[3]; int err;
} deal_job;
static void cb_deal(u128 c, int64_t pos, void *ctx) {
    (void)pos; deal_job *j = (deal_job *)ctx;
    kbuf *b = &j->out[(hkey(c) >> 40) % (uint64_t)j->T];
    if (b->n == b->cap) {
        int64_t nc = b->cap ? b->cap * 2 : 4096;
        u128 *nv = (u128 *)realloc(b->v, (size_t)nc * sizeof(u128));
        if (!nv) { j->err = 1; return; }
        b->v = nv; b->cap = nc;
    }
    b->v[b->n++] = c;
}
static void *deal_run(void *arg) {
    deal_job *j = (deal_job *)arg;
    for (int64_t r = j->r0; r < j->r1 && !j->err; r++)
        for_each_window(j->buf + j->offs[r], j->offs[r + 1] - j->offs[r], j->k, cb_deal, j);
    return NULL;
}
static void *tally_run(void *arg) {
    deal_job *j = (deal_job *)arg;
    int64_t total = 0;
    for (int t = 0; t < j->T; t++) total += j->all[t][j->tid].n;
    kmap m;
    if (kmap_init(&m, (uint64_t)(total / 2 + 16)) != 0) { j->err = 1; return NULL; }
    for (int t = 0; t < j->T; t++) {
        const kbuf *b = &j->all[t][j->tid];
        for (int64_t i = 0; i < b->n; i++) {
            uint64_t s = kmap_find(&m, b->v[i], 1);
            if (m.vals[s] != UINT32_MAX) m.vals[s]++;
        }
    }
    j->res[0] = m.n; j->res[1] = 0; j->res[2] = 0;
    for (uint64_t s = 0; s < m.cap; s++)
        if (m.used[s]) { j->res[1] += m.vals[s]; if (m.vals[s] >= j->min_count) j->res[2]++; }
    kmap_free(&m);
    return NULL;
}
int kdfo_count_tally_mt(const char *buf, const int64_t *offs, int64_t n_reads, int k, int threads,
                        uint32_t min_count, uint64_t *out) {
    if (k < 1 || k > 64) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    const int T = threads;
    deal_job *jobs = (deal_job *)calloc(T, sizeof(deal_job));
    pthread_t *th = (pthread_t *)calloc(T, sizeof(pthread_t));
    kbuf **all = (kbuf **)calloc(T, sizeof(kbuf *));
    int err = 0;
    for (int t = 0; t < T; t++) {
        all[t] = (kbuf *)calloc(T, sizeof(kbuf));
        jobs[t].buf = buf; jobs[t].offs = offs; jobs[t].k = k; jobs[t].tid = t; jobs[t].T = T;
        jobs[t].r0 = n_reads * t / T; jobs[t].r1 = n_reads * (t + 1) / T;
        jobs[t].out = all[t]; jobs[t].all = all; jobs[t].min_count = min_count;
        /* buffers sized for an even deal up front (they still grow if the deal is uneven) */
        const int64_t share = (offs[jobs[t].r1] - offs[jobs[t].r0]) / T;
        for (int p = 0; p < T; p++) {
            all[t][p].cap = share + share / 8 + 4096;
            all[t][p].v = (u128 *)malloc((size_t)all[t][p].cap * sizeof(u128));
            if (!all[t][p].v) { all[t][p].cap = 0; }
        }
    }
    for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, deal_run, &jobs[t]);
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); err |= jobs[t].err; }
    if (!err) {
        for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, tally_run, &jobs[t]);
        for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); err |= jobs[t].err; }
    }
    out[0] = out[1] = out[2] = 0;
    for (int t = 0; t < T; t++) {
        for (int q = 0; q < 3; q++) out[q] += jobs[t].res[q];
        for (int p = 0; p < T; p++) free(all[t][p].v);
        free(all[t]);
    }
    free(all); free(jobs); free(th);
    return err ? -1 : 0;
}

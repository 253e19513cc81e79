/*
 * mc_oracle.c -- CPU restatement of MetaCherchant's environment-finder hot path.
 * TEST INFRASTRUCTURE ONLY (see mc_oracle.h for the rules and the parity status).
 *
 * Every function cites the reference lines it restates:
 *   src/...  = /root/reference/src/...
 *   itmo!/x  = ru/ifmo/genetics/x in /root/reference/lib/itmo-assembler-src.jar
 */
#define _GNU_SOURCE
#include "mc_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ encoding */

/* itmo!/dna/DnaTools.java:31 NUCLEOTIDES = {'A','G','C','T'}; :46-64 fromChar */
int mco_code(int ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'G': case 'g': return 1;
    case 'C': case 'c': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

char mco_char(int code) { return "AGCT"[code & 3]; }

/* itmo!/utils/KmerUtils.java:12-22 reverseComplement(long kmer, long k) */
uint64_t mco_rc_packed(uint64_t kmer, int k)
{
    kmer = ((kmer & 0x3333333333333333ULL) << 2) | ((kmer & 0xccccccccccccccccULL) >> 2);
    kmer = ((kmer & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((kmer & 0xf0f0f0f0f0f0f0f0ULL) >> 4);
    kmer = ((kmer & 0x00ff00ff00ff00ffULL) << 8) | ((kmer & 0xff00ff00ff00ff00ULL) >> 8);
    kmer = ((kmer & 0x0000ffff0000ffffULL) << 16) | ((kmer & 0xffff0000ffff0000ULL) >> 16);
    kmer = ((kmer & 0x00000000ffffffffULL) << 32) | ((kmer & 0xffffffff00000000ULL) >> 32);
    kmer = ~kmer;
    return kmer >> (64 - 2 * k);
}

/* itmo!/dna/DnaTools.java:123-129 toLong + itmo!/utils/KmerUtils.java:58-60 getKmerKey
 * (== itmo!/dna/kmers/ShortKmer.java:54-56 toLong = Math.min(fwKmer, rcKmer), signed) */
int64_t mco_key31(const uint8_t *codes, int k)
{
    uint64_t fw = 0;
    for (int i = 0; i < k; i++) fw = (fw << 2) + codes[i];
    uint64_t rc = mco_rc_packed(fw, k);
    int64_t a = (int64_t)fw, b = (int64_t)rc;
    return a < b ? a : b;
}

/* src/utils/PolynomialHash.java:19-28 hash(Dna, start, end) */
int64_t mco_poly(const uint8_t *codes, int k)
{
    uint64_t fw = 1, rc = 1; /* Java long arithmetic wraps mod 2^64 */
    for (int i = 0; i < k; i++) {
        fw *= 5;
        rc *= 5;
        fw += codes[i];
        rc += (uint64_t)(3 ^ codes[k - 1 - i]);
    }
    int64_t a = (int64_t)fw, b = (int64_t)rc;
    return a < b ? a : b; /* Math.min on signed longs */
}

/* src/utils/FNV1AHash.java:8-9,33-42 */
int64_t mco_fnv1a(const uint8_t *codes, int k)
{
    const uint64_t basis = 14695981039346656037ULL; /* = -3750763034362895579L */
    const uint64_t prime = 1099511628211ULL;
    uint64_t fw = basis, rc = basis;
    for (int i = 0; i < k; i++) {
        fw ^= (uint64_t)codes[i];
        rc ^= (uint64_t)(3 ^ codes[k - 1 - i]);
        fw *= prime;
        rc *= prime;
    }
    int64_t a = (int64_t)fw, b = (int64_t)rc;
    return a < b ? a : b;
}

/* src/tools/EnvironmentFinderMain.java:128-136 (which loader) and
 * src/algo/OneSequenceCalculator.java:89-96 (getKmerKey) */
int64_t mco_key(const uint8_t *codes, int k, int mode)
{
    switch (mode) {
    case MCO_KEY_PACKED: return mco_key31(codes, k);
    case MCO_KEY_POLY: return mco_poly(codes, k);
    default: return mco_fnv1a(codes, k);
    }
}

void mco_pack(const uint8_t *codes, uint64_t n_bases, uint64_t *words)
{
    uint64_t nw = (n_bases + 31) / 32;
    memset(words, 0, nw * 8);
    for (uint64_t i = 0; i < n_bases; i++)
        words[i >> 5] |= (uint64_t)(codes[i] & 3) << (62 - 2 * (i & 31));
}

static inline uint8_t packed_at(const uint64_t *words, uint64_t i)
{
    return (uint8_t)((words[i >> 5] >> (62 - 2 * (i & 31))) & 3);
}

/* ------------------------------------------------------------------ table */
/* Value semantics of itmo!/structures/map/BigLong2ShortHashMap.java:68-77 ->
 * Long2ShortHashMap.java:119-175.  The reference's internal layout never
 * reaches any output so a plain open-addressing map is used.
 * Key 0 is stored out of band exactly like LongHashSet.FREE
 * (itmo!/structures/set/LongHashSet.java:33). */
struct mco_table {
    uint64_t cap, size; /* size excludes the free key */
    int64_t *keys;
    int16_t *vals;
    int has_free;
    int16_t free_val;
};

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

mco_table *mco_table_new(void)
{
    mco_table *t = calloc(1, sizeof *t);
    t->cap = 1u << 16;
    t->keys = calloc(t->cap, 8);
    t->vals = calloc(t->cap, 2);
    return t;
}

void mco_table_free(mco_table *t)
{
    if (!t) return;
    free(t->keys);
    free(t->vals);
    free(t);
}

static void table_grow(mco_table *t)
{
    uint64_t ncap = t->cap * 2;
    int64_t *nk = calloc(ncap, 8);
    int16_t *nv = calloc(ncap, 2);
    for (uint64_t i = 0; i < t->cap; i++) {
        if (!t->keys[i]) continue;
        uint64_t p = mix64((uint64_t)t->keys[i]) & (ncap - 1);
        while (nk[p]) p = (p + 1) & (ncap - 1);
        nk[p] = t->keys[i];
        nv[p] = t->vals[i];
    }
    free(t->keys);
    free(t->vals);
    t->keys = nk;
    t->vals = nv;
    t->cap = ncap;
}

/* itmo!/utils/NumUtils.java:21-26 addAndBound(short, short) */
static inline int16_t add_and_bound(int16_t v, int inc)
{
    if (v > 32767 - inc) return 32767;
    return (int16_t)(v + inc);
}

void mco_table_add(mco_table *t, int64_t key, int inc)
{
    if (key == 0) { /* Long2ShortHashMap.java:120-134 */
        t->free_val = add_and_bound(t->free_val, inc);
        t->has_free = 1;
        return;
    }
    uint64_t p = mix64((uint64_t)key) & (t->cap - 1);
    while (t->keys[p] && t->keys[p] != key) p = (p + 1) & (t->cap - 1);
    if (!t->keys[p]) { /* values start at 0: Long2ShortHashMap.java:35-38 */
        t->keys[p] = key;
        t->vals[p] = add_and_bound(0, inc);
        if (++t->size * 4 >= t->cap * 3) table_grow(t);
    } else {
        t->vals[p] = add_and_bound(t->vals[p], inc);
    }
}

int16_t mco_table_get(const mco_table *t, int64_t key)
{
    if (key == 0) return t->has_free ? t->free_val : -1; /* :162-167 */
    uint64_t p = mix64((uint64_t)key) & (t->cap - 1);
    while (t->keys[p] && t->keys[p] != key) p = (p + 1) & (t->cap - 1);
    return t->keys[p] ? t->vals[p] : -1; /* :168-174 */
}

uint64_t mco_table_size(const mco_table *t) { return t->size + (t->has_free ? 1 : 0); }

uint64_t mco_table_dump(const mco_table *t, int64_t *keys, int16_t *counts, uint64_t cap)
{
    uint64_t n = 0;
    if (t->has_free && n < cap) { keys[n] = 0; counts[n] = t->free_val; n++; }
    for (uint64_t i = 0; i < t->cap && n < cap; i++)
        if (t->keys[i]) { keys[n] = t->keys[i]; counts[n] = t->vals[i]; n++; }
    return n;
}

/* ------------------------------------------------------------------ counting */

/* src/io/IOUtils.java:201-214 (k<=31, !forcehash) and src/io/LargeKIOUtils.java:41-54:
 * every window of every read; reads shorter than k give nothing (minSeqLen = 0). */
uint64_t mco_count_reads(mco_table *t, const uint8_t *codes, const uint64_t *offsets,
                         uint64_t n_reads, int k, int mode)
{
    uint64_t windows = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        uint64_t b = offsets[r], e = offsets[r + 1];
        for (uint64_t i = b; i + (uint64_t)k <= e; i++) {
            mco_table_add(t, mco_key(codes + i, k, mode), 1);
            windows++;
        }
    }
    return windows;
}

/* rolling form for the packed k<=31 key: itmo!/dna/kmers/ShortKmer.java:68-71 shiftRight */
static uint64_t count_read_packed31(mco_table *t, const uint64_t *words, uint64_t b, uint64_t e, int k)
{
    if (e - b < (uint64_t)k) return 0;
    const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t fw = 0, rc = 0, n = 0;
    for (uint64_t i = b; i < e; i++) {
        uint64_t c = packed_at(words, i);
        fw = ((fw << 2) | c) & mask;
        rc = (rc >> 2) | ((3ULL - c) << (2 * k - 2));
        if (i - b + 1 >= (uint64_t)k) {
            int64_t a = (int64_t)fw, bb = (int64_t)rc;
            mco_table_add(t, a < bb ? a : bb, 1);
            n++;
        }
    }
    return n;
}

uint64_t mco_count_reads_packed(mco_table *t, const uint64_t *words, const uint64_t *offsets,
                                uint64_t n_reads, int k, int mode)
{
    uint64_t windows = 0;
    uint8_t buf[64];
    for (uint64_t r = 0; r < n_reads; r++) {
        uint64_t b = offsets[r], e = offsets[r + 1];
        if (mode == MCO_KEY_PACKED) {
            windows += count_read_packed31(t, words, b, e, k);
            continue;
        }
        for (uint64_t i = b; i + (uint64_t)k <= e; i++) {
            for (int j = 0; j < k; j++) buf[j] = packed_at(words, i + j);
            mco_table_add(t, mco_key(buf, k, mode), 1);
            windows++;
        }
    }
    return windows;
}

/* ------------------------------------------------------------------ CPU baseline (design restatement) */

typedef struct {
    pthread_mutex_t lock; /* ReentrantLock writeLock, itmo!/structures/set/LongHashSet.java:63 */
    uint64_t cap, size;
    int64_t *keys;
    int16_t *vals;
    int has_free;
    int16_t free_val;
} submap;

typedef struct {
    submap *maps;
    uint32_t mask;
} bigmap;

/* fastutil HashCommon.murmurHash3(int) -- MurmurHash3 fmix32; only picks the sub-map
 * (itmo!/structures/map/BigLong2ShortHashMap.java:69), never reaches an output. */
static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bU;
    h ^= h >> 13; h *= 0xc2b2ae35U;
    h ^= h >> 16;
    return h;
}

static void submap_grow(submap *m) /* Long2ShortHashMap.java:191-214 enlargeAndRehash */
{
    uint64_t ncap = m->cap * 2;
    int64_t *nk = calloc(ncap, 8);
    int16_t *nv = calloc(ncap, 2);
    for (uint64_t i = 0; i < m->cap; i++) {
        if (!m->keys[i]) continue;
        uint64_t p = mix64((uint64_t)m->keys[i]) & (ncap - 1);
        while (nk[p]) p = (p + 1) & (ncap - 1);
        nk[p] = m->keys[i];
        nv[p] = m->vals[i];
    }
    free(m->keys);
    free(m->vals);
    m->keys = nk;
    m->vals = nv;
    m->cap = ncap;
}

static inline void bigmap_add(bigmap *bm, int64_t key)
{
    submap *m = &bm->maps[fmix32((uint32_t)key) & bm->mask];
    pthread_mutex_lock(&m->lock); /* one lock acquire per occurrence, as the reference */
    if (key == 0) {
        m->free_val = add_and_bound(m->free_val, 1);
        m->has_free = 1;
    } else {
        uint64_t p = mix64((uint64_t)key) & (m->cap - 1);
        while (m->keys[p] && m->keys[p] != key) p = (p + 1) & (m->cap - 1);
        if (!m->keys[p]) {
            m->keys[p] = key;
            m->vals[p] = 1;
            if (++m->size * 4 >= m->cap * 3) submap_grow(m);
        } else {
            m->vals[p] = add_and_bound(m->vals[p], 1);
        }
    }
    pthread_mutex_unlock(&m->lock);
}

typedef struct {
    bigmap *bm;
    const uint64_t *words, *offsets;
    uint64_t n_reads;
    int k, mode;
    pthread_mutex_t *disp_lock; /* ReadsDispatcher.getWorkRange is synchronized */
    uint64_t *next_read;
    uint64_t windows;
} mt_worker;

static void *mt_worker_run(void *arg)
{
    mt_worker *w = arg;
    const uint64_t RANGE = 1u << 15; /* src/io/IOUtils.java:25 READS_WORK_RANGE_SIZE */
    uint8_t buf[64];
    for (;;) {
        pthread_mutex_lock(w->disp_lock);
        uint64_t b = *w->next_read;
        uint64_t e = b + RANGE < w->n_reads ? b + RANGE : w->n_reads;
        *w->next_read = e;
        pthread_mutex_unlock(w->disp_lock);
        if (b >= e) break;
        for (uint64_t r = b; r < e; r++) {
            uint64_t rb = w->offsets[r], re = w->offsets[r + 1];
            if (re - rb < (uint64_t)w->k) continue;
            if (w->mode == MCO_KEY_PACKED) {
                const int k = w->k;
                const uint64_t mask = (1ULL << (2 * k)) - 1;
                uint64_t fw = 0, rc = 0;
                for (uint64_t i = rb; i < re; i++) {
                    uint64_t c = packed_at(w->words, i);
                    fw = ((fw << 2) | c) & mask;
                    rc = (rc >> 2) | ((3ULL - c) << (2 * k - 2));
                    if (i - rb + 1 >= (uint64_t)k) {
                        int64_t a = (int64_t)fw, bb = (int64_t)rc;
                        bigmap_add(w->bm, a < bb ? a : bb);
                        w->windows++;
                    }
                }
            } else {
                /* the reference recomputes the hash from scratch per window:
                 * src/io/LargeKIOUtils.java:48-50 */
                for (uint64_t i = rb; i + (uint64_t)w->k <= re; i++) {
                    for (int j = 0; j < w->k; j++) buf[j] = packed_at(w->words, i + j);
                    bigmap_add(w->bm, mco_key(buf, w->k, w->mode));
                    w->windows++;
                }
            }
        }
    }
    return NULL;
}

uint64_t mco_count_reads_packed_mt(const uint64_t *words, const uint64_t *offsets, uint64_t n_reads,
                                   int k, int mode, int threads, uint64_t *n_distinct, double *seconds,
                                   mco_table **out_table)
{
    if (threads < 1) threads = 1;
    int lg = 0;
    while ((1 << (lg + 1)) <= threads) lg++;
    /* src/io/IOUtils.java:220-221: new BigLong2ShortHashMap((int)(ln P / ln 2) + 4, 12, true) */
    uint32_t nmaps = 1u << (lg + 4);
    bigmap bm;
    bm.mask = nmaps - 1;
    bm.maps = calloc(nmaps, sizeof(submap));
    for (uint32_t i = 0; i < nmaps; i++) {
        pthread_mutex_init(&bm.maps[i].lock, NULL);
        bm.maps[i].cap = 1u << 12;
        bm.maps[i].keys = calloc(bm.maps[i].cap, 8);
        bm.maps[i].vals = calloc(bm.maps[i].cap, 2);
    }
    pthread_mutex_t disp;
    pthread_mutex_init(&disp, NULL);
    uint64_t next = 0;
    mt_worker *ws = calloc((size_t)threads, sizeof *ws);
    pthread_t *th = calloc((size_t)threads, sizeof *th);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < threads; i++) {
        ws[i] = (mt_worker){&bm, words, offsets, n_reads, k, mode, &disp, &next, 0};
        pthread_create(&th[i], NULL, mt_worker_run, &ws[i]);
    }
    uint64_t windows = 0;
    for (int i = 0; i < threads; i++) {
        pthread_join(th[i], NULL); /* CountDownLatch.await, src/io/IOUtils.java:303 */
        windows += ws[i].windows;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    uint64_t nd = 0;
    mco_table *ot = out_table ? mco_table_new() : NULL;
    for (uint32_t i = 0; i < nmaps; i++) {
        submap *m = &bm.maps[i];
        nd += m->size + (m->has_free ? 1 : 0);
        if (ot) {
            if (m->has_free) mco_table_add(ot, 0, m->free_val);
            for (uint64_t j = 0; j < m->cap; j++)
                if (m->keys[j]) mco_table_add(ot, m->keys[j], m->vals[j]);
        }
        free(m->keys);
        free(m->vals);
        pthread_mutex_destroy(&m->lock);
    }
    if (n_distinct) *n_distinct = nd;
    if (out_table) *out_table = ot;
    free(bm.maps);
    free(ws);
    free(th);
    pthread_mutex_destroy(&disp);
    return windows;
}

/* ------------------------------------------------------------------ BFS */

/* insertion-ordered map  oriented k-mer (u128) -> index; plays distanceToKmer */
typedef struct {
    uint64_t cap, n;
    u128 *keys;    /* by slot; valid iff idx[slot] != UINT64_MAX */
    uint64_t *idx; /* slot -> entry index */
} omap;

static void omap_init(omap *m, uint64_t cap)
{
    m->cap = cap;
    m->n = 0;
    m->keys = malloc(cap * sizeof(u128));
    m->idx = malloc(cap * 8);
    memset(m->idx, 0xff, cap * 8);
}

static inline uint64_t omap_hash(u128 x) { return mix64((uint64_t)x ^ mix64((uint64_t)(x >> 64) + 0x9e3779b97f4a7c15ULL)); }

static uint64_t omap_find(const omap *m, u128 key)
{
    uint64_t p = omap_hash(key) & (m->cap - 1);
    while (m->idx[p] != UINT64_MAX) {
        if (m->keys[p] == key) return m->idx[p];
        p = (p + 1) & (m->cap - 1);
    }
    return UINT64_MAX;
}

static void omap_put_raw(omap *m, u128 key, uint64_t index)
{
    uint64_t p = omap_hash(key) & (m->cap - 1);
    while (m->idx[p] != UINT64_MAX) p = (p + 1) & (m->cap - 1);
    m->keys[p] = key;
    m->idx[p] = index;
}

static void omap_insert(omap *m, u128 key, uint64_t index)
{
    if ((m->n + 1) * 2 > m->cap) {
        omap o = *m;
        omap_init(m, o.cap * 2);
        m->n = o.n;
        for (uint64_t i = 0; i < o.cap; i++)
            if (o.idx[i] != UINT64_MAX) omap_put_raw(m, o.keys[i], o.idx[i]);
        free(o.keys);
        free(o.idx);
    }
    omap_put_raw(m, key, index);
    m->n++;
}

static void omap_free(omap *m)
{
    free(m->keys);
    free(m->idx);
}

static inline u128 kmask(int k) { return (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }

static void unpack_kmer(u128 v, int k, uint8_t *codes)
{
    for (int i = 0; i < k; i++) codes[i] = (uint8_t)((v >> (2 * (k - 1 - i))) & 3);
}

/* reads.get(getKmerKey(s)): src/algo/OneSequenceCalculator.java:89-96 */
static int16_t lookup(const mco_table *t, u128 v, int k, int mode, uint64_t *lookups)
{
    uint8_t codes[64];
    unpack_kmer(v, k, codes);
    (*lookups)++;
    return mco_table_get(t, mco_key(codes, k, mode));
}

/* src/utils/StringUtils.java:8-32: left = NUCLEOTIDES[i] + kmer[0..k-2]; right = kmer[1..] + NUCLEOTIDES[i];
 * all = L0,R0,L1,R1,L2,R2,L3,R3.  NUCLEOTIDES order is A,G,C,T = codes 0,1,2,3. */
static int neighbours(u128 v, int k, int dir, u128 *out)
{
    const u128 mask = kmask(k);
    if (dir == -1) {
        for (int c = 0; c < 4; c++) out[c] = (v >> 2) | ((u128)c << (2 * (k - 1)));
        return 4;
    }
    if (dir == 1) {
        for (int c = 0; c < 4; c++) out[c] = ((v << 2) | (u128)c) & mask;
        return 4;
    }
    for (int c = 0; c < 4; c++) {
        out[2 * c] = (v >> 2) | ((u128)c << (2 * (k - 1)));
        out[2 * c + 1] = ((v << 2) | (u128)c) & mask;
    }
    return 8;
}

typedef struct {
    uint64_t n, cap;
    u128 *kmer;
    int32_t *dist;
    int16_t *cov;
    uint8_t *last;
} dlist;

static void dlist_push(dlist *d, u128 v, int32_t dist, int16_t cov)
{
    if (d->n == d->cap) {
        d->cap = d->cap ? d->cap * 2 : 1024;
        d->kmer = realloc(d->kmer, d->cap * sizeof(u128));
        d->dist = realloc(d->dist, d->cap * 4);
        d->cov = realloc(d->cov, d->cap * 2);
        d->last = realloc(d->last, d->cap);
    }
    d->kmer[d->n] = v;
    d->dist[d->n] = dist;
    d->cov[d->n] = cov;
    d->last[d->n] = 0;
    d->n++;
}

/* src/algo/OneSequenceCalculator.java:154-220 runBfs + :241-262 runTrimPaths +
 * src/algo/TerminationMode.java:31-47 allowsAddition */
int mco_bfs(const mco_table *t, int k, int mode, const uint8_t *seed_codes, const uint64_t *seed_off,
            uint64_t n_seeds, int dir, int min_cov, int64_t max_kmers, int64_t max_radius, int trim,
            mco_bfs_result *out)
{
    memset(out, 0, sizeof *out);
    dlist D = {0};
    omap M;
    omap_init(&M, 1u << 12);
    uint64_t qcap = 1024, qn = 0, lookups = 0;
    uint64_t *queue = malloc(qcap * 8); /* indices into D; duplicates allowed (seed re-queue) */
    const u128 mask = kmask(k);

    /* :159-192 seeds: every window with occs >= minOccurences -> queue.add; distanceToKmer.put(kmer, 0) */
    for (uint64_t s = 0; s < n_seeds; s++) {
        uint64_t b = seed_off[s], e = seed_off[s + 1];
        for (uint64_t i = b; i + (uint64_t)k <= e; i++) {
            u128 v = 0;
            for (int j = 0; j < k; j++) v = (v << 2) | seed_codes[i + j];
            v &= mask;
            int16_t occs = lookup(t, v, k, mode, &lookups);
            if (occs >= min_cov) {
                uint64_t id = omap_find(&M, v);
                if (id == UINT64_MAX) {
                    id = D.n;
                    dlist_push(&D, v, 0, occs);
                    omap_insert(&M, v, id);
                } else {
                    D.dist[id] = 0; /* put() overwrites the value, position kept */
                }
                if (qn == qcap) { qcap *= 2; queue = realloc(queue, qcap * 8); }
                queue[qn++] = id;
            }
        }
    }
    if (qn == 0) { /* :193-196 */
        free(queue);
        omap_free(&M);
        return 1;
    }
    uint64_t head = 0;
    int32_t maxd = 0;
    while (head < qn) { /* :198-214 */
        uint64_t id = queue[head++];
        u128 v = D.kmer[id];
        int32_t distance = D.dist[id];
        u128 nb[8];
        int nn = neighbours(v, k, dir, nb);
        for (int j = 0; j < nn; j++) {
            int16_t occs = lookup(t, nb[j], k, mode, &lookups);
            if (occs >= min_cov) {
                /* TerminationMode.allowsAddition(distanceToKmer, neighbor, distance + 1) */
                int allowed = omap_find(&M, nb[j]) == UINT64_MAX;
                if (allowed && max_kmers >= 0 && (int64_t)D.n >= max_kmers) allowed = 0;
                if (allowed && max_radius >= 0 && (int64_t)distance + 1 > max_radius) allowed = 0;
                if (allowed) {
                    uint64_t nid = D.n;
                    dlist_push(&D, nb[j], distance + 1, occs);
                    omap_insert(&M, nb[j], nid);
                    if (qn == qcap) { qcap *= 2; queue = realloc(queue, qcap * 8); }
                    queue[qn++] = nid;
                    if (distance + 1 > maxd) maxd = distance + 1;
                } else {
                    D.last[id] = 1; /* lastKmers.add(kmer) */
                }
            }
        }
    }

    uint8_t *kept = malloc(D.n ? D.n : 1);
    memset(kept, 1, D.n ? D.n : 1);
    if (trim) { /* :241-262: reverse BFS from lastKmers through getNeighborsByDir(-dir), inside distanceToKmer */
        memset(kept, 0, D.n ? D.n : 1);
        uint64_t *tq = malloc((D.n ? D.n : 1) * 8), tn = 0, th = 0;
        for (uint64_t i = 0; i < D.n; i++)
            if (D.last[i]) { kept[i] = 1; tq[tn++] = i; }
        while (th < tn) {
            uint64_t id = tq[th++];
            u128 nb[8];
            int nn = neighbours(D.kmer[id], k, -dir, nb);
            for (int j = 0; j < nn; j++) {
                uint64_t nid = omap_find(&M, nb[j]);
                if (nid != UINT64_MAX && !kept[nid]) { kept[nid] = 1; tq[tn++] = nid; }
            }
        }
        free(tq);
    }

    out->n = D.n;
    out->hi = malloc((D.n ? D.n : 1) * 8);
    out->lo = malloc((D.n ? D.n : 1) * 8);
    for (uint64_t i = 0; i < D.n; i++) {
        out->hi[i] = (uint64_t)(D.kmer[i] >> 64);
        out->lo[i] = (uint64_t)D.kmer[i];
    }
    out->dist = D.dist;
    out->cov = D.cov;
    out->last = D.last;
    out->kept = kept;
    out->queue_len = qn;
    out->levels = (uint64_t)maxd;
    out->lookups = lookups;
    free(D.kmer);
    free(queue);
    omap_free(&M);
    return 0;
}

void mco_bfs_free(mco_bfs_result *r)
{
    free(r->hi); free(r->lo); free(r->dist); free(r->cov); free(r->last); free(r->kept);
    memset(r, 0, sizeof *r);
}

/* ------------------------------------------------------------------ synthetic inputs */

/* SplitMix64, n-th output (n >= 0) of the generator seeded with `seed` */
uint64_t mco_splitmix(uint64_t seed, uint64_t n)
{
    uint64_t z = seed + (n + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void mco_synth_genome(uint64_t seed, uint64_t n_bases, uint8_t *codes)
{
    for (uint64_t g = 0; g < n_bases; g++) codes[g] = (uint8_t)(mco_splitmix(seed, g) & 3);
}

#define MCO_ERR_STREAM 0xE44044E44044E440ULL

void mco_synth_reads(const uint8_t *genome, uint64_t n_contigs, uint64_t contig_len, uint64_t seed,
                     uint64_t first_read, uint64_t n_reads, int L, int err_per_10k, uint8_t *codes)
{
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t r = first_read + i;
        uint64_t x0 = mco_splitmix(seed, 2 * r), x1 = mco_splitmix(seed, 2 * r + 1);
        uint64_t contig = (x0 >> 33) % n_contigs;
        int strand = (int)(x0 & 1);
        uint64_t start = x1 % (contig_len - (uint64_t)L + 1);
        const uint8_t *g = genome + contig * contig_len + start;
        uint8_t *o = codes + i * (uint64_t)L;
        for (int j = 0; j < L; j++) {
            uint8_t b = strand ? (uint8_t)(3 ^ g[L - 1 - j]) : g[j];
            if (err_per_10k > 0) {
                uint64_t e = mco_splitmix(seed ^ MCO_ERR_STREAM, r * (uint64_t)L + (uint64_t)j);
                if ((int)(e % 10000) < err_per_10k) b = (uint8_t)((b + 1 + ((e >> 40) % 3)) & 3);
            }
            o[j] = b;
        }
    }
}

/*
 * mc_oracle.h -- CPU restatement of MetaCherchant's environment-finder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / the reported CPU
 * baseline.  The product path (metacherchant_amd/, libmcgpu.so) never links,
 * imports or calls it.
 *
 * Parity status: the Java reference cannot be built or run in the build
 * container (no JVM, lib/itmo-assembler.jar is a missing blob), so this is a
 * restatement that follows the reference line by line (citations below and in
 * mc_oracle.c).  It is pinned against the reference's own shipped data
 * (Hi-C_pipline/example_work_dir/output/1/merged/graph.txt and friends, see
 * tests/test_oracle_golden.py) at the level that data allows: k-mer set,
 * coverages, canonical orientation and java.util.HashMap bucket order.
 * Intra-bucket line order of that fixture comes from an older revision of the
 * reference and is NOT reproduced by the current sources: "parity unpinned"
 * for intra-bucket order only.
 *
 * Citation conventions: src/... = /root/reference/src/...;
 * itmo!/x = ru/ifmo/genetics/x inside /root/reference/lib/itmo-assembler-src.jar.
 */
#ifndef MC_ORACLE_H
#define MC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* key modes, chosen exactly as src/tools/EnvironmentFinderMain.java:128,157-169 */
enum { MCO_KEY_PACKED = 0, MCO_KEY_POLY = 1, MCO_KEY_FNV1A = 2 };

/* itmo!/dna/DnaTools.java:31,46-64: A=0 G=1 C=2 T=3 (case-insensitive); -1 otherwise */
int mco_code(int ch);
char mco_char(int code);

/* itmo!/utils/KmerUtils.java:12-22 */
uint64_t mco_rc_packed(uint64_t kmer, int k);
/* itmo!/dna/kmers/ShortKmer.java:54-56 / KmerUtils.java:58-60 -- signed min */
int64_t mco_key31(const uint8_t *codes, int k);
/* src/utils/PolynomialHash.java:19-28 */
int64_t mco_poly(const uint8_t *codes, int k);
/* src/utils/FNV1AHash.java:33-42 */
int64_t mco_fnv1a(const uint8_t *codes, int k);
int64_t mco_key(const uint8_t *codes, int k, int mode);

/* ---- counting table: value semantics of BigLong2ShortHashMap only ---- */
typedef struct mco_table mco_table;
mco_table *mco_table_new(void);
void mco_table_free(mco_table *t);
/* itmo!/structures/map/Long2ShortHashMap.java:119-157 + NumUtils.java:21-26 */
void mco_table_add(mco_table *t, int64_t key, int inc);
/* Long2ShortHashMap.java:160-175: -1 when absent */
int16_t mco_table_get(const mco_table *t, int64_t key);
uint64_t mco_table_size(const mco_table *t);
/* dump all (key,count) pairs, unordered; returns number written (<= cap) */
uint64_t mco_table_dump(const mco_table *t, int64_t *keys, int16_t *counts, uint64_t cap);

/*
 * src/io/IOUtils.java:201-214 / src/io/LargeKIOUtils.java:41-54:
 * for every read, for every window: addAndBound(key, 1).
 * reads are given as one byte per base (codes 0..3), concatenated;
 * offsets[n_reads+1] are base offsets.  Returns windows processed.
 */
uint64_t mco_count_reads(mco_table *t, const uint8_t *codes, const uint64_t *offsets,
                         uint64_t n_reads, int k, int mode);

/* same, but reading the 2-bit packed layout of include/mcgpu.h (MSB-first words) */
uint64_t mco_count_reads_packed(mco_table *t, const uint64_t *words, const uint64_t *offsets,
                                uint64_t n_reads, int k, int mode);

/*
 * CPU baseline: multi-threaded restatement of the reference's *design*
 * (src/io/IOUtils.java:217-248,283-315; src/io/ReadsDispatcher.java:34-53;
 * itmo!/structures/map/BigLong2ShortHashMap.java:44-76): 2^(floor(log2 P)+4)
 * lock-protected open-addressing sub-maps starting at 2^12 slots, doubling at
 * load 0.75, work items of 32768 reads handed out under one lock, P threads.
 * Returns windows processed; *n_distinct gets the final size; *seconds the wall
 * time of the counting loop only.
 */
uint64_t mco_count_reads_packed_mt(const uint64_t *words, const uint64_t *offsets, uint64_t n_reads,
                                   int k, int mode, int threads, uint64_t *n_distinct, double *seconds,
                                   mco_table **out_table /* may be NULL */);

/* pack codes -> MSB-first 2-bit words (layout of include/mcgpu.h) */
void mco_pack(const uint8_t *codes, uint64_t n_bases, uint64_t *words);

/* ---- BFS (src/algo/OneSequenceCalculator.java:154-262, TerminationMode.java:31-47) ---- */
typedef struct {
    uint64_t n;        /* number of distinct oriented k-mers in distanceToKmer (insertion order) */
    uint64_t *hi;      /* packed oriented k-mer, bases 0..k-1 MSB-first over 128 bits: hi = upper 64 */
    uint64_t *lo;
    int32_t *dist;
    int16_t *cov;      /* reads.get(key) */
    uint8_t *last;     /* member of lastKmers */
    uint8_t *kept;     /* survives runTrimPaths (all 1 when trim == 0) */
    uint64_t queue_len; /* queue.size() at the end, duplicates of seeds included */
    uint64_t levels;    /* max distance reached */
    uint64_t lookups;   /* reads.get calls */
} mco_bfs_result;

/*
 * seeds: n_seeds sequences, seed i = codes seed_codes[seed_off[i] .. seed_off[i+1]).
 * dir: -1 left, +1 right, 0 both (8 neighbours interleaved L,R per letter A,G,C,T).
 * max_kmers / max_radius: < 0 = unset.  Returns 0, or 1 when no seed k-mer passes
 * (OneSequenceCalculator.java:193-196 "fail").
 */
int mco_bfs(const mco_table *t, int k, int mode, const uint8_t *seed_codes, const uint64_t *seed_off,
            uint64_t n_seeds, int dir, int min_cov, int64_t max_kmers, int64_t max_radius, int trim,
            mco_bfs_result *out);
void mco_bfs_free(mco_bfs_result *r);

/* ---- synthetic inputs; spec in DESIGN.md "Synthetic workload" ---- */
uint64_t mco_splitmix(uint64_t seed, uint64_t n); /* n-th output (n>=0) of SplitMix64 seeded with seed */
/* genome: n_contigs contigs of contig_len bases each, base = out(seed, global_index) & 3 */
void mco_synth_genome(uint64_t seed, uint64_t n_bases, uint8_t *codes);
/* reads of fixed length L; err_per_10k = substitution rate in 1/10000 (0 = error-free) */
void mco_synth_reads(const uint8_t *genome, uint64_t n_contigs, uint64_t contig_len, uint64_t seed,
                     uint64_t first_read, uint64_t n_reads, int L, int err_per_10k, uint8_t *codes);

#ifdef __cplusplus
}
#endif
#endif

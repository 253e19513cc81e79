/*
 * mcgpu.h -- C ABI of libmcgpu.so: the MI355X (gfx950) implementation of MetaCherchant's
 * environment-finder hot path (k-mer counting over the read set + coverage-thresholded
 * de Bruijn BFS from the seed sequences).
 *
 * This is the drop-in boundary.  The reference has no FFI: the seam is ordinary Java calls
 * inside one JVM, so each entry point below names the Java method whose body it replaces
 * (src/... = the reference's src/ tree, itmo!/x = ru/ifmo/genetics/x in
 * lib/itmo-assembler-src.jar).  INTEGRATION.md shows the JNI stub a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MC_E* code; the message is available
 *     from mc_last_error().  The library never calls exit() and no C++ exception crosses the ABI
 *     (the reference throws ExecutionFailedException, itmo!/utils/tool/Tool.java:450-462).
 *   - plain pointers and sizes only.  Functions ending in _dev take pointers to device (HBM)
 *     memory of the context's device; all others take host pointers.  Work runs on the context's own
 *     HIP stream (or the one given to mc_set_stream) and every call returns with its results complete;
 *     device buffers a caller passes in must not have writes pending on OTHER streams (a memset, a
 *     copy, a collective's result): synchronise those first, or share the stream.
 *   - a context is not re-entrant for mutation (mc_add_*, mc_finalize_counts).  After
 *     mc_finalize_counts, mc_get* and mc_bfs are read-only on the table; calls on ONE context
 *     are serialised internally, several contexts may be used from several threads.
 *
 * Data layout of reads ("2-bit packed"):
 *   base code A=0 G=1 C=2 T=3 (itmo!/dna/DnaTools.java:31 -- note: not ACGT order);
 *   base i of the concatenated read set lives in words[i >> 5], bits (63 - 2*(i & 31)) .. (62 - 2*(i & 31))
 *   (first base = most significant, itmo!/utils/KmerUtils.java:24-39);
 *   read r covers bases [read_offsets[r], read_offsets[r+1]); read_offsets has n_reads + 1 entries;
 *   words[] must hold ceil(n_bases / 32) + 1 entries (one readable pad word).
 *   Reads must be pure ACGT: the N policy of the readers (FASTA records with N dropped, FASTQ
 *   split at phred < 1, itmo!/io/readers/FastaReader.java:54-76,
 *   itmo!/io/readers/FastaReaderFromXQSourceTrunc.java:61-95) is applied by the caller
 *   (metacherchant_amd's host reader does it) before packing.
 */
#ifndef MCGPU_H
#define MCGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_ABI_VERSION 14

/* error codes */
#define MC_OK 0
#define MC_EINVAL (-1)    /* bad argument */
#define MC_ENOMEM (-2)    /* host or device allocation failed */
#define MC_EHIP (-3)      /* HIP runtime error (message has the HIP error string) */
#define MC_ESTATE (-4)    /* call not allowed in this state (e.g. mc_get before mc_finalize_counts) */
#define MC_EOVERFLOW (-5) /* an internal capacity was exceeded and could not be grown */
#define MC_ENOSEED (-6)   /* no seed k-mer reaches min_cov: the reference's "fail"
                             (src/algo/OneSequenceCalculator.java:193-196) */
#define MC_ECHECK (-7)    /* debugging only: MC_BFS_SELFCHECK=1 was set and a finished walk broke an invariant of
                             runBfs (no duplicate vertices, coverages as in the table, queue order, every vertex
                             has a parent); the message lists what */

/* Key modes, chosen exactly as src/tools/EnvironmentFinderMain.java:128-136,157-169:
 * k <= 31 and !forcehash -> PACKED (key = min(fw, rc) of the 2-bit packed k-mer as signed long,
 * itmo!/dna/kmers/ShortKmer.java:54-56); otherwise a 64-bit strand-symmetric hash of the k-mer
 * (src/utils/PolynomialHash.java:19-28 default, src/utils/FNV1AHash.java:33-42 with --hash fnv1a);
 * colliding k-mers share a counter, as in the reference (src/io/LargeKIOUtils.java:46-49 adds the HASH of every window to the
 * map).  That holds for every internal form of the table: where hash keys are kept in minimizer bins of the k-mers' bases
 * (mc_get below), mc_finalize_counts joins the table's keys by key (csrc/dup_check.h), gives a key that two different k-mers
 * brought to two regions the SUM of its counters wherever it is read, and counts and exports it once (mc_stats.dup_keys,
 * dup_ms).  Environment MC_DUP_CHECK=0 skips the join (measurements only: a colliding pair -- expected n^2 / 2^65 of them among n
 * keys -- then keeps two counters, "Hashtable size" counts it twice, an export lists it twice; mc_stats.dup_unchecked says so). */
enum { MC_KEY_PACKED = 0, MC_KEY_POLY = 1, MC_KEY_FNV1A = 2 };

/* mc_config.flags.  MC_FLAG_SOLID_LIST: the k-mers at or above the coverage hint will be exported (a shard of a
 * multi-GPU run, mc_export_dev): the counting pass then lists them as it goes, which saves the export a table sweep. */
#define MC_FLAG_SOLID_LIST 1
/* MC_FLAG_GROUP_RCCL (mc_group_create only): the buckets travel between the devices through RCCL -- one communicator per
 * device inside the process (ncclCommInitAll), the exchange as grouped ncclSend / ncclRecv over xGMI -- instead of
 * peer-to-peer copies; the environment variable MC_GROUP_TRANSPORT=rccl|peer says the same.  librccl is loaded on demand. */
#define MC_FLAG_GROUP_RCCL 4

typedef struct mc_ctx mc_ctx;

typedef struct {
    int32_t k;              /* 1..31 for MC_KEY_PACKED, 1..63 for the hash modes */
    int32_t key_mode;       /* MC_KEY_* */
    int32_t device;         /* HIP device ordinal */
    int32_t flags;          /* MC_FLAG_* */
    uint64_t capacity_hint; /* expected number of distinct keys; 0 = start small and grow */
} mc_config;

/* Tool lifetime: one context = one k-mer table = the BigLong2ShortHashMap created by
 * src/io/IOUtils.java:220-221 / src/io/LargeKIOUtils.java:60-61. */
int mc_create(const mc_config *cfg, mc_ctx **out);
void mc_destroy(mc_ctx *ctx);
/* ctx may be NULL: returns the message of the last failed mc_create on this thread. */
const char *mc_last_error(const mc_ctx *ctx);
int mc_abi_version(void);

/* Empties the table (keeps its allocation): a fresh BigLong2ShortHashMap without paying hipMalloc again. */
int mc_clear(mc_ctx *ctx);

/* Optional: tells the context which --coverage the BFS will use (minOccurences,
 * src/tools/EnvironmentFinderMain.java:61-65, known before the reads are loaded).  Counting then
 * keeps the number of k-mers with count >= min_cov up to date as it goes, which saves mc_bfs* one
 * sweep over the table.  Results never depend on it; 0 turns it off. */
int mc_set_coverage_hint(mc_ctx *ctx, int min_cov);

/* Read pointers.  A context keeps the packed bases of every read it counts (its "read store", until mc_clear) and each
 * table slot remembers where one occurrence of its k-mer sits in it: the BFS reads its look-ahead from there (the bases
 * that follow an occurrence are the path the walk will most likely take; every guess is looked up, so results never
 * depend on it).  mc_set_read_pointers(ctx, 0) turns both off for the reads added from then on -- for the ranks of a
 * sharded run whose reads the BFS rank cannot see.  mc_share_read_store(ctx, from): a BFS-only context built by
 * mc_solid_from_pairs_dev reads the store of `from`, the context of the same device that counted (or extracted) this
 * rank's reads and whose pointers the pairs carry; `from` must outlive the BFS calls on ctx (NULL detaches). */
int mc_set_read_pointers(mc_ctx *ctx, int mode);
int mc_share_read_store(mc_ctx *ctx, mc_ctx *from);
/* ... and for the ranks of a sharded run whose reads the BFS rank CAN see, because they are brought to it: with pointers from the
 * walking rank's reads alone -- one read in eight at 8 GPUs -- the walk of 8 x configs[1] took 146 ms instead of 7.9 (its look-ahead
 * had a 3.75-fold read set to follow).  mc_set_read_pointers' mode: 0 no pointers; 1 this context's own store (the default); 2 a
 * store kept by another context: this one keeps nothing, but works out the pointers of the reads it extracts or counts as if
 * they were appended to a store at mc_read_store_tell() -- the caller copies the words there (mc_read_store_import_dev on the
 * context that keeps the store, at that position).  | MC_PTRS_ON_EVERY_RECORD: every record this context will be handed by
 * mc_add_superkmers*_dev / mc_add_keys_dev carries a pointer (the merge kernel need not wait for a later copy that has one).
 *   mc_read_store_seek    the next reads go (mode 1) or are deemed to go (mode 2) to position at_bases, a multiple of 32: ranks
 *                         that share one store take disjoint stretches of it; reserve_bases (mode 1): room up to there
 *   mc_read_store_tell    where the next reads will go
 *   mc_read_store_import_dev  n_words packed words (32 bases each, as in mc_add_reads_packed_dev, with their pad word) of another
 *                         rank's reads to position at_bases of this context's store */
#define MC_PTRS_ON_EVERY_RECORD 0x10
int mc_read_store_seek(mc_ctx *ctx, uint64_t at_bases, uint64_t reserve_bases);
uint64_t mc_read_store_tell(mc_ctx *ctx);
int mc_read_store_import_dev(mc_ctx *ctx, const uint64_t *d_words, uint64_t n_words, uint64_t at_bases);

/* Use the caller's HIP stream (a hipStream_t passed as void*) for all work of this context
 * instead of the context's own stream.  NULL restores the own stream. */
int mc_set_stream(mc_ctx *ctx, void *hip_stream);

/* ---- counting: replaces ReadsLoadWorker.process (src/io/IOUtils.java:201-214,
 * src/io/LargeKIOUtils.java:41-54): for every window of every read, addAndBound(key, 1).
 * May be called repeatedly (one call per batch / file); the result does not depend on call or
 * read order.  Reads shorter than k contribute nothing. */
int mc_add_reads_packed(mc_ctx *ctx, const uint64_t *words, const uint64_t *read_offsets, uint64_t n_reads);
int mc_add_reads_packed_dev(mc_ctx *ctx, const uint64_t *d_words, const uint64_t *d_read_offsets,
                            uint64_t n_reads, uint64_t n_bases);

/* One input file of --reads: ReadsWorker.run / ReadersUtils.readDnaLazyTrunc (src/io/ReadsWorker.java:29-41,
 * itmo!/io/ReadersUtils.java:27-53,104-121): format by extension (.fasta .fa .fn .fna / .fastq .fq, optionally
 * .gz or .bz2; .binq), FASTA records with N dropped whole, FASTQ and BINQ reads split where phred < 1 (quality
 * offset sniffed on the first 1000 records); every read (piece) is counted as by mc_add_reads_packed.  *n_reads (may be NULL) = reads
 * added ("N reads added").  Errors carry the reference's messages ("Can't detect file format for file ...").
 * Uncompressed FASTA / FASTQ text is tokenised on the device (csrc/tokenizer.h: the bytes cross the link in 256 MB
 * chunks, each tokenised and counted while the next one is copied); what the device declines, and every compressed
 * or .binq file, is parsed by the host (environment: MC_TOKENIZER=host parses every file on the host). */
int mc_add_reads_file(mc_ctx *ctx, const char *path, uint64_t *n_reads);

/* End of loadReads (src/io/IOUtils.java:217-248): waits for all queued counting work;
 * *n_distinct = hm.size() ("Hashtable size: N kmers", src/tools/EnvironmentFinderMain.java:137). */
int mc_finalize_counts(mc_ctx *ctx, uint64_t *n_distinct);

/* BigLong2ShortHashMap.get (itmo!/structures/map/BigLong2ShortHashMap.java:74-77 ->
 * Long2ShortHashMap.java:160-175): out[i] = count saturated at 32767
 * (itmo!/utils/NumUtils.java:21-26), or -1 when the key is absent.  Key 0 is legal.
 * A context with polynomial-hash keys of 33 .. 63 bases keeps its table in minimizer bins of the k-mers' BASES while reads
 * are counted into a table whose size something vouches for -- a capacity_hint that still holds, or, for the first batch into
 * an empty table, a sample of that batch -- (csrc/count_long.h), where a bare key does not say which region it lives in: mc_get
 * then answers all n queries with ONE sweep of the table (cost: the table's size, not n), and the first key stream
 * (mc_add_keys_dev, mc_add_pairs_dev, mc_load_kmers), mc_shard_export or batch that takes the direct kernel moves every key to
 * hash-prefix regions first, once and for good (a rebuild of the table: mc_stats.grows counts it), after which reads take the
 * per-window pipeline.  Results are the same either way: mc_finalize_counts has merged the counters of colliding k-mers (above). */
int mc_get(mc_ctx *ctx, const int64_t *keys, uint64_t n, int16_t *out);
int mc_get_dev(mc_ctx *ctx, const int64_t *d_keys, uint64_t n, int16_t *d_out);

/* getKmerKey (src/algo/OneSequenceCalculator.java:89-96) for packed oriented k-mers:
 * k-mer i is the 2k-bit number (hi[i] << 64 | lo[i]), first base most significant
 * (hi may be NULL when k <= 32).  Host pointers. */
int mc_kmer_keys(mc_ctx *ctx, const uint64_t *hi, const uint64_t *lo, uint64_t n, int64_t *out_keys);

/* ---- BFS: replaces OneSequenceCalculator.runBfs up to (not including) runTrimPaths
 * (src/algo/OneSequenceCalculator.java:154-214) with TerminationMode.allowsAddition
 * (src/algo/TerminationMode.java:31-47).
 *
 * Seeds are the k-windows of the seed sequences in file order (then Hi-C seeds when merging),
 * as oriented packed k-mers; the library looks their coverage up and queues those with
 * count >= min_cov (:159-192).  dir: -1 left neighbours, +1 right, 0 all eight interleaved
 * L(A) R(A) L(G) R(G) L(C) R(C) L(T) R(T) (src/utils/StringUtils.java:8-32).
 * max_kmers / max_radius: < 0 = unset (at least one must be set, EnvironmentFinderMain.java:171-175).
 *
 * The result lists the entries of distanceToKmer in insertion order (= BFS discovery order, seeds
 * first, duplicates once): the oriented k-mer, its distance, reads.get(key) and whether the vertex
 * is in lastKmers (needed by runTrimPaths, :241-262, which the host applies).  Arrays are
 * malloc()ed by the library; release with mc_bfs_result_free. */
typedef struct {
    uint64_t n;
    uint64_t *hi, *lo; /* oriented k-mers (hi all zero when k <= 32) */
    int32_t *dist;
    int16_t *cov;
    uint8_t *last;
    uint64_t levels;     /* largest distance assigned */
    uint64_t lookups;    /* table lookups issued (speculative ones included) */
    uint64_t rounds;     /* memory round trips on the critical path (speculation rounds + wide chunks) */
    double device_ms;    /* device time of the BFS kernels of the whole batch, HIP events */
} mc_bfs_result;

int mc_bfs(mc_ctx *ctx, const uint64_t *seed_hi, const uint64_t *seed_lo, uint64_t n_seeds, int dir,
           int min_cov, int64_t max_kmers, int64_t max_radius, mc_bfs_result *out);
void mc_bfs_result_free(mc_bfs_result *r);

/* Several independent BFS passes in ONE launch, one workgroup each: the two passes of
 * buildEnvironment with --bothdirs False (runBfs(-1); runBfs(+1), OneSequenceCalculator.java:137-144)
 * and the one-calculator-per-seed-sequence thread pool of EnvironmentFinderMain.java:218-225.
 * out[] has n_jobs entries; a job whose seeds all fail the threshold returns out[j].n == 0
 * (mc_bfs returns MC_ENOSEED for that). */
typedef struct {
    const uint64_t *seed_hi; /* may be NULL when k <= 32 */
    const uint64_t *seed_lo;
    uint64_t n_seeds;
    int32_t dir;
} mc_bfs_job;
int mc_bfs_batch(mc_ctx *ctx, const mc_bfs_job *jobs, uint32_t n_jobs, int min_cov, int64_t max_kmers,
                 int64_t max_radius, mc_bfs_result *out);

/* ---- table export / import: (key, count) pairs with count >= min_cov, unordered.
 * Also the record content of the reference's .kmers.bin (src/io/KmersLoadWorker.java:9,20-23)
 * and the payload of the multi-GPU gather.  With keys == NULL only *n_out is computed. */
int mc_export(mc_ctx *ctx, int min_cov, int64_t *keys, int16_t *counts, uint64_t cap, uint64_t *n_out);
/* d_hints (may be NULL): the 32-bit read pointer stored with each key -- where one of its occurrences sits in the
 * read store of the context that extracted it, used only to steer the BFS's look-ahead (never part of a result);
 * carry it along with the pairs so that a table rebuilt elsewhere walks as fast as the one that counted the reads
 * (mc_share_read_store). */
int mc_export_dev(mc_ctx *ctx, int min_cov, int64_t *d_keys, int16_t *d_counts, uint32_t *d_hints, uint64_t cap,
                  uint64_t *n_out);
/* table[key] = min(32767, table[key] + count) for each pair (saturating adds commute). */
int mc_add_pairs_dev(mc_ctx *ctx, const int64_t *d_keys, const int16_t *d_counts, const uint32_t *d_hints, uint64_t n);
/* The BFS side of the multi-GPU gather: turns an empty (new or cleared) context into a BFS-only one whose graph
 * is the given pairs with count >= min_cov -- the concatenated mc_export_dev output of every rank, owners being
 * disjoint so no key comes twice; entries with a smaller (or negative: padding) count are skipped.  The pairs go
 * straight into the BFS's own table; the counting table stays empty, so mc_get / mc_export on this context see
 * nothing and mc_bfs* accepts this min_cov only, until mc_clear.  *n_solid (may be NULL) = vertices kept.
 * What the reference does instead: the BFS reads the one shared map (src/algo/OneSequenceCalculator.java:203-204). */
int mc_solid_from_pairs_dev(mc_ctx *ctx, const int64_t *d_keys, const int16_t *d_counts, const uint32_t *d_hints, uint64_t n,
                            int min_cov, uint64_t *n_solid);

/* ---- multi-GPU building blocks (device pointers).  The read set is split across ranks; each
 * rank turns its reads into keys bucketed by owner rank, ranks exchange buckets (RCCL
 * all-to-all, done by the caller), and each rank counts the keys it owns. */
/* owner of a key among n_owners ranks; pure function, same on every rank */
uint32_t mc_key_owner(int64_t key, uint32_t n_owners);
/* Writes the keys of all windows grouped by owner into d_keys (capacity = total windows):
 * owner o's keys are d_keys[owner_offsets[o] .. owner_offsets[o+1]); owner_offsets (host,
 * n_owners + 1 entries) is filled on return.  d_hints (may be NULL, same capacity) receives the
 * speculation hint of every occurrence at the same index.  With d_keys == NULL only the offsets are
 * computed.  n_owners <= 512. */
int mc_extract_keys_dev(mc_ctx *ctx, const uint64_t *d_words, const uint64_t *d_read_offsets, uint64_t n_reads,
                        uint64_t n_bases, uint32_t n_owners, int64_t *d_keys, uint32_t *d_hints, uint64_t keys_cap,
                        uint64_t *owner_offsets);
/* addAndBound(key, 1) for each key; d_hints may be NULL */
int mc_add_keys_dev(mc_ctx *ctx, const int64_t *d_keys, const uint32_t *d_hints, uint64_t n);

/* The same split in the compact form the counting pipeline uses itself when keys are packed k-mers
 * of at least 23 bases: one 16-byte "super-k-mer" record per run of up to 16 consecutive windows of a
 * read that share their minimizer (two uint64 per record; layout in csrc/count_pipeline.h) plus one
 * 32-bit word per record (d_bins: the read pointer of the record's first window, 0 when the context keeps no read
 * store) -- about a seventh of the bytes of the key form.  All windows of a record have the same owner.
 *   mc_superkmer_capacity   records to provide room for, given the windows and reads of a batch;
 *                           0 when this context does not use the form (then use the key form above)
 *   mc_extract_superkmers_dev  as mc_extract_keys_dev; MC_EOVERFLOW when the reads yield more records
 *                           than mc_superkmer_capacity allows for (fall back to the key form)
 *   mc_add_superkmers_dev   addAndBound(key, 1) for every window of every record */
uint64_t mc_superkmer_capacity(mc_ctx *ctx, uint64_t n_windows, uint64_t n_reads);
int mc_extract_superkmers_dev(mc_ctx *ctx, const uint64_t *d_words, const uint64_t *d_read_offsets, uint64_t n_reads,
                              uint64_t n_bases, uint32_t n_owners, uint64_t *d_records, uint32_t *d_bins,
                              uint64_t records_cap, uint64_t *owner_offsets);
int mc_add_superkmers_dev(mc_ctx *ctx, const uint64_t *d_records, const uint32_t *d_bins, uint64_t n);
/* The binned form of the same exchange: the sender also does the first level of the OWNER's counting run, so that the owner's run
 * starts at its second level (the owner's pass that dealt flat received records to its level-1 buckets was 4.5 of the 14.4 ms a
 * rank of 8 x configs[1] counted for; the sender's packing pass does the ordering instead).  Owner o's records
 * d_records[owner_offsets[o] .. owner_offsets[o + 1]) come in the order of n_fine "fine buckets" of the records' bin words;
 * d_fine_counts (device, n_owners x n_fine, written here) says how many records every (owner, fine bucket) cell holds, and
 * owner_windows (host, n_owners) how many k-mer windows every owner's records hold.  What travels to owner o: its records, its
 * pointers (if any) and its row of n_fine counts.
 *   mc_superkmer_fine_buckets  the n_fine to extract with for n_owners owners whose tables are laid out like this context's (the
 *                           level-1 buckets of its counting run: a function of the table's size); 0: no binned form here
 *                           (tables without a second level, more than 16384 / n_owners buckets, MC_EXCHANGE_BINNED=0) -- use the flat form
 *   mc_add_superkmers_binned_dev  what an owner received: n records in n_parts parts lying back to back (part p = records
 *                           part_offsets[p] .. part_offsets[p + 1], part_offsets on the host, n_parts + 1 entries; a part is what one
 *                           rank sent of one chunk), part p in the order of its d_part_counts[p * n_fine ..] (device) fine buckets;
 *                           n_windows = the windows of all records (the senders' owner_windows, summed).  Same result as
 *                           mc_add_superkmers_dev on the same records; counts that do not add up to a part's length are MC_EINVAL.
 *                           Where this context's level-1 buckets are not unions of the fine buckets (its table grew to another
 *                           bucket count), or the parts are more than 1024 x np1 / n_fine, the records are counted as a flat stream. */
uint32_t mc_superkmer_fine_buckets(mc_ctx *ctx, uint32_t n_owners);
int mc_extract_superkmers_binned_dev(mc_ctx *ctx, const uint64_t *d_words, const uint64_t *d_read_offsets, uint64_t n_reads,
                                     uint64_t n_bases, uint32_t n_owners, uint32_t n_fine, uint64_t *d_records, uint32_t *d_bins,
                                     uint64_t records_cap, uint32_t *d_fine_counts, uint64_t *owner_offsets, uint64_t *owner_windows);
int mc_add_superkmers_binned_dev(mc_ctx *ctx, const uint64_t *d_records, const uint32_t *d_bins, uint64_t n, uint64_t n_windows,
                                 uint32_t n_fine, uint32_t n_parts, const uint64_t *part_offsets, const uint32_t *d_part_counts);

/* ---- the walk over several ranks' tables where they are.  After the exchange every rank's counting table holds the k-mers it
 * owns; instead of gathering the ones at or above --coverage into a second table on the rank that walks (mc_export_dev ->
 * mc_solid_from_pairs_dev: at configs[3]'s size that copy does not fit beside the rank's own table), the walking context
 * is given every rank's table and looks a k-mer up in its owner's, through peer access (contexts of one process) or an
 * IPC mapping (one process per GPU).  What the reference does: P threads read the one shared map,
 * src/algo/OneSequenceCalculator.java:203-204.
 *   mc_shard_export  after mc_finalize_counts: an opaque, fixed-size description of this context's table (geometry, an IPC
 *                    handle of its memory) to be sent to the walking rank by any means (an all-gather of 128 bytes).  The
 *                    table must stay as it is -- no counting, no mc_clear, no mc_destroy -- until the walker has detached.
 *   mc_shard_attach  on the walking context (which has counted its own share and keeps the read store the walk reads its
 *                    look-ahead from): shards[i] = rank i's handle, shards[self] its own.  by_minimizer != 0: the exchange
 *                    dealt super-k-mer records (mc_extract_superkmers_dev), so a k-mer is owned by the owner of its
 *                    minimizer; 0: it dealt keys (mc_key_owner).  From here on mc_bfs / mc_bfs_batch on this context walk the
 *                    union of the tables, any --coverage; results equal a single context's.  mc_get / mc_export still see
 *                    this context's own shard only.
 *                    An attachment names every table's block and geometry AS THEY WERE: any call that changes this context's
 *                    counts afterwards (mc_add_*, which may also move the table) drops it, and mc_bfs then fails with
 *                    MC_ESTATE until the tables are exported and attached again (or mc_shard_detach says one table is meant).
 *   mc_shard_detach  gives the mappings up (mc_clear and mc_destroy do so too). */
typedef struct { unsigned char bytes[128]; } mc_shard_handle;
int mc_shard_export(mc_ctx *ctx, mc_shard_handle *out);
int mc_shard_attach(mc_ctx *ctx, const mc_shard_handle *shards, uint32_t n_shards, uint32_t self, int by_minimizer);
int mc_shard_detach(mc_ctx *ctx);

/* ---- several GPUs of one node as ONE table, for a host that is one process (the native `metacherchant --devices 0,1,...`).
 * A context per device and one host thread per device; reads are dealt to the devices in equal contiguous shares, every
 * device buckets its share by owner (mc_extract_superkmers_dev / mc_extract_keys_dev), the buckets travel as peer-to-peer
 * copies over xGMI -- every (source, destination) pair at once -- and every device counts what it owns
 * (mc_add_superkmers_dev / mc_add_keys_dev); the walk runs on the first device and reads every device's table in place
 * (mc_shard_attach; without peer access between all devices, or with MC_GROUP_WALK=gather, the k-mers at or above the
 * threshold are gathered on the first device instead: mc_export_dev -> mc_solid_from_pairs_dev).  Results equal a single context's.
 * `cfg->device` is ignored; capacity_hint is the whole job's.  A device may be named more than once (shares of one GPU).
 * Replaces the same Java as the single-context calls: the P threads over one shared map of src/io/IOUtils.java:283-315 and
 * the work list of src/io/ReadsDispatcher.java:34-53.  ctx may be NULL in mc_group_last_error after a failed create. */
typedef struct mc_group mc_group;
int mc_group_create(const mc_config *cfg, const int32_t *devices, uint32_t n_devices, mc_group **out);
void mc_group_destroy(mc_group *g);
const char *mc_group_last_error(const mc_group *g);
int mc_group_set_coverage_hint(mc_group *g, int min_cov);
int mc_group_add_reads_packed(mc_group *g, const uint64_t *words, const uint64_t *read_offsets, uint64_t n_reads);
int mc_group_add_reads_file(mc_group *g, const char *path, uint64_t *n_reads);
int mc_group_finalize_counts(mc_group *g, uint64_t *n_distinct);
int mc_group_bfs_batch(mc_group *g, const mc_bfs_job *jobs, uint32_t n_jobs, int min_cov, int64_t max_kmers,
                       int64_t max_radius, mc_bfs_result *out);

/* ---- the table as a file: `kmer-counter`'s <name>.kmers.bin and <name>.stat.txt (src/tools/KmersCounter.java:87-121).
 * mc_save_kmers = IOUtils.printKmers (src/io/IOUtils.java:39-65): one 10-byte record per key with count >
 * threshold -- big-endian int64 key, big-endian int16 count (saturated) -- in table order (the reference's order
 * is its map's iteration order; nothing reads it), and the frequency histogram of ALL keys
 * ("# k-mer frequency<TAB>number of such k-mers", ascending, a blank line at the end); stat_path may be NULL.
 * Call after mc_finalize_counts.  *n_total = keys in the table, *n_written = records written.
 * mc_load_kmers = IOUtils.loadKmers (:94-126, src/io/KmersLoadWorker.java:9-23): addAndBound(key, count) for
 * every record with count > freq_threshold; *n_records = records read, *n_added = records added. */
int mc_save_kmers(mc_ctx *ctx, const char *bin_path, const char *stat_path, int threshold, uint64_t *n_total,
                  uint64_t *n_written);
int mc_load_kmers(mc_ctx *ctx, const char *path, int freq_threshold, uint64_t *n_records, uint64_t *n_added);

/* ---- measurement */
typedef struct {
    uint64_t windows;        /* k-mer occurrences counted so far */
    uint64_t count_launches; /* counting passes: launches of the direct kernel, or runs of the partitioned pipeline */
    double count_ms;         /* their summed device time (HIP events on the context's stream) */
    double count_total_ms;   /* device time of all counting-phase kernels */
    uint64_t table_slots;    /* current table capacity (slots) */
    uint64_t table_bytes;
    uint64_t grows;          /* number of table rebuilds */
    double p1_ms, p2_ms, p3_ms; /* partitioned pipeline: extract+scatter / scatter / merge kernels */
    uint64_t spill_keys;     /* keys that did not fit their bucket and took the direct kernel */
    uint64_t solid_kmers;    /* k-mers with count >= min_cov found by the last BFS set-up */
    uint64_t solid_sweeps;   /* table sweeps BFS set-ups needed to count them (0 with mc_set_coverage_hint) */
    uint64_t solid_list_builds; /* BFS set-ups and exports that took their entries from the list the merge kernel left, without sweeping the table */
    uint64_t long_runs;      /* pipeline runs that took long records (polynomial keys, k > 32, a capacity hint: csrc/count_long.h) */
    uint64_t dup_keys;       /* hash keys in minimizer bins: keys that different k-mers brought to more than one region, as the last
                                mc_finalize_counts found them (their counters are merged; 0 once the table has moved to hash-prefix regions) */
    uint64_t dup_checks;     /* joins of the table's keys by key that found that out (csrc/dup_check.h), ... */
    double dup_ms;           /* ... and their summed device time */
    uint64_t dup_unchecked;  /* 1: MC_DUP_CHECK=0 skipped the join for the table as it is now */
    uint64_t left_bins;      /* hash keys of 33 .. 63 bases: 0 while the table is in minimizer bins (reads travel as long records), else why
                                it moved to hash-prefix regions for good (reads take one record a window from then on, ~3 x the counting
                                time): 1 something came or asked by key (mc_add_keys_dev, mc_add_pairs_dev, mc_load_kmers, mc_shard_export,
                                a copy of the solid k-mers), 2 a batch of under 2^22 windows took the direct kernel, 3 nothing vouched for
                                the table's size when a batch came (no capacity_hint that still holds, and the table not empty), 4 bins
                                overflowed or the table had to grow, 5 more keys in several regions than the join's lists take */
    uint64_t binned_runs;    /* mc_add_superkmers_binned_dev calls whose run started at the second level (the others counted a flat stream) */
} mc_stats;
int mc_get_stats(mc_ctx *ctx, mc_stats *out);
int mc_reset_stats(mc_ctx *ctx);
/* Gives the counting pipeline's scratch (the record / key streams of the last run: about 1.3 bytes per base counted) and
 * the idle blocks of the process-wide pools back to the driver -- and, on a context that counted hash keys as long records, the
 * key streams of mc_finalize_counts' join (8 bytes a key, twice: 84 GB for configs[2]'s 4.58 G keys; a walk's check of its
 * look-ups by key then sweeps the table instead) --; the table, the read store and the list of solid k-mers
 * stay.  For callers that need the device's memory between counting and what follows (the next counting call allocates
 * its scratch again).  No counterpart in the reference: the JVM's collector does this for BigLong2ShortHashMap's
 * transient arrays (itmo!/structures/map/BigLong2ShortHashMap.java:46-70). */
int mc_trim(mc_ctx *ctx);
/* ... of a group: counts and bytes summed over its devices, times the largest of any device (they work side by side) */
int mc_group_get_stats(mc_group *g, mc_stats *out);

/* ---- synthetic workload of SURVEY.md section 8(d) (spec: DESIGN.md "Synthetic workload"),
 * generated straight into HBM so that benchmarks start with the reads resident.
 * Reads r = first_read .. first_read + n_reads - 1 of fixed length read_len drawn from
 * n_contigs contigs of contig_len bases; err_per_10k = substitution rate in 1/10000.
 * d_words needs ceil(n_reads*read_len/32) + 1 words, d_read_offsets n_reads + 1. */
int mc_synth_reads_dev(mc_ctx *ctx, uint64_t genome_seed, uint64_t n_contigs, uint64_t contig_len,
                       uint64_t read_seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                       uint32_t err_per_10k, uint64_t *d_words, uint64_t *d_read_offsets);
/* bases [start, start + n) of the synthetic genome as codes 0..3 (host buffer) */
int mc_synth_genome(uint64_t genome_seed, uint64_t start, uint64_t n, uint8_t *codes);

#ifdef __cplusplus
}
#endif
#endif /* MCGPU_H */

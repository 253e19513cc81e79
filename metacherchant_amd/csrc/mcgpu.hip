// libmcgpu.so -- MI355X (gfx950) k-mer counting + de Bruijn BFS behind the C ABI of include/mcgpu.h.
// Hand-written HIP for CDNA4: wave64, 16-byte table slots read with one dwordx4, memory-side
// 64-bit CAS / 32-bit add atomics, LDS-staged ordered compaction.  No MFMA: integer/hash work.
//
// Reference lines each piece replaces are cited at the definitions (src/... and itmo!/... as in
// include/mcgpu.h).  Nothing in this file links or calls oracle/.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// librccl is loaded at run time (RcclApi below); without its headers the library still builds, from the few declarations
// of <rccl/rccl.h> the transport uses
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
#endif

#include "../../include/mcgpu.h"
#include "kmer_device.h"
#include "bfs_device.h"
#include "count_pipeline.h"
#include "count_long.h"
#include "tokenizer.h"
#include "host/envfinder.h"

using namespace mc;

// ------------------------------------------------------------------------------------------ ctx

// Device scratch of the tokeniser (csrc/tokenizer.h), kept between calls: a file is read in chunks of the same size,
// and hipMalloc / hipFree of gigabyte buffers cost milliseconds each.  Best fit with at most 2x slack; at most 48 idle
// blocks (the smallest goes first).
struct DevPool {
    std::vector<std::pair<void *, size_t>> idle;
    hipError_t get(size_t bytes, void **out, size_t *got)
    {
        bytes = std::max<size_t>((bytes + 255) / 256 * 256, 256);
        size_t best = idle.size();
        for (size_t i = 0; i < idle.size(); i++)
            if (idle[i].second >= bytes && idle[i].second <= 2 * bytes + (1u << 20) && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
        if (best < idle.size()) {
            *out = idle[best].first;
            *got = idle[best].second;
            idle.erase(idle.begin() + (long)best);
            return hipSuccess;
        }
        *got = bytes;
        hipError_t e = hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory && !idle.empty()) {  // give the idle blocks back and try once more
            release();
            e = hipMalloc(out, bytes);
        }
        return e;
    }
    void put(void *p, size_t bytes)
    {
        idle.emplace_back(p, bytes);
        if (idle.size() > 48) {
            size_t small = 0;
            for (size_t i = 1; i < idle.size(); i++)
                if (idle[i].second < idle[small].second) small = i;
            (void)hipFree(idle[small].first);
            idle.erase(idle.begin() + (long)small);
        }
    }
    void release()
    {
        for (auto &b : idle) (void)hipFree(b.first);
        idle.clear();
    }
};
template <class T>
struct PoolBuf {  // RAII: a block of a DevPool
    T *p = nullptr;
    size_t bytes = 0;
    DevPool *pool = nullptr;
    PoolBuf() = default;
    PoolBuf(const PoolBuf &) = delete;
    PoolBuf &operator=(const PoolBuf &) = delete;
    ~PoolBuf() { if (p) pool->put(p, bytes); }
    hipError_t alloc(DevPool *pl, size_t n)
    {
        pool = pl;
        return pl->get(std::max<size_t>(n, 1) * sizeof(T), reinterpret_cast<void **>(&p), &bytes);
    }
};

// Table memory, kept across tables: fresh device memory comes zero-filled by the driver at ~30 GB/s (0.7 s for the 22 GB
// table of configs[1]), which a context without a capacity hint paid every time its table went to its real size.  A
// table that is given up goes here instead of back to the driver (the two largest idle blocks per device are kept) and the
// next table of about its size takes it; any allocation that fails for lack of memory empties the pool and tries again.
struct TablePool {
    std::mutex mu;
    std::map<int, DevPool> per_device;
    bool on = [] { const char *e = getenv("MC_TABLE_POOL"); return !(e && !strcmp(e, "0")); }();
    hipError_t get(int dev, size_t bytes, void **out, size_t *got)
    {
        if (!on) { *got = bytes; return hipMalloc(out, bytes); }
        std::lock_guard<std::mutex> g(mu);
        return per_device[dev].get(bytes, out, got);
    }
    void put(int dev, void *p, size_t bytes)
    {
        if (!p) return;
        if (!on) { (void)hipFree(p); return; }
        std::lock_guard<std::mutex> g(mu);
        DevPool &P = per_device[dev];
        P.put(p, bytes);
        while (P.idle.size() > 2) {
            size_t small = 0;
            for (size_t i = 1; i < P.idle.size(); i++)
                if (P.idle[i].second < P.idle[small].second) small = i;
            (void)hipFree(P.idle[small].first);
            P.idle.erase(P.idle.begin() + (long)small);
        }
    }
    void release(int dev)
    {
        std::lock_guard<std::mutex> g(mu);
        per_device[dev].release();
    }
};
static TablePool g_table_pool;

// The large scratch buffers of the counting pipeline (record and key streams: gigabytes per context), kept across contexts
// the same way: memory handed back with hipFree is reclaimed by the driver lazily and in bulk -- every third fresh context
// of a process that counted configs[1] stalled 1.5 - 4 s in its first launch while that happened.  Blocks of 64 MB or more
// go to this pool (DevPool's limits: best fit with at most 2x slack, 48 idle blocks); an allocation that fails for lack of
// memory empties both pools and tries again.
struct ScratchPool {
    static constexpr size_t MIN_BYTES = 64ull << 20;
    std::mutex mu;
    std::map<int, DevPool> per_device;
    bool on = [] { const char *e = getenv("MC_SCRATCH_POOL"); return !(e && !strcmp(e, "0")); }();
    hipError_t get(int dev, size_t bytes, void **out, size_t *got)
    {
        if (!on || bytes < MIN_BYTES) { *got = bytes; return hipMalloc(out, std::max<size_t>(bytes, 1)); }
        std::lock_guard<std::mutex> g(mu);
        return per_device[dev].get(bytes, out, got);
    }
    void put(int dev, void *p, size_t bytes)
    {
        if (!p) return;
        if (!on || bytes < MIN_BYTES) { (void)hipFree(p); return; }
        std::lock_guard<std::mutex> g(mu);
        DevPool &P = per_device[dev];
        P.put(p, bytes);
        for (;;) {  // at most max_idle bytes stay idle (MC_SCRATCH_POOL_GB, default 64): the largest blocks go first
            size_t total = 0, big = 0;
            for (size_t i = 0; i < P.idle.size(); i++) {
                total += P.idle[i].second;
                if (P.idle[i].second > P.idle[big].second) big = i;
            }
            if (total <= max_idle || P.idle.empty()) break;
            (void)hipFree(P.idle[big].first);
            P.idle.erase(P.idle.begin() + (long)big);
        }
    }
    void release(int dev)
    {
        std::lock_guard<std::mutex> g(mu);
        per_device[dev].release();
    }
    size_t max_idle = [] { const char *e = getenv("MC_SCRATCH_POOL_GB"); return (size_t)((e ? atof(e) : 64.0) * 1e9); }();
};
static ScratchPool g_scratch_pool;

// device buffers of one BFS job, kept in the context between calls
struct BfsJobBuffers {
    BfsState S{};
    uint64_t *d_seed_hi = nullptr, *d_seed_lo = nullptr;
    uint64_t seed_cap = 0;
    BfsJobBuffers() = default;
    BfsJobBuffers(const BfsJobBuffers &) = delete;
    BfsJobBuffers &operator=(const BfsJobBuffers &) = delete;
    void free_arrays()
    {
        (void)hipFree(S.hi); (void)hipFree(S.lo); (void)hipFree(S.dist); (void)hipFree(S.cov);
        (void)hipFree(S.flags); (void)hipFree(S.vis);
        S.hi = S.lo = nullptr; S.dist = nullptr; S.cov = nullptr; S.flags = nullptr; S.vis = nullptr;
    }
    ~BfsJobBuffers()
    {
        free_arrays();
        (void)hipFree(S.ctl);
        (void)hipFree(S.path);
        (void)hipFree(S.box);
        (void)hipFree(S.trace);
        (void)hipFree(d_seed_hi);
        (void)hipFree(d_seed_lo);
    }
};



struct mc_ctx {
    mc_config cfg{};
    std::mutex mu;
    std::string err;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the second scatter level of one piece of a batch runs here, next to the first level of the next piece (add_reads_partitioned)
    hipStream_t pipe_stream = nullptr;
    hipEvent_t ev_piece[8] = {}, ev_p2 = nullptr;
    hipEvent_t ev_seq[16] = {};  // behind every kernel of a per-window run in pieces (add_reads_partitioned_once)
    hipEvent_t ev_t[4] = {};  // P1 start, P1 end, P2 end, P3 end of a pipeline run enqueued without a host round trip in between
    // The read store: the packed bases of every read this context was given since the last mc_clear, batch after
    // batch (each starting on a word boundary).  Table slots point into it (kmer_device.h ptr_encode) and the BFS
    // reads its look-ahead from it.  rs_from: a BFS-only context (mc_solid_from_pairs_dev) borrows the store of the
    // context that counted this rank's reads (mc_share_read_store).
    uint64_t *rs_words = nullptr;
    uint64_t rs_cap_words = 0, rs_bases = 0;
    bool rs_enabled = true;
    // mc_set_read_pointers' other modes.  rs_virtual: this context keeps no store, but the reads it extracts or counts sit in ANOTHER
    // context's store (the walking rank's, mc_read_store_import_dev) from rs_bases on: their pointers are worked out as if they were
    // appended here, and only the bookkeeping is done.  all_ptrs: every record mc_add_superkmers*_dev is handed carries a pointer.
    // rs_hi_bases: how far the store is filled by imports (a real store; never below what rs_bases has reached).
    bool rs_virtual = false, all_ptrs = false;
    bool no_vleaf = false;  // (add_reads_partitioned: this batch's sample asked for a table beyond 2^19 regions; the first level runs again in the two-array form)
    uint64_t rs_hi_bases = 0;
    uint64_t rs_end() const { return std::max(rs_bases, rs_hi_bases); }
    mc_ctx *rs_from = nullptr;
    uint64_t cur_ptr_base = ~0ull;  // store position of base 0 of the batch being counted (~0: no pointers for it)
    uint32_t ptr_tries = 1;         // count_pipeline.h k_p3_merge: 16 while the records of other ranks (no pointers) are merged
    // The solid list P3 left in pipe.a_recs (count_pipeline.h P3Emit): valid for threshold cov_hint until anything
    // else touches the table or the pipeline buffers.
    bool solid_list_fresh = false;
    uint32_t solid_list_segs = 0;
    uint64_t solid_list_segcap = 0;

    // table
    Slot *slots = nullptr;
    size_t slots_bytes = 0;    // size of the block `slots` sits in (it may come from g_table_pool, a little larger than asked for)
    uint64_t n_regions = 0;    // regions of 2^sb slots: a power of two, or (minimizer-bin tables of >= 512 regions) any multiple of 512
    uint32_t rb = 0, sb = MC_REGION_LG;  // rb = log2(n_regions) when that is a power of two (else its floor)
    unsigned long long *d_ctr = nullptr;  // [0] n_used, [1] empty_cnt, [2] scratch counter, [3] solid n_used, [4..5] read summary, [6] keys with count >= cov_hint,
                                          // [7] parked additions, [8] the `fatal` flag (d_fatal points here), [10..13] read summary of a device batch
    unsigned long long *h_scratch = nullptr;  // 32 pinned words: where the small device-to-host copies land (a copy into pageable memory is staged)
    uint32_t *d_fatal = nullptr;
    uint64_t n_used_host = 0;
    bool finalized = false;

    // "solid" table (kmer_device.h): only the keys with count >= solid_cov, sparse.
    // Built lazily by mc_bfs_batch; the BFS never touches the counting table.
    Slot *solid = nullptr;
    uint32_t solid_lg = 0;
    int solid_cov = -1;  // -1: not built / stale
    bool solid_external = false;  // built by mc_solid_from_pairs_dev, not from this context's counting table
    bool bfs_direct = true;       // the BFS walks the counting table itself instead of a solid copy (MC_BFS_DIRECT=0: copy)
    bool solid_is_table = false;  // ... and does so now (solid_view)
    bool want_list = false;       // the merge kernel lists the solid keys (count_pipeline.h P3Emit): for exports, and for the copy
    int solid_external_cov = -1;
    double pending_solid_ms = 0;
    uint64_t n_solid = 0;
    // mc_set_coverage_hint: the merge kernel of the counting pipeline keeps d_ctr[6] = #keys with
    // count >= cov_hint, so ensure_solid needs no counting sweep.  Only additions that go through that
    // kernel maintain it; any other kind of addition clears solid_tracked until mc_clear.
    int cov_hint = 0;
    bool solid_tracked = true;

    mc_stats st{};
    std::vector<std::unique_ptr<BfsJobBuffers>> bfs_pool;

    char *pin[16] = {};                // pinned staging buffers of h2d_fast, made on first use
    DevPool tok_pool;                  // scratch of the device tokeniser
    int64_t extract_in_store = -1;     // mc_group: the reads the next mc_extract_*_dev call is given sit in the read store already, from this word on (consumed by that call)
    // Several GPUs: the walk of this context reads the counting tables of all ranks where they are (mc_shard_attach; mc_group
    // with peer access).  h_shards[i] describes rank i's table (this context's own among them), d_shards is the same array in
    // device memory, ipc_opened the mappings of other processes' tables this context holds.
    std::vector<ShardRef> h_shards;
    ShardRef *d_shards = nullptr;
    uint32_t shard_self = 0;
    bool shards_dropped = false;  // an attachment was dropped because the table changed: the next walk must not quietly read this rank's table alone
    int shard_owner_mm_k = 0;
    // A walker attaches the same tables step after step (bench.py: every step; the CLI: every batch of seeds): a mapping stays
    // open while its handle keeps coming (in_use: part of the current attachment) and is closed when it has not for a while.
    struct IpcMap { hipIpcMemHandle_t h; void *p; bool in_use; uint64_t addr, bytes; };  // (addr, bytes: the block in its owner's process -- a handle alone may be handed out again for another block)
    std::vector<IpcMap> ipc_opened;
    bool extract_by_minimizer = false; // mc_group: the next mc_extract_keys_dev call deals the keys to the owners of their minimizers (sk_owner), as the group's records are dealt (consumed by that call)
    uint4 *d_ovf_tmp = nullptr;        // pipe_drain_handed_on: the list moved aside while it is drained
    uint32_t *d_ovf_leaf_tmp = nullptr;
    uint64_t ovf_tmp_cap = 0, ovf_leaf_tmp_cap = 0;
    bool rs_copy_pending = false;      // a batch is on its way into the read store on pipe_stream (rs_append)
    // mc_bfs_batch: job states + seeds go up in one copy (pinned h_bfs_stage -> d_bfs_stage), results come back packed
    // (d_bfs_pack: a header block and the jobs' arrays back to back; h_bfs_hdr: the headers, pinned)
    char *h_bfs_stage = nullptr, *d_bfs_stage = nullptr, *d_bfs_pack = nullptr, *h_bfs_hdr = nullptr;
    uint64_t bfs_stage_cap = 0, bfs_pack_cap = 0, bfs_hdr_cap = 0;
    std::mutex pin_mu;                 // the pinned buffers serve one copy at a time
    hipStream_t pin_stream[8] = {};
    int mm_k = 0;        // != 0 (= k): regions are minimizer bins and reads are counted as super-k-mers (kmer_device.h)
    bool virgin = true;  // the table holds no key and its memory is not initialised yet
    int count_path = 0;  // 0 auto, 1 direct (atomics), 2 partitioned; MC_COUNT_PATH=direct|partition overrides
    bool sk_form = false;  // reads of this context can travel as super-k-mer records (set once; mm_k may be given up later)
    // scratch of the partitioned counting pipeline, kept between calls
    struct Pipe {
        uint64_t *a_keys = nullptr, *b_keys = nullptr, *spill_keys = nullptr;
        uint32_t *a_hints = nullptr, *b_hints = nullptr, *spill_hints = nullptr, *tile_first = nullptr;
        uint4 *a_recs = nullptr, *b_recs = nullptr, *spill_recs = nullptr;  // super-k-mer form (their bin words use a_hints / b_hints)
        uint64_t a_recs_cap = 0, b_recs_cap = 0, spill_recs_cap = 0;
        uint32_t *solid_cursors = nullptr;  // leaf fill levels of the solid-table build (minimizer-bin tables)
        uint64_t solid_cursors_cap = 0;
        uint32_t *emit_counts = nullptr;  // fill levels of the solid list's segments (one per P3 workgroup)
        uint32_t *cursors1 = nullptr, *seg_counts1 = nullptr, *cursors2 = nullptr, *leaf_state = nullptr, *leaf_new = nullptr, *flags = nullptr;  // cursors1: owner cursors (multi-GPU split); cursors2: leaf fill levels; flags: [0] spill lost, [1] any leaf failed, [2] a segment of the solid list overflowed
        unsigned long long *spill_count = nullptr;
        uint64_t a_cap = 0, b_cap = 0, spill_cap = 0, tiles1_cap = 0, leaves_cap = 0, segs1_cap = 0, a_hints_cap = 0, b_hints_cap = 0, cursors2_cap = 0;
        // the binned exchange (mc_extract_superkmers_binned_dev / mc_add_superkmers_binned_dev): every first-level workgroup's row of
        // (owner, fine bucket) counters; where every cell starts in the packed stream; where every listed segment of the second level starts
        uint32_t *skb_rows = nullptr;
        unsigned long long *skb_cell_start = nullptr, *skb_seg_start = nullptr, *skb_small = nullptr;  // skb_small: owner offsets / windows / part offsets / a flag
        uint64_t skb_rows_cap = 0, skb_cell_start_cap = 0, skb_seg_start_cap = 0, skb_small_cap = 0;
        void release(int dev)
        {   // (the large streams go to g_scratch_pool: ensure_buf took them from there)
            g_scratch_pool.put(dev, skb_rows, skb_rows_cap * 4); g_scratch_pool.put(dev, skb_cell_start, skb_cell_start_cap * 8);
            g_scratch_pool.put(dev, skb_seg_start, skb_seg_start_cap * 8); g_scratch_pool.put(dev, skb_small, skb_small_cap * 8);
            g_scratch_pool.put(dev, a_keys, a_cap * 8); g_scratch_pool.put(dev, b_keys, b_cap * 8); (void)hipFree(spill_keys);
            g_scratch_pool.put(dev, a_recs, a_recs_cap * sizeof(uint4)); g_scratch_pool.put(dev, b_recs, b_recs_cap * sizeof(uint4));
            (void)hipFree(spill_recs); (void)hipFree(solid_cursors); (void)hipFree(emit_counts);
            g_scratch_pool.put(dev, a_hints, a_hints_cap * 4); g_scratch_pool.put(dev, b_hints, b_hints_cap * 4); (void)hipFree(spill_hints); (void)hipFree(tile_first);
            (void)hipFree(cursors1); (void)hipFree(seg_counts1); (void)hipFree(cursors2); (void)hipFree(leaf_state); (void)hipFree(leaf_new); (void)hipFree(flags);
            // (spill_count lives behind flags, in the same allocation)
            *this = Pipe{};
        }
    } pipe;

    // Hash keys in minimizer bins: the join of the table's keys by key (dup_check.h) and what it found.
    struct Dup {
        // level 1: the merge kernel's (or the sweep's) key streams
        uint64_t *l1_keys = nullptr;
        uint64_t l1_words = 0;
        uint32_t *l1_counts = nullptr;
        uint64_t l1_counts_cap = 0;
        uint32_t l1_nseg = 0;
        uint64_t l1_cap = 0;
        bool l1_armed = false;       // the last long run's merge kernel collected into the streams (and nothing else has touched the table since)
        double expected_keys = 0;    // what the sample of the batch said the table will hold (a context without a hint)
        // levels 2 and 3
        uint64_t *l2_own = nullptr;  // where the pipeline's idle streams are too small to serve
        uint64_t l2_own_words = 0;
        uint32_t *l2_counts = nullptr;
        uint64_t l2_counts_cap = 0;
        uint32_t *flags = nullptr;            // [0] a level-1 segment overflowed, [1] a level-2 stream did
        unsigned long long *ctr = nullptr;    // [0] keys listed, [1] distinct keys in the set, [2] slots noted
        unsigned long long *list = nullptr;   // the listed keys (LIST_CAP)
        static constexpr uint64_t LIST_CAP = 1u << 16;
        // what the fix-up left: the set of keys held by more than one slot, their slots with their own counts
        DupSet set{nullptr, nullptr, nullptr, 0, nullptr};
        uint64_t set_slots = 0;
        DupTwin *tw = nullptr;
        uint64_t tw_cap = 0, n_tw = 0, n_keys = 0;
        bool merged = false;         // the noted slots hold their keys' sums now (else: their own counts)
        long long solid_delta = 0;   // what merging added to d_ctr[6] (keys at the coverage hint)
        bool checked = false;        // the table as it is has been joined
        DupL2 l2{nullptr, nullptr, 0, nullptr, 0, 0, 0, nullptr};  // the table's keys by key, as the last join left them (valid while l2_valid)
        bool l2_valid = false;
        // the check of a walk's "absent" look-ups by key (dup_check.h PhantomQ): queries, their order by sub-bucket, the hits
        unsigned long long *pq_mem = nullptr;
        uint64_t pq_cap = 0;
        uint32_t *pq_groups = nullptr;  // [G]: the first query of every sub-bucket's list
        uint64_t pq_groups_cap = 0;
    } dup;
    DupL1 dup_l1_view() const
    {
        if (!dup.l1_armed) return DupL1{nullptr, nullptr, 0, 0, nullptr};
        return DupL1{dup.l1_keys, dup.l1_counts, dup.l1_nseg, dup.l1_cap, dup.flags};
    }
    DupSet dup_filter() const { return dup.merged && dup.n_tw ? dup.set : DupSet{nullptr, nullptr, nullptr, 0, nullptr}; }

    uint64_t n_slots() const { return n_regions << sb; }
    SolidView solid_view() const
    {
        SolidView t;
        t.slots = solid;
        t.shift = 64 - solid_lg;
        t.rmask = (1u << 11) - 1;  // SOLID_REGION - 1
        t.mm_k = 0;
        t.n_regions = 0;
        if (solid_is_table) {  // the counting table itself
            t.slots = slots;
            t.shift = 64 - (rb + sb);
            t.rmask = (1u << sb) - 1;
            t.mm_k = mm_k;
            t.n_regions = (uint32_t)n_regions;
        }
        t.empty_cnt = d_ctr + 1;
        t.fatal = d_fatal;
        const mc_ctx *rs = rs_from ? rs_from : this;
        t.reads = rs->rs_end() && rs->rs_words ? rs->rs_words : nullptr;
        t.reads_bases = rs->rs_words ? rs->rs_end() : 0;
        t.shards = d_shards;
        t.n_shards = d_shards ? (uint32_t)h_shards.size() : 0u;
        t.owner_mm_k = shard_owner_mm_k;
        return t;
    }
    TableView view() const
    {
        TableView t;
        t.slots = slots;
        t.shift = 64 - (rb + sb);  // (hash-prefix regions: n_regions is a power of two)
        t.rmask = (1u << sb) - 1;
        t.n_regions = (uint32_t)n_regions;
        t.mm_k = mm_k;
        t.n_used = d_ctr;
        t.empty_cnt = d_ctr + 1;
        t.fatal = d_fatal;
        t.ovf = d_ovf;
        t.ovf_n = d_ctr + 7;
        t.ovf_cap = d_ovf ? OVF_CAP : 0;
        t.ovf_leaf = d_ovf_leaf;
        return t;
    }
    static constexpr uint64_t OVF_CAP = 1ull << 22;
    uint4 *d_ovf = nullptr;  // TableView::ovf
    uint32_t *d_ovf_leaf = nullptr;  // TableView::ovf_leaf
};

static thread_local std::string g_create_err;

static int fail(mc_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail((c), e_ == hipErrorOutOfMemory ? MC_ENOMEM : MC_EHIP, "%s: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

template <typename T>
struct DevBuf {  // RAII device buffer for temporaries
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
    hipError_t alloc(size_t n) { reset(); return hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T)); }
};

// ------------------------------------------------------------------------------------------ kernels: table

__global__ void k_fill_empty(Slot *slots, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 v;
        v.x = 0xFFFFFFFFu; v.y = 0xFFFFFFFFu; v.z = 0; v.w = 0;
        *reinterpret_cast<uint4 *>(slots + i) = v;
    }
}

// K2/K3: one wave per read, one lane per window; replaces the hot loop
// src/io/IOUtils.java:201-214 (ShortKmer.kmersOf + toLong + addAndBound) and
// src/io/LargeKIOUtils.java:41-54 (hasher.hash(dna, i, i+k) + addAndBound).
template <int MODE>
__global__ void __launch_bounds__(256) k_count_reads(const uint64_t *__restrict__ words,
                                                     const uint64_t *__restrict__ offsets, uint64_t r_begin,
                                                     uint64_t r_end, int k, TableView t, uint64_t ptr_base, uint32_t thr)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long n_new = 0;
    for (uint64_t r = r_begin + wave; r < r_end; r += n_waves) {
        const uint64_t b = offsets[r], e = offsets[r + 1];
        if (e - b < (uint64_t)k) continue;
        const uint64_t nwin = e - b - (uint64_t)k + 1;
        for (uint64_t w = lane; w < nwin; w += 64) {
            const Kmer v = extract_kmer(words, b + w, k);
            const uint64_t key = (uint64_t)key_of<MODE>(v, k);
            // (the inserter leaves its place in the read store; the occurrence ptr_pick names replaces it)
            n_new += table_add(t, key, 1u, ptr_base == ~0ull ? 0u : ptr_encode(ptr_base + b + w), nullptr, ptr_pick(key, thr >= 2 ? 1u : 0u, thr), ptr_pick_late(key, thr >= 2 ? 1u : 0u));
        }
    }
    wave_add_ull(t.n_used, n_new);
}

__global__ void k_add_keys(const int64_t *__restrict__ keys, uint64_t n, TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_new = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        n_new += table_add(t, (uint64_t)keys[i], 1u);
    wave_add_ull(t.n_used, n_new);
}

__global__ void k_add_pairs(const int64_t *__restrict__ keys, const int16_t *__restrict__ counts,
                            const uint32_t *__restrict__ hints, uint64_t n, TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_new = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (counts[i] > 0) n_new += table_add(t, (uint64_t)keys[i], (uint32_t)counts[i], hints ? hints[i] : 0u);
    wave_add_ull(t.n_used, n_new);
}

// the parked additions of TableView::ovf, once the table has been enlarged
// entry_leaf / leaf_state (may be null): an entry the merge kernels left (count_pipeline.h ovf_push) is skipped when its
// leaf was not committed -- that leaf is merged again and leaves its entries again
// region_of_leaf: hash keys in minimizer bins (count_long.h) -- the key does not say where it lives, the leaf it was handed on from
// does (one region a leaf; every entry of such a list comes from a leaf)
__global__ void k_add_parked(const uint4 *__restrict__ list, uint64_t n, TableView t, const uint32_t *__restrict__ entry_leaf = nullptr,
                             const uint32_t *__restrict__ leaf_state = nullptr, int region_of_leaf = 0, DupL1 d1 = DupL1{nullptr, nullptr, 0, 0, nullptr})
{   // d1: keys that are new to a table of hash keys in minimizer bins also go to the key streams of dup_check.h
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_new = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t lf = 0xFFFFFFFFu;
        if (entry_leaf && leaf_state) {
            lf = entry_leaf[i];
            if (lf != 0xFFFFFFFFu && leaf_state[lf] == 0) continue;
        }
        const uint4 e = list[i];
        const uint64_t key = ((uint64_t)e.y << 32) | e.x;
        if (region_of_leaf && lf != 0xFFFFFFFFu) {
            const uint32_t fresh = table_add_at(t, ((uint64_t)lf << MC_REGION_LG) | sk_home(key), key, e.z, e.w);
            n_new += fresh;
            if (fresh) dup_l1_extra(d1, key);
        } else n_new += table_add(t, key, e.z, e.w);
    }
    wave_add_ull(t.n_used, n_new);
}

// table rebuild into a larger table (the reference's enlargeAndRehash,
// itmo!/structures/map/Long2ShortHashMap.java:191-214, done for the whole table at once)
__global__ void k_rehash(const Slot *__restrict__ old_slots, uint64_t n_old, TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_new = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_old; i += stride) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(old_slots + i);
        const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
        if (key != EMPTY_KEY) n_new += table_add(t, key, raw.z, raw.w);
    }
    wave_add_ull(t.n_used, n_new);
}

// The solid table's regions (SOLID_REGION slots, 32 KB)
constexpr uint32_t SOLID_SB = 11, SOLID_REGION = 1u << SOLID_SB;
// (Round 3 also had k_build_solid_regions: a workgroup per solid region, fed from the counting regions whose hash prefix it
// shares.  It took every key to sit in its home counting region -- no longer so since additions that find their stretch full
// move on to the next region, kmer_device.h TABLE_CHAIN -- and lost the others: every solid table is now built
// by the sweep below, which asks no such thing.)

// K6: the solid table indexes by the key's own hash, so its regions draw from all over a counting table organised by
// minimizer bins (mc_ctx::mm_k) -- and from next door in one organised by hash prefixes, where a key may have been handed on.  The solid
// entries (a few % of the occurrences) therefore take the same route as the counting records:
// k_solid_emit sweeps the counting table and appends every entry with count >= min_cov to the level-1
// bucket of its solid slot (256 workgroups, each its own segment of every bucket), k_sk2_scatter
// splits the buckets into one leaf per solid region, k_solid_from_leaves assembles each region in
// LDS.  No fill pass, no global atomics.
__global__ void __launch_bounds__(PT_THREADS) k_solid_emit(const Slot *__restrict__ slots, uint64_t n_slots, int min_cov, uint32_t np1,
                                                            uint32_t *seg_counts, uint64_t cap, uint4 *out_recs, uint32_t *out_bins,
                                                            SkSpill sp)
{
    __shared__ SkCursors C;
    const uint32_t tid = threadIdx.x, n_buckets = np1;
    if (tid < PT_MAX_LEAVES2) { C.wcur[tid] = 0; C.cnt[tid] = 0; }
    __syncthreads();
    // Rows of 4 slots (64 bytes: one memory sector) are dealt to the workgroups (= segments) round-robin, as k_solid_emit_pairs
    // deals its rows: a table whose regions are hash prefixes is in the order of the very hash that picks the bucket here, and
    // whole tiles of it would put a bucket's entries into a few segments only (the segment capacities count on every segment
    // getting its share of every bucket).
    constexpr uint32_t WAVES = PT_THREADS / 64, ROW = 4, ROWS_PER_WAVE = 64 / ROW;
    const uint64_t n_rows = (n_slots + ROW - 1) / ROW;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    for (uint64_t t0 = 0; t0 * WAVES * ROWS_PER_WAVE * gridDim.x < n_rows; t0 += PT_ITEMS) {
        uint4 raws[PT_ITEMS];
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) {
            const uint64_t local = ((t0 + (uint64_t)j) * WAVES + wave) * ROWS_PER_WAVE + lane / ROW;  // this workgroup's row number
            const uint64_t i = (local * gridDim.x + blockIdx.x) * ROW + lane % ROW;
            raws[j] = i < n_slots ? *reinterpret_cast<const uint4 *>(slots + i) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) {
            const uint64_t key = ((uint64_t)raws[j].y << 32) | raws[j].x;
            if (key == EMPTY_KEY) continue;
            const uint32_t c = raws[j].z > 32767u ? 32767u : raws[j].z;
            if ((int)c < min_cov) continue;
            raws[j].z = c;
            const uint32_t bin = (uint32_t)(fmix64(key) >> 32);
            sk_emit(C, mulhi32(bin, np1), raws[j], bin, cap, (uint64_t)blockIdx.x * cap, (uint64_t)PT_SEGMENTS * cap, out_recs, out_bins, sp);
        }
        __syncthreads();
        if (tid < n_buckets) { C.wcur[tid] += C.cnt[tid]; C.cnt[tid] = 0; }
        __syncthreads();
    }
    if (tid < n_buckets) seg_counts[(uint64_t)tid * PT_SEGMENTS + blockIdx.x] = min(C.wcur[tid], (uint32_t)cap);
}

// The same first step for solid entries that arrive as (key, count, hint) arrays -- the thresholded shards of the
// other ranks (mc_solid_from_pairs_dev) -- instead of sitting in this context's counting table.
__global__ void __launch_bounds__(PT_THREADS) k_solid_emit_pairs(const int64_t *__restrict__ keys, const int16_t *__restrict__ counts,
                                                                  const uint32_t *__restrict__ hints, uint64_t n, int min_cov, uint32_t np1,
                                                                  uint32_t *seg_counts, uint64_t cap, uint4 *out_recs, uint32_t *out_bins,
                                                                  SkSpill sp, unsigned long long *empty_cnt)
{
    __shared__ SkCursors C;
    const uint32_t tid = threadIdx.x, n_buckets = np1;
    if (tid < PT_MAX_LEAVES2) { C.wcur[tid] = 0; C.cnt[tid] = 0; }
    __syncthreads();
    // Rows of 8 entries (64 bytes of keys: one memory sector) are dealt to the workgroups (= segments) round-robin.  An
    // export comes in table order, which for hash-prefix tables is the order of the very hash that picks the bucket
    // -- one sorted run per gathered shard -- and short arrays must not land in a few segments either: the segment
    // capacities count on each segment getting its share of every bucket, to within a row per shard.
    constexpr uint32_t WAVES = PT_THREADS / 64, ROW = 8, ROWS_PER_WAVE = 64 / ROW;
    const uint64_t n_rows = (n + ROW - 1) / ROW;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    for (uint64_t t0 = 0; t0 * WAVES * ROWS_PER_WAVE * gridDim.x < n_rows; t0 += PT_ITEMS) {
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) {
            const uint64_t local = ((t0 + (uint64_t)j) * WAVES + wave) * ROWS_PER_WAVE + lane / ROW;  // this workgroup's row number
            const uint64_t i = (local * gridDim.x + blockIdx.x) * ROW + lane % ROW;
            if (i >= n) continue;
            const int c = counts[i];
            if (c < min_cov || c < 0) continue;
            const uint64_t key = (uint64_t)keys[i];
            if (key == EMPTY_KEY) { atomicAdd(empty_cnt, (unsigned long long)c); continue; }  // kept out of band, as in the counting table
            const uint4 rec = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)c, hints ? hints[i] : 0u);
            const uint32_t bin = (uint32_t)(fmix64(key) >> 32);
            sk_emit(C, mulhi32(bin, np1), rec, bin, cap, (uint64_t)blockIdx.x * cap, (uint64_t)PT_SEGMENTS * cap, out_recs, out_bins, sp);
        }
        __syncthreads();
        if (tid < n_buckets) { C.wcur[tid] += C.cnt[tid]; C.cnt[tid] = 0; }
        __syncthreads();
    }
    if (tid < n_buckets) seg_counts[(uint64_t)tid * PT_SEGMENTS + blockIdx.x] = min(C.wcur[tid], (uint32_t)cap);
}

// ... and for the solid list P3 left behind (count_pipeline.h P3Emit): nseg segments of (key, count, hint) records.
// Workgroup b takes segments b, b + gridDim.x, ...: a segment holds what one P3 workgroup saw, regions from all over
// the table, so every workgroup here feeds all buckets evenly.
__global__ void __launch_bounds__(PT_THREADS) k_solid_emit_list(const uint4 *__restrict__ list, const uint32_t *__restrict__ list_counts,
                                                                 uint32_t nseg, uint64_t list_segcap, uint32_t np1, uint32_t *seg_counts,
                                                                 uint64_t cap, uint4 *out_recs, uint32_t *out_bins, SkSpill sp)
{
    __shared__ SkCursors C;
    const uint32_t tid = threadIdx.x, n_buckets = np1;
    if (tid < PT_MAX_LEAVES2) { C.wcur[tid] = 0; C.cnt[tid] = 0; }
    __syncthreads();
    for (uint32_t sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
        const uint32_t n = list_counts[sg];
        const uint4 *src = list + (uint64_t)sg * list_segcap;
        for (uint32_t i0 = 0; i0 < n; i0 += PT_TILE) {
            uint4 raws[PT_ITEMS];
#pragma unroll
            for (int j = 0; j < PT_ITEMS; j++) {
                const uint32_t i = i0 + tid + (uint32_t)j * PT_THREADS;
                raws[j] = i < n ? src[i] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < PT_ITEMS; j++) {
                const uint64_t key = ((uint64_t)raws[j].y << 32) | raws[j].x;
                if (key == EMPTY_KEY) continue;  // (padding of the last tile; the table never holds this key)
                const uint32_t bin = (uint32_t)(fmix64(key) >> 32);
                sk_emit(C, mulhi32(bin, np1), raws[j], bin, cap, (uint64_t)blockIdx.x * cap, (uint64_t)PT_SEGMENTS * cap, out_recs, out_bins, sp);
            }
            __syncthreads();
            if (tid < n_buckets) { C.wcur[tid] += C.cnt[tid]; C.cnt[tid] = 0; }
            __syncthreads();
        }
    }
    if (tid < n_buckets) seg_counts[(uint64_t)tid * PT_SEGMENTS + blockIdx.x] = min(C.wcur[tid], (uint32_t)cap);
}

__global__ void k_count_pairs(const int16_t *__restrict__ counts, const int64_t *__restrict__ keys, uint64_t n, int min_cov,
                              unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        m += counts[i] >= min_cov && counts[i] >= 0 && (uint64_t)keys[i] != EMPTY_KEY;
    wave_add_ull(out, m);
}

// one workgroup per solid region = one leaf (nseg segments of capacity seg_cap, as for k_p3_merge)
__global__ void __launch_bounds__(512) k_solid_from_leaves(const uint4 *__restrict__ leaf_recs, const uint32_t *__restrict__ seg_counts,
                                                           uint64_t seg_cap, uint32_t nseg, SolidView solid, uint32_t solid_lg)
{
    __shared__ Slot R[SOLID_REGION];
    __shared__ uint32_t overflow;
    const uint32_t tid = threadIdx.x;
    const uint64_t n_regions = 1ull << (solid_lg - SOLID_SB);
    for (uint64_t Q = blockIdx.x; Q < n_regions; Q += gridDim.x) {
        for (uint32_t i = tid; i < SOLID_REGION; i += 512) {
            R[i].key = EMPTY_KEY; R[i].count = 0; R[i].aux = 0;
        }
        if (tid == 0) overflow = 0;
        __syncthreads();
        for (uint32_t sgm = 0; sgm < nseg; sgm++) {
            const uint32_t n = min(seg_counts[Q * nseg + sgm], (uint32_t)seg_cap);
            const uint4 *recs = leaf_recs + (Q * nseg + sgm) * seg_cap;
            for (uint32_t i = tid; i < n; i += 512) {
                const uint4 raw = recs[i];
                const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
                uint32_t s = (uint32_t)(fmix64(key) >> (64 - solid_lg)) & (SOLID_REGION - 1);
                bool done = false;
                for (uint32_t probe = 0; probe < (SOLID_REGION < TABLE_MAX_PROBES ? SOLID_REGION : TABLE_MAX_PROBES); probe++) {  // (where a lookup gives the region up: kmer_device.h solid_probe_from)
                    if (atomicCAS(reinterpret_cast<unsigned long long *>(&R[s].key), (unsigned long long)EMPTY_KEY,
                                  (unsigned long long)key) == EMPTY_KEY) {
                        R[s].count = raw.z;
                        R[s].aux = raw.w;
                        done = true;
                        break;
                    }
                    s = (s + 1) & (SOLID_REGION - 1);
                }
                if (!done) atomicExch(&overflow, 1u);
            }
        }
        __syncthreads();
        if (overflow && tid == 0) atomicExch(solid.fatal, 1u);
        uint4 *dst = reinterpret_cast<uint4 *>(solid.slots + Q * SOLID_REGION);
        const uint4 *src = reinterpret_cast<const uint4 *>(R);
        for (uint32_t i = tid; i < SOLID_REGION; i += 512) dst[i] = src[i];
        __syncthreads();
    }
}

// windows = sum over reads of max(0, len - k + 1); also checks that the offsets never decrease
__global__ void k_reads_summary(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t k,
                                unsigned long long *out /* [0] windows, [1] violations; ends != 0: [2] first offset, [3] last offset */, int ends = 0)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long w = 0, bad = 0;
    // two reads a thread and step (three offsets, the middle one shared): half the loads of a read a thread
    for (uint64_t i = 2 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x); i < n_reads; i += 2 * stride) {
        const uint64_t a = offsets[i], b = offsets[i + 1];
        if (b < a) bad++;
        else if (b - a >= k) w += b - a - k + 1;
        if (i + 1 < n_reads) {
            const uint64_t c2 = offsets[i + 2];
            if (c2 < b) bad++;
            else if (c2 - b >= k) w += c2 - b - k + 1;
        }
    }
    // one pair of atomics a WORKGROUP, at most 512 of them: atomics on one address take their turns at ~7 ns each, and a pair per
    // wave of 2048 workgroups (16 000 of them) was 120 us for 10 M reads -- four times what reading the offsets takes
    __shared__ unsigned long long part[2][4];
    for (int o = 32; o > 0; o >>= 1) { w += __shfl_down(w, o); bad += __shfl_down(bad, o); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = w; part[1][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tw = 0, tb = 0;
        for (unsigned i = 0; i < (blockDim.x + 63) / 64 && i < 4; i++) { tw += part[0][i]; tb += part[1][i]; }
        if (tw) atomicAdd(out, tw);
        if (tb) atomicAdd(out + 1, tb);
    }
    if (ends && blockIdx.x == 0 && threadIdx.x == 0) { out[2] = offsets[0]; out[3] = offsets[n_reads]; }
}

// K4: BigLong2ShortHashMap.get for a batch of keys
__global__ void k_get(const int64_t *__restrict__ keys, uint64_t n, int16_t *__restrict__ out, TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (int16_t)table_get(t, (uint64_t)keys[i]);
}

// ... for a table of hash keys in minimizer bins (count_long.h), where a bare key does not say which region it lives in: the
// QUERIES go into a small set (qk: key, ~0 = free; qi: the first query that asked for it), the whole table is swept once and
// every slot whose key is in the set answers, queries that repeat a key copy the answer.  A sweep per call -- 20 GB in 5 ms --
// instead of a rebuild of the table that would end the long-record form for good (and cannot be afforded where the table is
// half the device's memory).
__global__ void k_getq_build(const int64_t *__restrict__ keys, uint64_t n, unsigned long long *qk, uint32_t *qi, uint64_t mask)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t key = (uint64_t)keys[i];
        if (key == EMPTY_KEY) continue;
        for (uint64_t s = fmix64(key) & mask;; s = (s + 1) & mask) {
            const unsigned long long old = atomicCAS(&qk[s], ~0ull, (unsigned long long)key);
            if (old == ~0ull) { qi[s] = (uint32_t)i; break; }
            if (old == key) break;
        }
    }
}
__global__ void k_getq_sweep(const Slot *__restrict__ slots, uint64_t n_slots, const unsigned long long *__restrict__ qk, const uint32_t *__restrict__ qi,
                             uint64_t mask, int16_t *__restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(slots + i);
        const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
        if (key == EMPTY_KEY) continue;
        for (uint64_t s = fmix64(key) & mask;; s = (s + 1) & mask) {
            const unsigned long long q = qk[s];
            if (q == ~0ull) break;
            if (q == key) { out[qi[s]] = (int16_t)(raw.z > 32767u ? 32767u : raw.z); break; }
        }
    }
}
__global__ void k_getq_finish(const int64_t *__restrict__ keys, uint64_t n, const unsigned long long *__restrict__ qk, const uint32_t *__restrict__ qi,
                              uint64_t mask, int16_t *__restrict__ out, const unsigned long long *__restrict__ empty_cnt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t key = (uint64_t)keys[i];
        if (key == EMPTY_KEY) {
            const unsigned long long c = *empty_cnt;
            out[i] = (int16_t)(c == 0 ? -1 : (c > 32767ull ? 32767 : (int)c));
            continue;
        }
        for (uint64_t s = fmix64(key) & mask;; s = (s + 1) & mask)
            if (qk[s] == key) {
                if (qi[s] != (uint32_t)i) out[i] = out[qi[s]];  // (the first query for this key got the answer)
                break;
            }
    }
}

// K6: (key, count) pairs with count >= min_cov; with keys == nullptr only counts them.
constexpr int EXP_THREADS = 256, EXP_ITEMS = 8, EXP_TILE = EXP_THREADS * EXP_ITEMS;

__global__ void __launch_bounds__(EXP_THREADS) k_export(const Slot *__restrict__ slots, uint64_t n_slots, int min_cov,
                                                        int64_t *__restrict__ keys, int16_t *__restrict__ counts,
                                                        uint32_t *__restrict__ hints, uint64_t cap, unsigned long long *cursor,
                                                        DupSet dq = DupSet{nullptr, nullptr, nullptr, 0, nullptr})
{
    // dq: hash keys in minimizer bins -- a key that two k-mers brought to two regions is counted / listed at its first slot only
    // (dup_check.h; every slot of it holds the sum of its counters)
    // A tile of slots is compacted in LDS (one LDS cursor update per wave and row), takes ONE slice of the output
    // with a single global atomic and leaves in coalesced stores: a global atomic per wave and row would put
    // n_slots / 64 updates on one address, which alone costs ~10 ns each.
    __shared__ uint4 buf[EXP_TILE];
    __shared__ uint32_t lcur;
    __shared__ unsigned long long gbase;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    unsigned long long counted = 0;
    const uint64_t n_tiles = (n_slots + EXP_TILE - 1) / EXP_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint4 raws[EXP_ITEMS];
#pragma unroll
        for (int j = 0; j < EXP_ITEMS; j++) {
            const uint64_t i = tile * EXP_TILE + (uint64_t)j * EXP_THREADS + tid;
            raws[j] = i < n_slots ? *reinterpret_cast<const uint4 *>(slots + i) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
        }
        if (!keys) {
#pragma unroll
            for (int j = 0; j < EXP_ITEMS; j++) {
                const uint64_t key = ((uint64_t)raws[j].y << 32) | raws[j].x;
                const int c = raws[j].z > 32767u ? 32767 : (int)raws[j].z;
                counted += key != EMPTY_KEY && c >= min_cov && !dup_is_shadow(dq, key, tile * EXP_TILE + (uint64_t)j * EXP_THREADS + tid) ? 1u : 0u;
            }
            continue;
        }
        if (tid == 0) lcur = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EXP_ITEMS; j++) {
            const uint64_t key = ((uint64_t)raws[j].y << 32) | raws[j].x;
            const int c = raws[j].z > 32767u ? 32767 : (int)raws[j].z;
            const bool take = key != EMPTY_KEY && c >= min_cov && !dup_is_shadow(dq, key, tile * EXP_TILE + (uint64_t)j * EXP_THREADS + tid);
            const unsigned long long m = __ballot(take);
            if (!m) continue;
            uint32_t base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if ((int)lane == leader) base = atomicAdd(&lcur, (uint32_t)__popcll(m));
            base = __shfl(base, leader);
            if (take) buf[base + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = make_uint4(raws[j].x, raws[j].y, (uint32_t)c, raws[j].w);
        }
        __syncthreads();
        const uint32_t n_out = lcur;
        if (tid == 0 && n_out) gbase = atomicAdd(cursor, (unsigned long long)n_out);
        __syncthreads();
        for (uint32_t i = tid; i < n_out; i += EXP_THREADS) {
            const unsigned long long pos = gbase + i;
            if (pos < cap) {
                const uint4 e = buf[i];
                keys[pos] = (int64_t)(((uint64_t)e.y << 32) | e.x);
                counts[pos] = (int16_t)e.z;
                if (hints) hints[pos] = e.w;
            }
        }
        __syncthreads();
    }
    if (!keys) wave_add_ull(cursor, counted);
}

// K6 from the solid list the merge kernel left (count_pipeline.h P3Emit): a segment takes its slice of the output
// with one atomic and is copied over -- 0.8 GB read instead of a sweep of the table.
__global__ void __launch_bounds__(256) k_export_list(const uint4 *__restrict__ list, const uint32_t *__restrict__ list_counts, uint32_t nseg,
                                                     uint64_t list_segcap, int64_t *__restrict__ keys, int16_t *__restrict__ counts,
                                                     uint32_t *__restrict__ hints, uint64_t cap, unsigned long long *cursor)
{
    __shared__ unsigned long long gbase;
    for (uint32_t sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
        const uint32_t n = list_counts[sg];
        if (threadIdx.x == 0 && n) gbase = atomicAdd(cursor, (unsigned long long)n);
        __syncthreads();
        const uint4 *src = list + (uint64_t)sg * list_segcap;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned long long pos = gbase + i;
            if (pos < cap) {
                const uint4 e = src[i];
                keys[pos] = (int64_t)(((uint64_t)e.y << 32) | e.x);
                counts[pos] = (int16_t)e.z;
                if (hints) hints[pos] = e.w;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ kernels: synthetic reads

constexpr uint64_t ERR_STREAM = 0xE44044E44044E440ull;

__host__ __device__ __forceinline__ uint32_t synth_read_base(uint64_t gseed, uint64_t n_contigs, uint64_t contig_len,
                                                             uint64_t rseed, uint64_t r, uint32_t L, uint32_t err,
                                                             uint32_t j)
{
    const uint64_t x0 = splitmix(rseed, 2 * r), x1 = splitmix(rseed, 2 * r + 1);
    const uint64_t contig = (x0 >> 33) % n_contigs;
    const bool strand = x0 & 1;
    const uint64_t start = x1 % (contig_len - L + 1);
    const uint64_t g = contig * contig_len + start + (strand ? (uint64_t)(L - 1 - j) : (uint64_t)j);
    uint32_t b = (uint32_t)(splitmix(gseed, g) & 3);
    if (strand) b ^= 3;
    if (err) {
        const uint64_t e = splitmix(rseed ^ ERR_STREAM, r * (uint64_t)L + j);
        if ((uint32_t)(e % 10000) < err) b = (b + 1 + (uint32_t)((e >> 40) % 3)) & 3;
    }
    return b;
}

// one thread per output word (32 bases)
__global__ void k_synth_reads(uint64_t gseed, uint64_t n_contigs, uint64_t contig_len, uint64_t rseed,
                              uint64_t first_read, uint64_t n_reads, uint32_t L, uint32_t err, uint64_t *words,
                              uint64_t n_words, uint64_t *offsets)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_bases = n_reads * (uint64_t)L;
    for (uint64_t w = tid; w < n_words; w += stride) {
        uint64_t v = 0;
        for (uint32_t i = 0; i < 32; i++) {
            const uint64_t p = w * 32 + i;
            if (p >= n_bases) break;
            const uint64_t r = p / L;
            const uint32_t j = (uint32_t)(p - r * L);
            v |= (uint64_t)synth_read_base(gseed, n_contigs, contig_len, rseed, first_read + r, L, err, j)
                 << (62 - 2 * i);
        }
        words[w] = v;
    }
    for (uint64_t r = tid; r <= n_reads; r += stride) offsets[r] = r * (uint64_t)L;
}

// ------------------------------------------------------------------------------------------ host: table management

static int grid_for(uint64_t work_items, int block, int max_blocks = 256 * 8)
{
    uint64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

// Number of regions for a table of at least `slots` slots.  Hash-prefix regions need a power of two; minimizer bins
// are numbered by multiplication (count_pipeline.h mulhi32), so those tables come in steps of 512 regions (2 M slots,
// 32 MB), of 1024 beyond 2^19 regions -- times 2^g beyond 2^20, where a leaf of the counting pipeline covers 2^g regions.
static uint64_t regions_for(const mc_ctx *c, uint64_t slots)
{
    uint64_t want = std::max<uint64_t>((slots + (1ull << c->sb) - 1) >> c->sb, 1);
    uint64_t p2 = 1;
    while (p2 < want) p2 <<= 1;
    if (!c->mm_k || want <= 512) return p2;
    // up to 2^19 regions: 512 level-1 buckets of up to 1024 leaves, regions in steps of 512; up to 2^20: 1024 buckets, steps of
    // 1024; beyond: a leaf covers 2^g regions and the steps double with it
    uint64_t step = ((want + 511) / 512) * 512 <= (512ull << 10) ? 512 : 1024;
    while (((want + step - 1) / step) * step > (step << 10)) step <<= 1;
    return ((want + step - 1) / step) * step;
}

// Slots for `keys` distinct k-mers at load `load` in a minimizer-bin table: never more than the pipeline has leaves for
// (2^20, one region each) while the load stays under 0.6, and beyond that 2^g regions per leaf for the smallest g that
// keeps it there -- every doubling of g doubles the merge kernel's sweeps over a leaf's records.
static uint64_t mm_slots_for(const mc_ctx *c, double keys, double load)
{
    uint64_t want = (uint64_t)(keys / load);
    const uint64_t one_leaf_each = 1ull << (SK_LEAVES_LG + c->sb);
    if (c->mm_k && want > one_leaf_each) {
        uint64_t cap_slots = one_leaf_each;
        while (keys > 0.6 * (double)cap_slots) cap_slots <<= 1;
        want = std::min(want, cap_slots);
    }
    return want;
}

static int table_alloc(mc_ctx *c, uint64_t n_regions)
{
    if (n_regions < 1) n_regions = 1;
    if (n_regions > (1ull << 24)) return fail(c, MC_EOVERFLOW, "k-mer table would need %llu regions", (unsigned long long)n_regions);
    c->n_regions = n_regions;
    c->rb = 0;
    while ((2ull << c->rb) <= n_regions) c->rb++;
    {
        void *blk = nullptr;
        size_t got = 0;
        hipError_t e = g_table_pool.get(c->cfg.device, c->n_slots() * sizeof(Slot), &blk, &got);
        if (e == hipErrorOutOfMemory) {  // (idle scratch blocks are given back first)
            (void)hipGetLastError();
            g_scratch_pool.release(c->cfg.device);
            e = g_table_pool.get(c->cfg.device, c->n_slots() * sizeof(Slot), &blk, &got);
        }
        HIPCHK(c, e);
        c->slots = static_cast<Slot *>(blk);
        c->slots_bytes = got;
    }
    c->virgin = true;  // filled lazily: the partitioned pipeline writes every region itself
    c->st.table_slots = c->n_slots();
    c->st.table_bytes = c->n_slots() * sizeof(Slot);
    return MC_OK;
}

// the direct kernels, lookups and exports need real (EMPTY-initialised) slots
namespace { void shard_detach_locked(mc_ctx *c); }
// The table is about to hold other counts (or to move: growing and rehashing give the old block back to the pool): results
// read before are void, and so is a walker's view of the ranks' tables (mc_shard_attach) -- it names this table's block
// and geometry as they were.  ADVICE r4: the attachment used to survive, and the next walk read a freed block.
static inline void counts_changed(mc_ctx *c)
{
    if (c->d_shards) { shard_detach_locked(c); c->shards_dropped = true; }
    c->finalized = false;
    c->dup.checked = false;
    c->dup.l2_valid = false;
}

static int materialize(mc_ctx *c)
{
    if (!c->virgin) return MC_OK;
    hipLaunchKernelGGL(k_fill_empty, dim3(grid_for(c->n_slots(), 256)), dim3(256), 0, c->stream, c->slots,
                       c->n_slots());
    HIPCHK(c, hipGetLastError());
    c->virgin = false;
    return MC_OK;
}

static int read_counters(mc_ctx *c, unsigned long long *n_used, uint32_t *fatal)
{
    unsigned long long *h = c->h_scratch;
    HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, 9 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_used = h[0];
    *fatal = (uint32_t)h[8];
    return MC_OK;
}

// the table's block goes back to the pool
static void table_release(mc_ctx *c, Slot *slots, size_t bytes) { g_table_pool.put(c->cfg.device, slots, bytes); }

// What table_grow / to_hash_regions put aside while they move the keys: the old table comes back, with its counters, if
// anything fails before the move is complete (and goes to the pool when it is).
struct TableSwap {
    mc_ctx *c;
    Slot *old;
    size_t old_bytes;
    uint64_t old_regions;
    uint32_t old_rb;
    bool old_virgin;
    int old_mm;
    unsigned long long old_used = 0;
    bool have_used = false;  // old_used has been read (only then is it written back when the swap is abandoned)
    bool done = false;
    explicit TableSwap(mc_ctx *ctx) : c(ctx), old(ctx->slots), old_bytes(ctx->slots_bytes), old_regions(ctx->n_regions), old_rb(ctx->rb), old_virgin(ctx->virgin), old_mm(ctx->mm_k) {}
    ~TableSwap()
    {
        if (done) { table_release(c, old, old_bytes); return; }
        if (c->slots && c->slots != old) table_release(c, c->slots, c->slots_bytes);  // the half-filled new table
        c->slots = old; c->slots_bytes = old_bytes; c->n_regions = old_regions; c->rb = old_rb; c->virgin = old_virgin; c->mm_k = old_mm;
        if (have_used) (void)hipMemcpy(c->d_ctr, &old_used, sizeof old_used, hipMemcpyHostToDevice);  // (n_used as it was)
    }
};

// Hash keys in minimizer bins (count_long.h: polynomial keys, k > 32, counted as long records): nothing that has only the key
// can find it there, and such a table cannot be rebuilt with another number of bins either.  Everything that works by key --
// mc_get, key streams, the direct kernel, growing -- first moves the table to hash-prefix regions, for good (by_key_ready).
static inline bool hash_bins(const mc_ctx *c) { return c->mm_k != 0 && c->cfg.key_mode != MC_KEY_PACKED; }
__global__ void k_ctr_add(unsigned long long *p, long long d) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += (unsigned long long)d; }
// The slots of keys that sit in several regions (dup_check.h) get their own counts back: before the table takes more reads (the
// merge kernel adds to whichever slot a k-mer's bin holds) and before its keys move to hash-prefix regions (where equal keys meet
// and their counts add up by themselves).
static int dup_unmerge(mc_ctx *c)
{
    mc_ctx::Dup &D = c->dup;
    if (!D.merged || !D.n_tw) { D.merged = false; return MC_OK; }
    hipLaunchKernelGGL(k_dup_apply, dim3((unsigned)((D.n_tw + 255) / 256)), dim3(256), 0, c->stream, c->slots, D.tw, D.n_tw, D.set, 0);
    HIPCHK(c, hipGetLastError());
    if (D.solid_delta) {
        hipLaunchKernelGGL(k_ctr_add, dim3(1), dim3(64), 0, c->stream, c->d_ctr + 6, -D.solid_delta);
        HIPCHK(c, hipGetLastError());
    }
    D.merged = false;
    D.solid_delta = 0;
    D.checked = false;
    return MC_OK;
}
// nothing is known about the table's keys any more (it is empty, or no longer in minimizer bins)
static void dup_forget(mc_ctx *c)
{
    mc_ctx::Dup &D = c->dup;
    D.merged = false; D.n_tw = 0; D.n_keys = 0; D.solid_delta = 0; D.checked = false; D.l1_armed = false; D.expected_keys = 0; D.l2_valid = false;
    c->st.dup_keys = 0;
    c->st.dup_unchecked = 0;
}
// why: what mc_stats.left_bins reports (include/mcgpu.h): 1 something came or asked by key (a key stream, mc_load_kmers, mc_shard_export, a
// copy of the solid k-mers), 2 a small batch took the direct kernel, 3 the long form declined a batch (nothing vouched for the table's
// size), 4 bins overflowed or the table had to grow, 5 more keys sat in several regions than the join's lists take
static int to_hash_regions(mc_ctx *c, int why = 1);
static inline int by_key_ready(mc_ctx *c, int why = 1) { return hash_bins(c) ? to_hash_regions(c, why) : MC_OK; }

static int table_grow(mc_ctx *c, uint64_t new_regions)
{
    if (hash_bins(c) && c->virgin) {  // nothing to move: another empty table, still in minimizer bins
        table_release(c, c->slots, c->slots_bytes);
        c->slots = nullptr;
        int rc = table_alloc(c, new_regions);
        if (!rc) c->st.grows++;
        return rc;
    }
    if (hash_bins(c)) {
        int rc = to_hash_regions(c, 4);
        if (rc) return rc;
        new_regions = regions_for(c, new_regions << c->sb);  // (a power of two now)
        if (new_regions <= c->n_regions) return MC_OK;
    }
    TableSwap sw(c);
    HIPCHK(c, hipMemcpyAsync(&sw.old_used, c->d_ctr, sizeof sw.old_used, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sw.have_used = true;
    const uint64_t old_n = c->n_slots();
    c->slots = nullptr;
    int rc = table_alloc(c, new_regions);
    if (rc) return rc;
    if (!sw.old_virgin) {  // (else: nothing to move)
        rc = materialize(c);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(unsigned long long), c->stream));  // n_used is recounted
        hipLaunchKernelGGL(k_rehash, dim3(grid_for(old_n, 256)), dim3(256), 0, c->stream, sw.old, old_n, c->view());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    sw.done = true;
    c->st.grows++;
    return MC_OK;
}

// When the k-mers of a read set crowd into single minimizer bins -- a genome covered more than ~1500-fold: the error
// variants of one locus share its minimizer and outnumber a region's 4096 slots -- no number of regions helps.  The
// context then gives up minimizer bins for good: regions by the key's own hash, as for hash keys and short k-mers
// (reads take the per-window pipeline from here on), with everything counted so far moved over.
static int to_hash_regions(mc_ctx *c, int why)
{
    if (!c->mm_k) return MC_OK;
    if (hash_bins(c)) c->st.left_bins = (uint64_t)why;  // (the long-record form ends here for good, and only mc_stats.grows used to show it)
    if (int urc = dup_unmerge(c)) return urc;  // (equal keys meet in the new table: their own counts add up there)
    TableSwap sw(c);
    HIPCHK(c, hipMemcpyAsync(&sw.old_used, c->d_ctr, sizeof sw.old_used, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sw.have_used = true;
    const unsigned long long used = sw.old_used;
    const uint64_t old_n = c->n_slots();
    c->mm_k = 0;
    c->slots = nullptr;
    int rc = table_alloc(c, regions_for(c, std::max<uint64_t>(old_n, 2 * used)));
    if (rc) return rc;
    if (!sw.old_virgin) {
        rc = materialize(c);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(unsigned long long), c->stream));  // n_used is recounted
        hipLaunchKernelGGL(k_rehash, dim3(grid_for(old_n, 256)), dim3(256), 0, c->stream, sw.old, old_n, c->view());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    sw.done = true;
    dup_forget(c);
    c->solid_tracked = false;
    c->solid_list_fresh = false;
    c->solid_cov = -1;
    c->st.grows++;
    return MC_OK;
}

// Additions that found their region full were parked (TableView::ovf): enlarge the table -- twice as many regions
// split the bins that were crowded together -- and add them again, until none is left.
static int drain_parked(mc_ctx *c)
{
    for (int attempt = 0;; attempt++) {
        unsigned long long n = 0;
        HIPCHK(c, hipMemcpyAsync(&n, c->d_ctr + 7, sizeof n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n == 0) return MC_OK;
        if (int brc = by_key_ready(c)) return brc;  // (the parked additions are keys)
        if (n > mc_ctx::OVF_CAP || attempt >= 8)
            return fail(c, MC_EOVERFLOW, "k-mer table regions keep overflowing (%llu additions parked); pass a capacity_hint "
                        "(distinct k-mers), or set MC_SUPERKMERS=0 for read sets that cover a genome thousands of times", n);
        DevBuf<uint4> tmp;
        HIPCHK(c, tmp.alloc(n));
        HIPCHK(c, hipMemcpyAsync(tmp.p, c->d_ovf, n * sizeof(uint4), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_ctr + 7, 0, sizeof(unsigned long long), c->stream));
        // (minimizer bins: if two doublings did not make room, the crowd shares one bin)
        int rc = (c->mm_k && attempt >= 2) ? to_hash_regions(c, 4) : table_grow(c, c->n_regions * 2);
        if (rc) return rc;
        c->solid_tracked = false;  // (these additions were not watched for crossing the coverage threshold)
        c->solid_list_fresh = false;
        hipLaunchKernelGGL(k_add_parked, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, tmp.p, (uint64_t)n, c->view());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
}

// Make room for `incoming` more key occurrences: returns how many of them may be inserted by the
// next launch without the load factor passing 0.85 even if every one is a new key.
static int table_reserve(mc_ctx *c, uint64_t incoming, uint64_t *allowed)
{
    int mrc = by_key_ready(c, 2);  // (what follows goes in by key: the direct kernel)
    if (!mrc) mrc = materialize(c);
    if (!mrc) mrc = drain_parked(c);  // (what the previous launch could not place)
    if (mrc) return mrc;
    unsigned long long used;
    uint32_t fatal;
    int rc = read_counters(c, &used, &fatal);
    if (rc) return rc;
    if (fatal) return fail(c, MC_EOVERFLOW, "a k-mer table region filled up (hash skew); table of %llu slots%s",
                           (unsigned long long)c->n_slots(), c->mm_k ? "; MC_SUPERKMERS=0 copes with read sets that cover a genome thousands of times" : "");
    c->n_used_host = used;
    const uint64_t max_launch = 1ull << 26;
    for (;;) {
        const uint64_t cap = c->n_slots();
        const uint64_t soft = (uint64_t)(0.70 * (double)cap), hard = (uint64_t)(0.85 * (double)cap);
        const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>(incoming, max_launch));
        if (used + want <= hard) {
            *allowed = want;
            return MC_OK;
        }
        const uint64_t room = hard > used ? hard - used : 0;
        if (used < soft && room >= cap / 16) {  // still worth a launch before rebuilding
            *allowed = room;
            return MC_OK;
        }
        rc = table_grow(c, c->n_regions * 2);
        if (rc) return rc;
    }
}

template <typename F>
static int timed(mc_ctx *c, double *acc_ms, F &&launch)
{
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    launch();
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *acc_ms += ms;
    return MC_OK;
}

static void launch_count(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t r0, uint64_t r1)
{
    const int block = 256;
    const int grid = grid_for((r1 - r0) * 64, block, 256 * 8);
    const TableView t = c->view();
    c->solid_tracked = false;
    c->solid_list_fresh = false;
    switch (c->cfg.key_mode) {
    case MC_KEY_PACKED:
        hipLaunchKernelGGL(k_count_reads<KEY_PACKED>, dim3(grid), dim3(block), 0, c->stream, d_words, d_off, r0, r1,
                           c->cfg.k, t, c->cur_ptr_base, (uint32_t)c->cov_hint);
        break;
    case MC_KEY_POLY:
        hipLaunchKernelGGL(k_count_reads<KEY_POLY>, dim3(grid), dim3(block), 0, c->stream, d_words, d_off, r0, r1,
                           c->cfg.k, t, c->cur_ptr_base, (uint32_t)c->cov_hint);
        break;
    default:
        hipLaunchKernelGGL(k_count_reads<KEY_FNV1A>, dim3(grid), dim3(block), 0, c->stream, d_words, d_off, r0, r1,
                           c->cfg.k, t, c->cur_ptr_base, (uint32_t)c->cov_hint);
    }
}

// hipMalloc that gives the pools' idle blocks back to the driver before it reports "out of memory"
static hipError_t dev_malloc(mc_ctx *c, void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        g_table_pool.release(c->cfg.device);
        g_scratch_pool.release(c->cfg.device);
        e = hipMalloc(p, bytes);
    }
    return e;
}

template <typename T>
static int ensure_buf(mc_ctx *c, T **p, uint64_t *cap, uint64_t need)
{   // (blocks of 64 MB and more come from and go to g_scratch_pool: *cap may come out above `need`)
    if (*cap >= need && *p) return MC_OK;
    if (*p) {  // (a block that goes back to the pool may be handed out at once: what is queued on it must be done)
        (void)hipStreamSynchronize(c->stream);
        if (c->pipe_stream) (void)hipStreamSynchronize(c->pipe_stream);
        g_scratch_pool.put(c->cfg.device, *p, *cap * sizeof(T));
    }
    *p = nullptr;
    *cap = 0;
    const size_t bytes = std::max<uint64_t>(need, 1) * sizeof(T);
    size_t got = 0;
    hipError_t e = g_scratch_pool.get(c->cfg.device, bytes, reinterpret_cast<void **>(p), &got);
    if (e == hipErrorOutOfMemory) {  // (idle blocks are given back first)
        (void)hipGetLastError();
        g_table_pool.release(c->cfg.device);
        g_scratch_pool.release(c->cfg.device);
        e = g_scratch_pool.get(c->cfg.device, bytes, reinterpret_cast<void **>(p), &got);
    }
    if (e != hipSuccess) {
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        *p = nullptr;
        return fail(c, e == hipErrorOutOfMemory ? MC_ENOMEM : MC_EHIP, "scratch of %.2f GB: %s (%.2f of %.2f GB free on the device)",
                    bytes / 1e9, hipGetErrorString(e), fr / 1e9, tot / 1e9);
    }
    *cap = std::max<uint64_t>(need, got / sizeof(T));
    return MC_OK;
}

// ------------------------------------------------------------------------------------------ equal keys in different regions (dup_check.h)

static bool dup_check_on()
{
    const char *e = getenv("MC_DUP_CHECK");  // (read on every call: the tests switch it)
    return !(e && !strcmp(e, "0"));
}

static int dup_small_bufs(mc_ctx *c)
{
    mc_ctx::Dup &D = c->dup;
    if (!D.flags) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&D.flags), 4 * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&D.ctr), 4 * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&D.list), mc_ctx::Dup::LIST_CAP * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(D.flags, 0, 4 * sizeof(uint32_t), c->stream));
    }
    return MC_OK;
}

// Before the join takes tens of GB beside a table that fills half the device: blocks that sit idle in the process-wide pools go back
// to the driver when less than `need` + 6 GB is free (an allocation that only just fits leaves a later kernel launch nothing for its
// own bookkeeping: configs[2] at full size after other contexts of the process had left their blocks in the pools).
static void dup_make_room(mc_ctx *c, uint64_t need_bytes)
{
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return; }
    if ((uint64_t)fr >= need_bytes + (6ull << 30)) return;
    (void)hipStreamSynchronize(c->stream);
    g_scratch_pool.release(c->cfg.device);
    g_table_pool.release(c->cfg.device);
}

// The key streams of the join go back to the scratch pool (mc_trim, or a result buffer that does not fit beside them: 84 GB at
// configs[2]'s full size): the next long run's merge kernel gets new ones, a walk's check by key sweeps the table instead.
static void dup_release_streams(mc_ctx *c)
{
    mc_ctx::Dup &D = c->dup;
    const int dev = c->cfg.device;
    (void)hipStreamSynchronize(c->stream);
    g_scratch_pool.put(dev, D.l1_keys, D.l1_words * 8); D.l1_keys = nullptr; D.l1_words = 0;
    g_scratch_pool.put(dev, D.l1_counts, D.l1_counts_cap * 4); D.l1_counts = nullptr; D.l1_counts_cap = 0;
    g_scratch_pool.put(dev, D.l2_own, D.l2_own_words * 8); D.l2_own = nullptr; D.l2_own_words = 0;
    D.l1_armed = false;
    if (D.l2.out_a && D.l2.out_a != reinterpret_cast<uint64_t *>(c->pipe.a_recs)) D.l2_valid = false;  // (streams borrowed from the pipeline stay)
}
// a result buffer for the caller: when the device is full, the join's streams and the pools' idle blocks make room
template <typename T>
static hipError_t alloc_result(mc_ctx *c, DevBuf<T> &b, size_t n)
{
    hipError_t e = b.alloc(n);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        dup_release_streams(c);
        g_scratch_pool.release(c->cfg.device);
        g_table_pool.release(c->cfg.device);
        e = b.alloc(n);
    }
    return e;
}

// capacity of a stream that is expected to take `mean` keys (they come hashed: Poisson, and a little more for the regions' unevenness)
static uint64_t dup_cap(double mean) { return (uint64_t)(mean * 1.12 + 7.0 * std::sqrt(mean) + 64.0); }

// The level-1 streams for a launch of k_p3_long with `grid` workgroups into a table that will hold about `keys` keys.  Not
// getting them is no error: the join then starts with a sweep of the table.
static void dup_arm(mc_ctx *c, double keys, uint32_t grid)
{
    mc_ctx::Dup &D = c->dup;
    D.l1_armed = false;
    static const bool fused = [] { const char *e = getenv("MC_DUP_FUSED"); return !(e && !strcmp(e, "0")); }();
    if (!fused || !dup_check_on() || keys < 1.0) return;
    const std::string keep = c->err;
    if (dup_small_bufs(c)) { c->err = keep; (void)hipGetLastError(); return; }
    const uint32_t nseg = grid + 1;  // (the last one: keys that enter the table outside the merge kernel)
    uint64_t cap = dup_cap(keys / ((double)DUP_B1 * (double)grid));
    if (const char *e = getenv("MC_DUP_L1_SCALE")) cap = std::max<uint64_t>(4, (uint64_t)((double)cap * atof(e)));  // (tests: streams that overflow)
    if (D.l1_words < (uint64_t)DUP_B1 * nseg * cap) dup_make_room(c, (uint64_t)DUP_B1 * nseg * cap * 8);
    if (ensure_buf(c, &D.l1_keys, &D.l1_words, (uint64_t)DUP_B1 * nseg * cap) || ensure_buf(c, &D.l1_counts, &D.l1_counts_cap, (uint64_t)DUP_B1 * nseg)) {
        c->err = keep;
        (void)hipGetLastError();
        return;
    }
    if (hipMemsetAsync(D.l1_counts, 0, (uint64_t)DUP_B1 * nseg * sizeof(uint32_t), c->stream) != hipSuccess ||
        hipMemsetAsync(D.flags, 0, sizeof(uint32_t), c->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    D.l1_nseg = nseg;
    D.l1_cap = cap;
    D.l1_armed = true;
}

// The listed keys' slots: one sweep of the table notes them, every one of them then holds the sum of its key's counters
// (dup_check.h); past what the lists take, the table gives up its minimizer bins instead -- equal keys meet in hash-prefix regions.
static int dup_fixup(mc_ctx *c, uint64_t n_listed)
{
    mc_ctx::Dup &D = c->dup;
    if (n_listed > mc_ctx::Dup::LIST_CAP) return to_hash_regions(c, 5);
    uint64_t want = 1024;
    while (want < 4 * n_listed) want <<= 1;
    if (want > D.set_slots) {
        if (D.set.qk) { (void)hipFree(D.set.qk); D.set.qk = nullptr; D.set_slots = 0; }
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&D.set.qk), want * 3 * sizeof(unsigned long long)));
        D.set_slots = want;
    }
    D.set.tot = D.set.qk + D.set_slots;
    D.set.prim = D.set.qk + 2 * D.set_slots;
    D.set.mask = D.set_slots - 1;
    D.set.n_keys = D.ctr + 1;
    const uint64_t tw_want = 8 * n_listed + 64;
    if (tw_want > D.tw_cap) {
        if (D.tw) { (void)hipFree(D.tw); D.tw = nullptr; D.tw_cap = 0; }
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&D.tw), tw_want * sizeof(DupTwin)));
        D.tw_cap = tw_want;
    }
    HIPCHK(c, hipMemsetAsync(D.set.qk, 0xFF, D.set_slots * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(D.set.tot, 0, D.set_slots * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(D.set.prim, 0xFF, D.set_slots * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(D.ctr + 1, 0, 2 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_dupq_build, dim3(grid_for(n_listed, 256)), dim3(256), 0, c->stream, D.list, n_listed, D.set);
    hipLaunchKernelGGL(k_dupq_sweep, dim3(grid_for(c->n_slots(), 256, 256 * 16)), dim3(256), 0, c->stream, c->slots, c->n_slots(), D.set, D.tw, D.ctr + 2, D.tw_cap);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_scratch + 20, D.ctr, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t n_keys = c->h_scratch[21], n_tw = c->h_scratch[22];
    if (n_tw > D.tw_cap) return to_hash_regions(c, 5);  // (a key in more slots than anybody planned for)
    D.n_keys = n_keys;
    D.n_tw = n_tw;
    D.solid_delta = 0;
    if (c->solid_tracked && c->cov_hint > 0 && n_tw) {  // d_ctr[6] counts KEYS at the threshold: once for a merged key, by its sum
        std::vector<DupTwin> tw(n_tw);
        std::vector<unsigned long long> tot(D.set_slots);
        HIPCHK(c, hipMemcpy(tw.data(), D.tw, n_tw * sizeof(DupTwin), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(tot.data(), D.set.tot, D.set_slots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::map<uint32_t, int> own_solid;
        for (const DupTwin &t : tw) own_solid[t.entry] += t.own >= (uint32_t)c->cov_hint ? 1 : 0;
        for (const auto &e : own_solid) D.solid_delta += (tot[e.first] >= (unsigned long long)c->cov_hint ? 1 : 0) - e.second;
    }
    if (n_tw) {
        hipLaunchKernelGGL(k_dup_apply, dim3((unsigned)((n_tw + 255) / 256)), dim3(256), 0, c->stream, c->slots, D.tw, n_tw, D.set, 1);
        if (D.solid_delta) hipLaunchKernelGGL(k_ctr_add, dim3(1), dim3(64), 0, c->stream, c->d_ctr + 6, D.solid_delta);
        HIPCHK(c, hipGetLastError());
        c->solid_list_fresh = false;  // (the merge kernel listed these keys by their own counts, once a slot)
    }
    D.merged = true;
    return MC_OK;
}

// mc_finalize_counts: a table of hash keys in minimizer bins is joined by key (dup_check.h) unless that has been done for the
// table as it is.
static int ensure_dups(mc_ctx *c)
{
    mc_ctx::Dup &D = c->dup;
    mc_ctx::Pipe &P = c->pipe;
    if (!hash_bins(c) || c->virgin) { dup_forget(c); return MC_OK; }
    if (D.checked) return MC_OK;
    if (!dup_check_on()) { D.checked = true; D.l1_armed = false; c->st.dup_unchecked = 1; return MC_OK; }
    c->st.dup_unchecked = 0;
    int rc = dup_unmerge(c);  // (only a table whose slots hold their own counts is joined: nothing has touched it since, but say so)
    if (rc) return rc;
    D.n_tw = 0; D.n_keys = 0;
    rc = dup_small_bufs(c);
    if (rc) return rc;
    unsigned long long n_used = 0;
    HIPCHK(c, hipMemcpyAsync(c->h_scratch + 20, c->d_ctr, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (D.l1_armed) HIPCHK(c, hipMemcpyAsync(c->h_scratch + 21, D.flags, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    n_used = c->h_scratch[20];
    bool have_l1 = D.l1_armed && (uint32_t)c->h_scratch[21] == 0;
    if (D.l1_armed && !have_l1 && getenv("MC_INGEST_DEBUG"))
        fprintf(stderr, "[join] a segment of the merge kernel's key streams overflowed (the table holds %llu keys): the first level by a sweep\n", (unsigned long long)c->h_scratch[20]);
    D.l1_armed = false;  // (whatever happens next: the streams serve one join)
    if (n_used < 2) { D.checked = true; c->st.dup_keys = 0; return MC_OK; }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    uint64_t n_listed = 0;
    for (int attempt = 0;; attempt++) {
        if (!have_l1) {  // level 1 by a sweep of the table: 1021 workgroups, each a segment of every bucket
            const uint32_t grid = 1021, nseg = grid + 1;  // (a region is 16 rows of the sweep, and a row's slots lie in 16 of the 256 buckets: not a multiple of 16)
            const uint64_t cap = dup_cap((double)n_used / ((double)DUP_B1 * grid));
            if (D.l1_words < (uint64_t)DUP_B1 * nseg * cap) dup_make_room(c, (uint64_t)DUP_B1 * nseg * cap * 8);
            rc = ensure_buf(c, &D.l1_keys, &D.l1_words, (uint64_t)DUP_B1 * nseg * cap);
            if (!rc) rc = ensure_buf(c, &D.l1_counts, &D.l1_counts_cap, (uint64_t)DUP_B1 * nseg);
            if (rc) return rc;
            HIPCHK(c, hipMemsetAsync(D.l1_counts, 0, (uint64_t)DUP_B1 * nseg * sizeof(uint32_t), c->stream));
            HIPCHK(c, hipMemsetAsync(D.flags, 0, sizeof(uint32_t), c->stream));
            D.l1_nseg = nseg;
            D.l1_cap = cap;
            hipLaunchKernelGGL(k_dup_sweep, dim3(grid), dim3(256), 0, c->stream, c->slots, c->n_slots(), DupL1{D.l1_keys, D.l1_counts, nseg, cap, D.flags});
            HIPCHK(c, hipGetLastError());
        }
        // level 2 into sub-buckets of ~1500 keys, every workgroup (bucket, slice of its segments) its own segment of each; the streams
        // borrow the pipeline's idle ones where those are large enough
        const uint32_t slices = std::max<uint32_t>(2u, (D.l1_nseg + DUP_MAX_SLICE_SEGS - 1) / DUP_MAX_SLICE_SEGS);
        if (slices > DUP_MAX_SLICES) return fail(c, MC_EINVAL, "internal: %u key-stream segments a bucket", D.l1_nseg);
        uint32_t f2_lg = 0;
        while (f2_lg < DUP_MAX_F2_LG && (double)n_used / ((double)DUP_B1 * (double)(1u << f2_lg)) > 1800.0) f2_lg++;
        const double per_sub = (double)n_used / ((double)DUP_B1 * (double)(1u << f2_lg));
        const uint64_t cap2 = dup_cap(per_sub / slices);
        const uint64_t per_bucket = ((uint64_t)slices << f2_lg) * cap2, need = per_bucket * DUP_B1;
        DupL2 l2{nullptr, nullptr, DUP_B1, nullptr, f2_lg, slices, cap2, D.flags + 1};
        {
            const uint64_t a_words = P.a_recs ? P.a_recs_cap * 2 : 0, b_words = P.b_recs ? P.b_recs_cap * 2 : 0;
            const uint64_t in_a = std::min<uint64_t>(a_words / per_bucket, DUP_B1);
            if (in_a + std::min<uint64_t>(b_words / per_bucket, DUP_B1) >= DUP_B1 && in_a > 0) {
                l2.out_a = reinterpret_cast<uint64_t *>(P.a_recs);
                l2.out_b = reinterpret_cast<uint64_t *>(P.b_recs);
                l2.split = (uint32_t)in_a;
                c->solid_list_fresh = false;  // (the solid list sat in a_recs)
            } else {
                if (D.l2_own_words < need) dup_make_room(c, need * 8);
                rc = ensure_buf(c, &D.l2_own, &D.l2_own_words, need);
                if (rc) return rc;
                l2.out_a = D.l2_own;
                l2.out_b = D.l2_own;
            }
        }
        const uint64_t n_cnt2 = ((uint64_t)DUP_B1 << f2_lg) * slices;
        rc = ensure_buf(c, &D.l2_counts, &D.l2_counts_cap, n_cnt2);
        if (rc) return rc;
        l2.counts = D.l2_counts;
        HIPCHK(c, hipMemsetAsync(D.flags + 1, 0, sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(D.ctr, 0, sizeof(unsigned long long), c->stream));
        const DupL1 l1{D.l1_keys, D.l1_counts, D.l1_nseg, D.l1_cap, D.flags};
        hipLaunchKernelGGL(k_dup_scatter, dim3(DUP_B1 * slices), dim3(DUP_THREADS), 0, c->stream, l1, l2);
        const DupOut lst{D.list, D.ctr, mc_ctx::Dup::LIST_CAP};
        // (an LDS set of 2^13 32-bit fingerprints, 32 KB, five workgroups a CU -- or, for sub-buckets beyond 4 500 keys, i.e. a table of
        // more than 1.2 G keys, of 2^15 in a workgroup of 1024 threads; beyond 19 600 keys a sub-bucket goes through the set in passes)
        if (per_sub <= 4500.0) hipLaunchKernelGGL((k_dup_find<13, 256>), dim3(std::min<uint32_t>(DUP_B1 << f2_lg, 256u * 40u)), dim3(256), 0, c->stream, l2, lst);
        else hipLaunchKernelGGL((k_dup_find<15, 1024>), dim3(std::min<uint32_t>(DUP_B1 << f2_lg, 256u * 8u)), dim3(1024), 0, c->stream, l2, lst);
        HIPCHK(c, hipGetLastError());
        D.l2 = l2;
        HIPCHK(c, hipMemcpyAsync(c->h_scratch + 20, D.ctr, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_scratch + 21, D.flags, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        uint32_t fl[2];
        memcpy(fl, c->h_scratch + 21, sizeof fl);
        if (!fl[0] && !fl[1]) { n_listed = c->h_scratch[20]; break; }
        // a stream overflowed (the merge kernel's were sized by a hint that fell short, or the keys are far from evenly spread): once
        // more from a sweep, whose streams are sized by the number of keys the table holds
        if (attempt >= 1 || !have_l1)
            return fail(c, MC_EOVERFLOW, "internal: the key streams of the duplicate-key join overflowed (%llu keys)", n_used);
        if (getenv("MC_INGEST_DEBUG")) fprintf(stderr, "[join] the merge kernel's key streams overflowed (sized for %.0f keys, the table holds %llu): once more from a sweep\n",
                                               c->cfg.capacity_hint ? (double)c->cfg.capacity_hint : D.expected_keys, n_used);
        have_l1 = false;
    }
    if (n_listed) {
        rc = dup_fixup(c, n_listed);
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->st.dup_ms += ms;
    c->st.dup_checks++;
    c->st.dup_keys = D.n_keys;
    D.checked = true;
    D.l2_valid = hash_bins(c);  // (the streams of level 2 hold the table's keys by key until the pipeline takes its buffers back)
    return MC_OK;
}

struct PipePlan {
    uint32_t b1 = 0, b2 = 1, g = 0;  // b1 level-1 buckets of b2 leaves each (counts, not bits); a leaf covers 2^g regions
    uint64_t np1 = 0, n_leaves = 0, cap1 = 0, cap2 = 0, spill_cap = 0, wb = 0;
    uint32_t nseg1 = PT_SEGMENTS;  // segments of every level-1 bucket = workgroups of the level-1 kernel
    uint32_t pieces = 1;           // the reads go through P1 / P2 in this many pieces (cap1, cap2: per piece); P3 sees `pieces` segments per leaf
    bool sk = false;  // the streams hold super-k-mer records; capacities are in records
    // compact: the streams hold 16-byte units and no pointer arrays (count_pipeline.h k_sk1w_extract<false, true>, k_sk2_scatter_compact):
    // every level-1 segment = workgroup took chunk_tiles consecutive tiles, the first of them at base position pos0
    bool compact = false;
    uint32_t chunk_tiles = 0;
    uint64_t pos0 = 0;
    bool guessed = false;  // no capacity hint vouches for the table's size: pipe_finish merges a sample of the leaves first
    // vleaf: compact records written before the table's size is known (a first batch without a hint): their leaf field holds the TEN
    // bits of the bin word below the level-1 bucket's -- 2^10 leaves a bucket whatever the table --, the table is then given a power
    // of two of regions (pipe_resize_by_sample) and the second level drops the bits it does not need
    bool vleaf = false;
    bool lng = false;      // long records (count_long.h): two 16-byte words a record in every stream
    // listed: there is no first level -- the records lie in the caller's buffer (in_recs / in_ptrs) in pieces that are in level-1 bucket
    // order already (the binned exchange), and the second level reads segment sg of a bucket from pipe.skb_seg_start[bucket * nseg1 + sg]
    bool listed = false;
    const uint4 *in_recs = nullptr;
    const uint32_t *in_ptrs = nullptr;
    SpillView sp{};
    SkSpill sks{};
};

// first_read[t] for every tile (count_pipeline.h): by the reads where they are short, by a search per tile otherwise
static void launch_tile_first(mc_ctx *c, const uint64_t *offs, uint64_t nr, uint64_t n_tiles, uint32_t *out, uint32_t tile_size)
{
    if (nr && n_tiles <= 8 * nr)  // (a read spans a few tiles at most on average: its thread's loop is short)
        hipLaunchKernelGGL(k_tile_first_read_by_reads, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream, offs, nr, n_tiles, out, tile_size);
    else
        hipLaunchKernelGGL(k_tile_first_read, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, c->stream, offs, nr, n_tiles, out, tile_size);
}

// the control words of a pipeline run, cleared in one launch (flags: 4 words + the 64-bit spill counter behind them)
__global__ void __launch_bounds__(256) k_pipe_reset(uint32_t *seg_counts1, uint64_t n1, uint32_t *cursors2, uint64_t n2, uint32_t *leaf_state,
                                                    uint32_t *leaf_new, uint64_t n_leaves, uint32_t *flags)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = t; i < n1; i += stride) seg_counts1[i] = 0;
    for (uint64_t i = t; i < n2; i += stride) cursors2[i] = 0;
    for (uint64_t i = t; i < n_leaves; i += stride) { leaf_state[i] = 0; leaf_new[i] = 0; }
    if (flags && t < 6) flags[t] = 0;
}

// How a table of c->n_regions regions splits into level-1 buckets and leaves (records: the super-k-mer pipeline; else keys): *n_leaves
// leaves of 2^*g regions each, *np1 level-1 buckets.  Returns what is wrong, or null.
static const char *plan_levels(const mc_ctx *c, bool records, uint64_t *n_leaves_out, uint32_t *g_out, uint64_t *np1_out)
{
    // leaves: the regions themselves up to 2^20 of them (2^18 when keys travel, not records), else 2^g regions per leaf;
    // level-1 buckets: up to 512, each of m2 <= 1024 (512) leaves (regions_for made the numbers divide)
    uint64_t n_leaves = c->n_regions;
    uint32_t g = 0;
    // (2^20 leaves at most in either form: a leaf that covers 2^g regions is swept 2^(g+1) times by the merge kernel -- with 2^18
    // leaves the 137 GB table of configs[2] had g = 3, and two thirds of that run were those sweeps)
    const uint32_t max_b2 = records ? PT_MAX_LEAVES2 : PT_MAX_BUCKETS_KEYS;  // (super-k-mer records: k_sk2_scatter; keys: k_p2_scatter)
    const uint32_t max_b1_big = records ? PT_MAX_BUCKETS1_SK : PT_MAX_BUCKETS_KEYS;  // (1024 level-1 buckets only when 512 do not do)
    while (n_leaves > (uint64_t)max_b1_big * max_b2) { n_leaves >>= 1; g++; }
    if ((n_leaves << g) != c->n_regions) return "the table's regions do not split into leaves";
    if (n_leaves < 4) return "partitioned counting needs at least 4 table regions";
    // (512 level-1 buckets while 512 x max_b2 leaves do, then 1024, then -- super-k-mer records only -- 2048)
    uint64_t np1 = PT_MAX_BUCKETS;
    while (np1 < max_b1_big && n_leaves > np1 * max_b2) np1 *= 2;
    np1 = std::min<uint64_t>(n_leaves, np1);
    if (records) if (const char *e = getenv("MC_SK_B1")) np1 = std::min<uint64_t>(n_leaves, std::min<uint64_t>(std::max<uint64_t>(strtoull(e, nullptr, 10), 1), max_b1_big));  // (tuning runs)
    while (n_leaves % np1) np1--;  // (a power of two, or 512 dividing a multiple of 512)
    if (n_leaves / np1 > max_b2) return "the leaves do not fit two scatter levels";
    *n_leaves_out = n_leaves; *g_out = g; *np1_out = np1;
    return nullptr;
}

// Makes sure the table can take a batch of wb key occurrences: with a capacity hint that still holds the table was sized for
// it; without one assume every eighth occurrence is a new key at most.  Either way the merge kernel
// reports regions that would overflow and the table is grown then.
static int pipe_reserve(mc_ctx *c, uint64_t wb, bool *hint_holds_out)
{
    unsigned long long used;
    uint32_t fatal;
    int rc = read_counters(c, &used, &fatal);
    if (rc) return rc;
    if (fatal) return fail(c, MC_EOVERFLOW, "a k-mer table region filled up (hash skew)");
    c->n_used_host = used;
    const bool hint_holds = c->cfg.capacity_hint && used < c->cfg.capacity_hint;
    if (!hint_holds && c->n_slots() < wb / 4) {
        const uint64_t want = regions_for(c, (uint64_t)(((double)used + (double)wb / 8.0) / 0.5));
        if (want > c->n_regions) {
            rc = table_grow(c, want);
            if (rc) return rc;
        }
    }
    *hint_holds_out = hint_holds;
    return MC_OK;
}

// Table capacity check, scratch buffers and cursors for one run of the partitioned pipeline over
// `wb` key occurrences.  n_records != 0: they travel as (an estimated) n_records super-k-mer records.
static int pipe_prepare(mc_ctx *c, uint64_t wb, PipePlan *pl, uint64_t n_records = 0, uint32_t nseg1 = PT_SEGMENTS, uint32_t pieces = 1,
                        bool level2_only = false, uint64_t compact_tiles = 0, bool lng = false, bool listed = false)
{   // listed: PipePlan::listed -- no first-level streams (the solid list alone may want pipe.a_recs), the caller has filled
    // pipe.seg_counts1 (np1 x nseg1 fill levels) and pipe.skb_seg_start already
    pl->lng = lng;
    pl->listed = listed;
    const uint64_t rw = lng ? 2 : 1;  // 16-byte words a record   // compact_tiles != 0: the caller's level-1 kernel is k_sk1w_extract over this many tiles in one piece and can write the compact
    // form: taken up when the rest of the run allows it (a second level through the staged kernel, one region per leaf, segments
    // short enough for 22-bit positions)   // level2_only: the level-1 scatter has run and the table has since been replaced by one of another size (pipe_resize_by_sample):
    // the plan of the second level and of the merge is made again for it; level 1 (buckets, segments, their fill levels, the
    // spill list) stays as it is -- the caller has checked that the new table splits into the same level-1 buckets
    pl->nseg1 = nseg1;
    pl->pieces = pieces;
    mc_ctx::Pipe &P = c->pipe;
    c->solid_list_fresh = false;  // the pipeline buffers are about to be reused
    c->dup.l2_valid = false;      // (... and with them what the key join left there)
    if (!level2_only) {
        bool hint_holds;
        int rc = pipe_reserve(c, wb, &hint_holds);
        if (rc) return rc;
        pl->guessed = !hint_holds;
    }
    uint64_t n_leaves, np1;
    uint32_t g;
    if (const char *why = plan_levels(c, n_records != 0, &n_leaves, &g, &np1)) return fail(c, MC_EINVAL, "internal: %s (%llu table regions)", why, (unsigned long long)c->n_regions);
    pl->b1 = (uint32_t)np1;               // level-1 buckets
    pl->b2 = (uint32_t)(n_leaves / np1);  // leaves per bucket; 1: the level-1 buckets already are the leaves, no P2
    pl->g = g;
    pl->wb = wb;
    pl->np1 = np1;
    pl->n_leaves = n_leaves;
    pl->sk = n_records != 0;
    if (pl->b2 <= 1) pl->pieces = pieces = 1;  // (no second level to overlap with)
    if (!level2_only) {
        const char *ce = getenv("MC_SK_COMPACT");  // (read on every run: the tests switch it)
        const bool compact_on = !(ce && !strcmp(ce, "0"));
        static const bool staged = [] { const char *e = getenv("MC_SK2_STAGED"); return !(e && !strcmp(e, "0")); }();
        const uint64_t chunk = (compact_tiles + nseg1 - 1) / std::max<uint32_t>(nseg1, 1);
        // (not where pipe_resize_by_sample may replace the table between the two levels: the leaf numbers inside the records
        // are those of the table the first level saw)
        pl->compact = compact_tiles && compact_on && staged && pl->sk && pl->b2 > 1 && pl->b2 <= (1u << (32 - SKC_REL_BITS)) && g == 0 && pieces == 1 &&
                      chunk * P1W_TILE <= (1ull << SKC_REL_BITS) && !(pl->guessed && c->virgin);
        // ... unless the table can be given a power of two of regions once the batch's size is known: 512 buckets of 2 .. 1024 leaves
        const char *ve = getenv("MC_SK_VLEAF");  // (read on every run: the tests switch it)
        const bool vleaf_on = !(ve && !strcmp(ve, "0"));
        pl->vleaf = false;
        if (!pl->compact && !lng && vleaf_on && !c->no_vleaf && compact_tiles && compact_on && staged && pl->sk && g == 0 && pieces == 1 && chunk * P1W_TILE <= (1ull << SKC_REL_BITS) &&
            pl->guessed && c->virgin && c->mm_k > 0 && np1 == PT_MAX_BUCKETS && c->n_regions <= (uint64_t)PT_MAX_BUCKETS * 1024) {
            uint64_t r2 = 2 * PT_MAX_BUCKETS;
            while (r2 < c->n_regions) r2 *= 2;
            if (r2 != c->n_regions) {  // (a virgin table: nothing to move)
                table_release(c, c->slots, c->slots_bytes);
                c->slots = nullptr;
                int rc = table_alloc(c, r2);
                if (rc) return rc;
                n_leaves = r2;
                pl->b2 = (uint32_t)(n_leaves / np1);
                pl->n_leaves = n_leaves;
            }
            pl->compact = pl->vleaf = true;
        }
        // (long records keep the leaf in their second word: 32 bits of position)
        // (... and the bin word, so the table may still be replaced between the levels: a virgin table without a hint qualifies)
        if (lng) pl->compact = compact_tiles && pl->sk && pl->b2 > 1 && pl->b2 <= 1024 && g == 0 && pieces == 1 && chunk * P1L_TILE < (1ull << 32) &&
                               !(pl->guessed && !c->virgin);
        pl->chunk_tiles = pl->compact ? (uint32_t)chunk : 0;
        if (listed) pl->compact = true;  // (the listed second level leaves the pointer in the record's first word too: no pointer arrays behind it)
    }
    const uint64_t units = (pl->sk ? n_records : wb) / pieces + (pieces > 1 ? 1024 : 0);  // records in the streams (of one piece)
    // (long records into 2^20 leaves and more -- configs[2] at full size, where the table leaves the streams 130 GB --: tighter streams)
    const bool tight = lng && n_leaves >= (1ull << 20);
    pl->cap1 = (uint64_t)((double)units / (double)pl->np1 / (double)nseg1 * (tight ? 1.12 : 1.25)) + (pl->sk ? 64 : 256);  // per segment
    const double mean_leaf = (double)units / (double)pl->n_leaves;
    // (records of one locus come in clumps -- one per read covering it -- so leaves vary more than Poisson)
    double sig2 = pl->sk ? (tight ? 12.0 : 32.0) : 8.0;
    if (const char *e = getenv("MC_CAP2_SIGMAS")) { const double v = atof(e); if (v >= 1.0 && v <= 64.0) sig2 = v; }  // (tuning runs: how far apart the leaves' streams lie)
    pl->cap2 = (uint64_t)(mean_leaf * 1.15 + sig2 * std::sqrt(mean_leaf) + 64.0);  // (a spilled record also costs the solid list, P3Emit)
    if (const char *e = getenv("MC_CAP2")) { const long v = atol(e); if (v >= 64) pl->cap2 = (uint64_t)v; }  // (tuning runs)
    pl->spill_cap = pl->sk ? std::max<uint64_t>(units * pieces / 16, 1u << 16) : std::max<uint64_t>(wb / 64, 1u << 20);
    if ((!listed && pl->np1 * nseg1 * pl->cap1 >= 0xFFFFFFFFull) || (uint64_t)pl->b2 * pl->cap2 * pieces >= 0xFFFFFFFFull)
        return fail(c, MC_EINVAL, "internal: partitioned batch too large for 32-bit bucket indices");
    // (np1, n_leaves as computed above)
    int rc;
    uint64_t dummy;
#define ENSURE(ptr, capvar, need) do { rc = ensure_buf(c, &(ptr), &(capvar), (need)); if (rc) return rc; } while (0)
    if (pl->sk) {
        if (!level2_only && !listed) ENSURE(P.a_recs, P.a_recs_cap, np1 * nseg1 * pl->cap1 * pieces * rw);
        if (listed && c->want_list) ENSURE(P.a_recs, P.a_recs_cap, units + units / 4);  // (the solid list's room, as a first level would have left it)
        // (long records into a table nothing vouches for: the second level's buffer first serves as the sample's scratch set, 64 MB)
        if (pl->b2 > 1) ENSURE(P.b_recs, P.b_recs_cap, std::max<uint64_t>(n_leaves * pl->cap2 * pieces * rw, lng && pl->guessed ? (1ull << 22) : 0));
        if (!level2_only) ENSURE(P.spill_recs, P.spill_recs_cap, pl->spill_cap * rw);
    } else {
        { uint64_t cap = P.a_cap; ENSURE(P.a_keys, cap, np1 * nseg1 * pl->cap1); P.a_cap = cap; }
        { uint64_t cap = P.b_cap; ENSURE(P.b_keys, cap, n_leaves * pl->cap2 * pieces); P.b_cap = cap; }
        { uint64_t cap = P.spill_cap; ENSURE(P.spill_keys, cap, pl->spill_cap); dummy = P.spill_cap; ENSURE(P.spill_hints, dummy, pl->spill_cap); P.spill_cap = cap; }
    }
    if (!pl->compact && !lng) {  // (the compact form keeps the read pointers inside the records)
        if (!level2_only && !listed) ENSURE(P.a_hints, P.a_hints_cap, np1 * nseg1 * pl->cap1 * pieces);
        ENSURE(P.b_hints, P.b_hints_cap, n_leaves * pl->cap2 * pieces);
    }
    if (!level2_only) ENSURE(P.seg_counts1, P.segs1_cap, np1 * nseg1 * pieces);
    ENSURE(P.cursors2, P.cursors2_cap, n_leaves * pieces);
    { uint64_t cap = P.leaves_cap; ENSURE(P.leaf_state, cap, n_leaves); dummy = P.leaves_cap; ENSURE(P.leaf_new, dummy, n_leaves); P.leaves_cap = cap; }
#undef ENSURE
    if (!P.flags) {  // four flags and, behind them, the spill counter: cleared by one fill, read by one copy
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&P.flags), 4 * sizeof(uint32_t) + sizeof(unsigned long long)));
        P.spill_count = reinterpret_cast<unsigned long long *>(P.flags + 4);
    }
    // (one launch clears all five: a fill each, 10 us apart, was 60 us in front of every run)
    hipLaunchKernelGGL(k_pipe_reset, dim3(256), dim3(256), 0, c->stream, P.seg_counts1, level2_only || listed ? 0 : np1 * nseg1 * pieces, P.cursors2, n_leaves * pieces,
                       P.leaf_state, P.leaf_new, n_leaves, level2_only ? nullptr : P.flags);
    HIPCHK(c, hipGetLastError());
    pl->sp = SpillView{P.spill_keys, P.spill_hints, P.spill_count, pl->spill_cap, P.flags};
    pl->sks = SkSpill{P.spill_recs, P.spill_count, pl->spill_cap, P.flags};
    return MC_OK;
}

// The occurrences the merge kernels handed on (count_pipeline.h ovf_push: no room within their stretch of the region) go
// into the table through the direct kernel, which follows the table's region chain; entries of leaves that were not
// committed are dropped (those leaves are merged again).  Only after a launch that has been over EVERY leaf: the regions
// behind must hold valid slots.  *n_parked (may be null): how many there were.
static int pipe_drain_handed_on(mc_ctx *c, uint64_t n_listed, bool lng = false)
{
    mc_ctx::Pipe &P = c->pipe;
    const uint64_t n = std::min<uint64_t>(n_listed, mc_ctx::OVF_CAP);
    if (n == 0) return MC_OK;
    // The direct kernel parks what IT cannot place (no room in the whole chain: a table far too small) in the same list:
    // the entries are moved aside first and the list starts again from 0 -- read and appended to in one launch, the
    // additions parked by the drain itself were wiped with the list's counter (15 000 of 21 M keys lost in a soak case with a
    // capacity hint a quarter of what the reads held).
    int rc = ensure_buf(c, &c->d_ovf_tmp, &c->ovf_tmp_cap, (uint64_t)mc_ctx::OVF_CAP);
    if (!rc) rc = ensure_buf(c, &c->d_ovf_leaf_tmp, &c->ovf_leaf_tmp_cap, (uint64_t)mc_ctx::OVF_CAP);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_ovf_tmp, c->d_ovf, n * sizeof(uint4), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ovf_leaf_tmp, c->d_ovf_leaf, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ctr + 7, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_add_parked, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d_ovf_tmp, n, c->view(), c->d_ovf_leaf_tmp, P.leaf_state,
                       lng && hash_bins(c) ? 1 : 0, lng && hash_bins(c) ? c->dup_l1_view() : DupL1{nullptr, nullptr, 0, 0, nullptr});
    HIPCHK(c, hipGetLastError());
    c->solid_tracked = false;  // (these additions were not watched for crossing the coverage threshold)
    c->solid_list_fresh = false;
    c->st.spill_keys += n;
    // what the drain parked: into a table of twice the regions, until nothing is left (pipe_finish sees the new size)
    for (int attempt = 0;; attempt++) {
        unsigned long long m = 0;
        HIPCHK(c, hipMemcpyAsync(c->h_scratch + 29, c->d_ctr + 7, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        m = c->h_scratch[29];
        if (m == 0) return MC_OK;
        if (m > mc_ctx::OVF_CAP || attempt >= 6)
            return fail(c, MC_EOVERFLOW, "k-mer table regions keep overflowing (%llu additions parked); pass a capacity_hint (distinct k-mers)", m);
        HIPCHK(c, hipMemcpyAsync(c->d_ovf_tmp, c->d_ovf, m * sizeof(uint4), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_ctr + 7, 0, sizeof(unsigned long long), c->stream));
        rc = table_grow(c, c->n_regions * 2);
        if (rc) return rc;
        hipLaunchKernelGGL(k_add_parked, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, c->d_ovf_tmp, (uint64_t)m, c->view());
        HIPCHK(c, hipGetLastError());
    }
}

// Nothing vouches for the table's size (no capacity hint) and it holds nothing yet: the level-1 scatter has just run, and
// its first bucket is a fair sample of the batch -- 1 / np1 of the minimizer bins, whatever the table's size.  Its
// distinct k-mers are counted into a scratch set, the table is replaced by one of the size they call for, and the second
// level and the merge are planned for that table: the batch is scattered once (before, the merge of 1024 leaves told the
// size, and then everything ran again: 31 ms of counting on configs[1] against 19 with a hint).
static int pipe_resize_by_sample(mc_ctx *c, PipePlan &pl, uint64_t n_records)
{
    mc_ctx::Pipe &P = c->pipe;
    if (!(pl.sk && pl.guessed && c->virgin && pl.b2 > 1 && pl.pieces == 1 && c->mm_k)) return MC_OK;
    static const bool off = getenv("MC_NO_SAMPLE_RESIZE") != nullptr;
    if (off) return MC_OK;
    constexpr uint64_t SET_SLOTS = 1ull << 23;  // 64 MB of P.b_recs, which the second level has not touched yet
    if (P.b_recs_cap * sizeof(uint4) < SET_SLOTS * 8) return MC_OK;
    uint64_t *set = reinterpret_cast<uint64_t *>(P.b_recs);
    HIPCHK(c, hipMemsetAsync(set, 0xFF, SET_SLOTS * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ctr + 2, 0, sizeof(unsigned long long), c->stream));
    if (pl.lng)
        hipLaunchKernelGGL(k_skl_sample_distinct, dim3(pl.nseg1), dim3(256), 0, c->stream, P.a_recs, P.seg_counts1, pl.cap1, c->cfg.k, set, SET_SLOTS - 1,
                           c->d_ctr + 2);
    else
        hipLaunchKernelGGL(k_sk_sample_distinct, dim3(pl.nseg1), dim3(256), 0, c->stream, P.a_recs, P.seg_counts1, pl.cap1, c->cfg.k, set, SET_SLOTS - 1,
                           c->d_ctr + 2);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_scratch + 24, c->d_ctr + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_scratch + 25, P.flags, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t fl[4];
    memcpy(fl, c->h_scratch + 25, sizeof fl);
    if (fl[0]) return MC_OK;  // (records were lost: pipe_finish sees the flag too and the batch is counted another way)
    const double est = (double)c->h_scratch[24] * (double)pl.np1 * 1.1 + 1024.0;
    c->dup.expected_keys = est;  // (what the key streams of dup_check.h are sized by in a context without a hint)
    if (est > 0.45 * (double)SET_SLOTS * (double)pl.np1) return MC_OK;  // (the scratch set was too full to count in: the old way)
    uint64_t want = regions_for(c, mm_slots_for(c, est, 0.36));
    if (pl.vleaf) {
        // a power of two of regions: the smaller of the two around `want` while it keeps the table under load 0.45 (0.36 x 1.25),
        // else the larger (load 0.23 at least); beyond 2^19 regions the records' ten leaf bits do not reach: the caller runs the
        // first level again in the two-array form
        uint64_t p2 = 2 * PT_MAX_BUCKETS;
        while (p2 * 2 <= want) p2 *= 2;
        if (est > 0.45 * (double)(p2 << c->sb)) p2 *= 2;
        uint64_t reach = (uint64_t)PT_MAX_BUCKETS * 1024;
        if (const char *e = getenv("MC_SK_VLEAF_MAX_REGIONS")) if (*e) reach = std::min<uint64_t>(reach, strtoull(e, nullptr, 10));  // (tests: the way back at a small size)
        if (p2 > reach) return 5;
        want = p2;
    }
    if (pl.lng && est > 0.40 * (double)(want << c->sb)) return 4;  // (hash keys need their bins roomy, mc_create: the caller takes the per-window form)
    if (pl.lng && want <= c->n_regions) { pl.guessed = false; return MC_OK; }  // (the table it was created with holds the batch)
    static const bool dbg = getenv("MC_INGEST_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[count] first bucket: %llu distinct k-mers, %.0f M expected in all; table of %llu regions, %llu wanted\n",
                     (unsigned long long)c->h_scratch[24], est / 1e6, (unsigned long long)c->n_regions, (unsigned long long)want);
    if (want != c->n_regions) {
        // the new table must split into the level-1 buckets the records are in
        if (want % pl.np1 || want / pl.np1 > PT_MAX_LEAVES2 || want / pl.np1 < 2) return MC_OK;
        table_release(c, c->slots, c->slots_bytes);
        c->slots = nullptr;
        int rc = table_alloc(c, want);
        if (rc) return rc;
        c->st.grows++;
        HIPCHK(c, hipMemsetAsync(c->d_ctr + 6, 0, sizeof(unsigned long long), c->stream));  // keys at the coverage threshold: none yet
        PipePlan p2 = pl;
        rc = pipe_prepare(c, pl.wb, &p2, n_records, pl.nseg1, 1, true, 0, pl.lng);
        if (rc) return rc;
        if (p2.np1 != pl.np1 || p2.cap1 != pl.cap1 || p2.g != 0) return fail(c, MC_EINVAL, "internal: the resized table does not keep the level-1 buckets");
        pl = p2;
    } else {
        // (the scratch set sat in the second level's buffer: nothing of it is read before it is written)
    }
    pl.guessed = false;  // the table was sized by what the batch holds
    return MC_OK;
}

// P2, P3 (with the retry after growing the table), spill drain and bookkeeping; ms1 = time of the
// level-1 scatter that filled the a_* buckets.  Returns 1 (nothing merged yet) when the streams
// overflowed even their spill list: the caller then counts the batch with the direct kernel.
static int pipe_finish(mc_ctx *c, PipePlan &pl, double ms1, bool p2_done = false, double ms2_exposed = 0, bool p1_pending = false,
                       bool may_rerun = false)
{   // p2_done: the caller ran P2 itself, piece by piece next to P1 (ms2_exposed = what of it outlasted P1)
    // p1_pending: the caller recorded ev_t[0], enqueued P1 and did not wait: P2 and P3 follow on the stream at once and
    // the host hears of all three together (a host round trip between two kernels leaves the device idle for tens of us)
    mc_ctx::Pipe &P = c->pipe;
    const uint64_t np1 = pl.np1, n_leaves = pl.n_leaves;
    const int k = c->cfg.k;
    int rc;
    double ms2 = ms2_exposed, ms3 = 0;
    // P3 reads the leaves: P2's output (one segment each), or P1's buckets directly when there is no second level
    const void *lk = pl.sk ? (pl.b2 > 1 ? (const void *)P.b_recs : (const void *)P.a_recs)
                           : (pl.b2 > 1 ? (const void *)P.b_keys : (const void *)P.a_keys);
    const uint32_t *lh = pl.compact ? nullptr : (pl.b2 > 1 ? P.b_hints : P.a_hints);  // (compact: the pointers sit in the records)
    const uint32_t *lc = pl.b2 > 1 ? P.cursors2 : P.seg_counts1;
    const uint64_t lcap = pl.b2 > 1 ? pl.cap2 : pl.cap1;
    const uint32_t lseg = pl.b2 > 1 ? pl.pieces : pl.nseg1;
    // The solid list (P3Emit): super-k-mer form with the leaves in b_recs, so that a_recs is free to take it, and a
    // threshold to track.  Each P3 workgroup owns a segment.
    const int p3_grid = (int)std::min<uint64_t>(n_leaves, 256 * 2 * 4 * (D2_THREADS < P3_THREADS ? P3_THREADS / D2_THREADS : 1));
    P3Emit emit{nullptr, nullptr, 0, P.flags + 2};
    static const bool no_list = getenv("MC_NO_SOLID_LIST") != nullptr;
    if (pl.sk && pl.b2 > 1 && c->mm_k && c->solid_tracked && c->cov_hint > 0 && !no_list && c->want_list) {
        if (!P.emit_counts) HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&P.emit_counts), 256 * 2 * 4 * (D2_THREADS < P3_THREADS ? P3_THREADS / D2_THREADS : 1) * sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(P.emit_counts, 0, 256 * 2 * 4 * (D2_THREADS < P3_THREADS ? P3_THREADS / D2_THREADS : 1) * sizeof(uint32_t), c->stream));
        emit.recs = P.a_recs;
        emit.counts = P.emit_counts;
        emit.seg_cap = std::min<uint64_t>(P.a_recs_cap / (uint64_t)p3_grid, 0xFFFFFFF0ull);
    }
    // super-k-mer records, one region and one segment per leaf: the merge kernel that merges identical records first
    // (count_pipeline.h k_p3_dedup; MC_P3_DEDUP=0: the general kernel)
    static const bool dedup_on = [] { const char *e = getenv("MC_P3_DEDUP"); return !(e && !strcmp(e, "0")); }();
    auto launch_p3_n = [&](uint32_t leaves, int grid, int virgin) {
#define P3_ARGS lk, lh, lc, lcap, lseg, leaves, pl.g, c->view(), virgin, P.leaf_state, P.leaf_new, P.flags + 1, \
                (uint32_t)(c->solid_tracked ? c->cov_hint : 0), c->d_ctr + 6, k, emit, c->ptr_tries, P.flags
#define P3D_ARGS static_cast<const uint4 *>(lk), lh, lc, lcap, leaves, c->view(), P.leaf_state, P.leaf_new, P.flags + 1, \
                 (uint32_t)(c->solid_tracked ? c->cov_hint : 0), c->d_ctr + 6, k, emit, c->ptr_tries, P.flags
        if (pl.lng) {
#define P3L_ARGS static_cast<const uint4 *>(lk), lc, lcap, leaves, c->view(), virgin, P.leaf_state, P.leaf_new, P.flags + 1, \
                 (uint32_t)(c->solid_tracked ? c->cov_hint : 0), c->d_ctr + 6, k, emit, P.flags, c->dup_l1_view()
            // (built for the k-mer lengths people run -- constant shifts, trip counts and powers of 5 --, and for any other from the argument)
            if (k == 63) hipLaunchKernelGGL(k_p3_long<63>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3L_ARGS);
            else if (k == 55) hipLaunchKernelGGL(k_p3_long<55>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3L_ARGS);
            else if (k == 47) hipLaunchKernelGGL(k_p3_long<47>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3L_ARGS);
            else if (k == 41) hipLaunchKernelGGL(k_p3_long<41>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3L_ARGS);
            else hipLaunchKernelGGL(k_p3_long<0>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3L_ARGS);
#undef P3L_ARGS
        } else if (pl.sk && pl.g == 0 && lseg == 1 && dedup_on) {
            const bool one_gpu = c->ptr_tries == 1 && (emit.recs == nullptr || !(c->solid_tracked && c->cov_hint > 0));
            if (virgin && one_gpu && k == 31) hipLaunchKernelGGL((k_p3_dedup<true, true, true>), dim3(grid), dim3(D2_THREADS), 0, c->stream, P3D_ARGS);
            else if (virgin && one_gpu) hipLaunchKernelGGL((k_p3_dedup<true, true>), dim3(grid), dim3(D2_THREADS), 0, c->stream, P3D_ARGS);
            else if (virgin) hipLaunchKernelGGL((k_p3_dedup<true, false>), dim3(grid), dim3(D2_THREADS), 0, c->stream, P3D_ARGS);
            else if (one_gpu) hipLaunchKernelGGL((k_p3_dedup<false, true>), dim3(grid), dim3(D2_THREADS), 0, c->stream, P3D_ARGS);
            else hipLaunchKernelGGL((k_p3_dedup<false, false>), dim3(grid), dim3(D2_THREADS), 0, c->stream, P3D_ARGS);
            // (it leaves the leaves of more than DD_MAX_CAP records alone: where the capacity allows such leaves, the general
            // kernel follows at once and takes what is left -- it skips the merged ones, ~20 us when that is all of them)
            if (lcap > DD_MAX_CAP) hipLaunchKernelGGL(k_p3_merge<true>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3_ARGS);
        } else if (pl.sk) {
            hipLaunchKernelGGL(k_p3_merge<true>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3_ARGS);
        } else {
            hipLaunchKernelGGL(k_p3_merge<false>, dim3(grid), dim3(P3_THREADS), 0, c->stream, P3_ARGS);
        }
#undef P3_ARGS
#undef P3D_ARGS
    };
    auto launch_p3 = [&] { launch_p3_n((uint32_t)n_leaves, p3_grid, c->virgin ? 1 : 0); };
    // hash keys in minimizer bins: the merge kernel also leaves every key it writes back in the key streams of dup_check.h, sized
    // by what is known about the number of keys the table will hold
    if (pl.lng && hash_bins(c) && pl.g == 0)
        dup_arm(c, c->cfg.capacity_hint ? (double)c->cfg.capacity_hint : c->dup.expected_keys, (uint32_t)p3_grid);
    else
        c->dup.l1_armed = false;
    uint32_t flags[3] = {0, 0, 0};
    unsigned long long n_spill = 0, n_handed_on = 0;
    const bool virgin0 = c->virgin;
    auto launch_p2 = [&] {
        static const bool staged = [] { const char *e = getenv("MC_SK2_STAGED"); return !(e && !strcmp(e, "0")); }();
        static bool told = getenv("MC_INGEST_DEBUG") == nullptr;
        if (!told) {  // (where the streams lie: the second level's time differs between processes that differ in nothing else)
            told = true;
            fprintf(stderr, "[count] level-1 stream %p (%llu records a segment), level-2 stream %p (%llu a leaf), table %p\n", (void *)P.a_recs,
                    (unsigned long long)pl.cap1, (void *)P.b_recs, (unsigned long long)pl.cap2, (void *)c->slots);
        }
        if (pl.lng)
            hipLaunchKernelGGL((k_sk2_scatter_compact<2, 2>), dim3((unsigned)np1), dim3(PT_THREADS), 0, c->stream, P.a_recs, pl.cap1,
                               P.seg_counts1, (uint32_t)np1, pl.b2, P.cursors2, pl.cap2, P.b_recs, pl.sks, pl.nseg1, c->cur_ptr_base, pl.pos0,
                               (uint64_t)pl.chunk_tiles * P1L_TILE);
        else if (pl.sk && pl.compact && !pl.listed)
            hipLaunchKernelGGL(k_sk2_scatter_compact<MC_SK2C_ITEMS>, dim3((unsigned)np1), dim3(PT_THREADS), 0, c->stream, P.a_recs, pl.cap1,
                               P.seg_counts1, (uint32_t)np1, pl.b2, P.cursors2, pl.cap2, P.b_recs, pl.sks, pl.nseg1, c->cur_ptr_base, pl.pos0,
                               (uint64_t)pl.chunk_tiles * P1W_TILE, pl.vleaf ? 10u - (uint32_t)__builtin_ctz(pl.b2) : 0u);
        else if (pl.listed)
            hipLaunchKernelGGL((k_sk2_scatter_staged<MC_SK2_ITEMS, true, false>), dim3((unsigned)np1), dim3(PT_THREADS), 0, c->stream, pl.in_recs, pl.in_ptrs, (uint64_t)0,
                               P.seg_counts1, (uint32_t)np1, pl.b1, pl.b2, P.cursors2, pl.cap2, P.b_recs, nullptr, pl.sks, pl.nseg1, P.skb_seg_start, nullptr);
        else if (pl.sk && staged && pl.pieces == 1)
            hipLaunchKernelGGL(k_sk2_scatter_staged<MC_SK2_ITEMS>, dim3((unsigned)np1), dim3(PT_THREADS), 0, c->stream, P.a_recs, P.a_hints, pl.cap1,
                               P.seg_counts1, (uint32_t)np1, pl.b1, pl.b2, P.cursors2, pl.cap2, P.b_recs, P.b_hints, pl.sks, pl.nseg1);
        else if (pl.sk)
            hipLaunchKernelGGL(k_sk2_scatter, dim3((unsigned)np1), dim3(PT_THREADS), 0, c->stream, P.a_recs, P.a_hints, pl.cap1,
                               P.seg_counts1, (uint32_t)np1, pl.b1, pl.b2, P.cursors2, pl.cap2, P.b_recs, P.b_hints, pl.sks, pl.nseg1);
        else
            hipLaunchKernelGGL(k_p2_scatter, dim3((unsigned)np1), dim3(PT_THREADS), 0, c->stream, P.a_keys, P.a_hints, pl.cap1,
                               P.seg_counts1, (uint32_t)np1, pl.b1, pl.b2, P.cursors2, pl.cap2, P.b_keys, P.b_hints, pl.sp, c->mm_k);
    };
    // how many distinct keys the batch holds, judged by the leaves merged so far (a table that held nothing before)
    auto resize_for_batch = [&](uint64_t sample_leaves) -> int {
        std::vector<uint32_t> st(sample_leaves), nw(sample_leaves);
        HIPCHK(c, hipMemcpy(st.data(), P.leaf_state, sample_leaves * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(nw.data(), P.leaf_new, sample_leaves * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t merged = 0, added = 0;
        for (uint64_t i = 0; i < sample_leaves; i++)
            if (st[i]) { merged++; added += nw[i]; }
        {   // occurrences the merged leaves handed on to the next region (ovf_push) are keys too, nearly all of them new
            unsigned long long ho = 0;
            HIPCHK(c, hipMemcpy(&ho, c->d_ctr + 7, sizeof ho, hipMemcpyDeviceToHost));
            added += std::min<uint64_t>(ho, mc_ctx::OVF_CAP);
            if (ho > mc_ctx::OVF_CAP) added += (uint64_t)((double)(ho - mc_ctx::OVF_CAP));
        }
        // (the leaves that overflowed hold more than the average: at least a full leaf each)
        const uint64_t per_full = (uint64_t)REGION_SLOTS << pl.g;
        const double est = ((double)added + (double)(sample_leaves - merged) * (double)per_full * 1.5) * ((double)n_leaves / (double)sample_leaves) * 1.1 + 1024.0;
        const double load = c->mm_k ? 0.36 : 0.6;
        if (merged == sample_leaves && est <= (c->mm_k ? 0.5 : 0.7) * (double)c->n_slots()) return MC_OK;  // it fits: carry on
        uint64_t want = regions_for(c, c->mm_k ? mm_slots_for(c, est, load) : (uint64_t)(est / load));
        if (want <= c->n_regions) want = regions_for(c, c->n_slots() * 2);
        static const bool dbg = getenv("MC_INGEST_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[count] table too small: %llu of %llu sampled leaves merged, %llu keys in them: %.0f M keys expected; new table %.1f GB\n",
                         (unsigned long long)merged, (unsigned long long)sample_leaves, (unsigned long long)added, est / 1e6, (double)(want << c->sb) * 16 / 1e9);
        table_release(c, c->slots, c->slots_bytes);
        c->slots = nullptr;
        int r = table_alloc(c, want);  // (fresh device memory comes zeroed by the driver at ~30 GB/s: 0.7 s for 22 GB, once)
        if (r) return r;
        c->st.grows++;
        HIPCHK(c, hipMemsetAsync(c->d_ctr + 6, 0, 2 * sizeof(unsigned long long), c->stream));  // keys at the coverage threshold: none yet; nothing handed on
        return 3;
    };
    if (may_rerun && virgin0 && pl.guessed && n_leaves >= 8192 && pl.b2 > 1 && !p2_done) {
        // Nothing vouches for the table's size (no capacity hint): merge 1024 leaves first -- the bin word is a mixed hash,
        // any range of leaves is a fair sample -- and see what they hold.  Costs a host round trip; a whole merge into a
        // table that turns out too small costs seven times the run.
        const uint64_t K = 1024;
        HIPCHK(c, hipEventRecord(c->ev_t[1], c->stream));
        launch_p2();
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev_t[2], c->stream));
        launch_p3_n((uint32_t)K, (int)K, 1);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(flags, P.flags, sizeof flags, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float f = 0;
        if (p1_pending) { HIPCHK(c, hipEventElapsedTime(&f, c->ev_t[0], c->ev_t[1])); ms1 = f; p1_pending = false; }
        HIPCHK(c, hipEventElapsedTime(&f, c->ev_t[1], c->ev_t[2]));
        ms2 += f;
        p2_done = true;
        if (flags[0]) return 1;  // (records lost: nothing was merged)
        rc = resize_for_batch(K);
        if (rc == 3) {
            c->st.p1_ms += ms1;
            c->st.p2_ms += ms2;
            c->st.count_ms += ms1 + ms2;
            c->st.count_total_ms += ms1 + ms2;
        }
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(P.flags + 1, 0, sizeof(uint32_t), c->stream));
    }
    {   // P2 and the first P3 back to back.  The capacities of the streams are estimates (and a few heavy keys can fill
        // a bucket's spill list alone): P3 looks at the "records lost" flag itself and merges nothing when it is set, so
        // that the caller can still count the batch another way.
        HIPCHK(c, hipEventRecord(c->ev_t[1], c->stream));
        if (pl.b2 > 1 && !p2_done) {
            launch_p2();
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipEventRecord(c->ev_t[2], c->stream));
        launch_p3();
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev_t[3], c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_scratch + 16, P.flags, 4 * sizeof(uint32_t) + sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_scratch + 19, c->d_ctr + 7, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(flags, c->h_scratch + 16, sizeof flags);
        n_spill = c->h_scratch[18];
        n_handed_on = c->h_scratch[19];
        float f = 0;
        if (p1_pending) { HIPCHK(c, hipEventElapsedTime(&f, c->ev_t[0], c->ev_t[1])); ms1 = f; }
        if (pl.b2 > 1 && !p2_done) { HIPCHK(c, hipEventElapsedTime(&f, c->ev_t[1], c->ev_t[2])); ms2 += f; }
        if (flags[0]) return 1;  // (nothing was merged)
        HIPCHK(c, hipEventElapsedTime(&f, c->ev_t[2], c->ev_t[3]));
        ms3 += f;
        c->virgin = false;
    }
    if (flags[1] && virgin0 && !flags[0] && may_rerun) {  // (may_rerun: the caller can run the batch again -- returns 3)
        // The table held nothing before this run and is too small for it (no capacity hint, or a poor one).  Rebuilding it
        // by rehashing and merging the leaves that are left in two sweeps each costs 40 times the run (785 against 19 ms
        // on configs[1]); what the merged leaves added says how many distinct keys the batch holds, so: a fresh table of
        // the right size, and the caller runs the batch again.
        rc = resize_for_batch(n_leaves);
        if (rc != 3) return rc ? rc : fail(c, MC_EOVERFLOW, "internal: a leaf overflowed in a table that should hold the batch");
        c->st.p1_ms += ms1;
        c->st.p2_ms += ms2;
        c->st.p3_ms += ms3;
        c->st.count_ms += ms1 + ms2 + ms3;
        c->st.count_total_ms += ms1 + ms2 + ms3;
        return 3;
    }
    // P3 again, with a larger table, while a region overflows
    for (int attempt = 0;; attempt++) {
        if (attempt > 0) {
            rc = timed(c, &ms3, [&] { launch_p3(); });
            if (rc) return rc;
            c->virgin = false;
            HIPCHK(c, hipMemcpy(flags, P.flags, sizeof flags, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(&n_handed_on, c->d_ctr + 7, sizeof n_handed_on, hipMemcpyDeviceToHost));
        }
        if (flags[0]) return fail(c, MC_EOVERFLOW, "internal: spill list of the partitioned counting pipeline overflowed");
        // (every leaf has been merged or, where the hand-on list ran full, left as a valid region: the list can go in)
        const bool bins_before = hash_bins(c);
        rc = pipe_drain_handed_on(c, n_handed_on, pl.lng);
        if (rc) return rc;
        n_handed_on = 0;
        if (pl.lng && bins_before && !hash_bins(c))  // (the drain moved the table to hash-prefix regions: n_used was recounted, the merged leaves' keys included)
            HIPCHK(c, hipMemsetAsync(P.leaf_new, 0, n_leaves * sizeof(uint32_t), c->stream));
        if (!pl.lng && (n_leaves << pl.g) < c->n_regions) {  // (the drain doubled the table: a leaf covers more regions now; n_used was recounted)
            while ((n_leaves << pl.g) < c->n_regions) pl.g++;
            HIPCHK(c, hipMemsetAsync(P.leaf_new, 0, n_leaves * sizeof(uint32_t), c->stream));
        }
        if (flags[2]) emit.recs = nullptr;  // a segment of the solid list overflowed: the BFS set-up sweeps the table instead
        if (!flags[1]) break;
        if (pl.lng || attempt >= 6 || pl.g >= 5) {
            // (long records: a table of hash keys in minimizer bins cannot be rebuilt with more bins -- it gives them up at the first leaf that fails)
            if (!(pl.sk && (c->mm_k || pl.lng)))
                return fail(c, MC_EOVERFLOW, "k-mer table regions keep overflowing; pass a capacity_hint (distinct k-mers)");
            // minimizer bins that no number of regions can hold: regions by the key's hash from here on, and the leaves
            // that were not merged go through the direct kernel, as many at a time as the table has room for
            rc = to_hash_regions(c, 4);
            if (rc) return rc;
            HIPCHK(c, hipMemsetAsync(P.leaf_new, 0, n_leaves * sizeof(uint32_t), c->stream));  // (n_used was recounted)
            std::vector<uint32_t> st(n_leaves), cnt(n_leaves * (uint64_t)lseg);
            HIPCHK(c, hipMemcpy(st.data(), P.leaf_state, n_leaves * sizeof(uint32_t), hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(cnt.data(), lc, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            uint32_t lo_leaf = 0;
            while (lo_leaf < n_leaves) {
                uint64_t allowed, need = 0;
                uint32_t hi_leaf = lo_leaf;
                rc = table_reserve(c, 1ull << 26, &allowed);
                if (rc) return rc;
                while (hi_leaf < n_leaves) {
                    uint64_t w = 0;
                    if (!st[hi_leaf])
                        for (uint32_t sg = 0; sg < lseg; sg++) w += std::min<uint64_t>(cnt[(uint64_t)hi_leaf * lseg + sg], lcap) * (pl.lng ? SKL_MAX_WINDOWS : SK_MAX_WINDOWS);
                    if (hi_leaf > lo_leaf && need + w > allowed) break;
                    need += w;
                    hi_leaf++;
                }
                if (need > allowed) {  // a single leaf above what one launch may add: make room for it
                    rc = table_reserve(c, need, &allowed);
                    if (rc) return rc;
                    if (need > allowed) return fail(c, MC_EOVERFLOW, "internal: a leaf of %llu k-mer occurrences does not fit one launch", (unsigned long long)need);
                }
                if (need && pl.lng)
                    hipLaunchKernelGGL(k_skl_add_unmerged, dim3(std::min<uint32_t>(hi_leaf - lo_leaf, 4096u)), dim3(256), 0, c->stream,
                                       static_cast<const uint4 *>(lk), lc, lcap, lo_leaf, hi_leaf, P.leaf_state, k, c->view());
                else if (need)
                    hipLaunchKernelGGL(k_sk_add_unmerged, dim3(std::min<uint32_t>(hi_leaf - lo_leaf, 4096u)), dim3(256), 0, c->stream,
                                       static_cast<const uint4 *>(lk), lh, lc, lcap, lseg, lo_leaf, hi_leaf, P.leaf_state, k, c->view());
                HIPCHK(c, hipGetLastError());
                lo_leaf = hi_leaf;
            }
            emit.recs = nullptr;
            break;
        }
        rc = table_grow(c, c->n_regions * 2);
        if (rc) return rc;
        pl.g++;
        HIPCHK(c, hipMemsetAsync(P.flags + 1, 0, sizeof(uint32_t), c->stream));
        // the rebuild recounted n_used, keys of the leaves merged so far included
        HIPCHK(c, hipMemsetAsync(P.leaf_new, 0, n_leaves * sizeof(uint32_t), c->stream));
    }
    hipLaunchKernelGGL(k_sum_leaf_new, dim3(64), dim3(256), 0, c->stream, P.leaf_new, (uint32_t)n_leaves, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    // what did not fit its bucket goes through the direct kernel (n_spill: read with the flags above)
    double ms4 = 0;
    if (n_spill) {
        const uint32_t thr = (uint32_t)(c->solid_tracked ? c->cov_hint : 0);
        uint64_t i = 0;
        if (pl.lng && hash_bins(c)) {
            // (the table was sized by the hint that let this run take long records, and the direct path's usual preparations would
            // move it out of its minimizer bins: the few records that found no room in the streams go in as they are, each to its bin)
            rc = timed(c, &ms4, [&] {
                hipLaunchKernelGGL(k_skl_add_records, dim3(grid_for(n_spill, 256)), dim3(256), 0, c->stream, P.spill_recs, (uint64_t)n_spill, k, c->view(), thr, c->d_ctr + 6,
                                   c->dup_l1_view());
            });
            if (rc) return rc;
            i = n_spill;
            unsigned long long parked = 0;  // what found no room in its bin's chain is a key without a leaf: only a by-key table takes it
            HIPCHK(c, hipMemcpy(&parked, c->d_ctr + 7, sizeof parked, hipMemcpyDeviceToHost));
            if (parked) { rc = drain_parked(c); if (rc) return rc; }
        }
        while (i < n_spill) {
            uint64_t allowed;
            rc = table_reserve(c, (n_spill - i) * (pl.lng ? SKL_MAX_WINDOWS : pl.sk ? SK_MAX_WINDOWS : 1), &allowed);
            if (rc) return rc;
            const uint64_t m = std::min<uint64_t>(pl.lng ? std::max<uint64_t>(allowed / SKL_MAX_WINDOWS, 1) : pl.sk ? std::max<uint64_t>(allowed / SK_MAX_WINDOWS, 1) : allowed, n_spill - i);
            rc = timed(c, &ms4, [&] {
                if (pl.lng)
                    hipLaunchKernelGGL(k_skl_add_records, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, P.spill_recs + 2 * i, m, k, c->view(), thr, c->d_ctr + 6,
                                       DupL1{nullptr, nullptr, 0, 0, nullptr});  // (by key: the table is in hash-prefix regions by now)
                else if (pl.sk)
                    hipLaunchKernelGGL(k_sk_add_records, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, P.spill_recs + i, (const uint32_t *)nullptr, m, k,
                                       c->view(), thr, c->d_ctr + 6);
                else
                    hipLaunchKernelGGL(k_add_keys_hint, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, P.spill_keys + i,
                                       P.spill_hints + i, m, c->view(), thr, c->d_ctr + 6);
            });
            if (rc) return rc;
            i += m;
        }
    }
    if (emit.recs && !n_spill) {  // (spilled records went into the table behind P3's back)
        c->solid_list_fresh = true;
        c->solid_list_segs = (uint32_t)p3_grid;
        c->solid_list_segcap = emit.seg_cap;
    }
    c->st.p1_ms += ms1;
    c->st.p2_ms += ms2;
    c->st.p3_ms += ms3;
    c->st.spill_keys += n_spill;
    c->st.count_ms += ms1 + ms2 + ms3 + ms4;
    c->st.count_total_ms += ms1 + ms2 + ms3 + ms4;
    c->st.count_launches++;
    c->st.windows += pl.wb;
    return MC_OK;
}

static void launch_p1_reads(mc_ctx *c, const uint64_t *d_words, const uint64_t *offs, uint64_t nr, uint64_t base0,
                            uint64_t end_abs, uint64_t n_tiles_abs, const uint32_t *tile_first, uint32_t b1, uint32_t *cursors,
                            uint64_t cap, uint64_t *out_keys, uint32_t *out_hints, const SpillView &sp, int owners_mode,
                            const uint64_t *bases)
{   // owners_mode: 0 = hash-prefix buckets (counting), 1 = count per owner, 2 = scatter per owner at `bases`
    // counting mode: exactly PT_SEGMENTS workgroups, each owns one segment of every bucket
    const int grid = owners_mode == 0 ? PT_SEGMENTS
                                      : (int)std::min<uint64_t>(std::max<uint64_t>(n_tiles_abs - base0 / PT_TILE, 1), 256);
    const int k = c->cfg.k;
    // (owner modes: the kernel's mm_k chooses the owner rule -- 0: mc_key_owner, the key's own hash; k: the owner of the key's minimizer)
    const int p1_mm_k = owners_mode == 0 ? c->mm_k : (c->extract_by_minimizer ? k : 0);
#define P1_ARGS d_words, offs, nr, base0, end_abs, n_tiles_abs, tile_first, k, b1, cursors, cap, out_keys, out_hints, c->d_ctr + 1, sp, bases, p1_mm_k, c->cur_ptr_base
#define P1_LAUNCH(MODE)                                                                                                  \
    do {                                                                                                                 \
        if (owners_mode == 0)                                                                                            \
            hipLaunchKernelGGL((k_p1_extract_scatter<MODE, false, false>), dim3(grid), dim3(PT_THREADS), 0, c->stream, P1_ARGS); \
        else if (owners_mode == 1)                                                                                       \
            hipLaunchKernelGGL((k_p1_extract_scatter<MODE, true, true>), dim3(grid), dim3(PT_THREADS), 0, c->stream, P1_ARGS);   \
        else                                                                                                             \
            hipLaunchKernelGGL((k_p1_extract_scatter<MODE, true, false>), dim3(grid), dim3(PT_THREADS), 0, c->stream, P1_ARGS);  \
    } while (0)
    switch (c->cfg.key_mode) {
    case MC_KEY_PACKED: P1_LAUNCH(KEY_PACKED); break;
    case MC_KEY_POLY: P1_LAUNCH(KEY_POLY); break;
    default: P1_LAUNCH(KEY_FNV1A);
    }
#undef P1_LAUNCH
#undef P1_ARGS
}

// super-k-mer records expected from `nr` reads with `wb` windows: runs of windows sharing a minimizer average
// (w + 1) / 2 windows in random sequence (w = k - SK_M + 1 minimizer candidates per window); sequencing errors
// and the 16-window limit cut some short (x 1.5 covers 1-2 % errors), and every read ends one
static uint64_t sk_records_bound(const mc_ctx *c, uint64_t wb, uint64_t nr)
{
    return (uint64_t)((double)wb * 2.0 / (double)(c->cfg.k - SK_M + 2) * 1.5) + 2 * nr + 1024;
}

// One batch of reads [r0, r1) through the partitioned pipeline (count_pipeline.h).  wb = its windows,
// base0 / end_abs = read_offsets[r0] / read_offsets[r1].
static int add_reads_partitioned_once(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t r0, uint64_t r1,
                                      uint64_t base0, uint64_t end_abs, uint64_t wb);
static uint64_t max_run_bases(const mc_ctx *c, double windows_per_base);
constexpr int RC_RESPLIT = 5;  // (internal) the run must be cut again: the table has left its minimizer bins under it and takes one record a window now
static int add_reads_partitioned(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t r0, uint64_t r1,
                                 uint64_t base0, uint64_t end_abs, uint64_t wb)
{
    // 3 from the pipeline: the (empty) table was too small and has been replaced by one of the right size -- once more
    unsigned long long empty0 = 0;  // occurrences of the key that looks like a free slot (hash keys only): counted apart, by P1
    const bool hashed = c->cfg.key_mode != MC_KEY_PACKED;
    if (hashed) {
        HIPCHK(c, hipMemcpyAsync(&empty0, c->d_ctr + 1, sizeof empty0, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int run = 0;; run++) {
        const int rc = add_reads_partitioned_once(c, d_words, d_off, r0, r1, base0, end_abs, wb);
        if (rc != 3) return rc;
        if (run >= 8) return fail(c, MC_EOVERFLOW, "k-mer table regions keep overflowing; pass a capacity_hint (distinct k-mers)");  // (a rerun multiplies the table by 4.5 at least)
        if (hashed) {
            HIPCHK(c, hipMemcpyAsync(c->d_ctr + 1, &empty0, sizeof empty0, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
}
// the reads [r0, r1) through the direct kernel, as many at a time as the table has room for
static int count_batch_direct(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t r0, uint64_t r1, uint64_t base0, uint64_t end_abs,
                              uint64_t wb)
{
    const uint64_t nr = r1 - r0;
    uint64_t r = r0;
    while (r < r1) {
        uint64_t allowed;
        int rc = table_reserve(c, wb, &allowed);
        if (rc) return rc;
        const uint64_t step = std::max<uint64_t>(1, std::min<uint64_t>(r1 - r, allowed / std::max<uint64_t>(1, (end_abs - base0) / nr + 1)));
        double ms = 0;
        rc = timed(c, &ms, [&] { launch_count(c, d_words, d_off, r, r + step); });
        if (rc) return rc;
        c->st.count_ms += ms;
        c->st.count_total_ms += ms;
        c->st.count_launches++;
        r += step;
    }
    c->st.windows += wb;
    return MC_OK;
}

// One run as long records (count_long.h).  4: the run does not qualify -- the caller moves the table to hash-prefix regions and
// takes the per-window form; otherwise what pipe_finish returns.
static int add_reads_long(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t r0, uint64_t r1, uint64_t base0, uint64_t end_abs,
                          uint64_t wb)
{
    mc_ctx::Pipe &P = c->pipe;
    const uint64_t nr = r1 - r0;
    if (c->cfg.key_mode != MC_KEY_POLY || c->cfg.k < SKL_MIN_K || c->cfg.k > SKL_MAX_K) return 4;
    unsigned long long used;
    uint32_t fatal;
    int rc = dup_unmerge(c);  // (slots that hold their keys' sums get their own counts back: the merge kernel adds to whichever slot a bin holds)
    if (!rc) rc = read_counters(c, &used, &fatal);
    if (rc) return rc;
    // (only a table sized for what it will hold -- by a hint that still holds, or, holding nothing yet, by a sample of this batch: it cannot grow)
    if (fatal || !((c->cfg.capacity_hint && used < c->cfg.capacity_hint) || c->virgin)) return 4;
    PipePlan pl;
    // (bin words of the two smallest hashes: runs two thirds as long -- 10.6 windows a record for 15.9 on configs[2]'s reads)
    const uint64_t n_records = c->mm_k < 0 ? sk_records_bound(c, wb, nr) * 8 / 5 : sk_records_bound(c, wb, nr);
    const uint64_t n_tiles_abs = (end_abs + P1L_TILE - 1) / P1L_TILE;
    rc = pipe_prepare(c, wb, &pl, n_records, (uint32_t)P1W_SEGMENTS, 1, false, n_tiles_abs - base0 / P1L_TILE, true);
    if (rc) return rc;
    if (!pl.compact || !hash_bins(c)) return 4;
    const uint64_t *offs = d_off + r0;
    rc = ensure_buf(c, &P.tile_first, &P.tiles1_cap, n_tiles_abs);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_t[0], c->stream));  // (no wait here: pipe_finish enqueues the second level and the merge right behind)
    launch_tile_first(c, offs, nr, n_tiles_abs, P.tile_first, P1L_TILE);
    pl.pos0 = base0 / P1L_TILE * P1L_TILE;
    if (getenv("MC_INGEST_DEBUG"))
        fprintf(stderr, "[count] long records: %u buckets x %u leaves, %u tiles a segment from base %llu, %llu records expected\n", pl.b1, pl.b2, pl.chunk_tiles,
                (unsigned long long)pl.pos0, (unsigned long long)n_records);
    const SklSpill sp{P.spill_recs, P.spill_count, pl.spill_cap, P.flags};
    hipLaunchKernelGGL(k_skl_extract, dim3(P1W_SEGMENTS), dim3(P1W_THREADS), 0, c->stream, d_words, offs, nr, base0, end_abs, n_tiles_abs, P.tile_first,
                       c->cfg.k, pl.b1, P.seg_counts1, pl.cap1, P.a_recs, sp, pl.chunk_tiles, c->mm_k < 0 ? 1u : 0u);
    HIPCHK(c, hipGetLastError());
    rc = pipe_resize_by_sample(c, pl, n_records);  // (no hint: the table is sized by what the first bucket holds; 4: its bins would be too full)
    if (rc) return rc;
    if (pl.guessed) return 4;  // (no sample could be taken: nothing vouches for the table)
    c->st.long_runs++;
    return pipe_finish(c, pl, 0, false, 0, true, false);
}

static int add_reads_partitioned_once(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t r0, uint64_t r1,
                                      uint64_t base0, uint64_t end_abs, uint64_t wb)
{
    mc_ctx::Pipe &P = c->pipe;
    PipePlan pl;
    const uint64_t nr = r1 - r0;
    if (hash_bins(c)) {
        int lrc = add_reads_long(c, d_words, d_off, r0, r1, base0, end_abs, wb);
        if (lrc == 1) return count_batch_direct(c, d_words, d_off, r0, r1, base0, end_abs, wb);  // (the streams lost records: nothing was merged)
        if (lrc != 4) return lrc;
        lrc = to_hash_regions(c, 3);
        if (lrc) return lrc;
        // The run was cut for long records (2^34 bases); one record a window takes far less (39 M reads of 150 bases
        // at k = 63 pass the 32-bit bucket indices).  The caller cuts what is left again, for the table as it is now.
        if (end_abs - base0 > max_run_bases(c, (double)wb / (double)std::max<uint64_t>(end_abs - base0, 1))) return RC_RESPLIT;
    }
    const uint64_t n_records = c->mm_k ? sk_records_bound(c, wb, nr) : 0;
    // MC_PIPE_PIECES=n (an experiment, off by default): the reads go through the two scatter levels in n pieces, the
    // second level of piece p on a side stream next to the first level of piece p + 1; the merge kernel then finds n
    // segments per leaf.  Measured on configs[1] (P1 + exposed P2 + P3): 1 piece 6.1 + 2.7 + 9.3 = 18.2 ms, 2 pieces
    // 7.0 + 1.4 + 9.5 = 17.9, 4 pieces 7.6 + 0.7 + 10.9 = 19.2 -- the two levels slow each other down by nearly what
    // the overlap hides, and the merge pays for the shorter segments.
    uint32_t pieces = 1;
    if (const char *e = getenv("MC_PIPE_PIECES")) pieces = (uint32_t)std::min<unsigned long>(8, std::max<unsigned long>(1, strtoul(e, nullptr, 10)));
    // One key per window (hash keys, short k): a large batch goes through the two scatter levels in pieces of ~2^30 windows
    // that REUSE the level-1 buffers, one after the other on the same stream, so that a run of twice the windows fits
    // the same scratch and the whole table is rewritten half as often (max_run_bases).
    // (only where the memory is needed: at 0.9 G windows two pieces cost 55 ms against 35 ms in one)
    if (!n_records) pieces = wb < (3ull << 29) ? 1u : (uint32_t)std::min<uint64_t>(8, (wb + (1ull << 30) - 1) >> 30);
    int rc = pipe_prepare(c, wb, &pl, n_records, n_records ? (uint32_t)P1W_SEGMENTS : (uint32_t)PT_SEGMENTS, pieces, false,
                          n_records != 0 && pieces == 1 ? (end_abs + P1W_TILE - 1) / P1W_TILE - base0 / P1W_TILE : 0);
    if (rc) return rc;
    pieces = pl.pieces;
    const uint64_t *offs = d_off + r0;
    // tiles are cut over the absolute base positions [0, end_abs); the ones before base0 hold no read of ours
    const uint32_t tile_size = pl.sk ? P1W_TILE : (uint32_t)PT_TILE;
    const uint64_t n_tiles_abs = (end_abs + tile_size - 1) / tile_size;
    rc = ensure_buf(c, &P.tile_first, &P.tiles1_cap, n_tiles_abs);
    if (rc) return rc;
    double ms1 = 0;
    if (pieces > 1 && !pl.sk) {
        const uint64_t t_first = base0 / tile_size, per = (n_tiles_abs - t_first + pieces - 1) / pieces;
        // (the two levels alternate on one stream; an event behind every kernel books each level's time where it belongs --
        // round 3 booked both under the first level and reported k_p2_scatter: 0.0 for configs[2] at full size)
        for (uint32_t i = 0; i < 2 * pieces; i++)
            if (!c->ev_seq[i]) HIPCHK(c, hipEventCreate(&c->ev_seq[i]));
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        launch_tile_first(c, offs, nr, n_tiles_abs, P.tile_first, tile_size);
        for (uint32_t pc = 0; pc < pieces; pc++) {
            const uint64_t lo_t = t_first + pc * per, hi_t = std::min<uint64_t>(n_tiles_abs, lo_t + per);
            if (lo_t < hi_t)
                launch_p1_reads(c, d_words, offs, nr, pc == 0 ? base0 : lo_t * tile_size, end_abs, hi_t, P.tile_first, pl.b1, P.seg_counts1, pl.cap1,
                                P.a_keys, P.a_hints, pl.sp, 0, nullptr);
            HIPCHK(c, hipEventRecord(c->ev_seq[2 * pc], c->stream));
            if (lo_t < hi_t)
                hipLaunchKernelGGL(k_p2_scatter, dim3((unsigned)pl.np1), dim3(PT_THREADS), 0, c->stream, P.a_keys, P.a_hints, pl.cap1, P.seg_counts1,
                                   (uint32_t)pl.np1, pl.b1, pl.b2, P.cursors2, pl.cap2, P.b_keys, P.b_hints, pl.sp, c->mm_k, pc, pieces);
            HIPCHK(c, hipEventRecord(c->ev_seq[2 * pc + 1], c->stream));
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventSynchronize(c->ev_seq[2 * pieces - 1]));
        double ms2 = 0;
        for (uint32_t pc = 0; pc < pieces; pc++) {
            float f1 = 0, f2 = 0;
            HIPCHK(c, hipEventElapsedTime(&f1, pc ? c->ev_seq[2 * pc - 1] : c->ev0, c->ev_seq[2 * pc]));
            HIPCHK(c, hipEventElapsedTime(&f2, c->ev_seq[2 * pc], c->ev_seq[2 * pc + 1]));
            ms1 += f1;
            ms2 += f2;
        }
        rc = pipe_finish(c, pl, ms1, true, ms2, false, true);
    } else if (pieces > 1) {
        const uint64_t t_first = base0 / tile_size, per = (n_tiles_abs - t_first + pieces - 1) / pieces;
        const uint64_t a_stride = pl.np1 * pl.nseg1 * pl.cap1, c_stride = pl.np1 * pl.nseg1;
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        launch_tile_first(c, offs, nr, n_tiles_abs, P.tile_first, tile_size);
        for (uint32_t pc = 0; pc < pieces; pc++) {
            const uint64_t lo_t = t_first + pc * per, hi_t = std::min<uint64_t>(n_tiles_abs, lo_t + per);
            if (lo_t < hi_t)
                hipLaunchKernelGGL(k_sk1w_extract<false>, dim3(P1W_SEGMENTS), dim3(P1W_THREADS), 0, c->stream, d_words, offs, nr,
                                   pc == 0 ? base0 : lo_t * tile_size, end_abs, hi_t, P.tile_first, c->cfg.k, pl.b1, P.seg_counts1 + pc * c_stride,
                                   pl.cap1, P.a_recs + pc * a_stride, P.a_hints + pc * a_stride, pl.sks, c->cur_ptr_base);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->ev_piece[pc], c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->pipe_stream, c->ev_piece[pc], 0));
            hipLaunchKernelGGL(k_sk2_scatter, dim3((unsigned)pl.np1), dim3(PT_THREADS), 0, c->pipe_stream, P.a_recs + pc * a_stride,
                               P.a_hints + pc * a_stride, pl.cap1, P.seg_counts1 + pc * c_stride, (uint32_t)pl.np1, pl.b1, pl.b2, P.cursors2,
                               pl.cap2, P.b_recs, P.b_hints, pl.sks, pl.nseg1, 0, pc, pieces);
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_p2, c->pipe_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_p2, 0));
        HIPCHK(c, hipEventSynchronize(c->ev_p2));
        HIPCHK(c, hipEventSynchronize(c->ev1));
        float f1 = 0, f2 = 0;
        HIPCHK(c, hipEventElapsedTime(&f1, c->ev0, c->ev1));
        HIPCHK(c, hipEventElapsedTime(&f2, c->ev1, c->ev_p2));  // what the last piece's second level adds behind the first levels
        ms1 = f1;
        rc = pipe_finish(c, pl, ms1, true, f2 > 0 ? f2 : 0, false, true);
    } else {
        HIPCHK(c, hipEventRecord(c->ev_t[0], c->stream));  // (no wait here: pipe_finish enqueues P2 and P3 right behind)
        launch_tile_first(c, offs, nr, n_tiles_abs, P.tile_first, tile_size);
        if (pl.sk && pl.compact) {
            pl.pos0 = base0 / P1W_TILE * P1W_TILE;
            if (getenv("MC_INGEST_DEBUG"))
                fprintf(stderr, "[count] compact records: %u buckets x %u leaves%s, %u tiles a segment from base %llu\n", pl.b1, pl.vleaf ? 1024u : pl.b2,
                        pl.vleaf ? " (whatever the table: it is sized behind the first level)" : "", pl.chunk_tiles, (unsigned long long)pl.pos0);
            hipLaunchKernelGGL((k_sk1w_extract<false, true>), dim3(P1W_SEGMENTS), dim3(P1W_THREADS), 0, c->stream, d_words, offs, nr, base0, end_abs,
                               n_tiles_abs, P.tile_first, c->cfg.k, pl.b1, P.seg_counts1, pl.cap1, P.a_recs, nullptr, pl.sks, c->cur_ptr_base,
                               pl.chunk_tiles, pl.vleaf ? 1024u : pl.b2);
        }
        else if (pl.sk)
            hipLaunchKernelGGL(k_sk1w_extract<false>, dim3(P1W_SEGMENTS), dim3(P1W_THREADS), 0, c->stream, d_words, offs, nr, base0, end_abs,
                               n_tiles_abs, P.tile_first, c->cfg.k, pl.b1, P.seg_counts1, pl.cap1, P.a_recs, P.a_hints, pl.sks, c->cur_ptr_base);
        else
            launch_p1_reads(c, d_words, offs, nr, base0, end_abs, n_tiles_abs, P.tile_first, pl.b1, P.seg_counts1, pl.cap1, P.a_keys,
                            P.a_hints, pl.sp, 0, nullptr);
        HIPCHK(c, hipGetLastError());
        rc = pipe_resize_by_sample(c, pl, n_records);
        if (rc == 5 && !c->no_vleaf) {  // (the batch wants a table beyond the reach of the records' leaf bits: once more, the two-array way)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (getenv("MC_INGEST_DEBUG")) fprintf(stderr, "[count] the batch wants a table beyond the reach of its records' ten leaf bits: the first level again, two arrays\n");
            c->no_vleaf = true;
            rc = add_reads_partitioned(c, d_words, d_off, r0, r1, base0, end_abs, wb);
            c->no_vleaf = false;
            return rc;
        }
        if (rc) return rc;
        rc = pipe_finish(c, pl, ms1, false, 0, true, true);
    }
    if (rc != 1) return rc;
    // (super-k-mer streams overflowed: unusually short runs) count this batch with the direct kernel instead
    return count_batch_direct(c, d_words, d_off, r0, r1, base0, end_abs, wb);
}


static int add_reads_partitioned_any(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t r0, uint64_t r1, uint64_t base0,
                                     uint64_t end_abs, uint64_t wb)
{
    return add_reads_partitioned(c, d_words, d_off, r0, r1, base0, end_abs, wb);
}

// A flat stream of keys (+ optional hints) through the partitioned pipeline: the keys a rank owns
// after the multi-GPU exchange.
static int add_keys_partitioned(mc_ctx *c, const uint64_t *d_keys, const uint32_t *d_hints, uint64_t n)
{
    mc_ctx::Pipe &P = c->pipe;
    PipePlan pl;
    int rc = by_key_ready(c);
    if (!rc) rc = pipe_prepare(c, n, &pl);
    if (rc) return rc;
    double ms1 = 0;
    rc = timed(c, &ms1, [&] {
        const uint64_t n_tiles = (n + PT_TILE - 1) / PT_TILE;
        (void)n_tiles;
        hipLaunchKernelGGL(k_p1_keys_scatter, dim3(PT_SEGMENTS), dim3(PT_THREADS), 0, c->stream,
                           d_keys, d_hints, n, pl.b1, P.seg_counts1, pl.cap1, P.a_keys, P.a_hints, c->d_ctr + 1, pl.sp, c->mm_k);
    });
    if (rc) return rc;
    rc = pipe_finish(c, pl, ms1);
    if (rc != 1) return rc;
    // (the streams overflowed: a few keys make up most of the batch) the direct kernel instead
    c->solid_tracked = false;
    c->solid_list_fresh = false;
    for (uint64_t i = 0; i < n;) {
        uint64_t allowed;
        rc = table_reserve(c, n - i, &allowed);
        if (rc) return rc;
        const uint64_t m = std::min<uint64_t>(allowed, n - i);
        if (d_hints)
            hipLaunchKernelGGL(k_add_keys_hint, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, d_keys + i, d_hints + i, m, c->view());
        else
            hipLaunchKernelGGL(k_add_keys, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, reinterpret_cast<const int64_t *>(d_keys) + i, m, c->view());
        HIPCHK(c, hipGetLastError());
        i += m;
    }
    return MC_OK;
}

// a stream of records through the direct kernel (what the partitioned runs fall back on when their streams overflow)
static int add_records_direct(mc_ctx *c, const uint4 *d_recs, const uint32_t *d_bins, uint64_t n)
{
    c->solid_tracked = false;
    c->solid_list_fresh = false;
    for (uint64_t i = 0; i < n;) {
        uint64_t allowed;
        int rc = table_reserve(c, (n - i) * SK_MAX_WINDOWS, &allowed);
        if (rc) return rc;
        const uint64_t m = std::min<uint64_t>(std::max<uint64_t>(allowed / SK_MAX_WINDOWS, 1), n - i);
        hipLaunchKernelGGL(k_sk_add_records, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, d_recs + i, d_bins + i, m, c->cfg.k, c->view(), 0u, c->d_ctr + 6);
        HIPCHK(c, hipGetLastError());
        i += m;
    }
    return MC_OK;
}

// A flat stream of super-k-mer records (+ bin words): what a rank owns after the multi-GPU exchange.
static int add_records_partitioned(mc_ctx *c, const uint4 *d_recs, const uint32_t *d_bins, uint64_t n)
{
    mc_ctx::Pipe &P = c->pipe;
    unsigned long long *sum = c->d_ctr + 2;
    HIPCHK(c, hipMemsetAsync(sum, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_sk_count_windows, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, d_recs, n, sum);
    HIPCHK(c, hipGetLastError());
    unsigned long long wb = 0;
    HIPCHK(c, hipMemcpy(&wb, sum, sizeof wb, hipMemcpyDeviceToHost));
    PipePlan pl;
    int rc = pipe_prepare(c, wb, &pl, n);
    if (rc) return rc;
    double ms1 = 0;
    rc = timed(c, &ms1, [&] {
        hipLaunchKernelGGL(k_sk1_records, dim3(PT_SEGMENTS), dim3(PT_THREADS), 0, c->stream, d_recs, d_bins, n, pl.b1, P.seg_counts1,
                           pl.cap1, P.a_recs, P.a_hints, pl.sks);
    });
    if (rc) return rc;
    rc = pipe_finish(c, pl, ms1);
    if (rc != 1) return rc;
    return add_records_direct(c, d_recs, d_bins, n);  // (the buckets overflowed)
}

// Records that arrive in level-1 bucket order already (the binned exchange, include/mcgpu.h mc_add_superkmers_binned_dev): part p of
// the buffer -- records part_off[p] .. part_off[p + 1] -- holds part_counts[p * fine + f] records of fine bucket f = mulhi32(bin word,
// fine), f ascending.  Where this table's level-1 buckets are unions of fine buckets the run starts at the second level, every part
// r = fine / np1 segments of every bucket; returns 2 where they are not (or the table has no second level, or the parts are too
// many): the caller counts the records as a flat stream.
static int add_records_binned(mc_ctx *c, const uint4 *d_recs, const uint32_t *d_ptrs, uint64_t n, uint64_t n_windows, uint32_t fine, uint32_t n_parts,
                              const uint64_t *part_off, const uint32_t *d_part_counts)
{
    mc_ctx::Pipe &P = c->pipe;
    {   // (the table may grow for the batch -- a context without a hint --: before the plan is looked at, as pipe_prepare would)
        bool hint_holds;
        int rc = pipe_reserve(c, n_windows, &hint_holds);
        if (rc) return rc;
    }
    uint64_t n_leaves, np1;
    uint32_t g;
    if (plan_levels(c, true, &n_leaves, &g, &np1)) return 2;
    if (n_leaves / np1 <= 1 || fine % np1 || n >= (1ull << 31)) return 2;
    const uint64_t r = fine / np1, nseg = (uint64_t)n_parts * r;
    if (nseg > (uint64_t)P1W_SEGMENTS || nseg == 0) return 2;
    int rc = ensure_buf(c, &P.seg_counts1, &P.segs1_cap, np1 * nseg);
    if (!rc) rc = ensure_buf(c, &P.skb_seg_start, &P.skb_seg_start_cap, np1 * nseg);
    if (!rc) rc = ensure_buf(c, &P.skb_small, &P.skb_small_cap, (uint64_t)P1W_SEGMENTS + 2 * PT_MAX_BUCKETS + 8);
    if (rc) return rc;
    // (the parts' starts go up through the context's pinned words; the kernel checks every part's counts against its length)
    std::vector<unsigned long long> po(part_off, part_off + n_parts + 1);
    po.push_back(0);  // [n_parts + 1]: the "counts do not add up" flag
    HIPCHK(c, hipMemcpyAsync(P.skb_small, po.data(), po.size() * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_skb_segments, dim3(n_parts), dim3(1024), 0, c->stream, P.skb_small, d_part_counts, fine, (uint32_t)r, (uint32_t)nseg, P.seg_counts1,
                       P.skb_seg_start, P.skb_small + n_parts + 1);
    HIPCHK(c, hipGetLastError());
    unsigned long long bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, P.skb_small + n_parts + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (po must outlive its copy, and nothing may read by counts that do not add up)
    if (bad) return fail(c, MC_EINVAL, "mc_add_superkmers_binned_dev: the fine-bucket counts of %llu part(s) do not add up to the parts' lengths", bad);
    if (!c->want_list && (P.a_recs || P.a_hints)) {
        // This run has no first level: the extraction's first-level streams (a fifth of the table at configs[3]'s size) go back to the
        // pool, where the second level's buffer -- about as large -- is taken from; the next extraction gets them back from there.
        HIPCHK(c, hipStreamSynchronize(c->stream));
        g_scratch_pool.put(c->cfg.device, P.a_recs, P.a_recs_cap * sizeof(uint4));
        g_scratch_pool.put(c->cfg.device, P.a_hints, P.a_hints_cap * 4);
        P.a_recs = nullptr; P.a_recs_cap = 0; P.a_hints = nullptr; P.a_hints_cap = 0;
        c->solid_list_fresh = false;
    }
    PipePlan pl;
    rc = pipe_prepare(c, n_windows, &pl, n, (uint32_t)nseg, 1, false, 0, false, true);
    if (rc) return rc;
    if (pl.np1 != np1 || pl.b2 <= 1) return fail(c, MC_EINVAL, "internal: the plan of a binned run changed under it");
    pl.in_recs = d_recs;
    pl.in_ptrs = d_ptrs;
    rc = pipe_finish(c, pl, 0.0);
    if (rc != 1) return rc;
    return add_records_direct(c, d_recs, d_ptrs, n);  // (the leaves' streams overflowed)
}

// Appends the packed bases [first_off, last_off) of a batch (word-aligned copy from the batch's own buffer) to the read
// store and sets cur_ptr_base so that store position = cur_ptr_base + position in the batch.  in_place >= 0: the batch
// already sits in the store from word `in_place` on (mc_add_reads_packed uploads straight into it).
static int rs_reserve(mc_ctx *c, uint64_t more_words)
{
    const uint64_t used = c->rs_words ? c->rs_end() / 32 : 0, need = std::max(c->rs_bases / 32 + more_words, used) + 2;  // (used: what must survive a move)
    if (need <= c->rs_cap_words) return MC_OK;
    uint64_t cap = std::max<uint64_t>(c->rs_cap_words * 2, std::max<uint64_t>(need, 1ull << 20));
    // (from the process-wide pool of large blocks, like the pipeline's scratch: a context's read store handed back with
    // hipFree is reclaimed lazily, and a later context's first launch waits for it)
    uint64_t *nw = nullptr;
    size_t got = 0;
    hipError_t e = g_scratch_pool.get(c->cfg.device, cap * 8, reinterpret_cast<void **>(&nw), &got);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        g_table_pool.release(c->cfg.device);
        g_scratch_pool.release(c->cfg.device);
        e = g_scratch_pool.get(c->cfg.device, cap * 8, reinterpret_cast<void **>(&nw), &got);
    }
    HIPCHK(c, e);
    cap = std::max<uint64_t>(cap, got / 8);
    if (used) HIPCHK(c, hipMemcpyAsync(nw, c->rs_words, used * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    g_scratch_pool.put(c->cfg.device, c->rs_words, c->rs_cap_words * 8);
    c->rs_words = nw;
    c->rs_cap_words = cap;
    return MC_OK;
}
static int rs_append(mc_ctx *c, const uint64_t *d_words, uint64_t first_off, uint64_t last_off, int64_t in_place = -1, bool side = false)
{
    c->cur_ptr_base = ~0ull;
    if ((!c->rs_enabled && !c->rs_virtual) || last_off <= first_off) return MC_OK;
    const uint64_t w0 = first_off / 32, w1 = (last_off + 31) / 32;  // words [w0, w1) hold the batch (+ a pad word behind)
    uint64_t at = c->rs_bases / 32;
    if (c->rs_virtual) {
        // (the words themselves travel to the context that keeps the store: mc_read_store_import_dev there, at rs_bases as it is now)
    } else if (in_place >= 0) {
        at = (uint64_t)in_place;
    } else {
        int rc = rs_reserve(c, w1 - w0 + 1);
        if (rc) return rc;
        if (side) {
            // on the side stream: the counting kernels read the caller's buffer, the walk reads the store -- the copy runs
            // beside the kernels (0.25 ms for 10 M reads) and add_reads_dev_locked waits for it before it returns
            HIPCHK(c, hipEventRecord(c->ev_piece[7], c->stream));  // (what sits in the stream before may still write the store: rs_reserve's copy)
            HIPCHK(c, hipStreamWaitEvent(c->pipe_stream, c->ev_piece[7], 0));
            HIPCHK(c, hipMemcpyAsync(c->rs_words + at, d_words + w0, (w1 - w0 + 1) * 8, hipMemcpyDeviceToDevice, c->pipe_stream));
            c->rs_copy_pending = true;
        } else {
            HIPCHK(c, hipMemcpyAsync(c->rs_words + at, d_words + w0, (w1 - w0 + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    c->cur_ptr_base = at * 32 - w0 * 32;  // (mod 2^64: the kernels add a batch position >= w0 * 32)
    c->rs_bases = (at + (w1 - w0)) * 32;
    return MC_OK;
}

// Bases one run of the partitioned pipeline takes.  Positions are 64-bit throughout; what is 32-bit is the index of a
// record inside the scatter levels' arrays (pipe_prepare checks it): super-k-mer records are ~0.18 per window, so 2^34
// bases (114 M reads of 150 bp) stay far below 2^32 records, while one 8-byte key per window (k > 31, hash keys) costs
// scratch by the window -- those runs take what half the free memory holds, 2^31 to 2^33 bases.
// Every run reads and rewrites the whole table, so fewer, larger runs are what a large read set wants.
static uint64_t max_run_bases(const mc_ctx *c, double windows_per_base)
{   // windows_per_base: of the read set at hand (88 / 150 for 150-base reads at k = 63): the scratch is per window
    static const uint64_t env = [] { const char *e = getenv("MC_MAX_RUN_BASES"); return e && *e ? strtoull(e, nullptr, 10) : 0ull; }();
    if (env) return std::max<uint64_t>(env, 1u << 20);
    if (c->mm_k) return 1ull << 34;
    // (tests: the per-window form's limit alone, so that a run cut for long records is too large for it)
    if (const char *e = getenv("MC_MAX_RUN_BASES_PER_WINDOW")) if (*e) return std::max<uint64_t>(strtoull(e, nullptr, 10), 1u << 20);
    // a key and a read pointer per window, in pieces (add_reads_partitioned): ~16.5 bytes of scratch per window.  Two
    // thirds of what the device has free may go there (the table is allocated already), between 2^31 and 2^33 bases: every
    // run reads and rewrites the whole table, so few, large runs (configs[2]: 2 instead of 4).
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = 0;
    fr += (c->pipe.a_cap + c->pipe.b_cap) * 12;  // (the scratch of the run before is ours to reuse)
    const double per_base = 16.5 * std::min(1.0, std::max(0.05, windows_per_base));
    return std::min<uint64_t>(1ull << 33, std::max<uint64_t>((1ull << 31) - (1ull << 24), (uint64_t)((double)fr * 0.66 / per_base)));
}

// counting with read offsets known on the host
static int add_reads_impl(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, const uint64_t *h_off,
                          uint64_t n_reads)
{
    const uint64_t k = (uint64_t)c->cfg.k;
    auto windows_of = [&](uint64_t a, uint64_t b) {
        uint64_t w = 0;
        for (uint64_t i = a; i < b; i++) {
            const uint64_t len = h_off[i + 1] - h_off[i];
            if (len >= k) w += len - k + 1;
        }
        return w;
    };
    const uint64_t total = windows_of(0, n_reads);
    const bool partition = c->count_path == 2 || (c->count_path == 0 && total >= (1ull << 22));
    if (partition) {
        uint64_t max_bases = max_run_bases(c, 1.0);
        uint64_t r = 0;
        while (r < n_reads) {
            uint64_t r1 = (uint64_t)(std::upper_bound(h_off + r, h_off + n_reads + 1, h_off[r] + max_bases) - h_off) - 1;
            if (r1 <= r) return fail(c, MC_EINVAL, "a single read of more than %llu bases is not supported", (unsigned long long)max_bases);
            if (r1 > n_reads) r1 = n_reads;
            const uint64_t wb = windows_of(r, r1);
            if (wb) {
                int rc = add_reads_partitioned_any(c, d_words, d_off, r, r1, h_off[r], h_off[r1], wb);
                if (rc == RC_RESPLIT) { max_bases = max_run_bases(c, 1.0); continue; }  // (nothing of [r, r1) was counted)
                if (rc) return rc;
            }
            r = r1;
        }
        counts_changed(c);
        c->solid_cov = -1; c->solid_external = false;
        return MC_OK;
    }
    uint64_t r = 0;
    while (r < n_reads) {
        uint64_t allowed;
        const uint64_t remaining_bases = h_off[n_reads] - h_off[r];
        int rc = table_reserve(c, remaining_bases, &allowed);
        if (rc) return rc;
        // largest r1 with windows(r..r1) <= allowed (windows <= bases)
        uint64_t r1 = r;
        const uint64_t target = h_off[r] + allowed;
        r1 = (uint64_t)(std::upper_bound(h_off + r, h_off + n_reads + 1, target) - h_off) - 1;
        if (r1 <= r) r1 = r + 1;  // a single read longer than the allowance: fine, still < 0.85 + one read
        if (r1 > n_reads) r1 = n_reads;
        const uint64_t win = windows_of(r, r1);
        double ms = 0;
        rc = timed(c, &ms, [&] { launch_count(c, d_words, d_off, r, r1); });
        if (rc) return rc;
        c->st.count_ms += ms;
        c->st.count_total_ms += ms;
        c->st.count_launches++;
        c->st.windows += win;
        r = r1;
    }
    counts_changed(c);
    c->solid_cov = -1; c->solid_external = false;
    return MC_OK;
}

// ------------------------------------------------------------------------------------------ C ABI

extern "C" {

int mc_abi_version(void) { return MC_ABI_VERSION; }

const char *mc_last_error(const mc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int mc_create(const mc_config *cfg, mc_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, MC_EINVAL, "mc_create: null argument");
    *out = nullptr;
    if (cfg->key_mode < MC_KEY_PACKED || cfg->key_mode > MC_KEY_FNV1A)
        return fail(nullptr, MC_EINVAL, "mc_create: unknown key_mode %d", cfg->key_mode);
    const int kmax = cfg->key_mode == MC_KEY_PACKED ? 31 : 63;
    if (cfg->k < 1 || cfg->k > kmax)
        return fail(nullptr, MC_EINVAL, "mc_create: k=%d out of range 1..%d for key_mode %d", cfg->k, kmax,
                    cfg->key_mode);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, MC_EHIP, "mc_create: no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= n_dev)
        return fail(nullptr, MC_EINVAL, "mc_create: device %d out of range (have %d)", cfg->device, n_dev);
    mc_ctx *c = new (std::nothrow) mc_ctx;
    if (!c) return fail(nullptr, MC_ENOMEM, "mc_create: out of host memory");
    c->cfg = *cfg;
#define CREATE_CHK(call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            fail(nullptr, MC_EHIP, "mc_create: %s: %s", #call, hipGetErrorString(e_));            \
            mc_destroy(c);                                                                        \
            return e_ == hipErrorOutOfMemory ? MC_ENOMEM : MC_EHIP;                               \
        }                                                                                         \
    } while (0)
    CREATE_CHK(hipSetDevice(cfg->device));
    CREATE_CHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CREATE_CHK(hipEventCreate(&c->ev0));
    CREATE_CHK(hipEventCreate(&c->ev1));
    CREATE_CHK(hipStreamCreateWithFlags(&c->pipe_stream, hipStreamNonBlocking));
    for (auto &e : c->ev_piece) CREATE_CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CREATE_CHK(hipEventCreate(&c->ev_p2));
    for (auto &e : c->ev_t) CREATE_CHK(hipEventCreate(&e));
    CREATE_CHK(hipMalloc(reinterpret_cast<void **>(&c->d_ctr), 16 * sizeof(unsigned long long)));
    CREATE_CHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_scratch), 32 * sizeof(unsigned long long), hipHostMallocDefault));
    c->d_fatal = reinterpret_cast<uint32_t *>(c->d_ctr + 8);  // (one copy brings the counters and the flag)
    CREATE_CHK(hipMalloc(reinterpret_cast<void **>(&c->d_ovf), mc_ctx::OVF_CAP * sizeof(uint4)));
    CREATE_CHK(hipMalloc(reinterpret_cast<void **>(&c->d_ovf_leaf), mc_ctx::OVF_CAP * sizeof(uint32_t)));
    CREATE_CHK(hipMemsetAsync(c->d_ctr, 0, 16 * sizeof(unsigned long long), c->stream));

#undef CREATE_CHK
    if (const char *e = getenv("MC_COUNT_PATH")) c->count_path = !strcmp(e, "direct") ? 1 : !strcmp(e, "partition") ? 2 : 0;
    if (const char *e = getenv("MC_BFS_DIRECT")) c->bfs_direct = strcmp(e, "0") != 0;
    c->want_list = (cfg->flags & MC_FLAG_SOLID_LIST) != 0 || !c->bfs_direct;
    // packed keys of at least SK_MIN_K bases: table regions = minimizer bins, reads counted as super-k-mers
    // (MC_SUPERKMERS=0 keeps the per-window pipeline: for A/B measurements)
    if (cfg->key_mode == MC_KEY_PACKED && cfg->k >= SK_MIN_K) c->mm_k = cfg->k;
    if (const char *e = getenv("MC_SUPERKMERS")) if (!strcmp(e, "0")) c->mm_k = 0;
    c->sk_form = c->mm_k != 0;
    // polynomial keys of 33 .. 63 bases in a table sized by a capacity hint: minimizer bins too, reads counted as long records
    // (count_long.h; hash_bins() above says what such a table cannot do and what happens then).  MC_LONG_RECORDS=0: the per-window pipeline.
    if (cfg->key_mode == MC_KEY_POLY && cfg->k >= SKL_MIN_K && cfg->k <= SKL_MAX_K) {
        const char *e = getenv("MC_LONG_RECORDS");
        if (!(e && !strcmp(e, "0"))) c->mm_k = cfg->k;
    }
    uint64_t want_slots = 1ull << 22;  // 4 M slots = 64 MB to start with
    if (cfg->capacity_hint) {
        // Load factor the hint is turned into.  Hash-prefix tables: 0.7 (regions are probed in LDS, a fuller table
        // costs little).  Minimizer-bin tables fill less evenly (the k-mers of one locus share a bin) and both the merge
        // kernel's probe loops and the walk's lookups pay for every extra probe, while a sparser table only costs its
        // write-back: measured on the 10 M-read workload (364 M keys), count + BFS per step at load 0.60 / 0.51 / 0.43 /
        // 0.36: 39.1 / 31.5 / 29.2 / 28.6 ms.  So 0.25 up to 64 M keys, + 0.05 per doubling, 0.36 at most -- but never
        // more than 2^20 regions while the load stays under 0.6: beyond that a leaf of the counting pipeline covers two
        // regions and the merge kernel sweeps each leaf twice.
        double load = 0.7;
        if (c->mm_k) load = std::min(0.36, std::max(0.25, 0.25 + 0.05 * std::log2((double)cfg->capacity_hint / (double)(64u << 20))));
        if (const char *e = getenv("MC_TABLE_LOAD")) { const double v = atof(e); if (v > 0.05 && v < 0.95) load = v; }  // (tuning runs)
        want_slots = std::max<uint64_t>(want_slots, (uint64_t)((double)cfg->capacity_hint / load));
        if (c->mm_k) want_slots = std::max<uint64_t>(1ull << 22, mm_slots_for(c, (double)cfg->capacity_hint, load));
        // Long records need their bins roomy: at k = 63 a region holds the k-mers of four or five loci (each with the error variants
        // of the ~30 reads that cover it), and 2^21 regions -- one a leaf, all the merge kernel takes -- at load 0.53 (configs[2] at
        // full size: 4.6 G keys, 137 GB) leave a few per cent of them overfull, more than the hand-on list holds; such a table cannot
        // be rebuilt either.  A table that cannot be roomy takes its bin words from the TWO smallest hashes of a window (count_long.h
        // skl_word2: mm_k < 0 -- half again as many, smaller loci a region: even enough up to load ~0.55, scripts/bin_model.py);
        // beyond that, and with MC_LONG_BINS=1 (tuning runs; 2: two smallest for every table), the per-window pipeline.
        const bool crowded = hash_bins(c) && (double)cfg->capacity_hint > 0.40 * (double)want_slots;
        const char *be = getenv("MC_LONG_BINS");
        const bool two = hash_bins(c) && (be ? !strcmp(be, "2") : crowded);
        if (two && (double)cfg->capacity_hint <= 0.56 * (double)want_slots) c->mm_k = -cfg->k;
        else if (crowded) {
            c->mm_k = 0;
            want_slots = std::max<uint64_t>(1ull << 22, (uint64_t)((double)cfg->capacity_hint / 0.7));
            if (const char *e = getenv("MC_TABLE_LOAD")) { const double v = atof(e); if (v > 0.05 && v < 0.95) want_slots = std::max<uint64_t>(1ull << 22, (uint64_t)((double)cfg->capacity_hint / v)); }
        }
    }
    int rc = table_alloc(c, regions_for(c, want_slots));
    if (rc) {
        g_create_err = c->err;
        mc_destroy(c);
        return rc;
    }
    *out = c;
    return MC_OK;
}

void mc_destroy(mc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    // Blocks go back to process-wide pools below and may be handed to another context at once (hipFree used to wait for the
    // device; a pool does not): everything this context still has queued must be done first -- on the caller's stream too
    // (mc_set_stream) and on the side stream of the pipeline.
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream && c->own_stream != c->stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->pipe_stream) (void)hipStreamSynchronize(c->pipe_stream);
    for (auto &m : c->ipc_opened) (void)hipIpcCloseMemHandle(m.p);
    if (c->d_shards) (void)hipFree(c->d_shards);
    if (c->slots) table_release(c, c->slots, c->slots_bytes);
    if (c->solid) (void)hipFree(c->solid);
    c->pipe.release(c->cfg.device);
    {
        mc_ctx::Dup &D = c->dup;
        g_scratch_pool.put(c->cfg.device, D.l1_keys, D.l1_words * 8);
        g_scratch_pool.put(c->cfg.device, D.l1_counts, D.l1_counts_cap * 4);
        g_scratch_pool.put(c->cfg.device, D.l2_own, D.l2_own_words * 8);
        g_scratch_pool.put(c->cfg.device, D.l2_counts, D.l2_counts_cap * 4);
        if (D.flags) (void)hipFree(D.flags);
        if (D.ctr) (void)hipFree(D.ctr);
        if (D.list) (void)hipFree(D.list);
        if (D.set.qk) (void)hipFree(D.set.qk);
        if (D.tw) (void)hipFree(D.tw);
        if (D.pq_mem) (void)hipFree(D.pq_mem);
        if (D.pq_groups) (void)hipFree(D.pq_groups);
    }
    c->bfs_pool.clear();
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    if (c->d_ovf) (void)hipFree(c->d_ovf);
    if (c->d_ovf_leaf) (void)hipFree(c->d_ovf_leaf);
    g_scratch_pool.put(c->cfg.device, c->d_ovf_tmp, c->ovf_tmp_cap * sizeof(uint4));
    if (c->d_ovf_leaf_tmp) (void)hipFree(c->d_ovf_leaf_tmp);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    g_scratch_pool.put(c->cfg.device, c->rs_words, c->rs_cap_words * 8);
    for (auto &e : c->ev_piece) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_seq) if (e) (void)hipEventDestroy(e);
    if (c->ev_p2) (void)hipEventDestroy(c->ev_p2);
    for (auto &e : c->ev_t) if (e) (void)hipEventDestroy(e);
    if (c->pipe_stream) { (void)hipStreamSynchronize(c->pipe_stream); (void)hipStreamDestroy(c->pipe_stream); }
    c->tok_pool.release();
    if (c->h_bfs_stage) (void)hipHostFree(c->h_bfs_stage);
    if (c->d_bfs_stage) (void)hipFree(c->d_bfs_stage);
    if (c->d_bfs_pack) (void)hipFree(c->d_bfs_pack);
    if (c->h_bfs_hdr) (void)hipHostFree(c->h_bfs_hdr);
    for (char *p : c->pin) if (p) (void)hipHostFree(p);
    for (hipStream_t st : c->pin_stream) if (st) (void)hipStreamDestroy(st);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int mc_clear(mc_ctx *c)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->virgin = true;
    if (c->d_shards) {  // (the tables an attachment describes are about to change; their mappings stay for the next attachment)
        for (auto &m : c->ipc_opened) m.in_use = false;
        (void)hipFree(c->d_shards);
        c->d_shards = nullptr; c->h_shards.clear(); c->shard_owner_mm_k = 0;
    }
    c->rs_bases = 0;  // (slots that pointed into the read store go with the table)
    c->rs_hi_bases = 0;
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, 9 * sizeof(unsigned long long), c->stream));  // (counters and the fatal flag: one fill)
    c->n_used_host = 0;
    c->finalized = false;
    c->shards_dropped = false;
    c->solid_cov = -1; c->solid_external = false;
    c->solid_tracked = true;
    c->solid_list_fresh = false;
    dup_forget(c);
    return MC_OK;
}

int mc_set_coverage_hint(mc_ctx *c, int min_cov)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (min_cov < 0 || min_cov > 32767) return fail(c, MC_EINVAL, "mc_set_coverage_hint: min_cov must be in 0..32767");
    if (min_cov != c->cov_hint && !c->virgin) c->solid_tracked = false;  // keys already counted were not tracked at this threshold
    if (min_cov != c->cov_hint) c->solid_list_fresh = false;
    c->cov_hint = min_cov;
    return MC_OK;
}

int mc_set_read_pointers(mc_ctx *c, int mode)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    const int m = mode & 0xF;
    if (m > 2 || (mode & ~(0xF | MC_PTRS_ON_EVERY_RECORD))) return fail(c, MC_EINVAL, "mc_set_read_pointers: mode %d", mode);
    c->rs_enabled = m == 1;
    c->rs_virtual = m == 2;
    c->all_ptrs = (mode & MC_PTRS_ON_EVERY_RECORD) != 0;
    return MC_OK;
}

int mc_read_store_seek(mc_ctx *c, uint64_t at_bases, uint64_t reserve_bases)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (at_bases % 32) return fail(c, MC_EINVAL, "mc_read_store_seek: positions are whole words (multiples of 32 bases)");
    if (!c->rs_enabled && !c->rs_virtual) return fail(c, MC_ESTATE, "mc_read_store_seek: this context keeps no read pointers");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->rs_enabled) {
        c->rs_hi_bases = c->rs_end();  // (what lies below stays readable)
        c->rs_bases = at_bases;
        const uint64_t want = std::max(reserve_bases, at_bases) / 32 + 2;
        if (want > c->rs_cap_words) {
            int rc = rs_reserve(c, want - at_bases / 32);
            if (rc) return rc;
        }
    } else {
        c->rs_bases = at_bases;
    }
    return MC_OK;
}

uint64_t mc_read_store_tell(mc_ctx *c)
{
    if (!c) return 0;
    std::lock_guard<std::mutex> g(c->mu);
    return c->rs_bases;
}

int mc_read_store_import_dev(mc_ctx *c, const uint64_t *d_words, uint64_t n_words, uint64_t at_bases)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->rs_enabled) return fail(c, MC_ESTATE, "mc_read_store_import_dev: this context keeps no read store");
    if (at_bases % 32 || (!d_words && n_words)) return fail(c, MC_EINVAL, "mc_read_store_import_dev: bad argument");
    if (n_words == 0) return MC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t at = at_bases / 32;
    if (at + n_words + 1 > c->rs_cap_words) {  // (mc_read_store_seek's reserve_bases is what makes this rare: a move copies the store)
        const uint64_t fill = c->rs_bases / 32;
        int rc = rs_reserve(c, at + n_words + 1 > fill ? at + n_words + 1 - fill : 0);
        if (rc) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(c->rs_words + at, d_words, n_words * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the caller's buffer may go as soon as this returns)
    c->rs_hi_bases = std::max(c->rs_hi_bases, (at + n_words) * 32);
    return MC_OK;
}

int mc_share_read_store(mc_ctx *c, mc_ctx *from)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (from == c) from = nullptr;
    if (from && from->cfg.device != c->cfg.device) return fail(c, MC_EINVAL, "mc_share_read_store: the contexts are on different devices");
    c->rs_from = from;
    return MC_OK;
}

int mc_set_stream(mc_ctx *c, void *hip_stream)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return MC_OK;
}

static int add_reads_dev_locked(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads, uint64_t n_bases,
                                int64_t in_store = -1);

// Host (pageable) memory to the device.  A plain hipMemcpy of pageable memory stages through one thread; a few
// threads each staging 8 MB pieces through their own pinned buffers and stream reach the link rate
// (scripts/microbench/hostreg.hip: 360 MB in 7-14 ms instead of 19 ms, or 176 ms for memory touched first here).
// fd >= 0: the bytes come from that file at file_off instead (pread straight into the pinned buffers: no page faults on
// a mapping).  Touches no state of the context but the pinned buffers (under their own lock), so the file reader can
// run it beside the kernels of the chunk before; returns a hipError_t-free verdict for the caller to report.
static bool h2d_pinned(mc_ctx *c, void *dst, const void *src, size_t bytes, int fd, uint64_t file_off)
{
    constexpr size_t CHUNK = 8u << 20;
    static constexpr int TMAX = 8;
    static const int T = [] { const char *e = getenv("MC_H2D_THREADS"); const int v = e && *e ? atoi(e) : 8; return std::min(std::max(v, 1), TMAX); }();
    std::lock_guard<std::mutex> g(c->pin_mu);
    if (hipSetDevice(c->cfg.device) != hipSuccess) return false;
    if (!c->pin[0]) {
        for (int i = 0; i < 2 * T; i++)
            if (hipHostMalloc(reinterpret_cast<void **>(&c->pin[i]), CHUNK) != hipSuccess) return false;
        for (int i = 0; i < T; i++)
            if (hipStreamCreateWithFlags(&c->pin_stream[i], hipStreamNonBlocking) != hipSuccess) return false;
    }
    const size_t n_chunks = (bytes + CHUNK - 1) / CHUNK;
    bool failed[TMAX] = {};
    const int device = c->cfg.device;
    std::vector<std::thread> th;
    for (int t = 0; t < T && (size_t)t < n_chunks; t++)
        th.emplace_back([&, t] {
            if (hipSetDevice(device) != hipSuccess) { failed[t] = true; return; }
            hipEvent_t ev[2];
            if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) { failed[t] = true; return; }
            bool used[2] = {false, false};
            int flip = 0;
            for (size_t ch = (size_t)t; ch < n_chunks; ch += T) {
                const size_t off = ch * CHUNK, len = std::min(CHUNK, bytes - off);
                if (used[flip] && hipEventSynchronize(ev[flip]) != hipSuccess) failed[t] = true;
                if (fd >= 0) {
                    for (size_t got = 0; got < len;) {
                        const ssize_t r = pread(fd, c->pin[2 * t + flip] + got, len - got, (off_t)(file_off + off + got));
                        if (r <= 0) { failed[t] = true; break; }
                        got += (size_t)r;
                    }
                } else {
                    memcpy(c->pin[2 * t + flip], static_cast<const char *>(src) + off, len);
                }
                if (hipMemcpyAsync(static_cast<char *>(dst) + off, c->pin[2 * t + flip], len, hipMemcpyHostToDevice, c->pin_stream[t]) != hipSuccess ||
                    hipEventRecord(ev[flip], c->pin_stream[t]) != hipSuccess)
                    failed[t] = true;
                used[flip] = true;
                flip ^= 1;
            }
            if (hipStreamSynchronize(c->pin_stream[t]) != hipSuccess) failed[t] = true;
            (void)hipEventDestroy(ev[0]);
            (void)hipEventDestroy(ev[1]);
        });
    for (auto &x : th) x.join();
    for (int t = 0; t < TMAX; t++)
        if (failed[t]) return false;
    return true;
}

static int h2d_fast(mc_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes < (32u << 20)) {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return MC_OK;
    }
    if (!h2d_pinned(c, dst, src, bytes, -1, 0)) return fail(c, MC_EHIP, "host-to-device copy failed");
    return MC_OK;
}

int mc_add_reads_packed(mc_ctx *c, const uint64_t *words, const uint64_t *off, uint64_t n_reads)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if ((!words || !off) && n_reads) return fail(c, MC_EINVAL, "mc_add_reads_packed: null pointer");
    if (n_reads == 0) return MC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t n_bases = off[n_reads];
    if (n_bases < off[0]) return fail(c, MC_EINVAL, "mc_add_reads_packed: read_offsets not monotone");
    const uint64_t n_words = (n_bases + 31) / 32 + 1;
    const uint64_t w_begin = off[0] / 32;  // offsets need not start at 0
    DevBuf<uint64_t> dw, doff;
    uint64_t *dst = nullptr;
    int64_t in_store = -1;
    int rc;
    if (c->rs_enabled) {  // straight into the read store: the kernels read the batch from there
        rc = rs_reserve(c, n_words - w_begin);
        if (rc) return rc;
        in_store = (int64_t)(c->rs_bases / 32);
        dst = c->rs_words + in_store;
    } else {
        HIPCHK(c, dw.alloc(n_words - w_begin));
        dst = dw.p;
    }
    HIPCHK(c, doff.alloc(n_reads + 1));
    rc = h2d_fast(c, dst, words + w_begin, (n_words - w_begin) * 8);
    if (rc) return rc;
    if (w_begin == 0) {
        rc = h2d_fast(c, doff.p, off, (n_reads + 1) * 8);
    } else {
        std::vector<uint64_t> rel(off, off + n_reads + 1);
        for (auto &x : rel) x -= w_begin * 32;
        rc = h2d_fast(c, doff.p, rel.data(), (n_reads + 1) * 8);
    }
    if (rc) return rc;
    // (monotone offsets are checked on the device, with the window count: k_reads_summary)
    rc = add_reads_dev_locked(c, dst, doff.p, n_reads, n_bases - w_begin * 32, in_store);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return rc;
}

// reads resident in HBM (the context's lock is held)
static int add_reads_dev_counted(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads, uint64_t n_bases, int64_t in_store);
static int add_reads_dev_locked(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads, uint64_t n_bases,
                                int64_t in_store)
{
    const int rc = add_reads_dev_counted(c, d_words, d_off, n_reads, n_bases, in_store);
    if (c->rs_copy_pending) {  // the caller's buffer is his again when this returns, and later work on the stream may read the store
        c->rs_copy_pending = false;
        HIPCHK(c, hipStreamSynchronize(c->pipe_stream));
    }
    return rc;
}
static int add_reads_dev_counted(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads, uint64_t n_bases, int64_t in_store)
{
    // summary on the device: total windows, monotone offsets, first and last offset
    unsigned long long *sum = c->d_ctr + 4;
    unsigned long long *sum4 = c->d_ctr + 10;  // windows, violations, first offset, last offset: one copy
    HIPCHK(c, hipMemsetAsync(sum4, 0, 2 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_reads_summary, dim3(grid_for((n_reads + 1) / 2, 256, 512)), dim3(256), 0, c->stream, d_off, n_reads,
                       (uint64_t)c->cfg.k, sum4, 1);
    HIPCHK(c, hipGetLastError());
    unsigned long long *hs = c->h_scratch + 20;
    HIPCHK(c, hipMemcpyAsync(hs, sum4, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t first_off = hs[2], last_off = hs[3];
    if (hs[1]) return fail(c, MC_EINVAL, "mc_add_reads_packed_dev: read_offsets not monotone");
    if (last_off != n_bases)
        return fail(c, MC_EINVAL, "mc_add_reads_packed_dev: read_offsets[n_reads]=%llu but n_bases=%llu",
                    (unsigned long long)last_off, (unsigned long long)n_bases);
    const uint64_t total = hs[0];
    {   // the batch joins the read store (the slots of its k-mers will point there)
        int rc = rs_append(c, d_words, first_off, last_off, in_store, true);
        if (rc) return rc;
        c->ptr_tries = 1;
    }
    const bool partition = c->count_path == 2 || (c->count_path == 0 && total >= (1ull << 22));
    const double wpb = last_off > first_off ? (double)total / (double)(last_off - first_off) : 1.0;  // windows per base
    if (partition && last_off - first_off < max_run_bases(c, wpb)) {  // one batch: no need for the offsets on the host
        int rc = total ? add_reads_partitioned_any(c, d_words, d_off, 0, n_reads, first_off, last_off, total) : MC_OK;
        if (rc != RC_RESPLIT) {  // (RC_RESPLIT: the long form declined and the batch is too large for one record a window: in runs, below)
            if (rc) return rc;
            counts_changed(c);
            c->solid_cov = -1; c->solid_external = false;
            return MC_OK;
        }
    }
    if (partition) {
        // several runs: cut at read boundaries found by bisection on the device's offsets (a few 8-byte copies), each
        // run's windows summed on the device -- the offsets of 100 M reads never travel to the host
        auto off_at = [&](uint64_t i, uint64_t *v) -> int {
            HIPCHK(c, hipMemcpyAsync(v, d_off + i, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            return MC_OK;
        };
        uint64_t r = 0, off_r = first_off;
        while (r < n_reads) {
            const uint64_t max_bases = max_run_bases(c, wpb), target = off_r + max_bases;
            uint64_t r1 = n_reads, off_r1 = last_off;
            if (last_off > target) {
                uint64_t lo = r, hi = n_reads;  // offsets[lo] <= target < offsets[hi]
                while (hi - lo > 1) {
                    const uint64_t mid = lo + (hi - lo) / 2;
                    uint64_t v;
                    int rc = off_at(mid, &v);
                    if (rc) return rc;
                    if (v <= target) lo = mid; else hi = mid;
                }
                r1 = lo;
                int rc = off_at(r1, &off_r1);
                if (rc) return rc;
            }
            if (r1 <= r) return fail(c, MC_EINVAL, "a single read of more than %llu bases is not supported", (unsigned long long)max_bases);
            HIPCHK(c, hipMemsetAsync(sum, 0, 2 * sizeof(unsigned long long), c->stream));
            hipLaunchKernelGGL(k_reads_summary, dim3(grid_for(r1 - r, 256, 512)), dim3(256), 0, c->stream, d_off + r, r1 - r, (uint64_t)c->cfg.k, sum);
            HIPCHK(c, hipGetLastError());
            unsigned long long wb = 0;
            HIPCHK(c, hipMemcpyAsync(&wb, sum, sizeof wb, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (wb) {
                int rc = add_reads_partitioned_any(c, d_words, d_off, r, r1, off_r, off_r1, wb);
                if (rc == RC_RESPLIT) continue;  // (nothing of [r, r1) was counted; the next cut is made for the table as it is now)
                if (rc) return rc;
            }
            r = r1;
            off_r = off_r1;
        }
        counts_changed(c);
        c->solid_cov = -1; c->solid_external = false;
        return MC_OK;
    }
    // the launch planner of the direct kernel needs the offsets on the host (8 bytes per read, once per call)
    std::vector<uint64_t> h_off(n_reads + 1);
    HIPCHK(c, hipMemcpyAsync(h_off.data(), d_off, (n_reads + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return add_reads_impl(c, d_words, d_off, h_off.data(), n_reads);
}

int mc_add_reads_packed_dev(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads,
                            uint64_t n_bases)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if ((!d_words || !d_off) && n_reads) return fail(c, MC_EINVAL, "mc_add_reads_packed_dev: null pointer");
    if (n_reads == 0) return MC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return add_reads_dev_locked(c, d_words, d_off, n_reads, n_bases);
}

int mc_add_keys_dev(mc_ctx *c, const int64_t *d_keys, const uint32_t *d_hints, uint64_t n)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!d_keys && n) return fail(c, MC_EINVAL, "mc_add_keys_dev: null pointer");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const bool partition = c->count_path == 2 || (c->count_path == 0 && n >= (1ull << 22));
    if (partition) {
        const uint64_t max_batch = (1ull << 31) - (1ull << 24);
        for (uint64_t i = 0; i < n; i += max_batch) {
            const uint64_t m = std::min<uint64_t>(max_batch, n - i);
            int rc = add_keys_partitioned(c, reinterpret_cast<const uint64_t *>(d_keys) + i, d_hints ? d_hints + i : nullptr, m);
            if (rc) return rc;
        }
    } else {
        c->solid_tracked = false;
        c->solid_list_fresh = false;
        uint64_t i = 0;
        while (i < n) {
            uint64_t allowed;
            int rc = table_reserve(c, n - i, &allowed);
            if (rc) return rc;
            const uint64_t m = std::min<uint64_t>(allowed, n - i);
            double ms = 0;
            rc = timed(c, &ms, [&] {
                if (d_hints)
                    hipLaunchKernelGGL(k_add_keys_hint, dim3(grid_for(m, 256)), dim3(256), 0, c->stream,
                                       reinterpret_cast<const uint64_t *>(d_keys) + i, d_hints + i, m, c->view());
                else
                    hipLaunchKernelGGL(k_add_keys, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, d_keys + i, m, c->view());
            });
            if (rc) return rc;
            c->st.count_ms += ms;
            c->st.count_total_ms += ms;
            c->st.count_launches++;
            c->st.windows += m;
            i += m;
        }
    }
    counts_changed(c);
    c->solid_cov = -1; c->solid_external = false;
    return MC_OK;
}

int mc_add_pairs_dev(mc_ctx *c, const int64_t *d_keys, const int16_t *d_counts, const uint32_t *d_hints, uint64_t n)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if ((!d_keys || !d_counts) && n) return fail(c, MC_EINVAL, "mc_add_pairs_dev: null pointer");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->solid_tracked = false;
    c->solid_list_fresh = false;
    uint64_t i = 0;
    while (i < n) {
        uint64_t allowed;
        int rc = table_reserve(c, n - i, &allowed);
        if (rc) return rc;
        const uint64_t m = std::min<uint64_t>(allowed, n - i);
        hipLaunchKernelGGL(k_add_pairs, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, d_keys + i, d_counts + i,
                           d_hints ? d_hints + i : nullptr, m, c->view());
        HIPCHK(c, hipGetLastError());
        i += m;
    }
    counts_changed(c);
    c->solid_cov = -1; c->solid_external = false;
    return MC_OK;
}

// ---- f1 on the device: csrc/tokenizer.h driven over one chunk of an uncompressed FASTA / FASTQ file

// exclusive scan of n 32-bit counts into 64-bit offsets; *total on the host
static int tok_scan(mc_ctx *c, const uint32_t *d_in, uint64_t n, unsigned long long *d_out, uint64_t *total)
{
    const uint64_t m = std::max<uint64_t>((n + tok::SCAN_TILE - 1) / tok::SCAN_TILE, 1);
    PoolBuf<unsigned long long> sums;
    HIPCHK(c, sums.alloc(&c->tok_pool, m + 1));
    hipLaunchKernelGGL(tok::k_scan_sums, dim3((unsigned)m), dim3(tok::T_THREADS), 0, c->stream, d_in, n, sums.p);
    hipLaunchKernelGGL(tok::k_scan_one, dim3(1), dim3(1024), 0, c->stream, sums.p, m, sums.p + m);
    hipLaunchKernelGGL(tok::k_scan_apply, dim3((unsigned)m), dim3(tok::T_THREADS), 0, c->stream, d_in, n, sums.p, d_out);
    HIPCHK(c, hipGetLastError());
    unsigned long long t = 0;
    HIPCHK(c, hipMemcpyAsync(&t, sums.p + m, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *total = t;
    return MC_OK;
}

// Text bytes [b, e) of a mapped file (a whole number of records) -> packed reads in HBM -> counted.  *declined: the
// device saw something the host parser has to deal with; nothing was added.  The context's lock is held.
// The chunks of one file, tokenised into the read store back to back and counted together when the file ends: one run
// of the counting pipeline (which reads and rewrites the whole table) instead of one per 256 MB of text.
struct TokPending {
    PoolBuf<uint64_t> off;     // read offsets of all chunks so far (+ the end), relative to the first chunk's first base
    uint64_t off_cap = 0;
    uint64_t reads = 0, bases = 0;
    int64_t base_word = -1;    // where the first chunk starts in the read store
};
static int tok_flush_locked(mc_ctx *c, TokPending &P)
{
    if (P.reads == 0) { P.base_word = -1; P.bases = 0; return MC_OK; }
    if ((uint64_t)P.base_word != c->rs_bases / 32)  // (the chunks sit behind the store's end until they are counted)
        return fail(c, MC_ESTATE, "mc_add_reads_file: the context took other reads while a file was being read");
    int rc = add_reads_dev_locked(c, c->rs_words + P.base_word, P.off.p, P.reads, P.bases, P.base_word);
    if (!rc) HIPCHK(c, hipStreamSynchronize(c->stream));
    P.reads = P.bases = 0;
    P.base_word = -1;
    return rc;
}

static int tokenize_chunk_locked(mc_ctx *c, const mch::PlainReadsFile &f, const char *b, const char *e, uint8_t *d_text, uint64_t *n_reads_out,
                                 bool *declined, TokPending *pend = nullptr)
{
    *declined = false;
    *n_reads_out = 0;
    const uint64_t n = (uint64_t)(e - b);
    if (n == 0) return MC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const bool dbg = getenv("MC_INGEST_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t1 = now();
    struct { uint8_t *p; } text{d_text};  // (padded with zero bytes to whole tiles of the newline passes: tok_text_bytes)
    {
        const uint64_t n_padded = (n + tok::T_TILE - 1) / tok::T_TILE * tok::T_TILE;
        if (n_padded > n) HIPCHK(c, hipMemsetAsync(text.p + n, 0, n_padded - n, c->stream));
    }
    int rc = MC_OK;
    PoolBuf<uint32_t> flags;
    HIPCHK(c, flags.alloc(&c->tok_pool, 1));
    HIPCHK(c, hipMemsetAsync(flags.p, 0, 4, c->stream));
    // pass 1: newline positions
    const uint64_t n_tiles = (n + tok::T_TILE - 1) / tok::T_TILE;
    if (n_tiles > 0x7FFFFFFFull) { *declined = true; return MC_OK; }
    PoolBuf<uint32_t> tile_counts;
    PoolBuf<unsigned long long> tile_off, nl;
    HIPCHK(c, tile_counts.alloc(&c->tok_pool, n_tiles));
    HIPCHK(c, tile_off.alloc(&c->tok_pool, n_tiles));
    hipLaunchKernelGGL(tok::k_nl_count, dim3((unsigned)n_tiles), dim3(tok::T_THREADS), 0, c->stream, text.p, tile_counts.p);
    uint64_t n_nl = 0;
    rc = tok_scan(c, tile_counts.p, n_tiles, tile_off.p, &n_nl);
    if (rc) return rc;
    HIPCHK(c, nl.alloc(&c->tok_pool, n_nl));
    hipLaunchKernelGGL(tok::k_nl_write, dim3((unsigned)n_tiles), dim3(tok::T_THREADS), 0, c->stream, text.p, tile_off.p, nl.p);
    HIPCHK(c, hipGetLastError());
    const uint64_t n_lines = n_nl + (e[-1] != '\n' ? 1 : 0);
    if (n_lines >= 0xFFFFFFF0ull) { *declined = true; return MC_OK; }

    auto read_flags = [&](uint32_t *out) -> int {
        HIPCHK(c, hipMemcpyAsync(out, flags.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return MC_OK;
    };
    // where the packed words go: straight into the read store when the context keeps one
    PoolBuf<uint64_t> own_words;
    PoolBuf<uint64_t> offsets;
    uint64_t *dst = nullptr;
    int64_t in_store = -1;
    uint64_t dst_base = 0;  // this chunk's first base in `dst` (deferred counting: behind the chunks before it)
    uint64_t *off_out = nullptr;
    const bool defer = pend != nullptr && c->rs_enabled;
    auto reserve_words = [&](uint64_t total_bases, uint64_t n_reads_chunk) -> int {
        const uint64_t n_words = (total_bases + 31) / 32 + 1;
        if (defer) {
            if (pend->base_word < 0) { pend->base_word = (int64_t)(c->rs_bases / 32); pend->bases = 0; pend->reads = 0; }
            if ((uint64_t)pend->base_word != c->rs_bases / 32)
                return fail(c, MC_ESTATE, "mc_add_reads_file: the context took other reads while a file was being read");
            dst_base = pend->bases;
            // (the caller reserved the store for the whole file: no reallocation may move the chunks packed so far)
            if ((uint64_t)pend->base_word + (dst_base + total_bases + 31) / 32 + 2 > c->rs_cap_words)
                return fail(c, MC_EINVAL, "internal: the read store was not reserved for the whole file");
            dst = c->rs_words + pend->base_word;
            in_store = pend->base_word;
            const uint64_t first_new = (dst_base + 31) / 32;  // words before it hold bases of earlier chunks
            HIPCHK(c, hipMemsetAsync(dst + first_new, 0, ((dst_base + total_bases + 31) / 32 + 1 - first_new) * 8, c->stream));
            const uint64_t need = pend->reads + n_reads_chunk + 1;
            if (need > pend->off_cap) {
                PoolBuf<uint64_t> bigger;
                const uint64_t cap = std::max<uint64_t>(need * 2, 1u << 20);
                HIPCHK(c, bigger.alloc(&c->tok_pool, cap));
                if (pend->reads) HIPCHK(c, hipMemcpyAsync(bigger.p, pend->off.p, (pend->reads + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                std::swap(pend->off.p, bigger.p);
                std::swap(pend->off.bytes, bigger.bytes);
                std::swap(pend->off.pool, bigger.pool);
                pend->off_cap = cap;
            }
            off_out = pend->off.p + pend->reads;
            return MC_OK;
        }
        if (c->rs_enabled) {
            int r = rs_reserve(c, n_words);
            if (r) return r;
            in_store = (int64_t)(c->rs_bases / 32);
            dst = c->rs_words + in_store;
        } else {
            HIPCHK(c, own_words.alloc(&c->tok_pool, n_words));
            dst = own_words.p;
        }
        HIPCHK(c, hipMemsetAsync(dst, 0, n_words * 8, c->stream));
        return MC_OK;
    };
    uint64_t n_reads = 0, total_bases = 0;
    uint32_t fl = 0;
    if (!f.fastq) {
        PoolBuf<uint32_t> line_hdr, line_len, keep_len, rec_first, rec_keep;
        PoolBuf<uint8_t> line_n, rec_n;
        PoolBuf<unsigned long long> hdr_before, rec_len, out_off, rec_out;
        HIPCHK(c, line_hdr.alloc(&c->tok_pool, n_lines));
        HIPCHK(c, line_len.alloc(&c->tok_pool, n_lines));
        HIPCHK(c, line_n.alloc(&c->tok_pool, n_lines));
        HIPCHK(c, hdr_before.alloc(&c->tok_pool, n_lines));
        hipLaunchKernelGGL(tok::k_fa_lines, dim3(grid_for(n_lines, 4, 1 << 14)), dim3(tok::T_THREADS), 0, c->stream, text.p, n, nl.p, n_nl, n_lines, line_hdr.p,
                           line_len.p, line_n.p, flags.p);
        uint64_t n_hdr = 0;
        rc = tok_scan(c, line_hdr.p, n_lines, hdr_before.p, &n_hdr);
        if (rc) return rc;
        rc = read_flags(&fl);
        if (rc) return rc;
        if (fl) { *declined = true; return MC_OK; }
        const uint64_t n_rec = n_hdr + 1;  // (record 0: the lines in front of the first header)
        HIPCHK(c, rec_n.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, rec_len.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, rec_first.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, rec_keep.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, rec_out.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, keep_len.alloc(&c->tok_pool, n_lines));
        HIPCHK(c, out_off.alloc(&c->tok_pool, n_lines));
        HIPCHK(c, hipMemsetAsync(rec_n.p, 0, n_rec, c->stream));
        HIPCHK(c, hipMemsetAsync(rec_len.p, 0, n_rec * 8, c->stream));
        HIPCHK(c, hipMemsetAsync(rec_first.p, 0xFF, n_rec * 4, c->stream));
        const int g = grid_for(n_lines, 256, 1 << 16);
        hipLaunchKernelGGL(tok::k_fa_records, dim3(g), dim3(256), 0, c->stream, hdr_before.p, line_hdr.p, line_len.p, line_n.p, n_lines, rec_n.p, rec_len.p,
                           rec_first.p);
        hipLaunchKernelGGL(tok::k_fa_keep, dim3(g), dim3(256), 0, c->stream, hdr_before.p, line_hdr.p, line_len.p, n_lines, rec_n.p, keep_len.p);
        hipLaunchKernelGGL(tok::k_fa_rec_keep, dim3(grid_for(n_rec, 256, 1 << 16)), dim3(256), 0, c->stream, rec_n.p, rec_len.p, n_rec, rec_keep.p);
        rc = tok_scan(c, keep_len.p, n_lines, out_off.p, &total_bases);
        if (rc) return rc;
        rc = tok_scan(c, rec_keep.p, n_rec, rec_out.p, &n_reads);
        if (rc) return rc;
        if (n_reads) {
            rc = reserve_words(total_bases, n_reads);
            if (rc) return rc;
            if (!off_out) { HIPCHK(c, offsets.alloc(&c->tok_pool, n_reads + 1)); off_out = offsets.p; }
            hipLaunchKernelGGL(tok::k_fa_offsets, dim3(grid_for(n_rec, 256, 1 << 16)), dim3(256), 0, c->stream, rec_keep.p, rec_out.p, rec_first.p, out_off.p,
                               n_rec, n_reads, total_bases, off_out, dst_base);
            hipLaunchKernelGGL(tok::k_fa_pack, dim3(grid_for(n_lines, 4, 1 << 14)), dim3(tok::T_THREADS), 0, c->stream, text.p, n, nl.p, n_nl,
                               n_lines, keep_len.p, out_off.p, dst, flags.p, dst_base);
            HIPCHK(c, hipGetLastError());
        }
    } else {
        if (n_lines % 4) { *declined = true; return MC_OK; }
        const uint64_t n_rec = n_lines / 4;
        PoolBuf<uint32_t> rec_pieces, rec_bases;
        PoolBuf<unsigned long long> piece_at, base_at;
        HIPCHK(c, rec_pieces.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, rec_bases.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, piece_at.alloc(&c->tok_pool, n_rec));
        HIPCHK(c, base_at.alloc(&c->tok_pool, n_rec));
        const int g = grid_for(n_rec, 4, 1 << 14);  // a wave per record
        hipLaunchKernelGGL(tok::k_fq_records, dim3(g), dim3(tok::T_THREADS), 0, c->stream, text.p, n, nl.p, n_nl, n_rec, f.offset, rec_pieces.p, rec_bases.p,
                           flags.p);
        rc = tok_scan(c, rec_pieces.p, n_rec, piece_at.p, &n_reads);
        if (rc) return rc;
        rc = tok_scan(c, rec_bases.p, n_rec, base_at.p, &total_bases);
        if (rc) return rc;
        rc = read_flags(&fl);
        if (rc) return rc;
        if (fl) { *declined = true; return MC_OK; }
        if (n_reads) {
            rc = reserve_words(total_bases, n_reads);
            if (rc) return rc;
            if (!off_out) { HIPCHK(c, offsets.alloc(&c->tok_pool, n_reads + 1)); off_out = offsets.p; }
            hipLaunchKernelGGL(tok::k_fq_emit, dim3(g), dim3(tok::T_THREADS), 0, c->stream, text.p, n, nl.p, n_nl, n_rec, f.offset, piece_at.p, base_at.p,
                               rec_bases.p, n_reads, total_bases, off_out, dst, dst_base);
            HIPCHK(c, hipGetLastError());
        }
    }
    rc = read_flags(&fl);
    if (rc) return rc;
    if (fl) { *declined = true; return MC_OK; }  // (cannot happen after the line pass; the read store was not advanced)
    const double t2 = now();
    if (n_reads && defer) {  // counted with the rest of the file (tok_flush_locked)
        pend->reads += n_reads;
        pend->bases += total_bases;
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (the text buffer goes back to its pool)
    } else if (n_reads) {
        rc = add_reads_dev_locked(c, dst, off_out, n_reads, total_bases, in_store);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (dbg)
        fprintf(stderr, "[ingest] device tokeniser: %.1f MB text, %llu lines, %llu reads, %llu bases: tokenise %.3f s, count %.3f s\n", n / 1e6,
                (unsigned long long)n_lines, (unsigned long long)n_reads, (unsigned long long)total_bases, t2 - t1, now() - t2);
    *n_reads_out = n_reads;
    return MC_OK;
}

int mc_add_reads_file(mc_ctx *c, const char *path, uint64_t *n_reads)
{
    if (!c) return MC_EINVAL;
    if (n_reads) *n_reads = 0;
    if (!path) return fail(c, MC_EINVAL, "mc_add_reads_file: null path");
    try {
        int rc = MC_OK;
        auto sink = [&](mch::PackedBatch &b) {
            if (rc == MC_OK) rc = mc_add_reads_packed(c, b.words.data(), b.offsets.data(), b.n_reads());
        };
        // Uncompressed FASTA / FASTQ: the bytes go to the device in chunks cut at record starts and are tokenised there
        // (csrc/tokenizer.h); a chunk the device declines goes through the host parser, as does any other kind of file
        // (MC_TOKENIZER=host: every file).  Host batches hold 2^20 reads; the context's lock is taken per batch / chunk.
        const char *mode = getenv("MC_TOKENIZER");
        mch::PlainReadsFile f;
        if (!(mode && !strcmp(mode, "host")) && mch::map_plain_reads(path, &f)) {
            // chunks of 256 MB: the bytes of chunk i + 1 cross the link (a helper thread, pinned staging buffers) while
            // the kernels tokenise and count chunk i
            const char *e_chunk = getenv("MC_TOKENIZER_CHUNK_BYTES");
            const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(e_chunk && *e_chunk ? strtoull(e_chunk, nullptr, 10) : (1ull << 28), 64), 3ull << 29);
            const bool dbg = getenv("MC_INGEST_DEBUG") != nullptr;
            auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            const double t_begin = now();
            std::vector<std::pair<const char *, const char *>> cuts;
            for (const char *b = f.p, *end = f.p + f.n; b < end;) {
                const char *e = (uint64_t)(end - b) <= chunk + chunk / 4 ? end : mch::plain_record_start(f, b + chunk);
                cuts.emplace_back(b, e);
                b = e;
            }
            struct Upload {
                PoolBuf<uint8_t> text;
                std::thread th;
                bool ok = true;
            };
            if (c->rs_enabled) {  // the read store grows once, not chunk by chunk (FASTA: a base a byte at most; FASTQ: half that)
                std::lock_guard<std::mutex> g(c->mu);
                HIPCHK(c, hipSetDevice(c->cfg.device));
                int r = rs_reserve(c, (f.fastq ? f.n / 2 : f.n) / 32 + 2 * cuts.size() + 2);
                if (r) return r;
            }
            std::unique_ptr<Upload> cur, nxt;
            auto start_upload = [&](size_t i, std::unique_ptr<Upload> &u) -> int {
                u.reset(new Upload);
                const uint64_t n = (uint64_t)(cuts[i].second - cuts[i].first);
                {
                    std::lock_guard<std::mutex> g(c->mu);
                    HIPCHK(c, hipSetDevice(c->cfg.device));
                    HIPCHK(c, u->text.alloc(&c->tok_pool, (n + tok::T_TILE - 1) / tok::T_TILE * tok::T_TILE));
                }
                Upload *up = u.get();
                const char *b = cuts[i].first;
                up->th = std::thread([c, up, b, n, &f] { up->ok = h2d_pinned(c, up->text.p, b, n, f.fd, (uint64_t)(b - f.p)); });
                return MC_OK;
            };
            auto finish = [&](std::unique_ptr<Upload> &u) {  // (the pool is the context's: blocks go back under its lock)
                if (!u) return;
                if (u->th.joinable()) u->th.join();
                std::lock_guard<std::mutex> g(c->mu);
                u.reset();
            };
            std::unique_ptr<TokPending> pend(new TokPending);  // (its pool block goes back under the context's lock)
            auto drop_pend = [&] {
                std::lock_guard<std::mutex> g(c->mu);
                pend.reset();
            };
            uint64_t total = 0;
            rc = cuts.empty() ? MC_OK : start_upload(0, cur);
            for (size_t i = 0; i < cuts.size() && rc == MC_OK; i++) {
                cur->th.join();
                if (!cur->ok) {
                    std::lock_guard<std::mutex> g(c->mu);
                    rc = fail(c, MC_EHIP, "mc_add_reads_file: host-to-device copy failed");
                    break;
                }
                if (i + 1 < cuts.size()) {
                    rc = start_upload(i + 1, nxt);
                    if (rc != MC_OK) break;
                }
                uint64_t got = 0;
                bool declined = false;
                {
                    std::lock_guard<std::mutex> g(c->mu);
                    rc = tokenize_chunk_locked(c, f, cuts[i].first, cuts[i].second, cur->text.p, &got, &declined, pend.get());
                    if (rc == MC_OK && declined) rc = tok_flush_locked(c, *pend);  // (the host's batches are appended behind what is counted)
                }
                if (rc != MC_OK) break;
                if (declined) {
                    try {
                        got = mch::parse_plain_range(f, cuts[i].first, cuts[i].second, 1u << 20, sink);
                    } catch (...) {
                        finish(cur);
                        finish(nxt);
                        drop_pend();
                        throw;
                    }
                    if (rc != MC_OK) break;
                }
                total += got;
                if (i == 0 && cuts.size() > 1) {
                    // No capacity hint that still holds: the first chunk says how many distinct k-mers a byte of this file
                    // brings, the file's size says how many chunks follow -- the table goes to its final size now, with one
                    // chunk's keys to move, instead of being rebuilt every other chunk (10 M reads with 1 % errors in six
                    // chunks: 110 ms of counting against ~45).  An over-estimate (the later chunks repeat k-mers of the first)
                    // costs memory, bounded by a third of what the device has free.
                    std::lock_guard<std::mutex> g(c->mu);
                    unsigned long long used = 0;
                    uint32_t fatal = 0;
                    rc = read_counters(c, &used, &fatal);
                    if (rc != MC_OK) break;
                    const bool hint_holds = c->cfg.capacity_hint && used < c->cfg.capacity_hint;
                    if (!hint_holds && used) {
                        const double scale = (double)f.n / (double)(cuts[0].second - cuts[0].first);
                        const double load = c->mm_k ? 0.36 : 0.6;
                        size_t fr = 0, tot = 0;
                        if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = 0;
                        uint64_t want_slots = (uint64_t)((double)used * scale / load);
                        want_slots = std::min<uint64_t>(want_slots, fr / 3 / sizeof(Slot));
                        const uint64_t want = regions_for(c, want_slots);
                        if (want > c->n_regions && want_slots > c->n_slots()) {
                            if (dbg) fprintf(stderr, "[ingest] %llu distinct k-mers after %.0f MB of %.0f MB: table to %.1f GB\n", used,
                                             (cuts[0].second - cuts[0].first) / 1e6, f.n / 1e6, (double)(want << c->sb) * sizeof(Slot) / 1e9);
                            rc = table_grow(c, want);
                            if (rc != MC_OK) break;
                            c->solid_list_fresh = false;
                        }
                    }
                }
                finish(cur);
                cur = std::move(nxt);
            }
            finish(cur);
            finish(nxt);
            if (rc == MC_OK) {
                std::lock_guard<std::mutex> g(c->mu);
                rc = tok_flush_locked(c, *pend);
            }
            drop_pend();
            if (rc != MC_OK) return rc;
            if (dbg) fprintf(stderr, "[ingest] device tokeniser: %zu chunk(s), %.1f MB in %.3f s\n", cuts.size(), f.n / 1e6, now() - t_begin);
            if (n_reads) *n_reads = total;
            return MC_OK;
        }
        const uint64_t n = mch::load_reads_file(path, 1u << 20, sink);
        if (rc != MC_OK) return rc;
        if (n_reads) *n_reads = n;
        return MC_OK;
    } catch (const mch::Error &e) {
        std::lock_guard<std::mutex> g(c->mu);
        return fail(c, MC_EINVAL, "%s", e.what());
    } catch (const std::bad_alloc &) {
        std::lock_guard<std::mutex> g(c->mu);
        return fail(c, MC_ENOMEM, "mc_add_reads_file: out of host memory");
    }
}

int mc_finalize_counts(mc_ctx *c, uint64_t *n_distinct)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    unsigned long long *h = c->h_scratch;
    {
        int rc = materialize(c);
        if (rc) return rc;
        // (one wait for all the counters: the parked additions' among them -- a wait of its own before, for a number that is 0
        // in every run that fits its table)
        HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, 9 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const bool parked = h[7] != 0;
        if (parked) rc = drain_parked(c);
        // hash keys in minimizer bins: a key that different k-mers brought to different regions gets the sum of its counters
        // wherever it is read, and counts once (dup_check.h; src/io/LargeKIOUtils.java:46-49 has one counter a hash)
        const bool joins = hash_bins(c) && !c->virgin && !c->dup.checked;
        if (!rc) rc = ensure_dups(c);
        if (rc) return rc;
        if (parked || joins) {  // (the counters may have moved)
            HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, 9 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    const uint32_t fatal = (uint32_t)h[8];
    if (fatal) return fail(c, MC_EOVERFLOW, "a k-mer table region filled up (hash skew)");
    c->n_used_host = h[0];
    c->finalized = true;
    const uint64_t shadows = c->dup.merged ? c->dup.n_tw - c->dup.n_keys : 0;  // (slots beyond a key's first)
    if (n_distinct) *n_distinct = h[0] - shadows + (h[1] ? 1 : 0);
    return MC_OK;
}

int mc_get_dev(mc_ctx *c, const int64_t *d_keys, uint64_t n, int16_t *d_out)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->finalized) return fail(c, MC_ESTATE, "mc_get: call mc_finalize_counts first");
    if ((!d_keys || !d_out) && n) return fail(c, MC_EINVAL, "mc_get_dev: null pointer");
    if (n == 0) return MC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (hash_bins(c) && !c->virgin) {  // (hash keys in minimizer bins: one sweep of the table answers all the queries, k_getq_*)
        if (n >= (1ull << 31)) return fail(c, MC_EINVAL, "mc_get: at most 2^31 - 1 keys a call on this table");
        uint64_t qn = 1024;
        while (qn < 2 * n) qn <<= 1;
        DevBuf<unsigned long long> qk;
        DevBuf<uint32_t> qi;
        HIPCHK(c, qk.alloc(qn));
        HIPCHK(c, qi.alloc(qn));
        HIPCHK(c, hipMemsetAsync(qk.p, 0xFF, qn * sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipMemsetAsync(d_out, 0xFF, n * sizeof(int16_t), c->stream));  // (-1: absent)
        hipLaunchKernelGGL(k_getq_build, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, d_keys, n, qk.p, qi.p, qn - 1);
        hipLaunchKernelGGL(k_getq_sweep, dim3(grid_for(c->n_slots(), 256, 256 * 16)), dim3(256), 0, c->stream, c->slots, c->n_slots(), qk.p, qi.p, qn - 1, d_out);
        hipLaunchKernelGGL(k_getq_finish, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, d_keys, n, qk.p, qi.p, qn - 1, d_out, c->d_ctr + 1);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return MC_OK;
    }
    if (int brc = by_key_ready(c)) return brc;
    hipLaunchKernelGGL(k_get, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, d_keys, n, d_out, c->view());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MC_OK;
}

int mc_get(mc_ctx *c, const int64_t *keys, uint64_t n, int16_t *out)
{
    if (!c) return MC_EINVAL;
    if ((!keys || !out) && n) return fail(c, MC_EINVAL, "mc_get: null pointer");
    if (n == 0) return MC_OK;
    DevBuf<int64_t> dk;
    DevBuf<int16_t> dout;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (!c->finalized) return fail(c, MC_ESTATE, "mc_get: call mc_finalize_counts first");
        HIPCHK(c, hipSetDevice(c->cfg.device));
        HIPCHK(c, alloc_result(c, dk, n));
        HIPCHK(c, alloc_result(c, dout, n));
        HIPCHK(c, hipMemcpyAsync(dk.p, keys, n * 8, hipMemcpyHostToDevice, c->stream));
    }
    int rc = mc_get_dev(c, dk.p, n, dout.p);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipMemcpy(out, dout.p, n * 2, hipMemcpyDeviceToHost));
    return MC_OK;
}

int mc_kmer_keys(mc_ctx *c, const uint64_t *hi, const uint64_t *lo, uint64_t n, int64_t *out_keys)
{
    if (!c) return MC_EINVAL;
    if ((!lo || !out_keys) && n) return fail(c, MC_EINVAL, "mc_kmer_keys: null pointer");
    for (uint64_t i = 0; i < n; i++) {
        const Kmer v{hi ? hi[i] : 0, lo[i]};
        out_keys[i] = key_of_mode(v, c->cfg.k, c->cfg.key_mode);
    }
    return MC_OK;
}

uint32_t mc_key_owner(int64_t key, uint32_t n_owners)
{
    return n_owners ? (uint32_t)((fmix64((uint64_t)key) & 0xFFFFFFFFull) % n_owners) : 0;
}

int mc_extract_keys_dev(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads, uint64_t n_bases,
                        uint32_t n_owners, int64_t *d_keys, uint32_t *d_hints, uint64_t cap, uint64_t *owner_offsets)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    struct ResetRule { mc_ctx *c; ~ResetRule() { c->extract_by_minimizer = false; } } reset_rule{c};  // (the group's choice holds for this call only)
    if (c->extract_by_minimizer && !(c->cfg.key_mode == MC_KEY_PACKED && c->cfg.k >= SK_MIN_K))
        return fail(c, MC_ESTATE, "internal: keys can only be dealt by minimizer where records are (packed keys, k >= %d)", SK_MIN_K);
    if (!owner_offsets || n_owners == 0 || n_owners > PT_MAX_BUCKETS)
        return fail(c, MC_EINVAL, "mc_extract_keys_dev: bad n_owners / owner_offsets");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    for (uint32_t o = 0; o <= n_owners; o++) owner_offsets[o] = 0;
    if (n_reads == 0) return MC_OK;
    mc_ctx::Pipe &P = c->pipe;
    uint64_t first_off = 0, last_off = 0;
    HIPCHK(c, hipMemcpyAsync(&first_off, d_off, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&last_off, d_off + n_reads, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (last_off != n_bases)
        return fail(c, MC_EINVAL, "mc_extract_keys_dev: read_offsets[n_reads]=%llu but n_bases=%llu",
                    (unsigned long long)last_off, (unsigned long long)n_bases);
    if (d_keys) {  // (the pass that writes the keys: their read pointers refer to this context's read store)
        const int64_t in_store = c->extract_in_store;  // (mc_group: tokenised on this device, in the store already)
        c->extract_in_store = -1;
        int rrc = rs_append(c, d_words, first_off, last_off, in_store);
        if (rrc) return rrc;
    } else {
        c->cur_ptr_base = ~0ull;
    }
    const uint64_t n_tiles_abs = (last_off + PT_TILE - 1) / PT_TILE;
    int rc = ensure_buf(c, &P.tile_first, &P.tiles1_cap, n_tiles_abs);
    if (rc) return rc;
    if (!P.cursors1) HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&P.cursors1), PT_MAX_BUCKETS * CURSOR1_STRIDE * sizeof(uint32_t)));
    DevBuf<uint64_t> d_bases;
    HIPCHK(c, d_bases.alloc(n_owners));
    const SpillView nosp{nullptr, nullptr, nullptr, 0, nullptr};
    // pass 1: how many keys every owner gets
    HIPCHK(c, hipMemsetAsync(P.cursors1, 0, PT_MAX_BUCKETS * CURSOR1_STRIDE * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_tile_first_read, dim3((unsigned)((n_tiles_abs + 255) / 256)), dim3(256), 0, c->stream, d_off, n_reads,
                       n_tiles_abs, P.tile_first);
    launch_p1_reads(c, d_words, d_off, n_reads, first_off, last_off, n_tiles_abs, P.tile_first, n_owners, P.cursors1, 0, nullptr,
                    nullptr, nosp, 1, nullptr);
    HIPCHK(c, hipGetLastError());
    std::vector<uint32_t> raw((size_t)n_owners * CURSOR1_STRIDE);
    HIPCHK(c, hipMemcpyAsync(raw.data(), P.cursors1, raw.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<uint64_t> bases(n_owners);
    for (uint32_t o = 0; o < n_owners; o++) {
        bases[o] = owner_offsets[o];
        owner_offsets[o + 1] = owner_offsets[o] + raw[(size_t)o * CURSOR1_STRIDE];
    }
    if (!d_keys) return MC_OK;
    if (owner_offsets[n_owners] > cap)
        return fail(c, MC_EINVAL, "mc_extract_keys_dev: %llu keys but capacity %llu",
                    (unsigned long long)owner_offsets[n_owners], (unsigned long long)cap);
    // pass 2: the same tiles again, now writing every owner's keys (and hints) into its packed range
    HIPCHK(c, hipMemcpyAsync(d_bases.p, bases.data(), n_owners * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(P.cursors1, 0, PT_MAX_BUCKETS * CURSOR1_STRIDE * sizeof(uint32_t), c->stream));
    launch_p1_reads(c, d_words, d_off, n_reads, first_off, last_off, n_tiles_abs, P.tile_first, n_owners, P.cursors1, 0,
                    reinterpret_cast<uint64_t *>(d_keys), d_hints, nosp, 2, d_bases.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MC_OK;
}

uint64_t mc_superkmer_capacity(mc_ctx *c, uint64_t n_windows, uint64_t n_reads)
{
    if (!c || !c->sk_form) return 0;
    return sk_records_bound(c, n_windows, n_reads);
}

int mc_extract_superkmers_dev(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads, uint64_t n_bases,
                              uint32_t n_owners, uint64_t *d_recs, uint32_t *d_bins, uint64_t cap, uint64_t *owner_offsets)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->sk_form) return fail(c, MC_ESTATE, "mc_extract_superkmers_dev: this context does not count through super-k-mers (mc_superkmer_capacity() == 0)");
    if (!owner_offsets || !d_recs || !d_bins || n_owners == 0 || n_owners > PT_MAX_BUCKETS)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_dev: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    for (uint32_t o = 0; o <= n_owners; o++) owner_offsets[o] = 0;
    if (n_reads == 0) return MC_OK;
    mc_ctx::Pipe &P = c->pipe;
    uint64_t first_off = 0, last_off = 0;
    HIPCHK(c, hipMemcpyAsync(&first_off, d_off, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&last_off, d_off + n_reads, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (last_off != n_bases)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_dev: read_offsets[n_reads]=%llu but n_bases=%llu",
                    (unsigned long long)last_off, (unsigned long long)n_bases);
    {
        const int64_t in_store = c->extract_in_store;  // (mc_group: tokenised on this device, in the store already)
        c->extract_in_store = -1;
        int rrc = rs_append(c, d_words, first_off, last_off, in_store);
        if (rrc) return rrc;
    }
    // The pieces go into pipe.a_recs, where the merge kernel may have left the list of solid k-mers (P3Emit): from here on
    // that list is gone, whether or not this rank then adds any record of its own (as pipe_prepare says for every run).
    c->solid_list_fresh = false;
    // windows <= bases: capacity of the (owner, segment) pieces from the same bound the caller sized its buffer with
    const uint64_t bound = sk_records_bound(c, n_bases - first_off, n_reads);
    const uint32_t nseg = P1W_SEGMENTS;
    uint64_t seg_cap = (uint64_t)((double)bound / (double)n_owners / (double)nseg * 1.25) + 64;
    {   // a workgroup (= segment) takes whole tiles, one per wave and round: with few tiles some segments get more than others, or any at all
        const uint64_t n_waves = (uint64_t)nseg * P1W_WAVES;
        const uint64_t tiles = (last_off - first_off + P1W_TILE - 1) / P1W_TILE + 1, per_wg = (tiles + n_waves - 1) / n_waves * P1W_WAVES;
        const double per_tile = (double)bound / (double)std::max<uint64_t>(tiles - 1, 1);
        seg_cap = std::max<uint64_t>(seg_cap, (uint64_t)((double)per_wg * per_tile / (double)n_owners * 1.3) + 64);
    }
    if ((uint64_t)n_owners * nseg * seg_cap >= 0xFFFFFFFFull)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_dev: batch too large (split the reads)");
    const uint64_t n_tiles_abs = (last_off + P1W_TILE - 1) / P1W_TILE;
    int rc = ensure_buf(c, &P.tile_first, &P.tiles1_cap, n_tiles_abs);
    if (!rc) rc = ensure_buf(c, &P.a_recs, &P.a_recs_cap, (uint64_t)n_owners * nseg * seg_cap);
    if (!rc) rc = ensure_buf(c, &P.a_hints, &P.a_hints_cap, (uint64_t)n_owners * nseg * seg_cap);
    if (!rc) rc = ensure_buf(c, &P.seg_counts1, &P.segs1_cap, (uint64_t)n_owners * nseg);
    if (rc) return rc;
    if (!P.flags) {  // four flags and, behind them, the spill counter: cleared by one fill, read by one copy
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&P.flags), 4 * sizeof(uint32_t) + sizeof(unsigned long long)));
        P.spill_count = reinterpret_cast<unsigned long long *>(P.flags + 4);
    }
    HIPCHK(c, hipMemsetAsync(P.flags, 0, 4 * sizeof(uint32_t) + sizeof(unsigned long long), c->stream));
    DevBuf<unsigned long long> d_piece, d_owner;
    HIPCHK(c, d_piece.alloc((uint64_t)n_owners * nseg));
    HIPCHK(c, d_owner.alloc(n_owners + 1));
    const SkSpill none{nullptr, P.spill_count, 0, P.flags};  // no spill list: an overflow is reported
    launch_tile_first(c, d_off, n_reads, n_tiles_abs, P.tile_first, P1W_TILE);
    hipLaunchKernelGGL(k_sk1w_extract<true>, dim3(nseg), dim3(P1W_THREADS), 0, c->stream, d_words, d_off, n_reads, first_off,
                       last_off, n_tiles_abs, P.tile_first, c->cfg.k, n_owners, P.seg_counts1, seg_cap, P.a_recs, P.a_hints, none, c->cur_ptr_base);
    hipLaunchKernelGGL(k_sk_pack_offsets, dim3(1), dim3(1024), 0, c->stream, P.seg_counts1, n_owners, d_piece.p, d_owner.p, nseg);
    hipLaunchKernelGGL(k_sk_pack, dim3(std::min<uint32_t>(n_owners * nseg, 4096)), dim3(256), 0, c->stream, P.a_recs, P.a_hints,
                       P.seg_counts1, seg_cap, d_piece.p, n_owners * nseg, reinterpret_cast<uint4 *>(d_recs), d_bins, cap);
    HIPCHK(c, hipGetLastError());
    std::vector<unsigned long long> h_owner(n_owners + 1);
    uint32_t lost = 0;
    HIPCHK(c, hipMemcpyAsync(h_owner.data(), d_owner.p, (n_owners + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&lost, P.flags, sizeof lost, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (lost) return fail(c, MC_EOVERFLOW, "mc_extract_superkmers_dev: more super-k-mers than expected (unusually short runs); use mc_extract_keys_dev");
    if (h_owner[n_owners] > cap)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_dev: %llu records but capacity %llu", h_owner[n_owners], (unsigned long long)cap);
    for (uint32_t o = 0; o <= n_owners; o++) owner_offsets[o] = h_owner[o];
    return MC_OK;
}

// fine buckets of a binned exchange for n_owners owners whose tables are laid out like this context's (0: no binned form), and the
// coarse buckets an owner's records are dealt to by the first level (*sub)
static uint32_t skb_fine_buckets(const mc_ctx *c, uint32_t n_owners, uint32_t *sub)
{
    static const bool off = [] { const char *e = getenv("MC_EXCHANGE_BINNED"); return e && !strcmp(e, "0"); }();
    if (off || !c->sk_form || !c->mm_k || n_owners == 0 || n_owners > PT_MAX_BUCKETS) return 0;
    uint64_t n_leaves, np1;
    uint32_t g;
    if (plan_levels(c, true, &n_leaves, &g, &np1)) return 0;
    if (n_leaves / np1 <= 1 || (uint64_t)n_owners * np1 > SKB_MAX_CELLS) return 0;
    uint32_t sb = 1;  // about 512 first-level buckets in all (what k_sk1w_extract keeps open on one GPU), a power of two that divides np1
    while ((uint64_t)n_owners * sb * 2 <= PT_MAX_BUCKETS && np1 % (sb * 2) == 0) sb *= 2;
    if (np1 / sb > PT_MAX_LEAVES2 || (uint64_t)n_owners * sb > PT_MAX_BUCKETS1_SK) return 0;
    if (sub) *sub = sb;
    return (uint32_t)np1;
}

uint32_t mc_superkmer_fine_buckets(mc_ctx *c, uint32_t n_owners)
{
    if (!c) return 0;
    std::lock_guard<std::mutex> g(c->mu);
    return skb_fine_buckets(c, n_owners, nullptr);
}

int mc_extract_superkmers_binned_dev(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads, uint64_t n_bases,
                                     uint32_t n_owners, uint32_t n_fine, uint64_t *d_recs, uint32_t *d_ptrs, uint64_t cap,
                                     uint32_t *d_fine_counts, uint64_t *owner_offsets, uint64_t *owner_windows)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->sk_form) return fail(c, MC_ESTATE, "mc_extract_superkmers_binned_dev: this context does not count through super-k-mers (mc_superkmer_capacity() == 0)");
    if (!owner_offsets || !owner_windows || !d_recs || !d_ptrs || !d_fine_counts || n_owners == 0 || n_owners > PT_MAX_BUCKETS)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_binned_dev: bad argument");
    // the coarse buckets of the first level: a power of two that divides n_fine, about 512 buckets in all
    uint32_t sub = 1;
    while ((uint64_t)n_owners * sub * 2 <= PT_MAX_BUCKETS && n_fine % (sub * 2) == 0) sub *= 2;
    if (n_fine == 0 || (uint64_t)n_owners * n_fine > SKB_MAX_CELLS || n_fine / sub > PT_MAX_LEAVES2 || (uint64_t)n_owners * sub > PT_MAX_BUCKETS1_SK)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_binned_dev: %u fine buckets for %u owners (mc_superkmer_fine_buckets says how many)", n_fine, n_owners);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    for (uint32_t o = 0; o <= n_owners; o++) owner_offsets[o] = 0;
    for (uint32_t o = 0; o < n_owners; o++) owner_windows[o] = 0;
    const uint64_t n_cells = (uint64_t)n_owners * n_fine;
    HIPCHK(c, hipMemsetAsync(d_fine_counts, 0, n_cells * sizeof(uint32_t), c->stream));
    if (n_reads == 0) return MC_OK;
    mc_ctx::Pipe &P = c->pipe;
    uint64_t first_off = 0, last_off = 0;
    HIPCHK(c, hipMemcpyAsync(&first_off, d_off, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&last_off, d_off + n_reads, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (last_off != n_bases)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_binned_dev: read_offsets[n_reads]=%llu but n_bases=%llu",
                    (unsigned long long)last_off, (unsigned long long)n_bases);
    {
        const int64_t in_store = c->extract_in_store;  // (mc_group: tokenised on this device, in the store already)
        c->extract_in_store = -1;
        int rrc = rs_append(c, d_words, first_off, last_off, in_store);
        if (rrc) return rrc;
    }
    c->solid_list_fresh = false;  // (the pieces go into pipe.a_recs: mc_extract_superkmers_dev says what that ends)
    const uint64_t bound = sk_records_bound(c, n_bases - first_off, n_reads);
    const uint32_t nseg = P1W_SEGMENTS, n_buckets = n_owners * sub;
    uint64_t seg_cap = (uint64_t)((double)bound / (double)n_buckets / (double)nseg * 1.25) + 64;
    {   // a workgroup (= segment) takes whole tiles, one per wave and round: with few tiles some segments get more than others, or any at all
        const uint64_t n_waves = (uint64_t)nseg * P1W_WAVES;
        const uint64_t tiles = (last_off - first_off + P1W_TILE - 1) / P1W_TILE + 1, per_wg = (tiles + n_waves - 1) / n_waves * P1W_WAVES;
        const double per_tile = (double)bound / (double)std::max<uint64_t>(tiles - 1, 1);
        seg_cap = std::max<uint64_t>(seg_cap, (uint64_t)((double)per_wg * per_tile / (double)n_buckets * 1.3) + 64);
    }
    if ((uint64_t)n_buckets * nseg * seg_cap >= 0xFFFFFFFFull)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_binned_dev: batch too large (split the reads)");
    const uint64_t n_tiles_abs = (last_off + P1W_TILE - 1) / P1W_TILE;
    int rc = ensure_buf(c, &P.tile_first, &P.tiles1_cap, n_tiles_abs);
    if (!rc) rc = ensure_buf(c, &P.a_recs, &P.a_recs_cap, (uint64_t)n_buckets * nseg * seg_cap);
    if (!rc) rc = ensure_buf(c, &P.a_hints, &P.a_hints_cap, (uint64_t)n_buckets * nseg * seg_cap);
    if (!rc) rc = ensure_buf(c, &P.seg_counts1, &P.segs1_cap, (uint64_t)n_buckets * nseg);
    if (!rc) rc = ensure_buf(c, &P.skb_rows, &P.skb_rows_cap, (uint64_t)nseg * n_cells);
    if (!rc) rc = ensure_buf(c, &P.skb_cell_start, &P.skb_cell_start_cap, n_cells);
    if (!rc) rc = ensure_buf(c, &P.skb_small, &P.skb_small_cap, (uint64_t)P1W_SEGMENTS + 2 * PT_MAX_BUCKETS + 8);
    if (rc) return rc;
    if (!P.flags) {  // four flags and, behind them, the spill counter: cleared by one fill, read by one copy
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&P.flags), 4 * sizeof(uint32_t) + sizeof(unsigned long long)));
        P.spill_count = reinterpret_cast<unsigned long long *>(P.flags + 4);
    }
    HIPCHK(c, hipMemsetAsync(P.flags, 0, 4 * sizeof(uint32_t) + sizeof(unsigned long long), c->stream));
    unsigned long long *d_owner = P.skb_small, *d_win = P.skb_small + n_owners + 1;  // owner offsets (n_owners + 1), owner windows (n_owners)
    HIPCHK(c, hipMemsetAsync(d_win, 0, n_owners * sizeof(unsigned long long), c->stream));
    const SkSpill none{nullptr, P.spill_count, 0, P.flags};  // no spill list: an overflow is reported
    launch_tile_first(c, d_off, n_reads, n_tiles_abs, P.tile_first, P1W_TILE);
    hipLaunchKernelGGL((k_sk1w_extract<true, false, true>), dim3(nseg), dim3(P1W_THREADS), 0, c->stream, d_words, d_off, n_reads, first_off,
                       last_off, n_tiles_abs, P.tile_first, c->cfg.k, n_owners, P.seg_counts1, seg_cap, P.a_recs, P.a_hints, none, c->cur_ptr_base,
                       0u, 0u, SkBinned{sub, n_fine, P.skb_rows, d_win});
    hipLaunchKernelGGL(k_skb_colsum, dim3((unsigned)((n_cells + 255) / 256), 16), dim3(256), 0, c->stream, P.skb_rows, nseg, (uint32_t)n_cells, d_fine_counts);
    hipLaunchKernelGGL(k_skb_offsets, dim3(1), dim3(1024), 0, c->stream, d_fine_counts, n_owners, n_fine, P.skb_cell_start, d_owner);
    // every (owner, coarse bucket)'s segments into the owner's fine buckets, at their exact places in the caller's buffer
    hipLaunchKernelGGL((k_sk2_scatter_staged<MC_SK2_ITEMS, false, true>), dim3(n_buckets), dim3(PT_THREADS), 0, c->stream, P.a_recs, P.a_hints, seg_cap,
                       P.seg_counts1, n_buckets, sub, n_fine / sub, nullptr, cap, reinterpret_cast<uint4 *>(d_recs), d_ptrs, none, nseg, nullptr, P.skb_cell_start);
    HIPCHK(c, hipGetLastError());
    std::vector<unsigned long long> h(2 * n_owners + 1);
    uint32_t lost = 0;
    HIPCHK(c, hipMemcpyAsync(h.data(), d_owner, (2 * n_owners + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&lost, P.flags, sizeof lost, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (lost) return fail(c, MC_EOVERFLOW, "mc_extract_superkmers_binned_dev: more super-k-mers than expected (unusually short runs); use mc_extract_keys_dev");
    if (h[n_owners] > cap)
        return fail(c, MC_EINVAL, "mc_extract_superkmers_binned_dev: %llu records but capacity %llu", h[n_owners], (unsigned long long)cap);
    for (uint32_t o = 0; o <= n_owners; o++) owner_offsets[o] = h[o];
    for (uint32_t o = 0; o < n_owners; o++) owner_windows[o] = h[n_owners + 1 + o];
    return MC_OK;
}

static int add_superkmers_locked(mc_ctx *c, const uint64_t *d_recs, const uint32_t *d_bins, uint64_t n);

int mc_add_superkmers_binned_dev(mc_ctx *c, const uint64_t *d_recs, const uint32_t *d_ptrs, uint64_t n, uint64_t n_windows, uint32_t n_fine,
                                 uint32_t n_parts, const uint64_t *part_offsets, const uint32_t *d_part_counts)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->sk_form) return fail(c, MC_ESTATE, "mc_add_superkmers_binned_dev: this context does not count through super-k-mers");
    if (n && (!d_recs || !d_ptrs || !part_offsets || !d_part_counts || n_parts == 0 || n_fine == 0)) return fail(c, MC_EINVAL, "mc_add_superkmers_binned_dev: bad argument");
    if (n) {
        bool ok = part_offsets[0] == 0 && part_offsets[n_parts] == n;
        for (uint32_t p = 0; p < n_parts && ok; p++) ok = part_offsets[p] <= part_offsets[p + 1];
        if (!ok) return fail(c, MC_EINVAL, "mc_add_superkmers_binned_dev: the parts must lie back to back over the %llu records", (unsigned long long)n);
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (n && c->mm_k) {
        c->ptr_tries = c->all_ptrs ? 1 : 16;  // (records of ranks that keep no read store carry no pointer)
        int rc = add_records_binned(c, reinterpret_cast<const uint4 *>(d_recs), d_ptrs, n, n_windows, n_fine, n_parts, part_offsets, d_part_counts);
        if (rc != 2) {
            if (rc) return rc;
            c->st.binned_runs++;
            counts_changed(c);
            c->solid_cov = -1; c->solid_external = false;
            return MC_OK;
        }
    }
    return add_superkmers_locked(c, d_recs, d_ptrs, n);  // (this table's buckets are not unions of the fine ones: records are records in any order)
}

int mc_add_superkmers_dev(mc_ctx *c, const uint64_t *d_recs, const uint32_t *d_bins, uint64_t n)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->sk_form) return fail(c, MC_ESTATE, "mc_add_superkmers_dev: this context does not count through super-k-mers");
    if ((!d_recs || !d_bins) && n) return fail(c, MC_EINVAL, "mc_add_superkmers_dev: null pointer");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return add_superkmers_locked(c, d_recs, d_bins, n);
}

static int add_superkmers_locked(mc_ctx *c, const uint64_t *d_recs, const uint32_t *d_bins, uint64_t n)
{
    c->ptr_tries = c->all_ptrs ? 1 : 16;  // (records of ranks that keep no read store carry no pointer)
    if (!c->mm_k) {  // the context gave up minimizer bins (to_hash_regions): the records are expanded by the direct kernel
        c->solid_tracked = false;
        c->solid_list_fresh = false;
        uint64_t i = 0;
        while (i < n) {
            uint64_t allowed;
            int rc = table_reserve(c, (n - i) * SK_MAX_WINDOWS, &allowed);
            if (rc) return rc;
            const uint64_t m = std::min<uint64_t>(std::max<uint64_t>(allowed / SK_MAX_WINDOWS, 1), n - i);
            hipLaunchKernelGGL(k_sk_add_records, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, reinterpret_cast<const uint4 *>(d_recs) + i, d_bins + i, m,
                               c->cfg.k, c->view(), 0u, c->d_ctr + 6);
            HIPCHK(c, hipGetLastError());
            i += m;
        }
        counts_changed(c);
        c->solid_cov = -1; c->solid_external = false;
        return MC_OK;
    }
    const uint64_t max_batch = 1ull << 28;  // records per pipeline run (32-bit bucket indices)
    for (uint64_t i = 0; i < n; i += max_batch) {
        const uint64_t m = std::min<uint64_t>(max_batch, n - i);
        int rc = add_records_partitioned(c, reinterpret_cast<const uint4 *>(d_recs) + i, d_bins + i, m);
        if (rc) return rc;
    }
    counts_changed(c);
    c->solid_cov = -1; c->solid_external = false;
    return MC_OK;
}

int mc_export_dev(mc_ctx *c, int min_cov, int64_t *d_keys, int16_t *d_counts, uint32_t *d_hints, uint64_t cap,
                  uint64_t *n_out)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->finalized) return fail(c, MC_ESTATE, "mc_export: call mc_finalize_counts first");
    if (!n_out) return fail(c, MC_EINVAL, "mc_export: n_out is null");
    if (d_keys && !d_counts) return fail(c, MC_EINVAL, "mc_export: counts is null");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    unsigned long long *cursor = c->d_ctr + 2;
    if (!d_keys && c->solid_tracked && c->cov_hint == min_cov && min_cov > 0 && !c->virgin) {
        cursor = c->d_ctr + 6;  // counting only, and k_p3_merge kept the number of keys at this threshold up to date
    } else if (d_keys && c->solid_list_fresh && c->solid_tracked && c->cov_hint == min_cov && min_cov > 0 && !c->virgin && c->mm_k) {
        // ... and listed them (the list stays: a BFS set-up on this context can still use it)
        HIPCHK(c, hipMemsetAsync(cursor, 0, sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(k_export_list, dim3(std::min<uint32_t>(c->solid_list_segs, 2048u)), dim3(256), 0, c->stream, c->pipe.a_recs,
                           c->pipe.emit_counts, c->solid_list_segs, c->solid_list_segcap, d_keys, d_counts, d_hints, cap, cursor);
        HIPCHK(c, hipGetLastError());
        c->st.solid_list_builds++;
    } else {
        HIPCHK(c, hipMemsetAsync(cursor, 0, sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(k_export, dim3(grid_for(c->n_slots(), 256)), dim3(256), 0, c->stream, c->slots, c->n_slots(),
                           min_cov, d_keys, d_counts, d_hints, cap, cursor, c->dup_filter());
        HIPCHK(c, hipGetLastError());
    }
    unsigned long long n = 0, empty_cnt = 0;
    HIPCHK(c, hipMemcpyAsync(&n, cursor, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&empty_cnt, c->d_ctr + 1, sizeof empty_cnt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the key equal to EMPTY_KEY is counted out of band (hash modes only)
    const int ec = empty_cnt > 32767ull ? 32767 : (int)empty_cnt;
    if (empty_cnt && ec >= min_cov) {
        if (d_keys && n < cap) {
            const int64_t kk = (int64_t)EMPTY_KEY;
            const int16_t cc = (int16_t)ec;
            HIPCHK(c, hipMemcpy(d_keys + n, &kk, 8, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(d_counts + n, &cc, 2, hipMemcpyHostToDevice));
            if (d_hints) HIPCHK(c, hipMemset(d_hints + n, 0, 4));
        }
        n++;
    }
    *n_out = n;
    if (d_keys && n > cap) return fail(c, MC_EINVAL, "mc_export: %llu pairs but capacity %llu", n, (unsigned long long)cap);
    return MC_OK;
}

int mc_export(mc_ctx *c, int min_cov, int64_t *keys, int16_t *counts, uint64_t cap, uint64_t *n_out)
{
    if (!c) return MC_EINVAL;
    if (!keys) return mc_export_dev(c, min_cov, nullptr, nullptr, nullptr, 0, n_out);
    DevBuf<int64_t> dk;
    DevBuf<int16_t> dc;
    {
        std::lock_guard<std::mutex> g(c->mu);
        HIPCHK(c, hipSetDevice(c->cfg.device));
        HIPCHK(c, alloc_result(c, dk, cap));
        HIPCHK(c, alloc_result(c, dc, cap));
    }
    int rc = mc_export_dev(c, min_cov, dk.p, dc.p, nullptr, cap, n_out);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipMemcpy(keys, dk.p, *n_out * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(counts, dc.p, *n_out * 2, hipMemcpyDeviceToHost));
    return MC_OK;
}

// ---- .kmers.bin (src/io/IOUtils.java:39-65 printKmers, :94-126 loadKmers, src/io/KmersLoadWorker.java:9-23)

int mc_save_kmers(mc_ctx *c, const char *bin_path, const char *stat_path, int threshold, uint64_t *n_total, uint64_t *n_written)
{
    if (!c) return MC_EINVAL;
    if (n_total) *n_total = 0;
    if (n_written) *n_written = 0;
    if (!bin_path) return fail(c, MC_EINVAL, "mc_save_kmers: null path");
    uint64_t n = 0;
    int rc = mc_export(c, 0, nullptr, nullptr, 0, &n);
    if (rc) return rc;
    std::vector<int64_t> keys(std::max<uint64_t>(n, 1));
    std::vector<int16_t> counts(std::max<uint64_t>(n, 1));
    if (n) {
        rc = mc_export(c, 0, keys.data(), counts.data(), n, &n);
        if (rc) return rc;
    }
    FILE *f = fopen(bin_path, "wb");
    if (!f) return fail(c, MC_EINVAL, "mc_save_kmers: cannot write %s", bin_path);
    std::vector<unsigned char> buf;
    buf.reserve(10u << 20);
    std::vector<uint64_t> hist(32768, 0);  // QuickQuantitativeStatistics<Short>: frequency -> number of such k-mers
    uint64_t good = 0;
    bool io_ok = true;
    for (uint64_t i = 0; i < n; i++) {
        const int16_t v = counts[i];
        hist[(size_t)v]++;
        if (v > threshold) {  // DataOutputStream.writeLong / writeShort: big-endian
            const uint64_t kk = (uint64_t)keys[i];
            for (int b = 7; b >= 0; b--) buf.push_back((unsigned char)(kk >> (8 * b)));
            buf.push_back((unsigned char)((uint16_t)v >> 8));
            buf.push_back((unsigned char)((uint16_t)v & 0xFF));
            good++;
            if (buf.size() >= (10u << 20)) { io_ok = io_ok && fwrite(buf.data(), 1, buf.size(), f) == buf.size(); buf.clear(); }
        }
    }
    if (!buf.empty()) io_ok = io_ok && fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    io_ok = (fclose(f) == 0) && io_ok;
    if (!io_ok) return fail(c, MC_EINVAL, "mc_save_kmers: error writing %s", bin_path);
    if (stat_path) {
        FILE *sf = fopen(stat_path, "w");
        if (!sf) return fail(c, MC_EINVAL, "mc_save_kmers: cannot write %s", stat_path);
        fputs("# k-mer frequency\tnumber of such k-mers\n", sf);
        for (size_t v = 0; v < hist.size(); v++)
            if (hist[v]) fprintf(sf, "%zu\t%llu\n", v, (unsigned long long)hist[v]);
        fputs("\n", sf);  // (println of a string that already ends with a newline)
        if (fclose(sf) != 0) return fail(c, MC_EINVAL, "mc_save_kmers: error writing %s", stat_path);
    }
    if (n_total) *n_total = n;
    if (n_written) *n_written = good;
    return MC_OK;
}

int mc_load_kmers(mc_ctx *c, const char *path, int freq_threshold, uint64_t *n_records, uint64_t *n_added)
{
    if (!c) return MC_EINVAL;
    if (n_records) *n_records = 0;
    if (n_added) *n_added = 0;
    if (!path) return fail(c, MC_EINVAL, "mc_load_kmers: null path");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(c, MC_EINVAL, "Failed to read from file %s", path);
    const size_t batch = 8u << 20;  // records per upload
    std::vector<unsigned char> raw(batch * 10);
    std::vector<int64_t> keys(batch);
    std::vector<int16_t> counts(batch);
    DevBuf<int64_t> dk;
    DevBuf<int16_t> dc;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (hipSetDevice(c->cfg.device) != hipSuccess || dk.alloc(batch) != hipSuccess || dc.alloc(batch) != hipSuccess) {
            fclose(f);
            return fail(c, MC_ENOMEM, "mc_load_kmers: device allocation failed");
        }
    }
    uint64_t total = 0, added = 0;
    int rc = MC_OK;
    for (;;) {
        const size_t got = fread(raw.data(), 1, raw.size(), f);
        if (got == 0) break;
        if (got % 10 != 0) { rc = fail(c, MC_EINVAL, "Can't load kmers from file %s: its length is not a multiple of the 10-byte record", path); break; }
        size_t m = 0;
        for (size_t i = 0; i < got; i += 10) {
            uint64_t kk = 0;
            for (int b = 0; b < 8; b++) kk = (kk << 8) | raw[i + b];
            const int16_t v = (int16_t)(((uint16_t)raw[i + 8] << 8) | raw[i + 9]);
            total++;
            if (v > freq_threshold) { keys[m] = (int64_t)kk; counts[m] = v; m++; }
        }
        if (m) {
            {
                std::lock_guard<std::mutex> g(c->mu);
                if (hipMemcpy(dk.p, keys.data(), m * 8, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(dc.p, counts.data(), m * 2, hipMemcpyHostToDevice) != hipSuccess) {
                    rc = fail(c, MC_EHIP, "mc_load_kmers: upload failed");
                    break;
                }
            }
            rc = mc_add_pairs_dev(c, dk.p, dc.p, nullptr, m);
            if (rc) break;
            added += m;
        }
    }
    fclose(f);
    if (rc) return rc;
    if (n_records) *n_records = total;
    if (n_added) *n_added = added;
    return MC_OK;
}

int mc_get_stats(mc_ctx *c, mc_stats *out)
{
    if (!c || !out) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    c->st.dup_keys = c->dup.merged ? c->dup.n_keys : 0;
    *out = c->st;
    return MC_OK;
}

int mc_reset_stats(mc_ctx *c)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    const uint64_t slots = c->st.table_slots, bytes = c->st.table_bytes, left = c->st.left_bins, unchecked = c->st.dup_unchecked;
    c->st = mc_stats{};
    c->st.table_slots = slots;
    c->st.table_bytes = bytes;
    c->st.left_bins = left;  // (states of the table, not counters)
    c->st.dup_unchecked = unchecked;
    return MC_OK;
}

int mc_trim(mc_ctx *c)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    mc_ctx::Pipe &P = c->pipe;
    const int dev = c->cfg.device;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (the blocks may be reused by another context at once: see mc_destroy)
    if (c->pipe_stream) HIPCHK(c, hipStreamSynchronize(c->pipe_stream));
    // (the list of solid k-mers sits in a_recs, its fill levels in emit_counts: they stay while the list is valid)
    uint4 *keep_recs = nullptr;
    uint64_t keep_cap = 0;
    uint32_t *keep_counts = nullptr;
    if (c->solid_list_fresh) {
        keep_recs = P.a_recs; keep_cap = P.a_recs_cap; keep_counts = P.emit_counts;
        P.a_recs = nullptr; P.a_recs_cap = 0; P.emit_counts = nullptr;
    }
    P.release(dev);
    P.a_recs = keep_recs; P.a_recs_cap = keep_cap; P.emit_counts = keep_counts;
    g_scratch_pool.put(dev, c->d_ovf_tmp, c->ovf_tmp_cap * sizeof(uint4));
    c->d_ovf_tmp = nullptr; c->ovf_tmp_cap = 0;
    dup_release_streams(c);
    g_scratch_pool.release(dev);
    g_table_pool.release(dev);
    return MC_OK;
}

int mc_synth_reads_dev(mc_ctx *c, uint64_t gseed, uint64_t n_contigs, uint64_t contig_len, uint64_t rseed,
                       uint64_t first_read, uint64_t n_reads, uint32_t L, uint32_t err, uint64_t *d_words,
                       uint64_t *d_off)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!d_words || !d_off || n_contigs == 0 || L == 0 || contig_len < L)
        return fail(c, MC_EINVAL, "mc_synth_reads_dev: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t n_words = (n_reads * (uint64_t)L + 31) / 32 + 1;
    hipLaunchKernelGGL(k_synth_reads, dim3(grid_for(n_words, 256)), dim3(256), 0, c->stream, gseed, n_contigs,
                       contig_len, rseed, first_read, n_reads, L, err, d_words, n_words, d_off);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MC_OK;
}

int mc_synth_genome(uint64_t gseed, uint64_t start, uint64_t n, uint8_t *codes)
{
    if (!codes && n) return MC_EINVAL;
    for (uint64_t i = 0; i < n; i++) codes[i] = (uint8_t)(splitmix(gseed, start + i) & 3);
    return MC_OK;
}

// ------------------------------------------------------------------------------------------ BFS driver

// The arrays of a BFS result are page-locked host memory from a process-wide pool (a block serves again once
// mc_bfs_result_free has returned it): the copies from the device run at the link rate without a staging pass, and a
// reused block costs no page faults -- 0.4 ms of a 29 ms step for two passes of 10^5 k-mers.  A 64-byte header in front
// of every array says what it is; pageable memory takes over when page-locked memory cannot be had.
namespace {
struct ResHeader {
    uint64_t kind;   // 1: page-locked block of the pool, 2: malloc
    uint64_t bytes;  // of the whole block
    uint64_t pad[6];
};
static_assert(sizeof(ResHeader) == 64, "arrays stay 64-byte aligned");
struct ResPool {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> idle;
    size_t idle_bytes = 0;
} g_res_pool;

void *res_alloc(size_t bytes)
{
    size_t want = 64 + std::max<size_t>(bytes, 1);
    want = (want + 4095) / 4096 * 4096;
    void *blk = nullptr;
    size_t got = 0;
    {
        std::lock_guard<std::mutex> g(g_res_pool.mu);
        size_t best = g_res_pool.idle.size();
        for (size_t i = 0; i < g_res_pool.idle.size(); i++) {
            const size_t b = g_res_pool.idle[i].second;
            if (b >= want && b <= 2 * want + (64u << 10) && (best == g_res_pool.idle.size() || b < g_res_pool.idle[best].second)) best = i;
        }
        if (best < g_res_pool.idle.size()) {
            blk = g_res_pool.idle[best].first;
            got = g_res_pool.idle[best].second;
            g_res_pool.idle_bytes -= got;
            g_res_pool.idle.erase(g_res_pool.idle.begin() + (long)best);
        }
    }
    uint64_t kind = 1;
    if (!blk) {
        got = want;
        if (hipHostMalloc(&blk, got) != hipSuccess) {
            (void)hipGetLastError();
            blk = malloc(got);
            kind = 2;
            if (!blk) return nullptr;
        }
    }
    ResHeader *h = static_cast<ResHeader *>(blk);
    h->kind = kind;
    h->bytes = got;
    return static_cast<char *>(blk) + 64;
}

void res_free(void *p)
{
    if (!p) return;
    ResHeader *h = reinterpret_cast<ResHeader *>(static_cast<char *>(p) - 64);
    if (h->kind == 2) { free(h); return; }
    std::lock_guard<std::mutex> g(g_res_pool.mu);
    if (g_res_pool.idle.size() >= 64 || g_res_pool.idle_bytes + h->bytes > (512u << 20)) {
        (void)hipHostFree(h);
        return;
    }
    g_res_pool.idle_bytes += h->bytes;
    g_res_pool.idle.emplace_back(h, (size_t)h->bytes);
}
}  // namespace

void mc_bfs_result_free(mc_bfs_result *r)
{
    if (!r) return;
    res_free(r->hi);  // (one block holds all five arrays: mc_bfs_batch)
    memset(r, 0, sizeof *r);
}

namespace {

int bfs_alloc(mc_ctx *c, BfsState &S, uint64_t dcap)
{
    S.dcap = dcap;
    uint64_t vcap = 1024;
    while (vcap < 4 * dcap) vcap <<= 1;
    S.bmask = vcap / 2 - 1;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.hi), dcap * 8));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.lo), dcap * 8));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.dist), dcap * 4));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.cov), dcap * 2));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.flags), dcap * 4));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.vis), vcap * 8));
    HIPCHK(c, hipMemsetAsync(S.vis, 0xFF, vcap * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(S.flags, 0, dcap * 4, c->stream));
    return MC_OK;
}

struct PairSource { const int64_t *keys; const int16_t *counts; const uint32_t *hints; uint64_t n; };

// Builds the solid table for n entries with count >= min_cov, taken from the counting table or (pairs != nullptr)
// from arrays of (key, count, hint).
int solid_build(mc_ctx *c, uint64_t n, int min_cov, double *ms, const PairSource *pairs, bool from_list = false)
{
    c->n_solid = n;
    c->st.solid_kmers = n;
    uint64_t factor = 4;  // slots per solid key (load factor <= 1/4)
    if (const char *e = getenv("MC_SOLID_FACTOR")) factor = std::max<uint64_t>(2, strtoull(e, nullptr, 10));
    uint32_t lg = c->sb;
    while (lg < 34 && (1ull << lg) < factor * n) lg++;
    if (lg < SOLID_SB + 1) lg = SOLID_SB + 1;
    if (!c->solid || lg != c->solid_lg) {
        if (c->solid) { (void)hipFree(c->solid); c->solid = nullptr; }
        HIPCHK(c, dev_malloc(c, reinterpret_cast<void **>(&c->solid), sizeof(Slot) << lg));
        c->solid_lg = lg;
    }
    // counting table organised by minimizer bins, or entries from outside: the solid entries are partitioned by
    // their own hash first
    const bool partitioned = true;
    mc_ctx::Pipe &P = c->pipe;
    const uint32_t q = lg - SOLID_SB, sb1 = std::min<uint32_t>(q, 9), sb2 = q - sb1;
    uint64_t scap1 = 0, scap2 = 0;
    int rc;
    if (partitioned) {
        const double m1 = (double)n / (double)(1ull << sb1) / (double)PT_SEGMENTS;
        const double m2 = (double)n / (double)(1ull << q);
        scap1 = (uint64_t)(m1 * 1.25 + 10.0 * std::sqrt(m1) + 64.0);
        scap1 += 64;  // rows of entries are dealt to the segments: one of them may bring a whole row more
        scap2 = (uint64_t)(m2 * 1.15 + 10.0 * std::sqrt(m2) + 64.0);
        const uint64_t need1 = (uint64_t)(1ull << sb1) * PT_SEGMENTS * scap1, need2 = sb2 ? (uint64_t)(1ull << q) * scap2 : 0;
        // The solid list sits in a_recs: the level-1 buckets then go to b_recs and the leaves, once the list has been
        // read, to a_recs -- which must not be reallocated for it (and a one-level build would have to read and
        // write a_recs at once): otherwise the table is swept as usual.
        if (from_list && (!sb2 || need2 > P.a_recs_cap)) from_list = false;  // (a_hints holds nothing of the list: it may grow, or come to be here -- compact runs do without it)
        uint4 **l1_recs = from_list ? &P.b_recs : &P.a_recs, **l2_recs = from_list ? &P.a_recs : &P.b_recs;
        uint32_t **l1_bins = from_list ? &P.b_hints : &P.a_hints, **l2_bins = from_list ? &P.a_hints : &P.b_hints;
        uint64_t *l1_recs_cap = from_list ? &P.b_recs_cap : &P.a_recs_cap, *l2_recs_cap = from_list ? &P.a_recs_cap : &P.b_recs_cap;
        uint64_t *l1_bins_cap = from_list ? &P.b_hints_cap : &P.a_hints_cap, *l2_bins_cap = from_list ? &P.a_hints_cap : &P.b_hints_cap;
        rc = ensure_buf(c, l1_recs, l1_recs_cap, need1);
        if (!rc) rc = ensure_buf(c, l1_bins, l1_bins_cap, need1);
        if (!rc) rc = ensure_buf(c, &P.seg_counts1, &P.segs1_cap, (uint64_t)(1ull << sb1) * PT_SEGMENTS);
        if (!rc && sb2) rc = ensure_buf(c, l2_recs, l2_recs_cap, need2);
        if (!rc && sb2) rc = ensure_buf(c, l2_bins, l2_bins_cap, need2);
        if (!rc && sb2) rc = ensure_buf(c, &P.solid_cursors, &P.solid_cursors_cap, 1ull << q);
        if (rc) return rc;
        if (!P.flags) {  // four flags and, behind them, the spill counter: cleared by one fill, read by one copy
            HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&P.flags), 4 * sizeof(uint32_t) + sizeof(unsigned long long)));
            P.spill_count = reinterpret_cast<unsigned long long *>(P.flags + 4);
        }
        HIPCHK(c, hipMemsetAsync(P.flags, 0, 4 * sizeof(uint32_t) + sizeof(unsigned long long), c->stream));
    }
    const bool list = partitioned && from_list && !pairs;
    uint4 *const r1 = list ? P.b_recs : P.a_recs, *const r2 = list ? P.a_recs : P.b_recs;
    uint32_t *const h1 = list ? P.b_hints : P.a_hints, *const h2 = list ? P.a_hints : P.b_hints;
    c->solid_list_fresh = false;  // (read below, then overwritten by the leaves)
    if (list) c->st.solid_list_builds++;
    rc = timed(c, ms, [&] {
        if (partitioned) {
            uint32_t *leaf_counts = P.solid_cursors;
            const SkSpill none{nullptr, P.spill_count, 0, P.flags};  // no spill list: an overflow is reported
            if (pairs)
                hipLaunchKernelGGL(k_solid_emit_pairs, dim3(PT_SEGMENTS), dim3(PT_THREADS), 0, c->stream, pairs->keys, pairs->counts,
                                   pairs->hints, pairs->n, min_cov, 1u << sb1, P.seg_counts1, scap1, r1, h1, none, c->d_ctr + 1);
            else if (list)
                hipLaunchKernelGGL(k_solid_emit_list, dim3(PT_SEGMENTS), dim3(PT_THREADS), 0, c->stream, P.a_recs, P.emit_counts,
                                   c->solid_list_segs, c->solid_list_segcap, 1u << sb1, P.seg_counts1, scap1, r1, h1, none);
            else
                hipLaunchKernelGGL(k_solid_emit, dim3(PT_SEGMENTS), dim3(PT_THREADS), 0, c->stream, c->slots, c->n_slots(), min_cov, 1u << sb1,
                                   P.seg_counts1, scap1, r1, h1, none);
            if (sb2)
                hipLaunchKernelGGL(k_sk2_scatter, dim3(1u << sb1), dim3(PT_THREADS), 0, c->stream, r1, h1, scap1,
                                   P.seg_counts1, 1u << sb1, 1u << sb1, 1u << sb2, leaf_counts, scap2, r2, h2, none, (uint32_t)PT_SEGMENTS, 1);
            hipLaunchKernelGGL(k_solid_from_leaves, dim3((unsigned)std::min<uint64_t>(1ull << q, 256 * 2 * 8)), dim3(512), 0, c->stream,
                               sb2 ? r2 : r1, sb2 ? leaf_counts : P.seg_counts1, sb2 ? scap2 : scap1,
                               sb2 ? 1u : (uint32_t)PT_SEGMENTS, c->solid_view(), lg);
        }
    });
    if (rc) return rc;
    uint32_t fatal = 0;
    HIPCHK(c, hipMemcpy(&fatal, c->d_fatal, sizeof fatal, hipMemcpyDeviceToHost));
    if (fatal) return fail(c, MC_EOVERFLOW, "a region of the solid k-mer table filled up (hash skew)");
    if (partitioned) {
        uint32_t lost = 0;
        HIPCHK(c, hipMemcpy(&lost, P.flags, sizeof lost, hipMemcpyDeviceToHost));
        if (lost) return fail(c, MC_EOVERFLOW, "internal: a bucket of the solid-table build overflowed (hash skew)");
    }
    c->solid_cov = min_cov;
    return MC_OK;
}

// Builds (or reuses) the solid table for this threshold.
int ensure_solid(mc_ctx *c, int min_cov, double *ms)
{
    if (c->d_shards || (c->bfs_direct && !c->solid_external)) {  // the walk looks its k-mers up in the counting table(s): nothing to build
        c->solid_is_table = true;
        c->solid_cov = min_cov;
        return MC_OK;
    }
    c->solid_is_table = false;
    if (c->solid_cov == min_cov && c->solid) return MC_OK;
    if (c->dup.merged && c->dup.n_tw)  // (a copy is built by key: the keys that sit in several regions meet in hash-prefix regions first)
        if (int brc = by_key_ready(c)) return brc;
    if (c->solid_external)
        return fail(c, MC_ESTATE, "this context's solid table came from mc_solid_from_pairs_dev at coverage %d; it serves that threshold only",
                    c->solid_external_cov);
    c->solid_cov = -1; c->solid_external = false;
    unsigned long long *cursor = c->d_ctr + 2;
    int rc;
    bool from_list = false;
    if (c->solid_tracked && c->cov_hint == min_cov && min_cov > 0 && !c->virgin) {
        cursor = c->d_ctr + 6;  // kept up to date by k_p3_merge
        from_list = c->solid_list_fresh && c->mm_k;  // ... which also listed those keys
    } else {
        HIPCHK(c, hipMemsetAsync(cursor, 0, sizeof(unsigned long long), c->stream));
        rc = timed(c, ms, [&] {
            hipLaunchKernelGGL(k_export, dim3(grid_for(c->n_slots(), 256)), dim3(256), 0, c->stream, c->slots,
                               c->n_slots(), min_cov, (int64_t *)nullptr, (int16_t *)nullptr, (uint32_t *)nullptr, (uint64_t)0, cursor);
        });
        if (rc) return rc;
        c->st.solid_sweeps++;
    }
    unsigned long long n = 0;
    HIPCHK(c, hipMemcpy(&n, cursor, sizeof n, hipMemcpyDeviceToHost));
    return solid_build(c, n, min_cov, ms, nullptr, from_list);
}

// What mc_bfs_batch used to do with 4 fills per job before the launch and 5 + 2 copies per job behind it (each a
// launch of its own, 10-25 us apart: 0.7 ms of a 9 ms walk on two jobs) is one kernel either side.
// k_bfs_reset: blockIdx.y = job.  full: index emptied, flags and control block cleared (a fresh run); always: the mailbox.
__global__ void __launch_bounds__(256) k_bfs_reset(const BfsState *__restrict__ states, int full)
{
    const BfsState S = states[blockIdx.y];
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    if (full) {
        uint4 *vis = reinterpret_cast<uint4 *>(S.vis);  // (bmask + 1 buckets of 16 bytes)
        for (uint64_t i = t; i <= S.bmask; i += stride) vis[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (uint64_t i = t; i < S.dcap; i += stride) S.flags[i] = 0;
        uint32_t *ctl = reinterpret_cast<uint32_t *>(S.ctl);
        for (uint64_t i = t; i < sizeof(BfsCtl) / 4; i += stride) ctl[i] = 0;
    }
    if (S.box) {
        uint32_t *box = reinterpret_cast<uint32_t *>(S.box);
        for (uint64_t i = t; i < sizeof(ScoutBox) / 4; i += stride) box[i] = 0;
    }
}

// the header of a job's packed results
struct BfsPackHdr {
    BfsCtl ctl;
    ScoutBox box;
    unsigned long long offset;  // of the job's arrays in the data block: hi[n] lo[n] dist[n] cov[n] last[n], each padded to 8 bytes
    unsigned long long levels;  // max distance
};
__host__ __device__ inline uint64_t bfs_pack_bytes(uint64_t n) { return 16 * n + ((4 * n + 7) & ~7ull) + ((2 * n + 7) & ~7ull) + ((n + 7) & ~7ull); }

// k_bfs_pack: blockIdx.y = job.  Control block and mailbox into the job's header, its arrays (the first ctl.n entries)
// into the data block behind the arrays of the jobs before it; `last` from bit 0 of the flags; the largest distance.
__global__ void __launch_bounds__(256) k_bfs_pack(const BfsState *__restrict__ states, uint32_t n_jobs, BfsPackHdr *hdr, char *data, uint64_t data_cap)
{
    (void)n_jobs;
    const uint32_t j = blockIdx.y;
    const BfsState S = states[j];
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    unsigned long long off = 0;
    for (uint32_t i = 0; i < j; i++) off += bfs_pack_bytes(ctl_ld(&states[i].ctl->n));
    const uint64_t n = ctl_ld(&S.ctl->n);
    if (blockIdx.x == 0) {
        uint32_t *h = reinterpret_cast<uint32_t *>(&hdr[j]);
        const uint32_t *ctl = reinterpret_cast<const uint32_t *>(S.ctl), *box = reinterpret_cast<const uint32_t *>(S.box);
        for (uint32_t i = threadIdx.x; i < sizeof(BfsCtl) / 4; i += 256) h[i] = __hip_atomic_load(&ctl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t i = threadIdx.x; i < sizeof(ScoutBox) / 4; i += 256)
            h[sizeof(BfsCtl) / 4 + i] = box ? __hip_atomic_load(&box[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (threadIdx.x == 0) hdr[j].offset = off;
    }
    if (off + bfs_pack_bytes(n) > data_cap) return;  // (the host sizes the block from the jobs' capacities: cannot happen)
    uint64_t *hi = reinterpret_cast<uint64_t *>(data + off), *lo = hi + n;
    int32_t *dist = reinterpret_cast<int32_t *>(lo + n);
    int16_t *cov = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(dist) + ((4 * n + 7) & ~7ull));
    uint8_t *last = reinterpret_cast<uint8_t *>(reinterpret_cast<char *>(cov) + ((2 * n + 7) & ~7ull));
    unsigned long long mx = 0;
    for (uint64_t i = t; i < n; i += stride) {
        hi[i] = S.hi[i];
        lo[i] = S.lo[i];
        const int32_t d = S.dist[i];
        dist[i] = d;
        cov[i] = S.cov[i];
        last[i] = (uint8_t)(S.flags[i] & 1u);
        mx = max(mx, (unsigned long long)(uint32_t)d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned long long)__shfl_down(mx, o));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&hdr[j].levels, mx);
}

// MC_BFS_SELFCHECK: the finished walks of a batch against the invariants of k_bfs_check; *report names what was found
int bfs_selfcheck(mc_ctx *c, uint32_t n_jobs, const std::vector<BfsCtl> &ctl, int min_cov, int64_t max_kmers, int64_t max_radius,
                  std::string *report)
{
    const SolidView t = c->solid_view();
    BfsCheck *d_out = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(BfsCheck)));
    for (uint32_t j = 0; j < n_jobs; j++) {
        const BfsState &S = c->bfs_pool[j]->S;
        const uint64_t n = ctl[j].n;
        if (n == 0) continue;
        uint64_t cap = 1024;
        while (cap < 4 * n) cap <<= 1;
        uint32_t *d_set = nullptr;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_set), cap * 4));
        HIPCHK(c, hipMemsetAsync(d_set, 0xFF, cap * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(d_out, 0, sizeof(BfsCheck), c->stream));
        const unsigned grid = (unsigned)std::min<uint64_t>(grid_for(n, 256), 4096);
        hipLaunchKernelGGL(k_bfs_check_index, dim3(grid), dim3(256), 0, c->stream, S, n, d_set, (uint32_t)(cap - 1), d_out);
        switch (c->cfg.key_mode) {
        case MC_KEY_PACKED:
            hipLaunchKernelGGL(k_bfs_check<KEY_PACKED>, dim3(grid), dim3(256), 0, c->stream, S, n, t, c->cfg.k, min_cov, (long long)max_kmers,
                               (long long)max_radius, d_set, (uint32_t)(cap - 1), d_out);
            break;
        case MC_KEY_POLY:
            hipLaunchKernelGGL(k_bfs_check<KEY_POLY>, dim3(grid), dim3(256), 0, c->stream, S, n, t, c->cfg.k, min_cov, (long long)max_kmers,
                               (long long)max_radius, d_set, (uint32_t)(cap - 1), d_out);
            break;
        default:
            hipLaunchKernelGGL(k_bfs_check<KEY_FNV1A>, dim3(grid), dim3(256), 0, c->stream, S, n, t, c->cfg.k, min_cov, (long long)max_kmers,
                               (long long)max_radius, d_set, (uint32_t)(cap - 1), d_out);
        }
        HIPCHK(c, hipGetLastError());
        BfsCheck h;
        HIPCHK(c, hipMemcpyAsync(&h, d_out, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(d_set);
        if (h.dup | h.cov | h.order | h.orphan | h.open) {
            char buf[256];
            snprintf(buf, sizeof buf, "job %u (dir %d, n %llu): %llu duplicate, %llu wrong coverage, %llu out of order, %llu without parent, %llu solid neighbours left out;",
                     j, S.dir, (unsigned long long)n, h.dup, h.cov, h.order, h.orphan, h.open);
            *report += buf;
            static const char *kind[] = {"?", "dup", "cov", "order", "orphan", "open"};
            for (uint32_t i = 0; i < std::min<uint32_t>(h.n_first, 16); i++) {
                snprintf(buf, sizeof buf, " [%s entry %u other %u dist %u]", kind[std::min<uint32_t>(h.first[i][0], 5)], h.first[i][1], h.first[i][2], h.first[i][3]);
                *report += buf;
            }
        }
    }
    (void)hipFree(d_out);
    return MC_OK;
}

// the rounds' trace of every job (MC_BFS_TRACE builds; nothing otherwise), oldest record first, as text
void bfs_trace_dump(mc_ctx *c, uint32_t n_jobs, const std::vector<BfsCtl> &ctl, const char *path, const std::string &why)
{
    FILE *f = fopen(path, "a");
    if (!f) return;
    if (!why.empty()) fprintf(f, "# %s\n", why.c_str());
    for (uint32_t j = 0; j < n_jobs; j++) {
        const BfsState &S = c->bfs_pool[j]->S;
        fprintf(f, "# job %u dir %d n %llu level %lld status %d rounds_narrow %llu slow %llu trace records %llu\n", j, S.dir, ctl[j].n, ctl[j].level, ctl[j].status,
                ctl[j].rounds_narrow, ctl[j].rounds_slow, ctl[j].trace_n);
        if (!S.trace || !ctl[j].trace_n) continue;
        std::vector<uint32_t> ring((size_t)BFS_TRACE_RECORDS * 8);
        if (hipMemcpy(ring.data(), S.trace, ring.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) continue;
        const unsigned long long tn = ctl[j].trace_n, first = tn > BFS_TRACE_RECORDS ? tn - BFS_TRACE_RECORDS : 0;
        for (unsigned long long r = first; r < tn; r++) {
            const uint32_t *w = &ring[(size_t)(r & (BFS_TRACE_RECORDS - 1)) * 8];
            const uint32_t kind = w[0] >> 24;
            if (kind == 8)
                fprintf(f, "%llu companion seq %u F %u team %u stopped %u levels %u %u budget %u stuck %u %u iters %u\n", r, w[0] & 0xFFFFFF, w[1] & 0xFF, (w[1] >> 8) & 0xFF,
                        (w[1] >> 16) & 1, w[2], w[3], w[4], w[5] & 0xFF, (w[5] >> 8) & 0xFF, w[6]);
            else
                fprintf(f, "%llu %s round %u n %u F %u H %u %s %u force_slow %u comp %u level %u pend %u %s %u seq %u open %u resp %08x %s %u %u\n", r,
                        kind == 1 ? "fast" : kind == 2 ? "slow" : "root_bad", w[0] & 0xFFFFFF, w[1], w[2] & 0xFF, (w[2] >> 8) & 0xFF, kind == 2 ? "n_new" : "J", (w[2] >> 16) & 0xFF,
                        (w[2] >> 24) & 1, (w[2] >> 25) & 1, w[3], w[4] & 0xFFFF, kind == 2 ? "F_next" : "bad_lvl", w[4] >> 16, w[5] & 0x7FFFFFFF, w[5] >> 31, w[6],
                        kind == 2 ? "n_after" : "plen0/ppos0", kind == 2 ? w[7] : w[7] & 0xFFFF, kind == 2 ? 0u : w[7] >> 16);
        }
    }
    fclose(f);
}

void launch_bfs(mc_ctx *c, hipStream_t stream, const BfsState *d_states, uint32_t n_jobs, int min_cov, int64_t max_kmers,
                int64_t max_radius, unsigned long long max_rounds, int companions)
{
    const SolidView t = c->solid_view();
    // companions: the scouts' kernel beside the walk's, on the side stream -- behind what the main stream has enqueued so far (the
    // mailboxes' reset), and the main stream takes nothing up behind the walk before the scouts have left (they leave when the walk
    // sets `quit`, which it does on every way out)
    hipStream_t side = companions && c->pipe_stream ? c->pipe_stream : nullptr;
    if (companions && !side) companions = 0;
    if (side) {
        if (hipEventRecord(c->ev_piece[0], stream) != hipSuccess || hipStreamWaitEvent(side, c->ev_piece[0], 0) != hipSuccess) { side = nullptr; companions = 0; }
    }
    // (SH: the walk over several ranks' tables, every look-up through its key's owner; the one-table kernel carries none of that)
#define BFS_LAUNCH(MODE)                                                                                                          \
    do {                                                                                                                          \
        if (t.n_shards > 1) {                                                                                                     \
            if (side) hipLaunchKernelGGL((k_bfs_scout<MODE, true>), dim3(n_jobs), dim3(BFS_THREADS), 0, side, d_states, t, c->cfg.k, min_cov); \
            hipLaunchKernelGGL((k_bfs<MODE, true>), dim3(n_jobs), dim3(BFS_THREADS), 0, stream, d_states, t, c->cfg.k, min_cov,    \
                               (long long)max_kmers, (long long)max_radius, max_rounds, companions);                              \
        } else {                                                                                                                  \
            if (side) hipLaunchKernelGGL((k_bfs_scout<MODE, false>), dim3(n_jobs), dim3(BFS_THREADS), 0, side, d_states, t, c->cfg.k, min_cov); \
            hipLaunchKernelGGL((k_bfs<MODE, false>), dim3(n_jobs), dim3(BFS_THREADS), 0, stream, d_states, t, c->cfg.k, min_cov,   \
                               (long long)max_kmers, (long long)max_radius, max_rounds, companions);                              \
        }                                                                                                                         \
    } while (0)
    switch (c->cfg.key_mode) {
    case MC_KEY_PACKED: BFS_LAUNCH(KEY_PACKED); break;
    case MC_KEY_POLY: BFS_LAUNCH(KEY_POLY); break;
    default: BFS_LAUNCH(KEY_FNV1A);
    }
#undef BFS_LAUNCH
    if (side) {
        (void)hipEventRecord(c->ev_piece[1], side);
        (void)hipStreamWaitEvent(stream, c->ev_piece[1], 0);
    }
}


// After a walk over a table of hash keys in minimizer bins: every look-up of it that came back "absent" is asked again by key
// (dup_check.h, "Look-ups of k-mers that were never counted").  *redo: a key turned up that a string nobody counted had asked for;
// it has been given a slot in that string's bin, the table has been joined again, and the walks must be repeated.
int phantom_verify(mc_ctx *c, uint32_t n_jobs, std::vector<std::unique_ptr<BfsJobBuffers>> &B, const std::vector<BfsCtl> &ctl, bool *redo)
{
    *redo = false;
    mc_ctx::Dup &D = c->dup;
    if (!hash_bins(c) || !c->solid_is_table || c->d_shards || !dup_check_on() || !D.checked) return MC_OK;
    const bool by_streams = D.l2_valid;  // (else: the join's streams are gone -- mc_trim, a pipeline run since --: one sweep of the table instead)
    uint64_t cap = 64;
    for (uint32_t j = 0; j < n_jobs; j++) cap += ctl[j].n * (B[j]->S.dir == 0 ? 8 : 4) + B[j]->S.n_seeds;
    if (cap >= (1ull << 32)) return fail(c, MC_EOVERFLOW, "mc_bfs: %llu look-ups to check by key", (unsigned long long)cap);
    const uint64_t hit_cap = 4096;
    const uint64_t words = 3 * cap + (cap + 1) / 2 + hit_cap / 2 + 8;  // key, hi, lo; order (32-bit); hits (32-bit); two counters
    if (D.pq_cap < words) {
        if (D.pq_mem) (void)hipFree(D.pq_mem);
        D.pq_mem = nullptr; D.pq_cap = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&D.pq_mem), words * 2 * 8));
        D.pq_cap = words * 2;
    }
    uint64_t set_slots = 1024;  // (the sweep form: the queries as a set, 12 bytes an entry in the same block)
    while (set_slots < 2 * cap) set_slots <<= 1;
    const uint32_t G = by_streams ? DUP_B1 << D.l2.f2_lg : (uint32_t)std::min<uint64_t>(3 * set_slots, 0xFFFFFFF0ull);
    if (!by_streams && 3 * set_slots > 0xFFFFFFF0ull) return fail(c, MC_EOVERFLOW, "mc_bfs: %llu look-ups to check by key", (unsigned long long)cap);
    if (D.pq_groups_cap < G) {
        if (D.pq_groups) (void)hipFree(D.pq_groups);
        D.pq_groups = nullptr; D.pq_groups_cap = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&D.pq_groups), (uint64_t)G * sizeof(uint32_t)));
        D.pq_groups_cap = G;
    }
    unsigned long long *ctr = D.pq_mem;  // [0] queries, [1] hits
    PhantomQ q{D.pq_mem + 8, D.pq_mem + 8 + cap, D.pq_mem + 8 + 2 * cap, ctr, cap};
    uint32_t *next = reinterpret_cast<uint32_t *>(D.pq_mem + 8 + 3 * cap);
    uint32_t *hits = next + 2 * ((cap + 1) / 2);
    uint32_t *head = D.pq_groups;
    HIPCHK(c, hipMemsetAsync(ctr, 0, 2 * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(head, 0xFF, (by_streams ? (uint64_t)G : 2 * set_slots) * sizeof(uint32_t), c->stream));  // (list heads, or the set's keys: ~0 = free)
    const SolidView t = c->solid_view();
    const int k = c->cfg.k;
    for (uint32_t j = 0; j < n_jobs; j++) {
        const BfsState &S = B[j]->S;
        const int nb = S.dir == 0 ? 8 : 4;
#define PQ_LAUNCH(MODE)                                                                                                                               \
        do {                                                                                                                                          \
            if (ctl[j].n) hipLaunchKernelGGL(k_phantom_queries<MODE>, dim3(grid_for(ctl[j].n * nb, 256)), dim3(256), 0, c->stream, S.hi, S.lo, (uint64_t)ctl[j].n, nb, S.dir, k, t, q); \
            if (S.n_seeds) hipLaunchKernelGGL(k_phantom_queries<MODE>, dim3(grid_for(S.n_seeds, 256)), dim3(256), 0, c->stream, S.seed_hi, S.seed_lo, (uint64_t)S.n_seeds, 0, S.dir, k, t, q); \
        } while (0)
        if (c->cfg.key_mode == MC_KEY_POLY) PQ_LAUNCH(KEY_POLY); else PQ_LAUNCH(KEY_FNV1A);
#undef PQ_LAUNCH
    }
    if (by_streams) {
        hipLaunchKernelGGL(k_pq_link, dim3(256), dim3(256), 0, c->stream, q.key, ctr, cap, D.l2.f2_lg, head, next);
        hipLaunchKernelGGL(k_pq_match, dim3(std::min<uint32_t>(G, 256u * 16u)), dim3(256), 0, c->stream, D.l2, head, next, q.key, hits, ctr + 1, hit_cap);
    } else {
        unsigned long long *qk = reinterpret_cast<unsigned long long *>(D.pq_groups);
        uint32_t *qidx = D.pq_groups + 2 * set_slots;
        hipLaunchKernelGGL(k_pq_set_build, dim3(256), dim3(256), 0, c->stream, q.key, ctr, cap, qk, qidx, set_slots - 1);
        hipLaunchKernelGGL(k_pq_sweep, dim3(grid_for(c->n_slots(), 256, 256 * 16)), dim3(256), 0, c->stream, c->slots, c->n_slots(), qk, qidx, set_slots - 1, hits, ctr + 1, hit_cap);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_scratch + 20, ctr, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint64_t n_hits = c->h_scratch[21];
    if (n_hits == 0) return MC_OK;
    // ---- a string nobody counted asked for a key the table holds: the key gets a slot in that string's bin and the join runs again
    int rc = dup_unmerge(c);
    if (rc) return rc;
    const uint64_t nh = std::min<uint64_t>(n_hits, hit_cap);
    hipLaunchKernelGGL(k_alias_insert, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, c->stream, hits, nh, q, c->view(), k);
    HIPCHK(c, hipGetLastError());
    rc = drain_parked(c);  // (no room in that bin's chain: the table gives up its bins, and every look-up is by key from then on)
    if (rc) return rc;
    D.checked = false; D.l1_armed = false; D.l2_valid = false;
    c->solid_tracked = false;  // (slots the merge kernel did not count)
    rc = ensure_dups(c);
    if (rc) return rc;
    *redo = true;
    return MC_OK;
}

}  // namespace

int mc_solid_from_pairs_dev(mc_ctx *c, const int64_t *d_keys, const int16_t *d_counts, const uint32_t *d_hints, uint64_t n,
                            int min_cov, uint64_t *n_solid)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (n_solid) *n_solid = 0;
    if ((!d_keys || !d_counts) && n) return fail(c, MC_EINVAL, "mc_solid_from_pairs_dev: null pointer");
    if (min_cov < 0) return fail(c, MC_EINVAL, "mc_solid_from_pairs_dev: negative coverage threshold");
    if (!c->virgin) return fail(c, MC_ESTATE, "mc_solid_from_pairs_dev: the context holds counts; call mc_clear first");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->solid_cov = -1;
    c->solid_external = false;
    c->solid_is_table = false;
    unsigned long long *cursor = c->d_ctr + 2;
    HIPCHK(c, hipMemsetAsync(cursor, 0, sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ctr + 1, 0, sizeof(unsigned long long), c->stream));  // the out-of-band count of EMPTY_KEY
    if (n) hipLaunchKernelGGL(k_count_pairs, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, d_counts, d_keys, n, min_cov, cursor);
    HIPCHK(c, hipGetLastError());
    unsigned long long m = 0;
    HIPCHK(c, hipMemcpyAsync(&m, cursor, sizeof m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const PairSource src{d_keys, d_counts, d_hints, n};
    double ms = 0;
    int rc = solid_build(c, m, min_cov, &ms, &src);
    if (rc) return rc;
    c->pending_solid_ms = ms;  // reported with the next BFS
    c->solid_external = true;
    c->solid_external_cov = min_cov;
    c->finalized = true;
    if (n_solid) *n_solid = m;
    return MC_OK;
}

int mc_bfs_batch(mc_ctx *c, const mc_bfs_job *jobs, uint32_t n_jobs, int min_cov, int64_t max_kmers,
                 int64_t max_radius, mc_bfs_result *out)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!out || !jobs || n_jobs == 0) return fail(c, MC_EINVAL, "mc_bfs: null argument");
    memset(out, 0, sizeof(mc_bfs_result) * n_jobs);
    if (!c->finalized) return fail(c, MC_ESTATE, "mc_bfs: call mc_finalize_counts first");
    if (c->shards_dropped)
        return fail(c, MC_ESTATE, "mc_bfs: the table changed after mc_shard_attach, which dropped the attachment: export and attach the ranks' tables again (or mc_shard_detach to walk this table alone)");
    if (max_kmers < 0 && max_radius < 0)
        return fail(c, MC_EINVAL, "At least one of --maxkmers and --maxradius parameters should be set");
    if (min_cov < 0)
        return fail(c, MC_EINVAL, "mc_bfs: negative coverage threshold (absent k-mers read as -1 and would pass)");
    if (n_jobs > 4096) return fail(c, MC_EINVAL, "mc_bfs: too many jobs in one batch");
    for (uint32_t j = 0; j < n_jobs; j++) {
        if (jobs[j].dir < -1 || jobs[j].dir > 1) return fail(c, MC_EINVAL, "mc_bfs: dir must be -1, 0 or +1");
        if (jobs[j].n_seeds && !jobs[j].seed_lo) return fail(c, MC_EINVAL, "mc_bfs: seed_lo is null");
        if (c->cfg.k > 32 && jobs[j].n_seeds && !jobs[j].seed_hi)
            return fail(c, MC_EINVAL, "mc_bfs: seed_hi is null with k > 32");
        if (max_kmers >= (int64_t)0x3FFFF000ll || jobs[j].n_seeds >= 0x3FFFF000ull)
            return fail(c, MC_EINVAL, "mc_bfs: more than 2^30 vertices requested");
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    double total_ms = c->pending_solid_ms;
    c->pending_solid_ms = 0;
    {
        int rc = ensure_solid(c, min_cov, &total_ms);
        // (hash keys in minimizer bins: the join of mc_finalize_counts, should anything have skipped it)
        if (!rc && hash_bins(c) && c->solid_is_table && !c->d_shards && dup_check_on() && !c->dup.checked) {
            c->dup.l1_armed = false;
            rc = ensure_dups(c);
            if (!rc) rc = ensure_solid(c, min_cov, &total_ms);
        }
        if (rc) return rc;
    }

    while (c->bfs_pool.size() < n_jobs) c->bfs_pool.emplace_back(new BfsJobBuffers);
    auto &B = c->bfs_pool;
    // ---- one upload: the job states, then every job's seeds (lo, hi)
    uint64_t stage_bytes = ((uint64_t)n_jobs * sizeof(BfsState) + 63) & ~63ull, pack_bytes = 0;
    std::vector<uint64_t> seed_at(n_jobs);
    for (uint32_t j = 0; j < n_jobs; j++) {
        seed_at[j] = stage_bytes;
        stage_bytes += 2 * std::max<uint64_t>(jobs[j].n_seeds, 1) * 8;
    }
    if (c->bfs_stage_cap < stage_bytes) {
        if (c->h_bfs_stage) (void)hipHostFree(c->h_bfs_stage);
        if (c->d_bfs_stage) (void)hipFree(c->d_bfs_stage);
        c->h_bfs_stage = c->d_bfs_stage = nullptr;
        c->bfs_stage_cap = 0;
        const uint64_t cap = std::max<uint64_t>(stage_bytes * 2, 1u << 16);
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_bfs_stage), cap, hipHostMallocDefault));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_bfs_stage), cap));
        c->bfs_stage_cap = cap;
    }
    for (uint32_t j = 0; j < n_jobs; j++) {
        BfsJobBuffers &J = *B[j];
        BfsState &S = J.S;
        const uint64_t ns = jobs[j].n_seeds;
        uint64_t *h_lo = reinterpret_cast<uint64_t *>(c->h_bfs_stage + seed_at[j]), *h_hi = h_lo + std::max<uint64_t>(ns, 1);
        if (ns) memcpy(h_lo, jobs[j].seed_lo, ns * 8);
        if (ns && jobs[j].seed_hi) memcpy(h_hi, jobs[j].seed_hi, ns * 8);
        const uint64_t dcap = max_kmers >= 0 ? std::max<uint64_t>((uint64_t)max_kmers, ns) + 2 * BFS_THREADS
                                             : std::max<uint64_t>(1ull << 20, ns + 2 * BFS_THREADS);
        if (S.dcap < dcap || S.dcap > 4 * dcap || !S.hi) {
            J.free_arrays();
            int rc = bfs_alloc(c, S, dcap);
            if (rc) return rc;
        }  // (reused arrays: k_bfs_reset empties the index and clears the flags)
        if (!S.ctl) HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.ctl), sizeof(BfsCtl)));
        if (!S.path) HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.path), (size_t)SCOUT_MAX_F * PATH_WORDS * 8));
        if (!S.box) HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.box), sizeof(ScoutBox)));
#ifdef MC_BFS_TRACE
        if (!S.trace) HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.trace), (size_t)BFS_TRACE_RECORDS * 32));
#endif
        uint64_t *d_lo = reinterpret_cast<uint64_t *>(c->d_bfs_stage + seed_at[j]);
        S.seed_hi = jobs[j].seed_hi ? d_lo + std::max<uint64_t>(ns, 1) : nullptr;
        S.seed_lo = d_lo;
        S.n_seeds = ns;
        S.dir = jobs[j].dir;
    }
    BfsState *const d_states = reinterpret_cast<BfsState *>(c->d_bfs_stage);
    BfsState *const h_states = reinterpret_cast<BfsState *>(c->h_bfs_stage);
    std::vector<BfsCtl> ctl(n_jobs);
    const unsigned long long max_rounds = 1ull << 17;  // bounds one launch; unfinished jobs are relaunched
    std::vector<std::array<unsigned long long, 4>> box_e(n_jobs, std::array<unsigned long long, 4>{0, 0, 0, 0});
    std::vector<unsigned long long> box_iters(n_jobs, 0), box_hops(n_jobs, 0), box_levels(n_jobs, 0), box_calls(n_jobs, 0), box_nf(n_jobs, 0), box_m0(n_jobs, 0);
    const uint64_t hdr_bytes = (uint64_t)n_jobs * sizeof(BfsPackHdr);
    BfsPackHdr *h_hdr = nullptr;
    for (int pass = 0;; pass++) {  // (a pass more where the check of the walk's look-ups by key gave the table a slot: phantom_verify)
    for (int launch = 0;; launch++) {
        for (uint32_t j = 0; j < n_jobs; j++) h_states[j] = B[j]->S;
        // (first launch: states and seeds; later ones, after a job's arrays grew: the states)
        HIPCHK(c, hipMemcpyAsync(c->d_bfs_stage, c->h_bfs_stage, launch == 0 ? stage_bytes : (uint64_t)n_jobs * sizeof(BfsState), hipMemcpyHostToDevice, c->stream));
        // Few jobs: each gets a second workgroup that scouts ahead while the first verifies (bfs_device.h ScoutBox).  The
        // pair must be on the chip together to gain anything (it is correct either way), so not for large batches.
        static const bool no_comp = getenv("MC_BFS_COMPANION") && !strcmp(getenv("MC_BFS_COMPANION"), "0");
        const int companions = !no_comp && n_jobs <= 64 && c->solid_view().reads != nullptr ? 1 : 0;
        hipLaunchKernelGGL(k_bfs_reset, dim3(n_jobs <= 8 ? 64 : 8, n_jobs), dim3(256), 0, c->stream, d_states, launch == 0 ? 1 : 0);
        HIPCHK(c, hipGetLastError());
        pack_bytes = 0;
        for (uint32_t j = 0; j < n_jobs; j++) pack_bytes += bfs_pack_bytes(B[j]->S.dcap);
        if (c->bfs_pack_cap < hdr_bytes + pack_bytes) {
            if (c->d_bfs_pack) (void)hipFree(c->d_bfs_pack);
            c->d_bfs_pack = nullptr;
            c->bfs_pack_cap = 0;
            HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_bfs_pack), hdr_bytes + pack_bytes));
            c->bfs_pack_cap = hdr_bytes + pack_bytes;
        }
        if (c->bfs_hdr_cap < hdr_bytes) {
            if (c->h_bfs_hdr) (void)hipHostFree(c->h_bfs_hdr);
            c->h_bfs_hdr = nullptr;
            c->bfs_hdr_cap = 0;
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_bfs_hdr), hdr_bytes * 2, hipHostMallocDefault));
            c->bfs_hdr_cap = hdr_bytes * 2;
        }
        // (the walk, the packing of its results and the copy of their headers are enqueued together: one wait)
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        launch_bfs(c, c->stream, d_states, n_jobs, min_cov, max_kmers, max_radius, max_rounds, companions);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        int rc;
        // ---- the results packed on the device: headers (control block, mailbox, offset, levels), then the jobs' arrays
        h_hdr = reinterpret_cast<BfsPackHdr *>(c->h_bfs_hdr);
        HIPCHK(c, hipMemsetAsync(c->d_bfs_pack, 0, hdr_bytes, c->stream));
        hipLaunchKernelGGL(k_bfs_pack, dim3(n_jobs <= 8 ? 64 : 8, n_jobs), dim3(256), 0, c->stream, d_states, n_jobs,
                           reinterpret_cast<BfsPackHdr *>(c->d_bfs_pack), c->d_bfs_pack + hdr_bytes, pack_bytes);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h_hdr, c->d_bfs_pack, hdr_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        {
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            total_ms += ms;
        }
        for (uint32_t j = 0; j < n_jobs; j++) {
            ctl[j] = h_hdr[j].ctl;
            if (!companions) continue;
            const ScoutBox &hb = h_hdr[j].box;
            box_iters[j] += hb.iters; box_e[j][0] += hb.e_stuck; box_e[j][1] += hb.e_nc0; box_e[j][2] += hb.e_budget; box_e[j][3] += hb.e_stop; box_hops[j] += hb.hops; box_levels[j] += hb.levels; box_calls[j] += hb.calls; box_nf[j] += hb.nf; box_m0[j] += hb.m0;
        }
        bool all_done = true;
        for (uint32_t j = 0; j < n_jobs; j++) {
            if (ctl[j].status == BFS_DONE) continue;
            all_done = false;
            if (ctl[j].status != BFS_NEED_GROW) continue;
            // grow distanceToKmer and its index (only reachable without --maxkmers)
            BfsState &S = B[j]->S;
            BfsState N = S;
            N.hi = N.lo = nullptr; N.dist = nullptr; N.cov = nullptr; N.flags = nullptr; N.vis = nullptr;  // (ctl and path stay)
            rc = bfs_alloc(c, N, S.dcap * 2);
            if (rc) return rc;
            const uint64_t n = ctl[j].n;
            HIPCHK(c, hipMemcpyAsync(N.hi, S.hi, n * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.lo, S.lo, n * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.dist, S.dist, n * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.cov, S.cov, n * 2, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.flags, S.flags, n * 4, hipMemcpyDeviceToDevice, c->stream));
            if (n) {
                hipLaunchKernelGGL(k_vis_rebuild, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, N, n);
                HIPCHK(c, hipGetLastError());
            }
            const int running = BFS_RUNNING;
            HIPCHK(c, hipMemcpyAsync(&S.ctl->status, &running, sizeof running, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            B[j]->free_arrays();
            S = N;
        }
        if (all_done) break;
    }
    bool redo = false;
    if (int vrc = phantom_verify(c, n_jobs, B, ctl, &redo)) return vrc;
    if (!redo) break;
    if (pass >= 8) return fail(c, MC_EOVERFLOW, "mc_bfs: the walk keeps finding keys that strings nobody counted ask for");
    }

    int ret = MC_OK;
    {   // debug: every walk checked on the device (bfs_device.h k_bfs_check); the rounds' trace of a tuning build written out
        const char *sc = getenv("MC_BFS_SELFCHECK");
        const char *dump = getenv("MC_BFS_TRACE_DUMP");
        std::string report;
        if (sc && *sc && *sc != '0') {
            int rc = bfs_selfcheck(c, n_jobs, ctl, min_cov, max_kmers, max_radius, &report);
            if (rc) return rc;
        }
        if (!report.empty() || (dump && *dump)) bfs_trace_dump(c, n_jobs, ctl, report.empty() ? dump : (dump && *dump ? dump : "/dev/stderr"), report);
        if (!report.empty()) return fail(c, MC_ECHECK, "mc_bfs: self-check failed: %s", report.c_str());
    }
    for (uint32_t j = 0; j < n_jobs; j++) {
        const uint64_t n = ctl[j].n;
        mc_bfs_result *o = &out[j];
        o->lookups = ctl[j].lookups;
        // (a scout hop: the read, then its k-mers; the members of a companion's team hop side by side)
        o->rounds = ctl[j].rounds_narrow + ctl[j].chunks_wide + 2 * (ctl[j].scout_hops + box_iters[j]);
        o->device_ms = total_ms;
        {
            static const bool stats = getenv("MC_BFS_STATS") != nullptr;
            if (stats) fprintf(stderr, "[bfs job %u] slow rounds: after a partly verified round %llu, first level mismatch %llu, nothing predicted %llu; companion runs ended: stuck %llu, no candidates %llu, budget %llu, aborted %llu\n",
                               j, ctl[j].slow_forced, ctl[j].slow_mismatch, ctl[j].slow_starved, box_e[j][0], box_e[j][1], box_e[j][2], box_e[j][3]);
            if (stats)
                fprintf(stderr, "[bfs job %u] n=%llu level=%lld rounds_narrow=%llu (slow %llu) chunks_wide=%llu scout calls=%llu hops=%llu (team hops %llu) levels=%llu notfound=%llu m0=%llu lookups=%llu\n", j,
                        ctl[j].n, ctl[j].level, ctl[j].rounds_narrow, ctl[j].rounds_slow, ctl[j].chunks_wide, ctl[j].scout_calls + box_calls[j], ctl[j].scout_hops + box_hops[j], box_iters[j],
                        ctl[j].scout_levels + box_levels[j], ctl[j].scout_nf + box_nf[j], ctl[j].scout_m0 + box_m0[j], ctl[j].lookups);
        }
#ifdef MC_BFS_TIMING
        {
            const double r_ = 0.01 / std::max(1ull, ctl[j].rounds_narrow);
            fprintf(stderr, "[bfs job %u] rounds_narrow=%llu (slow %llu) chunks_wide=%llu levels~%lld  us/round: gen+issue %.2f table %.2f vis+check %.2f accept %.2f publish %.2f (%.2f %.2f)\n",
                    j, ctl[j].rounds_narrow, ctl[j].rounds_slow, ctl[j].chunks_wide, ctl[j].level, ctl[j].tacc[0] * r_,
                    ctl[j].tacc[1] * r_, ctl[j].tacc[2] * r_, ctl[j].tacc[3] * r_, ctl[j].tacc[4] * r_, ctl[j].tacc[5] * r_,
                    ctl[j].tacc[6] * r_);
        }
#endif
        if (n == 0) continue;  // the reference's "fail": no seed k-mer passes (out[j].n == 0)
        // one block per job (mc_bfs_result_free releases it through `hi`), laid out as k_bfs_pack wrote it
        char *blk = static_cast<char *>(res_alloc(bfs_pack_bytes(n)));
        if (!blk) {
            ret = fail(c, MC_ENOMEM, "mc_bfs: out of host memory");
            break;
        }
        o->n = n;
        o->hi = reinterpret_cast<uint64_t *>(blk);
        o->lo = o->hi + n;
        o->dist = reinterpret_cast<int32_t *>(o->lo + n);
        o->cov = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(o->dist) + ((4 * n + 7) & ~7ull));
        o->last = reinterpret_cast<uint8_t *>(reinterpret_cast<char *>(o->cov) + ((2 * n + 7) & ~7ull));
        o->levels = h_hdr[j].levels;
        const hipError_t e = hipMemcpyAsync(blk, c->d_bfs_pack + hdr_bytes + h_hdr[j].offset, bfs_pack_bytes(n), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) {
            ret = fail(c, MC_EHIP, "mc_bfs: %s", hipGetErrorString(e));
            break;
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && ret == MC_OK) ret = fail(c, MC_EHIP, "mc_bfs: copying the results failed");
    if (ret != MC_OK)
        for (uint32_t j = 0; j < n_jobs; j++) mc_bfs_result_free(&out[j]);
    return ret;
}

int mc_bfs(mc_ctx *c, const uint64_t *seed_hi, const uint64_t *seed_lo, uint64_t n_seeds, int dir, int min_cov,
           int64_t max_kmers, int64_t max_radius, mc_bfs_result *out)
{
    if (!c) return MC_EINVAL;
    if (!out) return fail(c, MC_EINVAL, "mc_bfs: out is null");
    mc_bfs_job job{seed_hi, seed_lo, n_seeds, dir};
    int rc = mc_bfs_batch(c, &job, 1, min_cov, max_kmers, max_radius, out);
    if (rc) return rc;
    if (out->n == 0) return fail(c, MC_ENOSEED, "Could not find any k-mers of the target gene in the input");
    return MC_OK;
}

}  // extern "C"


// ---- the walk over several ranks' tables where they are (include/mcgpu.h mc_shard_*) -------------------------------------
namespace {
struct ShardWire {  // what a mc_shard_handle holds
    hipIpcMemHandle_t ipc;  // of the table's block (all zero: none could be made; the handle then only serves its own process)
    uint64_t addr, bytes;   // the table in the exporting process
    uint32_t shift, rmask, n_regions;
    int32_t mm_k;
    uint64_t empty;         // count of the key that equals EMPTY_KEY (hash modes)
    int32_t pid;
    int16_t device;
    uint8_t k, key_mode;
    uint64_t token;         // drawn once per process: ranks in separate PID namespaces can share a pid, and a foreign address must never be taken for a local one
    uint32_t magic, has_ipc;
};
static_assert(sizeof(ShardWire) == 128, "the wire format of a shard handle");
uint64_t process_token()
{
    static const uint64_t tok = [] {
        uint64_t v = 0;
        if (FILE *f = fopen("/dev/urandom", "rb")) { if (fread(&v, sizeof v, 1, f) != 1) v = 0; fclose(f); }
        v ^= ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)&v;
        return v ? v : 1;
    }();
    return tok;
}
static_assert(sizeof(ShardWire) <= sizeof(mc_shard_handle), "mc_shard_handle is too small");
constexpr uint32_t SHARD_MAGIC = 0x4D435348u;  // "MCSH"

// this context's counting table, ready to be read by another walker: initialised, nothing parked, counters read
int shard_describe(mc_ctx *c, ShardWire *w, bool want_ipc)
{
    memset(w, 0, sizeof *w);
    if (!c->finalized) return fail(c, MC_ESTATE, "mc_shard_export: call mc_finalize_counts first");
    if (c->solid_external) return fail(c, MC_ESTATE, "mc_shard_export: a BFS-only context has no counting table to share");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = by_key_ready(c);  // (the walkers of other ranks come with keys)
    if (!rc) rc = materialize(c);  // (a rank that owns nothing of the batch still offers a valid, empty table)
    if (rc) return rc;
    unsigned long long *h = c->h_scratch;
    HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, 9 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const TableView t = c->view();
    w->addr = (uint64_t)(uintptr_t)c->slots;
    w->bytes = c->n_slots() * sizeof(Slot);
    w->shift = t.shift; w->rmask = t.rmask; w->n_regions = t.n_regions; w->mm_k = t.mm_k;
    w->empty = h[1];
    w->pid = (int32_t)getpid(); w->device = (int16_t)c->cfg.device; w->k = (uint8_t)c->cfg.k; w->key_mode = (uint8_t)c->cfg.key_mode;
    w->token = process_token();
    w->magic = SHARD_MAGIC;
    if (want_ipc) {
        const hipError_t e = hipIpcGetMemHandle(&w->ipc, c->slots);
        w->has_ipc = e == hipSuccess;
        if (e != hipSuccess) (void)hipGetLastError();  // (the handle still serves this process)
    }
    return MC_OK;
}

void shard_detach_locked(mc_ctx *c)
{
    (void)hipSetDevice(c->cfg.device);
    for (auto &m : c->ipc_opened) m.in_use = false;  // (closed when the next attachments do not ask for them: shard_attach_locked)
    if (c->d_shards) (void)hipFree(c->d_shards);
    c->d_shards = nullptr;
    c->h_shards.clear();
    c->shard_owner_mm_k = 0;
}

int shard_attach_locked(mc_ctx *c, const ShardWire *w, uint32_t n, uint32_t self, int by_minimizer)
{
    shard_detach_locked(c);
    c->shards_dropped = false;
    if (!c->finalized) return fail(c, MC_ESTATE, "mc_shard_attach: call mc_finalize_counts first");
    if (n == 0 || n > 512 || self >= n) return fail(c, MC_EINVAL, "mc_shard_attach: bad number of shards / own index");
    if (by_minimizer && !(c->cfg.key_mode == MC_KEY_PACKED && c->cfg.k >= SK_MIN_K))
        return fail(c, MC_EINVAL, "mc_shard_attach: keys are dealt by minimizer only where reads travel as super-k-mer records (packed keys, k >= %d)", SK_MIN_K);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<ShardRef> refs(n);
    for (uint32_t i = 0; i < n; i++) {
        const ShardWire &x = w[i];
        if (x.magic != SHARD_MAGIC) return fail(c, MC_EINVAL, "mc_shard_attach: shard %u is not a handle of mc_shard_export", i);
        if (x.k != c->cfg.k || x.key_mode != c->cfg.key_mode) return fail(c, MC_EINVAL, "mc_shard_attach: shard %u was counted with k = %d, key mode %d", i, x.k, x.key_mode);
        Slot *slots = nullptr;
        if (i == self) {
            if (x.addr != (uint64_t)(uintptr_t)c->slots || x.token != process_token())
                return fail(c, MC_EINVAL, "mc_shard_attach: shard %u is not this context's table as it is now (export again after counting)", i);
            slots = c->slots;
        } else if (x.token == process_token()) {  // another context of this process: its pointer, through peer access when it is another device's
            slots = reinterpret_cast<Slot *>((uintptr_t)x.addr);
            if (x.device != c->cfg.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, c->cfg.device, x.device) != hipSuccess || !can)
                    return fail(c, MC_EHIP, "mc_shard_attach: GPU %d cannot read GPU %d's memory (no peer access)", c->cfg.device, x.device);
                const hipError_t e = hipDeviceEnablePeerAccess(x.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(c, MC_EHIP, "mc_shard_attach: hipDeviceEnablePeerAccess: %s", hipGetErrorString(e));
                (void)hipGetLastError();
            }
        } else {
            if (!x.has_ipc) { shard_detach_locked(c); return fail(c, MC_EHIP, "mc_shard_attach: shard %u comes from another process without an IPC handle", i); }
            void *p = nullptr;
            for (auto &m : c->ipc_opened)  // (the same table as last time: its mapping is still there)
                if (!m.in_use && memcmp(&m.h, &x.ipc, sizeof x.ipc) == 0 && m.addr == x.addr && m.bytes == x.bytes) { p = m.p; m.in_use = true; break; }
            if (!p) {
                const hipError_t e = hipIpcOpenMemHandle(&p, x.ipc, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) { shard_detach_locked(c); return fail(c, MC_EHIP, "mc_shard_attach: hipIpcOpenMemHandle (shard %u): %s", i, hipGetErrorString(e)); }
                c->ipc_opened.push_back(mc_ctx::IpcMap{x.ipc, p, true, x.addr, x.bytes});
            }
            slots = static_cast<Slot *>(p);
        }
        refs[i].slots = slots; refs[i].shift = x.shift; refs[i].rmask = x.rmask; refs[i].n_regions = x.n_regions; refs[i].mm_k = x.mm_k;
        refs[i].empty = x.empty;
    }
    if (hipMalloc(reinterpret_cast<void **>(&c->d_shards), n * sizeof(ShardRef)) != hipSuccess) { shard_detach_locked(c); return fail(c, MC_ENOMEM, "mc_shard_attach: out of device memory"); }
    if (hipMemcpy(c->d_shards, refs.data(), n * sizeof(ShardRef), hipMemcpyHostToDevice) != hipSuccess) { shard_detach_locked(c); return fail(c, MC_EHIP, "mc_shard_attach: upload failed"); }
    // Mappings nobody asked for this time are closed at once (kept until twice as many had gathered, a peer's table
    // that had been replaced -- grown, or given back to the driver -- stayed pinned through its mapping, up to 137 GB each).
    {
        std::vector<mc_ctx::IpcMap> keep;
        for (auto &m : c->ipc_opened) { if (m.in_use) keep.push_back(m); else (void)hipIpcCloseMemHandle(m.p); }
        c->ipc_opened.swap(keep);
    }
    c->h_shards = refs;
    c->shard_self = self;
    c->shard_owner_mm_k = by_minimizer ? c->cfg.k : 0;
    return MC_OK;
}
}  // namespace

extern "C" {

int mc_shard_export(mc_ctx *c, mc_shard_handle *out)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!out) return fail(c, MC_EINVAL, "mc_shard_export: out is null");
    ShardWire w;
    const int rc = shard_describe(c, &w, true);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    memcpy(out, &w, sizeof w);
    return MC_OK;
}

int mc_shard_attach(mc_ctx *c, const mc_shard_handle *shards, uint32_t n_shards, uint32_t self, int by_minimizer)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!shards) return fail(c, MC_EINVAL, "mc_shard_attach: shards is null");
    if (n_shards == 0 || n_shards > 512) return fail(c, MC_EINVAL, "mc_shard_attach: bad number of shards / own index");  // (before anything is sized by it)
    std::vector<ShardWire> w(n_shards);
    for (uint32_t i = 0; i < n_shards; i++) memcpy(&w[i], &shards[i], sizeof(ShardWire));
    return shard_attach_locked(c, w.data(), n_shards, self, by_minimizer);
}

int mc_shard_detach(mc_ctx *c)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    shard_detach_locked(c);
    c->shards_dropped = false;
    return MC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ several GPUs as one table
// mc_group: the native counterpart of metacherchant_amd/distributed.py, for a host that is one
// process: a context per device, one host thread per device for the work, and the exchange as peer-to-peer copies --
// every (source, destination) pair at once, so that all xGMI links of a GPU are busy together, which is what an
// all-to-all over RCCL's grouped send / recv does too.  Reads are dealt to the devices in equal contiguous shares, every
// device turns its share into super-k-mer records (keys for k < 23 / hash keys) bucketed by owner, the buckets travel,
// every device counts what it owns (disjoint key sets).  For the BFS the shards' k-mers at or above the threshold are
// gathered on the first device into a BFS-only context that borrows that device's read store (read pointers refer to the
// first device's reads: the others keep none).  What the reference does instead: P threads over one shared map
// (src/io/IOUtils.java:283-315, src/io/ReadsDispatcher.java:34-53).

// RCCL, loaded when a group asks for it (MC_GROUP_TRANSPORT=rccl / MC_FLAG_GROUP_RCCL): the library is large and a process
// that counts on one GPU, or moves its buckets with peer copies, never needs it.  Types and constants from <rccl/rccl.h>.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string *why)
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { *why = std::string("librccl.so.1 cannot be loaded: ") + dlerror(); return false; }
#define RCCL_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym)); if (!field) { *why = std::string("librccl has no ") + sym; return false; }
        RCCL_SYM(CommInitAll, "ncclCommInitAll") RCCL_SYM(CommDestroy, "ncclCommDestroy") RCCL_SYM(GroupStart, "ncclGroupStart")
        RCCL_SYM(GroupEnd, "ncclGroupEnd") RCCL_SYM(Send, "ncclSend") RCCL_SYM(Recv, "ncclRecv") RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
        return true;
    }
};
static RcclApi g_rccl;

struct mc_group {
    std::vector<mc_ctx *> ctx;  // one per device, in the order given
    mc_ctx *solid = nullptr;    // BFS-only context on the first device (n > 1)
    int solid_cov = -1;
    bool dirty = true;          // counts changed since the shards were gathered
    mc_config cfg{};
    std::string err;
    std::mutex mu;
    // how the buckets travel between the devices: peer-to-peer copies (every pair at once), or RCCL -- one communicator per
    // device in this process (ncclCommInitAll), the exchange as grouped ncclSend / ncclRecv, which is what an all-to-all is
    bool use_rccl = false;
    std::vector<ncclComm_t> comm;
    bool peer_all = true;       // every pair of distinct devices has peer access (else HIP stages those copies through the host)
    bool gather_reads = false;  // every device's reads are brought to the first device's store, and every record carries a pointer into it
};

namespace {

int gfail(mc_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    return code;
}

// runs f(rank) on one thread per rank; returns the first non-zero result
template <typename F>
int per_rank(size_t n, F &&f)
{
    std::vector<int> rc(n, 0);
    std::vector<std::thread> th;
    auto guarded = [&](size_t r) {  // (an exception must not leave a thread: std::terminate)
        try { rc[r] = f(r); } catch (const std::bad_alloc &) { rc[r] = MC_ENOMEM; } catch (...) { rc[r] = MC_EINVAL; }
    };
    for (size_t r = 1; r < n; r++) th.emplace_back([&, r] { guarded(r); });
    guarded(0);
    for (auto &t : th) t.join();
    for (int x : rc) if (x) return x;
    return MC_OK;
}

// bytes from a buffer of src's device to one of dst's (the same device: a plain device copy)
hipError_t peer_copy(void *d, const mc_ctx *dst, const void *s, const mc_ctx *src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    if (dst->cfg.device == src->cfg.device) return hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(d, dst->cfg.device, s, src->cfg.device, bytes, st);
}

}  // namespace

extern "C" {

const char *mc_group_last_error(const mc_group *g) { return g ? g->err.c_str() : g_create_err.c_str(); }

void mc_group_destroy(mc_group *g)
{
    if (!g) return;
    for (ncclComm_t cm : g->comm) if (cm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(cm);
    if (g->solid) mc_destroy(g->solid);
    for (mc_ctx *c : g->ctx) mc_destroy(c);
    delete g;
}

int mc_group_create(const mc_config *cfg, const int32_t *devices, uint32_t n_devices, mc_group **out)
{
    if (!cfg || !devices || !out || n_devices == 0 || n_devices > 64) return fail(nullptr, MC_EINVAL, "mc_group_create: bad argument");
    *out = nullptr;
    mc_group *g = new (std::nothrow) mc_group;
    if (!g) return fail(nullptr, MC_ENOMEM, "mc_group_create: out of host memory");
    g->cfg = *cfg;
    for (uint32_t r = 0; r < n_devices; r++) {
        mc_config c = *cfg;
        c.device = devices[r];
        c.capacity_hint = cfg->capacity_hint ? cfg->capacity_hint / n_devices + (1u << 20) : 0;  // owners hold equal shares of the keys
        if (n_devices > 1) c.flags |= MC_FLAG_SOLID_LIST;                                           // its solid k-mers will be exported
        mc_ctx *x = nullptr;
        const int rc = mc_create(&c, &x);
        if (rc) { mc_group_destroy(g); return rc; }
        g->ctx.push_back(x);
        // The BFS device reads its look-ahead from ITS store: the other devices' reads are brought there (group_import_reads; a file's
        // reads are tokenised there in the first place) and every device works its pointers out as if its reads sat in that store
        // -- a walk that had only the first device's reads to follow took 20 times as long at 8 devices (include/mcgpu.h
        // mc_set_read_pointers).  MC_EXCHANGE_GATHER_READS=0: the other devices' records carry no pointers.
        static const bool gather = [] { const char *e = getenv("MC_EXCHANGE_GATHER_READS"); return !(e && !strcmp(e, "0")); }();
        g->gather_reads = gather && n_devices > 1;
        if (n_devices > 1) (void)mc_set_read_pointers(x, g->gather_reads ? ((r == 0 ? 1 : 2) | MC_PTRS_ON_EVERY_RECORD) : (r == 0 ? 1 : 0));
    }
    // peer access between every pair of different devices (already enabled: fine).  A pair without it still works -- HIP
    // stages such copies through the host -- but at a fraction of the xGMI rate: say so once.
    for (uint32_t a = 0; a < n_devices; a++)
        for (uint32_t b = 0; b < n_devices; b++) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) != hipSuccess) can = 0;
            bool ok = false;
            if (can && hipSetDevice(devices[a]) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
            if (!ok && g->peer_all) {
                g->peer_all = false;
                fprintf(stderr, "WARN: no peer access from GPU %d to GPU %d: the exchange between them is staged through the host\n", devices[a], devices[b]);
            }
        }
    {   // transport of the exchange
        const char *e = getenv("MC_GROUP_TRANSPORT");
        g->use_rccl = (cfg->flags & MC_FLAG_GROUP_RCCL) != 0 || (e && !strcmp(e, "rccl"));
        if (e && !strcmp(e, "peer")) g->use_rccl = false;
        if (g->use_rccl && n_devices > 1) {
            for (uint32_t a = 0; a < n_devices; a++)
                for (uint32_t b = a + 1; b < n_devices; b++)
                    if (devices[a] == devices[b]) { mc_group_destroy(g); return fail(nullptr, MC_EINVAL, "mc_group_create: RCCL wants every rank on a GPU of its own (GPU %d is named twice); use the peer-copy transport for shares of one GPU", devices[a]); }
            std::string why;
            if (!g_rccl.load(&why)) { mc_group_destroy(g); return fail(nullptr, MC_EINVAL, "mc_group_create: %s", why.c_str()); }
            g->comm.assign(n_devices, nullptr);
            std::vector<int> devs(devices, devices + n_devices);
            const ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), (int)n_devices, devs.data());
            if (r != ncclSuccess) { mc_group_destroy(g); return fail(nullptr, MC_EHIP, "mc_group_create: ncclCommInitAll: %s", g_rccl.GetErrorString(r)); }
        } else {
            g->use_rccl = false;
        }
    }
    if (n_devices > 1) {
        mc_config c = *cfg;
        c.device = devices[0];
        c.capacity_hint = 1u << 20;
        c.flags = 0;
        const int rc = mc_create(&c, &g->solid);
        if (rc) { mc_group_destroy(g); return rc; }
    }
    *out = g;
    return MC_OK;
}

int mc_group_set_coverage_hint(mc_group *g, int min_cov)
{
    if (!g) return MC_EINVAL;
    for (mc_ctx *c : g->ctx) {
        const int rc = mc_set_coverage_hint(c, min_cov);
        if (rc) return gfail(g, rc, mc_last_error(c));
    }
    return MC_OK;
}

namespace {
// what one device brings to an exchange: its share of the reads (device pointers) and, filled on the way, what it sends and receives
struct GroupRank {
    DevBuf<uint64_t> dw, doff, send, recv;   // reads uploaded here (host batches); what goes out (records: 2 words each, or keys) and what came in
    DevBuf<uint32_t> send_p, recv_p;         // the read pointers that travel with them
    DevBuf<uint32_t> fc, recv_fc;            // binned records: a row of fine-bucket counts for every owner; the rows every source sent me
    std::vector<uint64_t> owner_off;         // owner o's piece of `send` = [owner_off[o], owner_off[o + 1])
    std::vector<uint64_t> owner_win;         // binned records: the windows of owner o's piece
    uint64_t n_recv = 0;
    const uint64_t *d_words = nullptr, *d_off = nullptr;  // the share: d_off[0 .. n_reads] index bases of d_words
    uint64_t n_reads = 0, n_bases = 0, windows = 0;       // n_bases = d_off[n_reads]; windows: k-mer occurrences, or an upper bound
    int64_t in_store = -1;                                // >= 0: the share sits in this device's read store from that word on
};

// n_words packed words of device `from`'s reads to word `at_word` of the first device's read store (where that device's pointers lead)
int group_import_reads(mc_group *g, size_t from, const uint64_t *src, uint64_t n_words, uint64_t at_word)
{
    mc_ctx *c0 = g->ctx[0];
    std::lock_guard<std::mutex> g0(c0->mu);
    if (hipSetDevice(c0->cfg.device) != hipSuccess) return MC_EHIP;
    if (at_word + n_words + 1 > c0->rs_cap_words) {
        const uint64_t fill = c0->rs_bases / 32;
        int rc = rs_reserve(c0, at_word + n_words + 1 > fill ? at_word + n_words + 1 - fill : 0);
        if (rc) return rc;
    }
    if (peer_copy(c0->rs_words + at_word, c0, src, g->ctx[from], n_words * 8, c0->stream) != hipSuccess || hipStreamSynchronize(c0->stream) != hipSuccess) return MC_EHIP;
    c0->rs_hi_bases = std::max(c0->rs_hi_bases, (at_word + n_words) * 32);
    return MC_OK;
}

// every device: its share -> records (keys) bucketed by owner; the exchange; every device counts what it owns
int group_exchange_count(mc_group *g, std::vector<GroupRank> &R)
{
    const size_t W = g->ctx.size();
    const bool sk = g->ctx[0]->sk_form;
    bool sk_batch = sk;  // this batch travels as super-k-mer records (false: as keys)
    // the binned form of the record exchange (include/mcgpu.h mc_extract_superkmers_binned_dev) where every device's table is laid
    // out for the same level-1 buckets: the sender orders an owner's records by them, and the owner's run starts at its second level
    uint32_t fine = 0;
    if (sk) {
        fine = mc_superkmer_fine_buckets(g->ctx[0], (uint32_t)W);
        for (size_t r = 1; r < W && fine; r++) if (mc_superkmer_fine_buckets(g->ctx[r], (uint32_t)W) != fine) fine = 0;
    }
    auto extract = [&](size_t r, bool as_records) -> int {
        mc_ctx *c = g->ctx[r];
        GroupRank &X = R[r];
        X.owner_off.assign(W + 1, 0);
        X.owner_win.assign(W, 0);
        if (hipSetDevice(c->cfg.device) != hipSuccess) return MC_EHIP;
        if (as_records && fine && X.fc.alloc((uint64_t)W * fine) != hipSuccess) return MC_ENOMEM;
        if (X.n_reads == 0 || X.windows == 0) {  // (a rank without reads still sends its rows: zeros)
            if (as_records && fine && (hipMemsetAsync(X.fc.p, 0, (uint64_t)W * fine * 4, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) return MC_EHIP;
            return MC_OK;
        }
        X.send.reset();
        X.send_p.reset();
        c->extract_in_store = X.in_store;
        if (as_records) {
            const uint64_t cap = mc_superkmer_capacity(c, X.windows, X.n_reads);
            if (X.send.alloc(cap * 2) != hipSuccess || X.send_p.alloc(cap) != hipSuccess) return MC_ENOMEM;
            if (fine)
                return mc_extract_superkmers_binned_dev(c, X.d_words, X.d_off, X.n_reads, X.n_bases, (uint32_t)W, fine, X.send.p, X.send_p.p, cap, X.fc.p,
                                                        X.owner_off.data(), X.owner_win.data());
            return mc_extract_superkmers_dev(c, X.d_words, X.d_off, X.n_reads, X.n_bases, (uint32_t)W, X.send.p, X.send_p.p, cap, X.owner_off.data());
        }
        if (X.send.alloc(X.windows) != hipSuccess || X.send_p.alloc(X.windows) != hipSuccess) return MC_ENOMEM;
        // A group that deals super-k-mer records deals by minimizer (sk_owner), and the owner of a k-mer must not depend on the
        // form a batch travels in: one batch as keys dealt by the key's own hash would split a k-mer's count over two shards
        // -- each under the threshold, perhaps, and the k-mer gone from the graph.
        c->extract_by_minimizer = sk;
        return mc_extract_keys_dev(c, X.d_words, X.d_off, X.n_reads, X.n_bases, (uint32_t)W, reinterpret_cast<int64_t *>(X.send.p), X.send_p.p, X.windows,
                                   X.owner_off.data());
    };
    std::vector<uint64_t> fill(W, 0);  // (where every device's reads go, or are deemed to go, in the shared store: a second extraction starts there again)
    if (g->gather_reads) for (size_t r = 0; r < W; r++) fill[r] = mc_read_store_tell(g->ctx[r]);
    int rc = per_rank(W, [&](size_t r) -> int { return extract(r, sk); });
    if (rc == MC_EOVERFLOW && sk) {
        if (g->gather_reads) for (size_t r = 0; r < W; r++) (void)mc_read_store_seek(g->ctx[r], fill[r], 0);
        // an owner's piece overflowed (reads of low complexity share one minimizer, hence one owner): this batch travels as
        // keys instead, on every rank alike -- one record per window, counted per owner first and then packed (no piece to
        // overflow), and dealt to the SAME owners: those of the keys' minimizers
        sk_batch = false;
        for (mc_ctx *c : g->ctx) c->err.clear();
        if (getenv("MC_INGEST_DEBUG")) fprintf(stderr, "[ingest] group: an owner's piece of the record form overflowed; this batch travels as keys, dealt by minimizer\n");
        rc = per_rank(W, [&](size_t r) -> int { return extract(r, false); });
    }
    if (rc) {
        for (mc_ctx *c : g->ctx) if (!c->err.empty()) return gfail(g, rc, c->err);
        return gfail(g, rc, "mc_group_add_reads_packed: extraction failed");
    }
    // ---- the exchange: owner d receives its piece of every rank's output, all pairs at once
    const size_t unit = sk_batch ? 16 : 8;
    const bool binned = sk_batch && fine != 0;
    if (g->use_rccl) {
        // RCCL: every device's receive buffer first, then ONE group of sends and receives over all pairs (what an all-to-all
        // with per-pair counts is), each communicator's calls on its own context's stream
        rc = per_rank(W, [&](size_t d) -> int {
            mc_ctx *c = g->ctx[d];
            if (hipSetDevice(c->cfg.device) != hipSuccess) return MC_EHIP;
            uint64_t total = 0;
            for (size_t r = 0; r < W; r++) total += R[r].owner_off[d + 1] - R[r].owner_off[d];
            R[d].n_recv = total;
            if (binned && R[d].recv_fc.alloc((uint64_t)W * fine) != hipSuccess) return MC_ENOMEM;
            if (total == 0) return MC_OK;
            return R[d].recv.alloc(total * (sk_batch ? 2 : 1)) == hipSuccess && R[d].recv_p.alloc(total) == hipSuccess ? MC_OK : MC_ENOMEM;
        });
        if (rc) return gfail(g, rc, "mc_group_add_reads_packed: no room for the received buckets");
        struct Zero { size_t d; uint64_t at, m; };
        std::vector<Zero> ptr_zero;
        ncclResult_t nr = g_rccl.GroupStart();
        std::vector<uint64_t> at(W, 0);
        for (size_t r = 0; r < W && nr == ncclSuccess; r++)       // source
            for (size_t d = 0; d < W && nr == ncclSuccess; d++) {  // owner
                const uint64_t o0 = R[r].owner_off[d], m = R[r].owner_off[d + 1] - o0;
                if (binned) {  // source r's row of counts for owner d
                    nr = g_rccl.Send(R[r].fc.p + d * fine, (size_t)fine * 4, ncclUint8, (int)d, g->comm[r], g->ctx[r]->stream);
                    if (nr == ncclSuccess) nr = g_rccl.Recv(R[d].recv_fc.p + r * fine, (size_t)fine * 4, ncclUint8, (int)r, g->comm[d], g->ctx[d]->stream);
                    if (nr != ncclSuccess) break;
                }
                if (m == 0) continue;
                nr = g_rccl.Send(reinterpret_cast<const char *>(R[r].send.p) + o0 * unit, m * unit, ncclUint8, (int)d, g->comm[r], g->ctx[r]->stream);
                if (nr == ncclSuccess) nr = g_rccl.Recv(reinterpret_cast<char *>(R[d].recv.p) + at[d] * unit, m * unit, ncclUint8, (int)r, g->comm[d], g->ctx[d]->stream);
                // (read pointers lead into the FIRST device's read store, the one the walk reads: the other devices keep none, and the
                // zeros they used to send -- a fifth of their bytes -- are filled in where they arrive)
                if (r == 0 || g->gather_reads) {  // (gather_reads: every device's reads are in that store, every record brings its pointer)
                    if (nr == ncclSuccess) nr = g_rccl.Send(R[r].send_p.p + o0, m * 4, ncclUint8, (int)d, g->comm[r], g->ctx[r]->stream);
                    if (nr == ncclSuccess) nr = g_rccl.Recv(R[d].recv_p.p + at[d], m * 4, ncclUint8, (int)r, g->comm[d], g->ctx[d]->stream);
                } else {
                    ptr_zero.push_back({d, at[d], m});
                }
                at[d] += m;
            }
        const ncclResult_t ne = g_rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        for (const Zero &z : ptr_zero) {
            if (hipSetDevice(g->ctx[z.d]->cfg.device) != hipSuccess || hipMemsetAsync(R[z.d].recv_p.p + z.at, 0, z.m * 4, g->ctx[z.d]->stream) != hipSuccess)
                return gfail(g, MC_EHIP, "mc_group_add_reads_packed: clearing the pointers of pointer-less records failed");
        }
        if (nr != ncclSuccess) return gfail(g, MC_EHIP, std::string("mc_group_add_reads_packed: RCCL exchange: ") + g_rccl.GetErrorString(nr));
        rc = per_rank(W, [&](size_t d) -> int {
            mc_ctx *c = g->ctx[d];
            if (hipSetDevice(c->cfg.device) != hipSuccess) return MC_EHIP;
            return hipStreamSynchronize(c->stream) == hipSuccess ? MC_OK : MC_EHIP;
        });
        if (rc) return gfail(g, rc, "mc_group_add_reads_packed: the RCCL exchange failed");
    } else
    rc = per_rank(W, [&](size_t d) -> int {
        mc_ctx *c = g->ctx[d];
        if (hipSetDevice(c->cfg.device) != hipSuccess) return MC_EHIP;
        uint64_t total = 0;
        for (size_t r = 0; r < W; r++) total += R[r].owner_off[d + 1] - R[r].owner_off[d];
        R[d].n_recv = total;
        if (total == 0) return MC_OK;
        if (R[d].recv.alloc(total * (sk_batch ? 2 : 1)) != hipSuccess || R[d].recv_p.alloc(total) != hipSuccess) return MC_ENOMEM;
        if (binned && R[d].recv_fc.alloc((uint64_t)W * fine) != hipSuccess) return MC_ENOMEM;
        uint64_t at = 0;
        for (size_t r = 0; r < W; r++) {
            const uint64_t o0 = R[r].owner_off[d], m = R[r].owner_off[d + 1] - o0;
            if (binned && peer_copy(R[d].recv_fc.p + r * fine, c, R[r].fc.p + d * fine, g->ctx[r], (size_t)fine * 4, c->stream) != hipSuccess) return MC_EHIP;
            // (pointers: from the first device only -- they lead into its read store; the others' would be zeros)
            if (peer_copy(reinterpret_cast<char *>(R[d].recv.p) + at * unit, c, reinterpret_cast<const char *>(R[r].send.p) + o0 * unit, g->ctx[r], m * unit, c->stream) != hipSuccess ||
                (r == 0 || g->gather_reads ? peer_copy(R[d].recv_p.p + at, c, R[r].send_p.p + o0, g->ctx[r], m * 4, c->stream)
                        : (m ? hipMemsetAsync(R[d].recv_p.p + at, 0, m * 4, c->stream) : hipSuccess)) != hipSuccess)
                return MC_EHIP;
            at += m;
        }
        return hipStreamSynchronize(c->stream) == hipSuccess ? MC_OK : MC_EHIP;
    });
    if (rc) return gfail(g, rc, "mc_group_add_reads_packed: the exchange between the devices failed");
    // ---- every device counts what it owns
    rc = per_rank(W, [&](size_t d) -> int {
        if (R[d].n_recv == 0) return MC_OK;
        if (binned) {  // every source's piece is a part, in the order the pieces were received
            std::vector<uint64_t> part_off(W + 1, 0);
            uint64_t windows = 0;
            for (size_t r = 0; r < W; r++) {
                part_off[r + 1] = part_off[r] + (R[r].owner_off[d + 1] - R[r].owner_off[d]);
                windows += R[r].owner_win[d];
            }
            return mc_add_superkmers_binned_dev(g->ctx[d], R[d].recv.p, R[d].recv_p.p, R[d].n_recv, windows, fine, (uint32_t)W, part_off.data(), R[d].recv_fc.p);
        }
        return sk_batch ? mc_add_superkmers_dev(g->ctx[d], R[d].recv.p, R[d].recv_p.p, R[d].n_recv)
                  : mc_add_keys_dev(g->ctx[d], reinterpret_cast<const int64_t *>(R[d].recv.p), R[d].recv_p.p, R[d].n_recv);
    });
    if (rc) {
        for (mc_ctx *c : g->ctx) if (!c->err.empty()) return gfail(g, rc, c->err);
        return gfail(g, rc, "mc_group_add_reads_packed: counting failed");
    }
    return MC_OK;
}
}  // namespace

namespace {
// a batch of packed reads in host memory: shares up to the devices, exchange, count (the caller holds the group's lock)
int group_add_host_batch(mc_group *g, const uint64_t *words, const uint64_t *off, uint64_t n_reads)
{
    const size_t W = g->ctx.size();
    const int k = g->cfg.k;
    std::vector<GroupRank> R(W);
    int rc = per_rank(W, [&](size_t r) -> int {
        mc_ctx *c = g->ctx[r];
        const uint64_t a = n_reads * r / W, b = n_reads * (r + 1) / W;
        if (a == b) return MC_OK;
        if (hipSetDevice(c->cfg.device) != hipSuccess) return MC_EHIP;
        const uint64_t w0 = off[a] / 32, w1 = (off[b] + 31) / 32 + 1, nb = off[b] - w0 * 32;
        std::vector<uint64_t> rel(off + a, off + b + 1);
        uint64_t windows = 0;
        for (uint64_t i = 0; i < b - a; i++) {
            const uint64_t len = rel[i + 1] - rel[i];
            if (len >= (uint64_t)k) windows += len - (uint64_t)k + 1;
        }
        for (auto &x : rel) x -= w0 * 32;
        if (R[r].dw.alloc(w1 - w0) != hipSuccess || R[r].doff.alloc(b - a + 1) != hipSuccess) return MC_ENOMEM;
        // (through the context's pinned staging buffers: a pageable copy runs at a third of the link's rate)
        if (h2d_fast(c, R[r].dw.p, words + w0, (w1 - w0) * 8) != MC_OK || h2d_fast(c, R[r].doff.p, rel.data(), (b - a + 1) * 8) != MC_OK) return MC_EHIP;
        R[r].d_words = R[r].dw.p; R[r].d_off = R[r].doff.p;
        R[r].n_reads = b - a; R[r].n_bases = nb; R[r].windows = windows;
        return MC_OK;
    });
    if (rc) {
        for (mc_ctx *c : g->ctx) if (!c->err.empty()) return gfail(g, rc, c->err);
        return gfail(g, rc, "mc_group_add_reads_packed: the reads did not reach the devices");
    }
    uint64_t end_word = 0;
    if (g->gather_reads) {
        // the shares one behind the other in the first device's store: its own where its extraction will append it, the others'
        // copied there from their devices; every other device is told where its share sits (its pointers lead there)
        mc_ctx *c0 = g->ctx[0];
        uint64_t at = 0;
        {
            std::lock_guard<std::mutex> g0(c0->mu);
            at = c0->rs_end() / 32;
        }
        std::vector<uint64_t> pos(W + 1, at);
        for (size_t r = 0; r < W; r++) pos[r + 1] = pos[r] + (R[r].n_reads ? (R[r].n_bases + 31) / 32 + 1 : 0);
        end_word = pos[W];
        if ((rc = mc_read_store_seek(c0, pos[0] * 32, end_word * 32 + 64)) != MC_OK) return gfail(g, rc, c0->err);
        rc = per_rank(W, [&](size_t r) -> int {
            if (r == 0 || R[r].n_reads == 0) return MC_OK;
            int x = group_import_reads(g, r, R[r].dw.p, (R[r].n_bases + 31) / 32 + 1, pos[r]);
            return x ? x : mc_read_store_seek(g->ctx[r], pos[r] * 32, 0);
        });
        if (rc) {
            for (mc_ctx *c : g->ctx) if (!c->err.empty()) return gfail(g, rc, c->err);
            return gfail(g, rc, "mc_group_add_reads_packed: the reads did not reach the first device's store");
        }
    }
    rc = group_exchange_count(g, R);
    if (g->gather_reads && rc == MC_OK && (rc = mc_read_store_seek(g->ctx[0], end_word * 32, 0)) != MC_OK) return gfail(g, rc, g->ctx[0]->err);  // (the next batch goes behind all of them)
    return rc;
}
}  // namespace

int mc_group_add_reads_packed(mc_group *g, const uint64_t *words, const uint64_t *off, uint64_t n_reads)
{
    if (!g) return MC_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    if ((!words || !off) && n_reads) return gfail(g, MC_EINVAL, "mc_group_add_reads_packed: null pointer");
    if (n_reads == 0) return MC_OK;
    g->dirty = true;
    if (g->ctx.size() == 1) {
        const int rc = mc_add_reads_packed(g->ctx[0], words, off, n_reads);
        return rc ? gfail(g, rc, mc_last_error(g->ctx[0])) : MC_OK;
    }
    return group_add_host_batch(g, words, off, n_reads);
}

int mc_group_add_reads_file(mc_group *g, const char *path, uint64_t *n_reads)
{
    if (!g) return MC_EINVAL;
    if (n_reads) *n_reads = 0;
    if (!path) return gfail(g, MC_EINVAL, "mc_group_add_reads_file: null path");
    if (g->ctx.size() == 1) {
        const int rc = mc_add_reads_file(g->ctx[0], path, n_reads);
        return rc ? gfail(g, rc, mc_last_error(g->ctx[0])) : MC_OK;
    }
    try {
        // Batches of 2^24 reads per device (2.5 G bases of 150-base reads: 0.6 GB of host memory per device): every batch is
        // one exchange and one counting run on every device, and a counting run rewrites the device's whole table -- with
        // 2^20 reads a batch, as before, a billion reads meant 119 rewrites of a 100 GB-class table per device.
        static const uint64_t per_dev = [] { const char *e = getenv("MC_GROUP_BATCH_READS"); return e && *e ? std::max<uint64_t>(strtoull(e, nullptr, 10), 1024) : 1ull << 24; }();
        int rc = MC_OK;
        // Uncompressed FASTA / FASTQ: tokenised on the FIRST device (csrc/tokenizer.h, chunk i + 1 crossing PCIe while chunk i
        // is tokenised), into its read store -- the one the walk reads --; when W x per_dev reads have gathered, or the file
        // ends, the other devices fetch their shares of the packed reads from it (device to device) and the batch is exchanged
        // and counted.  A chunk the device declines goes through the host reader like any other file.
        const char *mode = getenv("MC_TOKENIZER");
        mch::PlainReadsFile f;
        if (!(mode && !strcmp(mode, "host")) && g->ctx[0]->rs_enabled && mch::map_plain_reads(path, &f)) {
            std::lock_guard<std::mutex> lk(g->mu);
            g->dirty = true;
            mc_ctx *c0 = g->ctx[0];
            const size_t W = g->ctx.size();
            const char *e_chunk = getenv("MC_TOKENIZER_CHUNK_BYTES");
            const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(e_chunk && *e_chunk ? strtoull(e_chunk, nullptr, 10) : (1ull << 28), 64), 3ull << 29);
            std::vector<std::pair<const char *, const char *>> cuts;
            for (const char *b = f.p, *end = f.p + f.n; b < end;) {
                const char *e = (uint64_t)(end - b) <= chunk + chunk / 4 ? end : mch::plain_record_start(f, b + chunk);
                cuts.emplace_back(b, e);
                b = e;
            }
            {
                std::lock_guard<std::mutex> g0(c0->mu);
                HIPCHK(c0, hipSetDevice(c0->cfg.device));
                const int r = rs_reserve(c0, (f.fastq ? f.n / 2 : f.n) / 32 + 2 * cuts.size() + 2);
                if (r) return gfail(g, r, c0->err);
            }
            std::unique_ptr<TokPending> pend(new TokPending);
            auto drop_pend = [&] { std::lock_guard<std::mutex> g0(c0->mu); pend.reset(); };
            uint64_t total = 0, n_flushes = 0;
            // the reads gathered in `pend` so far: shares to the devices, exchange, count; the store then ends behind them
            auto flush = [&]() -> int {
                if (pend->reads == 0) { pend->base_word = -1; pend->bases = 0; return MC_OK; }
                n_flushes++;
                const uint64_t n = pend->reads, base_word = (uint64_t)pend->base_word;
                std::vector<uint64_t> cut_off(W + 1, 0);  // base offset (relative to the batch) of every share's first read
                {
                    std::lock_guard<std::mutex> g0(c0->mu);
                    if (hipSetDevice(c0->cfg.device) != hipSuccess) return gfail(g, MC_EHIP, "mc_group_add_reads_file: hipSetDevice");
                    for (size_t r = 0; r <= W; r++)
                        if (hipMemcpyAsync(&cut_off[r], pend->off.p + n * r / W, 8, hipMemcpyDeviceToHost, c0->stream) != hipSuccess) return gfail(g, MC_EHIP, "mc_group_add_reads_file: copy failed");
                    if (hipStreamSynchronize(c0->stream) != hipSuccess) return gfail(g, MC_EHIP, "mc_group_add_reads_file: copy failed");
                }
                std::vector<GroupRank> R(W);
                int frc = per_rank(W, [&](size_t r) -> int {
                    mc_ctx *c = g->ctx[r];
                    const uint64_t a = n * r / W, b = n * (r + 1) / W;
                    if (a == b) return MC_OK;
                    GroupRank &X = R[r];
                    X.n_reads = b - a;
                    X.n_bases = cut_off[r + 1];
                    X.windows = cut_off[r + 1] - cut_off[r];  // (an upper bound: a window per base)
                    if (r == 0) {  // in place: the share is where the walk will read it
                        X.d_words = c0->rs_words + base_word;
                        X.d_off = pend->off.p;
                        X.in_store = (int64_t)base_word;
                        return MC_OK;
                    }
                    // the words [w0, w1] of the share and its offsets come over; d_words is handed on as if it held the batch from
                    // its first word (the offsets stay relative to the batch: nothing below w0 is looked at)
                    if (hipSetDevice(c->cfg.device) != hipSuccess) return MC_EHIP;
                    // (with 512 words before the share's first: the extract kernels load whole tiles, up to 8191 bases ahead of it)
                    const uint64_t w0 = std::max<uint64_t>(cut_off[r] / 32, 512) - 512, w1 = (cut_off[r + 1] + 31) / 32 + 1;
                    if (X.dw.alloc(w1 - w0) != hipSuccess || X.doff.alloc(b - a + 1) != hipSuccess) return MC_ENOMEM;
                    if (peer_copy(X.dw.p, c, c0->rs_words + base_word + w0, c0, (w1 - w0) * 8, c->stream) != hipSuccess ||
                        peer_copy(X.doff.p, c, pend->off.p + a, c0, (b - a + 1) * 8, c->stream) != hipSuccess ||
                        hipStreamSynchronize(c->stream) != hipSuccess)
                        return MC_EHIP;
                    X.d_words = X.dw.p - w0;
                    X.d_off = X.doff.p;
                    // (the share sits in the first device's store: batch position p is store position base_word * 32 + p, which is
                    // what this device's pointers come to when its -- deemed -- fill stands at the share's first word)
                    if (g->gather_reads) return mc_read_store_seek(c, (base_word + cut_off[r] / 32) * 32, 0);
                    return MC_OK;
                });
                if (frc) return gfail(g, frc, "mc_group_add_reads_file: the shares did not reach the devices");
                frc = group_exchange_count(g, R);
                {   // the batch's reads stay in the first device's store, whoever counted them
                    std::lock_guard<std::mutex> g0(c0->mu);
                    c0->rs_bases = std::max<uint64_t>(c0->rs_bases, (base_word + (pend->bases + 31) / 32) * 32);
                }
                pend->reads = pend->bases = 0;
                pend->base_word = -1;
                return frc;
            };
            struct Upload {
                PoolBuf<uint8_t> text;
                std::thread th;
                bool ok = true;
            };
            std::unique_ptr<Upload> cur, nxt;
            auto start_upload = [&](size_t i, std::unique_ptr<Upload> &u) -> int {
                u.reset(new Upload);
                const uint64_t n = (uint64_t)(cuts[i].second - cuts[i].first);
                {
                    std::lock_guard<std::mutex> g0(c0->mu);
                    HIPCHK(c0, hipSetDevice(c0->cfg.device));
                    HIPCHK(c0, u->text.alloc(&c0->tok_pool, (n + tok::T_TILE - 1) / tok::T_TILE * tok::T_TILE));
                }
                Upload *up = u.get();
                const char *b = cuts[i].first;
                up->th = std::thread([c0, up, b, n, &f] { up->ok = h2d_pinned(c0, up->text.p, b, n, f.fd, (uint64_t)(b - f.p)); });
                return MC_OK;
            };
            auto finish = [&](std::unique_ptr<Upload> &u) {
                if (!u) return;
                if (u->th.joinable()) u->th.join();
                std::lock_guard<std::mutex> g0(c0->mu);
                u.reset();
            };
            rc = cuts.empty() ? MC_OK : start_upload(0, cur);
            if (rc) rc = gfail(g, rc, c0->err);
            for (size_t i = 0; i < cuts.size() && rc == MC_OK; i++) {
                cur->th.join();
                if (!cur->ok) { rc = gfail(g, MC_EHIP, "mc_group_add_reads_file: host-to-device copy failed"); break; }
                if (i + 1 < cuts.size()) {
                    rc = start_upload(i + 1, nxt);
                    if (rc) { rc = gfail(g, rc, c0->err); break; }
                }
                uint64_t got = 0;
                bool declined = false;
                {
                    std::lock_guard<std::mutex> g0(c0->mu);
                    rc = tokenize_chunk_locked(c0, f, cuts[i].first, cuts[i].second, cur->text.p, &got, &declined, pend.get());
                    if (rc) rc = gfail(g, rc, c0->err);
                }
                if (rc != MC_OK) break;
                if (declined) {  // (what was gathered goes first: the host's batches are appended behind it)
                    rc = flush();
                    if (rc != MC_OK) break;
                    int hrc = MC_OK;
                    try {
                        got = mch::parse_plain_range(f, cuts[i].first, cuts[i].second, W * per_dev, [&](mch::PackedBatch &b) {
                            if (hrc == MC_OK && b.n_reads()) hrc = group_add_host_batch(g, b.words.data(), b.offsets.data(), b.n_reads());
                        });
                    } catch (...) {
                        finish(cur);
                        finish(nxt);
                        drop_pend();
                        throw;
                    }
                    rc = hrc;
                    if (rc != MC_OK) break;
                }
                total += got;
                if (pend->reads >= W * per_dev) {
                    rc = flush();
                    if (rc != MC_OK) break;
                }
                finish(cur);
                cur = std::move(nxt);
            }
            finish(cur);
            finish(nxt);
            if (rc == MC_OK) rc = flush();
            drop_pend();
            if (rc != MC_OK) return rc;
            if (getenv("MC_INGEST_DEBUG"))
                fprintf(stderr, "[ingest] group: %zu chunk(s) tokenised on device %d, %llu reads in %llu exchange(s) over %zu devices\n", cuts.size(), c0->cfg.device,
                        (unsigned long long)total, (unsigned long long)n_flushes, W);
            if (n_reads) *n_reads = total;
            return MC_OK;
        }
        const uint64_t n = mch::load_reads_file(path, g->ctx.size() * per_dev, [&](mch::PackedBatch &b) {
            if (rc == MC_OK) rc = mc_group_add_reads_packed(g, b.words.data(), b.offsets.data(), b.n_reads());
        });
        if (rc != MC_OK) return rc;
        if (n_reads) *n_reads = n;
        return MC_OK;
    } catch (const mch::Error &e) {
        return gfail(g, MC_EINVAL, e.what());
    } catch (const std::bad_alloc &) {
        return gfail(g, MC_ENOMEM, "mc_group_add_reads_file: out of host memory");
    }
}

int mc_group_finalize_counts(mc_group *g, uint64_t *n_distinct)
{
    if (!g) return MC_EINVAL;
    uint64_t total = 0;
    for (mc_ctx *c : g->ctx) {
        uint64_t n = 0;
        const int rc = mc_finalize_counts(c, &n);
        if (rc) return gfail(g, rc, mc_last_error(c));
        total += n;  // owners are disjoint
    }
    if (n_distinct) *n_distinct = total;
    return MC_OK;
}

int mc_group_get_stats(mc_group *g, mc_stats *out)
{
    if (!g || !out) return MC_EINVAL;
    mc_stats t{};
    for (mc_ctx *c : g->ctx) {
        mc_stats s{};
        const int rc = mc_get_stats(c, &s);
        if (rc) return gfail(g, rc, mc_last_error(c));
        t.windows += s.windows; t.count_launches += s.count_launches; t.table_slots += s.table_slots; t.table_bytes += s.table_bytes;
        t.grows += s.grows; t.spill_keys += s.spill_keys; t.solid_kmers += s.solid_kmers; t.solid_sweeps += s.solid_sweeps;
        t.solid_list_builds += s.solid_list_builds;
        t.long_runs += s.long_runs;
        t.binned_runs += s.binned_runs;
        t.dup_keys += s.dup_keys; t.dup_checks += s.dup_checks; t.dup_ms = std::max(t.dup_ms, s.dup_ms); t.dup_unchecked |= s.dup_unchecked;
        t.left_bins = std::max(t.left_bins, s.left_bins);
        t.count_ms = std::max(t.count_ms, s.count_ms); t.count_total_ms = std::max(t.count_total_ms, s.count_total_ms);
        t.p1_ms = std::max(t.p1_ms, s.p1_ms); t.p2_ms = std::max(t.p2_ms, s.p2_ms); t.p3_ms = std::max(t.p3_ms, s.p3_ms);
    }
    *out = t;
    return MC_OK;
}

int mc_group_bfs_batch(mc_group *g, const mc_bfs_job *jobs, uint32_t n_jobs, int min_cov, int64_t max_kmers, int64_t max_radius,
                       mc_bfs_result *out)
{
    if (!g) return MC_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    const size_t W = g->ctx.size();
    if (W == 1) {
        const int rc = mc_bfs_batch(g->ctx[0], jobs, n_jobs, min_cov, max_kmers, max_radius, out);
        return rc ? gfail(g, rc, mc_last_error(g->ctx[0])) : MC_OK;
    }
    // ---- the walk reads every device's counting table where it is (round 4): the first context gets the other tables' addresses
    // (peer access; the same device for shares of one GPU) and looks a k-mer up in its owner's table.  No export, no copy of
    // the solid k-mers, no second table -- and what made configs[3] impossible: 5 G gathered solid k-mers do not fit beside
    // rank 0's counting table.  MC_GROUP_WALK=gather keeps round 3's way (and a group without peer
    // access between all its devices has no other).
    static const bool want_gather = getenv("MC_GROUP_WALK") && !strcmp(getenv("MC_GROUP_WALK"), "gather");
    if (!want_gather && g->peer_all) {
        mc_ctx *c0 = g->ctx[0];
        static const bool dbg = getenv("MC_INGEST_DEBUG") != nullptr;
        const auto t_0 = std::chrono::steady_clock::now();
        if (g->dirty || !c0->d_shards) {
            std::vector<ShardWire> w(W);
            int grc = per_rank(W, [&](size_t r) -> int {
                std::lock_guard<std::mutex> cl(g->ctx[r]->mu);
                return shard_describe(g->ctx[r], &w[r], false);
            });
            if (grc) {
                for (mc_ctx *c : g->ctx) if (!c->err.empty()) return gfail(g, grc, c->err);
                return gfail(g, grc, "mc_group_bfs_batch: a device's table could not be shared");
            }
            std::lock_guard<std::mutex> cl(c0->mu);
            grc = shard_attach_locked(c0, w.data(), (uint32_t)W, 0, c0->sk_form ? 1 : 0);
            if (grc) return gfail(g, grc, c0->err);
            g->dirty = false;
            g->solid_cov = -1;
        }
        const auto t_1 = std::chrono::steady_clock::now();
        const int rc = mc_bfs_batch(c0, jobs, n_jobs, min_cov, max_kmers, max_radius, out);
        if (dbg) fprintf(stderr, "[walk] group: tables described and attached in %.2f ms, %u walk(s) in %.2f ms\n", std::chrono::duration<double, std::milli>(t_1 - t_0).count(),
                         n_jobs, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_1).count());
        return rc ? gfail(g, rc, mc_last_error(c0)) : MC_OK;
    }
    if (g->dirty || g->solid_cov != min_cov) {
        // ---- gather: every shard's (key, count, pointer) with count >= min_cov, side by side on the first device.  The
        // shards are exported on all devices at once; the copies are then issued on the first device's stream with that
        // device current, and waited for once.
        std::vector<uint64_t> n(W, 0);
        struct Shard { DevBuf<int64_t> k; DevBuf<int16_t> c; DevBuf<uint32_t> p; uint64_t got = 0; };
        std::vector<Shard> sh(W);
        int grc = per_rank(W, [&](size_t r) -> int {
            mc_ctx *c = g->ctx[r];
            if (hipSetDevice(c->cfg.device) != hipSuccess) return MC_EHIP;
            int rc = mc_export_dev(c, min_cov, nullptr, nullptr, nullptr, 0, &n[r]);
            if (rc || n[r] == 0) return rc;
            if (sh[r].k.alloc(n[r]) != hipSuccess || sh[r].c.alloc(n[r]) != hipSuccess || sh[r].p.alloc(n[r]) != hipSuccess) return MC_ENOMEM;
            return mc_export_dev(c, min_cov, sh[r].k.p, sh[r].c.p, sh[r].p.p, n[r], &sh[r].got);
        });
        if (grc) {
            for (mc_ctx *c : g->ctx) if (!c->err.empty()) return gfail(g, grc, c->err);
            return gfail(g, grc, "mc_group_bfs_batch: exporting the shards failed");
        }
        uint64_t total = 0;
        for (size_t r = 0; r < W; r++) total += sh[r].got;
        mc_ctx *c0 = g->ctx[0];
        if (hipSetDevice(c0->cfg.device) != hipSuccess) return gfail(g, MC_EHIP, "hipSetDevice failed");
        DevBuf<int64_t> all_k;
        DevBuf<int16_t> all_c;
        DevBuf<uint32_t> all_p;
        if (all_k.alloc(total) != hipSuccess || all_c.alloc(total) != hipSuccess || all_p.alloc(total) != hipSuccess)
            return gfail(g, MC_ENOMEM, "mc_group_bfs_batch: no room for the gathered shards");
        uint64_t at = 0;
        for (size_t r = 0; r < W; r++) {
            const uint64_t got = sh[r].got;
            if (got == 0) continue;
            if (peer_copy(all_k.p + at, c0, sh[r].k.p, g->ctx[r], got * 8, c0->stream) != hipSuccess || peer_copy(all_c.p + at, c0, sh[r].c.p, g->ctx[r], got * 2, c0->stream) != hipSuccess ||
                peer_copy(all_p.p + at, c0, sh[r].p.p, g->ctx[r], got * 4, c0->stream) != hipSuccess)
                return gfail(g, MC_EHIP, "mc_group_bfs_batch: gathering the shards failed (peer copy)");
            at += got;
        }
        if (hipStreamSynchronize(c0->stream) != hipSuccess) return gfail(g, MC_EHIP, "mc_group_bfs_batch: gathering the shards failed (peer copy)");
        if (hipSetDevice(c0->cfg.device) != hipSuccess) return gfail(g, MC_EHIP, "hipSetDevice failed");
        int rc = mc_clear(g->solid);
        if (!rc) rc = mc_share_read_store(g->solid, c0);
        uint64_t kept = 0;
        if (!rc) rc = mc_solid_from_pairs_dev(g->solid, all_k.p, all_c.p, all_p.p, at, min_cov, &kept);
        if (rc) return gfail(g, rc, mc_last_error(g->solid));
        g->solid_cov = min_cov;
        g->dirty = false;
    }
    const int rc = mc_bfs_batch(g->solid, jobs, n_jobs, min_cov, max_kmers, max_radius, out);
    return rc ? gfail(g, rc, mc_last_error(g->solid)) : MC_OK;
}

}  // extern "C"

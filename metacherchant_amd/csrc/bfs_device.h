// K5: the de Bruijn BFS of src/algo/OneSequenceCalculator.java:154-214 on one workgroup per
// (seed set, direction) job.  All state lives in HBM, so a launch is resumable and the host can
// grow distanceToKmer between launches.
//
// The reference's BFS is strictly sequential: vertex by vertex in queue order, neighbour by
// neighbour in A,G,C,T order, and `maxkmers` is tested at each insertion
// (src/algo/TerminationMode.java:31-47).  What is reproduced exactly is therefore the ORDER:
// candidates get the rank (queue position of the parent, neighbour index); among equal k-mers the
// smallest rank wins; survivors are appended in rank order; the cap cuts that order.
//
// Two ways through a BFS level:
//   wide   -- the frontier has many vertices: chunks of BFS_THREADS candidates, one per thread,
//             de-duplicated in an LDS hash, block scan for the ordered append.
//   narrow -- the frontier is a handful of vertices (the usual case: metagenome graphs are mostly
//             linear, so the BFS is ~10^5 dependent levels of width 1).  A level-per-round-trip
//             walk is bound by HBM latency.  Every k-mer's slot points at one of its occurrences in
//             the read store, and the bases that follow it in that read are the path the walk will
//             most likely take: a SCOUT (one wave per walker) hops from read to read and writes the
//             predicted path down -- ~40-110 levels per two memory round trips -- and the rounds
//             then look up the neighbour sets of up to 128 predicted levels at once and append the
//             leading levels that are exactly what the sequential BFS would have found; anything
//             else falls back to an exact one-level replay (bfs_narrow).
#pragma once
#include "kmer_device.h"

namespace mc {

constexpr int BFS_THREADS = 512;          // 8 waves
constexpr int MAX_NODES = BFS_THREADS;    // neighbour sets looked up per round (levels x walkers x nb): one per thread
constexpr int MAX_DEPTH = 5;
constexpr int SCOUT_MAX_F = 8;                        // walkers that get a scout (one wave each)
constexpr uint32_t PATH_CAP = 1u << 15;               // levels one scout run may predict
constexpr uint32_t PATH_WORDS = (PATH_CAP + 64 + 96) / 32 + 4;  // the walker's own k bases + the levels, 32 bases per word
constexpr uint32_t PSEG_WORDS = 12;                   // path words a round needs: (31 + MAX_NODES / 4 + 63 + 1) bases and one to spare
constexpr uint32_t SCOUT_WORDS = 16;                  // read-store words a hop looks at
constexpr uint32_t SCOUT_MH = 64 + (63 - SK_M) + 4;   // minimizer hashes a hop looks at: 64 new vertices' last SK_M-mers + the ones inside the vertex in front (k - SK_M of them, k <= 63)
constexpr uint32_t SCOUT_BUDGET0 = 1024;              // levels of the first scout run; doubled after every run the rounds used up
#ifndef MC_TEAM_MAX
#define MC_TEAM_MAX 2   // waves of one walker's hop: MC_CAND_MAX candidate reads, each looked at by MC_TEAM_MAX / MC_CAND_MAX waves -- wave s of a
                        // candidate takes the levels 64 s + 1 .. 64 s + 64 past the tip.  Two waves per candidate (-DMC_TEAM_MAX=4: a hop of up
                        // to 128 levels of ONE read) were built and measured in round 4: the hops are cut by the reads' errors and ends, not
                        // by the 64 lanes (mean error-free run at 1 %: 100 bases), and four busy waves make a hop 6.2 us instead of 4.6:
                        // 9.8 ms against 9.4 on configs[1].  Left as an option.
#endif
#ifndef MC_CAND_MAX
#define MC_CAND_MAX 2   // candidate reads a hop follows (measured in round 2 on configs[1], one 64-level wave per candidate: 1 candidate 14.0 ms,
                        // 2: 11.2, 4: 11.8, 8: 14.3)
#endif
#ifndef MC_SEG_MIN
#define MC_SEG_MIN 8    // a hop's second stretch of 64 levels counts only when it holds at least this many (the next hop's candidates are
                        // the read pointers of its last vertices: too few of them and the hop after finds no read to follow)
#endif
constexpr uint32_t SCOUT_BACK = 192;                  // bases in front of a pointer's position a hop's words reach back to (a read that runs against the
                                                      // walk is followed up to 128 + 47 bases backwards)
#ifndef MC_SCOUT_WARM
#define MC_SCOUT_WARM 0   // a hop's lanes touch the read-store lines their pointers name (the next hop starts at one of them)
#endif
#ifndef MC_SCOUT_PROBES
#define MC_SCOUT_PROBES 4   // slots a scout's lookup requests at once
#endif
#ifndef MC_QUAD_MINIMIZER
#define MC_QUAD_MINIMIZER 1   // a round's four nodes of a level share the hashing of their vertex's SK_M-mers (0: every thread all 17)
#endif
#ifndef MC_ROUND_TAIL
#define MC_ROUND_TAIL 4   // slots a round's look-up requests at once when its first two did not decide it (8: measured in round 4, no difference)
#endif
#ifndef MC_ROUND_PROBES
#define MC_ROUND_PROBES 2   // slots a round's look-up requests at once (4 was measured in round 4: the 512 look-ups' wait 3.2 -> 3.5 us a round, the walk 7.7 -> 8.05 ms)
#endif
#ifndef MC_SCOUT_OPTIMISTIC
#define MC_SCOUT_OPTIMISTIC 0   // 1: a scout's lookup that finds neither its key nor a free slot among those takes the vertex for solid
#endif
#ifndef MC_CONS_NEIGHBOURS
#define MC_CONS_NEIGHBOURS 7   // scout_cons: a vertex's pointer is dropped when one of this many vertices nearer to the tip points into the same read
#endif
#ifndef MC_CONS_PAIRS
#define MC_CONS_PAIRS 0   // scout_cons: 1 = the longest agreement of any PAIR of candidate reads is the hop; 0 = a vote a level among all of them
#endif
#ifndef MC_CONS_LEVELS
#define MC_CONS_LEVELS 128   // levels a consensus hop may add: 64 (one stretch, rounds 4-5) or 128 (a second stretch with the same reads, no look-ups in between)
#endif
#ifndef MC_SCOUT_CONSENSUS
#define MC_SCOUT_CONSENSUS 1   // the companion's hops take the path from the CONSENSUS of up to CONS_N reads and look only the last CONS_TAIL
                               // vertices up (scout_cons); 0: round 2's hop -- every level of the better of two reads looked up (scout_eval)
#endif
constexpr uint32_t CONS_N = 8;       // candidate reads of a consensus hop
constexpr uint32_t CONS_WORDS = 12;  // read-store words staged per candidate: 384 bases from word (pos - CONS_BACK) / 32 on
constexpr uint32_t CONS_BACK = 160;  // ... enough for TWO stretches of 64 levels past a tip that sits within CONS_TAIL bases of pos, either way the read runs
                                     // (round 6: a hop that ran its 64 lanes out goes on for a second stretch with the same reads -- their words
                                     // are there, and so are the reads: the second longest of the ~6 that hold the tip reaches ~85 levels past it)
constexpr uint32_t CONS_TAIL = 24;   // levels at the end of a consensus whose vertices are looked up: the tip must be solid, and their read
                                     // pointers are the next hop's candidates (so a candidate's pointer is < CONS_TAIL levels behind its tip)
static_assert(CONS_N * CONS_WORDS == 96, "a lane per staged word, the first 32 lanes a second one");
static_assert(CONS_TAIL + 128 + 31 <= CONS_WORDS * 32 - CONS_BACK - 31 && CONS_TAIL + 128 <= CONS_BACK, "the staged words hold 128 levels either way");
constexpr int NARROW_CAND = 64;           // candidates per replayed level = lanes of one wave
constexpr int RH_SIZE = 1024;             // round-local LDS set (narrow, slow replay)
constexpr int WH_SIZE = 2 * BFS_THREADS;  // chunk-local LDS set (wide)
constexpr uint64_t VIS_EMPTY = ~0ull;
constexpr uint32_t LH_EMPTY = 0xFFFFFFFFu;

enum { BFS_RUNNING = 0, BFS_DONE = 1, BFS_NEED_GROW = 2 };

// ---- debugging aids of the walk (tuning builds only; the product library is built without them) ------------------------
// -DMC_BFS_FUZZ: every workgroup barrier of the walk is followed by a pause that differs from wave to wave and from time
// to time, so that the waves of a workgroup run the stretch behind it far out of step.  What is only ordered by "the other
// waves cannot be that late" breaks within a few walks instead of once in 10^7 (tests/test_gpu_bfs_race.py).
#ifdef MC_BFS_FUZZ
__device__ __forceinline__ void bfs_fuzz(uint32_t point)
{
    uint32_t x = (uint32_t)__builtin_readcyclecounter() ^ (point * 0x9E3779B1u) ^ ((threadIdx.x >> 6) * 0x85EBCA6Bu) ^ (blockIdx.x * 0xC2B2AE35u);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    if ((x & 3u) == 0) {
        const uint32_t n = (x & 0x40u) ? (x >> 8) & 63u : (x >> 8) & 7u;  // mostly a few hundred cycles, now and then a few thousand
        for (uint32_t i = 0; i < n; i++) __builtin_amdgcn_s_sleep(1);
    }
}
#define BFS_FUZZ(p) bfs_fuzz(p)
#else
#define BFS_FUZZ(p) do {} while (0)
#endif
#define BFS_SYNC() do { __syncthreads(); BFS_FUZZ(__LINE__); } while (0)
// Four places of the walk where round 3 let a wave decide about barriers from words another wave was about to overwrite
// (DESIGN.md section 3.3).  The product has the fixed forms below; round 3's forms live in csrc/test/bfs_old_race.h, which only a
// build with -DMC_BFS_OLD_RACE (tests/test_gpu_bfs_race.py: the fuzzed old kernel must fail) pulls in.
#ifdef MC_BFS_OLD_RACE
#include "test/bfs_old_race.h"
#else
#define BFS_DECIDED_SYNC() BFS_SYNC()
#define BFS_PATH_RESET_EARLY(L, tid) do {} while (0)
#define BFS_PATH_RESET(L, tid) do { if ((tid) < SCOUT_MAX_F) { (L).plen[tid] = 0; (L).ppos[tid] = 0; (L).pdone[tid] = 0; } } while (0)
#define BFS_CHUNK_LOOP(c0) for (unsigned long long c0 = ctl_ld(&ctl->c0);; c0 += BFS_THREADS)
#endif

// -DMC_BFS_TRACE: a ring of 8-word records per job, one per narrow round (tid 0) and one per run of the companion; the host
// dumps it when the self-check of a walk (k_bfs_check) finds something, or always with MC_BFS_TRACE_DUMP=<file>.
constexpr uint32_t BFS_TRACE_RECORDS = 1u << 16;
#ifdef MC_BFS_TRACE
#define BFS_TRACE(S, w0, w1, w2, w3, w4, w5, w6, w7)                                                              \
    do {                                                                                                          \
        if ((S).trace) {                                                                                          \
            const unsigned long long ti_ = atomicAdd(&(S).ctl->trace_n, 1ull) & (BFS_TRACE_RECORDS - 1);          \
            uint4 *tp_ = reinterpret_cast<uint4 *>((S).trace + 8 * ti_);                                          \
            tp_[0] = make_uint4((uint32_t)(w0), (uint32_t)(w1), (uint32_t)(w2), (uint32_t)(w3));                  \
            tp_[1] = make_uint4((uint32_t)(w4), (uint32_t)(w5), (uint32_t)(w6), (uint32_t)(w7));                  \
        }                                                                                                         \
    } while (0)
#else
#define BFS_TRACE(S, w0, w1, w2, w3, w4, w5, w6, w7) do {} while (0)
#endif

struct BfsCtl {
    unsigned long long n;       // |distanceToKmer|
    unsigned long long lb, le;  // current frontier = entries [lb, le)
    unsigned long long c0;      // next candidate rank inside the frontier (wide path)
    unsigned long long lookups;
    unsigned long long rounds_narrow, rounds_slow, chunks_wide, scout_hops, scout_levels, scout_calls, scout_nf, scout_m0, slow_mismatch, slow_starved, slow_forced;
    unsigned long long tacc[8];  // MC_BFS_TIMING builds: 10 ns ticks per phase of a narrow round
    unsigned long long trace_n;  // MC_BFS_TRACE builds: records written to BfsState::trace so far
    long long level;            // distance of the frontier
    int status;
    int seeds_done;
};

struct ScoutBox;
struct BfsState {
    uint64_t *hi, *lo;  // distanceToKmer keys in insertion order
    int32_t *dist;
    int16_t *cov;
    uint32_t *flags;    // bit0: in lastKmers; bit1: seed window queued more than once.  Pre-zeroed.
    uint64_t dcap;
    uint64_t *vis;      // index of the arrays above: buckets of two (fingerprint << 32 | index) entries
    uint64_t bmask;     // number of buckets - 1
    BfsCtl *ctl;
    uint64_t *path;     // SCOUT_MAX_F * PATH_WORDS words: the predicted paths of the walkers (scout_run)
    ScoutBox *box;      // mailbox between this job's workgroup and its scouting companion (nullptr: none)
    uint32_t *trace;    // MC_BFS_TRACE builds: BFS_TRACE_RECORDS records of 8 words (nullptr: none)
    const uint64_t *seed_hi, *seed_lo;
    uint64_t n_seeds;
    int dir;
};

// The walk state in BfsCtl is written by thread 0 and read by every thread behind a barrier.  Those reads decide what the
// whole workgroup does next, so every wave must see the same value: they are L1-bypassing loads (a wave that read a
// stale `c0` -- seen once the struct had grown over more cache lines -- expands other parents than its neighbours and,
// were it `lb` / `le`, would wait at a different barrier for good), and the stores write through.
template <typename T>
__device__ __forceinline__ T ctl_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T, typename V>
__device__ __forceinline__ void ctl_st(T *p, V v) { __hip_atomic_store(p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint64_t ld_sc1(const uint64_t *p)
{  // L1-bypassing load: data written earlier in this launch by an atomic or by another wave's store
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t ld_flags(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

// (hi == 0, every k <= 32: the inner hash is a constant)
__device__ __forceinline__ uint64_t vis_hash(const Kmer &v)
{
    constexpr uint64_t H0 = fmix64(0x9e3779b97f4a7c15ull);
    return fmix64(v.lo ^ (v.hi == 0 ? H0 : fmix64(v.hi + 0x9e3779b97f4a7c15ull)));
}

__device__ __forceinline__ bool vis_entry_is(const BfsState &S, uint64_t e, uint32_t fp, const Kmer &v)
{
    if ((uint32_t)(e >> 32) != fp) return false;
    const uint32_t idx = (uint32_t)e;
    return ld_sc1(&S.lo[idx]) == v.lo && ld_sc1(&S.hi[idx]) == v.hi;
}

// Is the oriented k-mer in distanceToKmer?  e0/e1 = the two entries of its home bucket (already loaded).
__device__ __forceinline__ bool vis_contains(const BfsState &S, const Kmer &v, uint64_t h, uint64_t e0, uint64_t e1)
{
    const uint32_t fp = (uint32_t)(h >> 32);
    uint64_t b = h & S.bmask;
    for (uint64_t probe = 0; probe <= S.bmask; probe++) {
        if (probe) {
            e0 = ld_sc1(&S.vis[2 * b]);
            e1 = ld_sc1(&S.vis[2 * b + 1]);
        }
        if (e0 == VIS_EMPTY) return false;
        if (vis_entry_is(S, e0, fp, v)) return true;
        if (e1 == VIS_EMPTY) return false;
        if (vis_entry_is(S, e1, fp, v)) return true;
        b = (b + 1) & S.bmask;
    }
    return false;
}

__device__ __forceinline__ bool vis_find(const BfsState &S, const Kmer &v)
{
    const uint64_t h = vis_hash(v);
    const uint64_t b = h & S.bmask;
    return vis_contains(S, v, h, ld_sc1(&S.vis[2 * b]), ld_sc1(&S.vis[2 * b + 1]));
}

// Index of a k-mer known to be present (seed bookkeeping only).
__device__ __forceinline__ long long vis_index_of(const BfsState &S, const Kmer &v)
{
    const uint64_t h = vis_hash(v);
    const uint32_t fp = (uint32_t)(h >> 32);
    uint64_t b = h & S.bmask;
    for (uint64_t probe = 0; probe <= S.bmask; probe++) {
        for (int i = 0; i < 2; i++) {
            const uint64_t e = ld_sc1(&S.vis[2 * b + i]);
            if (e == VIS_EMPTY) return -1;
            if (vis_entry_is(S, e, fp, v)) return (long long)(uint32_t)e;
        }
        b = (b + 1) & S.bmask;
    }
    return -1;
}

// Insert a k-mer known to be absent (callers de-duplicate first).  Entries are never removed, so a
// bucket fills front to back and "first entry empty" means the whole bucket is empty.
__device__ __forceinline__ void vis_insert(const BfsState &S, const Kmer &v, uint32_t idx)
{
    const uint64_t h = vis_hash(v);
    const uint64_t e = (h & 0xFFFFFFFF00000000ull) | idx;
    uint64_t b = h & S.bmask;
    for (uint64_t probe = 0; probe <= S.bmask; probe++) {
        for (int i = 0; i < 2; i++)
            if (atomicCAS(reinterpret_cast<unsigned long long *>(&S.vis[2 * b + i]), (unsigned long long)VIS_EMPTY,
                          (unsigned long long)e) == VIS_EMPTY)
                return;
        b = (b + 1) & S.bmask;
    }
}

// LDS set of candidate ids keyed by their k-mer; the smallest id of each k-mer stays.
// Returns the slot the caller's k-mer lives in.
__device__ __forceinline__ uint32_t lds_set_min(uint32_t *tab, uint32_t mask, const Kmer *kmers, const Kmer &v,
                                                uint32_t id)
{
    uint32_t s = (uint32_t)vis_hash(v) & mask;
    for (;;) {
        uint32_t cur = tab[s];
        if (cur == LH_EMPTY) {
            cur = atomicCAS(&tab[s], LH_EMPTY, id);
            if (cur == LH_EMPTY) return s;
        }
        const Kmer o = kmers[cur];  // an occupant is only ever replaced by a smaller id of the SAME k-mer
        if (o.lo == v.lo && o.hi == v.hi) {
            atomicMin(&tab[s], id);
            return s;
        }
        s = (s + 1) & mask;
    }
}

// block-wide exclusive scan of one flag per thread; *total = number of set flags
__device__ __forceinline__ uint32_t block_scan_flag(bool flag, uint32_t *lds_wave_tot, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1));
    if (lane == 0) lds_wave_tot[wv] = (uint32_t)__popcll(m);
    BFS_SYNC();
    uint32_t before = 0, tot = 0;
    const uint32_t n_waves = blockDim.x >> 6;
    for (uint32_t i = 0; i < n_waves; i++) {
        const uint32_t c = lds_wave_tot[i];
        if (i < wv) before += c;
        tot += c;
    }
    BFS_SYNC();
    *total = tot;
    return before + in_wave;
}

struct WideLds {
    Kmer kmer[BFS_THREADS];
    uint32_t set[WH_SIZE];
    uint32_t widx[BFS_THREADS];
    uint32_t wave_tot[BFS_THREADS / 64];
};

struct NarrowLds {
    Kmer kmer[MAX_NODES];
    Kmer pk[MAX_NODES];               // the vertices the walkers are expected to visit, [(level-1)*F + walker]
    Kmer root[NARROW_CAND];           // the walkers
    Kmer pub_k[MAX_NODES];            // accepted in the last round, to be indexed
    uint32_t pub_idx[MAX_NODES];
    uint32_t wptr[NARROW_CAND];       // read pointer stored with each walker's k-mer (0: none / not looked up yet)
    uint8_t wright[NARROW_CAND];      // the walker moves right (1) or left (0)
    uint32_t plen[SCOUT_MAX_F], ppos[SCOUT_MAX_F];  // predicted path of walker a: levels written / levels used up
    uint64_t pseg[SCOUT_MAX_F][PSEG_WORDS];         // the path words this round reads, from word ppos / 32 on
    uint64_t sw[SCOUT_MAX_F][SCOUT_WORDS];          // scout: the piece of the read store a hop looks at
    uint32_t mh[SCOUT_MAX_F][SCOUT_MH];             //        and the minimizer hashes along it
    uint32_t naux[MAX_NODES];         // read pointer found with each node's k-mer
    int16_t cov[MAX_NODES];
    uint8_t vis[MAX_NODES];
    uint32_t set[RH_SIZE];
    uint32_t fl_w[2][NARROW_CAND];    // frontier lists: index of the node inside its tree level
    uint32_t fl_idx[2][NARROW_CAND];  //                 index in distanceToKmer (bit31: re-queued seed)
    uint32_t new_id[NARROW_CAND * MAX_DEPTH];   // accepted this round: tree node
    uint32_t new_idx[NARROW_CAND * MAX_DEPTH];  //                      index in distanceToKmer
    // walk state, owned by wave 0, read by everyone after a barrier
    unsigned long long n, lb, le, rounds_left;
    long long level;
    uint32_t F;
    uint32_t bad_lvl, pend;
    uint32_t scout_budget, scout_skip, scout_wait, force_slow;
    uint32_t comp, req_seq, req_open, ready, root_bad;  // the companion (ScoutBox): usable, the last request, one is outstanding, its answer is in
    uint32_t pdone[SCOUT_MAX_F];              // the companion finished walker a's path
    unsigned long long hops, hop_levels, s_calls, s_nf, s_m0;
    int cur, status, any_dup_root;
};


// ---- wide path: one chunk of <= BFS_THREADS candidates in rank order.  parent == UINT64_MAX marks a
// seed window (src/algo/OneSequenceCalculator.java:159-192: queued when reads.get(key) >= minOccurences).
template <int MODE>
// (forceinline, like the walk's other parts: called out of line -- it has two call sites -- it takes S and t by reference, which puts
// both into scratch memory for the WHOLE kernel: every t.slots / t.reads then costs a scratch load)
__device__ __forceinline__ void bfs_chunk_wide(const BfsState &S, const SolidView &t, WideLds &L, int k, int min_cov,
                               long long max_kmers, bool radius_ok, bool have, const Kmer &cand, uint64_t parent,
                               int32_t new_dist, unsigned long long &lookups)
{
    BfsCtl *ctl = S.ctl;
    const uint32_t tid = threadIdx.x;
    const bool is_seed = parent == UINT64_MAX;
    L.kmer[tid] = cand;
    L.set[tid] = LH_EMPTY;
    L.set[tid + BFS_THREADS] = LH_EMPTY;
    int cov = -1;
    if (have) {
        cov = solid_get_kmer<MODE>(t, cand, k, (uint64_t)key_of<MODE>(cand, k));
        lookups++;
    }
    const bool solid = have && cov >= min_cov;
    const unsigned long long n_before = ctl_ld(&ctl->n);
    const bool capped = max_kmers >= 0 && (long long)n_before >= max_kmers;
    bool mark_last = false, contender = false;
    if (solid) {
        if (!is_seed && (ld_flags(&S.flags[parent]) & 2u)) mark_last = true;  // re-queued seed window: nothing is new to it
        if (!is_seed && (capped || !radius_ok)) {
            mark_last = true;  // allowsAddition() == false -> lastKmers.add(kmer)
        } else if (vis_find(S, cand)) {
            if (is_seed) {
                const long long f = vis_index_of(S, cand);
                if (f >= 0) atomicOr(&S.flags[f], 2u);
            } else {
                mark_last = true;
            }
        } else {
            contender = true;
        }
    }
    BFS_SYNC();
    uint32_t slot = 0;
    if (contender) slot = lds_set_min(L.set, WH_SIZE - 1, L.kmer, cand, tid);
    BFS_SYNC();
    const bool winner = contender && L.set[slot] == tid;
    if (contender && !winner && !is_seed) mark_last = true;  // an earlier rank inserts it first
    uint32_t total;
    const uint32_t pos = block_scan_flag(winner, L.wave_tot, &total);
    bool accepted = false;
    uint64_t idx = 0;
    if (winner) {
        idx = n_before + pos;
        accepted = is_seed || max_kmers < 0 || (long long)idx < max_kmers;  // distanceToKmer.size() >= threshold
        if (accepted) {
            S.hi[idx] = cand.hi;
            S.lo[idx] = cand.lo;
            S.dist[idx] = new_dist;
            S.cov[idx] = (int16_t)cov;
            vis_insert(S, cand, (uint32_t)idx);
        } else {
            mark_last = true;
        }
    }
    if (mark_last && !is_seed) atomicOr(&S.flags[parent], 1u);
    if (winner) L.widx[tid] = accepted ? (uint32_t)idx : LH_EMPTY;
    uint32_t n_acc;
    (void)block_scan_flag(accepted, L.wave_tot, &n_acc);  // (its barriers also publish widx)
    if (is_seed && contender && !winner) {  // the same seed window twice inside this chunk: it is re-queued
        const uint32_t wi = L.widx[L.set[slot]];
        if (wi != LH_EMPTY) atomicOr(&S.flags[wi], 2u);
    }
    if (tid == 0) ctl_st(&ctl->n, n_before + n_acc);
    BFS_SYNC();
}

// number of tree nodes under ONE root down to depth d: nb + nb^2 + ... + nb^d
__device__ __forceinline__ uint32_t tree_size(int nb, int d)
{
    uint32_t s = 0, p = 1;
    for (int j = 1; j <= d; j++) { p *= (uint32_t)nb; s += p; }
    return s;
}

#ifdef MC_BFS_TIMING
#define MC_STAMP(i)                                                            \
    do {                                                                       \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();      \
        if ((i) > 0) tacc[(i) > 0 ? (i)-1 : 0] += now_ - tlast;                \
        tlast = now_;                                                          \
    } while (0)
#else
#define MC_STAMP(i) do {} while (0)
#endif

// Exact level-by-level replay of one speculated round (wave 0).  Returns the number of accepted
// vertices; updates n/lb/le/level/cur/F in L.
__device__ __forceinline__ uint32_t replay_slow(const BfsState &S, NarrowLds &L, int d, int min_cov, long long max_kmers,
                                       long long max_radius, uint32_t lg, uint32_t flim, uint32_t *last_base)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nb = 1u << lg;
    unsigned long long n = L.n, lb = L.lb, le = L.le;
    long long level = L.level;
    int cur = L.cur;
    const uint32_t F = L.F;
    for (uint32_t i = lane; i < RH_SIZE; i += 64) L.set[i] = LH_EMPTY;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    uint32_t n_new = 0;  // accepted this round (uniform)
    uint32_t Fj = F;     // frontier of the level being expanded
    uint32_t base = 0, width = F << lg;
    for (int j = 1; j <= d; j++) {
        const uint32_t ncand = Fj << lg;  // <= 64
        const bool have = lane < ncand;
        const uint32_t p = lane >> lg, c = lane & (nb - 1);
        uint32_t id = 0, pidx = 0;
        bool pdup = false;
        int cov = -1;
        bool solid = false, gvis = false;
        Kmer v{0, 0};
        if (have) {
            const uint32_t pw = L.fl_w[cur][p];
            const uint32_t pi = L.fl_idx[cur][p];
            pidx = pi & 0x3FFFFFFFu;
            pdup = (pi & 0x80000000u) != 0;
            id = base + (pw << lg) + c;
            cov = L.cov[id];
            gvis = L.vis[id] != 0;
            v = L.kmer[id];
            solid = cov >= min_cov;
        }
        const bool capped = max_kmers >= 0 && (long long)n >= max_kmers;
        const bool radius_ok = max_radius < 0 || level + 1 <= max_radius;
        bool mark_last = false, contender = false;
        if (solid) {
            if (pdup) mark_last = true;
            if (capped || !radius_ok || gvis) mark_last = true;
            else contender = true;
        }
        uint32_t slot = 0;
        if (contender) slot = lds_set_min(L.set, RH_SIZE - 1, L.kmer, v, id);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const bool winner = contender && L.set[slot] == id;  // ids grow with (level, rank): earlier levels win
        if (contender && !winner) mark_last = true;
        const unsigned long long wm = __ballot(winner);
        const uint32_t pos = (uint32_t)__popcll(wm & ((1ull << lane) - 1));
        bool accepted = false;
        uint64_t idx = 0;
        if (winner) {
            idx = n + pos;
            accepted = max_kmers < 0 || (long long)idx < max_kmers;
            if (!accepted) mark_last = true;
        }
        const unsigned long long am = __ballot(accepted);
        const uint32_t n_acc = (uint32_t)__popcll(am);
        if (accepted) {
            const uint32_t apos = (uint32_t)__popcll(am & ((1ull << lane) - 1));  // == pos: the cap cuts a prefix
            S.hi[idx] = v.hi;
            S.lo[idx] = v.lo;
            S.dist[idx] = (int32_t)(level + 1);
            S.cov[idx] = (int16_t)cov;
            L.new_id[n_new + apos] = id;
            L.new_idx[n_new + apos] = (uint32_t)idx;
            L.fl_w[cur ^ 1][apos] = id - base;
            L.fl_idx[cur ^ 1][apos] = (uint32_t)idx;
        }
        if (mark_last) atomicOr(&S.flags[pidx], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        n += n_acc;
        n_new += n_acc;
        lb = le;
        le = n;
        level++;
        cur ^= 1;
        Fj = n_acc;
        base += width;
        width <<= lg;
        if (Fj == 0 || Fj > flim) break;
    }
    *last_base = base - (width >> lg);
    if (lane == 0) {
        L.n = n; L.lb = lb; L.le = le; L.level = level; L.cur = cur; L.F = Fj;
    }
    return n_new;
}

// lookup with the first two probe slots already loaded; *aux = the read pointer stored with the key (0 when absent)
__device__ __forceinline__ int solid_get2(const SolidView &tv, const TableRef &t, uint64_t key, uint64_t s0, uint4 a0, uint4 a1,
                                          uint32_t *aux)
{   // t: the table the key lives in (kmer_device.h solid_locate), s0 its home slot there; tv: the walk's view (the out-of-band key)
    slots_wanted(a0, a1);  // (every word of both slots is used from here on: the loads stay whole, see there)
    *aux = 0;
    if (key == EMPTY_KEY) return solid_get(tv, key);
    const uint64_t k0 = ((uint64_t)a0.y << 32) | a0.x;
    if (k0 == key) { *aux = a0.w; return a0.z > 32767u ? 32767 : (int)a0.z; }
    if (k0 == EMPTY_KEY) return -1;
    const uint64_t k1 = ((uint64_t)a1.y << 32) | a1.x;
    if (k1 == key) { *aux = a1.w; return a1.z > 32767u ? 32767 : (int)a1.z; }
    if (k1 == EMPTY_KEY) return -1;
    // both probes hit other keys: the rest of the sequence, four slots at a time
    return solid_probe_from<MC_ROUND_TAIL>(t, key, (s0 & ~(uint64_t)t.rmask) | ((s0 + 2) & t.rmask), 2, aux);
}

// ... with four probe slots already loaded (MC_ROUND_PROBES == 4: measured, not the default)
__device__ __forceinline__ int solid_get4l(const SolidView &tv, const TableRef &t, uint64_t key, uint64_t s0, uint4 a0, uint4 a1, uint4 a2, uint4 a3,
                                           uint32_t *aux)
{
    slots_wanted(a0, a1, a2, a3);
    *aux = 0;
    if (key == EMPTY_KEY) return solid_get(tv, key);
    const uint4 av[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t cur = ((uint64_t)av[i].y << 32) | av[i].x;
        if (cur == key) { *aux = av[i].w; return av[i].z > 32767u ? 32767 : (int)av[i].z; }
        if (cur == EMPTY_KEY) return -1;
    }
    return solid_probe_from(t, key, (s0 & ~(uint64_t)t.rmask) | ((s0 + 4) & t.rmask), 4, aux);
}

// 32 bases from base q of a packed word array on, first base on top (reads one word past the last one it needs)
__device__ __forceinline__ uint64_t bases32(const uint64_t *w, uint32_t q)
{
    const uint32_t wi = q >> 5, off = 2 * (q & 31);
    const uint64_t w0 = w[wi], w1 = w[wi + 1];
    return off ? ((w0 << off) | (w1 >> (64 - off))) : w0;
}

// The 32 bases that END right before base q (q may be < 32: the missing ones read as garbage), reverse-complemented:
// the complement of base q-1 on top, then q-2, ...
__device__ __forceinline__ uint64_t bases32_before_rc(const uint64_t *w, uint32_t q)
{
    const uint64_t x = q >= 32 ? bases32(w, q - 32) : (q ? bases32(w, 0) >> (2 * (32 - q)) : 0ull);
    return rc64_pairs(x);
}

__device__ __forceinline__ bool kmer_eq(const Kmer &a, const Kmer &b) { return a.lo == b.lo && a.hi == b.hi; }

// a value every lane holds alike, moved to scalar registers
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// the k-mer at base q of a packed word array / its reverse complement; packed keys (k <= 31) live in one word
template <int MODE>
__device__ __forceinline__ Kmer kmer_at(const uint64_t *w, uint64_t q, int k)
{
    if (MODE == KEY_PACKED) return Kmer{0, bases32(w, (uint32_t)q) >> (64 - 2 * k)};
    return extract_kmer(w, q, k);
}
template <int MODE>
__device__ __forceinline__ Kmer kmer_rc(const Kmer &v, int k)
{
    if (MODE == KEY_PACKED) return Kmer{0, rc_packed(v.lo, k)};
    return rc_kmer(v, k);
}

// lookup with the first four probe slots requested at once: the lanes of a wave look different keys up, and the
// slowest one decides -- at load 1/4 one in ~8 lookups needs a second probe, nearly none a fifth
__device__ __forceinline__ int solid_get4(const SolidView &tv, const TableRef &t, uint64_t key, uint32_t *aux, uint64_t s0)
{   // t, s0: where the key lives (solid_locate); tv: the walk's view
    *aux = 0;
    if (key == EMPTY_KEY) return solid_get(tv, key);
    const uint64_t base = s0 & ~(uint64_t)t.rmask;
    constexpr int NP = MC_SCOUT_PROBES;
    uint4 a[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) a[i] = *reinterpret_cast<const uint4 *>(t.slots + (base | ((s0 + i) & t.rmask)));
    if (NP == 4) slots_wanted(a[0], a[1], a[NP > 2 ? 2 : 0], a[NP > 3 ? 3 : 0]);  // (kmer_device.h: one round trip, as written)
    else if (NP == 2) slots_wanted(a[0], a[1]);
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const uint64_t cur = ((uint64_t)a[i].y << 32) | a[i].x;
        if (cur == key) { *aux = a[i].w; return a[i].z > 32767u ? 32767 : (int)a[i].z; }
        if (cur == EMPTY_KEY) return -1;
    }
#if MC_SCOUT_OPTIMISTIC
    // None of the NP slots held the key and none was free: the key sits further along (1 % of the solid k-mers at the table's
    // load) or is not there.  A scout only guesses: it takes the vertex for solid (without a pointer) and saves the second
    // round trip that the slowest of its 64 lanes would make the whole hop wait for (every other hop has such a lane); the
    // walk's rounds look every vertex up properly.
    return 32767;
#else
    return solid_probe_from(t, key, base | ((s0 + NP) & t.rmask), NP, aux);
#endif
}

// ---- one hop of a scout -------------------------------------------------------------------------------------------
// What one candidate read adds to a walker's predicted path (one wave; every lane calls it).
struct HopEval {
    uint32_t m;           // (uniform) levels added; 0: the candidate failed
    int why;              // (uniform) 0 ok, 1 the tip is not where the pointer says, 2 nothing solid behind it
    bool fwd;             // (uniform) the read runs the walker's way
    long long Q;          // (uniform) where the tip sits in the read store
    uint64_t e_hi, e_lo;  // (uniform) the m new bases, first on top
    Kmer K;               // (lane) the vertex lane + 1 levels past the tip, walk strand
    uint32_t aux;         // (lane) its read pointer
    bool other;           // (lane) ... which leads into another read than this one
    // n_cand > 0 was asked for: the next hop's candidates -- the pointers nearest to the tip that lead into other reads, one
    // per read -- and the read store's words around them, requested as soon as they are known (lanes 16 i .. 16 i + 15: the
    // words of candidate i, see scout_word_of)
    uint32_t nc, cptr[MC_CAND_MAX], cdelta[MC_CAND_MAX];  // (uniform)
    uint64_t word;                                         // (lane)
};

// word i < SCOUT_WORDS of the piece of the read store a hop looks at for pointer `cptr` (what scout_eval stages in sw[i])
__device__ __forceinline__ uint64_t scout_word_of(const SolidView &t, uint32_t cptr, uint32_t i)
{
    uint32_t span;
    const uint64_t lo = ptr_decode(cptr, &span);
    if (lo >= t.reads_bases) return 0;
    const uint64_t wlo = (lo > SCOUT_BACK ? lo - SCOUT_BACK : 0) >> 5, last_word = (t.reads_bases + 31) / 32;
    return t.reads[min(wlo + i, last_word)];
}

// sw, mh: this wave's LDS scratch (SCOUT_WORDS words of the read store; SCOUT_MH minimizer hashes)
template <int MODE, bool SH = false>
__device__ __forceinline__ void scout_eval(const SolidView &t, uint64_t *sw, uint32_t *mh, const Kmer &X, int k, int min_cov, uint32_t cptr,
                                           uint32_t delta, uint32_t want, HopEval &R, unsigned long long &lookups, unsigned long long *tsc = nullptr,
                                           bool staged = false, uint32_t n_cand = 0, uint32_t lane_base = 0)
{   // staged: sw already holds the SCOUT_WORDS words around the pointer (scout_words_of: the companion requests them a hop ahead)
    // lane_base: this wave looks at the vertices lane_base + 1 .. lane_base + 64 levels past the tip (the second wave of a candidate: 64)
#ifdef MC_SCOUT_TIMING
    unsigned long long ts_ = __builtin_amdgcn_s_memrealtime();
#define SC_STAMP(i) do { if (tsc) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[i] += n_ - ts_; ts_ = n_; } } while (0)
#else
#define SC_STAMP(i) do {} while (0)
    (void)tsc;
#endif
    const uint32_t lane = threadIdx.x & 63;
    R.m = 0; R.why = 1; R.fwd = true; R.Q = -1; R.e_hi = R.e_lo = 0; R.K = Kmer{0, 0}; R.aux = 0; R.other = false;
    R.nc = 0; R.word = 0;
#pragma unroll
    for (int i = 0; i < MC_CAND_MAX; i++) { R.cptr[i] = 0; R.cdelta[i] = 0; }
    uint32_t span;
    const uint64_t lo = ptr_decode(cptr, &span);
    if (lo >= t.reads_bases) return;  // (a pointer from elsewhere)
    const uint64_t last_word = (t.reads_bases + 31) / 32;  // the pad word
    // the piece of the read store around the occurrence
    const uint64_t wlo = (lo > SCOUT_BACK ? lo - SCOUT_BACK : 0) >> 5;
    if (!staged) {
        __builtin_amdgcn_wave_barrier();
        if (lane < SCOUT_WORDS) sw[lane] = t.reads[min(wlo + lane, last_word)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    SC_STAMP(0);
    const uint64_t base = wlo * 32;
    const Kmer Xr = kmer_rc<MODE>(X, k);
    // where is the tip?  `delta` bases after the pointer's k-mer in a read that runs our way, before it otherwise
    long long Q = -1;
    bool fwd = true;
    if (span == 1) {  // an exact pointer (the first 2^31 bases of the store): two places to look at, no search
        const uint64_t qf = lo + delta;
        const bool of = qf + (uint64_t)k <= t.reads_bases, orv = lo >= delta && lo - delta + (uint64_t)k <= t.reads_bases;
        const bool mf = of && kmer_eq(kmer_at<MODE>(sw, qf - base, k), X);
        const bool mr = !mf && orv && kmer_eq(kmer_at<MODE>(sw, lo - delta - base, k), Xr);
        if (mf) Q = (long long)qf;
        else if (mr) { Q = (long long)(lo - delta); fwd = false; }
        fwd = __builtin_amdgcn_readfirstlane((int)fwd) != 0;  // (every lane computed the same)
    } else
    for (uint32_t o0 = 0; o0 < span && Q < 0; o0 += 64) {
        const uint32_t o = o0 + lane;
        const uint64_t qf = lo + delta + o, qr = lo + o - delta;
        const bool of = o < span && qf + (uint64_t)k <= t.reads_bases, orv = o < span && lo + o >= delta && qr + (uint64_t)k <= t.reads_bases;
        const bool mf = of && kmer_eq(kmer_at<MODE>(sw, qf - base, k), X);
        const bool mr = orv && kmer_eq(kmer_at<MODE>(sw, qr - base, k), Xr);
        const unsigned long long bf = __ballot(mf), br = __ballot(mr);
        if (bf) { Q = (long long)(lo + delta + o0 + (uint32_t)__builtin_ctzll(bf)); fwd = true; }
        else if (br) { Q = (long long)(lo + o0 + (uint32_t)__builtin_ctzll(br)) - (long long)delta; fwd = false; }
    }
    if (Q < 0) return;
    Q = (long long)uni64((uint64_t)Q);  // (wave-uniform: what follows from it can run on the scalar unit)
    R.Q = Q; R.fwd = fwd; R.why = 2;
    // lane i: the vertex lane_base + i + 1 levels past the tip
    const uint32_t lv = lane_base + lane;
    Kmer K{0, 0};
    bool ok = lane < want;
    if (fwd) {
        const uint64_t pi = (uint64_t)Q + 1 + lv;
        ok = ok && pi + (uint64_t)k <= t.reads_bases && pi + (uint64_t)k <= (base + 32ull * (SCOUT_WORDS - 1));  // (inside the words at hand)
        if (ok) K = kmer_at<MODE>(sw, pi - base, k);
    } else {
        ok = ok && (uint64_t)Q >= (uint64_t)lv + 1 + base;
        if (ok) K = kmer_rc<MODE>(kmer_at<MODE>(sw, (uint64_t)Q - 1 - lv - base, k), k);
    }
    // the vertex in front of this wave's first one (the tip itself for the first wave of a candidate): its SK_M-mers are the
    // ones the first lanes' windows reach back into
    Kmer Xs = X;
    if (lane_base) {
        const bool okx = fwd ? (uint64_t)Q + lane_base + (uint64_t)k <= t.reads_bases : (uint64_t)Q >= (uint64_t)lane_base + base;
        if (!okx) return;  // (uniform: the read ends before this wave's stretch begins)
        Xs = fwd ? kmer_at<MODE>(sw, (uint64_t)Q + lane_base - base, k) : kmer_rc<MODE>(kmer_at<MODE>(sw, (uint64_t)Q - lane_base - base, k), k);
    }
    uint32_t aux = 0;
    int cov = -1;
    const uint64_t key = ok ? (uint64_t)key_of<MODE>(K, k) : 0;
    uint64_t s0;
    // h: the table this lane's vertex lives in -- SH (several GPUs): its owner's; else the walker's own (uniform values)
    TableRef h;
    if (MODE == KEY_PACKED && (SH ? t.owner_mm_k : t.mm_k)) {
        // The counting table's regions are minimizer bins (kmer_device.h).  The vertices of consecutive lanes overlap in
        // all but one base, so every SK_M-mer is hashed once -- a lane hashes the last one of its own vertex, the first
        // lanes also the ones inside the tip -- and a lane takes the minimum over its w of them (w = k - SK_M + 1).
        const uint32_t w = (uint32_t)k - SK_M + 1;
        auto mm_hash = [](uint32_t f) {  // sk_order of the canonical SK_M-mer (sk_rc_mmer on 32 bits: this is one lone wave's time)
            uint32_t r = __builtin_bitreverse32(f);
            r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
            r = (~r) >> (32 - 2 * SK_M);
            return sk_order(f < r ? f : r);
        };
        __builtin_amdgcn_wave_barrier();
        if (lane + 1 < w) mh[lane] = mm_hash((uint32_t)(Xs.lo >> (2 * (w - 2 - lane))) & SK_MMASK);  // the SK_M-mers of the vertex in front, from base lane + 1 on
        mh[w - 1 + lane] = ok ? mm_hash((uint32_t)K.lo & SK_MMASK) : SK_NONE;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t hmin = SK_NONE;
        if (w == 17) {  // k = 31: the reads requested together
            uint32_t hv[17];
#pragma unroll
            for (int i = 0; i < 17; i++) hv[i] = mh[lane + i];
#pragma unroll
            for (int i = 0; i < 17; i++) hmin = min(hmin, hv[i]);
        } else {
            for (uint32_t i = 0; i < w; i++) hmin = min(hmin, mh[lane + i]);
        }
        if (SH) s0 = solid_locate(t, key, h, true, hmin);
        else { h = own_table(t); s0 = ((((uint64_t)sk_bin(hmin) * t.n_regions) >> 32) << MC_REGION_LG) | sk_home(key); }
    } else if (MODE != KEY_PACKED && !SH && t.mm_k != 0) {
        // Hash keys in minimizer bins (count_long.h): the bin comes from the vertex's BASES, 19 .. 49 SK_M-mers each.  The same as above for
        // k-mers of two words: every SK_M-mer of the hop hashed once (a lane worked all 49 of its vertex out, solid_locate_kmer: most of
        // a hop's time at k = 63), a lane takes the smallest -- or the two smallest, as a multiset: mm_k < 0 -- of its w.
        const uint32_t w = (uint32_t)k - SK_M + 1, wq = w - 1;
        auto mm_hash = [](uint32_t f) {
            uint32_t r = __builtin_bitreverse32(f);
            r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
            r = (~r) >> (32 - 2 * SK_M);
            return sk_order(f < r ? f : r);
        };
        auto mer_at = [&](const Kmer &x, uint32_t p) -> uint32_t {  // SK_M-mer p of a k-mer: its bases p .. p + SK_M - 1
            const uint32_t o = 2u * (wq - p);
            const uint64_t ww = o >= 64 ? (x.hi >> (o - 64)) : (o == 0 ? x.lo : ((x.lo >> o) | (x.hi << (64 - o))));
            return (uint32_t)ww & SK_MMASK;
        };
        __builtin_amdgcn_wave_barrier();
        if (lane + 1 < w) mh[lane] = mm_hash(mer_at(Xs, lane + 1));  // the SK_M-mers of the vertex in front, from base lane + 1 on
        mh[w - 1 + lane] = ok ? mm_hash((uint32_t)K.lo & SK_MMASK) : SK_NONE;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t m_lo = SK_NONE, m_hi = SK_NONE;
        for (uint32_t i = 0; i < w; i++) {
            const uint32_t x = mh[lane + i];
            const uint32_t tm = max(m_lo, x);
            m_lo = min(m_lo, x);
            m_hi = min(m_hi, tm);
        }
        const uint32_t hm = t.mm_k < 0 ? m_lo ^ (m_hi * 0x85EBCA6Bu) : m_lo;  // (count_long.h skl_word2)
        h = own_table(t);
        s0 = ((((uint64_t)(sk_bin(hm) & 0xFFFFFF00u) * t.n_regions) >> 32) << MC_REGION_LG) | sk_home(key);  // (skl_bin)
#ifdef MC_BFS_CHECK_SLOTS
        if (ok) { TableRef h2; const uint64_t want = solid_locate_kmer<MODE>(t, K, k, key, h2);
                  if (want != s0) printf("[bfs] scout minimizer (hash keys): slot %llu, solid_locate_kmer says %llu (lane %u)\n", (unsigned long long)s0, (unsigned long long)want, lane); }
#endif
    } else {
        if (SH) s0 = solid_locate(t, key, h);
        else { h = own_table(t); s0 = solid_slot_of(t, key); }
    }
    SC_STAMP(1);
    if (ok) {
        cov = solid_get4(t, h, key, &aux, s0);
        lookups++;
    }
    SC_STAMP(2);
    // One of these pointers is where the next hop starts, and its first act is to fetch the read store's words around
    // it: every lane asks for the lines around ITS pointer now, so that they are on their way into the caches while this
    // hop is evaluated (the values are not used: `warm` only keeps the loads alive until the hop's end).
    uint64_t warm = 0;
#if MC_SCOUT_WARM
    if (aux) {
        uint32_t sp_;
        const uint64_t plo = ptr_decode(aux, &sp_);
        if (plo < t.reads_bases) {
            const uint64_t w0 = (plo > 128 ? plo - 128 : 0) >> 5;
            warm = t.reads[min(w0, last_word)] ^ t.reads[min(w0 + 8, last_word)] ^ t.reads[min(w0 + 15, last_word)];
        }
    }
#endif
    const unsigned long long solid_m = __ballot(ok && cov >= min_cov);
    const uint32_t m = solid_m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~solid_m);
    if (m == 0) {
        asm volatile("" ::"v"(warm));
        return;
    }
    bool other = aux != 0 && lane < m;
    if (other) {  // does the pointer lead back into this very read?
        uint32_t sp;
        const uint64_t at = ptr_decode(aux, &sp);
        const uint64_t here = fwd ? (uint64_t)Q + 1 + lv : (uint64_t)Q - 1 - lv;  // where this read holds lane's k-mer
        other = !(at <= here && here < at + sp);
    }
    if (n_cand) {
        uint32_t sp;
        const uint64_t apos = aux ? ptr_decode(aux, &sp) : 0;
        unsigned long long cm = __ballot(lane < m && lane + 48 >= m && other);
#pragma unroll
        for (int i = 0; i < MC_CAND_MAX; i++) {
            if (!cm || (uint32_t)i >= n_cand) break;
            const uint32_t j = 63u - (uint32_t)__builtin_clzll(cm);
            R.cptr[i] = (uint32_t)__builtin_amdgcn_readlane((int)aux, (int)j);
            R.cdelta[i] = m - 1 - j;
            R.nc = (uint32_t)i + 1;
            const uint64_t cpos = readlane64(apos, j);
            // lanes whose pointer sits in the same read at the matching distance add nothing
            const uint64_t d = (uint64_t)j - lane;  // (lanes above j are out of the mask already)
            cm &= ~__ballot(lane <= j && (apos + d == cpos || apos == cpos + d));
        }
        const uint32_t ci = lane / SCOUT_WORDS;
        uint32_t cp = 0;
#pragma unroll
        for (int i = 0; i < MC_CAND_MAX; i++) cp = ci == (uint32_t)i ? R.cptr[i] : cp;
        if (ci < R.nc) R.word = scout_word_of(t, cp, lane % SCOUT_WORDS);
    }
    // the m new bases, first on top: what follows the occurrence (precedes it, complemented)
    uint64_t e_hi, e_lo;
    if (fwd) {
        e_hi = bases32(sw, (uint32_t)((uint64_t)Q + lane_base + (uint64_t)k - base));
        e_lo = m > 32 ? bases32(sw, (uint32_t)((uint64_t)Q + lane_base + (uint64_t)k + 32 - base)) : 0;
    } else {
        e_hi = bases32_before_rc(sw, (uint32_t)((uint64_t)Q - lane_base - base));
        e_lo = m > 32 ? bases32_before_rc(sw, (uint32_t)((uint64_t)Q - lane_base - 32 - base)) : 0;
    }
    e_hi = uni64(e_hi);
    e_lo = uni64(e_lo);
    if (m < 32) e_hi &= ~0ull << (64 - 2 * m);
    if (m <= 32) e_lo = 0; else if (m < 64) e_lo &= ~0ull << (128 - 2 * m);
    R.m = m; R.why = 0; R.e_hi = e_hi; R.e_lo = e_lo; R.K = K; R.aux = aux; R.other = other;
    asm volatile("" ::"v"(warm));
    SC_STAMP(3);
}

__device__ __forceinline__ void st_u64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the tail of a path under construction: words P[0 .. wi) are complete, `tailw` holds the next tail_n < 32 bases
struct PathTail {
    uint64_t tailw;
    uint32_t tail_n, wi;
};
__device__ __forceinline__ PathTail path_open(uint64_t *P, const Kmer &X, int k, bool writer)
{   // the walker's own k bases open the path
    PathTail T;
    T.wi = 0;
    if (k <= 32) {
        T.tailw = X.lo << (64 - 2 * k);
        T.tail_n = (uint32_t)k;
        if (k == 32) { if (writer) P[0] = T.tailw; T.tailw = 0; T.tail_n = 0; T.wi = 1; }
    } else {
        const uint64_t top = (X.hi << (128 - 2 * k)) | (X.lo >> (2 * k - 64));
        if (writer) P[0] = top;
        T.tailw = X.lo << (128 - 2 * k);
        T.tail_n = (uint32_t)k - 32;
        T.wi = 1;
    }
    return T;
}
// appends the m <= 64 bases e_hi:e_lo (first on top, unused bits zero); the partial tail word is stored too, so that a
// reader may use every base appended so far
__device__ __forceinline__ void path_append(uint64_t *P, PathTail &T, uint64_t e_hi, uint64_t e_lo, uint32_t m, bool writer)
{
    const uint32_t sh = 2 * T.tail_n;  // tail_n < 32
    const uint64_t w0 = T.tailw | (sh ? e_hi >> sh : e_hi);
    const uint64_t w1 = sh ? ((e_hi << (64 - sh)) | (e_lo >> sh)) : e_lo;
    const uint64_t w2 = sh ? (e_lo << (64 - sh)) : 0;
    const uint32_t total = T.tail_n + m, nfull = total >> 5;
    if (writer) {  // (write-through stores: the reader may be a workgroup behind another L2)
        st_u64(&P[T.wi], w0);
        st_u64(&P[T.wi + 1], w1);
        st_u64(&P[T.wi + 2], w2);
        st_u64(&P[T.wi + 3], 0);
    }
    T.tailw = nfull == 0 ? w0 : (nfull == 1 ? w1 : w2);
    T.wi += nfull;
    T.tail_n = total & 31;
}

// The scout of walker `a` (one wave; every lane calls it).  From the walker's vertex it follows the reads: the slot of
// a k-mer points at one of its occurrences in the read store, the bases behind that occurrence (before it, when the read
// runs the other way) spell the next vertices, and lane i looks the i-th of them up -- two dependent memory round trips
// per hop of up to 64 levels.  The hop ends at the first k-mer that is not solid (a sequencing error, the end of the
// read); the next one starts from the last solid vertex with the read pointers found on the way (the nearest ones to
// the tip first, up to four tried).  The path is written, on the walker's own strand (for a walker that moves left: the
// reverse complement, so that the walk always appends), 32 bases per word into S.path: first the k bases of the
// walker's vertex, then one base per predicted level.  Only `solid` is checked here; whether each level is what the
// sequential BFS would find is the rounds' business.
template <int MODE, bool SH = false>
__device__ __forceinline__ void scout_run(const BfsState &S, const SolidView &t, NarrowLds &L, uint32_t a, int k, int min_cov, uint32_t budget,
                          unsigned long long &lookups)
{
    const uint32_t lane = threadIdx.x & 63;
    uint64_t *P = S.path + (uint64_t)a * PATH_WORDS;
    const bool right = L.wright[a] != 0;
    Kmer X = right ? L.root[a] : kmer_rc<MODE>(L.root[a], k);  // the walk strand: the scout always appends
    PathTail T = path_open(P, X, k, lane == 0);
    uint32_t cptr[4] = {0, 0, 0, 0}, cdelta[4] = {0, 0, 0, 0}, nc = 0;
    {
        uint32_t p0 = L.wptr[a];
        if (p0 == 0) {  // the walker's k-mer has not been looked up with its pointer yet
            (void)solid_get_kmer<MODE>(t, L.root[a], k, (uint64_t)key_of<MODE>(L.root[a], k), &p0);
            lookups++;
        }
        if (p0) { cptr[0] = p0; nc = 1; }
    }
    uint32_t levels = 0, hops = 0, n_nf = 0, n_m0 = 0;
    HopEval R;
    while (levels < budget && nc) {
        bool progressed = false;
        for (uint32_t ci = 0; ci < nc && !progressed; ci++) {
            scout_eval<MODE, SH>(t, L.sw[a], L.mh[a], X, k, min_cov, cptr[ci], cdelta[ci], min(64u, budget - levels), R, lookups);
            if (R.why == 1) { n_nf++; continue; }
            hops++;
            if (R.m == 0) { n_m0++; continue; }
            const uint32_t m = R.m;
            path_append(P, T, R.e_hi, R.e_lo, m, lane == 0);
            X.lo = readlane64(R.K.lo, m - 1);
            X.hi = readlane64(R.K.hi, m - 1);
            levels += m;
            // the next hop's candidates: the read pointers nearest to the tip that lead into OTHER reads than this one
            unsigned long long cm = __ballot(lane < m && lane + 48 >= m && R.other);
            nc = 0;
            while (cm && nc < 4) {
                const uint32_t j = 63u - (uint32_t)__builtin_clzll(cm);
                cm &= ~(1ull << j);
                cptr[nc] = (uint32_t)__builtin_amdgcn_readlane((int)R.aux, (int)j);
                cdelta[nc] = m - 1 - j;
                nc++;
            }
            progressed = true;
        }
        if (!progressed) break;
    }
    if (lane == 0) {
        L.plen[a] = levels;
        L.ppos[a] = 0;
        atomicAdd(&L.hops, (unsigned long long)hops);
        atomicAdd(&L.hop_levels, (unsigned long long)levels);
        atomicAdd(&L.s_calls, 1ull);
        atomicAdd(&L.s_nf, (unsigned long long)n_nf);
        atomicAdd(&L.s_m0, (unsigned long long)n_m0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (the rounds read the path with L1-bypassing loads)
}

// ---- the companion: a second workgroup per job that scouts WHILE the first one verifies ------------------------------
// The two talk through the job's ScoutBox in global memory.  The verifying workgroup posts a request -- its walkers,
// their read pointers, a budget -- under a new sequence number; the companion's waves form one TEAM per walker (all
// eight for a single walker): every member follows a different candidate read of the hop, the one that gets furthest
// extends the path, publishes the new length (resp[a]: request number, a "finished" bit, levels so far) and names the
// next candidates.  The verifier appends behind it and never waits for long: when no answer comes (the companion is
// not running: grids larger than the chip holds at once) it scouts for itself as above, and a companion nobody talks
// to leaves after a while.  A new request (the walk met something the prediction did not foresee) aborts the run.
struct ScoutBox {
    uint32_t req_seq, quit, F, budget;
    uint64_t root_hi[SCOUT_MAX_F], root_lo[SCOUT_MAX_F];
    uint32_t ptr[SCOUT_MAX_F], right[SCOUT_MAX_F];
    uint32_t resp[SCOUT_MAX_F];  // (seq & 0x7FFF) << 17 | finished << 16 | levels
    unsigned long long hops, levels, calls, nf, m0, iters, e_stuck, e_nc0, e_budget, e_stop;  // statistics (iters: team hops = round trips on the critical path / 2)
};
constexpr uint32_t BOX_WAIT_POLLS = 400;       // verifier: polls (~0.5 us each) before it gives the companion up
constexpr uint32_t BOX_IDLE_POLLS = 40000;     // companion: polls without a request before it leaves
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t box_resp(uint32_t seq, bool finished, uint32_t levels) { return ((seq & 0x7FFFu) << 17) | (finished ? 1u << 16 : 0u) | levels; }

static_assert(MC_CAND_MAX * SCOUT_WORDS <= 64, "a wave requests the words of all its candidates at once, SCOUT_WORDS lanes each");
static_assert(MC_TEAM_MAX % MC_CAND_MAX == 0 && MC_TEAM_MAX / MC_CAND_MAX <= 2, "one or two waves per candidate");
struct TeamLds {
    Kmer X[SCOUT_MAX_F];  // tips, walk strand
    PathTail T[SCOUT_MAX_F];
    uint32_t levels[SCOUT_MAX_F], nc[SCOUT_MAX_F], stuck[SCOUT_MAX_F], right[SCOUT_MAX_F];
    uint32_t cptr[SCOUT_MAX_F][8], cdelta[SCOUT_MAX_F][8];
    uint32_t reach[BFS_THREADS / 64];
    uint64_t ext[BFS_THREADS / 64][2];                    // the bases a candidate's second wave would add (e_hi, e_lo): its first wave appends them
    uint64_t sw[BFS_THREADS / 64][SCOUT_WORDS];
    uint64_t swn[SCOUT_MAX_F][MC_CAND_MAX][SCOUT_WORDS];  // the words around the candidates of the NEXT hop, requested by the wave that found them
    uint32_t staged[SCOUT_MAX_F];                         // ... are there (0 on a request's first hop)
    uint32_t mh[BFS_THREADS / 64][SCOUT_MH];
    uint32_t seq, quit, F, budget, stop;
    uint32_t published[SCOUT_MAX_F];  // levels whose path words are known to have arrived
    // consensus hops (scout_cons): one wave per walker, everything below is that wave's own
    uint64_t cw[SCOUT_MAX_F][CONS_N * CONS_WORDS];  // the read-store words around the candidates' pointers
    uint64_t cs[SCOUT_MAX_F][8];                    // the tip's k bases and the levels' bases behind them (two stretches of 64), 32 bases a word
    uint64_t cstr[SCOUT_MAX_F][CONS_N][2];          // the 64 bases each candidate holds for the levels past the tip, first on top
    uint32_t cnp[SCOUT_MAX_F][CONS_N], cnd[SCOUT_MAX_F][CONS_N];  // the next hop's candidates being picked: pointer, levels behind the tip
};


// ---- a consensus hop (round 4) ----------------------------------------------------------------------------------------
// What bounds a hop of scout_eval is the ERRORS of the one read it follows: at 1 % a read's next 64 k-mers are all right
// with probability 0.99^64 = 1/2, and the better of two reads gets 48 levels far (scripts/hop_model.py, 47.9 measured) -- for
// 128 table look-ups and two lone waves' ~600 instructions each.  But a guess needs no look-up: where two reads that hold the
// tip AGREE on the next base, that base is right (both wrong alike: 10^-4 a level), and which reads hold the tip the store's
// words tell.  So a hop here takes up to CONS_N candidate reads (the pointers of the last vertices, as before), lines them up at
// the tip (lane i: candidate i -- its 64 next bases as two words), and lane l votes on level l + 1: the path goes on while at
// least two reads that are still "alive" agree and no other base has as many votes.  Only the hop's LAST CONS_TAIL vertices are
// looked up: the new tip must be solid (the path is cut back to the last solid vertex: this is where a consensus that has run
// out of reads goes wrong), and their read pointers name the next hop's candidates.  With a single usable read (the first hop
// of a request, thin coverage) every level is looked up, as scout_eval does.
// One wave per walker, nothing shared with other waves: no workgroup barrier inside a request's hops.
struct ConsHop {
    uint32_t m;           // (uniform) levels added; 0: the hop failed
    int why;              // (uniform) 0 ok, 1 no candidate holds the tip, 2 nothing agreed on / solid behind it
    uint64_t e_hi, e_lo;  // (uniform) the first min(m, 64) new bases, first on top
    uint64_t f_hi, f_lo;  // (uniform) the bases of the levels 65 .. m (a hop's second stretch)
    Kmer X;               // (uniform) the new tip, walk strand
    uint32_t nc;          // (uniform) candidates found for the next hop (lanes < nc hold them)
    uint32_t used;        // (uniform) reads that held the tip
};

__device__ __forceinline__ uint64_t cons_first_word(uint64_t pos) { return (pos > CONS_BACK ? pos - CONS_BACK : 0) >> 5; }
__device__ __forceinline__ uint64_t cons_word_of(const SolidView &t, uint32_t cptr, uint32_t i)
{   // word i < CONS_WORDS of the piece of the read store a consensus hop stages for the (exact) pointer cptr
    if (cptr == 0) return 0;
    const uint64_t pos = (uint64_t)cptr - 1;
    if (pos >= t.reads_bases) return 0;
    return t.reads[min(cons_first_word(pos) + i, (t.reads_bases + 31) / 32)];
}

// the largest and the second largest of four byte counters, and which is the largest (ties: the lower index)
__device__ __forceinline__ uint32_t cons_top(uint32_t cnt, uint32_t &top, uint32_t &second)
{
    const uint32_t c0 = cnt & 255u, c1 = (cnt >> 8) & 255u, c2 = (cnt >> 16) & 255u, c3 = cnt >> 24;
    const uint32_t m01 = max(c0, c1), n01 = min(c0, c1), m23 = max(c2, c3), n23 = min(c2, c3);
    const bool hi = m23 > m01;
    top = hi ? m23 : m01;
    second = max(hi ? m01 : m23, hi ? n23 : n01);
    return hi ? (c3 > c2 ? 3u : 2u) : (c1 > c0 ? 1u : 0u);
}

// lane l holds the base of level l + 1 (`on`: it counts): the 64 bases as two words, first on top -- OR over the rows of 16 lanes
// (row_shr 1, 2, 4, 8; a row makes half a word)
__device__ __forceinline__ void cons_pack(uint32_t base, bool on, uint32_t lane, uint64_t &hi, uint64_t &lo)
{
    int v = on ? (int)(base << (30u - 2u * (lane & 15u))) : 0;
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    hi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 15) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 31);
    lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 47) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// c_ptr / c_del: lane i < nc holds candidate i (an exact read pointer, and how many levels behind the tip its k-mer sits); on
// return the next hop's candidates, and `word` = this lane's word of their pieces of the read store (requested, not waited for:
// the caller stores it into L.cw[g][lane] once it has done its own work).  staged: L.cw[g] holds the words of THESE candidates.
template <int MODE, bool SH = false>
__device__ __forceinline__ void scout_cons(const SolidView &t, TeamLds &L, uint32_t g, const Kmer &X, int k, int min_cov, uint32_t &c_ptr,
                                           uint32_t &c_del, uint32_t nc, uint32_t want, bool staged, ConsHop &R, uint64_t &word, uint64_t &word2,
                                           unsigned long long &lookups, unsigned long long *tsc = nullptr)
{   // want <= 128: two stretches of 64 levels; word / word2: this lane's (and, for lanes < 32, a second) word of the next hop's pieces
    static_assert(MODE == KEY_PACKED, "the tip and its levels in 64-bit words");
#ifdef MC_SCOUT_TIMING
    unsigned long long ts_ = __builtin_amdgcn_s_memrealtime();
#else
    (void)tsc;
#endif
    const uint32_t lane = threadIdx.x & 63;
    uint64_t *cw = L.cw[g];
    R.m = 0; R.why = 1; R.e_hi = R.e_lo = 0; R.f_hi = R.f_lo = 0; R.X = X; R.nc = 0; R.used = 0;
    word = 0; word2 = 0;
    // staged word `lane` belongs to candidate cdA (its word wiA), staged word 64 + lane (lanes < 32) to cdB
    const uint32_t cdA = lane / CONS_WORDS, wiA = lane - cdA * CONS_WORDS, cdB = (lane + 64u) / CONS_WORDS, wiB = lane + 64u - cdB * CONS_WORDS;
    if (!staged) {
        const uint32_t cpA = (uint32_t)__shfl((int)c_ptr, (int)cdA), cpB = (uint32_t)__shfl((int)c_ptr, (int)min(cdB, CONS_N - 1u));
        __builtin_amdgcn_wave_barrier();
        cw[lane] = cdA < nc ? cons_word_of(t, cpA, wiA) : 0ull;
        if (lane < 32u) cw[64u + lane] = cdB < nc ? cons_word_of(t, cpB, wiB) : 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- where does each candidate hold the tip, and which way does it run?  (lane i: candidate i.  32-bit places: every
    // pointer is exact, so the store has fewer than 2^31 bases)
    const uint32_t sk2 = 2u * (uint32_t)k, rb = (uint32_t)t.reads_bases;
    const Kmer Xr = kmer_rc<MODE>(X, k);
    uint32_t Qv = ~0u, wbv = 0;  // place of the tip in the store | 1 << 31: the read runs the walk's way (~0: not there); first staged base
    if (lane < nc && c_ptr) {
        const uint32_t pos = c_ptr - 1u;
        if (pos + (uint32_t)k <= rb) {
            const uint32_t wb = (pos > CONS_BACK ? pos - CONS_BACK : 0u) & ~31u;
            const uint64_t *sw = cw + CONS_WORDS * lane;
            wbv = wb;
            const uint32_t qf = pos + c_del;
            if (qf + (uint32_t)k <= rb && kmer_eq(kmer_at<MODE>(sw, qf - wb, k), X)) Qv = qf | 0x80000000u;
            else if (pos >= c_del + wb && kmer_eq(kmer_at<MODE>(sw, pos - c_del - wb, k), Xr)) Qv = pos - c_del;
        }
    }
    // two candidates that hold the tip at the same place are one read: its vote counts once (the candidates sit in the first
    // lanes of a row of sixteen: row_shr brings the earlier ones' places over, zeros from beyond the row's start)
    {
        static_assert(CONS_N == 8, "seven earlier candidates to compare with");
#define MC_CONS_EARLIER(o) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)Qv, 0x110 + (o), 0xF, 0xF, true) == Qv)
        // (| not ||: every lane takes part in every one of them)
        const bool dup = MC_CONS_EARLIER(1) | MC_CONS_EARLIER(2) | MC_CONS_EARLIER(3) | MC_CONS_EARLIER(4) | MC_CONS_EARLIER(5) |
                         MC_CONS_EARLIER(6) | MC_CONS_EARLIER(7);
#undef MC_CONS_EARLIER
        if (dup && Qv != 0u) Qv = ~0u;  // (0: what row_shr brings from beyond the row's start)
    }
    const uint32_t U = (uint32_t)__ballot(lane < CONS_N && Qv != ~0u);
    const uint32_t nU = (uint32_t)__popc(U);
    R.used = nU;
    if (nU == 0) return;
    R.why = 2;
    const bool single = nU == 1;
#ifdef MC_SCOUT_TIMING
    if (tsc) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[0] += n_ - ts_; ts_ = n_; }
#endif
    // ---- lane i < CONS_N: the 64 bases candidate i holds for the levels 1 .. 64, first on top, and how many of them lie in
    // its staged words.  The walk's next bases FOLLOW the tip in a read that runs its way and PRECEDE it, complemented, in
    // one that does not.
    uint32_t len = 0;
    uint64_t s_hi = 0, s_lo = 0;
    // add: 0 for the levels 1 .. 64, 64 for the second stretch (65 .. 128); live: this lane's candidate takes part
    auto cut = [&](uint32_t add, bool live, uint32_t &len_o, uint64_t &o_hi, uint64_t &o_lo) {
        len_o = 0; o_hi = 0; o_lo = 0;
        if (lane < CONS_N && Qv != ~0u && live) {
            const uint32_t Q = Qv & 0x7FFFFFFFu;
            const uint64_t *sw = cw + CONS_WORDS * lane;
            if (Qv >> 31) {
                const uint32_t off0 = Q + (uint32_t)k - wbv + add, lim = min(CONS_WORDS * 32u, rb - wbv);
                if (off0 < lim) { len_o = min(64u, lim - off0); o_hi = bases32(sw, off0); o_lo = bases32(sw, off0 + 32u); }
            } else {
                const uint32_t a = Q - wbv;  // staged bases in front of the tip (fewer than 64 only at the very start of the store)
                if (a >= 64u + add) { len_o = 64u; o_hi = rc64_pairs(bases32(sw, a - 32u - add)); o_lo = rc64_pairs(bases32(sw, a - 64u - add)); }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < CONS_N) { L.cstr[g][lane][0] = o_hi; L.cstr[g][lane][1] = o_lo; L.cnd[g][lane] = len_o; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    cut(0u, true, len, s_hi, s_lo);
    uint32_t m1 = 0;            // levels of the second stretch
    uint64_t f_hi = 0, f_lo = 0;
    uint32_t m0;
    uint64_t e_hi, e_lo;
#if MC_CONS_PAIRS
    // ---- lane (i, j): how far do candidates i < j AGREE?  The longest agreement of any pair is the hop.  (Measured against
    // the vote below: 0.78 us instead of 1.10, but 48 levels a hop instead of 54 -- a lone error of the best pair ends it.)
    {
        const uint32_t pi = lane >> 3, pj = lane & 7u;
        const uint64_t a_hi = L.cstr[g][pi][0], a_lo = L.cstr[g][pi][1], b_hi = L.cstr[g][pj][0], b_lo = L.cstr[g][pj][1];
        const uint32_t la = L.cnd[g][pi], lb = L.cnd[g][pj];
        const uint64_t x_hi = a_hi ^ b_hi, x_lo = a_lo ^ b_lo;
        uint32_t agree = x_hi ? (uint32_t)__builtin_clzll(x_hi) >> 1 : (x_lo ? 32u + ((uint32_t)__builtin_clzll(x_lo) >> 1) : 64u);
        agree = min(min(agree, want), min(la, lb));
        if (single ? pi != pj : pi >= pj) agree = 0;
        int v = (int)agree;
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true));
        m0 = (uint32_t)max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
                           max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
        if (m0 == 0) return;
        const uint32_t wl = (uint32_t)__builtin_ctzll(__ballot(agree == m0));
        e_hi = readlane64(a_hi, wl);
        e_lo = readlane64(a_lo, wl);
        if (m0 < 32u) { e_hi &= ~0ull << (64u - 2u * m0); e_lo = 0; }
        else if (m0 == 32u) e_lo = 0;
        else if (m0 < 64u) e_lo &= ~0ull << (128u - 2u * m0);
    }
#else
    // ---- a stretch of 64 levels: lane l votes on level l + 1 of it -- the candidates' bases there (4 bits each, 8: none) and a first
    // count, four byte counters.  v_hi:v_lo / vlen: lane i < CONS_N's candidate's bases for the stretch (cut above); alive_end: lane i <
    // CONS_N: up to which level of the stretch its candidate is alive; returns the levels the stretch adds, their bases in o_hi:o_lo
    uint32_t alive_end = 64u;
    auto vote = [&](const uint64_t v_hi, const uint64_t v_lo, const uint32_t vlen, const uint32_t want_s, const bool single_s, uint64_t &o_hi, uint64_t &o_lo) -> uint32_t {
        uint32_t bb = 0, cnt = 0;
        {
            const uint64_t *col = &L.cstr[g][0][lane >> 5];
            const uint32_t sh = 62u - 2u * (lane & 31u);
            uint64_t wv_[CONS_N];
#pragma unroll
            for (uint32_t i = 0; i < CONS_N; i++) wv_[i] = col[2u * i];  // (requested together)
#pragma unroll
            for (uint32_t i = 0; i < CONS_N; i++) {
                const uint32_t len_i = (uint32_t)__builtin_amdgcn_readlane((int)vlen, (int)i);
                const bool valid = lane < len_i;
                const uint32_t b = (uint32_t)(wv_[i] >> sh) & 3u;
                cnt += (valid ? 1u : 0u) << (8u * b);
                bb |= (valid ? b : 8u) << (4u * i);
            }
        }
        uint32_t top, second;
        uint32_t base = cons_top(cnt, top, second);
        bool ok = top >= 1;
        if (!single_s) {
            // ---- up to which level is each read alive?  A read stops being alive at its first disagreement with the first count's
            // winners that has a second one within the two levels behind it: a sequencing error is one lone disagreement, the end
            // of a read (the store holds the next read's bases behind it) a run of them.  Lane i < CONS_N compares candidate i's 64
            // bases with the winners all at once: a flag a level in the even bits (two bits a level, first level on top, so
            // "behind" is to the right), levels it holds no base for flagged too
            uint64_t p_hi, p_lo;
            cons_pack(base, true, lane, p_hi, p_lo);
            uint32_t end = 0;
            if (lane < CONS_N) {
                constexpr uint64_t EV = 0x5555555555555555ull;
                const uint64_t x_hi = v_hi ^ p_hi, x_lo = v_lo ^ p_lo;
                uint64_t d_hi = (x_hi | (x_hi >> 1)) & EV, d_lo = (x_lo | (x_lo >> 1)) & EV;
                if (vlen < 32u) { d_hi |= EV >> (2u * vlen); d_lo = EV; }
                else if (vlen < 64u) d_lo |= EV >> (2u * (vlen - 32u));
                const uint64_t f_hi_ = d_hi & ((d_hi << 2) | (d_lo >> 62) | (d_hi << 4) | (d_lo >> 60)), f_lo_ = d_lo & ((d_lo << 2) | (d_lo << 4));
                end = f_hi_ ? (uint32_t)__builtin_clzll(f_hi_) >> 1 : (f_lo_ ? 32u + ((uint32_t)__builtin_clzll(f_lo_) >> 1) : 64u);
                end = min(end, vlen);
            }
            alive_end = end;
            // ---- the count among the reads that are alive: the path goes on while two of them agree and no other base has as many votes
            uint32_t cnt2 = 0;
#pragma unroll
            for (uint32_t i = 0; i < CONS_N; i++) {
                const uint32_t end_i = (uint32_t)__builtin_amdgcn_readlane((int)end, (int)i);
                cnt2 += (lane < end_i ? 1u : 0u) << (8u * ((bb >> (4u * i)) & 3u));
            }
            base = cons_top(cnt2, top, second);
            ok = top >= 2 && top > second;
        }
        const unsigned long long okm = __ballot(ok && lane < want_s);
        const uint32_t ms = okm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okm);
        o_hi = 0; o_lo = 0;
        if (ms) cons_pack(base, lane < ms, lane, o_hi, o_lo);
        return ms;
    };
    m0 = vote(s_hi, s_lo, len, min(want, 64u), single, e_hi, e_lo);
    if (m0 == 0) return;
    // ---- the second stretch: the hop ran its 64 lanes out and the reads are still there -- the ones that were alive to the end of the
    // first stretch vote on the levels 65 .. 128 from the bases behind, no round trip in between (the look-ups come once, at the end)
    if (!single && m0 == 64u && want > 64u) {
        uint32_t len2;
        uint64_t t_hi, t_lo;
        cut(64u, alive_end >= 64u, len2, t_hi, t_lo);
        m1 = vote(t_hi, t_lo, len2, want - 64u, false, f_hi, f_lo);
    }
#endif
    // the tip's k bases and the levels behind them, 32 bases a word: vertex j levels past the tip = bases [j, j + k) of it
    uint64_t *cs = L.cs[g];
    __builtin_amdgcn_wave_barrier();
    const uint32_t mt = m0 + m1;  // the levels the votes added
    if (lane == 0) {
        cs[0] = (X.lo << (64u - sk2)) | (e_hi >> sk2);
        cs[1] = (e_hi << (64u - sk2)) | (e_lo >> sk2);
        cs[2] = (e_lo << (64u - sk2)) | (f_hi >> sk2);
        cs[3] = (f_hi << (64u - sk2)) | (f_lo >> sk2);
        cs[4] = f_lo << (64u - sk2);
        cs[5] = 0; cs[6] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef MC_SCOUT_TIMING
    if (tsc) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[1] += n_ - ts_; ts_ = n_; }
#endif
    // ---- the vertices that are looked up: lane l <-> level lo + l
    const uint32_t lo = single || mt <= CONS_TAIL ? 1u : mt - CONS_TAIL + 1u, n_lk = mt - lo + 1u;
    const bool lk = lane < n_lk;
    const uint32_t j = lo + lane;
    const Kmer K{0, lk ? bases32(cs, j) >> (64u - sk2) : 0ull};
    const uint64_t key = lk ? (uint64_t)key_of<MODE>(K, k) : 0;
    uint64_t s0;
    TableRef h;
    if (SH ? t.owner_mm_k : t.mm_k) {
        // the minimizer bins of the vertices (scout_eval says how): the SK_M-mers at bases lo .. m0 + w - 1, one or two a lane
        const uint32_t w = (uint32_t)k - SK_M + 1, n_mm = n_lk + w - 1;
        uint32_t *mh = L.mh[g];
        auto mm_hash = [](uint32_t f) {
            uint32_t r = __builtin_bitreverse32(f);
            r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
            r = (~r) >> (32 - 2 * SK_M);
            return sk_order(f < r ? f : r);
        };
        __builtin_amdgcn_wave_barrier();
        if (lane < n_mm) mh[lane] = mm_hash((uint32_t)(bases32(cs, lo + lane) >> (64 - 2 * SK_M)));
        if (64u + lane < n_mm) mh[64u + lane] = mm_hash((uint32_t)(bases32(cs, lo + 64u + lane) >> (64 - 2 * SK_M)));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t hmin = SK_NONE;
        if (lk) {
            if (w == 17) {
                uint32_t hv[17];
#pragma unroll
                for (int i = 0; i < 17; i++) hv[i] = mh[lane + i];
#pragma unroll
                for (int i = 0; i < 17; i++) hmin = min(hmin, hv[i]);
            } else {
                for (uint32_t i = 0; i < w; i++) hmin = min(hmin, mh[lane + i]);
            }
        }
        if (SH) s0 = solid_locate(t, key, h, true, hmin);
        else { h = own_table(t); s0 = ((((uint64_t)sk_bin(hmin) * t.n_regions) >> 32) << MC_REGION_LG) | sk_home(key); }
    } else {
        if (SH) s0 = solid_locate(t, key, h);
        else { h = own_table(t); s0 = solid_slot_of(t, key); }
    }
#ifdef MC_SCOUT_TIMING
    if (tsc) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[2] += n_ - ts_; ts_ = n_; }
#endif
    // (appending the levels in front of the looked-up ones to the path while the slots travel was measured: the wait for the slots
    // then waits for those write-through stores too, 0.8 -> 1.45 us)
    uint32_t aux = 0;
    int cov = -1;
    if (lk) {
        cov = solid_get4(t, h, key, &aux, s0);
        lookups++;
    }
#ifdef MC_SCOUT_TIMING
    if (tsc) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[3] += n_ - ts_; ts_ = n_; }
#endif
    const unsigned long long sm = __ballot(lk && cov >= min_cov);
    const uint32_t ps = sm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~sm);  // the looked-up vertices that are solid, from the first on
    // (a retry CONS_TAIL levels earlier when the first of them is not solid, and a hop that follows the first usable read alone
    // when no two agree on the very next base, were built and measured: 4 requests a walk instead of 9, but the loop around the
    // look-ups made every hop 0.3 us longer -- 8.1 ms against 7.8)
    const uint32_t m = lo - 1u + ps;  // (ps == 0 behind a consensus: its unchecked part stays, and the hop names no candidates)
    if (m == 0) return;

    // ---- the next hop's candidates: the pointers of the solid vertices nearest to the new tip, one a read where that is cheap to
    // tell (the same read holds neighbouring vertices at neighbouring places; what is left is dropped when the tip is looked for)
    bool el = lane < ps && aux != 0 && aux - 1u + (uint32_t)k <= rb && m - j < CONS_TAIL;
    {   // (through the staged words' LDS, which nobody reads any more: the neighbours' values in reads that travel together)
        uint32_t *scr = reinterpret_cast<uint32_t *>(cw);
        const uint32_t fw = el ? aux - j : 0xFFFFFF00u + 2u * lane, rv = el ? aux + j : 0xFFFFFF01u + 2u * lane;  // (pointers are below 2^31)
        __builtin_amdgcn_wave_barrier();
        scr[2u * lane] = fw;
        scr[2u * lane + 1u] = rv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool same = false;
#pragma unroll
        for (uint32_t o = 1; o <= MC_CONS_NEIGHBOURS; o++) {
            const uint32_t n = min(lane + o, 63u);
            const uint2 other = *reinterpret_cast<const uint2 *>(scr + 2u * n);
            same = same | (n != lane && (other.x == fw || other.y == rv));
        }
        el = el && !same;
    }
    const unsigned long long cm = __ballot(el);
    const uint32_t n_sel = min((uint32_t)__popcll(cm), CONS_N);
    const uint32_t rank = lane == 63u ? 0u : (uint32_t)__popcll(cm >> (lane + 1u));  // eligible lanes nearer to the tip
    __builtin_amdgcn_wave_barrier();
    if (el && rank < CONS_N) { L.cnp[g][rank] = aux; L.cnd[g][rank] = m - j; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    c_ptr = lane < n_sel ? L.cnp[g][lane] : 0u;
    c_del = lane < n_sel ? L.cnd[g][lane] : 0u;
    if (cdA < n_sel) word = cons_word_of(t, L.cnp[g][cdA], wiA);
    if (lane < 32u && cdB < n_sel) word2 = cons_word_of(t, L.cnp[g][cdB], wiB);
    {   // the bases of the levels that stay (the tail may have been cut back to the last solid vertex)
        auto keep = [](uint64_t &hi, uint64_t &lo_, uint32_t n) {
            if (n == 0) { hi = 0; lo_ = 0; }
            else if (n < 32) { hi &= ~0ull << (64u - 2u * n); lo_ = 0; }
            else if (n == 32) lo_ = 0;
            else if (n < 64) lo_ &= ~0ull << (128u - 2u * n);
        };
        keep(e_hi, e_lo, min(m, 64u));
        keep(f_hi, f_lo, m > 64u ? m - 64u : 0u);
    }
    R.m = m; R.why = 0; R.e_hi = e_hi; R.e_lo = e_lo; R.f_hi = f_hi; R.f_lo = f_lo; R.nc = n_sel;
    R.X = Kmer{0, bases32(cs, m) >> (64u - sk2)};
#ifdef MC_SCOUT_TIMING
    if (tsc) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[4] += n_ - ts_; ts_ = n_; }
#endif
}

template <int MODE, bool SH = false>
__device__ __forceinline__ void scout_companion(const BfsState &S, const SolidView &t, TeamLds &L, int k, int min_cov)
{
    ScoutBox *box = S.box;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr uint32_t W = BFS_THREADS / 64;
    uint32_t last_seq = 0, idle = 0;
    unsigned long long lookups = 0, hops = 0, n_nf = 0, n_m0 = 0, iters = 0;
    unsigned long long tsc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)tsc;
    for (;;) {
        if (tid == 0) { L.seq = ld_u32(&box->req_seq); L.quit = ld_u32(&box->quit); }
        BFS_SYNC();
        const uint32_t seq = L.seq, quit = L.quit;
        BFS_SYNC();
        if (quit) break;
        if (seq == last_seq) {
            if (++idle > BOX_IDLE_POLLS) break;
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        idle = 0;
        last_seq = seq;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // ---- the request
        if (tid == 0) { L.F = ld_u32(&box->F); L.budget = ld_u32(&box->budget); L.stop = 0; }
        if (tid < SCOUT_MAX_F) {
            const Kmer root{ld_sc1(&box->root_hi[tid]), ld_sc1(&box->root_lo[tid])};
            const uint32_t right = ld_u32(&box->right[tid]), p0 = ld_u32(&box->ptr[tid]);
            L.right[tid] = right;
            L.X[tid] = right ? root : kmer_rc<MODE>(root, k);
            L.levels[tid] = 0;
            L.published[tid] = 0;
            L.stuck[tid] = p0 == 0;  // (the verifier looks the pointer up before it asks)
            L.cptr[tid][0] = p0;
            L.cdelta[tid][0] = 0;
            L.nc[tid] = p0 ? 1 : 0;
        }
        BFS_SYNC();
        const uint32_t F = min(L.F, (uint32_t)SCOUT_MAX_F), budget = min(L.budget, PATH_CAP);
        if (F == 0) continue;
        bool cons_done = false;
        if constexpr (MC_SCOUT_CONSENSUS != 0 && MODE == KEY_PACKED) {
            if (t.reads_bases < PTR_EXACT_END) {  // (every read pointer names its place exactly: scout_cons looks for the tip nowhere else)
                // ---- consensus hops: wave g is walker g's scout, on its own until the request is done with
                cons_done = true;
                if (wv < F) {
                    uint64_t *P = S.path + (uint64_t)wv * PATH_WORDS;
                    Kmer X = L.X[wv];
                    PathTail T = path_open(P, X, k, lane == 0);
                    if (lane == 0) { P[T.wi] = T.tailw; P[T.wi + 1] = 0; }
                    uint32_t levels = 0, published = 0, nc = L.nc[wv];
                    uint32_t c_ptr = lane == 0 ? L.cptr[wv][0] : 0u, c_del = 0;
                    bool staged = false, stopped = false;
                    while (nc && levels < budget) {
                        uint32_t probe = seq;
                        if (lane == 0) probe = ld_u32(&box->req_seq) | (ld_u32(&box->quit) << 31);  // (used at the end of the hop)
                        ConsHop R;
                        uint64_t word, word2;
                        scout_cons<MODE, SH>(t, L, wv, X, k, min_cov, c_ptr, c_del, nc, min((uint32_t)MC_CONS_LEVELS, budget - levels), staged, R, word, word2, lookups, tsc);
                        if (tid == 0) iters++;
#ifdef MC_SCOUT_TIMING
                        unsigned long long tq_ = __builtin_amdgcn_s_memrealtime();
#endif
                        // the length the PREVIOUS hop added is published now: its path words were stored before this hop's round
                        // trips, and a wave's loads come back behind its earlier stores (see the other loop below)
                        if (lane == 0 && levels != published) st_u32(&box->resp[wv], box_resp(seq, false, levels));
                        published = levels;
                        if (R.why == 1) n_nf += lane == 0; else { hops += lane == 0; if (R.m == 0) n_m0 += lane == 0; }
                        if (R.m == 0) {
                            if (lane == 0) atomicAdd(&box->e_stuck, 1ull);
                            break;
                        }
                        path_append(P, T, R.e_hi, R.e_lo, min(R.m, 64u), lane == 0);
                        if (R.m > 64u) path_append(P, T, R.f_hi, R.f_lo, R.m - 64u, lane == 0);  // (the hop's second stretch)
                        levels += R.m;
                        X = R.X;
                        nc = R.nc;
                        if (nc == 0) {
                            if (lane == 0) atomicAdd(&box->e_nc0, 1ull);
                        } else {  // the words around the next hop's candidates: asked for inside the hop, they have travelled meanwhile
                            __builtin_amdgcn_wave_barrier();
                            L.cw[wv][lane] = word;
                            if (lane < 32u) L.cw[wv][64u + lane] = word2;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            staged = true;
                        }
#ifdef MC_SCOUT_TIMING
                        if (wv == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[5] += n_ - tq_; tsc[6]++; tsc[7] += ((unsigned long long)R.used << 32) + R.nc; }
#endif
                        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)probe) != seq) { stopped = true; break; }
                    }
                    if (lane == 0) {
                        L.levels[wv] = levels;
                        L.stuck[wv] = levels < budget;
                        if (stopped) L.stop = 1;
                    }
                }
            }
        }
        if (!cons_done) {
        uint32_t Tm = min(W, (uint32_t)MC_TEAM_MAX);  // team size: the largest power of two with F * Tm <= W
        while (Tm > 1 && F * Tm > W) Tm >>= 1;
        // a team = n_cd candidate reads x n_sg waves each: wave s of a candidate looks at the levels 64 s + 1 .. 64 s + 64 past the tip
        const uint32_t n_sg = Tm > (uint32_t)MC_CAND_MAX ? Tm / (uint32_t)MC_CAND_MAX : 1u, n_cd = Tm / n_sg;
        const uint32_t g = wv / Tm, u = wv - g * Tm, cd = u / n_sg, sg = u - cd * n_sg;
        const bool member = g < F;
        uint64_t *P = S.path + (uint64_t)g * PATH_WORDS;
        if (member && u == 0) {
            const PathTail T0 = path_open(P, L.X[g], k, lane == 0);
            if (lane == 0) { L.T[g] = T0; P[T0.wi] = T0.tailw; P[T0.wi + 1] = 0; }
        }
        BFS_SYNC();
        // ---- hops, all teams in step
        HopEval R;
        if (tid < SCOUT_MAX_F) L.staged[tid] = 0;
        BFS_SYNC();
        for (;;) {
            uint32_t probe = seq;
            if (tid == 0) probe = ld_u32(&box->req_seq) | (ld_u32(&box->quit) << 31);  // (used at the end of the hop)
            R.m = 0; R.e_hi = 0; R.e_lo = 0;
            const bool busy = member && !L.stuck[g] && L.levels[g] < budget;
            if (busy && cd < L.nc[g] && budget - L.levels[g] > 64u * sg) {
                const bool staged = L.staged[g] != 0;
                scout_eval<MODE, SH>(t, staged ? L.swn[g][cd] : L.sw[wv], L.mh[wv], L.X[g], k, min_cov, L.cptr[g][cd], L.cdelta[g][cd],
                                 min(64u, budget - L.levels[g] - 64u * sg), R, lookups, tsc, staged, n_cd, 64u * sg);
                if (sg == 0) { if (R.why == 1) n_nf += lane == 0; else { hops += lane == 0; if (R.m == 0) n_m0 += lane == 0; } }
            }
#ifdef MC_SCOUT_TIMING
            unsigned long long tq_ = __builtin_amdgcn_s_memrealtime();
#endif
            // What this wave hands to the next hop if the stretch it looked at ends the hop: the candidates it found (R.cptr) and
            // the read store's words around them, which it asked for inside scout_eval: they travel while the team finds its
            // winner (a hop's first act used to be to ask for them and wait: 0.5 us of its 4).
            const uint32_t my_nc = R.m ? R.nc : 0;
            const uint64_t my_word = R.word;
            if (lane == 0) { L.reach[wv] = R.m; L.ext[wv][0] = R.e_hi; L.ext[wv][1] = R.e_lo; }
            // The length the previous hop added is published now: its path words (write-through stores) were issued before
            // this hop's two round trips, and a wave's loads come back behind its earlier stores, so they have arrived -- a release
            // fence at agent scope right behind the stores would cost a cache write-back on every hop.  (The length only steers:
            // the walk's rounds check every level they take from a path against the table.)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const uint32_t lv_done = tid < F ? L.levels[tid] : 0;  // (read before the barrier: this hop's winner changes it behind it)
            BFS_SYNC();
            if (tid < F && lv_done != L.published[tid]) {
                st_u32(&box->resp[tid], box_resp(seq, false, lv_done));
                L.published[tid] = lv_done;
            }
            if (busy) {
                // how far each candidate got: its first wave's stretch and, when that one is whole, its second wave's -- unless that
                // holds too few vertices to name the next hop's candidates from (MC_SEG_MIN): the hop after covers them again
                uint32_t best = 0, best_c = 0, best_r0 = 0, best_r1 = 0;
                for (uint32_t i = 0; i < n_cd; i++) {
                    const uint32_t r0 = L.reach[g * Tm + i * n_sg], r1x = n_sg > 1 ? L.reach[g * Tm + i * n_sg + 1] : 0u;
                    const uint32_t r1 = (r0 == 64u && r1x >= (uint32_t)MC_SEG_MIN) ? r1x : 0u;
                    if (r0 + r1 > best) { best = r0 + r1; best_c = i; best_r0 = r0; best_r1 = r1; }
                }
                if (best == 0) {
                    if (u == 0 && lane == 0) { L.stuck[g] = 1; atomicAdd(&box->e_stuck, 1ull); }
                } else if (cd == best_c) {
                    if (sg == 0) {  // the candidate's first wave extends the path, by its own stretch and then by the second wave's
                        PathTail T = L.T[g];
                        path_append(P, T, R.e_hi, R.e_lo, best_r0, lane == 0);
                        if (best_r1) path_append(P, T, L.ext[wv + 1][0], L.ext[wv + 1][1], best_r1, lane == 0);
                        if (lane == 0) { L.T[g] = T; L.levels[g] = L.levels[g] + best; }
                    }
                    if (sg == (best_r1 ? 1u : 0u)) {  // the wave whose stretch ends the hop: the new tip, the next hop's candidates and their words
                        const uint32_t m = best_r1 ? best_r1 : best_r0;
                        Kmer X;
                        X.lo = readlane64(R.K.lo, m - 1);
                        X.hi = readlane64(R.K.hi, m - 1);
                        if (lane / SCOUT_WORDS < my_nc) L.swn[g][lane / SCOUT_WORDS][lane % SCOUT_WORDS] = my_word;
                        if (lane == 0) {
#pragma unroll
                            for (int i = 0; i < MC_CAND_MAX; i++) { L.cptr[g][i] = R.cptr[i]; L.cdelta[g][i] = R.cdelta[i]; }
                            L.X[g] = X; L.nc[g] = my_nc; L.staged[g] = 1;
                            if (my_nc == 0) { L.stuck[g] = 1; atomicAdd(&box->e_nc0, 1ull); }
                        }
                    }
                }
            }
            if (tid == 0 && probe != seq) L.stop = 1;
            if (tid == 0) iters++;
            BFS_SYNC();
#ifdef MC_SCOUT_TIMING
            { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tsc[4] += n_ - tq_; tsc[5]++; }
#endif
            bool any = false;
            for (uint32_t a = 0; a < F; a++) any = any || (!L.stuck[a] && L.levels[a] < budget);
            if (!any || L.stop) break;
        }
        }  // (!cons_done)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        BFS_SYNC();
        if (!L.stop && tid < F) st_u32(&box->resp[tid], box_resp(seq, true, L.levels[tid]));
        if (tid == 0) { atomicAdd(&box->calls, 1ull); if (L.stop) atomicAdd(&box->e_stop, 1ull); else if (L.levels[0] >= budget) atomicAdd(&box->e_budget, 1ull); }
        if (tid == 0) BFS_TRACE(S, 8u << 24 | (seq & 0xFFFFFF), F | (cons_done ? 1u : 0u) << 8 | L.stop << 16, L.levels[0], F > 1 ? L.levels[1] : 0u, budget, L.stuck[0] | (F > 1 ? L.stuck[1] : 0u) << 8, (uint32_t)iters, 0u);
        if (tid < F) atomicAdd(&box->levels, (unsigned long long)L.levels[tid]);
        BFS_SYNC();
    }
#ifdef MC_SCOUT_TIMING
    if (tid == 0 && tsc[6]) printf("[scout companion, consensus hops] %llu hops, us each: words + tip %.2f, consensus %.2f, vertices + minimizers %.2f, look up %.2f, candidates %.2f, append + store words %.2f; reads that held the tip %.2f a hop, candidates named %.2f\n", tsc[6], tsc[0] * 0.01 / tsc[6], tsc[1] * 0.01 / tsc[6], tsc[2] * 0.01 / tsc[6], tsc[3] * 0.01 / tsc[6], tsc[4] * 0.01 / tsc[6], tsc[5] * 0.01 / tsc[6], (double)(tsc[7] >> 32) / tsc[6], (double)(tsc[7] & 0xFFFFFFFFull) / tsc[6]);
    else if (tid == 0 && tsc[5]) printf("[scout companion] %llu iterations, us each: fetch words %.2f, find tip + hash %.2f, look up %.2f, rest of eval %.2f, barriers + winner %.2f\n", tsc[5], tsc[0] * 0.01 / tsc[5], tsc[1] * 0.01 / tsc[5], tsc[2] * 0.01 / tsc[5], tsc[3] * 0.01 / tsc[5], tsc[4] * 0.01 / tsc[5]);
#endif
    if (lane == 0) {
        atomicAdd(&box->hops, hops);
        atomicAdd(&box->nf, n_nf);
        atomicAdd(&box->m0, n_m0);
        if (tid == 0) atomicAdd(&box->iters, iters);
    }
    (void)lookups;
}

// The narrow walk.  All threads of the workgroup call it at a level boundary with a frontier of
// F <= NARROW_CAND / nb vertices ("walkers"); it returns when the frontier is empty (done), too
// wide, distanceToKmer is nearly full, or the round budget is used up, and hands the state back in *ctl.
//
// A round looks up, in ONE memory round trip, the neighbours of every walker and of the next
// H - 1 vertices each walker is EXPECTED to visit (its scout's predicted path): one tree node per
// thread.  Then it finds the leading levels J in which every walker's neighbourhood held exactly what
// the sequential BFS needs to add exactly the expected vertex (anything else that is solid there is
// already in distanceToKmer): those F*J vertices are appended in level-major order, which is the
// sequential discovery order.  The first level that holds anything else (a branch, a dead end, a cycle,
// the cap, the radius) is left to the exact one-level replay with the LDS set (replay_slow, wave 0).
// Predictions only steer the guess; every guess is checked against the table and the visited index.
template <int MODE, bool SH = false>
__device__ __forceinline__ void bfs_narrow(const BfsState &S, const SolidView &t, NarrowLds &L, int k, int min_cov,
                           long long max_kmers, long long max_radius, unsigned long long rounds_budget,
                           unsigned long long &lookups, bool companion)
{
    BfsCtl *ctl = S.ctl;
    ScoutBox *box = S.box;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int dir = S.dir;
    const int nb = dir == 0 ? 8 : 4;
    const uint32_t lg = dir == 0 ? 3 : 2;    // log2(nb)
    const uint32_t flim = NARROW_CAND / nb;  // widest frontier one wave replays
#ifdef MC_BFS_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
    unsigned long long rounds = 0, slow_rounds = 0;
    uint32_t div_f = 0, div_m = 0;  // div_m = ceil(2^16 / div_f)
    if (tid < 64) {
        const unsigned long long lb = ctl_ld(&ctl->lb), le = ctl_ld(&ctl->le);
        const long long level = ctl_ld(&ctl->level);
        const uint32_t F = (uint32_t)(le - lb);
        bool dup = false;
        if (lane < F) {
            const uint64_t di = lb + lane;
            if (level == 0) dup = (ld_flags(&S.flags[di]) & 2u) != 0;
            L.fl_w[0][lane] = lane;
            L.fl_idx[0][lane] = (uint32_t)di | (dup ? 0x80000000u : 0u);
            Kmer r;
            r.hi = S.hi[di];
            r.lo = S.lo[di];
            L.root[lane] = r;
            L.wptr[lane] = 0;
            L.wright[lane] = dir > 0 ? 1 : 0;
        }
        if (lane < SCOUT_MAX_F) { L.plen[lane] = 0; L.ppos[lane] = 0; L.pdone[lane] = 0; }
        const bool any = __ballot(dup) != 0;
        if (lane == 0) {
            L.n = ctl_ld(&ctl->n); L.lb = lb; L.le = le; L.level = level; L.F = F; L.cur = 0;
            L.status = BFS_RUNNING; L.any_dup_root = any ? 1 : 0; L.rounds_left = rounds_budget; L.pend = 0;
            L.scout_budget = SCOUT_BUDGET0; L.scout_skip = 0; L.scout_wait = 1; L.hops = 0; L.hop_levels = 0; L.s_calls = 0; L.s_nf = 0; L.s_m0 = 0;
            L.force_slow = dir == 0 ? 1 : 0;  // (both directions: which way a walker moves is known once it has stepped)
            L.comp = companion && box != nullptr ? 1 : 0;
            L.req_seq = L.comp ? ld_u32(&box->req_seq) : 0;
            L.req_open = 0;
            L.root_bad = 0;
        }
    }
    BFS_SYNC();

    // the path words the next round reads, for every walker (requested while the previous round still appends)
    auto load_pseg = [&](uint32_t F) {
        if (tid < F * PSEG_WORDS && F <= (uint32_t)SCOUT_MAX_F) {
            const uint32_t a = tid / PSEG_WORDS, w = tid - a * PSEG_WORDS;
            const uint32_t wd = (L.ppos[a] >> 5) + w;
            L.pseg[a][w] = wd < PATH_WORDS ? ld_sc1(&S.path[(uint64_t)a * PATH_WORDS + wd]) : 0ull;
        }
    };

    for (;;) {
        // ---- uniform decisions from the shared walk state
        const uint32_t F = L.F;
        const unsigned long long n = L.n;
        const long long level = L.level;
        const int cur = L.cur;
        const uint32_t pend = L.pend;
        const uint32_t skip0 = L.scout_skip, budget0 = L.scout_budget, fslow0 = L.force_slow;  // (tid 0 changes them mid-round)
        const uint32_t comp0 = L.comp, open0 = L.req_open, seq0 = L.req_seq;
        // what the companion has published meanwhile (applied at the end of the round; a little stale is fine)
        uint32_t resp_now = 0;
        if (comp0 && open0 && tid < F) resp_now = ld_u32(&box->resp[tid]);
        bool dec_skip = false;
        if (F == 0) { if (tid == 0) L.status = BFS_DONE; break; }
        if (F > flim) break;
        if (n + (unsigned long long)MAX_NODES > S.dcap) { if (tid == 0) L.status = BFS_NEED_GROW; break; }
        if (L.rounds_left == 0) break;
        rounds++;

        const bool capped0 = max_kmers >= 0 && (long long)n >= max_kmers;
        const long long room = max_radius < 0 ? (long long)PATH_CAP : max_radius - level;  // levels that may still add
        const uint32_t FN = F << lg;  // nodes per level
        const bool quad_mm = MC_QUAD_MINIMIZER != 0 && F == 1 && dir != 0 && (SH ? t.owner_mm_k : t.mm_k) != 0;  // (see where the nodes' slots are worked out)
        const bool quad_hash = MC_QUAD_MINIMIZER != 0 && F == 1;  // hash keys in minimizer bins: the same for the bins of the k-mers' bases, in any direction
        // x / F for x <= 512 and F <= 16 is (x * ceil(2^16 / F)) >> 16: integer divisions cost a lone wave ~30 instructions
        // each.  F rarely changes.
        if (F != div_f) { div_f = F; div_m = (65536u + F - 1) / F; }
        // Hcap: how many levels a round may speculate at all (nodes, radius, whole levels under the cap)
        uint32_t H = 1, Hcap = 1;
        long long lv_left = 0;  // whole levels that may still be added
        const bool may_spec = !capped0 && room >= 2 && !L.any_dup_root && !fslow0 && F <= (uint32_t)SCOUT_MAX_F && t.reads != nullptr;
        if (may_spec) {
            Hcap = (((uint32_t)MAX_NODES >> lg) * div_m) >> 16;  // MAX_NODES / FN
            lv_left = room;
            if (max_kmers >= 0) lv_left = min(lv_left, (long long)(((unsigned long long)max_kmers - n) / F));
            if ((long long)Hcap > lv_left) Hcap = (uint32_t)lv_left;
        }
        // ---- the scouts: when a walker's predicted path is used up
        uint32_t avail = 0;
        if (may_spec && Hcap >= 2) {
            avail = 0xFFFFFFFFu;
            for (uint32_t a = 0; a < F; a++) avail = min(avail, L.plen[a] - L.ppos[a]);
            if (avail == 0 && skip0 != 0) dec_skip = true;  // (applied at the end of the round, behind its barriers)
            bool inline_scout = avail == 0 && skip0 == 0 && !comp0;
            if (avail == 0 && skip0 == 0 && comp0) {
                // ---- the companion scouts: ask it (a new request when there is none, or the last one is used up), then
                // wait until every walker has something ahead of it (or is known to have nothing)
                bool used_up = true;
                for (uint32_t a = 0; a < F; a++) used_up = used_up && L.pdone[a] != 0;
                // `avail` and `used_up` decide which barriers a wave meets, and the threads that poll the mailbox below store
                // to the very words they were read from (plen, pdone): no wave may get that far while another still reads.
                // (Round 3 had no barrier here.  A wave that shares its SIMD with another kernel's waves can be microseconds
                // late: it then saw a path where the others saw none, left this branch and ran one barrier out of step with
                // its workgroup from there on -- the vertices of the last round entered the index too late for the look-ups
                // that needed them, and the walk appended vertices it already had: gpurun_out/soak_r3.log:80.)
                BFS_DECIDED_SYNC();
                uint32_t seq = seq0;
                if (!open0 || used_up) {
                    seq = seq0 + 1;
                    if (tid < F) {
                        uint32_t p0 = L.wptr[tid];
                        if (p0 == 0) {  // the walker's k-mer has not been looked up with its pointer yet
                            (void)solid_get_kmer<MODE>(t, L.root[tid], k, (uint64_t)key_of<MODE>(L.root[tid], k), &p0);
                            lookups++;
                            L.wptr[tid] = p0;
                        }
                        box->root_hi[tid] = L.root[tid].hi;
                        box->root_lo[tid] = L.root[tid].lo;
                        box->ptr[tid] = p0;
                        box->right[tid] = L.wright[tid];
                    }
                    if (tid == 0) { box->F = F; box->budget = (uint32_t)min((long long)PATH_CAP, lv_left); }
                    BFS_PATH_RESET_EARLY(L, tid);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    BFS_SYNC();
                    // behind the barrier: every wave has taken its decisions from the old values (the polling below starts
                    // with these threads' own stores, then a barrier)
                    BFS_PATH_RESET(L, tid);
                    if (tid == 0) { st_u32(&box->req_seq, seq); L.req_seq = seq; L.req_open = 1; }
                }
                bool ready = false;
                for (uint32_t polls = 0; polls <= BOX_WAIT_POLLS && !ready; polls++) {
                    if (tid < F) {
                        const uint32_t r = ld_u32(&box->resp[tid]);
                        if ((r >> 17) == (seq & 0x7FFFu)) { L.plen[tid] = r & 0xFFFFu; L.pdone[tid] = (r >> 16) & 1u; }
                    }
                    BFS_SYNC();
                    ready = true;
                    for (uint32_t a = 0; a < F; a++) ready = ready && (L.plen[a] > L.ppos[a] || L.pdone[a]);
                    BFS_SYNC();
                    if (!ready) __builtin_amdgcn_s_sleep(4);
                }
                if (ready) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    avail = 0xFFFFFFFFu;
                    for (uint32_t a = 0; a < F; a++) avail = min(avail, L.plen[a] - L.ppos[a]);
                    load_pseg(F);
                    BFS_SYNC();
                    if (tid == 0) {
                        if (avail == 0) { L.scout_skip = L.scout_wait; L.scout_wait = min(L.scout_wait * 2, 64u); L.req_open = 0; } else L.scout_wait = 1;
                    }
                } else {  // no answer: the companion is not there; this workgroup scouts for itself from now on
                    if (tid == 0) { L.comp = 0; L.req_open = 0; }
                    if (tid < SCOUT_MAX_F) { L.plen[tid] = 0; L.ppos[tid] = 0; L.pdone[tid] = 0; }
                    BFS_SYNC();
                    inline_scout = true;
                }
            }
            if (inline_scout) {
                const uint32_t budget = (uint32_t)min((long long)min(budget0, PATH_CAP), lv_left);
                BFS_DECIDED_SYNC();  // scout_run ends with stores to plen / ppos, which every wave read for `avail` above
                if (wv < F) scout_run<MODE, SH>(S, t, L, wv, k, min_cov, budget, lookups);
                BFS_SYNC();
                avail = 0xFFFFFFFFu;
                for (uint32_t a = 0; a < F; a++) avail = min(avail, L.plen[a]);
                load_pseg(F);
                BFS_SYNC();
                if (tid == 0) {
                    if (avail >= budget && budget0 < PATH_CAP) L.scout_budget = budget0 * 2;
                    // a walker nobody can predict (no pointer, a dead end ahead): plain levels for a while, longer every time
                    if (avail == 0) { L.scout_skip = L.scout_wait; L.scout_wait = min(L.scout_wait * 2, 64u); } else L.scout_wait = 1;
                }
            }
            H = max(1u, min(avail, Hcap));
        }
        const bool spec = H >= 2;
        if (tid == 0) L.bad_lvl = 0xFFFFFFFFu;

        // ---- speculate: node (i, a, c) = c-th neighbour of the vertex walker a is expected to reach after i-1 steps
        MC_STAMP(0);
        bool have = false;
        Kmer nk{0, 0};
        uint32_t ni = 0, na = 0;
        bool npred = false, nflip = false;
        int cov = -1;
        uint32_t naux = 0;
        uint64_t key = 0, s0 = 0;
        TableRef h = own_table(t);  // the table the node's k-mer lives in (SH, several GPUs: its owner's)
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0;
        bool root_bad = false;
        if (tid < H * FN) {
            const uint32_t tl = tid;
            const uint32_t lvl = ((tl >> lg) * div_m) >> 16;  // tl / FN
            ni = lvl + 1;
            const uint32_t r = tl - lvl * FN;
            na = r >> lg;
            const uint32_t c = r & (uint32_t)(nb - 1);
            Kmer v = L.root[na];
            if (spec) {
                const bool right = L.wright[na] != 0;
                const uint32_t q = (L.ppos[na] & 31u) + lvl;
                uint32_t hb;  // the base the walker's next step appends (walk strand)
                Kmer x;       // the walker's vertex after lvl steps, on its walk strand
                if (MODE == KEY_PACKED) {  // k <= 31: the vertex and the base behind it come out of one 32-base window
                    const uint64_t A = bases32(L.pseg[na], q);
                    x = Kmer{0, A >> (64 - 2 * k)};
                    hb = (uint32_t)(A >> (62 - 2 * k)) & 3u;
                } else {
                    x = extract_kmer(L.pseg[na], q, k);
                    const uint32_t qn = q + (uint32_t)k;
                    hb = (uint32_t)(L.pseg[na][qn >> 5] >> (62 - 2 * (qn & 31))) & 3u;
                }
                v = right ? x : kmer_rc<MODE>(x, k);
                // The path must start at the walker itself.  It does whenever the scout's words arrived in order; but a
                // path is only ever a guess, and a stale one -- another walker's, an earlier request's -- is a chain of the
                // graph that would pass every check below: the walker's own k-mer is the one thing they cannot share.
                root_bad = lvl == 0 && !kmer_eq(v, L.root[na]);  // (applied with the other checks, behind the barrier)
                const uint32_t ob = right ? hb : (3u ^ hb);  // ... as seen from the vertex itself
                const uint32_t cstar = dir == 0 ? (2 * ob + (right ? 1u : 0u)) : ob;
                npred = c == cstar;
            }
            if (MODE == KEY_PACKED) {  // everything in one 64-bit word, no branches (neighbour + key_of, specialised)
                const uint64_t kmask = (1ull << (2 * k)) - 1;
                const bool left = dir < 0 || (dir == 0 && !(c & 1));
                const uint64_t cc = dir == 0 ? (c >> 1) : c;
                const uint64_t xx = left ? ((v.lo >> 2) | (cc << (2 * (k - 1)))) : (((v.lo << 2) | cc) & kmask);
                nk = Kmer{0, xx};
                const uint64_t rcx = rc_packed(xx, k);
                nflip = rcx < xx;
                key = nflip ? rcx : xx;
            } else {
                nk = neighbour(v, k, dir, (int)c);
                key = (uint64_t)key_of<MODE>(nk, k, &nflip);
            }
            if (MODE == KEY_PACKED && quad_mm) {
                // The counting table's region is the bin of the k-mer's MINIMIZER (kmer_device.h), the smallest of the hashes of its
                // k - SK_M + 1 SK_M-mers -- 17 hashes a node when every thread works its own out (sk_hmin_of_kmer), two thirds of
                // what a round's threads compute before their look-ups go out.  One walker, one direction: the four nodes of a level
                // are the four lanes of a quad, neighbours of the SAME vertex v, and share all of v's SK_M-mers but one: each lane hashes
                // a quarter of those, the quad takes the smallest (two DPP steps), and a lane adds the one SK_M-mer that holds its own base.
                const uint32_t w1 = (uint32_t)k - SK_M;  // SK_M-mers of v that the neighbour keeps: v's all but its first (right) / last (left)
                auto mm_hash = [](uint32_t f) {
                    uint32_t r = __builtin_bitreverse32(f);
                    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
                    r = (~r) >> (32 - 2 * SK_M);
                    return sk_order(f < r ? f : r);
                };
                const bool lft = dir < 0;
                uint32_t hm = SK_NONE;
                for (uint32_t p = c + (lft ? 0u : 1u); p < w1 + (lft ? 0u : 1u); p += 4)  // (SK_M-mer p of v: its bases p .. p + SK_M - 1)
                    hm = min(hm, mm_hash((uint32_t)(v.lo >> (2u * ((uint32_t)k - SK_M - p))) & SK_MMASK));
                hm = min(hm, (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)hm, 0xB1, 0xF, 0xF, false));  // quad_perm [1, 0, 3, 2]
                hm = min(hm, (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)hm, 0x4E, 0xF, 0xF, false));  // quad_perm [2, 3, 0, 1]
                hm = min(hm, mm_hash((uint32_t)(lft ? nk.lo >> (2u * ((uint32_t)k - SK_M)) : nk.lo) & SK_MMASK));
                if (SH) s0 = solid_locate(t, key, h, true, hm);
                else s0 = ((((uint64_t)sk_bin(hm) * t.n_regions) >> 32) << MC_REGION_LG) | sk_home(key);
#ifdef MC_BFS_CHECK_SLOTS
                { TableRef h2 = own_table(t); const uint64_t want = SH ? solid_locate(t, key, h2) : solid_slot_of(t, key);
                  if (want != s0) printf("[bfs] quad minimizer: slot %llu, solid_slot_of says %llu (key %llx, level %u, c %u, dir %d)\n", (unsigned long long)s0, (unsigned long long)want, (unsigned long long)key, lvl, c, dir); }
#endif
            } else if (MODE != KEY_PACKED && !SH && t.mm_k != 0 && quad_hash) {
                // Hash keys in minimizer bins (count_long.h): the region is the bin of the k-mer's BASES -- 49 SK_M-mer hashes a node at
                // k = 63 when every thread works its own out (sk_hmin_of_kmer2), most of what a round computed (the CLI's walk at
                // k = 63: 19 ms on such a table against 12 on a hash-prefix one).  One walker: the nb nodes of a level are nb lanes side
                // by side, neighbours of the SAME vertex v, and share v's SK_M-mers but the one at either end: the lanes hash a share
                // each, the group takes the smallest (the two smallest, as a multiset, where the table's bin word is made of two:
                // mm_k < 0) over DPP, and a lane adds what only its side keeps and the SK_M-mer that holds its own base.
                const bool two = t.mm_k < 0;
                const uint32_t wq = (uint32_t)k - SK_M;  // SK_M-mer p of a k-mer: its bases p .. p + SK_M - 1, p = 0 .. wq
                auto mer_at = [&](const Kmer &x, uint32_t p) -> uint32_t {
                    const uint32_t o = 2u * (wq - p);
                    const uint64_t w = o >= 64 ? (x.hi >> (o - 64)) : (o == 0 ? x.lo : ((x.lo >> o) | (x.hi << (64 - o))));
                    return (uint32_t)w & SK_MMASK;
                };
                auto mm_hash = [](uint32_t f) {
                    uint32_t r = __builtin_bitreverse32(f);
                    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
                    r = (~r) >> (32 - 2 * SK_M);
                    return sk_order(f < r ? f : r);
                };
                uint32_t m_lo = SK_NONE, m_hi = SK_NONE;  // the two smallest so far
                auto take = [&](uint32_t x) { const uint32_t tm = max(m_lo, x); m_lo = min(m_lo, x); m_hi = min(m_hi, tm); };
                auto merge = [&](uint32_t o_lo, uint32_t o_hi) { const uint32_t tm = max(m_lo, o_lo); m_lo = min(m_lo, o_lo); m_hi = min(tm, min(m_hi, o_hi)); };
                const bool lft = dir < 0 || (dir == 0 && !(c & 1));
                if (dir != 0) {  // four lanes; the neighbours keep v's SK_M-mers 0 .. wq - 1 (left) or 1 .. wq (right)
                    for (uint32_t p = c + (lft ? 0u : 1u); p < wq + (lft ? 0u : 1u); p += 4) take(mm_hash(mer_at(v, p)));
                } else {         // eight lanes, left and right neighbours in turn: all keep 1 .. wq - 1
                    for (uint32_t p = 1u + c; p < wq; p += 8) take(mm_hash(mer_at(v, p)));
                }
                {
                    uint32_t o_lo = (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)m_lo, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
                    uint32_t o_hi = (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)m_hi, 0xB1, 0xF, 0xF, false);
                    merge(o_lo, o_hi);
                    o_lo = (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)m_lo, 0x4E, 0xF, 0xF, false);           // quad_perm [2, 3, 0, 1]
                    o_hi = (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)m_hi, 0x4E, 0xF, 0xF, false);
                    merge(o_lo, o_hi);
                    if (dir == 0) {
                        o_lo = (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)m_lo, 0x141, 0xF, 0xF, false);      // row_half_mirror: the other quad of my eight
                        o_hi = (uint32_t)__builtin_amdgcn_update_dpp((int)SK_NONE, (int)m_hi, 0x141, 0xF, 0xF, false);
                        merge(o_lo, o_hi);
                        take(mm_hash(mer_at(v, lft ? 0u : wq)));  // what only my side keeps of v
                    }
                }
                take(mm_hash(mer_at(nk, lft ? 0u : wq)));  // the SK_M-mer that holds my own base
                const uint32_t hm = two ? m_lo ^ (m_hi * 0x85EBCA6Bu) : m_lo;  // (count_long.h skl_word2)
                s0 = ((((uint64_t)(sk_bin(hm) & 0xFFFFFF00u) * t.n_regions) >> 32) << MC_REGION_LG) | sk_home(key);  // (skl_bin)
#ifdef MC_BFS_CHECK_SLOTS
                { TableRef h2 = own_table(t); const uint64_t want = solid_locate_kmer<MODE>(t, nk, k, key, h2);
                  if (want != s0) printf("[bfs] group minimizer (hash keys): slot %llu, solid_locate_kmer says %llu (level %u, c %u, dir %d)\n", (unsigned long long)s0, (unsigned long long)want, lvl, c, dir); }
#endif
            } else {
                if (MODE != KEY_PACKED && !SH && t.mm_k != 0) s0 = solid_locate_kmer<MODE>(t, nk, k, key, h);  // (hash keys in minimizer bins)
                else s0 = SH ? solid_locate(t, key, h) : solid_slot_of(t, key);
            }
            const uint64_t s1 = (s0 & ~(uint64_t)h.rmask) | ((s0 + 1) & h.rmask);
            a0 = *reinterpret_cast<const uint4 *>(h.slots + s0);
            a1 = *reinterpret_cast<const uint4 *>(h.slots + s1);
#if MC_ROUND_PROBES == 4
            a2 = *reinterpret_cast<const uint4 *>(h.slots + ((s0 & ~(uint64_t)h.rmask) | ((s0 + 2) & h.rmask)));
            a3 = *reinterpret_cast<const uint4 *>(h.slots + ((s0 & ~(uint64_t)h.rmask) | ((s0 + 3) & h.rmask)));
#endif
            lookups++;
            have = true;
        }
        // the previous round's vertices enter the index while this round's probes are in flight (taken from the top of
        // the workgroup, where threads usually hold no node)
        if (BFS_THREADS - 1 - tid < pend) vis_insert(S, L.pub_k[BFS_THREADS - 1 - tid], L.pub_idx[BFS_THREADS - 1 - tid]);
        MC_STAMP(1);
        if (have) {
#if MC_ROUND_PROBES == 4
            cov = solid_get4l(t, h, key, s0, a0, a1, a2, a3, &naux);
#else
            cov = solid_get2(t, h, key, s0, a0, a1, &naux);
#endif
            L.cov[tid] = (int16_t)cov;
            L.kmer[tid] = nk;
            L.naux[tid] = naux;
            if (npred) L.pk[(ni - 1) * F + na] = nk;
        }
        L.set[tid] = LH_EMPTY;
        L.set[tid + BFS_THREADS] = LH_EMPTY;
        BFS_SYNC();
        MC_STAMP(2);
        // is a solid node already in distanceToKmer when the sequential BFS meets it?  = in the index,
        // or one of the expected vertices that come earlier in level-major order.  The expected
        // vertices go into an LDS set keyed by k-mer that keeps the earliest position of each.
        const bool solid = have && cov >= min_cov;
        const uint32_t pos = have ? (ni - 1) * F + na : 0;  // my walker's place in level-major order at my level
        uint32_t myslot = 0;
        if (npred) myslot = lds_set_min(L.set, RH_SIZE - 1, L.pk, nk, pos);
        bool ind = false;
        if (solid) ind = vis_find(S, nk);
        BFS_SYNC();
        if (solid && !ind && H > 1) {
            if (npred) {
                ind = L.set[myslot] != pos;  // the same k-mer is expected earlier on some walker's path
            } else {  // a solid neighbour off the expected path: fine only if it is an expected vertex met earlier
                uint32_t sl = (uint32_t)vis_hash(nk) & (RH_SIZE - 1);
                for (;;) {
                    const uint32_t e = L.set[sl];
                    if (e == LH_EMPTY) break;
                    const Kmer o = L.pk[e];
                    if (o.lo == nk.lo && o.hi == nk.hi) { ind = e < pos; break; }
                    sl = (sl + 1) & (RH_SIZE - 1);
                }
            }
        }
        if (have) {
            L.vis[tid] = ind ? 1 : 0;  // (level 1 only matters: the one-level replay reads it)
            const bool ok = npred ? (solid && !ind) : (!solid || ind);
            if (!ok && H > 1) atomicMin(&L.bad_lvl, ni);
            if (root_bad) L.root_bad = 1;
        }
        BFS_SYNC();
        MC_STAMP(3);
        uint32_t J = 0;
        if (H > 1) J = min(H, L.bad_lvl - 1);
        if (L.root_bad) {
            // the path does not start at the walker: none of this round's nodes means anything (the one-level replay below
            // takes level 1 to be the walkers' own neighbours).  The paths are dropped and the next round is a plain one.
            BFS_SYNC();
            if (tid < SCOUT_MAX_F) { L.plen[tid] = 0; L.ppos[tid] = 0; L.pdone[tid] = 0; }
            if (tid == 0) {
                L.root_bad = 0;
                L.force_slow = 1;
                L.req_open = 0;
                L.pend = 0;  // (indexed above, in front of this round's first barrier)
                L.rounds_left--;
                if (dec_skip) L.scout_skip = skip0 - 1;
                atomicAdd(&ctl->slow_mismatch, 1ull);
                BFS_TRACE(S, 3u << 24 | (uint32_t)(rounds & 0xFFFFFF), n, F | H << 8, level, pend, seq0 | open0 << 31, resp_now, L.plen[0] | L.ppos[0] << 16);
            }
            BFS_SYNC();
            continue;
        }

        if (J >= 1) {
            // ---- levels 1..J are exactly "every walker steps to its expected vertex"
            if (have && ni <= J) {
                if (npred) {
                    const uint64_t idx = n + (unsigned long long)(ni - 1) * F + na;
                    S.hi[idx] = nk.hi;
                    S.lo[idx] = nk.lo;
                    S.dist[idx] = (int32_t)(level + ni);
                    S.cov[idx] = (int16_t)cov;
                    L.pub_k[(ni - 1) * F + na] = nk;
                    L.pub_idx[(ni - 1) * F + na] = (uint32_t)idx;
                    if (ni == J) {  // the walker's new position and the read pointer stored there
                        L.wptr[na] = naux;
                        L.fl_idx[cur ^ 1][na] = (uint32_t)idx;
                    }
                } else if (solid) {  // solid but already there: lastKmers.add(parent)
                    const uint32_t pidx = ni == 1 ? (L.fl_idx[cur][na] & 0x3FFFFFFFu)
                                                  : (uint32_t)(n + (unsigned long long)(ni - 2) * F + na);
                    atomicOr(&S.flags[pidx], 1u);
                }
            }
            BFS_SYNC();
            if (tid < F) {
                L.root[tid] = L.pk[(J - 1) * F + tid];
                L.ppos[tid] += J;
                L.fl_w[cur ^ 1][tid] = tid;
            }
            if (tid == 0) {
                const uint32_t n_new = F * J;
                L.n = n + n_new;
                L.lb = n + (unsigned long long)(J - 1) * F;
                L.le = n + n_new;
                L.level = level + J;
                L.cur = cur ^ 1;
                L.any_dup_root = 0;
                L.pend = n_new;  // indexed while the next round's lookups are in flight
                L.rounds_left--;
                if (J < H) L.force_slow = 1;  // the level behind them is something else: replayed exactly by the next round
                if (dec_skip) L.scout_skip = skip0 - 1;
                BFS_TRACE(S, 1u << 24 | (uint32_t)(rounds & 0xFFFFFF), n, F | H << 8 | J << 16 | fslow0 << 24 | comp0 << 25, level, pend | min(L.bad_lvl, 0xFFFFu) << 16,
                          seq0 | open0 << 31, resp_now, L.plen[0] | L.ppos[0] << 16);
            }
            if (comp0 && open0 && tid < F && L.req_seq == seq0 && (resp_now >> 17) == (seq0 & 0x7FFFu)) {  // how far the companion has got meanwhile (same request)
                if ((resp_now & 0xFFFFu) >= L.plen[tid]) {  // (the wait above may have seen a later answer already)
                    L.plen[tid] = resp_now & 0xFFFFu;
                    L.pdone[tid] = (resp_now >> 16) & 1u;
                }
            }
            BFS_SYNC();
            load_pseg(F);
            MC_STAMP(4);
        } else {
            // ---- exact one-level replay of level 1 (ids [0, FN) are the plain neighbour sets of the walkers)
            slow_rounds++;
            if (tid < FN && H > 1) L.vis[tid] = (solid && vis_find(S, nk)) ? 1 : 0;  // index only, no expectations
            BFS_SYNC();
            if (tid < 64) {
                uint32_t last_base = 0;
                const uint32_t n_new = replay_slow(S, L, 1, min_cov, max_kmers, max_radius, lg, flim, &last_base);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                for (uint32_t i = lane; i < n_new; i += 64) vis_insert(S, L.kmer[L.new_id[i]], L.new_idx[i]);
                const uint32_t Fn = L.F;
                const int c2 = L.cur;
                Kmer nr{0, 0};
                uint32_t np = 0;
                bool right = dir > 0;
                if (lane < Fn && Fn <= flim) {
                    const uint32_t id = last_base + L.fl_w[c2][lane];
                    nr = L.kmer[id];
                    if (dir == 0) right = (id & 1u) != 0;  // odd neighbour index = right neighbour
                    np = L.naux[id];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                if (lane < Fn && Fn <= flim) {
                    L.root[lane] = nr;
                    L.wptr[lane] = np;
                    L.wright[lane] = right ? 1 : 0;
                    L.fl_w[c2][lane] = lane;  // walkers are numbered 0..F-1 in the next round
                }
                if (lane < SCOUT_MAX_F) { L.plen[lane] = 0; L.ppos[lane] = 0; L.pdone[lane] = 0; }  // the walkers changed: their paths are void
                if (lane == 0) {
                    L.any_dup_root = 0;
                    L.pend = 0;
                    L.rounds_left--;
                    if (fslow0) L.scout_budget = SCOUT_BUDGET0;
                    if (fslow0) atomicAdd(&ctl->slow_forced, 1ull); else if (H > 1) atomicAdd(&ctl->slow_mismatch, 1ull); else atomicAdd(&ctl->slow_starved, 1ull);
                    L.force_slow = 0;
                    L.req_open = 0;  // (the companion's run, if any, is for walkers that no longer exist)
                    if (dec_skip) L.scout_skip = skip0 - 1;
                    BFS_TRACE(S, 2u << 24 | (uint32_t)(rounds & 0xFFFFFF), n, F | H << 8 | n_new << 16 | fslow0 << 24 | comp0 << 25, level, pend | Fn << 16,
                              seq0 | open0 << 31, resp_now, L.n);
                }
            }
            MC_STAMP(4);
        }
        BFS_SYNC();
        MC_STAMP(5);
    }
    BFS_SYNC();
    {   // vertices of the last fast round still waiting for the index
        const uint32_t pend = L.pend;
        if (tid < pend) vis_insert(S, L.pub_k[tid], L.pub_idx[tid]);
    }
    BFS_SYNC();
    if (tid == 0) {
        L.pend = 0;
        ctl_st(&ctl->n, L.n);
        ctl_st(&ctl->lb, L.lb);
        ctl_st(&ctl->le, L.le);
        ctl_st(&ctl->c0, 0);
        ctl_st(&ctl->level, L.level);
        ctl->rounds_narrow += rounds;
        ctl->rounds_slow += slow_rounds;
        ctl->scout_hops += L.hops;
        ctl->scout_levels += L.hop_levels;
        ctl->scout_calls += L.s_calls;
        ctl->scout_nf += L.s_nf;
        ctl->scout_m0 += L.s_m0;
        if (L.status != BFS_RUNNING) ctl_st(&ctl->status, L.status);
#ifdef MC_BFS_TIMING
        for (int i = 0; i < 8; i++) ctl->tacc[i] += tacc[i];
#endif
    }
    BFS_SYNC();
}

union BfsLds {
    WideLds w;
    NarrowLds n;
};

// The scouts of a batch's jobs, one workgroup each, as a kernel of their own beside k_bfs (on another stream): its registers are
// allotted for the hops alone.  (Rounds 2-4 ran them as the odd workgroups of k_bfs: one kernel, so one register allotment for the
// walk and the scouts together -- 177 vector registers and ~290 scalar values parked in vector lanes, every use of one an
// instruction of a lone wave.)  Correct whether or not the two grids are on the chip together: scout_companion says why.
template <int MODE, bool SH = false>
__global__ void __launch_bounds__(BFS_THREADS) k_bfs_scout(const BfsState *__restrict__ states, SolidView t, int k, int min_cov)
{
    __shared__ TeamLds lds;
    const BfsState S = states[blockIdx.x];
    if (S.box && t.reads) scout_companion<MODE, SH>(S, t, lds, k, min_cov);
}

template <int MODE, bool SH = false>
__global__ void __launch_bounds__(BFS_THREADS) k_bfs(const BfsState *__restrict__ states, SolidView t, int k,
                                                     int min_cov, long long max_kmers, long long max_radius,
                                                     unsigned long long max_rounds, int companions)
{
    // companions: k_bfs_scout runs beside this grid, a workgroup per job that scouts for this one (scout_companion)
    __shared__ BfsLds lds;
    const BfsState S = states[blockIdx.x];
    BfsCtl *ctl = S.ctl;
    const uint32_t tid = threadIdx.x;
    if (ctl_ld(&ctl->status) != BFS_RUNNING) {  // finished (or waiting for the host) in an earlier launch
        if (companions && S.box && tid == 0) st_u32(&S.box->quit, 1u);
        return;
    }
    unsigned long long lookups = 0, rounds_left = max_rounds, chunks = 0;
    const int dir = S.dir;
    const int nb = dir == 0 ? 8 : 4;
    const uint32_t flim = NARROW_CAND / nb;

    // seeds: every window with reads.get(key) >= minOccurences, in order (:159-192)
    if (!ctl_ld(&ctl->seeds_done)) {
        // (the chunk counter is read ONCE: behind the loop thread 0 resets it, and a wave that came late to a second read
        // would see 0 and start over, alone, one barrier out of step with the others)
        BFS_CHUNK_LOOP(c0) {
            if (c0 >= S.n_seeds) break;
            if (ctl_ld(&ctl->n) + BFS_THREADS > S.dcap) {
                if (tid == 0) ctl_st(&ctl->status, BFS_NEED_GROW);
                goto out;
            }
            if (rounds_left == 0) goto out;
            rounds_left--;
            chunks++;
            const uint64_t r = c0 + tid;
            const bool have = r < S.n_seeds;
            Kmer cand{0, 0};
            if (have) { cand.hi = S.seed_hi ? S.seed_hi[r] : 0; cand.lo = S.seed_lo[r]; }
            BFS_SYNC();
            bfs_chunk_wide<MODE>(S, t, lds.w, k, min_cov, -1, true, have, cand, UINT64_MAX, 0, lookups);
            if (tid == 0) ctl_st(&ctl->c0, c0 + BFS_THREADS);
            BFS_SYNC();
        }
        if (tid == 0) {
            ctl_st(&ctl->seeds_done, 1);
            ctl_st(&ctl->lb, 0);
            ctl_st(&ctl->le, ctl_ld(&ctl->n));
            ctl_st(&ctl->c0, 0);
            ctl_st(&ctl->level, 0);
        }
        BFS_SYNC();
    }

    for (;;) {
        const unsigned long long lb = ctl_ld(&ctl->lb), le = ctl_ld(&ctl->le);
        if (le == lb) {
            if (tid == 0) ctl_st(&ctl->status, BFS_DONE);
            break;
        }
        if (ctl_ld(&ctl->status) != BFS_RUNNING) break;
        if (rounds_left == 0) break;
        if (ctl_ld(&ctl->c0) == 0 && le - lb <= flim) {
            BFS_SYNC();
            bfs_narrow<MODE, SH>(S, t, lds.n, k, min_cov, max_kmers, max_radius, rounds_left, lookups, companions != 0);
            rounds_left = lds.n.rounds_left;
            BFS_SYNC();
            if (ctl_ld(&ctl->status) != BFS_RUNNING) break;  // done, or distanceToKmer must grow
            if (ctl_ld(&ctl->le) - ctl_ld(&ctl->lb) <= flim && ctl_ld(&ctl->le) != ctl_ld(&ctl->lb)) break;  // round budget used up: relaunch
            continue;
        }
        const long long level = ctl_ld(&ctl->level);
        const bool radius_ok = max_radius < 0 || level + 1 <= max_radius;  // newDistance > threshold -> false
        const unsigned long long ncand = (le - lb) * (unsigned long long)nb;
        BFS_CHUNK_LOOP(c0) {  // (read once: see the seeds' loop)
            if (c0 >= ncand) break;
            if (ctl_ld(&ctl->n) + BFS_THREADS > S.dcap) {
                if (tid == 0) ctl_st(&ctl->status, BFS_NEED_GROW);
                goto out;
            }
            if (rounds_left == 0) goto out;
            rounds_left--;
            chunks++;
            const unsigned long long rank = c0 + tid;
            const bool have = rank < ncand;
            Kmer cand{0, 0};
            uint64_t parent = 0;
            if (have) {
                parent = lb + rank / nb;
                const Kmer pv{S.hi[parent], S.lo[parent]};
                cand = neighbour(pv, k, dir, (int)(rank % nb));
            }
            BFS_SYNC();
            bfs_chunk_wide<MODE>(S, t, lds.w, k, min_cov, max_kmers, radius_ok, have, cand, parent,
                                 (int32_t)(level + 1), lookups);
            if (tid == 0) ctl_st(&ctl->c0, c0 + BFS_THREADS);
            BFS_SYNC();
        }
        if (tid == 0) {
            ctl_st(&ctl->lb, le);
            ctl_st(&ctl->le, ctl_ld(&ctl->n));
            ctl_st(&ctl->c0, 0);
            ctl_st(&ctl->level, level + 1);
        }
        BFS_SYNC();
    }
out:
    atomicAdd(&ctl->lookups, lookups);
    if (tid == 0) atomicAdd(&ctl->chunks_wide, chunks);
    if (companions && S.box && tid == 0) st_u32(&S.box->quit, 1u);  // (on every way out: the companion leaves with us)
}

// rebuild the index of distanceToKmer after the host enlarged it
__global__ void k_vis_rebuild(BfsState S, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Kmer v{S.hi[i], S.lo[i]};
        vis_insert(S, v, (uint32_t)i);
    }
}


// ---- a device-side check of a finished walk (debug: MC_BFS_SELFCHECK=1 makes mc_bfs_batch run it behind every job) ------
// What a result of OneSequenceCalculator.runBfs (src/algo/OneSequenceCalculator.java:154-214) must satisfy whatever the
// order of events on the device was:
//   dup      no two entries hold the same oriented k-mer (distanceToKmer is a map);
//   cov      every entry's coverage is what the table holds for its key, and >= minOccurences;
//   order    distances never decrease along the insertion order (a queue BFS), seeds first;
//   orphan   an entry with dist > 0 is a `dir`-neighbour of an entry with dist - 1 that comes before it;
//   open     (only when neither --maxkmers nor --maxradius cut the walk) every solid neighbour of an entry is an entry.
// The first violations are written down with the entries they concern.
struct BfsCheck {
    unsigned long long dup, cov, order, orphan, open;
    uint32_t n_first;
    uint32_t first[16][4];  // kind (1 dup, 2 cov, 3 order, 4 orphan, 5 open), entry, the other entry / the table's value, distance
};
constexpr uint32_t CHK_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ void bfs_check_note(BfsCheck *out, uint32_t kind, uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t i = atomicAdd(&out->n_first, 1u);
    if (i < 16) { out->first[i][0] = kind; out->first[i][1] = a; out->first[i][2] = b; out->first[i][3] = c; }
}

// entry holding k-mer v, or CHK_EMPTY
__device__ __forceinline__ uint32_t bfs_check_find(const BfsState &S, const uint32_t *set, uint32_t mask, const Kmer &v)
{
    uint32_t s = (uint32_t)vis_hash(v) & mask;
    for (uint32_t probe = 0; probe <= mask; probe++) {
        const uint32_t e = set[s];
        if (e == CHK_EMPTY) return CHK_EMPTY;
        if (S.lo[e] == v.lo && S.hi[e] == v.hi) return e;
        s = (s + 1) & mask;
    }
    return CHK_EMPTY;
}

// pass 1: an index of the result built from nothing but the result (set: mask + 1 words, all CHK_EMPTY)
__global__ void __launch_bounds__(256) k_bfs_check_index(BfsState S, uint64_t n, uint32_t *set, uint32_t mask, BfsCheck *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Kmer v{S.hi[i], S.lo[i]};
        uint32_t s = (uint32_t)vis_hash(v) & mask;
        for (uint32_t probe = 0; probe <= mask; probe++) {
            uint32_t e = set[s];
            if (e == CHK_EMPTY) e = atomicCAS(&set[s], CHK_EMPTY, (uint32_t)i);
            if (e == CHK_EMPTY) break;
            if (S.lo[e] == v.lo && S.hi[e] == v.hi) {
                atomicAdd(&out->dup, 1ull);
                bfs_check_note(out, 1, (uint32_t)max((uint64_t)e, i), (uint32_t)min((uint64_t)e, i), (uint32_t)S.dist[i]);
                break;
            }
            s = (s + 1) & mask;
        }
    }
}

// pass 2
template <int MODE>
__global__ void __launch_bounds__(256) k_bfs_check(BfsState S, uint64_t n, SolidView t, int k, int min_cov, long long max_kmers,
                                                   long long max_radius, const uint32_t *set, uint32_t mask, BfsCheck *out)
{
    const int dir = S.dir, nb = dir == 0 ? 8 : 4;
    const bool uncut = (max_kmers < 0 || (long long)n < max_kmers);  // the cap never refused an addition
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Kmer v{S.hi[i], S.lo[i]};
        const int32_t d = S.dist[i];
        const int have = solid_get_kmer<MODE>(t, v, k, (uint64_t)key_of<MODE>(v, k));
        if (have != (int)S.cov[i] || have < min_cov) {
            atomicAdd(&out->cov, 1ull);
            bfs_check_note(out, 2, (uint32_t)i, (uint32_t)have, (uint32_t)d);
        }
        if (i && S.dist[i - 1] > d) {
            atomicAdd(&out->order, 1ull);
            bfs_check_note(out, 3, (uint32_t)i, (uint32_t)S.dist[i - 1], (uint32_t)d);
        }
        if (d > 0) {  // some parent: v is the c-th dir-neighbour of p  <=>  p is one of v's neighbours the other way
            bool found = false;
            for (int c = 0; c < nb && !found; c++) {
                const Kmer p = neighbour(v, k, dir == 0 ? 0 : -dir, c);
                const uint32_t e = bfs_check_find(S, set, mask, p);
                found = e != CHK_EMPTY && e < i && S.dist[e] == d - 1;
            }
            if (!found) {
                atomicAdd(&out->orphan, 1ull);
                bfs_check_note(out, 4, (uint32_t)i, 0, (uint32_t)d);
            }
        }
        if (uncut && (max_radius < 0 || d + 1 <= max_radius)) {
            for (int c = 0; c < nb; c++) {
                const Kmer q = neighbour(v, k, dir, c);
                if (solid_get_kmer<MODE>(t, q, k, (uint64_t)key_of<MODE>(q, k)) >= min_cov && bfs_check_find(S, set, mask, q) == CHK_EMPTY) {
                    atomicAdd(&out->open, 1ull);
                    bfs_check_note(out, 5, (uint32_t)i, (uint32_t)c, (uint32_t)d);
                }
            }
        }
    }
}

}  // namespace mc

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
//             walk is bound by HBM latency, so wave 0 looks several levels ahead in one round
//             trip, guided by the read-context hints stored next to each k-mer, and falls back to
//             an exact one-level replay wherever the graph is not a plain path (bfs_narrow).
#pragma once
#include "kmer_device.h"

namespace mc {

constexpr int BFS_THREADS = 512;          // 8 waves
constexpr int MAX_NODES = BFS_THREADS;    // neighbour sets looked up per round (levels x walkers x nb): one per thread
constexpr int MAX_DEPTH = 5;
constexpr int NARROW_CAND = 64;           // candidates per replayed level = lanes of one wave
constexpr int RH_SIZE = 1024;             // round-local LDS set (narrow, slow replay)
constexpr int WH_SIZE = 2 * BFS_THREADS;  // chunk-local LDS set (wide)
constexpr uint64_t VIS_EMPTY = ~0ull;
constexpr uint32_t LH_EMPTY = 0xFFFFFFFFu;

enum { BFS_RUNNING = 0, BFS_DONE = 1, BFS_NEED_GROW = 2 };

struct BfsCtl {
    unsigned long long n;       // |distanceToKmer|
    unsigned long long lb, le;  // current frontier = entries [lb, le)
    unsigned long long c0;      // next candidate rank inside the frontier (wide path)
    unsigned long long lookups;
    unsigned long long rounds_narrow, rounds_slow, chunks_wide;
    unsigned long long tacc[8];  // MC_BFS_TIMING builds: 10 ns ticks per phase of a narrow round
    long long level;            // distance of the frontier
    int status;
    int seeds_done;
};

struct BfsState {
    uint64_t *hi, *lo;  // distanceToKmer keys in insertion order
    int32_t *dist;
    int16_t *cov;
    uint32_t *flags;    // bit0: in lastKmers; bit1: seed window queued more than once.  Pre-zeroed.
    uint64_t dcap;
    uint64_t *vis;      // index of the arrays above: buckets of two (fingerprint << 32 | index) entries
    uint64_t bmask;     // number of buckets - 1
    BfsCtl *ctl;
    const uint64_t *seed_hi, *seed_lo;
    uint64_t n_seeds;
    int dir;
};

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
    __syncthreads();
    uint32_t before = 0, tot = 0;
    const uint32_t n_waves = blockDim.x >> 6;
    for (uint32_t i = 0; i < n_waves; i++) {
        const uint32_t c = lds_wave_tot[i];
        if (i < wv) before += c;
        tot += c;
    }
    __syncthreads();
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
    uint64_t rhint[NARROW_CAND], nhint[NARROW_CAND];  // their oriented hints (walker_hint)
    Kmer sroot[2][NARROW_CAND];       // later stages of a round: where each walker is expected to be when its hint runs out,
    uint64_t shint[2][NARROW_CAND];   // and that vertex's own hint (length 0: no further stage); [stage & 1]
    uint64_t nslot[MAX_NODES];        // solid-table slot of each level-1 node (one-level replay)
    int16_t cov[MAX_NODES];
    uint8_t vis[MAX_NODES];
    uint8_t flip[MAX_NODES];
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
    int cur, status, any_dup_root;
};

union BfsLds {
    WideLds w;
    NarrowLds n;
};

// ---- wide path: one chunk of <= BFS_THREADS candidates in rank order.  parent == UINT64_MAX marks a
// seed window (src/algo/OneSequenceCalculator.java:159-192: queued when reads.get(key) >= minOccurences).
template <int MODE>
__device__ void bfs_chunk_wide(const BfsState &S, const SolidView &t, WideLds &L, int k, int min_cov,
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
        cov = solid_get(t, (uint64_t)key_of<MODE>(cand, k));
        lookups++;
    }
    const bool solid = have && cov >= min_cov;
    const unsigned long long n_before = ctl->n;
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
    __syncthreads();
    uint32_t slot = 0;
    if (contender) slot = lds_set_min(L.set, WH_SIZE - 1, L.kmer, cand, tid);
    __syncthreads();
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
    if (tid == 0) ctl->n = n_before + n_acc;
    __syncthreads();
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
__device__ inline uint32_t replay_slow(const BfsState &S, NarrowLds &L, int d, int min_cov, long long max_kmers,
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

// oriented hint of a walker: the bases it expects next, nearest first (bits 0..55), how many
// (bits 56..61), bit 62 = the walker moves right
__device__ __forceinline__ uint64_t walker_hint(uint64_t hr, uint64_t hl, bool flipped, bool right)
{
    uint64_t w = right ? (flipped ? hl : hr) : (flipped ? hr : hl);
    if (flipped) w = lh_complement(w);
    return (w & ~(3ull << 62)) | (right ? (1ull << 62) : 0);
}

// the vertex a walker reaches after i expected steps (i <= LHINT_MAX <= 28 < k is not required: i <= 28 and
// the shifts below stay under 64 bits)
__device__ __forceinline__ Kmer walker_at(const Kmer &root, int k, uint64_t hint, uint32_t i)
{
    if (i == 0) return root;
    const bool right = (hint >> 62) & 1;
    const uint32_t sh = 2 * i;  // 2 .. 56
    Kmer r;
    if (right) {  // ((root << 2i) | bases[0..i) with base 0 first) & mask
        const uint64_t blk = lh_block_forward(hint, i);
        r.hi = (root.hi << sh) | (root.lo >> (64 - sh));
        r.lo = (root.lo << sh) | blk;
        if (k <= 32) {
            r.hi = 0;
            if (k < 32) r.lo &= (1ull << (2 * k)) - 1;
        } else {
            r.hi &= (1ull << (2 * k - 64)) - 1;
        }
    } else {  // (root >> 2i) | (bases[i-1] ... bases[0]) << 2(k-i): the stored order already has base i-1 on top
        const uint64_t blk = hint & lh_mask(i);
        r.lo = (root.lo >> sh) | (root.hi << (64 - sh));
        r.hi = root.hi >> sh;
        const int pos = 2 * (k - (int)i);  // bit position of the block, >= 0 when i <= k
        if (pos >= 64) {
            r.hi |= blk << (pos - 64);
        } else {
            r.lo |= blk << pos;
            if (pos > 0 && pos + (int)sh > 64) r.hi |= blk >> (64 - pos);
        }
    }
    return r;
}

// lookup with the key/count halves of the first two probe slots already loaded
__device__ __forceinline__ int solid_get2(const SolidView &t, uint64_t key, uint64_t s0, uint64_t s1, const uint4 &a0,
                                          const uint4 &a1, uint64_t *found_slot)
{
    *found_slot = ~0ull;
    if (key == EMPTY_KEY) return solid_get(t, key);
    const uint64_t k0 = ((uint64_t)a0.y << 32) | a0.x;
    if (k0 == key) { *found_slot = s0; return a0.z > 32767u ? 32767 : (int)a0.z; }
    if (k0 == EMPTY_KEY) return -1;
    const uint64_t k1 = ((uint64_t)a1.y << 32) | a1.x;
    if (k1 == key) { *found_slot = s1; return a1.z > 32767u ? 32767 : (int)a1.z; }
    if (k1 == EMPTY_KEY) return -1;
    uint64_t s = s0;  // both probes hit other keys: walk the region
    const uint64_t base = s & ~(uint64_t)t.rmask;
    for (uint32_t probe = 0; probe <= t.rmask; probe++) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(t.slots + s);
        const uint64_t cur = ((uint64_t)raw.y << 32) | raw.x;
        if (cur == key) { *found_slot = s; return raw.z > 32767u ? 32767 : (int)raw.z; }
        if (cur == EMPTY_KEY) return -1;
        s = base | ((s + 1) & t.rmask);
    }
    return -1;
}

// The narrow walk.  All threads of the workgroup call it at a level boundary with a frontier of
// F <= NARROW_CAND / nb vertices ("walkers"); it returns when the frontier is empty (done), too
// wide, distanceToKmer is nearly full, or the round budget is used up, and hands the state back in *ctl.
//
// A round looks up, in ONE memory round trip, the neighbours of every walker and of the next
// H - 1 vertices each walker is EXPECTED to visit (the hint stored with the walker's k-mer says
// which bases followed it in the reads): one tree node per thread.  Then it finds the leading
// levels J in which every walker's neighbourhood held exactly what the sequential BFS needs to add
// exactly the expected vertex (anything else that is solid there is already in distanceToKmer):
// those F*J vertices are appended in level-major order, which is the sequential discovery order.
// The first level that holds anything else (a branch, a dead end, a wrong hint, a cycle, the cap,
// the radius) is left to the exact one-level replay with the LDS set (replay_slow, wave 0).
// Hints only steer the guess; every guess is checked against the table and the visited index.
template <int MODE>
__device__ void bfs_narrow(const BfsState &S, const SolidView &t, NarrowLds &L, int k, int min_cov,
                           long long max_kmers, long long max_radius, unsigned long long rounds_budget,
                           unsigned long long &lookups)
{
    BfsCtl *ctl = S.ctl;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
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
        const unsigned long long lb = ctl->lb, le = ctl->le;
        const long long level = ctl->level;
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
            L.rhint[lane] = dir > 0 ? (1ull << 62) : 0;  // no hint yet: the first round is a plain one-level round
        }
        const bool any = __ballot(dup) != 0;
        if (lane == 0) {
            L.n = ctl->n; L.lb = lb; L.le = le; L.level = level; L.F = F; L.cur = 0;
            L.status = BFS_RUNNING; L.any_dup_root = any ? 1 : 0; L.rounds_left = rounds_budget; L.pend = 0;
        }
    }
    __syncthreads();

    for (;;) {
        // ---- uniform decisions from the shared walk state
        const uint32_t F = L.F;
        const unsigned long long n = L.n;
        const long long level = L.level;
        const int cur = L.cur;
        const uint32_t pend = L.pend;
        if (F == 0) { if (tid == 0) L.status = BFS_DONE; break; }
        if (F > flim) break;
        if (n + (unsigned long long)MAX_NODES > S.dcap) { if (tid == 0) L.status = BFS_NEED_GROW; break; }
        if (L.rounds_left == 0) break;
        rounds++;

        const bool capped0 = max_kmers >= 0 && (long long)n >= max_kmers;
        const long long room = max_radius < 0 ? (long long)LHINT_MAX : max_radius - level;  // levels that may still add
        const uint32_t FN = F << lg;  // nodes per level
        // the round is as deep as the shortest hint among the walkers
        uint32_t hl_min = LHINT_MAX;
        for (uint32_t a = 0; a < F; a++) hl_min = min(hl_min, lh_len(L.rhint[a]));
        if (hl_min > (uint32_t)k) hl_min = (uint32_t)k;
        // x / F for x <= 512 and F <= 16 is (x * ceil(2^16 / F)) >> 16: integer divisions cost a lone wave ~30 instructions
        // each, and this loop is bound by how fast one wave issues instructions.  F rarely changes.
        if (F != div_f) { div_f = F; div_m = (65536u + F - 1) / F; }
        // Hcap: how many levels a round may speculate at all (nodes, radius, whole levels under the cap)
        uint32_t H = 1, Hcap = 1;
        if (hl_min >= 2 && !capped0 && room >= 1 && !L.any_dup_root) {
            Hcap = (((uint32_t)MAX_NODES >> lg) * div_m) >> 16;  // MAX_NODES / FN
            const long long room2 = max_radius < 0 ? (long long)(4 * LHINT_MAX) : room;
            if ((long long)Hcap > room2) Hcap = (uint32_t)room2;
            if (max_kmers >= 0) {  // whole levels under the cap: H <= (max_kmers - n) / F
                const unsigned long long rem = (unsigned long long)max_kmers - n;
                if (rem < (unsigned long long)Hcap * F) Hcap = (uint32_t)(((uint32_t)rem * div_m) >> 16);  // (rem < 512 here)
            }
            if (Hcap < 1) Hcap = 1;
            H = min(hl_min, Hcap);
        }
        // Further stages: when a walker's hint is used up before Hcap, the vertex it is expected to reach then brings
        // its own hint (its slot is among this stage's lookups), and the levels behind it are looked up in another
        // round trip of the same round -- the index work, checks and appends of a round are paid once for all stages.
        const bool spec = H >= 2;
        if (tid == 0) L.bad_lvl = 0xFFFFFFFFu;

        // ---- speculate: node (i, a, c) = c-th neighbour of the vertex walker a is expected to reach after i-1 steps
        MC_STAMP(0);
        bool have = false;
        Kmer nk{0, 0};
        uint32_t ni = 0, na = 0;
        bool npred = false, nflip = false;
        int cov = -1;
        uint64_t nslot = ~0ull;
        uint64_t key = 0, s0 = 0, s1 = 0;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        uint4 g0 = a0, g1 = a0;  // (nodes that seed the next stage: the hint words of both probed slots)
        bool seeds = false;
        // node tl of a stage of Hs levels: levels lvl0 + 1 .. of the walkers standing at roots[] with hints[]
        auto generate = [&](uint32_t tl, uint32_t lvl0, uint32_t Hs, const Kmer *roots, const uint64_t *hints, bool more) {
            const uint32_t lvl = ((tl >> lg) * div_m) >> 16;  // tl / FN
            ni = lvl0 + lvl + 1;
            const uint32_t r = tl - lvl * FN;
            na = r >> lg;
            const uint32_t c = r & (uint32_t)(nb - 1);
            const uint64_t h = hints[na];
            const bool right = (h >> 62) & 1;
            const uint32_t hb = (uint32_t)(h >> (2 * lvl)) & 3u;  // expected step
            const uint32_t cstar = dir == 0 ? (2 * hb + (right ? 1u : 0u)) : hb;
            npred = spec && c == cstar;
            if (MODE == KEY_PACKED) {
                // k <= 31: everything in one 64-bit word, no branches (walker_at + neighbour + key_of, specialised)
                const uint64_t root = roots[na].lo, kmask = (1ull << (2 * k)) - 1;
                const uint32_t sh = 2 * lvl;  // 0 .. 54
                const uint64_t fwd = ((root << sh) | lh_block_forward(h, lvl)) & kmask;                       // root followed by lvl bases
                const uint64_t bwd = (root >> sh) | ((h & lh_mask(lvl)) << (2 * ((uint32_t)k - lvl)));        // root preceded by them
                const uint64_t v = lvl == 0 ? root : (right ? fwd : bwd);
                const bool left = dir < 0 || (dir == 0 && !(c & 1));
                const uint64_t cc = dir == 0 ? (c >> 1) : c;
                const uint64_t x = left ? ((v >> 2) | (cc << (2 * (k - 1)))) : (((v << 2) | cc) & kmask);
                nk.hi = 0;
                nk.lo = x;
                const uint64_t rcx = rc_packed(x, k);
                nflip = rcx < x;
                key = nflip ? rcx : x;
            } else {
                const Kmer v = walker_at(roots[na], k, h, lvl);  // the expected path so far
                nk = neighbour(v, k, dir, (int)c);
                key = (uint64_t)key_of<MODE>(nk, k, &nflip);
            }
            s0 = solid_slot_of(t, key);
            s1 = (s0 & ~(uint64_t)t.rmask) | ((s0 + 1) & t.rmask);
            a0 = *reinterpret_cast<const uint4 *>(t.slots + s0);
            a1 = *reinterpret_cast<const uint4 *>(t.slots + s1);
            seeds = more && npred && lvl + 1 == Hs;
            if (seeds) {
                g0 = reinterpret_cast<const uint4 *>(t.slots + s0)[1];
                g1 = reinterpret_cast<const uint4 *>(t.slots + s1)[1];
            }
            lookups++;
        };
        {
            const Kmer *roots = L.root;
            const uint64_t *hints = L.rhint;
            uint32_t Hs = H, h_min = hl_min, lvl_done = 0, nt_done = 0;
            for (uint32_t st = 0;; st++) {
                // another stage follows when this one runs to the end of the walkers' hints and the budget is not used up
                const bool more = spec && Hs == h_min && lvl_done + Hs < Hcap;
                const bool mine = tid >= nt_done && tid < nt_done + Hs * FN;
                if (mine) generate(tid - nt_done, lvl_done, Hs, roots, hints, more);
                if (st == 0) {
                    // the previous round's vertices enter the index while this round's probes are in flight (taken
                    // from the top of the workgroup, where threads usually hold no node)
                    if (BFS_THREADS - 1 - tid < pend) vis_insert(S, L.pub_k[BFS_THREADS - 1 - tid], L.pub_idx[BFS_THREADS - 1 - tid]);
                    MC_STAMP(1);
                }
                if (mine) {
                    have = true;
                    cov = solid_get2(t, key, s0, s1, a0, a1, &nslot);
                    L.cov[tid] = (int16_t)cov;
                    L.kmer[tid] = nk;
                    if (npred) L.pk[(ni - 1) * F + na] = nk;
                    if (seeds) {  // the walker's expected position at the end of this stage, and the hint stored there
                        uint64_t h2 = 0;
                        if (cov >= min_cov && nslot != ~0ull) {
                            const bool right = (hints[na] >> 62) & 1;
                            uint64_t hr, hl;
                            if (nslot == s0) { hr = ((uint64_t)g0.y << 32) | g0.x; hl = ((uint64_t)g0.w << 32) | g0.z; }
                            else if (nslot == s1) { hr = ((uint64_t)g1.y << 32) | g1.x; hl = ((uint64_t)g1.w << 32) | g1.z; }
                            else { const SolidSlot *sl = t.slots + nslot; hr = sl->hr; hl = sl->hl; }
                            h2 = walker_hint(hr, hl, nflip, right);
                        }
                        L.sroot[(st + 1) & 1][na] = nk;
                        L.shint[(st + 1) & 1][na] = h2;
                    }
                }
                lvl_done += Hs;
                nt_done += Hs * FN;
                if (!more) break;
                __syncthreads();
                roots = L.sroot[(st + 1) & 1];
                hints = L.shint[(st + 1) & 1];
                h_min = LHINT_MAX;
                for (uint32_t a = 0; a < F; a++) h_min = min(h_min, lh_len(hints[a]));
                if (h_min > (uint32_t)k) h_min = (uint32_t)k;
                Hs = min(h_min, Hcap - lvl_done);
                if (Hs == 0) break;
            }
            H = lvl_done;
        }
        L.set[tid] = LH_EMPTY;
        L.set[tid + BFS_THREADS] = LH_EMPTY;
        __syncthreads();
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
        __syncthreads();
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
        }
        __syncthreads();
        MC_STAMP(3);
        uint32_t J = 0;
        if (H > 1) J = min(H, L.bad_lvl - 1);

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
                    if (ni == J) {  // the walker's new position and its own hint
                        const bool right = (L.rhint[na] >> 62) & 1;
                        const SolidSlot *sl = t.slots + nslot;
                        L.nhint[na] = nslot == ~0ull ? (right ? (1ull << 62) : 0) : walker_hint(sl->hr, sl->hl, nflip, right);
                        L.fl_idx[cur ^ 1][na] = (uint32_t)idx;
                    }
                } else if (solid) {  // solid but already there: lastKmers.add(parent)
                    const uint32_t pidx = ni == 1 ? (L.fl_idx[cur][na] & 0x3FFFFFFFu)
                                                  : (uint32_t)(n + (unsigned long long)(ni - 2) * F + na);
                    atomicOr(&S.flags[pidx], 1u);
                }
            }
            __syncthreads();
            if (tid < F) {
                L.root[tid] = L.pk[(J - 1) * F + tid];
                L.rhint[tid] = L.nhint[tid];
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
            }
            MC_STAMP(4);
        } else {
            // ---- exact one-level replay of level 1 (ids [0, FN) are the plain neighbour sets of the walkers)
            slow_rounds++;
            if (tid < FN) {
                L.nslot[tid] = nslot;
                L.flip[tid] = nflip ? 1 : 0;
                if (H > 1) L.vis[tid] = (solid && vis_find(S, nk)) ? 1 : 0;  // index only, no expectations
            }
            __syncthreads();
            if (tid < 64) {
                uint32_t last_base = 0;
                const uint32_t n_new = replay_slow(S, L, 1, min_cov, max_kmers, max_radius, lg, flim, &last_base);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                for (uint32_t i = lane; i < n_new; i += 64) vis_insert(S, L.kmer[L.new_id[i]], L.new_idx[i]);
                const uint32_t Fn = L.F;
                const int c2 = L.cur;
                Kmer nr{0, 0};
                uint64_t nh = 0;
                if (lane < Fn && Fn <= flim) {
                    const uint32_t id = last_base + L.fl_w[c2][lane];
                    nr = L.kmer[id];
                    const bool right = dir > 0 || (dir == 0 && (id & 1u));  // odd neighbour index = right neighbour
                    const uint64_t sl = L.nslot[id];
                    nh = sl == ~0ull ? (right ? (1ull << 62) : 0)
                                     : walker_hint(t.slots[sl].hr, t.slots[sl].hl, L.flip[id] != 0, right);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                if (lane < Fn && Fn <= flim) {
                    L.root[lane] = nr;
                    L.rhint[lane] = nh;
                    L.fl_w[c2][lane] = lane;  // walkers are numbered 0..F-1 in the next round
                }
                if (lane == 0) {
                    L.any_dup_root = 0;
                    L.pend = 0;
                    L.rounds_left--;
                }
            }
            MC_STAMP(4);
        }
        __syncthreads();
        MC_STAMP(5);
    }
    __syncthreads();
    {   // vertices of the last fast round still waiting for the index
        const uint32_t pend = L.pend;
        if (tid < pend) vis_insert(S, L.pub_k[tid], L.pub_idx[tid]);
    }
    __syncthreads();
    if (tid == 0) {
        L.pend = 0;
        ctl->n = L.n;
        ctl->lb = L.lb;
        ctl->le = L.le;
        ctl->c0 = 0;
        ctl->level = L.level;
        ctl->rounds_narrow += rounds;
        ctl->rounds_slow += slow_rounds;
        if (L.status != BFS_RUNNING) ctl->status = L.status;
#ifdef MC_BFS_TIMING
        for (int i = 0; i < 8; i++) ctl->tacc[i] += tacc[i];
#endif
    }
    __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(BFS_THREADS) k_bfs(const BfsState *__restrict__ states, SolidView t, int k,
                                                     int min_cov, long long max_kmers, long long max_radius,
                                                     unsigned long long max_rounds)
{
    __shared__ BfsLds lds;
    const BfsState S = states[blockIdx.x];
    BfsCtl *ctl = S.ctl;
    if (ctl->status != BFS_RUNNING) return;  // finished (or waiting for the host) in an earlier launch
    const uint32_t tid = threadIdx.x;
    unsigned long long lookups = 0, rounds_left = max_rounds, chunks = 0;
    const int dir = S.dir;
    const int nb = dir == 0 ? 8 : 4;
    const uint32_t flim = NARROW_CAND / nb;

    // seeds: every window with reads.get(key) >= minOccurences, in order (:159-192)
    if (!ctl->seeds_done) {
        for (;;) {
            const unsigned long long c0 = ctl->c0;
            if (c0 >= S.n_seeds) break;
            if (ctl->n + BFS_THREADS > S.dcap) {
                if (tid == 0) ctl->status = BFS_NEED_GROW;
                goto out;
            }
            if (rounds_left == 0) goto out;
            rounds_left--;
            chunks++;
            const uint64_t r = c0 + tid;
            const bool have = r < S.n_seeds;
            Kmer cand{0, 0};
            if (have) { cand.hi = S.seed_hi ? S.seed_hi[r] : 0; cand.lo = S.seed_lo[r]; }
            __syncthreads();
            bfs_chunk_wide<MODE>(S, t, lds.w, k, min_cov, -1, true, have, cand, UINT64_MAX, 0, lookups);
            if (tid == 0) ctl->c0 = c0 + BFS_THREADS;
            __syncthreads();
        }
        if (tid == 0) {
            ctl->seeds_done = 1;
            ctl->lb = 0;
            ctl->le = ctl->n;
            ctl->c0 = 0;
            ctl->level = 0;
        }
        __syncthreads();
    }

    for (;;) {
        const unsigned long long lb = ctl->lb, le = ctl->le;
        if (le == lb) {
            if (tid == 0) ctl->status = BFS_DONE;
            break;
        }
        if (ctl->status != BFS_RUNNING) break;
        if (rounds_left == 0) break;
        if (ctl->c0 == 0 && le - lb <= flim) {
            __syncthreads();
            bfs_narrow<MODE>(S, t, lds.n, k, min_cov, max_kmers, max_radius, rounds_left, lookups);
            rounds_left = lds.n.rounds_left;
            __syncthreads();
            if (ctl->status != BFS_RUNNING) break;  // done, or distanceToKmer must grow
            if (ctl->le - ctl->lb <= flim && ctl->le != ctl->lb) break;  // round budget used up: relaunch
            continue;
        }
        const long long level = ctl->level;
        const bool radius_ok = max_radius < 0 || level + 1 <= max_radius;  // newDistance > threshold -> false
        const unsigned long long ncand = (le - lb) * (unsigned long long)nb;
        for (;;) {
            const unsigned long long c0 = ctl->c0;
            if (c0 >= ncand) break;
            if (ctl->n + BFS_THREADS > S.dcap) {
                if (tid == 0) ctl->status = BFS_NEED_GROW;
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
            __syncthreads();
            bfs_chunk_wide<MODE>(S, t, lds.w, k, min_cov, max_kmers, radius_ok, have, cand, parent,
                                 (int32_t)(level + 1), lookups);
            if (tid == 0) ctl->c0 = c0 + BFS_THREADS;
            __syncthreads();
        }
        if (tid == 0) {
            ctl->lb = le;
            ctl->le = ctl->n;
            ctl->c0 = 0;
            ctl->level = level + 1;
        }
        __syncthreads();
    }
out:
    atomicAdd(&ctl->lookups, lookups);
    if (tid == 0) atomicAdd(&ctl->chunks_wide, chunks);
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

}  // namespace mc

// libmcgpu.so -- MI355X (gfx950) k-mer counting + de Bruijn BFS behind the C ABI of include/mcgpu.h.
// Hand-written HIP for CDNA4: wave64, 16-byte table slots read with one dwordx4, memory-side
// 64-bit CAS / 32-bit add atomics, LDS-staged ordered compaction.  No MFMA: integer/hash work.
//
// Reference lines each piece replaces are cited at the definitions (src/... and itmo!/... as in
// include/mcgpu.h).  Nothing in this file links or calls oracle/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mcgpu.h"
#include "kmer_device.h"

using namespace mc;

// ------------------------------------------------------------------------------------------ ctx

struct mc_ctx {
    mc_config cfg{};
    std::mutex mu;
    std::string err;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // table
    Slot *slots = nullptr;
    uint32_t rb = 0, sb = 12;  // 2^rb regions of 2^sb slots
    unsigned long long *d_ctr = nullptr;  // [0] n_used, [1] empty_cnt, [2] scratch counter
    uint32_t *d_fatal = nullptr;
    uint64_t n_used_host = 0;
    bool finalized = false;

    mc_stats st{};

    uint64_t n_slots() const { return 1ull << (rb + sb); }
    TableView view() const
    {
        TableView t;
        t.slots = slots;
        t.shift = 64 - (rb + sb);
        t.rmask = (1u << sb) - 1;
        t.n_used = d_ctr;
        t.empty_cnt = d_ctr + 1;
        t.fatal = d_fatal;
        return t;
    }
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
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T)); }
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
                                                     uint64_t r_end, int k, TableView t)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = r_begin + wave; r < r_end; r += n_waves) {
        const uint64_t b = offsets[r], e = offsets[r + 1];
        if (e - b < (uint64_t)k) continue;
        const uint64_t nwin = e - b - (uint64_t)k + 1;
        for (uint64_t w = lane; w < nwin; w += 64) {
            const Kmer v = extract_kmer(words, b + w, k);
            table_add(t, (uint64_t)key_of<MODE>(v, k), 1u);
        }
    }
}

__global__ void k_add_keys(const int64_t *__restrict__ keys, uint64_t n, TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        table_add(t, (uint64_t)keys[i], 1u);
}

__global__ void k_add_pairs(const int64_t *__restrict__ keys, const int16_t *__restrict__ counts, uint64_t n,
                            TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (counts[i] > 0) table_add(t, (uint64_t)keys[i], (uint32_t)counts[i]);
}

// table rebuild into a larger table (the reference's enlargeAndRehash,
// itmo!/structures/map/Long2ShortHashMap.java:191-214, done for the whole table at once)
__global__ void k_rehash(const Slot *__restrict__ old_slots, uint64_t n_old, TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_old; i += stride) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(old_slots + i);
        const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
        if (key != EMPTY_KEY) table_add(t, key, raw.z);
    }
}

// K4: BigLong2ShortHashMap.get for a batch of keys
__global__ void k_get(const int64_t *__restrict__ keys, uint64_t n, int16_t *__restrict__ out, TableView t)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (int16_t)table_get(t, (uint64_t)keys[i]);
}

// K6: (key, count) pairs with count >= min_cov; with keys == nullptr only counts them
__global__ void k_export(const Slot *__restrict__ slots, uint64_t n_slots, int min_cov, int64_t *__restrict__ keys,
                         int16_t *__restrict__ counts, uint64_t cap, unsigned long long *cursor)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(slots + i);
        const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
        if (key == EMPTY_KEY) continue;
        const int c = raw.z > 32767u ? 32767 : (int)raw.z;
        if (c < min_cov) continue;
        const unsigned long long pos = atomicAdd(cursor, 1ull);  // hipcc merges this into one add per wave
        if (keys && pos < cap) {
            keys[pos] = (int64_t)key;
            counts[pos] = (int16_t)c;
        }
    }
}

// ------------------------------------------------------------------------------------------ kernels: multi-GPU split

__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t n_owners)
{
    // bits disjoint from the slot index (which uses the TOP bits of fmix64(key))
    return (uint32_t)((fmix64(key) & 0xFFFFFFFFull) % n_owners);
}

template <int MODE, bool SCATTER>
__global__ void __launch_bounds__(256) k_extract_keys(const uint64_t *__restrict__ words,
                                                      const uint64_t *__restrict__ offsets, uint64_t n_reads, int k,
                                                      uint32_t n_owners, unsigned long long *cursors,
                                                      int64_t *__restrict__ out, uint64_t cap)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t b = offsets[r], e = offsets[r + 1];
        if (e - b < (uint64_t)k) continue;
        const uint64_t nwin = e - b - (uint64_t)k + 1;
        for (uint64_t w0 = 0; w0 < nwin; w0 += 64) {
            const bool act = w0 + lane < nwin;
            uint64_t key = 0;
            uint32_t own = 0xFFFFFFFFu;
            if (act) {
                key = (uint64_t)key_of<MODE>(extract_kmer(words, b + w0 + lane, k), k);
                own = owner_of(key, n_owners);
            }
            for (uint32_t o = 0; o < n_owners; o++) {  // one atomic per (wave, owner)
                const unsigned long long m = __ballot(act && own == o);
                if (!m) continue;
                const uint32_t cnt = (uint32_t)__popcll(m);
                unsigned long long base = 0;
                const int leader = __ffsll((long long)m) - 1;
                if ((int)lane == leader) base = atomicAdd(&cursors[o], (unsigned long long)cnt);
                base = __shfl(base, leader);
                if (SCATTER && act && own == o) {
                    const uint64_t pos = base + (uint64_t)__popcll(m & ((1ull << lane) - 1));
                    if (pos < cap) out[pos] = (int64_t)key;
                }
            }
        }
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

// ------------------------------------------------------------------------------------------ kernels: BFS

constexpr uint32_t V_EMPTY = 0xFFFFFFFFu, V_TOMB = 0xFFFFFFFEu, V_TEMP = 0x80000000u;
constexpr int BFS_THREADS = 1024;

enum { BFS_RUNNING = 0, BFS_DONE = 1, BFS_NEED_GROW = 2 };

struct BfsCtl {
    unsigned long long n;       // |distanceToKmer|
    unsigned long long lb, le;  // current frontier = entries [lb, le)
    unsigned long long c0;      // next candidate rank inside the frontier
    unsigned long long lookups;
    long long level;            // distance of the frontier
    int status;
    int seeds_done;
};

struct BfsState {
    uint64_t *hi, *lo;  // distanceToKmer keys in insertion order
    int32_t *dist;
    int16_t *cov;
    uint32_t *flags;    // bit0: in lastKmers; bit1: seed window queued more than once
    uint64_t dcap;
    uint32_t *vis;      // open-addressed set of indices into the arrays above
    uint64_t vmask;
    BfsCtl *ctl;
};

__device__ __forceinline__ uint32_t vis_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t vis_hash(const Kmer &v)
{
    return fmix64(v.lo ^ fmix64(v.hi + 0x9e3779b97f4a7c15ull));
}

// block-wide exclusive scan of one flag per thread; returns the block total in *total
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

// One chunk of <= BFS_THREADS candidate vertices in rank order: look up, deduplicate against
// distanceToKmer and among themselves (smallest rank wins, exactly the sequential order of
// src/algo/OneSequenceCalculator.java:198-214), append the survivors in rank order, apply
// TerminationMode.allowsAddition (src/algo/TerminationMode.java:31-47).
// parent == UINT64_MAX marks a seed window (:159-192).
template <int MODE>
__device__ void bfs_chunk(const BfsState &S, const TableView &t, int k, int min_cov, long long max_kmers,
                          bool radius_ok, bool have, const Kmer &cand, uint64_t parent, int32_t new_dist,
                          Kmer *lds_kmer, uint32_t *lds_tot, uint64_t rank0, unsigned long long &lookups)
{
    BfsCtl *ctl = S.ctl;
    const uint32_t tid = threadIdx.x;
    int cov = -1;
    if (have) {
        cov = table_get(t, (uint64_t)key_of<MODE>(cand, k));
        lookups++;
    }
    const bool solid = have && cov >= min_cov;
    const bool is_seed = parent == UINT64_MAX;
    const unsigned long long n_before = ctl->n;  // uniform: written only between barriers below
    const bool capped = max_kmers >= 0 && (long long)n_before >= max_kmers;
    lds_kmer[tid] = cand;
    if (solid && !is_seed && S.flags[parent] & 2u) atomicOr(&S.flags[parent], 1u);  // re-queued seed window
    __syncthreads();

    // ---- claim phase
    uint64_t slot = UINT64_MAX;
    bool contender = false;
    if (solid && !is_seed && (capped || !radius_ok)) {
        atomicOr(&S.flags[parent], 1u);  // allowsAddition() == false -> lastKmers.add(kmer)
    } else if (solid) {
        const uint32_t my = V_TEMP | tid;
        uint64_t s = vis_hash(cand) & S.vmask;
        for (uint64_t probe = 0; probe <= S.vmask; probe++) {
            uint32_t v = vis_load(&S.vis[s]);
            if (v == V_EMPTY) {
                v = atomicCAS(&S.vis[s], V_EMPTY, my);
                if (v == V_EMPTY) { slot = s; contender = true; break; }
            }
            if (v == V_TOMB) { s = (s + 1) & S.vmask; continue; }
            if (v & V_TEMP) {  // claimed in this chunk by candidate (v & ~V_TEMP); value only ever shrinks
                const Kmer o = lds_kmer[v & ~V_TEMP];
                if (o.lo == cand.lo && o.hi == cand.hi) {
                    atomicMin(&S.vis[s], my);
                    slot = s;
                    contender = true;
                    break;
                }
            } else if (S.lo[v] == cand.lo && S.hi[v] == cand.hi) {  // already in distanceToKmer
                if (is_seed) atomicOr(&S.flags[v], 2u); else atomicOr(&S.flags[parent], 1u);
                break;
            }
            s = (s + 1) & S.vmask;
        }
    }
    __syncthreads();

    // ---- resolve phase
    bool winner = false;
    if (contender) {
        winner = vis_load(&S.vis[slot]) == (V_TEMP | tid);
        if (!winner && !is_seed) atomicOr(&S.flags[parent], 1u);  // an earlier rank inserted it first
    }
    uint32_t total;
    const uint32_t pos = block_scan_flag(winner, lds_tot, &total);
    bool accepted = false;
    uint64_t idx = 0;
    if (winner) {
        idx = n_before + pos;
        accepted = is_seed || max_kmers < 0 || (long long)idx < max_kmers;  // distanceToKmer.size() < threshold
        if (accepted) {
            S.hi[idx] = cand.hi;
            S.lo[idx] = cand.lo;
            S.dist[idx] = new_dist;
            S.cov[idx] = (int16_t)cov;
            S.flags[idx] = 0;
            __hip_atomic_store(&S.vis[slot], (uint32_t)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(&S.vis[slot], V_TOMB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(&S.flags[parent], 1u);
        }
    }
    // losers of a seed chunk: the winner's index is final now only after the stores above land
    __syncthreads();
    if (contender && !winner && is_seed) {
        const uint32_t v = vis_load(&S.vis[slot]);
        if (!(v & V_TEMP)) atomicOr(&S.flags[v], 2u);
    }
    uint32_t n_acc;
    (void)block_scan_flag(accepted, lds_tot, &n_acc);
    if (tid == 0) ctl->n = n_before + n_acc;
    __syncthreads();
    (void)rank0;
}

// K5: persistent single-workgroup BFS; all state lives in HBM so the launch is resumable
// (bounded number of levels per launch; distanceToKmer can be grown by the host between launches).
template <int MODE>
__global__ void __launch_bounds__(BFS_THREADS) k_bfs(BfsState S, TableView t, int k, int dir, int min_cov,
                                                     long long max_kmers, long long max_radius,
                                                     const uint64_t *__restrict__ seed_hi,
                                                     const uint64_t *__restrict__ seed_lo, uint64_t n_seeds,
                                                     unsigned long long max_chunks)
{
    __shared__ Kmer lds_kmer[BFS_THREADS];
    __shared__ uint32_t lds_tot[BFS_THREADS / 64];
    BfsCtl *ctl = S.ctl;
    const uint32_t tid = threadIdx.x;
    unsigned long long lookups = 0, chunks = 0;
    const int nb = dir == 0 ? 8 : 4;

    // seeds: every window with reads.get(key) >= minOccurences, in order (:159-192)
    if (!ctl->seeds_done) {
        for (;;) {
            const unsigned long long c0 = ctl->c0;
            if (c0 >= n_seeds) break;
            if (ctl->n + BFS_THREADS > S.dcap) {
                if (tid == 0) ctl->status = BFS_NEED_GROW;
                goto out;
            }
            if (chunks++ >= max_chunks) goto out;
            const uint64_t r = c0 + tid;
            const bool have = r < n_seeds;
            Kmer cand{0, 0};
            if (have) { cand.hi = seed_hi ? seed_hi[r] : 0; cand.lo = seed_lo[r]; }
            __syncthreads();
            bfs_chunk<MODE>(S, t, k, min_cov, -1, true, have, cand, UINT64_MAX, 0, lds_kmer, lds_tot, c0, lookups);
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
            if (chunks++ >= max_chunks) goto out;
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
            bfs_chunk<MODE>(S, t, k, min_cov, max_kmers, radius_ok, have, cand, parent, (int32_t)(level + 1),
                            lds_kmer, lds_tot, c0, lookups);
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
}

__global__ void k_vis_rebuild(BfsState S, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Kmer v{S.hi[i], S.lo[i]};
        uint64_t s = vis_hash(v) & S.vmask;
        for (uint64_t probe = 0; probe <= S.vmask; probe++) {
            if (atomicCAS(&S.vis[s], V_EMPTY, (uint32_t)i) == V_EMPTY) break;
            s = (s + 1) & S.vmask;
        }
    }
}

// ------------------------------------------------------------------------------------------ host: table management

static int grid_for(uint64_t work_items, int block, int max_blocks = 256 * 8)
{
    uint64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

static int table_alloc(mc_ctx *c, uint32_t log2_slots)
{
    if (log2_slots < c->sb) log2_slots = c->sb;
    if (log2_slots > 36) return fail(c, MC_EOVERFLOW, "k-mer table would need 2^%u slots", log2_slots);
    c->rb = log2_slots - c->sb;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->slots), c->n_slots() * sizeof(Slot)));
    hipLaunchKernelGGL(k_fill_empty, dim3(grid_for(c->n_slots(), 256)), dim3(256), 0, c->stream, c->slots,
                       c->n_slots());
    HIPCHK(c, hipGetLastError());
    c->st.table_slots = c->n_slots();
    c->st.table_bytes = c->n_slots() * sizeof(Slot);
    return MC_OK;
}

static int read_counters(mc_ctx *c, unsigned long long *n_used, uint32_t *fatal)
{
    unsigned long long h[3];
    HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(fatal, c->d_fatal, sizeof *fatal, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_used = h[0];
    return MC_OK;
}

static int table_grow(mc_ctx *c, uint32_t new_log2)
{
    Slot *old = c->slots;
    const uint64_t old_n = c->n_slots();
    c->slots = nullptr;
    int rc = table_alloc(c, new_log2);
    if (rc) { c->slots = old; return rc; }
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, sizeof(unsigned long long), c->stream));  // n_used is recounted
    hipLaunchKernelGGL(k_rehash, dim3(grid_for(old_n, 256)), dim3(256), 0, c->stream, old, old_n, c->view());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(old));
    c->st.grows++;
    return MC_OK;
}

// Make room for `incoming` more key occurrences: returns how many of them may be inserted by the
// next launch without the load factor passing 0.85 even if every one is a new key.
static int table_reserve(mc_ctx *c, uint64_t incoming, uint64_t *allowed)
{
    unsigned long long used;
    uint32_t fatal;
    int rc = read_counters(c, &used, &fatal);
    if (rc) return rc;
    if (fatal) return fail(c, MC_EOVERFLOW, "a k-mer table region filled up (hash skew); table of %llu slots",
                           (unsigned long long)c->n_slots());
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
        rc = table_grow(c, c->rb + c->sb + 1);
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
    switch (c->cfg.key_mode) {
    case MC_KEY_PACKED:
        hipLaunchKernelGGL(k_count_reads<KEY_PACKED>, dim3(grid), dim3(block), 0, c->stream, d_words, d_off, r0, r1,
                           c->cfg.k, t);
        break;
    case MC_KEY_POLY:
        hipLaunchKernelGGL(k_count_reads<KEY_POLY>, dim3(grid), dim3(block), 0, c->stream, d_words, d_off, r0, r1,
                           c->cfg.k, t);
        break;
    default:
        hipLaunchKernelGGL(k_count_reads<KEY_FNV1A>, dim3(grid), dim3(block), 0, c->stream, d_words, d_off, r0, r1,
                           c->cfg.k, t);
    }
}

// counting with read offsets known on the host
static int add_reads_impl(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, const uint64_t *h_off,
                          uint64_t n_reads)
{
    const uint64_t k = (uint64_t)c->cfg.k;
    uint64_t r = 0;
    while (r < n_reads) {
        uint64_t allowed;
        const uint64_t remaining_bases = h_off[n_reads] - h_off[r];
        int rc = table_reserve(c, remaining_bases, &allowed);
        if (rc) return rc;
        // largest r1 with windows(r..r1) <= allowed (windows <= bases)
        uint64_t r1 = r, win = 0;
        const uint64_t target = h_off[r] + allowed;
        r1 = (uint64_t)(std::upper_bound(h_off + r, h_off + n_reads + 1, target) - h_off) - 1;
        if (r1 <= r) r1 = r + 1;  // a single read longer than the allowance: fine, still < 0.85 + one read
        if (r1 > n_reads) r1 = n_reads;
        for (uint64_t i = r; i < r1; i++) {
            const uint64_t len = h_off[i + 1] - h_off[i];
            if (len >= k) win += len - k + 1;
        }
        double ms = 0;
        rc = timed(c, &ms, [&] { launch_count(c, d_words, d_off, r, r1); });
        if (rc) return rc;
        c->st.count_ms += ms;
        c->st.count_total_ms += ms;
        c->st.count_launches++;
        c->st.windows += win;
        r = r1;
    }
    c->finalized = false;
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
    CREATE_CHK(hipMalloc(reinterpret_cast<void **>(&c->d_ctr), 3 * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc(reinterpret_cast<void **>(&c->d_fatal), sizeof(uint32_t)));
    CREATE_CHK(hipMemsetAsync(c->d_ctr, 0, 3 * sizeof(unsigned long long), c->stream));
    CREATE_CHK(hipMemsetAsync(c->d_fatal, 0, sizeof(uint32_t), c->stream));
#undef CREATE_CHK
    uint32_t lg = 22;  // 4 M slots = 64 MB to start with
    if (cfg->capacity_hint) {
        const double want = (double)cfg->capacity_hint / 0.5;
        while (lg < 36 && (double)(1ull << lg) < want) lg++;
    }
    int rc = table_alloc(c, lg);
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
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->slots) (void)hipFree(c->slots);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->d_fatal) (void)hipFree(c->d_fatal);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int mc_clear(mc_ctx *c)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    hipLaunchKernelGGL(k_fill_empty, dim3(grid_for(c->n_slots(), 256)), dim3(256), 0, c->stream, c->slots,
                       c->n_slots());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, 3 * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_fatal, 0, sizeof(uint32_t), c->stream));
    c->n_used_host = 0;
    c->finalized = false;
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

int mc_add_reads_packed(mc_ctx *c, const uint64_t *words, const uint64_t *off, uint64_t n_reads)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if ((!words || !off) && n_reads) return fail(c, MC_EINVAL, "mc_add_reads_packed: null pointer");
    if (n_reads == 0) return MC_OK;
    for (uint64_t i = 0; i < n_reads; i++)
        if (off[i + 1] < off[i]) return fail(c, MC_EINVAL, "mc_add_reads_packed: read_offsets not monotone at %llu",
                                             (unsigned long long)i);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const uint64_t n_bases = off[n_reads];
    const uint64_t n_words = (n_bases + 31) / 32 + 1;
    const uint64_t w_begin = off[0] / 32;  // offsets need not start at 0
    DevBuf<uint64_t> dw, doff;
    HIPCHK(c, dw.alloc(n_words - w_begin));
    HIPCHK(c, doff.alloc(n_reads + 1));
    std::vector<uint64_t> rel(off, off + n_reads + 1);
    for (auto &x : rel) x -= w_begin * 32;
    HIPCHK(c, hipMemcpyAsync(dw.p, words + w_begin, (n_words - w_begin) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(doff.p, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    int rc = add_reads_impl(c, dw.p, doff.p, rel.data(), n_reads);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return rc;
}

int mc_add_reads_packed_dev(mc_ctx *c, const uint64_t *d_words, const uint64_t *d_off, uint64_t n_reads,
                            uint64_t n_bases)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if ((!d_words || !d_off) && n_reads) return fail(c, MC_EINVAL, "mc_add_reads_packed_dev: null pointer");
    if (n_reads == 0) return MC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    // the launch planner needs the offsets on the host (8 bytes per read, once per call)
    std::vector<uint64_t> h_off(n_reads + 1);
    HIPCHK(c, hipMemcpyAsync(h_off.data(), d_off, (n_reads + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_off[n_reads] != n_bases)
        return fail(c, MC_EINVAL, "mc_add_reads_packed_dev: read_offsets[n_reads]=%llu but n_bases=%llu",
                    (unsigned long long)h_off[n_reads], (unsigned long long)n_bases);
    for (uint64_t i = 0; i < n_reads; i++)
        if (h_off[i + 1] < h_off[i])
            return fail(c, MC_EINVAL, "mc_add_reads_packed_dev: read_offsets not monotone at %llu",
                        (unsigned long long)i);
    return add_reads_impl(c, d_words, d_off, h_off.data(), n_reads);
}

int mc_add_keys_dev(mc_ctx *c, const int64_t *d_keys, uint64_t n)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!d_keys && n) return fail(c, MC_EINVAL, "mc_add_keys_dev: null pointer");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    uint64_t i = 0;
    while (i < n) {
        uint64_t allowed;
        int rc = table_reserve(c, n - i, &allowed);
        if (rc) return rc;
        const uint64_t m = std::min<uint64_t>(allowed, n - i);
        double ms = 0;
        rc = timed(c, &ms, [&] {
            hipLaunchKernelGGL(k_add_keys, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, d_keys + i, m, c->view());
        });
        if (rc) return rc;
        c->st.count_total_ms += ms;
        c->st.windows += m;
        i += m;
    }
    c->finalized = false;
    return MC_OK;
}

int mc_add_pairs_dev(mc_ctx *c, const int64_t *d_keys, const int16_t *d_counts, uint64_t n)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if ((!d_keys || !d_counts) && n) return fail(c, MC_EINVAL, "mc_add_pairs_dev: null pointer");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    uint64_t i = 0;
    while (i < n) {
        uint64_t allowed;
        int rc = table_reserve(c, n - i, &allowed);
        if (rc) return rc;
        const uint64_t m = std::min<uint64_t>(allowed, n - i);
        hipLaunchKernelGGL(k_add_pairs, dim3(grid_for(m, 256)), dim3(256), 0, c->stream, d_keys + i, d_counts + i, m,
                           c->view());
        HIPCHK(c, hipGetLastError());
        i += m;
    }
    c->finalized = false;
    return MC_OK;
}

int mc_finalize_counts(mc_ctx *c, uint64_t *n_distinct)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    unsigned long long h[3];
    uint32_t fatal;
    HIPCHK(c, hipMemcpyAsync(h, c->d_ctr, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&fatal, c->d_fatal, sizeof fatal, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (fatal) return fail(c, MC_EOVERFLOW, "a k-mer table region filled up (hash skew)");
    c->n_used_host = h[0];
    c->finalized = true;
    if (n_distinct) *n_distinct = h[0] + (h[1] ? 1 : 0);
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
        HIPCHK(c, dk.alloc(n));
        HIPCHK(c, dout.alloc(n));
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
                        uint32_t n_owners, int64_t *d_keys, uint64_t cap, uint64_t *owner_offsets)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!owner_offsets || n_owners == 0 || n_owners > 1024)
        return fail(c, MC_EINVAL, "mc_extract_keys_dev: bad n_owners / owner_offsets");
    (void)n_bases;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    DevBuf<unsigned long long> cur;
    HIPCHK(c, cur.alloc(n_owners));
    std::vector<unsigned long long> h(n_owners, 0);
    const int grid = grid_for(n_reads * 64, 256);
    const int k = c->cfg.k;
#define LAUNCH_EXTRACT(SC)                                                                                            \
    switch (c->cfg.key_mode) {                                                                                        \
    case MC_KEY_PACKED:                                                                                               \
        hipLaunchKernelGGL((k_extract_keys<KEY_PACKED, SC>), dim3(grid), dim3(256), 0, c->stream, d_words, d_off,     \
                           n_reads, k, n_owners, cur.p, d_keys, cap);                                                 \
        break;                                                                                                        \
    case MC_KEY_POLY:                                                                                                 \
        hipLaunchKernelGGL((k_extract_keys<KEY_POLY, SC>), dim3(grid), dim3(256), 0, c->stream, d_words, d_off,       \
                           n_reads, k, n_owners, cur.p, d_keys, cap);                                                 \
        break;                                                                                                        \
    default:                                                                                                          \
        hipLaunchKernelGGL((k_extract_keys<KEY_FNV1A, SC>), dim3(grid), dim3(256), 0, c->stream, d_words, d_off,      \
                           n_reads, k, n_owners, cur.p, d_keys, cap);                                                 \
    }
    // pass 1: histogram per owner
    HIPCHK(c, hipMemsetAsync(cur.p, 0, n_owners * sizeof(unsigned long long), c->stream));
    if (n_reads) { LAUNCH_EXTRACT(false) }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h.data(), cur.p, n_owners * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    owner_offsets[0] = 0;
    for (uint32_t o = 0; o < n_owners; o++) owner_offsets[o + 1] = owner_offsets[o] + h[o];
    if (owner_offsets[n_owners] > cap)
        return fail(c, MC_EINVAL, "mc_extract_keys_dev: %llu keys but capacity %llu",
                    (unsigned long long)owner_offsets[n_owners], (unsigned long long)cap);
    if (!d_keys) return MC_OK;
    // pass 2: scatter with cursors starting at the bucket offsets
    std::vector<unsigned long long> start(owner_offsets, owner_offsets + n_owners);
    HIPCHK(c, hipMemcpyAsync(cur.p, start.data(), n_owners * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    if (n_reads) { LAUNCH_EXTRACT(true) }
#undef LAUNCH_EXTRACT
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MC_OK;
}

int mc_export_dev(mc_ctx *c, int min_cov, int64_t *d_keys, int16_t *d_counts, uint64_t cap, uint64_t *n_out)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->finalized) return fail(c, MC_ESTATE, "mc_export: call mc_finalize_counts first");
    if (!n_out) return fail(c, MC_EINVAL, "mc_export: n_out is null");
    if (d_keys && !d_counts) return fail(c, MC_EINVAL, "mc_export: counts is null");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    unsigned long long *cursor = c->d_ctr + 2;
    HIPCHK(c, hipMemsetAsync(cursor, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_export, dim3(grid_for(c->n_slots(), 256)), dim3(256), 0, c->stream, c->slots, c->n_slots(),
                       min_cov, d_keys, d_counts, cap, cursor);
    HIPCHK(c, hipGetLastError());
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
    if (!keys) return mc_export_dev(c, min_cov, nullptr, nullptr, 0, n_out);
    DevBuf<int64_t> dk;
    DevBuf<int16_t> dc;
    {
        std::lock_guard<std::mutex> g(c->mu);
        HIPCHK(c, hipSetDevice(c->cfg.device));
        HIPCHK(c, dk.alloc(cap));
        HIPCHK(c, dc.alloc(cap));
    }
    int rc = mc_export_dev(c, min_cov, dk.p, dc.p, cap, n_out);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipMemcpy(keys, dk.p, *n_out * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(counts, dc.p, *n_out * 2, hipMemcpyDeviceToHost));
    return MC_OK;
}

int mc_get_stats(mc_ctx *c, mc_stats *out)
{
    if (!c || !out) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    *out = c->st;
    return MC_OK;
}

int mc_reset_stats(mc_ctx *c)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    const uint64_t slots = c->st.table_slots, bytes = c->st.table_bytes;
    c->st = mc_stats{};
    c->st.table_slots = slots;
    c->st.table_bytes = bytes;
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

void mc_bfs_result_free(mc_bfs_result *r)
{
    if (!r) return;
    free(r->hi); free(r->lo); free(r->dist); free(r->cov); free(r->last);
    memset(r, 0, sizeof *r);
}

struct BfsBuffers {
    BfsState S{};
    ~BfsBuffers()
    {
        (void)hipFree(S.hi); (void)hipFree(S.lo); (void)hipFree(S.dist); (void)hipFree(S.cov);
        (void)hipFree(S.flags); (void)hipFree(S.vis); (void)hipFree(S.ctl);
    }
};

static int bfs_alloc(mc_ctx *c, BfsState &S, uint64_t dcap)
{
    S.dcap = dcap;
    uint64_t vcap = 1024;
    while (vcap < 4 * dcap) vcap <<= 1;
    S.vmask = vcap - 1;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.hi), dcap * 8));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.lo), dcap * 8));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.dist), dcap * 4));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.cov), dcap * 2));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.flags), dcap * 4));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.vis), vcap * 4));
    HIPCHK(c, hipMemsetAsync(S.vis, 0xFF, vcap * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(S.flags, 0, dcap * 4, c->stream));
    return MC_OK;
}

int mc_bfs(mc_ctx *c, const uint64_t *seed_hi, const uint64_t *seed_lo, uint64_t n_seeds, int dir, int min_cov,
           int64_t max_kmers, int64_t max_radius, mc_bfs_result *out)
{
    if (!c) return MC_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!out) return fail(c, MC_EINVAL, "mc_bfs: out is null");
    memset(out, 0, sizeof *out);
    if (!c->finalized) return fail(c, MC_ESTATE, "mc_bfs: call mc_finalize_counts first");
    if (dir < -1 || dir > 1) return fail(c, MC_EINVAL, "mc_bfs: dir must be -1, 0 or +1");
    if (max_kmers < 0 && max_radius < 0)
        return fail(c, MC_EINVAL, "At least one of --maxkmers and --maxradius parameters should be set");
    if (min_cov < 0)
        return fail(c, MC_EINVAL, "mc_bfs: negative coverage threshold (absent k-mers read as -1 and would pass)");
    if (n_seeds && !seed_lo) return fail(c, MC_EINVAL, "mc_bfs: seed_lo is null");
    if (c->cfg.k > 32 && n_seeds && !seed_hi) return fail(c, MC_EINVAL, "mc_bfs: seed_hi is null with k > 32");
    if (n_seeds == 0) return fail(c, MC_ENOSEED, "Could not find any k-mers of the target gene in the input");
    if (max_kmers >= (int64_t)V_TEMP - 4096 || n_seeds >= (uint64_t)V_TEMP - 4096)
        return fail(c, MC_EINVAL, "mc_bfs: more than 2^31 vertices requested");
    HIPCHK(c, hipSetDevice(c->cfg.device));

    DevBuf<uint64_t> d_shi, d_slo;
    HIPCHK(c, d_slo.alloc(n_seeds));
    HIPCHK(c, hipMemcpyAsync(d_slo.p, seed_lo, n_seeds * 8, hipMemcpyHostToDevice, c->stream));
    if (seed_hi) {
        HIPCHK(c, d_shi.alloc(n_seeds));
        HIPCHK(c, hipMemcpyAsync(d_shi.p, seed_hi, n_seeds * 8, hipMemcpyHostToDevice, c->stream));
    }

    BfsBuffers B;
    BfsState &S = B.S;
    uint64_t dcap = max_kmers >= 0 ? std::max<uint64_t>((uint64_t)max_kmers, n_seeds) + 2 * BFS_THREADS
                                   : std::max<uint64_t>(1ull << 20, n_seeds + 2 * BFS_THREADS);
    int rc = bfs_alloc(c, S, dcap);
    if (rc) return rc;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&S.ctl), sizeof(BfsCtl)));
    HIPCHK(c, hipMemsetAsync(S.ctl, 0, sizeof(BfsCtl), c->stream));

    BfsCtl h{};
    double total_ms = 0;
    const unsigned long long max_chunks = 1ull << 16;  // bounds one launch; the loop below relaunches
    for (;;) {
        const TableView t = c->view();
        rc = timed(c, &total_ms, [&] {
            switch (c->cfg.key_mode) {
            case MC_KEY_PACKED:
                hipLaunchKernelGGL(k_bfs<KEY_PACKED>, dim3(1), dim3(BFS_THREADS), 0, c->stream, S, t, c->cfg.k, dir,
                                   min_cov, (long long)max_kmers, (long long)max_radius, d_shi.p, d_slo.p, n_seeds,
                                   max_chunks);
                break;
            case MC_KEY_POLY:
                hipLaunchKernelGGL(k_bfs<KEY_POLY>, dim3(1), dim3(BFS_THREADS), 0, c->stream, S, t, c->cfg.k, dir,
                                   min_cov, (long long)max_kmers, (long long)max_radius, d_shi.p, d_slo.p, n_seeds,
                                   max_chunks);
                break;
            default:
                hipLaunchKernelGGL(k_bfs<KEY_FNV1A>, dim3(1), dim3(BFS_THREADS), 0, c->stream, S, t, c->cfg.k, dir,
                                   min_cov, (long long)max_kmers, (long long)max_radius, d_shi.p, d_slo.p, n_seeds,
                                   max_chunks);
            }
        });
        if (rc) return rc;
        HIPCHK(c, hipMemcpy(&h, S.ctl, sizeof h, hipMemcpyDeviceToHost));
        if (h.status == BFS_DONE) break;
        if (h.status == BFS_NEED_GROW) {
            // grow distanceToKmer and its index; only reachable without --maxkmers
            BfsState N{};
            rc = bfs_alloc(c, N, S.dcap * 2);
            if (rc) return rc;
            N.ctl = S.ctl;
            HIPCHK(c, hipMemcpyAsync(N.hi, S.hi, h.n * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.lo, S.lo, h.n * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.dist, S.dist, h.n * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.cov, S.cov, h.n * 2, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(N.flags, S.flags, h.n * 4, hipMemcpyDeviceToDevice, c->stream));
            if (h.n) {
                hipLaunchKernelGGL(k_vis_rebuild, dim3(grid_for(h.n, 256)), dim3(256), 0, c->stream, N, (uint64_t)h.n);
                HIPCHK(c, hipGetLastError());
            }
            const int zero = BFS_RUNNING;
            HIPCHK(c, hipMemcpyAsync(&S.ctl->status, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            (void)hipFree(S.hi); (void)hipFree(S.lo); (void)hipFree(S.dist); (void)hipFree(S.cov);
            (void)hipFree(S.flags); (void)hipFree(S.vis);
            S = N;
        }
    }
    if (h.n == 0) return fail(c, MC_ENOSEED, "Could not find any k-mers of the target gene in the input");

    const uint64_t n = h.n;
    out->n = n;
    out->hi = static_cast<uint64_t *>(malloc(n * 8));
    out->lo = static_cast<uint64_t *>(malloc(n * 8));
    out->dist = static_cast<int32_t *>(malloc(n * 4));
    out->cov = static_cast<int16_t *>(malloc(n * 2));
    out->last = static_cast<uint8_t *>(malloc(n));
    std::vector<uint32_t> flags(n);
    if (!out->hi || !out->lo || !out->dist || !out->cov || !out->last) {
        mc_bfs_result_free(out);
        return fail(c, MC_ENOMEM, "mc_bfs: out of host memory");
    }
    HIPCHK(c, hipMemcpy(out->hi, S.hi, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out->lo, S.lo, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out->dist, S.dist, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out->cov, S.cov, n * 2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(flags.data(), S.flags, n * 4, hipMemcpyDeviceToHost));
    uint64_t levels = 0;
    for (uint64_t i = 0; i < n; i++) {
        out->last[i] = (uint8_t)(flags[i] & 1u);
        if ((uint64_t)out->dist[i] > levels) levels = (uint64_t)out->dist[i];
    }
    out->levels = levels;
    out->lookups = h.lookups;
    out->device_ms = total_ms;
    return MC_OK;
}

}  // extern "C"

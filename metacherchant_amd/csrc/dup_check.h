// Equal keys in different regions: what a table of HASH keys in minimizer bins (count_long.h) must find before anybody reads it.
//
// The reference adds every window's hash to one map (src/io/LargeKIOUtils.java:46-49 -> itmo!/structures/map/Long2ShortHashMap.java:119-157):
// two different k-mers with the same 64-bit PolynomialHash (src/utils/PolynomialHash.java:19-28) share a counter.  The long-record
// table files a key under the minimizer bin of its k-mer's BASES, so two such k-mers can sit in two regions with a counter each
// (expected number of such pairs: n^2 / 2^65 -- 0.57 for configs[2]'s 4.58 G keys, 0.006 scaled to 10 M reads).  Nothing local can tell:
// it takes a join of all the table's keys by key.  This file is that join, as a partition by key bits:
//
//   level 1   every occupied slot's key goes to one of DUP_B1 = 256 buckets by the top bits of sk_home_mix(key) -- written by the
//             merge kernel itself while a region goes back to HBM (count_long.h k_p3_long: the keys are in LDS then, and a region's
//             slots are in home-slot order, i.e. nearly in bucket order: runs of ~6 keys to consecutive addresses), or by a sweep of
//             the table (k_dup_sweep: what a context falls back to).  A workgroup owns one segment of every bucket.
//   level 2   k_dup_scatter: a bucket's keys, tile by tile through LDS, into F2 <= 1024 sub-buckets by the next bits; a workgroup
//             (bucket, slice of its segments) owns its own segment of every sub-bucket: no global atomics.
//   level 3   k_dup_find: a sub-bucket (~1 750 keys) goes into an LDS set of 32-bit fingerprints with one compare-and-swap a key; a
//             key whose fingerprint is taken is looked at again (the sub-bucket is read once more, 7e-4 of them), and a key that
//             really is there twice is listed.
//
// 8 bytes a key and level: 3.7 GB written (6.1 measured: runs of ~6 keys) + 3.7 GB read and 5.1 written + 3.8 GB read for configs[2]
// scaled (458 M keys): +1.5 ms in the merge kernel, 2.8 + 1.8 ms for the two levels (profiles/r06_*).  The list is
// empty in 994 runs of 1000 at that size; when it is not, mcgpu.hip (dup_fixup) sweeps the table once for the listed keys' slots,
// writes the SUM of a key's counters into each of them -- so the walk, which comes by bases, and mc_get's sweep read the reference's
// count wherever they look -- and remembers the slots: exports and "Hashtable size" count a key once, and the slots get their own
// counts back before the table takes more reads or moves to hash-prefix regions.
#pragma once
#include "kmer_device.h"

namespace mc {

constexpr uint32_t DUP_B1_LG = 8, DUP_B1 = 1u << DUP_B1_LG;
constexpr uint32_t DUP_MAX_F2_LG = 10;
constexpr int DUP_THREADS = 1024, DUP_ITEMS = 8, DUP_TILE = DUP_THREADS * DUP_ITEMS;

__host__ __device__ __forceinline__ uint32_t dup_b1(uint64_t key) { return sk_home_mix(key) >> (32 - DUP_B1_LG); }
__host__ __device__ __forceinline__ uint32_t dup_f2(uint64_t key, uint32_t f2_lg) { return f2_lg ? (sk_home_mix(key) << DUP_B1_LG) >> (32 - f2_lg) : 0u; }
// bits for the LDS set of level 3: not the ones the buckets were cut by
__host__ __device__ __forceinline__ uint32_t dup_mix3(uint64_t key)
{
    uint32_t x = (uint32_t)(key >> 32) * 0x9E3779B1u ^ (uint32_t)key * 0xC2B2AE35u;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15;
    return x;
}

// level-1 stream: keys[((b * nseg) + seg) * cap + i], counts[b * nseg + seg]; the last segment of every bucket takes the keys
// that enter the table outside the merge kernel (the drain of handed-on occurrences, spilled records), by a global atomic
struct DupL1 {
    uint64_t *keys;    // nullptr: nobody collects
    uint32_t *counts;
    uint32_t nseg;
    uint64_t cap;
    uint32_t *lost;    // set when a segment overflows: the stream is then not used
};

__device__ __forceinline__ void dup_l1_extra(const DupL1 &d, uint64_t key)
{
    if (!d.keys) return;
    const uint32_t b = dup_b1(key), sg = d.nseg - 1;
    const uint32_t pos = atomicAdd(&d.counts[b * d.nseg + sg], 1u);
    if (pos < d.cap) d.keys[((uint64_t)b * d.nseg + sg) * d.cap + pos] = key; else atomicExch(d.lost, 1u);
}

// level 1 by a sweep of the table (a context whose merge kernel did not collect, or whose segments overflowed)
__global__ void __launch_bounds__(256) k_dup_sweep(const Slot *__restrict__ slots, uint64_t n_slots, DupL1 d)
{
    __shared__ uint32_t cur[DUP_B1];
    const uint32_t tid = threadIdx.x;
    cur[tid] = 0;
    __syncthreads();
    // consecutive rows of 256 slots to consecutive workgroups: a workgroup's share is spread over the whole table
    const uint64_t n_rows = (n_slots + 255) / 256;
    for (uint64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const uint64_t i = row * 256 + tid;
        if (i >= n_slots) continue;
        const uint4 raw = *reinterpret_cast<const uint4 *>(slots + i);
        const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
        if (key == EMPTY_KEY) continue;
        const uint32_t b = dup_b1(key);
        const uint32_t pos = atomicAdd(&cur[b], 1u);
        if (pos < d.cap) d.keys[((uint64_t)b * d.nseg + blockIdx.x) * d.cap + pos] = key; else atomicExch(d.lost, 1u);
    }
    __syncthreads();
    d.counts[tid * d.nseg + blockIdx.x] = (uint32_t)min((uint64_t)cur[tid], d.cap);
}

// level 2.  Workgroup (b, s): slice s of bucket b's segments -> F2 sub-buckets; it owns segment s of every sub-bucket (b, f), so its
// fill levels live in LDS and no stream is shared (a first version took each tile's place in a shared stream with a global atomic a
// sub-bucket and tile: 57 M of them for configs[2] scaled, 3.2 ms for 7.4 GB).  Keys of (b, f, s) at bucket(b) + ((f * slices + s) * cap + i).
struct DupL2 {
    uint64_t *out_a, *out_b;   // buckets [0, split) in out_a, the others in out_b (the pipeline's two idle streams serve as one buffer)
    uint32_t split;
    uint32_t *counts;          // [(b << f2_lg | f) * slices + s]
    uint32_t f2_lg, slices;
    uint64_t cap;              // keys a segment holds
    uint32_t *lost;
    __host__ __device__ __forceinline__ uint64_t *bucket(uint32_t b) const
    {
        const uint64_t per = ((uint64_t)slices << f2_lg) * cap;
        return b < split ? out_a + (uint64_t)b * per : out_b + (uint64_t)(b - split) * per;
    }
};
constexpr uint32_t DUP_MAX_SLICE_SEGS = 960;   // segments of a bucket one workgroup of level 2 takes at most (the host launches enough slices; two workgroups' LDS a CU)
constexpr uint32_t DUP_MAX_SLICES = 8;
struct DupScatterLds {
    uint64_t sorted[DUP_TILE];
    uint32_t cnt[1u << DUP_MAX_F2_LG], base[1u << DUP_MAX_F2_LG], gcur[1u << DUP_MAX_F2_LG];
    uint32_t pre[DUP_MAX_SLICE_SEGS + 1];
    uint32_t wsum[DUP_THREADS / 64];
};
__global__ void __launch_bounds__(DUP_THREADS, 8) k_dup_scatter(DupL1 in, DupL2 out)
{
    __shared__ DupScatterLds L;
    const uint32_t tid = threadIdx.x, slices = out.slices, b = blockIdx.x / slices, sl = blockIdx.x % slices;
    const uint32_t per = (in.nseg + slices - 1) / slices;  // (the host keeps this <= DUP_MAX_SLICE_SEGS)
    const uint32_t s0 = min(sl * per, in.nseg), s1 = min(s0 + per, in.nseg), ns = s1 - s0;
    const uint32_t F2 = 1u << out.f2_lg;
    // prefix of the slice's segment fill levels: one thread a segment, then a wave scans
    if (tid < ns) L.pre[tid + 1] = (uint32_t)min((uint64_t)in.counts[b * in.nseg + s0 + tid], in.cap);
    if (tid == 0) L.pre[0] = 0;
    for (uint32_t f = tid; f < F2; f += DUP_THREADS) { L.cnt[f] = 0; L.gcur[f] = 0; }
    __syncthreads();
    if (tid < 64) {
        const uint32_t chunk = (ns + 63) / 64;
        uint32_t sum = 0;
        for (uint32_t i = 0; i < chunk; i++) { const uint32_t j = tid * chunk + i; if (j < ns) sum += L.pre[j + 1]; }
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)tid >= o) incl += v; }
        uint32_t run = incl - sum;
        for (uint32_t i = 0; i < chunk; i++) { const uint32_t j = tid * chunk + i; if (j < ns) { run += L.pre[j + 1]; L.pre[j + 1] = run; } }
    }
    __syncthreads();
    const uint32_t total = L.pre[ns];
    uint64_t *const ob = out.bucket(b);
    uint32_t seg = 0;  // the segment my item of the current tile lies in: g only grows, so the search goes on from the last find
    for (uint32_t t0 = 0; t0 < total; t0 += DUP_TILE) {
        uint64_t key[DUP_ITEMS];
        uint32_t rk[DUP_ITEMS];
        const uint32_t n = min((uint32_t)DUP_TILE, total - t0);
#pragma unroll
        for (int j = 0; j < DUP_ITEMS; j++) {
            const uint32_t g = t0 + (uint32_t)j * DUP_THREADS + tid;
            key[j] = EMPTY_KEY;
            if (g < total) {
                while (L.pre[seg + 1] <= g) seg++;  // (pre[seg] <= g < pre[seg + 1]; empty segments are stepped over)
                key[j] = in.keys[((uint64_t)b * in.nseg + s0 + seg) * in.cap + (g - L.pre[seg])];
            }
        }
#pragma unroll
        for (int j = 0; j < DUP_ITEMS; j++) rk[j] = key[j] != EMPTY_KEY ? atomicAdd(&L.cnt[dup_f2(key[j], out.f2_lg)], 1u) : 0u;
        __syncthreads();
        // exclusive scan of the sub-bucket counts (F2 <= 1024 = one a thread)
        const uint32_t c = tid < F2 ? L.cnt[tid] : 0u;
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)(tid & 63u) >= o) incl += v; }
        if ((tid & 63u) == 63u) L.wsum[tid >> 6] = incl;
        __syncthreads();
        if (tid < F2) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < (tid >> 6); w++) off += L.wsum[w];
            L.base[tid] = off + incl - c;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DUP_ITEMS; j++)
            if (key[j] != EMPTY_KEY) L.sorted[L.base[dup_f2(key[j], out.f2_lg)] + rk[j]] = key[j];
        __syncthreads();
        for (uint32_t i = tid; i < n; i += DUP_THREADS) {
            const uint64_t kk = L.sorted[i];
            const uint32_t f = dup_f2(kk, out.f2_lg);
            const uint64_t dst = (uint64_t)L.gcur[f] + (i - L.base[f]);
            if (dst < out.cap) ob[((uint64_t)f * slices + sl) * out.cap + dst] = kk; else atomicExch(out.lost, 1u);
        }
        __syncthreads();
        if (tid < F2) { L.gcur[tid] += L.cnt[tid]; L.cnt[tid] = 0; }
        __syncthreads();
    }
    for (uint32_t f = tid; f < F2; f += DUP_THREADS)
        out.counts[(((uint64_t)b << out.f2_lg) | f) * slices + sl] = (uint32_t)min((uint64_t)L.gcur[f], out.cap);
}

// level 3: a workgroup a sub-bucket (its `slices` segments); SET_LG: log2 of the LDS set's slots.  A sub-bucket of more keys than the
// set takes at load 0.6 goes through it in passes, each over a range of dup_mix3.
struct DupOut {
    unsigned long long *keys;   // the listed keys (a key held by three slots is listed twice: the fix-up takes them as a set)
    unsigned long long *n;      // how many were listed (may pass cap: the list is then incomplete and the caller says so)
    uint64_t cap;
};
constexpr int DUP_FIND_ITEMS = 8;
constexpr uint32_t DUP_MAX_CAND = 32;
// a second 32-bit word of the key for the set's entries (never 0: that is a free slot)
__host__ __device__ __forceinline__ uint32_t dup_fp(uint64_t key)
{
    uint32_t x = (uint32_t)key * 0x85EBCA6Bu + (uint32_t)(key >> 32) * 0x27D4EB2Fu;
    x ^= x >> 13; x *= 0x165667B1u;
    x ^= x >> 16;
    return x | 1u;
}
// <13, 256>: 32 KB of LDS, five workgroups a CU, sub-buckets up to 4 900 keys in one pass; <15, 1024>: 128 KB, one workgroup a CU, up to
// 19 600 (a table of more than 1.2 G keys: configs[2] at full size has 17 500 a sub-bucket)
template <int SET_LG, int DUP_FIND_THREADS>
__global__ void __launch_bounds__(DUP_FIND_THREADS) k_dup_find(DupL2 in, DupOut out)
{
    // The set holds 32-bit fingerprints: what bounds this kernel is the rate of LDS compare-and-swaps, and a 64-bit one costs twice a
    // 32-bit one (2.3 ms for configs[2] scaled with whole keys in the set).  Two keys with one fingerprint on one probe path (7e-4 a
    // sub-bucket) are told apart by looking: the key that found its fingerprint taken is noted, and the sub-bucket is read once more.
    __shared__ uint32_t set[1u << SET_LG];
    __shared__ unsigned long long cand[DUP_MAX_CAND];
    __shared__ uint32_t cand_n, cand_hits[DUP_MAX_CAND];
    constexpr uint32_t MASK = (1u << SET_LG) - 1u, ROOM = (uint32_t)((1u << SET_LG) * 6 / 10), ROUND = DUP_FIND_THREADS * DUP_FIND_ITEMS;
    const uint32_t tid = threadIdx.x, n_sub = DUP_B1 << in.f2_lg, slices = in.slices;
    // A sub-bucket is 14 KB: the next one's keys and the one after's fill levels travel while this one is worked on.
    auto src_of = [&](uint32_t sb) { return in.bucket(sb >> in.f2_lg) + (uint64_t)(sb & ((1u << in.f2_lg) - 1u)) * slices * in.cap; };
    auto load_counts = [&](uint32_t sb, uint32_t (&pre)[9]) {  // (<= DUP_MAX_SLICES = 8 scalar loads)
        pre[0] = 0;
#pragma unroll
        for (uint32_t s = 0; s < 8; s++) pre[s + 1] = pre[s] + (s < slices ? (uint32_t)min((uint64_t)in.counts[(uint64_t)sb * slices + s], in.cap) : 0u);
    };
    auto load_keys = [&](uint32_t sb, const uint32_t (&pre)[9], uint32_t i0, uint64_t (&key)[DUP_FIND_ITEMS]) {
        const uint64_t *src = src_of(sb);
#pragma unroll
        for (int j = 0; j < DUP_FIND_ITEMS; j++) {
            const uint32_t i = i0 + (uint32_t)j * DUP_FIND_THREADS + tid;
            key[j] = EMPTY_KEY;
            if (i < pre[8]) {  // (segment and offset without indexing the register array)
                uint32_t s = 0, first = 0;
#pragma unroll
                for (uint32_t q = 1; q < 8; q++) { const bool past = i >= pre[q]; s += past ? 1u : 0u; first = past ? pre[q] : first; }
                key[j] = src[(uint64_t)s * in.cap + (i - first)];
            }
        }
    };
    auto in_pass = [&](uint64_t kk, uint32_t passes, uint32_t p) {
        return kk != EMPTY_KEY && !(passes > 1 && (uint32_t)(((uint64_t)(dup_mix3(kk) >> 8) * passes) >> 24) != p);
    };
    auto insert = [&](const uint64_t (&key)[DUP_FIND_ITEMS], uint32_t passes, uint32_t p) {
#pragma unroll
        for (int j = 0; j < DUP_FIND_ITEMS; j++) {
            const uint64_t kk = key[j];
            if (!in_pass(kk, passes, p)) continue;
            const uint32_t f = dup_fp(kk);
            for (uint32_t s = dup_mix3(kk) & MASK;; s = (s + 1) & MASK) {
                const uint32_t old = atomicCAS(&set[s], 0u, f);
                if (old == 0u) break;
                if (old == f) {  // the same key, or -- rarely -- another one with this fingerprint: looked at below
                    const uint32_t at = atomicAdd(&cand_n, 1u);
                    if (at < DUP_MAX_CAND) cand[at] = kk;
                    else {  // (more than the note takes: listed as it is -- the fix-up's sweep drops a key it finds in one slot only)
                        const unsigned long long g = atomicAdd(out.n, 1ull);
                        if (g < out.cap) out.keys[g] = kk;
                    }
                    break;
                }
            }
        }
    };
    uint32_t sbA = blockIdx.x, sbB = sbA + gridDim.x, sbC = sbB + gridDim.x;
    uint32_t preA[9], preB[9], preC[9];
    uint64_t keyA[DUP_FIND_ITEMS], keyB[DUP_FIND_ITEMS];
    if (sbA < n_sub) { load_counts(sbA, preA); load_keys(sbA, preA, 0, keyA); }
    if (sbB < n_sub) load_counts(sbB, preB);
    while (sbA < n_sub) {
        if (sbB < n_sub) load_keys(sbB, preB, 0, keyB);
        if (sbC < n_sub) load_counts(sbC, preC);
        const uint32_t n = preA[8];
        const uint32_t passes = (n + ROOM - 1) / ROOM;
        for (uint32_t p = 0; p < passes; p++) {  // (one pass, and one round of it, unless the sub-bucket is far larger than planned)
            for (uint32_t i = tid; i <= MASK; i += DUP_FIND_THREADS) set[i] = 0u;
            if (tid < DUP_MAX_CAND) cand_hits[tid] = 0;
            if (tid == 0) cand_n = 0;
            __syncthreads();
            insert(keyA, passes, p);
            for (uint32_t i0 = ROUND; i0 < n; i0 += ROUND) {
                uint64_t more[DUP_FIND_ITEMS];
                load_keys(sbA, preA, i0, more);
                insert(more, passes, p);
            }
            __syncthreads();
            const uint32_t nc = min(cand_n, DUP_MAX_CAND);
            if (nc) {  // (uniform; 7e-4 of the sub-bucket without a key in two slots) how often does each noted key occur?
                for (uint32_t i0 = 0; i0 < n; i0 += ROUND) {
                    uint64_t again[DUP_FIND_ITEMS];
                    load_keys(sbA, preA, i0, again);
#pragma unroll
                    for (int j = 0; j < DUP_FIND_ITEMS; j++)
                        for (uint32_t q = 0; q < nc; q++)
                            if (again[j] == cand[q]) atomicAdd(&cand_hits[q], 1u);
                }
                __syncthreads();
                if (tid < nc && cand_hits[tid] >= 2) {
                    bool first = true;  // (a key in three slots was noted twice: listed once)
                    for (uint32_t q = 0; q < tid; q++) first = first && cand[q] != cand[tid];
                    if (first) {
                        const unsigned long long g = atomicAdd(out.n, 1ull);
                        if (g < out.cap) out.keys[g] = cand[tid];
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int j = 0; j < DUP_FIND_ITEMS; j++) keyA[j] = keyB[j];
#pragma unroll
        for (int q = 0; q < 9; q++) { preA[q] = preB[q]; preB[q] = preC[q]; }
        sbA = sbB; sbB = sbC; sbC += gridDim.x;
    }
}

// ---- the fix-up: the listed keys as a small set in global memory (qk: ~0 = free), per entry the sum of the key's counters, the
// lowest slot that holds it (the one exports count) and the number of slots
struct DupSet {
    unsigned long long *qk;
    unsigned long long *tot;
    unsigned long long *prim;
    uint64_t mask;               // slots - 1; 0 with qk == nullptr: no set
    unsigned long long *n_keys;  // distinct keys in the set
};
__device__ __forceinline__ bool dup_set_find(const DupSet &q, uint64_t key, uint64_t *at)
{
    for (uint64_t s = fmix64(key) & q.mask;; s = (s + 1) & q.mask) {
        const unsigned long long v = q.qk[s];
        if (v == key) { *at = s; return true; }
        if (v == ~0ull) return false;
    }
}
__global__ void k_dupq_build(const unsigned long long *__restrict__ keys, uint64_t n, DupSet q)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t key = keys[i];
        for (uint64_t s = fmix64(key) & q.mask;; s = (s + 1) & q.mask) {
            const unsigned long long old = atomicCAS(&q.qk[s], ~0ull, (unsigned long long)key);
            if (old == ~0ull) { atomicAdd(q.n_keys, 1ull); break; }
            if (old == key) break;
        }
    }
}
// every slot of a listed key: (slot index, its own count, the set entry) noted, count added to the entry's sum
struct DupTwin { unsigned long long slot; uint32_t own, entry; };
__global__ void k_dupq_sweep(const Slot *__restrict__ slots, uint64_t n_slots, DupSet q, DupTwin *tw, unsigned long long *n_tw, uint64_t tw_cap)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(slots + i);
        const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
        if (key == EMPTY_KEY) continue;
        uint64_t s;
        if (!dup_set_find(q, key, &s)) continue;
        atomicAdd(&q.tot[s], (unsigned long long)raw.z);
        atomicMin(&q.prim[s], (unsigned long long)i);
        const unsigned long long at = atomicAdd(n_tw, 1ull);
        if (at < tw_cap) tw[at] = DupTwin{i, raw.z, (uint32_t)s};
    }
}
// merged != 0: every noted slot takes its key's sum (counters stop at 2^30 as everywhere); 0: its own count back
__global__ void k_dup_apply(Slot *slots, const DupTwin *__restrict__ tw, uint64_t n, DupSet q, int merged)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DupTwin t = tw[i];
    const unsigned long long sum = q.tot[t.entry];
    slots[t.slot].count = merged ? (uint32_t)(sum > (1ull << 30) ? (1ull << 30) : sum) : t.own;
}

// what a sweep that counts or lists KEYS (exports, "Hashtable size") asks about a slot: is it a listed key's second (third, ...) slot?
__device__ __forceinline__ bool dup_is_shadow(const DupSet &q, uint64_t key, uint64_t slot)
{
    if (!q.qk) return false;
    uint64_t s;
    return dup_set_find(q, key, &s) && q.prim[s] != slot;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Look-ups of k-mers that were never counted.  The walk of a table of hash keys in minimizer bins looks a k-mer up in the bin of
// its BASES (kmer_device.h solid_locate_kmer).  The reference looks the HASH up (src/algo/OneSequenceCalculator.java:89-96,198-204):
// a string nobody counted whose hash equals a counted k-mer's is solid there -- n / 2^64 a look-up for a random string, one in four
// for the neighbours of a colliding pair (the rolling hash keeps colliding while the same bases leave and enter).  After a walk,
// every look-up it made that came back "absent" -- the neighbours of its vertices, its seeds -- is asked again BY KEY, against the
// table's keys as the join left them (DupL2: a few thousand keys a sub-bucket, only the sub-bucket of a query is read).  A hit gets a
// slot in the bin of the string that asked (count 0), the join runs again -- the key now sits in two slots, both take the sum -- and the
// walk is repeated.  In all but ~1e-5 of the runs there is no hit and this costs one small kernel and a scan of the queried sub-buckets.
struct PhantomQ {
    unsigned long long *key, *hi, *lo;   // the hash that was asked for and the string that asked
    unsigned long long *n;               // how many (may pass cap: the caller then asks again with more room)
    uint64_t cap;
};
// items: n k-mers (hi may be null: k <= 32); nb = 0: the k-mers themselves (seeds), else their nb neighbours in direction dir
template <int MODE>
__global__ void __launch_bounds__(256) k_phantom_queries(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int nb, int dir, int k,
                                                        SolidView t, PhantomQ q)
{
    const uint64_t per = nb ? (uint64_t)nb : 1, total = n * per, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += stride) {
        const uint64_t i = it / per;
        Kmer v{hi ? hi[i] : 0ull, lo[i]};
        const Kmer sq = nb ? neighbour(v, k, dir, (int)(it % per)) : v;
        const uint64_t key = (uint64_t)key_of<MODE>(sq, k);
        if (key == EMPTY_KEY) continue;  // (counted apart, by key)
        if (solid_get_kmer<MODE>(t, sq, k, key) >= 0) continue;
        const unsigned long long at = atomicAdd(q.n, 1ull);
        if (at < q.cap) { q.key[at] = key; q.hi[at] = sq.hi; q.lo[at] = sq.lo; }
    }
}
__host__ __device__ __forceinline__ uint32_t dup_sub(uint64_t key, uint32_t f2_lg) { return (dup_b1(key) << f2_lg) | dup_f2(key, f2_lg); }
// the queries of a sub-bucket as a list: head[sub-bucket] -> query -> next[query] ... (0xFFFFFFFF ends it; a walk of 1 900 vertices
// asks 15 000 times over 262 144 sub-buckets: sorting them by sub-bucket costs more than it saves)
// (n_ptr: the number of queries is read where it was counted -- the host enqueues the whole check without waiting for it)
__global__ void k_pq_link(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ n_ptr, uint64_t cap, uint32_t f2_lg, uint32_t *head, uint32_t *next)
{
    const uint64_t n = min((uint64_t)*n_ptr, cap);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) next[i] = atomicExch(&head[dup_sub(keys[i], f2_lg)], (uint32_t)i);
}
// a workgroup a sub-bucket that has queries: the queries in LDS (64 at a time), every key of the sub-bucket against them
__global__ void __launch_bounds__(256) k_pq_match(DupL2 in, const uint32_t *__restrict__ head, const uint32_t *__restrict__ next,
                                                  const unsigned long long *__restrict__ qkeys, uint32_t *hits, unsigned long long *n_hits, uint64_t hit_cap)
{
    __shared__ unsigned long long qk[64];
    __shared__ uint32_t qi[64];
    __shared__ uint32_t m_sh, cur_sh;
    const uint32_t tid = threadIdx.x, n_sub = DUP_B1 << in.f2_lg;
    for (uint32_t sb = blockIdx.x; sb < n_sub; sb += gridDim.x) {
        if (head[sb] == 0xFFFFFFFFu) continue;  // (uniform)
        const uint32_t b = sb >> in.f2_lg, f = sb & ((1u << in.f2_lg) - 1u);
        const uint64_t *src = in.bucket(b) + (uint64_t)f * in.slices * in.cap;
        __syncthreads();
        if (tid == 0) cur_sh = head[sb];
        for (;;) {
            __syncthreads();
            if (tid == 0) {  // the next (up to) 64 queries of the list
                uint32_t m = 0, cur = cur_sh;
                while (cur != 0xFFFFFFFFu && m < 64) { qi[m] = cur; qk[m] = qkeys[cur]; m++; cur = next[cur]; }
                m_sh = m; cur_sh = cur;
            }
            __syncthreads();
            const uint32_t m = m_sh;
            if (!m) break;
            for (uint32_t sg = 0; sg < in.slices; sg++) {
                const uint32_t n = (uint32_t)min((uint64_t)in.counts[(uint64_t)sb * in.slices + sg], in.cap);
                for (uint32_t i = tid; i < n; i += 256) {
                    const unsigned long long kk = src[(uint64_t)sg * in.cap + i];
                    for (uint32_t j = 0; j < m; j++)
                        if (qk[j] == kk) {
                            const unsigned long long at = atomicAdd(n_hits, 1ull);
                            if (at < hit_cap) hits[at] = qi[j];
                        }
                }
            }
        }
    }
}
// ... the same check where the join's streams are gone (mc_trim, a pipeline run since): the queries as a set in global memory, one
// sweep of the table, every slot whose key is in the set names the query that brought it (the first of several with one key).
__global__ void k_pq_set_build(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ n_ptr, uint64_t cap, unsigned long long *qk, uint32_t *qidx,
                               uint64_t mask)
{
    const uint64_t n = min((uint64_t)*n_ptr, cap);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t key = keys[i];
        for (uint64_t s = fmix64(key) & mask;; s = (s + 1) & mask) {
            const unsigned long long old = atomicCAS(&qk[s], ~0ull, (unsigned long long)key);
            if (old == ~0ull) { qidx[s] = (uint32_t)i; break; }
            if (old == key) break;
        }
    }
}
__global__ void k_pq_sweep(const Slot *__restrict__ slots, uint64_t n_slots, const unsigned long long *__restrict__ qk, const uint32_t *__restrict__ qidx, uint64_t mask,
                           uint32_t *hits, unsigned long long *n_hits, uint64_t hit_cap)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(slots + i);
        const uint64_t key = ((uint64_t)raw.y << 32) | raw.x;
        if (key == EMPTY_KEY) continue;
        for (uint64_t s = fmix64(key) & mask;; s = (s + 1) & mask) {
            const unsigned long long v = qk[s];
            if (v == ~0ull) break;
            if (v == key) {
                const unsigned long long at = atomicAdd(n_hits, 1ull);
                if (at < hit_cap) hits[at] = qidx[s];
                break;
            }
        }
    }
}
// a slot (count 0) for every hit's key in the bin of the string that asked for it
__global__ void k_alias_insert(const uint32_t *__restrict__ hits, uint64_t n, PhantomQ q, TableView t, int k)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_new = 0;
    if (i < n) {
        const uint32_t h = hits[i];
        const Kmer sq{q.hi[h], q.lo[h]};
        const uint64_t key = q.key[h];
        const uint64_t region = ((uint64_t)(sk_bin(sk_hmin_of_kmer2(sq, k, t.mm_k < 0)) & 0xFFFFFF00u) * t.n_regions) >> 32;  // (count_long.h skl_bin)
        n_new = table_add_at(t, (region << MC_REGION_LG) | sk_home(key), key, 0u, 0u);
    }
    wave_add_ull(t.n_used, n_new);
}

}  // namespace mc

// Super-k-mer form of the counting pipeline for k = 32 .. 63 with polynomial-hash keys (src/utils/PolynomialHash.java:19-28,
// src/io/LargeKIOUtils.java:41-88): the same idea as count_pipeline.h's 16-byte records, with room for the longer runs.
//
// While a context only ever counted reads through this pipeline, the regions of its table are the minimizer bins of
// the k-mers' BASES (the key is a hash of the bases and says nothing about them), so nothing can look a key up by its
// value alone: the host keeps such a table to the counting phase and the sweeps (export, save, the solid-table build
// of the BFS) and rebuilds it by key hash (mcgpu.hip to_hash_regions) before the first operation that needs to.
//
// Record, 32 bytes (two uint4 a, b):
//   a.x        bin word (sk_bin of the run's minimizer): bucket digits of P2, region of P3
//   a.y        read pointer (kmer_device.h ptr_encode) of the run's first window, or 0
//   a.w:a.z    bases 0 .. 31 of the run (first base on top)
//   b.y:b.x    bases 32 .. 63
//   b.w:b.z    bases 64 .. 77 in bits 63 .. 36; windows - 1 in bits 3 .. 0        (windows + k - 1 <= 16 + 62 bases)
// P3 rolls both strands' hashes along the run: one 64-bit multiplication per window (the reverse strand's hash loses
// its lowest term and is divided by 5, i.e. multiplied by 5^-1 mod 2^64), the rest shifts and adds.
#pragma once
#include "count_pipeline.h"

namespace mc {

struct SkLRec {
    uint4 a, b;
};
static_assert(sizeof(SkLRec) == 32, "two 16-byte stores");

struct SkLSpill {
    SkLRec *recs;
    unsigned long long *count;
    uint64_t cap;
    uint32_t *lost;
};

constexpr uint64_t POLY_INV5 = 0xCCCCCCCCCCCCCCCDull;  // 5 * POLY_INV5 == 1 (mod 2^64)
__host__ __device__ inline uint64_t pow5(int e)
{
    uint64_t p = 1;
    for (int i = 0; i < e; i++) p *= 5;
    return p;
}

__device__ __forceinline__ uint32_t skl_windows(const SkLRec &r) { return (r.b.z & 15u) + 1u; }

// f(key, j) for every window j of the record, in order; the polynomial hash of src/utils/PolynomialHash.java over both strands,
// key = Math.min on signed longs (kmer_device.h key_poly)
template <class F>
__device__ __forceinline__ void skl_expand_poly(const SkLRec &r, int k, uint64_t p5k, uint64_t p5km1, F &&f)
{
    const uint64_t q1 = ((uint64_t)r.a.w << 32) | r.a.z, q2 = ((uint64_t)r.b.y << 32) | r.b.x, q3 = ((uint64_t)r.b.w << 32) | r.b.z;
    auto base = [&](uint32_t i) -> uint32_t {
        const uint64_t q = i < 32 ? q1 : (i < 64 ? q2 : q3);
        return (uint32_t)(q >> (62 - 2 * (i & 31))) & 3u;
    };
    uint64_t fw = 1, rc = 1;
    for (int i = 0; i < k; i++) {
        fw = fw * 5 + base((uint32_t)i);
        rc = rc * 5 + (3u ^ base((uint32_t)(k - 1 - i)));
    }
    const uint32_t nw = skl_windows(r);
    for (uint32_t j = 0;; j++) {
        const int64_t a = (int64_t)fw, b = (int64_t)rc;
        f((uint64_t)(a < b ? a : b), j);
        if (j + 1 >= nw) break;
        const uint32_t out = base(j), in = base(j + (uint32_t)k);
        // fw = 5^k + sum_i b_i 5^(k-1-i): the first base (weight 5^(k-1)) leaves, the rest moves up, the new base enters at weight 1
        fw = (fw - p5k - (uint64_t)out * p5km1) * 5 + p5k + in;
        // rc = 5^k + sum_i (3^b_i) 5^i: the first base's term (weight 1) leaves, the rest moves down, the new one enters on top
        rc = (rc - p5k - (3u ^ out)) * POLY_INV5 + p5k + (uint64_t)(3u ^ in) * p5km1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// P1: reads -> records, scattered into np1 level-1 buckets.  A workgroup owns segment blockIdx.x of every bucket and
// takes tiles of SKL_TILE base positions; one thread per position.
constexpr int SKL_THREADS = 1024;
constexpr uint32_t SKL_TILE = 1024;
constexpr uint32_t SKL_HALO = 128;  // bases behind a tile that its windows and records reach into (k + 15 - 1 <= 77; whole words)
constexpr int SKL_SEGMENTS = 512;   // workgroups of the launch = segments of every level-1 bucket

struct SklLds {
    uint64_t W[(SKL_TILE + SKL_HALO) / 32 + 2];      // the tile's bases and the halo behind it
    uint32_t mh[SKL_TILE + 64];                      // sk_order of the canonical SK_M-mer that starts at lo + i
    uint32_t hmin[SKL_TILE + 1];                     // minimizer hash of the window that starts at lo + i
    uint64_t starts[(SKL_TILE + SKL_HALO) / 64 + 2]; // bit b: a read starts (or the batch ends) at lo + b
    uint64_t valid[SKL_TILE / 64 + 2];               // bit i: lo + i starts a window that lies inside one read
    uint64_t brk[SKL_TILE / 64 + 2];                 // bit i: a run of windows cannot continue INTO position i (natural start, or no window)
    uint32_t cnt[PT_MAX_BUCKETS], wcur[PT_MAX_BUCKETS];
};

__device__ __forceinline__ uint64_t bits64_at(const uint64_t *w, uint32_t bit)
{   // 64 bits of a little-endian bitmap starting at `bit`
    const uint32_t i = bit >> 6, s = bit & 63;
    return s ? (w[i] >> s) | (w[i + 1] << (64 - s)) : w[i];
}
__device__ __forceinline__ uint64_t bases64_at(const uint64_t *w, uint32_t base)
{   // 32 bases (top-aligned) starting at base `base` of a packed array
    const uint32_t i = base >> 5, s = 2 * (base & 31);
    return s ? (w[i] << s) | (w[i + 1] >> (64 - s)) : w[i];
}

__global__ void __launch_bounds__(SKL_THREADS) k_skl_extract(const uint64_t *__restrict__ words, const uint64_t *__restrict__ offsets, uint64_t n_reads,
                                                            uint64_t base_lo, uint64_t n_bases, uint64_t n_tiles,
                                                            const uint32_t *__restrict__ first_read, int k, uint32_t np1, uint32_t *seg_counts,
                                                            uint64_t cap, SkLRec *out, SkLSpill sp, uint64_t ptr_base)
{
    __shared__ SklLds L;
    const uint32_t tid = threadIdx.x;
    if (tid < PT_MAX_BUCKETS) { L.cnt[tid] = 0; L.wcur[tid] = 0; }
    const int w = k - SK_M + 1;                      // SK_M-mers per window (18 .. 49)
    const uint64_t last_word = (n_bases + 31) / 32;  // the pad word
    constexpr uint32_t NW = (SKL_TILE + SKL_HALO) / 32 + 2, NS = (SKL_TILE + SKL_HALO) / 64 + 2;
    for (uint64_t tile = base_lo / SKL_TILE + blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t lo = tile * (uint64_t)SKL_TILE, hi_pos = lo + SKL_TILE + SKL_HALO;
        __syncthreads();
        if (tid < NW) L.W[tid] = words[min(lo / 32 + tid, last_word)];
        if (tid >= 64 && tid < 64 + NS) L.starts[tid - 64] = 0;
        __syncthreads();
        // read starts in (lo, hi_pos): offsets ascend; offsets[n_reads] = the end of the batch counts as one
        for (uint64_t r = (uint64_t)first_read[tile] + tid;; r += SKL_THREADS) {
            const uint64_t off = r <= n_reads ? offsets[r] : ~0ull;
            if (off > lo && off < hi_pos) atomicOr(reinterpret_cast<unsigned long long *>(&L.starts[(off - lo) >> 6]), 1ull << ((off - lo) & 63));
            if (__syncthreads_or(off >= hi_pos)) break;
        }
        // minimizer candidates
        for (uint32_t i = tid; i < SKL_TILE + (uint32_t)w - 1; i += SKL_THREADS) {
            const uint32_t f = (uint32_t)(bases64_at(L.W, i) >> (64 - 2 * SK_M)), r = sk_rc_mmer(f);
            L.mh[i] = sk_order(f < r ? f : r);
        }
        __syncthreads();
        // windows: inside one read (no start among the k - 1 positions behind the first), of this batch
        const uint64_t p = lo + tid;
        const bool valid = p >= base_lo && p + (uint64_t)k <= n_bases && (bits64_at(L.starts, tid + 1) & ((1ull << (k - 1)) - 1)) == 0;
        uint32_t hm = SK_NONE;
        if (valid)
            for (int i = 0; i < w; i++) hm = min(hm, L.mh[tid + i]);
        L.hmin[tid] = hm;
        {
            const uint64_t vm = __ballot(valid);
            if ((tid & 63) == 0) L.valid[tid >> 6] = vm;
            if (tid == 0) { L.valid[SKL_TILE / 64] = 0; L.valid[SKL_TILE / 64 + 1] = 0; }
        }
        __syncthreads();
        // a run continues into position i only from a window at i - 1 of the same minimizer; the tile's first position never continues one
        const bool prev_same = tid > 0 && (L.valid[(tid - 1) >> 6] >> ((tid - 1) & 63) & 1) && L.hmin[tid - 1] == hm;
        {
            const uint64_t bm = __ballot(!valid || !prev_same);
            if ((tid & 63) == 0) L.brk[tid >> 6] = bm;
            if (tid == 0) { L.brk[SKL_TILE / 64] = ~0ull; L.brk[SKL_TILE / 64 + 1] = ~0ull; }
        }
        __syncthreads();
        if (valid) {
            // where my run began: the last break at or before me (it is a window: runs are made of windows)
            int wd = (int)(tid >> 6);
            uint64_t m = L.brk[wd] & (~0ull >> (63 - (tid & 63)));
            while (!m) m = L.brk[--wd];
            const uint32_t run_start = (uint32_t)wd * 64 + 63 - (uint32_t)__clzll((long long)m);
            if (((tid - run_start) & (SK_MAX_WINDOWS - 1)) == 0) {  // every 16th window of a run starts a record
                const uint32_t ahead = (uint32_t)(bits64_at(L.brk, tid + 1) & 0x7FFF) | 0x8000u;  // breaks at i + 1 .. i + 15
                const uint32_t nw = (uint32_t)__builtin_ctz(ahead) + 1;
                const uint32_t nb = (uint32_t)k + nw - 1;  // bases of the record (<= 78)
                uint64_t q1 = bases64_at(L.W, tid), q2 = bases64_at(L.W, tid + 32), q3 = bases64_at(L.W, tid + 64);
                if (nb < 64) q2 = nb > 32 ? q2 & (~0ull << (2 * (64 - nb))) : 0;
                q3 = nb > 64 ? q3 & (~0ull << (2 * (96 - nb))) : 0;
                q3 = (q3 & ~0xFull) | (nw - 1);
                SkLRec rec;
                rec.a.x = sk_bin(hm);
                rec.a.y = ptr_base == ~0ull ? 0u : ptr_encode(ptr_base + p);
                rec.a.z = (uint32_t)q1; rec.a.w = (uint32_t)(q1 >> 32);
                rec.b.x = (uint32_t)q2; rec.b.y = (uint32_t)(q2 >> 32);
                rec.b.z = (uint32_t)q3; rec.b.w = (uint32_t)(q3 >> 32);
                const uint32_t d = mulhi32(rec.a.x, np1);
                const uint64_t dst = (uint64_t)L.wcur[d] + atomicAdd(&L.cnt[d], 1u);
                if (dst < cap) {
                    out[((uint64_t)d * gridDim.x + blockIdx.x) * cap + dst] = rec;
                } else {
                    const unsigned long long i = atomicAdd(sp.count, 1ull);
                    if (i < sp.cap) sp.recs[i] = rec; else atomicExch(sp.lost, 1u);
                }
            }
        }
        __syncthreads();
        if (tid < np1) { L.wcur[tid] += L.cnt[tid]; L.cnt[tid] = 0; }
    }
    __syncthreads();
    if (tid < np1) seg_counts[(uint64_t)tid * gridDim.x + blockIdx.x] = min(L.wcur[tid], (uint32_t)min(cap, (uint64_t)0xFFFFFFFFu));
}

// ---------------------------------------------------------------------------------------------------------------
// P2: one workgroup per level-1 bucket; its segments are read as one stream and scattered by the next digits of the bin
// word into the bucket's m2 leaves (count_pipeline.h k_sk2_scatter, for 32-byte records).
struct Skl2Lds {
    uint32_t cnt[PT_MAX_LEAVES2], wcur[PT_MAX_LEAVES2];
    uint32_t seg_prefix[PT_THREADS + 1];
    uint32_t wave_tot[PT_THREADS / 64];
    uint32_t tile_seg;
};
__global__ void __launch_bounds__(PT_THREADS) k_skl2_scatter(const SkLRec *__restrict__ in, uint64_t seg_cap1, const uint32_t *__restrict__ seg_counts1,
                                                             uint32_t n_buckets1, uint32_t np1, uint32_t m2, uint32_t nseg_in, uint32_t *leaf_counts,
                                                             uint64_t cap2, SkLRec *out, SkLSpill sp)
{
    __shared__ Skl2Lds L;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t TILE2 = PT_THREADS * 2;
    for (uint32_t bucket = blockIdx.x; bucket < n_buckets1; bucket += gridDim.x) {
        __syncthreads();
        if (tid < PT_MAX_LEAVES2) { L.wcur[tid] = 0; L.cnt[tid] = 0; }
        {   // exclusive prefix of the bucket's segment fill levels (nseg_in <= PT_THREADS: one per thread)
            const uint32_t c = tid < nseg_in ? seg_counts1[(uint64_t)bucket * nseg_in + tid] : 0u;
            uint32_t x = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(x, o);
                if ((int)(tid & 63u) >= o) x += y;
            }
            if ((tid & 63u) == 63u) L.wave_tot[tid >> 6] = x;
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t i = 0; i < (tid >> 6); i++) before += L.wave_tot[i];
            if (tid < nseg_in) L.seg_prefix[tid] = before + x - c;
            if (tid == PT_THREADS - 1) L.seg_prefix[nseg_in] = before + x;
        }
        __syncthreads();
        const uint32_t total = L.seg_prefix[nseg_in];
        for (uint32_t first = 0; first < total; first += TILE2) {
            if (tid == 0) {  // segment of the tile's first record: largest sg with seg_prefix[sg] <= first
                uint32_t lo_s = 0, hi_s = nseg_in;
                while (hi_s - lo_s > 1) {
                    const uint32_t mid = (lo_s + hi_s) >> 1;
                    if (L.seg_prefix[mid] <= first) lo_s = mid; else hi_s = mid;
                }
                L.tile_seg = lo_s;
            }
            __syncthreads();
            SkLRec rec[2];
            bool have[2];
            uint32_t sg = L.tile_seg;  // (a thread's records ascend: the walk carries on from the previous one's segment)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t e = first + (uint32_t)j * PT_THREADS + tid;
                have[j] = e < total;
                if (have[j]) {
                    while (e >= L.seg_prefix[sg + 1]) sg++;
                    rec[j] = in[((uint64_t)bucket * nseg_in + sg) * seg_cap1 + (e - L.seg_prefix[sg])];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (have[j]) {
                    const uint32_t d = mulhi32(rec[j].a.x, np1 * m2) - bucket * m2;
                    const uint64_t dst = (uint64_t)L.wcur[d] + atomicAdd(&L.cnt[d], 1u);
                    if (dst < cap2) {
                        out[((uint64_t)bucket * m2 + d) * cap2 + dst] = rec[j];
                    } else {
                        const unsigned long long i = atomicAdd(sp.count, 1ull);
                        if (i < sp.cap) sp.recs[i] = rec[j]; else atomicExch(sp.lost, 1u);
                    }
                }
            __syncthreads();
            if (tid < m2) { L.wcur[tid] += L.cnt[tid]; L.cnt[tid] = 0; }
        }
        __syncthreads();
        if (tid < m2) leaf_counts[(uint64_t)bucket * m2 + tid] = min(L.wcur[tid], (uint32_t)min(cap2, (uint64_t)0xFFFFFFFFu));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// P3: one workgroup per leaf (count_pipeline.h k_p3_merge: region image in LDS, state / new-key bookkeeping per leaf, the
// two sweeps of a leaf that covers 2^g regions, the count of keys at the coverage threshold); a lane walks a record.
__global__ void __launch_bounds__(P3_THREADS) k_p3l_merge(const SkLRec *__restrict__ leaf_recs, const uint32_t *__restrict__ seg_counts, uint64_t seg_cap,
                                                         uint32_t nseg, uint32_t n_leaves, uint32_t g, TableView t, int virgin, uint32_t *leaf_state,
                                                         uint32_t *leaf_new, uint32_t *any_failed, uint32_t solid_thr, unsigned long long *n_solid,
                                                         int k, uint64_t p5k, uint64_t p5km1)
{
    __shared__ MergeLds L;
    const uint32_t tid = threadIdx.x;
    long long solid_delta = 0;
    unsigned long long n_empty = 0;
    const uint32_t ptr_from = solid_thr >= 2 ? 1u : 0u;
    for (uint32_t leaf = blockIdx.x; leaf < n_leaves; leaf += gridDim.x) {
        if (leaf_state[leaf]) continue;  // (uniform)
        bool leaf_ok = true;
        uint32_t new_total = 0;
        unsigned long long leaf_empty = 0;
        const int sweeps = g == 0 ? 1 : 2;
        for (int sweep = 0; sweep < sweeps && leaf_ok; sweep++) {
            const bool commit = g == 0 || sweep == 1;
            for (uint32_t sub = 0; sub < (1u << g); sub++) {
                const uint64_t region = ((uint64_t)leaf << g) | sub;
                Slot *gs = t.slots + region * REGION_SLOTS;
                int solid_before = 0;
                __syncthreads();
                for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                    if (virgin) {
                        L.key[i] = EMPTY_KEY; L.cnt[i] = 0; L.aux[i] = 0;
                    } else {
                        const uint4 raw = *reinterpret_cast<const uint4 *>(gs + i);
                        L.key[i] = ((uint64_t)raw.y << 32) | raw.x;
                        L.cnt[i] = min(raw.z, P3_COUNT_CAP);
                        L.aux[i] = raw.w;
                        solid_before += solid_thr && raw.z >= solid_thr;
                    }
                }
                if (tid == 0) { L.n_new = 0; L.overflow = 0; }
                __syncthreads();
                uint32_t my_new = 0;
                for (uint32_t sgm = 0; sgm < nseg; sgm++) {
                    const uint32_t n = min(seg_counts[(uint64_t)leaf * nseg + sgm], (uint32_t)min(seg_cap, (uint64_t)0xFFFFFFFFu));
                    const SkLRec *recs = leaf_recs + ((uint64_t)leaf * nseg + sgm) * seg_cap;
                    for (uint32_t r = tid; r < n; r += P3_THREADS) {
                        const SkLRec rec = recs[r];
                        if (g && mulhi32(rec.a.x, t.n_regions) != region) continue;
                        skl_expand_poly(rec, k, p5k, p5km1, [&](uint64_t key, uint32_t j) {
                            if (key == EMPTY_KEY) {  // (the table's free marker: counted apart, kmer_device.h empty_cnt)
                                if (commit) leaf_empty++;
                                return;
                            }
                            const uint32_t ptr = ptr_advance(rec.a.y, j);
                            if (!lds_region_add(L, key, ptr, sk_home(key), my_new, ptr_pick(key, ptr_from, solid_thr), ptr_pick_late(key, ptr_from)))
                                atomicExch(&L.overflow, 1u);
                        });
                    }
                }
                if (my_new) atomicAdd(&L.n_new, my_new);
                __syncthreads();
                const bool ovf = L.overflow != 0;
                if (ovf) leaf_ok = false;
                if (commit && !ovf) {
                    for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                        uint4 v;
                        const uint64_t kk = L.key[i];
                        v.x = (uint32_t)kk; v.y = (uint32_t)(kk >> 32);
                        v.z = min(L.cnt[i], P3_COUNT_CAP);
                        v.w = L.aux[i];
                        *reinterpret_cast<uint4 *>(gs + i) = v;
                        solid_delta += solid_thr && v.z >= solid_thr;
                    }
                    solid_delta -= solid_before;
                    if (tid == 0) new_total += L.n_new;
                }
                if (ovf) break;
            }
        }
        if (!leaf_ok && virgin) {  // nothing was there: leave valid empty regions behind
            for (uint32_t sub = 0; sub < (1u << g); sub++) {
                Slot *gs = t.slots + (((uint64_t)leaf << g) | sub) * REGION_SLOTS;
                for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                    uint4 v; v.x = 0xFFFFFFFFu; v.y = 0xFFFFFFFFu; v.z = 0; v.w = 0;
                    *reinterpret_cast<uint4 *>(gs + i) = v;
                }
            }
        }
        if (leaf_ok) n_empty += leaf_empty;
        if (tid == 0) {
            if (leaf_ok) { leaf_state[leaf] = 1; leaf_new[leaf] = new_total; } else atomicExch(any_failed, 1u);
        }
    }
    if (solid_thr) wave_add_ull(n_solid, (unsigned long long)solid_delta);
    wave_add_ull(t.empty_cnt, n_empty);
}

// records through the direct path (a table addressed by key hash): the spill list, the leaves P3 left unmerged after the
// table went back to hash regions, and batches too small for the pipeline never come here (they are counted per window)
__global__ void k_skl_add_records(const SkLRec *__restrict__ recs, uint64_t n, int k, uint64_t p5k, uint64_t p5km1, TableView t, uint32_t solid_thr,
                                  unsigned long long *n_solid)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_new = 0, n_cross = 0, n_empty = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const SkLRec rec = recs[i];
        skl_expand_poly(rec, k, p5k, p5km1, [&](uint64_t key, uint32_t j) {
            if (key == EMPTY_KEY) { n_empty++; return; }
            uint32_t before;
            n_new += table_add(t, key, 1u, ptr_advance(rec.a.y, j), &before);
            n_cross += crosses(before, 1u, solid_thr);
        });
    }
    wave_add_ull(t.n_used, n_new);
    wave_add_ull(t.empty_cnt, n_empty);
    if (solid_thr) wave_add_ull(n_solid, n_cross);
}

__global__ void k_skl_add_unmerged(const SkLRec *__restrict__ leaf_recs, const uint32_t *__restrict__ seg_counts, uint64_t seg_cap, uint32_t nseg,
                                   uint32_t leaf_lo, uint32_t leaf_hi, const uint32_t *__restrict__ leaf_state, int k, uint64_t p5k, uint64_t p5km1,
                                   TableView t, uint32_t solid_thr, unsigned long long *n_solid)
{
    unsigned long long n_new = 0, n_cross = 0, n_empty = 0;
    for (uint32_t leaf = leaf_lo + blockIdx.x; leaf < leaf_hi; leaf += gridDim.x) {
        if (leaf_state[leaf]) continue;
        for (uint32_t sgm = 0; sgm < nseg; sgm++) {
            const uint64_t n = min((uint64_t)seg_counts[(uint64_t)leaf * nseg + sgm], seg_cap);
            const SkLRec *recs = leaf_recs + ((uint64_t)leaf * nseg + sgm) * seg_cap;
            for (uint64_t r = threadIdx.x; r < n; r += blockDim.x) {
                const SkLRec rec = recs[r];
                skl_expand_poly(rec, k, p5k, p5km1, [&](uint64_t key, uint32_t j) {
                    if (key == EMPTY_KEY) { n_empty++; return; }
                    uint32_t before;
                    n_new += table_add(t, key, 1u, ptr_advance(rec.a.y, j), &before);
                    n_cross += crosses(before, 1u, solid_thr);
                });
            }
        }
    }
    wave_add_ull(t.n_used, n_new);
    wave_add_ull(t.empty_cnt, n_empty);
    if (solid_thr) wave_add_ull(n_solid, n_cross);
}

}  // namespace mc

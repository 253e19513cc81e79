// K2/K3, partitioned form: counting without one memory-side atomic per k-mer occurrence.
//
// The direct kernel (k_count_reads, mcgpu.hip) does what the reference does per occurrence
// (src/io/IOUtils.java:207-208: hm.addAndBound(key, 1)): a probe plus an atomic on a random 16-byte
// slot.  MI355X retires only ~2*10^10 scattered atomics per second chip-wide, far below what HBM
// can stream, so large batches take this route instead:
//
//   P1  extract + scatter   reads -> keys (+ the read-context hint of the occurrence), grouped into np1
//                           level-1 buckets by their bin word (mulhi32 below); tiles of 8192 windows staged
//                           in LDS and written out as one contiguous run per bucket.  Each of the 256 workgroups
//                           owns one segment of every bucket: fill levels live in LDS, no atomics,
//                           and a run's cache lines are completed by the workgroup that started them;
//   P2  scatter             one workgroup per level-1 bucket scatters it again into the bucket's m2 leaves:
//                           np1 * m2 = #leaves = #regions of the table (x 2^g for very large tables), so a
//                           leaf holds exactly the keys of one table region;
//   P3  merge               one workgroup per region: the region (4096 slots, 64 KB) lives in LDS, the
//                           leaf's keys are streamed in and counted with LDS atomics, the region goes
//                           back to HBM with plain coalesced stores.
//
// Every byte moved is a coalesced stream and the counting pipeline issues no global atomics at all
// (only the multi-GPU owner split reserves its packed output ranges with one atomic per tile and owner).
// Results are identical to the direct kernel: same slot placement rule (home slot from the hash,
// linear probing inside the region), saturation as in kmer_device.h.
#pragma once
#include <utility>
#include "kmer_device.h"

namespace mc {

constexpr int PT_THREADS = 1024;                    // workgroup of the scatter kernels
constexpr int PT_ITEMS = 8;
constexpr int PT_TILE = PT_THREADS * PT_ITEMS;      // 8192 windows / keys per tile
constexpr int PT_MAX_BUCKETS = 512;                 // fan-out of one scatter level (what a run takes unless the table needs more leaves)
constexpr int PT_MAX_BUCKETS_KEYS = 1024;           // ... at most, in the per-window pipeline (k_p1_extract_scatter / k_p2_scatter: a thread per bucket): 2^20 leaves
constexpr int PT_MAX_LEAVES2 = 1024;                // ... of the second level of the super-k-mer pipeline (k_sk2_scatter: an LDS cursor per leaf)
constexpr int PT_MAX_BUCKETS1_SK = 2048;             // level-1 buckets of the super-k-mer pipeline at most (512 unless the table needs more leaves, then 1024, then 2048)
constexpr int SK_LEAVES_LG = 21;                    // so up to 2048 x 1024 leaves of one region each (8.6 G slots of 16 bytes = 137 GB of table)
constexpr int P3_THREADS = 512;
constexpr uint32_t REGION_SLOTS = 1u << MC_REGION_LG;  // == 1 << mc_ctx::sb (4096: a 64 KB image in LDS)
constexpr uint32_t CURSOR1_STRIDE = 32;             // owner cursors of the multi-GPU split sit on separate 128-byte lines
constexpr int PT_SEGMENTS = 256;                    // workgroups of P1 (one per CU) = segments of every level-1 bucket

// Bucket digits.  A key's (or record's) 32-bit bin word places it by multiplication, not by bit prefix:
// region = (bin * n_regions) >> 32, so a table may have any number of regions; with np1 level-1 buckets of
// m2 leaves each (np1 * m2 leaves in all) bucket = (bin * np1) >> 32 and leaf = (bin * np1 * m2) >> 32 =
// bucket * m2 + leaf-in-bucket (floor(floor(x * np1 * m2) / m2) == floor(x * np1)).  Powers of two give bit prefixes.
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t bin, uint32_t n) { return (uint32_t)(((uint64_t)bin * n) >> 32); }

struct ScatterLds {
    uint64_t key[PT_TILE];
    uint32_t hint[PT_TILE];
    uint32_t rel[PT_TILE];                       // where the staged record goes, relative to out_base (~0 = spill)
    uint32_t cnt[PT_MAX_BUCKETS_KEYS], off[PT_MAX_BUCKETS_KEYS], gbase[PT_MAX_BUCKETS_KEYS];
    uint32_t wcur[PT_MAX_BUCKETS_KEYS];          // this workgroup's own fill level of every bucket
    uint32_t seg_prefix[PT_SEGMENTS + 1];        // P2: where each input segment starts in the bucket's stream
    uint32_t starts[(PT_TILE + 256) / 32 + 2];  // P1: bit per base position = "a read starts here"
    uint32_t wave_tot[PT_THREADS / 64];
    uint32_t n_valid, tile_seg;
    uint16_t polyF[256], polyR[256];             // P1, polynomial keys: kmer_device.h poly_hashes_tabled
    unsigned long long dbg[4];
};

struct SpillView {
    uint64_t *keys;
    uint32_t *hints;
    unsigned long long *count;
    uint64_t cap;
    uint32_t *lost;  // set when the spill list itself is full
};

__device__ __forceinline__ void spill_push(const SpillView &sp, uint64_t key, uint32_t hint)
{
    const unsigned long long i = atomicAdd(sp.count, 1ull);
    if (i < sp.cap) {
        sp.keys[i] = key;
        sp.hints[i] = hint;
    } else {
        atomicExch(sp.lost, 1u);
    }
}

// Shared tail of P1/P2: the tile's items are in registers; group them by digit in LDS and write one
// contiguous run per bucket.
//   counting pipeline (cursors == nullptr): every workgroup owns one SEGMENT of every bucket and
//     appends to it alone -- fill levels live in LDS (L.wcur), no global atomics, and the cache lines
//     of a run are completed by the same workgroup, so L2 writes whole lines.  Bucket d's piece for
//     this workgroup is out_*[out_base + d * bucket_stride ..] with room for `cap` records; what
//     does not fit goes to the spill list (drained by the direct kernel).
//   multi-GPU split (cursors + bases): exact bucket offsets from a COUNT_ONLY run of the same tiles,
//     global cursors count from 0 inside each bucket, output tightly packed; out_hints may be null.
template <bool COUNT_ONLY = false>
__device__ __forceinline__ void scatter_tile(ScatterLds &L, const uint64_t (&key)[PT_ITEMS], const uint32_t (&hint)[PT_ITEMS],
                                             const uint32_t (&dig)[PT_ITEMS], const bool (&valid)[PT_ITEMS],
                                             uint32_t n_buckets, uint32_t *cursors, uint64_t cap, uint64_t bucket_stride,
                                             uint64_t *out_keys, uint32_t *out_hints, uint64_t out_base, const SpillView &sp,
                                             const uint64_t *bases = nullptr)
{
    const uint32_t tid = threadIdx.x;
#ifdef MC_P1_TIMING
    unsigned long long ts_[5];
    ts_[0] = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t rank[PT_ITEMS];
#pragma unroll
    for (int j = 0; j < PT_ITEMS; j++) rank[j] = valid[j] ? atomicAdd(&L.cnt[dig[j]], 1u) : 0u;
    __syncthreads();
#ifdef MC_P1_TIMING
    ts_[1] = __builtin_amdgcn_s_memrealtime();
#endif
    // exclusive scan of cnt[0..n_buckets) (n_buckets <= PT_MAX_BUCKETS_KEYS <= blockDim)
    {
        const uint32_t lane = tid & 63, wv = tid >> 6;
        const uint32_t c = tid < n_buckets ? L.cnt[tid] : 0u;
        uint32_t x = c;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if ((int)lane >= o) x += y;
        }
        if (lane == 63) L.wave_tot[wv] = x;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t i = 0; i < wv; i++) before += L.wave_tot[i];
        if (tid < n_buckets) {
            L.off[tid] = before + x - c;
            if (cursors) {
                L.gbase[tid] = c ? atomicAdd(&cursors[(uint64_t)tid * CURSOR1_STRIDE], c) : 0u;  // one atomic per (tile, owner)
            } else {
                L.gbase[tid] = L.wcur[tid];
                L.wcur[tid] += c;
            }
        }
        if (tid == PT_THREADS - 1) L.n_valid = before + x;
    }
    __syncthreads();
    if (COUNT_ONLY) return;
#ifdef MC_P1_TIMING
    ts_[2] = __builtin_amdgcn_s_memrealtime();
#endif
    // stage by bucket; the destination of every record is computed here (while its bucket number is in a
    // register) so that the write-out below is three independent LDS reads and two stores per record
#pragma unroll
    for (int j = 0; j < PT_ITEMS; j++)
        if (valid[j]) {
            const uint32_t d = dig[j];
            const uint32_t pos = L.off[d] + rank[j];
            const uint64_t dst = (uint64_t)L.gbase[d] + rank[j];
            L.key[pos] = key[j];
            L.hint[pos] = hint[j];
            if (bases) L.rel[pos] = d;  // (multi-GPU split: 64-bit bases, resolved at write-out; rank recomputed)
            else L.rel[pos] = dst < cap ? (uint32_t)((uint64_t)d * bucket_stride + dst) : 0xFFFFFFFFu;
        }
    __syncthreads();
#ifdef MC_P1_TIMING
    ts_[3] = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t n = L.n_valid;
    for (uint32_t i = tid; i < n; i += PT_THREADS) {
        const uint32_t r = L.rel[i];
        if (bases) {
            const uint64_t at = bases[r] + L.gbase[r] + (i - L.off[r]);
            out_keys[at] = L.key[i];
            if (out_hints) out_hints[at] = L.hint[i];
        } else if (r != 0xFFFFFFFFu) {
            const uint64_t at = out_base + r;
            out_keys[at] = L.key[i];
            out_hints[at] = L.hint[i];
        } else {
            spill_push(sp, L.key[i], L.hint[i]);
        }
    }
    __syncthreads();
#ifdef MC_P1_TIMING
    ts_[4] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) for (int q = 0; q < 4; q++) L.dbg[q] += ts_[q + 1] - ts_[q];
#endif
}

// first read that can matter for the tile starting at base `lo`: largest r with offsets[r] <= lo
// (one thread per tile; the scatter kernel then walks forward from there)
__global__ void k_tile_first_read(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t n_tiles,
                                  uint32_t *__restrict__ first_read, uint32_t tile_size = PT_TILE)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint64_t lo = t * (uint64_t)tile_size;
    const uint64_t back = lo >= 64 ? lo - 64 : 0;  // hints look a few bases to the left of the tile
    uint64_t a = 0, b = n_reads;                    // invariant: offsets[a] <= back
    while (b - a > 1) {
        const uint64_t m = (a + b) >> 1;
        if (offsets[m] <= back) a = m; else b = m;
    }
    first_read[t] = (uint32_t)a;
}

// The same table from the reads' side, for short reads (a few per tile): read r names itself for the tiles whose `back`
// lies inside it -- two coalesced loads and usually one store or none per read instead of a 23-step dependent search per
// tile (84 -> ~20 us for 10 M reads of 150 bases).  Tiles in front of the first read get 0, as the search gives them.
__global__ void k_tile_first_read_by_reads(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t n_tiles,
                                           uint32_t *__restrict__ first_read, uint32_t tile_size)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t a = offsets[r], b = offsets[r + 1];
    if (b <= a && r != 0) return;  // (an empty read owns no position)
    // back(t) = max(t * tile_size - 64, 0) in [a, b)  <=>  t in [ceil((a + 64) / T), ceil((b + 64) / T)), and t = 0 where a == 0
    uint64_t t_lo = (a + 64 + tile_size - 1) / tile_size, t_hi = std::min<uint64_t>((b + 64 + tile_size - 1) / tile_size, n_tiles);
    if (r == 0) t_lo = 0;  // (tile 0, and whatever lies in front of the first read)
    else if (a == 0) t_lo = 0;
    for (uint64_t t = t_lo; t < t_hi; t++) first_read[t] = (uint32_t)r;
}

__device__ __forceinline__ uint64_t starts_window(const uint32_t *bits, uint32_t pos)
{  // 64 bits of the bitmap starting at bit `pos`
    const uint32_t w = pos >> 5, s = pos & 31;
    const uint64_t lo = ((uint64_t)bits[w + 1] << 32) | bits[w];
    const uint64_t hi = bits[w + 2];
    return s ? ((lo >> s) | (hi << (64 - s))) : lo;
}

__device__ __forceinline__ uint32_t base_or0(const uint64_t *__restrict__ words, uint64_t q, uint64_t n_bases)
{
    return q < n_bases ? (uint32_t)(words[q >> 5] >> (62 - 2 * (q & 31))) & 3u : 0u;
}

__device__ __forceinline__ uint64_t bits128(uint64_t lo, uint64_t hi, uint32_t from, uint32_t cnt)
{  // cnt <= 63 bits of the 128-bit word hi:lo starting at bit `from`
    if (cnt == 0) return 0;
    uint64_t x;
    if (from >= 64) x = hi >> (from - 64);
    else x = from ? ((lo >> from) | (hi << (64 - from))) : lo;
    return x & ((1ull << cnt) - 1);
}

// P1: tiles of PT_TILE consecutive base positions of the packed read set.  A thread owns PT_ITEMS
// CONSECUTIVE positions and rolls the window along them (the reference's ShortKmer.shiftRight,
// itmo!/dna/kmers/ShortKmer.java:68-71): one new base per step instead of a fresh extraction.
// bucket of a key: mulhi32(its bin word, np1) (counting pipeline), or its owner rank (multi-GPU split;
// low hash bits, disjoint from the bits that place it in the table)
// (owner_of: kmer_device.h)

// OWNERS: the np1 buckets are owner ranks instead of ranges of the bin word, EMPTY_KEY is an
// ordinary key, and the output is packed at `bases` (from a COUNT_ONLY run of the same kernel).
template <int MODE, bool OWNERS = false, bool COUNT_ONLY = false>
__global__ void __launch_bounds__(PT_THREADS) k_p1_extract_scatter(
    const uint64_t *__restrict__ words, const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t base_lo,
    uint64_t n_bases, uint64_t n_tiles, const uint32_t *__restrict__ first_read, int k, uint32_t np1, uint32_t *cursors, uint64_t cap,
    uint64_t *out_keys, uint32_t *out_hints, unsigned long long *empty_cnt, SpillView sp, const uint64_t *bases = nullptr,
    int mm_k = 0, uint64_t ptr_base = ~0ull)
{
    __shared__ ScatterLds L;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_buckets = np1;  // (owner ranks, or level-1 buckets)
    if (tid < PT_MAX_BUCKETS_KEYS) L.wcur[tid] = 0;  // (counting pipeline: gridDim.x == PT_SEGMENTS, segment = blockIdx.x)
    if (MODE == KEY_POLY && tid < 256) poly_tables_fill(L.polyF, L.polyR, tid);  // (published by the first tile's barriers)
#ifdef MC_P1_TIMING
    if (tid < 4) L.dbg[tid] = 0;
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tl = 0;
#define P1_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); if (i) tph[(i) > 0 ? (i)-1 : 0] += n_ - tl; tl = n_; } while (0)
#else
#define P1_STAMP(i) do {} while (0)
#endif
    constexpr int64_t MARGIN = 64;  // the bitmap starts this many bases left of the tile (virtually, for tile 0)
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    constexpr uint64_t INV5 = 0xCCCCCCCCCCCCCCCDull;  // 5 * INV5 == 1 (mod 2^64)
    uint64_t pow5_km1 = 1;
    for (int i = 1; i < k; i++) pow5_km1 *= 5;
    const uint64_t pow5_k = pow5_km1 * 5;
    for (uint64_t tile = base_lo / PT_TILE + blockIdx.x; tile < n_tiles; tile += gridDim.x) {  // reads cover [base_lo, n_bases)
        const uint64_t lo = tile * (uint64_t)PT_TILE;
        const int64_t bm_lo = (int64_t)lo - MARGIN;                    // position of bit 0 of the bitmap
        const uint64_t bm_hi = lo + PT_TILE + 128;                      // one past the last position it covers
        P1_STAMP(0);
        for (uint32_t i = tid; i < sizeof(L.starts) / 4; i += PT_THREADS) L.starts[i] = 0;
        if (tid < n_buckets) L.cnt[tid] = 0;
        __syncthreads();
        for (uint64_t r = (uint64_t)first_read[tile] + tid; r < n_reads; r += PT_THREADS) {
            const uint64_t s = offsets[r];
            if (s >= bm_hi) break;
            const int64_t rel = (int64_t)s - bm_lo;
            if (rel >= 0) atomicOr(&L.starts[(uint32_t)rel >> 5], 1u << ((uint32_t)rel & 31));
        }
        __syncthreads();
        P1_STAMP(1);
        uint64_t key[PT_ITEMS];
        uint32_t hint[PT_ITEMS], dig[PT_ITEMS];
        bool valid[PT_ITEMS];
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) { key[j] = 0; hint[j] = 0; dig[j] = 0; valid[j] = false; }
        const uint64_t p0 = lo + (uint64_t)tid * PT_ITEMS;
        if (p0 + (uint64_t)k <= n_bases) {
            // read-start bits around my positions: position q <-> bit q - (p0 - 8)
            const uint32_t wrel = (uint32_t)((int64_t)p0 - 8 - bm_lo);
            const uint64_t w_lo = starts_window(L.starts, wrel), w_hi = starts_window(L.starts, wrel + 64);
            Kmer v = extract_kmer(words, p0, k);
            uint64_t rc = MODE == KEY_PACKED ? rc_packed(v.lo, k) : 0;
            // polynomial keys roll too: src/utils/PolynomialHash.java:19-28 is h = 5^k + sum b_i 5^(k-1-i) in the ring
            // of 64-bit integers (and 5^k + sum (3 - b_i) 5^i for the other strand), so sliding the window by one
            // base is a handful of operations instead of 2k multiply-adds, with the same value bit for bit
            uint64_t hf = 1, hr = 1;
            // Polynomial keys, a thread whose eight windows all lie inside the data (all but the last tile's): the forward
            // hashes roll left to right, the reverse-strand ones RIGHT TO LEFT -- that way both rolls are "times 5, plus the
            // base that comes in, minus (4 + the base that goes out) x 5^k", a few shifts, adds and selects.  (Round 3 rolled
            // both left to right: the reverse hash then loses its LOWEST term and must be divided by 5 -- a 64-bit
            // multiplication by 5^-1 -- and both strands multiplied a base by 5^k or 5^(k-1): three 64-bit multiplications a
            // window at a quarter of the vector unit's rate, ~300 cycles, more than everything else a window costs here.)
            const bool poly_fast = MODE == KEY_POLY && p0 + (uint64_t)(PT_ITEMS - 1) + (uint64_t)k <= n_bases;
            uint64_t hr8[PT_ITEMS];
            uint32_t ob8 = 0, ib8 = 0;  // the windows' first bases / the bases behind them, 2 bits each, window 0 on top of 16 bits
            if (poly_fast) {
                // p0 is a multiple of 8: the eight bases from p0 (and from p0 + k) sit in at most two words
                const uint64_t qo = p0, qi = p0 + (uint64_t)k, last_w = (n_bases + 31) / 32;
                const uint64_t wo = words[qo >> 5];
                ob8 = (uint32_t)(wo >> (48 - 2 * (qo & 31))) & 0xFFFFu;  // (qo & 31 <= 24)
                const uint64_t wi0 = words[qi >> 5], wi1 = words[min((qi >> 5) + 1, last_w)];
                const uint32_t sh = 2 * (uint32_t)(qi & 31);
                ib8 = (uint32_t)((sh ? ((wi0 << sh) | (wi1 >> (64 - sh))) : wi0) >> 48);
                hf = poly_hash_f_tabled(v, k, L.polyF);
                hr8[PT_ITEMS - 1] = poly_hash_r_tabled(extract_kmer(words, p0 + PT_ITEMS - 1, k), k, L.polyR);
#pragma unroll
                for (int j = PT_ITEMS - 2; j >= 0; j--) {
                    const uint32_t bf = (ob8 >> (14 - 2 * j)) & 3u, inb = (ib8 >> (14 - 2 * j)) & 3u;
                    hr8[j] = hr8[j + 1] * 5 + (3u - bf) - poly_4x_times(3u - inb, pow5_k);
                }
            } else if (MODE == KEY_POLY) {
                poly_hashes_tabled(v, k, L.polyF, L.polyR, &hf, &hr);
            }
#pragma unroll
            for (int j = 0; j < PT_ITEMS; j++) {
                const uint64_t p = p0 + (uint64_t)j;
                if (p + (uint64_t)k > n_bases) break;
                if (poly_fast) hr = hr8[j];
                const uint32_t b = 8 + (uint32_t)j;
                // the window [p, p+k) lies inside one read iff no read starts at p+1 .. p+k-1
                if (p >= base_lo && bits128(w_lo, w_hi, b + 1, (uint32_t)(k - 1)) == 0) {
                    bool flipped;
                    if (MODE == KEY_PACKED) {
                        flipped = rc < v.lo;
                        key[j] = flipped ? rc : v.lo;
                    } else if (MODE == KEY_POLY) {
                        flipped = (int64_t)hr < (int64_t)hf;  // Math.min on signed longs
                        key[j] = flipped ? hr : hf;
                    } else {
                        key[j] = (uint64_t)key_of<MODE>(v, k, &flipped);
                    }
                    hint[j] = ptr_base == ~0ull ? 0u : ptr_encode(ptr_base + p);  // the occurrence's place in the read store
                    if (OWNERS) {
                        valid[j] = true;
                        // (mm_k != 0 here: the owner of the key's MINIMIZER, as the super-k-mer records of the same group are
                        // dealt -- a batch that falls back to keys must not send a k-mer to another owner than the records did)
                        dig[j] = mm_k ? sk_owner(sk_hmin_of_kmer(key[j], mm_k), n_buckets) : owner_of(key[j], n_buckets);
                    } else if (key[j] == EMPTY_KEY) {
                        atomicAdd(empty_cnt, 1ull);
                    } else {
                        valid[j] = true;
                        dig[j] = mulhi32(bin32_of(key[j], mm_k), np1);
                    }
                }
                // roll to p + 1: the window takes the first base after it, loses its first base
                if (poly_fast) {
                    if (j + 1 < PT_ITEMS) hf = hf * 5 + ((ib8 >> (14 - 2 * j)) & 3u) - poly_4x_times((ob8 >> (14 - 2 * j)) & 3u, pow5_k);
                    continue;
                }
                const uint32_t in = base_or0(words, p + (uint64_t)k, n_bases), out = base_at(v, k, 0);
                if (MODE == KEY_PACKED) {
                    v.lo = ((v.lo << 2) | in) & kmask;
                    rc = (rc >> 2) | ((uint64_t)(3u - in) << (2 * (k - 1)));
                } else {
                    v = neighbour(v, k, 1, (int)in);
                    if (MODE == KEY_POLY) {
                        hf = hf * 5 + in - (uint64_t)(out + 4u) * pow5_k;  // drop out * 5^(k-1) * 5 and the leading 5^(k+1), add 5^k
                        hr = pow5_k + (hr - pow5_k - (3u - out)) * INV5 + (uint64_t)(3u - in) * pow5_km1;
                    }
                }
            }
        }
        P1_STAMP(2);
        if (OWNERS)
            scatter_tile<COUNT_ONLY>(L, key, hint, dig, valid, n_buckets, cursors, 0, 0, out_keys, out_hints, 0, sp, bases);
        else
            scatter_tile(L, key, hint, dig, valid, n_buckets, nullptr, cap, (uint64_t)PT_SEGMENTS * cap, out_keys, out_hints,
                         (uint64_t)blockIdx.x * cap, sp);
        P1_STAMP(3);
    }
#ifdef MC_P1_TIMING
    if (!OWNERS && blockIdx.x == 7 && tid == 0) printf("[p1 block 7] us: bitmap %.1f items %.1f scatter %.1f  (rank %.1f scan %.1f stage %.1f writeout %.1f)\n", tph[0] * 0.01, tph[1] * 0.01, tph[2] * 0.01, L.dbg[0] * 0.01, L.dbg[1] * 0.01, L.dbg[2] * 0.01, L.dbg[3] * 0.01);
#endif
    if (!OWNERS && tid < n_buckets)  // how much of its segment of every bucket this workgroup filled
        cursors[(uint64_t)tid * PT_SEGMENTS + blockIdx.x] = min(L.wcur[tid], (uint32_t)cap);
}

// P1 for a flat stream of keys (+ optional hints) instead of reads: the receiving side of the
// multi-GPU exchange.  Tiles of PT_TILE consecutive entries.
__global__ void __launch_bounds__(PT_THREADS) k_p1_keys_scatter(const uint64_t *__restrict__ in_keys,
                                                                const uint32_t *__restrict__ in_hints, uint64_t n, uint32_t np1,
                                                                uint32_t *cursors, uint64_t cap, uint64_t *out_keys,
                                                                uint32_t *out_hints, unsigned long long *empty_cnt, SpillView sp,
                                                                int mm_k)
{
    __shared__ ScatterLds L;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_buckets = np1;
    const uint64_t n_tiles = (n + PT_TILE - 1) / PT_TILE;
    if (tid < PT_MAX_BUCKETS_KEYS) L.wcur[tid] = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (tid < n_buckets) L.cnt[tid] = 0;
        __syncthreads();
        uint64_t key[PT_ITEMS];
        uint32_t hint[PT_ITEMS], dig[PT_ITEMS];
        bool valid[PT_ITEMS];
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) {
            const uint64_t i = tile * PT_TILE + tid + (uint64_t)j * PT_THREADS;
            valid[j] = false;
            key[j] = 0; hint[j] = 0; dig[j] = 0;
            if (i < n) {
                key[j] = in_keys[i];
                hint[j] = in_hints ? in_hints[i] : 0u;
                if (key[j] == EMPTY_KEY) {
                    atomicAdd(empty_cnt, 1ull);
                } else {
                    valid[j] = true;
                    dig[j] = mulhi32(bin32_of(key[j], mm_k), np1);
                }
            }
        }
        scatter_tile(L, key, hint, dig, valid, n_buckets, nullptr, cap, (uint64_t)PT_SEGMENTS * cap, out_keys, out_hints,
                     (uint64_t)blockIdx.x * cap, sp);
    }
    if (tid < n_buckets) cursors[(uint64_t)tid * PT_SEGMENTS + blockIdx.x] = min(L.wcur[tid], (uint32_t)cap);
}

// P2: one workgroup per level-1 bucket (it owns all of the bucket's leaves, so again no global
// atomics and whole-line writes).  The bucket's PT_SEGMENTS input segments are read as one stream,
// PT_TILE records per tile, and scattered into the bucket's m2 leaves.
__global__ void __launch_bounds__(PT_THREADS) k_p2_scatter(const uint64_t *__restrict__ in_keys,
                                                           const uint32_t *__restrict__ in_hints, uint64_t seg_cap1,
                                                           const uint32_t *__restrict__ seg_counts1, uint32_t n_buckets1,
                                                           uint32_t np1, uint32_t m2, uint32_t *leaf_counts, uint64_t cap2,
                                                           uint64_t *out_keys, uint32_t *out_hints, SpillView sp, int mm_k,
                                                           uint32_t piece = 0, uint32_t n_pieces = 1)
{
    // piece / n_pieces: a large batch travels in n_pieces pieces that reuse the level-1 buffers one after the other;
    // every leaf then has n_pieces segments of capacity cap2 and this launch fills segment `piece` (as k_sk2_scatter)
    __shared__ ScatterLds L;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_buckets = m2;  // leaves per level-1 bucket
    for (uint32_t bucket = blockIdx.x; bucket < n_buckets1; bucket += gridDim.x) {
        __syncthreads();
        if (tid < PT_MAX_BUCKETS_KEYS) L.wcur[tid] = 0;
        if (tid == 0) {  // (257 additions: not worth a parallel scan)
            uint32_t acc = 0;
            for (int sgm = 0; sgm < PT_SEGMENTS; sgm++) {
                L.seg_prefix[sgm] = acc;
                acc += seg_counts1[(uint64_t)bucket * PT_SEGMENTS + sgm];
            }
            L.seg_prefix[PT_SEGMENTS] = acc;
        }
        __syncthreads();
        const uint32_t total = L.seg_prefix[PT_SEGMENTS];
        for (uint32_t first = 0; first < total; first += PT_TILE) {
            if (tid < n_buckets) L.cnt[tid] = 0;
            if (tid == 0) {  // segment of the tile's first record: largest sg with seg_prefix[sg] <= first
                uint32_t lo_s = 0, hi_s = PT_SEGMENTS;
                while (hi_s - lo_s > 1) {
                    const uint32_t mid = (lo_s + hi_s) >> 1;
                    if (L.seg_prefix[mid] <= first) lo_s = mid; else hi_s = mid;
                }
                L.tile_seg = lo_s;
            }
            __syncthreads();
            uint64_t key[PT_ITEMS];
            uint32_t hint[PT_ITEMS], dig[PT_ITEMS];
            bool valid[PT_ITEMS];
#pragma unroll
            for (int j = 0; j < PT_ITEMS; j++) {  // record first + j*PT_THREADS + tid: consecutive lanes, consecutive records
                const uint32_t e = first + (uint32_t)j * PT_THREADS + tid;
                valid[j] = e < total;
                key[j] = 0; hint[j] = 0; dig[j] = 0;
                if (valid[j]) {
                    uint32_t sg = L.tile_seg;
                    while (e >= L.seg_prefix[sg + 1]) sg++;  // a tile spans a few segments at most; empty ones are skipped
                    const uint64_t at = ((uint64_t)bucket * PT_SEGMENTS + sg) * seg_cap1 + (e - L.seg_prefix[sg]);
                    key[j] = in_keys[at];
                    hint[j] = in_hints[at];
                    dig[j] = mulhi32(bin32_of(key[j], mm_k), np1 * m2) - bucket * m2;
                }
            }
            scatter_tile(L, key, hint, dig, valid, n_buckets, nullptr, cap2, (uint64_t)n_pieces * cap2, out_keys, out_hints,
                         ((uint64_t)bucket * n_buckets * n_pieces + piece) * cap2, sp);
        }
        __syncthreads();
        if (tid < n_buckets) leaf_counts[((uint64_t)bucket * n_buckets + tid) * n_pieces + piece] = min(L.wcur[tid], (uint32_t)cap2);
    }
}


// =============================================================================================
// Super-k-mer form of the pipeline (packed keys, k >= SK_MIN_K: TableView::mm_k != 0).
//
// Regions of the table are minimizer bins (kmer_device.h), so all windows of a read that share their
// minimizer go to the same region, and consecutive windows mostly do.  P1 therefore emits one 16-byte
// RECORD per run of up to SK_MAX_WINDOWS such windows -- the run's bases plus, where the read has
// them -- instead of one 12-byte (key, read pointer) record per
// window: about 9 windows per record at k = 31, so the streams through P1/P2/P3 shrink ~7-fold.  P3
// expands the records back into keys and hints right before it counts them in LDS.
//
// Record, as the 128-bit number hi:lo --
//   hi bits 63..60  windows - 1
//   hi bits 59..0   the first 30 bases of the run (first base on top)
//   lo bits 63..32  bases 30 .. 45 (windows + k - 1 <= 16 + 30 bases in all; unused tail zero)
//   lo bits 31..0   the record's bin word (sk_bin of its minimizer): P2 and, after the table grew, P3 take their
//                   bucket digits from it
// A second stream carries the read pointer (kmer_device.h ptr_encode) of the record's first window, or 0.
constexpr uint32_t SK_MAX_WINDOWS = 16;  // 16 + (31 - 1) = 46 bases

__device__ __forceinline__ uint32_t sk_windows(uint64_t hi) { return (uint32_t)(hi >> 60) + 1u; }

// window j of a record -> its canonical key (j < sk_windows(hi), k <= 31)
__device__ __forceinline__ uint64_t sk_window_key(uint64_t lo, uint64_t hi, uint32_t j, int k)
{
    // T = the 46 bases top-aligned in 128 bits
    const uint64_t t_hi = (hi << 4) | (lo >> 60), t_lo = (lo >> 32) << 36;
    const uint32_t sh = 2 * j;  // <= 30
    const uint64_t fw = ((t_hi << sh) | ((t_lo >> 1) >> (63 - sh))) >> (64 - 2 * k), rc = rc_packed(fw, k);
    return rc < fw ? rc : fw;
}

struct SkSpill {
    uint4 *recs;
    unsigned long long *count;
    uint64_t cap;
    uint32_t *lost;
};

__device__ __forceinline__ void sk_spill_push(const SkSpill &sp, const uint4 &rec)
{
    const unsigned long long i = atomicAdd(sp.count, 1ull);
    if (i < sp.cap) sp.recs[i] = rec; else atomicExch(sp.lost, 1u);
}

// append one record to this workgroup's piece of bucket d (fill levels in LDS: wcur = before this
// tile, cnt = within it); runs of a bucket are short here, so the store is not staged through LDS
struct SkCursors {
    uint32_t cnt[PT_MAX_LEAVES2], wcur[PT_MAX_LEAVES2];
};
__device__ __forceinline__ void sk_emit(SkCursors &C, uint32_t d, const uint4 &rec, uint32_t bin, uint64_t cap, uint64_t base,
                                        uint64_t bucket_stride, uint4 *out_recs, uint32_t *out_bins, const SkSpill &sp)
{
    const uint64_t dst = (uint64_t)C.wcur[d] + atomicAdd(&C.cnt[d], 1u);
    if (dst < cap) {
        const uint64_t at = base + (uint64_t)d * bucket_stride + dst;
        out_recs[at] = rec;
        out_bins[at] = bin;
    } else {
        sk_spill_push(sp, rec);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// SK-P1, wave-autonomous form (the one the host launches).  The kernel above synchronises 16 waves six times per tile
// of 8192 positions and ran at 4 waves per SIMD with half its wave-cycles parked on barriers; here a WAVE owns a tile
// of P1W_TILE = 62 x 8 positions and nothing but the bucket fill levels is shared by the workgroup:
//   * a lane loads the three 64-bit words that hold bases p0-7 .. p0+59 around its 8 positions straight from global
//     memory (the next tile's words, read offsets and first read are requested one tile ahead);
//   * every 15-mer is hashed once: a lane hashes the 8 that start at its positions and takes the 16 that follow from the
//     next two lanes (lanes 62 and 63 own no window: they only supply those hashes, so a tile needs nothing from outside
//     its wave);
//   * read starts and run breaks live in a 176-byte piece of LDS per wave, ordered by wave-level fences only;
//   * a record goes out with one LDS atomic on the workgroup's fill level of its bucket and two scattered stores
//     (the record, and the read pointer of its first window: base ptr_base + position in the read store; ptr_base
//     == ~0: the context keeps no read store and the pointers are 0).
// Segments = workgroups = P1W_SEGMENTS.
#ifndef MC_P1W_MIN_WAVES
#define MC_P1W_MIN_WAVES 4   // waves per SIMD the register allocator must leave room for (tuning builds override it)
#endif
#ifndef MC_P1W_PREFETCH_WORDS
#define MC_P1W_PREFETCH_WORDS 1
#endif
#ifndef MC_P1W_COMPACT
#define MC_P1W_COMPACT 1   // records are built one per lane from a queue of their starts (0: every lane loops over its own starts)
#endif
#ifndef MC_P1W_THREADS
#define MC_P1W_THREADS 1024   // (16 waves: one workgroup a CU, 256 x 512 bucket lines open at a time -- 4.3 ms and 5.1 GB written on configs[1] against 4.5 ms and 5.8 GB with 512)
#endif
constexpr int P1W_THREADS = MC_P1W_THREADS;
constexpr int P1W_WAVES = P1W_THREADS / 64;
constexpr uint32_t P1W_TILE = 62 * PT_ITEMS;   // base positions per wave tile
constexpr int P1W_SEGMENTS = 1024;             // workgroups of the launch = segments of every level-1 bucket
static_assert(PT_MAX_LEAVES2 <= PT_THREADS && PT_MAX_BUCKETS_KEYS <= PT_THREADS, "one thread per leaf cursor");
static_assert(P1W_SEGMENTS <= PT_THREADS && P1W_SEGMENTS >= PT_SEGMENTS, "k_sk2_scatter scans one segment count per thread");

// inclusive prefix sum over the wave's lanes (row_shr 1, 2, 4, 8 inside a row of sixteen, then the rows' totals: row_bcast 15 / 31)
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

struct alignas(16) Sk1wLds {
    uint32_t wcur[PT_MAX_BUCKETS1_SK];  // this workgroup's fill level of every bucket
    uint32_t starts[P1W_WAVES][24];    // per wave: bit b <-> "a read starts at position lo - 64 + b" (704 bits used)
    uint32_t brk[P1W_WAVES][20];       // per wave, as bytes: byte 2 + lane = that lane's 8 break bits; bytes 0,1 = 0; bytes 66.. = 0xFF
#if MC_P1W_COMPACT
    uint64_t wst[P1W_WAVES][24];       // per wave: the tile's words (word 0 holds base lo - 7)
    uint32_t hst[P1W_WAVES][64 * PT_ITEMS];  // per wave: minimizer hash of every window of the tile
    uint16_t squeue[P1W_WAVES][64 * PT_ITEMS];  // per wave: the windows that start a record, in order
#endif
};

__device__ __forceinline__ uint32_t wave_from_next(uint32_t x)
{  // lane i <- lane i + 1 (lane 63: unspecified)
    return __shfl_down(x, 1);
}

// 64 bits starting at bit 2*q of the 192-bit string W0:W1:W2 (q <= 64); bits past W2 read as zero
__device__ __forceinline__ uint64_t p1w_bits(uint64_t W0, uint64_t W1, uint64_t W2, uint32_t q)
{
    const bool up = q >= 32;
    const uint64_t a = up ? W1 : W0, b = up ? W2 : W1;
    const uint32_t sh = 2 * (q & 31);
    return (a << sh) | ((b >> 1) >> (63 - sh));
}

// COMPACT: a record leaves as ONE 16-byte unit whose first word holds, instead of the bin word, what the second level needs of
// it -- the leaf inside the level-1 bucket (10 bits, `m2` leaves per bucket) -- and the position of its first window RELATIVE to
// the first base of the workgroup's tiles (22 bits): a workgroup then takes `chunk_tiles` CONSECUTIVE tiles (chunk_tiles x
// P1W_TILE <= 2^22), and the reader (k_sk2_scatter_compact) knows the segment = workgroup of every record.  out_ptrs is not
// written.  The 16-byte and 4-byte stores of the two-array form reach HBM as two partly filled sectors per record (the lines
// of 512 buckets x 1024 segments do not live in the L2 until they are full): 9.2 GB written for 2.7 GB on configs[1].
constexpr uint32_t SKC_REL_BITS = 22;
// BINNED (multi-GPU, with OWNERS): the records of owner o are dealt to `sub` buckets by the top of their bin word -- bucket
// o * sub + mulhi32(bin, sub), np1 * sub buckets in all -- so that a second pass (k_sk2_scatter_staged<.., false, true>) can put
// every owner's records in the order of the `fine` level-1 buckets of the OWNER's counting run (fine = a multiple of sub) and the
// owner needs no first level of its own.  That pass writes to exact places: this kernel counts, besides, the records of every
// (owner, fine bucket) cell and the windows of every owner -- in LDS, a row of np1 * fine counters a workgroup (fine_rows),
// summed by k_skb_offsets.
constexpr uint32_t SKB_MAX_CELLS = 16384;  // owners x fine buckets at most (64 KB of LDS beside the kernel's 62)
template <bool ON> struct Sk1wFine { uint32_t h[ON ? SKB_MAX_CELLS : 1]; uint32_t win[ON ? PT_MAX_BUCKETS : 1]; };
struct SkBinned { uint32_t sub, fine; uint32_t *fine_rows; unsigned long long *owner_windows; };
template <bool OWNERS, bool COMPACT = false, bool BINNED = false>
__global__ void __launch_bounds__(P1W_THREADS, MC_P1W_MIN_WAVES) k_sk1w_extract(
    const uint64_t *__restrict__ words, const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t base_lo,
    uint64_t n_bases, uint64_t n_tiles, const uint32_t *__restrict__ first_read, int k, uint32_t np1, uint32_t *seg_counts,
    uint64_t cap, uint4 *out_recs, uint32_t *out_ptrs, SkSpill sp, uint64_t ptr_base, uint32_t chunk_tiles = 0, uint32_t m2 = 0,
    SkBinned bn = SkBinned{1, 1, nullptr, nullptr})
{
    static_assert(!(OWNERS && COMPACT), "the compact form is for the single-GPU pipeline");
    static_assert(!BINNED || OWNERS, "fine buckets are the owners' business");
    __shared__ Sk1wLds L;
    __shared__ Sk1wFine<BINNED> F;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    for (uint32_t i = tid; i < PT_MAX_BUCKETS1_SK; i += P1W_THREADS) L.wcur[i] = 0;
    if (BINNED) {
        for (uint32_t i = tid; i < np1 * bn.fine; i += P1W_THREADS) F.h[i] = 0;
        for (uint32_t i = tid; i < np1; i += P1W_THREADS) F.win[i] = 0;
    }
    // bucket of a record (and, BINNED, its cell and its owner's windows counted)
    auto bucket_of = [&](uint32_t hsel, uint32_t bin, uint32_t n_win) -> uint32_t {
        if (BINNED) {
            const uint32_t o = sk_owner(hsel, np1);
            atomicAdd(&F.h[o * bn.fine + mulhi32(bin, bn.fine)], 1u);
            atomicAdd(&F.win[o], n_win);
            return o * bn.sub + mulhi32(bin, bn.sub);
        }
        return OWNERS ? sk_owner(hsel, np1) : mulhi32(bin, np1);
    };
    uint32_t *starts = L.starts[wv];
    uint32_t *brkw = L.brk[wv];
    uint8_t *brkb = reinterpret_cast<uint8_t *>(brkw);
#if MC_P1W_COMPACT
    uint64_t *wst = L.wst[wv];
    uint32_t *hst = L.hst[wv];
    uint16_t *squeue = L.squeue[wv];
    if (lane < 24) wst[lane] = 0;
#endif
    if (lane < 20) brkw[lane] = lane == 0 ? 0u : 0xFFFFFFFFu;  // (bytes 2 .. 65 are rewritten by every tile)
    __syncthreads();
    const int w = k - SK_M + 1;                       // SK_M-mers per window (9 .. 17)
    const uint64_t last_word = (n_bases + 31) / 32;   // the pad word
    const uint64_t seg_base = (uint64_t)blockIdx.x * cap, bucket_stride = (uint64_t)gridDim.x * cap;
    const uint64_t wave_id = (uint64_t)blockIdx.x * P1W_WAVES + wv, n_waves = (uint64_t)gridDim.x * P1W_WAVES;

    uint64_t pfW0 = 0, pfW1 = 0, pfW2 = 0, pf_off = ~0ull;
    uint32_t pf_first = 0;
    auto load_words = [&](uint64_t tile) {
        const int64_t p0 = (int64_t)(tile * P1W_TILE) + (int64_t)lane * PT_ITEMS;
        const int64_t wi0 = (p0 - 7) >> 5;  // (-1 for the first lane of tile 0: that word reads as zero)
        pfW0 = wi0 >= 0 ? words[min((uint64_t)wi0, last_word)] : 0ull;
        pfW1 = words[min((uint64_t)(wi0 + 1), last_word)];
        pfW2 = words[min((uint64_t)(wi0 + 2), last_word)];
    };
    auto prefetch = [&](uint64_t tile) {
        if (tile >= n_tiles) return;
        if (MC_P1W_PREFETCH_WORDS) load_words(tile);
        pf_first = first_read[tile];
        const uint64_t r = (uint64_t)pf_first + lane;
        pf_off = r < n_reads ? offsets[r] : ~0ull;
    };
    // tiles of this wave: every n_waves-th of the launch, or (COMPACT) every P1W_WAVES-th of the workgroup's own stretch
    const uint64_t chunk_lo = base_lo / P1W_TILE + (uint64_t)blockIdx.x * chunk_tiles;
    const uint64_t tile_end = COMPACT ? min(n_tiles, chunk_lo + chunk_tiles) : n_tiles;
    const uint64_t tile0 = COMPACT ? chunk_lo + wv : base_lo / P1W_TILE + wave_id;
    const uint64_t tile_step = COMPACT ? (uint64_t)P1W_WAVES : n_waves;
    prefetch(tile0);
    for (uint64_t tile = tile0; tile < tile_end; tile += tile_step) {
        const uint64_t lo = tile * (uint64_t)P1W_TILE;
        const int64_t bm_lo = (int64_t)lo - 64;
        const uint64_t bm_hi = lo + 640;
        if (!MC_P1W_PREFETCH_WORDS) load_words(tile);
        const uint64_t W0 = pfW0, W1 = pfW1, W2 = pfW2;
        uint64_t s = pf_off;
        const uint32_t my_first = pf_first;
        prefetch(tile + tile_step);
        const uint64_t p0 = lo + (uint64_t)lane * PT_ITEMS;
        const uint32_t off0 = (uint32_t)((int64_t)p0 - 32 * (((int64_t)p0 - 7) >> 5));  // base p0 inside W0:W1:W2 (8, 16, 24 or 32)

        // ---- read starts of this tile's neighbourhood
        if (lane < 24) starts[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint64_t r = (uint64_t)my_first + lane;; r += 64) {
            const bool in = r < n_reads && s < bm_hi;
            if (in) {
                const int64_t rel = (int64_t)s - bm_lo;
                if (rel >= 0) atomicOr(&starts[(uint32_t)rel >> 5], 1u << ((uint32_t)rel & 31));
            }
            if (__ballot(!in)) break;  // offsets ascend: the first lane past the range ends the walk
            s = r + 64 < n_reads ? offsets[r + 64] : ~0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t ws[4];  // bit b of ws[0..3] <-> a read starts at position p0 - 8 + b
        {
            const uint32_t bit0 = lane * 8 + 56, wd = bit0 >> 5, sh = bit0 & 31;
            const uint32_t d0 = starts[wd], d1 = starts[wd + 1], d2 = starts[wd + 2], d3 = starts[wd + 3], d4 = starts[wd + 4];
            ws[0] = __builtin_amdgcn_alignbit(d1, d0, sh);
            ws[1] = __builtin_amdgcn_alignbit(d2, d1, sh);
            ws[2] = __builtin_amdgcn_alignbit(d3, d2, sh);
            ws[3] = __builtin_amdgcn_alignbit(d4, d3, sh);
        }

        // ---- sk_order of the canonical SK_M-mers at my 8 positions, then of the 16 after them
        uint32_t hh[24];
        {
            const uint64_t A = p1w_bits(W0, W1, W2, off0);
            uint32_t f = (uint32_t)(A >> (64 - 2 * SK_M)), r = sk_rc_mmer(f);
            hh[0] = sk_order(f < r ? f : r);
#pragma unroll
            for (int i = 1; i < 8; i++) {
                const uint32_t nb = (uint32_t)(A >> (62 - 2 * (i + SK_M - 1))) & 3u;  // the base that enters
                f = ((f << 2) | nb) & SK_MMASK;
                r = (r >> 2) | ((3u - nb) << (2 * (SK_M - 1)));
                hh[i] = sk_order(f < r ? f : r);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) hh[8 + i] = wave_from_next(hh[i]);
#pragma unroll
            for (int i = 0; i < 8; i++) hh[16 + i] = wave_from_next(hh[8 + i]);
        }
        // minimizer hash of my 8 windows: window j = min hh[j .. j+w-1].  All eight ranges hold hh[7 .. w-1]; what differs
        // is a suffix of hh[0 .. 6] and a prefix of hh[w .. w+6].
        uint32_t hmin[PT_ITEMS];
#define SK_CASE(W)                                                                             \
    case W: {                                                                                  \
        uint32_t core = hh[7];                                                                 \
        _Pragma("unroll") for (int i = 8; i < W; i++) core = min(core, hh[i]);                 \
        uint32_t suf = SK_NONE;                                                                \
        hmin[7] = core;                                                                        \
        _Pragma("unroll") for (int j = 6; j >= 0; j--) { suf = min(suf, hh[j]); hmin[j] = min(suf, core); } \
        uint32_t pre = SK_NONE;                                                                \
        _Pragma("unroll") for (int j = 1; j < PT_ITEMS; j++) { pre = min(pre, hh[W + j - 1]); hmin[j] = min(hmin[j], pre); } \
    } break;
        switch (w) {
            SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
        default:
            SK_CASE(17)
        }
#undef SK_CASE

        // ---- which of my positions start a window, and where runs of equal minimizers break
        uint32_t valid_bits = 0;
        if (lane < 62 && p0 + (uint64_t)k <= n_bases) {
            // the window [p, p+k) lies inside one read iff no read starts at p+1 .. p+k-1: bit j of S = OR of the read-start bits
            // of p0 + j + 1 .. p0 + j + k - 1, for all eight windows at once (doubling shifts; round 3 tested the eight apart,
            // each behind 64-bit range checks: 90 instructions for what takes 30)
            uint64_t S = ((((uint64_t)ws[1]) << 32) | ws[0]) >> 9;  // bit b <-> a read starts at p0 + 1 + b
            uint32_t width = 1;
            for (; 2 * width <= (uint32_t)k - 1; width *= 2) S |= S >> width;  // (uniform: k - 1 is 22 .. 30, four rounds)
            S |= S >> ((uint32_t)k - 1 - width);
            const uint64_t jmax = n_bases - (uint64_t)k - p0;  // last j with p0 + j + k <= n_bases
            const uint32_t jm = jmax < 7 ? (uint32_t)jmax : 7u;
            const uint32_t j0 = p0 >= base_lo ? 0u : (base_lo - p0 < 8 ? (uint32_t)(base_lo - p0) : 8u);  // first j with p0 + j >= base_lo
            valid_bits = ~(uint32_t)S & ((2u << jm) - 1u) & ~((1u << j0) - 1u) & 0xFFu;
        }
        uint32_t brk_bits = 0;  // bit j: my window j breaks the run of equal minimizers (or is no window)
        {
            const uint32_t last = (valid_bits >> (PT_ITEMS - 1)) & 1u ? hmin[PT_ITEMS - 1] : SK_NONE;
            uint32_t prev = __shfl_up(last, 1);
            if (lane == 0) prev = SK_NONE;  // a tile always starts a run
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < PT_ITEMS; j++) {
                const bool v = (valid_bits >> j) & 1u;
                if (!v || prev == SK_NONE || prev != hmin[j]) bits |= 1u << j;
                prev = v ? hmin[j] : SK_NONE;
            }
            brkb[2 + lane] = (uint8_t)bits;  // window i of the tile <-> bit 16 + i of the wave's break bitmap
            brk_bits = bits;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // A window starts a record when it breaks the run of equal minimizers or sits a multiple of SK_MAX_WINDOWS behind the
        // run's first window.  Where did the run start?  The last lane below mine that holds a break is a ballot and a count of
        // leading zeros away, its break bits one cross-lane read; from there the run start is carried along my 8 windows.
        // (Round 3 looked the start up in the wave's bitmap only for windows of runs of 16 and more -- a branch with a loop over
        // LDS words in it, for each of the 8 windows apart.  One window in seven is that deep in its run, so every one of the
        // eight branches ran in nearly every tile with a handful of lanes in it: 330 of the kernel's 556 vector instructions
        // per tile.)
        uint32_t start_bits = 0;
        {
            const unsigned long long holders = __ballot(brk_bits != 0);  // (lane 0 is one: a tile always starts a run)
            const unsigned long long below = holders & ((1ull << lane) - 1ull);
            const uint32_t P = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
            const uint32_t bp = (uint32_t)__shfl((int)brk_bits, (int)P);
            uint32_t cur = P * PT_ITEMS + (31u - (uint32_t)__builtin_clz(bp | 1u));  // tile window of the last break before my first one (lane 0: unused)
#pragma unroll
            for (int j = 0; j < PT_ITEMS; j++) {
                const uint32_t pos = lane * PT_ITEMS + (uint32_t)j;
                cur = (brk_bits >> j) & 1u ? pos : cur;
                if (((valid_bits >> j) & 1u) && ((pos - cur) % SK_MAX_WINDOWS) == 0) start_bits |= 1u << j;
            }
        }

        // ---- records: a run is cut every SK_MAX_WINDOWS windows, counted from its first window
#if MC_P1W_COMPACT
        // A lane holds 0-4 record starts among its 8 positions, and a loop over them runs as long as the busiest lane's
        // (about 4 rounds of ~70 instructions for 0.9 records per lane).  Instead the tile's ~55 starts are lined up --
        // a lane writes the window numbers of its starts into the wave's queue at its prefix count -- and lane i builds
        // the i-th record of the tile from what the wave put in LDS for that: the tile's words, the minimizer hash of
        // every window, the break bitmap.
        {
            const uint32_t cnt = (uint32_t)__builtin_popcount(start_bits);
            const uint32_t incl = wave_incl_sum(cnt);  // (six DPP additions; as six cross-lane reads through the LDS port it was 24 instructions)
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t at = incl - cnt;
            for (uint32_t todo = start_bits; todo; todo &= todo - 1) squeue[at++] = (uint16_t)(lane * PT_ITEMS + (uint32_t)__builtin_ctz(todo));
            const uint32_t r7 = (uint32_t)(((int64_t)lo - 7) & 31);             // (lo - 7) = 32 * wbase + r7
            const uint32_t idx0 = (r7 + lane * PT_ITEMS) >> 5;                  // my W0 is word wbase + idx0
            wst[idx0] = W0; wst[idx0 + 1] = W1; wst[idx0 + 2] = W2;              // (lanes that hold the same word write the same value)
            *reinterpret_cast<uint4 *>(&hst[lane * PT_ITEMS]) = make_uint4(hmin[0], hmin[1], hmin[2], hmin[3]);
            *reinterpret_cast<uint4 *>(&hst[lane * PT_ITEMS + 4]) = make_uint4(hmin[4], hmin[5], hmin[6], hmin[7]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < total; i += 64) {
                const uint32_t wi = squeue[i];                                   // window number in the tile
                const uint32_t b0 = 17u + wi;                                    // break bits of the 15 windows behind it
                const uint32_t ahead = __builtin_amdgcn_alignbit(brkw[(b0 >> 5) + 1], brkw[b0 >> 5], b0 & 31u) & 0x7FFFu;
                const uint32_t n = ahead ? (uint32_t)__builtin_ctz(ahead) + 1u : SK_MAX_WINDOWS;
                const uint32_t len = n + (uint32_t)k - 1;  // bases of the run (<= 46)
                // X0:X1 = the 64 bases from the run's first base on; the record keeps the first `len`
                const uint32_t q = wi + 7u + r7, wq = q >> 5, sh = 2u * (q & 31u);
                const uint64_t A = wst[wq], B = wst[wq + 1], C = wst[wq + 2];
                const uint64_t X0 = (A << sh) | ((B >> 1) >> (63u - sh)), X1 = (B << sh) | ((C >> 1) >> (63u - sh));
                uint64_t hi = X0 >> 4;                                                  // bases 0 .. 29
                uint32_t tail = (uint32_t)(((X0 & 0xFull) << 28) | (X1 >> 36));         // bases 30 .. 45
                if (len <= 30) { hi &= ~0ull << (60 - 2 * len); tail = 0; }
                else tail &= len >= 46 ? ~0u : ~0u << (2 * (46 - len));
                hi |= (uint64_t)(n - 1) << 60;
                const uint32_t hsel = hst[wi];
                const uint32_t bin = sk_bin(hsel);
                const uint32_t d = bucket_of(hsel, bin, n);
                uint4 rec;
                rec.x = bin; rec.y = tail; rec.z = (uint32_t)hi; rec.w = (uint32_t)(hi >> 32);
                const uint64_t dst = atomicAdd(&L.wcur[d], 1u);
                if (dst < cap) {
                    const uint64_t o = seg_base + (uint64_t)d * bucket_stride + dst;
                    if (COMPACT) {
                        const uint32_t rel = (uint32_t)(tile - chunk_lo) * P1W_TILE + wi;
                        rec.x = rel | ((mulhi32(bin, np1 * m2) - d * m2) << SKC_REL_BITS);
                        out_recs[o] = rec;
                    } else {
                        out_recs[o] = rec;
                        out_ptrs[o] = ptr_base == ~0ull ? 0u : ptr_encode(ptr_base + lo + wi);
                    }
                } else {
                    sk_spill_push(sp, rec);
                }
            }
            __builtin_amdgcn_wave_barrier();  // (the next tile rewrites the staging area)
        }
#else
        uint64_t bw;  // bit b <-> window lane*8 - 16 + b
        {
            const uint32_t wd = lane >> 2, sh = 8 * (lane & 3);
            const uint32_t d0 = brkw[wd], d1 = brkw[wd + 1], d2 = brkw[wd + 2];
            bw = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32) | __builtin_amdgcn_alignbit(d1, d0, sh);
        }
        for (uint32_t todo = start_bits; todo; todo &= todo - 1) {
            const uint32_t j = (uint32_t)__builtin_ctz(todo);
            const uint32_t ahead = (uint32_t)(bw >> (17 + j)) & 0x7FFFu;  // breaks among the next 15 windows
            const uint32_t n = ahead ? (uint32_t)__builtin_ctz(ahead) + 1u : SK_MAX_WINDOWS;
            const uint32_t len = n + (uint32_t)k - 1;  // bases of the run (<= 46)
            // X0:X1 = the 64 bases from the run's first base on; the record keeps the first `len`
            const uint32_t q = off0 + j;  // 8 .. 39
            const uint64_t X0 = p1w_bits(W0, W1, W2, q), X1 = p1w_bits(W1, W2, 0ull, q);
            uint64_t hi = X0 >> 4;                                                  // bases 0 .. 29
            uint32_t tail = (uint32_t)(((X0 & 0xFull) << 28) | (X1 >> 36));         // bases 30 .. 45
            if (len <= 30) { hi &= ~0ull << (60 - 2 * len); tail = 0; }
            else tail &= len >= 46 ? ~0u : ~0u << (2 * (46 - len));
            hi |= (uint64_t)(n - 1) << 60;
            uint32_t hsel = hmin[0];
#pragma unroll
            for (int qq = 1; qq < PT_ITEMS; qq++) hsel = j == (uint32_t)qq ? hmin[qq] : hsel;
            const uint32_t bin = sk_bin(hsel);
            const uint32_t d = bucket_of(hsel, bin, n);
            uint4 rec;
            rec.x = bin; rec.y = tail; rec.z = (uint32_t)hi; rec.w = (uint32_t)(hi >> 32);
            const uint64_t dst = atomicAdd(&L.wcur[d], 1u);
            if (dst < cap) {
                const uint64_t at = seg_base + (uint64_t)d * bucket_stride + dst;
                if (COMPACT) {
                    const uint32_t rel = (uint32_t)(tile - chunk_lo) * P1W_TILE + lane * PT_ITEMS + j;
                    rec.x = rel | ((mulhi32(bin, np1 * m2) - d * m2) << SKC_REL_BITS);
                    out_recs[at] = rec;
                } else {
                    out_recs[at] = rec;
                    out_ptrs[at] = ptr_base == ~0ull ? 0u : ptr_encode(ptr_base + p0 + j);
                }
            } else {
                sk_spill_push(sp, rec);
            }
        }
#endif
    }
    __syncthreads();
    const uint32_t n_buckets = BINNED ? np1 * bn.sub : np1;
    for (uint32_t d = tid; d < n_buckets; d += P1W_THREADS)  // how much of its segment of every bucket this workgroup filled
        seg_counts[(uint64_t)d * gridDim.x + blockIdx.x] = (uint32_t)min((uint64_t)L.wcur[d], cap);
    if (BINNED) {
        for (uint32_t i = tid; i < np1 * bn.fine; i += P1W_THREADS) bn.fine_rows[(uint64_t)blockIdx.x * (np1 * bn.fine) + i] = F.h[i];
        for (uint32_t i = tid; i < np1; i += P1W_THREADS) if (F.win[i]) atomicAdd(&bn.owner_windows[i], (unsigned long long)F.win[i]);
    }
}

// BINNED, after the first level.  k_skb_colsum: cell[i] (zeroed by the caller) = records of (owner, fine bucket) cell i over all
// workgroups' rows -- blockIdx.y takes a stretch of rows.  k_skb_offsets (one workgroup): cell_start[i] = records in the cells
// before i (owners ascending, fine buckets ascending inside an owner: the order of the packed stream), owner_off[o] = where owner
// o's records start, owner_off[n_owners] = all of them.
__global__ void __launch_bounds__(256) k_skb_colsum(const uint32_t *__restrict__ fine_rows, uint32_t n_rows, uint32_t n, uint32_t *cell)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t per = (n_rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(n_rows, r0 + per);
    uint32_t v = 0;
    for (uint32_t r = r0; r < r1; r++) v += fine_rows[(uint64_t)r * n + i];
    if (v) atomicAdd(&cell[i], v);
}
__global__ void __launch_bounds__(1024) k_skb_offsets(const uint32_t *__restrict__ cell, uint32_t n_owners, uint32_t fine,
                                                      unsigned long long *cell_start, unsigned long long *owner_off)
{
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x, n = n_owners * fine;
    const uint32_t per = (n + 1023) / 1024, lo = min(n, tid * per), hi = min(n, lo + per);
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += cell[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long acc = 0;
        for (uint32_t i = 0; i < 1024; i++) { const unsigned long long v = part[i]; part[i] = acc; acc += v; }
        owner_off[n_owners] = acc;
    }
    __syncthreads();
    unsigned long long acc = part[tid];
    for (uint32_t i = lo; i < hi; i++) {
        cell_start[i] = acc;
        if (i % fine == 0) owner_off[i / fine] = acc;
        acc += cell[i];
    }
}

// The receiving side of a binned exchange: part p (what one rank sent of one chunk: part_off[p] onwards in the receive buffer, in the
// order of `fine` buckets with part_counts[p * fine + f] records each) becomes r = fine / np1 segments of every level-1 bucket of this
// rank's counting run -- segment p * r + f % r of bucket f / r (mulhi32(bin, fine) / r = mulhi32(bin, np1) when np1 divides fine).
// One workgroup a part.
__global__ void __launch_bounds__(1024) k_skb_segments(const unsigned long long *__restrict__ part_off, const uint32_t *__restrict__ part_counts,
                                                       uint32_t fine, uint32_t r, uint32_t nseg, uint32_t *seg_counts, unsigned long long *seg_start,
                                                       unsigned long long *bad)
{   // *bad: parts whose counts do not add up to their length (the second level would read past them)
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const uint32_t *cnt = part_counts + (uint64_t)p * fine;
    const uint32_t per = (fine + 1023) / 1024, lo = min(fine, tid * per), hi = min(fine, lo + per);
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long acc = part_off[p];
        for (uint32_t i = 0; i < 1024; i++) { const unsigned long long v = part[i]; part[i] = acc; acc += v; }
        if (acc != part_off[p + 1]) atomicAdd(bad, 1ull);
    }
    __syncthreads();
    unsigned long long acc = part[tid];
    for (uint32_t f = lo; f < hi; f++) {
        const uint64_t at = (uint64_t)(f / r) * nseg + (uint64_t)p * r + f % r;
        seg_counts[at] = cnt[f];
        seg_start[at] = acc;
        acc += cnt[f];
    }
}

// Multi-GPU split: packs the (owner, segment) pieces k_sk1_extract<true> filled into one stream ordered by
// owner.  k_sk_pack_offsets (one workgroup): piece_off[i] = records before piece i = owner * PT_SEGMENTS + seg,
// owner_off[o] = records before owner o, owner_off[n_owners] = all.  k_sk_pack: one workgroup per piece.
__global__ void __launch_bounds__(1024) k_sk_pack_offsets(const uint32_t *__restrict__ seg_counts, uint32_t n_owners,
                                                          unsigned long long *piece_off, unsigned long long *owner_off, uint32_t nseg)
{
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x, n = n_owners * nseg;
    const uint32_t per = (n + 1023) / 1024, lo = tid * per, hi = min(n, lo + per);
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += seg_counts[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long acc = 0;
        for (uint32_t i = 0; i < 1024; i++) { const unsigned long long v = part[i]; part[i] = acc; acc += v; }
        owner_off[n_owners] = acc;
    }
    __syncthreads();
    unsigned long long acc = part[tid];
    for (uint32_t i = lo; i < hi; i++) {
        piece_off[i] = acc;
        if (i % nseg == 0) owner_off[i / nseg] = acc;
        acc += seg_counts[i];
    }
}
__global__ void __launch_bounds__(256) k_sk_pack(const uint4 *__restrict__ recs, const uint32_t *__restrict__ bins,
                                                 const uint32_t *__restrict__ seg_counts, uint64_t seg_cap,
                                                 const unsigned long long *__restrict__ piece_off, uint32_t n_pieces,
                                                 uint4 *out_recs, uint32_t *out_bins, uint64_t out_cap)
{
    for (uint32_t p = blockIdx.x; p < n_pieces; p += gridDim.x) {
        const uint32_t n = seg_counts[p];
        const uint64_t src = (uint64_t)p * seg_cap, dst = piece_off[p];
        if (dst + n > out_cap) continue;  // (the host reports the overflow from owner_off)
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            out_recs[dst + i] = recs[src + i];
            out_bins[dst + i] = bins[src + i];
        }
    }
}

// Distinct canonical k-mers among the records of level-1 bucket 0 (segments 0 .. gridDim.x - 1 of capacity seg_cap), counted
// into a scratch set of 64-bit keys (all ones = free; mask + 1 slots, linear probing): the sample that sizes the table of a
// context without a capacity hint (mcgpu.hip pipe_resize_by_sample).  An estimate is all that is asked: a key that finds
// its probe sequence longer than 64 slots is counted as new.
__global__ void __launch_bounds__(256) k_sk_sample_distinct(const uint4 *__restrict__ recs, const uint32_t *__restrict__ seg_counts, uint64_t seg_cap, int k,
                                                            uint64_t *set, uint64_t mask, unsigned long long *n_distinct)
{   // (the records' first word -- bin word or the compact form's -- is not looked at)
    const uint32_t n = min(seg_counts[blockIdx.x], (uint32_t)seg_cap);
    const uint4 *seg = recs + (uint64_t)blockIdx.x * seg_cap;
    unsigned long long n_new = 0;
    for (uint32_t r = threadIdx.x; r < n; r += 256) {
        const uint4 rec = seg[r];
        const uint64_t lo = ((uint64_t)rec.y << 32) | rec.x, hi = ((uint64_t)rec.w << 32) | rec.z;
        for (uint32_t j = 0; j < sk_windows(hi); j++) {
            const uint64_t key = sk_window_key(lo, hi, j, k);
            uint64_t s = fmix64(key) & mask;
            bool placed = false;
            for (int probe = 0; probe < 64 && !placed; probe++) {
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&set[s]), ~0ull, (unsigned long long)key);
                if (old == ~0ull) { n_new++; placed = true; }
                else if (old == key) placed = true;
                else s = (s + 1) & mask;
            }
            if (!placed) n_new++;
        }
    }
    wave_add_ull(n_distinct, n_new);
}

// windows of a stream of records
__global__ void k_sk_count_windows(const uint4 *__restrict__ recs, uint64_t n, unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long w = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w += sk_windows(((uint64_t)recs[i].w << 32) | recs[i].z);
    wave_add_ull(out, w);
}

// SK-P1 for a flat stream of records (+ bin words): the receiving side of a multi-GPU exchange.
__global__ void __launch_bounds__(PT_THREADS) k_sk1_records(const uint4 *__restrict__ in_recs, const uint32_t *__restrict__ in_bins,
                                                            uint64_t n, uint32_t np1, uint32_t *seg_counts, uint64_t cap,
                                                            uint4 *out_recs, uint32_t *out_bins, SkSpill sp)
{
    __shared__ uint32_t wcur[PT_MAX_BUCKETS1_SK];  // this workgroup's fill level of every bucket
    const uint32_t tid = threadIdx.x, n_buckets = np1;
    for (uint32_t i = tid; i < PT_MAX_BUCKETS1_SK; i += PT_THREADS) wcur[i] = 0;
    __syncthreads();
    const uint64_t n_tiles = (n + PT_TILE - 1) / PT_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int j = 0; j < PT_ITEMS; j++) {
            const uint64_t i = tile * PT_TILE + tid + (uint64_t)j * PT_THREADS;
            if (i < n) {
                const uint4 rec = in_recs[i];  // (its bin word rides in rec.x; in_bins / out_bins carry the read pointers)
                const uint32_t d = mulhi32(rec.x, np1);
                const uint64_t dst = atomicAdd(&wcur[d], 1u);
                if (dst < cap) {
                    const uint64_t at = (uint64_t)blockIdx.x * cap + (uint64_t)d * ((uint64_t)PT_SEGMENTS * cap) + dst;
                    out_recs[at] = rec;
                    out_bins[at] = in_bins[i];
                } else {
                    sk_spill_push(sp, rec);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t d = tid; d < n_buckets; d += PT_THREADS) seg_counts[(uint64_t)d * PT_SEGMENTS + blockIdx.x] = (uint32_t)min((uint64_t)wcur[d], cap);
}

// SK-P2: one workgroup per level-1 bucket; its segments are read as one stream and
// scattered by the next m2 bits of the bin word into the bucket's leaves.
struct Sk2Lds {
    SkCursors C;
    uint32_t seg_prefix[P1W_SEGMENTS + 1];  // (P1W_SEGMENTS >= PT_SEGMENTS)
    uint32_t wave_tot[PT_THREADS / 64];
    uint32_t tile_seg;
};
__global__ void __launch_bounds__(PT_THREADS) k_sk2_scatter(const uint4 *__restrict__ in_recs, const uint32_t *__restrict__ in_bins,
                                                            uint64_t seg_cap1, const uint32_t *__restrict__ seg_counts1,
                                                            uint32_t n_buckets1, uint32_t np1, uint32_t m2, uint32_t *leaf_counts,
                                                            uint64_t cap2, uint4 *out_recs, uint32_t *out_bins, SkSpill sp, uint32_t nseg_in = PT_SEGMENTS,
                                                            int bin_of = 0, uint32_t piece = 0, uint32_t n_pieces = 1)
{
    // piece / n_pieces: the batch travels in n_pieces pieces (the level-1 pass of one runs next to this pass of the
    // one before it); every leaf then has n_pieces segments of capacity cap2 and this launch fills segment `piece`.
    // bin_of == 0: super-k-mer records -- the bucket digits come from the bin word inside the record (rec.x) and the
    // second stream (in_bins / out_bins) is the records' read pointers; != 0: other 16-byte payloads (the entries of the
    // solid-table build) whose bin word IS the second stream.
    __shared__ Sk2Lds L;
    const uint32_t tid = threadIdx.x, n_buckets = m2;  // leaves per level-1 bucket
    constexpr uint32_t TILE2 = PT_THREADS * 4;
    for (uint32_t bucket = blockIdx.x; bucket < n_buckets1; bucket += gridDim.x) {
        __syncthreads();
        if (tid < PT_MAX_LEAVES2) { L.C.wcur[tid] = 0; L.C.cnt[tid] = 0; }
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
            uint4 rec[4];
            uint32_t bin[4];
            bool have[4];
            uint32_t sg = L.tile_seg;  // (a thread's records ascend: the walk carries on from the previous one's segment)
#pragma unroll
            for (int j = 0; j < 4; j++) {  // loads first, then the scattered stores
                const uint32_t e = first + (uint32_t)j * PT_THREADS + tid;
                have[j] = e < total;
                if (have[j]) {
                    while (e >= L.seg_prefix[sg + 1]) sg++;
                    const uint64_t at = ((uint64_t)bucket * nseg_in + sg) * seg_cap1 + (e - L.seg_prefix[sg]);
                    rec[j] = in_recs[at];
                    bin[j] = in_bins[at];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (have[j])
                    sk_emit(L.C, mulhi32(bin_of ? bin[j] : rec[j].x, np1 * m2) - bucket * m2, rec[j], bin[j], cap2,
                            ((uint64_t)bucket * n_buckets * n_pieces + piece) * cap2, (uint64_t)n_pieces * cap2, out_recs, out_bins, sp);
            __syncthreads();
            if (tid < n_buckets) { L.C.wcur[tid] += L.C.cnt[tid]; L.C.cnt[tid] = 0; }
        }
        __syncthreads();
        if (tid < n_buckets) leaf_counts[((uint64_t)bucket * n_buckets + tid) * n_pieces + piece] = min(L.C.wcur[tid], (uint32_t)cap2);
    }
}

// 16 bytes as a native vector: copies of a HIP uint4 whose fields are then touched one by one go through a 12-byte memcpy into
// a private slot that ends up in LDS (48 KB of the staged scatter kernels' LDS were that); these stay in registers
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u ld_v4u(const uint4 *p) { return *reinterpret_cast<const v4u *>(p); }
__device__ __forceinline__ void st_v4u(uint4 *p, v4u v) { *reinterpret_cast<v4u *>(p) = v; }
__device__ __forceinline__ uint4 as_uint4(v4u v) { return make_uint4(v.x, v.y, v.z, v.w); }

// SK-P2 with the tile's records put in leaf order in LDS before they go out: a tile of 1024 x ITEMS records holds only
// a handful per leaf, and written one by one as they come (k_sk2_scatter) every 16-byte record and 4-byte pointer
// dirties a memory line of its own at a time of its own -- 7.0 GB reached HBM for 2.7 GB of records.  Here the records
// of one leaf leave together, from consecutive lanes to consecutive addresses, and the next tile continues where this
// one stopped.  Counting pipeline only (records + read pointers; digits from the record's bin word).
#ifndef MC_SK2_ITEMS
#define MC_SK2_ITEMS 4   // records per thread and tile: 4096-record tiles, 104 KB of LDS, one workgroup per CU (2: 2.6 ms, 4: 2.3, 5: 2.3, 6: 2.5; unstaged 2.9-3.0)
#endif
// IN_LISTED (the receiving side of a binned exchange, mcgpu.hip mc_add_superkmers_binned_dev): segment sg of a bucket starts at record
// in_start[bucket * nseg_in + sg] of in_recs / in_ptrs -- the pieces of what the other ranks sent, each in the order of the fine
// buckets -- instead of at (bucket * nseg_in + sg) * seg_cap1; and what leaves is the compact form's unit, the read pointer in the
// record's first word (the bin word has served once the leaf is known) -- out_ptrs is not written.
// OUT_LISTED (the sending side, mc_extract_superkmers_binned_dev): the buckets are (owner, coarse bucket) pairs -- bucket = owner *
// np1 + coarse --, the leaves the m2 fine buckets inside a coarse one, and leaf dd of a bucket goes to out_start[bucket * m2 + dd]
// onwards, an exact place (k_skb_offsets): no capacity, nothing spilled, no fill levels written.
template <int ITEMS, bool IN_LISTED, bool OUT_LISTED>
struct Sk2sLds {
    uint4 rec[PT_THREADS * ITEMS];
    uint32_t ptr[IN_LISTED ? 1 : PT_THREADS * ITEMS];
    uint16_t leaf[PT_THREADS * ITEMS];
    uint32_t cnt[PT_MAX_LEAVES2], wcur[PT_MAX_LEAVES2], off[PT_MAX_LEAVES2];
    uint32_t seg_prefix[P1W_SEGMENTS + 1];
    uint32_t wave_tot[PT_THREADS / 64];
    uint32_t tile_seg;
    unsigned long long seg_at[IN_LISTED ? P1W_SEGMENTS : 1];
    unsigned long long out_at[OUT_LISTED ? PT_MAX_LEAVES2 : 1];
};
template <int ITEMS, bool IN_LISTED = false, bool OUT_LISTED = false>
__global__ void __launch_bounds__(PT_THREADS) k_sk2_scatter_staged(const uint4 *__restrict__ in_recs, const uint32_t *__restrict__ in_ptrs,
                                                                   uint64_t seg_cap1, const uint32_t *__restrict__ seg_counts1,
                                                                   uint32_t n_buckets1, uint32_t np1, uint32_t m2, uint32_t *leaf_counts,
                                                                   uint64_t cap2, uint4 *out_recs, uint32_t *out_ptrs, SkSpill sp, uint32_t nseg_in,
                                                                   const unsigned long long *__restrict__ in_start = nullptr,
                                                                   const unsigned long long *__restrict__ out_start = nullptr)
{
    __shared__ Sk2sLds<ITEMS, IN_LISTED, OUT_LISTED> L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    constexpr uint32_t TILE = PT_THREADS * ITEMS;
    auto block_excl = [&](uint32_t c, uint32_t *total) -> uint32_t {  // exclusive scan of one value per thread
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if ((int)lane >= o) x += y;
        }
        __syncthreads();  // (wave_tot may still be read from the scan before)
        if (lane == 63u) L.wave_tot[wv] = x;
        __syncthreads();
        uint32_t before = 0, tot = 0;
        for (uint32_t i = 0; i < PT_THREADS / 64; i++) {
            const uint32_t t = L.wave_tot[i];
            if (i < wv) before += t;
            tot += t;
        }
        *total = tot;
        return before + x - c;
    };
    for (uint32_t bucket = blockIdx.x; bucket < n_buckets1; bucket += gridDim.x) {
        __syncthreads();
        if (tid < PT_MAX_LEAVES2) { L.wcur[tid] = 0; L.cnt[tid] = 0; }
        uint32_t total;
        {
            const uint32_t c = tid < nseg_in ? seg_counts1[(uint64_t)bucket * nseg_in + tid] : 0u;
            const uint32_t ex = block_excl(c, &total);
            if (tid < nseg_in) L.seg_prefix[tid] = ex;
            if (tid == 0) L.seg_prefix[nseg_in] = total;
            if (IN_LISTED) if (tid < nseg_in) L.seg_at[tid] = in_start[(uint64_t)bucket * nseg_in + tid];
            if (OUT_LISTED) if (tid < m2) L.out_at[tid] = out_start[(uint64_t)bucket * m2 + tid];
        }
        const uint32_t bdig = OUT_LISTED ? bucket % np1 : bucket;  // (the bucket's digit of the bin word)
        __syncthreads();
        for (uint32_t first = 0; first < total; first += TILE) {
            if (tid == 0) {  // segment of the tile's first record: largest sg with seg_prefix[sg] <= first
                uint32_t lo_s = 0, hi_s = nseg_in;
                while (hi_s - lo_s > 1) {
                    const uint32_t mid = (lo_s + hi_s) >> 1;
                    if (L.seg_prefix[mid] <= first) lo_s = mid; else hi_s = mid;
                }
                L.tile_seg = lo_s;
            }
            __syncthreads();
            v4u rec[ITEMS];
            uint32_t ptr[ITEMS], d[ITEMS], rank[ITEMS];
            bool have[ITEMS];
            uint32_t sg = L.tile_seg;
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const uint32_t e = first + (uint32_t)j * PT_THREADS + tid;
                have[j] = e < total;
                if (have[j]) {
                    while (e >= L.seg_prefix[sg + 1]) sg++;
                    const uint64_t at = (IN_LISTED ? (uint64_t)L.seg_at[sg] : ((uint64_t)bucket * nseg_in + sg) * seg_cap1) + (e - L.seg_prefix[sg]);
                    rec[j] = ld_v4u(&in_recs[at]);
                    ptr[j] = in_ptrs[at];
                }
            }
#pragma unroll
            for (int j = 0; j < ITEMS; j++)
                if (have[j]) {
                    d[j] = mulhi32(rec[j].x, np1 * m2) - bdig * m2;
                    if (IN_LISTED) rec[j].x = ptr[j];
                    rank[j] = atomicAdd(&L.cnt[d[j]], 1u);
                }
            __syncthreads();
            {   // where each leaf's run starts in the staging area
                uint32_t tot;
                const uint32_t ex = block_excl(tid < m2 ? L.cnt[tid] : 0u, &tot);
                if (tid < m2) L.off[tid] = ex;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; j++)
                if (have[j]) {
                    const uint32_t at = L.off[d[j]] + rank[j];
                    st_v4u(&L.rec[at], rec[j]);
                    if (!IN_LISTED) L.ptr[at] = ptr[j];
                    L.leaf[at] = (uint16_t)d[j];
                }
            __syncthreads();
            const uint32_t n_tile = min(TILE, total - first);
            for (uint32_t i = tid; i < n_tile; i += PT_THREADS) {
                const uint32_t dd = L.leaf[i];
                const uint64_t dst = (uint64_t)L.wcur[dd] + (i - L.off[dd]);
                if (OUT_LISTED) {
                    const uint64_t at = (uint64_t)L.out_at[dd] + dst;
                    if (at < cap2) {  // (here: the room of the caller's buffer; the host reports what did not fit)
                        st_v4u(&out_recs[at], ld_v4u(&L.rec[i]));
                        out_ptrs[at] = L.ptr[i];
                    }
                } else if (dst < cap2) {
                    const uint64_t at = ((uint64_t)bucket * m2 + dd) * cap2 + dst;
                    st_v4u(&out_recs[at], ld_v4u(&L.rec[i]));
                    if (!IN_LISTED) out_ptrs[at] = L.ptr[i];
                } else {
                    sk_spill_push(sp, L.rec[i]);
                }
            }
            __syncthreads();
            if (tid < m2) { L.wcur[tid] += L.cnt[tid]; L.cnt[tid] = 0; }
        }
        __syncthreads();
        if (!OUT_LISTED) if (tid < m2) leaf_counts[(uint64_t)bucket * m2 + tid] = min(L.wcur[tid], (uint32_t)cap2);
    }
}

// SK-P2 for k_sk1w_extract's COMPACT records (first word = position relative to the segment's first base | leaf in the bucket
// << 22): the same staging as k_sk2_scatter_staged, and what leaves is again ONE 16-byte unit per record -- first word = the
// read pointer of the record's first window, which only now, with the segment known, becomes absolute.  No pointer array on
// either side: 16 bytes in and 16 out per record against 20 and 20, and the staging area (72 KB) leaves room for two
// workgroups on a CU.  pos0 = position of segment 0's first base, seg_bases = bases per segment (chunk_tiles x P1W_TILE).
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): per-thread arrays indexed only by these stay in registers (a
// `#pragma unroll` loop with a data-dependent inner loop left them in scratch / LDS: 48 KB of the staged kernel's LDS were that)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef MC_SK2C_ITEMS
#define MC_SK2C_ITEMS 3   // records per thread and tile of k_sk2_scatter_compact: 3 = 72 KB of LDS and 64 registers, two workgroups on a CU (configs[1]: 1.61 ms; 4: 1.88, 5: 1.81)
#endif
template <int ITEMS, int RW = 1>
struct Sk2cLds {
    uint4 rec[PT_THREADS * ITEMS * RW];
    uint16_t leaf[PT_THREADS * ITEMS];
    uint32_t cnt[PT_MAX_LEAVES2], wcur[PT_MAX_LEAVES2], off[PT_MAX_LEAVES2];
    uint32_t seg_prefix[P1W_SEGMENTS + 1];
    uint32_t wave_tot[PT_THREADS / 64];
    uint32_t tile_seg;
};
// RW = 16-byte words per record: 1, or 2 for the long records of count_long.h (same first word; a spilled long record goes to
// the list two words a record)
template <int ITEMS, int RW = 1>
__global__ void __launch_bounds__(PT_THREADS, ITEMS * RW <= 3 ? 8 : 4) k_sk2_scatter_compact(const uint4 *__restrict__ in_recs, uint64_t seg_cap1,
                                                                       const uint32_t *__restrict__ seg_counts1, uint32_t n_buckets1, uint32_t m2,
                                                                       uint32_t *leaf_counts, uint64_t cap2, uint4 *out_recs, SkSpill sp,
                                                                       uint32_t nseg_in, uint64_t ptr_base, uint64_t pos0, uint64_t seg_bases, uint32_t leaf_shift = 0)
{   // leaf_shift (RW == 1): the records carry the leaf of a bucket of 2^10 leaves whatever the table -- the ten bits of the bin word
    // below the bucket's -- and this table, of a power of two of regions, has 2^(10 - leaf_shift) of them a bucket (a context without
    // a capacity hint: the table is sized between the two levels, mcgpu.hip pipe_resize_by_sample)
    __shared__ Sk2cLds<ITEMS, RW> L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    constexpr uint32_t TILE = PT_THREADS * ITEMS;
    auto block_excl = [&](uint32_t c, uint32_t *total) -> uint32_t {  // exclusive scan of one value per thread
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if ((int)lane >= o) x += y;
        }
        __syncthreads();  // (wave_tot may still be read from the scan before)
        if (lane == 63u) L.wave_tot[wv] = x;
        __syncthreads();
        uint32_t before = 0, tot = 0;
        for (uint32_t i = 0; i < PT_THREADS / 64; i++) {
            const uint32_t t = L.wave_tot[i];
            if (i < wv) before += t;
            tot += t;
        }
        *total = tot;
        return before + x - c;
    };
    for (uint32_t bucket = blockIdx.x; bucket < n_buckets1; bucket += gridDim.x) {
        __syncthreads();
        if (tid < PT_MAX_LEAVES2) { L.wcur[tid] = 0; L.cnt[tid] = 0; }
        uint32_t total;
        {
            const uint32_t c = tid < nseg_in ? seg_counts1[(uint64_t)bucket * nseg_in + tid] : 0u;
            const uint32_t ex = block_excl(c, &total);
            if (tid < nseg_in) L.seg_prefix[tid] = ex;
            if (tid == 0) L.seg_prefix[nseg_in] = total;
        }
        __syncthreads();
        for (uint32_t first = 0; first < total; first += TILE) {
            if (tid == 0) {  // segment of the tile's first record: largest sg with seg_prefix[sg] <= first
                uint32_t lo_s = 0, hi_s = nseg_in;
                while (hi_s - lo_s > 1) {
                    const uint32_t mid = (lo_s + hi_s) >> 1;
                    if (L.seg_prefix[mid] <= first) lo_s = mid; else hi_s = mid;
                }
                L.tile_seg = lo_s;
            }
            __syncthreads();
            v4u rec[ITEMS], rec2[ITEMS];
            // (dr: leaf << 16 | rank in the leaf's run of this tile, or NONE for a thread without a record -- one register an item where
            // leaf, rank, segment and a flag were four: at 64 registers a thread the kernel kept 16 of them in scratch memory)
            constexpr uint32_t NONE = 0xFFFFFFFFu;
            uint32_t dr[ITEMS];
            uint32_t sg = L.tile_seg;
            static_for<ITEMS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const uint32_t e = first + (uint32_t)j * PT_THREADS + tid;
                dr[j] = NONE;
                if (e < total) {
                    while (e >= L.seg_prefix[sg + 1]) sg++;
                    const uint64_t at = ((uint64_t)bucket * nseg_in + sg) * seg_cap1 + (e - L.seg_prefix[sg]);
                    rec[j] = ld_v4u(&in_recs[at * RW]);
                    if (RW == 2) rec2[j] = ld_v4u(&in_recs[at * RW + 1]);
                    dr[j] = sg;  // (the segment, until the record has arrived)
                }
            });
            static_for<ITEMS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if (dr[j] != NONE) {
                    // (long records: the bin word's top 24 bits sit in the second word above the window count -- the leaf is worked out here,
                    // for the table as it is now --, the position has the first word to itself)
                    const uint32_t dj = RW == 2 ? mulhi32(rec[j].y & 0xFFFFFF00u, n_buckets1 * m2) - bucket * m2 : (rec[j].x >> SKC_REL_BITS) >> leaf_shift;
                    if (RW == 2) rec[j].y &= 0xFFu;
                    const uint64_t pos = pos0 + (uint64_t)dr[j] * seg_bases + (RW == 2 ? rec[j].x : rec[j].x & ((1u << SKC_REL_BITS) - 1u));
                    rec[j].x = ptr_base == ~0ull ? 0u : ptr_encode(ptr_base + pos);
                    dr[j] = (dj << 16) | atomicAdd(&L.cnt[dj], 1u);  // (a tile holds TILE <= 5 * 1024 records: the rank fits 16 bits)
                }
            });
            __syncthreads();
            {   // where each leaf's run starts in the staging area
                uint32_t tot;
                const uint32_t ex = block_excl(tid < m2 ? L.cnt[tid] : 0u, &tot);
                if (tid < m2) L.off[tid] = ex;
            }
            __syncthreads();
            static_for<ITEMS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if (dr[j] != NONE) {
                    const uint32_t dj = dr[j] >> 16;
                    const uint32_t at = L.off[dj] + (dr[j] & 0xFFFFu);
                    st_v4u(&L.rec[at * RW], rec[j]);
                    if (RW == 2) st_v4u(&L.rec[at * RW + 1], rec2[j]);
                    L.leaf[at] = (uint16_t)dj;
                }
            });
            __syncthreads();
            const uint32_t n_tile = min(TILE, total - first);
            for (uint32_t i = tid; i < n_tile; i += PT_THREADS) {
                const uint32_t dd = L.leaf[i];
                const uint64_t dst = (uint64_t)L.wcur[dd] + (i - L.off[dd]);
                const v4u r = ld_v4u(&L.rec[i * RW]);
                if (RW == 1) {
                    if (dst < cap2) st_v4u(&out_recs[((uint64_t)bucket * m2 + dd) * cap2 + dst], r);
                    else sk_spill_push(sp, as_uint4(r));  // (the spill list's records go in without pointers and their first word is not looked at)
                } else {
                    const v4u r2 = ld_v4u(&L.rec[i * RW + 1]);
                    if (dst < cap2) {
                        const uint64_t o = (((uint64_t)bucket * m2 + dd) * cap2 + dst) * RW;
                        st_v4u(&out_recs[o], r);
                        st_v4u(&out_recs[o + 1], r2);
                    } else {
                        const unsigned long long si = atomicAdd(sp.count, 1ull);
                        if (si < sp.cap) { sp.recs[2 * si] = as_uint4(r); sp.recs[2 * si + 1] = as_uint4(r2); } else atomicExch(sp.lost, 1u);
                    }
                }
            }
            __syncthreads();
            if (tid < m2) { L.wcur[tid] += L.cnt[tid]; L.cnt[tid] = 0; }
        }
        __syncthreads();
        if (tid < m2) leaf_counts[(uint64_t)bucket * m2 + tid] = min(L.wcur[tid], (uint32_t)cap2);
    }
}

// drains the record spill list through the direct path (solid_thr / n_solid as in k_p3_merge)
__global__ void k_sk_add_records(const uint4 *__restrict__ recs, const uint32_t *__restrict__ ptrs, uint64_t n, int k, TableView t,
                                 uint32_t solid_thr, unsigned long long *n_solid)
{   // ptrs (may be null): the read pointer of each record's first window
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_new = 0, n_cross = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 rec = recs[i];
        const uint32_t p0 = ptrs ? ptrs[i] : 0u;
        const uint64_t lo = ((uint64_t)rec.y << 32) | rec.x, hi = ((uint64_t)rec.w << 32) | rec.z;
        for (uint32_t j = 0; j < sk_windows(hi); j++) {
            const uint64_t key = sk_window_key(lo, hi, j, k);
            uint32_t before;
            n_new += table_add(t, key, 1u, ptr_advance(p0, j), &before);
            n_cross += crosses(before, 1u, solid_thr);
        }
    }
    wave_add_ull(t.n_used, n_new);
    if (solid_thr) wave_add_ull(n_solid, n_cross);
}

// the records of the leaves [leaf_lo, leaf_hi) that k_p3_merge left unmerged, through the direct path (after the
// table has given up minimizer bins, mcgpu.hip to_hash_regions)
__global__ void k_sk_add_unmerged(const uint4 *__restrict__ leaf_recs, const uint32_t *__restrict__ leaf_ptrs,
                                  const uint32_t *__restrict__ seg_counts, uint64_t seg_cap,
                                  uint32_t nseg, uint32_t leaf_lo, uint32_t leaf_hi, const uint32_t *__restrict__ leaf_state, int k,
                                  TableView t)
{
    unsigned long long n_new = 0;
    for (uint32_t leaf = leaf_lo + blockIdx.x; leaf < leaf_hi; leaf += gridDim.x) {
        if (leaf_state[leaf]) continue;
        for (uint32_t sgm = 0; sgm < nseg; sgm++) {
            const uint64_t n = min((uint64_t)seg_counts[(uint64_t)leaf * nseg + sgm], seg_cap);
            const uint4 *recs = leaf_recs + ((uint64_t)leaf * nseg + sgm) * seg_cap;
            const uint32_t *ptrs = leaf_ptrs + ((uint64_t)leaf * nseg + sgm) * seg_cap;
            for (uint64_t r = threadIdx.x; r < n; r += blockDim.x) {
                const uint4 rec = recs[r];
                const uint32_t p0 = leaf_ptrs ? ptrs[r] : rec.x;  // (no array: the record's first word, k_sk2_scatter_compact)
                const uint64_t lo = ((uint64_t)rec.y << 32) | rec.x, hi = ((uint64_t)rec.w << 32) | rec.z;
                for (uint32_t j = 0; j < sk_windows(hi); j++) n_new += table_add(t, sk_window_key(lo, hi, j, k), 1u, ptr_advance(p0, j));
            }
        }
    }
    wave_add_ull(t.n_used, n_new);
}

// P3: one workgroup per leaf; a leaf covers 2^g consecutive table regions (g = 0 unless the table
// has more regions than leaves).  `virgin`: the table holds nothing yet and is not read.
// leaf_state[leaf]: 0 = to do, 1 = merged.  A leaf with a region that would overflow is left
// untouched and stays at 0 for the retry after the host enlarged the table.
// Counts are held to 2^30 when a region comes in and goes out; a launch adds < 2^30, so a counter never wraps.
// The read pointer of a slot (kmer_device.h) is written by ONE occurrence: the atomic add that counts an occurrence
// returns how many came before it, and the occurrence that finds ptr_from + r before it stores its own position --
// ptr_from = 0, or 1 when the coverage threshold is known to be above 1 (sequencing errors, most distinct k-mers, are
// seen once and never reach the BFS); r = a few bits of the key, below the threshold: the occurrences of neighbouring
// k-mers arrive in the same order (read by read), and if all took, say, their second one, the pointers along a stretch
// of the graph would all lead into the same read, which is no help once the walk has used that read up (ptr_pick).
// ptr_tries > 1 (records of other ranks carry no pointer): the next occurrences try too while the slot has none.
constexpr uint32_t P3_COUNT_CAP = 1u << 30;
// A region counts as full when an insertion has looked at this many slots (at the loads the host aims for probe chains
// stay below a few dozen); a region that really is full would otherwise cost 4096 probes per occurrence -- with no
// capacity hint that made a first, too small table 40 times slower than the run itself.
constexpr uint32_t P3_MAX_PROBES = REGION_SLOTS < TABLE_MAX_PROBES ? REGION_SLOTS : TABLE_MAX_PROBES;  // (the table's probing rule, kmer_device.h)
#ifndef MC_PTR_LATE
#define MC_PTR_LATE 0   // 1: a later occurrence (one of sixteen) replaces the early pointer (kmer_device.h ptr_pick_late): more different reads
                        // near a scout's tip, but + 0.6 ms in this kernel for 0.1 ms of walk on configs[1]
#endif

struct MergeLds {
    uint64_t key[REGION_SLOTS];
    uint32_t cnt[REGION_SLOTS];
    uint32_t aux[REGION_SLOTS];
    uint8_t dq[P3_THREADS / 64][64 * SK_MAX_WINDOWS];  // per wave: window -> lane holding its record (k_p3_merge<true>)
    uint32_t sq[P3_THREADS / 64][64];                  // per wave: lane -> read pointer of its record's first window
    uint32_t n_new, overflow;
    uint32_t emit_cur;  // fill level of this workgroup's segment of the solid list (P3Emit)
};

// The solid list: while a merged region goes back to HBM, its entries with count >= solid_thr are also appended,
// as (key, min(count, 32767), hint), to the workgroup's own segment of a list -- every batch rewrites every region, so
// after the last batch the list holds exactly what the BFS table is built from and the counting table need not be
// swept for it.  counts[wg] = the segment's fill level (kept across the retry launches of one batch).
struct P3Emit {
    uint4 *recs;       // nullptr: no list
    uint32_t *counts;  // one per workgroup of the launch
    uint64_t seg_cap;
    uint32_t *lost;    // set when a segment overflows: the list is then not used
};


// An occurrence that found no room within its stretch of the region held in LDS (the table's probing rule, kmer_device.h):
// it goes on to the region behind -- by the direct kernel, after this launch, from the list TableView::ovf (mcgpu.hip
// pipe_finish).  false: the list is full; the leaf is then not committed and merged again later, as before.
__device__ __forceinline__ bool ovf_push(const TableView &t, uint64_t key, uint32_t inc, uint32_t hint, uint32_t leaf)
{
    if (!t.ovf || !t.ovf_leaf) return false;
    const unsigned long long i = atomicAdd(t.ovf_n, 1ull);
    if (i >= t.ovf_cap) return false;
    t.ovf[i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), inc, hint);
    t.ovf_leaf[i] = leaf;  // (a leaf that fails after this -- the list ran full -- is merged again: its entries are then skipped)
    return true;
}

// Linear probing for `key` in an array of REGION_SLOTS 64-bit keys at LDS byte address `base`, from slot `home` on: the
// slot that holds the key when the loop ends, claimed with one LDS compare-and-swap per step when it was free (there is
// no load first: a CAS that finds another key is that load).  The compiler's loop spends ~20 scalar instructions a step
// on the bookkeeping of its exit conditions, and the slowest of 64 lanes decides the number of steps -- the scalar unit
// was the busiest part of the CU in the merge kernel; this one spends 6.  *n_new_wave += keys the WAVE inserted;
// *pending: lanes that found no slot in P3_MAX_PROBES steps (their key goes on to the next region: TableView::ovf).
__device__ __forceinline__ uint32_t lds_probe_claim(uint32_t base, uint32_t home, uint64_t key, uint32_t *n_new_wave, unsigned long long *pending,
                                                    uint32_t max_steps = P3_MAX_PROBES)
{   // max_steps: the caller has looked at P3_MAX_PROBES - max_steps slots in front of `home` itself
    uint32_t off = home * 8u, addr, cnt, t, it;
    unsigned long long old, sv, hit, pend;
    const unsigned long long empty = EMPTY_KEY;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b32 %[cnt], 0\n\t"
        "s_mov_b32 %[it], %[maxp]\n"
        "1:\n\t"
        "v_add_u32 %[addr], %[base], %[off]\n\t"
        "ds_cmpst_rtn_b64 %[old], %[addr], %[empty], %[key]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, %[old], %[empty]\n\t"
        "v_cmp_eq_u64 %[hit], %[old], %[key]\n\t"
        "s_bcnt1_i32_b64 %[t], vcc\n\t"
        "s_add_u32 %[cnt], %[cnt], %[t]\n\t"
        "s_or_b64 vcc, vcc, %[hit]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 2f\n\t"
        "v_add_u32 %[off], 8, %[off]\n\t"
        "v_and_b32 %[off], %[wrap], %[off]\n\t"
        "s_sub_u32 %[it], %[it], 1\n\t"
        "s_cmp_lg_u32 %[it], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_mov_b64 %[pend], exec\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [off] "+v"(off), [old] "=&v"(old), [addr] "=&v"(addr), [sv] "=&s"(sv), [hit] "=&s"(hit), [cnt] "=&s"(cnt), [t] "=&s"(t),
          [it] "=&s"(it), [pend] "=&s"(pend)
        : [base] "s"(base), [empty] "v"(empty), [key] "v"(key), [wrap] "s"((uint32_t)(REGION_SLOTS * 8u - 8u)), [maxp] "s"(max_steps)
        : "vcc", "scc", "memory");
    *n_new_wave += cnt;
    *pending = pend;
    return off >> 3;
}

// one occurrence of `key` into the region held in LDS; false when the region is full
__device__ __forceinline__ bool lds_region_add(MergeLds &L, uint64_t key, uint32_t hint, uint32_t home, uint32_t &my_new, uint32_t pick,
                                               uint32_t pick2)
{
    const uint32_t key_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint64_t *)L.key;
    uint32_t new_wave = 0;
    unsigned long long pending;
    const uint32_t s = lds_probe_claim(key_base, home, key, &new_wave, &pending);
    // (the probe loop counts the keys the WAVE inserted: the first of the lanes that came here books them)
    if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)__ballot(true)) - 1u) my_new += new_wave;
    if ((pending >> (threadIdx.x & 63u)) & 1ull) return false;
    const uint32_t seen = atomicAdd(&L.cnt[s], 1u);
    if (hint && (seen == pick || seen == pick2 || L.aux[s] == 0)) L.aux[s] = hint;  // (racy on purpose: any occurrence's pointer will do)
    return true;
}

// A leaf's records sit in `nseg` segments of capacity seg_cap: 1 after P2, PT_SEGMENTS when the table
// is so small that P1's buckets already are the leaves.
// SK: the streams hold super-k-mer records (leaf_keys = uint4 records, leaf_hints = their bin words, see below).
template <bool SK>
__global__ void __launch_bounds__(P3_THREADS) k_p3_merge(const void *__restrict__ leaf_keys,
                                                         const uint32_t *__restrict__ leaf_hints,
                                                         const uint32_t *__restrict__ seg_counts, uint64_t seg_cap,
                                                         uint32_t nseg, uint32_t n_leaves, uint32_t g, TableView t, int virgin,
                                                         uint32_t *leaf_state, uint32_t *leaf_new, uint32_t *any_failed,
                                                         uint32_t solid_thr, unsigned long long *n_solid, int k, P3Emit emit,
                                                         uint32_t ptr_tries, const uint32_t *lost = nullptr)
{
    // lost: the scatter levels' "records lost" flag -- the host enqueues this kernel behind them without looking; with
    // the flag set nothing may be merged (the batch is counted another way)
    if (lost && *lost) return;
    __shared__ MergeLds L;
    const uint32_t tid = threadIdx.x;
#ifdef MC_P3_TIMING
    unsigned long long tph[4] = {0, 0, 0, 0}, tl = __builtin_amdgcn_s_memrealtime(), n_lv = 0;
#define P3_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tph[i] += n_ - tl; tl = n_; } while (0)
#else
#define P3_STAMP(i) do {} while (0)
#endif
    const bool emitting = emit.recs != nullptr && solid_thr != 0;
    if (tid == 0) L.emit_cur = emitting ? emit.counts[blockIdx.x] : 0u;  // (published by the first barrier below)
    // solid_thr != 0: keep *n_solid = number of keys with count >= solid_thr up to date (the coverage
    // threshold the BFS will ask for, mc_set_coverage_hint), which saves the BFS set-up a table sweep
    long long solid_delta = 0;
    // A leaf starts with three dependent global reads (its state, its record count, its first records).  State and
    // count of the NEXT leaf are requested when this one starts, its first records once this one's are merged (their
    // addresses depend on the count: the waves share a leaf's records evenly), so that none of them costs waiting.
    const uint32_t wv = tid >> 6, lane = tid & 63u;
    constexpr uint32_t N_WAVES = P3_THREADS / 64;
    uint32_t st_nxt = blockIdx.x < n_leaves ? leaf_state[blockIdx.x] : 1u;
    uint32_t n0_nxt = 0;
    uint4 pre_nxt = make_uint4(0, 0, 0, 0);
    uint32_t pre_bin_nxt = 0;
    auto fetch_first = [&](uint32_t lf, uint32_t n0) {  // lane's record of the wave's first batch in leaf lf
        if constexpr (SK) {
            const uint32_t per = (n0 + N_WAVES - 1) / N_WAVES, r = wv * per + lane, end = (wv + 1) * per < n0 ? (wv + 1) * per : n0;
            if (r < end) {
                pre_nxt = static_cast<const uint4 *>(leaf_keys)[(uint64_t)lf * nseg * seg_cap + r];
                pre_bin_nxt = leaf_hints ? leaf_hints[(uint64_t)lf * nseg * seg_cap + r] : pre_nxt.x;  // (the record's read pointer; no array: the record's first word, k_sk2_scatter_compact)
            }
        }
    };
    if constexpr (SK) {
        if (blockIdx.x < n_leaves) {
            n0_nxt = min(seg_counts[(uint64_t)blockIdx.x * nseg], (uint32_t)seg_cap);
            fetch_first(blockIdx.x, n0_nxt);
        }
    }
    for (uint32_t leaf = blockIdx.x; leaf < n_leaves; leaf += gridDim.x) {
        const uint32_t st_cur = st_nxt, n0_cur = n0_nxt, pre_bin_cur = pre_bin_nxt;
        const uint4 pre_cur = pre_nxt;
        const uint32_t nl = leaf + gridDim.x;
        if (nl < n_leaves) {
            st_nxt = leaf_state[nl];
            if constexpr (SK) n0_nxt = min(seg_counts[(uint64_t)nl * nseg], (uint32_t)seg_cap);
        }
        bool first_fetched = false;
        if (st_cur) {  // uniform
            if (nl < n_leaves) fetch_first(nl, n0_nxt);
            continue;
        }

        // g == 0: the leaf is one region: merge and commit unless it overflows.  g > 0 (after the host
        // enlarged the table): a first sweep only checks that every sub-region fits, a second one commits,
        // so a leaf is never merged partially.
        bool leaf_ok = true;
        uint32_t new_total = 0;  // (thread 0) keys this leaf added to the table
        const int sweeps = g == 0 ? 1 : 2;
        for (int sweep = 0; sweep < sweeps && leaf_ok; sweep++) {
            const bool commit = g == 0 || sweep == 1;
            for (uint32_t sub = 0; sub < (1u << g); sub++) {
                const uint64_t region = ((uint64_t)leaf << g) | sub;
                Slot *gs = t.slots + region * REGION_SLOTS;
                int solid_before = 0;
                const uint4 pre = pre_cur;
                const uint32_t pre_bin = pre_bin_cur;
                for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                    if (virgin) {
                        L.key[i] = EMPTY_KEY; L.cnt[i] = 0; L.aux[i] = 0;
                    } else {
                        const uint4 raw = *reinterpret_cast<const uint4 *>(gs + i);
                        L.key[i] = ((uint64_t)raw.y << 32) | raw.x;
                        L.cnt[i] = min(raw.z, P3_COUNT_CAP);
                        L.aux[i] = raw.w;
                        solid_before += solid_thr && raw.z >= solid_thr;  // (empty slots hold count 0)
                    }
                }
                if (tid == 0) { L.n_new = 0; L.overflow = 0; }
                __syncthreads();
                P3_STAMP(0);
                uint32_t my_new = 0;
                for (uint32_t sgm = 0; sgm < nseg; sgm++) {
                const uint32_t n = (SK && sgm == 0) ? n0_cur : min(seg_counts[(uint64_t)leaf * nseg + sgm], (uint32_t)seg_cap);
                if constexpr (SK) {
                    // A wave takes 64 records at a time and spreads their WINDOWS over its lanes: records hold 1-16
                    // windows, and a lane that walked its own record left the wave waiting for the longest one (lane
                    // use ~40 %).  Each lane writes its lane number into the wave's queue once per window of its
                    // record; window w of the batch then belongs to the record of lane dq[w], whose words come over
                    // by ds_bpermute, and is cut out of the record directly (no rolling state).  The next batch's
                    // records are in flight meanwhile.
                    const uint4 *recs = static_cast<const uint4 *>(leaf_keys) + ((uint64_t)leaf * nseg + sgm) * seg_cap;
                    const uint32_t *ptrs = leaf_hints ? leaf_hints + ((uint64_t)leaf * nseg + sgm) * seg_cap : nullptr;
                    const uint32_t ptr_from = solid_thr >= 2 ? 1u : 0u, ptr_last = ptr_from + (MC_PTR_LATE ? 19 : 3) + ptr_tries - 1;
                    uint8_t *dq = L.dq[wv];
                    uint32_t *sq = L.sq[wv];
                    // the waves share the segment's records evenly (in batches of 64): the barrier behind the merge
                    // waits for the slowest wave
                    const uint32_t per = (n + N_WAVES - 1) / N_WAVES, r_lo = wv * per, r_hi = r_lo + per < n ? r_lo + per : n;
                    uint4 nxt = sgm == 0 ? pre : (r_lo + lane < r_hi ? recs[r_lo + lane] : make_uint4(0, 0, 0, 0));
                    uint32_t nxt_ptr = sgm == 0 ? pre_bin : (r_lo + lane < r_hi ? (ptrs ? ptrs[r_lo + lane] : nxt.x) : 0u);
                    for (uint32_t b0 = r_lo; b0 < r_hi; b0 += 64) {  // wave-uniform
                        if (L.overflow) break;  // (the leaf will not be committed: no point in merging the rest of it)
                        const uint32_t r = b0 + lane;
                        const uint4 rec = nxt;
                        const uint32_t rptr = nxt_ptr;
                        if (r + 64 < r_hi) {
                            nxt = recs[r + 64];
                            nxt_ptr = ptrs ? ptrs[r + 64] : nxt.x;
                        }
                        // (g != 0 with compact records -- the table grew under a run that had planned one region per leaf: the
                        // bin word is worked out again from the record's first window)
                        const bool mine = r < r_hi && !(g && mulhi32(ptrs ? rec.x : bin32_of(sk_window_key(((uint64_t)rec.y << 32) | rec.x, ((uint64_t)rec.w << 32) | rec.z, 0, k), k),
                                                                     t.n_regions) != region);
                        const uint32_t nw = mine ? sk_windows(((uint64_t)rec.w << 32) | rec.z) : 0u;
                        uint32_t incl = nw;  // inclusive scan of the window counts
#pragma unroll
                        for (uint32_t o = 1; o < 64; o <<= 1) {
                            const uint32_t y = __shfl_up(incl, o);
                            if (lane >= o) incl += y;
                        }
                        const uint32_t excl = incl - nw, total = __shfl(incl, 63);
                        for (uint32_t i = 0; i < nw; i++) dq[excl + i] = (uint8_t)lane;
                        sq[lane] = rptr;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        // the record words of chunk c + 1 are on their way while chunk c probes the region
                        uint32_t src_n = lane < total ? dq[lane] : lane;
                        uint32_t ny = __shfl(rec.y, src_n), nz = __shfl(rec.z, src_n), nw4 = __shfl(rec.w, src_n);
                        uint32_t ne = __shfl(excl, src_n);
                        for (uint32_t base = 0; base < total; base += 64) {
                            const uint32_t w = base + lane;
                            const bool act = w < total;
                            const uint32_t ry = ny, rz = nz, rw = nw4, src_c = src_n;
                            const uint32_t j = w - ne;
                            if (base + 64 < total) {
                                src_n = w + 64 < total ? dq[w + 64] : lane;
                                ny = __shfl(rec.y, src_n); nz = __shfl(rec.z, src_n); nw4 = __shfl(rec.w, src_n);
                                ne = __shfl(excl, src_n);
                            }
                            if (!act) continue;
                            // T = the record's 46 bases top-aligned in 128 bits (sk_window_key, with the bin word never moved)
                            const uint64_t hi = ((uint64_t)rw << 32) | rz;
                            const uint64_t t_hi = (hi << 4) | (ry >> 28), t_lo = (uint64_t)ry << 36;
                            const uint32_t sh = 2 * j;  // (j <= 15)
                            const uint64_t fw = ((t_hi << sh) | ((t_lo >> 1) >> (63 - sh))) >> (64 - 2 * k), rc = rc_packed(fw, k);
                            const uint64_t key = rc < fw ? rc : fw;
                            uint32_t s = sk_home(key);
                            bool done = false;
                            for (uint32_t probe = 0; probe < P3_MAX_PROBES; probe++) {
                                uint64_t cur = L.key[s];
                                if (cur == EMPTY_KEY) {
                                    cur = atomicCAS(reinterpret_cast<unsigned long long *>(&L.key[s]), (unsigned long long)EMPTY_KEY,
                                                    (unsigned long long)key);
                                    if (cur == EMPTY_KEY) { my_new++; cur = key; }
                                }
                                if (cur == key) { done = true; break; }
                                s = (s + 1) & (REGION_SLOTS - 1);
                            }
                            if (done) {
                                const uint32_t seen = atomicAdd(&L.cnt[s], 1u);
                                if (seen >= ptr_from && seen <= ptr_last) {
#if MC_PTR_LATE
                                    const uint32_t first = ptr_pick(key, ptr_from, solid_thr), late = ptr_pick_late(key, ptr_from);
#else
                                    const uint32_t first = ptr_pick(key, ptr_from, solid_thr), late = first;
#endif
                                    if ((seen >= first && seen < first + ptr_tries) || seen == late) {
                                        const uint32_t p0 = sq[src_c];
                                        if (p0 && (seen == first || seen == late || L.aux[s] == 0)) L.aux[s] = ptr_advance(p0, j);
                                    }
                                }
                            } else if (commit) {  // (a sweep that only checks, g > 0, leaves it to the one that commits)
                                if (!ovf_push(t, key, 1u, ptr_advance(sq[src_c], j), leaf)) atomicExch(&L.overflow, 1u);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();  // (the queue is rewritten by the next batch)
                    }
                } else {
                const uint64_t *keys = static_cast<const uint64_t *>(leaf_keys) + ((uint64_t)leaf * nseg + sgm) * seg_cap;
                const uint32_t *hints = leaf_hints + ((uint64_t)leaf * nseg + sgm) * seg_cap;
                for (uint32_t i0 = tid; i0 < n; i0 += 4 * P3_THREADS) {
                    if (L.overflow) break;  // (the leaf will not be committed)
                    uint64_t kk[4];
                    uint32_t hh[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {  // four independent loads in flight per thread
                        const uint32_t i = i0 + (uint32_t)u * P3_THREADS;
                        kk[u] = i < n ? keys[i] : EMPTY_KEY;  // (the streams never hold EMPTY_KEY)
                        hh[u] = i < n ? hints[i] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint64_t key = kk[u];
                        if (key == EMPTY_KEY) continue;
                        const uint64_t gslot = slot_of(t, key);
                        if (g && (gslot >> MC_REGION_LG) != region) continue;
                        const bool done = lds_region_add(L, key, hh[u], (uint32_t)gslot & (REGION_SLOTS - 1), my_new,
                                                         ptr_pick(key, solid_thr >= 2 ? 1u : 0u, solid_thr), ptr_pick_late(key, solid_thr >= 2 ? 1u : 0u));
                        if (!done && commit && !ovf_push(t, key, 1u, hh[u], leaf)) atomicExch(&L.overflow, 1u);
                    }
                }
                }
                }
                if (!first_fetched && nl < n_leaves) {  // the next leaf's first records: its count has long arrived
                    fetch_first(nl, n0_nxt);
                    first_fetched = true;
                }
                P3_STAMP(1);
                if (my_new) atomicAdd(&L.n_new, my_new);
                __syncthreads();
                P3_STAMP(2);
                const bool ovf = L.overflow != 0;
                if (ovf) leaf_ok = false;
                if (commit && !ovf) {
                    for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                        uint4 v;
                        const uint64_t kk = L.key[i];
                        v.x = (uint32_t)kk; v.y = (uint32_t)(kk >> 32);
                        v.z = min(L.cnt[i], P3_COUNT_CAP);  // counters stop at 2^30 here (anything above 32767 reads the same)
                        v.w = L.aux[i];
                        *reinterpret_cast<uint4 *>(gs + i) = v;
                        const bool solid = solid_thr && v.z >= solid_thr;
                        solid_delta += solid;
                        if (emitting) {  // (uniform; the loop's trip count is the same for every lane)
                            const unsigned long long m = __ballot(solid);
                            if (m) {
                                uint32_t base = 0;
                                const int leader = __ffsll((long long)m) - 1;
                                if ((int)(tid & 63u) == leader) base = atomicAdd(&L.emit_cur, (uint32_t)__popcll(m));
                                base = __shfl(base, leader);
                                if (solid) {
                                    const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << (tid & 63u)) - 1));
                                    if (pos < emit.seg_cap) {
                                        v.z = min(v.z, 32767u);
                                        emit.recs[(uint64_t)blockIdx.x * emit.seg_cap + pos] = v;
                                    } else {
                                        atomicExch(emit.lost, 1u);
                                    }
                                }
                            }
                        }
                    }
                    solid_delta -= solid_before;
                    if (tid == 0) new_total += L.n_new;
                } else if (commit && ovf && virgin) {  // (g == 0) nothing was there: leave a valid empty region behind
                    for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                        uint4 v; v.x = 0xFFFFFFFFu; v.y = 0xFFFFFFFFu; v.z = 0; v.w = 0;
                        *reinterpret_cast<uint4 *>(gs + i) = v;
                    }
                }
                __syncthreads();
                if (ovf) break;
            }
        }
        if (tid == 0) {
            if (leaf_ok) { leaf_state[leaf] = 1; leaf_new[leaf] = new_total; } else atomicExch(any_failed, 1u);
        }
        P3_STAMP(3);
#ifdef MC_P3_TIMING
        n_lv++;
#endif
    }
#ifdef MC_P3_TIMING
    if (blockIdx.x == 7 && (tid == 0 || tid == 448)) printf("[p3 block 7 thread %u] %llu leaves, us per leaf: init %.2f merge(own) %.2f wait-for-others %.2f writeback+rest %.2f\n", tid, n_lv, tph[0] * 0.01 / n_lv, tph[1] * 0.01 / n_lv, tph[2] * 0.01 / n_lv, tph[3] * 0.01 / n_lv);
#endif
    if (solid_thr) wave_add_ull(n_solid, (unsigned long long)solid_delta);  // (two's complement: deltas may be negative)
    if (emitting) {
        __syncthreads();
        if (tid == 0) emit.counts[blockIdx.x] = (uint64_t)L.emit_cur < emit.seg_cap ? L.emit_cur : (uint32_t)emit.seg_cap;  // (seg_cap < 2^32)
    }
}

// =============================================================================================
// P3 for super-k-mer records, one region per leaf, one segment per leaf (the headline case): k_p3_dedup.
//
// The records of a leaf come from the ~25 reads that cover each of its ~20 genome loci, and a read that covers a
// locus's run of windows whole and without a sequencing error yields the SAME record as every other such read of its
// strand: on configs[1] (30-fold depth, 1 % errors) 54 % of a leaf's records are copies of another one and they hold
// 51 % of its windows (error-free reads: 82 %).  So the records are first merged in LDS -- a 1024-slot record table
// per workgroup, claimed by a 16-bit fingerprint with one LDS CAS, verified word for word after a barrier (a record
// whose fingerprint meets another record's, or that finds no slot in 8 probes, simply stays on its own: exact either
// way) -- and only the DISTINCT records are expanded into windows, each window adding the record's number of copies.
//
// The region image is 12 bytes a slot instead of 16: key, and one word holding count (17 bits: a leaf adds < 2^16 to
// a count that came in clamped to 32767, see DD_MAX_CAP) and, above it, where the key's read pointer is found when the
// region goes back to HBM (0 = none).  That field is written for exactly one window -- the one whose addition carries the
// count across ptr_pick + 1, which only one addition does; with ptr_tries > 1 (records of other ranks carry no pointer)
// the next crossings try too, with a CAS.  A record slot keeps the pointers of its first three copies, and bits of the
// slot choose among them (with one pointer per record all k-mers of a stretch led into the same read, and a scout that
// had used that read up found no other: + 1 ms of walk).  Table layout, probing and results are those of k_p3_merge.
constexpr uint32_t DD_PROBES = 8;
constexpr uint32_t DD_CNT_BITS = 17, DD_CNT_MASK = (1u << DD_CNT_BITS) - 1;
constexpr uint32_t DD_MAX_CAP = 4096;    // records per leaf: 32767 + 4096 * 16 windows < 2^17, and copies < 2^13 (round 4: 2048 -- the leaves of
                                         // error-free reads, whose table is small, hold ~2650 and all went to the general kernel)
constexpr uint32_t DD_NONE = 0xFFFFFFFFu, DD_OWNER = 0x80000000u;

__device__ __forceinline__ uint32_t dd_hash(uint32_t y, uint32_t z, uint32_t w)
{
    uint32_t h = (y ^ (z * 0x9E3779B1u));
    h = (h ^ (h >> 15)) * 0x85EBCA6Bu;
    h ^= w * 0xC2B2AE35u;
    h = (h ^ (h >> 13)) * 0x27D4EB2Fu;
    return h ^ (h >> 16);
}

// Round 5: the kernel rebuilt around what round 4's counters and its ISA showed (the round-4 kernel: a window cost a byte-table read,
// FIVE cross-lane reads (ds_bpermute: 24 ticks of the SIMD's LDS port each), 21 vector instructions for its two strands
// and a probe loop of its own; the prefix over a wave's records was six dependent ds_bpermutes; the eight waves' shares of
// a leaf differed by a quarter; and -- what decided -- the kernel's time follows the number of instructions its waves issue,
// scalar ones included: 3.9 G of them at a quarter of an instruction a cycle and SIMD, 45 % scalar, most of those the
// compiler's bookkeeping of divergent branches).  Here
//   * the unit of work is a PAIR of windows of one record: its lane reads the record itself from the record table (one
//     ds_read_b128 -- lanes of one record read the same address: a broadcast), cuts the 32 bases the pair spans out once,
//     reverse-complements them once, and both windows' strands are shifts and masks of those two words;
//   * the two keys go through ONE probe loop (lds_probe_claim2: both compare-and-swaps of a step in flight together), the
//     two count additions likewise;
//   * the units of ALL distinct records of the leaf are lined up in one list (a DPP prefix per wave, one LDS atomic per wave
//     for its place) and every wave takes an eighth of the list;
//   * which occurrence leaves its pointer is decided by two bits of the home-slot hash the window needs anyway.
//   * what the compiler turned into exec-mask bookkeeping is written without branches or by hand: the record table's claims
//     (lds_tag_claim), the unit list (lds_line_up8), loops of known trip count unrolled, inserted keys counted when the region
//     goes back (occupied slots after - before) instead of in every probe step.
// 7.3 -> 5.9 ms on configs[1] (DESIGN.md section 3.1 has the steps and the instruction budget).
#ifndef MC_D2_THREADS
#define MC_D2_THREADS 512
#endif
#ifndef MC_D2_UL_CAP
#define MC_D2_UL_CAP 1976
#endif
constexpr int D2_THREADS = MC_D2_THREADS;           // threads of the workgroup that merges a leaf
#ifndef MC_D2_SLOTS
#define MC_D2_SLOTS 1024
#endif
#ifndef MC_D2_WAVES_PER_EU
#define MC_D2_WAVES_PER_EU (MC_D2_THREADS / 128)   // two workgroups on a CU: 4 waves a SIMD with 512 threads (128 registers), 8 with 1024 (64)
#endif
constexpr uint32_t D2_SLOTS = MC_D2_SLOTS;          // its record table (a leaf of configs[1] holds ~250 distinct records)
constexpr int D2_NQ = D2_SLOTS / D2_THREADS;        // record table slots, and records of a round, per thread
static_assert(D2_NQ * D2_THREADS == (int)D2_SLOTS && (D2_NQ == 1 || D2_NQ == 2), "one or two record slots per thread");
constexpr uint32_t D2_UL_CAP = MC_D2_UL_CAP;        // units queued at a time (a leaf of configs[1] holds ~1300)
constexpr uint32_t D2_CP_MASK = 0x1FFFu;  // copies of a record: <= DD_MAX_CAP = 4096 (the tag word keeps 15 bits for them)
static_assert(32767u + DD_MAX_CAP * SK_MAX_WINDOWS < (1u << DD_CNT_BITS) && DD_MAX_CAP <= D2_CP_MASK, "a leaf's additions fit the packed counters");

struct alignas(16) DedupLds {
    uint64_t key[REGION_SLOTS];         // at LDS address 0: the probe loop's addresses are offsets into it
    uint32_t ca[REGION_SLOTS];          // count | (1 + (record slot << 4 | window)) << DD_CNT_BITS
    uint4 drec[D2_SLOTS];               // {d0, d1, d2 | windows - 1, fingerprint << 16 | 0x8000 | copies}: the bases top-aligned in d0:d1:d2; .w == 0: free
    uint32_t dptr[3][D2_SLOTS];
    uint16_t ul[D2_UL_CAP];             // record slot << 3 | pair number; behind what a wave has read of it, and from its end down: pointers (32 bits)
    uint32_t n_new, overflow, emit_cur, n_units;
    uint32_t want_used, pad_[3];        // pointers noted from the list's end down (the waves' own parts ran out)
};
static_assert(sizeof(DedupLds) * (1024 / D2_THREADS) <= 160 * 1024, "sixteen waves of the merge kernel on a CU");

// kmer_device.h rc64_pairs word by word: the bits reversed, the two bits of every base swapped back (one v_bfi), complemented
__device__ __forceinline__ uint32_t rc32_pairs_fast(uint32_t x)
{
    const uint32_t r = __builtin_bitreverse32(x);
    return ~(((r << 1) & 0xAAAAAAAAu) | ((r >> 1) & ~0xAAAAAAAAu));
}
__device__ __forceinline__ uint64_t rc64_pairs_fast(uint64_t x)
{
    return ((uint64_t)rc32_pairs_fast((uint32_t)x) << 32) | rc32_pairs_fast((uint32_t)(x >> 32));
}

// lds_probe_claim for two keys a lane: offA / offB = byte offsets of the home slots in the key array AT LDS ADDRESS 0, mA / mB =
// the lanes that hold a first / a second key.  A lane is done with a key when it found it or claimed a free slot.  Two steps
// with both compare-and-swaps in flight together settle 95 % of the keys; the slowest of a wave's 128 keys needs 4.6 steps
// (measured on a model of configs[1]'s leaves), so what is left then -- a lane's first key, or its second one when the first
// is settled -- goes through a loop of ONE key a lane (half the instructions a step), and the few lanes that still owe
// their second key take it once more.  On return offX = the slot's byte offset, *pendX = lanes whose key found no room in
// P3_MAX_PROBES steps.  (Keys inserted are counted when the region goes back: occupied slots after - before.)
__device__ __forceinline__ void lds_probe_claim2(uint32_t &offA, uint32_t &offB, uint64_t keyA, uint64_t keyB, unsigned long long mA,
                                                 unsigned long long mB, unsigned long long *pendA, unsigned long long *pendB)
{
    unsigned long long oldA, oldB, keyX, sv, hit, mX2, pA, pB;
    uint32_t offX, it;
    const unsigned long long empty = EMPTY_KEY;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        ".rept 2\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "ds_cmpst_rtn_b64 %[oldA], %[offA], %[empty], %[keyA]\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "ds_cmpst_rtn_b64 %[oldB], %[offB], %[empty], %[keyB]\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_cmp_eq_u64 vcc, %[oldA], %[empty]\n\t"
        "v_cmp_eq_u64 %[hit], %[oldA], %[keyA]\n\t"
        "s_or_b64 vcc, vcc, %[hit]\n\t"
        "s_andn2_b64 %[mA], %[mA], vcc\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "v_add_u32 %[offA], 8, %[offA]\n\t"
        "v_and_b32 %[offA], %[wrap], %[offA]\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, %[oldB], %[empty]\n\t"
        "v_cmp_eq_u64 %[hit], %[oldB], %[keyB]\n\t"
        "s_or_b64 vcc, vcc, %[hit]\n\t"
        "s_andn2_b64 %[mB], %[mB], vcc\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "v_add_u32 %[offB], 8, %[offB]\n\t"
        "v_and_b32 %[offB], %[wrap], %[offB]\n\t"
        ".endr\n\t"
        // what is left, one key a lane: the first key of the lanes in mA, the second of those in mB & ~mA
        "s_andn2_b64 %[mX2], %[mB], %[mA]\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "v_mov_b64 %[keyX], %[keyA]\n\t"
        "v_mov_b32 %[offX], %[offA]\n\t"
        "s_mov_b64 exec, %[mX2]\n\t"
        "v_mov_b64 %[keyX], %[keyB]\n\t"
        "v_mov_b32 %[offX], %[offB]\n\t"
        "s_or_b64 exec, %[mA], %[mX2]\n\t"
        "s_mov_b32 %[it], %[maxp]\n"
        "1:\n\t"
        "ds_cmpst_rtn_b64 %[oldA], %[offX], %[empty], %[keyX]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, %[oldA], %[empty]\n\t"
        "v_cmp_eq_u64 %[hit], %[oldA], %[keyX]\n\t"
        "s_or_b64 vcc, vcc, %[hit]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 2f\n\t"
        "v_add_u32 %[offX], 8, %[offX]\n\t"
        "v_and_b32 %[offX], %[wrap], %[offX]\n\t"
        "s_sub_u32 %[it], %[it], 1\n\t"
        "s_cmp_lg_u32 %[it], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_and_b64 %[pA], exec, %[mA]\n\t"
        "s_and_b64 %[pB], exec, %[mX2]\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "v_mov_b32 %[offA], %[offX]\n\t"
        "s_mov_b64 exec, %[mX2]\n\t"
        "v_mov_b32 %[offB], %[offX]\n\t"
        // the lanes whose two keys were both left: their second key (one step in five has such a lane)
        "s_and_b64 exec, %[mA], %[mB]\n\t"
        "s_cbranch_execz 4f\n\t"
        "s_mov_b32 %[it], %[maxp]\n"
        "3:\n\t"
        "ds_cmpst_rtn_b64 %[oldB], %[offB], %[empty], %[keyB]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u64 vcc, %[oldB], %[empty]\n\t"
        "v_cmp_eq_u64 %[hit], %[oldB], %[keyB]\n\t"
        "s_or_b64 vcc, vcc, %[hit]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 4f\n\t"
        "v_add_u32 %[offB], 8, %[offB]\n\t"
        "v_and_b32 %[offB], %[wrap], %[offB]\n\t"
        "s_sub_u32 %[it], %[it], 1\n\t"
        "s_cmp_lg_u32 %[it], 0\n\t"
        "s_cbranch_scc1 3b\n"
        "4:\n\t"
        "s_or_b64 %[pB], %[pB], exec\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [offA] "+v"(offA), [offB] "+v"(offB), [mA] "+s"(mA), [mB] "+s"(mB), [oldA] "=&v"(oldA), [oldB] "=&v"(oldB), [keyX] "=&v"(keyX),
          [offX] "=&v"(offX), [sv] "=&s"(sv), [hit] "=&s"(hit), [mX2] "=&s"(mX2), [pA] "=&s"(pA), [pB] "=&s"(pB), [it] "=&s"(it)
        // (the two steps above count: a key may sit P3_MAX_PROBES slots from its home at most -- where a look-up stops, kmer_device.h.
        // Round 5's first build let the loops run P3_MAX_PROBES steps MORE: in a region crowded enough, a key two slots beyond what
        // table_get examines; the randomised soak found the walk that missed it)
        : [empty] "v"(empty), [keyA] "v"(keyA), [keyB] "v"(keyB), [wrap] "s"((uint32_t)(REGION_SLOTS * 8u - 8u)), [maxp] "s"((uint32_t)(P3_MAX_PROBES - 2u))
        : "vcc", "scc", "memory");
    static_assert(P3_MAX_PROBES > 2, "two probe steps outside the loops");
    *pendA = pA;
    *pendB = pB;
}

// A record claims a place in the record table by its tag (fingerprint << 16 | 0x8000): up to DD_PROBES compare-and-swaps on
// the tag words from byte address `addr` on (record table of 16-byte entries, a power of two of them, aligned to its
// size).  On return addr = the tag word of the place, *claimed = lanes that took a free one (they store their record
// there), *none = lanes that met DD_PROBES other records.  (The compiler's loop spent 35 instructions a step on this.)
__device__ __forceinline__ void lds_tag_claim(uint32_t &addr, uint32_t tag, unsigned long long *claimed, unsigned long long *none)
{
    unsigned long long sv, hit, cl, nn;
    uint32_t old, it, nxt;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 %[cl], 0\n\t"
        "s_mov_b32 %[it], %[maxp]\n"
        "1:\n\t"
        "ds_cmpst_rtn_b32 %[old], %[addr], %[zero], %[tag]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32 vcc, 0, %[old]\n\t"
        "v_and_b32 %[old], 0xffff8000, %[old]\n\t"
        "s_or_b64 %[cl], %[cl], vcc\n\t"
        "v_cmp_eq_u32 %[hit], %[old], %[tag]\n\t"
        "s_or_b64 vcc, vcc, %[hit]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 2f\n\t"
        "v_add_u32 %[nxt], 16, %[addr]\n\t"
        "v_bfi_b32 %[addr], %[wrap], %[nxt], %[addr]\n\t"
        "s_sub_u32 %[it], %[it], 1\n\t"
        "s_cmp_lg_u32 %[it], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_mov_b64 %[nn], exec\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [addr] "+v"(addr), [old] "=&v"(old), [nxt] "=&v"(nxt), [sv] "=&s"(sv), [hit] "=&s"(hit), [cl] "=&s"(cl), [nn] "=&s"(nn), [it] "=&s"(it)
        : [zero] "v"(0u), [tag] "v"(tag), [wrap] "s"((uint32_t)(D2_SLOTS * 16u - 1u)), [maxp] "s"((uint32_t)DD_PROBES)
        : "vcc", "scc", "memory");
    *claimed = cl;
    *none = nn;
}

// the n <= 8 consecutive numbers val, val + 1, ... as 16-bit words from LDS byte address dst on: eight compare-and-store
// steps under a shrinking exec mask (the compiler's loop of the same took seven instructions a trip)
__device__ __forceinline__ void lds_line_up8(uint32_t dst, uint32_t val, uint32_t n)
{
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_u32 vcc, 0, %[n]\n\t"
        "ds_write_b16 %[dst], %[val]\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "v_cmpx_lt_u32 vcc, 1, %[n]\n\t"
        "ds_write_b16 %[dst], %[val] offset:2\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "v_cmpx_lt_u32 vcc, 2, %[n]\n\t"
        "ds_write_b16 %[dst], %[val] offset:4\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "v_cmpx_lt_u32 vcc, 3, %[n]\n\t"
        "ds_write_b16 %[dst], %[val] offset:6\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "v_cmpx_lt_u32 vcc, 4, %[n]\n\t"
        "ds_write_b16 %[dst], %[val] offset:8\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "v_cmpx_lt_u32 vcc, 5, %[n]\n\t"
        "ds_write_b16 %[dst], %[val] offset:10\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "v_cmpx_lt_u32 vcc, 6, %[n]\n\t"
        "ds_write_b16 %[dst], %[val] offset:12\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "v_cmpx_lt_u32 vcc, 7, %[n]\n\t"
        "ds_write_b16 %[dst], %[val] offset:14\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [val] "+v"(val), [sv] "=&s"(sv)
        : [dst] "v"(dst), [n] "v"(n)
        : "vcc", "memory");
}

// ONE_GPU: the run of one GPU on its own reads -- every record carries its pointer (ptr_tries == 1) and no list of solid k-mers
// is kept for a gather (P3Emit): the branches of the other case leave the loops (uniform ones cost scalar instructions too).
// K31: k == 31 -- the windows' shifts and masks are constants (a pair of windows is exactly the 32 bases of the two words)
template <bool VIRGIN, bool ONE_GPU, bool K31 = false>
__global__ void __launch_bounds__(D2_THREADS, MC_D2_WAVES_PER_EU) k_p3_dedup(const uint4 *__restrict__ leaf_recs, const uint32_t *__restrict__ leaf_ptrs,
                                                          const uint32_t *__restrict__ leaf_counts, uint64_t seg_cap, uint32_t n_leaves,
                                                          TableView t, uint32_t *leaf_state, uint32_t *leaf_new, uint32_t *any_failed,
                                                          uint32_t solid_thr, unsigned long long *n_solid, int k, P3Emit emit,
                                                          uint32_t ptr_tries_arg, const uint32_t *lost)
{
    const uint32_t ptr_tries = ONE_GPU ? 1u : ptr_tries_arg;
    if (lost && *lost) return;
    __shared__ DedupLds L;
    const uint32_t tid = threadIdx.x, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = tid & 63u;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint64_t *)L.key != 0u) {  // (uniform; never: the kernel's one LDS object)
        if (tid == 0) atomicExch(any_failed, 1u);
        return;
    }
    const bool emitting = !ONE_GPU && emit.recs != nullptr && solid_thr != 0;
    if (tid == 0) L.emit_cur = emitting ? emit.counts[blockIdx.x] : 0u;
    auto clear_records = [&] {  // the record table's tags and pointer fields (trip counts known: no loop bookkeeping)
#pragma unroll
        for (int it = 0; it < D2_NQ; it++) L.drec[tid + (uint32_t)it * D2_THREADS].w = 0;
        static_assert((3 * D2_SLOTS / 4) % D2_THREADS == 0 || (3 * D2_SLOTS / 4) < D2_THREADS || D2_THREADS == 512, "dptr in whole steps");
#pragma unroll
        for (int it = 0; it < (int)((3 * D2_SLOTS / 4 + D2_THREADS - 1) / D2_THREADS); it++) {
            const uint32_t i = tid + (uint32_t)it * D2_THREADS;
            if (i < 3 * D2_SLOTS / 4) reinterpret_cast<uint4 *>(L.dptr)[i] = make_uint4(0, 0, 0, 0);
        }
    };
    clear_records();
    long long solid_delta = 0;
    static_assert(offsetof(DedupLds, drec) % (D2_SLOTS * 16u) == 0, "lds_tag_claim wraps inside the aligned record table");
    const uint32_t drec_w = (uint32_t)offsetof(DedupLds, drec) + 12u;  // byte address of the first tag word (the struct sits at LDS address 0)
    const uint32_t ptr_from = solid_thr >= 2 ? 1u : 0u;
    const uint32_t sh_a = K31 ? 2u : 64u - 2u * (uint32_t)k, sh_b = K31 ? 0u : 62u - 2u * (uint32_t)k;  // (k <= 31: a pair of windows spans k + 1 <= 32 bases)
    const uint64_t kmask = ~0ull >> sh_a;
    uint32_t pick_tbl = 0;  // ptr_pick's choice for the four values of its two bits
    for (uint32_t r = 0; r < 4; r++) pick_tbl |= (ptr_pick((uint64_t)r, ptr_from, solid_thr) - ptr_from) << (2 * r);

    uint32_t cur_leaf = 0;
    // one occurrence set of `key` (inc copies) by the plain rule: records without a place in the record table (a fingerprint
    // met another record's, or eight occupied places: one leaf in a thousand has such a record)
    auto add_plain = [&](uint64_t key, uint32_t inc) {
        uint32_t s = sk_home(key);
#pragma unroll 1
        for (uint32_t p = 0; p < P3_MAX_PROBES; p++) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&L.key[s]), (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == EMPTY_KEY || old == key) {
                atomicAdd(&L.ca[s], inc);
                return;
            }
            s = (s + 1u) & (REGION_SLOTS - 1u);
        }
        if (!ovf_push(t, key, inc, 0u, cur_leaf)) atomicExch(&L.overflow, 1u);
    };

#ifdef MC_P3_TIMING
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memrealtime(), n_lv = 0;
#define P3E_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tph[i] += n_ - tl; tl = n_; } while (0)
#else
#define P3E_STAMP(i) do {} while (0)
#endif
    uint32_t st_nxt = 1u, n_nxt = 0;
    uint4 pre_rec[D2_NQ];
    uint32_t pre_ptr[D2_NQ];
#pragma unroll
    for (int q = 0; q < D2_NQ; q++) { pre_rec[q] = make_uint4(0, 0, 0, 0); pre_ptr[q] = 0; }
    auto fetch = [&](uint32_t lf, uint32_t n) {  // this thread's first records of leaf lf
#pragma unroll
        for (int q = 0; q < D2_NQ; q++) {
            const uint32_t r = tid + (uint32_t)q * D2_THREADS;
            if (r < n) {
                pre_rec[q] = leaf_recs[(uint64_t)lf * seg_cap + r];
                pre_ptr[q] = leaf_ptrs ? leaf_ptrs[(uint64_t)lf * seg_cap + r] : pre_rec[q].x;  // (no array: the record's first word, k_sk2_scatter_compact)
            }
        }
    };
    if (blockIdx.x < n_leaves) {
        st_nxt = leaf_state[blockIdx.x];
        n_nxt = min(leaf_counts[blockIdx.x], (uint32_t)seg_cap);
        fetch(blockIdx.x, n_nxt);
    }
    __syncthreads();
    for (uint32_t leaf = blockIdx.x; leaf < n_leaves; leaf += gridDim.x) {
        const uint32_t st_cur = st_nxt, n0 = n_nxt;
        uint4 cur_rec[D2_NQ];
        uint32_t cur_ptr[D2_NQ];
#pragma unroll
        for (int q = 0; q < D2_NQ; q++) { cur_rec[q] = pre_rec[q]; cur_ptr[q] = pre_ptr[q]; }
        const uint32_t nl = leaf + gridDim.x;
        if (nl < n_leaves) {
            st_nxt = leaf_state[nl];
            n_nxt = min(leaf_counts[nl], (uint32_t)seg_cap);
        }
        if (st_cur || n0 > DD_MAX_CAP) {  // uniform; a leaf of more records than the packed counters are safe for is left
            if (nl < n_leaves) fetch(nl, n_nxt);  // to k_p3_merge, which the host enqueues behind this kernel
            continue;
        }
        P3E_STAMP(7);
        Slot *gs = t.slots + (uint64_t)leaf * REGION_SLOTS;
        const uint4 *recs = leaf_recs + (uint64_t)leaf * seg_cap;
        const uint32_t *ptrs = leaf_ptrs + (uint64_t)leaf * seg_cap;
        int solid_before = 0;
        uint32_t occupied = 0;  // (uniform) slots of the region that hold a key: after the merge, less before it = keys inserted
        if (VIRGIN) {  // 16 bytes a store
            static_assert((REGION_SLOTS / 4) % D2_THREADS == 0, "the image is cleared in whole steps");
#pragma unroll
            for (int it = 0; it < (int)(REGION_SLOTS / 2 / D2_THREADS); it++) reinterpret_cast<uint4 *>(L.key)[tid + (uint32_t)it * D2_THREADS] = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
            for (int it = 0; it < (int)(REGION_SLOTS / 4 / D2_THREADS); it++) reinterpret_cast<uint4 *>(L.ca)[tid + (uint32_t)it * D2_THREADS] = make_uint4(0, 0, 0, 0);
        } else {
#pragma unroll 2
            for (int it = 0; it < (int)(REGION_SLOTS / D2_THREADS); it++) {  // (all eight loads in flight cost 32 registers)
                const uint32_t i = tid + (uint32_t)it * D2_THREADS;
                const uint4 raw = *reinterpret_cast<const uint4 *>(gs + i);
                L.key[i] = ((uint64_t)raw.y << 32) | raw.x;
                L.ca[i] = min(raw.z, 32767u);  // (anything above reads the same: kmer_device.h table_get)
                solid_before += solid_thr && raw.z >= solid_thr;  // (empty slots hold count 0)
                occupied -= (uint32_t)__popcll(__ballot(raw.z != 0));  // (a key in the table has been counted at least once)
            }
        }
        if (tid == 0) { L.n_new = 0; L.overflow = 0; L.n_units = 0; L.want_used = 0; }
        cur_leaf = leaf;
        // ---- A: the records into the record table, 1024 at a time (nearly always all of them)
        for (uint32_t base = 0; base < n0; base += D2_SLOTS) {  // uniform
            uint32_t e0[D2_NQ], e1[D2_NQ], e2[D2_NQ], rptr[D2_NQ], sl[D2_NQ];
            bool have[D2_NQ];
#pragma unroll
            for (int q = 0; q < D2_NQ; q++) {
                const uint32_t r = base + (uint32_t)q * D2_THREADS + tid;
                have[q] = r < n0;
                sl[q] = DD_NONE;
                uint4 rec;
                if (base == 0) { rec = cur_rec[q]; rptr[q] = cur_ptr[q]; }
                else if (have[q]) { rec = recs[r]; rptr[q] = leaf_ptrs ? ptrs[r] : rec.x; }
                else { rec = make_uint4(0, 0, 0, 0); rptr[q] = 0; }
                // the record's 46 bases top-aligned in three words, windows - 1 in the four bits below them
                e0[q] = __builtin_amdgcn_alignbit(rec.w, rec.z, 28);
                e1[q] = __builtin_amdgcn_alignbit(rec.z, rec.y, 28);
                e2[q] = (rec.y << 4) | (rec.w >> 28);
                if (have[q]) {
                    const uint32_t h = dd_hash(rec.y, rec.z, rec.w), tag = (h & 0xFFFF0000u) | 0x8000u;
                    uint32_t at = drec_w + ((h & (D2_SLOTS - 1)) << 4);
                    unsigned long long claimed, none;
                    lds_tag_claim(at, tag, &claimed, &none);
                    const uint32_t slot = (at - drec_w) >> 4;
                    if ((claimed >> lane) & 1ull) {  // this copy's words are what the others are compared with (its pointer fields were cleared with the table)
                        L.drec[slot].x = e0[q]; L.drec[slot].y = e1[q]; L.drec[slot].z = e2[q];
                        sl[q] = slot | DD_OWNER;
                    } else if (!((none >> lane) & 1ull)) {
                        sl[q] = slot;
                    }
                }
            }
            P3E_STAMP(0);
            __syncthreads();
            P3E_STAMP(1);
#pragma unroll
            for (int q = 0; q < D2_NQ; q++) {
                // (one 16-byte read and no short cuts: the compiler made three dependent reads behind three branches of the comparison)
                const bool has = sl[q] != DD_NONE;
                const uint32_t slot = has ? sl[q] & (D2_SLOTS - 1u) : 0u;
                const uint4 e = L.drec[slot];
                const bool same = has & (((e.x ^ e0[q]) | (e.y ^ e1[q]) | (e.z ^ e2[q])) == 0u);  // (its own words for the copy that claimed the place)
                if (same) {
                    // the first three copies that carry a read pointer are remembered (the occurrences of neighbouring
                    // k-mers arrive in the same order: kmer_device.h ptr_pick says why one pointer per record is too few)
                    const uint32_t c = atomicAdd(&L.drec[slot].w, 1u) & D2_CP_MASK;
                    if (rptr[q] != 0u) {
                        if (c < 3u) {
                            L.dptr[c][slot] = rptr[q];
                        } else if (ptr_tries > 1) {  // (copies of other ranks carry none: a later one fills a field that stayed empty)
                            for (uint32_t f = 0; f < 3u; f++)
                                if (L.dptr[f][slot] == 0) { L.dptr[f][slot] = rptr[q]; break; }
                        }
                    }
                }
                // a record without a place (another record with its fingerprint, or eight occupied places) goes into the region
                // image as it is: one copy of every window, no pointer
                const bool alone = have[q] & !same;
                if (__ballot(alone)) {  // uniform; one leaf in a thousand
                    if (alone) {
                        const uint32_t nw = (e2[q] & 15u) + 1u;
                        const uint64_t top = ((uint64_t)e0[q] << 32) | e1[q];
#pragma unroll 1
                        for (uint32_t j = 0; j < nw; j++) {
                            const uint32_t sh = 2u * j;
                            const uint64_t fw = ((top << sh) | (uint64_t)((e2[q] >> 1) >> (31u - sh))) >> sh_a, rc = rc_packed(fw, k);
                            add_plain(rc < fw ? rc : fw, 1u);
                        }
                    }
                }
            }
        }
        // ---- the pairs of windows of all distinct records, lined up (a record table slot belongs to one thread; its tag and
        // window count were written before the barrier above, and the copies are only read behind the next one)
        uint32_t u_cnt[D2_NQ], u_at[D2_NQ];
        {
            uint32_t mine = 0;
#pragma unroll
            for (int q = 0; q < D2_NQ; q++) {
                const uint32_t slot = tid + (uint32_t)q * D2_THREADS;
                const uint2 zw = *reinterpret_cast<const uint2 *>(&L.drec[slot].z);
                u_cnt[q] = zw.y ? ((zw.x & 15u) + 2u) >> 1 : 0u;
                mine += u_cnt[q];
            }
            const uint32_t incl = wave_incl_sum(mine);
            uint32_t wbase = 0;
            if (lane == 63u && incl) wbase = atomicAdd(&L.n_units, incl);
            wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, 63);
            u_at[0] = wbase + incl - mine;
#pragma unroll
            for (int q = 1; q < D2_NQ; q++) u_at[q] = u_at[q - 1] + u_cnt[q - 1];
        }
        auto line_up = [&](uint32_t r0) {  // the units r0 .. r0 + D2_UL_CAP - 1 into the list
            // (uniform: a wave whose units all fall into the round -- nearly always -- stores without looking)
            const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readlane((int)u_at[0], 0);
            const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readlane((int)(u_at[D2_NQ - 1] + u_cnt[D2_NQ - 1]), 63);
            const bool inside = w_lo >= r0 && w_hi <= r0 + D2_UL_CAP;
#pragma unroll
            for (int q = 0; q < D2_NQ; q++) {
                const uint32_t val = (tid + (uint32_t)q * D2_THREADS) << 3;
                if (inside) {
                    lds_line_up8((uint32_t)offsetof(DedupLds, ul) + 2u * (u_at[q] - r0), val, u_cnt[q]);  // (the struct sits at LDS address 0)
                } else {
                    for (uint32_t u = 0; u < u_cnt[q]; u++) {
                        const uint32_t at = u_at[q] + u - r0;
                        if (at < D2_UL_CAP) L.ul[at] = (uint16_t)(val | u);
                    }
                }
            }
        };
        line_up(0);
        P3E_STAMP(2);
        __syncthreads();
        P3E_STAMP(3);
        // ---- B: the pairs into the region image, an eighth of the list per wave
        const uint32_t n_units = L.n_units;
        const bool one_round = n_units <= D2_UL_CAP;
        uint32_t *wl = reinterpret_cast<uint32_t *>(L.ul);
        const uint32_t tail_lo = (n_units + 1u) >> 1;  // the list's 32-bit words behind the units
        // the pointer of window j of the record in slot rs: one of the record's (up to three) copies' pointers, chosen by bits of the
        // region slot sg, moved on to the window
        auto window_ptr = [&](uint32_t rs, uint32_t j, uint32_t sg) {
            const uint32_t r = sg & 3u, f = r == 3u ? 0u : r;
            uint32_t p0 = L.dptr[f][rs];
            if (p0 == 0) p0 = L.dptr[0][rs];
            if (ptr_tries > 1 && p0 == 0) p0 = L.dptr[1][rs] ? L.dptr[1][rs] : L.dptr[2][rs];  // (fields fill in the order of ALL copies there)
            return p0 - 1u < 0x7FFFFFEFu ? p0 + j : ptr_advance(p0, j);  // (exact pointers: kmer_device.h ptr_advance's first case)
        };
        auto note_ptr = [&](uint32_t sg, uint32_t at, uint32_t pw) {  // region slot sg finds its pointer in word `at` of the list
            wl[at] = pw;
            if (pw) {
                const uint32_t occ = (at + 1u) << DD_CNT_BITS;
                if (ptr_tries == 1) {
                    atomicOr(&L.ca[sg], occ);
                } else {  // (records of other ranks carry no pointer: the first occurrence that has one takes the field)
                    for (int a = 0; a < 4; a++) {
                        const uint32_t cur = L.ca[sg];
                        if ((cur >> DD_CNT_BITS) != 0 || atomicCAS(&L.ca[sg], cur, cur | occ) == cur) break;
                    }
                }
            }
        };
        for (uint32_t r0 = 0; r0 < n_units; r0 += D2_UL_CAP) {  // uniform; nearly always one round
            if (r0) {
                __syncthreads();
                line_up(r0);
                __syncthreads();
            }
            const uint32_t nr = min(n_units - r0, D2_UL_CAP);
            const uint32_t lo = (nr * wv) / (D2_THREADS / 64), hi = (nr * (wv + 1u)) / (D2_THREADS / 64);
            const uint32_t wbeg = (lo + 1u) >> 1;
            uint32_t wcur = wbeg;  // (uniform) the wave's notes: 32-bit words of its own part of the list, behind what it has read
#ifdef MC_P3_NOB   // (timing experiments: the tables of such builds are not usable)
            if (n_leaves) continue;
#endif
#pragma unroll 1
            for (uint32_t b = lo; b < hi; b += 64) {  // uniform
                const uint32_t idx = b + lane;
                const bool va = idx < hi;
                const uint32_t ue = va ? (uint32_t)L.ul[idx] : 0u;
                const uint32_t slot = ue >> 3, u = ue & 7u;
                const uint4 e = L.drec[slot];
                const uint32_t nw = (e.z & 15u) + 1u, ja = 2u * u, cp = e.w & D2_CP_MASK;
                const bool vb = va && ja + 1u < nw;
                // the 32 bases from the pair's first one on, and their reverse complement
                const uint32_t sh = 4u * u;  // <= 28
                const uint64_t X = ((((uint64_t)e.x << 32) | e.y) << sh) | (uint64_t)((e.z >> 1) >> (31u - sh));
                const uint64_t R = rc64_pairs_fast(X);
                const uint64_t fa = X >> sh_a, fb = (X >> sh_b) & kmask, ra = R & kmask, rb = (R >> 2) & kmask;
                const uint64_t keyA = ra < fa ? ra : fa, keyB = rb < fb ? rb : fb;
                const uint32_t hxA = sk_home_mix(keyA), hxB = sk_home_mix(keyB);
                uint32_t offA = (hxA >> (32 - MC_REGION_LG)) << 3, offB = (hxB >> (32 - MC_REGION_LG)) << 3;
                unsigned long long pendA, pendB;
                // (lane masks straight from and to conditions: __ballot(int) and a shifted-mask test cost two and three vector
                // instructions each where the hardware needs none)
                const unsigned long long mvA = __builtin_amdgcn_ballot_w64(va), mvB = __builtin_amdgcn_ballot_w64(vb);
                lds_probe_claim2(offA, offB, keyA, keyB, mvA, mvB, &pendA, &pendB);
                const bool okA = __builtin_amdgcn_inverse_ballot_w64(mvA & ~pendA), okB = __builtin_amdgcn_inverse_ballot_w64(mvB & ~pendB);
                // (a lane without a key adds nothing to a word of its own)
                const uint32_t sA = okA ? offA >> 3 : lane, sB = okB ? offB >> 3 : lane;
                const uint32_t befA = atomicAdd(&L.ca[sA], okA ? cp : 0u) & DD_CNT_MASK;
                const uint32_t befB = atomicAdd(&L.ca[sB], okB ? cp : 0u) & DD_CNT_MASK;
                // the count an occurrence leaves its pointer at (kmer_device.h ptr_pick: here by two bits of the home-slot hash)
                const uint32_t firstA = ptr_from + 1u + ((pick_tbl >> ((hxA >> 17) & 6u)) & 3u);
                const uint32_t firstB = ptr_from + 1u + ((pick_tbl >> ((hxB >> 17) & 6u)) & 3u);
                const bool wantA = okA && befA < firstA + ptr_tries - 1u && befA + cp >= firstA;  // (one addition per key when ptr_tries == 1)
                const bool wantB = okB && befB < firstB + ptr_tries - 1u && befB + cp >= firstB;
                const unsigned long long wmA = __builtin_amdgcn_ballot_w64(wantA), wmB = __builtin_amdgcn_ballot_w64(wantB);
                if (wmA | wmB) {  // uniform
                    const uint32_t wnA = (uint32_t)__popcll(wmA), wnB = (uint32_t)__popcll(wmB);
                    if (one_round) {
                        // ... is noted in the part of the list this wave has used up: (slot of the region, slot of the record, window)
                        const uint32_t wlim = min(b + 64u, hi) >> 1;
                        const uint32_t iA = wcur + __builtin_amdgcn_mbcnt_hi((uint32_t)(wmA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmA, 0u));
                        const uint32_t iB = wcur + wnA + __builtin_amdgcn_mbcnt_hi((uint32_t)(wmB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmB, 0u));
                        if (wantA && iA < wlim) wl[iA] = sA << 14 | slot << 4 | ja;
                        if (wantB && iB < wlim) wl[iB] = sB << 14 | slot << 4 | (ja + 1u);
                        if (wcur + wnA + wnB > wlim) {  // uniform
                            // More crossings than the wave has read units: error-free reads -- few distinct records, every window
                            // of them a k-mer that reaches the threshold at once.  Those pointers are worked out here and noted
                            // from the END of the list down, where no unit lies (an LDS counter hands the words out).
                            const bool oA = wantA && iA >= wlim, oB = wantB && iB >= wlim;
                            const unsigned long long omA = __builtin_amdgcn_ballot_w64(oA), omB = __builtin_amdgcn_ballot_w64(oB);
                            const uint32_t onA = (uint32_t)__popcll(omA), on = onA + (uint32_t)__popcll(omB);
                            uint32_t ob = 0;
                            if (lane == 0) ob = atomicAdd(&L.want_used, on);
                            ob = (uint32_t)__builtin_amdgcn_readfirstlane((int)ob);
                            const uint32_t top = D2_UL_CAP / 2u - 1u;
                            const uint32_t kA = ob + __builtin_amdgcn_mbcnt_hi((uint32_t)(omA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)omA, 0u));
                            const uint32_t kB = ob + onA + __builtin_amdgcn_mbcnt_hi((uint32_t)(omB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)omB, 0u));
                            // (what finds no room there either leaves no pointer: a hint less)
                            if (oA && kA <= top && top - kA >= tail_lo) note_ptr(sA, top - kA, window_ptr(slot, ja, sA));
                            if (oB && kB <= top && top - kB >= tail_lo) note_ptr(sB, top - kB, window_ptr(slot, ja + 1u, sB));
                        }
                        wcur = min(wcur + wnA + wnB, wlim);
                    } else {
                        // a leaf of more units than the list holds: (record slot, window) beside the count, worked out when the region goes back
                        const uint32_t occA = (((slot << 4) | ja) + 1u) << DD_CNT_BITS, occB = occA + (1u << DD_CNT_BITS);
                        // (records of other ranks carry no pointer: theirs must not take the field)
                        const bool hp = ptr_tries == 1 || (L.dptr[0][slot] | L.dptr[1][slot] | L.dptr[2][slot]) != 0;
                        if (wantA && hp)
                            for (int a = 0; a < 4; a++) {
                                const uint32_t cur = L.ca[sA];
                                if ((cur >> DD_CNT_BITS) != 0 || atomicCAS(&L.ca[sA], cur, cur | occA) == cur) break;
                            }
                        if (wantB && hp)
                            for (int a = 0; a < 4; a++) {
                                const uint32_t cur = L.ca[sB];
                                if ((cur >> DD_CNT_BITS) != 0 || atomicCAS(&L.ca[sB], cur, cur | occB) == cur) break;
                            }
                    }
                }
                if (pendA | pendB) {  // uniform; a region that is full (the table's chain takes the occurrence: TableView::ovf)
                    if (__builtin_amdgcn_inverse_ballot_w64(pendA) && !ovf_push(t, keyA, cp, 0u, cur_leaf)) atomicExch(&L.overflow, 1u);
                    if (__builtin_amdgcn_inverse_ballot_w64(pendB) && !ovf_push(t, keyB, cp, 0u, cur_leaf)) atomicExch(&L.overflow, 1u);
                }
            }
            // the wave's notes become pointers, a lane each: one of the record's (up to three) copies' pointers, moved on to the
            // window; the region slot gets the note's place, and the write-back finds the pointer there
#pragma unroll 1
            for (uint32_t i0 = wbeg; i0 < wcur; i0 += 64) {  // uniform; one turn
                const uint32_t i = i0 + lane;
                if (i < wcur) {
                    const uint32_t e = wl[i], rs = (e >> 4) & (D2_SLOTS - 1u), j = e & 15u, sg = e >> 14;
                    note_ptr(sg, i, window_ptr(rs, j, sg));
                }
            }
        }
        if (nl < n_leaves) fetch(nl, n_nxt);  // the next leaf's first records: its count has long arrived
        P3E_STAMP(4);
        __syncthreads();
        P3E_STAMP(5);
        const bool ovf = L.overflow != 0;
        if (!ovf) {
            // four slots of a thread at a time, their LDS words requested together; FAST: the pointers are in the list
            auto write_back = [&](auto fast) {
                constexpr bool FAST = decltype(fast)::value;
#pragma unroll
                for (int it = 0; it < (int)(REGION_SLOTS / (4 * D2_THREADS)); it++) {
                    const uint32_t i0 = tid + (uint32_t)it * 4u * D2_THREADS;
                    uint64_t kk[4];
                    uint32_t cc[4], pp[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t i = i0 + (uint32_t)u * D2_THREADS;
                        kk[u] = L.key[i];
                        cc[u] = L.ca[i];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t o = cc[u] >> DD_CNT_BITS;
                        pp[u] = 0;
                        if (o) {
                            if (FAST) {
                                pp[u] = wl[o - 1u];
                            } else {  // the pointer of one of the record's copies, by bits of the key; the first copy's when that one has none
                                const uint32_t sl = (o - 1u) >> 4, j = (o - 1u) & 15u;
                                const uint32_t r = ((uint32_t)kk[u] >> 7) & 3u, f = r == 3u ? 0u : r;
                                uint32_t p0 = L.dptr[f][sl];
                                if (p0 == 0) p0 = L.dptr[0][sl];
                                if (ptr_tries > 1 && p0 == 0) p0 = L.dptr[1][sl] ? L.dptr[1][sl] : L.dptr[2][sl];
                                pp[u] = ptr_advance(p0, j);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t i = i0 + (uint32_t)u * D2_THREADS;
                        uint4 v;
                        v.x = (uint32_t)kk[u]; v.y = (uint32_t)(kk[u] >> 32);
                        v.z = cc[u] & DD_CNT_MASK;
                        v.w = pp[u];
                        if (!VIRGIN && v.w == 0) v.w = reinterpret_cast<const uint32_t *>(gs + i)[3];  // the pointer the slot had
#ifdef MC_P3_NOWB
                        if (v.z == 0x12345u)
#endif
                        *reinterpret_cast<uint4 *>(gs + i) = v;
                        occupied += (uint32_t)__popcll(__ballot(v.z != 0));  // (a key in the image has been counted at least once)
                        const bool solid = solid_thr && v.z >= solid_thr;
                        solid_delta += solid;
                        if (emitting) {  // (uniform; the loop's trip count is the same for every lane)
                            const unsigned long long m = __ballot(solid);
                            if (m) {
                                uint32_t ebase = 0;
                                const int leader = __ffsll((long long)m) - 1;
                                if ((int)lane == leader) ebase = atomicAdd(&L.emit_cur, (uint32_t)__popcll(m));
                                ebase = __shfl(ebase, leader);
                                if (solid) {
                                    const uint32_t pos = ebase + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                                    if (pos < emit.seg_cap) {
                                        v.z = min(v.z, 32767u);
                                        emit.recs[(uint64_t)blockIdx.x * emit.seg_cap + pos] = v;
                                    } else {
                                        atomicExch(emit.lost, 1u);
                                    }
                                }
                            }
                        }
                    }
                }
            };
            if (one_round) write_back(std::true_type{}); else write_back(std::false_type{});
            if (lane == 0 && occupied) atomicAdd(&L.n_new, occupied);  // (two's complement: a wave may have loaded more keys than it writes back)
            solid_delta -= solid_before;
        } else if (VIRGIN) {  // nothing was there: leave a valid empty region behind
#pragma unroll
            for (int it = 0; it < (int)(REGION_SLOTS / D2_THREADS); it++) {
                const uint32_t i = tid + (uint32_t)it * D2_THREADS;
                uint4 v; v.x = 0xFFFFFFFFu; v.y = 0xFFFFFFFFu; v.z = 0; v.w = 0;
                *reinterpret_cast<uint4 *>(gs + i) = v;
            }
        }
        clear_records();
        P3E_STAMP(6);
        __syncthreads();
        if (tid == 0) {  // (behind the barrier: the waves' key counts are in)
            if (!ovf) { leaf_state[leaf] = 1; leaf_new[leaf] = L.n_new; } else atomicExch(any_failed, 1u);
        }
#ifdef MC_P3_TIMING
        n_lv++;
#endif
    }
#ifdef MC_P3_TIMING
    if (blockIdx.x == 7 && (tid == 0 || tid == 448)) printf("[p3d block 7 thread %u] %llu leaves, us per leaf: init+A1 %.2f wait %.2f A2+line-up %.2f wait %.2f B %.2f wait %.2f writeback %.2f wait+next %.2f\n", tid, n_lv, tph[0] * 0.01 / n_lv, tph[1] * 0.01 / n_lv, tph[2] * 0.01 / n_lv, tph[3] * 0.01 / n_lv, tph[4] * 0.01 / n_lv, tph[5] * 0.01 / n_lv, tph[6] * 0.01 / n_lv, tph[7] * 0.01 / n_lv);
#endif
    if (solid_thr) wave_add_ull(n_solid, (unsigned long long)solid_delta);  // (two's complement: deltas may be negative)
    if (emitting) {
        __syncthreads();
        if (tid == 0) emit.counts[blockIdx.x] = (uint64_t)L.emit_cur < emit.seg_cap ? L.emit_cur : (uint32_t)emit.seg_cap;
    }
}

// n_used += sum(leaf_new): one atomic per workgroup instead of one per region on a single hot address
__global__ void k_sum_leaf_new(const uint32_t *__restrict__ leaf_new, uint32_t n, unsigned long long *n_used)
{
    unsigned long long v = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v += leaf_new[i];
    wave_add_ull(n_used, v);
}

// drains the spill list (and serves as the direct path for key streams that carry a hint)
__global__ void k_add_keys_hint(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ hints, uint64_t n,
                                TableView t, uint32_t solid_thr = 0, unsigned long long *n_solid = nullptr)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_new = 0, n_cross = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t before;
        n_new += table_add(t, keys[i], 1u, hints[i], &before);
        n_cross += crosses(before, 1u, solid_thr);
    }
    wave_add_ull(t.n_used, n_new);
    if (solid_thr) wave_add_ull(n_solid, n_cross);
}

}  // namespace mc

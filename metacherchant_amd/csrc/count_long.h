// Long records: the super-k-mer form of the counting pipeline for polynomial-hash keys, 32 < k <= 64 (configs[2]: k = 63).
//
// The per-window form moves one 12-byte (key, read pointer) record per window through two scatter levels and the merge:
// 65 GB a step for configs[2] scaled to 10 M reads, and round 5 measured those two levels at the memory system's rate already.
// A hash key says nothing about its bases, but the reads do: the REGION of a key is the bin of its k-mer's canonical minimizer
// (kmer_device.h, as for packed keys), worked out where the bases are at hand -- here, and in the walk (solid_locate_kmer) --
// and the key itself (src/utils/PolynomialHash.java:19-28, the smaller of the two strands' hashes) is rolled along a record
// by the merge kernel.  What travels is one 32-byte record per run of up to 32 windows that share their minimizer:
//
//   r0.x  level 1 -> 2 (k_skl_extract): position of the first window relative to the workgroup's first base (the compact form
//         of count_pipeline.h, with all 32 bits for the position); level 2 -> merge: the read pointer of the first window
//   r0.y  windows - 1 | (level 1 -> 2 only) the top 24 bits of the record's bin word << 8: the second level takes its leaf from
//         them (the table may have been replaced by one of another size since the first level ran: mcgpu.hip pipe_resize_by_sample),
//         and so that everybody agrees on a key's region, the REGION of a hash key in minimizer bins is worked out from those 24
//         bits alone wherever it is worked out (skl_bin)
//   r0.w:r0.z, r1.y:r1.x, r1.w:r1.z   the run's bases (windows + k - 1 <= 96), first base on top, unused tail zero
//
// About 16 windows a record at k = 63 on reads with 1 % errors: the streams shrink ~8-fold.  A bare key does not say where it lives
// in such a table: mc_get answers by one sweep of the table (mcgpu.hip k_getq_*), and before key streams of other ranks, the direct
// kernel or a shard export mcgpu.hip moves the table to hash-prefix regions (by_key_ready).
#pragma once
#include "count_pipeline.h"
#include "dup_check.h"

namespace mc {

constexpr uint32_t SKL_MAX_WINDOWS = 32;
constexpr uint32_t P1L_LANES = 57;                      // lanes of a wave's tile that own windows; the 7 above them only supply SK_M-mer hashes
constexpr uint32_t P1L_TILE = P1L_LANES * PT_ITEMS;     // 456 base positions per wave tile
constexpr int SKL_MIN_K = 33, SKL_MAX_K = 63;           // (two-word k-mers; mc_create takes hash keys up to k = 63)
static_assert((SKL_MAX_K - SK_M) / 8 + 1 <= 64 - (int)P1L_LANES, "k_skl_extract: a window reaches at most that many lanes up");

// the bin word of a hash key's minimizer: sk_bin with its low byte cleared (see the record's second word)
__host__ __device__ __forceinline__ uint32_t skl_bin(uint32_t hmin) { return sk_bin(hmin) & 0xFFFFFF00u; }

// TWO SMALLEST (TableView::mm_k < 0, mcgpu.hip mc_create): the word that picks a key's bin is made of the two smallest sk_order
// values among the window's canonical SK_M-mers (as a multiset: a value that occurs twice is both) instead of the smallest alone.
// Runs of windows that share it are two thirds as long, and a region holds the k-mers of half again as many, smaller stretches of
// the genome: scripts/bin_model.py -- at load 0.53 the fullest of 4 200 regions is 87 % full (116 % with the smallest alone, 1.6 % of
// the regions above 90 %), which is what lets configs[2] at FULL size (4.6 G keys in the 2^21 regions the merge kernel takes) travel
// as long records at all.  Either strand of a k-mer holds the same multiset, so the word is a function of the key.
__host__ __device__ __forceinline__ uint32_t skl_word2(uint32_t lo, uint32_t hi) { return lo ^ (hi * 0x85EBCA6Bu); }
// (lo, hi) <- the two smallest of (lo, hi) and (blo, bhi)
__device__ __forceinline__ void skl_merge2(uint32_t &lo, uint32_t &hi, uint32_t blo, uint32_t bhi)
{
    const uint32_t t = max(lo, blo);
    lo = min(lo, blo);
    hi = min(t, min(hi, bhi));
}

struct SklSpill {
    uint4 *recs;  // two per record
    unsigned long long *count;
    uint64_t cap;
    uint32_t *lost;
};
__device__ __forceinline__ void skl_spill_push(const SklSpill &sp, const uint4 &r0, const uint4 &r1)
{
    const unsigned long long i = atomicAdd(sp.count, 1ull);
    if (i < sp.cap) { sp.recs[2 * i] = r0; sp.recs[2 * i + 1] = r1; } else atomicExch(sp.lost, 1u);
}

// base i (0 = first) of a record's 96 bases X0:X1:X2
__device__ __forceinline__ uint32_t skl_base(uint64_t X0, uint64_t X1, uint64_t X2, uint32_t i)
{
    const uint64_t w = i < 32 ? X0 : (i < 64 ? X1 : X2);
    return (uint32_t)(w >> (62u - 2u * (i & 31u))) & 3u;
}
// the k-mer (hi:lo, right-aligned) that starts at base 0 of X0:X1 (32 < k <= 64)
__device__ __forceinline__ Kmer skl_first_kmer(uint64_t X0, uint64_t X1, int k)
{
    Kmer v;
    const int s = 128 - 2 * k;  // 0 .. 62
    if (s == 0) { v.hi = X0; v.lo = X1; }
    else { v.hi = X0 >> s; v.lo = (X1 >> s) | (X0 << (64 - s)); }
    return v;
}
// one step along the read for both strands' hashes (kmer_device.h key_poly): the window loses base `out` in front and gains `in`
// behind.  fw = 5^k + sum b_i 5^(k-1-i)  ->  5 fw - (4 + out) 5^k + in;   rc = 5^k + sum (3 - b_j) 5^j  ->  (rc - (3 - out)) / 5 +
// (4 + (3 - in)) 5^(k-1), the division exact, i.e. a multiplication by 5^-1 mod 2^64.
constexpr uint64_t POLY_INV5 = 0xCCCCCCCCCCCCCCCDull;
static_assert(POLY_INV5 * 5ull == 1ull, "5^-1 mod 2^64");
__device__ __forceinline__ void poly_roll(uint64_t &hf, uint64_t &hr, uint32_t out, uint32_t in, uint64_t p_k, uint64_t p_km1)
{
    hf = hf * 5ull - poly_4x_times(out, p_k) + in;
    hr = (hr - (3u ^ out)) * POLY_INV5 + poly_4x_times(3u ^ in, p_km1);
}
__host__ __device__ inline uint64_t pow5(int e) { uint64_t p = 1; for (int i = 0; i < e; i++) p *= 5ull; return p; }

// pointer of the window j <= 31 bases behind the one `aux` names (kmer_device.h ptr_advance serves 16-window records)
__device__ __forceinline__ uint32_t ptr_advance_long(uint32_t aux, uint32_t j)
{
    if (aux == 0) return 0;
    const uint64_t v = (uint64_t)aux - 1;
    if (v + SKL_MAX_WINDOWS < PTR_EXACT_END) return aux + j;
    uint32_t span;
    return ptr_encode(ptr_decode(aux, &span) + j);  // (a granule: its first base + j names a granule at most one short of the window's)
}

// ---------------------------------------------------------------------------------------------------------------
// Level 1: k_sk1w_extract (count_pipeline.h) for windows of 33 .. 64 bases, compact records only.  What differs:
//   * a window holds w = k - SK_M + 1 = 19 .. 50 SK_M-mers, reaching up to seven lanes up: a lane hashes the 8 SK_M-mers at its
//     own positions, leaves their running minima pre[0..7] in the wave's staging area, and a window's minimizer is
//     min(own suffix, the whole blocks between -- one value a lane, read across lanes --, the prefix of the block it ends in);
//     so 57 lanes of a tile own windows and the tile is 456 positions;
//   * "no read starts inside the window" looks 63 positions ahead: the doubling shifts run on 128 bits;
//   * a run is cut every 32 windows and its record holds up to 96 bases in three words.
struct alignas(16) Sk1lLds {  // Sk1wLds (count_pipeline.h) + a second staging array for the second-smallest hashes
    uint32_t wcur[PT_MAX_BUCKETS1_SK];
    uint32_t starts[P1W_WAVES][24];
    uint32_t brk[P1W_WAVES][20];
    uint64_t wst[P1W_WAVES][24];
    uint32_t hst[P1W_WAVES][64 * PT_ITEMS];
    uint16_t squeue[P1W_WAVES][64 * PT_ITEMS];
    uint32_t hst2[P1W_WAVES][64 * PT_ITEMS];
};
__global__ void __launch_bounds__(P1W_THREADS, MC_P1W_MIN_WAVES) k_skl_extract(
    const uint64_t *__restrict__ words, const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t base_lo,
    uint64_t n_bases, uint64_t n_tiles, const uint32_t *__restrict__ first_read, int k, uint32_t np1, uint32_t *seg_counts,
    uint64_t cap, uint4 *out_recs, SklSpill sp, uint32_t chunk_tiles, uint32_t two)
{   // two: the bin word from the two smallest hashes of a window (skl_word2)
    __shared__ Sk1lLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    for (uint32_t i = tid; i < PT_MAX_BUCKETS1_SK; i += P1W_THREADS) L.wcur[i] = 0;
    uint32_t *starts = L.starts[wv];
    uint32_t *brkw = L.brk[wv];
    uint8_t *brkb = reinterpret_cast<uint8_t *>(brkw);
    uint64_t *wst = L.wst[wv];
    uint32_t *hst = L.hst[wv], *hst2 = L.hst2[wv];
    uint16_t *squeue = L.squeue[wv];
    if (lane < 24) wst[lane] = 0;
    if (lane < 20) brkw[lane] = lane == 0 ? 0u : 0xFFFFFFFFu;  // (bytes 2 .. 65 are rewritten by every tile)
    __syncthreads();
    const uint32_t wm1 = (uint32_t)k - SK_M;            // SK_M-mers of a window - 1 (18 .. 49)
    const uint32_t qd = wm1 >> 3, rd = wm1 & 7u;        // the window of position j ends rd + j SK_M-mers into the block qd lanes up
    const uint64_t last_word = (n_bases + 31) / 32;     // the pad word
    const uint64_t seg_base = (uint64_t)blockIdx.x * cap, bucket_stride = (uint64_t)gridDim.x * cap;

    uint64_t pfW0 = 0, pfW1 = 0, pfW2 = 0, pf_off = ~0ull;
    uint32_t pf_first = 0;
    auto prefetch = [&](uint64_t tile) {
        if (tile >= n_tiles) return;
        const int64_t p0 = (int64_t)(tile * P1L_TILE) + (int64_t)lane * PT_ITEMS;
        const int64_t wi0 = (p0 - 7) >> 5;  // (-1 for the first lane of tile 0: that word reads as zero)
        pfW0 = wi0 >= 0 ? words[min((uint64_t)wi0, last_word)] : 0ull;
        pfW1 = words[min((uint64_t)(wi0 + 1), last_word)];
        pfW2 = words[min((uint64_t)(wi0 + 2), last_word)];
        pf_first = first_read[tile];
        const uint64_t r = (uint64_t)pf_first + lane;
        pf_off = r < n_reads ? offsets[r] : ~0ull;
    };
    const uint64_t chunk_lo = base_lo / P1L_TILE + (uint64_t)blockIdx.x * chunk_tiles;
    const uint64_t tile_end = min(n_tiles, chunk_lo + chunk_tiles);
    prefetch(chunk_lo + wv);
    for (uint64_t tile = chunk_lo + wv; tile < tile_end; tile += P1W_WAVES) {
        const uint64_t lo = tile * (uint64_t)P1L_TILE;
        const int64_t bm_lo = (int64_t)lo - 64;
        const uint64_t bm_hi = lo + 640;
        const uint64_t W0 = pfW0, W1 = pfW1, W2 = pfW2;
        uint64_t s = pf_off;
        const uint32_t my_first = pf_first;
        prefetch(tile + P1W_WAVES);
        const uint64_t p0 = lo + (uint64_t)lane * PT_ITEMS;
        const uint32_t off0 = (uint32_t)((int64_t)p0 - 32 * (((int64_t)p0 - 7) >> 5));  // base p0 inside W0:W1:W2 (8 .. 38)

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
        uint64_t S_lo, S_hi;  // bit b <-> a read starts at position p0 + 1 + b
        {
            const uint32_t bit0 = lane * 8 + 56, wd = bit0 >> 5, sh = bit0 & 31;
            const uint32_t d0 = starts[wd], d1 = starts[wd + 1], d2 = starts[wd + 2], d3 = starts[wd + 3], d4 = starts[wd + 4];
            const uint32_t ws0 = __builtin_amdgcn_alignbit(d1, d0, sh), ws1 = __builtin_amdgcn_alignbit(d2, d1, sh);
            const uint32_t ws2 = __builtin_amdgcn_alignbit(d3, d2, sh), ws3 = __builtin_amdgcn_alignbit(d4, d3, sh);
            const uint64_t a = ((uint64_t)ws1 << 32) | ws0, b = ((uint64_t)ws3 << 32) | ws2;  // bit b of b:a <-> a read starts at p0 - 8 + b
            S_lo = (a >> 9) | (b << 55);
            S_hi = b >> 9;
        }

        // ---- sk_order of the canonical SK_M-mers at my 8 positions; running minima from both ends
        uint32_t hh[PT_ITEMS], suf[PT_ITEMS], hmin[PT_ITEMS];
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
            if (!two) {
                uint32_t pre[PT_ITEMS];
                pre[0] = hh[0];
#pragma unroll
                for (int i = 1; i < 8; i++) pre[i] = min(pre[i - 1], hh[i]);
                suf[7] = hh[7];
#pragma unroll
                for (int j = 6; j >= 0; j--) suf[j] = min(suf[j + 1], hh[j]);
                *reinterpret_cast<uint4 *>(&hst[lane * PT_ITEMS]) = make_uint4(pre[0], pre[1], pre[2], pre[3]);
                *reinterpret_cast<uint4 *>(&hst[lane * PT_ITEMS + 4]) = make_uint4(pre[4], pre[5], pre[6], pre[7]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t m8 = pre[7];
                uint32_t M = SK_NONE;  // the whole blocks strictly between mine and the one the windows end in
                for (uint32_t d = 1; d < qd; d++) M = min(M, (uint32_t)__shfl_down((int)m8, d));
                const uint32_t m8A = (uint32_t)__shfl_down((int)m8, qd);
                const uint32_t at = (lane + qd) * PT_ITEMS + rd;
#pragma unroll
                for (int j = 0; j < PT_ITEMS; j++) {
                    uint32_t x = hst[min(at + (uint32_t)j, 64u * PT_ITEMS - 1u)];  // (the lanes that own no window read whatever is there)
                    if (rd + (uint32_t)j >= 8u) x = min(x, m8A);
                    hmin[j] = min(min(suf[j], M), x);
                }
            } else {
                // the same with PAIRS: the two smallest of every prefix and suffix of my 8 hashes, of the whole blocks, of the end block
                uint32_t plo[PT_ITEMS], phi[PT_ITEMS], suh[PT_ITEMS];
                plo[0] = hh[0]; phi[0] = SK_NONE;
#pragma unroll
                for (int i = 1; i < 8; i++) { plo[i] = plo[i - 1]; phi[i] = phi[i - 1]; skl_merge2(plo[i], phi[i], hh[i], SK_NONE); }
                suf[7] = hh[7]; suh[7] = SK_NONE;
#pragma unroll
                for (int j = 6; j >= 0; j--) { suf[j] = suf[j + 1]; suh[j] = suh[j + 1]; skl_merge2(suf[j], suh[j], hh[j], SK_NONE); }
                *reinterpret_cast<uint4 *>(&hst[lane * PT_ITEMS]) = make_uint4(plo[0], plo[1], plo[2], plo[3]);
                *reinterpret_cast<uint4 *>(&hst[lane * PT_ITEMS + 4]) = make_uint4(plo[4], plo[5], plo[6], plo[7]);
                *reinterpret_cast<uint4 *>(&hst2[lane * PT_ITEMS]) = make_uint4(phi[0], phi[1], phi[2], phi[3]);
                *reinterpret_cast<uint4 *>(&hst2[lane * PT_ITEMS + 4]) = make_uint4(phi[4], phi[5], phi[6], phi[7]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t tl = plo[7], th = phi[7];
                uint32_t Ml = SK_NONE, Mh = SK_NONE;
                for (uint32_t d = 1; d < qd; d++) skl_merge2(Ml, Mh, (uint32_t)__shfl_down((int)tl, d), (uint32_t)__shfl_down((int)th, d));
                const uint32_t Al = (uint32_t)__shfl_down((int)tl, qd), Ah = (uint32_t)__shfl_down((int)th, qd);
                const uint32_t at = (lane + qd) * PT_ITEMS + rd;
#pragma unroll
                for (int j = 0; j < PT_ITEMS; j++) {
                    const uint32_t idx = min(at + (uint32_t)j, 64u * PT_ITEMS - 1u);
                    uint32_t xl = hst[idx], xh = hst2[idx];
                    if (rd + (uint32_t)j >= 8u) skl_merge2(xl, xh, Al, Ah);
                    uint32_t rl = suf[j], rh = suh[j];
                    skl_merge2(rl, rh, Ml, Mh);
                    skl_merge2(rl, rh, xl, xh);
                    hmin[j] = skl_word2(rl, rh);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the staging area now takes the windows' minimizers)
        }

        // ---- which of my positions start a window, and where runs of equal minimizers break
        uint32_t valid_bits = 0;
        if (lane < P1L_LANES && p0 + (uint64_t)k <= n_bases) {
            // window j lies inside one read iff no read starts at p0 + j + 1 .. p0 + j + k - 1: bit j of the OR of S >> 0 .. k - 2
            auto shr_or = [&](uint32_t n) {
                if (n == 0) return;
                S_lo |= (S_lo >> n) | (S_hi << (64u - n));
                S_hi |= S_hi >> n;
            };
            uint32_t width = 1;
            for (; 2 * width <= (uint32_t)k - 1; width *= 2) shr_or(width);  // (uniform: five or six rounds)
            shr_or((uint32_t)k - 1 - width);
            const uint64_t jmax = n_bases - (uint64_t)k - p0;  // last j with p0 + j + k <= n_bases
            const uint32_t jm = jmax < 7 ? (uint32_t)jmax : 7u;
            const uint32_t j0 = p0 >= base_lo ? 0u : (base_lo - p0 < 8 ? (uint32_t)(base_lo - p0) : 8u);  // first j with p0 + j >= base_lo
            valid_bits = ~(uint32_t)S_lo & ((2u << jm) - 1u) & ~((1u << j0) - 1u) & 0xFFu;
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
        uint32_t start_bits = 0;  // a window starts a record when it breaks the run or sits a multiple of 32 windows behind the run's first
        {
            const unsigned long long holders = __ballot(brk_bits != 0);  // (lane 0 is one: a tile always starts a run)
            const unsigned long long below = holders & ((1ull << lane) - 1ull);
            const uint32_t P = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
            const uint32_t bp = (uint32_t)__shfl((int)brk_bits, (int)P);
            uint32_t cur = P * PT_ITEMS + (31u - (uint32_t)__builtin_clz(bp | 1u));
#pragma unroll
            for (int j = 0; j < PT_ITEMS; j++) {
                const uint32_t pos = lane * PT_ITEMS + (uint32_t)j;
                cur = (brk_bits >> j) & 1u ? pos : cur;
                if (((valid_bits >> j) & 1u) && ((pos - cur) % SKL_MAX_WINDOWS) == 0) start_bits |= 1u << j;
            }
        }

        // ---- records: the tile's starts lined up, lane i builds the i-th record (k_sk1w_extract says why)
        {
            const uint32_t cnt = (uint32_t)__builtin_popcount(start_bits);
            const uint32_t incl = wave_incl_sum(cnt);
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
                const uint32_t b0 = 17u + wi;                                    // break bits of the 31 windows behind it
                const uint32_t ahead = __builtin_amdgcn_alignbit(brkw[(b0 >> 5) + 1], brkw[b0 >> 5], b0 & 31u) & 0x7FFFFFFFu;
                const uint32_t n = ahead ? (uint32_t)__builtin_ctz(ahead) + 1u : SKL_MAX_WINDOWS;
                const uint32_t len = n + (uint32_t)k - 1;  // bases of the run (33 .. 95)
                const uint32_t q = wi + 7u + r7, wq = q >> 5, sh = 2u * (q & 31u);
                const uint64_t A = wst[wq], B = wst[wq + 1], C = wst[wq + 2], D = wst[wq + 3];
                const uint64_t X0 = (A << sh) | ((B >> 1) >> (63u - sh));
                uint64_t X1 = (B << sh) | ((C >> 1) >> (63u - sh)), X2 = (C << sh) | ((D >> 1) >> (63u - sh));
                if (len < 64) { X1 &= ~0ull << (2 * (64 - len)); X2 = 0; }   // (len >= 33)
                else if (len == 64) X2 = 0;
                else if (len < 96) X2 &= ~0ull << (2 * (96 - len));
                const uint32_t hsel = hst[wi];
                const uint32_t bin = skl_bin(hsel);
                const uint32_t d = mulhi32(bin, np1);
                const uint32_t rel = (uint32_t)(tile - chunk_lo) * P1L_TILE + wi;
                uint4 r0, r1;
                r0.x = rel;
                r0.y = (n - 1) | bin;
                r0.z = (uint32_t)X0; r0.w = (uint32_t)(X0 >> 32);
                r1.x = (uint32_t)X1; r1.y = (uint32_t)(X1 >> 32); r1.z = (uint32_t)X2; r1.w = (uint32_t)(X2 >> 32);
                const uint64_t dst = atomicAdd(&L.wcur[d], 1u);
                if (dst < cap) {
                    const uint64_t o = seg_base + (uint64_t)d * bucket_stride + dst;
                    out_recs[2 * o] = r0;
                    out_recs[2 * o + 1] = r1;
                } else {
                    skl_spill_push(sp, r0, r1);
                }
            }
            __builtin_amdgcn_wave_barrier();  // (the next tile rewrites the staging area)
        }
    }
    __syncthreads();
    for (uint32_t d = tid; d < np1; d += P1W_THREADS)  // how much of its segment of every bucket this workgroup filled
        seg_counts[(uint64_t)d * gridDim.x + blockIdx.x] = (uint32_t)min((uint64_t)L.wcur[d], cap);
}

// ---------------------------------------------------------------------------------------------------------------
// Merge: one workgroup per leaf = table region (count_pipeline.h k_p3_merge, whose layout, probing, pointer rule, hand-on list
// and solid list these are).  A lane takes a record, starts both strands' hashes of its first window four bases at a time from
// two 256-entry LDS tables (kmer_device.h poly_hashes_tabled) and rolls them along its windows.
#ifndef MC_SKL_CHUNK
#define MC_SKL_CHUNK 8   // windows a lane of the merge kernel takes of a record
#endif
constexpr uint32_t SKL_CHUNK = MC_SKL_CHUNK, SKL_CPR = SKL_MAX_WINDOWS / SKL_CHUNK;  // chunks (lanes) per record
static_assert(SKL_CPR * SKL_CHUNK == SKL_MAX_WINDOWS && SKL_CHUNK * 2 * (SKL_CPR - 1) < 64, "a chunk starts less than a word into the record");
constexpr uint32_t SKL_ROUND = P3_THREADS;           // records at hand at a time: one a thread (merged where equal, their chunks lined up)
constexpr uint32_t SKL_DTAB = 1024;                  // slots of the table that finds equal records
constexpr uint32_t SKL_TASKS = SKL_ROUND * SKL_CPR;
struct alignas(16) LongLds {
    uint64_t key[REGION_SLOTS];
    uint32_t cnt[REGION_SLOTS];
    uint32_t aux[REGION_SLOTS];
    uint16_t polyF[256], polyR[256];
    uint16_t tasks[SKL_TASKS];          // record (of the 512 at hand) << 2 | chunk
    uint32_t dtab[SKL_DTAB];            // 0: free; fingerprint << 16 | record + 1
    uint16_t copies[SKL_ROUND];         // of a record that stands for its equals, itself included
    uint32_t n_new, overflow, emit_cur, n_empty;
    uint32_t n_tasks, pad_[3];
    uint32_t dcur[DUP_B1];              // fill levels of this workgroup's segments of the key stream (dup_check.h level 1)
};
static_assert(2 * sizeof(LongLds) <= 160 * 1024, "two workgroups of the long merge kernel on a CU");

// both strands' hashes of the k-mer at the top of X0:X1 (kmer_device.h poly_hashes_tabled, whose values these are): the bytes
// of either strand sit at fixed places of X0:X1 and of the right-aligned k-mer, so the sixteen steps are unrolled with constant
// shifts instead of moving two 128-bit words along (a third of what a chunk of 8 windows costs was this start-up)
__device__ __forceinline__ void poly_start(uint64_t X0, uint64_t X1, uint64_t X2, int k, const uint16_t *polyF, const uint16_t *polyR, uint64_t *hf_out,
                                           uint64_t *hr_out)
{
    const int n4 = k >> 2, rem = k & 3;
    const Kmer v = skl_first_kmer(X0, X1, k);
    uint64_t hf = 1, hr = 1;
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i < n4) {  // (uniform)
            const uint32_t fb = (uint32_t)((i < 8 ? X0 : X1) >> (56 - 8 * (i & 7))) & 0xFFu;
            const uint32_t rb = (uint32_t)((i < 8 ? v.lo : v.hi) >> (8 * (i & 7))) & 0xFFu;
            hf = hf * 625ull + polyF[fb];
            hr = hr * 625ull + polyR[rb];
        }
    for (int e = 0; e < rem; e++) {
        hf = hf * 5ull + skl_base(X0, X1, X2, (uint32_t)(4 * n4 + e));
        hr = hr * 5ull + (3u ^ skl_base(X0, X1, X2, (uint32_t)(rem - 1 - e)));
    }
    *hf_out = hf;
    *hr_out = hr;
}

// KT: the k-mer length when the kernel is built for one (63: configs[2]) -- shifts, trip counts and the powers of 5 are constants
// then -- or 0: any length, from the argument.
template <int KT>
__global__ void __launch_bounds__(P3_THREADS) k_p3_long(const uint4 *__restrict__ leaf_recs, const uint32_t *__restrict__ leaf_counts, uint64_t cap,
                                                        uint32_t n_leaves, TableView t, int virgin, uint32_t *leaf_state, uint32_t *leaf_new,
                                                        uint32_t *any_failed, uint32_t solid_thr, unsigned long long *n_solid, int k_arg, P3Emit emit,
                                                        const uint32_t *lost, DupL1 d1)
{   // d1: every key that goes back to HBM is also appended to the workgroup's segment of its key bucket (dup_check.h: the join by
    // key that finds a key two different k-mers brought to two regions starts from these streams instead of a sweep of the table)
    if (lost && *lost) return;  // (the scatter levels lost records: the batch is counted another way)
    const int k = KT ? KT : k_arg;
    __shared__ LongLds L;
    const uint32_t tid = threadIdx.x;
    if (d1.keys && tid < DUP_B1) L.dcur[tid] = d1.counts[tid * d1.nseg + blockIdx.x];  // (a launch goes on where the last one stopped; published by the barriers below)
    const bool emitting = emit.recs != nullptr && solid_thr != 0;
    if (tid < 256) poly_tables_fill(L.polyF, L.polyR, tid);
    if (tid == 0) L.emit_cur = emitting ? emit.counts[blockIdx.x] : 0u;  // (published by the first barrier below)
    const uint64_t p_k = pow5(k), p_km1 = pow5(k - 1);
    const uint32_t key_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint64_t *)L.key;
    const uint32_t ptr_from = solid_thr >= 2 ? 1u : 0u;
    long long solid_delta = 0;
    unsigned long long empty_total = 0;  // (thread 0) occurrences of the key that looks like a free slot, in committed leaves
    for (uint32_t leaf = blockIdx.x; leaf < n_leaves; leaf += gridDim.x) {
        if (leaf_state[leaf]) continue;  // uniform
        Slot *gs = t.slots + (uint64_t)leaf * REGION_SLOTS;
        const uint32_t n = min(leaf_counts[leaf], (uint32_t)cap);
        const uint4 *recs = leaf_recs + 2 * (uint64_t)leaf * cap;
        // a thread's own record of the first 512: requested before the region image is set up
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
        if (tid < n) { q0 = recs[2 * tid]; q1 = recs[2 * tid + 1]; }
        int solid_before = 0;
        uint32_t occ = 0;  // occupied slots when the region goes back - when it came: the keys this leaf added
        for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
            if (virgin) {
                L.key[i] = EMPTY_KEY; L.cnt[i] = 0; L.aux[i] = 0;
            } else {
                const uint4 raw = *reinterpret_cast<const uint4 *>(gs + i);
                occ -= (raw.x & raw.y) != 0xFFFFFFFFu;
                L.key[i] = ((uint64_t)raw.y << 32) | raw.x;
                L.cnt[i] = min(raw.z, P3_COUNT_CAP);
                L.aux[i] = raw.w;
                solid_before += solid_thr && raw.z >= solid_thr;  // (empty slots hold count 0)
            }
        }
        for (uint32_t i = tid; i < SKL_DTAB; i += P3_THREADS) L.dtab[i] = 0;
        L.copies[tid] = 1;
        if (tid == 0) { L.n_new = 0; L.overflow = 0; L.n_empty = 0; L.n_tasks = 0; }
        __syncthreads();
        uint32_t my_empty = 0;
        // A lane takes a CHUNK of SKL_CHUNK windows of a record.  With a lane a record three of the eight waves had work, for 32
        // steps, and -- what cost most -- the ~30 reads that cover a locus give as many near-identical records, whose lanes came to
        // the SAME key at the same step: every compare-and-swap and every count addition ten lanes deep on one LDS address (20 ms
        // for configs[2]'s 10 M reads against 6 for the per-window merge).  The chunks that hold windows are lined up first (one
        // LDS atomic a record for its place: chunks of a record next to each other, so the lanes of a wave mostly hold different keys).
        for (uint32_t rbase = 0; rbase < n; rbase += SKL_ROUND) {
            // Equal records first: a read that covers a locus's run of windows whole and without an error yields the same record as
            // every other such read -- 0.99^94 = 39 % of configs[2]'s records, a dozen of each -- and one of them stands for all,
            // each of its windows adding their number.  A thread registers its record's fingerprint in a small table with one
            // compare-and-swap; who finds the fingerprint there compares the records word for word (a record that finds neither
            // its equal nor a free slot in four probes stays on its own: exact either way).
            if (rbase) {
                q0 = make_uint4(0, 0, 0, 0); q1 = q0;
                if (rbase + tid < n) { q0 = recs[2 * (rbase + tid)]; q1 = recs[2 * (rbase + tid) + 1]; }
            }
            const bool have = rbase + tid < n;
            const uint32_t nw_own = have ? (q0.y & 0xFFu) + 1u : 0u;
            uint32_t cand = 0xFFFFFFFFu;  // the record mine may be a copy of
            if (have) {
                uint32_t h = q0.z * 0x9E3779B1u ^ q0.w;
                h = (h ^ (h >> 15)) * 0x85EBCA6Bu ^ q1.x;
                h = (h ^ (h >> 13)) * 0xC2B2AE35u ^ q1.y;
                h = (h ^ (h >> 16)) * 0x27D4EB2Fu ^ q1.z;
                h = (h ^ (h >> 15)) * 0x165667B1u ^ q1.w ^ nw_own;
                h ^= h >> 16;
                const uint32_t mine_e = (h & 0xFFFF0000u) | (tid + 1u);
                uint32_t sl = h & (SKL_DTAB - 1u);
#pragma unroll
                for (int pr = 0; pr < 4; pr++) {
                    const uint32_t old = atomicCAS(&L.dtab[sl], 0u, mine_e);
                    if (old == 0u) break;  // registered
                    if ((old & 0xFFFF0000u) == (mine_e & 0xFFFF0000u)) { cand = (old & 0xFFFFu) - 1u; break; }
                    sl = (sl + 1u) & (SKL_DTAB - 1u);
                }
            }
            bool copy = false;
            if (cand != 0xFFFFFFFFu) {
                const uint4 o0 = recs[2 * (rbase + cand)], o1 = recs[2 * (rbase + cand) + 1];
                copy = ((o0.y ^ q0.y) & 0xFFu) == 0 && o0.z == q0.z && o0.w == q0.w && o1.x == q1.x && o1.y == q1.y && o1.z == q1.z && o1.w == q1.w;
            }
            if (copy) atomicAdd(reinterpret_cast<uint32_t *>(&L.copies[cand & ~1u]), (cand & 1u) ? 0x10000u : 1u);  // (16-bit counters, two a word; < 512 copies)
            {   // (the barrier behind the task list also stands between these additions and the tasks that read them)
                const uint32_t nch = copy ? 0u : (nw_own + SKL_CHUNK - 1) / SKL_CHUNK;
                if (nch) {
                    const uint32_t at = atomicAdd(&L.n_tasks, nch);
                    for (uint32_t c = 0; c < nch; c++) L.tasks[at + c] = (uint16_t)((tid << 2) | c);
                }
            }
            __syncthreads();
            const uint32_t nt = L.n_tasks;
            for (uint32_t T = tid; T - tid < nt; T += P3_THREADS) {  // (uniform trip count)
                const bool mine = T < nt;
                if (!__ballot(mine)) continue;
                const uint32_t task = mine ? L.tasks[T] : 0u;
                const uint32_t r = rbase + (task >> 2), j0 = (task & 3u) * SKL_CHUNK;
                uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
                if (mine) { r0 = recs[2 * r]; r1 = recs[2 * r + 1]; }
                const uint32_t nw = mine ? (r0.y & 0xFFu) + 1u : 0u, p0 = r0.x;
                const uint32_t cp = mine ? L.copies[task >> 2] : 0u;
                const uint32_t cnt = j0 < nw ? min(SKL_CHUNK, nw - j0) : 0u;
                uint64_t X0 = ((uint64_t)r0.w << 32) | r0.z, X1 = ((uint64_t)r1.y << 32) | r1.x, X2 = ((uint64_t)r1.w << 32) | r1.z;
                if (j0) {  // the bases from the chunk's first window on (j0 = 8, 16, 24)
                    const uint32_t sh = 2 * j0;
                    X0 = (X0 << sh) | (X1 >> (64 - sh));
                    X1 = (X1 << sh) | (X2 >> (64 - sh));
                    X2 <<= sh;
                }
                // the bases that leave (from the chunk's first on) and the ones that enter (from base k on) at the top of a word each
                uint64_t outs = X0, ins;
                {
                    const uint32_t sh = 2u * ((uint32_t)k - 32u);  // (k > 32: base k lies in X1, or is X2's first)
                    ins = sh < 64u ? ((X1 << sh) | ((X2 >> 1) >> (63u - sh))) : X2;
                }
                uint64_t hf = 0, hr = 0;
                if (cnt) poly_start(X0, X1, X2, k, L.polyF, L.polyR, &hf, &hr);
                // the pointer of the chunk's first window, once; window j of the chunk is j bases on where pointers name places, and
                // within the slack of the granule where they name granules (kmer_device.h ptr_advance: the same rule)
                const uint32_t pbase = ptr_advance_long(p0, j0);
                const uint32_t pstep = pbase != 0 && (uint64_t)pbase - 1 + SKL_MAX_WINDOWS < PTR_EXACT_END ? 0xFFFFFFFFu : 0u;
#pragma unroll 1
                for (uint32_t j = 0; j < SKL_CHUNK; j++) {
                    const bool act = j < cnt;
                    if (!__ballot(act)) break;
                    const uint64_t key = (int64_t)hf < (int64_t)hr ? hf : hr;  // (Math.min on signed longs)
                    const bool odd_key = key == EMPTY_KEY;  // (the key that looks like a free slot: counted apart)
                    const bool ins_it = act && !odd_key;
                    const uint32_t home = sk_home(key);
                    // which of the key's occurrences leaves its pointer (kmer_device.h ptr_pick): worked out for every lane, so that what
                    // follows the count's addition is one comparison and one masked store (as nested branches the rule was a quarter
                    // of the loop's instructions, most of them the compiler's bookkeeping of who is in and who is out)
                    const uint32_t pk = ptr_pick(key, ptr_from, solid_thr);
                    unsigned long long old0 = EMPTY_KEY;
                    if (ins_it) old0 = atomicCAS(reinterpret_cast<unsigned long long *>(&L.key[home]), (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                    poly_roll(hf, hr, (uint32_t)(outs >> 62), (uint32_t)(ins >> 62), p_k, p_km1);  // (one step past the last window does no harm)
                    outs <<= 2;
                    ins <<= 2;
                    my_empty += act && odd_key ? cp : 0u;
                    uint32_t s = home;
                    bool fits = ins_it;
                    if (ins_it && old0 != EMPTY_KEY && old0 != key) {  // on from the slot behind, TABLE_MAX_PROBES slots in all (where look-ups look)
                        uint32_t unused = 0;
                        unsigned long long pending;
                        s = lds_probe_claim(key_base, (home + 1u) & (REGION_SLOTS - 1u), key, &unused, &pending, P3_MAX_PROBES - 1u);
                        fits = !((pending >> (tid & 63u)) & 1ull);
                    }
                    const uint32_t ptr = pbase + (j & pstep);
                    if (fits) {
                        const uint32_t seen = atomicAdd(&L.cnt[s], cp);
                        if (pbase != 0 && seen <= pk && pk < seen + cp) L.aux[s] = ptr;  // (the occurrence that leaves its pointer is one of these cp)
                    } else if (ins_it) {
                        if (!ovf_push(t, key, cp, ptr, leaf)) atomicExch(&L.overflow, 1u);
                    }
                }
            }
            if (rbase + SKL_ROUND < n) {  // (uniform) the list and the table of records are rewritten
                __syncthreads();
                for (uint32_t i = tid; i < SKL_DTAB; i += P3_THREADS) L.dtab[i] = 0;
                L.copies[tid] = 1;
                if (tid == 0) L.n_tasks = 0;
                __syncthreads();
            }
        }
        if (my_empty) atomicAdd(&L.n_empty, my_empty);
        __syncthreads();
        const bool ovf = L.overflow != 0;
        if (!ovf) {
            for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                uint4 v;
                const uint64_t kk = L.key[i];
                v.x = (uint32_t)kk; v.y = (uint32_t)(kk >> 32);
                v.z = min(L.cnt[i], P3_COUNT_CAP);
                v.w = L.aux[i];
                occ += kk != EMPTY_KEY;
                *reinterpret_cast<uint4 *>(gs + i) = v;
                if (d1.keys && kk != EMPTY_KEY) {  // (slots are in home-slot order: the lanes of a bucket sit side by side)
                    const uint32_t b = dup_b1(kk);
                    const uint32_t pos = atomicAdd(&L.dcur[b], 1u);
                    if (pos < d1.cap) d1.keys[((uint64_t)b * d1.nseg + blockIdx.x) * d1.cap + pos] = kk; else atomicExch(d1.lost, 1u);
                }
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
            const uint32_t wsum = wave_incl_sum(occ);  // (two's complement: a thread's own difference may be negative)
            if ((tid & 63u) == 63u) atomicAdd(&L.n_new, wsum);
            __syncthreads();
        } else if (virgin) {  // nothing was there: leave a valid empty region behind
            for (uint32_t i = tid; i < REGION_SLOTS; i += P3_THREADS) {
                uint4 v; v.x = 0xFFFFFFFFu; v.y = 0xFFFFFFFFu; v.z = 0; v.w = 0;
                *reinterpret_cast<uint4 *>(gs + i) = v;
            }
        }
        if (tid == 0) {
            if (!ovf) { leaf_state[leaf] = 1; leaf_new[leaf] = L.n_new; empty_total += L.n_empty; } else atomicExch(any_failed, 1u);
        }
        __syncthreads();
    }
    if (tid == 0 && empty_total) atomicAdd(t.empty_cnt, empty_total);
    if (solid_thr) wave_add_ull(n_solid, (unsigned long long)solid_delta);  // (two's complement: deltas may be negative)
    if (d1.keys && tid < DUP_B1) d1.counts[tid * d1.nseg + blockIdx.x] = (uint32_t)min((uint64_t)L.dcur[tid], d1.cap);  // (the leaf loop ends in a barrier)
    if (emitting) {
        __syncthreads();
        if (tid == 0) emit.counts[blockIdx.x] = (uint64_t)L.emit_cur < emit.seg_cap ? L.emit_cur : (uint32_t)emit.seg_cap;
    }
}

// home slot of window keys of a record in the table `t`: the record's minimizer bin while the table has minimizer bins
__device__ __forceinline__ uint64_t skl_region_base(const TableView &t, uint64_t X0, uint64_t X1, int k)
{
    const uint32_t hm = sk_hmin_of_kmer2(skl_first_kmer(X0, X1, k), k, t.mm_k < 0);  // (every window of a record has the same bin word)
    return (((uint64_t)skl_bin(hm) * t.n_regions) >> 32) << MC_REGION_LG;
}

// one record's windows through the direct path: the spill list of the scatter levels (n records at `recs`, pointerless), or the
// leaves the merge kernel left unmerged (below)
__device__ __forceinline__ void skl_add_record(const TableView &t, const uint4 &r0, const uint4 &r1, uint32_t p0, int k, uint64_t p_k, uint64_t p_km1,
                                               uint32_t solid_thr, unsigned long long &n_new, unsigned long long &n_cross, const DupL1 &d1)
{
    const uint32_t nw = (r0.y & 0xFFu) + 1u;
    const uint64_t X0 = ((uint64_t)r0.w << 32) | r0.z, X1 = ((uint64_t)r1.y << 32) | r1.x, X2 = ((uint64_t)r1.w << 32) | r1.z;
    const Kmer v = skl_first_kmer(X0, X1, k);
    uint64_t hf = 1, hr = 1;
    for (int i = 0; i < k; i++) {
        hf = hf * 5 + base_at(v, k, i);
        hr = hr * 5 + (3u ^ base_at(v, k, k - 1 - i));
    }
    const uint64_t rbase = t.mm_k ? skl_region_base(t, X0, X1, k) : 0;
    for (uint32_t j = 0; j < nw; j++) {
        const uint64_t key = (int64_t)hf < (int64_t)hr ? hf : hr;
        uint32_t before;
        const uint32_t fresh = table_add_at(t, key == EMPTY_KEY ? 0 : (t.mm_k ? (rbase | sk_home(key)) : slot_of(t, key)), key, 1u, ptr_advance_long(p0, j), &before);
        n_new += fresh;
        if (fresh) dup_l1_extra(d1, key);  // (a key the merge kernel's streams do not hold)
        n_cross += crosses(before, 1u, solid_thr);
        if (j + 1 < nw) poly_roll(hf, hr, skl_base(X0, X1, X2, j), skl_base(X0, X1, X2, j + (uint32_t)k), p_k, p_km1);
    }
}

__global__ void k_skl_add_records(const uint4 *__restrict__ recs, uint64_t n, int k, TableView t, uint32_t solid_thr, unsigned long long *n_solid, DupL1 d1)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t p_k = pow5(k), p_km1 = pow5(k - 1);
    unsigned long long n_new = 0, n_cross = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        skl_add_record(t, recs[2 * i], recs[2 * i + 1], 0u, k, p_k, p_km1, solid_thr, n_new, n_cross, d1);
    wave_add_ull(t.n_used, n_new);
    if (solid_thr) wave_add_ull(n_solid, n_cross);
}

// Distinct keys among the records of level-1 bucket 0 (count_pipeline.h k_sk_sample_distinct, for long records): the sample
// that sizes the table of a context without a capacity hint.
__global__ void __launch_bounds__(256) k_skl_sample_distinct(const uint4 *__restrict__ recs, const uint32_t *__restrict__ seg_counts, uint64_t seg_cap, int k,
                                                             uint64_t *set, uint64_t mask, unsigned long long *n_distinct)
{
    const uint32_t n = min(seg_counts[blockIdx.x], (uint32_t)seg_cap);
    const uint4 *seg = recs + 2 * (uint64_t)blockIdx.x * seg_cap;
    const uint64_t p_k = pow5(k), p_km1 = pow5(k - 1);
    unsigned long long n_new = 0;
    for (uint32_t r = threadIdx.x; r < n; r += 256) {
        const uint4 r0 = seg[2 * r], r1 = seg[2 * r + 1];
        const uint32_t nw = (r0.y & 0xFFu) + 1u;
        const uint64_t X0 = ((uint64_t)r0.w << 32) | r0.z, X1 = ((uint64_t)r1.y << 32) | r1.x, X2 = ((uint64_t)r1.w << 32) | r1.z;
        const Kmer v = skl_first_kmer(X0, X1, k);
        uint64_t hf = 1, hr = 1;
        for (int i = 0; i < k; i++) {
            hf = hf * 5 + base_at(v, k, i);
            hr = hr * 5 + (3u ^ base_at(v, k, k - 1 - i));
        }
        for (uint32_t j = 0; j < nw; j++) {
            const uint64_t key = (int64_t)hf < (int64_t)hr ? hf : hr;
            uint64_t s = fmix64(key) & mask;
            bool placed = key == ~0ull;
            for (int probe = 0; probe < 64 && !placed; probe++) {
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&set[s]), ~0ull, (unsigned long long)key);
                if (old == ~0ull) { n_new++; placed = true; }
                else if (old == key) placed = true;
                else s = (s + 1) & mask;
            }
            if (!placed) n_new++;
            if (j + 1 < nw) poly_roll(hf, hr, skl_base(X0, X1, X2, j), skl_base(X0, X1, X2, j + (uint32_t)k), p_k, p_km1);
        }
    }
    wave_add_ull(n_distinct, n_new);
}

// the records of the leaves [leaf_lo, leaf_hi) that k_p3_long left unmerged (leaf_state == 0), after the table has given up
// minimizer bins (mcgpu.hip to_hash_regions)
__global__ void k_skl_add_unmerged(const uint4 *__restrict__ leaf_recs, const uint32_t *__restrict__ leaf_counts, uint64_t cap, uint32_t leaf_lo,
                                   uint32_t leaf_hi, const uint32_t *__restrict__ leaf_state, int k, TableView t)
{
    const uint64_t p_k = pow5(k), p_km1 = pow5(k - 1);
    unsigned long long n_new = 0, n_cross = 0;
    for (uint32_t leaf = leaf_lo + blockIdx.x; leaf < leaf_hi; leaf += gridDim.x) {
        if (leaf_state[leaf]) continue;
        const uint64_t n = min((uint64_t)leaf_counts[leaf], cap);
        const uint4 *recs = leaf_recs + 2 * (uint64_t)leaf * cap;
        for (uint64_t r = threadIdx.x; r < n; r += blockDim.x) skl_add_record(t, recs[2 * r], recs[2 * r + 1], recs[2 * r].x, k, p_k, p_km1, 0u, n_new, n_cross, DupL1{nullptr, nullptr, 0, 0, nullptr});
    }
    wave_add_ull(t.n_used, n_new);
}

}  // namespace mc

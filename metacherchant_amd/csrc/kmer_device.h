// Device-side k-mer arithmetic for gfx950 (wave64).  Every function names the reference code
// whose result it must reproduce bit for bit (src/... = reference src/, itmo!/x =
// ru/ifmo/genetics/x in lib/itmo-assembler-src.jar).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

constexpr int KEY_PACKED = 0, KEY_POLY = 1, KEY_FNV1A = 2;
constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;  // never a packed key (those are < 2^62)

struct Kmer {  // oriented k-mer, 2k bits right-aligned in 128, first base most significant
    uint64_t hi, lo;
};

__host__ __device__ constexpr __forceinline__ uint64_t fmix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// SplitMix64, n-th output of the generator seeded with `seed` (DESIGN.md "Synthetic workload")
__host__ __device__ __forceinline__ uint64_t splitmix(uint64_t seed, uint64_t n)
{
    uint64_t z = seed + (n + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// itmo!/utils/KmerUtils.java:12-22 reverseComplement(kmer, k): reverse the 2-bit groups,
// complement, right-align.  v_bfrev reverses single bits, so swap the bits of each pair back.
__host__ __device__ __forceinline__ uint64_t rc_packed(uint64_t x, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r = __brevll(x);
#else
    uint64_t r = x;
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    r = ((r & 0x3333333333333333ull) << 2) | ((r >> 2) & 0x3333333333333333ull);
    r = ((r & 0x0f0f0f0f0f0f0f0full) << 4) | ((r >> 4) & 0x0f0f0f0f0f0f0f0full);
    r = ((r & 0x00ff00ff00ff00ffull) << 8) | ((r >> 8) & 0x00ff00ff00ff00ffull);
    r = ((r & 0x0000ffff0000ffffull) << 16) | ((r >> 16) & 0x0000ffff0000ffffull);
    r = (r << 32) | (r >> 32);
#endif
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    return (~r) >> (64 - 2 * k);
}

__host__ __device__ __forceinline__ uint32_t base_at(const Kmer &v, int k, int i)
{  // i-th base, 0 = first
    const int sh = 2 * (k - 1 - i);
    return (uint32_t)((sh >= 64 ? (v.hi >> (sh - 64)) : (v.lo >> sh)) & 3);
}

// All key functions also report `flipped`: whether the reverse-complement strand supplied the key
// (only used to orient the speculation hints stored next to the key; never part of a result).

// itmo!/dna/kmers/ShortKmer.java:54-56 toLong = Math.min(fwKmer, rcKmer) on signed longs
// (both are < 2^62 for k <= 31, so the unsigned min is the same number)
__host__ __device__ __forceinline__ int64_t key_packed(uint64_t fw, int k, bool *flipped = nullptr)
{
    const int64_t a = (int64_t)fw, b = (int64_t)rc_packed(fw, k);
    if (flipped) *flipped = b < a;
    return a < b ? a : b;
}

// key_poly's two hashes four bases at a time: h <- h * 5^4 + T[byte] with T[b0 b1 b2 b3] = 125 b0 + 25 b1 + 5 b2 + b3, so a
// k-mer costs k / 4 table look-ups per strand instead of k multiply-adds each with its own base extraction (the counting
// pipeline's P1 does this once per thread and tile, then rolls: at k = 63 the start-up was most of the kernel).
// polyF[byte] serves the forward strand (bytes from the top), polyR[byte] the other one (bytes from the bottom,
// complemented, last base first); same values as key_poly bit for bit (ring arithmetic mod 2^64).
__host__ __device__ inline void poly_tables_fill(uint16_t *polyF, uint16_t *polyR, uint32_t i)
{   // entry i < 256 of both tables
    const uint32_t b0 = (i >> 6) & 3, b1 = (i >> 4) & 3, b2 = (i >> 2) & 3, b3 = i & 3;
    polyF[i] = (uint16_t)(125 * b0 + 25 * b1 + 5 * b2 + b3);
    polyR[i] = (uint16_t)(125 * (3u ^ b3) + 25 * (3u ^ b2) + 5 * (3u ^ b1) + (3u ^ b0));
}
__host__ __device__ inline void poly_hashes_tabled(const Kmer &v, int k, const uint16_t *polyF, const uint16_t *polyR, uint64_t *hf_out,
                                                   uint64_t *hr_out)
{
    // forward: the k bases top-aligned in 128 bits, eaten from the top
    uint64_t t_hi, t_lo;
    if (k <= 32) { t_hi = v.lo << (64 - 2 * k); t_lo = 0; }
    else { t_hi = (v.hi << (128 - 2 * k)) | (v.lo >> (2 * k - 64)); t_lo = v.lo << (128 - 2 * k); }
    uint64_t b_hi = k <= 32 ? 0 : v.hi, b_lo = v.lo;  // reverse: eaten from the bottom
    uint64_t hf = 1, hr = 1;
    const int n4 = k >> 2;
    for (int i = 0; i < n4; i++) {
        hf = hf * 625 + polyF[t_hi >> 56];
        t_hi = (t_hi << 8) | (t_lo >> 56);
        t_lo <<= 8;
        hr = hr * 625 + polyR[b_lo & 0xFF];
        b_lo = (b_lo >> 8) | (b_hi << 56);
        b_hi >>= 8;
    }
    for (int i = n4 * 4; i < k; i++) {
        hf = hf * 5 + (t_hi >> 62);
        t_hi = (t_hi << 2) | (t_lo >> 62);
        t_lo <<= 2;
        hr = hr * 5 + (3u ^ (uint32_t)(b_lo & 3));
        b_lo = (b_lo >> 2) | (b_hi << 62);
        b_hi >>= 2;
    }
    *hf_out = hf;
    *hr_out = hr;
}

// one strand each (the extraction kernel starts its forward hashes at a thread's first window and its reverse ones at its last)
__host__ __device__ inline uint64_t poly_hash_f_tabled(const Kmer &v, int k, const uint16_t *polyF)
{
    uint64_t t_hi, t_lo;
    if (k <= 32) { t_hi = v.lo << (64 - 2 * k); t_lo = 0; }
    else { t_hi = (v.hi << (128 - 2 * k)) | (v.lo >> (2 * k - 64)); t_lo = v.lo << (128 - 2 * k); }
    uint64_t hf = 1;
    const int n4 = k >> 2;
    for (int i = 0; i < n4; i++) {
        hf = hf * 625 + polyF[t_hi >> 56];
        t_hi = (t_hi << 8) | (t_lo >> 56);
        t_lo <<= 8;
    }
    for (int i = n4 * 4; i < k; i++) {
        hf = hf * 5 + (t_hi >> 62);
        t_hi = (t_hi << 2) | (t_lo >> 62);
        t_lo <<= 2;
    }
    return hf;
}
__host__ __device__ inline uint64_t poly_hash_r_tabled(const Kmer &v, int k, const uint16_t *polyR)
{
    uint64_t b_hi = k <= 32 ? 0 : v.hi, b_lo = v.lo;
    uint64_t hr = 1;
    const int n4 = k >> 2;
    for (int i = 0; i < n4; i++) {
        hr = hr * 625 + polyR[b_lo & 0xFF];
        b_lo = (b_lo >> 8) | (b_hi << 56);
        b_hi >>= 8;
    }
    for (int i = n4 * 4; i < k; i++) {
        hr = hr * 5 + (3u ^ (uint32_t)(b_lo & 3));
        b_lo = (b_lo >> 2) | (b_hi << 62);
        b_hi >>= 2;
    }
    return hr;
}
// src/utils/PolynomialHash.java:19-28
__host__ __device__ inline int64_t key_poly_plain(const Kmer &v, int k, bool *flipped = nullptr)
{
    uint64_t fw = 1, rc = 1;
    for (int i = 0; i < k; i++) {
        fw = fw * 5 + base_at(v, k, i);
        rc = rc * 5 + (3u ^ base_at(v, k, k - 1 - i));
    }
    const int64_t a = (int64_t)fw, b = (int64_t)rc;
    if (flipped) *flipped = b < a;
    return a < b ? a : b;  // Math.min on signed longs
}

// The two 256-entry tables of poly_hashes_tabled as constants of the code object: a key worked out from scratch on the device --
// every node of a round of the walk at k > 31, every window of the direct kernel -- takes k / 4 look-ups a strand instead of k
// multiply-adds with a base extraction each (~1 400 instructions a key at k = 63: most of what a round of the walk computed).
struct PolyTabs {
    uint16_t f[256], r[256];
    constexpr PolyTabs() : f(), r()
    {
        for (uint32_t i = 0; i < 256; i++) {
            const uint32_t b0 = (i >> 6) & 3, b1 = (i >> 4) & 3, b2 = (i >> 2) & 3, b3 = i & 3;
            f[i] = (uint16_t)(125 * b0 + 25 * b1 + 5 * b2 + b3);
            r[i] = (uint16_t)(125 * (3u ^ b3) + 25 * (3u ^ b2) + 5 * (3u ^ b1) + (3u ^ b0));
        }
    }
};
__device__ const PolyTabs g_poly_tabs = PolyTabs();

// src/utils/PolynomialHash.java:19-28 (key_poly_plain above is the loop as written there; same values)
__host__ __device__ inline int64_t key_poly(const Kmer &v, int k, bool *flipped = nullptr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t fw, rc;
    poly_hashes_tabled(v, k, g_poly_tabs.f, g_poly_tabs.r, &fw, &rc);
    const int64_t a = (int64_t)fw, b = (int64_t)rc;
    if (flipped) *flipped = b < a;
    return a < b ? a : b;  // Math.min on signed longs
#else
    return key_poly_plain(v, k, flipped);
#endif
}

// (4 + x) * p for x = 0 .. 3 without a multiplication (p = 5^k: what a rolling step of either strand's hash takes away)
__host__ __device__ __forceinline__ uint64_t poly_4x_times(uint32_t x, uint64_t p)
{
    return 4 * p + ((x & 1u) ? p : 0ull) + ((x & 2u) ? 2 * p : 0ull);
}

// src/utils/FNV1AHash.java:8-9,33-42
__host__ __device__ inline int64_t key_fnv1a(const Kmer &v, int k, bool *flipped = nullptr)
{
    const uint64_t prime = 1099511628211ull;
    uint64_t fw = 14695981039346656037ull, rc = 14695981039346656037ull;
    for (int i = 0; i < k; i++) {
        fw = (fw ^ (uint64_t)base_at(v, k, i)) * prime;
        rc = (rc ^ (uint64_t)(3u ^ base_at(v, k, k - 1 - i))) * prime;
    }
    const int64_t a = (int64_t)fw, b = (int64_t)rc;
    if (flipped) *flipped = b < a;
    return a < b ? a : b;
}

// src/algo/OneSequenceCalculator.java:89-96 getKmerKey / src/tools/EnvironmentFinderMain.java:128
template <int MODE>
__host__ __device__ __forceinline__ int64_t key_of(const Kmer &v, int k, bool *flipped = nullptr)
{
    if (MODE == KEY_PACKED) return key_packed(v.lo, k, flipped);
    if (MODE == KEY_POLY) return key_poly(v, k, flipped);
    return key_fnv1a(v, k, flipped);
}

__host__ __device__ inline int64_t key_of_mode(const Kmer &v, int k, int mode)
{
    if (mode == KEY_PACKED) return key_of<KEY_PACKED>(v, k);
    if (mode == KEY_POLY) return key_of<KEY_POLY>(v, k);
    return key_of<KEY_FNV1A>(v, k);
}

// The k-mer starting at base p of the packed read set (layout: include/mcgpu.h).
__device__ __forceinline__ Kmer extract_kmer(const uint64_t *__restrict__ words, uint64_t p, int k)
{
    const uint64_t wi = p >> 5;
    const int off = 2 * (int)(p & 31);
    const uint64_t w0 = words[wi], w1 = words[wi + 1];  // wi + 1 is at worst the pad word
    const uint64_t a = off ? ((w0 << off) | (w1 >> (64 - off))) : w0;  // bases p .. p+31
    Kmer r;
    if (k <= 32) {
        r.hi = 0;
        r.lo = a >> (64 - 2 * k);
        return r;
    }
    const uint64_t w2 = (off + 2 * k > 128) ? words[wi + 2] : 0;  // only touched when it holds real bases
    const uint64_t b = off ? ((w1 << off) | (w2 >> (64 - off))) : w1;  // bases p+32 .. p+63
    const int s = 128 - 2 * k;  // 2 .. 62
    r.hi = a >> s;
    r.lo = (a << (64 - s)) | (b >> s);
    return r;
}

// Neighbours in the reference's order (src/utils/StringUtils.java:8-32, NUCLEOTIDES = A,G,C,T =
// codes 0..3): dir -1: j-th left neighbour = code j + kmer[0..k-2]; dir +1: kmer[1..] + code j;
// dir 0: index 2c = left(c), 2c+1 = right(c).
__host__ __device__ __forceinline__ Kmer neighbour(const Kmer &v, int k, int dir, int j)
{
    const bool left = dir < 0 || (dir == 0 && !(j & 1));
    const uint64_t c = (uint64_t)(dir == 0 ? (j >> 1) : j);
    Kmer r;
    if (left) {  // (v >> 2) | c << 2(k-1)
        r.lo = (v.lo >> 2) | (v.hi << 62);
        r.hi = v.hi >> 2;
        const int sh = 2 * (k - 1);
        if (sh >= 64) r.hi |= c << (sh - 64); else r.lo |= c << sh;
    } else {  // ((v << 2) | c) & mask
        r.hi = (v.hi << 2) | (v.lo >> 62);
        r.lo = (v.lo << 2) | c;
        if (k <= 32) {
            r.hi = 0;
            if (k < 32) r.lo &= (1ull << (2 * k)) - 1;
        } else {
            r.hi &= (1ull << (2 * k - 64)) - 1;
        }
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// Read pointers (Slot::aux).  The context keeps the packed bases of every read it was given (the "read store",
// mcgpu.hip) and a slot remembers WHERE one occurrence of its key sits in it.  The BFS uses that only to GUESS the
// next vertices of a linear stretch -- the bases that follow the occurrence in its read are the path a walker will
// most likely take -- and looks every guess up, so a missing, stale or wrong pointer can cost time but never change a
// result.  32 bits: 0 = none; v = aux - 1 < 2^31: the occurrence starts at base v of the store, exactly (14 M reads of
// 150 bases).  Beyond that the value names a GRANULE of the store and the reader matches the k-mer against every offset of
// it (+ PTR_SLACK, see ptr_advance), in tiers, so that a store a few times the exact range still gets fine pointers:
//   v in [2^31,            2^31 + 2^30)            granules of   4 bases   positions 2.1 G ..   6.4 G
//   v in [2^31 + 2^30,     2^31 + 2^30 + 2^29)     granules of  16 bases             6.4 G ..  15.0 G
//   v in [2^31 + 3 * 2^29, 2^32 - 2)               granules of  64 bases            15.0 G ..  49 G   (beyond: no pointer)
// (one tier of 64-base granules from 2^31 on, as it was, made the walk over 50 M reads twice as long as over 10 M.)
constexpr uint64_t PTR_EXACT_END = 1ull << 31;
constexpr uint32_t PTR_SLACK = 16;
constexpr uint32_t PTR_T1_LG = 2, PTR_T2_LG = 4, PTR_T3_LG = 6;  // (a hop looks at 512 bases around a pointer: 64 + PTR_SLACK candidate offsets is what fits)
constexpr uint64_t PTR_T1_N = 1ull << 30, PTR_T2_N = 1ull << 29, PTR_T3_N = (1ull << 29) - 2;
constexpr uint64_t PTR_T1_POS = PTR_EXACT_END, PTR_T2_POS = PTR_T1_POS + (PTR_T1_N << PTR_T1_LG), PTR_T3_POS = PTR_T2_POS + (PTR_T2_N << PTR_T2_LG);
__host__ __device__ __forceinline__ uint32_t ptr_encode(uint64_t pos)
{
    if (pos < PTR_EXACT_END) return (uint32_t)pos + 1u;
    uint64_t v;
    if (pos < PTR_T2_POS) v = PTR_EXACT_END + ((pos - PTR_T1_POS) >> PTR_T1_LG);
    else if (pos < PTR_T3_POS) v = PTR_EXACT_END + PTR_T1_N + ((pos - PTR_T2_POS) >> PTR_T2_LG);
    else {
        const uint64_t g = (pos - PTR_T3_POS) >> PTR_T3_LG;
        if (g >= PTR_T3_N) return 0u;
        v = PTR_EXACT_END + PTR_T1_N + PTR_T2_N + g;
    }
    return (uint32_t)v + 1u;
}
// first base of the range the occurrence starts in; *span = number of candidate offsets
__host__ __device__ __forceinline__ uint64_t ptr_decode(uint32_t aux, uint32_t *span)
{
    const uint64_t v = (uint64_t)aux - 1;
    if (v < PTR_EXACT_END) { *span = 1; return v; }
    const uint64_t w = v - PTR_EXACT_END;
    if (w < PTR_T1_N) { *span = (1u << PTR_T1_LG) + PTR_SLACK; return PTR_T1_POS + (w << PTR_T1_LG); }
    if (w < PTR_T1_N + PTR_T2_N) { *span = (1u << PTR_T2_LG) + PTR_SLACK; return PTR_T2_POS + ((w - PTR_T1_N) << PTR_T2_LG); }
    *span = (1u << PTR_T3_LG) + PTR_SLACK;
    return PTR_T3_POS + ((w - PTR_T1_N - PTR_T2_N) << PTR_T3_LG);
}
// pointer of the window j <= 15 bases after the window a pointer names (windows of one super-k-mer record)
__host__ __device__ __forceinline__ uint32_t ptr_advance(uint32_t aux, uint32_t j)
{
    if (aux == 0) return 0;
    const uint64_t v = (uint64_t)aux - 1;
    if (v + 16 < PTR_EXACT_END) return aux + j;
    if (v < PTR_EXACT_END) return ptr_encode(v + j);  // (the last exact positions: window j may lie in the first granule)
    return aux;  // (a granule: the reader's range has PTR_SLACK to spare)
}

// Which occurrence of a key (counted from 0 in the order the counting kernels meet them) leaves its pointer:
// ptr_from + r with r = a few bits of the key, r < min(4, thr - ptr_from) so that every key that reaches the coverage
// threshold thr has one.  The occurrences of neighbouring k-mers arrive in the same order (read by read): if all took
// the same one, the pointers along a stretch of the graph would all lead into one read.
__host__ __device__ __forceinline__ uint32_t ptr_pick(uint64_t key, uint32_t ptr_from, uint32_t solid_thr)
{   // which occurrence (counted from 0) leaves its pointer: ptr_from + r, r < min(4, thr - ptr_from) so that every key that reaches the threshold has one
    const uint32_t room = solid_thr > ptr_from ? solid_thr - ptr_from : 1u;
    const uint32_t r = (uint32_t)(key ^ (key >> 9) ^ (key >> 23)) & 3u;
    // r % room for room = 1, 2, 3 without a division (the compiler makes one of float operations, and the merge kernel
    // comes here for the first occurrences of every key)
    const uint32_t m = room >= 4 ? r : (room == 3 ? (r == 3 ? 0u : r) : (room == 2 ? (r & 1u) : 0u));
    return ptr_from + m;
}

// ... and a later occurrence, one of sixteen, replaces it when the key has that many: the early arrivals of all the
// k-mers of a stretch are the same few reads, and a scout that has just used those up wants pointers into others.
__host__ __device__ __forceinline__ uint32_t ptr_pick_late(uint64_t key, uint32_t ptr_from)
{
    return ptr_from + 4 + ((uint32_t)((key >> 3) ^ (key >> 17) ^ (key >> 41)) & 15u);
}

// reverse complement of an oriented k-mer of up to 64 bases (two words, right-aligned)
__host__ __device__ __forceinline__ uint64_t rc64_pairs(uint64_t x)
{  // reverses the 32 2-bit groups of a word and complements them
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r = __brevll(x);
#else
    uint64_t r = x;
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    r = ((r & 0x3333333333333333ull) << 2) | ((r >> 2) & 0x3333333333333333ull);
    r = ((r & 0x0f0f0f0f0f0f0f0full) << 4) | ((r >> 4) & 0x0f0f0f0f0f0f0f0full);
    r = ((r & 0x00ff00ff00ff00ffull) << 8) | ((r >> 8) & 0x00ff00ff00ff00ffull);
    r = ((r & 0x0000ffff0000ffffull) << 16) | ((r >> 16) & 0x0000ffff0000ffffull);
    r = (r << 32) | (r >> 32);
#endif
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    return ~r;
}
__host__ __device__ __forceinline__ Kmer rc_kmer(const Kmer &v, int k)
{
    Kmer r;
    if (k <= 32) {
        r.hi = 0;
        r.lo = rc64_pairs(v.lo) >> (64 - 2 * k);
        return r;
    }
    // 128-bit value hi:lo reversed pairwise = rc(lo):rc(hi); right-align by 128 - 2k (0 .. 62)
    const uint64_t a = rc64_pairs(v.lo), b = rc64_pairs(v.hi);
    const int s = 128 - 2 * k;
    if (s == 0) { r.hi = a; r.lo = b; }
    else { r.hi = a >> s; r.lo = (b >> s) | (a << (64 - s)); }
    return r;
}

// ---------------------------------------------------------------------------------------------
// The k-mer table in HBM (layout free: SURVEY.md F7).  2^rb regions of RS = 2^sb slots; a key
// lives in region = top rb bits of fmix64(key), starting at offset = next sb bits, linear
// probing that wraps inside the region.  A slot is 16 bytes {u64 key; u32 count; u32 aux} so one
// dwordx4 load returns key, count and read pointer together and a region is one contiguous run of lines.
struct Slot {
    uint64_t key;
    uint32_t count;
    uint32_t aux;
};

// ---------------------------------------------------------------------------------------------
// Minimizer bins.  With packed keys and k >= SK_MIN_K the REGION of a key is not taken from the
// key's own hash but from its canonical minimizer: the SK_M-mer x inside the k-mer whose
// sk_order(min(x, rc(x))) is smallest.  Both strands of a k-mer hold the same canonical SK_M-mers,
// so the bin is a function of the key; and consecutive windows of a read mostly share their
// minimizer, which is what lets the counting pipeline move "super-k-mers" (a run of windows in one
// 16-byte record) instead of one record per window (count_pipeline.h).  Nothing of this reaches a
// result: it only decides where in the table a key lives.
constexpr int SK_M = 15;
constexpr int SK_MIN_K = 23;  // shorter k-mers: runs too short to pay; regions from fmix64(key) as for hash keys
constexpr uint32_t SK_MMASK = (1u << (2 * SK_M)) - 1;
constexpr uint32_t SK_NONE = 0xFFFFFFFFu;  // "no window here" in arrays of minimizer hashes

__host__ __device__ __forceinline__ uint32_t sk_order(uint32_t canon_mmer)
{  // a bijection of 32-bit words: random-looking total order of the SK_M-mers (ties impossible below 2^30).  It never gives
   // SK_NONE for an SK_M-mer: the one word it maps there is 0xCCFF8DF3, and canonical 15-mers are below 2^30 (round 3 tested
   // every hash against it: two of the thirteen instructions a base position costs the extraction kernel)
    uint32_t x = canon_mmer * 0x9E3779B1u;
    x ^= x >> 15;
    return x;
}
static_assert(SK_M == 15, "sk_order's image of the SK_M-mers must not hold SK_NONE: check again for another SK_M");
__host__ __device__ __forceinline__ uint32_t sk_bin(uint32_t hmin)
{  // the minimum of many hashes is small: mix again before taking top bits as a bin number
    uint32_t x = hmin;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// owner rank of everything that shares a minimizer (multi-GPU split): mixed differently from sk_bin, so that
// the keys one rank owns still spread over all regions of its table
__host__ __device__ __forceinline__ uint32_t sk_owner(uint32_t hmin, uint32_t n_owners)
{
    return sk_bin(hmin ^ 0x5BD1E995u) % n_owners;
}
__host__ __device__ __forceinline__ uint32_t sk_rc_mmer(uint32_t x)
{
    return (uint32_t)rc_packed((uint64_t)x, SK_M);
}
// smallest sk_order over the canonical SK_M-mers of a k-mer (either strand gives the same value)
__host__ __device__ inline uint32_t sk_hmin_of_kmer(uint64_t fw, int k)
{
    uint32_t best = SK_NONE;
    for (int i = 0; i + SK_M <= k; i++) {
        const uint32_t f = (uint32_t)(fw >> (2 * (k - SK_M - i))) & SK_MMASK, r = sk_rc_mmer(f);
        const uint32_t h = sk_order(f < r ? f : r);
        best = h < best ? h : best;
    }
    return best;
}

// ... of a k-mer of up to 64 bases (hi:lo, right-aligned).  Hash keys say nothing about their bases: where such keys live in
// minimizer bins (k > 32 with polynomial keys, count_pipeline.h "long records") every look-up brings the k-mer itself.
__host__ __device__ inline uint32_t sk_hmin_of_kmer2(const Kmer &v, int k, bool two = false)
{   // two: the word made of the TWO smallest values, as a multiset (count_long.h skl_word2) -- tables with mm_k < 0
    // (the bases are taken from the k-mer's END -- two bits off the bottom of hi:lo a step --: the set of SK_M-mers is the same from
    // either side, and base_at's two variable 64-bit shifts a base were a third of what a look-up of the walk computed)
    uint64_t lo = v.lo, hi = v.hi;
    uint32_t f = 0, r = 0, best = SK_NONE, second = SK_NONE;
    for (int i = 0; i < k; i++) {
        const uint32_t b = (uint32_t)lo & 3u;
        lo = (lo >> 2) | (hi << 62);
        hi >>= 2;
        f = (f >> 2) | (b << (2 * (SK_M - 1)));   // the SK_M-mer that STARTS at this base, first base on top
        r = ((r << 2) | (3u - b)) & SK_MMASK;     // ... and its reverse complement
        if (i >= SK_M - 1) {
            const uint32_t h = sk_order(f < r ? f : r);
            const uint32_t t = h > best ? h : best;
            best = h < best ? h : best;
            second = t < second ? t : second;
        }
    }
    return two ? best ^ (second * 0x85EBCA6Bu) : best;
}

struct TableView {
    Slot *slots;
    uint32_t shift;   // 64 - (rb + sb)
    uint32_t rmask;   // RS - 1
    uint32_t n_regions;  // any number when regions are minimizer bins, a power of two otherwise
    int mm_k;         // 0: region from fmix64(key); else k: region from the key's minimizer bin
    unsigned long long *n_used;     // distinct keys stored in slots
    unsigned long long *empty_cnt;  // occurrences of the key that equals EMPTY_KEY (hash modes only)
    uint32_t *fatal;                // set when a region is full and the overflow list cannot take the addition either
    // Additions that found their region full (minimizer bins fill unevenly: the error variants of a deeply covered
    // locus share its bin) wait here as (key, inc, hint) until the host has enlarged the table (mc_finalize_counts).
    uint4 *ovf;
    unsigned long long *ovf_n;
    uint64_t ovf_cap;
    // The merge kernels of the counting pipeline use the same list for occurrences that go on to the next region
    // (count_pipeline.h ovf_push): ovf_leaf[i] = the leaf entry i came from, so that the entries of a leaf that was not
    // committed after all are dropped when the list is drained (0xFFFFFFFF: an entry of table_add's own, always kept).
    uint32_t *ovf_leaf;
};

// home slot inside a minimizer-bin region: 12 well-mixed bits of the key, cheaper than fmix64 (the
// merge kernel of the counting pipeline computes it once per k-mer occurrence)
#ifndef MC_REGION_LG
#define MC_REGION_LG 12   // log2 of the slots of a table region = of the merge kernel's LDS image (tuning builds override it)
#endif
__host__ __device__ __forceinline__ uint32_t sk_home_mix(uint64_t key)
{   // (the home slot is the top MC_REGION_LG bits; the merge kernel takes two more bits below them for its pointer rule)
    uint32_t x = (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA6Bu;
    x ^= x >> 15; x *= 0xC2B2AE35u;
    return x;
}
__host__ __device__ __forceinline__ uint32_t sk_home(uint64_t key) { return sk_home_mix(key) >> (32 - MC_REGION_LG); }

// 32 bits whose TOP bits number the region of a key (and, in the counting pipeline, its buckets)
__host__ __device__ __forceinline__ uint32_t bin32_of(uint64_t key, int mm_k)
{
    return mm_k ? sk_bin(sk_hmin_of_kmer(key, mm_k)) : (uint32_t)(fmix64(key) >> 32);
}

__host__ __device__ __forceinline__ uint64_t slot_of(const TableView &t, uint64_t key)
{
    if (t.mm_k == 0) return fmix64(key) >> t.shift;
    const uint64_t region = ((uint64_t)sk_bin(sk_hmin_of_kmer(key, t.mm_k)) * t.n_regions) >> 32;
    return (region << MC_REGION_LG) | sk_home(key);  // (count_pipeline.h REGION_SLOTS)
}

// Probing rule of the table, the same for every kernel that inserts or looks up: a key sits within TABLE_MAX_PROBES slots of
// its home slot (linear probing that wraps inside the region), or -- when those slots held no free one at the time it came
// -- within as many slots of the same offset in the region behind, and so on through TABLE_CHAIN regions.  A lookup stops at
// the first free slot; it moves on to the next region only after TABLE_MAX_PROBES occupied ones.  Minimizer-bin regions
// fill unevenly (a 300-fold covered genome puts 7 loci' worth of error k-mers into one region in fifty), and a table
// that cannot be doubled any more -- 137 GB on a 288 GB device -- used to have no way out.
constexpr uint32_t TABLE_MAX_PROBES = 128;
constexpr uint32_t TABLE_CHAIN = 4;
// first slot of the region behind the one that starts at `base` (regions of rmask + 1 slots, n_regions of them)
__host__ __device__ __forceinline__ uint64_t next_region_base(uint64_t base, uint32_t rmask, uint64_t n_regions)
{
    const uint64_t nb = base + (uint64_t)rmask + 1;
    return nb >= n_regions * ((uint64_t)rmask + 1) ? 0 : nb;
}

// addAndBound(key, inc) with the saturation deferred to read time (count is 32-bit here; a
// counter that already reached 2^31 is left alone, launches add < 2^30 each, so it never wraps
// and min(32767, count) equals the reference's saturating short, itmo!/utils/NumUtils.java:21-26).
// Returns 1 when the key was new: callers add these up and publish them with wave_add_ull once per
// wave (one hot counter address hammered by every insert costs more than the inserts themselves).
// *before (optional): the key's count before this addition (0 for a new key).
// table_add_at: the same from home slot `s` on (callers that know the key's region otherwise: hash keys in minimizer bins)
__device__ __forceinline__ uint32_t table_add_at(const TableView &t, uint64_t s, uint64_t key, uint32_t inc, uint32_t hint = 0,
                                                 uint32_t *before = nullptr, uint32_t pick = 0, uint32_t pick2 = 0)
{   // pick != 0: the occurrences that find `pick` / `pick2` before them replace the inserter's read pointer with their own (ptr_pick)
    if (before) *before = 0;
    if (key == EMPTY_KEY) {
        const unsigned long long old = atomicAdd(t.empty_cnt, (unsigned long long)inc);
        if (before) *before = old > 0x80000000ull ? 0x80000000u : (uint32_t)old;
        return 0;
    }
    uint64_t base = s & ~(uint64_t)t.rmask;
    // (TABLE_MAX_PROBES slots from the home slot, then the same stretch of the region behind, ...: see there)
    const uint64_t home = s & t.rmask;
    const uint32_t max_probes = t.rmask + 1 < TABLE_MAX_PROBES ? t.rmask + 1 : TABLE_MAX_PROBES;
    for (uint32_t hop = 0; hop < TABLE_CHAIN; hop++, base = next_region_base(base, t.rmask, t.n_regions), s = base | home)
    for (uint32_t probe = 0; probe < max_probes; probe++) {
        Slot *p = t.slots + s;
        const uint4 raw = *reinterpret_cast<const uint4 *>(p);
        uint64_t cur = ((uint64_t)raw.y << 32) | raw.x;
        if (cur == EMPTY_KEY) {
            // a stale EMPTY is harmless: the CAS runs at the coherent memory side and decides
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&p->key), (unsigned long long)EMPTY_KEY,
                            (unsigned long long)key);
            if (cur == EMPTY_KEY) {
                // the inserter alone also writes the hint: one 64-bit add on {count, aux} (aux was 0, and
                // count never carries into it: counters stop growing at 2^31)
                const unsigned long long old =
                    atomicAdd(reinterpret_cast<unsigned long long *>(&p->count), ((unsigned long long)hint << 32) | inc);
                if (before) *before = (uint32_t)old;  // (another thread may have counted this key in between)
                return 1;
            }
            if (cur == key) {
                const uint32_t old = atomicAdd(&p->count, inc);
                if (before) *before = old;
                if (pick && hint && (old == pick || old == pick2)) p->aux = hint;
                return 0;
            }
        } else if (cur == key) {
            if (raw.z < 0x80000000u) {
                const uint32_t old = atomicAdd(&p->count, inc);
                if (before) *before = old;
                if (pick && hint && (old == pick || old == pick2)) p->aux = hint;
            } else if (before) {
                *before = raw.z;
            }
            return 0;
        }
        s = base | ((s + 1) & t.rmask);
    }
    if (t.ovf) {  // the region is full: park the addition
        const unsigned long long i = atomicAdd(t.ovf_n, 1ull);
        if (i < t.ovf_cap) {
            t.ovf[i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), inc, hint);
            if (t.ovf_leaf) t.ovf_leaf[i] = 0xFFFFFFFFu;
            return 0;
        }
    }
    atomicExch(t.fatal, 1u);
    return 0;
}

__device__ __forceinline__ uint32_t table_add(const TableView &t, uint64_t key, uint32_t inc, uint32_t hint = 0,
                                              uint32_t *before = nullptr, uint32_t pick = 0, uint32_t pick2 = 0)
{
    return table_add_at(t, key == EMPTY_KEY ? 0 : slot_of(t, key), key, inc, hint, before, pick, pick2);
}

// does an addition of `inc` to a count that was `before` carry it over the threshold? (both clamp at 32767 when read)
__device__ __forceinline__ uint32_t crosses(uint32_t before, uint32_t inc, uint32_t thr)
{
    return thr && before < thr && (uint64_t)before + inc >= thr;
}

// ---------------------------------------------------------------------------------------------
// The "solid" table: a BFS-only context's copy of the keys with count >= --coverage (the gathered shards of a
// multi-GPU run; MC_BFS_DIRECT=0 builds it on one GPU too), at a load factor
// <= 1/4 so that its (mostly negative) lookups end at the first probe.  Same 16-byte slots as the counting table
// {key, count (already saturated), read pointer}, regions of SOLID_REGION slots, always indexed by the key's own
// hash (a BFS lookup must not pay for a minimizer).  The view also carries the read store the pointers refer to.
// One rank's counting table as the walk on another device sees it (several GPUs: the walk reads the owners' tables where
// they are -- peer access inside one process, hipIpc between processes -- instead of a gathered copy: mcgpu.hip mc_group,
// mc_shard_attach).  32 bytes; an array of them sits in the walker's device memory.
struct ShardRef {
    Slot *slots;
    uint32_t shift, rmask, n_regions;
    int mm_k;
    unsigned long long empty;  // the shard's count of the key that equals EMPTY_KEY (hash modes), as it was when the shard was attached
};

struct SolidView {
    Slot *slots;
    uint32_t shift;  // 64 - log2(#slots)
    uint32_t rmask;  // region size - 1 (probing wraps inside a region, as in the counting table)
    const unsigned long long *empty_cnt;
    uint32_t *fatal;
    const uint64_t *reads;  // packed bases of the read store (nullptr: none), one readable pad word behind them
    uint64_t reads_bases;
    // The BFS of the context that counted the reads walks the counting table itself (no copy to build): then regions
    // are those of the TableView -- minimizer bins when mm_k != 0 -- and counts are clamped when read, as by table_get.
    int mm_k;
    uint32_t n_regions;
    // n_shards > 1: the table is the union of n_shards ranks' tables; a key lives in its owner's -- the owner of its
    // minimizer (owner_mm_k = k: the exchange dealt super-k-mer records, sk_owner) or of its own hash (0: owner_of) --
    // and every lookup goes there (solid_locate); the fields above then describe the walker's own shard.
    const ShardRef *shards;
    uint32_t n_shards;
    int owner_mm_k;
};

// owner rank of a key dealt by its own hash (the multi-GPU split of key streams: low hash bits, disjoint from the bits that
// place the key in its owner's table); pure function, same on every rank -- mc_key_owner
__host__ __device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t n_owners)
{
    return (uint32_t)((fmix64(key) & 0xFFFFFFFFull) % n_owners);
}

__device__ __forceinline__ uint64_t solid_slot_of(const SolidView &t, uint64_t key)
{
    if (t.mm_k == 0) return fmix64(key) >> t.shift;
    const uint64_t region = ((uint64_t)sk_bin(sk_hmin_of_kmer(key, t.mm_k)) * t.n_regions) >> 32;
    return (region << MC_REGION_LG) | sk_home(key);
}

// What a lookup needs of the table it goes to -- the walker's own or, with several GPUs, the key's owner's.  Built field by
// field: a whole-struct copy of the kernel argument `t` made the compiler keep `t` in scratch memory, and every t.slots /
// t.reads of the walk became a scratch load (k_bfs went from 9.4 to 10.7 ms on configs[1] before this was noticed).
struct TableRef {
    Slot *slots;
    uint32_t shift, rmask, n_regions;
    int mm_k;
    const unsigned long long *empty_cnt;
};
__device__ __forceinline__ TableRef own_table(const SolidView &t) { return TableRef{t.slots, t.shift, t.rmask, t.n_regions, t.mm_k, t.empty_cnt}; }
__device__ __forceinline__ TableRef shard_table(const SolidView &t, uint32_t o)
{
    const ShardRef &r = t.shards[o];
    return TableRef{r.slots, r.shift, r.rmask, r.n_regions, r.mm_k, &r.empty};
}
// Where `key` lives: h = the table that holds it (t's own on one GPU), the result its home slot there.
// have_hmin / hmin: the key's minimizer hash when the caller has it already (sk_hmin_of_kmer is 17 hashes at k = 31).
__device__ __forceinline__ uint64_t solid_locate(const SolidView &t, uint64_t key, TableRef &h, bool have_hmin = false, uint32_t hmin = 0)
{
    if (t.n_shards <= 1) {
        h = own_table(t);
        if (t.mm_k == 0) return fmix64(key) >> t.shift;
        const uint32_t hm = have_hmin ? hmin : sk_hmin_of_kmer(key, t.mm_k);
        return ((((uint64_t)sk_bin(hm) * t.n_regions) >> 32) << MC_REGION_LG) | sk_home(key);
    }
    uint32_t hm = 0;
    if (t.owner_mm_k) hm = have_hmin ? hmin : sk_hmin_of_kmer(key, t.owner_mm_k);
    h = shard_table(t, t.owner_mm_k ? sk_owner(hm, t.n_shards) : owner_of(key, t.n_shards));
    if (h.mm_k == 0) return fmix64(key) >> h.shift;
    if (!t.owner_mm_k) hm = sk_hmin_of_kmer(key, h.mm_k);
    return ((((uint64_t)sk_bin(hm) * h.n_regions) >> 32) << MC_REGION_LG) | sk_home(key);
}

// The probe sequence of `key` from slot s on, four slots to a memory round trip: a lookup in the walk is one link of a
// dependent chain, and the lanes of a wave look different keys up -- the longest sequence among them decides, one round
// trip per step (a table a little fuller than planned made the walk twice as long when this went slot by slot).
// count (saturated) or -1; *aux (may be null) = the slot's read pointer; n_done = slots the caller has looked at already.
// Probe slots that were requested together are wanted together, whole.  Left to itself the compiler (read in the code object of
// k_bfs, round 4) fetches a slot's key alone and its count and read pointer in a SECOND, dependent load for the lanes whose key
// matched, and sinks each further slot's load below the test that ends the loop at the slot before it: a lookup that the source
// shows as one round trip made two to five.  An empty asm that names every word pins the loads where they are written.
__device__ __forceinline__ void slots_wanted([[maybe_unused]] uint4 &a, [[maybe_unused]] uint4 &b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MC_NO_SLOTS_WANTED)
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
#endif
}
__device__ __forceinline__ void slots_wanted([[maybe_unused]] uint4 &a, [[maybe_unused]] uint4 &b, [[maybe_unused]] uint4 &c, [[maybe_unused]] uint4 &d)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MC_NO_SLOTS_WANTED)
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w),
                      "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
#endif
}

template <int CHUNK = 4>
__device__ __forceinline__ int solid_probe_from(const TableRef &t, uint64_t key, uint64_t s, uint32_t n_done, uint32_t *aux)
{   // CHUNK: slots requested together (4, or 8 where many lanes wait for the slowest of them: a walk's round)
    uint64_t base = s & ~(uint64_t)t.rmask;
    const uint64_t n_regions = t.n_regions ? (uint64_t)t.n_regions : ((1ull << (64 - t.shift)) / ((uint64_t)t.rmask + 1));
    const uint64_t home = (s - n_done) & t.rmask;
    const uint32_t max_probes = t.rmask + 1 < TABLE_MAX_PROBES ? t.rmask + 1 : TABLE_MAX_PROBES;
    for (uint32_t hop = 0; hop < TABLE_CHAIN; hop++) {
        for (uint32_t probe = hop ? 0 : n_done; probe < max_probes; probe += CHUNK) {
            uint4 a[CHUNK];
#pragma unroll
            for (int i = 0; i < CHUNK; i++) a[i] = *reinterpret_cast<const uint4 *>(t.slots + (base | ((s + i) & t.rmask)));
            slots_wanted(a[0], a[1], a[2], a[3]);
            if (CHUNK == 8) slots_wanted(a[CHUNK - 4], a[CHUNK - 3], a[CHUNK - 2], a[CHUNK - 1]);
#pragma unroll
            for (int i = 0; i < CHUNK; i++) {
                if (probe + (uint32_t)i >= max_probes) break;  // (a free slot BEHIND the stretch says nothing: the key may have moved on)
                const uint64_t cur = ((uint64_t)a[i].y << 32) | a[i].x;
                if (cur == key) {
                    if (aux) *aux = a[i].w;
                    return a[i].z > 32767u ? 32767 : (int)a[i].z;
                }
                if (cur == EMPTY_KEY) return -1;
            }
            s = base | ((s + CHUNK) & t.rmask);
        }
        // no free slot in the whole stretch: the key may have been handed on to the next region (table_add)
        base = next_region_base(base, t.rmask, n_regions);
        s = base | home;
    }
    return -1;
}

// count (saturated) or -1; *aux (may be null) = the slot's read pointer
__device__ __forceinline__ int solid_get(const SolidView &t, uint64_t key, uint32_t *aux = nullptr)
{
    if (aux) *aux = 0;
    if (key == EMPTY_KEY) {
        const unsigned long long c = t.n_shards > 1 ? t.shards[owner_of(key, t.n_shards)].empty : *t.empty_cnt;  // (hash modes only: dealt by the key's hash)
        return c == 0 ? -1 : (c > 32767ull ? 32767 : (int)c);
    }
    TableRef h;  // (several GPUs: the owner's table)
    const uint64_t s0 = solid_locate(t, key, h);
    return solid_probe_from(h, key, s0, 0, aux);
}

// The same for the k-mer `v` whose key is `key`, in any key mode: a hash key says nothing about its bases, so where such keys
// live in minimizer bins (t.mm_k != 0 with MODE != KEY_PACKED: the long-record counting pipeline) the bin comes from v itself.
template <int MODE>
__device__ __forceinline__ uint64_t solid_locate_kmer(const SolidView &t, const Kmer &v, int k, uint64_t key, TableRef &h)
{
    if (MODE != KEY_PACKED && t.n_shards <= 1 && t.mm_k != 0) {
        h = own_table(t);  // (the bin word of a hash key: its low byte cleared, count_long.h skl_bin)
        return ((((uint64_t)(sk_bin(sk_hmin_of_kmer2(v, k, t.mm_k < 0)) & 0xFFFFFF00u) * t.n_regions) >> 32) << MC_REGION_LG) | sk_home(key);
    }
    return solid_locate(t, key, h);
}
template <int MODE>
__device__ __forceinline__ int solid_get_kmer(const SolidView &t, const Kmer &v, int k, uint64_t key, uint32_t *aux = nullptr)
{
    if (MODE == KEY_PACKED || t.n_shards > 1 || t.mm_k == 0 || key == EMPTY_KEY) return solid_get(t, key, aux);
    if (aux) *aux = 0;
    TableRef h;
    const uint64_t s0 = solid_locate_kmer<MODE>(t, v, k, key, h);
    return solid_probe_from(h, key, s0, 0, aux);
}

// sum over the wave, one atomic per wave.  Every lane of the wave must call it (convergent).
__device__ __forceinline__ void wave_add_ull(unsigned long long *p, unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(p, v);
}

// BigLong2ShortHashMap.get: -1 when absent, else min(32767, count)
__device__ __forceinline__ int table_get(const TableView &t, uint64_t key)
{
    if (key == EMPTY_KEY) {
        const unsigned long long c = *t.empty_cnt;
        return c == 0 ? -1 : (c > 32767ull ? 32767 : (int)c);
    }
    uint64_t s = slot_of(t, key);
    uint64_t base = s & ~(uint64_t)t.rmask;
    const uint64_t home = s & t.rmask;
    const uint32_t max_probes = t.rmask + 1 < TABLE_MAX_PROBES ? t.rmask + 1 : TABLE_MAX_PROBES;
    for (uint32_t hop = 0; hop < TABLE_CHAIN; hop++, base = next_region_base(base, t.rmask, t.n_regions), s = base | home)
    for (uint32_t probe = 0; probe < max_probes; probe++) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(t.slots + s);
        const uint64_t cur = ((uint64_t)raw.y << 32) | raw.x;
        if (cur == key) return raw.z > 32767u ? 32767 : (int)raw.z;
        if (cur == EMPTY_KEY) return -1;
        s = base | ((s + 1) & t.rmask);
    }
    return -1;
}

}  // namespace mc

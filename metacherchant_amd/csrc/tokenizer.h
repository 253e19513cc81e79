// f1: uncompressed FASTA / FASTQ text -> 2-bit packed reads ON THE GPU.
//
// The reference parses inside its dispatcher's lock, one thread for the whole file (src/io/ReadsDispatcher.java:34-53 ->
// itmo!/io/readers/FastaReader.java:54-104, FastqReader.java:53-82, FastaReaderFromXQSourceTrunc.java:61-95); the host
// reader of csrc/host/envfinder.cpp does it on all cores (~5 Gbases/s).  Here the file's bytes go to HBM as they are and
// a handful of data-parallel passes do the rest, with the reference's policy:
//   FASTA  a line that starts with '>' or ';' ends the record before it; the other lines of a record are concatenated
//          (a trailing '\r' comes off); a record with an N / n anywhere is dropped whole; empty records give nothing;
//   FASTQ  records of four lines (@id, bases, +, qualities); a base that is N n . or has phred < 1 (quality char minus
//          offset, 6 bits) ends a piece and is dropped, every non-empty piece is a read.
// Passes: (1) newline positions (count per tile, scan, write); (2) a wave per line (FASTA) or per record (FASTQ)
// classifies it and measures what it keeps; (3) scans turn the measures into read numbers and base offsets; (4) a wave
// per line / record packs the bases that stay, through LDS words of its own.  Anything out of the ordinary -- a byte that
// is no base, a FASTQ record whose lines do not look like one, a quality char outside [offset, 126] -- raises a flag and
// the CALLER reads the file with the host parser instead, which reproduces the reference's behaviour (and messages) in
// those cases.  Nothing here decides a result differently from the host reader: tests compare the tables.
#pragma once
#include "kmer_device.h"

namespace mc {
namespace tok {

constexpr int T_THREADS = 256;
constexpr uint32_t T_BYTES = 32;                        // bytes per thread of the newline passes
constexpr uint32_t T_TILE = T_THREADS * T_BYTES;        // 8192 bytes per workgroup
constexpr uint32_t SCAN_TILE = 4096;                    // elements per workgroup of the generic scan
enum { TOK_BAD_CHAR = 1, TOK_BAD_STRUCTURE = 2, TOK_BAD_QUALITY = 4 };

__device__ __forceinline__ int base_code(uint8_t c)
{   // A0 G1 C2 T3 (itmo!/dna/DnaTools.java:31), either case; -1: not a base
    switch (c | 0x20) {
    case 'a': return 0;
    case 'g': return 1;
    case 'c': return 2;
    case 't': return 3;
    default: return -1;
    }
}

// ---- a scan of 32-bit counts into 64-bit offsets: tile sums, one workgroup over the sums, tiles again
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds_wave, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if ((int)lane >= o) x += y;
    }
    if (lane == 63) lds_wave[wv] = x;
    __syncthreads();
    uint32_t before = 0, tot = 0;
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t c = lds_wave[i];
        if (i < wv) before += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return before + x - v;
}

__global__ void __launch_bounds__(T_THREADS) k_scan_sums(const uint32_t *__restrict__ in, uint64_t n, unsigned long long *tile_sums)
{
    __shared__ uint32_t lw[T_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < SCAN_TILE; i += T_THREADS) s += base + i < n ? in[base + i] : 0u;
    uint32_t tot;
    (void)block_excl_scan(s, lw, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// exclusive scan of `m` 64-bit values in place by ONE workgroup; total -> *total
__global__ void __launch_bounds__(1024) k_scan_one(unsigned long long *v, uint64_t m, unsigned long long *total)
{
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t b = 0; b < m; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const unsigned long long x = i < m ? v[i] : 0ull;
        part[threadIdx.x] = x;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele
            const unsigned long long y = threadIdx.x >= o ? part[threadIdx.x - o] : 0ull;
            __syncthreads();
            part[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < m) v[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(T_THREADS) k_scan_apply(const uint32_t *__restrict__ in, uint64_t n, const unsigned long long *__restrict__ tile_off,
                                                         unsigned long long *out)
{
    __shared__ uint32_t lw[T_THREADS / 64];
    constexpr uint32_t PER = SCAN_TILE / T_THREADS;  // consecutive elements per thread
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * PER;
    uint32_t vals[PER], s = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) { vals[i] = base + i < n ? in[base + i] : 0u; s += vals[i]; }
    uint32_t tot;
    uint64_t at = tile_off[blockIdx.x] + block_excl_scan(s, lw, &tot);
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
        if (base + i < n) out[base + i] = at;
        at += vals[i];
    }
}

// ---- pass 1: where the newlines are.  The text buffer is padded with zero bytes to a whole number of tiles, so a
// thread takes its 32 bytes as two 16-byte loads (a wave reads 2 KB in a row).
__device__ __forceinline__ uint32_t nl_bits4(uint32_t v)
{   // bit i set when byte i of v is '\n' (exact: no borrow between bytes)
    const uint32_t x = v ^ 0x0A0A0A0Au;
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu) >> 7;  // 1 at bit 8i for a zero byte i
    return (z | z >> 7 | z >> 14 | z >> 21) & 0xFu;
}
__device__ __forceinline__ uint32_t nl_bits32(const uint8_t *__restrict__ t, uint64_t base)
{
    const uint4 a = *reinterpret_cast<const uint4 *>(t + base), b = *reinterpret_cast<const uint4 *>(t + base + 16);
    return nl_bits4(a.x) | nl_bits4(a.y) << 4 | nl_bits4(a.z) << 8 | nl_bits4(a.w) << 12 | nl_bits4(b.x) << 16 | nl_bits4(b.y) << 20 |
           nl_bits4(b.z) << 24 | nl_bits4(b.w) << 28;
}

__global__ void __launch_bounds__(T_THREADS) k_nl_count(const uint8_t *__restrict__ t, uint32_t *tile_counts)
{
    __shared__ uint32_t lw[T_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * T_TILE + (uint64_t)threadIdx.x * T_BYTES;
    uint32_t tot;
    (void)block_excl_scan((uint32_t)__popc(nl_bits32(t, base)), lw, &tot);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(T_THREADS) k_nl_write(const uint8_t *__restrict__ t, const unsigned long long *__restrict__ tile_off, unsigned long long *nl)
{
    __shared__ uint32_t lw[T_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * T_TILE + (uint64_t)threadIdx.x * T_BYTES;
    uint32_t mask = nl_bits32(t, base);
    uint32_t tot;
    uint64_t at = tile_off[blockIdx.x] + block_excl_scan((uint32_t)__popc(mask), lw, &tot);
    for (; mask; mask &= mask - 1) nl[at++] = base + (uint32_t)__builtin_ctz(mask);
}

// ---- a wave at a time: lines and records are short (a read), so a WAVE takes one -- its lanes read 64 bytes in a row
// and a ballot tells every lane which of them hold a base that stays.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint64_t lanes_below() { return (1ull << (threadIdx.x & 63)) - 1; }
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    return (uint64_t)(uint32_t)__shfl((int)(uint32_t)v, src) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src) << 32;
}

// Lines first .. first + N - 1 (N <= 4) of the text: lane i < N ends up with [s, e) of line first + i -- the '\n' and one
// trailing '\r' excluded -- and the line's first byte (0 for an empty line).
__device__ __forceinline__ void wave_line_spans(const uint8_t *__restrict__ t, uint64_t n, const unsigned long long *__restrict__ nl, uint64_t n_nl,
                                                uint64_t first, int N, uint64_t *s, uint64_t *e, uint8_t *c0)
{
    const int lane = threadIdx.x & 63;
    uint64_t v = 0;  // lane i <= N: the position of the newline in front of line first + i  (-1: the start of the text)
    if (lane <= N) {
        const uint64_t j = first + (uint64_t)lane;
        v = j == 0 ? ~0ull : j - 1 < n_nl ? nl[j - 1] : n;
    }
    const uint64_t nxt = shfl64(v, lane < 63 ? lane + 1 : 63);
    uint64_t ss = v + 1, ee = nxt;
    uint8_t first_byte = 0;
    if (lane < N) {
        if (ee > n) ee = n;  // (only the line after the last newline)
        if (ss > ee) ss = ee;
        if (ee > ss && t[ee - 1] == '\r') ee--;
        if (ee > ss) first_byte = t[ss];
    }
    *s = ss;
    *e = ee;
    *c0 = first_byte;
}

// The bases a wave keeps go out through 65 words of LDS of its own: lanes OR their two bits in, and a flush stores the
// words that lie wholly inside what the wave wrote since the last flush and ORs the two at the ends into the output
// (which starts zeroed), where a neighbouring read may have bits too.
constexpr uint32_t WP_WORDS = 65;
struct WavePacker {
    uint64_t *lds, *words;
    uint64_t base0, at;  // output bases [base0, at) are in the LDS words
    __device__ __forceinline__ void init(uint64_t *l, uint64_t *w)
    {
        lds = l;
        words = w;
        base0 = at = 0;
        for (uint32_t i = threadIdx.x & 63; i < WP_WORDS; i += 64) lds[i] = 0;
        wave_sync();
    }
    __device__ __forceinline__ void open(uint64_t b) { base0 = at = b; }
    __device__ __forceinline__ void flush()
    {
        wave_sync();
        const uint64_t w0 = base0 >> 5, nw = at > base0 ? ((at + 31) >> 5) - w0 : 0;
        for (uint64_t i = threadIdx.x & 63; i < nw; i += 64) {
            const uint64_t v = lds[i], W = w0 + i;
            if (W * 32 >= base0 && W * 32 + 32 <= at) words[W] = v;
            else if (v) atomicOr(reinterpret_cast<unsigned long long *>(&words[W]), (unsigned long long)v);
            lds[i] = 0;
        }
        wave_sync();
        base0 = at;
    }
    // every lane of the wave calls this; `good` lanes hold a base (code 0..3); returns where the lane's base went
    __device__ __forceinline__ uint64_t put(bool good, int code)
    {
        if (at - (base0 & ~31ull) + 64 > (uint64_t)WP_WORDS * 32) flush();
        const uint64_t gm = __ballot(good);
        const uint64_t b = at + (uint64_t)__popcll(gm & lanes_below());
        if (good && code) atomicOr(reinterpret_cast<unsigned long long *>(&lds[(b >> 5) - (base0 >> 5)]), (unsigned long long)code << (62 - 2 * (b & 31)));
        at += (uint64_t)__popcll(gm);
        return b;
    }
};

__device__ __forceinline__ uint64_t wave_index() { return ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; }
__device__ __forceinline__ uint64_t wave_count() { return ((uint64_t)gridDim.x * blockDim.x) >> 6; }

// ---- FASTA, pass 2: a wave per line
__global__ void __launch_bounds__(T_THREADS) k_fa_lines(const uint8_t *__restrict__ t, uint64_t n, const unsigned long long *__restrict__ nl, uint64_t n_nl,
                                                       uint64_t n_lines, uint32_t *line_hdr, uint32_t *line_len, uint8_t *line_n, uint32_t *flags)
{
    const int lane = threadIdx.x & 63;
    uint32_t bad = 0;
    for (uint64_t j = wave_index(); j < n_lines; j += wave_count()) {
        uint64_t s, e;
        uint8_t c0;
        wave_line_spans(t, n, nl, n_nl, j, 1, &s, &e, &c0);
        s = shfl64(s, 0);
        e = shfl64(e, 0);
        c0 = (uint8_t)__shfl((int)c0, 0);
        const bool hdr = e > s && (c0 == '>' || c0 == ';');
        bool has_n = false;
        if (!hdr) {
            for (uint64_t i = s + lane; i < e; i += 64) {
                const uint8_t c = t[i];
                if (c == 'N' || c == 'n') has_n = true;
                else if (base_code(c) < 0) bad |= TOK_BAD_CHAR;
            }
            has_n = __ballot(has_n) != 0;
            if (e - s > 0xFFFFFFF0ull) bad |= TOK_BAD_STRUCTURE;  // (a line of 4 G bases: not for this path)
        }
        if (lane == 0) {
            line_hdr[j] = hdr ? 1u : 0u;
            line_len[j] = hdr ? 0u : (uint32_t)(e - s);
            line_n[j] = has_n ? 1 : 0;
        }
    }
    if (bad) atomicOr(flags, bad);
}

// pass 3a: per record (rec = headers at or before the line): does it hold an N, how long is it, which line opens it
__global__ void k_fa_records(const unsigned long long *__restrict__ hdr_before, const uint32_t *__restrict__ line_hdr, const uint32_t *__restrict__ line_len,
                             const uint8_t *__restrict__ line_n, uint64_t n_lines, uint8_t *rec_n, unsigned long long *rec_len, uint32_t *rec_first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_lines; j += stride) {
        const uint32_t len = line_len[j];
        if (len == 0) continue;
        const uint64_t rec = hdr_before[j] + line_hdr[j];  // (exclusive scan + own flag; a header line has len 0 anyway)
        if (line_n[j]) rec_n[rec] = 1;
        atomicAdd(&rec_len[rec], (unsigned long long)len);
        atomicMin(&rec_first[rec], (uint32_t)j);
    }
}

// pass 3b: what a line contributes once the records with N are gone
__global__ void k_fa_keep(const unsigned long long *__restrict__ hdr_before, const uint32_t *__restrict__ line_hdr, const uint32_t *__restrict__ line_len,
                          uint64_t n_lines, const uint8_t *__restrict__ rec_n, uint32_t *keep_len)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_lines; j += stride) {
        const uint64_t rec = hdr_before[j] + line_hdr[j];
        keep_len[j] = rec_n[rec] ? 0u : line_len[j];
    }
}

__global__ void k_fa_rec_keep(const uint8_t *__restrict__ rec_n, const unsigned long long *__restrict__ rec_len, uint64_t n_rec, uint32_t *rec_keep)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += stride) rec_keep[r] = !rec_n[r] && rec_len[r] ? 1u : 0u;
}

__global__ void k_fa_offsets(const uint32_t *__restrict__ rec_keep, const unsigned long long *__restrict__ rec_out, const uint32_t *__restrict__ rec_first,
                             const unsigned long long *__restrict__ out_off, uint64_t n_rec, uint64_t n_reads, uint64_t total_bases, uint64_t *offsets,
                             uint64_t dst_base)
{   // dst_base: where this chunk's first base goes in the output (the chunks of a file are packed back to back)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += stride)
        if (rec_keep[r]) offsets[rec_out[r]] = dst_base + out_off[rec_first[r]];
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n_reads] = dst_base + total_bases;
}

// pass 4: a wave per line packs it
__global__ void __launch_bounds__(T_THREADS) k_fa_pack(const uint8_t *__restrict__ t, uint64_t n, const unsigned long long *__restrict__ nl, uint64_t n_nl,
                                                      uint64_t n_lines, const uint32_t *__restrict__ keep_len, const unsigned long long *__restrict__ out_off,
                                                      uint64_t *words, uint32_t *flags, uint64_t dst_base)
{
    __shared__ uint64_t lds[T_THREADS / 64][WP_WORDS];
    const int lane = threadIdx.x & 63;
    WavePacker wp;
    wp.init(lds[threadIdx.x >> 6], words);
    uint32_t bad = 0;
    for (uint64_t j = wave_index(); j < n_lines; j += wave_count()) {
        const uint32_t len = keep_len[j];
        if (len == 0) continue;
        uint64_t s, e;
        uint8_t c0;
        wave_line_spans(t, n, nl, n_nl, j, 1, &s, &e, &c0);
        s = shfl64(s, 0);
        wp.open(dst_base + out_off[j]);
        for (uint32_t q = 0; q < len; q += 64) {
            const bool in = q + lane < len;
            const int c = in ? base_code(t[s + q + lane]) : -1;
            if (in && c < 0) bad |= TOK_BAD_CHAR;
            (void)wp.put(in && c >= 0, c);
        }
        wp.flush();
    }
    if (bad) atomicOr(flags, bad);
}

// ---- FASTQ: a wave per record (lines 4r .. 4r + 3)
__device__ __forceinline__ bool fq_bad_base(uint8_t c, uint8_t q, int offset, uint32_t *flags_acc)
{
    if (c == 'N' || c == 'n' || c == '.') return true;
    if (base_code(c) < 0) { *flags_acc |= TOK_BAD_CHAR; return true; }
    if ((int)q < offset || q > 126) { *flags_acc |= TOK_BAD_QUALITY; return true; }
    return (((int)q - offset) & 63) < 1;  // the phred lives in 6 bits (DnaQBuilder.java:32-35): 64 wraps to 0
}

struct FqRecord {
    uint64_t s1, s3;  // where the bases and the qualities start
    uint64_t len;
    bool ok;
};
__device__ __forceinline__ FqRecord fq_record(const uint8_t *__restrict__ t, uint64_t n, const unsigned long long *__restrict__ nl, uint64_t n_nl, uint64_t r)
{
    uint64_t s, e;
    uint8_t c0;
    wave_line_spans(t, n, nl, n_nl, 4 * r, 4, &s, &e, &c0);
    const uint64_t len = e - s;
    FqRecord R;
    R.s1 = shfl64(s, 1);
    R.s3 = shfl64(s, 3);
    R.len = shfl64(len, 1);
    const uint64_t len3 = shfl64(len, 3);
    const int m0 = __shfl((int)c0, 0), m2 = __shfl((int)c0, 2);
    R.ok = m0 == '@' && m2 == '+' && R.len == len3 && R.len <= 0xFFFFFFF0ull;
    return R;
}

// pass 2: structure, pieces and kept bases of every record
__global__ void __launch_bounds__(T_THREADS) k_fq_records(const uint8_t *__restrict__ t, uint64_t n, const unsigned long long *__restrict__ nl, uint64_t n_nl,
                                                         uint64_t n_rec, int offset, uint32_t *rec_pieces, uint32_t *rec_bases, uint32_t *flags)
{
    const int lane = threadIdx.x & 63;
    uint32_t bad = 0;
    for (uint64_t r = wave_index(); r < n_rec; r += wave_count()) {
        const FqRecord R = fq_record(t, n, nl, n_nl, r);
        uint32_t pieces = 0, bases = 0;
        if (!R.ok) {
            bad |= TOK_BAD_STRUCTURE;
        } else {
            uint64_t carry = 0;  // was the base in front of this stretch of 64 a good one
            for (uint64_t q = 0; q < R.len; q += 64) {
                const bool in = q + lane < R.len;
                const bool good = in && !fq_bad_base(t[R.s1 + q + lane], t[R.s3 + q + lane], offset, &bad);
                const uint64_t gm = __ballot(good);
                pieces += (uint32_t)__popcll(gm & ~(gm << 1 | carry));
                bases += (uint32_t)__popcll(gm);
                carry = gm >> 63;
            }
        }
        if (lane == 0) {
            rec_pieces[r] = pieces;
            rec_bases[r] = bases;
        }
    }
    if (bad) atomicOr(flags, bad);
}

// pass 4: the read offset of every piece, and its bases packed
__global__ void __launch_bounds__(T_THREADS) k_fq_emit(const uint8_t *__restrict__ t, uint64_t n, const unsigned long long *__restrict__ nl, uint64_t n_nl,
                                                      uint64_t n_rec, int offset, const unsigned long long *__restrict__ piece_at,
                                                      const unsigned long long *__restrict__ base_at, const uint32_t *__restrict__ rec_bases, uint64_t n_reads,
                                                      uint64_t total_bases, uint64_t *offsets, uint64_t *words, uint64_t dst_base)
{
    __shared__ uint64_t lds[T_THREADS / 64][WP_WORDS];
    const int lane = threadIdx.x & 63;
    WavePacker wp;
    wp.init(lds[threadIdx.x >> 6], words);
    uint32_t dummy = 0;
    for (uint64_t r = wave_index(); r < n_rec; r += wave_count()) {
        if (rec_bases[r] == 0) continue;
        const FqRecord R = fq_record(t, n, nl, n_nl, r);
        uint64_t p = piece_at[r], carry = 0;
        wp.open(dst_base + base_at[r]);
        for (uint64_t q = 0; q < R.len; q += 64) {
            const bool in = q + lane < R.len;
            const uint8_t c = in ? t[R.s1 + q + lane] : 0;
            const bool good = in && !fq_bad_base(c, t[R.s3 + q + lane], offset, &dummy);
            const uint64_t gm = __ballot(good);
            const uint64_t starts = gm & ~(gm << 1 | carry);
            const uint64_t b = wp.put(good, good ? base_code(c) : 0);
            if (starts >> lane & 1) offsets[p + (uint64_t)__popcll(starts & lanes_below())] = b;
            p += (uint64_t)__popcll(starts);
            carry = gm >> 63;
        }
        wp.flush();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n_reads] = dst_base + total_bases;
}

}  // namespace tok
}  // namespace mc

"""ctypes binding of oracle/libmcoracle.so (the C restatement, mc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KEY_PACKED, KEY_POLY, KEY_FNV1A = 0, 1, 2


class _BfsResult(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("hi", C.POINTER(C.c_uint64)),
        ("lo", C.POINTER(C.c_uint64)),
        ("dist", C.POINTER(C.c_int32)),
        ("cov", C.POINTER(C.c_int16)),
        ("last", C.POINTER(C.c_uint8)),
        ("kept", C.POINTER(C.c_uint8)),
        ("queue_len", C.c_uint64),
        ("levels", C.c_uint64),
        ("lookups", C.c_uint64),
    ]


def build():
    """(Re)build libmcoracle.so with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libmcoracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libmcoracle.so")
    src = os.path.join(_HERE, "mc_oracle.c")
    alt = os.environ.get("MCO_LIB")  # (tests/test_host_sanitizers.py: the AddressSanitizer build, `make asan`, with libasan preloaded)
    if alt:
        path = alt
    elif not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
        build()
    L = C.CDLL(path)
    u8p, u64p, i64p, i16p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int64),
                             C.POINTER(C.c_int16))
    L.mco_code.restype = C.c_int
    L.mco_rc_packed.restype = C.c_uint64
    L.mco_rc_packed.argtypes = [C.c_uint64, C.c_int]
    for f in (L.mco_key31, L.mco_poly, L.mco_fnv1a):
        f.restype = C.c_int64
        f.argtypes = [u8p, C.c_int]
    L.mco_key.restype = C.c_int64
    L.mco_key.argtypes = [u8p, C.c_int, C.c_int]
    L.mco_table_new.restype = C.c_void_p
    L.mco_table_free.argtypes = [C.c_void_p]
    L.mco_table_add.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.mco_table_get.restype = C.c_int16
    L.mco_table_get.argtypes = [C.c_void_p, C.c_int64]
    L.mco_table_size.restype = C.c_uint64
    L.mco_table_size.argtypes = [C.c_void_p]
    L.mco_table_dump.restype = C.c_uint64
    L.mco_table_dump.argtypes = [C.c_void_p, i64p, i16p, C.c_uint64]
    L.mco_count_reads.restype = C.c_uint64
    L.mco_count_reads.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, C.c_int, C.c_int]
    L.mco_count_reads_packed.restype = C.c_uint64
    L.mco_count_reads_packed.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64, C.c_int, C.c_int]
    L.mco_count_reads_packed_mt.restype = C.c_uint64
    L.mco_count_reads_packed_mt.argtypes = [u64p, u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, u64p,
                                            C.POINTER(C.c_double), C.POINTER(C.c_void_p)]
    L.mco_pack.argtypes = [u8p, C.c_uint64, u64p]
    L.mco_bfs.restype = C.c_int
    L.mco_bfs.argtypes = [C.c_void_p, C.c_int, C.c_int, u8p, u64p, C.c_uint64, C.c_int, C.c_int, C.c_int64,
                          C.c_int64, C.c_int, C.POINTER(_BfsResult)]
    L.mco_bfs_free.argtypes = [C.POINTER(_BfsResult)]
    L.mco_splitmix.restype = C.c_uint64
    L.mco_splitmix.argtypes = [C.c_uint64, C.c_uint64]
    L.mco_synth_genome.argtypes = [C.c_uint64, C.c_uint64, u8p]
    L.mco_synth_reads.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                  C.c_int, u8p]
    _LIB = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate("AGCT"):
    _CODE[ord(_ch)] = _i
    _CODE[ord(_ch.lower())] = _i


def encode(s):
    """ASCII -> codes A0 G1 C2 T3 (itmo!/dna/DnaTools.java:46-64); raises on anything else."""
    a = _CODE[np.frombuffer(s.encode("ascii") if isinstance(s, str) else s, dtype=np.uint8)]
    if (a == 255).any():
        raise ValueError("non-ACGT character")
    return np.ascontiguousarray(a)


def decode(codes):
    return bytes(np.frombuffer(b"AGCT", dtype=np.uint8)[np.asarray(codes, dtype=np.uint8)]).decode()


def key(codes, k, mode):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    return int(lib().mco_key(_p(codes, C.c_uint8), k, mode))


def pack(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    words = np.zeros((len(codes) + 31) // 32 + 1, dtype=np.uint64)  # +1 pad word (see include/mcgpu.h)
    lib().mco_pack(_p(codes, C.c_uint8), len(codes), _p(words, C.c_uint64))
    return words


class Table:
    def __init__(self, handle=None):
        self.h = C.c_void_p(handle) if handle is not None else C.c_void_p(lib().mco_table_new())

    def __del__(self):
        if getattr(self, "h", None):
            lib().mco_table_free(self.h)
            self.h = None

    def add(self, key_, inc=1):
        lib().mco_table_add(self.h, key_, inc)

    def get(self, key_):
        return int(lib().mco_table_get(self.h, key_))

    def get_many(self, keys):
        L = lib()
        return np.array([L.mco_table_get(self.h, int(x)) for x in keys], dtype=np.int16)

    def size(self):
        return int(lib().mco_table_size(self.h))

    def dump(self):
        n = self.size()
        keys = np.zeros(n, dtype=np.int64)
        cnt = np.zeros(n, dtype=np.int16)
        m = lib().mco_table_dump(self.h, _p(keys, C.c_int64), _p(cnt, C.c_int16), n)
        assert m == n
        o = np.argsort(keys, kind="stable")
        return keys[o], cnt[o]

    def count_reads(self, codes, offsets, k, mode):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return int(lib().mco_count_reads(self.h, _p(codes, C.c_uint8), _p(offsets, C.c_uint64),
                                         len(offsets) - 1, k, mode))

    def count_reads_packed(self, words, offsets, k, mode):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return int(lib().mco_count_reads_packed(self.h, _p(words, C.c_uint64), _p(offsets, C.c_uint64),
                                                len(offsets) - 1, k, mode))


def count_reads_packed_mt(words, offsets, k, mode, threads, want_table=False):
    words = np.ascontiguousarray(words, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nd = C.c_uint64(0)
    sec = C.c_double(0)
    tab = C.c_void_p(0)
    w = lib().mco_count_reads_packed_mt(_p(words, C.c_uint64), _p(offsets, C.c_uint64), len(offsets) - 1, k,
                                        mode, threads, C.byref(nd), C.byref(sec),
                                        C.byref(tab) if want_table else None)
    return int(w), int(nd.value), float(sec.value), (Table(tab.value) if want_table else None)


def bfs(table, k, mode, seeds, direction, min_cov, max_kmers=-1, max_radius=-1, trim=False):
    """seeds: list of code arrays.  Returns None on 'fail', else a dict of numpy arrays in
    distanceToKmer insertion order."""
    off = np.zeros(len(seeds) + 1, dtype=np.uint64)
    for i, s in enumerate(seeds):
        off[i + 1] = off[i] + len(s)
    allc = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.uint8) for s in seeds])
                                if seeds else np.zeros(0, dtype=np.uint8))
    if len(allc) == 0:
        allc = np.zeros(1, dtype=np.uint8)
    r = _BfsResult()
    rc = lib().mco_bfs(table.h, k, mode, _p(allc, C.c_uint8), _p(off, C.c_uint64), len(seeds), direction,
                       min_cov, max_kmers, max_radius, 1 if trim else 0, C.byref(r))
    if rc != 0:
        return None
    n = int(r.n)

    def arr(ptr, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dtype=dt)

    out = dict(hi=arr(r.hi, np.uint64), lo=arr(r.lo, np.uint64), dist=arr(r.dist, np.int32),
               cov=arr(r.cov, np.int16), last=arr(r.last, np.uint8), kept=arr(r.kept, np.uint8),
               queue_len=int(r.queue_len), levels=int(r.levels), lookups=int(r.lookups))
    lib().mco_bfs_free(C.byref(r))
    return out


def kmer_string(hi, lo, k):
    v = (int(hi) << 64) | int(lo)
    return "".join("AGCT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def splitmix(seed, n):
    return int(lib().mco_splitmix(seed, n))


def synth_genome(seed, n_bases):
    g = np.zeros(n_bases, dtype=np.uint8)
    lib().mco_synth_genome(seed, n_bases, _p(g, C.c_uint8))
    return g


def synth_reads(genome, n_contigs, contig_len, seed, first_read, n_reads, L, err_per_10k):
    genome = np.ascontiguousarray(genome, dtype=np.uint8)
    out = np.zeros(n_reads * L, dtype=np.uint8)
    lib().mco_synth_reads(_p(genome, C.c_uint8), n_contigs, contig_len, seed, first_read, n_reads, L,
                          err_per_10k, _p(out, C.c_uint8))
    return out

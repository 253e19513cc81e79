"""ctypes binding of libmcgpu.so (include/mcgpu.h).  Thin: argument marshalling and error
translation only.  There is no CPU fallback: if the HIP library is missing or no MI355X is
present, the calls raise."""
import ctypes as C
import os

import numpy as np

from . import build as _build

KEY_PACKED, KEY_POLY, KEY_FNV1A = 0, 1, 2
FLAG_SOLID_LIST = 1  # mc_config.flags: this context is a shard whose solid k-mers will be exported
MC_ENOSEED = -6


class McError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmcgpu error %d: %s" % (code, msg))
        self.code = code


class _Config(C.Structure):
    _fields_ = [("k", C.c_int32), ("key_mode", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32),
                ("capacity_hint", C.c_uint64)]


class _BfsResult(C.Structure):
    _fields_ = [("n", C.c_uint64), ("hi", C.POINTER(C.c_uint64)), ("lo", C.POINTER(C.c_uint64)),
                ("dist", C.POINTER(C.c_int32)), ("cov", C.POINTER(C.c_int16)), ("last", C.POINTER(C.c_uint8)),
                ("levels", C.c_uint64), ("lookups", C.c_uint64), ("rounds", C.c_uint64), ("device_ms", C.c_double)]


class _BfsJob(C.Structure):
    _fields_ = [("seed_hi", C.POINTER(C.c_uint64)), ("seed_lo", C.POINTER(C.c_uint64)), ("n_seeds", C.c_uint64),
                ("dir", C.c_int32)]


class _ResultOwner:
    """Keeps one mc_bfs_result alive for the numpy views of its arrays; frees it with the last of them."""

    def __init__(self, lib, res):
        self._lib = lib
        self._res = _BfsResult()
        C.memmove(C.byref(self._res), C.byref(res), C.sizeof(_BfsResult))

    def __del__(self):
        try:
            self._lib.mc_bfs_result_free(C.byref(self._res))
        except Exception:  # (interpreter shutdown)
            pass


class Stats(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("count_launches", C.c_uint64), ("count_ms", C.c_double),
                ("count_total_ms", C.c_double), ("table_slots", C.c_uint64), ("table_bytes", C.c_uint64),
                ("grows", C.c_uint64), ("p1_ms", C.c_double), ("p2_ms", C.c_double), ("p3_ms", C.c_double),
                ("spill_keys", C.c_uint64), ("solid_kmers", C.c_uint64), ("solid_sweeps", C.c_uint64),
                ("solid_list_builds", C.c_uint64), ("long_runs", C.c_uint64), ("dup_keys", C.c_uint64),
                ("dup_checks", C.c_uint64), ("dup_ms", C.c_double), ("dup_unchecked", C.c_uint64), ("left_bins", C.c_uint64),
                ("binned_runs", C.c_uint64)]


# every symbol include/mcgpu.h declares; tests check that the library exports all of them
EXPORTS = [
    "mc_abi_version", "mc_create", "mc_destroy", "mc_clear", "mc_set_coverage_hint", "mc_set_read_pointers", "mc_share_read_store", "mc_last_error", "mc_set_stream", "mc_add_reads_packed",
    "mc_add_reads_packed_dev", "mc_add_reads_file", "mc_finalize_counts", "mc_get", "mc_get_dev", "mc_kmer_keys", "mc_bfs", "mc_bfs_batch",
    "mc_bfs_result_free", "mc_export", "mc_export_dev", "mc_add_pairs_dev", "mc_solid_from_pairs_dev", "mc_save_kmers", "mc_load_kmers", "mc_key_owner", "mc_extract_keys_dev",
    "mc_group_create", "mc_group_destroy", "mc_group_last_error", "mc_group_set_coverage_hint", "mc_group_add_reads_packed", "mc_group_add_reads_file",
    "mc_group_finalize_counts", "mc_group_bfs_batch", "mc_group_get_stats", "mc_add_keys_dev", "mc_superkmer_capacity", "mc_extract_superkmers_dev", "mc_add_superkmers_dev",
    "mc_superkmer_fine_buckets", "mc_extract_superkmers_binned_dev", "mc_add_superkmers_binned_dev",
    "mc_read_store_seek", "mc_read_store_tell", "mc_read_store_import_dev", "mc_get_stats", "mc_reset_stats", "mc_trim", "mc_synth_reads_dev", "mc_synth_genome",
    "mc_shard_export", "mc_shard_attach", "mc_shard_detach",
]

_LIB = None


def lib_path():
    """The product library; MC_LIB=<path> selects a tuning build instead (metacherchant_amd/build.py build_lib(variant=...))."""
    return os.environ.get("MC_LIB") or _build.LIB


def load():
    """Loads libmcgpu.so (does not touch the GPU).  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("libmcgpu.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(needs hipcc); there is no CPU fallback")
    # PyTorch-ROCm ships its own HIP runtime; it must be the first one initialised in a process that
    # uses both (the other order leaves torch with "No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, u64, i64, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_int
    u64p, i64p, i16p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int16)
    L.mc_abi_version.restype = i32
    L.mc_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    L.mc_destroy.argtypes = [vp]
    L.mc_destroy.restype = None
    L.mc_last_error.argtypes = [vp]
    L.mc_last_error.restype = C.c_char_p
    L.mc_set_stream.argtypes = [vp, vp]
    L.mc_clear.argtypes = [vp]
    L.mc_set_coverage_hint.argtypes = [vp, i32]
    L.mc_set_read_pointers.argtypes = [vp, i32]
    L.mc_share_read_store.argtypes = [vp, vp]
    L.mc_add_reads_packed.argtypes = [vp, u64p, u64p, u64]
    L.mc_add_reads_packed_dev.argtypes = [vp, vp, vp, u64, u64]
    L.mc_add_reads_file.argtypes = [vp, C.c_char_p, u64p]
    L.mc_finalize_counts.argtypes = [vp, u64p]
    L.mc_save_kmers.argtypes = [vp, C.c_char_p, C.c_char_p, i32, u64p, u64p]
    L.mc_load_kmers.argtypes = [vp, C.c_char_p, i32, u64p, u64p]
    L.mc_get.argtypes = [vp, i64p, u64, i16p]
    L.mc_get_dev.argtypes = [vp, vp, u64, vp]
    L.mc_kmer_keys.argtypes = [vp, u64p, u64p, u64, i64p]
    L.mc_bfs.argtypes = [vp, u64p, u64p, u64, i32, i32, i64, i64, C.POINTER(_BfsResult)]
    L.mc_bfs_batch.argtypes = [vp, C.POINTER(_BfsJob), C.c_uint32, i32, i64, i64, C.POINTER(_BfsResult)]
    L.mc_bfs_result_free.argtypes = [C.POINTER(_BfsResult)]
    L.mc_bfs_result_free.restype = None
    L.mc_export.argtypes = [vp, i32, i64p, i16p, u64, u64p]
    L.mc_export_dev.argtypes = [vp, i32, vp, vp, vp, u64, u64p]
    L.mc_add_pairs_dev.argtypes = [vp, vp, vp, vp, u64]
    L.mc_solid_from_pairs_dev.argtypes = [vp, vp, vp, vp, u64, i32, u64p]
    L.mc_key_owner.argtypes = [i64, C.c_uint32]
    L.mc_key_owner.restype = C.c_uint32
    L.mc_extract_keys_dev.argtypes = [vp, vp, vp, u64, u64, C.c_uint32, vp, vp, u64, u64p]
    L.mc_add_keys_dev.argtypes = [vp, vp, vp, u64]
    L.mc_superkmer_capacity.argtypes = [vp, u64, u64]
    L.mc_superkmer_capacity.restype = u64
    L.mc_extract_superkmers_dev.argtypes = [vp, vp, vp, u64, u64, C.c_uint32, vp, vp, u64, u64p]
    L.mc_add_superkmers_dev.argtypes = [vp, vp, vp, u64]
    L.mc_read_store_seek.argtypes = [vp, u64, u64]
    L.mc_read_store_tell.argtypes = [vp]
    L.mc_read_store_tell.restype = u64
    L.mc_read_store_import_dev.argtypes = [vp, vp, u64, u64]
    L.mc_superkmer_fine_buckets.argtypes = [vp, C.c_uint32]
    L.mc_superkmer_fine_buckets.restype = C.c_uint32
    L.mc_extract_superkmers_binned_dev.argtypes = [vp, vp, vp, u64, u64, C.c_uint32, C.c_uint32, vp, vp, u64, vp, u64p, u64p]
    L.mc_add_superkmers_binned_dev.argtypes = [vp, vp, vp, u64, u64, C.c_uint32, C.c_uint32, u64p, vp]
    L.mc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mc_reset_stats.argtypes = [vp]
    L.mc_trim.argtypes = [vp]
    L.mc_synth_reads_dev.argtypes = [vp, u64, u64, u64, u64, u64, u64, C.c_uint32, C.c_uint32, vp, vp]
    L.mc_synth_genome.argtypes = [u64, u64, u64, C.POINTER(C.c_uint8)]
    if hasattr(L, "mc_shard_export"):  # (a tuning build of an older revision, MC_LIB: scripts/gpu_variants.sh)
        L.mc_shard_export.argtypes = [vp, C.c_char_p]
        L.mc_shard_attach.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32, i32]
        L.mc_shard_detach.argtypes = [vp]
    _LIB = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _dptr(x):
    """device pointer of a torch tensor (or a raw int).  The library works on its own HIP stream, so whatever
    torch still has queued for the tensor (the fill of a torch.zeros, a copy, a collective's result) must be
    done before the pointer is handed over: the tensor's current stream is synchronised here."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if getattr(x, "is_cuda", False):
        import torch
        torch.cuda.current_stream(x.device).synchronize()
    return C.c_void_p(x.data_ptr())


class Context:
    """One k-mer table on one GPU = the BigLong2ShortHashMap of one tool run."""

    def __init__(self, k, key_mode=KEY_PACKED, device=0, capacity_hint=0, flags=0):
        self._L = load()
        self.k, self.key_mode, self.device = k, key_mode, device
        cfg = _Config(k, key_mode, device, flags, capacity_hint)
        h = C.c_void_p()
        rc = self._L.mc_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise McError(rc, (self._L.mc_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.mc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise McError(rc, (self._L.mc_last_error(self._h) or b"").decode())

    def clear(self):
        self._chk(self._L.mc_clear(self._h))

    def set_coverage_hint(self, min_cov):
        """mc_set_coverage_hint: counting keeps #(count >= min_cov) current, BFS set-up skips a table sweep."""
        self._chk(self._L.mc_set_coverage_hint(self._h, int(min_cov)))

    PTRS_NONE, PTRS_OWN_STORE, PTRS_STORE_ELSEWHERE, PTRS_ON_EVERY_RECORD = 0, 1, 2, 0x10

    def set_read_pointers(self, mode):
        """mc_set_read_pointers: False / 0 no pointers, True / 1 this context's own read store, 2 a store kept by another context
        (read_store_tell / read_store_import_dev); | PTRS_ON_EVERY_RECORD: every record it is handed carries a pointer."""
        self._chk(self._L.mc_set_read_pointers(self._h, int(mode)))

    def read_store_seek(self, at_bases, reserve_bases=0):
        self._chk(self._L.mc_read_store_seek(self._h, int(at_bases), int(reserve_bases)))

    def read_store_tell(self):
        return int(self._L.mc_read_store_tell(self._h))

    def read_store_import_dev(self, d_words, n_words, at_bases):
        self._chk(self._L.mc_read_store_import_dev(self._h, _dptr(d_words), int(n_words), int(at_bases)))

    def share_read_store(self, other):
        """mc_share_read_store: this (BFS-only) context reads its look-ahead from `other`'s read store."""
        self._chk(self._L.mc_share_read_store(self._h, other._h if other is not None else None))

    def set_stream(self, stream_ptr):
        self._chk(self._L.mc_set_stream(self._h, C.c_void_p(stream_ptr) if stream_ptr else None))

    # ---- counting
    def add_reads_packed(self, words, offsets):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = len(offsets) - 1
        if n_reads > 0 and len(words) < (int(offsets[-1]) + 31) // 32 + 1:
            raise ValueError("words[] must hold ceil(n_bases/32) + 1 entries")
        self._chk(self._L.mc_add_reads_packed(self._h, _p(words, C.c_uint64), _p(offsets, C.c_uint64), n_reads))

    def add_reads_packed_dev(self, d_words, d_offsets, n_reads, n_bases):
        self._chk(self._L.mc_add_reads_packed_dev(self._h, _dptr(d_words), _dptr(d_offsets), n_reads, n_bases))

    def add_keys_dev(self, d_keys, n, d_hints=None):
        self._chk(self._L.mc_add_keys_dev(self._h, _dptr(d_keys), _dptr(d_hints), n))

    def add_pairs_dev(self, d_keys, d_counts, n, d_hints=None):
        self._chk(self._L.mc_add_pairs_dev(self._h, _dptr(d_keys), _dptr(d_counts), _dptr(d_hints), n))

    def solid_from_pairs_dev(self, d_keys, d_counts, n, min_cov, d_hints=None):
        """BFS-only context from the gathered (key, count, hint) pairs with count >= min_cov; returns how many."""
        m = C.c_uint64(0)
        self._chk(self._L.mc_solid_from_pairs_dev(self._h, _dptr(d_keys), _dptr(d_counts), _dptr(d_hints), n, min_cov, C.byref(m)))
        return int(m.value)

    def finalize(self):
        n = C.c_uint64(0)
        self._chk(self._L.mc_finalize_counts(self._h, C.byref(n)))
        return int(n.value)

    # ---- lookups
    def get(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.zeros(len(keys), dtype=np.int16)
        self._chk(self._L.mc_get(self._h, _p(keys, C.c_int64), len(keys), _p(out, C.c_int16)))
        return out

    def get_dev(self, d_keys, n, d_out):
        self._chk(self._L.mc_get_dev(self._h, _dptr(d_keys), n, _dptr(d_out)))

    def kmer_keys(self, hi, lo):
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        hi = np.ascontiguousarray(hi, dtype=np.uint64) if hi is not None else None
        out = np.zeros(len(lo), dtype=np.int64)
        self._chk(self._L.mc_kmer_keys(self._h, _p(hi, C.c_uint64) if hi is not None else None,
                                       _p(lo, C.c_uint64), len(lo), _p(out, C.c_int64)))
        return out

    # ---- BFS
    def bfs_batch(self, jobs, min_cov, max_kmers=-1, max_radius=-1):
        """jobs: list of (seed_hi or None, seed_lo, direction).  All passes run in one launch, one
        workgroup each.  Returns a list with, per job, None when no seed k-mer reaches min_cov (the
        reference's 'fail'), else a dict of numpy arrays in distanceToKmer insertion order."""
        n = len(jobs)
        arr_jobs = (_BfsJob * n)()
        keep = []
        for i, (hi, lo, d) in enumerate(jobs):
            lo = np.ascontiguousarray(lo, dtype=np.uint64)
            hi = np.ascontiguousarray(hi if hi is not None else np.zeros(len(lo)), dtype=np.uint64)
            keep.append((hi, lo))
            arr_jobs[i] = _BfsJob(_p(hi, C.c_uint64), _p(lo, C.c_uint64), len(lo), d)
        res = (_BfsResult * n)()
        self._chk(self._L.mc_bfs_batch(self._h, arr_jobs, n, min_cov, max_kmers, max_radius, res))
        out = []
        for i in range(n):
            r = res[i]
            m = int(r.n)
            if m == 0:
                out.append(None)
                continue

            # the arrays are views of the library's (page-locked) result memory, which goes back to the library when the
            # last of them is collected: no second copy of 10^5 vertices per pass
            owner = _ResultOwner(self._L, r)

            def arr(ptr, ctype, dt):
                buf = (ctype * m).from_address(C.addressof(ptr.contents))
                buf._owner = owner  # (the numpy array keeps `buf` alive as its base)
                return np.frombuffer(buf, dtype=dt)

            out.append(dict(hi=arr(r.hi, C.c_uint64, np.uint64), lo=arr(r.lo, C.c_uint64, np.uint64), dist=arr(r.dist, C.c_int32, np.int32),
                            cov=arr(r.cov, C.c_int16, np.int16), last=arr(r.last, C.c_uint8, np.uint8), levels=int(r.levels),
                            lookups=int(r.lookups), rounds=int(r.rounds), device_ms=float(r.device_ms)))
        return out

    def bfs(self, seed_hi, seed_lo, direction, min_cov, max_kmers=-1, max_radius=-1):
        """One runBfs pass; None when no seed k-mer reaches min_cov."""
        return self.bfs_batch([(seed_hi, seed_lo, direction)], min_cov, max_kmers, max_radius)[0]

    # ---- export
    def export(self, min_cov=0):
        n = C.c_uint64(0)
        self._chk(self._L.mc_export(self._h, min_cov, None, None, 0, C.byref(n)))
        cap = int(n.value)
        keys = np.zeros(cap, dtype=np.int64)
        cnt = np.zeros(cap, dtype=np.int16)
        if cap:
            self._chk(self._L.mc_export(self._h, min_cov, _p(keys, C.c_int64), _p(cnt, C.c_int16), cap, C.byref(n)))
        o = np.argsort(keys, kind="stable")
        return keys[o], cnt[o]

    def export_count(self, min_cov=0):
        n = C.c_uint64(0)
        self._chk(self._L.mc_export_dev(self._h, min_cov, None, None, None, 0, C.byref(n)))
        return int(n.value)

    def export_dev(self, min_cov, d_keys, d_counts, cap, d_hints=None):
        n = C.c_uint64(0)
        self._chk(self._L.mc_export_dev(self._h, min_cov, _dptr(d_keys), _dptr(d_counts), _dptr(d_hints), cap, C.byref(n)))
        return int(n.value)

    # ---- multi-GPU building blocks
    def extract_keys_dev(self, d_words, d_offsets, n_reads, n_bases, n_owners, d_keys, cap, d_hints=None):
        off = np.zeros(n_owners + 1, dtype=np.uint64)
        self._chk(self._L.mc_extract_keys_dev(self._h, _dptr(d_words), _dptr(d_offsets), n_reads, n_bases, n_owners,
                                              _dptr(d_keys), _dptr(d_hints), cap, _p(off, C.c_uint64)))
        return off

    def add_reads_file(self, path):
        """One --reads file (FASTA / FASTQ, optionally .gz) with the reference's reader policies; returns the reads added."""
        n = C.c_uint64(0)
        self._chk(self._L.mc_add_reads_file(self._h, os.fsencode(path), C.byref(n)))
        return int(n.value)

    def save_kmers(self, bin_path, stat_path=None, threshold=0):
        """<name>.kmers.bin (+ <name>.stat.txt): returns (keys in the table, records written)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.mc_save_kmers(self._h, os.fsencode(bin_path), os.fsencode(stat_path) if stat_path else None,
                                        threshold, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def load_kmers(self, path, freq_threshold=0):
        """Adds the records of a .kmers.bin file; returns (records read, records added)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.mc_load_kmers(self._h, os.fsencode(path), freq_threshold, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def superkmer_capacity(self, n_windows, n_reads):
        """Records to make room for when a batch of reads is split with extract_superkmers_dev; 0 when this
        context counts window by window (hash keys, k < 23): use extract_keys_dev then."""
        return int(self._L.mc_superkmer_capacity(self._h, int(n_windows), int(n_reads)))

    def extract_superkmers_dev(self, d_words, d_offsets, n_reads, n_bases, n_owners, d_recs, d_bins, cap):
        """d_recs: int64 tensor of shape (cap, 2); d_bins: int32 tensor of cap entries.  Returns the owner offsets."""
        off = np.zeros(n_owners + 1, dtype=np.uint64)
        self._chk(self._L.mc_extract_superkmers_dev(self._h, _dptr(d_words), _dptr(d_offsets), n_reads, n_bases, n_owners,
                                                    _dptr(d_recs), _dptr(d_bins), cap, _p(off, C.c_uint64)))
        return off

    def add_superkmers_dev(self, d_recs, d_bins, n):
        self._chk(self._L.mc_add_superkmers_dev(self._h, _dptr(d_recs), _dptr(d_bins), n))

    # ---- the binned form of the exchange (include/mcgpu.h mc_extract_superkmers_binned_dev): the sender does the owner's first level
    def superkmer_fine_buckets(self, n_owners):
        """fine buckets to extract with for n_owners owners laid out like this context; 0: use the flat form"""
        return int(self._L.mc_superkmer_fine_buckets(self._h, int(n_owners)))

    def extract_superkmers_binned_dev(self, d_words, d_offsets, n_reads, n_bases, n_owners, n_fine, d_recs, d_bins, cap, d_fine_counts):
        """as extract_superkmers_dev; d_fine_counts: int32 tensor of n_owners x n_fine entries (written).  Returns (owner offsets,
        windows of every owner's records)."""
        off = np.zeros(n_owners + 1, dtype=np.uint64)
        win = np.zeros(n_owners, dtype=np.uint64)
        self._chk(self._L.mc_extract_superkmers_binned_dev(self._h, _dptr(d_words), _dptr(d_offsets), n_reads, n_bases, n_owners, n_fine,
                                                           _dptr(d_recs), _dptr(d_bins), cap, _dptr(d_fine_counts), _p(off, C.c_uint64), _p(win, C.c_uint64)))
        return off, win

    def add_superkmers_binned_dev(self, d_recs, d_bins, n, n_windows, n_fine, part_offsets, d_part_counts):
        """part_offsets: n_parts + 1 record offsets (host); d_part_counts: int32 tensor of n_parts x n_fine entries"""
        po = np.ascontiguousarray(part_offsets, dtype=np.uint64)
        self._chk(self._L.mc_add_superkmers_binned_dev(self._h, _dptr(d_recs), _dptr(d_bins), n, int(n_windows), n_fine, len(po) - 1,
                                                       _p(po, C.c_uint64), _dptr(d_part_counts)))

    # ---- the walk over several ranks' tables in place (include/mcgpu.h mc_shard_*)
    SHARD_HANDLE_BYTES = 128

    def shard_export(self):
        """this context's table as 128 opaque bytes for the walking rank (after finalize; keep the table as it is meanwhile)"""
        buf = C.create_string_buffer(self.SHARD_HANDLE_BYTES)
        self._chk(self._L.mc_shard_export(self._h, buf))
        return buf.raw

    def shard_attach(self, handles, self_index, by_minimizer):
        """handles[i] = rank i's shard_export() (this context's own at self_index): bfs / bfs_batch then walk all the tables"""
        blob = b"".join(handles)
        assert len(blob) == self.SHARD_HANDLE_BYTES * len(handles)
        self._chk(self._L.mc_shard_attach(self._h, blob, len(handles), self_index, 1 if by_minimizer else 0))

    def shard_detach(self):
        self._chk(self._L.mc_shard_detach(self._h))

    # ---- measurement / synthetic data
    def stats(self):
        s = Stats()
        self._chk(self._L.mc_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._chk(self._L.mc_reset_stats(self._h))

    def trim(self):
        """mc_trim: the counting pipeline's scratch and the pools' idle blocks go back to the driver."""
        self._chk(self._L.mc_trim(self._h))

    def synth_reads_dev(self, genome_seed, n_contigs, contig_len, read_seed, first_read, n_reads, read_len,
                        err_per_10k, d_words, d_offsets):
        self._chk(self._L.mc_synth_reads_dev(self._h, genome_seed, n_contigs, contig_len, read_seed, first_read,
                                             n_reads, read_len, err_per_10k, _dptr(d_words), _dptr(d_offsets)))


def key_owner(key, n_owners):
    return int(load().mc_key_owner(int(key), n_owners))


def synth_genome(genome_seed, start, n):
    out = np.zeros(n, dtype=np.uint8)
    load().mc_synth_genome(genome_seed, start, n, _p(out, C.c_uint8))
    return out

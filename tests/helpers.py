"""Shared input builders for the parity tests (seeded, small enough for the CPU oracle)."""
import os

import numpy as np

from oracle import pyoracle as po

GENOME_SEED = 20240531
READ_SEED = 42


def synth_case(n_contigs, contig_len, n_reads, L=150, err=100, first_read=0):
    """Synthetic genome + fixed-length reads exactly as DESIGN.md 'Synthetic workload'."""
    genome = po.synth_genome(GENOME_SEED, n_contigs * contig_len)
    reads = po.synth_reads(genome, n_contigs, contig_len, READ_SEED, first_read, n_reads, L, err)
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * L
    return genome, reads, offsets


def ragged_case(rng, n_reads, max_len=220, genome_len=5000):
    """Reads of random length 0..max_len cut from a small random genome (so k-mers repeat)."""
    genome = rng.integers(0, 4, genome_len).astype(np.uint8)
    lens = rng.integers(0, max_len + 1, n_reads)
    lens[:4] = [0, 1, 30, 31]
    starts = [int(rng.integers(0, genome_len - l + 1)) for l in lens]
    reads = []
    for s, l in zip(starts, lens):
        r = genome[s:s + l]
        if rng.integers(0, 2):
            r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r)
    codes = np.concatenate(reads) if reads else np.zeros(0, dtype=np.uint8)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    return genome, codes, offsets


def oracle_table(codes, offsets, k, mode):
    t = po.Table()
    n = t.count_reads(codes, offsets, k, mode)
    return t, n


def seed_windows(codes, k):
    """All k-windows of a seed sequence as packed oriented k-mers (hi, lo)."""
    n = len(codes) - k + 1
    hi = np.zeros(max(n, 0), dtype=np.uint64)
    lo = np.zeros(max(n, 0), dtype=np.uint64)
    for i in range(max(n, 0)):
        v = 0
        for c in codes[i:i + k]:
            v = (v << 2) | int(c)
        hi[i] = v >> 64
        lo[i] = v & 0xFFFFFFFFFFFFFFFF
    return hi, lo


def assert_bfs_equal(got, want):
    assert (got is None) == (want is None)
    if got is None:
        return
    for f in ("hi", "lo", "dist", "cov", "last"):
        if not np.array_equal(got[f], want[f]):  # say where: the first entries that differ, with their distances
            g, w = np.asarray(got[f]), np.asarray(want[f])
            m = min(len(g), len(w))
            idx = np.nonzero(g[:m] != w[:m])[0][:8]
            raise AssertionError("%s: %d vs %d entries, first differences at %s: got %s want %s (dist %s, n=%d)" % (
                f, len(g), len(w), idx, g[idx], w[idx], np.asarray(want["dist"])[idx], len(w)))
    assert got["levels"] == want["levels"]


def hic_case_reads(ref_example_dir, n_loci=40):
    """Reads for the --hicseq tests (the reference's own test, tests/EnvironmentFinderMainTest.java:23-45, counts WGS reads
    that are not shipped): the example's plasmid (first 30 kb) tiled without errors at 7.5-fold coverage, which holds the
    --seq gene; and, since the shipped Hi-C sequences share no 31-mer with the plasmid, `n_loci` of them each inside a
    locus of its own (80 random bases on either side), tiled at up to 10-fold coverage that thins out towards the locus
    ends, so that a walk from a Hi-C seed runs into the flanks and stops where the coverage falls under the threshold."""
    import numpy as np
    from oracle import host_oracle as ho
    from oracle import pyoracle as po
    plasmid = ho.read_fasta_reads(os.path.join(ref_example_dir, "salmonella_pls.fasta"))[0][:30000]
    reads = [plasmid[s:s + 150] for s in range(0, len(plasmid) - 150 + 1, 20)]
    hic, _ = ho.rich_fasta_read(os.path.join(ref_example_dir, "selected_reads.fasta"))
    rng = np.random.default_rng(2024)
    for i in range(n_loci):
        h = hic[i * (len(hic) // n_loci)]
        locus = po.decode(rng.integers(0, 4, 80).astype(np.uint8)) + h + po.decode(rng.integers(0, 4, 80).astype(np.uint8))
        reads += [locus[s:s + 100] for s in range(0, len(locus) - 100 + 1, 10)]
    return reads

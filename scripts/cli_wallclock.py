"""Wall clock of the native CLI end to end (FASTA from the page cache -> graph.txt, seqs.fasta, graph.gfa, tsvs)
on a synthetic 30x data set.  Usage: [CLI_K=63] [NOHINT=1] [MC_LONG_RECORDS=0] python scripts/cli_wallclock.py [n_reads]"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
K = os.environ.get("CLI_K", "31")  # (above 31 the CLI hashes its k-mers: polynomial keys; NOHINT=1: without --capacity-hint)
L = 150
rng = np.random.default_rng(0)
lut = np.frombuffer(b"AGCT", dtype=np.uint8)
genome = rng.integers(0, 4, n * L // 30).astype(np.uint8)
starts = rng.integers(0, len(genome) - L, n)
tmp = os.environ.get("TMPDIR", "/tmp")
fa, seq = os.path.join(tmp, "cli_reads.fasta"), os.path.join(tmp, "cli_seed.fasta")
with open(fa, "wb") as f:
    for i in range(n):
        f.write(b">r%d\n%s\n" % (i, lut[genome[starts[i]:starts[i] + L]].tobytes()))
with open(seq, "wb") as f:
    f.write(b">seed\n%s\n" % lut[genome[100000:101000]].tobytes())
cli = os.path.join(ROOT, "metacherchant_amd", "lib", "metacherchant")
out, wd = os.path.join(tmp, "cli_out"), os.path.join(tmp, "cli_wd")
for rep in range(2):
    t0 = time.time()
    p = subprocess.run([cli, "--tool", "environment-finder", "-k", K, "--coverage", "5", "--reads", fa, "--seq", seq, "--output", out,
                        "--work-dir", wd, "--maxkmers", "100000", "--force"] + ([] if os.environ.get("NOHINT") else ["--capacity-hint", str(len(genome) + (1 << 20))]),
                       capture_output=True, text=True)
    t1 = time.time()
    assert p.returncode == 0, p.stderr[-2000:]
print("CLI wall clock %.2f s for %d reads (%.1f MB FASTA)" % (t1 - t0, n, os.path.getsize(fa) / 1e6))
print(open(os.path.join(wd, "metrics.json")).read().strip())
print("graph.txt lines:", sum(1 for _ in open(os.path.join(out, "seed", "graph.txt"))))

"""Randomised end-to-end comparison of the native CLI (C++ host + HIP library, on the GPU) with the oracle pipeline: random
read files (FASTA / FASTQ with quality splits and Ns, plain / .gz / .bz2), k, key mode, seeds (present, absent, several)
and options; every output file must be byte-identical.  Usage: python scripts/soak_cli.py [iterations] [seed]"""
import bz2
import gzip
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from metacherchant_amd import build
from oracle import host_oracle as ho
from oracle import pyoracle as po
from tests.helpers import synth_case
from tests.test_gpu_cli import _assert_same_tree

build.build_all()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def oracle_run(read_files, k, mode, seqs, comments, out_dir, **kw):
    t = po.Table()
    for p in read_files:
        inner = p[:-3] if p.endswith(".gz") else p[:-4] if p.endswith(".bz2") else p
        reads = ho.read_fastq_reads(p) if inner.endswith((".fastq", ".fq")) else ho.read_fasta_reads(p)
        if not reads:
            continue
        codes = np.concatenate([po.encode(r) for r in reads])
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads])
        t.count_reads(codes, off, k, mode)
    return t, ho.environment_finder(t, k, mode, seqs, comments, out_dir, **kw)


for it in range(iters):
    k = int(rng.choice([15, 21, 25, 31, 31, 41, 63]))
    hashed = k > 31 or bool(rng.integers(0, 5) == 0)
    hname = str(rng.choice(["poly", "fnv1a"]))
    mode = po.KEY_PACKED if not hashed else (po.KEY_POLY if hname == "poly" else po.KEY_FNV1A)
    L = int(rng.choice([100, 150]))
    n = int(rng.integers(2000, 9000))
    contigs, clen = int(rng.integers(1, 3)), int(rng.choice([8000, 20000, 40000]))
    genome, reads, _ = synth_case(contigs, clen, n, L, int(rng.choice([0, 50, 100])), first_read=it * 100000)
    with tempfile.TemporaryDirectory() as tmp:
        files, cut = [], n // 2 if rng.integers(0, 2) else n
        for fi, (a, b) in enumerate([(0, cut), (cut, n)]):
            if a == b:
                continue
            fastq = bool(rng.integers(0, 2))
            ext = str(rng.choice([".fastq", ".fq"])) if fastq else str(rng.choice([".fasta", ".fa", ".fna"]))
            comp = str(rng.choice(["", "", ".gz", ".bz2"]))
            path = os.path.join(tmp, "reads_%d%s%s" % (fi, ext, comp))
            lines = []
            for i in range(a, b):
                s = po.decode(reads[i * L:(i + 1) * L])
                if rng.integers(0, 9) == 0:
                    p = int(rng.integers(0, L))
                    s = s[:p] + "N" + s[p + 1:]
                if fastq:
                    q = ["I"] * L
                    if rng.integers(0, 6) == 0:
                        q[int(rng.integers(0, L))] = "!"
                    lines.append("@r%d\n%s\n+\n%s\n" % (i, s, "".join(q)))
                else:
                    lines.append(">r%d\n%s\n%s\n" % (i, s[:60], s[60:]))
            data = "".join(lines).encode()
            with open(path, "wb") as f:
                f.write(gzip.compress(data) if comp == ".gz" else bz2.compress(data) if comp == ".bz2" else data)
            files.append(path)
        seq = os.path.join(tmp, "genes.fasta")
        n_seq = int(rng.integers(1, 4))
        with open(seq, "w") as f:
            for si in range(n_seq):
                if rng.integers(0, 5) == 0:
                    body = po.decode(rng.integers(0, 4, 100).astype(np.uint8))  # absent from the reads
                else:
                    a = int(rng.integers(0, contigs * clen - 400))
                    body = po.decode(genome[a:a + int(rng.integers(k, 350))])
                f.write(">gene%d extra words\n%s\n" % (si, body))
        kw, extra = {}, []
        cov = int(rng.integers(1, 5)); kw["coverage"] = cov; extra += ["--coverage", str(cov)]
        if rng.integers(0, 2):
            mk = int(rng.choice([200, 2000, 50000])); kw["max_kmers"] = mk; extra += ["--maxkmers", str(mk)]
        if "max_kmers" not in kw or rng.integers(0, 2):
            mr = int(rng.choice([20, 100, 400])); kw["max_radius"] = mr; extra += ["--maxradius", str(mr)]
        if rng.integers(0, 2):
            kw["bothdirs"] = True; extra += ["--bothdirs"]
        if rng.integers(0, 2):
            kw["trim"] = True; extra += ["--trim"]
        if rng.integers(0, 3) == 0:
            kw["merge"] = True; extra += ["--merge"]
        hic_comments = None
        if rng.integers(0, 3) == 0:  # --hicseq: more seeds with --merge, only the directory names without it (EnvironmentFinderMain.java:149)
            hicf = os.path.join(tmp, "hic.fasta")
            n_hic = n_seq + int(rng.integers(0, 6))
            with open(hicf, "w") as f:
                for hi_ in range(n_hic):
                    a = int(rng.integers(0, contigs * clen - 200))
                    f.write(">hic%d of %d\n%s\n" % (hi_, it, po.decode(genome[a:a + int(rng.integers(k, 150))])))
            hic_seqs, hic_comments = ho.rich_fasta_read(hicf)
            extra += ["--hicseq", hicf]
            if kw.get("merge"):
                kw["hic_seqs"] = hic_seqs
        cl = int(rng.choice([1, 10, 50])); kw["chunk_length"] = cl; extra += ["--chunklength", str(cl)]
        if hashed:
            extra += ["--hash", hname] + ([] if k > 31 else ["--forcehash"])
        env_extra = {}
        if os.environ.get("SOAK_DEVICES") and rng.integers(0, 2) == 0:  # the native multi-device driver on shares of this GPU: same files out
            extra += ["--devices", ",".join(["0"] * int(rng.integers(2, 4)))]
            if rng.integers(0, 2) == 0:  # several exchanges a file; small tokeniser chunks
                env_extra = {"MC_GROUP_BATCH_READS": str(int(rng.choice([1024, 3000]))), "MC_TOKENIZER_CHUNK_BYTES": str(int(rng.choice([40000, 300000])))}
        out, want = os.path.join(tmp, "out"), os.path.join(tmp, "want")
        cmd = [build.CLI, "-k", str(k), "--reads"] + files + ["--seq", seq, "-o", out, "-w", os.path.join(tmp, "wd"), "--force"] + extra
        print("it %d: %s" % (it, " ".join(cmd[1:])), flush=True)
        if os.environ.get("SOAK_ORACLE_ONLY"):
            import time
            t0 = time.time()
            seqs, comments = ho.rich_fasta_read(seq)
            t, res = oracle_run(files, k, mode, seqs, comments, want, **kw)
            print("   oracle alone: %.1f s, table %d" % (time.time() - t0, t.size()), flush=True)
            for fpath in files:  # the host reader alone (no GPU)
                r = subprocess.run([build.HOSTTEST, "reads", fpath], capture_output=True, text=True, timeout=60)
                print("   host reader %s: rc %d, %d reads" % (os.path.basename(fpath), r.returncode, len(r.stdout.splitlines())), flush=True)
            continue
        if it < int(os.environ.get("SOAK_FROM", "0")):
            continue
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=int(os.environ.get("SOAK_CLI_TIMEOUT", "120")), env=dict(os.environ, **env_extra))
        except subprocess.TimeoutExpired as e:
            err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
            raise SystemExit("it %d: the CLI did not finish; its log ends:\n%s" % (it, err[-1500:]))
        assert p.returncode == 0, (it, cmd, p.stderr[-2000:])
        seqs, comments = ho.rich_fasta_read(seq)
        if hic_comments is not None:
            comments = hic_comments
        t, res = oracle_run(files, k, mode, seqs, comments, want, **kw)
        assert "Hashtable size: %d kmers" % t.size() in p.stderr, (it, t.size(), p.stderr[-500:])
        _assert_same_tree(res, out, want)
    print("it %d ok: k=%d %s files=%s %s" % (it, k, "hash:" + hname if hashed else "packed", [os.path.basename(x) for x in files], " ".join(extra)), flush=True)
print("cli soak ok: %d iterations" % iters)

"""metacherchant --devices 0,0,0 against one device on a small read set: where the walk's time goes (MC_INGEST_DEBUG prints)."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metacherchant_amd import build
from oracle import pyoracle as po
from tests.helpers import synth_case
build.build_all()
cli = os.path.join(ROOT, "metacherchant_amd", "lib", "metacherchant")
tmp = tempfile.mkdtemp()
genome, reads, _ = synth_case(2, 40000, 12000, 150, 60)
fa = os.path.join(tmp, "a.fasta")
with open(fa, "w") as f:
    for i in range(12000):
        f.write(">r%d\n%s\n" % (i, po.decode(reads[i * 150:(i + 1) * 150])))
seq = os.path.join(tmp, "g.fasta")
open(seq, "w").write(">g1\n%s\n" % po.decode(genome[5000:5400]))
for dev in (None, "0,0,0"):
    cmd = [cli, "-k", "31", "-i", fa, "--seq", seq, "-o", os.path.join(tmp, "o"), "-w", os.path.join(tmp, "w"), "--force", "--maxkmers", "4000", "--coverage", "3"]
    if dev:
        cmd += ["--devices", dev]
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MC_INGEST_DEBUG="1"))
    print(dev, p.returncode, open(os.path.join(tmp, "w", "metrics.json")).read().strip())
    print("\n".join(l for l in p.stderr.splitlines() if "[walk]" in l or "[bfs]" in l))

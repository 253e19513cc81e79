cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, subprocess, numpy as np
n=400000; L=150
rng=np.random.default_rng(0); lut=np.frombuffer(b"AGCT",dtype=np.uint8)
genome=rng.integers(0,4,n*L//30).astype(np.uint8); starts=rng.integers(0,len(genome)-L,n)
with open("/tmp/r.fasta","wb") as f:
    for i in range(n): f.write(b">r%d\n%s\n"%(i,lut[genome[starts[i]:starts[i]+L]].tobytes()))
with open("/tmp/s.fasta","wb") as f: f.write(b">seed\n%s\n"%lut[genome[100000:101000]].tobytes())
cli="metacherchant_amd/lib/metacherchant"
p=subprocess.run([cli,"--tool","environment-finder","-k","63","--coverage","5","--reads","/tmp/r.fasta","--seq","/tmp/s.fasta","--output","/tmp/o","--work-dir","/tmp/w","--maxkmers","100000","--force"],capture_output=True,text=True)
print("rc",p.returncode); print(p.stderr[-1500:]); print(p.stdout[-500:])
PY

#!/bin/bash
# Byte-for-byte comparison with the REAL reference, for a box that has what this image lacks:
# a JVM (>= 8) and the reference's own jar.  Nothing here is used by the tests or the product.
#
#   MC_REFERENCE_JAR=/path/to/metacherchant.jar scripts/compare_with_java.sh [n_reads] [k] [extra environment-finder flags...]
#
# Generates the synthetic workload of DESIGN.md section 3.4 (10 x 5 Mb contigs scaled down to n_reads at 30x), writes it
# as FASTA, runs `java -jar $MC_REFERENCE_JAR --tool environment-finder` and the native `metacherchant` on the same
# files with the same flags, and diffs graph.txt, graph.gfa, seqs.fasta and tsvs/* of every output directory.
# A third leg, when MC_PATCHED_JAR names a jar built from the reference with integration/patches/* applied and
# integration/java/gpu/McGpu.java + integration/jni/mcgpu_jni.c built (integration/README.md): the same command with
# MC_GPU_DEVICE=0, i.e. the original Java host over the C ABI, diffed against the other two.
# Exit status 0 = every file identical.  The Java log's timestamps around "Loading file" ... "Hashtable size" ...
# "Finished processing all sequences!" are printed as the reference's phase times on this box's cores.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
N=${1:-200000}; K=${2:-31}; shift $(( $# > 2 ? 2 : $# )) || true
EXTRA=("$@")
[ ${#EXTRA[@]} -eq 0 ] && EXTRA=(--coverage 5 --maxkmers 100000 --bothdirs False)
command -v java >/dev/null || { echo "no java on PATH: this script needs a JVM (the graft image has none)"; exit 2; }
[ -f "${MC_REFERENCE_JAR:-}" ] || { echo "set MC_REFERENCE_JAR to the reference's metacherchant.jar"; exit 2; }
CLI="$ROOT/metacherchant_amd/lib/metacherchant"
[ -x "$CLI" ] || python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()"
W="$(mktemp -d)"; trap 'rm -rf "$W"' EXIT
python3 - "$ROOT" "$W" "$N" <<'PY'
import sys
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
from oracle import pyoracle as po
L = 150
glen = max(20000, n * L // 30)
genome = po.synth_genome(20240531, glen)
reads = po.synth_reads(genome, 1, glen, 42, 0, n, L, 100)
with open(w + "/reads.fasta", "w") as f:
    for i in range(n):
        f.write(">r%d\n%s\n" % (i, po.decode(reads[i * L:(i + 1) * L])))
a = min(10000, glen // 2)
with open(w + "/seed.fasta", "w") as f:
    f.write(">seed\n%s\n" % po.decode(genome[a:a + 500]))
PY
CORES=$(nproc)
t0=$(date +%s.%N)
java -jar "$MC_REFERENCE_JAR" --tool environment-finder -k "$K" --reads "$W/reads.fasta" --seq "$W/seed.fasta" \
     --output "$W/java_out" --work-dir "$W/java_wd" -p "$CORES" --force "${EXTRA[@]}" > "$W/java.stdout" 2> "$W/java.log"
t1=$(date +%s.%N)
"$CLI" --tool environment-finder -k "$K" --reads "$W/reads.fasta" --seq "$W/seed.fasta" \
     --output "$W/hip_out" --work-dir "$W/hip_wd" --force "${EXTRA[@]}" 2> "$W/hip.log"
t2=$(date +%s.%N)
echo "reference (JVM, $CORES cores): $(echo "$t1 - $t0" | bc) s wall; native (MI355X): $(echo "$t2 - $t1" | bc) s wall"
grep -E "Loading file|Hashtable size|Finished processing" "$W/java_wd/log" 2>/dev/null | cut -c1-120 || true
rc=0
while IFS= read -r f; do
    rel="${f#$W/java_out/}"
    if cmp -s "$f" "$W/hip_out/$rel"; then echo "identical  $rel"; else echo "DIFFERENT  $rel"; rc=1; fi
done < <(find "$W/java_out" -type f \( -name graph.txt -o -name graph.gfa -o -name seqs.fasta -o -name '*.tsv' \) | sort)
if [ -f "${MC_PATCHED_JAR:-}" ]; then   # the Java host over the C ABI (integration/: McGpu.java, mcgpu_jni.c, patches)
    MC_GPU_DEVICE=0 java -Djava.library.path="$ROOT/metacherchant_amd/lib" -jar "$MC_PATCHED_JAR" --tool environment-finder -k "$K" \
         --reads "$W/reads.fasta" --seq "$W/seed.fasta" --output "$W/jni_out" --work-dir "$W/jni_wd" -p "$CORES" --force "${EXTRA[@]}" > "$W/jni.stdout" 2> "$W/jni.log"
    while IFS= read -r f; do
        rel="${f#$W/java_out/}"
        if cmp -s "$f" "$W/jni_out/$rel"; then echo "identical (JNI host)  $rel"; else echo "DIFFERENT (JNI host)  $rel"; rc=1; fi
    done < <(find "$W/java_out" -type f \( -name graph.txt -o -name graph.gfa -o -name seqs.fasta -o -name '*.tsv' \) | sort)
fi
[ $rc -eq 0 ] && echo "all output files byte-identical with the reference" || echo "MISMATCH: see above"
exit $rc

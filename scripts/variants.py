#!/usr/bin/env python3
"""Tuning builds of libmcgpu.so next to the product library, and a runner that benches each of them in one gpurun call.

  python scripts/variants.py build name1:-DA=1,-DB=2 name2:-DC=3 ...   (here: hipcc cross-compiles, 4 at a time)
  python scripts/variants.py run [bench args...]                       (on the GPU box: every lib/libmcgpu_*.so + the product)

`run` prints one line per library: ms per step, count / BFS split and the pipeline kernels' times."""
import glob
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metacherchant_amd import build as b  # noqa: E402


def do_build(specs):
    def one(spec):
        name, _, defs = spec.partition(":")
        defines = [d[2:] if d.startswith("-D") else d for d in defs.split(",") if d]
        return b.build_lib(force=True, verbose=True, variant=name, defines=defines)
    with ThreadPoolExecutor(4) as ex:
        for out in ex.map(one, specs):
            print("built", out)


def do_run(args):
    only = os.environ.get("MC_VARIANTS")
    libs = [b.LIB] + sorted(glob.glob(os.path.join(b.LIBDIR, "libmcgpu_*.so")))
    for lib in libs:
        name = os.path.basename(lib)[len("libmcgpu"):-3].lstrip("_") or "product"
        if only and name not in only.split(","):
            continue
        env = dict(os.environ, MC_LIB=lib)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--skip-no-hint"] + args, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode or not line:
            print("%-16s FAILED rc=%d %s" % (name, p.returncode, (p.stderr or p.stdout)[-400:].replace("\n", " | ")), flush=True)
            continue
        j = json.loads(line[-1])
        r = j["roofline"] or {}
        print("%-16s %.2f ms/step  count %.2f  bfs %.2f  %s  distinct %d reached %d" % (
            name, j["ms_per_step"], r.get("count_ms_per_step", 0), j["bfs"]["ms_per_step"],
            " ".join("%s=%.2f" % (k.replace("k_", ""), v) for k, v in (r.get("kernel_ms") or {}).items()),
            j["distinct_kmers"], j["bfs"]["reached"]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "build":
        do_build(sys.argv[2:])
    elif len(sys.argv) >= 2 and sys.argv[1] == "run":
        do_run(sys.argv[2:])
    else:
        raise SystemExit(__doc__)

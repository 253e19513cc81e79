"""Builds the native parts in-tree: libmcgpu.so (HIP, gfx950) and the C++ host tool.

hipcc cross-compiles without a GPU.  Outputs land in metacherchant_amd/lib/ (git-ignored, but they
travel to the GPU box with the gpurun snapshot).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmcgpu.so")
CLI = os.path.join(LIBDIR, "metacherchant")

HIP_SOURCES = ["mcgpu.hip", os.path.join("host", "envfinder.cpp")]  # (the read-file entry point uses the host reader)
# every header under csrc/ (mcgpu.hip includes them all; a stale library would travel to the GPU box unnoticed)
HIP_DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("host", "envfinder.h"), os.path.join("test", "bfs_old_race.h"),
                                                                      os.path.join(ROOT, "include", "mcgpu.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libmcgpu.so)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def build_lib(force=False, verbose=False, variant=None, defines=()):
    """variant/defines: a tuning build next to the product library (lib/libmcgpu_<variant>.so, compiled with the given
    -D flags); MC_LIB=<path> makes native.load() use it (scripts/variants.py)."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [d if os.path.isabs(d) else os.path.join(CSRC, d) for d in HIP_DEPS]
    out = LIB if not variant else os.path.join(LIBDIR, "libmcgpu_%s.so" % variant)
    if not force and not _stale(out, deps):
        return out
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
           "-I", os.path.join(ROOT, "include"), "-o", out] + ["-D" + d for d in defines] + srcs + ["-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


HOSTTEST = os.path.join(LIBDIR, "mc_hosttest")


def build_host(force=False, verbose=False):
    """C++ host side: the `metacherchant` CLI (links libmcgpu.so) and the CPU-only `mc_hosttest`."""
    hdir = os.path.join(CSRC, "host")
    os.makedirs(LIBDIR, exist_ok=True)
    common = [os.path.join(hdir, "envfinder.cpp")]
    hdrs = [os.path.join(hdir, "envfinder.h"), os.path.join(ROOT, "include", "mcgpu.h")]
    flags = ["-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include")]
    if force or _stale(HOSTTEST, common + hdrs + [os.path.join(hdir, "hosttest.cpp")]):
        cmd = ["g++"] + flags + ["-o", HOSTTEST, os.path.join(hdir, "hosttest.cpp")] + common + ["-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if os.path.exists(LIB) and (force or _stale(CLI, common + hdrs + [os.path.join(hdir, "main.cpp"), LIB])):
        cmd = ["g++"] + flags + ["-o", CLI, os.path.join(hdir, "main.cpp")] + common + [
            "-L", LIBDIR, "-lmcgpu", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


def build_host_sanitized(kind, force=False, verbose=False):
    """mc_hosttest under a CPU sanitizer (kind: "asan" = address + undefined behaviour, "tsan" = threads): the host code --
    readers on several threads, the replay of java.util.HashMap with its tree bins, compaction, writers -- run by
    tests/test_host_sanitizers.py.  (No GPU sanitizer runs on this pool; the HIP side has the device self-check instead.)"""
    hdir = os.path.join(CSRC, "host")
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "mc_hosttest_" + kind)
    srcs = [os.path.join(hdir, "hosttest.cpp"), os.path.join(hdir, "envfinder.cpp")]
    hdrs = [os.path.join(hdir, "envfinder.h"), os.path.join(ROOT, "include", "mcgpu.h")]
    san = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "tsan": ["-fsanitize=thread"]}[kind]
    if force or _stale(out, srcs + hdrs):
        cmd = ["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include")] + san + [
            "-o", out] + srcs + ["-lz", "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


# Tuning builds the GPU tests and scripts load with MC_LIB.  `fuzz` (the walk with a pause behind every barrier) is what the
# gating test of tests/test_gpu_bfs_race.py runs: build_all makes it beside the product library.  The others are made when
# somebody asks for them (build_variant): `fuzz_old` = round 3's racy walk fuzzed the same way (the opt-in half of that test,
# MC_RUN_OLD_RACE=1), `trace_old` and `sctime` what scripts/gpu_bfs_hunt.sh and scripts/gpu_r4_bfs.sh load.
VARIANTS = {
    "fuzz": ("MC_BFS_FUZZ", "MC_BFS_TRACE"),
    "fuzz_old": ("MC_BFS_FUZZ", "MC_BFS_TRACE", "MC_BFS_OLD_RACE"),
    "trace_old": ("MC_BFS_TRACE", "MC_BFS_OLD_RACE"),
    "sctime": ("MC_SCOUT_TIMING",),
}
DEFAULT_VARIANTS = ("fuzz",)


def build_variant(name, force=False, verbose=False):
    """lib/libmcgpu_<name>.so for a name of VARIANTS (a minute of hipcc when it is missing or stale)"""
    return build_lib(force, verbose, variant=name, defines=VARIANTS[name])


def build_variants(force=False, verbose=False, names=DEFAULT_VARIANTS):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(len(names), 1)) as ex:
        return list(ex.map(lambda n: build_variant(n, force, verbose), names))


def build_all(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:  # (hipcc runs a minute per library: the product and the fuzzed walk side by side)
        v = ex.submit(build_variants, force, verbose)
        build_lib(force, verbose)
        v.result()
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force=True, verbose=True)

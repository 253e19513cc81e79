"""The CPU side under sanitizers (`-m "not gpu"`): the C++ host code (csrc/host/envfinder.cpp: readers on
several threads, the replay of java.util.HashMap with its tree bins and their removals, trim, compaction, the writers,
environment-finder-multi) built with AddressSanitizer + UndefinedBehaviorSanitizer runs every test of tests/test_host_cpp.py
and the joins of tests/test_multi_string_model.py; the parallel ingest runs under ThreadSanitizer; and the C oracle
(oracle/mc_oracle.c, its multi-threaded counting port included) runs the golden tests as an ASan build loaded into a child
interpreter.  A report of any of them fails the run: the sanitizers exit non-zero (halt_on_error, abort on undefined
behaviour), and every caller checks exit codes.  GPU sanitizers do not exist on this pool: the device side has
MC_BFS_SELFCHECK and the parity suites instead."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child_pytest(args, env):
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def test_host_code_under_address_and_ub_sanitizers():
    from metacherchant_amd import build
    exe = build.build_host_sanitized("asan")
    out = _child_pytest(["tests/test_host_cpp.py", "tests/test_multi_string_model.py"],
                        {"MC_HOSTTEST": exe, "ASAN_OPTIONS": "halt_on_error=1:detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    assert " passed" in out and "failed" not in out


def test_parallel_ingest_under_thread_sanitizer():
    from metacherchant_amd import build
    exe = build.build_host_sanitized("tsan")
    out = _child_pytest(["tests/test_host_cpp.py", "-k", "parallel_ingest or reads"], {"MC_HOSTTEST": exe, "TSAN_OPTIONS": "halt_on_error=1:exitcode=66"})
    assert " passed" in out and "failed" not in out


def test_c_oracle_under_address_sanitizer():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libmcoracle_asan.so"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan.so to preload")
    out = _child_pytest(["tests/test_oracle_golden.py", "tests/test_oracle_pins.py"],
                        {"MCO_LIB": os.path.join(ROOT, "oracle", "libmcoracle_asan.so"), "LD_PRELOAD": libasan,
                         "ASAN_OPTIONS": "halt_on_error=1:detect_leaks=0", "UBSAN_OPTIONS": "halt_on_error=1"})  # (the interpreter itself leaks by design)
    assert " passed" in out and "failed" not in out

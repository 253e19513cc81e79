"""CPU: the C-ABI shared library loads and exports every entry point include/mcgpu.h declares,
and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from metacherchant_amd import build, native
    build.build_lib()
    return native.load()


def _declared():
    text = open(os.path.join(ROOT, "include", "mcgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from metacherchant_amd import native
    names = _declared()
    assert len(names) >= 24
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(native.EXPORTS) == names  # the Python binding tracks the header
    assert lib.mc_abi_version() == 14


def test_host_only_entry_points_work_without_gpu(lib):
    from metacherchant_amd import native
    import numpy as np
    assert all(0 <= native.key_owner(k, 8) < 8 for k in (0, 1, -5, 2**62, -2**63))
    assert native.key_owner(12345, 1) == 0
    g = native.synth_genome(20240531, 0, 1000)
    from oracle import pyoracle as po
    assert np.array_equal(g, po.synth_genome(20240531, 1000))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import metacherchant_amd as m
    with pytest.raises(m.McError) as e:
        m.Context(31)
    assert "no HIP device" in str(e.value)

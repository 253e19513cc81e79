"""metacherchant_amd -- MI355X-native implementation of MetaCherchant's environment-finder hot path.

The product is libmcgpu.so (hand-written HIP for gfx950 behind the C ABI of include/mcgpu.h) plus
the C++ host tool; this package only binds it for tests, benchmarks and multi-GPU orchestration.
"""
from . import native  # noqa: F401
from .native import KEY_FNV1A, KEY_PACKED, KEY_POLY, Context, McError  # noqa: F401

__all__ = ["native", "Context", "McError", "KEY_PACKED", "KEY_POLY", "KEY_FNV1A"]

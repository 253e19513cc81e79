"""The hot kernels' compiled resources, read from the code object inside libmcgpu.so (no GPU needed).  A kernel of the walk that
touches scratch memory is a regression: in round 4 one out-of-line call (a function taking the walk's state by reference) put the
kernel arguments into scratch for the whole of k_bfs, every t.slots / t.reads became a scratch load, and the walk went from 9.4
to 10.7 ms before anyone looked at `private_segment_fixed_size`."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_notes(tmp_path):
    from metacherchant_amd import build
    lib = build.build_lib()
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm's llvm tools are not here")
    fat, co = str(tmp_path / "fatbin"), str(tmp_path / "k.co")
    subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib, str(tmp_path / "stripped.so")])
    subprocess.check_call([tools[1], "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    text = subprocess.check_output([tools[2], "--notes", co], text=True)
    kernels = {}
    for block in text.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|group_segment_fixed_size|vgpr_spill_count):\s+(\d+)", block)}
    return kernels


def test_the_walk_and_the_merge_kernel_use_no_scratch_memory(tmp_path):
    kernels = _kernel_notes(tmp_path)
    walk = {n: r for n, r in kernels.items() if "k_bfsILi" in n or "k_bfs_scoutILi" in n}
    assert len(walk) == 12  # the walk and its scouts' kernel: three key modes x (one table, several ranks' tables) each
    for name, r in list(walk.items()) + [(n, r) for n, r in kernels.items() if "k_p3_dedup" in n or "k_sk1w_extract" in n]:
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, (name, r)
    # the merge kernel's two workgroups a CU: 2 x 81.8 KB of the CU's 160 KB of LDS, and at most 128 registers for 4 waves a SIMD
    for name, r in kernels.items():
        if "k_p3_dedup" in name:
            assert 2 * r["group_segment_fixed_size"] <= 160 * 1024 and r["vgpr_count"] <= 128, (name, r)


def test_the_key_join_kernels_use_no_scratch_and_fit_their_workgroups(tmp_path):
    """csrc/dup_check.h: level 2 keeps two workgroups of 1024 threads on a CU (80 KB of LDS, 64 registers), level 3 five of 256."""
    kernels = _kernel_notes(tmp_path)
    dup = {n: r for n, r in kernels.items() if "k_dup_" in n or "k_pq_" in n or "k_phantom_queries" in n or "k_dupq_" in n}
    assert len(dup) >= 9, sorted(dup)
    for name, r in dup.items():
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, (name, r)
        if "k_dup_scatter" in name:
            assert 2 * r["group_segment_fixed_size"] <= 160 * 1024 and r["vgpr_count"] <= 64, (name, r)
        if "k_dup_findILi13E" in name:
            assert 4 * r["group_segment_fixed_size"] <= 160 * 1024 and r["vgpr_count"] <= 128, (name, r)
        if "k_dup_findILi15E" in name:  # (one workgroup of 1024 threads a CU: 4 waves a SIMD)
            assert r["group_segment_fixed_size"] <= 160 * 1024 and r["vgpr_count"] <= 128, (name, r)


def test_the_long_record_kernels_fit_two_workgroups_a_cu(tmp_path):
    """csrc/count_long.h: the first level and the merge kernel of long records use no scratch memory, and the merge kernel's
    region image, chunk list and record table leave room for two workgroups a CU (160 KB of LDS, 128 registers a wave at 4 a SIMD)."""
    kernels = _kernel_notes(tmp_path)
    long_k = {n: r for n, r in kernels.items() if "k_p3_longILi" in n or "k_skl_extract" in n or "k_sk2_scatter_compactILi2ELi2E" in n}
    assert sum("k_p3_longILi" in n for n in long_k) == 5 and len(long_k) == 7, sorted(long_k)  # built for k = 63, 55, 47, 41, and for any k
    for name, r in long_k.items():
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, (name, r)
        if "k_p3_long" in name:
            assert 2 * r["group_segment_fixed_size"] <= 160 * 1024 and r["vgpr_count"] <= 128, (name, r)


def test_the_binned_exchange_kernels_use_no_scratch_and_fit_a_cu(tmp_path):
    """The multi-GPU record exchange in its binned form (count_pipeline.h): the first level with its 64 KB histogram of (owner, fine bucket)
    cells beside the 62 KB it has anyway, and the staged second level in its three forms -- segments where the first level left them,
    segments where they were received (`in listed`), exact output places (`out listed`) -- one workgroup of 1024 threads a CU each."""
    kernels = _kernel_notes(tmp_path)
    staged = {n: r for n, r in kernels.items() if "k_sk2_scatter_stagedILi" in n}
    assert len(staged) == 3, sorted(staged)
    binned = {n: r for n, r in kernels.items() if "k_sk1w_extractILb1ELb0ELb1E" in n}
    assert len(binned) == 1, sorted(kernels)
    for name, r in list(staged.items()) + list(binned.items()) + [(n, r) for n, r in kernels.items() if "k_skb_" in n]:
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, (name, r)
        assert r["group_segment_fixed_size"] <= 160 * 1024 and r["vgpr_count"] <= 128, (name, r)

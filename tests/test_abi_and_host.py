"""CPU-only checks: the C-ABI library loads and exports every symbol include/orbhip.h declares,
error paths that need no GPU behave, and the host-side helpers (sharding, byte model) are right."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "orbhip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbhip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from visual_slam_amd import orbhip
    L = orbhip.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libORBhip.so does not export %s" % n
    assert sorted(orbhip.EXPORTS) == names


def test_strerror_and_argument_errors_need_no_gpu():
    from visual_slam_amd import orbhip
    L = orbhip.lib()
    assert L.orbhip_strerror(0) == b"ok"
    assert b"capacity" in L.orbhip_strerror(orbhip.ERR_CAPACITY)
    h = C.c_void_p()
    assert L.orbhip_create(None, C.byref(h)) == orbhip.ERR_ARG
    bad = orbhip.Params(0, 1.2, 8, 20, 7, 0, 0, 0)
    assert L.orbhip_create(C.byref(bad), C.byref(h)) == orbhip.ERR_ARG
    bad = orbhip.Params(1000, 1.0, 8, 20, 7, 0, 0, 0)
    assert L.orbhip_create(C.byref(bad), C.byref(h)) == orbhip.ERR_ARG
    assert L.orbhip_destroy(None) == orbhip.ERR_ARG
    assert L.orbhip_get_levels(None) == orbhip.ERR_ARG


def test_no_cpu_fallback_without_device(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    from visual_slam_amd import orbhip
    with pytest.raises(orbhip.OrbHipError) as ei:
        orbhip.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ei.value.code in (orbhip.ERR_NO_DEVICE, orbhip.ERR_HIP)


def test_keypoint_record_layout():
    from visual_slam_amd import orbhip
    assert orbhip.KP_DTYPE.itemsize == 28 and orbhip.KP_DTYPE.names == ("x", "y", "size", "angle", "response", "octave", "class_id")
    assert C.sizeof(orbhip.Params) == 32


def test_shard_ranges_partition_the_batch():
    from visual_slam_amd.sharding import shard_range
    for n, world, group in [(1024, 8, 1), (512, 8, 2), (10, 4, 2), (3, 8, 1), (0, 2, 1), (1026, 4, 2)]:
        spans = [shard_range(n, world, r, group) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert all(s % group == 0 for s in sizes) and max(sizes) - min(sizes) <= group
    with pytest.raises(ValueError):
        shard_range(5, 2, 0, group=2)


def test_algorithmic_byte_model():
    import bench
    expect = {"C1_tum": 4352822, "C2_kitti": 6621136, "C3_euroc": 5101414, "C4_1080p_2000f": 28246032, "C5_4k_8000f": 181243512}
    for name, total in expect.items():
        w, h, nf, sc, nl, _, _ = bench.WORKLOADS[name]
        parts, tot = bench.algorithmic_bytes(w, h, sc, nl, nf)
        assert tot == total == sum(parts.values())

"""Golden vectors (tests/golden/*.npz, written by tools/gen_golden.py with the CPU oracle).

CPU: the oracle still reproduces them. GPU: the HIP path reproduces them through the C ABI.
"""
import zlib
from pathlib import Path

import numpy as np
import pytest

GOLDEN = sorted((Path(__file__).resolve().parent / "golden").glob("*.npz"))


def _params(g):
    p = g["params"]
    return int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4])


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_oracle_reproduces_golden(path):
    from oracle import orb_oracle as O
    g = np.load(path)
    prm = _params(g)
    e = O.OracleExtractor(*prm)
    kp, desc = e.extract(g["image"])
    assert np.array_equal(kp, g["keypoints"]) and np.array_equal(desc, g["descriptors"])
    assert [zlib.crc32(e.level(l, with_border=True).tobytes()) for l in range(prm[2])] == g["level_crc32"].tolist()
    assert [len(e.candidates(l)) for l in range(prm[2])] == g["candidates_per_level"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_hip_reproduces_golden(path):
    from tests.parity import DESC_MIN_EXACT, compare_outputs
    from visual_slam_amd import orbhip
    g = np.load(path)
    prm = _params(g)
    ex = orbhip.ORBextractor(*prm)
    kp, desc = ex(g["image"])
    r = compare_outputs(kp, desc, g["keypoints"], g["descriptors"])
    assert r["desc_exact"] >= DESC_MIN_EXACT, r
    assert [zlib.crc32(ex.pyramid_level(l, with_border=True).tobytes()) for l in range(prm[2])] == g["level_crc32"].tolist()
    assert [len(ex.debug_candidates(l)) for l in range(prm[2])] == g["candidates_per_level"].tolist()

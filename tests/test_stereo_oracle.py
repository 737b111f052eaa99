"""Stereo matching (reference src/Frame.cc:481-653): properties of the CPU oracle on synthetic pairs
with known disparities, and the golden fixtures under tests/golden/stereo/ (written by
tools/gen_golden.py with the oracle; the reference has no fixtures for this step either)."""
from pathlib import Path

import numpy as np
import pytest

from visual_slam_amd import synth

GOLDEN = sorted((Path(__file__).resolve().parent / "golden" / "stereo").glob("*.npz"))
FX, BF = 718.856, 386.1448


def _params(g):
    p = g["params"]
    return int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4])


def _oracle_pair(left, right, prm, mb=BF / FX, mbf=BF):
    from oracle import orb_oracle as O
    eL, eR = O.OracleExtractor(*prm), O.OracleExtractor(*prm)
    kl, dl = eL.extract(left)
    kr, dr = eR.extract(right)
    return kl, kr, O.stereo_match(eL, eR, kl, dl, kr, dr, mb, mbf)


def test_uniform_shift_gives_that_disparity():
    d = 7
    left, right = synth.stereo_pair("blocks", 480, 360, 31, (d,))
    kl, kr, (ur, dp, kept) = _oracle_pair(left, right, (600, 1.2, 6, 20, 7))
    m = ur >= 0
    assert kept == m.sum() and kept > 100
    disp = kl["x"][m] - ur[m]
    # the SAD window moves in whole pixels of the keypoint's level and the parabola adds at most one
    scale = np.float32(1.2) ** kl["octave"][m]
    assert np.all(np.abs(disp - d) <= 2.0 * scale)
    assert np.median(np.abs(disp - d)) < 0.5
    # depth = mbf / disparity (src/Frame.cc:634) in float32
    assert np.array_equal(dp[m], np.float32(BF) / disp.astype(np.float32))
    assert np.all(dp[~m] == -1)


def test_no_match_when_right_image_is_flat():
    left = synth.frame("blocks", 320, 240, 32)
    right = np.full_like(left, 90)
    kl, kr, (ur, dp, kept) = _oracle_pair(left, right, (300, 1.2, 4, 20, 7))
    assert len(kr) == 0 and kept == 0 and np.all(ur == -1) and np.all(dp == -1)


def test_matches_beyond_max_disparity_are_dropped():
    # maxD = mbf / mb = fx: with fx = 10 a 20-px disparity is out of range (src/Frame.cc:513,531,553)
    left, right = synth.stereo_pair("blocks", 320, 240, 33, (20,))
    kl, kr, (ur, dp, kept) = _oracle_pair(left, right, (300, 1.2, 4, 20, 7), mb=np.float32(0.5), mbf=np.float32(5.0))
    m = ur >= 0
    assert np.all(kl["x"][m] - ur[m] < 10.0)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_oracle_reproduces_stereo_golden(path):
    g = np.load(path)
    kl, kr, (ur, dp, kept) = _oracle_pair(g["left"], g["right"], _params(g), float(g["mb"]), float(g["mbf"]))
    assert kept == int(g["kept"])
    assert np.array_equal(ur.view(np.uint32), g["uright"].view(np.uint32))
    assert np.array_equal(dp.view(np.uint32), g["depth"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_hip_reproduces_stereo_golden(path):
    from visual_slam_amd import orbhip
    g = np.load(path)
    prm = _params(g)
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    kl, dl = gL(g["left"], fetch_pyramid=False)
    kr, dr = gR(g["right"], fetch_pyramid=False)
    ur, dp, kept = orbhip.ComputeStereoMatches(gL, gR, kl, dl, kr, dr, float(g["mb"]), float(g["mbf"]))
    assert kept == int(g["kept"])
    assert np.array_equal(ur.view(np.uint32), g["uright"].view(np.uint32))
    assert np.array_equal(dp.view(np.uint32), g["depth"].view(np.uint32))

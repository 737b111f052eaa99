"""ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:402-517): oracle properties on a
frame pair with a known image motion, and GPU parity."""
import numpy as np
import pytest

from visual_slam_amd import synth

KITTI = (718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0, 0.0, 0.0)
TUM1 = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314)


def shifted(img, dx, dy):
    """The scene moved by (dx, dy) pixels (content that leaves the frame is replaced by the old border)."""
    out = img.copy()
    h, w = img.shape
    out[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)] = img[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)]
    return out


def frames(w, h, index, dx, dy, cam, prm=(1000, 1.2, 8, 20, 7)):
    from oracle import orb_oracle as O
    a = synth.frame("blocks", w, h, index)
    b = shifted(a, dx, dy)
    e = O.OracleExtractor(*prm)
    k1, d1 = e.extract(a)
    k2, d2 = e.extract(b)
    bounds = O.image_bounds(cam, w, h)
    u1 = O.undistort_grid(cam, bounds, k1)[0]
    u2 = O.undistort_grid(cam, bounds, k2)[0]
    prev = np.stack([u1["x"], u1["y"]], axis=1)             # Tracking seeds vbPrevMatched with F1's keypoints (src/Tracking.cc:569-571)
    return (u1, d1), (u2, d2), bounds, prev


def test_oracle_recovers_a_known_shift():
    from oracle import orb_oracle as O
    (u1, d1), (u2, d2), bounds, prev = frames(640, 480, 61, 13, -8, KITTI)
    m12, prev2, n = O.search_for_initialization(u1, d1, u2, d2, bounds, prev)
    assert n == (m12 >= 0).sum() and n > 100
    m = m12 >= 0
    assert np.all(u1["octave"][m] == 0) and np.all(u2["octave"][m12[m]] == 0)      # level-0 keypoints only
    dx = u2["x"][m12[m]] - u1["x"][m]
    dy = u2["y"][m12[m]] - u1["y"][m]
    assert np.mean((np.abs(dx - 13) <= 1) & (np.abs(dy + 8) <= 1)) > 0.95
    # a match index is used once, and vbPrevMatched follows the matches
    assert len(np.unique(m12[m])) == m.sum()
    assert np.array_equal(prev2[m, 0], u2["x"][m12[m]]) and np.array_equal(prev2[~m], prev[~m])
    # nothing inside a 5-px window when the motion is 13 px
    m12s, _, ns = O.search_for_initialization(u1, d1, u2, d2, bounds, prev, window=5)
    assert ns < n // 4


@pytest.mark.gpu
@pytest.mark.parametrize("cam,w,h,dx,dy,ori", [(KITTI, 1241, 376, 9, 3, True), (TUM1, 640, 480, -17, 11, True), (KITTI, 752, 480, 30, 0, False)])
def test_hip_search_for_initialization_matches_oracle(cam, w, h, dx, dy, ori):
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    (u1, d1), (u2, d2), bounds, prev = frames(w, h, 62, dx, dy, cam, (2000, 1.2, 8, 20, 7))
    om, op, on = O.search_for_initialization(u1, d1, u2, d2, bounds, prev, 100, 0.9, ori)
    ex = orbhip.ORBextractor(2000, 1.2, 8, 20, 7)
    gm, gp, gn = orbhip.SearchForInitialization(ex, u1, d1, u2, d2, bounds, prev, 100, 0.9, ori)
    assert on > 100 and gn == on
    assert np.array_equal(gm, om)
    assert np.array_equal(gp.view(np.uint32), op.view(np.uint32))
    # a second round, as Tracking does frame after frame with the updated vbPrevMatched
    om2, op2, on2 = O.search_for_initialization(u1, d1, u2, d2, bounds, op, 100, 0.9, ori)
    gm2, gp2, gn2 = orbhip.SearchForInitialization(ex, u1, d1, u2, d2, bounds, gp, 100, 0.9, ori)
    assert gn2 == on2 and np.array_equal(gm2, om2)

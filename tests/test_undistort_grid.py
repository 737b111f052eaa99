"""Keypoint undistortion and the 64x48 feature grid (reference Frame::UndistortKeyPoints,
Frame::AssignFeaturesToGrid, Frame::ComputeImageBounds: src/Frame.cc:422-478, 255-270, 403-411)."""
import numpy as np
import pytest

from visual_slam_amd import synth

TUM1 = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314)  # Examples/Monocular/TUM1.yaml
KITTI = (718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0, 0.0, 0.0)                                        # Examples/Stereo/KITTI00-02.yaml


def test_oracle_undistort_inverts_the_distortion_model():
    """Known-answer: distort ideal pixels with the forward model, undistortPoints must return them."""
    from oracle import orb_oracle as O
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = TUM1
    rng = np.random.default_rng(3)
    ideal = np.stack([rng.uniform(40, 600, 500), rng.uniform(40, 440, 500)], axis=1)
    x, y = (ideal[:, 0] - cx) / fx, (ideal[:, 1] - cy) / fy
    r2 = x * x + y * y
    radial = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    distorted = np.stack([xd * fx + cx, yd * fy + cy], axis=1).astype(np.float32)
    back = O.undistort_points(TUM1, distorted)
    # five iterations of the fixed point leave a small residual that grows with the radius
    assert np.abs(back - ideal).max() < 0.15
    centre = r2 < 0.1
    assert np.abs(back[centre] - ideal[centre]).max() < 2e-3
    # no distortion: the identity up to float rounding of (u - cx) / fx * fx + cx
    same = O.undistort_points(KITTI, distorted)
    assert np.abs(same - distorted).max() < 1e-3


def test_oracle_bounds_and_grid_rules():
    from oracle import orb_oracle as O
    assert O.image_bounds(KITTI, 1241, 376).tolist() == [0.0, 1241.0, 0.0, 376.0]
    b = O.image_bounds(TUM1, 640, 480)
    assert 0 < b[0] < 30 and 610 < b[1] < 640 and 0 < b[2] < 30 and 450 < b[3] < 480   # k1 > 0: the corners move inwards
    # grid: 3 keypoints in known cells of an undistorted 640x480 frame (cell = round(x * 64/640), round(y * 48/480))
    kp = np.zeros(4, O.KP_DTYPE)
    kp["x"] = [5.0, 5.0, 320.0, 639.9]
    kp["y"] = [5.0, 5.0, 240.0, 479.9]
    un, gs, gi = O.undistort_grid(KITTI, O.image_bounds(KITTI, 640, 480), kp)
    assert np.array_equal(un, kp)
    cell = lambda x, y: x * 48 + y
    assert gi[gs[cell(1, 1)]:gs[cell(1, 1) + 1]].tolist() == [0, 1]      # round(0.5) = 1: half away from zero
    assert gi[gs[cell(32, 24)]:gs[cell(32, 24) + 1]].tolist() == [2]
    assert gs[-1] == 3                                                   # the last one rounds to column 64: outside


@pytest.mark.gpu
@pytest.mark.parametrize("cam,w,h", [(TUM1, 640, 480), (KITTI, 1241, 376)])
def test_hip_undistort_grid_matches_oracle(cam, w, h):
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    ex = orbhip.ORBextractor(2000, 1.2, 8, 20, 7)
    kp, desc = ex(synth.frame("blocks", w, h, 51), fetch_pyramid=False)
    assert len(kp) > 1500
    ob = O.image_bounds(cam, w, h)
    gb = orbhip.ComputeImageBounds(cam, w, h)
    assert np.array_equal(ob.view(np.uint32), gb.view(np.uint32))
    oun, ogs, ogi = O.undistort_grid(cam, ob, kp)
    gun, ggs, ggi = orbhip.UndistortAndAssignToGrid(ex, kp, cam, gb)
    assert np.array_equal(oun["x"].view(np.uint32), gun["x"].view(np.uint32))
    assert np.array_equal(oun["y"].view(np.uint32), gun["y"].view(np.uint32))
    for fld in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(oun[fld], gun[fld])
    assert np.array_equal(ogs, ggs) and np.array_equal(ogi, ggi)
    if cam[4] != 0:
        assert np.abs(gun["x"] - kp["x"]).max() > 1.0                # the distortion really moves points


@pytest.mark.gpu
def test_hip_undistort_grid_empty():
    from visual_slam_amd import orbhip
    ex = orbhip.ORBextractor(500, 1.2, 8, 20, 7)
    ex(synth.frame("blocks", 320, 240, 52), fetch_pyramid=False)
    un, gs, gi = orbhip.UndistortAndAssignToGrid(ex, np.zeros(0, orbhip.KP_DTYPE), KITTI, orbhip.ComputeImageBounds(KITTI, 320, 240))
    assert len(un) == 0 and len(gi) == 0 and not gs.any()

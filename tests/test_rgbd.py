"""Frame::ComputeStereoFromRGBD (reference src/Frame.cc:655-672) behind the depth conversion of
Tracking::GrabImageRGBD: oracle known answers and GPU parity (mvuRight, mvDepth bit for bit)."""
import numpy as np
import pytest

from visual_slam_amd import synth

TUM1 = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314)
MBF = 40.0


def make(seed=3):
    from oracle import orb_oracle as O
    w, h = 640, 480
    e = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    kp, _ = e.extract(synth.frame("blocks", w, h, 55))
    bounds = O.image_bounds(TUM1, w, h)
    un = O.undistort_grid(TUM1, bounds, kp)[0]
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 40000, (h, w)).astype(np.uint16)
    raw[rng.random((h, w)) < 0.2] = 0                                       # holes of the sensor
    return kp, un, raw


def test_oracle_rgbd_known_answers():
    from oracle import orb_oracle as O
    kp, un, raw = make()
    factor = np.float32(1.0 / 5000.0)                                       # TUM: DepthMapFactor 5000
    ur, dp = O.stereo_from_rgbd(kp, un, raw, factor, MBF)
    d = raw[kp["y"].astype(int), kp["x"].astype(int)].astype(np.float32) * factor
    assert np.array_equal(dp, np.where(d > 0, d, np.float32(-1)).astype(np.float32))
    good = d > 0
    assert good.sum() > 300 and (~good).sum() > 50
    assert np.array_equal(ur[good], (un["x"][good] - np.float32(MBF) / d[good]).astype(np.float32)) and np.all(ur[~good] == -1)
    # a float map with factor 1 is taken as it is; with another factor it is scaled
    fl = (raw.astype(np.float32) * factor).astype(np.float32)
    ur1, dp1 = O.stereo_from_rgbd(kp, un, fl, 1.0, MBF)
    assert np.array_equal(dp1, dp) and np.array_equal(ur1, ur)
    ur2, dp2 = O.stereo_from_rgbd(kp, un, fl, 2.0, MBF)
    assert np.array_equal(dp2[good], dp[good] * np.float32(2))


@pytest.mark.gpu
@pytest.mark.parametrize("as_float,factor", [(False, 1.0 / 5000.0), (True, 1.0), (True, 0.5), (False, 1.0)])
def test_hip_rgbd_matches_oracle(as_float, factor):
    import torch
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    kp, un, raw = make()
    depth = (raw.astype(np.float32) / np.float32(5000.0)).astype(np.float32) if as_float else raw
    n = 3
    cap = len(kp) + 7
    ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7)
    dev = torch.device("cuda", 0)
    kps = np.zeros((n, cap), orbhip.KP_DTYPE); uns = np.zeros((n, cap), orbhip.KP_DTYPE)
    counts = np.array([len(kp), len(kp) // 2, 0], np.int32)
    for i in range(n):
        kps[i, :len(kp)] = kp; uns[i, :len(un)] = un
    maps = np.stack([depth, depth[::-1].copy(), depth])
    d_kp = torch.from_numpy(kps.view(np.uint8).reshape(n, cap, 28)).to(dev); d_un = torch.from_numpy(uns.view(np.uint8).reshape(n, cap, 28)).to(dev)
    d_cnt = torch.from_numpy(counts).to(dev)
    d_map = torch.from_numpy(maps.view(np.uint8).reshape(n, -1)).to(dev)
    d_ur = torch.zeros((n, cap), dtype=torch.float32, device=dev); d_dp = torch.zeros((n, cap), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    orbhip.stereo_from_rgbd_batch_device(ex, n, d_kp.data_ptr(), d_un.data_ptr(), d_cnt.data_ptr(), cap, d_map.data_ptr(), not as_float,
                                         640, 480, maps.strides[1], maps.strides[0], factor, MBF, d_ur.data_ptr(), d_dp.data_ptr())
    ex.synchronize()
    hur, hdp = d_ur.cpu().numpy(), d_dp.cpu().numpy()
    for i in range(n):
        c = counts[i]
        our, odp = O.stereo_from_rgbd(kp[:c], un[:c], maps[i], factor, MBF)
        assert np.array_equal(hur[i, :c].view(np.uint32), our.view(np.uint32)), i
        assert np.array_equal(hdp[i, :c].view(np.uint32), odp.view(np.uint32)), i
        assert np.all(hur[i, c:] == -1) and np.all(hdp[i, c:] == -1)
    assert (hdp[0] > 0).sum() > 300


def test_oracle_unproject_known_answers():
    from oracle import orb_oracle as O
    kp, un, raw = make()
    ur, dp = O.stereo_from_rgbd(kp, un, raw, 1.0 / 5000.0, MBF)
    ident = np.concatenate([np.eye(3).ravel(), [0, 0, 0]]).astype(np.float32)
    world, valid = O.unproject_stereo(un, dp, ident, TUM1[:4])
    assert np.array_equal(valid, (dp > 0).astype(np.uint8)) and np.all(world[valid == 0] == 0)
    g = valid == 1
    # re-projection gives the undistorted keypoint back, the depth is the z coordinate
    assert np.array_equal(world[g, 2], dp[g])
    assert np.allclose(world[g, 0] / world[g, 2] * TUM1[0] + TUM1[2], un["x"][g], atol=1e-2)
    assert np.allclose(world[g, 1] / world[g, 2] * TUM1[1] + TUM1[3], un["y"][g], atol=1e-2)
    # a pose moves the points rigidly
    c, s_ = np.cos(0.3), np.sin(0.3)
    pose = np.array([c, 0, s_, 0, 1, 0, -s_, 0, c, 1.0, -2.0, 0.5], np.float32)
    w2, _ = O.unproject_stereo(un, dp, pose, TUM1[:4])
    R = pose[:9].reshape(3, 3).astype(np.float64)
    assert np.allclose(w2[g], world[g].astype(np.float64) @ R.T + pose[9:12], atol=1e-4)


@pytest.mark.gpu
def test_hip_unproject_matches_oracle():
    import torch
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    kp, un, raw = make()
    ur, dp = O.stereo_from_rgbd(kp, un, raw, 1.0 / 5000.0, MBF)
    n, cap = 2, len(kp) + 5
    c, s_ = np.cos(-0.2), np.sin(-0.2)
    poses = np.stack([np.concatenate([np.eye(3).ravel(), [0, 0, 0]]), [c, -s_, 0, s_, c, 0, 0, 0, 1, 0.3, 0.1, -2.0]]).astype(np.float32)
    uns = np.zeros((n, cap), orbhip.KP_DTYPE); dps = np.full((n, cap), -1.0, np.float32)
    counts = np.array([len(kp), len(kp) - 100], np.int32)
    for i in range(n):
        uns[i, :len(un)] = un; dps[i, :len(dp)] = dp
    ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7)
    dev = torch.device("cuda", 0)
    d_un = torch.from_numpy(uns.view(np.uint8).reshape(n, cap, 28)).to(dev); d_dp = torch.from_numpy(dps).to(dev)
    d_cnt = torch.from_numpy(counts).to(dev); d_pose = torch.from_numpy(poses).to(dev)
    d_w = torch.zeros((n, cap, 3), dtype=torch.float32, device=dev); d_v = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    orbhip.unproject_stereo_batch_device(ex, n, d_un.data_ptr(), d_dp.data_ptr(), d_cnt.data_ptr(), cap, d_pose.data_ptr(), TUM1[:4],
                                         d_w.data_ptr(), d_v.data_ptr())
    ex.synchronize()
    hw, hv = d_w.cpu().numpy(), d_v.cpu().numpy()
    for i in range(n):
        cnt = counts[i]
        ow, ov = O.unproject_stereo(un[:cnt], dp[:cnt], poses[i], TUM1[:4])
        assert np.array_equal(hv[i, :cnt], ov) and np.all(hv[i, cnt:] == 0)
        assert np.array_equal(hw[i, :cnt].view(np.uint32), ow.view(np.uint32)), i
    assert hv[0].sum() > 300

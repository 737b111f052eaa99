"""GPU parity of the stereo matching step (reference src/Frame.cc:481-653) against the CPU oracle.

Synthetic rectified pairs: the right image is the left image with its content moved left by a
per-band disparity, so true matches exist at known disparities.
"""
import numpy as np
import pytest

from visual_slam_amd import synth

pytestmark = pytest.mark.gpu

PARAMS = (2000, 1.2, 8, 20, 7)
FX, BF = 718.856, 386.1448          # KITTI 00-02 calibration (Examples/Stereo/KITTI00-02.yaml)


def make_pair(kind, w, h, index, disparities=(3, 12, 40)):
    return synth.stereo_pair(kind, w, h, index, disparities)


def run_pair(kind, w, h, index, params=PARAMS):
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    left, right = make_pair(kind, w, h, index)
    oL, oR = O.OracleExtractor(*params), O.OracleExtractor(*params)
    okl, odl = oL.extract(left)
    okr, odr = oR.extract(right)
    ou, od, on = O.stereo_match(oL, oR, okl, odl, okr, odr, BF / FX, BF)
    gL, gR = orbhip.ORBextractor(*params), orbhip.ORBextractor(*params)
    gkl, gdl = gL(left, fetch_pyramid=False)
    gkr, gdr = gR(right, fetch_pyramid=False)
    gu, gd, gn = orbhip.ComputeStereoMatches(gL, gR, gkl, gdl, gkr, gdr, BF / FX, BF)
    return (okl, ou, od, on), (gkl, gu, gd, gn)


@pytest.mark.parametrize("kind,w,h", [("blocks", 1241, 376), ("blocks", 752, 480), ("lowc", 640, 480), ("noise", 320, 240)])
def test_stereo_matches_bit_exact(kind, w, h):
    (okl, ou, od, on), (gkl, gu, gd, gn) = run_pair(kind, w, h, 0)
    assert len(okl) == len(gkl)
    assert np.array_equal(okl["x"], gkl["x"]) and np.array_equal(okl["y"], gkl["y"])
    assert gn == on
    # float results come from the same float32 operations in the same order: bit-exact
    assert np.array_equal(ou.view(np.uint32), gu.view(np.uint32))
    assert np.array_equal(od.view(np.uint32), gd.view(np.uint32))
    if kind == "blocks":
        assert on > 100                                  # the synthetic pair really has matches
        m = gu >= 0
        disp = gkl["x"][m] - gu[m]
        assert np.all(disp >= 0) and np.all(disp < BF / (BF / FX))


def test_stereo_no_right_keypoints():
    from visual_slam_amd import orbhip
    left = synth.batch("blocks", 640, 480, 1)[0]
    right = np.full_like(left, 100)
    gL, gR = orbhip.ORBextractor(1000, 1.2, 8, 20, 7), orbhip.ORBextractor(1000, 1.2, 8, 20, 7)
    kl, dl = gL(left, fetch_pyramid=False)
    kr, dr = gR(right, fetch_pyramid=False)
    assert len(kr) == 0
    u, d, n = orbhip.ComputeStereoMatches(gL, gR, kl, dl, kr, dr, BF / FX, BF)
    assert n == 0 and np.all(u == -1) and np.all(d == -1)


def test_stereo_rejects_mismatched_instances():
    from visual_slam_amd import orbhip
    a = synth.batch("blocks", 640, 480, 1)[0]
    b = synth.batch("blocks", 320, 240, 1)[0]
    gL, gR = orbhip.ORBextractor(500, 1.2, 8, 20, 7), orbhip.ORBextractor(500, 1.2, 8, 20, 7)
    kl, dl = gL(a, fetch_pyramid=False)
    kr, dr = gR(b, fetch_pyramid=False)
    with pytest.raises(orbhip.OrbHipError):
        orbhip.ComputeStereoMatches(gL, gR, kl, dl, kr, dr, BF / FX, BF)


def test_stereo_batch_device_matches_oracle():
    """Device-resident batch entry point: three pairs in one pass, outputs compared per pair."""
    import torch
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    w, h, prm = 640, 480, (1000, 1.2, 8, 20, 7)
    pairs = [synth.stereo_pair("blocks", w, h, i, (4, 17)) for i in range(3)]
    dev = torch.device("cuda", 0)
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    cap = gL.max_keypoints()
    mk = lambda: (torch.empty((3, cap, 7), dtype=torch.float32, device=dev), torch.empty((3, cap, 32), dtype=torch.uint8, device=dev),
                  torch.zeros((3,), dtype=torch.int32, device=dev))
    (kl, dl, cl), (kr, dr, cr) = mk(), mk()
    ur = torch.empty((3, cap), dtype=torch.float32, device=dev)
    dp = torch.empty((3, cap), dtype=torch.float32, device=dev)
    nm = torch.zeros((3,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    gL.extract_batch_device(L.data_ptr(), 3, w, h, w, w * h, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), cap)
    gR.extract_batch_device(R.data_ptr(), 3, w, h, w, w * h, kr.data_ptr(), dr.data_ptr(), cr.data_ptr(), cap)
    orbhip.stereo_match_batch_device(gL, gR, 3, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), kr.data_ptr(), dr.data_ptr(),
                                     cr.data_ptr(), cap, BF / FX, BF, ur.data_ptr(), dp.data_ptr(), nm.data_ptr())
    gL.synchronize(); gR.synchronize()
    torch.cuda.synchronize()
    for i, (left, right) in enumerate(pairs):
        oL, oR = O.OracleExtractor(*prm), O.OracleExtractor(*prm)
        okl, odl = oL.extract(left)
        okr, odr = oR.extract(right)
        ou, od, on = O.stereo_match(oL, oR, okl, odl, okr, odr, BF / FX, BF)
        n = int(cl[i].item())
        assert n == len(okl) and int(nm[i].item()) == on
        assert np.array_equal(ur[i, :n].cpu().numpy().view(np.uint32), ou.view(np.uint32))
        assert np.array_equal(dp[i, :n].cpu().numpy().view(np.uint32), od.view(np.uint32))


def test_one_pair_of_instances_across_input_sizes():
    """The same two instances extract a small pair, match it and build the feature grid, then extract a
    larger pair (the pinned input staging grows) and do all of it again: the matcher's scratch buffers
    must survive the growth of the input staging."""
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    cam = (718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0, 0.0, 0.0)
    prm = (1000, 1.2, 8, 20, 7)
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    for kind, w, h, idx in (("blocks", 320, 240, 3), ("blocks", 752, 480, 4), ("noise", 320, 240, 5)):
        left, right = make_pair(kind, w, h, idx)
        oL, oR = O.OracleExtractor(*prm), O.OracleExtractor(*prm)
        okl, odl = oL.extract(left)
        okr, odr = oR.extract(right)
        ou, od, on = O.stereo_match(oL, oR, okl, odl, okr, odr, BF / FX, BF)
        gkl, gdl = gL(left, fetch_pyramid=False)
        gkr, gdr = gR(right, fetch_pyramid=False)
        gu, gd, gn = orbhip.ComputeStereoMatches(gL, gR, gkl, gdl, gkr, gdr, BF / FX, BF)
        assert gn == on
        assert np.array_equal(ou.view(np.uint32), gu.view(np.uint32))
        assert np.array_equal(od.view(np.uint32), gd.view(np.uint32))
        ob = O.image_bounds(cam, w, h)
        oun, ogs, ogi = O.undistort_grid(cam, ob, okl)
        gun, ggs, ggi = orbhip.UndistortAndAssignToGrid(gL, gkl, cam, orbhip.ComputeImageBounds(cam, w, h))
        assert np.array_equal(oun["x"].view(np.uint32), gun["x"].view(np.uint32))
        assert np.array_equal(ogs, ggs) and np.array_equal(ogi, ggi)


def test_extract_and_match_in_one_call_across_passes():
    """orbhip_extract_stereo_batch_device on 5 pairs with passes of 2: extraction and matching run pass by pass,
    every pair's outputs equal the oracle's; the plain matcher call on more pairs than a pass holds is refused."""
    import torch
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    w, h, prm, n = 640, 480, (1000, 1.2, 8, 20, 7), 5
    pairs = [synth.stereo_pair("blocks", w, h, 20 + i, (4, 17)) for i in range(n)]
    dev = torch.device("cuda", 0)
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    gL, gR = orbhip.ORBextractor(*prm, max_batch=2), orbhip.ORBextractor(*prm, max_batch=2)
    cap = gL.max_keypoints()
    mk = lambda: (torch.empty((n, cap, 7), dtype=torch.float32, device=dev), torch.empty((n, cap, 32), dtype=torch.uint8, device=dev),
                  torch.zeros((n,), dtype=torch.int32, device=dev))
    (kl, dl, cl), (kr, dr, cr) = mk(), mk()
    ur = torch.empty((n, cap), dtype=torch.float32, device=dev)
    dp = torch.empty((n, cap), dtype=torch.float32, device=dev)
    nm = torch.zeros((n,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    orbhip.extract_stereo_batch_device(gL, gR, L.data_ptr(), R.data_ptr(), n, w, h, w, w * h, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(),
                                       kr.data_ptr(), dr.data_ptr(), cr.data_ptr(), cap, BF / FX, BF, ur.data_ptr(), dp.data_ptr(), nm.data_ptr())
    gL.synchronize(); gR.synchronize()
    torch.cuda.synchronize()
    for i, (left, right) in enumerate(pairs):
        oL, oR = O.OracleExtractor(*prm), O.OracleExtractor(*prm)
        okl, odl = oL.extract(left)
        okr, odr = oR.extract(right)
        ou, od, on = O.stereo_match(oL, oR, okl, odl, okr, odr, BF / FX, BF)
        k = int(cl[i].item())
        assert k == len(okl) and int(cr[i].item()) == len(okr) and int(nm[i].item()) == on
        assert np.array_equal(kl[i, :k, 0].cpu().numpy(), okl["x"]) and np.array_equal(dl[i, :k].cpu().numpy(), odl)
        assert np.array_equal(ur[i, :k].cpu().numpy().view(np.uint32), ou.view(np.uint32))
        assert np.array_equal(dp[i, :k].cpu().numpy().view(np.uint32), od.view(np.uint32))
    with pytest.raises(orbhip.OrbHipError) as ei:             # the last pass held one pair only
        orbhip.stereo_match_batch_device(gL, gR, n, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), kr.data_ptr(), dr.data_ptr(),
                                         cr.data_ptr(), cap, BF / FX, BF, ur.data_ptr(), dp.data_ptr(), nm.data_ptr())
    assert ei.value.code == orbhip.ERR_STATE

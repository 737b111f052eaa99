"""The stereo matching entry point that allocates nothing (orbhip_stereo_match_batch_device_noalloc):
its results must equal those of orbhip_stereo_match_batch_device bit for bit, on two calls in a row
of which the second is smaller (the static scratch keeps the first call's contents)."""
import numpy as np
import pytest

from visual_slam_amd import synth

pytestmark = pytest.mark.gpu

FX, BF = 718.856, 386.1448          # KITTI 00-02 calibration (Examples/Stereo/KITTI00-02.yaml)


def _run(gL, gR, pairs, w, h, noalloc):
    import torch
    from visual_slam_amd import orbhip
    n = len(pairs)
    dev = torch.device("cuda", 0)
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    cap = gL.max_keypoints()
    mk = lambda: (torch.empty((n, cap, 7), dtype=torch.float32, device=dev), torch.empty((n, cap, 32), dtype=torch.uint8, device=dev),
                  torch.zeros((n,), dtype=torch.int32, device=dev))
    (kl, dl, cl), (kr, dr, cr) = mk(), mk()
    ur = torch.full((n, cap), -7.0, dtype=torch.float32, device=dev)
    dp = torch.full((n, cap), -7.0, dtype=torch.float32, device=dev)
    nm = torch.zeros((n,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    gL.extract_batch_device(L.data_ptr(), n, w, h, w, w * h, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), cap)
    gR.extract_batch_device(R.data_ptr(), n, w, h, w, w * h, kr.data_ptr(), dr.data_ptr(), cr.data_ptr(), cap)
    orbhip.stereo_match_batch_device(gL, gR, n, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), kr.data_ptr(), dr.data_ptr(),
                                     cr.data_ptr(), cap, BF / FX, BF, ur.data_ptr(), dp.data_ptr(), nm.data_ptr(), noalloc=noalloc)
    gL.synchronize(); gR.synchronize()
    torch.cuda.synchronize()
    cnt = cl.cpu().numpy()
    return cnt, [ur[i, :cnt[i]].cpu().numpy() for i in range(n)], [dp[i, :cnt[i]].cpu().numpy() for i in range(n)], nm.cpu().numpy()


def test_noalloc_equals_allocating_entry_point_twice():
    from visual_slam_amd import orbhip
    prm = (1000, 1.2, 8, 20, 7)
    # first call: 5 pairs of 640x480; second call: 2 pairs of 320x240 (fewer pairs, fewer keypoints, fewer row bins)
    for npairs, w, h, first in ((5, 640, 480, 0), (2, 320, 240, 40)):
        pairs = [synth.stereo_pair("blocks", w, h, first + i, (4, 17)) for i in range(npairs)]
        gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
        ref = _run(gL, gR, pairs, w, h, noalloc=False)
        got = _run(gL, gR, pairs, w, h, noalloc=True)
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[3], got[3])
        assert ref[3].sum() > 50 * npairs                       # the pairs really match
        for i in range(npairs):
            assert np.array_equal(ref[1][i].view(np.uint32), got[1][i].view(np.uint32))
            assert np.array_equal(ref[2][i].view(np.uint32), got[2][i].view(np.uint32))


def test_noalloc_refuses_batches_beyond_its_scratch():
    import torch
    from visual_slam_amd import orbhip
    prm = (300, 1.2, 4, 20, 7)
    w, h = 160, 120
    pairs = [synth.stereo_pair("blocks", w, h, 3 + i, (4,)) for i in range(9)]
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    _run(gL, gR, pairs, w, h, noalloc=True)                     # a pass of 9 pairs exists on both instances
    one = torch.zeros((16,), dtype=torch.int32, device="cuda:0")
    # 9 pairs x 65535 slots exceed ORBHIP_NOALLOC_MAX_SLOTS: refused before anything is launched or read
    with pytest.raises(orbhip.OrbHipError) as ei:
        orbhip.stereo_match_batch_device(gL, gR, 9, one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(),
                                         one.data_ptr(), 65535, BF / FX, BF, one.data_ptr(), one.data_ptr(), None, noalloc=True)
    assert ei.value.code == orbhip.ERR_CAPACITY

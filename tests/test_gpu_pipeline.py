"""The device entry points chained as a stereo tracking front-end would chain them, on one stream
and without a host round trip in between: rectification + extraction of both eyes at two time steps,
stereo matching (reference src/Frame.cc:481-653), undistortion (src/Frame.cc:422-449), unprojection
(src/Frame.cc:674-687), the
motion-model matcher (src/ORBmatcher.cc:1324-1445) between the two steps and the local-map matcher
(src/ORBmatcher.cc:46-127) on the second. Every stage is compared
with the oracle run stage by stage on the host: besides the arithmetic this checks the ordering of
the calls against each other (streams, workspaces that later passes overwrite)."""
import numpy as np
import pytest

from visual_slam_amd import synth

KITTI = (718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0, 0.0, 0.0)
Z = 0.54 * 718.856 / 14.0          # depth that a 14-px disparity means for this rig


def shifted(img, dx, dy):
    out = img.copy()
    h, w = img.shape
    out[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)] = img[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)]
    return out


@pytest.mark.gpu
def test_stereo_front_end_chain_matches_oracle_stage_by_stage():
    import torch
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    n, w, h, prm = 3, 640, 360, (800, 1.2, 8, 20, 7)
    fx, fy, cx, cy = KITTI[:4]
    bf = 0.54 * fx
    mb = bf / fx
    motion = [(6, -3), (-9, 2), (3, 5)]
    mx, my = synth.rectify_maps(w, h, 0.8, 0.04, (1.3, -0.7))
    # raw images: time 0 and time 1 (image motion) of the left eye; the right eye sees the content moved by the disparity
    L0 = np.stack([synth.frame("blocks", w, h, 300 + i) for i in range(n)])
    L1 = np.stack([shifted(L0[i], *motion[i]) for i in range(n)])
    R0 = np.stack([shifted(L0[i], -14, 0) for i in range(n)])
    R1 = np.stack([shifted(L1[i], -14, 0) for i in range(n)])
    bounds = orbhip.ComputeImageBounds(KITTI, w, h)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    sp = stream.cuda_stream
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    gL.set_rectification(mx, my); gR.set_rectification(mx, my)
    cap = gL.max_keypoints()
    lib = orbhip.lib()
    cam_c = orbhip._cam(np.array(KITTI, np.float32)); b_c = np.ascontiguousarray(bounds, np.float32)

    def outputs():
        return (torch.zeros((n, cap, 7), dtype=torch.float32, device=dev), torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev),
                torch.zeros((n,), dtype=torch.int32, device=dev))

    with torch.cuda.stream(stream):
        dimg = {k: torch.from_numpy(v).to(dev) for k, v in (("L0", L0), ("R0", R0), ("L1", L1), ("R1", R1))}
        res = {k: outputs() for k in dimg}
        un = {k: torch.zeros((n, cap, 7), dtype=torch.float32, device=dev) for k in ("L0", "L1")}
        ur = {k: torch.zeros((n, cap), dtype=torch.float32, device=dev) for k in ("0", "1")}
        dp = {k: torch.zeros((n, cap), dtype=torch.float32, device=dev) for k in ("0", "1")}
        nm = {k: torch.zeros((n,), dtype=torch.int32, device=dev) for k in ("0", "1")}
        gs = torch.zeros((n, orbhip.GRID_CELLS + 1), dtype=torch.int32, device=dev)
        gi = torch.zeros((n, cap), dtype=torch.int32, device=dev)
        match = torch.zeros((n, cap), dtype=torch.int32, device=dev)
        nmatch = torch.zeros((n,), dtype=torch.int32, device=dev)
    stream.synchronize()
    # time step t: both eyes, stereo matching, undistortion of the left keypoints -- all enqueued, no waits
    for t in ("0", "1"):
        for eye, g in (("L", gL), ("R", gR)):
            k, d, c = res[eye + t]
            g.extract_batch_device(dimg[eye + t].data_ptr(), n, w, h, w, w * h, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap, stream=sp)
        (kl, dl, cl), (kr, dr, cr) = res["L" + t], res["R" + t]
        orbhip.stereo_match_batch_device(gL, gR, n, kl.data_ptr(), dl.data_ptr(), cl.data_ptr(), kr.data_ptr(), dr.data_ptr(),
                                         cr.data_ptr(), cap, mb, bf, ur[t].data_ptr(), dp[t].data_ptr(), nm[t].data_ptr(), stream=sp)
        gL._check(lib.orbhip_undistort_grid_batch_device(gL._h, n, kl.data_ptr(), cl.data_ptr(), cap, orbhip._ptr(cam_c), orbhip._ptr(b_c),
                                                         un["L" + t].data_ptr(), gs.data_ptr(), gi.data_ptr(), sp))
    stream.synchronize()
    # map points of step 0: left keypoints with a stereo depth, back-projected on the device (Frame::UnprojectStereo,
    # the last frame's camera = the world frame); usable and observed where a depth exists
    ident = np.tile(np.concatenate([np.eye(3).ravel(), [0, 0, 0]]).astype(np.float32), (n, 1))
    Tcw = np.stack([np.concatenate([np.eye(3).ravel(), [dx * Z / fx, dy * Z / fy, 0.0]]) for dx, dy in motion]).astype(np.float32)
    with torch.cuda.stream(stream):
        d_ident, dT = torch.from_numpy(ident).to(dev), torch.from_numpy(Tcw).to(dev)
        dw = torch.zeros((n, cap, 3), dtype=torch.float32, device=dev)
        dvalid = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
    stream.synchronize()
    orbhip.unproject_stereo_batch_device(gL, n, un["L0"].data_ptr(), dp["0"].data_ptr(), res["L0"][2].data_ptr(), cap, d_ident.data_ptr(),
                                         KITTI[:4], dw.data_ptr(), dvalid.data_ptr(), stream=sp)
    with torch.cuda.stream(stream):
        df = dvalid * 3
    stream.synchronize()
    hu0, hc0 = un["L0"].cpu().numpy(), res["L0"][2].cpu().numpy()
    world, flags = dw.cpu().numpy(), df.cpu().numpy()
    prm_p = orbhip.projection_params(KITTI[:4], bf, bounds, 7.0, True)
    orbhip.search_by_projection_batch_device(gL, n, cap, un["L0"].data_ptr(), dw.data_ptr(), df.data_ptr(), res["L0"][1].data_ptr(),
                                             res["L0"][2].data_ptr(), un["L1"].data_ptr(), res["L1"][1].data_ptr(), ur["1"].data_ptr(),
                                             res["L1"][2].data_ptr(), dT.data_ptr(), None, prm_p, match.data_ptr(), nmatch.data_ptr(),
                                             stream=sp)
    stream.synchronize()
    # Tracking::SearchLocalPoints next: the local map = the points of step 0 twice over (shuffled), projected with
    # the same motion; keypoints that the motion-model matcher gave an observed point are occupied
    rng = np.random.default_rng(5)
    hmatch = match.cpu().numpy()
    occupied = ((hmatch >= 0) & (np.take_along_axis(flags, np.maximum(hmatch, 0), axis=1) >= 2)).astype(np.uint8)
    cap_mp = 2 * cap
    mp = np.zeros((n, cap_mp), orbhip.TRACK_DTYPE)
    dmp = np.zeros((n, cap_mp, 32), np.uint8)
    cnt_mp = np.zeros(n, np.int32)
    hd0 = res["L0"][1].cpu().numpy()
    for i in range(n):
        src = np.tile(np.arange(int(hc0[i])), 2); rng.shuffle(src)
        m = len(src)
        mp["proj_x"][i, :m] = hu0[i, src, 0] + motion[i][0] + rng.normal(0, 1.0, m)
        mp["proj_y"][i, :m] = hu0[i, src, 1] + motion[i][1] + rng.normal(0, 1.0, m)
        mp["proj_xr"][i, :m] = mp["proj_x"][i, :m] - 14.0
        mp["view_cos"][i, :m] = np.where(rng.random(m) < 0.5, 0.9995, 0.9)
        mp["level"][i, :m] = hu0[i, src].view(orbhip.KP_DTYPE)["octave"].reshape(-1)
        mp["flags"][i, :m] = 1 | ((rng.random(m) < 0.7).astype(np.int32) << 1)
        dmp[i, :m] = hd0[i, src]
        cnt_mp[i] = m
    with torch.cuda.stream(stream):
        d_mp = torch.from_numpy(mp.view(np.uint8).reshape(n, cap_mp, 24)).to(dev)
        d_dmp, d_cmp, d_occ = (torch.from_numpy(a).to(dev) for a in (dmp, cnt_mp, occupied))
        lmatch = torch.zeros((n, cap), dtype=torch.int32, device=dev)
        lnm = torch.zeros((n,), dtype=torch.int32, device=dev)
    stream.synchronize()
    orbhip.search_map_points_batch_device(gL, n, cap_mp, d_mp.data_ptr(), d_dmp.data_ptr(), d_cmp.data_ptr(), cap, un["L1"].data_ptr(),
                                          res["L1"][1].data_ptr(), ur["1"].data_ptr(), d_occ.data_ptr(), res["L1"][2].data_ptr(),
                                          bounds, 3.0, 0.8, lmatch.data_ptr(), lnm.data_ptr(), stream=sp)
    stream.synchronize()

    # ---- the same chain on the host, stage by stage
    total_stereo = total_proj = total_local = 0
    for i in range(n):
        o = {}
        for key, raw in (("L0", L0), ("R0", R0), ("L1", L1), ("R1", R1)):
            ex = O.OracleExtractor(*prm)
            kp, desc = ex.extract(O.remap_linear(raw[i], mx, my))
            o[key] = (ex, kp, desc)
            c = int(res[key][2][i].item())
            assert c == len(kp), (i, key)
            assert np.array_equal(res[key][1][i, :c].cpu().numpy(), desc), (i, key)
        st = {}
        for t in ("0", "1"):
            (eL, kl, dl), (eR, kr, dr) = o["L" + t], o["R" + t]
            ou, od, on = O.stereo_match(eL, eR, kl, dl, kr, dr, mb, bf)
            c = len(kl)
            assert int(nm[t][i].item()) == on
            assert np.array_equal(ur[t][i, :c].cpu().numpy().view(np.uint32), ou.view(np.uint32)), (i, t)
            assert np.array_equal(dp[t][i, :c].cpu().numpy().view(np.uint32), od.view(np.uint32)), (i, t)
            st[t] = (ou, od)
            total_stereo += on
        u0 = O.undistort_grid(KITTI, bounds, o["L0"][1])[0]
        u1 = O.undistort_grid(KITTI, bounds, o["L1"][1])[0]
        c0, c1 = len(u0), len(u1)
        ow, ov = O.unproject_stereo(u0, st["0"][1], ident[i], KITTI[:4])
        assert np.array_equal(world[i, :c0].view(np.uint32), ow.view(np.uint32)) and np.array_equal(flags[i, :c0], ov * 3), i
        om, on = O.search_by_projection(u0, world[i, :c0], flags[i, :c0], o["L0"][2], u1, o["L1"][2], st["1"][0], Tcw[i],
                                        KITTI[:4], bf, bounds, o["L0"][0].table(0), 7.0, 0, True)
        assert int(nmatch[i].item()) == on, i
        assert np.array_equal(match[i, :c1].cpu().numpy(), om), i
        total_proj += on
        lm, ln = O.search_map_points(mp[i, :cnt_mp[i]], dmp[i, :cnt_mp[i]], u1, o["L1"][2], st["1"][0], occupied[i, :c1], bounds,
                                     o["L1"][0].table(0), 3.0, 0.8)
        assert int(lnm[i].item()) == ln, i
        assert np.array_equal(lmatch[i, :c1].cpu().numpy(), lm), i
        total_local += ln
    assert total_stereo > 6 * 100 and total_proj > 3 * 30 and total_local > 3 * 100

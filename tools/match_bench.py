#!/usr/bin/env python3
"""Throughput of the device matchers on a batch of frame pairs that stays in device memory:
extraction of both frames, undistortion, then ORBmatcher::SearchByProjection(CurrentFrame, LastFrame)
for every pair in one launch. Prints one JSON line; the CPU figure is the oracle's restatement of the
same function on the same pairs (one thread).

Usage: python tools/match_bench.py [--pairs 256] [--width 1241 --height 376] [--steps 5]
"""
import argparse
import ctypes as C
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from visual_slam_amd import orbhip, synth  # noqa: E402

KITTI = (718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0, 0.0, 0.0)
Z = 6.0


def shifted(img, dx, dy):
    out = img.copy()
    h, w = img.shape
    out[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)] = img[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)]
    return out


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--th", type=float, default=15.0)
    ap.add_argument("--cpu-pairs", type=int, default=16)
    ap.add_argument("--observed", type=float, default=0.8, help="fraction of map points with observations")
    ap.add_argument("--local-copies", type=int, default=2, help="local map = this many copies of the last frame's keypoints")
    args = ap.parse_args()
    n, w, h = args.pairs, args.width, args.height
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    last = np.stack([synth.frame("blocks", w, h, 900 + i) for i in range(n)])
    motion = [((i * 7) % 23 - 11, (i * 5) % 13 - 6) for i in range(n)]
    cur = np.stack([shifted(last[i], *motion[i]) for i in range(n)])
    ex = orbhip.ORBextractor(args.nfeatures, 1.2, 8, 20, 7)
    cap = ex.max_keypoints()
    fx, fy, cx, cy = KITTI[:4]
    bounds = orbhip.ComputeImageBounds(KITTI, w, h)
    cam = np.array(KITTI, np.float32)
    with torch.cuda.stream(stream):
        dl = torch.from_numpy(last).to(dev); dc = torch.from_numpy(cur).to(dev)
        mk = lambda: (torch.zeros((n, cap, 7), dtype=torch.float32, device=dev), torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev),
                      torch.zeros((n,), dtype=torch.int32, device=dev))
        (k1, d1, c1), (k2, d2, c2) = mk(), mk()
        u1 = torch.zeros_like(k1); u2 = torch.zeros_like(k2)
        gs = torch.zeros((n, orbhip.GRID_CELLS + 1), dtype=torch.int32, device=dev)
        gi = torch.zeros((n, cap), dtype=torch.int32, device=dev)
        match = torch.zeros((n, cap), dtype=torch.int32, device=dev)
        nm = torch.zeros((n,), dtype=torch.int32, device=dev)
    stream.synchronize()
    sp = stream.cuda_stream
    L = orbhip.lib()
    cam_c = orbhip._cam(cam); b_c = np.ascontiguousarray(bounds, np.float32)
    ex.extract_batch_device(dl.data_ptr(), n, w, h, w, w * h, k1.data_ptr(), d1.data_ptr(), c1.data_ptr(), cap, stream=sp)
    ex._check(L.orbhip_undistort_grid_batch_device(ex._h, n, k1.data_ptr(), c1.data_ptr(), cap, orbhip._ptr(cam_c), orbhip._ptr(b_c),
                                                   u1.data_ptr(), gs.data_ptr(), gi.data_ptr(), sp))
    ex.extract_batch_device(dc.data_ptr(), n, w, h, w, w * h, k2.data_ptr(), d2.data_ptr(), c2.data_ptr(), cap, stream=sp)
    ex._check(L.orbhip_undistort_grid_batch_device(ex._h, n, k2.data_ptr(), c2.data_ptr(), cap, orbhip._ptr(cam_c), orbhip._ptr(b_c),
                                                   u2.data_ptr(), gs.data_ptr(), gi.data_ptr(), sp))
    stream.synchronize()
    # map points: the last frame's keypoints on the plane z = Z; the current pose explains the image motion
    hu1 = u1.cpu().numpy()
    world = np.stack([(hu1[..., 0] - cx) * Z / fx, (hu1[..., 1] - cy) * Z / fy, np.full(hu1.shape[:2], Z)], axis=2).astype(np.float32)
    rng = np.random.default_rng(11)
    flags = ((rng.random((n, cap)) < 0.85).astype(np.uint8) | ((rng.random((n, cap)) < args.observed).astype(np.uint8) << 1))
    Tcw = np.stack([np.concatenate([np.eye(3).ravel(), [dx * Z / fx, dy * Z / fy, 0.0]]) for dx, dy in motion]).astype(np.float32)
    with torch.cuda.stream(stream):
        dw = torch.from_numpy(world).to(dev); df = torch.from_numpy(flags).to(dev); dT = torch.from_numpy(Tcw).to(dev)
    stream.synchronize()
    prm = orbhip.projection_params(KITTI[:4], 0.5 * fx, bounds, args.th, True)

    def step():
        orbhip.search_by_projection_batch_device(ex, n, cap, u1.data_ptr(), dw.data_ptr(), df.data_ptr(), d1.data_ptr(), c1.data_ptr(),
                                                 u2.data_ptr(), d2.data_ptr(), None, c2.data_ptr(), dT.data_ptr(), None, prm,
                                                 match.data_ptr(), nm.data_ptr(), stream=sp)
    step(); stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(args.steps):
            step()
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    hm, hn = match.cpu().numpy(), nm.cpu().numpy()
    out = {"op": "SearchByProjection(CurrentFrame, LastFrame)", "pairs": n, "image": [w, h], "th": args.th,
           "ms_per_batch": round(ms, 3), "pairs_per_s": round(n / ms * 1e3, 1), "mean_matches": float(hn.mean())}
    # oracle on the first pairs: parity and the CPU time
    try:
        from oracle import orb_oracle as O
        hu2, hd1, hd2 = u2.cpu().numpy(), d1.cpu().numpy(), d2.cpu().numpy()
        hc1, hc2 = c1.cpu().numpy(), c2.cpu().numpy()
        sf = np.array(ex.GetScaleFactors(), np.float32)
        m = min(args.cpu_pairs, n)
        ok, t = True, 0.0
        for i in range(m):
            a1 = np.zeros(hc1[i], orbhip.KP_DTYPE); a1[:] = hu1[i, :hc1[i]].view(orbhip.KP_DTYPE).reshape(-1)
            a2 = np.zeros(hc2[i], orbhip.KP_DTYPE); a2[:] = hu2[i, :hc2[i]].view(orbhip.KP_DTYPE).reshape(-1)
            t0 = time.perf_counter()
            om, on = O.search_by_projection(a1, world[i, :hc1[i]], flags[i, :hc1[i]], hd1[i, :hc1[i]], a2, hd2[i, :hc2[i]], None, Tcw[i],
                                            KITTI[:4], 0.5 * fx, bounds, sf, args.th, 0, True)
            t += time.perf_counter() - t0
            ok = ok and on == hn[i] and np.array_equal(om, hm[i, :hc2[i]])
        out["oracle_parity_first_pairs"] = bool(ok)
        if m > 0: out["cpu_oracle_pairs_per_s_one_thread"] = round(m / t, 1)
    except ImportError:
        pass
    print(json.dumps(out))

    # ---- ORBmatcher::SearchByProjection(F, vpMapPoints, th): a local map of local_copies x the last frame's keypoints
    K = args.local_copies
    hc1 = c1.cpu().numpy(); hk1 = k1.cpu().numpy(); hd1 = d1.cpu().numpy()
    cap_mp = K * cap
    mp = np.zeros((n, cap_mp), orbhip.TRACK_DTYPE)
    dmp = np.zeros((n, cap_mp, 32), np.uint8)
    cnt_mp = np.zeros(n, np.int32)
    for i in range(n):
        c = int(hc1[i])
        src = np.tile(np.arange(c), K); rng.shuffle(src)
        dx, dy = motion[i]
        mp["proj_x"][i, :len(src)] = hu1[i, src, 0] + dx + rng.normal(0, 1.5, len(src))
        mp["proj_y"][i, :len(src)] = hu1[i, src, 1] + dy + rng.normal(0, 1.5, len(src))
        mp["proj_xr"][i, :len(src)] = mp["proj_x"][i, :len(src)] - 0.5 * fx / Z
        mp["view_cos"][i, :len(src)] = np.where(rng.random(len(src)) < 0.5, 0.9995, 0.93)
        mp["level"][i, :len(src)] = hk1[i, src].view(orbhip.KP_DTYPE)["octave"].reshape(-1)
        mp["flags"][i, :len(src)] = 1 | ((rng.random(len(src)) < args.observed).astype(np.int32) << 1)
        dmp[i, :len(src)] = hd1[i, src]
        cnt_mp[i] = len(src)
    with torch.cuda.stream(stream):
        d_mp = torch.from_numpy(mp.view(np.uint8).reshape(n, cap_mp, 24)).to(dev); d_dmp = torch.from_numpy(dmp).to(dev)
        d_cmp = torch.from_numpy(cnt_mp).to(dev)
    stream.synchronize()

    def step_local():
        orbhip.search_map_points_batch_device(ex, n, cap_mp, d_mp.data_ptr(), d_dmp.data_ptr(), d_cmp.data_ptr(), cap, u2.data_ptr(),
                                              d2.data_ptr(), None, None, c2.data_ptr(), bounds, 3.0, 0.8, match.data_ptr(), nm.data_ptr(),
                                              stream=sp)
    step_local(); stream.synchronize()
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(args.steps):
            step_local()
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    hm, hn = match.cpu().numpy(), nm.cpu().numpy()
    out = {"op": "SearchByProjection(F, vpMapPoints)", "frames": n, "image": [w, h], "th": 3.0, "map_points_per_frame": float(cnt_mp.mean()),
           "ms_per_batch": round(ms, 3), "frames_per_s": round(n / ms * 1e3, 1), "mean_matches": float(hn.mean())}
    try:
        from oracle import orb_oracle as O
        hu2, hd2, hc2 = u2.cpu().numpy(), d2.cpu().numpy(), c2.cpu().numpy()
        sf = np.array(ex.GetScaleFactors(), np.float32)
        m = min(args.cpu_pairs, n)
        ok, t = True, 0.0
        for i in range(m):
            a2 = np.zeros(hc2[i], orbhip.KP_DTYPE); a2[:] = hu2[i, :hc2[i]].view(orbhip.KP_DTYPE).reshape(-1)
            t0 = time.perf_counter()
            om, on = O.search_map_points(mp[i, :cnt_mp[i]], dmp[i, :cnt_mp[i]], a2, hd2[i, :hc2[i]], None, None, bounds, sf, 3.0, 0.8)
            t += time.perf_counter() - t0
            ok = ok and on == hn[i] and np.array_equal(om, hm[i, :hc2[i]])
        out["oracle_parity_first_frames"] = bool(ok)
        if m > 0: out["cpu_oracle_frames_per_s_one_thread"] = round(m / t, 1)
    except ImportError:
        pass
    print(json.dumps(out))

    # ---- ORBmatcher::SearchByBoW(pKF, F): last frame as key frame, synthetic node ids (100 nodes, a hash of descriptor bits)
    def nodes(d):
        v = (d[..., 0].astype(np.uint32) | (d[..., 7].astype(np.uint32) << 8)) & 0xFFF
        return (v * 2654435761 % 100).astype(np.uint32) * 37 + 1000
    hd2 = d2.cpu().numpy()
    nk, nf = nodes(hd1), nodes(hd2)
    fl = (flags & 1).astype(np.uint8)
    with torch.cuda.stream(stream):
        d_nk, d_nf, d_fl = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (nk, nf, fl))
    stream.synchronize()

    def step_bow():
        orbhip.search_by_bow_batch_device(ex, n, cap, k1.data_ptr(), d1.data_ptr(), d_nk.data_ptr(), d_fl.data_ptr(), c1.data_ptr(),
                                          k2.data_ptr(), d2.data_ptr(), d_nf.data_ptr(), c2.data_ptr(), 0.7, True, match.data_ptr(),
                                          nm.data_ptr(), stream=sp)
    step_bow(); stream.synchronize()
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(args.steps):
            step_bow()
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    hm, hn = match.cpu().numpy(), nm.cpu().numpy()
    out = {"op": "SearchByBoW(pKF, F)", "pairs": n, "image": [w, h], "nodes": 100, "ms_per_batch": round(ms, 3),
           "pairs_per_s": round(n / ms * 1e3, 1), "mean_matches": float(hn.mean())}
    try:
        from oracle import orb_oracle as O
        hk1, hk2 = k1.cpu().numpy(), k2.cpu().numpy()
        hc1, hc2 = c1.cpu().numpy(), c2.cpu().numpy()
        m = min(args.cpu_pairs, n)
        ok, t = True, 0.0
        for i in range(m):
            a1 = np.zeros(hc1[i], orbhip.KP_DTYPE); a1[:] = hk1[i, :hc1[i]].view(orbhip.KP_DTYPE).reshape(-1)
            a2 = np.zeros(hc2[i], orbhip.KP_DTYPE); a2[:] = hk2[i, :hc2[i]].view(orbhip.KP_DTYPE).reshape(-1)
            t0 = time.perf_counter()
            om, on = O.search_by_bow(a1, hd1[i, :hc1[i]], nk[i, :hc1[i]], fl[i, :hc1[i]], a2, hd2[i, :hc2[i]], nf[i, :hc2[i]], 0.7, True)
            t += time.perf_counter() - t0
            ok = ok and on == hn[i] and np.array_equal(om, hm[i, :hc2[i]])
        out["oracle_parity_first_pairs"] = bool(ok)
        if m > 0: out["cpu_oracle_pairs_per_s_one_thread"] = round(m / t, 1)
    except ImportError:
        pass
    print(json.dumps(out))

    # ---- the remaining entry points, device time only (their parity is covered by tests/): 256 frames or pairs per launch
    L = orbhip.lib()

    def timed(fn):
        fn(); stream.synchronize()
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(args.steps):
                fn()
            e1.record(stream)
        stream.synchronize()
        return e0.elapsed_time(e1) / args.steps

    # Frame::isInFrustum on the local map of the section above
    mpw = np.zeros((n, cap_mp), orbhip.MAPPOINT_DTYPE)
    for i in range(n):
        m = int(cnt_mp[i])
        zz = np.full(m, Z, np.float32)
        mpw["pos"][i, :m] = np.stack([(mp["proj_x"][i, :m] - cx) * Z / fx, (mp["proj_y"][i, :m] - cy) * Z / fy, zz], axis=1)
        mpw["normal"][i, :m] = (0, 0, 1)
        mpw["min_distance"][i, :m] = 1.0; mpw["max_distance"][i, :m] = 40.0
        mpw["flags"][i, :m] = 3
    pose = np.tile(np.concatenate([np.eye(3).ravel(), [0, 0, 0], [0, 0, 0]]).astype(np.float32), (n, 1))
    with torch.cuda.stream(stream):
        d_mpw = torch.from_numpy(mpw.view(np.uint8).reshape(n, cap_mp, 36)).to(dev); d_pose = torch.from_numpy(pose).to(dev)
        d_track = torch.zeros((n, cap_mp, 24), dtype=torch.uint8, device=dev); d_nvis = torch.zeros((n,), dtype=torch.int32, device=dev)
    stream.synchronize()
    pp = orbhip.projection_params(KITTI[:4], 0.5 * fx, bounds, 0.0, False)
    ms = timed(lambda: orbhip.is_in_frustum_batch_device(ex, n, cap_mp, d_mpw.data_ptr(), d_cmp.data_ptr(), d_pose.data_ptr(), pp, 0.5,
                                                         d_track.data_ptr(), d_nvis.data_ptr(), stream=sp))
    print(json.dumps({"op": "Frame::isInFrustum", "frames": n, "map_points_per_frame": float(cnt_mp.mean()), "ms_per_batch": round(ms, 3),
                      "frames_per_s": round(n / ms * 1e3, 1), "mean_in_view": float(d_nvis.cpu().numpy().mean())}))

    # SearchForInitialization on the (last, current) pairs
    prev = hu1[..., :2].astype(np.float32).copy()
    with torch.cuda.stream(stream):
        d_prev0 = torch.from_numpy(prev).to(dev); d_prev = d_prev0.clone()
        m12 = torch.zeros((n, cap), dtype=torch.int32, device=dev)
    stream.synchronize()
    b_f = np.ascontiguousarray(bounds, np.float32)

    def step_init():
        d_prev.copy_(d_prev0)
        ex._check(L.orbhip_search_for_initialization_batch_device(ex._h, n, u1.data_ptr(), d1.data_ptr(), c1.data_ptr(), u2.data_ptr(),
                                                                  d2.data_ptr(), c2.data_ptr(), cap, orbhip._ptr(b_f), d_prev.data_ptr(), 100,
                                                                  0.9, 1, m12.data_ptr(), nm.data_ptr(), sp))
    with torch.cuda.stream(stream):
        ms = timed(step_init)
    print(json.dumps({"op": "SearchForInitialization(F1, F2)", "pairs": n, "window": 100, "ms_per_batch": round(ms, 3),
                      "pairs_per_s": round(n / ms * 1e3, 1), "mean_matches": float(nm.cpu().numpy().mean())}))

    # the relocalisation and loop-closing variants of the projection matcher on the (last, current) pairs
    maxd = (Z * np.array(ex.GetScaleFactors(), np.float32)[hk1.view(orbhip.KP_DTYPE)["octave"].reshape(n, cap)]).astype(np.float32)
    drange = np.stack([maxd / 3.5832, maxd], axis=2).astype(np.float32)
    nrm = np.zeros((n, cap, 3), np.float32); nrm[..., 2] = 1.0
    with torch.cuda.stream(stream):
        d_rng = torch.from_numpy(drange).to(dev); d_nrm = torch.from_numpy(nrm).to(dev)
    stream.synchronize()
    prm_kf = orbhip.projection_params(KITTI[:4], 0.0, bounds, 10.0, True)
    ms = timed(lambda: ex._check(L.orbhip_search_by_projection_keyframe_batch_device(
        ex._h, n, cap, u1.data_ptr(), dw.data_ptr(), d_rng.data_ptr(), df.data_ptr(), d1.data_ptr(), c1.data_ptr(), u2.data_ptr(),
        d2.data_ptr(), None, c2.data_ptr(), dT.data_ptr(), C.byref(prm_kf), 100, match.data_ptr(), nm.data_ptr(), sp)))
    print(json.dumps({"op": "SearchByProjection(CurrentFrame, pKF, sAlreadyFound)", "pairs": n, "th": 10, "ms_per_batch": round(ms, 3),
                      "pairs_per_s": round(n / ms * 1e3, 1), "mean_matches": float(nm.cpu().numpy().mean())}))
    ms = timed(lambda: ex._check(L.orbhip_search_by_projection_sim3_batch_device(
        ex._h, n, cap, dw.data_ptr(), d_nrm.data_ptr(), d_rng.data_ptr(), df.data_ptr(), d1.data_ptr(), c1.data_ptr(), u2.data_ptr(),
        d2.data_ptr(), None, c2.data_ptr(), dT.data_ptr(), C.byref(prm_kf), match.data_ptr(), nm.data_ptr(), sp)))
    print(json.dumps({"op": "SearchByProjection(pKF, Scw, vpPoints, vpMatched)", "pairs": n, "th": 10, "ms_per_batch": round(ms, 3),
                      "pairs_per_s": round(n / ms * 1e3, 1), "mean_matches": float(nm.cpu().numpy().mean())}))

    # SearchByBoW between two key frames
    with torch.cuda.stream(stream):
        d_f2 = torch.ones((n, cap), dtype=torch.uint8, device=dev)
    stream.synchronize()
    ms = timed(lambda: ex._check(L.orbhip_search_by_bow_keyframes_batch_device(
        ex._h, n, cap, k1.data_ptr(), d1.data_ptr(), d_nk.data_ptr(), d_fl.data_ptr(), c1.data_ptr(), k2.data_ptr(), d2.data_ptr(),
        d_nf.data_ptr(), d_f2.data_ptr(), c2.data_ptr(), 0.75, 1, match.data_ptr(), nm.data_ptr(), sp)))
    print(json.dumps({"op": "SearchByBoW(pKF1, pKF2)", "pairs": n, "nodes": 100, "ms_per_batch": round(ms, 3),
                      "pairs_per_s": round(n / ms * 1e3, 1), "mean_matches": float(nm.cpu().numpy().mean())}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Random-parameter sweep of the device paths next to the extraction (rectification, stereo matching,
undistortion and grid, the matchers) against the oracle. One-off robustness tool: prints the failing draws, exit status 1 if any.

Usage: python tools/fuzz_next_rows.py [--cases 60] [--seed 1]
"""
import argparse
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import orb_oracle as O  # noqa: E402
from visual_slam_amd import orbhip, synth  # noqa: E402


def shifted(img, dx, dy):
    out = img.copy()
    h, w = img.shape
    out[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)] = img[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        w = int(rng.integers(160, 900)); h = int(rng.integers(120, 600))
        nfeat = int(rng.choice([300, 800, 1500]))
        kind = str(rng.choice(["blocks", "lowc", "noise"]))
        prm = (nfeat, 1.2, int(rng.integers(3, 9)), 20, 7)
        img = synth.frame(kind, w, h, 1000 + case)
        tag = "case %d %dx%d %s n%d l%d" % (case, w, h, kind, nfeat, prm[2])
        try:
            o = O.OracleExtractor(*prm)
            ex = orbhip.ORBextractor(*prm)
        except (RuntimeError, orbhip.OrbHipError) as e:
            print(tag, "skipped:", e); continue
        # rectification with random strength, partly far outside the source
        mx, my = synth.rectify_maps(w, h, float(rng.uniform(-8, 8)), float(rng.uniform(-0.5, 0.5)),
                                    (float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40))))
        if rng.random() < 0.3:
            mx = mx + rng.uniform(-0.01, 0.01, mx.shape).astype(np.float32)     # rounding ties of the 1/32 grid
        rect = O.remap_linear(img, mx, my)
        ex.set_rectification(mx, my)
        try:
            kp, desc = ex(img)
            okp, odesc = o.extract(rect)
            ok = np.array_equal(ex.pyramid_level(0, with_border=True), o.level(0, with_border=True)) and len(kp) == len(okp) and \
                np.array_equal(desc, odesc)
        except (RuntimeError, orbhip.OrbHipError) as e:
            print(tag, "extract refused:", e); ex.set_rectification(None, None); continue
        if not ok:
            bad += 1; print(tag, "RECTIFICATION MISMATCH")
        ex.set_rectification(None, None)
        # stereo matching of (img, img moved by a random disparity) and undistortion + grid of the left keypoints
        try:
            exR = orbhip.ORBextractor(*prm); oR = O.OracleExtractor(*prm)
            disp = int(rng.integers(1, 40))
            right = shifted(img, -disp, int(rng.integers(-1, 2)))
            kl, dl = ex(img); kr, dr = exR(right)
            okl, odl = o.extract(img); okr, odr = oR.extract(right)
            fxs = float(rng.uniform(300, 900)); bfs = fxs * float(rng.uniform(0.05, 0.6))
            gu, gd, gn = orbhip.ComputeStereoMatches(ex, exR, kl, dl, kr, dr, bfs / fxs, bfs)
            ou, od, on_ = O.stereo_match(o, oR, okl, odl, okr, odr, bfs / fxs, bfs)
            if gn != on_ or not np.array_equal(gu.view(np.uint32), ou.view(np.uint32)) or not np.array_equal(gd.view(np.uint32), od.view(np.uint32)):
                bad += 1; print(tag, "STEREO MISMATCH", gn, on_)
            camd = (fxs, fxs * float(rng.uniform(0.9, 1.1)), w / 2.0, h / 2.0, float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.2, 0.2)),
                    float(rng.uniform(-0.002, 0.002)), float(rng.uniform(-0.002, 0.002)), float(rng.uniform(-0.1, 0.1)))
            bd = O.image_bounds(camd, w, h)
            if bd[1] > bd[0] and bd[3] > bd[2] and len(okl):
                gb = orbhip.ComputeImageBounds(camd, w, h)
                gun, ggs, ggi = orbhip.UndistortAndAssignToGrid(ex, kl, camd, gb)
                oun, ogs, ogi = O.undistort_grid(camd, bd, okl)
                if not np.array_equal(gb.view(np.uint32), bd.view(np.uint32)) or not np.array_equal(gun.view(np.uint8), oun.view(np.uint8)) or \
                        not np.array_equal(ggs, ogs) or not np.array_equal(ggi, ogi):
                    bad += 1; print(tag, "UNDISTORT / GRID MISMATCH")
        except (RuntimeError, orbhip.OrbHipError) as e:
            print(tag, "stereo refused:", e)
        # matchers on (img, shifted img)
        dx, dy = int(rng.integers(-20, 21)), int(rng.integers(-12, 13))
        cam = (float(rng.uniform(300, 900)), float(rng.uniform(300, 900)), w / 2.0 + float(rng.uniform(-20, 20)), h / 2.0, 0.0, 0, 0, 0, 0)
        if rng.random() < 0.5:
            cam = cam[:4] + (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.2, 0.2)), 0.001, -0.002, 0.0)
        k1, d1 = o.extract(img)
        k2, d2 = o.extract(shifted(img, dx, dy))
        if len(k1) == 0 or len(k2) == 0:
            continue
        bounds = O.image_bounds(cam, w, h)
        if not (bounds[1] > bounds[0] and bounds[3] > bounds[2]):
            continue                                         # a distortion so strong that the frame folds over: no grid
        u1 = O.undistort_grid(cam, bounds, k1)[0]; u2 = O.undistort_grid(cam, bounds, k2)[0]
        Z = float(rng.uniform(2, 20))
        fx, fy, cx, cy = cam[:4]
        world = np.stack([(u1["x"] - cx) * Z / fx, (u1["y"] - cy) * Z / fy, np.full(len(u1), Z)], axis=1).astype(np.float32)
        world += rng.normal(0, 0.01, world.shape).astype(np.float32)
        a = float(rng.uniform(-0.02, 0.02))
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Tcw = np.concatenate([R.ravel(), [dx * Z / fx, dy * Z / fy, float(rng.uniform(-0.3, 0.3))]]).astype(np.float32)
        flags = ((rng.random(len(u1)) < 0.8).astype(np.uint8) | ((rng.random(len(u1)) < rng.random()).astype(np.uint8) << 1))
        ur = None
        if rng.random() < 0.5:
            ur = (u2["x"] - 0.4 * fx / Z + rng.normal(0, 3, len(u2))).astype(np.float32)
            ur[rng.random(len(u2)) < 0.3] = -1
        th = float(rng.choice([3, 7, 15, 30])); mode = int(rng.integers(0, 3)); ori = bool(rng.random() < 0.7)
        sf = o.table(0)
        om, on = O.search_by_projection(u1, world, flags, d1, u2, d2, ur, Tcw, cam[:4], 0.4 * fx, bounds, sf, th, mode, ori)
        gm, gn = orbhip.SearchByProjection(ex, u1, world, flags, d1, u2, d2, ur, Tcw, cam[:4], 0.4 * fx, bounds, th, mode, ori)
        if gn != on or not np.array_equal(gm, om):
            bad += 1; print(tag, "PROJECTION MISMATCH", gn, on, int((gm != om).sum()))
        prev = np.stack([u1["x"], u1["y"]], axis=1)
        win = int(rng.choice([20, 100])); ratio = float(rng.choice([0.7, 0.9]))
        om12, op, on2 = O.search_for_initialization(u1, d1, u2, d2, bounds, prev, win, ratio, ori)
        gm12, gp, gn2 = orbhip.SearchForInitialization(ex, u1, d1, u2, d2, bounds, prev, win, ratio, ori)
        if gn2 != on2 or not np.array_equal(gm12, om12) or not np.array_equal(gp.view(np.uint32), op.view(np.uint32)):
            bad += 1; print(tag, "INITIALIZATION MISMATCH", gn2, on2)
        # the relocalisation variant of the projection matcher: distance ranges around the true depth, some far off
        ref = Z * rng.choice([1.0, 1.0, 1.3, 0.7, 4.0, 0.1], len(u1))
        maxd = (ref * sf[u1["octave"]]).astype(np.float32)
        drange = np.stack([maxd / sf[-1], maxd], axis=1).astype(np.float32)
        occ_kf = (rng.random(len(u2)) < rng.uniform(0, 0.5)).astype(np.uint8) if rng.random() < 0.7 else None
        odist = int(rng.choice([40, 64, 100]))
        om4, on4 = O.search_by_projection_keyframe(u1, world, drange, flags & 1, d1, u2, d2, occ_kf, Tcw, cam[:4], bounds, sf, prm[1], th,
                                                   odist, ori)
        gm4, gn4 = orbhip.SearchByProjectionKeyFrame(ex, u1, world, drange, flags & 1, d1, u2, d2, occ_kf, Tcw, cam[:4], bounds, th, odist, ori)
        if gn4 != on4 or not np.array_equal(gm4, om4):
            bad += 1; print(tag, "KEYFRAME PROJECTION MISMATCH", gn4, on4, int((gm4 != om4).sum()))
        # the loop-closing variant: a similarity with a random scale, normals around the viewing direction
        Ow_ = -(R.T @ Tcw[9:12].astype(np.float64))
        view = world.astype(np.float64) - Ow_
        nrm = view / np.maximum(np.linalg.norm(view, axis=1, keepdims=True), 1e-9) + rng.normal(0, 0.5, view.shape)
        nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
        Scw = (Tcw.astype(np.float64) * float(rng.uniform(0.3, 3.0))).astype(np.float32)
        th_i = int(rng.choice([3, 4, 10]))
        om6, on6 = O.search_by_projection_sim3(world, nrm, drange, flags & 1, d1, u2, d2, occ_kf, Scw, cam[:4], bounds, sf, prm[1], th_i)
        gm6, gn6 = orbhip.SearchByProjectionSim3(ex, world, nrm, drange, flags & 1, d1, u2, d2, occ_kf, Scw, cam[:4], bounds, th_i)
        if gn6 != on6 or not np.array_equal(gm6, om6):
            bad += 1; print(tag, "SIM3 PROJECTION MISMATCH", gn6, on6, int((gm6 != om6).sum()))
        # the bag-of-words matcher with synthetic node ids (a hash of descriptor bits; now and then a feature without a node)
        nn = int(rng.choice([1, 7, 100, 1000]))
        def nodes(d):
            v = (d[:, 0].astype(np.uint32) | (d[:, 9].astype(np.uint32) << 8)) & 0xFFF
            out = (v * 2654435761 % nn).astype(np.uint32) * 11 + 5
            out[rng.random(len(d)) < 0.03] = 0xFFFFFFFF
            return out
        nk, nf = nodes(d1), nodes(d2)
        ratio_b = float(rng.choice([0.6, 0.7, 0.9]))
        if max(len(u1), len(u2)) <= 4096:
            om5, on5 = O.search_by_bow(u1, d1, nk, flags & 1, k2, d2, nf, ratio_b, ori)
            gm5, gn5 = orbhip.SearchByBoW(ex, u1, d1, nk, flags & 1, k2, d2, nf, ratio_b, ori)
            if gn5 != on5 or not np.array_equal(gm5, om5):
                bad += 1; print(tag, "BOW MISMATCH", gn5, on5, int((gm5 != om5).sum()))
        # the local-map matcher: the first view's keypoints (several copies, shuffled) as map points
        copies = int(rng.integers(1, 4))
        src = np.tile(np.arange(len(u1)), copies); rng.shuffle(src)
        mp = np.zeros(len(src), O.TRACK_DTYPE)
        jit = float(rng.uniform(0, 4))
        mp["proj_x"] = u1["x"][src] + dx + rng.normal(0, jit, len(src)); mp["proj_y"] = u1["y"][src] + dy + rng.normal(0, jit, len(src))
        mp["proj_xr"] = mp["proj_x"] - 0.4 * fx / Z
        mp["view_cos"] = rng.choice(np.array([0.9995, 0.998, 0.93, 0.9981], np.float32), len(src))
        mp["level"] = u1["octave"][src] + rng.integers(-1, 2, len(src))             # now and then outside the pyramid
        mp["flags"] = (rng.random(len(src)) < 0.9).astype(np.int32) | ((rng.random(len(src)) < rng.random()).astype(np.int32) << 1)
        dmp = d1[src].copy()
        dmp ^= ((rng.random(dmp.shape) < rng.uniform(0, 0.08)) * (1 << rng.integers(0, 8, dmp.shape))).astype(np.uint8)
        occ = (rng.random(len(u2)) < rng.uniform(0, 0.3)).astype(np.uint8) if rng.random() < 0.7 else None
        thl = float(rng.choice([1, 3, 5, 8])); ratio = float(rng.choice([0.6, 0.8, 0.9]))
        try:
            gm3, gn3 = orbhip.SearchLocalPoints(ex, mp, dmp, u2, d2, ur, occ, bounds, thl, ratio)
        except orbhip.OrbHipError as e:
            print(tag, "local map refused:", e); continue
        om3, on3 = O.search_map_points(mp, dmp, u2, d2, ur, occ, bounds, sf, thl, ratio)
        if gn3 != on3 or not np.array_equal(gm3, om3):
            bad += 1; print(tag, "LOCAL MAP MISMATCH", gn3, on3, int((gm3 != om3).sum()))
    print("fuzz: %d cases, %d mismatches" % (args.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

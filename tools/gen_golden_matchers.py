#!/usr/bin/env python3
"""Regression fixtures for the matcher restatements: one small fixed scenario per function, the oracle's
result reduced to (nmatches, sha1 of the result array). Produced by the oracle itself (nothing in the
reference provides vectors for these functions), so they guard the oracle and the device against
accidental change, they do not pin either to the reference.

Usage: python tools/gen_golden_matchers.py > tests/golden/matchers.json
"""
import hashlib
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from visual_slam_amd import synth  # noqa: E402

CAM = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
W, H, PRM = 752, 480, (1000, 1.2, 8, 20, 7)
Z = 5.0


def shifted(img, dx, dy):
    out = img.copy()
    h, w = img.shape
    out[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)] = img[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)]
    return out


def scenario(O):
    """Everything the matchers need, from seeded generators only."""
    rng = np.random.default_rng(20240)
    a = synth.frame("blocks", W, H, 4242)
    dx, dy = 7, -4
    e = O.OracleExtractor(*PRM)
    k1, d1 = e.extract(a)
    k2, d2 = e.extract(shifted(a, dx, dy))
    bounds = O.image_bounds(CAM, W, H)
    u1 = O.undistort_grid(CAM, bounds, k1)[0]
    u2 = O.undistort_grid(CAM, bounds, k2)[0]
    fx, fy, cx, cy = CAM[:4]
    sf = e.table(0)
    n1, n2 = len(u1), len(u2)
    world = np.stack([(u1["x"] - cx) * Z / fx, (u1["y"] - cy) * Z / fy, np.full(n1, Z)], axis=1).astype(np.float32)
    Tcw = np.concatenate([np.eye(3).ravel(), [dx * Z / fx, dy * Z / fy, 0.0]]).astype(np.float32)
    flags = ((rng.random(n1) < 0.85).astype(np.uint8) | ((rng.random(n1) < 0.7).astype(np.uint8) << 1))
    uright = (u2["x"] - 0.11 * fx / Z).astype(np.float32)
    uright[rng.random(n2) < 0.3] = -1.0
    maxd = (Z * rng.choice([1.0, 1.0, 1.3, 0.7, 4.0], n1) * sf[u1["octave"]]).astype(np.float32)
    drange = np.stack([maxd / sf[-1], maxd], axis=1).astype(np.float32)
    occupied = (rng.random(n2) < 0.25).astype(np.uint8)
    view = world + Tcw[9:12]
    nrm = view / np.linalg.norm(view, axis=1, keepdims=True) + rng.normal(0, 0.4, (n1, 3))
    normal = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    src = np.tile(np.arange(n1), 2); rng.shuffle(src)
    mp = np.zeros(len(src), O.TRACK_DTYPE)
    mp["proj_x"] = u1["x"][src] + dx + rng.normal(0, 1.5, len(src)); mp["proj_y"] = u1["y"][src] + dy + rng.normal(0, 1.5, len(src))
    mp["proj_xr"] = mp["proj_x"] - 0.11 * fx / Z
    mp["view_cos"] = np.where(rng.random(len(src)) < 0.5, 0.9995, 0.93)
    mp["level"] = u1["octave"][src]
    mp["flags"] = (rng.random(len(src)) < 0.9).astype(np.int32) | ((rng.random(len(src)) < 0.7).astype(np.int32) << 1)
    mpw = np.zeros(n1, O.MAPPOINT_DTYPE)
    mpw["pos"] = world; mpw["normal"] = normal; mpw["min_distance"] = drange[:, 0]; mpw["max_distance"] = drange[:, 1]
    mpw["flags"] = flags.astype(np.int32)

    def nodes(d):
        v = (d[:, 0].astype(np.uint32) | (d[:, 7].astype(np.uint32) << 8)) & 0xFFF
        return (v * 2654435761 % 64).astype(np.uint32) * 5 + 9
    pose15 = np.concatenate([Tcw, -Tcw[9:12]]).astype(np.float32)
    return dict(u1=u1, d1=d1, u2=u2, d2=d2, k1=k1, k2=k2, bounds=bounds, sf=sf, world=world, Tcw=Tcw, flags=flags, uright=uright,
                drange=drange, occupied=occupied, normal=normal, mp=mp, dmp=d1[src].copy(), mpw=mpw, pose15=pose15,
                nk=nodes(d1), nf=nodes(d2), f2=(rng.random(n2) < 0.8).astype(np.uint8), Scw=(Tcw * np.float32(1.25)).astype(np.float32),
                prev=np.stack([u1["x"], u1["y"]], axis=1).astype(np.float32))


def digest(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_all(s, B):
    """B: the oracle module or an adapter with the same function names; returns {name: [count, sha1]}."""
    out = {}
    m, p, n = B.search_for_initialization(s["u1"], s["d1"], s["u2"], s["d2"], s["bounds"], s["prev"], 100, 0.9, True)
    out["search_for_initialization"] = [int(n), digest(m.astype(np.int32)), digest(p.astype(np.float32))]
    m, n = B.search_by_projection(s["u1"], s["world"], s["flags"], s["d1"], s["u2"], s["d2"], s["uright"], s["Tcw"], CAM[:4], 0.11 * CAM[0],
                                  s["bounds"], s["sf"], 7.0, 0, True)
    out["search_by_projection"] = [int(n), digest(m.astype(np.int32))]
    m, n = B.search_by_projection_keyframe(s["u1"], s["world"], s["drange"], s["flags"] & 1, s["d1"], s["u2"], s["d2"], s["occupied"],
                                           s["Tcw"], CAM[:4], s["bounds"], s["sf"], 1.2, 10.0, 100, True)
    out["search_by_projection_keyframe"] = [int(n), digest(m.astype(np.int32))]
    m, n = B.search_by_projection_sim3(s["world"], s["normal"], s["drange"], s["flags"] & 1, s["d1"], s["u2"], s["d2"], s["occupied"],
                                       s["Scw"], CAM[:4], s["bounds"], s["sf"], 1.2, 10)
    out["search_by_projection_sim3"] = [int(n), digest(m.astype(np.int32))]
    m, n = B.search_map_points(s["mp"], s["dmp"], s["u2"], s["d2"], s["uright"], s["occupied"], s["bounds"], s["sf"], 3.0, 0.8)
    out["search_map_points"] = [int(n), digest(m.astype(np.int32))]
    t, n = B.is_in_frustum(s["mpw"], s["pose15"], CAM[:4], 0.11 * CAM[0], s["bounds"], 1.2, 8, 0.5)
    out["is_in_frustum"] = [int(n), digest(t)]
    m, n = B.search_by_bow(s["u1"], s["d1"], s["nk"], s["flags"] & 1, s["k2"], s["d2"], s["nf"], 0.7, True)
    out["search_by_bow"] = [int(n), digest(m.astype(np.int32))]
    m, n = B.search_by_bow_keyframes(s["u1"], s["d1"], s["nk"], s["flags"] & 1, s["u2"], s["d2"], s["nf"], s["f2"], 0.75, True)
    out["search_by_bow_keyframes"] = [int(n), digest(m.astype(np.int32))]
    return out


if __name__ == "__main__":
    from oracle import orb_oracle as O
    print(json.dumps(run_all(scenario(O), O), indent=1))

#!/usr/bin/env python3
"""Latency of the host-array (one item per call) entry points a tracking front end makes per frame besides the
extraction: keypoint undistortion + grid, the motion-model matcher, the frustum test. KITTI-sized frame, 2000 features,
4000 map points. Medians of 40 calls through the Python mirror (its own array handling included).
Usage: python tools/host_call_latency.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from visual_slam_amd import orbhip, synth      # noqa: E402

w, h = 1241, 376
ex = orbhip.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=1)
img = synth.frame("blocks", w, h, 3)
kp, desc = ex(img, fetch_pyramid=False)
cam = (718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0, 0.0, 0.0)
bounds = (0.0, float(w), 0.0, float(h))


def med(fn, n=40):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3


print("extraction (orbhip_extract): %.3f ms" % med(lambda: ex(img, fetch_pyramid=False)))
print("UndistortKeyPoints + AssignFeaturesToGrid (orbhip_undistort_grid), %d keypoints: %.3f ms"
      % (len(kp), med(lambda: orbhip.UndistortAndAssignToGrid(ex, kp, cam, bounds))))
n = len(kp)
rng = np.random.default_rng(1)
world = np.stack([(kp["x"] - cam[2]) / cam[0] * 10.0, (kp["y"] - cam[3]) / cam[1] * 10.0, np.full(n, 10.0)], axis=1).astype(np.float32)
flags = np.ones(n, np.uint8)
Tcw = np.eye(4, dtype=np.float32)[:3].copy()
print("SearchByProjection(CurrentFrame, LastFrame) (orbhip_search_by_projection), %d x %d: %.3f ms"
      % (n, n, med(lambda: orbhip.SearchByProjection(ex, kp, world, flags, desc, kp, desc, None, Tcw, cam[:4], 386.1448, bounds, 15.0, 0))))

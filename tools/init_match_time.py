#!/usr/bin/env python3
"""Time of one ORBmatcher::SearchForInitialization call through the host-array entry point (copies
included) next to the oracle on one host thread, on a KITTI-sized frame pair with a known shift.

Usage: python tools/init_match_time.py
"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from tests.test_search_init import frames, KITTI
from visual_slam_amd import orbhip
from oracle import orb_oracle as O
(u1, d1), (u2, d2), bounds, prev = frames(1241, 376, 62, 9, 3, KITTI, (2000, 1.2, 8, 20, 7))
ex = orbhip.ORBextractor(2000, 1.2, 8, 20, 7)
orbhip.SearchForInitialization(ex, u1, d1, u2, d2, bounds, prev)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); r = orbhip.SearchForInitialization(ex, u1, d1, u2, d2, bounds, prev); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); o = O.search_for_initialization(u1, d1, u2, d2, bounds, prev); tc = time.perf_counter() - t0
print("device call median %.3f ms, oracle %.3f ms, matches %d / %d" % (np.median(ts) * 1e3, tc * 1e3, r[2], o[2]))

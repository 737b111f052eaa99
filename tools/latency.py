#!/usr/bin/env python3
"""Single-frame latency of the host-buffer entry point (orbhip_extract), the way a SLAM front end
calls the extractor: one image in, keypoints and descriptors out, per call.
Usage: python tools/latency.py [C1_tum|C2_kitti|C3_euroc|C4_1080p_2000f] [--pyramid]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench                                   # noqa: E402
from visual_slam_amd import orbhip, synth      # noqa: E402

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "C4_1080p_2000f")
w, h, nfeat, scale, levels, ini, mn = bench.WORKLOADS[name]
fetch = "--pyramid" in sys.argv
ex = orbhip.ORBextractor(nfeat, scale, levels, ini, mn, max_batch=1)
imgs = synth.batch("blocks", w, h, 8)
for im in imgs[:3]:
    ex(im, fetch_pyramid=fetch)
ts = []
for k in range(40):
    t0 = time.perf_counter()
    kp, desc = ex(imgs[k % 8], fetch_pyramid=fetch)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("%s single-frame latency (host in, host out%s): median %.3f ms, p90 %.3f ms, min %.3f ms, %d keypoints"
      % (name, ", pyramid fetched" if fetch else "", np.median(ts), np.percentile(ts, 90), ts.min(), len(kp)))

#!/usr/bin/env python3
"""Single-frame latency of the host-buffer entry point (orbhip_extract), the way a SLAM front end
calls the extractor: one image in, keypoints and descriptors out, per call.
Usage: python tools/latency.py [C1_tum|C2_kitti|C3_euroc|C4_1080p_2000f] [--pyramid]
       python tools/latency.py [workload] --stereo     one rectified pair per call: the two extractions on two threads as in
                                                       the reference's Frame constructor (src/Frame.cc:114-117), then
                                                       ComputeStereoMatches; and the same pair through orbhip_extract_stereo_batch"""
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
if "--stereo" in sys.argv:
    import threading
    fx, bf = 435.2, 47.9
    exl = orbhip.ORBextractor(nfeat, scale, levels, ini, mn, max_batch=1)
    exr = orbhip.ORBextractor(nfeat, scale, levels, ini, mn, max_batch=1)
    pairs = [synth.stereo_pair("blocks", w, h, 10 + i, (3, 12, 40)) for i in range(8)]

    def one(l, r):
        out = {}
        tl = threading.Thread(target=lambda: out.__setitem__("l", exl(l, fetch_pyramid=False)))
        tr = threading.Thread(target=lambda: out.__setitem__("r", exr(r, fetch_pyramid=False)))
        tl.start(); tr.start(); tl.join(); tr.join()
        (kl, dl), (kr, dr) = out["l"], out["r"]
        return orbhip.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, bf / fx, bf)
    for l, r in pairs[:3]:
        one(l, r)
    ts = []
    for k in range(40):
        t0 = time.perf_counter()
        u, d, m = one(*pairs[k % 8])
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("%s stereo pair latency, two Python threads + ComputeStereoMatches (host in, host out): median %.3f ms, p90 %.3f ms, min %.3f ms, %d matches"
          % (name, np.median(ts), np.percentile(ts, 90), ts.min(), m))
    te, tm = [], []
    for k in range(43):
        l, r = pairs[k % 8]
        t0 = time.perf_counter()
        kl, dl = exl(l, fetch_pyramid=False)
        kr, dr = exr(r, fetch_pyramid=False)
        t1 = time.perf_counter()
        u, d, m = orbhip.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, bf / fx, bf)
        if k >= 3:
            te.append(t1 - t0); tm.append(time.perf_counter() - t1)
    print("%s stereo pair latency, left then right extraction on one thread: median %.3f ms; ComputeStereoMatches alone: median %.3f ms"
          % (name, np.median(te) * 1e3, np.median(tm) * 1e3))
    import ctypes as C
    cap = exl.max_keypoints(w, h)
    kl, kr = np.zeros((1, cap), orbhip.KP_DTYPE), np.zeros((1, cap), orbhip.KP_DTYPE)
    dl, dr = np.zeros((1, cap, 32), np.uint8), np.zeros((1, cap, 32), np.uint8)
    cl, cr, nm = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
    ur, dep = np.zeros((1, cap), np.float32), np.zeros((1, cap), np.float32)
    ts = []
    for k in range(43):
        l, r = pairs[k % 8]
        t0 = time.perf_counter()
        rc = orbhip.lib().orbhip_extract_stereo_batch(exl._h, exr._h, l.ctypes.data, r.ctypes.data, 1, w, h, w, w * h, kl.ctypes.data,
                                                      dl.ctypes.data, cl.ctypes.data, kr.ctypes.data, dr.ctypes.data, cr.ctypes.data, cap,
                                                      C.c_float(bf / fx), C.c_float(bf), ur.ctypes.data, dep.ctypes.data, nm.ctypes.data)
        assert rc == 0, orbhip.lib().orbhip_last_error(exl._h)
        if k >= 3:
            ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("%s stereo pair latency, one orbhip_extract_stereo_batch call (host in, host out): median %.3f ms, p90 %.3f ms, min %.3f ms, %d matches"
          % (name, np.median(ts), np.percentile(ts, 90), ts.min(), nm[0]))
    sys.exit(0)
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

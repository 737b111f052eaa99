#!/usr/bin/env python3
"""Rate of the host batch entry point (images in host memory in, results in host memory out) over chunk sizes,
chunks in flight and packing threads, for pageable and page-locked input.
Usage: python tools/hostfed_sweep.py [workload] [frames]
       python tools/hostfed_sweep.py --stereo [workload] [pairs]     (the stereo batch call: both sides up, extracted and
                                                                      matched on the device, all results down)"""
import itertools
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench                                   # noqa: E402
import torch                                   # noqa: E402
from visual_slam_amd import orbhip, synth      # noqa: E402

stereo = "--stereo" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--stereo"]
name = argv[0] if argv else ("C3_euroc" if stereo else "C4_1080p_2000f")
nh = int(argv[1]) if len(argv) > 1 else 512
w, h, nfeat, scale, levels, ini, mn = bench.WORKLOADS[name]
if stereo:
    # pairs: the right image is the left one shifted by a few columns (what bench.py --stereo feeds)
    base = synth.batch("blocks", w, h, 16)
    left = torch.from_numpy(np.stack([np.roll(base[i % 16], (37 * (i // 16)) % h, axis=0) for i in range(nh)]))
    right = torch.from_numpy(np.roll(left.numpy(), -7, axis=2).copy())
    pl, pr = left.pin_memory(), right.pin_memory()
    dst = torch.empty((nh, h, w), dtype=torch.uint8, device="cuda")
    dst.copy_(pl, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dst.copy_(pl, non_blocking=True)
    torch.cuda.synchronize()
    h2d = 3 * pl.numel() / (time.perf_counter() - t0) / 1e9
    print("%s, %d pairs: H2D copy %.1f GB/s = %.0f pairs/s of images" % (name, nh, h2d, h2d * 1e9 / (2 * w * h)), flush=True)
    fx, bf = 435.2, 47.9
    import ctypes as C
    for pipe in ("1", "0"):
        os.environ["ORBHIP_HOST_PIPELINE"] = pipe
        eL, eR = orbhip.ORBextractor(nfeat, scale, levels, ini, mn), orbhip.ORBextractor(nfeat, scale, levels, ini, mn)
        cap = eL.max_keypoints(w, h)
        kl, kr = np.zeros((nh, cap), orbhip.KP_DTYPE), np.zeros((nh, cap), orbhip.KP_DTYPE)
        dl, dr = np.zeros((nh, cap, 32), np.uint8), np.zeros((nh, cap, 32), np.uint8)
        cl, cr, nm = np.zeros(nh, np.int32), np.zeros(nh, np.int32), np.zeros(nh, np.int32)
        ur, dep = np.zeros((nh, cap), np.float32), np.zeros((nh, cap), np.float32)
        out = []
        for a, b in ((left, right), (pl, pr)):
            a, b = a.numpy(), b.numpy()
            best = 0.0
            for rep in range(3):
                t0 = time.perf_counter()
                rc = orbhip.lib().orbhip_extract_stereo_batch(eL._h, eR._h, a.ctypes.data, b.ctypes.data, nh, w, h, w, w * h,
                                                              kl.ctypes.data, dl.ctypes.data, cl.ctypes.data, kr.ctypes.data, dr.ctypes.data,
                                                              cr.ctypes.data, cap, C.c_float(bf / fx), C.c_float(bf), ur.ctypes.data,
                                                              dep.ctypes.data, nm.ctypes.data)
                dt = time.perf_counter() - t0
                assert rc == 0, orbhip.lib().orbhip_last_error(eL._h)
                if rep:
                    best = max(best, nh / dt)
            out.append(best)
        print("pipeline %s: pageable %8.0f  pinned %8.0f pairs/s, %.0f keypoints and %.0f matches per pair"
              % (pipe, out[0], out[1], cl.mean(), nm.mean()), flush=True)
        eL.close(); eR.close()
    sys.exit(0)
base = synth.batch("blocks", w, h, 16)
pageable = torch.from_numpy(np.stack([np.roll(base[i % 16], (37 * (i // 16)) % h, axis=0) for i in range(nh)]))
pinned = pageable.pin_memory()
dst = torch.empty((nh, h, w), dtype=torch.uint8, device="cuda")
dst.copy_(pinned, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    dst.copy_(pinned, non_blocking=True)
torch.cuda.synchronize()
h2d = 3 * pinned.numel() / (time.perf_counter() - t0) / 1e9
print("%s, %d frames: H2D copy %.1f GB/s = %.0f frames/s" % (name, nh, h2d, h2d * 1e9 / (w * h)), flush=True)
chunks = [int(c) for c in os.environ.get("SWEEP_CHUNKS", "0,16,64,128").split(",")]
lanes = [int(c) for c in os.environ.get("SWEEP_LANES", "3,4").split(",")]
threads = [int(c) for c in os.environ.get("SWEEP_THREADS", "4,8").split(",")]
for chunk, nl, nt in itertools.product(chunks, lanes, threads):
    os.environ.update(ORBHIP_HOST_CHUNK=str(chunk), ORBHIP_HOST_LANES=str(nl), ORBHIP_PACK_THREADS=str(nt))
    ex = orbhip.ORBextractor(nfeat, scale, levels, ini, mn)
    cap = ex.max_keypoints(w, h)
    kp = np.zeros((nh, cap), orbhip.KP_DTYPE); desc = np.zeros((nh, cap, 32), np.uint8); cnt = np.zeros(nh, np.int32)
    out = []
    for t in (pageable, pinned):
        arr = t.numpy()
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            rc = orbhip.lib().orbhip_extract_batch(ex._h, arr.ctypes.data, nh, w, h, w, w * h, kp.ctypes.data, desc.ctypes.data,
                                                   cnt.ctypes.data, cap)
            dt = time.perf_counter() - t0
            assert rc == 0, orbhip.lib().orbhip_last_error(ex._h)
            if rep:
                best = max(best, nh / dt)
        out.append(best)
    print("chunk %4d lanes %d pack threads %d: pageable %8.0f  pinned %8.0f frames/s" % (chunk, nl, nt, out[0], out[1]), flush=True)
    ex.close()

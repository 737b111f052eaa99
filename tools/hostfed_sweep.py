#!/usr/bin/env python3
"""Rate of the host batch entry point (images in host memory in, results in host memory out) over chunk sizes,
chunks in flight and packing threads, for pageable and page-locked input.
Usage: python tools/hostfed_sweep.py [workload] [frames]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "C4_1080p_2000f"
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w, h, nfeat, scale, levels, ini, mn = bench.WORKLOADS[name]
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

#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/ with the CPU oracle.

The reference has no fixtures of its own and cannot be built here (it needs OpenCV), so these
vectors are produced by the oracle (oracle/orb_oracle.cpp), which is pinned by the known-answer
tests in tests/test_oracle_kat.py. Each .npz holds the input image, the extractor parameters, the
expected keypoints and descriptors, the per-level candidate counts and a CRC of every bordered
pyramid level.  Usage: python tools/gen_golden.py
"""
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import orb_oracle as O          # noqa: E402
from visual_slam_amd import synth           # noqa: E402

CASES = [
    # name, kind, w, h, index, (nfeatures, scale, levels, iniTh, minTh)
    ("blocks_240x180", "blocks", 240, 180, 11, (300, 1.2, 4, 20, 7)),
    ("noise_200x150", "noise", 200, 150, 12, (250, 1.2, 4, 20, 7)),
    ("lowc_256x192", "lowc", 256, 192, 13, (300, 1.2, 5, 20, 7)),
    ("blocks_deep_320x200", "blocks", 320, 200, 14, (400, 1.1, 8, 12, 7)),
]

out = ROOT / "tests" / "golden"
out.mkdir(parents=True, exist_ok=True)
for name, kind, w, h, idx, prm in CASES:
    img = synth.frame(kind, w, h, idx)
    e = O.OracleExtractor(*prm)
    kp, desc = e.extract(img)
    nl = prm[2]
    crcs = np.array([zlib.crc32(e.level(l, with_border=True).tobytes()) for l in range(nl)], np.uint32)
    ncand = np.array([len(e.candidates(l)) for l in range(nl)], np.int32)
    np.savez_compressed(out / (name + ".npz"), image=img, params=np.array(prm, np.float64), keypoints=kp,
                        descriptors=desc, level_crc32=crcs, candidates_per_level=ncand)
    print(name, len(kp), ncand.tolist())

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


# ---- stereo matching (oracle restatement of reference src/Frame.cc:481-653): tests/golden/stereo/*.npz
STEREO_CASES = [
    # name, kind, w, h, index, disparities per horizontal band, extractor parameters
    ("blocks_320x240", "blocks", 320, 240, 21, (3, 9, 20), (400, 1.2, 6, 20, 7)),
    ("lowc_288x216", "lowc", 288, 216, 22, (5,), (300, 1.2, 4, 20, 7)),
]
FX, BF = 718.856, 386.1448          # Examples/Stereo/KITTI00-02.yaml
sout = out / "stereo"
sout.mkdir(exist_ok=True)
for name, kind, w, h, idx, disps, prm in STEREO_CASES:
    left, right = synth.stereo_pair(kind, w, h, idx, disps)
    eL, eR = O.OracleExtractor(*prm), O.OracleExtractor(*prm)
    kl, dl = eL.extract(left)
    kr, dr = eR.extract(right)
    ur, dp, kept = O.stereo_match(eL, eR, kl, dl, kr, dr, BF / FX, BF)
    np.savez_compressed(sout / (name + ".npz"), left=left, right=right, params=np.array(prm, np.float64),
                        mb=np.float32(BF / FX), mbf=np.float32(BF), uright=ur, depth=dp, kept=np.int32(kept))
    print("stereo", name, len(kl), len(kr), kept)

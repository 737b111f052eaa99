#!/usr/bin/env python3
"""Random-parameter sweep of the extraction itself against the oracle, stage by stage (pyramid with
border, FAST candidates per level, keypoints, descriptors), single frames and small batches. The
draws are those of tests/test_gpu_parity.py::test_random_parameters with another seed and count.

Usage: python tools/fuzz_extract.py [--cases 200] [--seed 7] [--device]
"""
import argparse
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import orb_oracle as O  # noqa: E402
from tests.parity import DESC_MIN_EXACT, compare_outputs  # noqa: E402
from tests.test_gpu_parity import _random_configs  # noqa: E402
from visual_slam_amd import orbhip, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--device", action="store_true", help="also run every frame device-resident (level 0 read in place)")
    args = ap.parse_args()
    bad = refused = 0
    for case in _random_configs(args.cases, args.seed):
        w, h, nf, sf, nl, ini, mn, kind, index = case
        tag = "%dx%d n%d s%.2f l%d t%d-%d %s" % case[:8]
        img = synth.frame(kind, w, h, index)
        try:
            o = O.OracleExtractor(nf, sf, nl, ini, mn)
            okp, odesc = o.extract(img)
        except RuntimeError:
            try:
                orbhip.ORBextractor(nf, sf, nl, ini, mn)(img)
                bad += 1; print(tag, "ORACLE REFUSES, DEVICE DOES NOT")
            except orbhip.OrbHipError:
                refused += 1
            continue
        try:
            g = orbhip.ORBextractor(nf, sf, nl, ini, mn)
            gkp, gdesc = g(img)
        except orbhip.OrbHipError as e:
            refused += 1; print(tag, "device refuses:", e); continue
        ok = all(np.array_equal(g.pyramid_level(l, with_border=True), o.level(l, with_border=True)) and
                 np.array_equal(g.debug_candidates(l), o.candidates(l)) for l in range(nl))
        r = compare_outputs(gkp, gdesc, okp, odesc)
        ok = ok and r["desc_exact"] >= DESC_MIN_EXACT
        # the same frame inside a batch of three (other frames around it)
        batch = np.stack([synth.frame(kind, w, h, index + 1), img, synth.frame("blocks", w, h, index + 2)])
        bk, bd = g.extract_batch(batch)[1]
        ok = ok and np.array_equal(bk, gkp) and np.array_equal(bd, gdesc)
        # the same frame cropped to a multiple of 4 columns, device-resident, rows padded by a random multiple of 4:
        # level 0 is read in place from that buffer (tests/test_gpu_inplace.py); against the oracle on the crop
        w4 = w & ~3
        if args.device and w4 >= 64:
            from tests.test_gpu_inplace import _run_device
            crop = np.ascontiguousarray(img[:, :w4])
            try:
                ckp, cdesc = O.OracleExtractor(nf, sf, nl, ini, mn).extract(crop)
                res, _keep = _run_device(orbhip.ORBextractor(nf, sf, nl, ini, mn), crop[None], pad=4 * (index % 4), offset=4 * (index % 3))
                dk, dd = res[0]
                ok = ok and len(dk) == len(ckp) and np.array_equal(dk[:, 0], ckp["x"]) and np.array_equal(dk[:, 1], ckp["y"]) and \
                    np.array_equal(dk[:, 3], ckp["angle"]) and np.mean(np.all(dd == cdesc, axis=1)) >= DESC_MIN_EXACT
            except (RuntimeError, orbhip.OrbHipError):
                pass                                              # the crop is one of the shapes both sides refuse
        if not ok:
            bad += 1; print(tag, "MISMATCH", r)
    print("fuzz_extract: %d cases, %d refused by both or by the device with a message, %d mismatches" % (args.cases, refused, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

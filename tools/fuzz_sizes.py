#!/usr/bin/env python3
"""Random sweep of image SIZES and aspect ratios up to the 16384-pixel limit against the oracle (the other
sweep, tools/fuzz_extract.py, stays between 200 x 160 and 900 x 700 pixels).

Every draw runs in two steps, so that a wrong keypoint position shows as a message and not as a memory
fault of the descriptor kernel: first the extraction without that kernel (ORBHIP_DEBUG_SKIP_DESCRIBE),
whose FAST candidates and quadtree keypoints are checked (candidate lists equal to the oracle's, every
keypoint inside the detectable area of its level); only then the whole extraction, compared with the
oracle's output.

Usage: python tools/fuzz_sizes.py [--cases 60] [--seed 1] [--max-mp 24]
"""
import argparse
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import orb_oracle as O  # noqa: E402
from tests.parity import DESC_MIN_EXACT, compare_outputs  # noqa: E402
from visual_slam_amd import orbhip, synth  # noqa: E402


def draws(n, seed, max_mp):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        w = int(round(np.exp(rng.uniform(np.log(64), np.log(16384)))))
        h = int(round(np.exp(rng.uniform(np.log(64), np.log(16384)))))
        if h > 1.9 * w and rng.random() < 0.8:
            w, h = h, w                                        # (both sides refuse most portrait strips: nIni == 0)
        if w * h > max_mp * 1e6:
            continue
        nf = int(rng.choice([100, 500, 1000, 2000, 5000, 12000]))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nl = int(rng.integers(1, 13))
        ini = int(rng.choice([12, 20, 20, 40]))
        mn = int(rng.choice([5, 7, 7, ini]))
        kind = str(rng.choice(["blocks", "blocks", "noise", "lowc"]))
        out.append((w, h, nf, sf, nl, ini, min(mn, ini), kind, 500 + len(out)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-mp", type=float, default=24.0, help="largest image area drawn, in megapixels")
    ap.add_argument("--variants", action="store_true",
                    help="also run every accepted draw device-resident with padded rows and an offset start (level 0 read in place "
                         "where the alignment allows) and as a 3-channel image through the gray conversion: both must equal the plain result")
    args = ap.parse_args()
    if args.variants:
        import torch                                           # (before the library's first HIP call: torch brings its own runtime)
        torch.cuda.init()
    bad = refused = 0
    for case in draws(args.cases, args.seed, args.max_mp):
        w, h, nf, sf, nl, ini, mn, kind, index = case
        tag = "%dx%d n%d s%.2f l%d t%d-%d %s" % case[:8]
        img = synth.frame(kind, w, h, index)
        o = O.OracleExtractor(nf, sf, nl, ini, mn)
        try:
            okp, odesc = o.extract(img)
        except RuntimeError:
            okp = None
        os.environ["ORBHIP_DEBUG_SKIP_DESCRIBE"] = "1"
        try:
            g = orbhip.ORBextractor(nf, sf, nl, ini, mn)
            g(img, fetch_pyramid=False)
        except orbhip.OrbHipError as e:
            os.environ["ORBHIP_DEBUG_SKIP_DESCRIBE"] = "0"
            if okp is None:
                refused += 1
            else:
                bad += 1; print(tag, "DEVICE REFUSES, ORACLE DOES NOT:", e, flush=True)
            continue
        os.environ["ORBHIP_DEBUG_SKIP_DESCRIBE"] = "0"
        if okp is None:
            bad += 1; print(tag, "ORACLE REFUSES, DEVICE DOES NOT", flush=True); continue
        sane = True
        for l in range(nl):
            gc, oc = g.debug_candidates(l), o.candidates(l)
            d = g.debug_level_keypoints(l).astype(int)
            lh, lw = o.level(l).shape
            inside = not len(d) or (d[:, 0].min() >= 19 and d[:, 1].min() >= 19 and d[:, 0].max() < lw - 19 and d[:, 1].max() < lh - 19)
            if not (len(gc) == len(oc) and np.array_equal(gc, oc) and inside):
                sane = False; print(tag, "level %d: candidates equal %s, keypoints inside %s" % (l, len(gc) == len(oc) and bool(np.array_equal(gc, oc)), inside), flush=True)
        if not sane:
            bad += 1; continue
        g = orbhip.ORBextractor(nf, sf, nl, ini, mn)
        try:
            gkp, gdesc = g(img, fetch_pyramid=False)
        except orbhip.OrbHipError as e:
            bad += 1; print(tag, "DEVICE REFUSES THE FULL RUN:", e, flush=True); continue
        try:
            r = compare_outputs(gkp, gdesc, okp, odesc)
        except AssertionError as e:
            bad += 1; print(tag, "MISMATCH:", e, flush=True); continue
        if r["desc_exact"] < DESC_MIN_EXACT:
            bad += 1; print(tag, "MISMATCH", r, flush=True)
        elif len(gkp) and index % 10 == 0:
            print(tag, "ok,", len(gkp), "keypoints", flush=True)
        if args.variants:
            import torch
            dev = torch.device("cuda:0")
            pad, off = 4 * (index % 5), 4 * (index % 3) + (index % 7 == 0)      # (an odd offset now and then: the copied route)
            rs = w + pad
            buf = torch.full((off + h * rs,), 0xA5, dtype=torch.uint8, device=dev)
            buf[off:].view(h, rs)[:, :w] = torch.from_numpy(img).to(dev)
            cap = g.max_keypoints(w, h)
            kp = torch.zeros((1, cap, 7), dtype=torch.float32, device=dev)
            de = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
            cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            g2 = orbhip.ORBextractor(nf, sf, nl, ini, mn)
            try:
                g2.extract_batch_device(buf.data_ptr() + off, 1, w, h, rs, h * rs, kp.data_ptr(), de.data_ptr(), cnt.data_ptr(), cap)
                g2.synchronize()
                torch.cuda.synchronize()
                n = int(cnt[0].item())
                k = kp[0, :n].cpu().numpy()
                same = n == len(gkp) and np.array_equal(k[:, 0], gkp["x"]) and np.array_equal(k[:, 1], gkp["y"]) and \
                    np.array_equal(k[:, 3], gkp["angle"]) and np.array_equal(de[0, :n].cpu().numpy(), gdesc)
                if not same:
                    bad += 1; print(tag, "DEVICE-RESIDENT RESULT DIFFERS (pad %d, offset %d)" % (pad, off), flush=True)
            except orbhip.OrbHipError as e:
                bad += 1; print(tag, "DEVICE-RESIDENT RUN REFUSED:", e, flush=True)
            try:
                ckp, cdesc = orbhip.ORBextractor(nf, sf, nl, ini, mn).extract_color(np.repeat(img[:, :, None], 3, axis=2), rgb=bool(index & 1), fetch_pyramid=False)
                if not (len(ckp) == len(gkp) and np.array_equal(ckp["x"], gkp["x"]) and np.array_equal(ckp["y"], gkp["y"]) and np.array_equal(cdesc, gdesc)):
                    bad += 1; print(tag, "COLOUR INPUT RESULT DIFFERS", flush=True)
            except orbhip.OrbHipError as e:
                bad += 1; print(tag, "COLOUR RUN REFUSED:", e, flush=True)
    print("fuzz_sizes: %d cases, %d refused by both, %d problems" % (args.cases, refused, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

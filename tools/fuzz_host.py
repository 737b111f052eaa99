#!/usr/bin/env python3
"""Random sweep of the host batch entry points (orbhip_extract_batch, orbhip_extract_stereo_batch): image sizes, batch
sizes, row and frame strides, chunk sizes, chunks in flight, packing threads, pageable and page-locked memory, against
the per-frame (per-pair) calls on other handles. Every case runs in this process; the switches are read when a handle
is created.

Usage: python tools/fuzz_host.py [--cases 60] [--seed 1]"""
import argparse
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch                                   # noqa: E402
from visual_slam_amd import orbhip, synth      # noqa: E402

BF, FX = 386.1448, 718.856


def make(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        w = int(rng.integers(120, 700)); h = int(rng.integers(100, 500))
        n = int(rng.integers(8, 60))
        nfeat = int(rng.integers(100, 900)); levels = int(rng.integers(2, 9))
        pad = int(rng.choice([0, 0, 3, 16, 61])); gap = int(rng.choice([0, 0, 2, 7]))
        stereo = bool(rng.integers(0, 2))
        pinned = bool(rng.integers(0, 2))
        env = {"ORBHIP_HOST_CHUNK": str(int(rng.choice([0, 1, 3, 5, 16]))), "ORBHIP_HOST_LANES": str(int(rng.integers(2, 5))),
               "ORBHIP_PACK_THREADS": str(int(rng.integers(1, 6)))}
        prm = (nfeat, 1.2, levels, 20, 7)
        tag = "case %d %dx%d n%d f%d l%d pad%d gap%d %s %s %s" % (case, w, h, n, nfeat, levels, pad, gap, "stereo" if stereo else "mono",
                                                                      "pinned" if pinned else "pageable", env)
        try:
            big = np.zeros((n, h + gap, w + pad), np.uint8)
            if stereo:
                pairs = [synth.stereo_pair("blocks", w, h, 1000 * case + i, (3, 12, 40)) for i in range(n)]
                bigr = np.zeros_like(big)
                for i, (l, r) in enumerate(pairs):
                    big[i, :h, :w] = l; bigr[i, :h, :w] = r
            else:
                for i in range(n):
                    big[i, :h, :w] = synth.frame("blocks" if i % 3 else "lowc", w, h, 1000 * case + i)
            if pinned:
                big = torch.from_numpy(big).pin_memory().numpy()
                if stereo:
                    bigr = torch.from_numpy(bigr).pin_memory().numpy()
            view = big[:, :h, :w]
            cap_of = lambda ex: ex.max_keypoints(w, h)
            if not stereo:
                ref_ex = orbhip.ORBextractor(*prm)
                ref = [ref_ex(np.ascontiguousarray(v), fetch_pyramid=False) for v in view]
                g = make(env, lambda: orbhip.ORBextractor(*prm))
                cap = cap_of(g)
                kp = np.zeros((n, cap), orbhip.KP_DTYPE); desc = np.zeros((n, cap, 32), np.uint8); cnt = np.zeros(n, np.int32)
                rc = orbhip.lib().orbhip_extract_batch(g._h, view.ctypes.data, n, w, h, view.strides[1], view.strides[0],
                                                       kp.ctypes.data, desc.ctypes.data, cnt.ctypes.data, cap)
                assert rc == 0, orbhip.lib().orbhip_last_error(g._h)
                for i in range(n):
                    assert cnt[i] == len(ref[i][0]), "count of frame %d" % i
                    assert np.array_equal(kp[i, :cnt[i]], ref[i][0]) and np.array_equal(desc[i, :cnt[i]], ref[i][1]), "frame %d" % i
                g.close(); ref_ex.close()
            else:
                viewr = bigr[:, :h, :w]
                rl, rr = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
                ref = []
                for l, r in zip(view, viewr):
                    kl, dl = rl(np.ascontiguousarray(l), fetch_pyramid=False)
                    kr, dr = rr(np.ascontiguousarray(r), fetch_pyramid=False)
                    u, d, m = orbhip.ComputeStereoMatches(rl, rr, kl, dl, kr, dr, BF / FX, BF)
                    ref.append((kl, dl, kr, dr, u, d, m))
                gl, gr = make(env, lambda: (orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)))
                cap = cap_of(gl)
                kl, kr = np.zeros((n, cap), orbhip.KP_DTYPE), np.zeros((n, cap), orbhip.KP_DTYPE)
                dl, dr = np.zeros((n, cap, 32), np.uint8), np.zeros((n, cap, 32), np.uint8)
                cl, cr, nm = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
                ur, dep = np.zeros((n, cap), np.float32), np.zeros((n, cap), np.float32)
                import ctypes as C
                rc = orbhip.lib().orbhip_extract_stereo_batch(gl._h, gr._h, view.ctypes.data, viewr.ctypes.data, n, w, h, view.strides[1],
                                                              view.strides[0], kl.ctypes.data, dl.ctypes.data, cl.ctypes.data, kr.ctypes.data,
                                                              dr.ctypes.data, cr.ctypes.data, cap, C.c_float(BF / FX), C.c_float(BF),
                                                              ur.ctypes.data, dep.ctypes.data, nm.ctypes.data)
                assert rc == 0, orbhip.lib().orbhip_last_error(gl._h)
                for i in range(n):
                    a = ref[i]
                    assert cl[i] == len(a[0]) and cr[i] == len(a[2]), "counts of pair %d" % i
                    assert np.array_equal(kl[i, :cl[i]], a[0]) and np.array_equal(dl[i, :cl[i]], a[1]), "left of pair %d" % i
                    assert np.array_equal(kr[i, :cr[i]], a[2]) and np.array_equal(dr[i, :cr[i]], a[3]), "right of pair %d" % i
                    assert np.array_equal(ur[i, :cl[i]].view(np.uint32), a[4].view(np.uint32)), "uright of pair %d" % i
                    assert np.array_equal(dep[i, :cl[i]].view(np.uint32), a[5].view(np.uint32)), "depth of pair %d" % i
                    assert nm[i] == a[6], "matches of pair %d" % i
                for x in (gl, gr, rl, rr):
                    x.close()
            print(tag, "ok", flush=True)
        except orbhip.OrbHipError as ex:
            print(tag, "refused:", ex, flush=True)
        except AssertionError as ex:
            bad += 1
            print(tag, "MISMATCH:", ex, flush=True)
    print("fuzz_host: %d cases, %d mismatches" % (args.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

"""Deterministic synthetic 8-bit gray frames for tests and benchmarks.

Three distributions (seed = 0x5EED0000 + frame index, numpy PCG64):
  blocks : mid-gray canvas + W*H/600 random axis-aligned rectangles (side 8..96 px, gray U[0,255])
           + uniform noise in [-4,4]  (many true corners, natural-image-like)
  noise  : IID U[0,255]                (candidate explosion stress for the quadtree)
  lowc   : `blocks` compressed to amplitude <= 15 around 128 (forces the minimum-threshold retry)
  const  : all pixels equal            (zero-keypoint path)
Dataset-shaped frames (TUM 640x480, KITTI 1241x376, EuRoC 752x480) are `blocks` at those sizes;
the datasets themselves are not available offline.
"""
import numpy as np

SEED_BASE = 0x5EED0000


def frame(kind, w, h, index=0):
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + index))
    if kind == "noise":
        return rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    if kind == "const":
        return np.full((h, w), 100, np.uint8)
    if kind not in ("blocks", "lowc"):
        raise ValueError(kind)
    img = np.full((h, w), 128, np.int16)
    n = max(1, (w * h) // 600)
    xs = rng.integers(-48, w, size=n)
    ys = rng.integers(-48, h, size=n)
    ws = rng.integers(8, 97, size=n)
    hs = rng.integers(8, 97, size=n)
    gs = rng.integers(0, 256, size=n)
    for x, y, rw, rh, g in zip(xs, ys, ws, hs, gs):
        img[max(y, 0):max(y + rh, 0), max(x, 0):max(x + rw, 0)] = g
    img += rng.integers(-4, 5, size=(h, w), dtype=np.int16)
    if kind == "lowc":
        img = 128 + (img - 128) * 15 // 128
    return np.clip(img, 0, 255).astype(np.uint8)


def batch(kind, w, h, n, start=0):
    return np.stack([frame(kind, w, h, start + i) for i in range(n)])


def stereo_pair(kind, w, h, index=0, disparities=(3, 12, 40)):
    """A rectified synthetic pair: the right image is the left one with the content of each
    horizontal band moved left by that band's disparity (the vacated columns keep the left pixels)."""
    left = frame(kind, w, h, index)
    right = np.empty_like(left)
    for rows, d in zip(np.array_split(np.arange(h), len(disparities)), disparities):
        r0, r1 = rows[0], rows[-1] + 1
        right[r0:r1, :w - d] = left[r0:r1, d:]
        right[r0:r1, w - d:] = left[r0:r1, w - d:]
    return left, right


def rectify_maps(w, h, angle_deg=1.7, k=0.09, shift=(3.3, -2.6)):
    """Float32 maps (map_x, map_y) of the kind cv::initUndistortRectifyMap produces for a stereo rig:
    a small rotation, a radial term and a shift; parts of the rectified image look outside the source."""
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    cx, cy, f = w / 2.0 - 0.5, h / 2.0 - 0.5, 0.9 * w
    a = np.deg2rad(angle_deg)
    xn, yn = (x - cx) / f, (y - cy) / f
    xr, yr = np.cos(a) * xn - np.sin(a) * yn, np.sin(a) * xn + np.cos(a) * yn
    d = 1.0 + k * (xr * xr + yr * yr)
    return (xr * d * f + cx + shift[0]).astype(np.float32), (yr * d * f + cy + shift[1]).astype(np.float32)

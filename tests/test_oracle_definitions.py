"""The oracle's image primitives against independent numpy formulations of their PUBLISHED DEFINITIONS,
on random images.

The known-answer tests (tests/test_oracle_kat.py) pin single worked examples; these pin the bulk
behaviour from the other side: every function here is written from the definition a paper or the
OpenCV documentation gives (segment test, bilinear interpolation at pixel centres, a normalised
Gaussian), not from the oracle's code and not in its arithmetic, so a misreading shared by the oracle
and the device kernels (which are checked against the oracle, never against OpenCV: neither the
reference nor this repository holds OpenCV vectors) would have to be shared by these as well.
What integer rounding leaves open is stated as a tolerance in each test.
"""
import numpy as np
import pytest

from oracle import orb_oracle as O

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]       # Rosten & Drummond's Bresenham circle of radius 3 (x, y)


def images(seed, n, lo=24, hi=72):
    """Random images of three kinds: noise, smoothed noise (few corners), blocks with noise (edges and corners)."""
    rng = np.random.default_rng(seed)
    for i in range(n):
        w, h = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        kind = i % 3
        if kind == 0:
            img = rng.integers(0, 256, (h, w))
        elif kind == 1:
            a = rng.integers(0, 256, (h + 4, w + 4)).astype(np.float64)
            img = sum(a[dy:dy + h, dx:dx + w] for dy in range(5) for dx in range(5)) / 25.0
        else:
            img = np.full((h, w), 128.0)
            for _ in range(6):
                x0, y0 = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
                img[y0:y0 + int(rng.integers(4, h)), x0:x0 + int(rng.integers(4, w))] = rng.integers(0, 256)
            img = img + rng.integers(-4, 5, (h, w))
        yield np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ---- FAST: the segment test, literally ------------------------------------------------------------------
def segment_test(img, t):
    """Mask of the pixels (3 px inside the image) with 9 contiguous circle pixels all brighter than I + t or
    all darker than I - t (Rosten & Drummond, "Machine learning for high-speed corner detection", N = 9)."""
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    ring = [I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE]
    brighter = [r > c + t for r in ring]
    darker = [r < c - t for r in ring]
    out = np.zeros(c.shape, bool)
    for start in range(16):
        b = np.ones(c.shape, bool)
        d = np.ones(c.shape, bool)
        for k in range(9):
            b &= brighter[(start + k) % 16]
            d &= darker[(start + k) % 16]
        out |= b | d
    return out


def fast9_by_definition(img, threshold):
    """Corners of cv::FAST(img, threshold, nonmaxSuppression=true) from the documentation alone: a corner
    passes the segment test at `threshold`; its score is the largest threshold at which it still passes;
    it is kept if its score is strictly greater than the scores of its 8 neighbours (non-corners score 0).
    Raster order, (x, y, score)."""
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    for t in range(255):                                    # a pixel passes at every threshold up to its score
        m = segment_test(img, t)
        if not m.any():
            break
        score[3:h - 3, 3:w - 3][m] = t
    corner = np.zeros((h, w), bool)
    corner[3:h - 3, 3:w - 3] = segment_test(img, threshold)
    s = np.where(corner, score, 0)
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not corner[y, x]:
                continue
            nb = s[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if s[y, x] > nb.max():
                out.append((x, y, int(s[y, x])))
    return out


@pytest.mark.parametrize("threshold", [7, 20])
def test_fast9_is_the_segment_test_with_strict_nonmax_suppression(threshold):
    n_corners = 0
    for img in images(11 + threshold, 12):
        got = [tuple(int(v) for v in r) for r in O.fast9(img, threshold).tolist()]
        want = fast9_by_definition(img, threshold)
        assert got == want
        n_corners += len(want)
    assert n_corners > 100


# ---- resize: bilinear interpolation at pixel centres ----------------------------------------------------
def resize_by_definition(src, dw, dh):
    """INTER_LINEAR as documented: destination pixel centre (dx + 0.5) * sw / dw - 0.5 in source coordinates,
    the two nearest source pixels per axis weighted by distance, coordinates clamped to the image. Float."""
    sh, sw = src.shape
    S = src.astype(np.float64)
    fx = (np.arange(dw) + 0.5) * sw / dw - 0.5
    fy = (np.arange(dh) + 0.5) * sh / dh - 0.5
    x0 = np.floor(fx).astype(int); ax = fx - x0
    y0 = np.floor(fy).astype(int); ay = fy - y0
    x0c, x1c = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)
    y0c, y1c = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    top = S[y0c][:, x0c] * (1 - ax) + S[y0c][:, x1c] * ax
    bot = S[y1c][:, x0c] * (1 - ax) + S[y1c][:, x1c] * ax
    return top * (1 - ay)[:, None] + bot * ay[:, None]


@pytest.mark.parametrize("factor", [1.2, 1.1, 1.5, 2.0 / 1.0001])
def test_resize_is_bilinear_at_pixel_centres(factor):
    worst = 0.0
    for img in images(5, 9, 40, 120):
        sh, sw = img.shape
        dw, dh = int(round(sw / factor)), int(round(sh / factor))
        got = O.resize_linear_u8(img, dw, dh).astype(np.float64)
        want = resize_by_definition(img, dw, dh)
        # 11-bit weights, two rounding shifts: within one grey level of the real-valued interpolation
        worst = max(worst, float(np.abs(got - want).max()))
    assert worst <= 1.0, worst
    # the mean error: the vertical pass truncates twice before its rounding shift, an eighth of a grey level on average
    img = next(images(6, 2, 100, 120))
    got = O.resize_linear_u8(img, 80, 70).astype(np.float64)
    assert -0.2 < (got - resize_by_definition(img, 80, 70)).mean() < 0.02


# ---- blur: a normalised Gaussian, sigma 2, 7 taps, reflected border -------------------------------------
def blur_by_definition(img):
    k = np.exp(-np.arange(-3, 4) ** 2 / (2 * 2.0 ** 2))
    k /= k.sum()                                            # getGaussianKernel(7, 2)
    P = np.pad(img.astype(np.float64), 3, mode="reflect")   # BORDER_REFLECT_101: the edge pixel is not repeated
    h, w = img.shape
    Pv = sum(k[i] * P[:, i:i + w] for i in range(7))        # along the rows, every row of the padded image
    return sum(k[i] * Pv[i:i + h, :] for i in range(7))     # then along the columns


@pytest.mark.parametrize("variant,bias", [(O.BLUR_LEGACY, 257.0 / 256.0), (O.BLUR_FIXEDPOINT, 1.0)])
def test_blur_is_a_gaussian_of_sigma_two(variant, bias):
    """Both variants are 8-bit kernels (each tap up to 0.45 / 256 off the real one, times seven taps and two passes
    of values up to 255): within a grey level and a half of the real-valued Gaussian on noise images, the
    legacy one (taps that sum to 257 / 256) scaled by that sum."""
    worst = 0.0
    for img in images(21, 9, 30, 90):
        got = O.gaussian_blur7(img, variant).astype(np.float64)
        want = np.minimum(blur_by_definition(img) * bias, 255.0)
        worst = max(worst, float(np.abs(got - want).max()))
    assert worst <= 1.5, worst


# ---- orientation: the intensity centroid ---------------------------------------------------------------
def test_ic_angle_is_the_intensity_centroid_direction():
    """Rublee et al., "ORB: an efficient alternative to SIFT or SURF": theta = atan2(m01, m10) over a circular
    patch of radius 15 (the reference's patch: |u| <= umax[|v|]); fastAtan2 is accurate to 0.3 degrees."""
    umax = O.OracleExtractor(500, 1.2, 4, 20, 7).umax()
    for img in images(33, 9, 48, 80):
        h, w = img.shape
        x, y = w // 2, h // 2
        m10 = m01 = 0.0
        for v in range(-15, 16):
            for u in range(-int(umax[abs(v)]), int(umax[abs(v)]) + 1):
                m10 += u * float(img[y + v, x + u])
                m01 += v * float(img[y + v, x + u])
        if abs(m10) + abs(m01) < 1:
            continue
        want = np.degrees(np.arctan2(m01, m10)) % 360.0
        got = O.ic_angle(img, x, y)
        d = abs(got - want)
        assert min(d, 360.0 - d) <= 0.31, (got, want)


# ---- descriptor: steered BRIEF -------------------------------------------------------------------------
def test_descriptor_is_steered_brief_of_the_pattern():
    """Bit i of the descriptor compares the image at the i-th pattern pair rotated by the keypoint angle
    (rounded to the nearest pixel): tau(p; x, y) = 1 iff p(x) < p(y) (Calonder et al., BRIEF; Rublee et al.)."""
    import pathlib
    import re
    txt = (pathlib.Path(__file__).resolve().parent.parent / "include" / "orb_pattern_31.inc").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//.*", "", txt)
    pat = np.array([int(v) for v in re.findall(r"-?\d+", txt)], np.int64).reshape(256, 4)
    rng = np.random.default_rng(3)
    flips = total = 0
    for img in images(44, 6, 60, 90):
        h, w = img.shape
        x, y = w // 2, h // 2
        for angle in rng.uniform(0, 360, 8):
            a, b = np.cos(np.radians(angle)), np.sin(np.radians(angle))
            want = np.zeros(32, np.uint8)
            for i in range(256):
                x0, y0, x1, y1 = pat[i]
                v0 = img[y + int(np.rint(x0 * b + y0 * a)), x + int(np.rint(x0 * a - y0 * b))]
                v1 = img[y + int(np.rint(x1 * b + y1 * a)), x + int(np.rint(x1 * a - y1 * b))]
                if v0 < v1:
                    want[i // 8] |= 1 << (i % 8)
            got = O.descriptor(img, x, y, float(np.float32(angle)))
            flips += int(np.unpackbits(got ^ want).sum())
            total += 256
    # float32 against float64 rotation: a rounded coordinate may differ for a handful of points
    assert flips <= total * 0.002, (flips, total)


# ---- gray conversion ------------------------------------------------------------------------------------
def test_gray_is_the_documented_luma():
    """cvtColor RGB/BGR -> GRAY: Y = 0.299 R + 0.587 G + 0.114 B (OpenCV documentation), rounded; fixed-point
    weights leave at most one grey level."""
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for rgb in (False, True):
        r, g, b = (img[:, :, 0], img[:, :, 1], img[:, :, 2]) if rgb else (img[:, :, 2], img[:, :, 1], img[:, :, 0])
        want = 0.299 * r + 0.587 * g + 0.114 * b
        got = O.cvt_gray(img, rgb).astype(np.float64)
        assert np.abs(got - want).max() <= 1.0
        assert abs((got - want).mean()) < 0.05

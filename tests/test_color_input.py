"""Colour input: the cvtColor to gray that precedes extraction in the reference
(Tracking::GrabImage*, src/Tracking.cc:166-191) fused into the level-0 kernel.

Known answers of the 8-bit conversion (OpenCV 2.4 .. 3.x: (B*1868 + G*9617 + R*4899 + 8192) >> 14):
pure red 76, pure green 150, pure blue 29, white 255, and gray stays gray because the weights sum to 2^14.
"""
import numpy as np
import pytest

from visual_slam_amd import synth


def color_image(w, h, index, cn):
    """Three decorrelated planes (different synthetic frames) interleaved; alpha, if any, is noise."""
    planes = [synth.frame("blocks", w, h, index + 7 * k) for k in range(3)]
    if cn == 4:
        planes.append(synth.frame("noise", w, h, index + 99))
    return np.ascontiguousarray(np.stack(planes, axis=-1))


def test_oracle_cvt_gray_known_answers():
    from oracle import orb_oracle as O
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [17, 17, 17], [200, 200, 200]]], np.uint8)
    assert O.cvt_gray(px, rgb=True)[0].tolist() == [76, 150, 29, 255, 0, 17, 200]
    assert O.cvt_gray(px, rgb=False)[0].tolist() == [29, 150, 76, 255, 0, 17, 200]
    # 4 channels: alpha is ignored
    px4 = np.concatenate([px, np.full((1, 7, 1), 123, np.uint8)], axis=-1)
    assert np.array_equal(O.cvt_gray(px4, rgb=True), O.cvt_gray(px, rgb=True))
    # against the formula in numpy on random pixels
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    assert np.array_equal(O.cvt_gray(img, rgb=True), ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("cn,rgb,w,h", [(3, True, 640, 480), (3, False, 752, 480), (4, True, 333, 257), (4, False, 1241, 376)])
def test_hip_color_matches_oracle(cn, rgb, w, h):
    from oracle import orb_oracle as O
    from tests.parity import DESC_MIN_EXACT, compare_outputs
    from visual_slam_amd import orbhip
    img = color_image(w, h, 40, cn)
    gray = O.cvt_gray(img, rgb)
    prm = (1000, 1.2, 8, 20, 7)
    o = O.OracleExtractor(*prm)
    okp, odesc = o.extract(gray)
    ex = orbhip.ORBextractor(*prm)
    kp, desc = ex.extract_color(img, rgb)
    # level 0 of the pyramid is the converted image, border included
    assert np.array_equal(ex.pyramid_level(0, with_border=True), o.level(0, with_border=True))
    r = compare_outputs(kp, desc, okp, odesc)
    assert r["n"] > 300 and r["desc_exact"] >= DESC_MIN_EXACT, r
    # and the gray entry point on the converted image gives the same result
    kp2, desc2 = ex(gray)
    assert np.array_equal(kp2, kp) and np.array_equal(desc2, desc)


@pytest.mark.gpu
def test_hip_color_rejects_bad_channel_count():
    from visual_slam_amd import orbhip
    ex = orbhip.ORBextractor(500, 1.2, 8, 20, 7)
    with pytest.raises(orbhip.OrbHipError):
        ex.extract_color(np.zeros((100, 100, 2), np.uint8), True)

"""Shared comparison helpers for the GPU parity tests (HIP path vs CPU oracle)."""
import numpy as np

ANGLE_TOL_DEG = 1e-3          # north-star tolerance for the intensity-centroid angle
DESC_MIN_EXACT = 0.999        # at least 99.9 % of descriptors bit-exact


def compare_outputs(gkp, gdesc, okp, odesc):
    """Returns a dict of parity figures; raises AssertionError on a hard mismatch."""
    assert len(gkp) == len(okp), "keypoint count %d vs oracle %d" % (len(gkp), len(okp))
    n = len(gkp)
    if n == 0:
        return dict(n=0, angle_max=0.0, desc_exact=1.0, desc_bits=0)
    for fld in ("x", "y", "size", "response", "octave", "class_id"):
        bad = np.nonzero(gkp[fld] != okp[fld])[0]
        assert bad.size == 0, "field %s differs at %s: %s vs %s" % (fld, bad[:5], gkp[fld][bad[:5]], okp[fld][bad[:5]])
    da = np.abs(gkp["angle"].astype(np.float64) - okp["angle"].astype(np.float64))
    da = np.minimum(da, 360.0 - da)
    assert da.max() <= ANGLE_TOL_DEG, "angle differs by %g deg" % da.max()
    same = np.all(gdesc == odesc, axis=1)
    bits = int(np.unpackbits(np.bitwise_xor(gdesc, odesc), axis=1).sum())
    return dict(n=n, angle_max=float(da.max()), angle_exact=float(np.mean(gkp["angle"] == okp["angle"])),
                desc_exact=float(same.mean()), desc_bits=bits, desc_bad=np.nonzero(~same)[0])

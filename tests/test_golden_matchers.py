"""Regression fixtures of the matcher functions (tests/golden/matchers.json, made by tools/gen_golden_matchers.py
with the oracle: nothing in the reference provides vectors for them, so the fixture guards against accidental
change and does not pin anything to the reference). The oracle must reproduce it on the CPU, the device
through the C ABI on the GPU."""
import importlib.util
import json
import pathlib

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("gen_golden_matchers", ROOT / "tools" / "gen_golden_matchers.py")
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
GOLD = json.loads((ROOT / "tests" / "golden" / "matchers.json").read_text())


def test_oracle_reproduces_the_matcher_fixture():
    from oracle import orb_oracle as O
    got = gen.run_all(gen.scenario(O), O)
    assert got == GOLD
    assert all(v[0] > 50 for v in GOLD.values())                       # every function matched something


class DeviceAdapter:
    """The device entry points under the oracle's function names and argument order."""

    def __init__(self, ex):
        from visual_slam_amd import orbhip
        self.ex, self.h = ex, orbhip

    def search_for_initialization(self, k1, d1, k2, d2, bounds, prev, win, ratio, ori):
        return self.h.SearchForInitialization(self.ex, k1, d1, k2, d2, bounds, prev, win, ratio, ori)

    def search_by_projection(self, k1, world, flags, d1, k2, d2, ur, Tcw, cam4, mbf, bounds, sf, th, mode, ori):
        return self.h.SearchByProjection(self.ex, k1, world, flags, d1, k2, d2, ur, Tcw, cam4, mbf, bounds, th, mode, ori)

    def search_by_projection_keyframe(self, k1, world, rng_, flags, d1, k2, d2, occ, Tcw, cam4, bounds, sf, scale, th, odist, ori):
        return self.h.SearchByProjectionKeyFrame(self.ex, k1, world, rng_, flags, d1, k2, d2, occ, Tcw, cam4, bounds, th, odist, ori)

    def search_by_projection_sim3(self, world, normal, rng_, flags, d1, k2, d2, occ, Scw, cam4, bounds, sf, scale, th):
        return self.h.SearchByProjectionSim3(self.ex, world, normal, rng_, flags, d1, k2, d2, occ, Scw, cam4, bounds, th)

    def search_map_points(self, mp, dmp, k2, d2, ur, occ, bounds, sf, th, ratio):
        return self.h.SearchLocalPoints(self.ex, mp, dmp, k2, d2, ur, occ, bounds, th, ratio)

    def is_in_frustum(self, mpw, pose, cam4, mbf, bounds, scale, levels, lim):
        return self.h.IsInFrustum(self.ex, mpw, pose, cam4, mbf, bounds, lim)

    def search_by_bow(self, k1, d1, n1, f1, k2, d2, n2, ratio, ori):
        return self.h.SearchByBoW(self.ex, k1, d1, n1, f1, k2, d2, n2, ratio, ori)

    def search_by_bow_keyframes(self, k1, d1, n1, f1, k2, d2, n2, f2, ratio, ori):
        return self.h.SearchByBoWKeyFrames(self.ex, k1, d1, n1, f1, k2, d2, n2, f2, ratio, ori)


@pytest.mark.gpu
def test_device_reproduces_the_matcher_fixture():
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    ex = orbhip.ORBextractor(*gen.PRM)
    got = gen.run_all(gen.scenario(O), DeviceAdapter(ex))   # the scenario (extraction of the two views) comes from the oracle
    assert got == GOLD

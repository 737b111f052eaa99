"""The single-frame host call replayed from a captured HIP graph.

The library captures the copy up, the kernel launches and the copies down of ORBextractor::operator()
(src/ORBextractor.cc:1043-1105, one image per call, the way a SLAM front end drives it) the second time a
configuration is seen and replays the graph from then on. These tests call one handle many times and check
every call against the oracle, across the events that must retire a captured graph: another image size,
another output capacity, rectification maps set and cleared, colour input, per-kernel timing switched on,
and the bordered pyramid asked for after a replayed call. ORBHIP_GRAPH=0 is the plain path for comparison.

The oracle is pinned by analytic known answers only (the reference holds no vectors): parity unpinned.
"""
import os

import numpy as np
import pytest

from tests.parity import DESC_MIN_EXACT, compare_outputs
from visual_slam_amd import synth

PRM = (800, 1.2, 8, 20, 7)


def _check(g, o, img, levels=(0, 2)):
    okp, odesc = o.extract(img)
    gkp, gdesc = g(img, fetch_pyramid=False)
    r = compare_outputs(gkp, gdesc, okp, odesc)
    assert r["desc_exact"] >= DESC_MIN_EXACT, r
    for l in levels:                                             # level 0 is rebuilt from the staged image on request
        assert np.array_equal(g.pyramid_level(l, with_border=True), o.level(l, with_border=True)), l


@pytest.mark.gpu
def test_repeated_single_frame_calls_follow_the_oracle():
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    g, o = orbhip.ORBextractor(*PRM), O.OracleExtractor(*PRM)
    for i in range(5):                                           # plain, capture, replay, replay, replay
        _check(g, o, synth.frame("blocks" if i % 2 else "noise", 640, 480, 40 + i))
    for i in range(3):                                           # another size: new plan, new graph
        _check(g, o, synth.frame("blocks", 752, 480, 50 + i))
    for i in range(3):                                           # back again
        _check(g, o, synth.frame("lowc", 640, 480, 60 + i))
    g.enable_timing(True)                                        # timed calls run the plain way
    _check(g, o, synth.frame("blocks", 640, 480, 70))
    assert sum(ms for ms, _ in g.timing().values()) > 0
    g.enable_timing(False)
    for i in range(2):
        _check(g, o, synth.frame("blocks", 640, 480, 71 + i))
    # a constant frame between textured ones: no keypoints, then keypoints again
    kp, desc = g(np.full((480, 640), 90, np.uint8), fetch_pyramid=False)
    assert len(kp) == 0
    _check(g, o, synth.frame("blocks", 640, 480, 75))


@pytest.mark.gpu
def test_graph_and_plain_path_agree_with_maps_and_colour():
    from visual_slam_amd import orbhip
    w, h = 640, 480
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    map_x = (xx + 1.75 + 0.002 * yy).astype(np.float32)
    map_y = (yy - 0.5 + 0.001 * xx).astype(np.float32)
    frames = [synth.frame("blocks", w, h, 80 + i) for i in range(4)]
    colour = [np.stack([f, np.roll(f, 3, 1), np.roll(f, 5, 0)], axis=2) for f in frames]
    out = {}
    for graph in ("1", "0"):
        old = os.environ.get("ORBHIP_GRAPH")
        os.environ["ORBHIP_GRAPH"] = graph
        try:
            g = orbhip.ORBextractor(*PRM)
        finally:
            if old is None:
                del os.environ["ORBHIP_GRAPH"]
            else:
                os.environ["ORBHIP_GRAPH"] = old
        res = []
        for f in frames[:3]:
            res.append(g(f, fetch_pyramid=False))
        g.set_rectification(map_x, map_y)                        # the graph of the unrectified calls must not be replayed
        for f in frames:
            res.append(g(f, fetch_pyramid=False))
        g.set_rectification(None, None)
        for f in frames[:3]:
            res.append(g(f, fetch_pyramid=False))
        for c in colour:                                         # three channels: another staging layout
            res.append(g.extract_color(c, rgb=False, fetch_pyramid=False))
        for f in frames[:2]:
            res.append(g(f, fetch_pyramid=False))
        out[graph] = res
    assert len(out["1"]) == len(out["0"])
    for i, ((ka, da), (kb, db)) in enumerate(zip(out["1"], out["0"])):
        assert np.array_equal(ka, kb) and np.array_equal(da, db), i
    # rectified and unrectified results differ, or the maps were not in effect
    assert not np.array_equal(out["1"][0][0], out["1"][3][0])

"""ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (reference src/ORBmatcher.cc:156-285): oracle properties and
GPU parity. The vocabulary transform is not part of the path: node ids are synthetic (a hash of a few
descriptor bits, so that the two views of a feature mostly share a node, as a vocabulary would arrange)."""
import numpy as np
import pytest

from visual_slam_amd import synth


def shifted(img, dx, dy):
    out = img.copy()
    h, w = img.shape
    out[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)] = img[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)]
    return out


def nodes_of(desc, nnodes, rng, missing=0.02):
    # a feature's node from 12 of its descriptor bits: equal descriptors share a node, close ones often do
    v = (desc[:, 0].astype(np.uint32) | (desc[:, 7].astype(np.uint32) << 8)) & 0xFFF
    node = (v * 2654435761 % nnodes).astype(np.uint32) * 37 + 1000        # sparse ids, like word ids of an inner level
    node[rng.random(len(desc)) < missing] = 0xFFFFFFFF
    return node


def scenario(w, h, index, dx, dy, nnodes=100, seed=5, prm=(2000, 1.2, 8, 20, 7)):
    from oracle import orb_oracle as O
    a = synth.frame("blocks", w, h, index)
    e = O.OracleExtractor(*prm)
    k1, d1 = e.extract(a)
    k2, d2 = e.extract(shifted(a, dx, dy))
    rng = np.random.default_rng(seed)
    flags = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    return dict(k1=k1, d1=d1, n1=nodes_of(d1, nnodes, rng), flags=flags, k2=k2, d2=d2, n2=nodes_of(d2, nnodes, rng), dx=dx, dy=dy)


def run_oracle(s, ratio=0.7, ori=True):
    from oracle import orb_oracle as O
    return O.search_by_bow(s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], ratio, ori)


def test_oracle_bow_properties():
    s = scenario(640, 480, 95, 6, -3)
    m, n = run_oracle(s, 0.7, True)
    got = m >= 0
    assert n == got.sum() and n > 200
    assert np.all(s["flags"][m[got]] & 1)                                 # only features with a map point are matched
    assert np.all(s["n1"][m[got]] == s["n2"][got])                        # inside a common node
    assert np.all(s["n2"][got] != 0xFFFFFFFF)
    assert len(np.unique(m[got])) == got.sum()                             # a key-frame feature is used once
    # the matched pairs are the same scene points: positions differ by the shift
    ex = s["k2"]["x"][got] - s["k1"]["x"][m[got]] - s["dx"] * 1.0
    ey = s["k2"]["y"][got] - s["k1"]["y"][m[got]] - s["dy"] * 1.0
    assert np.mean((np.abs(ex) < 3) & (np.abs(ey) < 3)) > 0.9
    # the orientation test only removes matches; a looser ratio only adds candidates for acceptance
    m0, n0 = run_oracle(s, 0.7, False)
    assert n0 >= n
    m9, n9 = run_oracle(s, 0.9, False)
    assert n9 >= n0
    # one node for everything: brute force over the whole frame, still one-to-one
    one = dict(s, n1=np.zeros_like(s["n1"]), n2=np.zeros_like(s["n2"]))
    mo, no = run_oracle(one, 0.7, True)
    assert no > 0 and len(np.unique(mo[mo >= 0])) == (mo >= 0).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,dx,dy,nnodes,ratio,ori,nfeat", [
    (1241, 376, 6, 2, 100, 0.7, True, 2000),
    (640, 480, -7, 5, 1000, 0.75, True, 1000),
    (752, 480, 4, -3, 10, 0.7, False, 1500),
    (640, 480, 2, 2, 1, 0.9, True, 3000),
])
def test_hip_search_by_bow_matches_oracle(w, h, dx, dy, nnodes, ratio, ori, nfeat):
    from visual_slam_amd import orbhip
    s = scenario(w, h, 96, dx, dy, nnodes=nnodes, prm=(nfeat, 1.2, 8, 20, 7))
    om, on = run_oracle(s, ratio, ori)
    ex = orbhip.ORBextractor(nfeat, 1.2, 8, 20, 7)
    gm, gn = orbhip.SearchByBoW(ex, s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], ratio, ori)
    assert on > 50 and gn == on
    assert np.array_equal(gm, om)


@pytest.mark.gpu
def test_hip_search_by_bow_edge_cases():
    from visual_slam_amd import orbhip
    s = scenario(640, 480, 97, 3, 1, nnodes=50, prm=(1000, 1.2, 8, 20, 7))
    ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7)
    # identical frames: every feature's twin sits in its node
    same = dict(s, k2=s["k1"], d2=s["d1"], n2=s["n1"])
    om, on = run_oracle(same, 0.7, True)
    gm, gn = orbhip.SearchByBoW(ex, same["k1"], same["d1"], same["n1"], same["flags"], same["k2"], same["d2"], same["n2"], 0.7, True)
    assert gn == on and np.array_equal(gm, om) and on > 300
    # no common node, no map points, empty sides
    apart = dict(s, n2=s["n2"] + 1)
    gm, gn = orbhip.SearchByBoW(ex, apart["k1"], apart["d1"], apart["n1"], apart["flags"], apart["k2"], apart["d2"], apart["n2"])
    assert gn == 0 and np.all(gm == -1) and run_oracle(apart)[1] == 0
    gm, gn = orbhip.SearchByBoW(ex, s["k1"], s["d1"], s["n1"], np.zeros_like(s["flags"]), s["k2"], s["d2"], s["n2"])
    assert gn == 0 and np.all(gm == -1)
    gm, gn = orbhip.SearchByBoW(ex, s["k1"][:0], s["d1"][:0], s["n1"][:0], s["flags"][:0], s["k2"], s["d2"], s["n2"])
    assert gn == 0 and np.all(gm == -1)
    # sizes around the padding of the on-chip sort
    for n in (63, 64, 65, 1023, 1025):
        cut = dict(s, k1=s["k1"][:n], d1=s["d1"][:n], n1=s["n1"][:n], flags=s["flags"][:n])
        om, on = run_oracle(cut, 0.8, True)
        gm, gn = orbhip.SearchByBoW(ex, cut["k1"], cut["d1"], cut["n1"], cut["flags"], cut["k2"], cut["d2"], cut["n2"], 0.8, True)
        assert gn == on and np.array_equal(gm, om), n


@pytest.mark.gpu
def test_hip_search_by_bow_large_frames_in_global_memory():
    """Frames with more than 4096 features (the 8000-feature configuration): the sort arrays move to global memory;
    both forms of the function, results identical to the oracle's."""
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    prm = (8000, 1.2, 8, 20, 7)
    s = scenario(1920, 1080, 99, 5, -2, nnodes=400, prm=prm)
    assert len(s["k1"]) > 4096 and len(s["k2"]) > 4096
    ex = orbhip.ORBextractor(*prm)
    om, on = run_oracle(s, 0.7, True)
    gm, gn = orbhip.SearchByBoW(ex, s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], 0.7, True)
    assert on > 1000 and gn == on and np.array_equal(gm, om)
    f2 = (np.random.default_rng(3).random(len(s["k2"])) < 0.7).astype(np.uint8)
    om, on = O.search_by_bow_keyframes(s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], f2, 0.75, True)
    gm, gn = orbhip.SearchByBoWKeyFrames(ex, s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], f2, 0.75, True)
    assert on > 500 and gn == on and np.array_equal(gm, om)


# ---- SearchByBoW(pKF1, pKF2, vpMatches12) (reference src/ORBmatcher.cc:519-652)

def test_oracle_bow_keyframes_properties():
    from oracle import orb_oracle as O
    s = scenario(640, 480, 98, 4, 2)
    rng = np.random.default_rng(8)
    f2 = (rng.random(len(s["k2"])) < 0.7).astype(np.uint8)
    m, n = O.search_by_bow_keyframes(s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], f2, 0.75, True)
    got = m >= 0
    assert n == got.sum() and n > 100
    assert np.all(s["flags"][got] & 1) and np.all(f2[m[got]] & 1)          # both features carry a good map point
    assert np.all(s["n1"][got] == s["n2"][m[got]])
    assert len(np.unique(m[got])) == got.sum()
    # against the frame variant with all of side 2 eligible: the same pairs except for the strict TH_LOW test
    allf = np.ones(len(s["k2"]), np.uint8)
    m12, n12 = O.search_by_bow_keyframes(s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], allf, 0.7, False)
    mf, nf = O.search_by_bow(s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], 0.7, False)
    inv = np.full(len(s["k1"]), -1, np.int32)
    inv[mf[mf >= 0]] = np.nonzero(mf >= 0)[0]
    differ = np.nonzero(inv != m12)[0]
    assert n12 <= nf and len(differ) <= nf - n12 + 5


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,nnodes,ratio,ori", [(1241, 376, 100, 0.75, True), (640, 480, 5, 0.9, False), (752, 480, 1000, 0.6, True)])
def test_hip_search_by_bow_keyframes_matches_oracle(w, h, nnodes, ratio, ori):
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    s = scenario(w, h, 99, 5, -2, nnodes=nnodes, prm=(1500, 1.2, 8, 20, 7))
    rng = np.random.default_rng(9)
    f2 = (rng.random(len(s["k2"])) < 0.75).astype(np.uint8)
    om, on = O.search_by_bow_keyframes(s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], f2, ratio, ori)
    ex = orbhip.ORBextractor(1500, 1.2, 8, 20, 7)
    gm, gn = orbhip.SearchByBoWKeyFrames(ex, s["k1"], s["d1"], s["n1"], s["flags"], s["k2"], s["d2"], s["n2"], f2, ratio, ori)
    assert on > 30 and gn == on and np.array_equal(gm, om)
    # identical key frames with every point good; one side empty
    ones = np.ones(len(s["k1"]), np.uint8)
    om, on = O.search_by_bow_keyframes(s["k1"], s["d1"], s["n1"], ones, s["k1"], s["d1"], s["n1"], ones, ratio, ori)
    gm, gn = orbhip.SearchByBoWKeyFrames(ex, s["k1"], s["d1"], s["n1"], ones, s["k1"], s["d1"], s["n1"], ones, ratio, ori)
    assert gn == on and np.array_equal(gm, om)
    gm, gn = orbhip.SearchByBoWKeyFrames(ex, s["k1"], s["d1"], s["n1"], ones, s["k2"][:0], s["d2"][:0], s["n2"][:0], f2[:0], ratio, ori)
    assert gn == 0 and np.all(gm == -1) and len(gm) == len(s["k1"])

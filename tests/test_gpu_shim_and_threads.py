"""GPU tests of the drop-in boundary: the C++ ORB_SLAM2::ORBextractor class over the C ABI, and the
reference's threading contract (two extractor instances on two threads for a stereo pair,
src/Frame.cc:114-117)."""
import subprocess
import threading
from pathlib import Path

import numpy as np
import pytest

from visual_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_cpp_class_shim_matches_oracle():
    exe = ROOT / "shim" / "shim_test"
    assert exe.exists(), "shim/shim_test is built by __graft_entry__.build() (make -C shim)"
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim ok" in out.stdout


def test_two_instances_on_two_threads_like_stereo():
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    left, right = synth.frame("blocks", 1241, 376, 21), synth.frame("blocks", 1241, 376, 22)
    exl, exr = orbhip.ORBextractor(2000, 1.2, 8, 20, 7), orbhip.ORBextractor(2000, 1.2, 8, 20, 7)
    res, errors = {}, []

    def work(name, ex, img):
        try:
            for _ in range(8):                                   # (the second call captures a graph, later ones replay it)
                res[name] = ex(img)
        except Exception as exc:                                 # an exception in a thread would otherwise pass unseen
            errors.append((name, repr(exc)))

    th = [threading.Thread(target=work, args=("l", exl, left)), threading.Thread(target=work, args=("r", exr, right))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    o = O.OracleExtractor(2000, 1.2, 8, 20, 7)
    for name, img, ex in (("l", left, exl), ("r", right, exr)):
        okp, odesc = o.extract(img)
        kp, desc = res[name]
        assert len(kp) == len(okp)
        for fld in ("x", "y", "octave", "response", "size"):
            assert np.array_equal(kp[fld], okp[fld]), (name, fld)
        assert np.mean(np.all(desc == odesc, axis=1)) >= 0.999
        for l in range(8):   # each instance keeps its own pyramid (ComputeStereoMatches reads both)
            assert np.array_equal(ex.mvImagePyramid[l], o.level(l)), (name, l)


@pytest.mark.gpu
def test_two_passes_in_flight_give_the_same_results():
    """ORBHIP_LANES=2 splits a large batch into passes on two internal streams, ORBHIP_OVERLAP=1 (the default
    for large passes; ORBHIP_OVERLAP=2 forces it for any size) runs FAST on a second stream beside the resize chain,
    the first ORBHIP_FAST_SPLIT levels as launches of their own (all read at handle creation, so they run in child
    processes); the outputs must equal the single-stream ones."""
    import os
    import subprocess
    import sys
    import tempfile
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from visual_slam_amd import orbhip, synth\n"
        "imgs = synth.batch('blocks', 320, 240, 80)\n"
        "ex = orbhip.ORBextractor(300, 1.2, 6, 20, 7, max_batch=64)\n"
        "res = ex.extract_batch(imgs)\n"
        "np.savez(sys.argv[1], n=np.array([len(k) for k, d in res]), kp=np.concatenate([k for k, d in res]),\n"
        "         desc=np.concatenate([d for k, d in res]))\n"
    ) % str(root)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for lanes, overlap, split in (("1", "0", "2"), ("2", "0", "2"), ("1", "2", "2"), ("2", "2", "2"), ("1", "2", "1"), ("1", "2", "4"),
                                      ("2", "2", "16")):
            path = os.path.join(tmp, "lanes%s_%s_%s.npz" % (lanes, overlap, split))
            # (the host batch call in one piece, so that it runs as passes of 64 and 16 frames through the device entry point)
            env = dict(os.environ, ORBHIP_LANES=lanes, ORBHIP_OVERLAP=overlap, ORBHIP_FAST_SPLIT=split, ORBHIP_HOST_PIPELINE="0")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=240)
            with np.load(path) as z:
                out[lanes + overlap + "/" + split] = {k: z[k] for k in z.files}
    assert out["10/2"]["n"].sum() > 80 * 100
    for mode in out:
        for k in ("n", "kp", "desc"):
            assert np.array_equal(out["10/2"][k], out[mode][k]), (mode, k)

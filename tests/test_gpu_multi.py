"""GPU tests of the host batch pipeline, the multi-handle entry points and the sharded bench launch.

The host batch call (Frame::ExtractORB, reference src/Frame.cc:272-278, for many images) cuts a batch into chunks
that overlap packing, copy up, extraction and copy down; the multi-handle calls shard a batch over several
handles with one host thread each (the reference's two extractor threads, src/Frame.cc:114-117). Both must return
exactly what frame-by-frame calls on one handle return.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from visual_slam_amd import synth

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
PRM = (600, 1.2, 6, 20, 7)
FX, BF = 718.856, 386.1448


def _with_env(env, make):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return make()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _frames(n, w=320, h=240):
    kinds = ["blocks", "noise", "lowc", "const", "blocks"]
    return np.stack([synth.frame(kinds[i % len(kinds)], w, h, 200 + i) for i in range(n)])


def _singles(orbhip, imgs, prm=PRM):
    single = orbhip.ORBextractor(*prm)
    return [single(im, fetch_pyramid=False) for im in imgs]


def _same(res, ref):
    assert len(res) == len(ref)
    for i, ((ka, da), (kb, db)) in enumerate(zip(res, ref)):
        assert np.array_equal(ka, kb) and np.array_equal(da, db), "frame %d" % i


def test_pipelined_host_batch_spans_many_chunks():
    """41 frames in chunks of 5: nine chunks over three input buffers, the last one short."""
    from visual_slam_amd import orbhip
    imgs = _frames(41)
    ref = _singles(orbhip, imgs)
    g = _with_env({"ORBHIP_HOST_CHUNK": "5", "ORBHIP_PACK_THREADS": "3"}, lambda: orbhip.ORBextractor(*PRM))
    _same(g.extract_batch(imgs), ref)
    _same(g.extract_batch(imgs[:17]), ref[:17])                 # same handle again, fewer frames
    # the pyramid of the last chunk (one frame) is still there
    from oracle import orb_oracle as O
    o = O.OracleExtractor(*PRM)
    o.extract(imgs[15])
    assert np.array_equal(g.pyramid_level(2, frame=0), o.level(2))
    # the unpipelined route gives the same
    g0 = _with_env({"ORBHIP_HOST_PIPELINE": "0"}, lambda: orbhip.ORBextractor(*PRM))
    _same(g0.extract_batch(imgs), ref)


def test_pipelined_host_batch_default_chunks_and_padded_rows():
    """Default chunking (a quarter of the batch each) on images whose rows are not 16-byte multiples, taken from a
    wider buffer (row stride larger than the width)."""
    from visual_slam_amd import orbhip
    big = np.stack([synth.frame("blocks", 400, 260, 300 + i) for i in range(12)])
    roi = big[:, 7:247, 33:366]                                 # 333 x 240 views, row stride 400
    ref = _singles(orbhip, [np.ascontiguousarray(r) for r in roi])
    g = orbhip.ORBextractor(*PRM)
    cap = g.max_keypoints(333, 240)
    kp = np.zeros((12, cap), orbhip.KP_DTYPE); desc = np.zeros((12, cap, 32), np.uint8); cnt = np.zeros(12, np.int32)
    rc = orbhip.lib().orbhip_extract_batch(g._h, roi.ctypes.data, 12, 333, 240, roi.strides[1], roi.strides[0],
                                           kp.ctypes.data, desc.ctypes.data, cnt.ctypes.data, cap)
    assert rc == 0, orbhip.lib().orbhip_last_error(g._h)
    _same([(kp[i, :cnt[i]], desc[i, :cnt[i]]) for i in range(12)], ref)


def test_pipelined_host_batch_from_pinned_memory():
    """Images in page-locked memory are copied from where they lie (no packing step)."""
    import torch
    from visual_slam_amd import orbhip
    imgs = _frames(24)
    ref = _singles(orbhip, imgs)
    pinned = torch.from_numpy(imgs).pin_memory()
    g = _with_env({"ORBHIP_HOST_CHUNK": "4"}, lambda: orbhip.ORBextractor(*PRM))
    _same(g.extract_batch(pinned.numpy()), ref)


def test_pipelined_host_batch_from_pinned_views():
    """Page-locked images that are views into a wider buffer (row stride larger than the width), with and without a gap
    between the frames: packed by the host threads like pageable ones (only the staging layout is copied from where it
    lies); the trace switch that prints the timeline is on."""
    import torch
    from visual_slam_amd import orbhip
    big = torch.from_numpy(np.stack([synth.frame("blocks", 400, 260, 500 + i) for i in range(12)])).pin_memory().numpy()
    g = _with_env({"ORBHIP_HOST_TRACE": "1"}, lambda: orbhip.ORBextractor(*PRM))
    for y0, y1 in ((0, 260), (9, 249)):                         # frames one behind the other / with rows between them
        roi = big[:, y0:y1, 31:364]                             # 333 columns, row stride 400
        h = y1 - y0
        ref = _singles(orbhip, [np.ascontiguousarray(r) for r in roi])
        cap = g.max_keypoints(333, h)
        kp = np.zeros((12, cap), orbhip.KP_DTYPE); desc = np.zeros((12, cap, 32), np.uint8); cnt = np.zeros(12, np.int32)
        rc = orbhip.lib().orbhip_extract_batch(g._h, roi.ctypes.data, 12, 333, h, roi.strides[1], roi.strides[0],
                                               kp.ctypes.data, desc.ctypes.data, cnt.ctypes.data, cap)
        assert rc == 0, orbhip.lib().orbhip_last_error(g._h)
        _same([(kp[i, :cnt[i]], desc[i, :cnt[i]]) for i in range(12)], ref)


def test_capacity_error_is_reported_by_the_pipeline():
    from visual_slam_amd import orbhip
    imgs = _frames(10)
    g = orbhip.ORBextractor(*PRM)
    cap = 50                                                    # far below what a frame yields
    kp = np.zeros((10, cap), orbhip.KP_DTYPE); desc = np.zeros((10, cap, 32), np.uint8); cnt = np.zeros(10, np.int32)
    rc = orbhip.lib().orbhip_extract_batch(g._h, imgs.ctypes.data, 10, 320, 240, 320, 320 * 240,
                                           kp.ctypes.data, desc.ctypes.data, cnt.ctypes.data, cap)
    assert rc == orbhip.ERR_CAPACITY
    _same(g.extract_batch(imgs), _singles(orbhip, imgs))        # the handle works on


@pytest.mark.parametrize("nhandles", [2, 4])
def test_multi_handles_equal_one_handle(nhandles):
    """The frames of a batch sharded over several handles (here all on device 0, one host thread each)."""
    from visual_slam_amd import orbhip
    imgs = _frames(11)                                          # uneven shards
    ref = _singles(orbhip, imgs)
    hs = [orbhip.ORBextractor(*PRM) for _ in range(nhandles)]
    _same(orbhip.extract_batch_multi(hs, imgs), ref)
    # shards large enough for the chunk pipeline inside every handle's thread
    imgs = _frames(37)
    _same(orbhip.extract_batch_multi(hs, imgs), _singles(orbhip, imgs))
    # the partition is the one the Python sharding helper describes
    from visual_slam_amd import sharding
    for i in range(nhandles):
        assert orbhip.shard_range(37, nhandles, i) == sharding.shard_range(37, nhandles, i)
    with pytest.raises(orbhip.OrbHipError):
        orbhip.extract_batch_multi([hs[0], hs[0]], imgs)        # one handle cannot serve two threads


def test_current_device_is_left_alone():
    import ctypes as C
    import torch
    from visual_slam_amd import orbhip
    hip = C.CDLL("libamdhip64.so")
    dev = C.c_int(-1)
    g = orbhip.ORBextractor(*PRM, device=0)
    g(_frames(1)[0], fetch_pyramid=False)
    assert hip.hipGetDevice(C.byref(dev)) == 0 and dev.value == torch.cuda.current_device()


def test_stereo_host_batch_and_multi():
    """Pairs in host memory through the batch call (one pair of handles, then two pairs of handles) against the
    per-pair calls."""
    from visual_slam_amd import orbhip
    prm = (800, 1.2, 8, 20, 7)
    pairs = [synth.stereo_pair("blocks", 640, 480, 40 + i, (3, 12, 40)) for i in range(5)]
    L = np.stack([p[0] for p in pairs]); R = np.stack([p[1] for p in pairs])
    ref = []
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    for l, r in pairs:
        kl, dl = gL(l, fetch_pyramid=False)
        kr, dr = gR(r, fetch_pyramid=False)
        u, d, n = orbhip.ComputeStereoMatches(gL, gR, kl, dl, kr, dr, BF / FX, BF)
        ref.append((kl, dl, kr, dr, u, d, n))
        assert n > 50

    def check(res):
        assert len(res) == len(ref)
        for i, (a, b) in enumerate(zip(res, ref)):
            for x, y in zip(a[:4], b[:4]):
                assert np.array_equal(x, y), i
            assert np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32)), i
            assert np.array_equal(a[5].view(np.uint32), b[5].view(np.uint32)), i
            assert a[6] == b[6], i

    hL, hR = orbhip.ORBextractor(*prm, max_batch=2), orbhip.ORBextractor(*prm, max_batch=2)   # three passes
    check(orbhip.extract_stereo_batch(hL, hR, L, R, BF / FX, BF))
    lefts = [orbhip.ORBextractor(*prm) for _ in range(2)]
    rights = [orbhip.ORBextractor(*prm) for _ in range(2)]
    check(orbhip.extract_stereo_batch(lefts, rights, L, R, BF / FX, BF))
    with pytest.raises(orbhip.OrbHipError):
        orbhip.extract_stereo_batch(hL, hL, L, R, BF / FX, BF)  # one instance cannot be both sides


def test_stereo_host_batch_pipelined():
    """Thirteen pairs in host memory in chunks of three (five chunks over three input buffers, the last one short),
    from pageable and from page-locked memory, and in one piece, against the per-pair calls; then the pyramids of the
    last chunk are still there for a separate matching call."""
    import torch
    from visual_slam_amd import orbhip
    prm = (600, 1.2, 8, 20, 7)
    pairs = [synth.stereo_pair("blocks", 480, 360, 70 + i, (3, 12, 40)) for i in range(13)]
    L = np.stack([p[0] for p in pairs]); R = np.stack([p[1] for p in pairs])
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    ref = []
    for l, r in pairs:
        kl, dl = gL(l, fetch_pyramid=False)
        kr, dr = gR(r, fetch_pyramid=False)
        u, d, n = orbhip.ComputeStereoMatches(gL, gR, kl, dl, kr, dr, BF / FX, BF)
        ref.append((kl, dl, kr, dr, u, d, n))
    assert sum(r[6] for r in ref) > 13 * 40

    def check(res):
        assert len(res) == len(ref)
        for i, (a, b) in enumerate(zip(res, ref)):
            for x, y in zip(a[:4], b[:4]):
                assert np.array_equal(x, y), i
            assert np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32)), i
            assert np.array_equal(a[5].view(np.uint32), b[5].view(np.uint32)), i
            assert a[6] == b[6], i

    hL, hR = _with_env({"ORBHIP_HOST_CHUNK": "3"}, lambda: (orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)))
    check(orbhip.extract_stereo_batch(hL, hR, L, R, BF / FX, BF))
    Lp, Rp = torch.from_numpy(L).pin_memory().numpy(), torch.from_numpy(R).pin_memory().numpy()
    check(orbhip.extract_stereo_batch(hL, hR, Lp, Rp, BF / FX, BF))
    check(orbhip.extract_stereo_batch(hL, hR, L[:9], R[:9], BF / FX, BF) + ref[9:])   # same handles, fewer pairs
    # the last chunk (pairs 6, 7, 8 of the nine) left its pyramids behind: a separate matching call reads its first pair's
    kl, dl, kr, dr = ref[6][:4]
    u, d, n = orbhip.ComputeStereoMatches(hL, hR, kl, dl, kr, dr, BF / FX, BF)
    assert n == ref[6][6] and np.array_equal(u.view(np.uint32), ref[6][4].view(np.uint32))
    oL, oR = _with_env({"ORBHIP_HOST_PIPELINE": "0"}, lambda: (orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)))
    check(orbhip.extract_stereo_batch(oL, oR, L, R, BF / FX, BF))


def test_graph_replay_restores_its_level0_view():
    """The single-frame call replays a captured graph whose kernels read level 0 from the handle's own staging
    buffer. Other calls on the handle in between (a device batch read in place from the caller's buffer, a colour
    frame) leave another level-0 view in the workspace; the replay must put its own back, or the pyramid and the
    stereo refinement read from the wrong place."""
    import torch
    from oracle import orb_oracle as O
    from visual_slam_amd import orbhip
    prm = (800, 1.2, 8, 20, 7)
    w, h = 640, 480
    left, right = synth.stereo_pair("blocks", w, h, 7, (3, 12, 40))
    other = synth.frame("noise", w, h, 5)
    gL, gR = orbhip.ORBextractor(*prm), orbhip.ORBextractor(*prm)
    for _ in range(3):                                          # plain, capture, replay
        gL(left, fetch_pyramid=False)
        gR(right, fetch_pyramid=False)
    # another image through the device batch entry point of the left handle: read in place from this tensor
    d_other = torch.from_numpy(other).cuda()
    cap = gL.max_keypoints(w, h)
    d_kp = torch.empty((1, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.empty((1, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
    gL.extract_batch_device(d_other.data_ptr(), 1, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap)
    gL.synchronize()
    del d_other                                                 # the caller's buffer goes away
    torch.cuda.empty_cache()
    # and a colour frame through the right handle (level 0 copied into its plane)
    gR.extract_color(np.stack([other] * 3, axis=2), rgb=False, fetch_pyramid=False)
    kl, dl = gL(left, fetch_pyramid=False)                      # replays
    kr, dr = gR(right, fetch_pyramid=False)
    oL, oR = O.OracleExtractor(*prm), O.OracleExtractor(*prm)
    okl, odl = oL.extract(left)
    okr, odr = oR.extract(right)
    from tests.parity import DESC_MIN_EXACT, compare_outputs
    assert compare_outputs(kl, dl, okl, odl)["desc_exact"] >= DESC_MIN_EXACT
    assert compare_outputs(kr, dr, okr, odr)["desc_exact"] >= DESC_MIN_EXACT
    gu, gd, gn = orbhip.ComputeStereoMatches(gL, gR, kl, dl, kr, dr, BF / FX, BF)
    ou, od, on = O.stereo_match(oL, oR, okl, odl, okr, odr, BF / FX, BF)
    assert gn == on and on > 50
    assert np.array_equal(ou.view(np.uint32), gu.view(np.uint32)) and np.array_equal(od.view(np.uint32), gd.view(np.uint32))
    for side, g, o in (("left", gL, oL), ("right", gR, oR)):
        assert np.array_equal(g.pyramid_level(0, with_border=True), o.level(0, with_border=True)), side


def test_bench_two_ranks_on_one_device():
    """bench.py launched as the driver launches it for N > 1 (one process per rank through torch.distributed.run),
    rehearsed on the one GPU of this box with the gloo backend: the strong-scaling preset shards 1024 frames."""
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
           "--preset", "c4_strong_1024", "--steps", "2", "--warmup", "1"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["frames_per_step"] == 1024 and line["config"]["frames_per_gpu_per_step"] == 512
    assert line["value"] > 0 and line["cpu_baseline"] is None

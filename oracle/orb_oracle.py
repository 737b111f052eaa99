"""ctypes binding of the CPU oracle (liborb_oracle.so). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the HIP extraction path never does.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "liborb_oracle.so"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])

BLUR_LEGACY, BLUR_FIXEDPOINT = 0, 1
TIE_CREATION_ORDER, TIE_HEAP_ADDRESS = 0, 1
FMA_OFF, FMA_ON = 0, 1


def build(force=False):
    """Compile the oracle with g++ (seconds). No-op when the library is newer than its sources."""
    srcs = [_HERE / "orb_oracle.cpp", _HERE / "orb_oracle.h", _HERE.parent / "include" / "orb_pattern_31.inc"]
    if not force and _LIB.exists() and all(_LIB.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return _LIB
    subprocess.check_call(["make", "-C", str(_HERE), "-B", "liborb_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB.exists():
            build()
        L = C.CDLL(str(_LIB))
        u8p, f32p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orbo_create.restype = C.c_void_p
        L.orbo_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orbo_destroy.argtypes = [C.c_void_p]
        L.orbo_extract.restype = C.c_int
        L.orbo_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, i32p, C.c_void_p]
        L.orbo_levels.argtypes = [C.c_void_p]
        L.orbo_scale_factor.restype = C.c_float
        L.orbo_scale_factor.argtypes = [C.c_void_p]
        L.orbo_table.argtypes = [C.c_void_p, C.c_int, f32p]
        L.orbo_features_per_level.argtypes = [C.c_void_p, i32p]
        L.orbo_umax.argtypes = [C.c_void_p, i32p]
        L.orbo_level_size.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
        L.orbo_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        L.orbo_blurred_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orbo_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orbo_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orbo_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        L.orbo_border_reflect101.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.orbo_fast9.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_int]
        L.orbo_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.orbo_fast_atan2.restype = C.c_float
        L.orbo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orbo_ic_angle.restype = C.c_float
        L.orbo_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int]
        L.orbo_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orbo_distribute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orbo_extract_batch_mt.restype = C.c_double
        L.orbo_extract_batch_mt.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                            C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, i32p]
        L.orbo_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orbo_undistort_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orbo_image_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orbo_undistort_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbo_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                     C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orbo_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orbo_projection_level_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
        L.orbo_remap_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t]
        L.orbo_search_map_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.orbo_is_in_frustum.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_int,
                                         C.c_float, C.c_void_p]
        L.orbo_search_by_projection_keyframe.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + \
            [C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orbo_search_by_bow.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orbo_search_by_bow_keyframes.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orbo_search_by_projection_sim3.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + \
            [C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orbo_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_float,
                                            C.c_float, C.c_void_p, C.c_void_p]
        L.orbo_unproject_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbo_libm_logf.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orbo_libm_sincos.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orbo_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """CPU restatement of ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc)."""

    def __init__(self, nfeatures, scale_factor, nlevels, ini_th, min_th,
                 blur_variant=BLUR_LEGACY, tie_mode=TIE_CREATION_ORDER, fma_mode=FMA_OFF):
        self._h = lib().orbo_create(nfeatures, scale_factor, nlevels, ini_th, min_th, blur_variant, tie_mode, fma_mode)
        if not self._h:
            raise ValueError("bad extractor parameters")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:     # (at interpreter shutdown the module may already be gone)
            _lib.orbo_destroy(self._h)
            self._h = None

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        # a level can keep 4 nodes per initial node whatever its quota, and has at most width / 30 initial nodes
        cap = self.nfeatures + (68 + 4 * (w // 30 + 1)) * self.nlevels + 64
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = lib().orbo_extract(self._h, _ptr(img), w, h, img.strides[0], _ptr(kp), cap, C.byref(n), _ptr(desc))
        if rc:
            raise RuntimeError("orbo_extract failed: %d" % rc)
        return kp[:n.value].copy(), desc[:n.value].copy()

    def table(self, which):
        out = np.zeros(self.nlevels, np.float32)
        lib().orbo_table(self._h, which, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def scale_factor(self):
        return lib().orbo_scale_factor(self._h)

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        lib().orbo_features_per_level(self._h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def umax(self):
        out = np.zeros(16, np.int32)
        lib().orbo_umax(self._h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        if lib().orbo_level_size(self._h, level, C.byref(w), C.byref(h)):
            raise IndexError(level)
        return w.value, h.value

    def level(self, level, with_border=False):
        w, h = self.level_size(level)
        b = 38 if with_border else 0
        out = np.zeros((h + b, w + b), np.uint8)
        lib().orbo_level_copy(self._h, level, _ptr(out), out.strides[0], int(with_border))
        return out

    def blurred(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        return out if lib().orbo_blurred_copy(self._h, level, _ptr(out), out.strides[0]) == 1 else None

    def candidates(self, level):
        n = lib().orbo_level_candidates(self._h, level, None, 0)
        out = np.zeros(max(n, 1), CAND_DTYPE)
        lib().orbo_level_candidates(self._h, level, _ptr(out), n)
        return out[:n]

    def level_keypoints(self, level):
        n = lib().orbo_level_keypoints(self._h, level, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib().orbo_level_keypoints(self._h, level, _ptr(out), n)
        return out[:n]


def stereo_match(left, right, kp_l, desc_l, kp_r, desc_r, mb, mbf):
    """Frame::ComputeStereoMatches (reference src/Frame.cc:481-653) on the last extractions of two
    OracleExtractor instances. Returns (mvuRight, mvDepth, matches kept)."""
    kp_l = np.ascontiguousarray(kp_l, KP_DTYPE); kp_r = np.ascontiguousarray(kp_r, KP_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, np.uint8); desc_r = np.ascontiguousarray(desc_r, np.uint8)
    uright = np.zeros(max(len(kp_l), 1), np.float32)
    depth = np.zeros(max(len(kp_l), 1), np.float32)
    n = lib().orbo_stereo_match(left._h, right._h, _ptr(kp_l), _ptr(desc_l), len(kp_l), _ptr(kp_r), _ptr(desc_r), len(kp_r),
                                mb, mbf, _ptr(uright), _ptr(depth))
    if n < 0:
        raise RuntimeError("orbo_stereo_match failed: %d" % n)
    return uright[:len(kp_l)], depth[:len(kp_l)], n


GRID_CELLS = 64 * 48


def _cam(cam):
    return np.ascontiguousarray(cam, np.float32).reshape(9)       # fx, fy, cx, cy, k1, k2, p1, p2, k3


def undistort_points(cam, pts):
    """cv::undistortPoints(pts, pts, K, distCoeffs, noArray(), K) for an (n, 2) float32 array."""
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros_like(pts)
    lib().orbo_undistort_points(_ptr(_cam(cam)), _ptr(pts), len(pts), _ptr(out))
    return out


def image_bounds(cam, width, height):
    out = np.zeros(4, np.float32)
    lib().orbo_image_bounds(_ptr(_cam(cam)), width, height, _ptr(out))
    return out


def undistort_grid(cam, bounds, kp):
    """Frame::UndistortKeyPoints + Frame::AssignFeaturesToGrid: returns (mvKeysUn, grid_start, grid_idx)."""
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    un = np.zeros(max(len(kp), 1), KP_DTYPE)
    gs = np.zeros(GRID_CELLS + 1, np.int32)
    gi = np.zeros(max(len(kp), 1), np.int32)
    lib().orbo_undistort_grid(_ptr(_cam(cam)), _ptr(np.ascontiguousarray(bounds, np.float32)), _ptr(kp), len(kp), _ptr(un), _ptr(gs), _ptr(gi))
    return un[:len(kp)], gs, gi[:gs[-1]]


def search_for_initialization(kp1, desc1, kp2, desc2, bounds, prev_matched, window=100, nnratio=0.9, check_orientation=True):
    """ORBmatcher::SearchForInitialization: returns (vnMatches12, updated vbPrevMatched, nmatches)."""
    kp1 = np.ascontiguousarray(kp1, KP_DTYPE); kp2 = np.ascontiguousarray(kp2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    prev = np.array(prev_matched, np.float32).reshape(-1, 2).copy()
    m12 = np.full(max(len(kp1), 1), -1, np.int32)
    n = lib().orbo_search_for_initialization(_ptr(kp1), _ptr(desc1), len(kp1), _ptr(kp2), _ptr(desc2), len(kp2),
                                             _ptr(np.ascontiguousarray(bounds, np.float32)), _ptr(prev), int(window),
                                             float(nnratio), int(bool(check_orientation)), _ptr(m12))
    return m12[:len(kp1)], prev, n


def projection_level_mode(Tcw_cur, Tcw_last, mb, mono):
    """bForward / bBackward of SearchByProjection(CurrentFrame, LastFrame): 0 neither, 1 forward, 2 backward.
    T* = 12 floats: Rcw row-major, then tcw."""
    a = np.ascontiguousarray(Tcw_cur, np.float32).reshape(12); b = np.ascontiguousarray(Tcw_last, np.float32).reshape(12)
    return lib().orbo_projection_level_mode(_ptr(a), _ptr(b), float(mb), int(bool(mono)))


def search_by_projection(kp_last, world_last, flags_last, desc_last, kp_cur, desc_cur, uright_cur, Tcw, cam4, mbf, bounds,
                         scale_factors, th, level_mode=0, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono): returns (match_cur, nmatches) with
    match_cur[i2] = index of the last-frame keypoint whose map point current keypoint i2 received, or -1."""
    kp_last = np.ascontiguousarray(kp_last, KP_DTYPE); kp_cur = np.ascontiguousarray(kp_cur, KP_DTYPE)
    world = np.ascontiguousarray(world_last, np.float32).reshape(-1, 3)
    flags = np.ascontiguousarray(flags_last, np.uint8)
    d1 = np.ascontiguousarray(desc_last, np.uint8); d2 = np.ascontiguousarray(desc_cur, np.uint8)
    ur = None if uright_cur is None else np.ascontiguousarray(uright_cur, np.float32)
    m = np.full(max(len(kp_cur), 1), -1, np.int32)
    n = lib().orbo_search_by_projection(_ptr(kp_last), _ptr(world), _ptr(flags), _ptr(d1), len(kp_last), _ptr(kp_cur), _ptr(d2),
                                        None if ur is None else _ptr(ur), len(kp_cur),
                                        _ptr(np.ascontiguousarray(Tcw, np.float32).reshape(12)),
                                        _ptr(np.ascontiguousarray(cam4, np.float32)), float(mbf),
                                        _ptr(np.ascontiguousarray(bounds, np.float32)),
                                        _ptr(np.ascontiguousarray(scale_factors, np.float32)), float(th), int(level_mode),
                                        int(bool(check_orientation)), _ptr(m))
    return m[:len(kp_cur)], n


def remap_linear(img, map_x, map_y):
    """cv::remap(img, dst, map_x, map_y, INTER_LINEAR) (constant border 0) of an (H, W) uint8 image; maps are float32."""
    img = np.ascontiguousarray(img, np.uint8)
    mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
    out = np.zeros(mx.shape, np.uint8)
    lib().orbo_remap_linear(_ptr(img), img.shape[1], img.shape[0], img.strides[0], _ptr(mx), _ptr(my), mx.shape[1], mx.shape[0],
                            _ptr(out), out.strides[0])
    return out


def cvt_gray(img, rgb):
    """cv::cvtColor(img, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) of an (H, W, 3|4) uint8 image."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, cn = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orbo_cvt_gray(_ptr(img), w, h, img.strides[0], cn, int(bool(rgb)), _ptr(out), out.strides[0])
    return out


def resize_linear_u8(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orbo_resize_linear_u8(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dw, dh, dst.strides[0])
    return dst


def border_reflect101(src, border):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().orbo_border_reflect101(_ptr(src), w, h, src.strides[0], _ptr(dst), dst.strides[0], border)
    return dst


def fast9(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    n = lib().orbo_fast9(_ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, None, 0)
    out = np.zeros(max(n, 1), CAND_DTYPE)
    lib().orbo_fast9(_ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, _ptr(out), n)
    return out[:n]


def gaussian_blur7(img, variant=BLUR_LEGACY):
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros_like(img)
    lib().orbo_gaussian_blur7(_ptr(img), img.shape[1], img.shape[0], img.strides[0], _ptr(dst), dst.strides[0], variant)
    return dst


def fast_atan2(y, x):
    return lib().orbo_fast_atan2(float(y), float(x))


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orbo_ic_angle(_ptr(img), img.shape[1], img.shape[0], img.strides[0], x, y)


def descriptor(img, x, y, angle_deg, fma_mode=FMA_OFF):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().orbo_descriptor(_ptr(img), img.shape[1], img.shape[0], img.strides[0], x, y, angle_deg, fma_mode, _ptr(out))
    return out


def distribute(cands, min_x, max_x, min_y, max_y, n_wanted, tie_mode=TIE_CREATION_ORDER):
    c = np.zeros(len(cands), CAND_DTYPE)
    for i, (x, y, r) in enumerate(cands):
        c[i] = (x, y, r)
    out = np.zeros(max(len(cands), 1), CAND_DTYPE)
    n = lib().orbo_distribute(_ptr(c), len(c), min_x, max_x, min_y, max_y, n_wanted, tie_mode, _ptr(out), len(out))
    if n < 0:
        raise RuntimeError("orbo_distribute failed: %d" % n)
    return out[:n]


def extract_batch_mt(params, imgs, nthreads):
    """CPU baseline: imgs is (F,H,W) uint8; returns (seconds, counts)."""
    imgs = np.ascontiguousarray(imgs, np.uint8)
    f, h, w = imgs.shape
    counts = np.zeros(f, np.int32)
    nfeat, sf, nl, ini, mn = params
    dt = lib().orbo_extract_batch_mt(nfeat, sf, nl, ini, mn, BLUR_LEGACY, _ptr(imgs), f, w, h,
                                     imgs.strides[1], imgs.strides[0], nthreads, counts.ctypes.data_as(C.POINTER(C.c_int)))
    return dt, counts


def libm_sincos(first_bits, count):
    """(cos, sin) float32 arrays of the host C library for `count` consecutive float bit patterns."""
    c = np.empty(count, np.float32)
    s = np.empty(count, np.float32)
    lib().orbo_libm_sincos(int(first_bits), int(count), _ptr(c), _ptr(s))
    return c, s


TRACK_DTYPE = np.dtype([("proj_x", np.float32), ("proj_y", np.float32), ("proj_xr", np.float32), ("view_cos", np.float32),
                        ("level", np.int32), ("flags", np.int32)])


def search_map_points(mp, desc_mp, kp, desc, uright, occupied, bounds, scale_factors, th, nnratio=0.8):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th): returns (match, nmatches) with match[idx] = index of the
    map point assigned to keypoint idx, or -1."""
    mp = np.ascontiguousarray(mp, TRACK_DTYPE); kp = np.ascontiguousarray(kp, KP_DTYPE)
    d1 = np.ascontiguousarray(desc_mp, np.uint8); d2 = np.ascontiguousarray(desc, np.uint8)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    m = np.full(max(len(kp), 1), -1, np.int32)
    n = lib().orbo_search_map_points(_ptr(mp), _ptr(d1), len(mp), _ptr(kp), _ptr(d2), None if ur is None else _ptr(ur),
                                     None if oc is None else _ptr(oc), len(kp), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                     _ptr(sf), len(sf), float(th), float(nnratio), _ptr(m))
    return m[:len(kp)], n


MAPPOINT_DTYPE = np.dtype([("pos", np.float32, 3), ("normal", np.float32, 3), ("min_distance", np.float32),
                           ("max_distance", np.float32), ("flags", np.int32)])


def is_in_frustum(mp, pose, cam4, mbf, bounds, scale_factor, nlevels, viewing_cos_limit=0.5):
    """Frame::isInFrustum for an array of map points: returns (track records, number in view)."""
    mp = np.ascontiguousarray(mp, MAPPOINT_DTYPE)
    out = np.zeros(max(len(mp), 1), TRACK_DTYPE)
    n = lib().orbo_is_in_frustum(_ptr(mp), len(mp), _ptr(np.ascontiguousarray(pose, np.float32).reshape(15)),
                                 _ptr(np.ascontiguousarray(cam4, np.float32)), float(mbf), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                 float(np.float32(scale_factor)), int(nlevels), float(viewing_cos_limit), _ptr(out))
    return out[:len(mp)], n


def libm_logf(first_bits, count, stride=1):
    """float32 logf of the host C library for the float bit patterns first_bits + i * stride."""
    out = np.empty(count, np.float32)
    lib().orbo_libm_logf(int(first_bits), int(count), int(stride), _ptr(out))
    return out


def search_by_projection_keyframe(kp_kf, world_kf, dist_range_kf, flags_kf, desc_kf, kp_cur, desc_cur, occupied_cur, Tcw, cam4, bounds,
                                  scale_factors, scale_factor, th, orb_dist, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): returns (match_cur, nmatches)."""
    kp_kf = np.ascontiguousarray(kp_kf, KP_DTYPE); kp_cur = np.ascontiguousarray(kp_cur, KP_DTYPE)
    world = np.ascontiguousarray(world_kf, np.float32).reshape(-1, 3)
    rng_ = np.ascontiguousarray(dist_range_kf, np.float32).reshape(-1, 2)
    flags = np.ascontiguousarray(flags_kf, np.uint8)
    d1 = np.ascontiguousarray(desc_kf, np.uint8); d2 = np.ascontiguousarray(desc_cur, np.uint8)
    oc = None if occupied_cur is None else np.ascontiguousarray(occupied_cur, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    m = np.full(max(len(kp_cur), 1), -1, np.int32)
    n = lib().orbo_search_by_projection_keyframe(_ptr(kp_kf), _ptr(world), _ptr(rng_), _ptr(flags), _ptr(d1), len(kp_kf), _ptr(kp_cur),
                                                 _ptr(d2), None if oc is None else _ptr(oc), len(kp_cur),
                                                 _ptr(np.ascontiguousarray(Tcw, np.float32).reshape(12)),
                                                 _ptr(np.ascontiguousarray(cam4, np.float32)), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                                 _ptr(sf), float(np.float32(scale_factor)), len(sf), float(th), int(orb_dist),
                                                 int(bool(check_orientation)), _ptr(m))
    return m[:len(kp_cur)], n


def search_by_bow(kp_kf, desc_kf, node_kf, flags_kf, kp_f, desc_f, node_f, nnratio=0.7, check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches): returns (match_f, nmatches), match_f[i2] = key-frame feature or -1."""
    kp_kf = np.ascontiguousarray(kp_kf, KP_DTYPE); kp_f = np.ascontiguousarray(kp_f, KP_DTYPE)
    d1 = np.ascontiguousarray(desc_kf, np.uint8); d2 = np.ascontiguousarray(desc_f, np.uint8)
    n1 = np.ascontiguousarray(node_kf, np.uint32); n2 = np.ascontiguousarray(node_f, np.uint32)
    fl = np.ascontiguousarray(flags_kf, np.uint8)
    m = np.full(max(len(kp_f), 1), -1, np.int32)
    n = lib().orbo_search_by_bow(_ptr(kp_kf), _ptr(d1), _ptr(n1), _ptr(fl), len(kp_kf), _ptr(kp_f), _ptr(d2), _ptr(n2), len(kp_f),
                                 float(nnratio), int(bool(check_orientation)), _ptr(m))
    return m[:len(kp_f)], n


def search_by_bow_keyframes(kp1, desc1, node1, flags1, kp2, desc2, node2, flags2, nnratio=0.75, check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12): returns (match12, nmatches), match12[i1] = feature of key frame 2 or -1."""
    kp1 = np.ascontiguousarray(kp1, KP_DTYPE); kp2 = np.ascontiguousarray(kp2, KP_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    n1 = np.ascontiguousarray(node1, np.uint32); n2 = np.ascontiguousarray(node2, np.uint32)
    f1 = np.ascontiguousarray(flags1, np.uint8); f2 = np.ascontiguousarray(flags2, np.uint8)
    m = np.full(max(len(kp1), 1), -1, np.int32)
    n = lib().orbo_search_by_bow_keyframes(_ptr(kp1), _ptr(d1), _ptr(n1), _ptr(f1), len(kp1), _ptr(kp2), _ptr(d2), _ptr(n2), _ptr(f2),
                                           len(kp2), float(nnratio), int(bool(check_orientation)), _ptr(m))
    return m[:len(kp1)], n


def search_by_projection_sim3(world, normal, dist_range, flags, desc, kp_kf, desc_kf, occupied_kf, Scw, cam4, bounds, scale_factors,
                              scale_factor, th):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th): returns (match_kf, nmatches)."""
    kp_kf = np.ascontiguousarray(kp_kf, KP_DTYPE)
    w = np.ascontiguousarray(world, np.float32).reshape(-1, 3); nr = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
    rg = np.ascontiguousarray(dist_range, np.float32).reshape(-1, 2)
    fl = np.ascontiguousarray(flags, np.uint8)
    d1 = np.ascontiguousarray(desc, np.uint8); d2 = np.ascontiguousarray(desc_kf, np.uint8)
    oc = None if occupied_kf is None else np.ascontiguousarray(occupied_kf, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    m = np.full(max(len(kp_kf), 1), -1, np.int32)
    n = lib().orbo_search_by_projection_sim3(_ptr(w), _ptr(nr), _ptr(rg), _ptr(fl), _ptr(d1), len(w), _ptr(kp_kf), _ptr(d2),
                                             None if oc is None else _ptr(oc), len(kp_kf),
                                             _ptr(np.ascontiguousarray(Scw, np.float32).reshape(12)),
                                             _ptr(np.ascontiguousarray(cam4, np.float32)), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                             _ptr(sf), float(np.float32(scale_factor)), len(sf), int(th), _ptr(m))
    return m[:len(kp_kf)], n


def stereo_from_rgbd(kp, kp_un, depth, depth_map_factor, mbf):
    """Frame::ComputeStereoFromRGBD behind GrabImageRGBD's conversion: depth is an (H, W) uint16 or float32 map.
    Returns (mvuRight, mvDepth)."""
    kp = np.ascontiguousarray(kp, KP_DTYPE); kp_un = np.ascontiguousarray(kp_un, KP_DTYPE)
    depth = np.ascontiguousarray(depth)
    assert depth.dtype in (np.uint16, np.float32)
    ur = np.zeros(max(len(kp), 1), np.float32); dp = np.zeros(max(len(kp), 1), np.float32)
    lib().orbo_stereo_from_rgbd(_ptr(kp), _ptr(kp_un), len(kp), _ptr(depth), int(depth.dtype == np.uint16), depth.shape[1], depth.shape[0],
                                depth.strides[0], float(np.float32(depth_map_factor)), float(np.float32(mbf)), _ptr(ur), _ptr(dp))
    return ur[:len(kp)], dp[:len(kp)]


def unproject_stereo(kp_un, depth, pose_wc, cam4):
    """Frame::UnprojectStereo for every keypoint: returns (world [n, 3], valid [n])."""
    kp_un = np.ascontiguousarray(kp_un, KP_DTYPE); depth = np.ascontiguousarray(depth, np.float32)
    n = len(kp_un)
    world = np.zeros((max(n, 1), 3), np.float32); valid = np.zeros(max(n, 1), np.uint8)
    lib().orbo_unproject_stereo(_ptr(kp_un), _ptr(depth), n, _ptr(np.ascontiguousarray(pose_wc, np.float32).reshape(12)),
                                _ptr(np.ascontiguousarray(cam4, np.float32)), _ptr(world), _ptr(valid))
    return world[:n], valid[:n]

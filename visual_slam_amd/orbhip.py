"""ctypes binding of libORBhip and a host class with the reference's ORBextractor interface.

`ORBextractor` mirrors ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111): same
constructor arguments, `__call__(image, mask)` for operator(), the six accessors and
`mvImagePyramid`. Everything is computed by the HIP library; there is no CPU fallback and the
import fails loudly when the library is missing.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("ORBHIP_LIB", _PKG / "libORBhip.so"))   # override only for kernel experiments

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])
assert KP_DTYPE.itemsize == 28

BLUR_LEGACY, BLUR_FIXEDPOINT = 0, 1
OK, ERR_ARG, ERR_EMPTY, ERR_CAPACITY, ERR_GEOMETRY, ERR_HIP, ERR_NO_DEVICE, ERR_STATE = 0, -1, -2, -3, -4, -5, -6, -7

# every symbol include/orbhip.h declares
EXPORTS = [
    "orbhip_create", "orbhip_destroy", "orbhip_extract", "orbhip_extract_batch_device",
    "orbhip_extract_batch", "orbhip_synchronize", "orbhip_max_keypoints", "orbhip_max_keypoints_for_image", "orbhip_get_levels",
    "orbhip_get_scale_factor", "orbhip_get_scale_factors", "orbhip_get_inverse_scale_factors",
    "orbhip_get_scale_sigma_squares", "orbhip_get_inverse_scale_sigma_squares",
    "orbhip_get_features_per_level", "orbhip_get_umax", "orbhip_pyramid_info", "orbhip_pyramid_copy",
    "orbhip_debug_candidates", "orbhip_debug_level_keypoints", "orbhip_enable_timing", "orbhip_get_timing", "orbhip_get_timing_busy", "orbhip_strerror",
    "orbhip_last_error", "orbhip_stereo_match", "orbhip_stereo_match_batch_device",
    "orbhip_stereo_match_batch_device_noalloc", "orbhip_extract_stereo_batch_device",
    "orbhip_extract_color", "orbhip_extract_batch_device_color",
    "orbhip_image_bounds", "orbhip_undistort_grid", "orbhip_undistort_grid_batch_device",
    "orbhip_search_for_initialization", "orbhip_search_for_initialization_batch_device",
    "orbhip_search_by_projection", "orbhip_search_by_projection_batch_device", "orbhip_projection_level_mode",
    "orbhip_set_rectification", "orbhip_debug_sincos",
    "orbhip_search_map_points", "orbhip_search_map_points_batch_device",
    "orbhip_is_in_frustum", "orbhip_is_in_frustum_batch_device", "orbhip_debug_logf",
    "orbhip_search_by_projection_keyframe", "orbhip_search_by_projection_keyframe_batch_device",
    "orbhip_search_by_bow", "orbhip_search_by_bow_batch_device",
    "orbhip_search_by_bow_keyframes", "orbhip_search_by_bow_keyframes_batch_device",
    "orbhip_extract_batch_multi", "orbhip_shard_range", "orbhip_extract_stereo_batch", "orbhip_extract_stereo_batch_multi",
    "orbhip_stereo_from_rgbd_batch_device", "orbhip_unproject_stereo_batch_device", "orbhip_decompose_sim3", "orbhip_search_by_projection_sim3", "orbhip_search_by_projection_sim3_batch_device",
]


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("device", C.c_int),
                ("blur_variant", C.c_int), ("max_batch", C.c_int)]


class OrbHipError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().orbhip_strerror(code).decode()
        super().__init__("libORBhip: %s (%d)%s" % (msg, code, (": " + detail) if detail else ""))


_lib = None


def lib():
    """Loads libORBhip.so from the package directory (build it with visual_slam_amd/build.py)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError("libORBhip.so is missing: run `python -m visual_slam_amd.build` (hipcc, gfx950)")
        L = C.CDLL(str(LIB_PATH))
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orbhip_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
        L.orbhip_destroy.argtypes = [vp]
        L.orbhip_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_int, ip, vp]
        L.orbhip_extract_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                  vp, vp, vp, C.c_int, vp]
        L.orbhip_extract_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp, vp, vp, C.c_int]
        L.orbhip_synchronize.argtypes = [vp]
        L.orbhip_max_keypoints.argtypes = [vp]
        L.orbhip_max_keypoints_for_image.argtypes = [vp, C.c_int, C.c_int]
        L.orbhip_get_levels.argtypes = [vp]
        L.orbhip_get_scale_factor.argtypes = [vp]
        L.orbhip_get_scale_factor.restype = C.c_float
        for name in ("orbhip_get_scale_factors", "orbhip_get_inverse_scale_factors",
                     "orbhip_get_scale_sigma_squares", "orbhip_get_inverse_scale_sigma_squares"):
            getattr(L, name).argtypes = [vp, fp]
        L.orbhip_get_features_per_level.argtypes = [vp, ip]
        L.orbhip_get_umax.argtypes = [vp, ip]
        L.orbhip_pyramid_info.argtypes = [vp, C.c_int, ip, ip]
        L.orbhip_pyramid_copy.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, C.c_int]
        L.orbhip_debug_candidates.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.orbhip_debug_level_keypoints.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.orbhip_enable_timing.argtypes = [vp, C.c_int]
        L.orbhip_get_timing.argtypes = [vp, C.POINTER(C.c_char_p), fp, ip, C.c_int]
        L.orbhip_get_timing_busy.argtypes = [vp, fp, C.c_int]
        L.orbhip_strerror.argtypes = [C.c_int]
        L.orbhip_strerror.restype = C.c_char_p
        L.orbhip_last_error.argtypes = [vp]
        L.orbhip_last_error.restype = C.c_char_p
        L.orbhip_extract_color.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, vp, C.c_int, ip, vp]
        L.orbhip_extract_batch_device_color.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                                        vp, vp, vp, C.c_int, vp]
        L.orbhip_image_bounds.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orbhip_undistort_grid.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
        L.orbhip_undistort_grid_batch_device.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
        L.orbhip_search_for_initialization.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_int, vp, ip]
        L.orbhip_search_for_initialization_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int,
                                                                    C.c_float, C.c_int, vp, vp, vp]
        L.orbhip_search_by_projection.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, ip]
        L.orbhip_search_by_projection_batch_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                               vp, vp, vp]
        L.orbhip_projection_level_mode.argtypes = [vp, vp, C.c_float, C.c_int]
        L.orbhip_set_rectification.argtypes = [vp, vp, vp, C.c_int, C.c_int]
        L.orbhip_debug_sincos.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp]
        L.orbhip_search_by_projection_keyframe.argtypes = [vp] * 6 + [C.c_int] + [vp] * 3 + [C.c_int, vp, vp, C.c_int, vp, ip]
        L.orbhip_search_by_projection_keyframe_batch_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 12 + [C.c_int, vp, vp, vp]
        L.orbhip_search_by_bow.argtypes = [vp] * 5 + [C.c_int] + [vp] * 3 + [C.c_int, C.c_float, C.c_int, vp, ip]
        L.orbhip_search_by_bow_batch_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 9 + [C.c_float, C.c_int, vp, vp, vp]
        L.orbhip_search_by_bow_keyframes.argtypes = [vp] * 5 + [C.c_int] + [vp] * 4 + [C.c_int, C.c_float, C.c_int, vp, ip]
        L.orbhip_search_by_bow_keyframes_batch_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 10 + [C.c_float, C.c_int, vp, vp, vp]
        L.orbhip_decompose_sim3.argtypes = [vp, vp]
        L.orbhip_search_by_projection_sim3.argtypes = [vp] * 6 + [C.c_int] + [vp] * 3 + [C.c_int, vp, vp, vp, ip]
        L.orbhip_search_by_projection_sim3_batch_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 14 + [vp]
        L.orbhip_stereo_from_rgbd_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                                           C.c_size_t, C.c_float, C.c_float, vp, vp, vp]
        L.orbhip_unproject_stereo_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                                           vp, vp, vp]
        L.orbhip_debug_logf.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
        L.orbhip_is_in_frustum.argtypes = [vp, vp, C.c_int, vp, vp, C.c_float, vp, ip]
        L.orbhip_is_in_frustum_batch_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_float, vp, vp, vp]
        L.orbhip_search_map_points.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_float, C.c_float, vp, ip]
        L.orbhip_search_map_points_batch_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp,
                                                            C.c_float, C.c_float, vp, vp, vp]
        L.orbhip_stereo_match.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, ip]
        L.orbhip_stereo_match_batch_device.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float,
                                                       vp, vp, vp, vp]
        L.orbhip_stereo_match_batch_device_noalloc.argtypes = L.orbhip_stereo_match_batch_device.argtypes
        L.orbhip_extract_stereo_batch_device.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                         vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp, vp]
        L.orbhip_extract_batch_multi.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp, vp, vp, C.c_int]
        L.orbhip_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip, ip]
        L.orbhip_extract_stereo_batch.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                  vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]
        L.orbhip_extract_stereo_batch_multi.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                        vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


GRID_CELLS = 64 * 48          # FRAME_GRID_COLS x FRAME_GRID_ROWS (include/Frame.h:37-38)


def _cam(cam):
    return np.ascontiguousarray(cam, np.float32).reshape(9)       # fx, fy, cx, cy, k1, k2, p1, p2, k3


def ComputeImageBounds(cam, width, height):
    """Frame::ComputeImageBounds (reference src/Frame.cc:450-478): mnMinX, mnMaxX, mnMinY, mnMaxY."""
    out = np.zeros(4, np.float32)
    rc = lib().orbhip_image_bounds(_ptr(_cam(cam)), width, height, _ptr(out))
    if rc:
        raise OrbHipError(rc)
    return out


def UndistortAndAssignToGrid(ex, kp, cam, bounds):
    """Frame::UndistortKeyPoints + Frame::AssignFeaturesToGrid (reference src/Frame.cc:422-449, 255-270)
    on the device: returns (mvKeysUn, grid_start[64*48+1], grid_idx)."""
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    n = len(kp)
    un = np.zeros(max(n, 1), KP_DTYPE)
    gs = np.zeros(GRID_CELLS + 1, np.int32)
    gi = np.zeros(max(n, 1), np.int32)
    ex._check(lib().orbhip_undistort_grid(ex._h, _ptr(kp), n, _ptr(_cam(cam)), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                          _ptr(un), _ptr(gs), _ptr(gi)))
    return un[:n], gs, gi[:gs[-1]]


def SearchForInitialization(ex, kp1, desc1, kp2, desc2, bounds, prev_matched, window=100, nnratio=0.9, check_orientation=True):
    """ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:402-517) on the device: returns
    (vnMatches12, updated vbPrevMatched, nmatches)."""
    kp1 = np.ascontiguousarray(kp1, KP_DTYPE); kp2 = np.ascontiguousarray(kp2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    prev = np.array(prev_matched, np.float32).reshape(-1, 2).copy()
    m12 = np.full(max(len(kp1), 1), -1, np.int32)
    n = C.c_int(0)
    ex._check(lib().orbhip_search_for_initialization(ex._h, _ptr(kp1), _ptr(desc1), len(kp1), _ptr(kp2), _ptr(desc2), len(kp2),
                                                     _ptr(np.ascontiguousarray(bounds, np.float32)), _ptr(prev), int(window),
                                                     float(nnratio), int(bool(check_orientation)), _ptr(m12), C.byref(n)))
    return m12[:len(kp1)], prev, n.value


class ProjectionParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float),
                ("bounds", C.c_float * 4), ("th", C.c_float), ("check_orientation", C.c_int)]


def projection_params(cam4, mbf, bounds, th, check_orientation=True):
    b = [float(v) for v in bounds]
    return ProjectionParams(float(cam4[0]), float(cam4[1]), float(cam4[2]), float(cam4[3]), float(mbf),
                            (C.c_float * 4)(*b), float(th), int(bool(check_orientation)))


def ProjectionLevelMode(Tcw_cur, Tcw_last, mb, mono):
    """bForward / bBackward of SearchByProjection(CurrentFrame, LastFrame) (reference src/ORBmatcher.cc:1330-1337):
    0 neither, 1 forward, 2 backward. Poses as 12 floats: R row-major, then t."""
    a = np.ascontiguousarray(Tcw_cur, np.float32).reshape(12); b = np.ascontiguousarray(Tcw_last, np.float32).reshape(12)
    return lib().orbhip_projection_level_mode(_ptr(a), _ptr(b), float(mb), int(bool(mono)))


def SearchByProjection(ex, kp_last, world_last, flags_last, desc_last, kp_cur, desc_cur, uright_cur, Tcw, cam4, mbf, bounds, th,
                       level_mode=0, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (reference src/ORBmatcher.cc:1324-1445)
    on the device. Returns (match_cur, nmatches): match_cur[i2] = index of the last-frame keypoint whose
    map point current keypoint i2 received, or -1."""
    kp_last = np.ascontiguousarray(kp_last, KP_DTYPE); kp_cur = np.ascontiguousarray(kp_cur, KP_DTYPE)
    world = np.ascontiguousarray(world_last, np.float32).reshape(-1, 3)
    flags = np.ascontiguousarray(flags_last, np.uint8)
    d1 = np.ascontiguousarray(desc_last, np.uint8); d2 = np.ascontiguousarray(desc_cur, np.uint8)
    ur = None if uright_cur is None else np.ascontiguousarray(uright_cur, np.float32)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    prm = projection_params(cam4, mbf, bounds, th, check_orientation)
    m = np.full(max(len(kp_cur), 1), -1, np.int32)
    n = C.c_int(0)
    ex._check(lib().orbhip_search_by_projection(ex._h, _ptr(kp_last), _ptr(world), _ptr(flags), _ptr(d1), len(kp_last),
                                                _ptr(kp_cur), _ptr(d2), None if ur is None else _ptr(ur), len(kp_cur),
                                                _ptr(T), int(level_mode), C.byref(prm), _ptr(m), C.byref(n)))
    return m[:len(kp_cur)], n.value


def SearchByProjectionKeyFrame(ex, kp_kf, world_kf, dist_range_kf, flags_kf, desc_kf, kp_cur, desc_cur, occupied_cur, Tcw, cam4,
                               bounds, th, orb_dist, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (reference
    src/ORBmatcher.cc:1447-1564, relocalisation) on the device. Returns (match_cur, nmatches)."""
    kp_kf = np.ascontiguousarray(kp_kf, KP_DTYPE); kp_cur = np.ascontiguousarray(kp_cur, KP_DTYPE)
    world = np.ascontiguousarray(world_kf, np.float32).reshape(-1, 3)
    rng_ = np.ascontiguousarray(dist_range_kf, np.float32).reshape(-1, 2)
    flags = np.ascontiguousarray(flags_kf, np.uint8)
    d1 = np.ascontiguousarray(desc_kf, np.uint8); d2 = np.ascontiguousarray(desc_cur, np.uint8)
    oc = None if occupied_cur is None else np.ascontiguousarray(occupied_cur, np.uint8)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(12)
    prm = projection_params(cam4, 0.0, bounds, th, check_orientation)
    m = np.full(max(len(kp_cur), 1), -1, np.int32)
    n = C.c_int(0)
    ex._check(lib().orbhip_search_by_projection_keyframe(ex._h, _ptr(kp_kf), _ptr(world), _ptr(rng_), _ptr(flags), _ptr(d1), len(kp_kf),
                                                         _ptr(kp_cur), _ptr(d2), None if oc is None else _ptr(oc), len(kp_cur),
                                                         _ptr(T), C.byref(prm), int(orb_dist), _ptr(m), C.byref(n)))
    return m[:len(kp_cur)], n.value


def SearchByBoW(ex, kp_kf, desc_kf, node_kf, flags_kf, kp_f, desc_f, node_f, nnratio=0.7, check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (reference src/ORBmatcher.cc:156-285) on the device; node_*:
    vocabulary node of every feature (0xFFFFFFFF: none). Returns (match_f, nmatches)."""
    kp_kf = np.ascontiguousarray(kp_kf, KP_DTYPE); kp_f = np.ascontiguousarray(kp_f, KP_DTYPE)
    d1 = np.ascontiguousarray(desc_kf, np.uint8); d2 = np.ascontiguousarray(desc_f, np.uint8)
    n1 = np.ascontiguousarray(node_kf, np.uint32); n2 = np.ascontiguousarray(node_f, np.uint32)
    fl = np.ascontiguousarray(flags_kf, np.uint8)
    m = np.full(max(len(kp_f), 1), -1, np.int32)
    n = C.c_int(0)
    ex._check(lib().orbhip_search_by_bow(ex._h, _ptr(kp_kf), _ptr(d1), _ptr(n1), _ptr(fl), len(kp_kf), _ptr(kp_f), _ptr(d2), _ptr(n2),
                                         len(kp_f), float(nnratio), int(bool(check_orientation)), _ptr(m), C.byref(n)))
    return m[:len(kp_f)], n.value


def SearchByBoWKeyFrames(ex, kp1, desc1, node1, flags1, kp2, desc2, node2, flags2, nnratio=0.75, check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (reference src/ORBmatcher.cc:519-652) on the device.
    Returns (match12, nmatches): match12[i1] = feature of key frame 2 whose map point vpMatches12[i1] receives, or -1."""
    kp1 = np.ascontiguousarray(kp1, KP_DTYPE); kp2 = np.ascontiguousarray(kp2, KP_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    n1 = np.ascontiguousarray(node1, np.uint32); n2 = np.ascontiguousarray(node2, np.uint32)
    f1 = np.ascontiguousarray(flags1, np.uint8); f2 = np.ascontiguousarray(flags2, np.uint8)
    m = np.full(max(len(kp1), 1), -1, np.int32)
    n = C.c_int(0)
    ex._check(lib().orbhip_search_by_bow_keyframes(ex._h, _ptr(kp1), _ptr(d1), _ptr(n1), _ptr(f1), len(kp1), _ptr(kp2), _ptr(d2), _ptr(n2),
                                                   _ptr(f2), len(kp2), float(nnratio), int(bool(check_orientation)), _ptr(m), C.byref(n)))
    return m[:len(kp1)], n.value


def search_by_bow_batch_device(ex, nframes, cap, d_kp_kf, d_desc_kf, d_node_kf, d_flags_kf, d_cnt_kf, d_kp_f, d_desc_f, d_node_f,
                               d_cnt_f, nnratio, check_orientation, d_match_f, d_nmatches, stream=None):
    """Raw-pointer batch call of SearchByBoW on device-resident key frame / frame pairs."""
    ex._check(lib().orbhip_search_by_bow_batch_device(ex._h, nframes, cap, d_kp_kf, d_desc_kf, d_node_kf, d_flags_kf, d_cnt_kf, d_kp_f,
                                                      d_desc_f, d_node_f, d_cnt_f, float(nnratio), int(bool(check_orientation)),
                                                      d_match_f, d_nmatches, stream))


def SearchByProjectionSim3(ex, world, normal, dist_range, flags, desc, kp_kf, desc_kf, occupied_kf, Scw, cam4, bounds, th):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (reference src/ORBmatcher.cc:287-400, loop
    closing) on the device. Scw: 12 floats, sR row-major then t. Returns (match_kf, nmatches)."""
    kp_kf = np.ascontiguousarray(kp_kf, KP_DTYPE)
    w = np.ascontiguousarray(world, np.float32).reshape(-1, 3); nr = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
    rg = np.ascontiguousarray(dist_range, np.float32).reshape(-1, 2)
    fl = np.ascontiguousarray(flags, np.uint8)
    d1 = np.ascontiguousarray(desc, np.uint8); d2 = np.ascontiguousarray(desc_kf, np.uint8)
    oc = None if occupied_kf is None else np.ascontiguousarray(occupied_kf, np.uint8)
    S = np.ascontiguousarray(Scw, np.float32).reshape(12)
    prm = projection_params(cam4, 0.0, bounds, float(int(th)), False)
    m = np.full(max(len(kp_kf), 1), -1, np.int32)
    n = C.c_int(0)
    ex._check(lib().orbhip_search_by_projection_sim3(ex._h, _ptr(w), _ptr(nr), _ptr(rg), _ptr(fl), _ptr(d1), len(w), _ptr(kp_kf), _ptr(d2),
                                                     None if oc is None else _ptr(oc), len(kp_kf), _ptr(S), C.byref(prm), _ptr(m),
                                                     C.byref(n)))
    return m[:len(kp_kf)], n.value


def search_by_projection_batch_device(ex, nframes, cap, d_kp_last, d_world_last, d_flags_last, d_desc_last, d_cnt_last,
                                      d_kp_cur, d_desc_cur, d_uright_cur, d_cnt_cur, d_Tcw, d_level_mode, prm,
                                      d_match_cur, d_nmatches, stream=None):
    """Raw-pointer batch call of SearchByProjection on device-resident frame pairs (prm: ProjectionParams)."""
    ex._check(lib().orbhip_search_by_projection_batch_device(ex._h, nframes, cap, d_kp_last, d_world_last, d_flags_last,
                                                             d_desc_last, d_cnt_last, d_kp_cur, d_desc_cur, d_uright_cur,
                                                             d_cnt_cur, d_Tcw, d_level_mode, C.byref(prm), d_match_cur,
                                                             d_nmatches, stream))


TRACK_DTYPE = np.dtype([("proj_x", np.float32), ("proj_y", np.float32), ("proj_xr", np.float32), ("view_cos", np.float32),
                        ("level", np.int32), ("flags", np.int32)])                  # orbhip_track_point


MAPPOINT_DTYPE = np.dtype([("pos", np.float32, 3), ("normal", np.float32, 3), ("min_distance", np.float32),
                           ("max_distance", np.float32), ("flags", np.int32)])        # orbhip_map_point


def IsInFrustum(ex, mp, pose, cam4, mbf, bounds, viewing_cos_limit=0.5):
    """Frame::isInFrustum (reference src/Frame.cc:294-350) for an array of map points (MAPPOINT_DTYPE) on the
    device. pose: 15 floats, Rcw row-major, tcw, Ow. Returns (TRACK_DTYPE records, number in view)."""
    mp = np.ascontiguousarray(mp, MAPPOINT_DTYPE)
    out = np.zeros(max(len(mp), 1), TRACK_DTYPE)
    prm = projection_params(cam4, mbf, bounds, 0.0, False)
    n = C.c_int(0)
    ex._check(lib().orbhip_is_in_frustum(ex._h, _ptr(mp), len(mp), _ptr(np.ascontiguousarray(pose, np.float32).reshape(15)),
                                         C.byref(prm), float(viewing_cos_limit), _ptr(out), C.byref(n)))
    return out[:len(mp)], n.value


def is_in_frustum_batch_device(ex, nframes, cap_mp, d_mp, d_cnt_mp, d_pose, prm, viewing_cos_limit, d_track, d_nvisible, stream=None):
    """Raw-pointer batch call of IsInFrustum on device-resident map points (prm: ProjectionParams)."""
    ex._check(lib().orbhip_is_in_frustum_batch_device(ex._h, nframes, cap_mp, d_mp, d_cnt_mp, d_pose, C.byref(prm),
                                                      float(viewing_cos_limit), d_track, d_nvisible, stream))


def SearchLocalPoints(ex, mp, desc_mp, kp, desc, uright, occupied, bounds, th, nnratio=0.8):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) (reference src/ORBmatcher.cc:46-127, the matcher of
    Tracking::SearchLocalPoints) on the device. mp: TRACK_DTYPE records of the map points (what Frame::isInFrustum
    left in them, flags bit 0 = in view and not bad, bit 1 = has observations). Returns (match, nmatches):
    match[idx] = index of the map point keypoint idx received, or -1."""
    mp = np.ascontiguousarray(mp, TRACK_DTYPE); kp = np.ascontiguousarray(kp, KP_DTYPE)
    d1 = np.ascontiguousarray(desc_mp, np.uint8); d2 = np.ascontiguousarray(desc, np.uint8)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    m = np.full(max(len(kp), 1), -1, np.int32)
    n = C.c_int(0)
    ex._check(lib().orbhip_search_map_points(ex._h, _ptr(mp), _ptr(d1), len(mp), _ptr(kp), _ptr(d2), None if ur is None else _ptr(ur),
                                             None if oc is None else _ptr(oc), len(kp), _ptr(np.ascontiguousarray(bounds, np.float32)),
                                             float(th), float(nnratio), _ptr(m), C.byref(n)))
    return m[:len(kp)], n.value


def search_map_points_batch_device(ex, nframes, cap_mp, d_mp, d_desc_mp, d_cnt_mp, cap_kp, d_kp, d_desc, d_uright, d_occupied,
                                   d_cnt_kp, bounds, th, nnratio, d_match, d_nmatches, stream=None):
    """Raw-pointer batch call of SearchLocalPoints on device-resident frames."""
    b = np.ascontiguousarray(bounds, np.float32)
    ex._check(lib().orbhip_search_map_points_batch_device(ex._h, nframes, cap_mp, d_mp, d_desc_mp, d_cnt_mp, cap_kp, d_kp, d_desc,
                                                          d_uright, d_occupied, d_cnt_kp, _ptr(b), float(th), float(nnratio),
                                                          d_match, d_nmatches, stream))


def stereo_from_rgbd_batch_device(ex, nframes, d_kp, d_kp_un, d_counts, cap, d_depth, depth_is_u16, width, height, row_stride,
                                  frame_stride, depth_map_factor, mbf, d_uright, d_depth_out, stream=None):
    """Frame::ComputeStereoFromRGBD (reference src/Frame.cc:655-672) on device-resident frames and depth maps."""
    ex._check(lib().orbhip_stereo_from_rgbd_batch_device(ex._h, nframes, d_kp, d_kp_un, d_counts, cap, d_depth, int(bool(depth_is_u16)),
                                                         width, height, row_stride, frame_stride, float(depth_map_factor), float(mbf),
                                                         d_uright, d_depth_out, stream))


def unproject_stereo_batch_device(ex, nframes, d_kp_un, d_depth, d_counts, cap, d_pose_wc, cam4, d_world, d_valid, stream=None):
    """Frame::UnprojectStereo (reference src/Frame.cc:674-687) for all keypoints of device-resident frames."""
    ex._check(lib().orbhip_unproject_stereo_batch_device(ex._h, nframes, d_kp_un, d_depth, d_counts, cap, d_pose_wc, float(cam4[0]),
                                                         float(cam4[1]), float(cam4[2]), float(cam4[3]), d_world, d_valid, stream))


def ComputeStereoMatches(left, right, kp_l, desc_l, kp_r, desc_r, mb, mbf):
    """Frame::ComputeStereoMatches (reference src/Frame.cc:481-653) for the pair the two ORBextractor
    instances extracted last. Returns (mvuRight, mvDepth, matches kept)."""
    kp_l = np.ascontiguousarray(kp_l, KP_DTYPE); kp_r = np.ascontiguousarray(kp_r, KP_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, np.uint8); desc_r = np.ascontiguousarray(desc_r, np.uint8)
    n_l, n_r = len(kp_l), len(kp_r)
    uright = np.full(max(n_l, 1), -1.0, np.float32)
    depth = np.full(max(n_l, 1), -1.0, np.float32)
    nm = C.c_int(0)
    left._check(lib().orbhip_stereo_match(left._h, right._h, _ptr(kp_l), _ptr(desc_l), n_l, _ptr(kp_r), _ptr(desc_r), n_r,
                                          mb, mbf, _ptr(uright), _ptr(depth), C.byref(nm)))
    return uright[:n_l], depth[:n_l], nm.value


def stereo_match_batch_device(left, right, nframes, d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap,
                              mb, mbf, d_uright, d_depth, d_nmatch=None, stream=None, noalloc=False):
    """Raw-pointer batch call on the device-resident outputs of two extract_batch_device calls.
    noalloc: the entry point that works on the library's static scratch and allocates nothing."""
    fn = lib().orbhip_stereo_match_batch_device_noalloc if noalloc else lib().orbhip_stereo_match_batch_device
    left._check(fn(left._h, right._h, nframes, d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap, mb, mbf,
                   d_uright, d_depth, d_nmatch, stream))


def extract_stereo_batch_device(left, right, d_imgs_l, d_imgs_r, nframes, w, h, row_stride, frame_stride,
                                d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap, mb, mbf,
                                d_uright, d_depth, d_nmatch=None, stream=None):
    """Both extractions and the stereo matching of a device-resident batch of pairs of any size (raw pointers)."""
    left._check(lib().orbhip_extract_stereo_batch_device(left._h, right._h, d_imgs_l, d_imgs_r, nframes, w, h, row_stride, frame_stride,
                                                         d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap, mb, mbf,
                                                         d_uright, d_depth, d_nmatch, stream))


def shard_range(nframes, nshards, index, group=1):
    """[lo, hi) of the frames shard `index` owns (orbhip_shard_range: the partition the multi-handle calls use)."""
    lo, hi = C.c_int(), C.c_int()
    rc = lib().orbhip_shard_range(nframes, nshards, index, group, C.byref(lo), C.byref(hi))
    if rc:
        raise OrbHipError(rc)
    return lo.value, hi.value


def extract_batch_multi(extractors, images):
    """operator() over a (F,H,W) uint8 host array, the frames sharded over several handles (one per GPU, or
    several on one GPU), one host thread per handle inside the library. Returns the per-frame results in order."""
    images = np.ascontiguousarray(images, np.uint8)
    f, h, w = images.shape
    cap = extractors[0].max_keypoints(w, h)
    kp = np.zeros((f, cap), KP_DTYPE)
    desc = np.zeros((f, cap, 32), np.uint8)
    counts = np.zeros(f, np.int32)
    handles = (C.c_void_p * len(extractors))(*[x._h for x in extractors])
    extractors[0]._check(lib().orbhip_extract_batch_multi(handles, len(extractors), _ptr(images), f, w, h, images.strides[1],
                                                          images.strides[0], _ptr(kp), _ptr(desc), _ptr(counts), cap))
    return [(kp[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(f)]


def extract_stereo_batch(left, right, images_l, images_r, mb, mbf):
    """Both extractions and Frame::ComputeStereoMatches for (P,H,W) host arrays of rectified pairs. `left` / `right` are
    one instance each, or equally long lists of instances (pairs sharded over them). Returns per pair
    (kp_l, desc_l, kp_r, desc_r, mvuRight, mvDepth, matches)."""
    images_l = np.ascontiguousarray(images_l, np.uint8); images_r = np.ascontiguousarray(images_r, np.uint8)
    assert images_l.shape == images_r.shape
    p, h, w = images_l.shape
    multi = isinstance(left, (list, tuple))
    first = left[0] if multi else left
    cap = first.max_keypoints(w, h)
    kl, kr = np.zeros((p, cap), KP_DTYPE), np.zeros((p, cap), KP_DTYPE)
    dl, dr = np.zeros((p, cap, 32), np.uint8), np.zeros((p, cap, 32), np.uint8)
    cl, cr, nm = np.zeros(p, np.int32), np.zeros(p, np.int32), np.zeros(p, np.int32)
    ur, dep = np.full((p, cap), -1.0, np.float32), np.full((p, cap), -1.0, np.float32)
    tail = (_ptr(images_l), _ptr(images_r), p, w, h, images_l.strides[1], images_l.strides[0], _ptr(kl), _ptr(dl), _ptr(cl),
            _ptr(kr), _ptr(dr), _ptr(cr), cap, mb, mbf, _ptr(ur), _ptr(dep), _ptr(nm))
    if multi:
        hl = (C.c_void_p * len(left))(*[x._h for x in left])
        hr = (C.c_void_p * len(right))(*[x._h for x in right])
        first._check(lib().orbhip_extract_stereo_batch_multi(hl, hr, len(left), *tail))
    else:
        first._check(lib().orbhip_extract_stereo_batch(left._h, right._h, *tail))
    return [(kl[i, :cl[i]].copy(), dl[i, :cl[i]].copy(), kr[i, :cr[i]].copy(), dr[i, :cr[i]].copy(),
             ur[i, :cl[i]].copy(), dep[i, :cl[i]].copy(), int(nm[i])) for i in range(p)]


class ORBextractor:
    """Drop-in counterpart of ORB_SLAM2::ORBextractor backed by libORBhip."""

    HARRIS_SCORE, FAST_SCORE = 0, 1   # include/ORBextractor.h:49

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0,
                 blur_variant=BLUR_LEGACY, max_batch=0):
        self._h = C.c_void_p()
        p = Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, blur_variant, max_batch)
        rc = lib().orbhip_create(C.byref(p), C.byref(self._h))
        if rc:
            self._h = None
            raise OrbHipError(rc)
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.mvImagePyramid = []

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:     # (at interpreter shutdown the module may already be gone)
            _lib.orbhip_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise OrbHipError(rc, lib().orbhip_last_error(self._h).decode())

    # ---- operator() ---------------------------------------------------------------------------
    def __call__(self, image, mask=None, fetch_pyramid=True):
        """operator()(image, mask, keypoints, descriptors): returns (keypoints, descriptors).

        An empty image returns (None, None), leaving the caller's previous outputs as they were,
        like the reference (src/ORBextractor.cc:1046-1047). The mask is ignored, as there.
        """
        if image is None or image.size == 0:
            return None, None
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("image must be 8-bit single channel (CV_8UC1)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints(w, h)
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(lib().orbhip_extract(self._h, _ptr(image), w, h, image.strides[0], _ptr(kp), cap, C.byref(n), _ptr(desc)))
        if fetch_pyramid:
            self.mvImagePyramid = [self.pyramid_level(l) for l in range(self.nlevels)]
        return kp[:n.value].copy(), desc[:n.value].copy()

    def extract_color(self, image, rgb, fetch_pyramid=True):
        """operator() on an (H, W, 3|4) uint8 image, with the cvtColor to gray of
        Tracking::GrabImage* (reference src/Tracking.cc:166-191) done on the device; rgb = mbRGB."""
        image = np.ascontiguousarray(image, np.uint8)
        h, w, cn = image.shape
        cap = self.max_keypoints(w, h)
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(lib().orbhip_extract_color(self._h, _ptr(image), w, h, image.strides[0], cn, int(bool(rgb)), _ptr(kp), cap,
                                               C.byref(n), _ptr(desc)))
        if fetch_pyramid:
            self.mvImagePyramid = [self.pyramid_level(l) for l in range(self.nlevels)]
        return kp[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """operator() over a (F,H,W) uint8 array of independent frames (host buffers)."""
        images = np.ascontiguousarray(images, np.uint8)
        f, h, w = images.shape
        cap = self.max_keypoints(w, h)
        kp = np.zeros((f, cap), KP_DTYPE)
        desc = np.zeros((f, cap, 32), np.uint8)
        counts = np.zeros(f, np.int32)
        self._check(lib().orbhip_extract_batch(self._h, _ptr(images), f, w, h, images.strides[1], images.strides[0],
                                               _ptr(kp), _ptr(desc), _ptr(counts), cap))
        return [(kp[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(f)]

    def extract_batch_device(self, d_imgs, nframes, w, h, row_stride, frame_stride, d_kp, d_desc, d_counts, cap, stream=None):
        """Raw-pointer batch call on device-resident frames; asynchronous on `stream`."""
        self._check(lib().orbhip_extract_batch_device(self._h, d_imgs, nframes, w, h, row_stride, frame_stride,
                                                      d_kp, d_desc, d_counts, cap, stream))

    def extract_batch_device_color(self, d_imgs, nframes, w, h, row_stride, frame_stride, channels, rgb, d_kp, d_desc, d_counts,
                                   cap, stream=None):
        """Raw-pointer batch call on device-resident interleaved colour frames (strides in bytes)."""
        self._check(lib().orbhip_extract_batch_device_color(self._h, d_imgs, nframes, w, h, row_stride, frame_stride,
                                                            channels, int(bool(rgb)), d_kp, d_desc, d_counts, cap, stream))

    def set_rectification(self, map_x=None, map_y=None):
        """cv::remap(im, imRect, M1, M2, INTER_LINEAR) in front of every extraction (reference
        Examples/Stereo/stereo_euroc.cc:97-98,136-137): map_x / map_y are (H, W) float32 arrays; None clears."""
        if map_x is None and map_y is None:
            self._check(lib().orbhip_set_rectification(self._h, None, None, 0, 0))
            return
        mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
        if mx.ndim != 2 or mx.shape != my.shape:
            raise ValueError("map_x and map_y must be 2-D arrays of the same shape")
        self._check(lib().orbhip_set_rectification(self._h, _ptr(mx), _ptr(my), mx.shape[1], mx.shape[0]))

    def synchronize(self):
        self._check(lib().orbhip_synchronize(self._h))

    def max_keypoints(self, width=None, height=None):
        """Output capacity to provide: for any camera format, or (with a size) for images of that size,
        which matters for wide strips (see include/orbhip.h)."""
        if width is None:
            return lib().orbhip_max_keypoints(self._h)
        return lib().orbhip_max_keypoints_for_image(self._h, int(width), int(height))

    # ---- accessors (include/ORBextractor.h:63-83) ----------------------------------------------
    def GetLevels(self):
        return lib().orbhip_get_levels(self._h)

    def GetScaleFactor(self):
        return lib().orbhip_get_scale_factor(self._h)

    def _table(self, fn):
        out = np.zeros(self.nlevels, np.float32)
        self._check(fn(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def GetScaleFactors(self):
        return self._table(lib().orbhip_get_scale_factors)

    def GetInverseScaleFactors(self):
        return self._table(lib().orbhip_get_inverse_scale_factors)

    def GetScaleSigmaSquares(self):
        return self._table(lib().orbhip_get_scale_sigma_squares)

    def GetInverseScaleSigmaSquares(self):
        return self._table(lib().orbhip_get_inverse_scale_sigma_squares)

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        self._check(lib().orbhip_get_features_per_level(self._h, out.ctypes.data_as(C.POINTER(C.c_int))))
        return out

    def umax(self):
        out = np.zeros(16, np.int32)
        self._check(lib().orbhip_get_umax(self._h, out.ctypes.data_as(C.POINTER(C.c_int))))
        return out

    # ---- mvImagePyramid --------------------------------------------------------------------------
    def pyramid_level(self, level, frame=0, with_border=False):
        w, h = C.c_int(), C.c_int()
        self._check(lib().orbhip_pyramid_info(self._h, level, C.byref(w), C.byref(h)))
        b = 38 if with_border else 0
        out = np.zeros((h.value + b, w.value + b), np.uint8)
        self._check(lib().orbhip_pyramid_copy(self._h, frame, level, _ptr(out), out.strides[0], int(with_border)))
        return out

    # ---- diagnostics ---------------------------------------------------------------------------
    def debug_candidates(self, level, frame=0):
        n = lib().orbhip_debug_candidates(self._h, frame, level, None, 0)
        if n < 0:
            raise OrbHipError(n)
        out = np.zeros(max(n, 1), CAND_DTYPE)
        lib().orbhip_debug_candidates(self._h, frame, level, _ptr(out), n)
        return out[:n]

    def debug_level_keypoints(self, level, frame=0, cap=65536):
        """(n, 3) uint16 array of x, y, score: what the quadtree left for `level`, in list order."""
        out = np.zeros((cap, 3), np.uint16)
        n = lib().orbhip_debug_level_keypoints(self._h, frame, level, _ptr(out), cap)
        if n < 0:
            raise OrbHipError(n)
        return out[:min(n, cap)]

    def enable_timing(self, on=True):
        self._check(lib().orbhip_enable_timing(self._h, int(on)))

    def timing(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        launches = (C.c_int * 16)()
        n = lib().orbhip_get_timing(self._h, names, ms, launches, 16)
        return {names[i].decode(): (ms[i], launches[i]) for i in range(n)}

    def timing_busy(self):
        """Per kernel class: milliseconds during which at least one of its launches was running (launches of a
        class can run side by side on two streams; `timing()` adds their durations up)."""
        names = (C.c_char_p * 16)()
        busy = (C.c_float * 16)()
        n = lib().orbhip_get_timing(self._h, names, None, None, 16)
        lib().orbhip_get_timing_busy(self._h, busy, 16)
        return {names[i].decode(): busy[i] for i in range(n)}

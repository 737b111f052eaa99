/*
 * orb_oracle — CPU restatement of the reference ORB extractor. TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker and the timed CPU baseline. It is never part of the
 * shipped extraction path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.
 *
 * It restates ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc, include/ORBextractor.h)
 * and the OpenCV 3.x generic C++ routines that class calls (resize INTER_LINEAR 8U,
 * copyMakeBorder REFLECT_101, FAST 9_16 + NMS, GaussianBlur 7x7 sigma 2, fastAtan2, cvRound).
 * OpenCV is not vendored by the reference and is absent from the build image, and the
 * reference has no tests or golden vectors, so this restatement is PARITY UNPINNED against a
 * real OpenCV build; it is pinned only by the analytic known-answer tests under tests/.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 28-byte record as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct orbo_keypoint {
    float x, y, size, angle, response;
    int octave, class_id;
} orbo_keypoint;

typedef struct orbo_candidate { /* one FAST corner handed to the quadtree, window-relative coords */
    int x, y, response;
} orbo_candidate;

typedef struct orbo_extractor orbo_extractor;

enum { ORBO_BLUR_LEGACY = 0, ORBO_BLUR_FIXEDPOINT = 1 };
enum { ORBO_TIE_CREATION_ORDER = 0, ORBO_TIE_HEAP_ADDRESS = 1 };
enum { ORBO_FMA_OFF = 0, ORBO_FMA_ON = 1 };

/* ORBextractor::ORBextractor (src/ORBextractor.cc:410-470). Returns NULL on bad arguments. */
orbo_extractor* orbo_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                            int min_th_fast, int blur_variant, int tie_mode, int fma_mode);
void orbo_destroy(orbo_extractor* e);

/* ORBextractor::operator() (src/ORBextractor.cc:1043-1105). kp/desc may be NULL to only count.
 * Returns 0, or a negative code: -1 empty image, -2 capacity too small, -3 unsupported geometry
 * (the reference divides by zero for these). *n receives the keypoint count. */
int orbo_extract(orbo_extractor* e, const uint8_t* img, int w, int h, size_t stride,
                 orbo_keypoint* kp, int cap, int* n, uint8_t* desc);

/* Tables (accessors at include/ORBextractor.h:63-83 and the protected vectors). */
int orbo_levels(const orbo_extractor* e);
float orbo_scale_factor(const orbo_extractor* e);
/* which: 0 mvScaleFactor, 1 mvInvScaleFactor, 2 mvLevelSigma2, 3 mvInvLevelSigma2 */
void orbo_table(const orbo_extractor* e, int which, float* out);
void orbo_features_per_level(const orbo_extractor* e, int* out);
void orbo_umax(const orbo_extractor* e, int* out16);

/* State of the last orbo_extract call on this instance. */
int orbo_level_size(const orbo_extractor* e, int level, int* w, int* h);
/* Copies level pixels; with_border != 0 copies the (w+38)x(h+38) bordered buffer. */
int orbo_level_copy(const orbo_extractor* e, int level, uint8_t* dst, size_t dst_stride, int with_border);
/* Blurred level (the image the descriptors were sampled from); 0 if the level had no keypoints. */
int orbo_blurred_copy(const orbo_extractor* e, int level, uint8_t* dst, size_t dst_stride);
/* Candidates handed to DistributeOctTree for a level, in the reference's order. Returns count. */
int orbo_level_candidates(const orbo_extractor* e, int level, orbo_candidate* out, int cap);
/* Keypoints of one level after DistributeOctTree, level coordinates. Returns count. */
int orbo_level_keypoints(const orbo_extractor* e, int level, orbo_keypoint* out, int cap);

/* Stand-alone stages, for known-answer tests. */
void orbo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride,
                           uint8_t* dst, int dw, int dh, size_t dstride);
void orbo_border_reflect101(const uint8_t* src, int w, int h, size_t sstride,
                            uint8_t* dst, size_t dstride, int border);
/* cv::FAST(img, kps, threshold, nonmax=true), TYPE_9_16. Returns count (may exceed cap). */
int orbo_fast9(const uint8_t* img, int w, int h, size_t stride, int threshold,
               orbo_candidate* out, int cap);
void orbo_gaussian_blur7(const uint8_t* src, int w, int h, size_t sstride,
                         uint8_t* dst, size_t dstride, int blur_variant);
float orbo_fast_atan2(float y, float x);
/* IC_Angle at (x, y) of a plain image; the 15-px disc must lie inside the image. */
float orbo_ic_angle(const uint8_t* img, int w, int h, size_t stride, int x, int y);
/* computeOrbDescriptor at (x, y) with the given angle (degrees) on an (already blurred) image;
 * the 19-px reach must lie inside the image. */
void orbo_descriptor(const uint8_t* img, int w, int h, size_t stride, int x, int y, float angle_deg,
                     int fma_mode, uint8_t* out32);
/* DistributeOctTree on a hand-built candidate set (coords relative to minX/minY). Returns count. */
int orbo_distribute(const orbo_candidate* cand, int ncand, int min_x, int max_x, int min_y, int max_y,
                    int n_wanted, int tie_mode, orbo_candidate* out, int cap);

/* CPU baseline: nthreads workers, one extractor instance each, frames dealt round-robin.
 * counts[nframes] receives per-frame keypoint counts; returns elapsed seconds (steady clock). */
double orbo_extract_batch_mt(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                             int min_th_fast, int blur_variant, const uint8_t* imgs, int nframes,
                             int w, int h, size_t row_stride, size_t frame_stride, int nthreads,
                             int* counts);

/* cv::cvtColor to gray of an interleaved 3- or 4-channel 8-bit image (src/Tracking.cc:166-191);
 * rgb != 0: channel 0 is red. OpenCV 2.4 .. 3.x 14-bit fixed point. */
void orbo_cvt_gray(const uint8_t* src, int w, int h, size_t sstride, int channels, int rgb,
                   uint8_t* dst, size_t dstride);

/* cv::undistortPoints with R = I, P = K (src/Frame.cc:438,463); cam = {fx, fy, cx, cy, k1, k2, p1, p2, k3},
 * points as interleaved x, y floats. */
void orbo_undistort_points(const float* cam, const float* src_xy, int n, float* dst_xy);
/* Frame::ComputeImageBounds (src/Frame.cc:450-478): bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}. */
void orbo_image_bounds(const float* cam, int width, int height, float* bounds);
/* Frame::UndistortKeyPoints + Frame::AssignFeaturesToGrid (src/Frame.cc:422-449, 255-270, 403-411);
 * grid_start has 64*48 + 1 entries, grid_idx up to n. */
void orbo_undistort_grid(const float* cam, const float* bounds, const orbo_keypoint* kp, int n,
                         orbo_keypoint* kp_un, int* grid_start, int* grid_idx);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:402-517) on the undistorted keypoints and
 * descriptors of an initial (1) and a current (2) frame; prev_matched = vbPrevMatched (in/out),
 * matches12 = vnMatches12 (out). Returns nmatches. */
int orbo_search_for_initialization(const orbo_keypoint* k1, const uint8_t* d1, int n1,
                                   const orbo_keypoint* k2, const uint8_t* d2, int n2, const float* bounds,
                                   float* prev_matched, int windowSize, float mfNNratio, int mbCheckOrientation,
                                   int* matches12);

/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1447-1564). */
int orbo_search_by_projection_keyframe(const orbo_keypoint* k_kf, const float* world_kf, const float* dist_range_kf,
                                       const uint8_t* flags_kf, const uint8_t* desc_kf, int n1, const orbo_keypoint* k_cur,
                                       const uint8_t* desc_cur, const uint8_t* occupied_cur, int n2, const float* Tcw,
                                       const float* cam, const float* bounds, const float* scale_factors, float scale_factor,
                                       int nlevels, float th, int ORBdist, int mbCheckOrientation, int* match_cur);
/* Frame::UnprojectStereo (src/Frame.cc:674-687) for arrays of keypoints. */
void orbo_unproject_stereo(const orbo_keypoint* kp_un, const float* depth, int n, const float* pose_wc, const float* cam,
                           float* world, uint8_t* valid);
/* Frame::ComputeStereoFromRGBD (src/Frame.cc:655-672) with the depth conversion of Tracking::GrabImageRGBD. */
void orbo_stereo_from_rgbd(const orbo_keypoint* kp, const orbo_keypoint* kp_un, int n, const void* depth, int depth_u16, int width,
                           int height, size_t row_stride, float mDepthMapFactor, float mbf, float* mvuRight, float* mvDepth);
/* ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:287-400). */
int orbo_search_by_projection_sim3(const float* world, const float* normal, const float* dist_range, const uint8_t* flags,
                                   const uint8_t* desc, int n1, const orbo_keypoint* k_kf, const uint8_t* desc_kf,
                                   const uint8_t* occupied_kf, int n2, const float* Scw, const float* cam, const float* bounds,
                                   const float* scale_factors, float scale_factor, int nlevels, int th, int* match_kf);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:156-285). */
int orbo_search_by_bow(const orbo_keypoint* k_kf, const uint8_t* desc_kf, const uint32_t* node_kf, const uint8_t* flags_kf, int n1,
                       const orbo_keypoint* k_f, const uint8_t* desc_f, const uint32_t* node_f, int n2, float mfNNratio,
                       int mbCheckOrientation, int* match_f);
/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (src/ORBmatcher.cc:519-652). */
int orbo_search_by_bow_keyframes(const orbo_keypoint* k1, const uint8_t* desc1, const uint32_t* node1, const uint8_t* flags1, int n1,
                                 const orbo_keypoint* k2, const uint8_t* desc2, const uint32_t* node2, const uint8_t* flags2, int n2,
                                 float mfNNratio, int mbCheckOrientation, int* match12);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:46-127). */
typedef struct orbo_track_point { float proj_x, proj_y, proj_xr, view_cos; int32_t level, flags; } orbo_track_point;
int orbo_search_map_points(const orbo_track_point* mp, const uint8_t* desc_mp, int n1, const orbo_keypoint* k,
                           const uint8_t* desc, const float* uright, const uint8_t* occupied, int n2,
                           const float* bounds, const float* scale_factors, int nlevels, float th, float mfNNratio, int* match);
/* Frame::isInFrustum (src/Frame.cc:294-350) for arrays of map points. */
typedef struct orbo_map_point { float pos[3], normal[3], min_distance, max_distance; int32_t flags; } orbo_map_point;
int orbo_is_in_frustum(const orbo_map_point* mp, int n, const float* pose, const float* cam, float mbf,
                       const float* bounds, float scale_factor, int nlevels, float viewingCosLimit, orbo_track_point* track);
/* The host C library's cosf / sinf (what src/ORBextractor.cc:113-114 calls) for `count` consecutive
 * float bit patterns. */
void orbo_libm_logf(uint32_t first_bits, uint32_t count, uint32_t stride, float* out);
void orbo_libm_sincos(uint32_t first_bits, uint32_t count, float* out_cos, float* out_sin);
/* cv::remap(src, dst, map1, map2, INTER_LINEAR) with constant border 0 for an 8-bit gray image and
 * CV_32FC1 maps of size dw x dh (Examples/Stereo/stereo_euroc.cc:136-137). */
void orbo_remap_linear(const uint8_t* src, int sw, int sh, size_t sstride, const float* map_x, const float* map_y,
                       int dw, int dh, uint8_t* dst, size_t dstride);

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1324-1445) on
 * plain arrays; see orb_oracle.cpp for the meaning of every argument. Returns nmatches. */
int orbo_search_by_projection(const orbo_keypoint* k_last, const float* world_last, const uint8_t* flags_last,
                              const uint8_t* desc_last, int n1, const orbo_keypoint* k_cur, const uint8_t* desc_cur,
                              const float* uright_cur, int n2, const float* Tcw, const float* cam, float mbf,
                              const float* bounds, const float* scale_factors, float th, int level_mode,
                              int mbCheckOrientation, int* match_cur);
/* bForward / bBackward of the same function (src/ORBmatcher.cc:1330-1337): 0 neither, 1 forward, 2 backward. */
int orbo_projection_level_mode(const float* Tcw_cur, const float* Tcw_last, float mb, int bMono);

/* Frame::ComputeStereoMatches (reference src/Frame.cc:481-653) on the last extractions of a left and
 * a right instance: kl/dl/nl and kr/dr/nr are their keypoints and descriptors, mb / mbf the stereo
 * baseline (metres) and baseline times fx. uright[nl] / depth[nl] receive mvuRight / mvDepth
 * (-1 = none). Returns the number of matches kept. */
int orbo_stereo_match(const orbo_extractor* L, const orbo_extractor* R,
                      const orbo_keypoint* kl, const uint8_t* dl, int nl,
                      const orbo_keypoint* kr, const uint8_t* dr, int nr,
                      float mb, float mbf, float* uright, float* depth);

#ifdef __cplusplus
}
#endif
#endif

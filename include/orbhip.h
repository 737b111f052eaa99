/*
 * orbhip.h — C ABI of libORBhip, the MI355X (gfx950) ORB feature extractor.
 *
 * The reference has no FFI layer: its boundary for this path is the C++ class
 * ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111). Every entry point below is
 * what a thin C++ class with that exact interface binds to (shim/ORBextractor.{h,cc} is that
 * class); each declaration cites the reference member it replaces.
 *
 * Conventions: plain pointers and sizes, int status codes (0 = ORBHIP_OK, negative = error),
 * no exceptions cross the ABI. A handle owns one HIP stream and all device workspaces; distinct
 * handles may be used concurrently from different threads (the reference runs two extractor
 * instances on two threads for stereo, src/Frame.cc:114-117), one handle may not. Every entry point
 * makes its handle's device current for the duration of the call and restores the calling thread's
 * device before it returns.
 */
#ifndef ORBHIP_H
#define ORBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBHIP_OK 0
#define ORBHIP_ERR_ARG (-1)          /* null pointer / bad parameter */
#define ORBHIP_ERR_EMPTY (-2)        /* empty image: outputs untouched (src/ORBextractor.cc:1046-1047) */
#define ORBHIP_ERR_CAPACITY (-3)     /* output or internal candidate capacity exceeded */
#define ORBHIP_ERR_GEOMETRY (-4)     /* image shape the reference itself cannot process (division by zero at src/ORBextractor.cc:543-545) or larger than 16384 px */
#define ORBHIP_ERR_HIP (-5)          /* a HIP runtime call failed; see orbhip_last_error() */
#define ORBHIP_ERR_NO_DEVICE (-6)    /* no usable gfx950 device */
#define ORBHIP_ERR_STATE (-7)        /* call not valid before an extraction */

/* Blur arithmetic of cv::GaussianBlur(7x7, sigma 2) (src/ORBextractor.cc:1086), which depends on
 * the OpenCV version the reference is linked with. */
#define ORBHIP_BLUR_LEGACY 0         /* OpenCV 2.4 .. 3.4.1: taps {18,34,49,55,49,34,18}/256 */
#define ORBHIP_BLUR_FIXEDPOINT 1     /* OpenCV >= 3.4.2: taps forced to sum 256, round half up */

/* Same layout as cv::KeyPoint (28 bytes): the C++ class can memcpy these into its vector. */
typedef struct orbhip_keypoint {
    float x, y;      /* pt, level coordinates times mvScaleFactor[octave] (src/ORBextractor.cc:1095-1101) */
    float size;      /* int(31 * mvScaleFactor[octave])                  (src/ORBextractor.cc:837,846) */
    float angle;     /* intensity-centroid angle, degrees in [0,360)     (src/ORBextractor.cc:77-104) */
    float response;  /* FAST corner score                                (cv::FAST, src/ORBextractor.cc:809) */
    int octave;      /* pyramid level                                    (src/ORBextractor.cc:845) */
    int class_id;    /* -1 */
} orbhip_keypoint;

typedef struct orbhip_params {
    int nfeatures;        /* ORBextractor.nFeatures  */
    float scale_factor;   /* ORBextractor.scaleFactor */
    int nlevels;          /* ORBextractor.nLevels    */
    int ini_th_fast;      /* ORBextractor.iniThFAST  */
    int min_th_fast;      /* ORBextractor.minThFAST  */
    int device;           /* HIP device ordinal */
    int blur_variant;     /* ORBHIP_BLUR_* */
    int max_batch;        /* frames processed per internal pass of the batch entry points (0 = default: 512, more of images smaller than 1920 x 1080, as many pixels per pass; also capped by a ~24 GB workspace limit) */
} orbhip_params;

typedef struct orbhip_extractor orbhip_extractor;

/* ORBextractor::ORBextractor (src/ORBextractor.cc:410-470): scale tables, per-level feature
 * quotas, the rBRIEF pattern and the circular-patch row limits are computed here once. */
int orbhip_create(const orbhip_params* params, orbhip_extractor** out);
/* ORBextractor::~ORBextractor (include/ORBextractor.h:54). */
int orbhip_destroy(orbhip_extractor* e);

/* ORBextractor::operator()(image, mask, keypoints, descriptors) (src/ORBextractor.cc:1043-1105)
 * for one 8-bit gray host image. `kp` (capacity `cap` records) and `desc` (cap x 32 bytes) are
 * host buffers; *n receives the keypoint count. The mask argument of the reference is ignored
 * there (include/ORBextractor.h:58) and has no counterpart here. An empty image returns
 * ORBHIP_ERR_EMPTY and leaves the outputs untouched. From the second call with the same image size
 * and capacity on, the copy up, the kernel launches and the copies down are replayed as one captured
 * HIP graph (the call is launch-bound at camera-frame sizes); ORBHIP_GRAPH=0 in the environment, read
 * by orbhip_create, keeps the plain launches. */
int orbhip_extract(orbhip_extractor* e, const uint8_t* img, int width, int height, size_t stride,
                   orbhip_keypoint* kp, int cap, int* n, uint8_t* desc);

/* The same operator() applied to `nframes` independent frames that already live in device memory
 * (frame f starts at d_imgs + f*frame_stride, rows `row_stride` bytes apart). Outputs are device
 * buffers: d_kp[nframes][cap], d_desc[nframes][cap][32], d_counts[nframes]. Work is enqueued on
 * `stream` (a hipStream_t, NULL = the handle's own stream) and the call returns without waiting;
 * frames beyond max_batch are processed in consecutive passes. This is the entry point the
 * multi-GPU batch driver shards over (one handle per device, no collective).
 * Level 0 in place: the reference copies the image into a bordered level-0 buffer
 * (src/ORBextractor.cc:1113-1128); when d_imgs, row_stride and frame_stride are multiples of 4 and
 * width is too, the image itself serves as level 0 and the copy is skipped (same results; the
 * environment variable ORBHIP_INPLACE_LEVEL0=0, read by orbhip_create, forces the copy). The images
 * must then stay unchanged not only until the enqueued work has run, as for any asynchronous call, but
 * for as long as level 0 of this pass is used: a later orbhip_stereo_match_batch_device* on this handle
 * and orbhip_pyramid_copy read it from d_imgs. */
int orbhip_extract_batch_device(orbhip_extractor* e, const uint8_t* d_imgs, int nframes, int width,
                                int height, size_t row_stride, size_t frame_stride,
                                orbhip_keypoint* d_kp, uint8_t* d_desc, int* d_counts, int cap,
                                void* stream);

/* Host-buffer variant of the batch call: what Frame::ExtractORB (src/Frame.cc:272-278) does per image, for
 * `nframes` images in host memory; waits for the results. Batches of 8 frames and more run as a pipeline of
 * chunks (about 64 MB of images each, ORBHIP_HOST_CHUNK=n frames overrides) over three internal streams, one
 * per stage: while one chunk is packed into page-locked memory by a few host threads (ORBHIP_PACK_THREADS,
 * default half of the CPUs the process may use, at most 4), the previous one travels to the device, the one
 * before is extracted and the results of the one before that travel back and are handed to the caller's arrays
 * by helper threads (ORBHIP_HOST_LANES=2..4 chunks in flight, default 3; ORBHIP_HOST_TRACE=1 prints the timeline
 * of a call on stderr). Images that already lie in page-locked memory (hipHostMalloc / hipHostRegister) with rows
 * of the staging pitch (the width rounded up to 16 bytes, frames one behind the other) are copied from where
 * they are, without the packing step. ORBHIP_HOST_PIPELINE=0 (read by orbhip_create) restores one copy up, all passes, one copy down.
 * After the call orbhip_pyramid_copy sees the frames of the last chunk only. */
int orbhip_extract_batch(orbhip_extractor* e, const uint8_t* imgs, int nframes, int width, int height,
                         size_t row_stride, size_t frame_stride, orbhip_keypoint* kp, uint8_t* desc,
                         int* counts, int cap);

/* Several GPUs (or several handles on one GPU) from one process. The reference runs its two extractor instances
 * on two threads inside one process (src/Frame.cc:114-117); this is the same arrangement for a batch: the frames
 * are cut into `nhandles` contiguous ranges (orbhip_shard_range), handle i extracts range i on its own device
 * with its own host thread, and every thread writes its frames' results where they belong in the caller's
 * arrays: no collective, no exchange between devices, a host-side gather only. Arguments and results are those
 * of orbhip_extract_batch (gray images); the handles must be distinct and configured alike. Returns the first
 * failing handle's status (its message through orbhip_last_error(handles[0])). Like every entry point of this
 * library the call leaves the calling thread's current HIP device as it found it. */
int orbhip_extract_batch_multi(orbhip_extractor* const* handles, int nhandles, const uint8_t* imgs, int nframes,
                               int width, int height, size_t row_stride, size_t frame_stride,
                               orbhip_keypoint* kp, uint8_t* desc, int* counts, int cap);
/* The range [*lo, *hi) of `nframes` frames that shard `index` of `nshards` owns: contiguous, sizes differing by at
 * most one group of `group` frames (group = 2 keeps the two images of an interleaved stereo pair together). */
int orbhip_shard_range(int nframes, int nshards, int index, int group, int* lo, int* hi);

/* The same two calls for interleaved 8-bit colour images: the cvtColor to gray that precedes the
 * extraction in the reference (Tracking::GrabImageStereo / GrabImageRGBD / GrabImageMonocular,
 * src/Tracking.cc:166-191,211-222,249-262: CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY)
 * happens on the device while the level-0 plane is written, in the 8-bit arithmetic of OpenCV
 * 2.4 .. 3.x (gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14). channels is 3 or 4 (alpha ignored),
 * rgb != 0 means channel 0 is red (mbRGB, src/Tracking.cc:126-131). Strides are in bytes. */
int orbhip_extract_color(orbhip_extractor* e, const uint8_t* img, int width, int height, size_t stride,
                         int channels, int rgb, orbhip_keypoint* kp, int cap, int* n, uint8_t* desc);
int orbhip_extract_batch_device_color(orbhip_extractor* e, const uint8_t* d_imgs, int nframes, int width,
                                      int height, size_t row_stride, size_t frame_stride, int channels, int rgb,
                                      orbhip_keypoint* d_kp, uint8_t* d_desc, int* d_counts, int cap, void* stream);

/* Waits for everything enqueued by the handle; reports a deferred capacity/geometry error. */
int orbhip_synchronize(orbhip_extractor* e);

/* Upper bound of keypoints one frame can yield (sum over levels of quota + slack of the quadtree,
 * src/ORBextractor.cc:669-673,728-731): size `cap` with it. Holds for images whose levels have at most 16
 * initial quadtree nodes each (round(width / height) of a level's detection window, src/ORBextractor.cc:543:
 * every camera format); the first round of the distribution yields up to 4 nodes per initial node whatever
 * the quota (:594-665), so wide strips can exceed it: */
int orbhip_max_keypoints(const orbhip_extractor* e);
/* The bound for images of one size (any aspect ratio the extractor accepts). */
int orbhip_max_keypoints_for_image(const orbhip_extractor* e, int width, int height);

/* Accessors (include/ORBextractor.h:63-83). Array getters fill nlevels floats. */
int orbhip_get_levels(const orbhip_extractor* e);                               /* GetLevels */
float orbhip_get_scale_factor(const orbhip_extractor* e);                       /* GetScaleFactor */
int orbhip_get_scale_factors(const orbhip_extractor* e, float* out);            /* GetScaleFactors */
int orbhip_get_inverse_scale_factors(const orbhip_extractor* e, float* out);    /* GetInverseScaleFactors */
int orbhip_get_scale_sigma_squares(const orbhip_extractor* e, float* out);      /* GetScaleSigmaSquares */
int orbhip_get_inverse_scale_sigma_squares(const orbhip_extractor* e, float* out); /* GetInverseScaleSigmaSquares */
int orbhip_get_features_per_level(const orbhip_extractor* e, int* out);         /* mnFeaturesPerLevel (include/ORBextractor.h:103) */
int orbhip_get_umax(const orbhip_extractor* e, int* out16);                     /* umax (include/ORBextractor.h:105) */

/* mvImagePyramid (include/ORBextractor.h:85; filled by ComputePyramid, src/ORBextractor.cc:1107-1132).
 * Valid after an extraction, until the next one on this handle. `frame` indexes the frames of the
 * last pass (0 for orbhip_extract). with_border != 0 copies the (w+38)x(h+38) buffer whose inner
 * ROI the reference exposes. dst is a host buffer. */
int orbhip_pyramid_info(const orbhip_extractor* e, int level, int* width, int* height);
int orbhip_pyramid_copy(orbhip_extractor* e, int frame, int level, uint8_t* dst, size_t dst_stride,
                        int with_border);

/* Diagnostics for parity tests: the FAST corners handed to the quadtree for one level of one frame
 * of the last pass (x, y relative to the detection window origin (16,16); response = score), in
 * the reference's order (cell row, cell column, then raster; src/ORBextractor.cc:789-829).
 * Returns the count (which may exceed cap) or a negative error. */
typedef struct orbhip_candidate { int x, y, response; } orbhip_candidate;
int orbhip_debug_candidates(orbhip_extractor* e, int frame, int level, orbhip_candidate* out, int cap);
/* Diagnostic: the keypoints the quadtree left for one level of one frame of the last pass, in list order
 * (the output of DistributeOctTree, src/ORBextractor.cc:539-763, before orientation and descriptors):
 * x, y (level coordinates) and FAST score per keypoint, three 16-bit values each. Returns the count
 * (which may exceed cap), or a negative status. */
int orbhip_debug_level_keypoints(orbhip_extractor* e, int frame, int level, uint16_t* out_xys, int cap);

/* Diagnostics for parity tests: the cosine and sine the descriptor kernel uses for its rotation (the
 * reference calls cosf / sinf of the keypoint angle in radians, src/ORBextractor.cc:113-114) for the
 * `count` floats whose bit patterns start at first_bits, written to device arrays; asynchronous on
 * `stream` (NULL: the handle's). A test compares them with the host C library over [0, 2*pi]. */
int orbhip_debug_sincos(orbhip_extractor* e, uint32_t first_bits, uint32_t count, float* d_cos, float* d_sin, void* stream);

/* Diagnostics for parity tests: the logarithm orbhip_is_in_frustum uses for MapPoint::PredictScale (the
 * reference calls log() on a float, src/MapPoint.cc:409) for the `count` positive normal floats whose
 * bit patterns are first_bits + i * stride, written to a device array. */
int orbhip_debug_logf(orbhip_extractor* e, uint32_t first_bits, uint32_t count, uint32_t stride, float* d_out, void* stream);

/* Per-kernel device time of the last batch call, measured with HIP events on the stream the
 * kernels ran on. Enable first; names[i] / ms[i] are filled up to `cap` entries; returns the
 * number of kernel classes. Enabling adds event records (and a wait in this call) only. */
int orbhip_enable_timing(orbhip_extractor* e, int on);
int orbhip_get_timing(orbhip_extractor* e, const char** names, float* ms, int* launches, int cap);
/* The same classes: the time during which at least one launch of the class was running. Launches of one class
 * can run side by side on two internal streams (FAST follows the resize chain level by level), and then the sum
 * that orbhip_get_timing reports counts that time more than once. (No counterpart in the reference.) */
int orbhip_get_timing_busy(orbhip_extractor* e, float* busy_ms, int cap);

/* Rectification in front of the extraction: cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) as the
 * stereo mains apply it to both images of every frame (Examples/Stereo/stereo_euroc.cc:136-137,
 * Examples/ROS/ORB_SLAM2/src/ros_stereo.cc:161-162) with the CV_32F maps of
 * cv::initUndistortRectifyMap (stereo_euroc.cc:97-98). map_x / map_y are host arrays of
 * height x width floats (M1, M2: source column and row of every rectified pixel). Once set, every
 * extraction call on this handle takes a gray image of that size, rectifies it on the way into
 * pyramid level 0 (constant border 0 outside the source, as cv::remap's defaults) and works on the
 * result; colour input and other sizes are refused. NULL, NULL clears the maps. Waits for the
 * device; not to be called while a batch of this handle is in flight. */
int orbhip_set_rectification(orbhip_extractor* e, const float* map_x, const float* map_y, int width, int height);

/* Frame::ComputeStereoMatches (src/Frame.cc:481-653): for every left keypoint the right-image
 * column of its match (mvuRight) and its depth (mvDepth), -1 where there is none. `left` and
 * `right` are the two extractor instances of a stereo rig (mpORBextractorLeft / Right,
 * src/Frame.cc:114-117) after they have extracted the two images of the same `nframes` pairs in
 * one pass each (same image size, same parameters, same device); their pyramids are read where
 * the extraction left them in device memory (the reference reads mvImagePyramid of both,
 * src/Frame.cc:488,578,590,595). d_kp_* / d_desc_* / d_cnt_* are the outputs of
 * orbhip_extract_batch_device ([nframes][cap] records, counts per frame); mb is the baseline in
 * metres and mbf the baseline times fx (src/Frame.cc:144,511-513). d_uright / d_depth are
 * [nframes][cap] floats, d_nmatch[nframes] receives the matches kept per pair (may be NULL).
 * Enqueued on `stream` (NULL = left's own stream), which must be the stream the two extractions
 * were enqueued on or be ordered after them. */
int orbhip_stereo_match_batch_device(orbhip_extractor* left, orbhip_extractor* right, int nframes,
                                     const orbhip_keypoint* d_kp_l, const uint8_t* d_desc_l, const int* d_cnt_l,
                                     const orbhip_keypoint* d_kp_r, const uint8_t* d_desc_r, const int* d_cnt_r,
                                     int cap, float mb, float mbf, float* d_uright, float* d_depth, int* d_nmatch,
                                     void* stream);

/* The same matching for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): it has no counterpart in the reference. Arguments and
 * results are those of orbhip_stereo_match_batch_device; the kernels are the same. Its scratch
 * (right-image records, per-keypoint SAD values, row-bin offsets: about 10.6 MB) is a static array in
 * the library's device image, one copy per GPU, whose address is taken once per device when a
 * handle is created there; the call itself allocates and frees nothing. Batches beyond the limits
 * below are refused with ORBHIP_ERR_CAPACITY. All calls on one GPU share that GPU's array (the
 * kernels overwrite every element they read, nothing has to be cleared between calls): calls that
 * overlap in time on one GPU race, also from different handles or streams. Per-kernel timing
 * (orbhip_enable_timing) does not cover this call. */
#define ORBHIP_NOALLOC_MAX_PAIRS 4096              /* nframes */
#define ORBHIP_NOALLOC_MAX_SLOTS (1 << 19)         /* nframes * cap */
#define ORBHIP_NOALLOC_MAX_BIN_ENTRIES (1 << 17)   /* nframes * ((image height >> 4) + 3) */
int orbhip_stereo_match_batch_device_noalloc(orbhip_extractor* left, orbhip_extractor* right, int nframes,
                                     const orbhip_keypoint* d_kp_l, const uint8_t* d_desc_l, const int* d_cnt_l,
                                     const orbhip_keypoint* d_kp_r, const uint8_t* d_desc_r, const int* d_cnt_r,
                                     int cap, float mb, float mbf, float* d_uright, float* d_depth, int* d_nmatch,
                                     void* stream);
/* Extraction of both images of `nframes` rectified pairs and their stereo matching in one call: what the
 * stereo constructor of Frame does per pair (two ExtractORB calls, src/Frame.cc:114-117, then
 * ComputeStereoMatches, src/Frame.cc:481-653), for a device-resident batch of any size. A batch larger than
 * one internal pass is processed pass by pass (left pass, right pass, matching of those pairs), so the
 * matcher always finds the pyramids of the pairs it works on. Gray images, both sides with the same size
 * and strides; outputs as in orbhip_extract_batch_device (both sides) and orbhip_stereo_match_batch_device.
 * Asynchronous on `stream` (NULL = left's own stream). */
int orbhip_extract_stereo_batch_device(orbhip_extractor* left, orbhip_extractor* right,
                                       const uint8_t* d_imgs_l, const uint8_t* d_imgs_r, int nframes, int width, int height,
                                       size_t row_stride, size_t frame_stride,
                                       orbhip_keypoint* d_kp_l, uint8_t* d_desc_l, int* d_cnt_l,
                                       orbhip_keypoint* d_kp_r, uint8_t* d_desc_r, int* d_cnt_r, int cap,
                                       float mb, float mbf, float* d_uright, float* d_depth, int* d_nmatch, void* stream);
/* The same for pairs in host memory, results into host arrays ([npairs][cap] records per side, [npairs][cap]
 * floats of mvuRight / mvDepth, nmatch[npairs] or NULL); waits for the results. From 8 pairs on it runs as the
 * chunk pipeline of orbhip_extract_batch (a chunk: pairs whose two sides together are about 64 MB of images; the
 * same environment switches). Afterwards a separate matching call finds the pyramids of the last chunk only. */
int orbhip_extract_stereo_batch(orbhip_extractor* left, orbhip_extractor* right, const uint8_t* imgs_l, const uint8_t* imgs_r,
                                int npairs, int width, int height, size_t row_stride, size_t frame_stride,
                                orbhip_keypoint* kp_l, uint8_t* desc_l, int* cnt_l, orbhip_keypoint* kp_r, uint8_t* desc_r, int* cnt_r,
                                int cap, float mb, float mbf, float* uright, float* depth, int* nmatch);
/* ... sharded over `nhandles` pairs of instances (left[i], right[i] on one device each) like
 * orbhip_extract_batch_multi: contiguous ranges of pairs, both images of a pair on the same device (the matcher
 * reads both pyramids there), one host thread per pair of handles, no exchange between devices. */
int orbhip_extract_stereo_batch_multi(orbhip_extractor* const* left, orbhip_extractor* const* right, int nhandles,
                                      const uint8_t* imgs_l, const uint8_t* imgs_r, int npairs, int width, int height,
                                      size_t row_stride, size_t frame_stride,
                                      orbhip_keypoint* kp_l, uint8_t* desc_l, int* cnt_l, orbhip_keypoint* kp_r, uint8_t* desc_r, int* cnt_r,
                                      int cap, float mb, float mbf, float* uright, float* depth, int* nmatch);
/* The same for one pair with host arrays (the last orbhip_extract of each instance): kp/desc as
 * returned by orbhip_extract, uright / depth receive n_l floats. Waits for the result. */
int orbhip_stereo_match(orbhip_extractor* left, orbhip_extractor* right,
                        const orbhip_keypoint* kp_l, const uint8_t* desc_l, int n_l,
                        const orbhip_keypoint* kp_r, const uint8_t* desc_r, int n_r,
                        float mb, float mbf, float* uright, float* depth, int* nmatch);

/* Camera of a Frame: mK (fx, fy, cx, cy) and mDistCoef (k1, k2, p1, p2, k3), the float values
 * Tracking::Tracking reads from the settings file (src/Tracking.cc:50-75). */
typedef struct orbhip_camera { float fx, fy, cx, cy, k1, k2, p1, p2, k3; } orbhip_camera;

#define ORBHIP_GRID_COLS 64   /* FRAME_GRID_COLS, include/Frame.h:38 */
#define ORBHIP_GRID_ROWS 48   /* FRAME_GRID_ROWS, include/Frame.h:37 */

/* Frame::ComputeImageBounds (src/Frame.cc:450-478): bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY} of a
 * width x height image (the four corners through cv::undistortPoints when k1 != 0). Host only. */
int orbhip_image_bounds(const orbhip_camera* cam, int width, int height, float bounds[4]);

/* Frame::UndistortKeyPoints (src/Frame.cc:422-449) and Frame::AssignFeaturesToGrid (src/Frame.cc:255-270
 * with PosInGrid :403-411) for the keypoints of `nframes` frames in device memory (the outputs of
 * orbhip_extract_batch_device). d_kp_un[nframes][cap] receives mvKeysUn (records with pt replaced;
 * a copy when k1 == 0, as in the reference). The grid comes back in compressed form: cell
 * mGrid[x][y] of frame f holds d_grid_idx[f][s .. e) with s = d_grid_start[f][x*48 + y] and
 * e = d_grid_start[f][x*48 + y + 1] (64*48 + 1 entries per frame), keypoint indices ascending as
 * the reference pushes them. Enqueued on `stream` (NULL = the handle's own). */
int orbhip_undistort_grid_batch_device(orbhip_extractor* e, int nframes, const orbhip_keypoint* d_kp,
                                       const int* d_counts, int cap, const orbhip_camera* cam,
                                       const float bounds[4], orbhip_keypoint* d_kp_un, int* d_grid_start,
                                       int* d_grid_idx, void* stream);
/* The same for one frame with host arrays (n keypoints; grid_start has 64*48 + 1 entries, grid_idx n). */
int orbhip_undistort_grid(orbhip_extractor* e, const orbhip_keypoint* kp, int n, const orbhip_camera* cam,
                          const float bounds[4], orbhip_keypoint* kp_un, int* grid_start, int* grid_idx);

/* Frame::ComputeStereoFromRGBD (src/Frame.cc:655-672) for frames in device memory, with the depth scaling
 * of Tracking::GrabImageRGBD in front of it (imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) when the
 * factor differs from 1 or the map is not CV_32F): d_depth holds nframes depth maps (unsigned 16-bit
 * when depth_is_u16, else float; strides in bytes), d_kp the raw keypoints (mvKeys: the depth is read at
 * their truncated position), d_kp_un the undistorted ones (mvKeysUn). d_uright / d_depth_out
 * [nframes][cap] receive mvuRight and mvDepth (-1 where the depth is not positive). A keypoint outside
 * the map gets -1 (the reference would read out of bounds). Asynchronous on `stream`. */
int orbhip_stereo_from_rgbd_batch_device(orbhip_extractor* e, int nframes, const orbhip_keypoint* d_kp, const orbhip_keypoint* d_kp_un,
                                         const int* d_counts, int cap, const void* d_depth, int depth_is_u16, int width, int height,
                                         size_t row_stride, size_t frame_stride, float depth_map_factor, float mbf,
                                         float* d_uright, float* d_depth_out, void* stream);

/* Frame::UnprojectStereo (src/Frame.cc:674-687) for every keypoint of nframes frames in device memory:
 * d_kp_un = mvKeysUn, d_depth = mvDepth ([nframes][cap], as the stereo / RGB-D calls above write it),
 * d_pose_wc = per frame mRwc row-major (9 floats) and mOw (3). d_world[nframes][cap][3] receives the world
 * positions, d_valid[nframes][cap] 1 where the depth is positive (the reference returns an empty Mat
 * otherwise; the position is zeroed there). flags for orbhip_search_by_projection_batch_device can be
 * built from d_valid. Asynchronous on `stream`. */
int orbhip_unproject_stereo_batch_device(orbhip_extractor* e, int nframes, const orbhip_keypoint* d_kp_un, const float* d_depth,
                                         const int* d_counts, int cap, const float* d_pose_wc, float fx, float fy, float cx, float cy,
                                         float* d_world, uint8_t* d_valid, void* stream);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
 * (src/ORBmatcher.cc:402-517; called with ORBmatcher(0.9, true) and windowSize 100 at
 * src/Tracking.cc:588-589) for `nframes` frame pairs in device memory. d_kp_un1 / d_kp_un2 are the
 * undistorted keypoints of the initial and of the current frames (mvKeysUn, e.g. from
 * orbhip_undistort_grid_batch_device), d_desc* / d_cnt* their descriptors and counts, bounds the
 * image bounds of the frames' grid (orbhip_image_bounds). d_prev_matched[nframes][cap][2] holds
 * vbPrevMatched (x, y per F1 keypoint) and is updated as the reference updates it;
 * d_matches12[nframes][cap] receives vnMatches12 (-1 = none), d_nmatches[nframes] the return value. */
int orbhip_search_for_initialization_batch_device(orbhip_extractor* e, int nframes,
                                                  const orbhip_keypoint* d_kp_un1, const uint8_t* d_desc1, const int* d_cnt1,
                                                  const orbhip_keypoint* d_kp_un2, const uint8_t* d_desc2, const int* d_cnt2,
                                                  int cap, const float bounds[4], float* d_prev_matched, int window_size,
                                                  float nnratio, int check_orientation, int* d_matches12, int* d_nmatches,
                                                  void* stream);
/* The same for one pair with host arrays; prev_matched has n1 (x, y) pairs, matches12 n1 ints. */
int orbhip_search_for_initialization(orbhip_extractor* e, const orbhip_keypoint* kp_un1, const uint8_t* desc1, int n1,
                                     const orbhip_keypoint* kp_un2, const uint8_t* desc2, int n2, const float bounds[4],
                                     float* prev_matched, int window_size, float nnratio, int check_orientation,
                                     int* matches12, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1324-1445), the matcher of Tracking::TrackWithMotionModel (src/Tracking.cc:1332,1337),
 * for `nframes` (last, current) frame pairs in device memory. Frames and map points stay with the
 * caller; the call sees them as arrays of [nframes][cap] records:
 *   last frame     d_kp_last     mvKeysUn (octave and angle are read)
 *                  d_world_last  GetWorldPos() of mvpMapPoints[i], 3 floats
 *                  d_flags_last  bit 0: mvpMapPoints[i] exists and mvbOutlier[i] is false;
 *                                bit 1: that point has Observations() > 0
 *                  d_desc_last   GetDescriptor() of the point, 32 bytes
 *   current frame  d_kp_cur      mvKeysUn;  d_desc_cur  mDescriptors;
 *                  d_uright_cur  mvuRight, or NULL for a monocular frame (all -1)
 *   d_Tcw          [nframes][12] CurrentFrame.mTcw as Rcw (row-major) followed by tcw
 *   d_level_mode   [nframes] 0 / 1 (bForward) / 2 (bBackward) as orbhip_projection_level_mode
 *                  returns it, or NULL for 0 everywhere (a monocular tracker)
 * CurrentFrame.mvpMapPoints is taken to be all NULL on entry, as the caller leaves it
 * (src/Tracking.cc:1323,1336). d_match_cur[nframes][cap] receives, per current keypoint, the index
 * of the last-frame keypoint whose map point it was given, or -1 (the caller stores
 * LastFrame.mvpMapPoints[index] there); d_nmatches[nframes] the return value. The scale factors
 * are those of the handle `e` (CurrentFrame.mvScaleFactors). The fork's bounding-box bookkeeping
 * inside the loop (matches_in_box, points_in_box, src/ORBmatcher.cc:1401-1408) does not influence
 * the matching and stays with the caller. */
typedef struct orbhip_projection_params {
    float fx, fy, cx, cy;     /* CurrentFrame.fx .. cy */
    float mbf;                /* CurrentFrame.mbf */
    float bounds[4];          /* mnMinX, mnMaxX, mnMinY, mnMaxY (orbhip_image_bounds) */
    float th;                 /* window factor: 15 mono / 7 stereo, doubled on the retry (src/Tracking.cc:1327-1337) */
    int check_orientation;    /* mbCheckOrientation */
} orbhip_projection_params;
int orbhip_search_by_projection_batch_device(orbhip_extractor* e, int nframes, int cap,
                                             const orbhip_keypoint* d_kp_last, const float* d_world_last,
                                             const uint8_t* d_flags_last, const uint8_t* d_desc_last, const int* d_cnt_last,
                                             const orbhip_keypoint* d_kp_cur, const uint8_t* d_desc_cur,
                                             const float* d_uright_cur, const int* d_cnt_cur, const float* d_Tcw,
                                             const int* d_level_mode, const orbhip_projection_params* prm,
                                             int* d_match_cur, int* d_nmatches, void* stream);
/* The same for one pair with host arrays (n1 last-frame and n2 current keypoints); waits for the result. */
int orbhip_search_by_projection(orbhip_extractor* e, const orbhip_keypoint* kp_last, const float* world_last,
                                const uint8_t* flags_last, const uint8_t* desc_last, int n1,
                                const orbhip_keypoint* kp_cur, const uint8_t* desc_cur, const float* uright_cur, int n2,
                                const float* Tcw, int level_mode, const orbhip_projection_params* prm,
                                int* match_cur, int* nmatches);
/* bForward / bBackward of the same function (src/ORBmatcher.cc:1330-1337) from the two poses
 * (12 floats each: R row-major, t), the baseline mb and bMono: 0 neither, 1 forward, 2 backward.
 * Host only. */
int orbhip_projection_level_mode(const float* Tcw_cur, const float* Tcw_last, float mb, int mono);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound,
 * const float th, const int ORBdist) (src/ORBmatcher.cc:1447-1564, relocalisation, src/Tracking.cc
 * Tracking::Relocalization) on plain arrays, laid out like orbhip_search_by_projection_batch_device.
 * Per key-frame keypoint i: d_kp_kf = pKF->mvKeysUn (the angle is read), d_world_kf = GetWorldPos()
 * of GetMapPointMatches()[i], d_dist_range_kf = {mfMinDistance, mfMaxDistance} of that point,
 * d_flags_kf bit 0 = "the point exists, is not bad and is not in sAlreadyFound", d_desc_kf =
 * GetDescriptor(). Per current keypoint: mvKeysUn, descriptor, d_occupied_cur = 1 where
 * CurrentFrame.mvpMapPoints[i2] is not NULL on entry (NULL: none). d_Tcw: 12 floats per frame.
 * prm: camera, bounds, th, check_orientation (mbf is not read). d_match_cur[i2] = index of the
 * key-frame keypoint whose point was assigned to current keypoint i2 by this call, else -1. */
int orbhip_search_by_projection_keyframe_batch_device(orbhip_extractor* e, int nframes, int cap,
                                                      const orbhip_keypoint* d_kp_kf, const float* d_world_kf,
                                                      const float* d_dist_range_kf, const uint8_t* d_flags_kf,
                                                      const uint8_t* d_desc_kf, const int* d_cnt_kf,
                                                      const orbhip_keypoint* d_kp_cur, const uint8_t* d_desc_cur,
                                                      const uint8_t* d_occupied_cur, const int* d_cnt_cur, const float* d_Tcw,
                                                      const orbhip_projection_params* prm, int orb_dist,
                                                      int* d_match_cur, int* d_nmatches, void* stream);
/* The same for one frame / key frame with host arrays; waits for the result. */
int orbhip_search_by_projection_keyframe(orbhip_extractor* e, const orbhip_keypoint* kp_kf, const float* world_kf,
                                         const float* dist_range_kf, const uint8_t* flags_kf, const uint8_t* desc_kf, int n1,
                                         const orbhip_keypoint* kp_cur, const uint8_t* desc_cur, const uint8_t* occupied_cur, int n2,
                                         const float* Tcw, const orbhip_projection_params* prm, int orb_dist,
                                         int* match_cur, int* nmatches);

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
 * vector<MapPoint*>& vpMatched, int th) (src/ORBmatcher.cc:287-400, loop closing) on plain arrays.
 * Per candidate map point: GetWorldPos(), GetNormal(), {mfMinDistance, mfMaxDistance}, flags bit 0 =
 * "not bad and not already in vpMatched", GetDescriptor(). Key frame: mvKeysUn (x, y, octave),
 * descriptors, occupied = 1 where vpMatched[idx] is not NULL on entry (NULL: none). d_Tcw: per frame the
 * 12 floats orbhip_decompose_sim3 makes of Scw (rotation and translation with the scale divided out,
 * src/ORBmatcher.cc:296-299); the host-array call takes Scw itself (3 x 4, row-major: sR | t... laid out
 * as R row-major then t, like the poses elsewhere). prm: camera, bounds, th (mbf and check_orientation
 * are not read). d_match_kf[idx] = index of the map point this call put into vpMatched[idx], else -1. */
int orbhip_decompose_sim3(const float* Scw, float* Tcw);
int orbhip_search_by_projection_sim3_batch_device(orbhip_extractor* e, int nframes, int cap, const float* d_world,
                                                  const float* d_normal, const float* d_dist_range, const uint8_t* d_flags,
                                                  const uint8_t* d_desc, const int* d_cnt_pts, const orbhip_keypoint* d_kp_kf,
                                                  const uint8_t* d_desc_kf, const uint8_t* d_occupied_kf, const int* d_cnt_kf,
                                                  const float* d_Tcw, const orbhip_projection_params* prm, int* d_match_kf,
                                                  int* d_nmatches, void* stream);
int orbhip_search_by_projection_sim3(orbhip_extractor* e, const float* world, const float* normal, const float* dist_range,
                                     const uint8_t* flags, const uint8_t* desc, int n1, const orbhip_keypoint* kp_kf,
                                     const uint8_t* desc_kf, const uint8_t* occupied_kf, int n2, const float* Scw,
                                     const orbhip_projection_params* prm, int* match_kf, int* nmatches);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:156-285;
 * Tracking::TrackReferenceKeyFrame and relocalisation) on plain arrays. The DBoW2 feature vectors
 * (pKF->mFeatVec, F.mFeatVec: vocabulary node -> feature indices) arrive as one node id per feature
 * (0xFFFFFFFF: the feature is in no node); computing them (the vocabulary transform) stays with the
 * caller. Key frame: d_kp_kf = mvKeysUn (angle), descriptors, node ids, d_flags_kf bit 0 =
 * "GetMapPointMatches()[i] exists and is not bad". Frame: d_kp_f = mvKeys (angle), descriptors, node
 * ids. nnratio = mfNNratio, check_orientation = mbCheckOrientation. d_match_f[f][i2] = index of the
 * key-frame feature whose map point vpMapPointMatches[i2] receives, -1 = NULL; d_nmatches[f] = the
 * returned count. Frames of up to 4096 keypoint slots are sorted on chip, larger ones (up to 65535) in a slice of global memory. */
int orbhip_search_by_bow_batch_device(orbhip_extractor* e, int nframes, int cap, const orbhip_keypoint* d_kp_kf,
                                      const uint8_t* d_desc_kf, const uint32_t* d_node_kf, const uint8_t* d_flags_kf,
                                      const int* d_cnt_kf, const orbhip_keypoint* d_kp_f, const uint8_t* d_desc_f,
                                      const uint32_t* d_node_f, const int* d_cnt_f, float nnratio, int check_orientation,
                                      int* d_match_f, int* d_nmatches, void* stream);
/* The same for one key frame / frame with host arrays; waits for the result. */
int orbhip_search_by_bow(orbhip_extractor* e, const orbhip_keypoint* kp_kf, const uint8_t* desc_kf, const uint32_t* node_kf,
                         const uint8_t* flags_kf, int n1, const orbhip_keypoint* kp_f, const uint8_t* desc_f,
                         const uint32_t* node_f, int n2, float nnratio, int check_orientation, int* match_f, int* nmatches);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:519-652,
 * loop detection) in the same form: both sides give mvKeysUn, descriptors, node ids and flags (bit 0 =
 * "the feature has a map point that is not bad"). d_match12[f][i1] = feature of key frame 2 whose map
 * point vpMatches12[i1] receives, -1 = NULL. */
int orbhip_search_by_bow_keyframes_batch_device(orbhip_extractor* e, int nframes, int cap, const orbhip_keypoint* d_kp1,
                                                const uint8_t* d_desc1, const uint32_t* d_node1, const uint8_t* d_flags1,
                                                const int* d_cnt1, const orbhip_keypoint* d_kp2, const uint8_t* d_desc2,
                                                const uint32_t* d_node2, const uint8_t* d_flags2, const int* d_cnt2, float nnratio,
                                                int check_orientation, int* d_match12, int* d_nmatches, void* stream);
int orbhip_search_by_bow_keyframes(orbhip_extractor* e, const orbhip_keypoint* kp1, const uint8_t* desc1, const uint32_t* node1,
                                   const uint8_t* flags1, int n1, const orbhip_keypoint* kp2, const uint8_t* desc2,
                                   const uint32_t* node2, const uint8_t* flags2, int n2, float nnratio, int check_orientation,
                                   int* match12, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)
 * (src/ORBmatcher.cc:46-127, called by Tracking::SearchLocalPoints) on plain arrays.
 * A map point is described by what that function reads from it: the fields Frame::isInFrustum
 * (src/Frame.cc:294-350) left in the MapPoint and two flags.
 *   proj_x, proj_y, proj_xr  mTrackProjX, mTrackProjY, mTrackProjXR
 *   view_cos                 mTrackViewCos          level   mnTrackScaleLevel
 *   flags                    bit 0: mbTrackInView && !isBad(); bit 1: Observations() > 0
 * A map point whose level is outside [0, nlevels) is ignored (the reference would index
 * mvScaleFactors out of range). Per frame: cap_mp slots of map points with their descriptors
 * (GetDescriptor(), 32 bytes each) and d_cnt_mp[f] of them in use, in the order of vpMapPoints;
 * cap_kp slots of keypoints (mvKeysUn: x, y, octave are read), descriptors, d_uright (mvuRight, NULL:
 * monocular, all -1), d_occupied (1 where F.mvpMapPoints[idx] holds a point with Observations() > 0 on
 * entry, NULL: none) and d_cnt_kp[f]. bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY} (host array, read
 * during the call); mvScaleFactors are the handle's; nnratio is the matcher's mfNNratio.
 * Output: d_match[f][idx] = index of the map point the reference assigns to F.mvpMapPoints[idx]
 * (the last one, if several), -1 where it assigns none; d_nmatches[f] = the returned count.
 * Asynchronous on `stream` (NULL: the handle's). One workgroup per frame; its state (17 bytes per map
 * point slot and 17 per keypoint slot) lives on chip while it fits 140 KB and in a global-memory slice
 * beyond that (slower, same results). */
typedef struct orbhip_track_point {
    float proj_x, proj_y, proj_xr, view_cos;
    int32_t level, flags;
} orbhip_track_point;
int orbhip_search_map_points_batch_device(orbhip_extractor* e, int nframes, int cap_mp, const orbhip_track_point* d_mp,
                                          const uint8_t* d_desc_mp, const int* d_cnt_mp, int cap_kp,
                                          const orbhip_keypoint* d_kp, const uint8_t* d_desc, const float* d_uright,
                                          const uint8_t* d_occupied, const int* d_cnt_kp, const float* bounds, float th,
                                          float nnratio, int* d_match, int* d_nmatches, void* stream);
/* The same for one frame with host arrays (n1 map points, n2 keypoints); waits for the result. */
int orbhip_search_map_points(orbhip_extractor* e, const orbhip_track_point* mp, const uint8_t* desc_mp, int n1,
                             const orbhip_keypoint* kp, const uint8_t* desc, const float* uright, const uint8_t* occupied, int n2,
                             const float* bounds, float th, float nnratio, int* match, int* nmatches);

/* Frame::isInFrustum(MapPoint*, viewingCosLimit) (src/Frame.cc:294-350) with MapPoint::PredictScale
 * (src/MapPoint.cc:401-417) and Get{Min,Max}DistanceInvariance (:372-382) for arrays of map points: the
 * loop of Tracking::SearchLocalPoints that precedes the search above. Per map point: GetWorldPos(),
 * GetNormal(), mfMinDistance, mfMaxDistance and flags (bit 0: test this point, i.e. it is not bad and
 * was not seen by this frame already; bit 1: Observations() > 0, passed through). Per frame: d_pose
 * = 15 floats, mRcw row-major, mtcw, mOw (Frame::UpdatePoseMatrices). prm: camera, mbf and image
 * bounds (th and check_orientation are not read); mfLogScaleFactor and mnScaleLevels are the
 * handle's. d_track[f][i] receives what isInFrustum stores in the MapPoint (flags bit 0 =
 * mbTrackInView), ready for orbhip_search_map_points_batch_device; d_nvisible[f] (may be NULL) the
 * number of points in view (nToMatch). Asynchronous on `stream` (NULL: the handle's). */
typedef struct orbhip_map_point {
    float pos[3], normal[3], min_distance, max_distance;
    int32_t flags;
} orbhip_map_point;
int orbhip_is_in_frustum_batch_device(orbhip_extractor* e, int nframes, int cap_mp, const orbhip_map_point* d_mp, const int* d_cnt_mp,
                                      const float* d_pose, const orbhip_projection_params* prm, float viewing_cos_limit,
                                      orbhip_track_point* d_track, int* d_nvisible, void* stream);
/* The same for one frame with host arrays; waits for the result. */
int orbhip_is_in_frustum(orbhip_extractor* e, const orbhip_map_point* mp, int n, const float* pose,
                         const orbhip_projection_params* prm, float viewing_cos_limit, orbhip_track_point* track, int* nvisible);

const char* orbhip_strerror(int code);
const char* orbhip_last_error(const orbhip_extractor* e);

#ifdef __cplusplus
}
#endif
#endif

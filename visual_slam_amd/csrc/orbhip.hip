// libORBhip — host side of the C ABI declared in include/orbhip.h.
//
// Mirrors ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc): the constructor tables are
// computed here exactly as the reference computes them; operator() becomes a fixed sequence of
// kernel launches over a pass of up to max_batch frames. There is no CPU fallback: every entry
// point fails with an error code when HIP is unavailable.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbhip.h"
#include "orb_device.h"
#include "orb_kernels.h"
#include "orb_stereo.h"
#include "orb_grid.h"
#include "orb_match.h"
#include "orb_frustum.h"

namespace {

inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

enum KernelClass { KC_LEVEL0 = 0, KC_RESIZE, KC_FAST, KC_QUADTREE, KC_DESCRIBE, KC_STEREO, KC_COUNT };
const char* const kKernelNames[KC_COUNT] = {"k_level0", "k_resize", "k_fast", "k_quadtree", "k_describe", "k_stereo"};

struct TimedSpan { int kc; hipEvent_t a, b; };

// Device buffers of one pass. A handle owns two of them so that consecutive passes of a large
// batch can run on two streams at once: the FAST / descriptor kernels are bound by the vector
// ALUs, the pyramid kernels by memory and the quadtree by latency, so passes in different stages
// fill each other's idle units.
struct Workspace {
    uint8_t* d_pyr = nullptr;
    uint2* d_seg = nullptr;        // [frames][seg_per_frame]   FAST output segments
    int* d_seg_count = nullptr;    // [frames][segc_per_frame]
    // keys of the quadtree, [frames][keys_per_frame] each: position x | y << 16, FAST score, label word
    uint32_t* d_keys_xy = nullptr;
    uint8_t* d_keys_score = nullptr;
    uint32_t* d_keys_label = nullptr;   // (16 bits of every slot in use where a level's node arrays fit LDS)
    int* d_key_count = nullptr;    // [frames][nlevels]
    LevelKp* d_lvl_kp = nullptr;   // [frames][kp_per_frame]
    int* d_lvl_cnt = nullptr;      // [frames][nlevels]
    uint32_t* d_qt_spill = nullptr;   // [frames][levels that spill][qt_spill_words] node arrays of k_quadtree that do not fit in LDS
    hipStream_t stream = nullptr;  // used when two passes are in flight
    hipEvent_t done = nullptr;
    // Level 0 of the last pass on this workspace: read in place from the caller's images (v0.org set) or
    // copied into its plane. A pass read in place remembers its source so that a later request for the
    // bordered level-0 image (orbhip_pyramid_copy) can still build the plane.
    Level0View v0 = {nullptr, 0, 0, 0};
    const uint8_t* l0_src = nullptr;
    size_t l0_row_stride = 0, l0_frame_stride = 0;
    bool l0_in_plane = true;
};
const int kLanes = 4;
// Level-0 quotas up to this run the quadtree with 128 threads per workgroup: 640 x 480 / 1000 features (quota 217) 0.193
// against 0.239 ms per 1024 frames at 256 threads, 752 x 480 / 1200 (261) 0.216 against 0.243; 1241 x 376 / 2000 (434) 0.393
// against 0.336, 1080p / 2000 1.13 against 0.82
const int kQtSmallQuota = 300;
const int kQtLatencyGroups = 256;   // (frames x levels) up to this: a pass that cannot fill the chip
const double kOverlapMinPixels = 5.0e7;   // passes of at least this many input pixels run FAST on a second stream beside the resize chain;
const double kOverlapMinPixelsPair = 1.0e8;   // the two sides of a stereo batch already run beside each other: larger passes only
const int kMinLaneFrames = 16;     // batches below twice this run as one pass on the caller's stream
const int kPipelineMinFrames = 8;  // host batches from this size on are cut into chunks that overlap copy, extraction and copy back

} // namespace

// Scratch of orbhip_stereo_match_batch_device_noalloc: the right-image records (16 bytes per keypoint
// slot), the SAD of every left keypoint's match plus one spare counter per pair (4 bytes per slot),
// and the row-bin offsets. It lives in the library's device image, so every GPU the library runs on
// holds its own copy at its own address until the process ends; the name below is only the host-side
// stand-in of the symbol: host code uses the address hipGetSymbolAddress reports for the current
// device (noalloc_scratch_ptr), never the name.
#define NOALLOC_SAD_BYTES ((size_t)(ORBHIP_NOALLOC_MAX_SLOTS + ORBHIP_NOALLOC_MAX_PAIRS) * sizeof(int))
#define NOALLOC_RINFO_BYTES ((size_t)ORBHIP_NOALLOC_MAX_SLOTS * sizeof(int4))
#define NOALLOC_BINS_BYTES ((size_t)ORBHIP_NOALLOC_MAX_BIN_ENTRIES * sizeof(int))
static __device__ __align__(256) unsigned char scratch[NOALLOC_RINFO_BYTES + NOALLOC_SAD_BYTES + NOALLOC_BINS_BYTES];

namespace {
const int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {nullptr};   // device address of `scratch` on every device seen so far

// Address of `scratch` on `device` (which must be the current device), resolved once per device.
hipError_t noalloc_scratch_ptr(int device, void** out) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t st = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (st != hipSuccess) return st;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
} // namespace

static int default_pack_threads() {
    int c = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) c = CPU_COUNT(&set);
    return std::max(1, std::min(c / 2, 4));
}

struct orbhip_extractor {
    orbhip_params prm;
    double scale_factor_d;
    std::vector<float> sf, inv_sf, sigma2, inv_sigma2;
    std::vector<int> per_level;
    int umax[16];
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // plan for the current image size
    int plan_w = 0, plan_h = 0, plan_frames = 0;
    PlanDev plan;
    size_t pyr_bytes_per_frame = 0;
    PlanDev* d_plan = nullptr;
    RsTab* d_tabs = nullptr;
    Workspace ws[kLanes];
    int plan_lanes = 0;            // workspaces allocated for the current plan
    int qt_threads = 0;            // ORBHIP_QT_THREADS=128|256: the quadtree's workgroup size whatever the frame size (0: by frame size)
    bool debug_skip_describe = false;   // ORBHIP_DEBUG_SKIP_DESCRIBE=1: stop after the quadtree (orbhip_debug_level_keypoints shows its output)
    bool debug_sync = false;       // ORBHIP_DEBUG_SYNC=1: wait after every group of launches and name the one that failed
    int max_lanes = 1;             // ORBHIP_LANES=2..4: that many passes of a large batch in flight (a few % more throughput)
    hipEvent_t ev_fork = nullptr;
    size_t qt_lds_limit = 160 * 1024 - 256;   // node arrays of a level beyond this many bytes live in global memory (ORBHIP_QT_LDS_LIMIT)
    int fast_pack = 1;             // ORBHIP_FAST_PACK=0: every tile of k_fast is a plain one (no bands in the last tile column)
    int inplace_l0 = 1;            // ORBHIP_INPLACE_LEVEL0=0: always copy gray input into the level-0 plane
    std::vector<int> l1_base;      // per 4-pixel group of level 1's plane: 4-aligned first source column of its taps
    int stereo_fork = 1;           // ORBHIP_STEREO_FORK=0: the right side of a stereo pass runs behind the left one on the same stream
    bool in_pair = false;          // set around the passes of a stereo batch call (the overlap rule differs)
    int overlap = 1;               // ORBHIP_OVERLAP=0: FAST of level 0 never runs beside the resize chain on aux_stream; 2: for passes of any size
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_l0 = nullptr, ev_f0 = nullptr;
    hipEvent_t ev_lv[ORB_MAX_LEVELS] = {};   // level l of the pass on the main stream is complete
    int fast_split_levels = 2;     // ORBHIP_FAST_SPLIT=n: with `overlap`, the first n levels' FAST tiles are launches of their own
    hipEvent_t ev_up[kLanes] = {}, ev_run[kLanes] = {}, ev_down[kLanes] = {};   // per input buffer of the pipelined host batch call: copied up, extracted, results down
    hipEvent_t ev_busy = nullptr;    // end of the last extraction call: a call on another stream waits for it
    hipEvent_t ev_order = nullptr;   // orders work of another instance (stereo matching) against this one's stream
    hipStream_t last_stream = nullptr;   // stream of the last batch call
    int* d_status = nullptr;       // bit0 candidate overflow, bit1 output capacity
    void* noalloc_scratch = nullptr;   // address of the library's static scratch on this handle's device
    float* d_map_x = nullptr;      // rectification maps (orbhip_set_rectification), map_w x map_h floats each
    float* d_map_y = nullptr;
    int map_w = 0, map_h = 0;
    size_t qt_lds = 0;
    size_t qt_spill_words = 0;     // words of one (frame, level) slice of the quadtree's global spill buffer
    int last_pass_frames = 0;
    int last_ws = 0;               // workspace of the last pass (pyramid / candidate read-back)
    unsigned long long pass_gen = 0;       // counts passes: tags the host copy of a frame's pyramid
    uint8_t* h_pyr = nullptr;      // pinned host copy of all levels of one frame (orbhip_pyramid_copy)
    size_t h_pyr_bytes = 0;
    unsigned long long h_pyr_gen = 0;
    int h_pyr_frame = -1;
    // The single-frame host call (what a SLAM front end makes per camera frame) is launch-bound: its copy up,
    // (behind the copy up, which goes band by band as the image is packed) its kernel launches and its copies down are captured into a HIP graph the second time a configuration
    // is seen and replayed from then on (ORBHIP_GRAPH=0: never). config_gen counts everything that
    // invalidates the captured pointers: a new plan, new staging buffers, new rectification maps.
    int use_graph = 1;
    int host_pipeline = 1;         // ORBHIP_HOST_PIPELINE=0: the host batch call runs as one copy up, all passes, one copy down
    bool host_trace = false;       // ORBHIP_HOST_TRACE=1: the pipelined host batch call prints its timeline on stderr
    int host_lanes = 3;            // ORBHIP_HOST_LANES=2..4: chunks of the pipelined host batch call in flight (input buffers)
    int pack_threads = 1;          // ORBHIP_PACK_THREADS=n: host threads that pack pageable images into page-locked staging (default: half the CPUs the process may use, at most 4)
    int host_chunk = 0;            // ORBHIP_HOST_CHUNK=n: frames per chunk of the pipelined host batch call (0: about 64 MB of images)
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey { int w, h, channels, rgb, cap; unsigned long long gen; } graph_key = {0, 0, 0, 0, 0, 0}, graph_seen = {0, 0, 0, 0, 0, 0};
    unsigned long long config_gen = 1;
    // what run_pass noted about level 0 in ws[0] while the graph was captured (the captured kernels carry that view by
    // value): written back on every replay, since other calls on the handle leave their own view there in between
    Level0View graph_v0 = {nullptr, 0, 0, 0};
    const uint8_t* graph_l0_src = nullptr;
    size_t graph_l0_row_stride = 0, graph_l0_frame_stride = 0;
    // staging for the host entry points
    uint8_t* d_in = nullptr; size_t d_in_bytes = 0;
    uint8_t* h_in = nullptr; size_t h_in_bytes = 0;      // pinned: rows are packed here, then one copy up
    uint8_t* h_out = nullptr; size_t h_out_bytes = 0;    // pinned: counts, keypoints, descriptors in one copy down each
    float* d_out_kp = nullptr; uint8_t* d_out_desc = nullptr; int* d_out_cnt = nullptr;
    size_t d_out_frames = 0; int d_out_cap = 0;
    // stereo matching (owned by the left instance of a pair)
    int* d_st_sad = nullptr; size_t d_st_sad_n = 0;      // [frames][cap] SAD of the match, -1 = none
    int4* d_st_rinfo = nullptr;                          // [frames][cap] row band / octave / x of the right keypoints, binned by row
    int* d_st_bins = nullptr; size_t d_st_bins_n = 0;    // [frames][nbins + 1] record range of every row bin
    uint8_t* d_st_tmp = nullptr; size_t d_st_tmp_bytes = 0;   // staging of the host-array entry point
    uint8_t* h_st_tmp = nullptr; size_t h_st_tmp_bytes = 0;   // its page-locked twin
    uint8_t* d_match_spill = nullptr; size_t d_match_spill_bytes = 0;   // matcher state of frames too large for LDS
    uint8_t* d_st_host = nullptr; uint8_t* h_st_host = nullptr; size_t st_host_bytes = 0;   // mvuRight / mvDepth / match counts of the host stereo batch call
    // timing
    bool timing = false;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> event_pool;
    float kc_ms[KC_COUNT] = {0};
    float kc_busy_ms[KC_COUNT] = {0};
    int kc_launches[KC_COUNT] = {0};
};

namespace {

#define HIP_TRY(e, call)                                                                     \
    do {                                                                                     \
        hipError_t _s = (call);                                                              \
        if (_s != hipSuccess) {                                                              \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_s);                    \
            return ORBHIP_ERR_HIP;                                                           \
        }                                                                                    \
    } while (0)

// Every entry point works on its handle's device and leaves the calling thread's current device as it
// found it: a host that drives several GPUs from one process (one handle per device, as the reference runs
// its two extractor instances side by side, src/Frame.cc:114-117) never finds it changed under its feet.
struct DeviceScope {
    int prev = -1;
    hipError_t st;
    explicit DeviceScope(int dev) {
        int cur = -1;
        st = hipGetDevice(&cur);
        if (st == hipSuccess && cur != dev) {
            st = hipSetDevice(dev);
            if (st == hipSuccess) prev = cur;
        }
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define DEVICE_SCOPE(e)                         \
    DeviceScope _device_scope((e)->device);     \
    HIP_TRY(e, _device_scope.st)

template <typename T>
void dfree(T*& p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

void free_plan(orbhip_extractor* e) {
    e->config_gen++;
    dfree(e->d_plan); dfree(e->d_tabs);
    for (Workspace& w : e->ws) {
        dfree(w.d_pyr); dfree(w.d_seg); dfree(w.d_seg_count);
        dfree(w.d_keys_xy); dfree(w.d_keys_score); dfree(w.d_keys_label); dfree(w.d_key_count); dfree(w.d_lvl_kp); dfree(w.d_lvl_cnt); dfree(w.d_qt_spill);
    }
    e->plan_w = e->plan_h = e->plan_frames = e->plan_lanes = 0;
}

// Tap tables of cv::resize(8UC1, INTER_LINEAR): same construction as OpenCV's xofs/ialpha and
// yofs/ibeta, with the clamping the row/column loops apply folded into the indices.
void build_resize_tabs(int sw, int sh, int dw, int dh, RsTab* xt, RsTab* yt) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xt[dx].s0 = (uint16_t)sx;
        xt[dx].s1 = (uint16_t)std::min(sx + 1, sw - 1);
        xt[dx].a0 = sat_short(cv_round((1.f - fx) * 2048));
        xt[dx].a1 = sat_short(cv_round(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yt[dy].s0 = (uint16_t)std::min(std::max(sy, 0), sh - 1);
        yt[dy].s1 = (uint16_t)std::min(std::max(sy + 1, 0), sh - 1);
        yt[dy].a0 = sat_short(cv_round((1.f - fy) * 2048));
        yt[dy].a1 = sat_short(cv_round(fy * 2048));
    }
}

int reflect101_host(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

void blur_taps(int variant, int taps[4]) {   // cv::getGaussianKernel(7, 2, CV_32F) scaled by 256
    double k[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; float t = (float)std::exp(-0.5 / 4.0 * x * x); k[i] = t; sum += t; }
    const double scale = 1. / sum;
    int q[7], qs = 0;
    for (int i = 0; i < 7; i++) { q[i] = cv_round((float)(k[i] * scale) * 256.f); qs += q[i]; }
    if (variant == ORBHIP_BLUR_FIXEDPOINT) q[3] -= qs - 256;
    taps[0] = q[3]; taps[1] = q[2]; taps[2] = q[1]; taps[3] = q[0];
}

// Plan and workspaces for passes of up to `frames` frames, `lanes` of them in flight.
int ensure_plan(orbhip_extractor* e, int w, int h, int frames, int lanes) {
    if (e->plan_w == w && e->plan_h == h && e->plan_frames >= frames && e->plan_lanes >= lanes) return ORBHIP_OK;
    if (w > 16384 || h > 16384) { e->err = "image larger than 16384 px"; return ORBHIP_ERR_GEOMETRY; }
    free_plan(e);
    PlanDev& P = e->plan;
    std::memset(&P, 0, sizeof(P));
    const int nl = e->prm.nlevels;
    P.nlevels = nl;
    P.ini_th = e->prm.ini_th_fast;
    P.min_th = e->prm.min_th_fast;
    blur_taps(e->prm.blur_variant, P.blur_taps);
    P.blur_ties_even = e->prm.blur_variant == ORBHIP_BLUR_LEGACY;
    std::memcpy(P.umax, e->umax, sizeof(P.umax));
    for (int a = 0; a < 16; a++)
        for (int i = 0; i < 31; i++) {
            const int u = i - 15;
            if (std::abs(u) > e->umax[a]) continue;
            P.disc_w[a][i >> 2] |= 1u << (8 * (i & 3));
            P.disc_w[a][8 + (i >> 2)] |= (uint32_t)(u + 15) << (8 * (i & 3));
        }
    {
        // Rotated test points keep their distance from the keypoint (at most sqrt(338) for this
        // pattern) and are rounded to the nearest pixel: a point on row ry has |rx| <= 0.5 +
        // sqrt(R^2 - (|ry| - 0.5)^2), and the vertical 7-tap pass at (rx, ry) reads the row sums of
        // rows ry-3 .. ry+3 at column rx. Row / column index of the sums = offset + 22 / + 19.
        static const signed char pat[1024] = {
#include "../../include/orb_pattern_31.inc"
        };
        int r2 = 0;
        for (int i = 0; i < 512; i++) r2 = std::max(r2, pat[2 * i] * pat[2 * i] + pat[2 * i + 1] * pat[2 * i + 1]);
        const double R = std::sqrt((double)r2) + 1e-3;
        const int ry_max = (int)std::floor(R + 0.5);
        // first / last column group a row needs (row = dy + 22; rows nobody reads have lo > hi)
        int glo[46], ghi[46];
        for (int r = 0; r < 46; r++) { glo[r] = 1; ghi[r] = 0; }
        for (int dy = -ry_max - 3; dy <= ry_max + 3; dy++) {
            const int ry = std::max(std::abs(dy) - 3, 0);
            const double t = R * R - std::max(ry - 0.5, 0.0) * std::max(ry - 0.5, 0.0);
            if (t < 0 || dy + 22 < 0 || dy + 22 > 44) continue;
            const int m = std::min((int)std::floor(0.5 + std::sqrt(t)), 19);
            glo[dy + 22] = (19 - m) >> 2; ghi[dy + 22] = std::min((19 + m) >> 2, 9);
        }
        // an item covers the same four columns of an even row and the row below it (their sums share a dword
        // of the column-major array)
        int n = 0;
        for (int r = 0; r < 44; r += 2) {
            const bool a = glo[r] <= ghi[r], b = glo[r + 1] <= ghi[r + 1];
            if (!a && !b) continue;
            const int lo = a && b ? std::min(glo[r], glo[r + 1]) : a ? glo[r] : glo[r + 1];
            const int hi = a && b ? std::max(ghi[r], ghi[r + 1]) : a ? ghi[r] : ghi[r + 1];
            for (int g = lo; g <= hi; g++)
                if (n < ORB_HP_ITEMS) P.hp_items[n++] = (uint32_t)(r * DS_RAW_P + 4 * g) | ((uint32_t)((4 * g * DS_HP_P + r) * 2) << 16);
                else { e->err = "internal: blur work list overflow"; return ORBHIP_ERR_ARG; }
        }
        if (glo[44] <= ghi[44]) { e->err = "internal: blur work list expects an even number of rows"; return ORBHIP_ERR_ARG; }
        for (int i = n; i < ORB_HP_ITEMS; i++) P.hp_items[i] = P.hp_items[i - n];   // repeats write the same values again
    }
    size_t pyr_off = 0, keys_off = 0, seg_off = 0;
    int segc_off = 0, kp_off = 0, tab_off = 0;
    size_t max_qt_words = 0, spill_words = 0;
    for (int l = 0; l < nl; l++) {
        LevelGeom& g = P.lv[l];
        g.w = cv_round((float)w * e->inv_sf[l]);          // src/ORBextractor.cc:1111-1112
        g.h = cv_round((float)h * e->inv_sf[l]);
        if (g.w < 1 || g.h < 1) { e->err = "pyramid level collapses to zero size"; return ORBHIP_ERR_GEOMETRY; }
        g.pitch = (ORB_OX + g.w + ORB_EDGE + 63) / 64 * 64;
        g.rows = g.h + 2 * ORB_EDGE;
        g.plane_bytes = (unsigned long long)g.pitch * g.rows;
        g.pyr_off = pyr_off;
        pyr_off += g.plane_bytes * frames;
        g.maxbx = g.w - ORB_EDGE + 3;                      // src/ORBextractor.cc:775-776
        g.maxby = g.h - ORB_EDGE + 3;
        const float width = (float)(g.maxbx - ORB_MINB), height = (float)(g.maxby - ORB_MINB);
        g.ncols = width > 0 ? (int)(width / 30.f) : 0;     // src/ORBextractor.cc:784-787
        g.nrows = height > 0 ? (int)(height / 30.f) : 0;
        if (g.ncols < 1 || g.nrows < 1) {
            // the reference visits no cell on such a level: it contributes no keypoints
            g.ncols = g.nrows = 0; g.wcell = g.hcell = 1;
            g.n_ini = 0; g.hx = 1.f; g.node_cap = 1; g.keys_cap = 1;
            g.ntx = g.nty = 0;
            g.fast_ntiles = g.fast_ntx_plain = 0; g.fast_bands = 1; g.fast_lpb = 64;
        } else {
            g.wcell = (int)std::ceil(width / g.ncols);
            g.hcell = (int)std::ceil(height / g.nrows);
            g.wcell_magic = (unsigned)((1ull << 32) / (unsigned)g.wcell + 1);
            g.hcell_magic = (unsigned)((1ull << 32) / (unsigned)g.hcell + 1);
            g.wcell_m22 = ((1u << 22) + g.wcell - 1) / g.wcell;
            g.hcell_m22 = ((1u << 22) + g.hcell - 1) / g.hcell;
            // operand bounds of the kernels' 24-bit multiplies that depend on the image size (listed at umul24 in orb_device.h)
            if ((long long)g.w * g.wcell >= (1 << 22) || (long long)g.h * g.hcell >= (1 << 22) ||
                (long long)g.h * g.ncols >= (1 << 24) || g.plane_bytes >= (1ull << 31)) {
                e->err = "image too large for the cell arithmetic"; return ORBHIP_ERR_GEOMETRY;
            }
            g.n_ini = (int)std::round(width / (float)(g.maxby - ORB_MINB));   // src/ORBextractor.cc:543
            if (g.n_ini < 1) { e->err = "image too narrow for the quadtree (nIni == 0)"; return ORBHIP_ERR_GEOMETRY; }
            g.hx = width / (float)g.n_ini;                                  // src/ORBextractor.cc:545
            // strict local maxima inside a cell: at most one per 2x2 block
            const int dw = g.w - 2 * ORB_EDGE, dh = g.h - 2 * ORB_EDGE;
            g.keys_cap = ((dw + 1) / 2 + g.ncols) * ((dh + 1) / 2 + g.nrows);
            g.node_cap = std::max(e->per_level[l] + 3, 4 * g.n_ini) + 1;
            g.ntx = (g.w - ORB_EDGE - ORB_MINB + FT_COLS - 1) / FT_COLS;
            g.nty = (dh + FT_ROWS - 1) / FT_ROWS;
            // the last tile column: lanes its output columns need plus two halo lanes; a wave holds 64 / that
            // many bands (tile rows side by side). Level 0 stays plain: it may be read in place from an image
            // that has no rows below it, and a band's loads are not clamped to the level.
            const int rest = (g.w - ORB_EDGE - ORB_MINB) - FT_COLS * (g.ntx - 1);
            g.fast_lpb = (rest + 3) / 4 + 2;
            g.fast_bands = (l == 0 || e->fast_pack == 0) ? 1 : 64 / g.fast_lpb;
            if (g.fast_bands < 2) { g.fast_bands = 1; g.fast_lpb = 64; }
            g.fast_ntx_plain = g.fast_bands > 1 ? g.ntx - 1 : g.ntx;
            g.fast_ntiles = g.fast_ntx_plain * g.nty + (g.fast_bands > 1 ? (g.nty + g.fast_bands - 1) / g.fast_bands : 0);
        }
        g.seg_off = seg_off; seg_off += (size_t)g.fast_ntiles * 4 * FT_SEGCAP;
        g.segc_off = segc_off; segc_off += g.fast_ntiles * 4;
        g.keys_off = keys_off; keys_off += g.keys_cap;
        g.qt_words = std::max(18 * g.node_cap, g.fast_ntiles * 4 + 1);
        {
            // node arrays (18 words per node) plus the per-cell bit mask: in LDS when they fit, else in a global slice
            const size_t words = (size_t)g.qt_words + (size_t)(g.ncols * g.nrows + 31) / 32;
            if (words * 4 > e->qt_lds_limit) {
                P.qt_spill_slot[l] = P.qt_nspill++;
                spill_words = std::max(spill_words, (words + 63) / 64 * 64);
            } else {
                P.qt_spill_slot[l] = -1;
                max_qt_words = std::max(max_qt_words, words);
            }
        }
        g.quota = e->per_level[l];
        g.kp_off = kp_off; kp_off += g.node_cap;
        g.scale = e->sf[l];
        g.patch_size = (float)(int)(31 * e->sf[l]);        // src/ORBextractor.cc:837
        // resize tap tables are indexed by device column / row (border reflection folded in)
        g.xtab_off = tab_off; tab_off += (ORB_OX + g.w + ORB_EDGE + 3) / 4 * 4;
        g.ytab_off = tab_off; tab_off += (g.rows + 1) & ~1;
    }
    P.fast_cum[0][0] = P.fast_cum[1][0] = 0;
    for (int l = 0; l < nl; l++) {
        const unsigned plain = (unsigned)(P.lv[l].fast_ntx_plain * P.lv[l].nty), packed = (unsigned)P.lv[l].fast_ntiles - plain;
        P.fast_cum[0][l + 1] = P.fast_cum[0][l] + plain;
        P.fast_cum[1][l + 1] = P.fast_cum[1][l] + packed;
        P.fast_tiles_magic[0][l] = grid_magic(plain);
        P.fast_tiles_magic[1][l] = grid_magic(packed);
        P.fast_ntx_magic[l] = grid_magic((unsigned)P.lv[l].fast_ntx_plain);
        // the magic divisions in k_fast are exact for n * d < 2^32
        const unsigned tiles = std::max(plain, packed);
        if ((unsigned long long)tiles * frames * std::max(tiles, 1u) >= (1ull << 32)) {
            e->err = "too many FAST tiles per pass"; return ORBHIP_ERR_GEOMETRY;
        }
    }
    if ((unsigned long long)(P.fast_cum[0][nl] + P.fast_cum[1][nl]) * frames >= (1ull << 31)) { e->err = "too many FAST tiles per pass"; return ORBHIP_ERR_GEOMETRY; }
    P.seg_per_frame = std::max(seg_off, (size_t)1);
    P.segc_per_frame = std::max(segc_off, 1);
    P.keys_per_frame = keys_off;
    P.kp_per_frame = kp_off;
    e->pyr_bytes_per_frame = pyr_off / frames;
    e->qt_lds = max_qt_words * 4;
    e->qt_spill_words = spill_words;
    std::vector<RsTab> tabs((size_t)std::max(tab_off, 1));
    for (int l = 1; l < nl; l++) {
        const LevelGeom& g = P.lv[l];
        std::vector<RsTab> xt((size_t)g.w), yt((size_t)g.h);
        build_resize_tabs(P.lv[l - 1].w, P.lv[l - 1].h, g.w, g.h, xt.data(), yt.data());
        const int ncol = (ORB_OX + g.w + ORB_EDGE + 3) / 4 * 4;
        for (int c = 0; c < ncol; c++) tabs[g.xtab_off + c] = xt[reflect101_host(c - ORB_OX, g.w)];
        for (int r = 0; r < g.rows; r++) tabs[g.ytab_off + r] = yt[reflect101_host(r - ORB_OY, g.h)];
        // k_resize feeds the 4 pixels of a lane from one 12-byte load at a 4-aligned source offset;
        // its shorter variant needs their taps (two bytes each) inside 8 bytes from the leftmost one
        int narrow = 1;
        for (int c = 0; c < ncol; c += 4) {
            int lo = 1 << 30, hi = 0;
            for (int i = 0; i < 4; i++) { lo = std::min(lo, (int)tabs[g.xtab_off + c + i].s0); hi = std::max(hi, (int)tabs[g.xtab_off + c + i].s0); }
            if (hi + 1 - (lo & ~3) > 11) { e->err = "scale factor too large for the resize kernel (must be <= 2)"; return ORBHIP_ERR_ARG; }
            if (hi + 1 - lo > 7) narrow = 0;
        }
        P.lv[l].rs_narrow = narrow;
        if (l == 1) {
            e->l1_base.assign((size_t)ncol / 4, 0);
            for (int c = 0; c < ncol; c += 4) {
                int lo = 1 << 30;
                for (int i = 0; i < 4; i++) lo = std::min(lo, (int)tabs[g.xtab_off + c + i].s0);
                e->l1_base[(size_t)c / 4] = lo & ~3;
            }
        }
    }
    DEVICE_SCOPE(e);
    HIP_TRY(e, hipMalloc(&e->d_plan, sizeof(PlanDev)));
    HIP_TRY(e, hipMalloc(&e->d_tabs, tabs.size() * sizeof(RsTab)));
    for (int k = 0; k < lanes; k++) {
        Workspace& W = e->ws[k];
        HIP_TRY(e, hipMalloc(&W.d_pyr, pyr_off + 64));
        HIP_TRY(e, hipMalloc(&W.d_seg, (size_t)P.seg_per_frame * frames * sizeof(uint2)));
        HIP_TRY(e, hipMalloc(&W.d_seg_count, (size_t)P.segc_per_frame * frames * sizeof(int)));
        HIP_TRY(e, hipMalloc(&W.d_keys_xy, (size_t)P.keys_per_frame * frames * sizeof(uint32_t)));
        HIP_TRY(e, hipMalloc(&W.d_keys_score, (size_t)P.keys_per_frame * frames));
        HIP_TRY(e, hipMalloc(&W.d_keys_label, (size_t)P.keys_per_frame * frames * sizeof(uint32_t)));
        HIP_TRY(e, hipMalloc(&W.d_key_count, (size_t)frames * nl * sizeof(int)));
        HIP_TRY(e, hipMalloc(&W.d_lvl_kp, (size_t)frames * P.kp_per_frame * sizeof(LevelKp)));
        HIP_TRY(e, hipMalloc(&W.d_lvl_cnt, (size_t)frames * nl * sizeof(int)));
        if (P.qt_nspill) HIP_TRY(e, hipMalloc(&W.d_qt_spill, (size_t)frames * P.qt_nspill * spill_words * sizeof(uint32_t)));
        HIP_TRY(e, hipMemsetAsync(W.d_lvl_kp, 0, (size_t)frames * P.kp_per_frame * sizeof(LevelKp), e->stream));
    }
    HIP_TRY(e, hipMemcpyAsync(e->d_plan, &P, sizeof(PlanDev), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->d_tabs, tabs.data(), tabs.size() * sizeof(RsTab), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    e->plan_lanes = lanes;
    if (e->qt_lds > 64 * 1024)
    {
        HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree<false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->qt_lds));
        HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->qt_lds));
    }
    e->plan_w = w; e->plan_h = h; e->plan_frames = frames;
    return ORBHIP_OK;
}

hipEvent_t get_event(orbhip_extractor* e) {
    if (!e->event_pool.empty()) { hipEvent_t ev = e->event_pool.back(); e->event_pool.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

struct SpanGuard {   // records a pair of events around a group of launches when timing is on
    orbhip_extractor* e; hipStream_t s; int kc; hipEvent_t a = nullptr;
    SpanGuard(orbhip_extractor* e_, hipStream_t s_, int kc_) : e(e_), s(s_), kc(kc_) {
        if (e->timing) { a = get_event(e); (void)hipEventRecord(a, s); }
    }
    ~SpanGuard() {
        if (e->timing) { hipEvent_t b = get_event(e); (void)hipEventRecord(b, s); e->spans.push_back(TimedSpan{kc, a, b}); e->kc_launches[kc]++; }
    }
};

// Level 0 of `frames` frames into its plane (src/ORBextractor.cc:1113-1128 for level 0: the image inside
// its 19-pixel reflected border), through the rectification maps or the gray conversion when asked.
void launch_level0(orbhip_extractor* e, Workspace& W, hipStream_t s, const uint8_t* d_imgs, int frames, size_t row_stride,
                   size_t frame_stride, int channels, int rgb, bool use_maps) {
    const PlanDev& P = e->plan;
    const dim3 blk2(64, 4);
    {
        const LevelGeom& g = P.lv[0];
        const int n16 = (ORB_OX + g.w + ORB_EDGE + 15) / 16;
        const int aligned16 = ((uintptr_t)d_imgs % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0);
        // 16-byte groups [i_lo, i_hi) lie inside the image: straight dwordx4 copies when the
        // source is aligned; the groups that touch the border go through the reflecting byte path
        const int i_lo = ORB_OX / 16, i_hi = (ORB_OX + g.w) / 16;
        if (use_maps && e->d_map_x) {                          // gray input through the rectification maps
            const int ndw = n16 * 4;
            hipLaunchKernelGGL(k_level0_remap, dim3((ndw + 63) / 64, (g.rows + 3) / 4, (frames + RM_FR - 1) / RM_FR), blk2, 0, s,
                               d_imgs, row_stride, frame_stride, frames, e->d_map_x, e->d_map_y, W.d_pyr, g, ndw,
                               (int)(((uintptr_t)d_imgs % 4 == 0) && (row_stride % 4 == 0) && (frame_stride % 4 == 0)));
        } else if (channels != 1) {
            const dim3 grid((g.rows * n16 + 255) / 256, frames);
            if (channels == 3)
                hipLaunchKernelGGL(k_level0_color<3>, grid, dim3(256), 0, s, d_imgs, row_stride, frame_stride, W.d_pyr, g, n16, rgb, aligned16);
            else
                hipLaunchKernelGGL(k_level0_color<4>, grid, dim3(256), 0, s, d_imgs, row_stride, frame_stride, W.d_pyr, g, n16, rgb, aligned16);
        } else if (i_hi > i_lo && g.h > ORB_EDGE && g.w > ORB_OX) {
            // the border groups of a row (the first i_lo and those from i_hi on) ride in an extra block per row band
            const dim3 grid((i_hi - i_lo + 63) / 64 + 1, (g.rows + 4 * L0_RPT - 1) / (4 * L0_RPT), frames);
            if (aligned16)
                hipLaunchKernelGGL(k_level0_interior<true>, grid, blk2, 0, s, d_imgs, row_stride, frame_stride, W.d_pyr, g, i_lo, i_hi,
                                   n16 - (i_hi - i_lo));
            else
                hipLaunchKernelGGL(k_level0_interior<false>, grid, blk2, 0, s, d_imgs, row_stride, frame_stride, W.d_pyr, g, i_lo, i_hi,
                                   n16 - (i_hi - i_lo));
        } else {
            hipLaunchKernelGGL(k_level0_edges, dim3((g.rows * n16 + 255) / 256, frames), dim3(256), 0, s,
                               d_imgs, row_stride, frame_stride, W.d_pyr, g, n16, n16, n16);
        }
    }
}

// ORBHIP_DEBUG_SYNC: the stream is drained after a group of launches so that a failing kernel is named.
static int debug_sync_point(orbhip_extractor* e, hipStream_t s, const char* what) {
    if (!e->debug_sync) return ORBHIP_OK;
    const hipError_t st = hipStreamSynchronize(s);
    if (st == hipSuccess) return ORBHIP_OK;
    static thread_local char msg[160];
    std::snprintf(msg, sizeof msg, "%s failed: %s", what, hipGetErrorString(st));
    e->err = msg;
    return ORBHIP_ERR_HIP;
}

// One pass: `frames` frames starting at device pointer d_imgs; outputs written at frame index frame0.
// channels: 1 (gray), 3 or 4 (interleaved colour, converted to gray on the way in); rgb: channel 0 is red.
// row_readable: bytes of every row (the last row of the last frame included) that belong to the buffer.
int run_pass(orbhip_extractor* e, Workspace& W, hipStream_t s, const uint8_t* d_imgs, int frames, size_t row_stride,
             size_t frame_stride, size_t row_readable, int channels, int rgb, float* d_kp, uint8_t* d_desc, int* d_counts,
             int cap, int frame0) {
    const PlanDev& P = e->plan;
    const int nl = P.nlevels;
    const dim3 blk2(64, 4);
    // A gray image whose rows the kernels can read as aligned dwords already is level 0: FAST, the resize
    // to level 1, the descriptors and the stereo refinement read it where it lies, and the copy into a
    // bordered plane is skipped (the kernels never read past row_readable bytes of a row).
    const int xlim = (int)(std::min(row_readable, row_stride) & ~(size_t)3);
    const bool inplace = e->inplace_l0 && channels == 1 && !e->d_map_x && (uintptr_t)d_imgs % 4 == 0 && row_stride % 4 == 0 &&
                         frame_stride % 4 == 0 && xlim >= P.lv[0].w && xlim >= 16 &&
                         (unsigned long long)row_stride * (unsigned long long)P.lv[0].h < (1ull << 31) &&
                         (nl == 1 || P.lv[1].rs_narrow);
    W.v0 = inplace ? Level0View{d_imgs, (unsigned long long)frame_stride, (int)row_stride, xlim} : Level0View{nullptr, 0, 0, 0};
    W.l0_src = d_imgs; W.l0_row_stride = row_stride; W.l0_frame_stride = frame_stride;
    W.l0_in_plane = !inplace;
    {
        SpanGuard t(e, s, KC_LEVEL0);
        if (!inplace) launch_level0(e, W, s, d_imgs, frames, row_stride, frame_stride, channels, rgb, true);
    }
    if (int rc = debug_sync_point(e, s, "level 0")) return rc;
    // FAST is bound by vector issue, the resize chain by memory: with `overlap` FAST runs on a second stream beside
    // the chain, level by level (below). Measured per pass against one stream: 1080p x 1024 frames +4 to +6 %, x 64 +4 %,
    // x 32 +2 %, x 16 +1 %, x 8 -2 %; 1241 x 376 x 1024 +6 %, x 32 -5 %; VGA x 1024 +2 to +3 %, x 256 +2 %, x 64 -3 %, x 16 -10 %:
    // the fork and join cost more than a short chain can hide, hence the smallest pass. The two sides of a stereo batch
    // run beside each other on two streams already; four streams pay from larger passes on (per side: 752 x 480 x 512 +3 %,
    // 1241 x 376 x 256 +4 %, 1080p x 64 +3 %; 752 x 480 x 64 -4 %).
    const double px0 = (double)P.lv[0].w * P.lv[0].h;
    const bool large = (double)frames * px0 >= (e->in_pair ? kOverlapMinPixelsPair : kOverlapMinPixels);
    const bool beside = e->overlap && nl > 1 && (e->overlap > 1 || large);
    // With `beside` the plain FAST tiles run on the second stream, level by level as the chain on the main stream
    // delivers the levels: the first `ksplit` levels as launches of their own, the rest in one launch behind the
    // chain's last kernel; the packed tiles (the narrow last tile column of every level: a short launch) run on the
    // main stream behind the chain, beside them. The kernel takes the first tile of its launch as an argument.
    const int ksplit = beside ? std::min(std::max(e->fast_split_levels, 1), nl) : 0;
    auto fast_plain = [&](hipStream_t st, int l0, int l1) {    // plain tiles of levels l0 .. l1-1
        const unsigned first = P.fast_cum[0][l0] * (unsigned)frames, count = P.fast_cum[0][l1] * (unsigned)frames - first;
        if (!count) return;
        SpanGuard t(e, st, KC_FAST);
        hipLaunchKernelGGL(k_fast<false>, dim3(count), dim3(256), 0, st, W.d_pyr, e->d_plan, frames, W.d_seg, W.d_seg_count, e->d_status, first, W.v0);
    };
    if (beside) {
        HIP_TRY(e, hipEventRecord(e->ev_l0, s));
        HIP_TRY(e, hipStreamWaitEvent(e->aux_stream, e->ev_l0, 0));
        fast_plain(e->aux_stream, 0, 1);
    }
    for (int l = 1; l < nl; l++) {
        {
            SpanGuard t(e, s, KC_RESIZE);
            const LevelGeom& g = P.lv[l];
            // (only the level itself: 4-pixel groups ORB_OX / 4 .. c4_end - 1 of the plane's rows ORB_OY .. ORB_OY + h - 1)
            const int c4_begin = ORB_OX / 4, c4_end = c4_begin + (g.w + 3) / 4;
            const int nbx = (c4_end + 63) / 64;
            const dim3 grid(nbx, (g.h + 4 * RS_RPT - 1) / (4 * RS_RPT), frames);
            if (l == 1 && inplace) {
                // the source is the caller's image: same kernel, the image's strides in place of the plane's, and the
                // block columns with a lane whose 12-byte window would pass the readable part of a row run the
                // clamping code (normally the last one: the right edge and the border columns reflected onto it)
                LevelGeom gs = P.lv[0];
                gs.pitch = (int)row_stride;
                gs.plane_bytes = frame_stride;
                const uint8_t* src = reinterpret_cast<const uint8_t*>((uintptr_t)d_imgs - ((size_t)ORB_OY * row_stride + ORB_OX));
                int first_clamp = nbx;
                for (int c4 = c4_begin; c4 < c4_end && c4 < (int)e->l1_base.size(); c4++)
                    if (e->l1_base[(size_t)c4] + 12 > xlim) { first_clamp = c4 / 64; break; }
                hipLaunchKernelGGL((k_resize<true, true>), grid, blk2, 0, s, src, W.d_pyr + g.pyr_off, gs, g, e->d_tabs, c4_end, first_clamp, xlim - 12);
            } else if (g.rs_narrow)
                hipLaunchKernelGGL((k_resize<true, false>), grid, blk2, 0, s, W.d_pyr + P.lv[l - 1].pyr_off, W.d_pyr + g.pyr_off, P.lv[l - 1], g,
                                   e->d_tabs, c4_end, 0, 0);
            else
                hipLaunchKernelGGL((k_resize<false, false>), grid, blk2, 0, s, W.d_pyr + P.lv[l - 1].pyr_off, W.d_pyr + g.pyr_off, P.lv[l - 1], g,
                                   e->d_tabs, c4_end, 0, 0);
        }
        if (int rc = debug_sync_point(e, s, "k_resize")) return rc;
        if (beside && (l < ksplit || l == nl - 1)) {           // level l is there (l == nl - 1: all of them)
            HIP_TRY(e, hipEventRecord(e->ev_lv[l], s));
            HIP_TRY(e, hipStreamWaitEvent(e->aux_stream, e->ev_lv[l], 0));
            if (l < ksplit) fast_plain(e->aux_stream, l, l + 1);
            if (l == nl - 1 && ksplit < nl) fast_plain(e->aux_stream, std::max(ksplit, 1), nl);
        }
    }
    if (!beside) fast_plain(s, 0, nl);
    if (P.fast_cum[1][nl]) {
        SpanGuard t(e, s, KC_FAST);
        hipLaunchKernelGGL(k_fast<true>, dim3(P.fast_cum[1][nl] * (unsigned)frames), dim3(256), 0, s,
                           W.d_pyr, e->d_plan, frames, W.d_seg, W.d_seg_count, e->d_status, 0u, W.v0);
    }
    if (beside) {
        HIP_TRY(e, hipEventRecord(e->ev_f0, e->aux_stream));
        HIP_TRY(e, hipStreamWaitEvent(s, e->ev_f0, 0));
    }
    if (int rc = debug_sync_point(e, s, "k_fast")) return rc;
    {
        SpanGuard t(e, s, KC_QUADTREE);
        // Threads per (frame, level): 128 where the lists stay short and the pass fills the chip, 256 otherwise (a pass of
        // a few frames leaves most of the chip idle, and what counts then is the time of one workgroup: one 1080p frame
        // host to host 0.43 ms with 128 threads, 0.38 with 256, 0.37 with 1024). See k_quadtree.
        const int qt = e->qt_threads ? e->qt_threads : (frames * nl > kQtLatencyGroups && P.lv[0].quota <= kQtSmallQuota) ? 128 : 256;
        typedef void (*QtFn)(const PlanDev*, const uint2*, const int*, uint32_t*, uint8_t*, uint32_t*, int*, LevelKp*, int*, int*, uint32_t*,
                             unsigned long long);
        if (P.qt_nspill < nl) {
            QtFn fn = qt == 128 ? k_quadtree<false, 128> : k_quadtree<false, 256>;
            hipLaunchKernelGGL(fn, dim3(frames, nl), dim3(qt), e->qt_lds, s,
                               e->d_plan, W.d_seg, W.d_seg_count, W.d_keys_xy, W.d_keys_score, W.d_keys_label, W.d_key_count, W.d_lvl_kp, W.d_lvl_cnt,
                               e->d_status, static_cast<uint32_t*>(nullptr), 0ull);
        }
        if (P.qt_nspill) {
            QtFn fn = qt == 128 ? k_quadtree<true, 128> : k_quadtree<true, 256>;
            hipLaunchKernelGGL(fn, dim3(frames, nl), dim3(qt), 0, s,
                               e->d_plan, W.d_seg, W.d_seg_count, W.d_keys_xy, W.d_keys_score, W.d_keys_label, W.d_key_count, W.d_lvl_kp, W.d_lvl_cnt,
                               e->d_status, W.d_qt_spill, (unsigned long long)e->qt_spill_words);
        }
    }
    if (int rc = debug_sync_point(e, s, "k_quadtree")) return rc;
    if (e->debug_skip_describe) HIP_TRY(e, hipMemsetAsync(d_counts + frame0, 0, (size_t)frames * sizeof(int), s));
    else {
        SpanGuard t(e, s, KC_DESCRIBE);
        const int maxkp = std::min(cap, P.kp_per_frame);
        const int ngroups = (maxkp + 3) / 4;                  // 4 keypoints per workgroup
        const unsigned nwg = (unsigned)ngroups * 8u * (unsigned)((frames + 7) / 8);
        if ((unsigned long long)nwg * (8ull * ngroups) >= (1ull << 32)) { e->err = "too many keypoints per pass"; return ORBHIP_ERR_CAPACITY; }
        hipLaunchKernelGGL(k_describe, dim3(nwg), dim3(256), 0, s,
                           e->d_plan, W.d_pyr, W.d_lvl_kp, W.d_lvl_cnt, d_kp, d_desc, d_counts, cap, frame0, e->d_status,
                           frames, ngroups, grid_magic(8u * (unsigned)ngroups), W.v0);
    }
    if (int rc = debug_sync_point(e, s, "k_describe")) return rc;
    HIP_TRY(e, hipGetLastError());
    e->last_pass_frames = frames;
    e->last_ws = (int)(&W - e->ws);
    e->pass_gen++;
    return ORBHIP_OK;
}

int check_status(orbhip_extractor* e) {
    int st = 0;
    HIP_TRY(e, hipMemcpy(&st, e->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        (void)hipMemset(e->d_status, 0, sizeof(int));
        e->err = (st & 1) ? "internal candidate capacity exceeded" : (st & 4) ? "internal: quadtree round limit reached" : "output capacity (cap) too small";
        return ORBHIP_ERR_CAPACITY;
    }
    return ORBHIP_OK;
}

// The staging block of the host-array entry points (one item per call: a front end's per-frame calls): a device block
// (d_st_tmp) and its page-locked twin. The inputs are placed in the twin at the offsets of their device arrays and go up
// in one copy, the outputs come down in one copy each, and one wait ends the call.
int staged_ensure(orbhip_extractor* e, size_t need) {
    if (need <= e->d_st_tmp_bytes && need <= e->h_st_tmp_bytes) return ORBHIP_OK;
    HIP_TRY(e, hipDeviceSynchronize());
    dfree(e->d_st_tmp);
    e->d_st_tmp_bytes = 0;
    if (e->h_st_tmp) (void)hipHostFree(e->h_st_tmp);
    e->h_st_tmp = nullptr; e->h_st_tmp_bytes = 0;
    HIP_TRY(e, hipMalloc(&e->d_st_tmp, need));
    e->d_st_tmp_bytes = need;
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_st_tmp), need, hipHostMallocDefault));
    e->h_st_tmp_bytes = need;
    return ORBHIP_OK;
}

struct Staged {
    orbhip_extractor* e;
    uint8_t *d, *h;
    size_t hi = 0;                                             // end of the inputs placed so far
    struct Out { void* dst; size_t off, bytes; } outs[6];
    int nout = 0;
    explicit Staged(orbhip_extractor* x) : e(x), d(x->d_st_tmp), h(x->h_st_tmp) {}
    void up(const void* dptr, const void* src, size_t bytes) {
        const size_t off = (size_t)(static_cast<const uint8_t*>(dptr) - d);
        if (bytes) std::memcpy(h + off, src, bytes);
        hi = std::max(hi, off + bytes);
    }
    hipError_t send() { return hi ? hipMemcpyAsync(d, h, hi, hipMemcpyHostToDevice, e->stream) : hipSuccess; }
    void down(void* dst, const void* dptr, size_t bytes) {
        if (nout < 6) outs[nout++] = Out{dst, (size_t)(static_cast<const uint8_t*>(dptr) - d), bytes};
    }
    hipError_t fetch() {
        for (int i = 0; i < nout; i++)
            if (outs[i].bytes) {
                const hipError_t st = hipMemcpyAsync(h + outs[i].off, d + outs[i].off, outs[i].bytes, hipMemcpyDeviceToHost, e->stream);
                if (st != hipSuccess) return st;
            }
        const hipError_t st = hipStreamSynchronize(e->stream);
        if (st != hipSuccess) return st;
        for (int i = 0; i < nout; i++)
            if (outs[i].bytes) std::memcpy(outs[i].dst, h + outs[i].off, outs[i].bytes);
        return hipSuccess;
    }
};

// A thread that is joined when its owner goes out of scope (every return path of the pipelines below).
struct HelperThread {
    std::thread t;
    void join() { if (t.joinable()) t.join(); }
    ~HelperThread() { join(); }
};

// body(first, n) over [first, first + count) in up to `threads` slices of at least 8 items, a thread each (the caller's included).
template <typename F>
void run_sliced(int threads, int first, int count, F&& body) {
    const int nt = std::max(1, std::min(threads, count / 8));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back([&body, first, count, t, nt] { body(first + count * t / nt, count * (t + 1) / nt - count * t / nt); });
    body(first, count / nt);
    for (std::thread& t : th) t.join();
}

void collect_timing(orbhip_extractor* e) {
    // kc_ms: the spans of a class added up; kc_busy_ms: the time during which at least one span of the class was open
    // (launches of one class run side by side on two streams when FAST follows the resize chain level by level)
    std::vector<std::pair<float, float>> iv[KC_COUNT];
    const hipEvent_t base = e->spans.empty() ? nullptr : e->spans[0].a;
    for (TimedSpan& sp : e->spans) {
        float ms = 0, ta = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            e->kc_ms[sp.kc] += ms;
            if (hipEventElapsedTime(&ta, base, sp.a) == hipSuccess) iv[sp.kc].emplace_back(ta, ta + ms);
            else e->kc_busy_ms[sp.kc] += ms;
        }
    }
    for (int kc = 0; kc < KC_COUNT; kc++) {
        std::sort(iv[kc].begin(), iv[kc].end());
        float lo = 0, hi = -1;
        for (const auto& x : iv[kc]) {
            if (hi < lo || x.first > hi) { if (hi >= lo) e->kc_busy_ms[kc] += hi - lo; lo = x.first; hi = x.second; }
            else hi = std::max(hi, x.second);
        }
        if (hi >= lo) e->kc_busy_ms[kc] += hi - lo;
    }
    for (TimedSpan& sp : e->spans) {
        e->event_pool.push_back(sp.a);
        e->event_pool.push_back(sp.b);
    }
    e->spans.clear();
}

} // namespace

extern "C" {

int orbhip_create(const orbhip_params* p, orbhip_extractor** out) {
    if (!p || !out) return ORBHIP_ERR_ARG;
    *out = nullptr;
    if (p->nfeatures < 1 || p->nlevels < 1 || p->nlevels > ORB_MAX_LEVELS || !(p->scale_factor > 1.0f) ||
        p->ini_th_fast < 1 || p->min_th_fast < 1 || p->ini_th_fast > 255 || p->min_th_fast > p->ini_th_fast)
        return ORBHIP_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || p->device < 0 || p->device >= ndev) return ORBHIP_ERR_NO_DEVICE;
    orbhip_extractor* e = new orbhip_extractor;
    e->prm = *p;
    if (e->prm.max_batch < 1) e->prm.max_batch = 0;            // by image size: see pass_budget
    e->device = p->device;
    const int nl = p->nlevels;
    // src/ORBextractor.cc:412-431 (scaleFactor is a double member initialised from the float argument)
    e->scale_factor_d = p->scale_factor;
    e->sf.resize(nl); e->sigma2.resize(nl); e->inv_sf.resize(nl); e->inv_sigma2.resize(nl);
    e->sf[0] = 1.0f; e->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        e->sf[i] = (float)(e->sf[i - 1] * e->scale_factor_d);
        e->sigma2[i] = e->sf[i] * e->sf[i];
    }
    for (int i = 0; i < nl; i++) {
        e->inv_sf[i] = 1.0f / e->sf[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    // src/ORBextractor.cc:435-446
    e->per_level.resize(nl);
    const float factor = (float)(1.0f / e->scale_factor_d);
    float desired = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        e->per_level[l] = cv_round(desired);
        sum += e->per_level[l];
        desired *= factor;
    }
    e->per_level[nl - 1] = std::max(p->nfeatures - sum, 0);
    // src/ORBextractor.cc:454-469
    int um[18] = {0};
    const int hp = 15;
    int v, v0, vmax = cv_floor(hp * std::sqrt(2.f) / 2 + 1), vmin = cv_ceil(hp * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) um[v] = cv_round(std::sqrt((double)hp * hp - v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) {
        while (um[v0] == um[v0 + 1]) ++v0;
        um[v] = v0;
        ++v0;
    }
    std::memcpy(e->umax, um, sizeof(e->umax));
    DeviceScope device_scope(e->device);
    if (device_scope.st != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&e->d_status, sizeof(int)) != hipSuccess || hipMemset(e->d_status, 0, sizeof(int)) != hipSuccess) {
        delete e;
        return ORBHIP_ERR_HIP;
    }
    // the static scratch of the no-allocation entry point: its address on this device, taken once per device
    if (noalloc_scratch_ptr(e->device, &e->noalloc_scratch) != hipSuccess) { orbhip_destroy(e); return ORBHIP_ERR_HIP; }
    bool ok = hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_l0, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_f0, hipEventDisableTiming) == hipSuccess &&

              hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_busy, hipEventDisableTiming) == hipSuccess;
    for (Workspace& w : e->ws)
        ok = ok && hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&w.done, hipEventDisableTiming) == hipSuccess;
    // Streams of a process share a few hardware queues (four by default), and work on two streams that share one
    // runs in order. Measured for the six streams created above, in creation order: the 3rd and 4th share a queue, so
    // do the 2nd and 5th, the 1st and 6th have one each. The lanes are numbered so that the first two (and the
    // handle's own stream) lie on different queues.
    if (ok) std::swap(e->ws[1].stream, e->ws[kLanes - 1].stream);
    for (hipEvent_t& ev : e->ev_lv) ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < kLanes; k++)
        ok = ok && hipEventCreateWithFlags(&e->ev_up[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&e->ev_run[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&e->ev_down[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) { orbhip_destroy(e); return ORBHIP_ERR_HIP; }
    if (const char* v = std::getenv("ORBHIP_DEBUG_SKIP_DESCRIBE")) e->debug_skip_describe = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_QT_THREADS")) { const int t = std::atoi(v); e->qt_threads = (t == 128 || t == 256) ? t : 0; }
    if (const char* v = std::getenv("ORBHIP_DEBUG_SYNC")) e->debug_sync = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_LANES")) e->max_lanes = std::min(kLanes, std::max(1, std::atoi(v)));
    if (const char* v = std::getenv("ORBHIP_OVERLAP")) e->overlap = std::min(2, std::max(0, std::atoi(v)));
    if (const char* v = std::getenv("ORBHIP_STEREO_FORK")) e->stereo_fork = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_FAST_SPLIT")) e->fast_split_levels = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("ORBHIP_INPLACE_LEVEL0")) e->inplace_l0 = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_FAST_PACK")) e->fast_pack = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_GRAPH")) e->use_graph = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_HOST_PIPELINE")) e->host_pipeline = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_HOST_CHUNK")) e->host_chunk = std::max(0, std::atoi(v));
    e->pack_threads = default_pack_threads();
    if (const char* v = std::getenv("ORBHIP_PACK_THREADS")) e->pack_threads = std::min(32, std::max(1, std::atoi(v)));
    if (const char* v = std::getenv("ORBHIP_HOST_TRACE")) e->host_trace = std::atoi(v) != 0;
    if (const char* v = std::getenv("ORBHIP_HOST_LANES")) e->host_lanes = std::min(kLanes, std::max(2, std::atoi(v)));
    if (const char* v = std::getenv("ORBHIP_QT_LDS_LIMIT")) e->qt_lds_limit = std::min<size_t>(160 * 1024 - 256, (size_t)std::max(std::atoi(v), 1024));
    *out = e;
    return ORBHIP_OK;
}

int orbhip_destroy(orbhip_extractor* e) {
    if (!e) return ORBHIP_ERR_ARG;
    DeviceScope device_scope(e->device);
    (void)hipDeviceSynchronize();
    collect_timing(e);
    for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
    if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
    free_plan(e);
    dfree(e->d_status); dfree(e->d_in); dfree(e->d_out_kp); dfree(e->d_out_desc); dfree(e->d_out_cnt);
    dfree(e->d_st_sad); dfree(e->d_st_rinfo); dfree(e->d_st_bins); dfree(e->d_st_tmp); dfree(e->d_match_spill);
    dfree(e->d_map_x); dfree(e->d_map_y); dfree(e->d_st_host);
    if (e->h_st_host) (void)hipHostFree(e->h_st_host);
    if (e->h_st_tmp) (void)hipHostFree(e->h_st_tmp);
    if (e->h_in) (void)hipHostFree(e->h_in);
    if (e->h_out) (void)hipHostFree(e->h_out);
    if (e->h_pyr) (void)hipHostFree(e->h_pyr);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    for (Workspace& w : e->ws) {
        if (w.stream) (void)hipStreamDestroy(w.stream);
        if (w.done) (void)hipEventDestroy(w.done);
    }
    for (int k = 0; k < kLanes; k++)
        for (hipEvent_t ev : {e->ev_up[k], e->ev_run[k], e->ev_down[k]})
            if (ev) (void)hipEventDestroy(ev);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_l0) (void)hipEventDestroy(e->ev_l0);
    if (e->ev_f0) (void)hipEventDestroy(e->ev_f0);
    for (hipEvent_t ev : e->ev_lv) if (ev) (void)hipEventDestroy(ev);
    if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
    if (e->ev_order) (void)hipEventDestroy(e->ev_order);
    if (e->ev_busy) (void)hipEventDestroy(e->ev_busy);
    delete e;
    return ORBHIP_OK;
}

// Frames one pass may hold: max_batch, but keep the workspaces (about 20 bytes per input pixel and frame:
// pyramid planes, FAST output segments, quadtree keys) under ~24 GB, and the descriptor kernel's
// one-dimensional grid (8 * groups of 4 keypoints * frames / 8 workgroups) small enough for its 32-bit
// multiply-high division, which is exact while grid * 8 * groups < 2^32.
static int pass_budget(const orbhip_extractor* e, int width, int height) {
    const double ws_per_frame = 20.0 * (double)width * (double)height;
    const int ws_limit = std::max(1, (int)(24e9 / ws_per_frame));
    const unsigned long long groups = ((unsigned long long)orbhip_max_keypoints(e) + 3) / 4;
    const int grid_limit = (int)std::min<unsigned long long>(1u << 20, 8ull * (((1ull << 26) - 1) / (groups * groups)));
    // (the frame index is the y or z coordinate of several grids: at most 65535)
    // Default pass: 512 frames, more of small images (as many pixels as 512 frames of 1080p: the dozen dependent launches
    // of a pass cost their gaps once per pass; 640 x 480: 379 / 400 / 409 k frames/s with passes of 512 / 1024 / 2048).
    const int max_batch = e->prm.max_batch > 0 ? e->prm.max_batch
                                               : std::max(512, (int)(512.0 * 1920.0 * 1080.0 / ((double)width * (double)height)));
    return std::max(1, std::min(std::min(std::min(max_batch, ws_limit), std::max(grid_limit, 8)), 65535));
}

static int extract_device_impl(orbhip_extractor* e, const uint8_t* d_imgs, int nframes, int width, int height,
                               size_t row_stride, size_t frame_stride, size_t row_readable, int channels, int rgb,
                               orbhip_keypoint* d_kp, uint8_t* d_desc, int* d_counts, int cap, void* stream) {
    if (!e || !d_kp || !d_desc || !d_counts || cap < 1 || nframes < 0 || (channels != 1 && channels != 3 && channels != 4))
        return ORBHIP_ERR_ARG;
    if (!d_imgs || width <= 0 || height <= 0 || nframes == 0) return ORBHIP_ERR_EMPTY;
    if (row_stride < (size_t)width * channels) return ORBHIP_ERR_ARG;
    if (e->d_map_x && (channels != 1 || width != e->map_w || height != e->map_h || width < 2 || height < 2)) {
        e->err = "rectification maps are set: the input must be a gray image of the maps' size";
        return ORBHIP_ERR_ARG;
    }
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    // A handle has one set of workspaces: a call on another stream than the previous one waits for
    // that call's work before it touches them.
    // (ev_busy was recorded at the end of that call, so the earlier stream itself is not touched again)
    if (e->last_stream && e->last_stream != s) HIP_TRY(e, hipStreamWaitEvent(s, e->ev_busy, 0));
    e->last_stream = s;
    const int budget = pass_budget(e, width, height);          // frames in flight
    // Large batches run as passes of half the budget, alternating between two workspaces on two
    // internal streams that fork from and join back into the caller's stream. (Per-kernel timing
    // then measures every kernel as it runs, next to the other pass's kernels.)
    const int lanes = std::max(1, std::min(e->max_lanes, std::min(budget, nframes / kMinLaneFrames)));
    const int chunk = lanes > 1 ? std::min((nframes + lanes - 1) / lanes, budget / lanes) : std::min(nframes, budget);
    if (e->plan_w != width || e->plan_h != height || e->plan_frames < chunk || e->plan_lanes < lanes) {
        // a new plan frees the old workspaces: nothing may still be running on them
        HIP_TRY(e, hipDeviceSynchronize());
        int rc = ensure_plan(e, width, height, std::max(chunk, e->plan_w == width && e->plan_h == height ? e->plan_frames : 0),
                             std::max(lanes, e->plan_w == width && e->plan_h == height ? e->plan_lanes : 0));
        if (rc) return rc;
    }
    if (lanes > 1) {
        HIP_TRY(e, hipEventRecord(e->ev_fork, s));
        for (int k = 0; k < lanes; k++) HIP_TRY(e, hipStreamWaitEvent(e->ws[k].stream, e->ev_fork, 0));
    }
    int pass = 0;
    for (int f0 = 0; f0 < nframes; f0 += chunk, pass++) {
        const int n = std::min(chunk, nframes - f0);
        Workspace& W = e->ws[pass % lanes];
        int rc = run_pass(e, W, lanes > 1 ? W.stream : s, d_imgs + (size_t)f0 * frame_stride, n, row_stride, frame_stride,
                          row_readable, channels, rgb, reinterpret_cast<float*>(d_kp), d_desc, d_counts, cap, f0);
        if (rc) return rc;
    }
    if (lanes > 1)
        for (int k = 0; k < lanes; k++) {
            HIP_TRY(e, hipEventRecord(e->ws[k].done, e->ws[k].stream));
            HIP_TRY(e, hipStreamWaitEvent(s, e->ws[k].done, 0));
        }
    HIP_TRY(e, hipEventRecord(e->ev_busy, s));
    return ORBHIP_OK;
}

int orbhip_extract_batch_device(orbhip_extractor* e, const uint8_t* d_imgs, int nframes, int width, int height,
                                size_t row_stride, size_t frame_stride, orbhip_keypoint* d_kp, uint8_t* d_desc,
                                int* d_counts, int cap, void* stream) {
    // (the caller owns `width` bytes of every row; what lies behind the last row's is not known to be readable)
    return extract_device_impl(e, d_imgs, nframes, width, height, row_stride, frame_stride, (size_t)std::max(width, 0), 1, 0, d_kp, d_desc,
                               d_counts, cap, stream);
}

int orbhip_extract_batch_device_color(orbhip_extractor* e, const uint8_t* d_imgs, int nframes, int width, int height,
                                      size_t row_stride, size_t frame_stride, int channels, int rgb,
                                      orbhip_keypoint* d_kp, uint8_t* d_desc, int* d_counts, int cap, void* stream) {
    if (channels != 3 && channels != 4) return ORBHIP_ERR_ARG;
    return extract_device_impl(e, d_imgs, nframes, width, height, row_stride, frame_stride, (size_t)std::max(width, 0) * channels,
                               channels, rgb != 0, d_kp, d_desc, d_counts, cap, stream);
}

int orbhip_synchronize(orbhip_extractor* e) {
    if (!e) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    collect_timing(e);
    return check_status(e);
}

// Worker threads for packing a chunk of caller images into pinned memory: half of the CPUs this process may
// run on, at most 4 (ORBHIP_PACK_THREADS overrides; a GPU box reports the whole machine in
// hardware_concurrency, the affinity mask is what the process really has).
// Rows [r0, r1) of the `n` frames starting at `src` (rows counted through all frames) into the staging layout
// (rows of dev_row bytes, frames of height rows). The caller owns width_bytes of every source row, not the
// padding behind it.
static void pack_rows(const uint8_t* src, size_t row_stride, size_t frame_stride, uint8_t* dst, size_t dev_row, int height,
                      size_t width_bytes, size_t r0, size_t r1) {
    while (r0 < r1) {
        const size_t f = r0 / (size_t)height, y0 = r0 - f * (size_t)height;
        const size_t y1 = std::min((size_t)height, y0 + (r1 - r0));
        const uint8_t* s = src + f * frame_stride + y0 * row_stride;
        uint8_t* d = dst + (f * (size_t)height + y0) * dev_row;
        if (row_stride == dev_row) std::memcpy(d, s, (y1 - y0 - 1) * row_stride + width_bytes);
        else for (size_t y = y0; y < y1; y++, s += row_stride, d += dev_row) std::memcpy(d, s, width_bytes);
        r0 += y1 - y0;
    }
}

static void pack_frames(const uint8_t* src, size_t row_stride, size_t frame_stride, uint8_t* dst, size_t dev_row, int height,
                        size_t width_bytes, int n, int threads) {
    const size_t rows = (size_t)n * (size_t)height;
    const size_t bytes = rows * width_bytes;
    const int nt = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, bytes >> 22));   // a thread per 4 MB at least
    if (nt <= 1) { pack_rows(src, row_stride, frame_stride, dst, dev_row, height, width_bytes, 0, rows); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    for (int t = 1; t < nt; t++)
        th.emplace_back(pack_rows, src, row_stride, frame_stride, dst, dev_row, height, width_bytes, rows * t / nt, rows * (t + 1) / nt);
    pack_rows(src, row_stride, frame_stride, dst, dev_row, height, width_bytes, 0, rows / nt);
    for (std::thread& t : th) t.join();
}

// true when `p` is page-locked host memory the GPU can copy from directly (hipHostMalloc / hipHostRegister)
static bool host_pointer_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// Staging buffers of the host entry points: in_bytes of device and pinned input, outputs of out_frames frames
// (device) and their pinned copy: [status word, 16 bytes][counts][keypoints][descriptors].
struct HostStage { uint8_t *h_status, *h_cnt, *h_kp, *h_desc; size_t kp_bytes, desc_bytes; };
static int host_staging(orbhip_extractor* e, size_t in_bytes, bool need_pinned_in, int out_frames, int cap, HostStage* hs) {
    if (in_bytes > e->d_in_bytes) {
        HIP_TRY(e, hipDeviceSynchronize());
        // a pass that read its level 0 in place from the old staging buffer loses it
        for (Workspace& w : e->ws)
            if (w.l0_src && w.l0_src >= e->d_in && w.l0_src < e->d_in + e->d_in_bytes) {
                w.v0 = Level0View{nullptr, 0, 0, 0}; w.l0_src = nullptr; w.l0_in_plane = true;
                e->last_pass_frames = 0;
            }
        dfree(e->d_in);
        e->d_in_bytes = 0;
        HIP_TRY(e, hipMalloc(&e->d_in, in_bytes));
        e->d_in_bytes = in_bytes;
        e->config_gen++;
    }
    if (need_pinned_in && in_bytes > e->h_in_bytes) {
        HIP_TRY(e, hipDeviceSynchronize());
        e->config_gen++;
        if (e->h_in) (void)hipHostFree(e->h_in);
        e->h_in = nullptr; e->h_in_bytes = 0;
        HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_in), in_bytes, hipHostMallocDefault));
        e->h_in_bytes = in_bytes;
    }
    if ((size_t)out_frames > e->d_out_frames || cap != e->d_out_cap) {
        HIP_TRY(e, hipDeviceSynchronize());
        e->config_gen++;
        dfree(e->d_out_kp); dfree(e->d_out_desc); dfree(e->d_out_cnt);
        e->d_out_frames = 0;
        HIP_TRY(e, hipMalloc(&e->d_out_kp, (size_t)out_frames * cap * sizeof(orbhip_keypoint)));
        HIP_TRY(e, hipMalloc(&e->d_out_desc, (size_t)out_frames * cap * 32));
        HIP_TRY(e, hipMalloc(&e->d_out_cnt, (size_t)out_frames * sizeof(int)));
        e->d_out_frames = out_frames; e->d_out_cap = cap;
    }
    hs->kp_bytes = (size_t)out_frames * cap * sizeof(orbhip_keypoint);
    hs->desc_bytes = (size_t)out_frames * cap * 32;
    const size_t cnt_bytes = ((size_t)out_frames * sizeof(int) + 15) & ~(size_t)15;
    const size_t total = 16 + cnt_bytes + hs->kp_bytes + hs->desc_bytes;
    if (total > e->h_out_bytes) {
        HIP_TRY(e, hipDeviceSynchronize());
        e->config_gen++;
        if (e->h_out) (void)hipHostFree(e->h_out);
        e->h_out = nullptr; e->h_out_bytes = 0;
        HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_out), total, hipHostMallocDefault));
        e->h_out_bytes = total;
    }
    hs->h_status = e->h_out;
    hs->h_cnt = e->h_out + 16;
    hs->h_kp = hs->h_cnt + cnt_bytes;
    hs->h_desc = hs->h_kp + hs->kp_bytes;
    return ORBHIP_OK;
}

// The status word arrives with the results (one more small copy on the stream, no blocking copy of its own).
static int host_status(orbhip_extractor* e, const HostStage& hs, hipStream_t s) {
    const int st = *reinterpret_cast<const volatile int*>(hs.h_status);
    if (!st) return ORBHIP_OK;
    (void)hipMemsetAsync(e->d_status, 0, sizeof(int), s);
    e->err = (st & 1) ? "internal candidate capacity exceeded" : (st & 4) ? "internal: quadtree round limit reached" : "output capacity (cap) too small";
    return ORBHIP_ERR_CAPACITY;
}

static void unpack_frames(const HostStage& hs, int f0, int n, int cap, orbhip_keypoint* kp, uint8_t* desc, int* counts) {
    std::memcpy(counts + f0, hs.h_cnt + (size_t)f0 * sizeof(int), (size_t)n * sizeof(int));
    for (int f = f0; f < f0 + n; f++) {
        const int m = std::min(counts[f], cap);
        if (m <= 0) continue;
        std::memcpy(kp + (size_t)f * cap, hs.h_kp + (size_t)f * cap * sizeof(orbhip_keypoint), (size_t)m * sizeof(orbhip_keypoint));
        std::memcpy(desc + (size_t)f * cap * 32, hs.h_desc + (size_t)f * cap * 32, (size_t)m * 32);
    }
}

// A batch of host images as a pipeline of chunks over three streams, one per stage: copies up, kernels, copies down.
// While chunk k's images are packed into pinned memory by a few host threads (or read straight from the caller's buffer
// when that is page-locked), chunk k-1 is on its way up, chunk k-2 is being extracted and the results of chunk k-3 come
// down and are handed to the caller's arrays. What Frame::ExtractORB does per image (src/Frame.cc:272-278) for many
// images in host memory; the transfer over PCIe is what bounds it, so the copies up follow each other without a gap.
// (One stream per stage, not one per chunk: streams of a process share a few hardware queues, work on two streams
// that share one runs in order, and a chunk's copy up would then wait for another chunk's kernels and copy down.)
// A stage's input buffer is free again when the host has seen the results of the chunk that used it.
static int extract_host_pipelined(orbhip_extractor* e, const uint8_t* imgs, int nframes, int width, int height,
                                  size_t row_stride, size_t frame_stride, int channels, int rgb, orbhip_keypoint* kp,
                                  uint8_t* desc, int* counts, int cap) {
    const size_t width_bytes = (size_t)width * channels;
    const size_t dev_row = (width_bytes + 15) & ~(size_t)15, dev_frame = dev_row * height;
    const int budget = pass_budget(e, width, height);
    int chunk = e->host_chunk > 0 ? e->host_chunk : (int)std::max<size_t>(4, ((size_t)64 << 20) / dev_frame);
    chunk = std::min(chunk, (nframes + 3) / 4);                // four chunks at least, so that the stages overlap
    chunk = std::max(1, std::min(chunk, budget));
    const int nchunks = (nframes + chunk - 1) / chunk;
    const int nb = std::min(e->host_lanes, nchunks);           // input buffers: chunks in flight
    // page-locked images in the staging layout are copied from where they lie; any other layout is packed by the host
    // threads first (a strided copy from host memory runs row by row: 1241-byte rows move at a hundredth of the rate)
    const bool pinned = row_stride == dev_row && frame_stride == dev_frame && host_pointer_is_pinned(imgs);
    HostStage hs;
    int rc = host_staging(e, (size_t)nb * chunk * dev_frame, !pinned, nframes, cap, &hs);
    if (rc) return rc;
    // earlier work of this handle (on whatever stream) is over before the workspaces are touched
    if (e->last_stream) HIP_TRY(e, hipEventSynchronize(e->ev_busy));
    if (e->plan_w != width || e->plan_h != height || e->plan_frames < chunk || e->plan_lanes < 1) {
        HIP_TRY(e, hipDeviceSynchronize());
        const bool same = e->plan_w == width && e->plan_h == height;
        rc = ensure_plan(e, width, height, std::max(chunk, same ? e->plan_frames : 0), std::max(1, same ? e->plan_lanes : 0));
        if (rc) return rc;
    }
    if (e->d_map_x && (channels != 1 || width != e->map_w || height != e->map_h || width < 2 || height < 2)) {
        e->err = "rectification maps are set: the input must be a gray image of the maps' size";
        return ORBHIP_ERR_ARG;
    }
    auto fail = [&](int code) { (void)hipDeviceSynchronize(); return code; };   // nothing of this call stays in flight
    const hipStream_t s_up = e->ws[0].stream, s_run = e->stream, s_down = e->ws[1].stream;   // three different hardware queues
    Workspace& W = e->ws[0];                                   // the kernels of all chunks run one after another
    // ORBHIP_HOST_TRACE=1: where the call's time goes, chunk by chunk (host clock and device events), on stderr
    const bool trace = e->host_trace;
    const auto t_call = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    struct TraceEvents : std::vector<hipEvent_t> {             // destroyed on every return path
        ~TraceEvents() { for (hipEvent_t ev : *this) if (ev) (void)hipEventDestroy(ev); }
    } tev;
    std::vector<double> thost;
    if (trace) {
        tev.resize((size_t)nchunks * 4 + 1, nullptr);
        for (hipEvent_t& ev : tev) (void)hipEventCreate(&ev);
        thost.assign((size_t)nchunks * 4, 0.0);
        (void)hipEventRecord(tev[(size_t)nchunks * 4], s_up);
    }
    // (the results of a chunk are handed to the caller's arrays by helper threads while this one packs the next chunk:
    // with small images, many keypoints per image byte, the host's copying is what bounds the call)
    HelperThread helper;
    for (int k = 0; k < nchunks + nb; k++) {
        const int b = k % nb;
        helper.join();
        if (k >= nb) {                                         // chunk k - nb used this buffer: its results are down
            if (hipEventSynchronize(e->ev_down[b]) != hipSuccess) { e->err = "a chunk of the batch failed on the device"; return fail(ORBHIP_ERR_HIP); }
            if (trace) thost[(size_t)(k - nb) * 4 + 2] = now_ms();
            const int j0 = (k - nb) * chunk;
            double* const t_done = trace ? &thost[(size_t)(k - nb) * 4 + 3] : nullptr;
            helper.t = std::thread([&, j0, t_done](int count) {
                run_sliced(e->pack_threads, j0, count, [&](int first, int n) { unpack_frames(hs, first, n, cap, kp, desc, counts); });
                if (t_done) *t_done = now_ms();
            }, std::min(chunk, nframes - j0));
        }
        if (k >= nchunks) continue;
        if (trace) thost[(size_t)k * 4] = now_ms();
        const int f0 = k * chunk, n = std::min(chunk, nframes - f0);
        uint8_t* d_in = e->d_in + (size_t)b * chunk * dev_frame;
        const uint8_t* src = imgs + (size_t)f0 * frame_stride;
        hipError_t st = hipSuccess;
        if (!pinned) pack_frames(src, row_stride, frame_stride, e->h_in + (size_t)b * chunk * dev_frame, dev_row, height, width_bytes, n, e->pack_threads);
        if (trace) (void)hipEventRecord(tev[(size_t)k * 4], s_up);
        if (pinned) {
            st = hipMemcpyAsync(d_in, src, (size_t)n * dev_frame - (dev_row - width_bytes), hipMemcpyHostToDevice, s_up);
        } else {
            st = hipMemcpyAsync(d_in, e->h_in + (size_t)b * chunk * dev_frame, (size_t)n * dev_frame, hipMemcpyHostToDevice, s_up);
        }
        if (trace) (void)hipEventRecord(tev[(size_t)k * 4 + 1], s_up);
        if (st == hipSuccess) st = hipEventRecord(e->ev_up[b], s_up);
        if (st == hipSuccess) st = hipStreamWaitEvent(s_run, e->ev_up[b], 0);
        if (st != hipSuccess) { e->err = std::string("copy of a chunk to the device: ") + hipGetErrorString(st); return fail(ORBHIP_ERR_HIP); }
        rc = run_pass(e, W, s_run, d_in, n, dev_row, dev_frame, dev_row, channels, rgb, e->d_out_kp, e->d_out_desc, e->d_out_cnt, cap, f0);
        if (rc) return fail(rc);
        if (trace) (void)hipEventRecord(tev[(size_t)k * 4 + 2], s_run);
        st = hipEventRecord(e->ev_run[b], s_run);
        if (st == hipSuccess) st = hipStreamWaitEvent(s_down, e->ev_run[b], 0);
        const size_t o = (size_t)f0 * cap;
        if (st == hipSuccess) st = hipMemcpyAsync(hs.h_cnt + (size_t)f0 * sizeof(int), e->d_out_cnt + f0, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s_down);
        if (st == hipSuccess) st = hipMemcpyAsync(hs.h_kp + o * sizeof(orbhip_keypoint), e->d_out_kp + o * 7, (size_t)n * cap * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, s_down);
        if (st == hipSuccess) st = hipMemcpyAsync(hs.h_desc + o * 32, e->d_out_desc + o * 32, (size_t)n * cap * 32, hipMemcpyDeviceToHost, s_down);
        if (trace) (void)hipEventRecord(tev[(size_t)k * 4 + 3], s_down);
        if (st == hipSuccess) st = hipEventRecord(e->ev_down[b], s_down);
        if (st != hipSuccess) { e->err = std::string("copy of a chunk's results: ") + hipGetErrorString(st); return fail(ORBHIP_ERR_HIP); }
        if (trace) thost[(size_t)k * 4 + 1] = now_ms();
    }
    helper.join();
    if (trace) {
        (void)hipDeviceSynchronize();
        std::fprintf(stderr, "orbhip host batch: %d frames, %d chunks of %d, %d in flight, %s input, %.3f ms\n", nframes, nchunks, chunk, nb,
                     pinned ? "page-locked" : "pageable", now_ms());
        std::fprintf(stderr, "  chunk: host [pack+enqueue from..to | results seen, unpacked]   device [copy up from..to, kernels to, copy down to] (ms)\n");
        for (int k = 0; k < nchunks; k++) {
            float d[4] = {0, 0, 0, 0};
            for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&d[i], tev[(size_t)nchunks * 4], tev[(size_t)k * 4 + i]);
            std::fprintf(stderr, "  %3d: host [%7.3f..%7.3f | %7.3f, %7.3f]   device [%7.3f..%7.3f, %7.3f, %7.3f]\n", k, thost[(size_t)k * 4],
                         thost[(size_t)k * 4 + 1], thost[(size_t)k * 4 + 2], thost[(size_t)k * 4 + 3], d[0], d[1], d[2], d[3]);
        }
    }
    // every chunk's results are down (the loop waited for each one's event), so the kernels are over: the status word of the whole batch
    HIP_TRY(e, hipMemcpyAsync(hs.h_status, e->d_status, sizeof(int), hipMemcpyDeviceToHost, s_run));
    HIP_TRY(e, hipStreamSynchronize(s_run));
    e->last_stream = s_run;
    HIP_TRY(e, hipEventRecord(e->ev_busy, s_run));
    collect_timing(e);
    return host_status(e, hs, s_run);
}

// The end of a host call that was enqueued with `defer`: waits for the handle's stream and hands the results over.
static int extract_host_finish(orbhip_extractor* e, const HostStage& hs, int nframes, int cap, orbhip_keypoint* kp, uint8_t* desc, int* counts) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    collect_timing(e);
    const int rc = host_status(e, hs, e->stream);
    if (rc) return rc;
    unpack_frames(hs, 0, nframes, cap, kp, desc, counts);
    return ORBHIP_OK;
}

// defer: the call returns with everything enqueued on the handle's stream (copies down included) and leaves the wait
// and the hand-over to extract_host_finish (small batches only: the chunk pipeline waits by itself).
static int extract_host_impl(orbhip_extractor* e, const uint8_t* imgs, int nframes, int width, int height,
                             size_t row_stride, size_t frame_stride, int channels, int rgb, orbhip_keypoint* kp,
                             uint8_t* desc, int* counts, int cap, HostStage* defer = nullptr) {
    if (!e || !kp || !desc || !counts || cap < 1 || nframes < 0 || (channels != 1 && channels != 3 && channels != 4))
        return ORBHIP_ERR_ARG;
    if (!imgs || width <= 0 || height <= 0 || nframes == 0) return ORBHIP_ERR_EMPTY;
    if (row_stride < (size_t)width * channels) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    if (nframes >= kPipelineMinFrames && e->host_pipeline && !defer)
        return extract_host_pipelined(e, imgs, nframes, width, height, row_stride, frame_stride, channels, rgb, kp, desc, counts, cap);
    const size_t dev_row = ((size_t)width * channels + 15) & ~(size_t)15, dev_frame = dev_row * height;
    const size_t need = dev_frame * nframes;
    HostStage hs;
    int rc = host_staging(e, need, true, nframes, cap, &hs);
    if (rc) return rc;
    const size_t kp_bytes = hs.kp_bytes, desc_bytes = hs.desc_bytes;
    // Images up, in row bands of about half a megabyte: a band travels while the next one is packed into the
    // page-locked buffer (one thread packs a camera frame in about the time its copy takes; a 2-D copy from
    // host memory would go row by row). Images that lie in page-locked memory with the staging buffer's row pitch are
    // copied from there.
    if (e->last_stream && e->last_stream != e->stream) HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_busy, 0));
    e->last_stream = e->stream;
    {
        const size_t width_bytes = (size_t)width * channels;
        if (row_stride == dev_row && host_pointer_is_pinned(imgs)) {   // (rows of another pitch are packed: a strided copy runs row by row)
            for (int f = 0; f < nframes; f++)
                HIP_TRY(e, hipMemcpyAsync(e->d_in + (size_t)f * dev_frame, imgs + (size_t)f * frame_stride, dev_frame - (dev_row - width_bytes),
                                          hipMemcpyHostToDevice, e->stream));
        } else {
            const size_t rows = (size_t)nframes * (size_t)height;
            const size_t band = std::max<size_t>(16, ((size_t)512 << 10) / dev_row);
            for (size_t r0 = 0; r0 < rows; r0 += band) {
                const size_t r1 = std::min(rows, r0 + band);
                pack_rows(imgs, row_stride, frame_stride, e->h_in, dev_row, height, width_bytes, r0, r1);
                HIP_TRY(e, hipMemcpyAsync(e->d_in + r0 * dev_row, e->h_in + r0 * dev_row, (r1 - r0) * dev_row, hipMemcpyHostToDevice, e->stream));
            }
        }
    }
    uint8_t* h_cnt = hs.h_cnt, *h_kp = hs.h_kp, *h_desc = hs.h_desc;
    // One frame through a plan that exists: replay the captured graph, or capture it when this configuration
    // comes a second time (the first call of a configuration builds plan and buffers and runs the plain way).
    const orbhip_extractor::GraphKey key = {width, height, channels, rgb, cap, e->config_gen};
    auto same = [](const orbhip_extractor::GraphKey& a, const orbhip_extractor::GraphKey& b) {
        return a.w == b.w && a.h == b.h && a.channels == b.channels && a.rgb == b.rgb && a.cap == b.cap && a.gen == b.gen;
    };
    const bool graphable = e->use_graph && nframes == 1 && !e->timing && e->overlap < 2 && e->plan_w == width && e->plan_h == height &&
                           e->plan_frames >= 1 && e->plan_lanes >= 1 &&
                           !(e->d_map_x && (channels != 1 || width != e->map_w || height != e->map_h || width < 2 || height < 2));
    bool replayed = false;
    if (graphable && !(e->graph_exec && same(key, e->graph_key)) && same(key, e->graph_seen)) {
        hipGraph_t graph = nullptr;
        HIP_TRY(e, hipStreamSynchronize(e->stream));                 // (the images are up)
        if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
        // (relaxed mode: only launches and copies on this stream are issued between here and the end of the capture,
        // and other threads - a second instance at work, src/Frame.cc:114-117 - go on making whatever calls they make)
        hipError_t st = hipStreamBeginCapture(e->stream, hipStreamCaptureModeRelaxed);
        int prc = st == hipSuccess ? run_pass(e, e->ws[0], e->stream, e->d_in, 1, dev_row, dev_frame, dev_row, channels, rgb, e->d_out_kp,
                                              e->d_out_desc, e->d_out_cnt, cap, 0) : ORBHIP_ERR_HIP;
        if (prc == ORBHIP_OK) {
            st = hipMemcpyAsync(h_cnt, e->d_out_cnt, sizeof(int), hipMemcpyDeviceToHost, e->stream);
            if (st == hipSuccess) st = hipMemcpyAsync(h_kp, e->d_out_kp, kp_bytes, hipMemcpyDeviceToHost, e->stream);
            if (st == hipSuccess) st = hipMemcpyAsync(h_desc, e->d_out_desc, desc_bytes, hipMemcpyDeviceToHost, e->stream);
            if (st == hipSuccess) st = hipMemcpyAsync(hs.h_status, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream);
        }
        const hipError_t end = hipStreamEndCapture(e->stream, &graph);   // always: the stream must leave capture mode
        if (prc == ORBHIP_OK && st == hipSuccess && end == hipSuccess && graph &&
            hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            e->graph_key = key;
            const Workspace& W = e->ws[0];
            e->graph_v0 = W.v0; e->graph_l0_src = W.l0_src;
            e->graph_l0_row_stride = W.l0_row_stride; e->graph_l0_frame_stride = W.l0_frame_stride;
        } else {
            e->graph_exec = nullptr;                               // capture is not available here: plain launches from now on
            e->use_graph = 0;
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
    }
    if (graphable && e->graph_exec && same(key, e->graph_key)) {
        if (e->last_stream && e->last_stream != e->stream) HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_busy, 0));
        e->last_stream = e->stream;
        HIP_TRY(e, hipGraphLaunch(e->graph_exec, e->stream));
        HIP_TRY(e, hipEventRecord(e->ev_busy, e->stream));
        // what run_pass notes on the host for a pass (the level-0 view is the captured one, whatever another
        // call on this handle left in the workspace since)
        Workspace& W = e->ws[0];
        W.v0 = e->graph_v0; W.l0_src = e->graph_l0_src;
        W.l0_row_stride = e->graph_l0_row_stride; W.l0_frame_stride = e->graph_l0_frame_stride;
        W.l0_in_plane = W.v0.org == nullptr;
        e->last_pass_frames = 1; e->last_ws = 0; e->pass_gen++;
        replayed = true;
    }
    e->graph_seen = key;
    if (!replayed) {
        // (the staging buffer's rows are dev_row bytes long, all of them the library's own)
        rc = extract_device_impl(e, e->d_in, nframes, width, height, dev_row, dev_frame, dev_row, channels, rgb,
                                 reinterpret_cast<orbhip_keypoint*>(e->d_out_kp), e->d_out_desc, e->d_out_cnt, cap, nullptr);
        if (rc) return rc;
        HIP_TRY(e, hipMemcpyAsync(h_cnt, e->d_out_cnt, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipMemcpyAsync(h_kp, e->d_out_kp, kp_bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipMemcpyAsync(h_desc, e->d_out_desc, desc_bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipMemcpyAsync(hs.h_status, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    }
    if (defer) { *defer = hs; return ORBHIP_OK; }
    return extract_host_finish(e, hs, nframes, cap, kp, desc, counts);
}

int orbhip_extract_batch(orbhip_extractor* e, const uint8_t* imgs, int nframes, int width, int height,
                         size_t row_stride, size_t frame_stride, orbhip_keypoint* kp, uint8_t* desc,
                         int* counts, int cap) {
    return extract_host_impl(e, imgs, nframes, width, height, row_stride, frame_stride, 1, 0, kp, desc, counts, cap);
}

int orbhip_shard_range(int nframes, int nshards, int index, int group, int* lo, int* hi) {
    if (!lo || !hi || nframes < 0 || nshards < 1 || index < 0 || index >= nshards || group < 1 || nframes % group) return ORBHIP_ERR_ARG;
    const int ngroups = nframes / group, base = ngroups / nshards, rem = ngroups % nshards;
    const int first = index * base + std::min(index, rem);
    *lo = first * group;
    *hi = (first + base + (index < rem ? 1 : 0)) * group;
    return ORBHIP_OK;
}

int orbhip_extract_batch_multi(orbhip_extractor* const* handles, int nhandles, const uint8_t* imgs, int nframes, int width, int height,
                               size_t row_stride, size_t frame_stride, orbhip_keypoint* kp, uint8_t* desc, int* counts, int cap) {
    if (!handles || nhandles < 1 || !kp || !desc || !counts || cap < 1 || nframes < 0) return ORBHIP_ERR_ARG;
    for (int i = 0; i < nhandles; i++) {
        if (!handles[i]) return ORBHIP_ERR_ARG;
        for (int j = 0; j < i; j++) if (handles[j] == handles[i]) return ORBHIP_ERR_ARG;   // a handle serves one thread at a time
    }
    if (!imgs || width <= 0 || height <= 0 || nframes == 0) return ORBHIP_ERR_EMPTY;
    // one host thread per handle, each with its own contiguous range of frames and its own slices of the outputs:
    // no exchange between the handles, the "gather" is that every thread writes where its frames belong
    std::vector<int> rcs((size_t)nhandles, ORBHIP_OK);
    auto work = [&](int i) {
        int lo = 0, hi = 0;
        orbhip_shard_range(nframes, nhandles, i, 1, &lo, &hi);
        if (hi <= lo) return;
        rcs[(size_t)i] = extract_host_impl(handles[i], imgs + (size_t)lo * frame_stride, hi - lo, width, height, row_stride, frame_stride, 1, 0,
                                           kp + (size_t)lo * cap, desc + (size_t)lo * cap * 32, counts + lo, cap);
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nhandles; i++) th.emplace_back(work, i);
    work(0);
    for (std::thread& t : th) t.join();
    for (int i = 0; i < nhandles; i++)
        if (rcs[(size_t)i]) {
            if (i) handles[0]->err = "handle " + std::to_string(i) + ": " + handles[i]->err;
            return rcs[(size_t)i];
        }
    return ORBHIP_OK;
}

int orbhip_extract(orbhip_extractor* e, const uint8_t* img, int width, int height, size_t stride,
                   orbhip_keypoint* kp, int cap, int* n, uint8_t* desc) {
    if (!e || !n) return ORBHIP_ERR_ARG;
    int count = 0;
    int rc = extract_host_impl(e, img, 1, width, height, stride, stride * (size_t)std::max(height, 0), 1, 0, kp, desc, &count, cap);
    if (rc == ORBHIP_OK) *n = count;
    return rc;
}

int orbhip_extract_color(orbhip_extractor* e, const uint8_t* img, int width, int height, size_t stride, int channels, int rgb,
                         orbhip_keypoint* kp, int cap, int* n, uint8_t* desc) {
    if (!e || !n || (channels != 3 && channels != 4)) return ORBHIP_ERR_ARG;
    int count = 0;
    int rc = extract_host_impl(e, img, 1, width, height, stride, stride * (size_t)std::max(height, 0), channels, rgb != 0, kp, desc,
                               &count, cap);
    if (rc == ORBHIP_OK) *n = count;
    return rc;
}

int orbhip_max_keypoints(const orbhip_extractor* e) {
    if (!e) return ORBHIP_ERR_ARG;
    // per level the list stops growing once it reaches its quota; one split adds at most 3 nodes,
    // and the first round can produce 4 nodes per initial node (at most ~10 initial nodes for any
    // aspect ratio the tables support)
    int total = 0;
    for (int l = 0; l < e->prm.nlevels; l++) total += std::max(e->per_level[l] + 3, 64) + 1;
    return total;
}

int orbhip_max_keypoints_for_image(const orbhip_extractor* e, int width, int height) {
    if (!e || width < 1 || height < 1) return ORBHIP_ERR_ARG;
    // The bound of orbhip_max_keypoints assumes at most 16 initial nodes per level. The distribution's first
    // round yields up to 4 nodes per initial node whatever the quota (src/ORBextractor.cc:594-665), and a level
    // has round(width / height) initial nodes of its detection window (:543): wide strips exceed that bound.
    int total = 0;
    for (int l = 0; l < e->prm.nlevels; l++) {
        const int w = cv_round((float)width * e->inv_sf[l]), h = cv_round((float)height * e->inv_sf[l]);
        const float ww = (float)(w - 2 * ORB_MINB), wh = (float)(h - 2 * ORB_MINB);
        if (ww < 30.f || wh < 30.f) continue;                  // no cell, no keypoints
        const int n_ini = (int)std::round(ww / (float)(h - 2 * ORB_MINB));
        total += std::max(std::max(e->per_level[l] + 3, 4 * n_ini), 64) + 1;
    }
    return std::max(total, 1);
}

int orbhip_get_levels(const orbhip_extractor* e) { return e ? e->prm.nlevels : ORBHIP_ERR_ARG; }
float orbhip_get_scale_factor(const orbhip_extractor* e) { return e ? (float)e->scale_factor_d : 0.f; }
static int copy_table(const orbhip_extractor* e, const std::vector<float>& t, float* out) {
    if (!e || !out) return ORBHIP_ERR_ARG;
    std::memcpy(out, t.data(), t.size() * sizeof(float));
    return ORBHIP_OK;
}
int orbhip_get_scale_factors(const orbhip_extractor* e, float* out) { return e ? copy_table(e, e->sf, out) : ORBHIP_ERR_ARG; }
int orbhip_get_inverse_scale_factors(const orbhip_extractor* e, float* out) { return e ? copy_table(e, e->inv_sf, out) : ORBHIP_ERR_ARG; }
int orbhip_get_scale_sigma_squares(const orbhip_extractor* e, float* out) { return e ? copy_table(e, e->sigma2, out) : ORBHIP_ERR_ARG; }
int orbhip_get_inverse_scale_sigma_squares(const orbhip_extractor* e, float* out) { return e ? copy_table(e, e->inv_sigma2, out) : ORBHIP_ERR_ARG; }
int orbhip_get_features_per_level(const orbhip_extractor* e, int* out) {
    if (!e || !out) return ORBHIP_ERR_ARG;
    std::memcpy(out, e->per_level.data(), e->per_level.size() * sizeof(int));
    return ORBHIP_OK;
}
int orbhip_get_umax(const orbhip_extractor* e, int* out16) {
    if (!e || !out16) return ORBHIP_ERR_ARG;
    std::memcpy(out16, e->umax, sizeof(e->umax));
    return ORBHIP_OK;
}

int orbhip_pyramid_info(const orbhip_extractor* e, int level, int* width, int* height) {
    if (!e || !width || !height || level < 0 || level >= e->prm.nlevels) return ORBHIP_ERR_ARG;
    if (!e->plan_w) return ORBHIP_ERR_STATE;
    *width = e->plan.lv[level].w;
    *height = e->plan.lv[level].h;
    return ORBHIP_OK;
}

int orbhip_pyramid_copy(orbhip_extractor* e, int frame, int level, uint8_t* dst, size_t dst_stride, int with_border) {
    if (!e || !dst || level < 0 || level >= e->prm.nlevels) return ORBHIP_ERR_ARG;
    if (!e->plan_w || frame < 0 || frame >= e->last_pass_frames) return ORBHIP_ERR_STATE;
    const LevelGeom& g = e->plan.lv[level];
    const int b = with_border ? ORB_EDGE : 0;
    if (dst_stride < (size_t)(g.w + 2 * b)) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    // The first request for a frame after a pass brings all levels of that frame into pinned host
    // memory (one asynchronous copy per level, one wait: callers such as the C++ class ask for every
    // level in turn); the rows are then cut out on the host (a 2-D copy into pageable memory would
    // go row by row).
    if (e->h_pyr_gen != e->pass_gen || e->h_pyr_frame != frame) {
        HIP_TRY(e, hipDeviceSynchronize());
        Workspace& W = e->ws[e->last_ws];
        if (!W.l0_in_plane) {
            // the pass read level 0 in place: build its bordered plane now, from the images the pass was given
            // (they must still be there: see orbhip_extract_batch_device in the header)
            if (!W.l0_src) return ORBHIP_ERR_STATE;
            launch_level0(e, W, e->stream, W.l0_src, e->last_pass_frames, W.l0_row_stride, W.l0_frame_stride, 1, 0, false);
            HIP_TRY(e, hipGetLastError());
            W.l0_in_plane = true;
        }
        if (e->pyr_bytes_per_frame > e->h_pyr_bytes) {
            if (e->h_pyr) (void)hipHostFree(e->h_pyr);
            e->h_pyr = nullptr; e->h_pyr_bytes = 0;
            HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_pyr), e->pyr_bytes_per_frame, hipHostMallocDefault));
            e->h_pyr_bytes = e->pyr_bytes_per_frame;
        }
        size_t off = 0;
        for (int l = 0; l < e->prm.nlevels; l++) {
            const LevelGeom& gl = e->plan.lv[l];
            HIP_TRY(e, hipMemcpyAsync(e->h_pyr + off, e->ws[e->last_ws].d_pyr + gl.pyr_off + (size_t)frame * gl.plane_bytes,
                                      gl.plane_bytes, hipMemcpyDeviceToHost, e->stream));
            off += gl.plane_bytes;
        }
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        e->h_pyr_gen = e->pass_gen;
        e->h_pyr_frame = frame;
    }
    size_t off = 0;
    for (int l = 0; l < level; l++) off += e->plan.lv[l].plane_bytes;
    // The planes hold the levels themselves; the 19-pixel border the reference keeps around every level
    // (copyMakeBorder with BORDER_REFLECT_101, src/ORBextractor.cc:1120-1128) is rebuilt here.
    const uint8_t* lvl = e->h_pyr + off + (size_t)ORB_OY * g.pitch + ORB_OX;
    for (int y = -b; y < g.h + b; y++) {
        const uint8_t* srow = lvl + (size_t)reflect101_host(y, g.h) * g.pitch;
        uint8_t* drow = dst + (size_t)(y + b) * dst_stride;
        std::memcpy(drow + b, srow, (size_t)g.w);
        for (int x = 1; x <= b; x++) {
            drow[b - x] = srow[reflect101_host(-x, g.w)];
            drow[b + g.w - 1 + x] = srow[reflect101_host(g.w - 1 + x, g.w)];
        }
    }
    return ORBHIP_OK;
}

int orbhip_debug_candidates(orbhip_extractor* e, int frame, int level, orbhip_candidate* out, int cap) {
    if (!e || level < 0 || level >= e->prm.nlevels) return ORBHIP_ERR_ARG;
    if (!e->plan_w || frame < 0 || frame >= e->last_pass_frames) return ORBHIP_ERR_STATE;
    const PlanDev& P = e->plan;
    const LevelGeom& g = P.lv[level];
    if (g.ncols < 1) return 0;
    DEVICE_SCOPE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    int n = 0;
    const Workspace& W = e->ws[e->last_ws];
    HIP_TRY(e, hipMemcpy(&n, W.d_key_count + frame * P.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    n = std::min(n, g.keys_cap);
    const size_t kbase = (size_t)frame * P.keys_per_frame + g.keys_off;
    const bool wide = P.qt_spill_slot[level] >= 0;              // 32-bit label words where the node arrays live in global memory
    std::vector<uint32_t> xy((size_t)std::max(n, 1)), lab((size_t)std::max(n, 1));
    std::vector<uint8_t> sc((size_t)std::max(n, 1));
    HIP_TRY(e, hipMemcpy(xy.data(), W.d_keys_xy + kbase, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(sc.data(), W.d_keys_score + kbase, (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(lab.data(), W.d_keys_label + kbase, (size_t)n * (wide ? 4 : 2), hipMemcpyDeviceToHost));
    struct Item { unsigned ord; orbhip_candidate c; };
    std::vector<Item> items;
    for (int i = 0; i < n; i++) {
        // keys the quadtree dropped by the per-cell threshold rule (src/ORBextractor.cc:808-816) carry a label word of all ones
        const uint32_t label = wide ? lab[(size_t)i] : reinterpret_cast<const uint16_t*>(lab.data())[i];
        if (label == (wide ? 0xFFFFFFFFu : 0xFFFFu)) continue;
        const int x = xy[(size_t)i] & 0xFFFF, y = xy[(size_t)i] >> 16, s = sc[(size_t)i];
        const int cx = (x - 3) / g.wcell, cy = (y - 3) / g.hcell;
        const unsigned ord = (unsigned)(((cy * g.ncols + cx) * g.hcell + (y - 3 - cy * g.hcell)) * g.wcell + (x - 3 - cx * g.wcell));
        items.push_back(Item{ord, orbhip_candidate{x, y, s}});
    }
    std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.ord < b.ord; });
    for (size_t i = 0; i < items.size() && (int)i < cap; i++) out[i] = items[i].c;
    return (int)items.size();
}

int orbhip_debug_level_keypoints(orbhip_extractor* e, int frame, int level, uint16_t* out_xys, int cap) {
    if (!e || level < 0 || level >= e->prm.nlevels) return ORBHIP_ERR_ARG;
    if (!e->plan_w || frame < 0 || frame >= e->last_pass_frames) return ORBHIP_ERR_STATE;
    const PlanDev& P = e->plan;
    const LevelGeom& g = P.lv[level];
    DEVICE_SCOPE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    const Workspace& W = e->ws[e->last_ws];
    int n = 0;
    HIP_TRY(e, hipMemcpy(&n, W.d_lvl_cnt + frame * P.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    const int m = std::max(0, std::min(std::min(n, g.node_cap), cap));
    std::vector<LevelKp> kp((size_t)std::max(m, 1));
    HIP_TRY(e, hipMemcpy(kp.data(), W.d_lvl_kp + (size_t)frame * P.kp_per_frame + g.kp_off, (size_t)m * sizeof(LevelKp), hipMemcpyDeviceToHost));
    for (int i = 0; i < m; i++) { out_xys[3 * i] = kp[(size_t)i].x; out_xys[3 * i + 1] = kp[(size_t)i].y; out_xys[3 * i + 2] = kp[(size_t)i].score; }
    return n;
}

int orbhip_enable_timing(orbhip_extractor* e, int on) {
    if (!e) return ORBHIP_ERR_ARG;
    e->timing = on != 0;
    std::memset(e->kc_ms, 0, sizeof(e->kc_ms));
    std::memset(e->kc_busy_ms, 0, sizeof(e->kc_busy_ms));
    std::memset(e->kc_launches, 0, sizeof(e->kc_launches));
    return ORBHIP_OK;
}

int orbhip_get_timing(orbhip_extractor* e, const char** names, float* ms, int* launches, int cap) {
    if (!e) return ORBHIP_ERR_ARG;
    DeviceScope device_scope(e->device);
    (void)hipDeviceSynchronize();
    collect_timing(e);
    for (int i = 0; i < KC_COUNT && i < cap; i++) {
        if (names) names[i] = kKernelNames[i];
        if (ms) ms[i] = e->kc_ms[i];
        if (launches) launches[i] = e->kc_launches[i];
    }
    return KC_COUNT;
}

int orbhip_get_timing_busy(orbhip_extractor* e, float* busy_ms, int cap) {
    if (!e) return ORBHIP_ERR_ARG;
    DeviceScope device_scope(e->device);
    (void)hipDeviceSynchronize();
    collect_timing(e);
    for (int i = 0; i < KC_COUNT && i < cap; i++)
        if (busy_ms) busy_ms[i] = e->kc_busy_ms[i];
    return KC_COUNT;
}

// Argument and state checks shared by the two batch entry points of the stereo matching.
static int stereo_batch_check(orbhip_extractor* left, orbhip_extractor* right, int nframes,
                              const orbhip_keypoint* d_kp_l, const uint8_t* d_desc_l, const int* d_cnt_l,
                              const orbhip_keypoint* d_kp_r, const uint8_t* d_desc_r, const int* d_cnt_r,
                              int cap, float mb, const float* d_uright, const float* d_depth) {
    if (!left || !right || !d_kp_l || !d_desc_l || !d_cnt_l || !d_kp_r || !d_desc_r || !d_cnt_r || !d_uright || !d_depth ||
        cap < 1 || cap > 65535 || nframes < 1 || !(mb > 0.f))
        return ORBHIP_ERR_ARG;
    orbhip_extractor* e = left;
    if (left == right) {
        e->err = "left and right must be two instances (each keeps the pyramids of its own side)";
        return ORBHIP_ERR_ARG;
    }
    if (!left->plan_w || !right->plan_w || nframes > left->last_pass_frames || nframes > right->last_pass_frames) {
        e->err = "stereo matching needs a pass of at least nframes frames on both instances";
        return ORBHIP_ERR_STATE;
    }
    if (left->device != right->device || left->plan_w != right->plan_w || left->plan_h != right->plan_h ||
        left->prm.nlevels != right->prm.nlevels || left->prm.scale_factor != right->prm.scale_factor) {
        e->err = "left and right instances differ in device, image size or pyramid parameters";
        return ORBHIP_ERR_ARG;
    }
    // the level geometry of the two plans is the same function of (size, parameters); plane
    // offsets depend on the frames a plan was made for, so each side is addressed with its own plan
    if (left->plan_frames != right->plan_frames) {
        e->err = "left and right instances must be used with the same pass size";
        return ORBHIP_ERR_ARG;
    }
    return ORBHIP_OK;
}

// Row bins of the right keypoints: bins per frame, and the height of a row band (at most
// 2 * ceil(2 * scale) + 2 rows).
static void stereo_bins(const orbhip_extractor* left, int* nbins, int* band) {
    *nbins = (left->plan.lv[0].h >> ST_BIN_SHIFT) + 2;
    *band = 2 * (int)std::ceil(2.0f * left->sf.back()) + 2;
}

// The three launches of the stereo matching on scratch buffers the caller of this function chose:
// sad [nframes * cap + nframes] ints, rinfo [nframes * cap] records, bins [nframes * (nbins + 1)] ints.
// The kernels write every scratch element before they read it: nothing has to be cleared.
static int stereo_batch_launch(orbhip_extractor* left, orbhip_extractor* right, int nframes,
                               const orbhip_keypoint* d_kp_l, const uint8_t* d_desc_l, const int* d_cnt_l,
                               const orbhip_keypoint* d_kp_r, const uint8_t* d_desc_r, const int* d_cnt_r,
                               int cap, float mb, float mbf, float* d_uright, float* d_depth, int* d_nmatch,
                               hipStream_t s, int* sad, int4* rinfo, int* bins, bool timed) {
    orbhip_extractor* e = left;
    int nbins, band;
    stereo_bins(left, &nbins, &band);
    // the two extractions may have run on other streams than `s`: wait for them here, and make their
    // streams wait for the matching before a later pass overwrites the pyramids it reads
    orbhip_extractor* const sides[2] = {left, right};
    for (orbhip_extractor* x : sides)
        if (x->last_stream && x->last_stream != s) {
            HIP_TRY(e, hipEventRecord(x->ev_order, x->last_stream));
            HIP_TRY(e, hipStreamWaitEvent(s, x->ev_order, 0));
        }
    const int ngroups = (cap + 3) / 4;                         // 4 left keypoints per workgroup
    const unsigned nwg = (unsigned)ngroups * 8u * (unsigned)((nframes + 7) / 8);
    if ((unsigned long long)nwg * (8ull * ngroups) >= (1ull << 32)) { e->err = "too many keypoints per pass"; return ORBHIP_ERR_CAPACITY; }
    const bool was_timing = e->timing;
    if (!timed) e->timing = false;                             // the timing spans take events from a pool that may have to grow
    {
    SpanGuard t(e, s, KC_STEREO);
    hipLaunchKernelGGL(k_stereo_prep, dim3(nframes), dim3(256), 0, s,
                       left->d_plan, reinterpret_cast<const StKp*>(d_kp_r), d_cnt_r, cap, nbins, rinfo, bins);
    hipLaunchKernelGGL(k_stereo_match, dim3(nwg), dim3(256), 0, s,
                       left->d_plan, rinfo, bins, nbins, band, left->ws[left->last_ws].d_pyr, right->ws[right->last_ws].d_pyr,
                       reinterpret_cast<const StKp*>(d_kp_l), d_desc_l, d_cnt_l,
                       reinterpret_cast<const StKp*>(d_kp_r), d_desc_r, cap, mb, mbf, d_uright, d_depth, sad,
                       nframes, ngroups, grid_magic(8u * (unsigned)ngroups),
                       left->ws[left->last_ws].v0, right->ws[right->last_ws].v0);
    hipLaunchKernelGGL(k_stereo_median, dim3(nframes), dim3(256), 0, s, d_cnt_l, cap, sad, d_uright, d_depth,
                       d_nmatch ? d_nmatch : sad + (size_t)nframes * cap);
    }
    e->timing = was_timing;
    HIP_TRY(e, hipGetLastError());
    for (orbhip_extractor* x : sides)
        if (x->last_stream && x->last_stream != s) {
            HIP_TRY(e, hipEventRecord(x->ev_order, s));
            HIP_TRY(e, hipStreamWaitEvent(x->last_stream, x->ev_order, 0));
        }
    return ORBHIP_OK;
}

// Scratch of the allocating entry points of the stereo matching, grown on demand (owned by the left instance).
static int stereo_scratch(orbhip_extractor* e, int nframes, int cap) {
    const size_t need = (size_t)nframes * cap + (size_t)nframes;   // + a counter per frame when the caller wants none
    if (need > e->d_st_sad_n) {
        HIP_TRY(e, hipDeviceSynchronize());
        dfree(e->d_st_sad); dfree(e->d_st_rinfo);
        e->d_st_sad_n = 0;
        HIP_TRY(e, hipMalloc(&e->d_st_sad, need * sizeof(int)));
        HIP_TRY(e, hipMalloc(&e->d_st_rinfo, (size_t)nframes * cap * sizeof(int4)));
        e->d_st_sad_n = need;
    }
    int nbins, band;
    stereo_bins(e, &nbins, &band);
    if ((size_t)nframes * (nbins + 1) > e->d_st_bins_n) {
        HIP_TRY(e, hipDeviceSynchronize());
        dfree(e->d_st_bins);
        e->d_st_bins_n = 0;
        HIP_TRY(e, hipMalloc(&e->d_st_bins, (size_t)nframes * (nbins + 1) * sizeof(int)));
        e->d_st_bins_n = (size_t)nframes * (nbins + 1);
    }
    return ORBHIP_OK;
}

int orbhip_stereo_match_batch_device(orbhip_extractor* left, orbhip_extractor* right, int nframes,
                                     const orbhip_keypoint* d_kp_l, const uint8_t* d_desc_l, const int* d_cnt_l,
                                     const orbhip_keypoint* d_kp_r, const uint8_t* d_desc_r, const int* d_cnt_r,
                                     int cap, float mb, float mbf, float* d_uright, float* d_depth, int* d_nmatch,
                                     void* stream) {
    int rc = stereo_batch_check(left, right, nframes, d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap, mb, d_uright, d_depth);
    if (rc) return rc;
    orbhip_extractor* e = left;
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    rc = stereo_scratch(left, nframes, cap);
    if (rc) return rc;
    return stereo_batch_launch(left, right, nframes, d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap, mb, mbf,
                               d_uright, d_depth, d_nmatch, s, e->d_st_sad, e->d_st_rinfo, e->d_st_bins, true);
}

int orbhip_stereo_match_batch_device_noalloc(orbhip_extractor* left, orbhip_extractor* right, int nframes,
                                             const orbhip_keypoint* d_kp_l, const uint8_t* d_desc_l, const int* d_cnt_l,
                                             const orbhip_keypoint* d_kp_r, const uint8_t* d_desc_r, const int* d_cnt_r,
                                             int cap, float mb, float mbf, float* d_uright, float* d_depth, int* d_nmatch,
                                             void* stream) {
    int rc = stereo_batch_check(left, right, nframes, d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap, mb, d_uright, d_depth);
    if (rc) return rc;
    orbhip_extractor* e = left;
    int nbins, band;
    stereo_bins(left, &nbins, &band);
    if (nframes > ORBHIP_NOALLOC_MAX_PAIRS || (unsigned long long)nframes * cap > ORBHIP_NOALLOC_MAX_SLOTS ||
        (unsigned long long)nframes * (nbins + 1) > ORBHIP_NOALLOC_MAX_BIN_ENTRIES) {
        e->err = "batch too large for the static scratch of the no-allocation entry point (ORBHIP_NOALLOC_MAX_* in orbhip.h)";
        return ORBHIP_ERR_CAPACITY;
    }
    if (!e->noalloc_scratch) { e->err = "static scratch address unknown"; return ORBHIP_ERR_STATE; }
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    // carve the device's copy of the static array (its address, not the symbol's name)
    unsigned char* base = static_cast<unsigned char*>(e->noalloc_scratch);
    int4* rinfo = reinterpret_cast<int4*>(base);
    int* sad = reinterpret_cast<int*>(base + NOALLOC_RINFO_BYTES);
    int* bins = reinterpret_cast<int*>(base + NOALLOC_RINFO_BYTES + NOALLOC_SAD_BYTES);
    return stereo_batch_launch(left, right, nframes, d_kp_l, d_desc_l, d_cnt_l, d_kp_r, d_desc_r, d_cnt_r, cap, mb, mbf,
                               d_uright, d_depth, d_nmatch, s, sad, rinfo, bins, false);
}

int orbhip_extract_stereo_batch_device(orbhip_extractor* left, orbhip_extractor* right,
                                       const uint8_t* d_imgs_l, const uint8_t* d_imgs_r, int nframes, int width, int height,
                                       size_t row_stride, size_t frame_stride,
                                       orbhip_keypoint* d_kp_l, uint8_t* d_desc_l, int* d_cnt_l,
                                       orbhip_keypoint* d_kp_r, uint8_t* d_desc_r, int* d_cnt_r, int cap,
                                       float mb, float mbf, float* d_uright, float* d_depth, int* d_nmatch, void* stream) {
    if (!left || !right || !d_kp_l || !d_desc_l || !d_cnt_l || !d_kp_r || !d_desc_r || !d_cnt_r || !d_uright || !d_depth ||
        cap < 1 || cap > 65535 || nframes < 0 || !(mb > 0.f))
        return ORBHIP_ERR_ARG;
    if (!d_imgs_l || !d_imgs_r || width <= 0 || height <= 0 || nframes == 0) return ORBHIP_ERR_EMPTY;
    if (row_stride < (size_t)width) return ORBHIP_ERR_ARG;
    orbhip_extractor* e = left;
    if (left == right) {
        e->err = "left and right must be two instances (the right pass would overwrite the left pyramids before they are matched)";
        return ORBHIP_ERR_ARG;
    }
    if (left->device != right->device || left->prm.nlevels != right->prm.nlevels || left->prm.scale_factor != right->prm.scale_factor) {
        e->err = "left and right instances differ in device or pyramid parameters";
        return ORBHIP_ERR_ARG;
    }
    orbhip_extractor* const sides[2] = {left, right};
    for (orbhip_extractor* x : sides)
        if (x->d_map_x && (width != x->map_w || height != x->map_h || width < 2 || height < 2)) {
            e->err = "rectification maps are set: the input must be a gray image of the maps' size";
            return ORBHIP_ERR_ARG;
        }
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    for (orbhip_extractor* x : sides) {
        if (x->last_stream && x->last_stream != s) HIP_TRY(e, hipStreamWaitEvent(s, x->ev_busy, 0));
        x->last_stream = s;
    }
    // One pass of both instances at a time, matched before the next pass overwrites the pyramids: the
    // matcher reads them where the extraction left them.
    const int chunk = std::min(nframes, std::min(pass_budget(left, width, height), pass_budget(right, width, height)));
    for (orbhip_extractor* x : sides)
        if (x->plan_w != width || x->plan_h != height || x->plan_frames < chunk || x->plan_lanes < 1) {
            HIP_TRY(e, hipDeviceSynchronize());
            const bool same = x->plan_w == width && x->plan_h == height;
            int rc = ensure_plan(x, width, height, std::max(chunk, same ? x->plan_frames : 0), std::max(1, same ? x->plan_lanes : 0));
            if (rc) { if (x != e) e->err = x->err; return rc; }
        }
    if (left->plan_frames != right->plan_frames) {
        // plane offsets depend on the frames a plan was made for: bring both instances to the same pass size
        const int frames = std::max(left->plan_frames, right->plan_frames);
        HIP_TRY(e, hipDeviceSynchronize());
        for (orbhip_extractor* x : sides) {
            int rc = ensure_plan(x, width, height, frames, std::max(1, x->plan_lanes));
            if (rc) { if (x != e) e->err = x->err; return rc; }
        }
    }
    int rc = stereo_scratch(left, chunk, cap);
    if (rc) return rc;
    struct PairScope {                                         // the passes below are those of a stereo pair
        orbhip_extractor *a, *b;
        PairScope(orbhip_extractor* l, orbhip_extractor* r) : a(l), b(r) { a->in_pair = b->in_pair = true; }
        ~PairScope() { a->in_pair = b->in_pair = false; }
    } pair_scope(left, right);
    // The two sides of a pass are independent until they are matched: the right one runs on the right instance's own
    // stream, forked from and joined into `s` (ORBHIP_STEREO_FORK=0: one after the other on `s`).
    const hipStream_t sr = (e->stereo_fork && right->stream != s) ? right->stream : s;
    for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int n = std::min(chunk, nframes - f0);
        if (sr != s) {                                         // (behind the matching of the previous pass, which reads the right pyramids)
            HIP_TRY(e, hipEventRecord(e->ev_fork, s));
            HIP_TRY(e, hipStreamWaitEvent(sr, e->ev_fork, 0));
        }
        rc = run_pass(left, left->ws[0], s, d_imgs_l + (size_t)f0 * frame_stride, n, row_stride, frame_stride, (size_t)width, 1, 0,
                      reinterpret_cast<float*>(d_kp_l), d_desc_l, d_cnt_l, cap, f0);
        if (rc) return rc;
        rc = run_pass(right, right->ws[0], sr, d_imgs_r + (size_t)f0 * frame_stride, n, row_stride, frame_stride, (size_t)width, 1, 0,
                      reinterpret_cast<float*>(d_kp_r), d_desc_r, d_cnt_r, cap, f0);
        if (rc) { e->err = right->err; return rc; }
        if (sr != s) {
            HIP_TRY(e, hipEventRecord(right->ws[0].done, sr));
            HIP_TRY(e, hipStreamWaitEvent(s, right->ws[0].done, 0));
        }
        const size_t o = (size_t)f0 * cap;
        rc = stereo_batch_launch(left, right, n, d_kp_l + o, d_desc_l + o * 32, d_cnt_l + f0, d_kp_r + o, d_desc_r + o * 32, d_cnt_r + f0,
                                 cap, mb, mbf, d_uright + o, d_depth + o, d_nmatch ? d_nmatch + f0 : nullptr, s,
                                 e->d_st_sad, e->d_st_rinfo, e->d_st_bins, true);
        if (rc) return rc;
    }
    for (orbhip_extractor* x : sides) HIP_TRY(e, hipEventRecord(x->ev_busy, s));
    return ORBHIP_OK;
}

int orbhip_stereo_match(orbhip_extractor* left, orbhip_extractor* right,
                        const orbhip_keypoint* kp_l, const uint8_t* desc_l, int n_l,
                        const orbhip_keypoint* kp_r, const uint8_t* desc_r, int n_r,
                        float mb, float mbf, float* uright, float* depth, int* nmatch) {
    if (!left || !right || n_l < 0 || n_r < 0 || !uright || !depth) return ORBHIP_ERR_ARG;
    if (n_l == 0) { if (nmatch) *nmatch = 0; return ORBHIP_OK; }
    if (!kp_l || !desc_l || (n_r > 0 && (!kp_r || !desc_r))) return ORBHIP_ERR_ARG;
    orbhip_extractor* e = left;
    DEVICE_SCOPE(e);
    const int cap = std::max(std::max(n_l, n_r), 1);
    const size_t kpb = (size_t)cap * sizeof(orbhip_keypoint), db = (size_t)cap * 32, fb = (size_t)cap * sizeof(float);
    // One staging block on the device and its page-locked twin: [counts, 64 bytes][mvuRight][mvDepth] (what comes back,
    // one copy) then [left keypoints][right keypoints][left descriptors][right descriptors] (what goes up with the
    // counts, one copy of the whole block); one wait at the end. (A front end calls this once per frame.)
    const size_t out_bytes = 64 + 2 * fb, need = out_bytes + 2 * kpb + 2 * db;
    if (need > e->d_st_tmp_bytes || need > e->h_st_tmp_bytes) {
        HIP_TRY(e, hipDeviceSynchronize());
        dfree(e->d_st_tmp);
        e->d_st_tmp_bytes = 0;
        if (e->h_st_tmp) (void)hipHostFree(e->h_st_tmp);
        e->h_st_tmp = nullptr; e->h_st_tmp_bytes = 0;
        HIP_TRY(e, hipMalloc(&e->d_st_tmp, need));
        e->d_st_tmp_bytes = need;
        HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_st_tmp), need, hipHostMallocDefault));
        e->h_st_tmp_bytes = need;
    }
    uint8_t* const d = e->d_st_tmp;
    uint8_t* const hbuf = e->h_st_tmp;
    int* d_cnt = reinterpret_cast<int*>(d);                        // [0] n_l, [1] n_r, [2] matches
    float* d_u = reinterpret_cast<float*>(d + 64);
    float* d_d = reinterpret_cast<float*>(d + 64 + fb);
    const size_t o_kl = out_bytes, o_kr = o_kl + kpb, o_dl = o_kr + kpb, o_dr = o_dl + db;
    const int cnt[3] = {n_l, n_r, 0};
    std::memcpy(hbuf, cnt, sizeof(cnt));
    std::memcpy(hbuf + o_kl, kp_l, (size_t)n_l * sizeof(orbhip_keypoint));
    std::memcpy(hbuf + o_dl, desc_l, (size_t)n_l * 32);
    if (n_r > 0) {
        std::memcpy(hbuf + o_kr, kp_r, (size_t)n_r * sizeof(orbhip_keypoint));
        std::memcpy(hbuf + o_dr, desc_r, (size_t)n_r * 32);
    }
    // the two extractions (host calls that waited for their results, possibly on other streams) are complete
    hipStream_t s = e->stream;
    orbhip_extractor* const sides[2] = {left, right};
    for (orbhip_extractor* x : sides)
        if (x->last_stream && x->last_stream != s) HIP_TRY(e, hipStreamWaitEvent(s, x->ev_busy, 0));
    HIP_TRY(e, hipMemcpyAsync(d, hbuf, need, hipMemcpyHostToDevice, s));
    int rc = orbhip_stereo_match_batch_device(left, right, 1, reinterpret_cast<orbhip_keypoint*>(d + o_kl), d + o_dl, d_cnt,
                                              reinterpret_cast<orbhip_keypoint*>(d + o_kr), d + o_dr, d_cnt + 1, cap, mb, mbf,
                                              d_u, d_d, d_cnt + 2, s);
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(hbuf, d, out_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    std::memcpy(uright, hbuf + 64, (size_t)n_l * sizeof(float));
    std::memcpy(depth, hbuf + 64 + fb, (size_t)n_l * sizeof(float));
    if (nmatch) *nmatch = reinterpret_cast<const int*>(hbuf)[2];
    return ORBHIP_OK;
}

// mvuRight, mvDepth and the match count per pair of a host stereo batch: device arrays and their pinned copy, owned
// by the left instance ([uright][depth] of npairs * cap floats each, then npairs counts).
static int stereo_host_outputs(orbhip_extractor* e, int npairs, int cap, size_t* fl_bytes, size_t* st_bytes) {
    *fl_bytes = (size_t)npairs * cap * sizeof(float);
    *st_bytes = 2 * *fl_bytes + (((size_t)npairs * sizeof(int) + 15) & ~(size_t)15);
    if (*st_bytes > e->st_host_bytes) {
        HIP_TRY(e, hipDeviceSynchronize());
        dfree(e->d_st_host);
        if (e->h_st_host) (void)hipHostFree(e->h_st_host);
        e->h_st_host = nullptr; e->st_host_bytes = 0;
        HIP_TRY(e, hipMalloc(&e->d_st_host, *st_bytes));
        HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_st_host), *st_bytes, hipHostMallocDefault));
        e->st_host_bytes = *st_bytes;
    }
    return ORBHIP_OK;
}

// A batch of stereo pairs in host memory as the same pipeline of chunks as extract_host_pipelined (one stream per
// stage: copies up, kernels, copies down; the left instance's streams and events carry it): a chunk is so many pairs
// that its two sides together are about 64 MB of images; both sides of a chunk are extracted and matched on the device
// (orbhip_extract_stereo_batch_device) while the neighbouring chunks travel.
static int extract_stereo_host_pipelined(orbhip_extractor* left, orbhip_extractor* right, const uint8_t* imgs_l, const uint8_t* imgs_r,
                                         int npairs, int width, int height, size_t row_stride, size_t frame_stride,
                                         orbhip_keypoint* kp_l, uint8_t* desc_l, int* cnt_l, orbhip_keypoint* kp_r, uint8_t* desc_r,
                                         int* cnt_r, int cap, float mb, float mbf, float* uright, float* depth, int* nmatch) {
    orbhip_extractor* e = left;
    orbhip_extractor* const sides[2] = {left, right};
    const uint8_t* const imgs[2] = {imgs_l, imgs_r};
    const size_t width_bytes = (size_t)width;
    const size_t dev_row = (width_bytes + 15) & ~(size_t)15, dev_frame = dev_row * height;
    const int budget = std::min(pass_budget(left, width, height), pass_budget(right, width, height));
    int chunk = e->host_chunk > 0 ? e->host_chunk : (int)std::max<size_t>(4, ((size_t)32 << 20) / dev_frame);
    chunk = std::min(chunk, (npairs + 3) / 4);
    chunk = std::max(1, std::min(chunk, budget));
    const int nchunks = (npairs + chunk - 1) / chunk;
    const int nb = std::min(e->host_lanes, nchunks);
    const bool staged_layout = row_stride == dev_row && frame_stride == dev_frame;   // (see extract_host_pipelined)
    const bool pinned[2] = {staged_layout && host_pointer_is_pinned(imgs_l), staged_layout && host_pointer_is_pinned(imgs_r)};
    HostStage hst[2];
    for (int k = 0; k < 2; k++) {
        const int rc = host_staging(sides[k], (size_t)nb * chunk * dev_frame, !pinned[k], npairs, cap, &hst[k]);
        if (rc) { if (k) e->err = right->err; return rc; }
    }
    size_t fl_bytes = 0, st_bytes = 0;
    int rc = stereo_host_outputs(e, npairs, cap, &fl_bytes, &st_bytes);
    if (rc) return rc;
    float* const d_ur = reinterpret_cast<float*>(e->d_st_host);
    float* const d_dep = reinterpret_cast<float*>(e->d_st_host + fl_bytes);
    int* const d_nm = reinterpret_cast<int*>(e->d_st_host + 2 * fl_bytes);
    float* const h_ur = reinterpret_cast<float*>(e->h_st_host);
    float* const h_dep = reinterpret_cast<float*>(e->h_st_host + fl_bytes);
    int* const h_nm = reinterpret_cast<int*>(e->h_st_host + 2 * fl_bytes);
    // earlier work of the two handles (on whatever stream) is over before their buffers are touched
    for (orbhip_extractor* x : sides)
        if (x->last_stream) HIP_TRY(e, hipEventSynchronize(x->ev_busy));
    auto fail = [&](int code) { (void)hipDeviceSynchronize(); return code; };   // nothing of this call stays in flight
    const hipStream_t s_up = e->ws[0].stream, s_run = e->stream, s_down = e->ws[1].stream;
    auto unpack = [&](int j0, int n) {
        unpack_frames(hst[0], j0, n, cap, kp_l, desc_l, cnt_l);
        unpack_frames(hst[1], j0, n, cap, kp_r, desc_r, cnt_r);
        for (int f = j0; f < j0 + n; f++) {
            const int m = std::min(cnt_l[f], cap);
            if (m <= 0) continue;
            std::memcpy(uright + (size_t)f * cap, h_ur + (size_t)f * cap, (size_t)m * sizeof(float));
            std::memcpy(depth + (size_t)f * cap, h_dep + (size_t)f * cap, (size_t)m * sizeof(float));
        }
        if (nmatch) std::memcpy(nmatch + j0, h_nm + j0, (size_t)n * sizeof(int));
    };
    // (the results of a chunk are handed to the caller's arrays by a helper thread while this one packs the next chunk:
    // with pageable images the host's copying, not the transfer, is what bounds this call)
    HelperThread helper;
    for (int k = 0; k < nchunks + nb; k++) {
        const int b = k % nb;
        helper.join();
        if (k >= nb) {                                         // chunk k - nb used this buffer: its results are down
            if (hipEventSynchronize(e->ev_down[b]) != hipSuccess) { e->err = "a chunk of the batch failed on the device"; return fail(ORBHIP_ERR_HIP); }
            const int j0 = (k - nb) * chunk;
            helper.t = std::thread([&unpack, e](int first, int count) { run_sliced(e->pack_threads, first, count, unpack); },
                                   j0, std::min(chunk, npairs - j0));
        }
        if (k >= nchunks) continue;
        const int f0 = k * chunk, n = std::min(chunk, npairs - f0);
        hipError_t st = hipSuccess;
        uint8_t* d_in[2];
        for (int side = 0; side < 2 && st == hipSuccess; side++) {
            orbhip_extractor* x = sides[side];
            d_in[side] = x->d_in + (size_t)b * chunk * dev_frame;
            const uint8_t* src = imgs[side] + (size_t)f0 * frame_stride;
            if (!pinned[side]) {
                uint8_t* h_in = x->h_in + (size_t)b * chunk * dev_frame;
                pack_frames(src, row_stride, frame_stride, h_in, dev_row, height, width_bytes, n, e->pack_threads);
                st = hipMemcpyAsync(d_in[side], h_in, (size_t)n * dev_frame, hipMemcpyHostToDevice, s_up);
            } else {
                st = hipMemcpyAsync(d_in[side], src, (size_t)n * dev_frame - (dev_row - width_bytes), hipMemcpyHostToDevice, s_up);
            }
        }
        if (st == hipSuccess) st = hipEventRecord(e->ev_up[b], s_up);
        if (st == hipSuccess) st = hipStreamWaitEvent(s_run, e->ev_up[b], 0);
        if (st != hipSuccess) { e->err = std::string("copy of a chunk to the device: ") + hipGetErrorString(st); return fail(ORBHIP_ERR_HIP); }
        const size_t o = (size_t)f0 * cap;
        rc = orbhip_extract_stereo_batch_device(left, right, d_in[0], d_in[1], n, width, height, dev_row, dev_frame,
                                                reinterpret_cast<orbhip_keypoint*>(left->d_out_kp) + o, left->d_out_desc + o * 32, left->d_out_cnt + f0,
                                                reinterpret_cast<orbhip_keypoint*>(right->d_out_kp) + o, right->d_out_desc + o * 32, right->d_out_cnt + f0,
                                                cap, mb, mbf, d_ur + o, d_dep + o, d_nm + f0, s_run);
        if (rc) return fail(rc);
        st = hipEventRecord(e->ev_run[b], s_run);
        if (st == hipSuccess) st = hipStreamWaitEvent(s_down, e->ev_run[b], 0);
        for (int side = 0; side < 2 && st == hipSuccess; side++) {
            orbhip_extractor* x = sides[side];
            const HostStage& h = hst[side];
            st = hipMemcpyAsync(h.h_cnt + (size_t)f0 * sizeof(int), x->d_out_cnt + f0, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s_down);
            if (st == hipSuccess) st = hipMemcpyAsync(h.h_kp + o * sizeof(orbhip_keypoint), x->d_out_kp + o * 7, (size_t)n * cap * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, s_down);
            if (st == hipSuccess) st = hipMemcpyAsync(h.h_desc + o * 32, x->d_out_desc + o * 32, (size_t)n * cap * 32, hipMemcpyDeviceToHost, s_down);
        }
        if (st == hipSuccess) st = hipMemcpyAsync(h_ur + o, d_ur + o, (size_t)n * cap * sizeof(float), hipMemcpyDeviceToHost, s_down);
        if (st == hipSuccess) st = hipMemcpyAsync(h_dep + o, d_dep + o, (size_t)n * cap * sizeof(float), hipMemcpyDeviceToHost, s_down);
        if (st == hipSuccess) st = hipMemcpyAsync(h_nm + f0, d_nm + f0, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s_down);
        if (st == hipSuccess) st = hipEventRecord(e->ev_down[b], s_down);
        if (st != hipSuccess) { e->err = std::string("copy of a chunk's results: ") + hipGetErrorString(st); return fail(ORBHIP_ERR_HIP); }
    }
    helper.join();
    // every chunk's results are down, so the kernels are over: the status words of the whole batch
    for (int side = 0; side < 2; side++)
        HIP_TRY(e, hipMemcpyAsync(hst[side].h_status, sides[side]->d_status, sizeof(int), hipMemcpyDeviceToHost, s_run));
    HIP_TRY(e, hipStreamSynchronize(s_run));
    collect_timing(left); collect_timing(right);
    rc = host_status(left, hst[0], s_run);
    if (rc) return rc;
    rc = host_status(right, hst[1], s_run);
    if (rc) { e->err = right->err; return rc; }
    return ORBHIP_OK;
}

int orbhip_extract_stereo_batch(orbhip_extractor* left, orbhip_extractor* right, const uint8_t* imgs_l, const uint8_t* imgs_r,
                                int npairs, int width, int height, size_t row_stride, size_t frame_stride,
                                orbhip_keypoint* kp_l, uint8_t* desc_l, int* cnt_l, orbhip_keypoint* kp_r, uint8_t* desc_r, int* cnt_r,
                                int cap, float mb, float mbf, float* uright, float* depth, int* nmatch) {
    if (!left || !right || left == right || !kp_l || !desc_l || !cnt_l || !kp_r || !desc_r || !cnt_r || !uright || !depth ||
        cap < 1 || cap > 65535 || npairs < 0 || !(mb > 0.f))
        return ORBHIP_ERR_ARG;
    if (!imgs_l || !imgs_r || width <= 0 || height <= 0 || npairs == 0) return ORBHIP_ERR_EMPTY;
    if (row_stride < (size_t)width) return ORBHIP_ERR_ARG;
    orbhip_extractor* e = left;
    if (left->device != right->device) { e->err = "left and right instances live on different devices"; return ORBHIP_ERR_ARG; }
    DEVICE_SCOPE(e);
    if (npairs >= kPipelineMinFrames && e->host_pipeline)
        return extract_stereo_host_pipelined(left, right, imgs_l, imgs_r, npairs, width, height, row_stride, frame_stride,
                                             kp_l, desc_l, cnt_l, kp_r, desc_r, cnt_r, cap, mb, mbf, uright, depth, nmatch);
    if (npairs == 1) {
        // One pair per call is what a stereo front end makes per frame (src/Frame.cc:114-122: two extractions on two
        // threads, then the matching). Each side goes through its handle's single-frame route (a replayed graph from
        // the second call on) on its handle's own stream, nothing waits in between; the matching follows on the left
        // stream behind the right side's kernels, and one wait per stream ends the call.
        HostStage hl1, hr1;
        int rc1 = extract_host_impl(left, imgs_l, 1, width, height, row_stride, frame_stride, 1, 0, kp_l, desc_l, cnt_l, cap, &hl1);
        if (rc1) return rc1;
        rc1 = extract_host_impl(right, imgs_r, 1, width, height, row_stride, frame_stride, 1, 0, kp_r, desc_r, cnt_r, cap, &hr1);
        if (rc1) { e->err = right->err; (void)hipStreamSynchronize(left->stream); return rc1; }
        auto fail1 = [&](int code) { (void)hipStreamSynchronize(left->stream); (void)hipStreamSynchronize(right->stream); return code; };
        size_t fl1 = 0, st1 = 0;
        rc1 = stereo_host_outputs(e, 1, cap, &fl1, &st1);
        if (rc1) return fail1(rc1);
        float* const d_ur1 = reinterpret_cast<float*>(e->d_st_host);
        float* const d_dep1 = reinterpret_cast<float*>(e->d_st_host + fl1);
        int* const d_nm1 = reinterpret_cast<int*>(e->d_st_host + 2 * fl1);
        hipError_t hst = hipEventRecord(right->ev_order, right->stream);
        if (hst == hipSuccess) hst = hipStreamWaitEvent(left->stream, right->ev_order, 0);
        if (hst != hipSuccess) { e->err = std::string("stereo pair: ") + hipGetErrorString(hst); return fail1(ORBHIP_ERR_HIP); }
        rc1 = orbhip_stereo_match_batch_device(left, right, 1, reinterpret_cast<orbhip_keypoint*>(left->d_out_kp), left->d_out_desc, left->d_out_cnt,
                                               reinterpret_cast<orbhip_keypoint*>(right->d_out_kp), right->d_out_desc, right->d_out_cnt, cap, mb, mbf,
                                               d_ur1, d_dep1, d_nm1, left->stream);
        if (rc1) return fail1(rc1);
        hst = hipMemcpyAsync(e->h_st_host, e->d_st_host, st1, hipMemcpyDeviceToHost, left->stream);
        if (hst != hipSuccess) { e->err = std::string("stereo pair: ") + hipGetErrorString(hst); return fail1(ORBHIP_ERR_HIP); }
        rc1 = extract_host_finish(right, hr1, 1, cap, kp_r, desc_r, cnt_r);
        if (rc1) { e->err = right->err; (void)hipStreamSynchronize(left->stream); return rc1; }
        rc1 = extract_host_finish(left, hl1, 1, cap, kp_l, desc_l, cnt_l);
        if (rc1) return rc1;
        const int m = std::min(cnt_l[0], cap);
        if (m > 0) {
            std::memcpy(uright, e->h_st_host, (size_t)m * sizeof(float));
            std::memcpy(depth, e->h_st_host + fl1, (size_t)m * sizeof(float));
        }
        if (nmatch) nmatch[0] = *reinterpret_cast<const int*>(e->h_st_host + 2 * fl1);
        return ORBHIP_OK;
    }
    const size_t dev_row = ((size_t)width + 15) & ~(size_t)15, dev_frame = dev_row * height;
    HostStage hl, hr;
    int rc = host_staging(left, dev_frame * npairs, true, npairs, cap, &hl);
    if (rc) return rc;
    rc = host_staging(right, dev_frame * npairs, true, npairs, cap, &hr);
    if (rc) { e->err = right->err; return rc; }
    size_t fl_bytes = 0, st_bytes = 0;
    rc = stereo_host_outputs(e, npairs, cap, &fl_bytes, &st_bytes);
    if (rc) return rc;
    float* d_ur = reinterpret_cast<float*>(e->d_st_host);
    float* d_dep = reinterpret_cast<float*>(e->d_st_host + fl_bytes);
    int* d_nm = reinterpret_cast<int*>(e->d_st_host + 2 * fl_bytes);
    hipStream_t s = e->stream;
    orbhip_extractor* const sides[2] = {left, right};
    const uint8_t* const imgs[2] = {imgs_l, imgs_r};
    for (int k = 0; k < 2; k++) {
        pack_frames(imgs[k], row_stride, frame_stride, sides[k]->h_in, dev_row, height, (size_t)width, npairs, e->pack_threads);
        HIP_TRY(e, hipMemcpyAsync(sides[k]->d_in, sides[k]->h_in, dev_frame * npairs, hipMemcpyHostToDevice, s));
    }
    rc = orbhip_extract_stereo_batch_device(left, right, left->d_in, right->d_in, npairs, width, height, dev_row, dev_frame,
                                            reinterpret_cast<orbhip_keypoint*>(left->d_out_kp), left->d_out_desc, left->d_out_cnt,
                                            reinterpret_cast<orbhip_keypoint*>(right->d_out_kp), right->d_out_desc, right->d_out_cnt, cap,
                                            mb, mbf, d_ur, d_dep, d_nm, s);
    if (rc) { (void)hipDeviceSynchronize(); return rc; }
    const HostStage* const hs[2] = {&hl, &hr};
    for (int k = 0; k < 2; k++) {
        HIP_TRY(e, hipMemcpyAsync(hs[k]->h_cnt, sides[k]->d_out_cnt, (size_t)npairs * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpyAsync(hs[k]->h_kp, sides[k]->d_out_kp, hs[k]->kp_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpyAsync(hs[k]->h_desc, sides[k]->d_out_desc, hs[k]->desc_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpyAsync(hs[k]->h_status, sides[k]->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(e, hipMemcpyAsync(e->h_st_host, e->d_st_host, st_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    collect_timing(left); collect_timing(right);
    rc = host_status(left, hl, s);
    if (rc) return rc;
    rc = host_status(right, hr, s);
    if (rc) { e->err = right->err; return rc; }
    unpack_frames(hl, 0, npairs, cap, kp_l, desc_l, cnt_l);
    unpack_frames(hr, 0, npairs, cap, kp_r, desc_r, cnt_r);
    const float* h_ur = reinterpret_cast<const float*>(e->h_st_host);
    const float* h_dep = reinterpret_cast<const float*>(e->h_st_host + fl_bytes);
    for (int f = 0; f < npairs; f++) {
        const int m = std::min(cnt_l[f], cap);
        if (m <= 0) continue;
        std::memcpy(uright + (size_t)f * cap, h_ur + (size_t)f * cap, (size_t)m * sizeof(float));
        std::memcpy(depth + (size_t)f * cap, h_dep + (size_t)f * cap, (size_t)m * sizeof(float));
    }
    if (nmatch) std::memcpy(nmatch, e->h_st_host + 2 * fl_bytes, (size_t)npairs * sizeof(int));
    return ORBHIP_OK;
}

int orbhip_extract_stereo_batch_multi(orbhip_extractor* const* left, orbhip_extractor* const* right, int nhandles,
                                      const uint8_t* imgs_l, const uint8_t* imgs_r, int npairs, int width, int height,
                                      size_t row_stride, size_t frame_stride,
                                      orbhip_keypoint* kp_l, uint8_t* desc_l, int* cnt_l, orbhip_keypoint* kp_r, uint8_t* desc_r, int* cnt_r,
                                      int cap, float mb, float mbf, float* uright, float* depth, int* nmatch) {
    if (!left || !right || nhandles < 1 || npairs < 0 || cap < 1) return ORBHIP_ERR_ARG;
    for (int i = 0; i < nhandles; i++) {
        if (!left[i] || !right[i]) return ORBHIP_ERR_ARG;
        for (int j = 0; j < i; j++)
            if (left[j] == left[i] || right[j] == right[i] || left[j] == right[i] || right[j] == left[i]) return ORBHIP_ERR_ARG;
    }
    if (!imgs_l || !imgs_r || width <= 0 || height <= 0 || npairs == 0) return ORBHIP_ERR_EMPTY;
    // both images of a pair go to the same pair of handles (the matcher reads both pyramids on one device)
    std::vector<int> rcs((size_t)nhandles, ORBHIP_OK);
    auto work = [&](int i) {
        int lo = 0, hi = 0;
        orbhip_shard_range(npairs, nhandles, i, 1, &lo, &hi);
        if (hi <= lo) return;
        const size_t o = (size_t)lo * cap;
        rcs[(size_t)i] = orbhip_extract_stereo_batch(left[i], right[i], imgs_l + (size_t)lo * frame_stride, imgs_r + (size_t)lo * frame_stride,
                                                     hi - lo, width, height, row_stride, frame_stride, kp_l + o, desc_l + o * 32, cnt_l + lo,
                                                     kp_r + o, desc_r + o * 32, cnt_r + lo, cap, mb, mbf, uright + o, depth + o,
                                                     nmatch ? nmatch + lo : nullptr);
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nhandles; i++) th.emplace_back(work, i);
    work(0);
    for (std::thread& t : th) t.join();
    for (int i = 0; i < nhandles; i++)
        if (rcs[(size_t)i]) {
            if (i) left[0]->err = "handle pair " + std::to_string(i) + ": " + left[i]->err;
            return rcs[(size_t)i];
        }
    return ORBHIP_OK;
}

int orbhip_image_bounds(const orbhip_camera* cam, int width, int height, float bounds[4]) {
    if (!cam || !bounds || width < 1 || height < 1) return ORBHIP_ERR_ARG;
    if (cam->k1 != 0.0f) {                                     // src/Frame.cc:453-470
        const CamDev c = {cam->fx, cam->fy, cam->cx, cam->cy, cam->k1, cam->k2, cam->p1, cam->p2, cam->k3};
        float x[4] = {0.0f, (float)width, 0.0f, (float)width}, y[4] = {0.0f, 0.0f, (float)height, (float)height};
        for (int i = 0; i < 4; i++) undistort_point(c, x[i], y[i]);
        bounds[0] = std::min(x[0], x[2]); bounds[1] = std::max(x[1], x[3]);
        bounds[2] = std::min(y[0], y[1]); bounds[3] = std::max(y[2], y[3]);
    } else {                                                   // src/Frame.cc:471-477
        bounds[0] = 0.0f; bounds[1] = (float)width; bounds[2] = 0.0f; bounds[3] = (float)height;
    }
    return ORBHIP_OK;
}

int orbhip_stereo_from_rgbd_batch_device(orbhip_extractor* e, int nframes, const orbhip_keypoint* d_kp, const orbhip_keypoint* d_kp_un,
                                         const int* d_counts, int cap, const void* d_depth, int depth_is_u16, int width, int height,
                                         size_t row_stride, size_t frame_stride, float depth_map_factor, float mbf,
                                         float* d_uright, float* d_depth_out, void* stream) {
    if (!e || nframes < 1 || cap < 1 || !d_kp || !d_kp_un || !d_counts || !d_depth || width < 1 || height < 1 || !d_uright || !d_depth_out ||
        row_stride < (size_t)width * (depth_is_u16 ? 2 : 4))
        return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    // src/Tracking.cc (GrabImageRGBD): the map is converted when the factor is not 1 or the type is not CV_32F
    const int convert = (std::fabs(depth_map_factor - 1.0f) > 1e-5) || depth_is_u16;
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((cap + 255) / 256, nframes), dim3(256), 0, s, reinterpret_cast<const GridKp*>(d_kp),
                       reinterpret_cast<const GridKp*>(d_kp_un), d_counts, cap, static_cast<const uint8_t*>(d_depth), row_stride,
                       frame_stride, width, height, depth_is_u16 ? 1 : 0, convert, depth_map_factor, mbf, d_uright, d_depth_out);
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_unproject_stereo_batch_device(orbhip_extractor* e, int nframes, const orbhip_keypoint* d_kp_un, const float* d_depth,
                                         const int* d_counts, int cap, const float* d_pose_wc, float fx, float fy, float cx, float cy,
                                         float* d_world, uint8_t* d_valid, void* stream) {
    if (!e || nframes < 1 || cap < 1 || !d_kp_un || !d_depth || !d_counts || !d_pose_wc || !d_world || !d_valid || !(fx != 0.0f) ||
        !(fy != 0.0f))
        return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;          // src/Frame.cc: invfx = 1.0f/fx in the constructors
    hipLaunchKernelGGL(k_unproject_stereo, dim3((cap + 255) / 256, nframes), dim3(256), 0, s, reinterpret_cast<const GridKp*>(d_kp_un),
                       d_depth, d_counts, cap, d_pose_wc, cx, cy, invfx, invfy, d_world, d_valid);
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_undistort_grid_batch_device(orbhip_extractor* e, int nframes, const orbhip_keypoint* d_kp,
                                       const int* d_counts, int cap, const orbhip_camera* cam,
                                       const float bounds[4], orbhip_keypoint* d_kp_un, int* d_grid_start,
                                       int* d_grid_idx, void* stream) {
    if (!e || nframes < 1 || !d_kp || !d_counts || cap < 1 || !cam || !bounds || !d_kp_un || !d_grid_start || !d_grid_idx ||
        !(bounds[1] > bounds[0]) || !(bounds[3] > bounds[2]))
        return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    const CamDev c = {cam->fx, cam->fy, cam->cx, cam->cy, cam->k1, cam->k2, cam->p1, cam->p2, cam->k3};
    // mfGridElementWidthInv / mfGridElementHeightInv (src/Frame.cc:128-129)
    const float inv_w = static_cast<float>(GRID_COLS) / (bounds[1] - bounds[0]);
    const float inv_h = static_cast<float>(GRID_ROWS) / (bounds[3] - bounds[2]);
    hipLaunchKernelGGL(k_undistort_grid, dim3(nframes), dim3(256), 0, s, reinterpret_cast<const GridKp*>(d_kp), d_counts, cap, c,
                       bounds[0], bounds[2], inv_w, inv_h, reinterpret_cast<GridKp*>(d_kp_un), d_grid_start, d_grid_idx);
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_undistort_grid(orbhip_extractor* e, const orbhip_keypoint* kp, int n, const orbhip_camera* cam,
                          const float bounds[4], orbhip_keypoint* kp_un, int* grid_start, int* grid_idx) {
    if (!e || n < 0 || !cam || !bounds || !grid_start || (n > 0 && (!kp || !kp_un || !grid_idx))) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    const int cap = std::max(n, 1);
    const size_t kpb = (size_t)cap * sizeof(orbhip_keypoint), gsb = (size_t)(GRID_CELLS + 1) * sizeof(int), gib = (size_t)cap * sizeof(int);
    const size_t need = 64 + 2 * kpb + gsb + gib;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p); p += 64;
    orbhip_keypoint* d_k = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    orbhip_keypoint* d_u = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    int* d_gs = reinterpret_cast<int*>(p); p += gsb;
    int* d_gi = reinterpret_cast<int*>(p);
    io.up(d_cnt, &n, sizeof(int));
    if (n > 0) io.up(d_k, kp, (size_t)n * sizeof(orbhip_keypoint));
    HIP_TRY(e, io.send());
    int rc = orbhip_undistort_grid_batch_device(e, 1, d_k, d_cnt, cap, cam, bounds, d_u, d_gs, d_gi, nullptr);
    if (rc) return rc;
    if (n > 0) {
        io.down(kp_un, d_u, (size_t)n * sizeof(orbhip_keypoint));
        io.down(grid_idx, d_gi, (size_t)n * sizeof(int));
    }
    io.down(grid_start, d_gs, gsb);
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

int orbhip_search_for_initialization_batch_device(orbhip_extractor* e, int nframes,
                                                  const orbhip_keypoint* d_kp_un1, const uint8_t* d_desc1, const int* d_cnt1,
                                                  const orbhip_keypoint* d_kp_un2, const uint8_t* d_desc2, const int* d_cnt2,
                                                  int cap, const float bounds[4], float* d_prev_matched, int window_size,
                                                  float nnratio, int check_orientation, int* d_matches12, int* d_nmatches,
                                                  void* stream) {
    if (!e || nframes < 1 || !d_kp_un1 || !d_desc1 || !d_cnt1 || !d_kp_un2 || !d_desc2 || !d_cnt2 || cap < 1 || cap > 65534 ||
        !bounds || !(bounds[1] > bounds[0]) || !(bounds[3] > bounds[2]) || !d_prev_matched || !d_matches12 || !d_nmatches)
        return ORBHIP_ERR_ARG;
    const size_t lds = (size_t)MT_LDS_PER_KP * cap + 16;
    if (lds > 160 * 1024 - 512) { e->err = "too many keypoints per frame for the on-chip matcher state"; return ORBHIP_ERR_CAPACITY; }
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    const float inv_w = static_cast<float>(GRID_COLS) / (bounds[1] - bounds[0]);
    const float inv_h = static_cast<float>(GRID_ROWS) / (bounds[3] - bounds[2]);
    // the 16-wave kernel with candidate lists when its state fits (140 KB: about 3900 keypoints per frame), else one wave per pair
    const size_t lds_par = (size_t)MI_LDS_PER_KP * cap + 16;
    if (lds_par <= 140 * 1024) {
        if (lds_par > 40 * 1024)
            HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_init_par), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_par));
        hipLaunchKernelGGL(k_search_init_par, dim3(nframes), dim3(MP_THREADS), lds_par, s,
                           reinterpret_cast<const GridKp*>(d_kp_un1), d_desc1, d_cnt1, reinterpret_cast<const GridKp*>(d_kp_un2), d_desc2,
                           d_cnt2, cap, bounds[0], bounds[2], inv_w, inv_h, (float)window_size, nnratio, check_orientation,
                           reinterpret_cast<float2*>(d_prev_matched), d_matches12, d_nmatches);
    } else {
        if (lds > 64 * 1024)
            HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_init), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_search_init, dim3(nframes), dim3(64), lds, s,
                           reinterpret_cast<const GridKp*>(d_kp_un1), d_desc1, d_cnt1, reinterpret_cast<const GridKp*>(d_kp_un2), d_desc2,
                           d_cnt2, cap, bounds[0], bounds[2], inv_w, inv_h, (float)window_size, nnratio, check_orientation,
                           reinterpret_cast<float2*>(d_prev_matched), d_matches12, d_nmatches);
    }
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_for_initialization(orbhip_extractor* e, const orbhip_keypoint* kp_un1, const uint8_t* desc1, int n1,
                                     const orbhip_keypoint* kp_un2, const uint8_t* desc2, int n2, const float bounds[4],
                                     float* prev_matched, int window_size, float nnratio, int check_orientation,
                                     int* matches12, int* nmatches) {
    if (!e || n1 < 0 || n2 < 0 || !bounds || !nmatches) return ORBHIP_ERR_ARG;
    if (n1 == 0) { *nmatches = 0; return ORBHIP_OK; }
    if (!kp_un1 || !desc1 || !prev_matched || !matches12 || (n2 > 0 && (!kp_un2 || !desc2))) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    const int cap = std::max(std::max(n1, n2), 1);
    const size_t kpb = (size_t)cap * sizeof(orbhip_keypoint), db = (size_t)cap * 32;
    const size_t pb = (size_t)cap * 2 * sizeof(float), mb = (size_t)cap * sizeof(int);
    const size_t need = 64 + 2 * kpb + 2 * db + pb + mb;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p); p += 64;            // [0] n1, [1] n2, [2] matches
    orbhip_keypoint* d_k1 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    orbhip_keypoint* d_k2 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    uint8_t* d_d1 = p; p += db;
    uint8_t* d_d2 = p; p += db;
    float* d_prev = reinterpret_cast<float*>(p); p += pb;
    int* d_m12 = reinterpret_cast<int*>(p);
    const int cnt[3] = {n1, n2, 0};
    io.up(d_cnt, cnt, sizeof(cnt));
    io.up(d_k1, kp_un1, (size_t)n1 * sizeof(orbhip_keypoint));
    io.up(d_d1, desc1, (size_t)n1 * 32);
    io.up(d_prev, prev_matched, (size_t)n1 * 2 * sizeof(float));
    if (n2 > 0) {
        io.up(d_k2, kp_un2, (size_t)n2 * sizeof(orbhip_keypoint));
        io.up(d_d2, desc2, (size_t)n2 * 32);
    }
    HIP_TRY(e, io.send());
    int rc = orbhip_search_for_initialization_batch_device(e, 1, d_k1, d_d1, d_cnt, d_k2, d_d2, d_cnt + 1, cap, bounds, d_prev,
                                                           window_size, nnratio, check_orientation, d_m12, d_cnt + 2, nullptr);
    if (rc) return rc;
    io.down(prev_matched, d_prev, (size_t)n1 * 2 * sizeof(float));
    io.down(matches12, d_m12, (size_t)n1 * sizeof(int));
    io.down(nmatches, d_cnt + 2, sizeof(int));
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

int orbhip_set_rectification(orbhip_extractor* e, const float* map_x, const float* map_y, int width, int height) {
    if (!e) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    HIP_TRY(e, hipDeviceSynchronize());                        // no pass may still read the old maps
    e->config_gen++;                                           // a captured graph holds the old maps (or none)
    dfree(e->d_map_x); dfree(e->d_map_y);
    e->map_w = e->map_h = 0;
    if (!map_x && !map_y) return ORBHIP_OK;
    if (!map_x || !map_y || width < 2 || height < 2) return ORBHIP_ERR_ARG;
    const size_t bytes = (size_t)width * height * sizeof(float);
    HIP_TRY(e, hipMalloc(&e->d_map_x, bytes));
    HIP_TRY(e, hipMalloc(&e->d_map_y, bytes));
    HIP_TRY(e, hipMemcpy(e->d_map_x, map_x, bytes, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_map_y, map_y, bytes, hipMemcpyHostToDevice));
    e->map_w = width; e->map_h = height;
    return ORBHIP_OK;
}

int orbhip_debug_sincos(orbhip_extractor* e, uint32_t first_bits, uint32_t count, float* d_cos, float* d_sin, void* stream) {
    if (!e || !d_cos || !d_sin || count == 0 || count > (1u << 30)) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    hipLaunchKernelGGL(k_debug_sincos, dim3((count + 255u) / 256u), dim3(256), 0, s, first_bits, count, d_cos, d_sin);
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_projection_level_mode(const float* Tcw_cur, const float* Tcw_last, float mb, int mono) {
    if (!Tcw_cur || !Tcw_last) return ORBHIP_ERR_ARG;
    // twc = -Rcw^T tcw, tlc = Rlw twc + tlw (src/ORBmatcher.cc:1330-1335): CV_32F matrix products; the
    // transposed one goes through the general path (accumulated in double, stored as float)
    float twc[3];
    for (int r = 0; r < 3; r++) {
        double acc = 0;
        for (int k = 0; k < 3; k++) acc += (double)Tcw_cur[3 * k + r] * (double)Tcw_cur[9 + k];
        twc[r] = (float)(-acc);
    }
    // Rlw * twc + tlw has no transposed operand: the 3x3 by 3x1 product takes the small-matrix path
    // like Rcw * x3Dw + tcw (float products summed in float, the addend added in double)
    const float rz = Tcw_last[6] * twc[0] + Tcw_last[7] * twc[1] + Tcw_last[8] * twc[2];
    const float tlc_z = (float)((double)rz + (double)Tcw_last[11]);
    if (mono) return 0;
    if (tlc_z > mb) return 1;                                  // bForward
    if (-tlc_z > mb) return 2;                                 // bBackward
    return 0;
}

// Global-memory state of the matchers for frames whose state does not fit a workgroup's LDS: `bytes` per frame.
static int match_spill(orbhip_extractor* e, size_t bytes) {
    if (bytes > e->d_match_spill_bytes) {
        HIP_TRY(e, hipDeviceSynchronize());
        dfree(e->d_match_spill);
        e->d_match_spill_bytes = 0;
        HIP_TRY(e, hipMalloc(&e->d_match_spill, bytes));
        e->d_match_spill_bytes = bytes;
    }
    return ORBHIP_OK;
}

int orbhip_search_by_projection_batch_device(orbhip_extractor* e, int nframes, int cap,
                                             const orbhip_keypoint* d_kp_last, const float* d_world_last,
                                             const uint8_t* d_flags_last, const uint8_t* d_desc_last, const int* d_cnt_last,
                                             const orbhip_keypoint* d_kp_cur, const uint8_t* d_desc_cur,
                                             const float* d_uright_cur, const int* d_cnt_cur, const float* d_Tcw,
                                             const int* d_level_mode, const orbhip_projection_params* prm,
                                             int* d_match_cur, int* d_nmatches, void* stream) {
    if (!e || nframes < 1 || cap < 1 || cap > 65534 || !d_kp_last || !d_world_last || !d_flags_last || !d_desc_last || !d_cnt_last ||
        !d_kp_cur || !d_desc_cur || !d_cnt_cur || !d_Tcw || !prm || !d_match_cur || !d_nmatches ||
        !(prm->bounds[1] > prm->bounds[0]) || !(prm->bounds[3] > prm->bounds[2]))
        return ORBHIP_ERR_ARG;
    const size_t lds = (size_t)MP_LDS_PER_KP * cap + 16;
    const bool spill = lds > 160 * 1024 - 20 * 1024;                // the state then lives in global memory
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    if (!spill && lds > 40 * 1024)
        HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_proj<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProjArgs a;
    a.fx = prm->fx; a.fy = prm->fy; a.cx = prm->cx; a.cy = prm->cy; a.mbf = prm->mbf;
    a.min_x = prm->bounds[0]; a.max_x = prm->bounds[1]; a.min_y = prm->bounds[2]; a.max_y = prm->bounds[3];
    a.inv_w = static_cast<float>(GRID_COLS) / (prm->bounds[1] - prm->bounds[0]);   // src/Frame.cc:128-129
    a.inv_h = static_cast<float>(GRID_ROWS) / (prm->bounds[3] - prm->bounds[2]);
    a.th = prm->th;
    a.check_orientation = prm->check_orientation;
    ScaleTab sf;
    for (int l = 0; l < ORB_MAX_LEVELS; l++) sf.v[l] = e->sf[std::min(l, e->prm.nlevels - 1)];
    if (spill) {
        const size_t stride = (lds + 255) & ~(size_t)255;
        int rc = match_spill(e, stride * (size_t)nframes);
        if (rc) return rc;
        hipLaunchKernelGGL((k_search_proj<0, true>), dim3(nframes), dim3(MP_THREADS), 0, s, reinterpret_cast<const GridKp*>(d_kp_last), d_world_last,
                           d_flags_last, d_desc_last, d_cnt_last, reinterpret_cast<const GridKp*>(d_kp_cur), d_desc_cur, d_uright_cur,
                           d_cnt_cur, cap, d_Tcw, d_level_mode, a, sf, KeyframeArgs{nullptr, nullptr, nullptr, 0, 0.0f, 0}, d_match_cur, d_nmatches, e->d_match_spill, stride);
    } else {
        hipLaunchKernelGGL(k_search_proj<0>, dim3(nframes), dim3(MP_THREADS), lds, s, reinterpret_cast<const GridKp*>(d_kp_last), d_world_last,
                           d_flags_last, d_desc_last, d_cnt_last, reinterpret_cast<const GridKp*>(d_kp_cur), d_desc_cur, d_uright_cur,
                           d_cnt_cur, cap, d_Tcw, d_level_mode, a, sf, KeyframeArgs{nullptr, nullptr, nullptr, 0, 0.0f, 0}, d_match_cur, d_nmatches);
    }
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_by_projection(orbhip_extractor* e, const orbhip_keypoint* kp_last, const float* world_last,
                                const uint8_t* flags_last, const uint8_t* desc_last, int n1,
                                const orbhip_keypoint* kp_cur, const uint8_t* desc_cur, const float* uright_cur, int n2,
                                const float* Tcw, int level_mode, const orbhip_projection_params* prm,
                                int* match_cur, int* nmatches) {
    if (!e || n1 < 0 || n2 < 0 || !Tcw || !prm || !nmatches) return ORBHIP_ERR_ARG;
    if (n2 > 0 && !match_cur) return ORBHIP_ERR_ARG;
    for (int i = 0; i < n2; i++) match_cur[i] = -1;
    if (n1 == 0 || n2 == 0) { *nmatches = 0; return ORBHIP_OK; }
    if (!kp_last || !world_last || !flags_last || !desc_last || !kp_cur || !desc_cur) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    const int cap = std::max(n1, n2);
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t kpb = up((size_t)cap * sizeof(orbhip_keypoint)), db = up((size_t)cap * 32), wb = up((size_t)cap * 12);
    const size_t fb = up((size_t)cap), ub = up((size_t)cap * 4), mb = up((size_t)cap * 4);
    const size_t need = 128 + 2 * kpb + 2 * db + wb + fb + ub + mb;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p);                     // [0] n1, [1] n2, [2] matches, [3] level mode
    float* d_T = reinterpret_cast<float*>(p + 64); p += 128;
    orbhip_keypoint* d_k1 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    orbhip_keypoint* d_k2 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    uint8_t* d_d1 = p; p += db;
    uint8_t* d_d2 = p; p += db;
    float* d_w = reinterpret_cast<float*>(p); p += wb;
    uint8_t* d_f = p; p += fb;
    float* d_u = reinterpret_cast<float*>(p); p += ub;
    int* d_m = reinterpret_cast<int*>(p);
    const int cnt[4] = {n1, n2, 0, level_mode};
    io.up(d_cnt, cnt, sizeof(cnt));
    io.up(d_T, Tcw, 12 * sizeof(float));
    io.up(d_k1, kp_last, (size_t)n1 * sizeof(orbhip_keypoint));
    io.up(d_d1, desc_last, (size_t)n1 * 32);
    io.up(d_w, world_last, (size_t)n1 * 12);
    io.up(d_f, flags_last, (size_t)n1);
    io.up(d_k2, kp_cur, (size_t)n2 * sizeof(orbhip_keypoint));
    io.up(d_d2, desc_cur, (size_t)n2 * 32);
    if (uright_cur) io.up(d_u, uright_cur, (size_t)n2 * 4);
    HIP_TRY(e, io.send());
    int rc = orbhip_search_by_projection_batch_device(e, 1, cap, d_k1, d_w, d_f, d_d1, d_cnt, d_k2, d_d2, uright_cur ? d_u : nullptr,
                                                      d_cnt + 1, d_T, d_cnt + 3, prm, d_m, d_cnt + 2, nullptr);
    if (rc) return rc;
    io.down(match_cur, d_m, (size_t)n2 * sizeof(int));
    io.down(nmatches, d_cnt + 2, sizeof(int));
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

int orbhip_search_by_projection_keyframe_batch_device(orbhip_extractor* e, int nframes, int cap,
                                                      const orbhip_keypoint* d_kp_kf, const float* d_world_kf,
                                                      const float* d_dist_range_kf, const uint8_t* d_flags_kf,
                                                      const uint8_t* d_desc_kf, const int* d_cnt_kf,
                                                      const orbhip_keypoint* d_kp_cur, const uint8_t* d_desc_cur,
                                                      const uint8_t* d_occupied_cur, const int* d_cnt_cur, const float* d_Tcw,
                                                      const orbhip_projection_params* prm, int orb_dist,
                                                      int* d_match_cur, int* d_nmatches, void* stream) {
    if (!e || nframes < 1 || cap < 1 || cap > 65534 || !d_kp_kf || !d_world_kf || !d_dist_range_kf || !d_flags_kf || !d_desc_kf ||
        !d_cnt_kf || !d_kp_cur || !d_desc_cur || !d_cnt_cur || !d_Tcw || !prm || !d_match_cur || !d_nmatches || orb_dist < 0 ||
        !(prm->bounds[1] > prm->bounds[0]) || !(prm->bounds[3] > prm->bounds[2]))
        return ORBHIP_ERR_ARG;
    const size_t lds = (size_t)MP_LDS_PER_KP * cap + 16;
    const bool spill = lds > 160 * 1024 - 20 * 1024;                // the state then lives in global memory
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    if (!spill && lds > 40 * 1024)
        HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_proj<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProjArgs a;
    a.fx = prm->fx; a.fy = prm->fy; a.cx = prm->cx; a.cy = prm->cy; a.mbf = prm->mbf;
    a.min_x = prm->bounds[0]; a.max_x = prm->bounds[1]; a.min_y = prm->bounds[2]; a.max_y = prm->bounds[3];
    a.inv_w = static_cast<float>(GRID_COLS) / (prm->bounds[1] - prm->bounds[0]);   // src/Frame.cc:128-129
    a.inv_h = static_cast<float>(GRID_ROWS) / (prm->bounds[3] - prm->bounds[2]);
    a.th = prm->th;
    a.check_orientation = prm->check_orientation;
    ScaleTab sf;
    for (int l = 0; l < ORB_MAX_LEVELS; l++) sf.v[l] = e->sf[std::min(l, e->prm.nlevels - 1)];
    const KeyframeArgs kf{reinterpret_cast<const float2*>(d_dist_range_kf), d_occupied_cur, nullptr, orb_dist,
                          std::log((float)e->scale_factor_d), e->prm.nlevels};
    if (spill) {
        const size_t stride = (lds + 255) & ~(size_t)255;
        int rc = match_spill(e, stride * (size_t)nframes);
        if (rc) return rc;
        hipLaunchKernelGGL((k_search_proj<1, true>), dim3(nframes), dim3(MP_THREADS), 0, s, reinterpret_cast<const GridKp*>(d_kp_kf), d_world_kf,
                           d_flags_kf, d_desc_kf, d_cnt_kf, reinterpret_cast<const GridKp*>(d_kp_cur), d_desc_cur, (const float*)nullptr,
                           d_cnt_cur, cap, d_Tcw, (const int*)nullptr, a, sf, kf, d_match_cur, d_nmatches, e->d_match_spill, stride);
    } else {
        hipLaunchKernelGGL(k_search_proj<1>, dim3(nframes), dim3(MP_THREADS), lds, s, reinterpret_cast<const GridKp*>(d_kp_kf), d_world_kf,
                           d_flags_kf, d_desc_kf, d_cnt_kf, reinterpret_cast<const GridKp*>(d_kp_cur), d_desc_cur, (const float*)nullptr,
                           d_cnt_cur, cap, d_Tcw, (const int*)nullptr, a, sf, kf, d_match_cur, d_nmatches);
    }
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_decompose_sim3(const float* Scw, float* Tcw) {
    if (!Scw || !Tcw) return ORBHIP_ERR_ARG;
    // src/ORBmatcher.cc:296-299: scw = sqrt(sRcw.row(0).dot(sRcw.row(0))) (dot in double, scw a float), then
    // sRcw / scw and t / scw as Mat / scalar: every element times the float value of 1.0 / scw
    const double d = (double)Scw[0] * (double)Scw[0] + (double)Scw[1] * (double)Scw[1] + (double)Scw[2] * (double)Scw[2];
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    for (int i = 0; i < 12; i++) Tcw[i] = Scw[i] * inv;
    return ORBHIP_OK;
}

int orbhip_search_by_projection_sim3_batch_device(orbhip_extractor* e, int nframes, int cap, const float* d_world,
                                                  const float* d_normal, const float* d_dist_range, const uint8_t* d_flags,
                                                  const uint8_t* d_desc, const int* d_cnt_pts, const orbhip_keypoint* d_kp_kf,
                                                  const uint8_t* d_desc_kf, const uint8_t* d_occupied_kf, const int* d_cnt_kf,
                                                  const float* d_Tcw, const orbhip_projection_params* prm, int* d_match_kf,
                                                  int* d_nmatches, void* stream) {
    if (!e || nframes < 1 || cap < 1 || cap > 65534 || !d_world || !d_normal || !d_dist_range || !d_flags || !d_desc || !d_cnt_pts ||
        !d_kp_kf || !d_desc_kf || !d_cnt_kf || !d_Tcw || !prm || !d_match_kf || !d_nmatches ||
        !(prm->bounds[1] > prm->bounds[0]) || !(prm->bounds[3] > prm->bounds[2]))
        return ORBHIP_ERR_ARG;
    const size_t lds = (size_t)MP_LDS_PER_KP * cap + 16;
    const bool spill = lds > 160 * 1024 - 20 * 1024;                // the state then lives in global memory
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    if (!spill && lds > 40 * 1024)
        HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_proj<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProjArgs a;
    a.fx = prm->fx; a.fy = prm->fy; a.cx = prm->cx; a.cy = prm->cy; a.mbf = prm->mbf;
    a.min_x = prm->bounds[0]; a.max_x = prm->bounds[1]; a.min_y = prm->bounds[2]; a.max_y = prm->bounds[3];
    a.inv_w = static_cast<float>(GRID_COLS) / (prm->bounds[1] - prm->bounds[0]);   // src/KeyFrame.cc: the frame's grid is copied
    a.inv_h = static_cast<float>(GRID_ROWS) / (prm->bounds[3] - prm->bounds[2]);
    a.th = prm->th;
    a.check_orientation = 0;
    ScaleTab sf;
    for (int l = 0; l < ORB_MAX_LEVELS; l++) sf.v[l] = e->sf[std::min(l, e->prm.nlevels - 1)];
    const KeyframeArgs kf{reinterpret_cast<const float2*>(d_dist_range), d_occupied_kf, d_normal, MT_TH_LOW,
                          std::log((float)e->scale_factor_d), e->prm.nlevels};
    if (spill) {
        const size_t stride = (lds + 255) & ~(size_t)255;
        int rc = match_spill(e, stride * (size_t)nframes);
        if (rc) return rc;
        hipLaunchKernelGGL((k_search_proj<2, true>), dim3(nframes), dim3(MP_THREADS), 0, s, (const GridKp*)nullptr, d_world, d_flags, d_desc,
                           d_cnt_pts, reinterpret_cast<const GridKp*>(d_kp_kf), d_desc_kf, (const float*)nullptr, d_cnt_kf, cap, d_Tcw,
                           (const int*)nullptr, a, sf, kf, d_match_kf, d_nmatches, e->d_match_spill, stride);
    } else {
        hipLaunchKernelGGL(k_search_proj<2>, dim3(nframes), dim3(MP_THREADS), lds, s, (const GridKp*)nullptr, d_world, d_flags, d_desc,
                           d_cnt_pts, reinterpret_cast<const GridKp*>(d_kp_kf), d_desc_kf, (const float*)nullptr, d_cnt_kf, cap, d_Tcw,
                           (const int*)nullptr, a, sf, kf, d_match_kf, d_nmatches);
    }
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_by_projection_sim3(orbhip_extractor* e, const float* world, const float* normal, const float* dist_range,
                                     const uint8_t* flags, const uint8_t* desc, int n1, const orbhip_keypoint* kp_kf,
                                     const uint8_t* desc_kf, const uint8_t* occupied_kf, int n2, const float* Scw,
                                     const orbhip_projection_params* prm, int* match_kf, int* nmatches) {
    if (!e || n1 < 0 || n2 < 0 || !Scw || !prm || !nmatches) return ORBHIP_ERR_ARG;
    if (n2 > 0 && !match_kf) return ORBHIP_ERR_ARG;
    for (int i = 0; i < n2; i++) match_kf[i] = -1;
    if (n1 == 0 || n2 == 0) { *nmatches = 0; return ORBHIP_OK; }
    if (!world || !normal || !dist_range || !flags || !desc || !kp_kf || !desc_kf) return ORBHIP_ERR_ARG;
    float Tcw[12];
    orbhip_decompose_sim3(Scw, Tcw);
    DEVICE_SCOPE(e);
    const int cap = std::max(n1, n2);
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t kpb = up((size_t)cap * sizeof(orbhip_keypoint)), db = up((size_t)cap * 32), wb = up((size_t)cap * 12);
    const size_t rb = up((size_t)cap * 8), fb = up((size_t)cap), mb = up((size_t)cap * 4);
    const size_t need = 128 + kpb + 2 * db + 2 * wb + rb + 2 * fb + mb;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p);                     // [0] n1, [1] n2, [2] matches
    float* d_T = reinterpret_cast<float*>(p + 64); p += 128;
    orbhip_keypoint* d_k2 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    uint8_t* d_d1 = p; p += db;
    uint8_t* d_d2 = p; p += db;
    float* d_w = reinterpret_cast<float*>(p); p += wb;
    float* d_n = reinterpret_cast<float*>(p); p += wb;
    float* d_r = reinterpret_cast<float*>(p); p += rb;
    uint8_t* d_f = p; p += fb;
    uint8_t* d_o = p; p += fb;
    int* d_m = reinterpret_cast<int*>(p);
    const int cnt[3] = {n1, n2, 0};
    io.up(d_cnt, cnt, sizeof(cnt));
    io.up(d_T, Tcw, 12 * sizeof(float));
    io.up(d_d1, desc, (size_t)n1 * 32);
    io.up(d_w, world, (size_t)n1 * 12);
    io.up(d_n, normal, (size_t)n1 * 12);
    io.up(d_r, dist_range, (size_t)n1 * 8);
    io.up(d_f, flags, (size_t)n1);
    io.up(d_k2, kp_kf, (size_t)n2 * sizeof(orbhip_keypoint));
    io.up(d_d2, desc_kf, (size_t)n2 * 32);
    if (occupied_kf) io.up(d_o, occupied_kf, (size_t)n2);
    HIP_TRY(e, io.send());
    int rc = orbhip_search_by_projection_sim3_batch_device(e, 1, cap, d_w, d_n, d_r, d_f, d_d1, d_cnt, d_k2, d_d2,
                                                           occupied_kf ? d_o : nullptr, d_cnt + 1, d_T, prm, d_m, d_cnt + 2, nullptr);
    if (rc) return rc;
    io.down(match_kf, d_m, (size_t)n2 * sizeof(int));
    io.down(nmatches, d_cnt + 2, sizeof(int));
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

int orbhip_search_by_projection_keyframe(orbhip_extractor* e, const orbhip_keypoint* kp_kf, const float* world_kf,
                                         const float* dist_range_kf, const uint8_t* flags_kf, const uint8_t* desc_kf, int n1,
                                         const orbhip_keypoint* kp_cur, const uint8_t* desc_cur, const uint8_t* occupied_cur, int n2,
                                         const float* Tcw, const orbhip_projection_params* prm, int orb_dist,
                                         int* match_cur, int* nmatches) {
    if (!e || n1 < 0 || n2 < 0 || !Tcw || !prm || !nmatches) return ORBHIP_ERR_ARG;
    if (n2 > 0 && !match_cur) return ORBHIP_ERR_ARG;
    for (int i = 0; i < n2; i++) match_cur[i] = -1;
    if (n1 == 0 || n2 == 0) { *nmatches = 0; return ORBHIP_OK; }
    if (!kp_kf || !world_kf || !dist_range_kf || !flags_kf || !desc_kf || !kp_cur || !desc_cur) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    const int cap = std::max(n1, n2);
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t kpb = up((size_t)cap * sizeof(orbhip_keypoint)), db = up((size_t)cap * 32), wb = up((size_t)cap * 12);
    const size_t rb = up((size_t)cap * 8), fb = up((size_t)cap), mb = up((size_t)cap * 4);
    const size_t need = 128 + 2 * kpb + 2 * db + wb + rb + 2 * fb + mb;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p);                     // [0] n1, [1] n2, [2] matches
    float* d_T = reinterpret_cast<float*>(p + 64); p += 128;
    orbhip_keypoint* d_k1 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    orbhip_keypoint* d_k2 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    uint8_t* d_d1 = p; p += db;
    uint8_t* d_d2 = p; p += db;
    float* d_w = reinterpret_cast<float*>(p); p += wb;
    float* d_r = reinterpret_cast<float*>(p); p += rb;
    uint8_t* d_f = p; p += fb;
    uint8_t* d_o = p; p += fb;
    int* d_m = reinterpret_cast<int*>(p);
    const int cnt[3] = {n1, n2, 0};
    io.up(d_cnt, cnt, sizeof(cnt));
    io.up(d_T, Tcw, 12 * sizeof(float));
    io.up(d_k1, kp_kf, (size_t)n1 * sizeof(orbhip_keypoint));
    io.up(d_d1, desc_kf, (size_t)n1 * 32);
    io.up(d_w, world_kf, (size_t)n1 * 12);
    io.up(d_r, dist_range_kf, (size_t)n1 * 8);
    io.up(d_f, flags_kf, (size_t)n1);
    io.up(d_k2, kp_cur, (size_t)n2 * sizeof(orbhip_keypoint));
    io.up(d_d2, desc_cur, (size_t)n2 * 32);
    if (occupied_cur) io.up(d_o, occupied_cur, (size_t)n2);
    HIP_TRY(e, io.send());
    int rc = orbhip_search_by_projection_keyframe_batch_device(e, 1, cap, d_k1, d_w, d_r, d_f, d_d1, d_cnt, d_k2, d_d2,
                                                               occupied_cur ? d_o : nullptr, d_cnt + 1, d_T, prm, orb_dist, d_m,
                                                               d_cnt + 2, nullptr);
    if (rc) return rc;
    io.down(match_cur, d_m, (size_t)n2 * sizeof(int));
    io.down(nmatches, d_cnt + 2, sizeof(int));
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

static int search_by_bow_launch(orbhip_extractor* e, bool kfkf, int nframes, int cap, const orbhip_keypoint* d_kp1,
                                const uint8_t* d_desc1, const uint32_t* d_node1, const uint8_t* d_flags1, const int* d_cnt1,
                                const orbhip_keypoint* d_kp2, const uint8_t* d_desc2, const uint32_t* d_node2,
                                const uint8_t* d_flags2, const int* d_cnt2, float nnratio, int check_orientation,
                                int* d_match, int* d_nmatches, void* stream) {
    if (!e || nframes < 1 || cap < 1 || !d_kp1 || !d_desc1 || !d_node1 || !d_flags1 || !d_cnt1 || !d_kp2 || !d_desc2 ||
        !d_node2 || !d_cnt2 || !d_match || !d_nmatches || (kfkf && !d_flags2))
        return ORBHIP_ERR_ARG;
    if (cap > 65535) { e->err = "the bag-of-words matcher indexes features with 16 bits (cap <= 65535)"; return ORBHIP_ERR_CAPACITY; }
    int npad = 64;
    while (npad < cap) npad <<= 1;
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    const GridKp* k1 = reinterpret_cast<const GridKp*>(d_kp1);
    const GridKp* k2 = reinterpret_cast<const GridKp*>(d_kp2);
    if (npad > 4096) {
        // the sort arrays of such frames live in global memory (the same kernel, 26 bytes per padded slot and pair)
        const size_t stride = ((size_t)(MB_LDS_PER_SLOT + 4) * npad + 255) / 256 * 256;
        int rc = match_spill(e, stride * (size_t)nframes);
        if (rc) return rc;
        if (kfkf)
            hipLaunchKernelGGL((k_search_bow<true, true>), dim3(nframes), dim3(MP_THREADS), 0, s, k1, d_desc1, d_node1, d_flags1, d_cnt1, k2, d_desc2,
                               d_node2, d_flags2, d_cnt2, cap, npad, nnratio, check_orientation, d_match, d_nmatches, e->d_match_spill, stride);
        else
            hipLaunchKernelGGL((k_search_bow<false, true>), dim3(nframes), dim3(MP_THREADS), 0, s, k1, d_desc1, d_node1, d_flags1, d_cnt1, k2, d_desc2,
                               d_node2, d_flags2, d_cnt2, cap, npad, nnratio, check_orientation, d_match, d_nmatches, e->d_match_spill, stride);
        HIP_TRY(e, hipGetLastError());
        return ORBHIP_OK;
    }
    const size_t lds = (size_t)MB_LDS_PER_SLOT * npad;
    const void* fn = kfkf ? reinterpret_cast<const void*>(k_search_bow<true, false>) : reinterpret_cast<const void*>(k_search_bow<false, false>);
    if (lds > 40 * 1024) HIP_TRY(e, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (kfkf)
        hipLaunchKernelGGL((k_search_bow<true, false>), dim3(nframes), dim3(MP_THREADS), lds, s, k1, d_desc1, d_node1, d_flags1, d_cnt1, k2, d_desc2,
                           d_node2, d_flags2, d_cnt2, cap, npad, nnratio, check_orientation, d_match, d_nmatches, static_cast<uint8_t*>(nullptr), (size_t)0);
    else
        hipLaunchKernelGGL((k_search_bow<false, false>), dim3(nframes), dim3(MP_THREADS), lds, s, k1, d_desc1, d_node1, d_flags1, d_cnt1, k2, d_desc2,
                           d_node2, d_flags2, d_cnt2, cap, npad, nnratio, check_orientation, d_match, d_nmatches, static_cast<uint8_t*>(nullptr), (size_t)0);
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_by_bow_batch_device(orbhip_extractor* e, int nframes, int cap, const orbhip_keypoint* d_kp_kf,
                                      const uint8_t* d_desc_kf, const uint32_t* d_node_kf, const uint8_t* d_flags_kf,
                                      const int* d_cnt_kf, const orbhip_keypoint* d_kp_f, const uint8_t* d_desc_f,
                                      const uint32_t* d_node_f, const int* d_cnt_f, float nnratio, int check_orientation,
                                      int* d_match_f, int* d_nmatches, void* stream) {
    return search_by_bow_launch(e, false, nframes, cap, d_kp_kf, d_desc_kf, d_node_kf, d_flags_kf, d_cnt_kf, d_kp_f, d_desc_f, d_node_f,
                                nullptr, d_cnt_f, nnratio, check_orientation, d_match_f, d_nmatches, stream);
}

int orbhip_search_by_bow_keyframes_batch_device(orbhip_extractor* e, int nframes, int cap, const orbhip_keypoint* d_kp1,
                                                const uint8_t* d_desc1, const uint32_t* d_node1, const uint8_t* d_flags1,
                                                const int* d_cnt1, const orbhip_keypoint* d_kp2, const uint8_t* d_desc2,
                                                const uint32_t* d_node2, const uint8_t* d_flags2, const int* d_cnt2, float nnratio,
                                                int check_orientation, int* d_match12, int* d_nmatches, void* stream) {
    return search_by_bow_launch(e, true, nframes, cap, d_kp1, d_desc1, d_node1, d_flags1, d_cnt1, d_kp2, d_desc2, d_node2, d_flags2,
                                d_cnt2, nnratio, check_orientation, d_match12, d_nmatches, stream);
}

static int search_by_bow_host(orbhip_extractor* e, bool kfkf, const orbhip_keypoint* kp1, const uint8_t* desc1, const uint32_t* node1,
                              const uint8_t* flags1, int n1, const orbhip_keypoint* kp2, const uint8_t* desc2,
                              const uint32_t* node2, const uint8_t* flags2, int n2, float nnratio, int check_orientation,
                              int* match, int* nmatches) {
    if (!e || n1 < 0 || n2 < 0 || !nmatches) return ORBHIP_ERR_ARG;
    const int nout = kfkf ? n1 : n2;                           // the result is indexed by side 1 for two key frames, by the frame otherwise
    if (nout > 0 && !match) return ORBHIP_ERR_ARG;
    for (int i = 0; i < nout; i++) match[i] = -1;
    if (n1 == 0 || n2 == 0) { *nmatches = 0; return ORBHIP_OK; }
    if (!kp1 || !desc1 || !node1 || !flags1 || !kp2 || !desc2 || !node2 || (kfkf && !flags2)) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    const int cap = std::max(n1, n2);
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t kpb = up((size_t)cap * sizeof(orbhip_keypoint)), db = up((size_t)cap * 32), nb = up((size_t)cap * 4), fb = up((size_t)cap);
    const size_t need = 64 + 2 * kpb + 2 * db + 3 * nb + 2 * fb;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p); p += 64;           // [0] n1, [1] n2, [2] matches
    orbhip_keypoint* d_k1 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    orbhip_keypoint* d_k2 = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    uint8_t* d_d1 = p; p += db;
    uint8_t* d_d2 = p; p += db;
    uint32_t* d_n1 = reinterpret_cast<uint32_t*>(p); p += nb;
    uint32_t* d_n2 = reinterpret_cast<uint32_t*>(p); p += nb;
    int* d_m = reinterpret_cast<int*>(p); p += nb;
    uint8_t* d_f1 = p; p += fb;
    uint8_t* d_f2 = p;
    const int cnt[3] = {n1, n2, 0};
    io.up(d_cnt, cnt, sizeof(cnt));
    io.up(d_k1, kp1, (size_t)n1 * sizeof(orbhip_keypoint));
    io.up(d_d1, desc1, (size_t)n1 * 32);
    io.up(d_n1, node1, (size_t)n1 * 4);
    io.up(d_f1, flags1, (size_t)n1);
    io.up(d_k2, kp2, (size_t)n2 * sizeof(orbhip_keypoint));
    io.up(d_d2, desc2, (size_t)n2 * 32);
    io.up(d_n2, node2, (size_t)n2 * 4);
    if (kfkf) io.up(d_f2, flags2, (size_t)n2);
    HIP_TRY(e, io.send());
    int rc = search_by_bow_launch(e, kfkf, 1, cap, d_k1, d_d1, d_n1, d_f1, d_cnt, d_k2, d_d2, d_n2, kfkf ? d_f2 : nullptr, d_cnt + 1,
                                  nnratio, check_orientation, d_m, d_cnt + 2, nullptr);
    if (rc) return rc;
    io.down(match, d_m, (size_t)nout * sizeof(int));
    io.down(nmatches, d_cnt + 2, sizeof(int));
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

int orbhip_search_by_bow(orbhip_extractor* e, const orbhip_keypoint* kp_kf, const uint8_t* desc_kf, const uint32_t* node_kf,
                         const uint8_t* flags_kf, int n1, const orbhip_keypoint* kp_f, const uint8_t* desc_f,
                         const uint32_t* node_f, int n2, float nnratio, int check_orientation, int* match_f, int* nmatches) {
    return search_by_bow_host(e, false, kp_kf, desc_kf, node_kf, flags_kf, n1, kp_f, desc_f, node_f, nullptr, n2, nnratio,
                              check_orientation, match_f, nmatches);
}

int orbhip_search_by_bow_keyframes(orbhip_extractor* e, const orbhip_keypoint* kp1, const uint8_t* desc1, const uint32_t* node1,
                                   const uint8_t* flags1, int n1, const orbhip_keypoint* kp2, const uint8_t* desc2,
                                   const uint32_t* node2, const uint8_t* flags2, int n2, float nnratio, int check_orientation,
                                   int* match12, int* nmatches) {
    return search_by_bow_host(e, true, kp1, desc1, node1, flags1, n1, kp2, desc2, node2, flags2, n2, nnratio, check_orientation,
                              match12, nmatches);
}

int orbhip_search_map_points_batch_device(orbhip_extractor* e, int nframes, int cap_mp, const orbhip_track_point* d_mp,
                                          const uint8_t* d_desc_mp, const int* d_cnt_mp, int cap_kp,
                                          const orbhip_keypoint* d_kp, const uint8_t* d_desc, const float* d_uright,
                                          const uint8_t* d_occupied, const int* d_cnt_kp, const float* bounds, float th,
                                          float nnratio, int* d_match, int* d_nmatches, void* stream) {
    if (!e || nframes < 1 || cap_mp < 1 || cap_kp < 1 || cap_kp > 65534 || !d_mp || !d_desc_mp || !d_cnt_mp || !d_kp || !d_desc ||
        !d_cnt_kp || !bounds || !d_match || !d_nmatches || !(bounds[1] > bounds[0]) || !(bounds[3] > bounds[2]) || !(nnratio > 0.0f))
        return ORBHIP_ERR_ARG;
    static_assert(sizeof(orbhip_track_point) == sizeof(TrackPt), "track point layout");
    const size_t lds = (size_t)ML_LDS_PER_MP * cap_mp + (size_t)ML_LDS_PER_KP * cap_kp + 32;
    const bool spill = lds > 160 * 1024 - 20 * 1024;                // the state then lives in global memory
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    if (!spill && lds > 40 * 1024)
        HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_search_map<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ScaleTab sf;
    for (int l = 0; l < ORB_MAX_LEVELS; l++) sf.v[l] = e->sf[std::min(l, e->prm.nlevels - 1)];
    const float inv_w = static_cast<float>(GRID_COLS) / (bounds[1] - bounds[0]);   // src/Frame.cc:128-129
    const float inv_h = static_cast<float>(GRID_ROWS) / (bounds[3] - bounds[2]);
    // the map point lists come first in the carve-up and are written 16 bytes at a time: 16 * cap_mp keeps every later array aligned
    if (spill) {
        const size_t stride = (lds + 255) & ~(size_t)255;
        int rc = match_spill(e, stride * (size_t)nframes);
        if (rc) return rc;
        hipLaunchKernelGGL(k_search_map<true>, dim3(nframes), dim3(MP_THREADS), 0, s, reinterpret_cast<const TrackPt*>(d_mp), d_desc_mp, d_cnt_mp,
                           cap_mp, reinterpret_cast<const GridKp*>(d_kp), d_desc, d_uright, d_occupied, d_cnt_kp, cap_kp, bounds[0], bounds[2],
                           inv_w, inv_h, th, nnratio, sf, e->prm.nlevels, d_match, d_nmatches, e->d_match_spill, stride);
    } else {
        hipLaunchKernelGGL(k_search_map<false>, dim3(nframes), dim3(MP_THREADS), lds, s, reinterpret_cast<const TrackPt*>(d_mp), d_desc_mp, d_cnt_mp,
                           cap_mp, reinterpret_cast<const GridKp*>(d_kp), d_desc, d_uright, d_occupied, d_cnt_kp, cap_kp, bounds[0], bounds[2],
                           inv_w, inv_h, th, nnratio, sf, e->prm.nlevels, d_match, d_nmatches);
    }
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_map_points(orbhip_extractor* e, const orbhip_track_point* mp, const uint8_t* desc_mp, int n1,
                             const orbhip_keypoint* kp, const uint8_t* desc, const float* uright, const uint8_t* occupied, int n2,
                             const float* bounds, float th, float nnratio, int* match, int* nmatches) {
    if (!e || n1 < 0 || n2 < 0 || !bounds || !nmatches) return ORBHIP_ERR_ARG;
    if (n2 > 0 && !match) return ORBHIP_ERR_ARG;
    for (int i = 0; i < n2; i++) match[i] = -1;
    if (n1 == 0 || n2 == 0) { *nmatches = 0; return ORBHIP_OK; }
    if (!mp || !desc_mp || !kp || !desc) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t tb = up((size_t)n1 * sizeof(orbhip_track_point)), d1b = up((size_t)n1 * 32);
    const size_t kpb = up((size_t)n2 * sizeof(orbhip_keypoint)), d2b = up((size_t)n2 * 32), ub = up((size_t)n2 * 4), ob = up((size_t)n2);
    const size_t need = 64 + tb + d1b + kpb + d2b + 2 * ub + ob;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p); p += 64;           // [0] n1, [1] n2, [2] matches
    orbhip_track_point* d_t = reinterpret_cast<orbhip_track_point*>(p); p += tb;
    uint8_t* d_d1 = p; p += d1b;
    orbhip_keypoint* d_k = reinterpret_cast<orbhip_keypoint*>(p); p += kpb;
    uint8_t* d_d2 = p; p += d2b;
    float* d_u = reinterpret_cast<float*>(p); p += ub;
    int* d_m = reinterpret_cast<int*>(p); p += ub;
    uint8_t* d_o = p;
    const int cnt[3] = {n1, n2, 0};
    io.up(d_cnt, cnt, sizeof(cnt));
    io.up(d_t, mp, (size_t)n1 * sizeof(orbhip_track_point));
    io.up(d_d1, desc_mp, (size_t)n1 * 32);
    io.up(d_k, kp, (size_t)n2 * sizeof(orbhip_keypoint));
    io.up(d_d2, desc, (size_t)n2 * 32);
    if (uright) io.up(d_u, uright, (size_t)n2 * 4);
    if (occupied) io.up(d_o, occupied, (size_t)n2);
    HIP_TRY(e, io.send());
    int rc = orbhip_search_map_points_batch_device(e, 1, n1, d_t, d_d1, d_cnt, n2, d_k, d_d2, uright ? d_u : nullptr,
                                                   occupied ? d_o : nullptr, d_cnt + 1, bounds, th, nnratio, d_m, d_cnt + 2, nullptr);
    if (rc) return rc;
    io.down(match, d_m, (size_t)n2 * sizeof(int));
    io.down(nmatches, d_cnt + 2, sizeof(int));
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

int orbhip_debug_logf(orbhip_extractor* e, uint32_t first_bits, uint32_t count, uint32_t stride, float* d_out, void* stream) {
    if (!e || !d_out || count == 0 || count > (1u << 30) || stride == 0) return ORBHIP_ERR_ARG;
    const unsigned long long last = (unsigned long long)first_bits + (unsigned long long)(count - 1) * stride;
    if (first_bits < 0x00800000u || last >= 0x7f800000ull) return ORBHIP_ERR_ARG;   // positive normal floats only
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    hipLaunchKernelGGL(k_debug_logf, dim3((count + 255u) / 256u), dim3(256), 0, s, first_bits, count, stride, d_out);
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_is_in_frustum_batch_device(orbhip_extractor* e, int nframes, int cap_mp, const orbhip_map_point* d_mp, const int* d_cnt_mp,
                                      const float* d_pose, const orbhip_projection_params* prm, float viewing_cos_limit,
                                      orbhip_track_point* d_track, int* d_nvisible, void* stream) {
    if (!e || nframes < 1 || cap_mp < 1 || !d_mp || !d_cnt_mp || !d_pose || !prm || !d_track) return ORBHIP_ERR_ARG;
    static_assert(sizeof(orbhip_map_point) == sizeof(MapPt), "map point layout");
    DEVICE_SCOPE(e);
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    FrustumArgs a;
    a.fx = prm->fx; a.fy = prm->fy; a.cx = prm->cx; a.cy = prm->cy; a.mbf = prm->mbf;
    a.min_x = prm->bounds[0]; a.max_x = prm->bounds[1]; a.min_y = prm->bounds[2]; a.max_y = prm->bounds[3];
    a.viewing_cos_limit = viewing_cos_limit;
    a.log_scale_factor = std::log((float)e->scale_factor_d);   // mfLogScaleFactor (src/Frame.cc: log(mfScaleFactor), the float overload)
    a.nlevels = e->prm.nlevels;
    if (d_nvisible) HIP_TRY(e, hipMemsetAsync(d_nvisible, 0, (size_t)nframes * sizeof(int), s));
    hipLaunchKernelGGL(k_in_frustum, dim3((cap_mp + 255) / 256, nframes), dim3(256), 0, s, reinterpret_cast<const MapPt*>(d_mp), d_cnt_mp,
                       cap_mp, d_pose, a, reinterpret_cast<TrackPt*>(d_track), d_nvisible);
    HIP_TRY(e, hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_is_in_frustum(orbhip_extractor* e, const orbhip_map_point* mp, int n, const float* pose,
                         const orbhip_projection_params* prm, float viewing_cos_limit, orbhip_track_point* track, int* nvisible) {
    if (!e || n < 0 || !pose || !prm) return ORBHIP_ERR_ARG;
    if (nvisible) *nvisible = 0;
    if (n == 0) return ORBHIP_OK;
    if (!mp || !track) return ORBHIP_ERR_ARG;
    DEVICE_SCOPE(e);
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t mb = up((size_t)n * sizeof(orbhip_map_point)), tb = up((size_t)n * sizeof(orbhip_track_point));
    const size_t need = 128 + mb + tb;
    if (int rc0 = staged_ensure(e, need)) return rc0;
    uint8_t* p = e->d_st_tmp;
    Staged io(e);
    int* d_cnt = reinterpret_cast<int*>(p);                     // [0] n, [1] visible
    float* d_pose = reinterpret_cast<float*>(p + 64); p += 128;
    orbhip_map_point* d_m = reinterpret_cast<orbhip_map_point*>(p); p += mb;
    orbhip_track_point* d_t = reinterpret_cast<orbhip_track_point*>(p);
    io.up(d_cnt, &n, sizeof(int));
    io.up(d_pose, pose, 15 * sizeof(float));
    io.up(d_m, mp, (size_t)n * sizeof(orbhip_map_point));
    HIP_TRY(e, io.send());
    int rc = orbhip_is_in_frustum_batch_device(e, 1, n, d_m, d_cnt, d_pose, prm, viewing_cos_limit, d_t, d_cnt + 1, nullptr);
    if (rc) return rc;
    io.down(track, d_t, (size_t)n * sizeof(orbhip_track_point));
    if (nvisible) io.down(nvisible, d_cnt + 1, sizeof(int));
    HIP_TRY(e, io.fetch());
    return ORBHIP_OK;
}

const char* orbhip_strerror(int code) {
    switch (code) {
        case ORBHIP_OK: return "ok";
        case ORBHIP_ERR_ARG: return "invalid argument";
        case ORBHIP_ERR_EMPTY: return "empty image";
        case ORBHIP_ERR_CAPACITY: return "capacity exceeded";
        case ORBHIP_ERR_GEOMETRY: return "unsupported image geometry";
        case ORBHIP_ERR_HIP: return "HIP runtime error";
        case ORBHIP_ERR_NO_DEVICE: return "no usable HIP device";
        case ORBHIP_ERR_STATE: return "no extraction has run yet";
        default: return "unknown error";
    }
}

const char* orbhip_last_error(const orbhip_extractor* e) { return e ? e->err.c_str() : ""; }

} // extern "C"

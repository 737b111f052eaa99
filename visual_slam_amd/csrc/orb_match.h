// ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:402-517) on device-resident
// keypoints and descriptors: the windowed Hamming search between the level-0 keypoints of the
// initial frame F1 and the keypoints of the current frame F2 that monocular initialisation runs
// (src/Tracking.cc:588-589), with Frame::GetFeaturesInArea (src/Frame.cc:351-401),
// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1611-1627) and ComputeThreeMaxima (:1566-1607).
//
// The reference loop over i1 is sequential by construction: a candidate i2 is skipped when an
// earlier i1 already holds it at a distance not larger, and a later, closer i1 takes it over. One
// wave therefore owns one frame pair and walks i1 in order; inside a step the 64 lanes share the
// scan over F2's keypoints. GetFeaturesInArea visits grid cells in (x, y) order and a cell's
// keypoints in index order; every keypoint inside the window lies in a visited cell, so the
// candidate set is "in the grid, octave 0, |dx| < r and |dy| < r" and the reference's first-wins
// rule for equal distances is the minimum over (distance, cell, index).
#pragma once
#include <hip/hip_runtime.h>
#include "orb_device.h"
#include "orb_grid.h"
#include "orb_libm.h"

#define MT_TH_LOW 50          // ORBmatcher::TH_LOW (src/ORBmatcher.cc:39)
#define MT_HISTO 30           // ORBmatcher::HISTO_LENGTH (src/ORBmatcher.cc:40)

// Minimum of a 64-bit key over the 64 lanes, the same value in every lane. DPP row shifts and row
// broadcasts (as in wave_scan_incl; lanes without a source keep the identity ~0), no LDS traffic.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#define WMIN_STEP(CTRL, ROWS)                                                                                   \
    {                                                                                                           \
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)lo, CTRL, ROWS, 0xf, false);        \
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)hi, CTRL, ROWS, 0xf, false);        \
        const bool less = ohi < hi || (ohi == hi && olo < lo);                                                  \
        lo = less ? olo : lo;                                                                                   \
        hi = less ? ohi : hi;                                                                                   \
    }
    WMIN_STEP(0x111, 0xf)   // row_shr:1
    WMIN_STEP(0x112, 0xf)   // row_shr:2
    WMIN_STEP(0x114, 0xf)   // row_shr:4
    WMIN_STEP(0x118, 0xf)   // row_shr:8: lane 15 of every row holds the row's minimum
    WMIN_STEP(0x142, 0xa)   // row_bcast:15 into rows 1, 3
    WMIN_STEP(0x143, 0xc)   // row_bcast:31 into rows 2, 3: lane 63 holds the minimum
#undef WMIN_STEP
    const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63), rhi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return ((unsigned long long)rhi << 32) | rlo;
}

// Grid (frames), one wave per workgroup. Dynamic LDS per F2 keypoint: position (8 bytes), grid cell
// or -1 when it can never be a candidate (2), the distance and the F1 index that currently hold it
// (2 + 2), a slot of the compact list of eligible F2 keypoints (2); per F1 keypoint the rotation bin
// it was matched in (1): MT_LDS_PER_KP bytes in all.
#define MT_LDS_PER_KP 17
__global__ __launch_bounds__(64) void k_search_init(const GridKp* __restrict__ kp1, const uint8_t* __restrict__ desc1,
                                                    const int* __restrict__ cnt1, const GridKp* __restrict__ kp2,
                                                    const uint8_t* __restrict__ desc2, const int* __restrict__ cnt2, int cap,
                                                    float min_x, float min_y, float inv_w, float inv_h, float window,
                                                    float nnratio, int check_orientation, float2* __restrict__ prev_matched,
                                                    int* __restrict__ matches12, int* __restrict__ nmatches_out) {
    extern __shared__ __align__(16) uint8_t mt_smem[];
    float2* pos2 = reinterpret_cast<float2*>(mt_smem);        // undistorted position of every F2 keypoint
    int16_t* cell2 = reinterpret_cast<int16_t*>(mt_smem + 8 * (size_t)cap);   // its grid cell; -1: outside the grid or octave > 0
    uint16_t* matched_dist = reinterpret_cast<uint16_t*>(cell2 + cap);   // vMatchedDistance (0xFFFF = INT_MAX)
    uint16_t* matches21 = matched_dist + cap;                 // vnMatches21 (0xFFFF = -1)
    uint16_t* elig = matches21 + cap;                         // indices of the F2 keypoints that can be candidates at all
    uint8_t* bin_of = reinterpret_cast<uint8_t*>(elig + cap);   // rotation bin of i1 when it was matched, 0xFF = none
    __shared__ int s_hist[MT_HISTO];
    const int lane = threadIdx.x, f = blockIdx.x;
    const size_t o = (size_t)f * cap;
    const int n1 = min(cnt1[f], cap), n2 = min(cnt2[f], cap);
    const GridKp* K1 = kp1 + o;
    const GridKp* K2 = kp2 + o;
    int* m12 = matches12 + o;
    float2* prev = prev_matched + o;
    int nelig = 0;
    for (int i0 = 0; i0 < n2; i0 += 64) {                      // wave-uniform trip count
        const int i = i0 + lane;
        bool ok = false;
        if (i < n2) {
            const GridKp k2 = K2[i];
            const int cell = grid_cell(k2.x, k2.y, min_x, min_y, inv_w, inv_h);
            ok = cell >= 0 && k2.octave == 0;                  // in the grid; minLevel = maxLevel = 0
            pos2[i] = make_float2(k2.x, k2.y);
            cell2[i] = (int16_t)(ok ? cell : -1);
            matched_dist[i] = 0xFFFF;
            matches21[i] = 0xFFFF;
        }
        const unsigned long long m = __ballot(ok);
        if (ok) elig[nelig + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = (uint16_t)i;
        nelig += __popcll(m);
    }
    for (int i = lane; i < n1; i += 64) { bin_of[i] = 0xFF; m12[i] = -1; }
    if (lane < MT_HISTO) s_hist[lane] = 0;
    const float factor = 1.0f / MT_HISTO;
    const float r = window;
    // LDS operations of the one wave complete in issue order: no barriers are needed below
    for (int i1 = 0; i1 < n1; i1++) {
        const GridKp k1 = K1[i1];                             // wave-uniform
        if (k1.octave > 0) continue;                          // src/ORBmatcher.cc:418-420
        const float2 p = prev[i1];
        // Frame::GetFeaturesInArea(x, y, windowSize, 0, 0): the early returns for an empty cell range
        const int min_cx = max(0, (int)floorf((p.x - min_x - r) * inv_w));
        const int max_cx = min(GRID_COLS - 1, (int)ceilf((p.x - min_x + r) * inv_w));
        const int min_cy = max(0, (int)floorf((p.y - min_y - r) * inv_h));
        const int max_cy = min(GRID_ROWS - 1, (int)ceilf((p.y - min_y + r) * inv_h));
        if (min_cx >= GRID_COLS || max_cx < 0 || min_cy >= GRID_ROWS || max_cy < 0) continue;
        const uint4* d1p = reinterpret_cast<const uint4*>(desc1 + (o + i1) * 32);
        const uint4 da = d1p[0], db = d1p[1];
        // every lane: best and second-best (with multiplicity) over its share of the candidates
        unsigned long long best = ~0ull;                      // distance << 40 | cell << 20 | index
        unsigned second = 0x7FFFFFFFu;
        for (int j = lane; j < nelig; j += 64) {
            const int i2 = elig[j];
            const int cell = cell2[i2];
            const float2 q = pos2[i2];
            if (!(fabsf(q.x - p.x) < r && fabsf(q.y - p.y) < r)) continue;
            const uint4* d2p = reinterpret_cast<const uint4*>(desc2 + (o + i2) * 32);
            const uint4 a = d2p[0], b = d2p[1];
            const unsigned dist = __popc(a.x ^ da.x) + __popc(a.y ^ da.y) + __popc(a.z ^ da.z) + __popc(a.w ^ da.w) +
                                  __popc(b.x ^ db.x) + __popc(b.y ^ db.y) + __popc(b.z ^ db.z) + __popc(b.w ^ db.w);
            if ((unsigned)matched_dist[i2] <= dist) continue; // src/ORBmatcher.cc:441-442
            const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)cell << 20) | (unsigned)i2;
            if (key < best) { second = min(second, (unsigned)(best >> 40) >= 0xFFFFFFu ? 0x7FFFFFFFu : (unsigned)(best >> 40)); best = key; }
            else second = min(second, dist);
        }
        // wave: the best entry, then the smallest distance among all other entries
        const unsigned long long wbest = wave_min_u64(best);
        if (wbest == ~0ull) continue;                          // vIndices2 empty, or every candidate skipped: bestDist stays INT_MAX
        unsigned other = best == wbest ? second : min(second, (unsigned)(best >> 40) >= 0xFFFFFFu ? 0x7FFFFFFFu : (unsigned)(best >> 40));
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) other = min(other, (unsigned)__shfl_xor((int)other, s, 64));
        const int bestDist = (int)(wbest >> 40), bestIdx2 = (int)(wbest & 0xFFFFF);
        const int bestDist2 = (int)other;                     // 0x7FFFFFFF = INT_MAX: no second candidate
        if (bestDist <= MT_TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {   // src/ORBmatcher.cc:456-458
            if (lane == 0) {
                const int old = matches21[bestIdx2];
                if (old != 0xFFFF) m12[old] = -1;              // the earlier owner loses the match
                m12[i1] = bestIdx2;
                matches21[bestIdx2] = (uint16_t)i1;
                matched_dist[bestIdx2] = (uint16_t)bestDist;
                if (check_orientation) {
                    float rot = k1.angle - K2[bestIdx2].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == MT_HISTO) bin = 0;
                    bin_of[i1] = (uint8_t)bin;
                    s_hist[bin]++;
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                             // lane 0's global stores to m12 precede the reads below (same wave)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (check_orientation) {
        // ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607) on the bin populations (entries of
        // matches that were taken over later still count, as in the reference)
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < MT_HISTO; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = lane; i < n1; i += 64) {
            const int b = bin_of[i];
            if (b != 0xFF && b != ind1 && b != ind2 && b != ind3) m12[i] = -1;   // src/ORBmatcher.cc:494-507
        }
    }
    // vbPrevMatched follows the matches (src/ORBmatcher.cc:511-514); count them
    int count = 0;
    for (int i = lane; i < n1; i += 64) {
        const int m = m12[i];
        if (m >= 0) { prev[i] = make_float2(K2[m].x, K2[m].y); count++; }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) count += __shfl_xor(count, s, 64);
    if (lane == 0) nmatches_out[f] = count;
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
// (reference src/ORBmatcher.cc:1324-1445; Tracking::TrackWithMotionModel, src/Tracking.cc:1332,1337)
// on plain arrays: the map points seen by the last frame are projected with the current pose and
// matched against the current keypoints inside a window (Frame::GetFeaturesInArea, src/Frame.cc:351-401).
//
// The reference loop over the last frame's keypoints is sequential only through one rule: a
// current keypoint that already holds a point with observations is skipped. The kernel therefore
// splits the work. First every last-frame keypoint finds, in parallel, its best few candidates with
// that rule ignored (one wave per keypoint, lanes over the grid cells of the window; the current
// keypoints are bucketed by grid cell in LDS first). Then one wave walks the keypoints in order:
// the first listed candidate that is still free is the reference's answer (the minimum over a
// superset that lies in the subset); only when the whole list is taken and may be incomplete is the
// search repeated with the rule applied. The traversal
// order of the reference (cells by column then row, indices ascending inside a cell, first of equal
// distances wins) is the minimum over (distance, cell, index).
// One workgroup (MP_THREADS / 64 waves) per frame pair: the searches wait on descriptor loads, so
// many waves per pair hide that latency.
// ------------------------------------------------------------------------------------------------
#define MP_TH_HIGH 100        // ORBmatcher::TH_HIGH (src/ORBmatcher.cc:38)
#define MP_LDS_PER_KP 50      // dynamic LDS bytes per keypoint slot (see the carve-up below)
#define MP_LIST 4             // candidates kept per last-frame keypoint
#define MP_END 0xFFFFFFFEu    // list terminator: there are no further acceptable candidates
#define MP_NONE 0xFFFFFFFFu    // no candidate / list terminator: further candidates are unknown
#define MP_THREADS 1024
#define MP_WAVES (MP_THREADS / 64)

// Exclusive prefix sum of a[0..n) in place by the whole workgroup; returns the total.
__device__ __forceinline__ uint32_t mp_block_excl_scan(uint32_t* a, int n, uint32_t* s_part) {
    const int tid = threadIdx.x;
    const int chunk = (n + MP_THREADS - 1) / MP_THREADS;
    const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += a[i];
    const uint32_t inc = wave_scan_incl(s);
    if ((tid & 63) == 63) s_part[tid >> 6] = inc;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < MP_WAVES; w++) {
        if (w < (tid >> 6)) off += s_part[w];
        total += s_part[w];
    }
    uint32_t run = off + inc - s;
    for (int i = lo; i < hi; i++) { const uint32_t v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

struct ScaleTab { float v[ORB_MAX_LEVELS]; };                 // mvScaleFactors

struct ProjArgs {
    float fx, fy, cx, cy, mbf;
    float min_x, max_x, min_y, max_y, inv_w, inv_h;
    float th;
    int check_orientation;
};

// KEYFRAME: the variant of relocalisation, ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF,
// sAlreadyFound, th, ORBdist) (reference src/ORBmatcher.cc:1447-1564). "Last frame" then reads "key frame";
// what differs: no test of the depth's sign, the point's distance to the camera centre Ow = -Rcw^T tcw
// must lie in its scale-invariance range and predicts the level (MapPoint::PredictScale) in place
// of the keypoint's octave, levels [level - 1, level + 1] always, no mvuRight test, the acceptance
// threshold is ORBdist, and a current keypoint that holds any map point is skipped: the ones that
// hold one on entry (occupied2) and every one assigned here.
// MODE 2, SIM3: ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) of loop closing
// (reference src/ORBmatcher.cc:287-400) on top of that variant: side 1 is a plain list of map points
// (no keypoints), side 2 the key frame; the pose is Scw with its scale divided out (done by the host);
// the depth must not be negative, the projection is formed as fx * (xc * invz) + cx with a float
// reciprocal, KeyFrame::IsInImage excludes the upper bounds, the viewing direction must lie within 60
// degrees of the point's normal, levels [level - 1, level], threshold TH_LOW, no rotation check.
struct KeyframeArgs {
    const float2* dist_range1;    // per key-frame keypoint: mfMinDistance, mfMaxDistance of its map point
    const uint8_t* occupied2;     // per current keypoint: CurrentFrame.mvpMapPoints[i2] != NULL on entry (may be NULL)
    const float* normal1;         // SIM3: GetNormal() of every map point
    int orb_dist;
    float log_scale_factor;
    int nlevels;
};

// SPILL: frames with more keypoints than the workgroup's LDS can describe (beyond ~2800 per frame) keep the same state in
// a slice of global memory (`spill + frame * spill_stride`); the code is the same, slower, and the workgroup's barriers
// order its own global accesses.
template <int MODE, bool SPILL = false>
__global__ __launch_bounds__(MP_THREADS) void k_search_proj(const GridKp* __restrict__ kp1, const float* __restrict__ world1,
                                                     const uint8_t* __restrict__ flags1, const uint8_t* __restrict__ desc1,
                                                     const int* __restrict__ cnt1, const GridKp* __restrict__ kp2,
                                                     const uint8_t* __restrict__ desc2, const float* __restrict__ uright2,
                                                     const int* __restrict__ cnt2, int cap, const float* __restrict__ Tcw,
                                                     const int* __restrict__ level_mode, ProjArgs a, ScaleTab sf, KeyframeArgs kf,
                                                     int* __restrict__ match2, int* __restrict__ nmatches_out,
                                                     uint8_t* __restrict__ spill = nullptr, size_t spill_stride = 0) {
    constexpr bool KEYFRAME = MODE != 0, SIM3 = MODE == 2;
    extern __shared__ __align__(16) uint8_t mp_lds[];
    uint8_t* const mp_smem = SPILL ? spill + (size_t)blockIdx.x * spill_stride : mp_lds;
    uint32_t* list1 = reinterpret_cast<uint32_t*>(mp_smem);                  // last: MP_LIST x (distance << 16 | current index), best first, occupancy ignored
    float2* pos2 = reinterpret_cast<float2*>(list1 + MP_LIST * (size_t)cap);  // current keypoints: position
    float2* uv1 = pos2 + cap;                                                // last keypoints: projection (u, v)
    float* ur2 = reinterpret_cast<float*>(uv1 + cap);                        // current: mvuRight
    float* ur1 = ur2 + cap;                                                  // last: u - mbf / z
    uint16_t* sorted2 = reinterpret_cast<uint16_t*>(ur1 + cap);              // current indices bucketed by grid cell
    uint16_t* cell2 = sorted2 + cap;                                         // current: grid cell, 0xFFFF = outside
    uint16_t* taken1 = cell2 + cap;                                          // last: the current keypoint it was assigned to
    uint8_t* oct2 = reinterpret_cast<uint8_t*>(taken1 + cap);                // current: octave
    uint8_t* occ2 = oct2 + cap;                                              // current: holds a point with observations
    uint8_t* st1 = occ2 + cap;                                               // last: 1 usable | 2 observed | octave << 2
    uint8_t* bin1 = st1 + cap;                                               // last: rotation bin of its match, 0xFF = none
    __shared__ uint32_t s_cur[GRID_CELLS];                                   // histogram, then running cursor
    __shared__ uint16_t s_start[GRID_CELLS + 2];
    __shared__ uint32_t s_magic[GRID_ROWS + 1];
    __shared__ uint32_t s_tmp[MP_WAVES];
    __shared__ int s_hist[MT_HISTO];
    __shared__ int s_made;
    const int tid = threadIdx.x, lane = tid & 63, f = blockIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t o = (size_t)f * cap;
    const int n1 = min(cnt1[f], cap), n2 = min(cnt2[f], cap);
    const GridKp* K1 = kp1 + o;
    const GridKp* K2 = kp2 + o;
    const int mode = level_mode ? level_mode[f] : 0;
    const float* T = Tcw + 12 * (size_t)f;
    int* m2 = match2 + o;

    for (int c = tid; c < GRID_CELLS; c += MP_THREADS) s_cur[c] = 0;
    if (tid >= 1 && tid <= GRID_ROWS) s_magic[tid] = ((1u << 20) + tid - 1) / tid;   // n / d == n * magic >> 20 for n * d < 2^20
    if (tid < MT_HISTO) s_hist[tid] = 0;
    if (tid == 0) s_made = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += MP_THREADS) {
        const GridKp k = K2[i];
        const int cell = grid_cell(k.x, k.y, a.min_x, a.min_y, a.inv_w, a.inv_h);
        pos2[i] = make_float2(k.x, k.y);
        ur2[i] = uright2 ? uright2[o + i] : -1.0f;
        cell2[i] = (uint16_t)(cell < 0 ? 0xFFFF : cell);
        oct2[i] = (uint8_t)k.octave;
        occ2[i] = (uint8_t)(KEYFRAME && kf.occupied2 ? kf.occupied2[o + i] != 0 : 0);
        m2[i] = -1;                                            // nothing assigned by this call yet
        if (cell >= 0) atomicAdd(&s_cur[cell], 1u);
    }
    // projections of the last frame's map points (src/ORBmatcher.cc:1343-1361)
    const float r00 = T[0], r01 = T[1], r02 = T[2], r10 = T[3], r11 = T[4], r12 = T[5], r20 = T[6], r21 = T[7], r22 = T[8];
    const float t0 = T[9], t1 = T[10], t2 = T[11];
    for (int i = tid; i < n1; i += MP_THREADS) {
        uint8_t st = 0;
        const uint8_t fl = flags1[o + i];
        if (fl & 1) {
            const float X = world1[3 * (o + i)], Y = world1[3 * (o + i) + 1], Z = world1[3 * (o + i) + 2];
            // Rcw * x3Dw + tcw as the small-matrix product computes it: float products summed in
            // float, the translation added in double
            const float xc = (float)((double)(r00 * X + r01 * Y + r02 * Z) + (double)t0);
            const float yc = (float)((double)(r10 * X + r11 * Y + r12 * Z) + (double)t1);
            const float zc = (float)((double)(r20 * X + r21 * Y + r22 * Z) + (double)t2);
            const float invzc = SIM3 ? __fdiv_rn(1.0f, zc) : (float)(1.0 / (double)zc);
            const float u = SIM3 ? a.fx * (xc * invzc) + a.cx : a.fx * xc * invzc + a.cx;
            const float v = SIM3 ? a.fy * (yc * invzc) + a.cy : a.fy * yc * invzc + a.cy;
            bool ok = SIM3 ? (!(zc < 0.0f) && u >= a.min_x && u < a.max_x && v >= a.min_y && v < a.max_y)   // src/ORBmatcher.cc:324-337
                           : ((KEYFRAME || !(invzc < 0)) && u == u && v == v && !(u < a.min_x || u > a.max_x) && !(v < a.min_y || v > a.max_y));
            int lvl = 0;
            if (!KEYFRAME) lvl = K1[i].octave & (ORB_MAX_LEVELS - 1);
            if (KEYFRAME) {
                // Ow = -Rcw^T tcw as the matrix expression computes it (accumulated in double, stored as float);
                // PO = x3Dw - Ow in float, cv::norm in double (src/ORBmatcher.cc:1452,1480-1490)
                const float ox = (float)(-((double)r00 * (double)t0 + (double)r10 * (double)t1 + (double)r20 * (double)t2));
                const float oy = (float)(-((double)r01 * (double)t0 + (double)r11 * (double)t1 + (double)r21 * (double)t2));
                const float oz = (float)(-((double)r02 * (double)t0 + (double)r12 * (double)t1 + (double)r22 * (double)t2));
                const float px = X - ox, py = Y - oy, pz = Z - oz;
                const float dist3 = (float)sqrt((double)px * (double)px + (double)py * (double)py + (double)pz * (double)pz);
                const float2 range = kf.dist_range1[o + i];
                ok = ok && !(dist3 < 0.8f * range.x || dist3 > 1.2f * range.y);
                if (SIM3) {                                    // PO.dot(Pn) < 0.5 * dist: both sides double (src/ORBmatcher.cc:349-352)
                    const float* nr = kf.normal1 + 3 * (o + i);
                    const double dot = (double)px * (double)nr[0] + (double)py * (double)nr[1] + (double)pz * (double)nr[2];
                    ok = ok && !(dot < 0.5 * (double)dist3);
                }
                lvl = predict_scale(range.y, dist3, kf.log_scale_factor, kf.nlevels);
            }
            if (ok) {
                uv1[i] = make_float2(u, v);
                ur1[i] = u - a.mbf * invzc;
                st = (uint8_t)(1 | (KEYFRAME ? 2 : (fl & 2)) | (lvl << 2));
            }
        }
        st1[i] = st;
        bin1[i] = 0xFF;
        list1[MP_LIST * i] = MP_END;
    }
    __syncthreads();
    const uint32_t total = mp_block_excl_scan(s_cur, GRID_CELLS, s_tmp);
    for (int c = tid; c < GRID_CELLS; c += MP_THREADS) s_start[c] = (uint16_t)s_cur[c];
    if (tid == 0) s_start[GRID_CELLS] = (uint16_t)total;
    __syncthreads();
    for (int i = tid; i < n2; i += MP_THREADS) {
        const int cell = cell2[i];
        if (cell != 0xFFFF) sorted2[atomicAdd(&s_cur[cell], 1u)] = (uint16_t)i;
    }
    __syncthreads();

    // The best candidates of last-frame keypoint i over the window, by the whole wave: e[0..return)
    // in the reference's order of preference, then a terminator (MP_END: nothing acceptable follows,
    // MP_NONE: unknown) unless all MP_LIST slots are used. use_occ: skip current keypoints that hold
    // a point with observations (src/ORBmatcher.cc:1377-1379).
    auto find_cands = [&](int i, int st, bool use_occ, uint32_t (&e)[MP_LIST]) {
#pragma unroll
        for (int r = 0; r < MP_LIST; r++) e[r] = MP_END;
        const float2 p = uv1[i];
        const float ur = ur1[i];
        const int oct = st >> 2;
        const float r = a.th * sf.v[oct];
        const int lo = mode == 1 ? oct : mode == 2 ? 0 : oct - 1;           // octave range of GetFeaturesInArea's level test
        const int hi = mode == 1 ? 255 : (mode == 2 || SIM3) ? oct : oct + 1;
        const int min_cx = max(0, (int)floorf((p.x - a.min_x - r) * a.inv_w));
        const int max_cx = min(GRID_COLS - 1, (int)ceilf((p.x - a.min_x + r) * a.inv_w));
        const int min_cy = max(0, (int)floorf((p.y - a.min_y - r) * a.inv_h));
        const int max_cy = min(GRID_ROWS - 1, (int)ceilf((p.y - a.min_y + r) * a.inv_h));
        if (min_cx >= GRID_COLS || max_cx < 0 || min_cy >= GRID_ROWS || max_cy < 0 || max_cx < min_cx || max_cy < min_cy) return;
        const int ncy = max_cy - min_cy + 1, ncell = (max_cx - min_cx + 1) * ncy;
        const uint32_t magic = s_magic[ncy];
        const uint4* d1p = reinterpret_cast<const uint4*>(desc1 + (o + i) * 32);
        const uint4 da = d1p[0], db = d1p[1];
        // per lane: its two smallest keys  distance << 40 | cell << 20 | index  and how many it saw
        unsigned long long b1 = ~0ull, b2 = ~0ull;
        int seen = 0;
        for (int c = lane; c < ncell; c += 64) {
            const int qx = (int)(((uint32_t)c * magic) >> 20);
            const int cell = (min_cx + qx) * GRID_ROWS + min_cy + (c - qx * ncy);
            const int ke = s_start[cell + 1];
            for (int k = s_start[cell]; k < ke; k++) {
                const int i2 = sorted2[k];
                const int o2 = oct2[i2];
                if (o2 < lo || o2 > hi) continue;
                const float2 q = pos2[i2];
                if (!(fabsf(q.x - p.x) < r && fabsf(q.y - p.y) < r)) continue;
                if ((use_occ || KEYFRAME) && occ2[i2]) continue;   // (before the ordered walk occ2 holds the occupancy on entry)
                const float u2 = ur2[i2];
                if (!KEYFRAME && u2 > 0 && fabsf(ur - u2) > r) continue;    // src/ORBmatcher.cc:1380-1385
                const uint4* d2p = reinterpret_cast<const uint4*>(desc2 + (o + i2) * 32);
                const uint4 x = d2p[0], y = d2p[1];
                const unsigned dist = __popc(x.x ^ da.x) + __popc(x.y ^ da.y) + __popc(x.z ^ da.z) + __popc(x.w ^ da.w) +
                                      __popc(y.x ^ db.x) + __popc(y.y ^ db.y) + __popc(y.z ^ db.z) + __popc(y.w ^ db.w);
                if ((int)dist > (KEYFRAME ? kf.orb_dist : MP_TH_HIGH)) continue;   // can never be accepted (src/ORBmatcher.cc:1396,1520)
                const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)cell << 20) | (unsigned)i2;
                seen++;
                if (key < b1) { b2 = b1; b1 = key; }
                else if (key < b2) b2 = key;
            }
        }
        // the wave's smallest keys one after the other; a lane that has handed out both keys it kept
        // and saw more ends the list, because its third key might come next
        int used = 0;
#pragma unroll
        for (int r = 0; r < MP_LIST; r++) {
            const unsigned long long m = wave_min_u64(b1);
            if (m == ~0ull) break;                             // e[r] stays MP_END
            e[r] = (uint32_t)((m >> 40) << 16) | (uint32_t)(m & 0xFFFF);
            bool dry = false;
            if (b1 == m) { used++; b1 = b2; b2 = ~0ull; dry = used == 2 && seen > 2; }
            if (__ballot(dry)) { if (r + 1 < MP_LIST) e[r + 1] = MP_NONE; break; }
        }
    };

    for (int i = wv; i < n1; i += MP_WAVES) {
        const int st = __builtin_amdgcn_readfirstlane((int)st1[i]);
        if (!(st & 1)) continue;
        uint32_t e[MP_LIST];
        find_cands(i, st, false, e);
        if (lane == 0) *reinterpret_cast<uint4*>(&list1[MP_LIST * i]) = make_uint4(e[0], e[1], e[2], e[3]);
    }
    __syncthreads();
    if (wv == 0) {
        // the reference's loop order; LDS operations of the one wave complete in issue order
        // States and lists of 64 keypoints at a time are fetched into registers (one per lane) and handed out
        // with v_readlane; the occupancy of a keypoint's listed candidates is read by lanes 0..3 in one round trip.
        int made = 0;
        for (int i0 = 0; i0 < n1; i0 += 64) {
          const int il = min(i0 + lane, n1 - 1);
          const int stv = st1[il];
          const uint4 lv = *reinterpret_cast<const uint4*>(&list1[MP_LIST * il]);
          // keypoints without a usable point or with an empty, complete list decide nothing: only the others are visited
          unsigned long long todo = __ballot(i0 + lane < n1 && (stv & 1) && lv.x != MP_END);
          while (todo) {
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int i = i0 + k;
            const int st = __builtin_amdgcn_readlane(stv, k);                 // the same in every lane: scalar control flow
            const uint32_t le[MP_LIST] = {(uint32_t)__builtin_amdgcn_readlane((int)lv.x, k), (uint32_t)__builtin_amdgcn_readlane((int)lv.y, k),
                                          (uint32_t)__builtin_amdgcn_readlane((int)lv.z, k), (uint32_t)__builtin_amdgcn_readlane((int)lv.w, k)};
            const uint32_t myc = lane == 0 ? le[0] : lane == 1 ? le[1] : lane == 2 ? le[2] : le[3];
            const int myo = (lane < MP_LIST && myc < MP_END) ? (int)occ2[myc & 0xFFFF] : 0;
            const int oc[MP_LIST] = {__builtin_amdgcn_readlane(myo, 0), __builtin_amdgcn_readlane(myo, 1),
                                     __builtin_amdgcn_readlane(myo, 2), __builtin_amdgcn_readlane(myo, 3)};
            uint32_t b = MP_NONE;                              // MP_NONE after the walk: the list did not decide
#pragma unroll
            for (int r = 0; r < MP_LIST; r++) {
                const uint32_t c = le[r];
                if (c >= MP_END) { b = c; break; }
                if (!oc[r]) { b = c; break; }                  // still free: the reference's choice
            }
            if (b == MP_NONE) {                                // every listed candidate was taken meanwhile: search again
                uint32_t e[MP_LIST];
                find_cands(i, st, true, e);
                b = e[0];
            }
            if (b >= MP_END) continue;                         // bestDist > TH_HIGH or no candidate
            const int i2 = (int)(b & 0xFFFF);
            made++;
            if (lane == 0) {
                m2[i2] = i;                                    // a point without observations is simply overwritten
                occ2[i2] = (uint8_t)((st >> 1) & 1);
                taken1[i] = (uint16_t)i2;
                bin1[i] = 0xFE;                                // matched; its rotation bin is filled in below, in parallel
            }
          }
        }
        if (lane == 0) s_made = made;
    }
    __syncthreads();                                           // wave 0's stores to m2 are complete and visible to the workgroup
    int bad = 0;
    if (a.check_orientation) {
        // rotation histogram of every assignment made (src/ORBmatcher.cc:1409-1418)
        const float factor = 1.0f / MT_HISTO;
        for (int i = tid; i < n1; i += MP_THREADS) {
            if (bin1[i] != 0xFE) continue;
            float rot = K1[i].angle - K2[taken1[i]].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == MT_HISTO) bin = 0;
            bin1[i] = (uint8_t)bin;
            atomicAdd(&s_hist[bin], 1);
        }
        __syncthreads();
        // ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607); every assignment counts, overwritten ones too
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < MT_HISTO; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = tid; i < n1; i += MP_THREADS) {
            const int b = bin1[i];
            if (b != 0xFF && b != ind1 && b != ind2 && b != ind3) { m2[taken1[i]] = -1; bad++; }   // src/ORBmatcher.cc:1430-1441
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) bad += __shfl_xor(bad, s, 64);
    if (lane == 0) s_tmp[wv] = (uint32_t)bad;
    __syncthreads();
    if (tid == 0) {
        int bad_all = 0;
        for (int w = 0; w < MP_WAVES; w++) bad_all += (int)s_tmp[w];
        nmatches_out[f] = s_made - bad_all;
    }
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchForInitialization again (reference src/ORBmatcher.cc:402-517), organised like
// k_search_proj: the loop over F1 is sequential only through vMatchedDistance (a candidate that an
// earlier keypoint holds at a distance not larger is skipped). First every level-0 keypoint of F1
// lists, in parallel and with that rule ignored, its best few candidates in the reference's order of
// preference (distance, cell, index); candidates that can decide nothing are left out (distance
// above TH_LOW and large enough that, as a second best, it lets every acceptable best through).
// Then one wave walks F1 in order: the first two listed candidates that the rule does not skip are
// the reference's best and second best; a list that ends too early is replaced by a search with the
// rule applied. One workgroup (MP_THREADS / 64 waves) per frame pair; k_search_init above remains
// for frames with more keypoints than this kernel's LDS state holds.
// ------------------------------------------------------------------------------------------------
#define MI_LDS_PER_KP 36      // dynamic LDS bytes per keypoint slot

__global__ __launch_bounds__(MP_THREADS) void k_search_init_par(const GridKp* __restrict__ kp1, const uint8_t* __restrict__ desc1,
                                                                const int* __restrict__ cnt1, const GridKp* __restrict__ kp2,
                                                                const uint8_t* __restrict__ desc2, const int* __restrict__ cnt2,
                                                                int cap, float min_x, float min_y, float inv_w, float inv_h,
                                                                float window, float nnratio, int check_orientation,
                                                                float2* __restrict__ prev_matched, int* __restrict__ matches12,
                                                                int* __restrict__ nmatches_out) {
    extern __shared__ __align__(16) uint8_t mi_smem[];
    uint32_t* list1 = reinterpret_cast<uint32_t*>(mi_smem);                  // F1: MP_LIST x (distance << 16 | F2 index), best first
    float2* pos2 = reinterpret_cast<float2*>(list1 + MP_LIST * (size_t)cap);  // F2: position
    uint16_t* sorted2 = reinterpret_cast<uint16_t*>(pos2 + cap);             // F2 level-0 indices bucketed by grid cell
    uint16_t* cell2 = sorted2 + cap;                                         // F2: grid cell, 0xFFFF = never a candidate
    uint16_t* matched_dist = cell2 + cap;                                    // vMatchedDistance (0xFFFF = INT_MAX)
    uint16_t* matches21 = matched_dist + cap;                                // vnMatches21 (0xFFFF = -1)
    uint16_t* taken1 = matches21 + cap;                                      // F1: the F2 keypoint it was matched to (at that time)
    uint8_t* bin1 = reinterpret_cast<uint8_t*>(taken1 + cap);                // F1: rotation bin of that match, 0xFE = pending, 0xFF = none
    uint8_t* lvl0 = bin1 + cap;                                              // F1: 1 = octave 0 (the others are not matched, src/ORBmatcher.cc:418-420)
    __shared__ uint32_t s_cur[GRID_CELLS];
    __shared__ uint16_t s_start[GRID_CELLS + 2];
    __shared__ uint32_t s_magic[GRID_ROWS + 1];
    __shared__ uint32_t s_tmp[MP_WAVES];
    __shared__ int s_hist[MT_HISTO];
    const int tid = threadIdx.x, lane = tid & 63, f = blockIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t o = (size_t)f * cap;
    const int n1 = min(cnt1[f], cap), n2 = min(cnt2[f], cap);
    const GridKp* K1 = kp1 + o;
    const GridKp* K2 = kp2 + o;
    int* m12 = matches12 + o;
    float2* prev = prev_matched + o;
    const float r = window;

    for (int c = tid; c < GRID_CELLS; c += MP_THREADS) s_cur[c] = 0;
    if (tid >= 1 && tid <= GRID_ROWS) s_magic[tid] = ((1u << 20) + tid - 1) / tid;
    if (tid < MT_HISTO) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += MP_THREADS) {
        const GridKp k = K2[i];
        int cell = grid_cell(k.x, k.y, min_x, min_y, inv_w, inv_h);
        if (k.octave != 0) cell = -1;                          // minLevel = maxLevel = 0 (src/ORBmatcher.cc:423)
        pos2[i] = make_float2(k.x, k.y);
        cell2[i] = (uint16_t)(cell < 0 ? 0xFFFF : cell);
        matched_dist[i] = 0xFFFF;
        matches21[i] = 0xFFFF;
        if (cell >= 0) atomicAdd(&s_cur[cell], 1u);
    }
    for (int i = tid; i < n1; i += MP_THREADS) { bin1[i] = 0xFF; m12[i] = -1; list1[MP_LIST * i] = MP_END; lvl0[i] = K1[i].octave == 0; }
    __syncthreads();
    const uint32_t total = mp_block_excl_scan(s_cur, GRID_CELLS, s_tmp);
    for (int c = tid; c < GRID_CELLS; c += MP_THREADS) s_start[c] = (uint16_t)s_cur[c];
    if (tid == 0) s_start[GRID_CELLS] = (uint16_t)total;
    __syncthreads();
    for (int i = tid; i < n2; i += MP_THREADS) {
        const int cell = cell2[i];
        if (cell != 0xFFFF) sorted2[atomicAdd(&s_cur[cell], 1u)] = (uint16_t)i;
    }
    __syncthreads();

    // Visits the candidates of F1 keypoint i1 (F2 level-0 keypoints inside the window around its
    // vbPrevMatched entry, Frame::GetFeaturesInArea src/Frame.cc:351-401), lanes over the cells.
    auto for_candidates = [&](int i1, auto&& body) __attribute__((always_inline)) {
        const float2 p = prev[i1];
        const int min_cx = max(0, (int)floorf((p.x - min_x - r) * inv_w));
        const int max_cx = min(GRID_COLS - 1, (int)ceilf((p.x - min_x + r) * inv_w));
        const int min_cy = max(0, (int)floorf((p.y - min_y - r) * inv_h));
        const int max_cy = min(GRID_ROWS - 1, (int)ceilf((p.y - min_y + r) * inv_h));
        if (min_cx >= GRID_COLS || max_cx < 0 || min_cy >= GRID_ROWS || max_cy < 0 || max_cx < min_cx || max_cy < min_cy) return;
        const int ncy = max_cy - min_cy + 1, ncell = (max_cx - min_cx + 1) * ncy;
        const uint32_t magic = s_magic[ncy];
        const uint4* d1p = reinterpret_cast<const uint4*>(desc1 + (o + i1) * 32);
        const uint4 da = d1p[0], db = d1p[1];
        for (int c = lane; c < ncell; c += 64) {
            const int qx = (int)(((uint32_t)c * magic) >> 20);
            const int cell = (min_cx + qx) * GRID_ROWS + min_cy + (c - qx * ncy);
            const int ke = s_start[cell + 1];
            for (int k = s_start[cell]; k < ke; k++) {
                const int i2 = sorted2[k];
                const float2 q = pos2[i2];
                if (!(fabsf(q.x - p.x) < r && fabsf(q.y - p.y) < r)) continue;
                const uint4* d2p = reinterpret_cast<const uint4*>(desc2 + (o + i2) * 32);
                const uint4 x = d2p[0], y = d2p[1];
                const unsigned dist = __popc(x.x ^ da.x) + __popc(x.y ^ da.y) + __popc(x.z ^ da.z) + __popc(x.w ^ da.w) +
                                      __popc(y.x ^ db.x) + __popc(y.y ^ db.y) + __popc(y.z ^ db.z) + __popc(y.w ^ db.w);
                body(i2, cell, dist);
            }
        }
    };

    // ---- phase A: candidate lists, one wave per F1 keypoint
    for (int i1 = wv; i1 < n1; i1 += MP_WAVES) {
        if (!__builtin_amdgcn_readfirstlane((int)lvl0[i1])) continue;
        unsigned long long b1 = ~0ull, b2 = ~0ull;             // per lane: its two smallest keys, and how many it saw
        int seen = 0;
        for_candidates(i1, [&](int i2, int cell, unsigned dist) {
            // a candidate beyond TH_LOW is no acceptable best, and as a second best it matters only
            // if some acceptable best fails the ratio test against it
            if (dist > MT_TH_LOW && (float)MT_TH_LOW < (float)dist * nnratio) return;
            const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)cell << 20) | (unsigned)i2;
            seen++;
            if (key < b1) { b2 = b1; b1 = key; }
            else if (key < b2) b2 = key;
        });
        uint32_t e[MP_LIST];
#pragma unroll
        for (int q = 0; q < MP_LIST; q++) e[q] = MP_END;
        int used = 0;
#pragma unroll
        for (int q = 0; q < MP_LIST; q++) {
            const unsigned long long m = wave_min_u64(b1);
            if (m == ~0ull) break;
            e[q] = (uint32_t)((m >> 40) << 16) | (uint32_t)(m & 0xFFFF);
            bool dry = false;
            if (b1 == m) { used++; b1 = b2; b2 = ~0ull; dry = used == 2 && seen > 2; }
            if (__ballot(dry)) { if (q + 1 < MP_LIST) e[q + 1] = MP_NONE; break; }
        }
        if (lane == 0) *reinterpret_cast<uint4*>(&list1[MP_LIST * i1]) = make_uint4(e[0], e[1], e[2], e[3]);
    }
    __syncthreads();

    // ---- phase B: the reference's loop order, one wave, scalar control flow on LDS state
    if (wv == 0) {
        for (int i1 = 0; i1 < n1; i1++) {
            if (!__builtin_amdgcn_readfirstlane((int)lvl0[i1])) continue;
            int bestDist = 0x7FFFFFFF, bestDist2 = 0x7FFFFFFF, bestIdx2 = -1;
            bool decided = false;
            uint32_t term = MP_NONE, last_real = MP_NONE;      // how the list ended (MP_NONE also stands for "all slots used")
            for (int q = 0; q < MP_LIST; q++) {
                const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)list1[MP_LIST * i1 + q]);
                if (c >= MP_END) { term = c; break; }
                last_real = c;
                const int d = (int)(c >> 16), i2 = (int)(c & 0xFFFF);
                if (__builtin_amdgcn_readfirstlane((int)matched_dist[i2]) <= d) continue;   // src/ORBmatcher.cc:441-442
                if (bestIdx2 < 0) { bestDist = d; bestIdx2 = i2; }
                else { bestDist2 = d; decided = true; break; }
            }
            if (!decided) {
                if (term == MP_END) decided = true;            // the list was complete: what was found is all there is
                else if (bestIdx2 >= 0 && last_real != MP_NONE && (float)bestDist < (float)(int)(last_real >> 16) * nnratio)
                    decided = true;                            // every unlisted candidate lies at or beyond the last listed distance
            }
            if (!decided) {
                // search again with the rule applied: exact best (distance, cell, index) and second-best distance
                unsigned long long best = ~0ull;
                unsigned second = 0x7FFFFFFFu;
                for_candidates(i1, [&](int i2, int cell, unsigned dist) {
                    if ((unsigned)matched_dist[i2] <= dist) return;
                    const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)cell << 20) | (unsigned)i2;
                    if (key < best) { second = min(second, best == ~0ull ? 0x7FFFFFFFu : (unsigned)(best >> 40)); best = key; }
                    else second = min(second, dist);
                });
                const unsigned long long wbest = wave_min_u64(best);
                if (wbest == ~0ull) continue;
                unsigned other = best == wbest ? second : min(second, best == ~0ull ? 0x7FFFFFFFu : (unsigned)(best >> 40));
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) other = min(other, (unsigned)__shfl_xor((int)other, s, 64));
                bestDist = (int)(wbest >> 40); bestIdx2 = (int)(wbest & 0xFFFFF); bestDist2 = (int)other;
            }
            if (bestIdx2 < 0) continue;
            if (bestDist <= MT_TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {   // src/ORBmatcher.cc:456-458
                if (lane == 0) {
                    const int old = matches21[bestIdx2];
                    if (old != 0xFFFF) m12[old] = -1;          // the earlier owner loses the match
                    m12[i1] = bestIdx2;
                    matches21[bestIdx2] = (uint16_t)i1;
                    matched_dist[bestIdx2] = (uint16_t)bestDist;
                    taken1[i1] = (uint16_t)bestIdx2;
                    bin1[i1] = 0xFE;
                }
            }
        }
    }
    __syncthreads();                                           // wave 0's stores to m12 are complete and visible to the workgroup
    if (check_orientation) {
        const float factor = 1.0f / MT_HISTO;
        for (int i = tid; i < n1; i += MP_THREADS) {
            if (bin1[i] != 0xFE) continue;
            float rot = K1[i].angle - K2[taken1[i]].angle;     // src/ORBmatcher.cc:466-473
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == MT_HISTO) bin = 0;
            bin1[i] = (uint8_t)bin;
            atomicAdd(&s_hist[bin], 1);
        }
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;   // ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607)
        for (int i = 0; i < MT_HISTO; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = tid; i < n1; i += MP_THREADS) {
            const int b = bin1[i];
            if (b < 0xFE && b != ind1 && b != ind2 && b != ind3) m12[i] = -1;   // src/ORBmatcher.cc:494-507
        }
    }
    // vbPrevMatched follows the matches (src/ORBmatcher.cc:511-514); count them
    int count = 0;
    for (int i = tid; i < n1; i += MP_THREADS) {
        const int m = m12[i];                                  // written by wave 0 or by this same thread
        if (m >= 0) { prev[i] = make_float2(K2[m].x, K2[m].y); count++; }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) count += __shfl_xor(count, s, 64);
    if (lane == 0) s_tmp[wv] = (uint32_t)count;
    __syncthreads();
    if (tid == 0) {
        int all = 0;
        for (int w = 0; w < MP_WAVES; w++) all += (int)s_tmp[w];
        nmatches_out[f] = all;
    }
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th)
// (reference src/ORBmatcher.cc:46-127; Tracking::SearchLocalPoints, src/Tracking.cc) on plain
// arrays: every local map point that Frame::isInFrustum marked visible (its projection, predicted
// pyramid level and viewing cosine are inputs, as the reference reads them from the MapPoint) is
// matched against the frame's keypoints inside a window whose size depends on the viewing angle
// (RadiusByViewingCos, src/ORBmatcher.cc:129-135) and the predicted level; the best candidate is
// accepted when it is close enough and, if the second best lies on the same level, distinctly
// better than it.
//
// The loop over the map points is sequential only through one rule: a keypoint that holds a point
// with observations is skipped, and an accepted match puts such a point there. Occupancy only ever
// grows, so the work is split as in k_search_proj: first every map point lists, in parallel, its
// best few candidates under the occupancy of the call's start, in the reference's order of
// preference (distance, cell, index: strict comparisons make the first of equal distances the best
// and the second of them the second best). Then one wave walks the map points in order: the first
// two listed candidates that are still free are the reference's best and second best; a list that
// ends before it decides is replaced by a search under the current occupancy.
// Candidates that can decide nothing are left out of the lists: beyond TH_HIGH none is an
// acceptable best, and as a second best it matters only if an acceptable best fails the ratio test
// against it.
// ------------------------------------------------------------------------------------------------
struct TrackPt { float x, y, xr, view_cos; int level, flags; };   // flags: 1 = mbTrackInView and not bad, 2 = Observations() > 0

#define ML_LDS_PER_MP 17      // dynamic LDS bytes per map point slot
#define ML_LDS_PER_KP 17      // and per keypoint slot

template <bool SPILL = false>
__global__ __launch_bounds__(MP_THREADS) void k_search_map(const TrackPt* __restrict__ mp, const uint8_t* __restrict__ desc1,
                                                           const int* __restrict__ cnt1, int cap1,
                                                           const GridKp* __restrict__ kp2, const uint8_t* __restrict__ desc2,
                                                           const float* __restrict__ uright2, const uint8_t* __restrict__ occupied2,
                                                           const int* __restrict__ cnt2, int cap2, float min_x, float min_y,
                                                           float inv_w, float inv_h, float th, float nnratio, ScaleTab sf, int nlevels,
                                                           int* __restrict__ match2, int* __restrict__ nmatches_out,
                                                           uint8_t* __restrict__ spill = nullptr, size_t spill_stride = 0) {
    extern __shared__ __align__(16) uint8_t ml_lds[];
    uint8_t* const ml_smem = SPILL ? spill + (size_t)blockIdx.x * spill_stride : ml_lds;   // as in k_search_proj
    uint32_t* list1 = reinterpret_cast<uint32_t*>(ml_smem);                  // map point: MP_LIST x (distance << 16 | keypoint), best first
    float2* pos2 = reinterpret_cast<float2*>(list1 + MP_LIST * (size_t)cap1);  // keypoint: position
    float* ur2 = reinterpret_cast<float*>(pos2 + cap2);                      // mvuRight
    uint16_t* sorted2 = reinterpret_cast<uint16_t*>(ur2 + cap2);             // keypoint indices bucketed by grid cell
    uint16_t* cell2 = sorted2 + cap2;                                        // grid cell, 0xFFFF = outside
    uint8_t* oo2 = reinterpret_cast<uint8_t*>(cell2 + cap2);                 // octave | 0x80 when it holds a point with observations
    uint8_t* st1 = oo2 + cap2;                                               // map point: 1 usable | 2 observed
    __shared__ uint32_t s_cur[GRID_CELLS];
    __shared__ uint16_t s_start[GRID_CELLS + 2];
    __shared__ uint32_t s_magic[GRID_ROWS + 1];
    __shared__ uint32_t s_tmp[MP_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, f = blockIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t o1 = (size_t)f * cap1, o2 = (size_t)f * cap2;
    const int n1 = min(cnt1[f], cap1), n2 = min(cnt2[f], cap2);
    const TrackPt* P = mp + o1;
    const GridKp* K2 = kp2 + o2;
    int* m2 = match2 + o2;

    for (int c = tid; c < GRID_CELLS; c += MP_THREADS) s_cur[c] = 0;
    if (tid >= 1 && tid <= GRID_ROWS) s_magic[tid] = ((1u << 20) + tid - 1) / tid;
    __syncthreads();
    for (int i = tid; i < n2; i += MP_THREADS) {
        const GridKp k = K2[i];
        const int cell = grid_cell(k.x, k.y, min_x, min_y, inv_w, inv_h);
        pos2[i] = make_float2(k.x, k.y);
        ur2[i] = uright2 ? uright2[o2 + i] : -1.0f;
        cell2[i] = (uint16_t)(cell < 0 ? 0xFFFF : cell);
        oo2[i] = (uint8_t)((k.octave & 0x7F) | ((occupied2 && occupied2[o2 + i] != 0) ? 0x80 : 0));
        m2[i] = -1;
        if (cell >= 0) atomicAdd(&s_cur[cell], 1u);
    }
    for (int i = tid; i < n1; i += MP_THREADS) {
        const int fl = P[i].flags, lv = P[i].level;
        st1[i] = (uint8_t)(((fl & 1) && lv >= 0 && lv < nlevels) ? (1 | (fl & 2)) : 0);
        list1[MP_LIST * i] = MP_END;
    }
    __syncthreads();
    const uint32_t total = mp_block_excl_scan(s_cur, GRID_CELLS, s_tmp);
    for (int c = tid; c < GRID_CELLS; c += MP_THREADS) s_start[c] = (uint16_t)s_cur[c];
    if (tid == 0) s_start[GRID_CELLS] = (uint16_t)total;
    __syncthreads();
    for (int i = tid; i < n2; i += MP_THREADS) {
        const int cell = cell2[i];
        if (cell != 0xFFFF) sorted2[atomicAdd(&s_cur[cell], 1u)] = (uint16_t)i;
    }
    __syncthreads();

    // The two smallest keys  distance << 40 | cell << 20 | index  every lane finds among its share of
    // the free candidates of map point i (lanes over the cells of the window), and how many it saw.
    // prune: leave out the candidates that can decide nothing (see above).
    auto scan = [&](int i, bool prune, unsigned long long& b1, unsigned long long& b2, int& seen) __attribute__((always_inline)) {
        b1 = b2 = ~0ull;
        seen = 0;
        const TrackPt t = P[i];                                // wave-uniform
        float r = (double)t.view_cos > 0.998 ? 2.5f : 4.0f;    // RadiusByViewingCos
        if (th != 1.0f) r *= th;                               // src/ORBmatcher.cc:50,66-67
        r = r * sf.v[t.level];
        const int lo = t.level - 1, hi = t.level;              // GetFeaturesInArea(.., nPredictedLevel-1, nPredictedLevel)
        const int min_cx = max(0, (int)floorf((t.x - min_x - r) * inv_w));
        const int max_cx = min(GRID_COLS - 1, (int)ceilf((t.x - min_x + r) * inv_w));
        const int min_cy = max(0, (int)floorf((t.y - min_y - r) * inv_h));
        const int max_cy = min(GRID_ROWS - 1, (int)ceilf((t.y - min_y + r) * inv_h));
        if (min_cx >= GRID_COLS || max_cx < 0 || min_cy >= GRID_ROWS || max_cy < 0 || max_cx < min_cx || max_cy < min_cy) return;
        const int ncy = max_cy - min_cy + 1, ncell = (max_cx - min_cx + 1) * ncy;
        const uint32_t magic = s_magic[ncy];
        const uint4* d1p = reinterpret_cast<const uint4*>(desc1 + (o1 + i) * 32);
        const uint4 da = d1p[0], db = d1p[1];
        for (int c = lane; c < ncell; c += 64) {
            const int qx = (int)(((uint32_t)c * magic) >> 20);
            const int cell = (min_cx + qx) * GRID_ROWS + min_cy + (c - qx * ncy);
            const int ke = s_start[cell + 1];
            for (int k = s_start[cell]; k < ke; k++) {
                const int i2 = sorted2[k];
                const int oo = oo2[i2];
                if (oo & 0x80) continue;                       // src/ORBmatcher.cc:88-90
                const int oc = oo;
                if (oc < lo || oc > hi) continue;
                const float2 q = pos2[i2];
                if (!(fabsf(q.x - t.x) < r && fabsf(q.y - t.y) < r)) continue;
                const float u2 = ur2[i2];
                if (u2 > 0 && fabsf(t.xr - u2) > r) continue;  // src/ORBmatcher.cc:92-97
                const uint4* d2p = reinterpret_cast<const uint4*>(desc2 + (o2 + i2) * 32);
                const uint4 x = d2p[0], y = d2p[1];
                const unsigned dist = __popc(x.x ^ da.x) + __popc(x.y ^ da.y) + __popc(x.z ^ da.z) + __popc(x.w ^ da.w) +
                                      __popc(y.x ^ db.x) + __popc(y.y ^ db.y) + __popc(y.z ^ db.z) + __popc(y.w ^ db.w);
                if (prune && dist > MP_TH_HIGH && !((float)MP_TH_HIGH > nnratio * (float)dist)) continue;
                const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)cell << 20) | (unsigned)i2;
                seen++;
                if (key < b1) { b2 = b1; b1 = key; }
                else if (key < b2) b2 = key;
            }
        }
    };

    // ---- phase A: candidate lists, one wave per map point
    for (int i = wv; i < n1; i += MP_WAVES) {
        if (!(__builtin_amdgcn_readfirstlane((int)st1[i]) & 1)) continue;
        unsigned long long b1, b2;
        int seen;
        scan(i, true, b1, b2, seen);
        uint32_t e[MP_LIST];
#pragma unroll
        for (int q = 0; q < MP_LIST; q++) e[q] = MP_END;
        int used = 0;
#pragma unroll
        for (int q = 0; q < MP_LIST; q++) {
            const unsigned long long m = wave_min_u64(b1);
            if (m == ~0ull) break;
            e[q] = (uint32_t)((m >> 40) << 16) | (uint32_t)(m & 0xFFFF);
            bool dry = false;                                  // a lane that handed out both keys it kept and saw more: its third might come next
            if (b1 == m) { used++; b1 = b2; b2 = ~0ull; dry = used == 2 && seen > 2; }
            if (__ballot(dry)) { if (q + 1 < MP_LIST) e[q + 1] = MP_NONE; break; }
        }
        if (lane == 0) *reinterpret_cast<uint4*>(&list1[MP_LIST * i]) = make_uint4(e[0], e[1], e[2], e[3]);
    }
    __syncthreads();

    // ---- phase B: the reference's loop order, one wave, scalar control flow on LDS state. The states and
    // lists of 64 map points at a time are fetched into registers (one per lane) and handed out with
    // v_readlane, so that only the occupancy look-ups remain as dependent LDS reads.
    int made = 0;
    if (wv == 0) {
        for (int i0 = 0; i0 < n1; i0 += 64) {
          const int il = min(i0 + lane, n1 - 1);
          const int stv = st1[il];
          const uint4 lv = *reinterpret_cast<const uint4*>(&list1[MP_LIST * il]);
          // points that are not usable or whose list is empty and complete decide nothing: only the others are visited
          unsigned long long todo = __ballot(i0 + lane < n1 && (stv & 1) && lv.x != MP_END);
          while (todo) {
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int i = i0 + k;
            const int st = __builtin_amdgcn_readlane(stv, k);
            const uint32_t le[MP_LIST] = {(uint32_t)__builtin_amdgcn_readlane((int)lv.x, k), (uint32_t)__builtin_amdgcn_readlane((int)lv.y, k),
                                          (uint32_t)__builtin_amdgcn_readlane((int)lv.z, k), (uint32_t)__builtin_amdgcn_readlane((int)lv.w, k)};
            // octave and occupancy of the (up to four) listed keypoints in one LDS round trip: lane q reads entry q's
            const uint32_t myc = lane == 0 ? le[0] : lane == 1 ? le[1] : lane == 2 ? le[2] : le[3];
            const int myo = (lane < MP_LIST && myc < MP_END) ? (int)oo2[myc & 0xFFFF] : 0;
            const int lo_[MP_LIST] = {__builtin_amdgcn_readlane(myo, 0), __builtin_amdgcn_readlane(myo, 1),
                                      __builtin_amdgcn_readlane(myo, 2), __builtin_amdgcn_readlane(myo, 3)};
            uint32_t best = MP_NONE, second = MP_NONE;         // distance << 16 | keypoint
            int best_lv = 0, second_lv = 0;
            bool decided = false;
            uint32_t term = MP_NONE, last_real = MP_NONE;      // how the list ended (MP_NONE also stands for "all slots used")
#pragma unroll
            for (int q = 0; q < MP_LIST; q++) {
                const uint32_t c = le[q];
                if (c >= MP_END) { term = c; break; }
                last_real = c;
                if (lo_[q] & 0x80) continue;
                if (best == MP_NONE) {
                    best = c; best_lv = lo_[q];
                    if ((c >> 16) > MP_TH_HIGH) { decided = true; break; }   // nothing closer is free: no match
                } else { second = c; second_lv = lo_[q]; decided = true; break; }
            }
            if (!decided) {
                if (term == MP_END) decided = true;            // the list was complete: what was found is all there is
                else if (best != MP_NONE && last_real != MP_NONE && !((float)(int)(best >> 16) > nnratio * (float)(int)(last_real >> 16)))
                    decided = true;                            // whatever the second best is, it lies at or beyond the last listed distance
            }
            if (!decided) {
                unsigned long long b1, b2;
                int seen;
                scan(i, false, b1, b2, seen);
                const unsigned long long k1 = wave_min_u64(b1);
                if (b1 == k1) b1 = b2;                          // keys are unique: one lane holds it
                const unsigned long long k2 = wave_min_u64(b1);
                best = k1 == ~0ull ? MP_NONE : (uint32_t)((k1 >> 40) << 16) | (uint32_t)(k1 & 0xFFFF);
                second = k2 == ~0ull ? MP_NONE : (uint32_t)((k2 >> 40) << 16) | (uint32_t)(k2 & 0xFFFF);
                best_lv = best == MP_NONE ? 0 : __builtin_amdgcn_readfirstlane((int)oo2[best & 0xFFFF]);
                second_lv = second == MP_NONE ? 0 : __builtin_amdgcn_readfirstlane((int)oo2[second & 0xFFFF]);
            }
            if (best == MP_NONE) continue;
            const int bestDist = (int)(best >> 16), bestIdx = (int)(best & 0xFFFF);
            if (bestDist > MP_TH_HIGH) continue;               // src/ORBmatcher.cc:116
            if (second != MP_NONE) {
                if ((best_lv & 0x7F) == (second_lv & 0x7F) && (float)bestDist > nnratio * (float)(int)(second >> 16)) continue;   // src/ORBmatcher.cc:118-119
            }
            made++;
            if (lane == 0) {
                m2[bestIdx] = i;                               // a point without observations is simply overwritten
                if (st & 2) oo2[bestIdx] = (uint8_t)(best_lv | 0x80);
            }
          }
        }
        if (lane == 0) nmatches_out[f] = made;
    }
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
// (reference src/ORBmatcher.cc:156-285; Tracking::TrackReferenceKeyFrame and relocalisation) on plain
// arrays. The DBoW2 feature vectors (vocabulary node -> ascending feature indices, nodes ascending)
// arrive as one node id per feature; the matching itself is Hamming work: inside every node the two
// frames share, each key-frame feature that has a map point takes the closest frame feature of
// that node that no earlier feature took, if it is close enough and distinctly closer than the
// second closest.
//
// A frame feature lies in exactly one node, so the nodes are independent and only the loop inside
// a node is ordered. The workgroup sorts (node, index) keys of both sides in LDS (bitonic), cuts
// the key-frame side into runs of equal node and deals the runs to its waves; a wave walks its run in
// order, the lanes sharing the scan over the node's frame features (first of equal distances wins:
// minimum over (distance, position in the node)).
// ------------------------------------------------------------------------------------------------
#define MB_NONE_NODE 0xFFFFFFFFu   // node id of a feature that is in no feature vector
#define MB_LDS_PER_SLOT 22          // dynamic LDS bytes per (power-of-two padded) feature slot

// Bitonic sort of a[0..n) (n a power of two) in LDS by the whole workgroup, ascending.
__device__ __forceinline__ void mb_sort(unsigned long long* a, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < n / 2; t += MP_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // the lower index of the pair
                const int p = i | j;
                const unsigned long long x = a[i], y = a[p];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
    }
}

// KFKF: ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (reference
// src/ORBmatcher.cc:519-652, loop detection): side 2 is a key frame too, only its features with a
// good map point (flags2) are candidates, the acceptance test is bestDist1 < TH_LOW, and the result
// is indexed by the feature of side 1 (match[i1] = feature of side 2).
//
// SPILL: frames with more than 4096 features keep the sort arrays in a slice of global memory per pair
// (26 bytes per padded slot) instead of LDS; the code is the same, the workgroup's barriers order its own
// global accesses. Feature indices travel in 16 bits of the sort keys: at most 65535 features per frame.
template <bool KFKF, bool SPILL = false>
__global__ __launch_bounds__(MP_THREADS) void k_search_bow(const GridKp* __restrict__ kp1, const uint8_t* __restrict__ desc1,
                                                           const uint32_t* __restrict__ node1, const uint8_t* __restrict__ flags1,
                                                           const int* __restrict__ cnt1, const GridKp* __restrict__ kp2,
                                                           const uint8_t* __restrict__ desc2, const uint32_t* __restrict__ node2,
                                                           const uint8_t* __restrict__ flags2,
                                                           const int* __restrict__ cnt2, int cap, int npad, float nnratio,
                                                           int check_orientation, int* __restrict__ match2, int* __restrict__ nmatches_out,
                                                           uint8_t* __restrict__ spill = nullptr, size_t spill_stride = 0) {
    extern __shared__ __align__(16) uint8_t mb_smem[];
    uint8_t* const smem = SPILL ? spill + (size_t)blockIdx.x * spill_stride : mb_smem;
    unsigned long long* A = reinterpret_cast<unsigned long long*>(smem);      // key frame: node << 16 | index, sorted
    unsigned long long* B = A + npad;                                          // frame: the same
    uint32_t* run_start = reinterpret_cast<uint32_t*>(B + npad);              // scan of the run heads, then first position of every run
    uint8_t* taken2 = reinterpret_cast<uint8_t*>(run_start + npad);           // frame feature already holds a match (KFKF: or has no good map point)
    uint8_t* bin2 = taken2 + npad;                                            // rotation bin of the match of the output slot, 0xFF = none
    uint32_t* runs = SPILL ? reinterpret_cast<uint32_t*>(bin2 + npad) : run_start;   // first position of every run (LDS: in place of the flags)
    __shared__ uint32_t s_tmp[MP_WAVES];
    __shared__ int s_hist[MT_HISTO];
    __shared__ int s_count[2];                                                // matches made, matches removed
    const int tid = threadIdx.x, lane = tid & 63, f = blockIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t o = (size_t)f * cap;
    const int n1 = min(cnt1[f], cap), n2 = min(cnt2[f], cap);
    int* m2 = match2 + o;
    for (int i = tid; i < npad; i += MP_THREADS) {
        const uint32_t na = i < n1 ? node1[o + i] : MB_NONE_NODE, nb = i < n2 ? node2[o + i] : MB_NONE_NODE;
        A[i] = ((unsigned long long)na << 16) | (unsigned)i;
        B[i] = ((unsigned long long)nb << 16) | (unsigned)i;
        taken2[i] = (uint8_t)(KFKF ? (i < n2 ? !(flags2[o + i] & 1) : 1) : 0);   // src/ORBmatcher.cc:571-577
        bin2[i] = 0xFF;
        if (i < (KFKF ? n1 : n2)) m2[i] = -1;
    }
    if (tid < MT_HISTO) s_hist[tid] = 0;
    if (tid < 2) s_count[tid] = 0;
    __syncthreads();
    mb_sort(A, npad);
    mb_sort(B, npad);
    // runs of equal node on the key-frame side (features without a node sort to the end and form no run)
    for (int i = tid; i < npad; i += MP_THREADS) {
        const uint32_t nd = (uint32_t)(A[i] >> 16);
        run_start[i] = (nd != MB_NONE_NODE && (i == 0 || (uint32_t)(A[i - 1] >> 16) != nd)) ? 1u : 0u;
    }
    __syncthreads();
    uint32_t mine[4];                                          // LDS: npad <= 4096, at most 4 positions per thread
    const int per = npad / MP_THREADS > 0 ? npad / MP_THREADS : 1;
    // exclusive scan over consecutive chunks (thread t owns positions t*per .. t*per+per-1)
    uint32_t s = 0;
    if (SPILL) {
        for (int k = 0; k < per; k++) s += run_start[tid * per + k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = tid * per + k; mine[k] = (k < per && i < npad) ? run_start[i] : 0u; s += mine[k]; }
    }
    const uint32_t inc = wave_scan_incl(s);
    if (lane == 63) s_tmp[wv] = inc;
    __syncthreads();
    uint32_t off = 0, nruns = 0;
#pragma unroll
    for (int w = 0; w < MP_WAVES; w++) { if (w < wv) off += s_tmp[w]; nruns += s_tmp[w]; }
    uint32_t run = off + inc - s;
    __syncthreads();                                           // every thread has read its flags: the array is reused for the run starts
    if (SPILL) {
        for (int k = 0; k < per; k++) {                        // (a separate array: the flags are read again here)
            const int i = tid * per + k;
            if (run_start[i]) { runs[run] = (uint32_t)i; run++; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = tid * per + k;
            if (mine[k]) { run_start[run] = (uint32_t)i; run++; }
        }
    }
    __syncthreads();

    int made = 0;
    for (uint32_t r = wv; r < nruns; r += MP_WAVES) {
        const int start = (int)runs[r];
        const uint32_t nd = (uint32_t)(A[start] >> 16);
        // the frame's features of this node: B[fs .. fe)
        int lo = 0, hi = npad;                                 // lower bound of nd << 16
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)(B[mid] >> 16) < nd) lo = mid + 1; else hi = mid; }
        const int fs = lo;
        hi = npad;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)(B[mid] >> 16) <= nd) lo = mid + 1; else hi = mid; }
        const int fe = lo;
        if (fe == fs) continue;                                // the frame has no feature in this node
        for (int ia = start; ia < npad && (uint32_t)(A[ia] >> 16) == nd; ia++) {
            const int i1 = (int)(A[ia] & 0xFFFF);
            if (!(flags1[o + i1] & 1)) continue;               // no map point, or a bad one (src/ORBmatcher.cc:188-194)
            const uint4* d1p = reinterpret_cast<const uint4*>(desc1 + (o + i1) * 32);
            const uint4 da = d1p[0], db = d1p[1];
            uint32_t best = 0xFFFFFFFFu, second = 0x7FFFFFFFu;  // distance << 16 | position in the node; second-best distance
            for (int c = fs + lane; c < fe; c += 64) {
                const int i2 = (int)(B[c] & 0xFFFF);
                if (taken2[i2]) continue;                      // src/ORBmatcher.cc:206-207
                const uint4* d2p = reinterpret_cast<const uint4*>(desc2 + (o + i2) * 32);
                const uint4 x = d2p[0], y = d2p[1];
                const unsigned dist = __popc(x.x ^ da.x) + __popc(x.y ^ da.y) + __popc(x.z ^ da.z) + __popc(x.w ^ da.w) +
                                      __popc(y.x ^ db.x) + __popc(y.y ^ db.y) + __popc(y.z ^ db.z) + __popc(y.w ^ db.w);
                const uint32_t key = (dist << 16) | (uint32_t)(c - fs);
                if (key < best) { second = min(second, best == 0xFFFFFFFFu ? 0x7FFFFFFFu : best >> 16); best = key; }
                else second = min(second, dist);
            }
            uint32_t wbest = best;
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) wbest = min(wbest, (uint32_t)__shfl_xor((int)wbest, sh, 64));
            if (wbest == 0xFFFFFFFFu) continue;                // every feature of the node is taken: bestDist1 stays 256
            uint32_t other = best == wbest ? second : min(second, best == 0xFFFFFFFFu ? 0x7FFFFFFFu : best >> 16);
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) other = min(other, (uint32_t)__shfl_xor((int)other, sh, 64));
            const int bestDist1 = (int)(wbest >> 16);
            const int bestDist2 = other == 0x7FFFFFFFu ? 256 : (int)other;
            if ((KFKF ? bestDist1 < MT_TH_LOW : bestDist1 <= MT_TH_LOW) && (float)bestDist1 < nnratio * (float)bestDist2) {   // src/ORBmatcher.cc:225-227, 595-597
                const int i2 = (int)(B[fs + (int)(wbest & 0xFFFF)] & 0xFFFF);
                made++;
                if (lane == 0) {
                    taken2[i2] = 1;
                    const int slot = KFKF ? i1 : i2;            // vpMatches12[idx1] / vpMapPointMatches[bestIdxF]
                    m2[slot] = KFKF ? i2 : i1;
                    if (check_orientation) {
                        float rot = kp1[o + i1].angle - kp2[o + i2].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * (1.0f / MT_HISTO));
                        if (bin == MT_HISTO) bin = 0;
                        bin2[slot] = (uint8_t)bin;
                        atomicAdd(&s_hist[bin], 1);
                    }
                }
            }
        }
    }
    if (lane == 0 && made) atomicAdd(&s_count[0], made);
    __syncthreads();                                           // all matches are made, their stores visible to the workgroup
    int bad = 0;
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;   // ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607)
        for (int i = 0; i < MT_HISTO; i++) {
            const int sz = s_hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = tid; i < (KFKF ? n1 : n2); i += MP_THREADS) {
            const int b = bin2[i];
            if (b != 0xFF && b != ind1 && b != ind2 && b != ind3) { m2[i] = -1; bad++; }   // src/ORBmatcher.cc:272-281, 639-648
        }
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) bad += __shfl_xor(bad, sh, 64);
    if (lane == 0 && bad) atomicAdd(&s_count[1], bad);
    __syncthreads();
    if (tid == 0) nmatches_out[f] = s_count[0] - s_count[1];
}

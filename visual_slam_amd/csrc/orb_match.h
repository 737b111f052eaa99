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

#define MT_TH_LOW 50          // ORBmatcher::TH_LOW (src/ORBmatcher.cc:39)
#define MT_HISTO 30           // ORBmatcher::HISTO_LENGTH (src/ORBmatcher.cc:40)

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, s, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), s, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
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

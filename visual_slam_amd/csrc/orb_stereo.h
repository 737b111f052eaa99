// Stereo matching of the keypoints of a rectified pair, Frame::ComputeStereoMatches of the reference
// (src/Frame.cc:481-653), on the keypoints / descriptors / pyramids that the two extractions left
// in device memory.
//
//   k_stereo_prep    one thread per right keypoint: its row band, octave and x as one 16-byte record
//   k_stereo_match   one wave per left keypoint:
//     - candidates are the right keypoints whose row band [floor(y - 2 s), ceil(y + 2 s)] contains
//       the left keypoint's row, whose octave differs by at most one and whose x lies in
//       [uL - maxD, uL] (src/Frame.cc:498-508, 548-553). The reference keeps per-row index lists;
//       the winner is the smallest Hamming distance below TH_HIGH, first index on ties
//       (src/Frame.cc:537-563), which is a minimum over (distance, index) and needs no lists:
//       every lane tests a strided share of all right keypoints.
//     - 11x11 SAD (both patches minus their centre pixel) at 11 horizontal offsets in the pyramid
//       level of the left keypoint, parabola through the best three, disparity test
//       (src/Frame.cc:566-637). 121 (row, offset) work items over the lanes, partial sums added in LDS.
//   k_stereo_median  one workgroup per frame: median of the SAD values of the matches, matches at
//       or above 1.5 * 1.4 * median are dropped (src/Frame.cc:640-652).
#pragma once
#include <hip/hip_runtime.h>
#include "orb_device.h"

#define ST_TH_HIGH 100        // ORBmatcher::TH_HIGH (src/ORBmatcher.cc:38)
#define ST_TH_LOW 50          // ORBmatcher::TH_LOW  (src/ORBmatcher.cc:39)
#define ST_CAND 256           // candidate list of one left keypoint in LDS

struct StKp { float x, y, size, angle, response; int octave, class_id; };   // cv::KeyPoint record (28 bytes)

// One workgroup per frame: the right keypoints as 16-byte records {minr, maxr, octave | index << 8,
// bits of x} (row band as in src/Frame.cc:498-508), counting-sorted into bins of ST_BIN_ROWS rows
// by minr. binstart[f][b] .. binstart[f][b + 1] is the record range of bin b. A left keypoint in
// row v can only match records with minr in [v - band, v], i.e. a few consecutive bins.
#define ST_BIN_SHIFT 4
#define ST_BIN_ROWS (1 << ST_BIN_SHIFT)
#define ST_MAX_BINS 1040          // 16384 rows / 16 + slack
__global__ __launch_bounds__(256) void k_stereo_prep(const PlanDev* __restrict__ plan, const StKp* __restrict__ kp_r,
                                                     const int* __restrict__ cnt_r, int cap, int nbins,
                                                     int4* __restrict__ rinfo, int* __restrict__ binstart) {
    __shared__ uint32_t s_hist[ST_MAX_BINS + 1];
    __shared__ uint32_t s_tmp[4];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int nr = min(cnt_r[f], cap);
    const StKp* K = kp_r + (size_t)f * cap;
    for (int b = tid; b <= nbins; b += 256) s_hist[b] = 0;
    __syncthreads();
    auto record = [&](int i, int& bin) -> int4 {
        const StKp k = K[i];
        const float r = 2.0f * plan->lv[k.octave].scale;
        const int minr = (int)floorf(k.y - r), maxr = (int)ceilf(k.y + r);
        bin = min(max(minr >> ST_BIN_SHIFT, 0), nbins - 1);
        return make_int4(minr, maxr, k.octave | (i << 8), __float_as_int(k.x));
    };
    for (int i = tid; i < nr; i += 256) { int bin; record(i, bin); atomicAdd(&s_hist[bin], 1u); }
    __syncthreads();
    const uint32_t total = block_excl_scan<256>(s_hist, nbins, s_tmp);
    if (tid == 0) s_hist[nbins] = total;
    __syncthreads();
    int* bs = binstart + (size_t)f * (nbins + 1);
    for (int b = tid; b <= nbins; b += 256) bs[b] = (int)s_hist[b];
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        int bin;
        const int4 rec = record(i, bin);
        rinfo[(size_t)f * cap + atomicAdd(&s_hist[bin], 1u)] = rec;
    }
}

__global__ __launch_bounds__(256) void k_stereo_match(const PlanDev* __restrict__ plan, const int4* __restrict__ rinfo,
                                                      const int* __restrict__ binstart, int nbins, int band,
                                                      const uint8_t* __restrict__ pyr_l, const uint8_t* __restrict__ pyr_r,
                                                      const StKp* __restrict__ kp_l, const uint8_t* __restrict__ desc_l,
                                                      const int* __restrict__ cnt_l,
                                                      const StKp* __restrict__ kp_r, const uint8_t* __restrict__ desc_r,
                                                      int cap, float mb, float mbf,
                                                      float* __restrict__ uright, float* __restrict__ depth,
                                                      int* __restrict__ sad_out, int nframes, int ngroups,
                                                      unsigned ngroups8_magic, Level0View v0l, Level0View v0r) {
    __shared__ int s_dist[4][12];
    __shared__ uint16_t s_cand[4][ST_CAND];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // One-dimensional grid: 8 consecutive workgroups (one per XCD, each with its own L2) take the
    // same group of 4 left keypoints in 8 consecutive pairs: an XCD stays with one pair at a time,
    // whose two pyramids and right-image records then sit in its L2.
    const unsigned id = blockIdx.x;
    const unsigned fb = ngroups8_magic ? __umulhi(id, ngroups8_magic) : id;    // id / (8 * ngroups)
    const unsigned within = id - fb * (unsigned)(8 * ngroups);
    const int f = (int)(fb * 8 + (within & 7));
    if (f >= nframes) return;
    const int iL = (int)(within >> 3) * 4 + wv;
    const int nl = min(cnt_l[f], cap);
    if (iL >= nl) return;                                      // waves are independent: no workgroup barrier below
    const size_t o = (size_t)f * cap;
    const StKp kL = kp_l[o + iL];
    const int levelL = kL.octave;
    const float vL = kL.y, uL = kL.x;
    float out_u = -1.0f, out_d = -1.0f;                        // src/Frame.cc:483-484
    int out_sad = -1;
    const float minD = 0.f, maxD = __fdiv_rn(mbf, mb);         // src/Frame.cc:511-513
    const float minU = uL - maxD, maxU = uL - minD;
    const int row = (int)vL;
    // ---- best right keypoint by descriptor distance
    uint32_t best = 0xFFFFFFFFu;                               // distance << 16 | index
    if (!(maxU < 0)) {
        const uint4* dLp = reinterpret_cast<const uint4*>(desc_l + (o + iL) * 32);
        const uint4 dl0 = dLp[0], dl1 = dLp[1];
        auto hamming = [&](int iR) {
            const uint4* dRp = reinterpret_cast<const uint4*>(desc_r + (o + iR) * 32);
            const uint4 a = dRp[0], b = dRp[1];
            const int dist = __popc(a.x ^ dl0.x) + __popc(a.y ^ dl0.y) + __popc(a.z ^ dl0.z) + __popc(a.w ^ dl0.w) +
                             __popc(b.x ^ dl1.x) + __popc(b.y ^ dl1.y) + __popc(b.z ^ dl1.z) + __popc(b.w ^ dl1.w);
            if (dist < ST_TH_HIGH) best = min(best, ((uint32_t)dist << 16) | (uint32_t)iR);
        };
        // Scan of the bins that can hold a candidate: 4 records per lane and trip, loads first. The
        // indices that pass the geometric tests are compacted into a list in LDS so that their
        // descriptors can then be fetched 64 at a time; what does not fit is handled on the spot.
        uint16_t* cand = s_cand[wv];
        int ncand = 0;
        const int* bs = binstart + (size_t)f * (nbins + 1);
        const int r_lo = bs[min(max((row - band) >> ST_BIN_SHIFT, 0), nbins - 1)];
        const int r_hi = bs[min(max(row >> ST_BIN_SHIFT, 0), nbins - 1) + 1];
        for (int base = r_lo; base < r_hi; base += 4 * 64) {     // wave-uniform trip count: ncand stays uniform
            int4 ri[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int p = base + lane + 64 * u;
                ri[u] = make_int4(1, 0, 0, 0);                 // empty row band: never a candidate
                if (p < r_hi) ri[u] = rinfo[o + p];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int iR = ri[u].z >> 8, octR = ri[u].z & 0xFF;
                const float uR = __int_as_float(ri[u].w);
                const bool ok = row >= ri[u].x && row <= ri[u].y && octR >= levelL - 1 && octR <= levelL + 1 &&
                                uR >= minU && uR <= maxU;
                const unsigned long long m = __ballot(ok);
                if (ok) {
                    const int pos = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (pos < ST_CAND) cand[pos] = (uint16_t)iR; else hamming(iR);
                }
                ncand += __popcll(m);
            }
        }
        ncand = min(ncand, ST_CAND);
        for (int c0 = 0; c0 < ncand; c0 += 64)
            if (c0 + lane < ncand) hamming(cand[c0 + lane]);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, s, 64));
    const int bestDist = best == 0xFFFFFFFFu ? ST_TH_HIGH : (int)(best >> 16);
    const int thOrbDist = (ST_TH_HIGH + ST_TH_LOW) / 2;        // src/Frame.cc:486
    if (bestDist < thOrbDist) {
        // ---- sub-pixel match by correlation (src/Frame.cc:566-637)
        const int bestIdxR = (int)(best & 0xFFFF);
        const float uR0 = kp_r[o + bestIdxR].x;
        const LevelGeom& g = plan->lv[levelL];
        const float scaleFactor = __fdiv_rn(1.0f, g.scale);    // mvInvScaleFactors[octave]
        const float scaleduL = roundf(kL.x * scaleFactor);
        const float scaledvL = roundf(kL.y * scaleFactor);
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (!(iniu < 0 || endu >= (float)g.w)) {
            // pixel (0, 0) of the level on either side: in its plane, or (level 0 read in place) in the caller's image;
            // every byte read below lies inside the level (rows around a keypoint, columns checked by iniu / endu)
            const size_t plane_org = g.pyr_off + (size_t)f * g.plane_bytes + (size_t)ORB_OY * g.pitch + ORB_OX;
            const bool inl = levelL == 0 && v0l.org != nullptr, inr = levelL == 0 && v0r.org != nullptr;
            const uint8_t* PL = inl ? v0l.org + (size_t)f * v0l.frame_bytes : pyr_l + plane_org;
            const uint8_t* PR = inr ? v0r.org + (size_t)f * v0r.frame_bytes : pyr_r + plane_org;
            const int pitch_l = inl ? v0l.pitch : g.pitch, pitch_r = inr ? v0r.pitch : g.pitch;
            const int yl0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w), xr00 = (int)(scaleduR0 - L - w);
            int* dist = s_dist[wv];
            if (lane < 11) dist[lane] = 0;
            // (24-bit multiplies run at full rate; a plane is far below 2^24 rows x 2^24 bytes of pitch)
            const int cL = PL[__mul24(yl0 + w, pitch_l) + xl0 + w];
            for (int t = lane; t < 121; t += 64) {
                const int i = __mul24(t, 745) >> 13, k = t - i * 11;   // t / 11 for t < 121: patch row, offset index (incR = k - 5)
                const uint8_t* rl = PL + (__mul24(yl0 + i, pitch_l) + xl0);
                const uint8_t* rr = PR + (__mul24(yl0 + i, pitch_r) + xr00 + k);
                const int cR = PR[__mul24(yl0 + w, pitch_r) + xr00 + k + w];
                // |(l - cL) - (r - cR)| = |(l + da) - (r + db)| with da = max(cR - cL, 0), db = max(cL - cR, 0):
                // both sides stay non-negative, so the unsigned sum-of-absolute-differences instruction applies
                const uint32_t da = (uint32_t)max(cR - cL, 0), db = (uint32_t)max(cL - cR, 0);
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 11; j++) acc = __usad((uint32_t)rl[j] + da, (uint32_t)rr[j] + db, acc);
                atomicAdd(&dist[k], (int)acc);
            }
            // LDS operations of one wave complete in order: the sums are final when read back
            const int dk = lane < 11 ? dist[lane] : 0x7FFFFF;
            uint32_t bk = ((uint32_t)dk << 8) | (uint32_t)lane; // smallest distance, first offset on ties
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) bk = min(bk, (uint32_t)__shfl_xor((int)bk, s, 64));
            const int kbest = (int)(bk & 0xFF), bestincR = kbest - L;
            if (!(bestincR == -L || bestincR == L)) {
                const float dist1 = (float)dist[kbest - 1], dist2 = (float)dist[kbest], dist3 = (float)dist[kbest + 1];
                const float deltaR = __fdiv_rn(dist1 - dist3, 2.0f * (dist1 + dist3 - 2.0f * dist2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = g.scale * ((float)scaleduR0 + (float)bestincR + deltaR);
                    float disparity = uL - bestuR;
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = (float)0.01;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        out_d = __fdiv_rn(mbf, disparity);
                        out_u = bestuR;
                        out_sad = (int)(bk >> 8);
                    }
                }
            }
        }
    }
    if (lane == 0) {
        uright[o + iL] = out_u;
        depth[o + iL] = out_d;
        sad_out[o + iL] = out_sad;
    }
}

// One workgroup per frame. nmatch[f] receives the number of matches kept.
__global__ __launch_bounds__(256) void k_stereo_median(const int* __restrict__ cnt_l, int cap, const int* __restrict__ sad,
                                                       float* __restrict__ uright, float* __restrict__ depth,
                                                       int* __restrict__ nmatch) {
    __shared__ int s_cnt;
    const int tid = threadIdx.x, f = blockIdx.x;
    const int nl = min(cnt_l[f], cap);
    const size_t o = (size_t)f * cap;
    auto count_le = [&](int v) -> int {                         // matches with SAD <= v (v = -1: none)
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int c = 0;
        for (int i = tid; i < nl; i += 256) { const int s = sad[o + i]; c += s >= 0 && s <= v; }
        if (c) atomicAdd(&s_cnt, c);
        __syncthreads();
        const int r = s_cnt;
        __syncthreads();
        return r;
    };
    const int m = count_le(0x7FFFFFFF);
    if (m == 0) { if (tid == 0) nmatch[f] = 0; return; }
    // element m/2 of the sorted distances = smallest v with count(<= v) >= m/2 + 1
    int lo = -1, hi = 121 * 510;                                // count_le(lo) < need <= count_le(hi)
    const int need = m / 2 + 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (count_le(mid) >= need) hi = mid; else lo = mid;
    }
    const float median = (float)hi;
    const float thDist = 1.5f * 1.4f * median;                  // src/Frame.cc:642
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int kept = 0;
    for (int i = tid; i < nl; i += 256) {
        const int s = sad[o + i];
        if (s < 0) continue;
        if ((float)s < thDist) kept++;
        else { uright[o + i] = -1.0f; depth[o + i] = -1.0f; }
    }
    if (kept) atomicAdd(&s_cnt, kept);
    __syncthreads();
    if (tid == 0) nmatch[f] = s_cnt;
}

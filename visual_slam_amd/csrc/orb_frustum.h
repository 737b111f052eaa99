// Frame::isInFrustum (reference src/Frame.cc:294-350) for arrays of map points: projection with
// the frame's pose, image bounds, the scale-invariance distance range, the viewing angle and the
// predicted pyramid level (MapPoint::PredictScale, src/MapPoint.cc:401-417). The outputs are the
// fields the reference stores in the MapPoint for ORBmatcher::SearchByProjection(F, vpMapPoints, th).
//
// Every operation is the one the reference's expression performs: Rcw * P + tcw is OpenCV's
// small-matrix product for CV_32F (float products summed in float, the translation added in
// double, one rounding), cv::norm and Mat::dot of three floats accumulate in double, and logf is the
// host C library's, restated in orb_libm.h.
#pragma once
#include <hip/hip_runtime.h>
#include "orb_device.h"
#include "orb_libm.h"
#include "orb_match.h"

struct MapPt { float pos[3], normal[3], min_distance, max_distance; int flags; };   // flags: 1 = test it, 2 = Observations() > 0

struct FrustumArgs {
    float fx, fy, cx, cy, mbf;
    float min_x, max_x, min_y, max_y;
    float viewing_cos_limit, log_scale_factor;
    int nlevels;
};

// pose: 15 floats per frame: Rcw row-major (9), tcw (3), Ow = -Rcw^T tcw (3) (Frame::UpdatePoseMatrices).
__global__ __launch_bounds__(256) void k_in_frustum(const MapPt* __restrict__ mp, const int* __restrict__ cnt, int cap,
                                                    const float* __restrict__ pose, FrustumArgs a,
                                                    TrackPt* __restrict__ out, int* __restrict__ nvisible) {
    const int f = blockIdx.y;
    const int n = min(cnt[f], cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* T = pose + 15 * (size_t)f;
    bool vis = false;
    if (i < n) {
        const MapPt p = mp[(size_t)f * cap + i];
        TrackPt t = {0.0f, 0.0f, 0.0f, 0.0f, 0, p.flags & 2};
        if (p.flags & 1) {
            const float X = p.pos[0], Y = p.pos[1], Z = p.pos[2];
            const float xc = (float)((double)(T[0] * X + T[1] * Y + T[2] * Z) + (double)T[9]);
            const float yc = (float)((double)(T[3] * X + T[4] * Y + T[5] * Z) + (double)T[10]);
            const float zc = (float)((double)(T[6] * X + T[7] * Y + T[8] * Z) + (double)T[11]);
            bool ok = !(zc < 0.0f);                            // src/Frame.cc:308
            const float invz = __fdiv_rn(1.0f, zc);
            const float u = a.fx * xc * invz + a.cx;
            const float v = a.fy * yc * invz + a.cy;
            ok = ok && u == u && v == v;                       // a point in the camera centre: undefined in the reference
            ok = ok && !(u < a.min_x || u > a.max_x) && !(v < a.min_y || v > a.max_y);
            const float max_d = 1.2f * p.max_distance, min_d = 0.8f * p.min_distance;   // src/MapPoint.cc:372-382
            const float px = X - T[12], py = Y - T[13], pz = Z - T[14];                 // PO = P - mOw
            const float dist = (float)sqrt((double)px * (double)px + (double)py * (double)py + (double)pz * (double)pz);   // cv::norm
            ok = ok && !(dist < min_d || dist > max_d);
            const double dot = (double)px * (double)p.normal[0] + (double)py * (double)p.normal[1] + (double)pz * (double)p.normal[2];
            const float view_cos = (float)(dot / (double)dist);
            ok = ok && !(view_cos < a.viewing_cos_limit);
            if (ok) {
                const int scale = predict_scale(p.max_distance, dist, a.log_scale_factor, a.nlevels);
                t.x = u; t.y = v; t.xr = u - a.mbf * invz; t.view_cos = view_cos; t.level = scale;
                t.flags |= 1;
                vis = true;
            }
        }
        out[(size_t)f * cap + i] = t;
    }
    if (nvisible) {
        const unsigned long long m = __ballot(vis);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&nvisible[f], __popcll(m));
    }
}

// Diagnostic: logf_glibc of the floats whose bit patterns are first_bits + i * stride, i < count, so
// that a test can compare the device arithmetic with the host's logf over whole ranges.
__global__ __launch_bounds__(256) void k_debug_logf(unsigned first_bits, unsigned count, unsigned stride, float* __restrict__ out) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < count) out[i] = logf_glibc(__uint_as_float(first_bits + i * stride));
}

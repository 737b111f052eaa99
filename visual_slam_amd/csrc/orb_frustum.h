// Frame::isInFrustum (reference src/Frame.cc:294-350) for arrays of map points: projection with
// the frame's pose, image bounds, the scale-invariance distance range, the viewing angle and the
// predicted pyramid level (MapPoint::PredictScale, src/MapPoint.cc:401-417). The outputs are the
// fields the reference stores in the MapPoint for ORBmatcher::SearchByProjection(F, vpMapPoints, th).
//
// Every operation is the one the reference's expression performs: Rcw * P + tcw is OpenCV's
// small-matrix product for CV_32F (float products summed in float, the translation added in
// double, one rounding), cv::norm and Mat::dot of three floats accumulate in double, and logf is the
// host C library's (glibc 2.28+, sysdeps/ieee754/flt-32/e_logf.c: a 16-entry table and a cubic, all
// in double), restated operation by operation: a host program compared the restatement with libm for
// every positive normal float, with and without fused multiply-adds, and found no difference.
#pragma once
#include <hip/hip_runtime.h>
#include "orb_device.h"
#include "orb_match.h"

struct MapPt { float pos[3], normal[3], min_distance, max_distance; int flags; };   // flags: 1 = test it, 2 = Observations() > 0

struct FrustumArgs {
    float fx, fy, cx, cy, mbf;
    float min_x, max_x, min_y, max_y;
    float viewing_cos_limit, log_scale_factor;
    int nlevels;
};

// logf of a positive, finite, normal float.
__device__ __forceinline__ float logf_glibc(float x) {
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0,
                             0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                             0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                             -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3, -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                             -0x1.252f438e10c1ep-5, 0x0p+0, 0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3, 0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2, 0x1.4043057b6ee09p-2};
    const double ln2 = 0x1.62e42fefa39efp-1;
    const double a0 = -0x1.00ea348b88334p-2, a1 = 0x1.5575b0be00b6ap-2, a2 = -0x1.ffffef20a4123p-2;
    const uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int)tmp >> 23;
    const double z = (double)__uint_as_float(ix - (tmp & 0xff800000u));
    const double r = z * invc[i] - 1.0;
    const double y0 = logc[i] + (double)k * ln2;
    const double r2 = r * r;
    double y = a1 * r + a2;
    y = a0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

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
                // MapPoint::PredictScale: ceil(log(mfMaxDistance / dist) / mfLogScaleFactor), clamped to the pyramid
                const float ratio = __fdiv_rn(p.max_distance, dist);
                int scale = 0;                                 // what the x86 conversion of a non-finite value ends in after the clamp
                if (ratio >= 1.17549435e-38f && ratio < __builtin_inff()) {
                    const float c = ceilf(__fdiv_rn(logf_glibc(ratio), a.log_scale_factor));
                    if (c >= 1.0f) scale = c < (float)a.nlevels ? (int)c : a.nlevels - 1;
                }
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

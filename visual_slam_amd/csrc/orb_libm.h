// Host C library functions restated for the device where the reference's result depends on their
// last bit, and MapPoint::PredictScale on top of logf.
//
// logf is glibc's (2.28+, sysdeps/ieee754/flt-32/e_logf.c: a 16-entry table and a cubic, all in
// double), operation by operation: a host program compared the restatement with libm for every
// positive normal float, with and without fused multiply-adds, and found no difference.
#pragma once
#include <hip/hip_runtime.h>

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

// MapPoint::PredictScale (reference src/MapPoint.cc:401-417): ceil(log(mfMaxDistance / dist) / mfLogScaleFactor)
// clamped to [0, nlevels). A ratio that is not a positive normal number ends, through the x86
// conversion of a non-finite value, at level 0.
__device__ __forceinline__ int predict_scale(float max_distance, float dist, float log_scale_factor, int nlevels) {
    const float ratio = __fdiv_rn(max_distance, dist);
    int scale = 0;
    if (ratio >= 1.17549435e-38f && ratio < __builtin_inff()) {
        const float c = ceilf(__fdiv_rn(logf_glibc(ratio), log_scale_factor));
        if (c >= 1.0f) scale = c < (float)nlevels ? (int)c : nlevels - 1;
    }
    return scale;
}

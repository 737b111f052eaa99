// ORB_SLAM2::ORBextractor on top of libORBhip (see shim/ORBextractor.h). Every method forwards to
// the C ABI entry point that replaces the corresponding part of the reference's
// src/ORBextractor.cc; no image processing happens on the host.
#include "ORBextractor.h"

#include <cassert>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../include/orbhip.h"

namespace ORB_SLAM2
{

static_assert(sizeof(cv::KeyPoint) == sizeof(orbhip_keypoint), "cv::KeyPoint must be the 28-byte record libORBhip writes");

static void check(int rc, orbhip_extractor* h, const char* what)
{
    if (rc != ORBHIP_OK)
        throw std::runtime_error(std::string("ORBextractor (libORBhip) ") + what + ": " + orbhip_strerror(rc) +
                                 (h ? std::string(" - ") + orbhip_last_error(h) : std::string()));
}

// reference: src/ORBextractor.cc:410-470 (tables are computed inside orbhip_create)
ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels,
                           int _iniThFAST, int _minThFAST):
    mpHandle(0), mbFetchPyramid(true), nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels),
    iniThFAST(_iniThFAST), minThFAST(_minThFAST)
{
    orbhip_params p;
    std::memset(&p, 0, sizeof(p));
    p.nfeatures = _nfeatures; p.scale_factor = _scaleFactor; p.nlevels = _nlevels;
    p.ini_th_fast = _iniThFAST; p.min_th_fast = _minThFAST;
    const char* dev = std::getenv("ORBHIP_DEVICE");
    p.device = dev ? std::atoi(dev) : 0;
    const char* blur = std::getenv("ORBHIP_BLUR_VARIANT");
    p.blur_variant = blur ? std::atoi(blur) : ORBHIP_BLUR_LEGACY;
    p.max_batch = 1;
    check(orbhip_create(&p, &mpHandle), 0, "create");
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    orbhip_get_scale_factors(mpHandle, mvScaleFactor.data());
    orbhip_get_inverse_scale_factors(mpHandle, mvInvScaleFactor.data());
    orbhip_get_scale_sigma_squares(mpHandle, mvLevelSigma2.data());
    orbhip_get_inverse_scale_sigma_squares(mpHandle, mvInvLevelSigma2.data());
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor()
{
    if (mpHandle) orbhip_destroy(mpHandle);
}

// reference: src/ORBextractor.cc:1043-1105
void ORBextractor::operator()( cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints,
                      cv::OutputArray _descriptors)
{
    (void)_mask;
    if(_image.empty())
        return;

    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1 );

    const int cap = orbhip_max_keypoints_for_image(mpHandle, image.cols, image.rows);
    std::vector<orbhip_keypoint> kps(cap);
    std::vector<unsigned char> desc((size_t)cap * 32);
    int n = 0;
    check(orbhip_extract(mpHandle, image.data, image.cols, image.rows, image.step, kps.data(), cap, &n, desc.data()),
          mpHandle, "extract");

    if( n == 0 )
        _descriptors.release();
    else
    {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat descriptors = _descriptors.getMat();
        for (int i = 0; i < n; i++)
            std::memcpy(descriptors.ptr(i), &desc[(size_t)i * 32], 32);
    }
    _keypoints.resize(n);
    if (n) std::memcpy(static_cast<void*>(_keypoints.data()), kps.data(), (size_t)n * sizeof(orbhip_keypoint));

    // mvImagePyramid[l] is the inner ROI of a (w+38)x(h+38) buffer, as ComputePyramid builds it
    // (src/ORBextractor.cc:1113-1115); Frame::ComputeStereoMatches slices it (src/Frame.cc:578,595).
    if (mbFetchPyramid)
    {
        for (int level = 0; level < nlevels; ++level)
        {
            int w = 0, h = 0;
            check(orbhip_pyramid_info(mpHandle, level, &w, &h), mpHandle, "pyramid_info");
            cv::Mat temp(h + 38, w + 38, CV_8UC1);
            check(orbhip_pyramid_copy(mpHandle, 0, level, temp.data, temp.step, 1), mpHandle, "pyramid_copy");
            mvImagePyramid[level] = temp(cv::Rect(19, 19, w, h));
        }
    }
}

void ORBextractor::SetRectification(const cv::Mat& M1, const cv::Mat& M2)
{
    if (M1.empty() && M2.empty())
    {
        check(orbhip_set_rectification(mpHandle, NULL, NULL, 0, 0), mpHandle, "set_rectification");
        return;
    }
    // dense rows of floats
    std::vector<float> mx((size_t)M1.rows * M1.cols), my((size_t)M2.rows * M2.cols);
    for (int y = 0; y < M1.rows; y++) std::memcpy(&mx[(size_t)y * M1.cols], M1.ptr(y), (size_t)M1.cols * sizeof(float));
    for (int y = 0; y < M2.rows; y++) std::memcpy(&my[(size_t)y * M2.cols], M2.ptr(y), (size_t)M2.cols * sizeof(float));
    check(M1.rows == M2.rows && M1.cols == M2.cols ? 0 : ORBHIP_ERR_ARG, mpHandle, "set_rectification (map sizes)");
    check(orbhip_set_rectification(mpHandle, mx.data(), my.data(), M1.cols, M1.rows), mpHandle, "set_rectification");
}

// Frame::ComputeStereoMatches (reference src/Frame.cc:481-653)
void ORBextractor::ComputeStereoMatches(ORBextractor& left, ORBextractor& right,
                                        const std::vector<cv::KeyPoint>& keysLeft, const cv::Mat& descLeft,
                                        const std::vector<cv::KeyPoint>& keysRight, const cv::Mat& descRight,
                                        float mb, float mbf, std::vector<float>& uRight, std::vector<float>& depth)
{
    const int nl = (int)keysLeft.size(), nr = (int)keysRight.size();
    uRight.assign(nl, -1.0f);                                  // src/Frame.cc:483-484
    depth.assign(nl, -1.0f);
    if (nl == 0) return;
    // descriptor rows are 32 contiguous bytes each; a Mat may pad rows, so pack them
    std::vector<unsigned char> dl((size_t)nl * 32), dr((size_t)(nr > 0 ? nr : 1) * 32);
    for (int i = 0; i < nl; i++) std::memcpy(&dl[(size_t)i * 32], descLeft.ptr(i), 32);
    for (int i = 0; i < nr; i++) std::memcpy(&dr[(size_t)i * 32], descRight.ptr(i), 32);
    int kept = 0;
    check(orbhip_stereo_match(left.mpHandle, right.mpHandle,
                              reinterpret_cast<const orbhip_keypoint*>(keysLeft.data()), dl.data(), nl,
                              reinterpret_cast<const orbhip_keypoint*>(keysRight.data()), dr.data(), nr,
                              mb, mbf, uRight.data(), depth.data(), &kept),
          left.mpHandle, "stereo_match");
}

} //namespace ORB_SLAM

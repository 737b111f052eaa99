// Drop-in replacement of the reference header include/ORBextractor.h: the same class, the same
// public interface (constructor, operator(), the six accessors, mvImagePyramid, the score enum),
// implemented on top of libORBhip's C ABI (include/orbhip.h). Frame.cc / Tracking.cc of the
// reference compile and link against it unchanged: replace src/ORBextractor.cc by
// shim/ORBextractor.cc in the build and link libORBhip.so.
//
// Differences a caller can observe:
//  * ExtractorNode (reference include/ORBextractor.h:32-43) is an implementation detail of the
//    CPU quadtree and is not provided; nothing outside ORBextractor.cc uses it.
//  * The protected helpers (ComputePyramid, ComputeKeyPointsOctTree, DistributeOctTree) do not
//    exist: the whole operator() runs on the GPU.
//  * SetPyramidFetch(false) (extension) skips the device-to-host copy of the pyramid for callers
//    that never read mvImagePyramid (monocular / RGB-D tracking).
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>

#ifdef ORBHIP_SHIM_MINI_CV
#include "mini_cv.h"
#else
#include <opencv2/core/core.hpp>
#endif

struct orbhip_extractor;

namespace ORB_SLAM2
{

class ORBextractor
{
public:

    enum {HARRIS_SCORE=0, FAST_SCORE=1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels,
                 int iniThFAST, int minThFAST);

    ~ORBextractor();

    // Compute the ORB features and descriptors on an image (mask is ignored, as in the reference).
    void operator()( cv::InputArray image, cv::InputArray mask,
      std::vector<cv::KeyPoint>& keypoints,
      cv::OutputArray descriptors);

    int inline GetLevels(){
        return nlevels;}

    float inline GetScaleFactor(){
        return scaleFactor;}

    std::vector<float> inline GetScaleFactors(){
        return mvScaleFactor;
    }

    std::vector<float> inline GetInverseScaleFactors(){
        return mvInvScaleFactor;
    }

    std::vector<float> inline GetScaleSigmaSquares(){
        return mvLevelSigma2;
    }

    std::vector<float> inline GetInverseScaleSigmaSquares(){
        return mvInvLevelSigma2;
    }

    std::vector<cv::Mat> mvImagePyramid;

    void SetPyramidFetch(bool on) { mbFetchPyramid = on; }

    // Rectification in front of the extraction (extension): the cv::remap(im, imRect, M1, M2,
    // cv::INTER_LINEAR) of the stereo mains (reference Examples/Stereo/stereo_euroc.cc:97-98, 136-137)
    // runs on the GPU on the way into pyramid level 0. M1 / M2 are the CV_32F maps of
    // cv::initUndistortRectifyMap; empty Mats switch it off again.
    void SetRectification(const cv::Mat& M1, const cv::Mat& M2);

    // The C handle, for the other entry points of orbhip.h (undistortion + grid, matchers).
    orbhip_extractor* Handle() { return mpHandle; }

    // Frame::ComputeStereoMatches (reference src/Frame.cc:481-653) on the GPU, for the pair of images
    // the two instances extracted last: fills mvuRight / mvDepth (-1 = no match). With it the body
    // of Frame::ComputeStereoMatches is one call and the pyramids need not be fetched
    // (SetPyramidFetch(false)).
    static void ComputeStereoMatches(ORBextractor& left, ORBextractor& right,
                                     const std::vector<cv::KeyPoint>& keysLeft, const cv::Mat& descLeft,
                                     const std::vector<cv::KeyPoint>& keysRight, const cv::Mat& descRight,
                                     float mb, float mbf, std::vector<float>& uRight, std::vector<float>& depth);

private:
    ORBextractor(const ORBextractor&);
    ORBextractor& operator=(const ORBextractor&);

    orbhip_extractor* mpHandle;
    bool mbFetchPyramid;

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;

    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;
};

} //namespace ORB_SLAM

#endif

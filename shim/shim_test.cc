// Exercises shim/ORBextractor.{h,cc} (built against mini_cv.h) on the GPU and checks it against
// the CPU oracle through the class API the reference's Frame / Tracking use. Test program only.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ORBextractor.h"
#include "../oracle/orb_oracle.h"

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main()
{
    const int W = 752, H = 480;
    cv::Mat img(H, W, CV_8UC1);
    unsigned seed = 7;
    std::memset(img.data, 128, (size_t)W * H);
    for (int k = 0; k < W * H / 600; k++) {
        int x = (int)(lcg(seed) % W), y = (int)(lcg(seed) % H), w = 8 + (int)(lcg(seed) % 89), h = 8 + (int)(lcg(seed) % 89);
        unsigned char g = (unsigned char)(lcg(seed) & 255);
        for (int yy = y; yy < y + h && yy < H; yy++) std::memset(img.ptr(yy) + x, g, (size_t)((x + w < W ? x + w : W) - x));
    }
    ORB_SLAM2::ORBextractor ex(1200, 1.2f, 8, 20, 7);
    CHECK(ex.GetLevels() == 8);
    CHECK(ex.GetScaleFactor() == 1.2f);
    std::vector<float> sf = ex.GetScaleFactors(), isf = ex.GetInverseScaleFactors(), s2 = ex.GetScaleSigmaSquares(), is2 = ex.GetInverseScaleSigmaSquares();
    CHECK(sf.size() == 8 && isf.size() == 8 && s2.size() == 8 && is2.size() == 8);
    for (int i = 0; i < 8; i++) { CHECK(isf[i] == 1.0f / sf[i]); CHECK(s2[i] == sf[i] * sf[i]); CHECK(is2[i] == 1.0f / s2[i]); }

    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    ex(img, cv::Mat(), kps, desc);
    CHECK(kps.size() > 600 && desc.rows == (int)kps.size() && desc.cols == 32);

    // oracle through its C interface
    orbo_extractor* o = orbo_create(1200, 1.2f, 8, 20, 7, ORBO_BLUR_LEGACY, ORBO_TIE_CREATION_ORDER, ORBO_FMA_OFF);
    std::vector<orbo_keypoint> okp(4096);
    std::vector<unsigned char> odesc(4096 * 32);
    int on = 0;
    CHECK(orbo_extract(o, img.data, W, H, img.step, okp.data(), 4096, &on, odesc.data()) == 0);
    CHECK(on == (int)kps.size());
    int desc_bad = 0;
    for (int i = 0; i < on; i++) {
        CHECK(kps[i].pt.x == okp[i].x && kps[i].pt.y == okp[i].y && kps[i].octave == okp[i].octave);
        CHECK(kps[i].response == okp[i].response && kps[i].size == okp[i].size && kps[i].class_id == -1);
        float da = kps[i].angle - okp[i].angle; if (da < 0) da = -da;
        CHECK(da <= 1e-3f);
        desc_bad += std::memcmp(desc.ptr(i), &odesc[(size_t)i * 32], 32) != 0;
    }
    CHECK(desc_bad * 1000 <= on);
    // mvImagePyramid: ROI inside a bordered buffer, REFLECT_101 border, pixels equal to the oracle's
    for (int l = 0; l < 8; l++) {
        const cv::Mat& m = ex.mvImagePyramid[l];
        int w = 0, h = 0;
        orbo_level_size(o, l, &w, &h);
        CHECK(m.cols == w && m.rows == h && (int)m.step == w + 38);
        std::vector<unsigned char> ref((size_t)(w + 38) * (h + 38));
        orbo_level_copy(o, l, ref.data(), (size_t)(w + 38), 1);
        CHECK(std::memcmp(m.data - 19 * m.step - 19, ref.data(), ref.size()) == 0);
        CHECK(m.ptr(5)[-3] == m.ptr(5)[3] && m.ptr(-2)[7] == m.ptr(2)[7]);
    }
    // stereo matching through the class against the oracle's restatement of Frame::ComputeStereoMatches
    {
        const int d = 9;
        cv::Mat imgR(H, W, CV_8UC1);
        for (int y = 0; y < H; y++) {
            std::memcpy(imgR.ptr(y), img.ptr(y) + d, (size_t)(W - d));
            std::memcpy(imgR.ptr(y) + (W - d), img.ptr(y) + (W - d), (size_t)d);
        }
        ORB_SLAM2::ORBextractor exR(1200, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> kpsR;
        cv::Mat descR;
        exR(imgR, cv::Mat(), kpsR, descR);
        const float fx = 435.2f, bf = 47.9f;
        std::vector<float> uR, dep;
        ORB_SLAM2::ORBextractor::ComputeStereoMatches(ex, exR, kps, desc, kpsR, descR, bf / fx, bf, uR, dep);
        orbo_extractor* oR = orbo_create(1200, 1.2f, 8, 20, 7, ORBO_BLUR_LEGACY, ORBO_TIE_CREATION_ORDER, ORBO_FMA_OFF);
        std::vector<orbo_keypoint> okpR(4096);
        std::vector<unsigned char> odescR(4096 * 32);
        int onR = 0;
        CHECK(orbo_extract(oR, imgR.data, W, H, imgR.step, okpR.data(), 4096, &onR, odescR.data()) == 0);
        CHECK(onR == (int)kpsR.size());
        std::vector<float> ouR(on), odep(on);
        const int okept = orbo_stereo_match(o, oR, okp.data(), odesc.data(), on, okpR.data(), odescR.data(), onR, bf / fx, bf,
                                            ouR.data(), odep.data());
        int matched = 0;
        for (int i = 0; i < on; i++) {
            CHECK(std::memcmp(&uR[i], &ouR[i], 4) == 0 && std::memcmp(&dep[i], &odep[i], 4) == 0);
            matched += uR[i] >= 0;
        }
        CHECK(matched == okept && matched > 200);
        orbo_destroy(oR);
        std::printf("shim stereo ok: %d of %d left keypoints matched\n", matched, on);
    }
    // rectification through the class: level 0 is cv::remap of the input as the oracle restates it
    {
        cv::Mat M1(H, W, CV_32FC1), M2(H, W, CV_32FC1);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                reinterpret_cast<float*>(M1.ptr(y))[x] = x + 2.3f + 0.01f * y;
                reinterpret_cast<float*>(M2.ptr(y))[x] = y - 1.6f + 0.02f * x;
            }
        std::vector<float> mx((size_t)W * H), my((size_t)W * H);
        for (int y = 0; y < H; y++) {
            std::memcpy(&mx[(size_t)y * W], M1.ptr(y), (size_t)W * 4);
            std::memcpy(&my[(size_t)y * W], M2.ptr(y), (size_t)W * 4);
        }
        std::vector<unsigned char> rect((size_t)W * H);
        orbo_remap_linear(img.data, W, H, img.step, mx.data(), my.data(), W, H, rect.data(), (size_t)W);
        ORB_SLAM2::ORBextractor exq(1200, 1.2f, 8, 20, 7);
        exq.SetRectification(M1, M2);
        std::vector<cv::KeyPoint> kq;
        cv::Mat dq;
        exq(img, cv::Mat(), kq, dq);
        int bad = 0;
        for (int y = 0; y < H; y++) bad += std::memcmp(exq.mvImagePyramid[0].ptr(y), &rect[(size_t)y * W], (size_t)W) != 0;
        CHECK(bad == 0 && kq.size() > 300);
        exq.SetRectification(cv::Mat(), cv::Mat());
        std::vector<cv::KeyPoint> kq2;
        exq(img, cv::Mat(), kq2, dq);
        CHECK(kq2.size() == kps.size());
        std::printf("shim rectification ok: %d keypoints on the rectified image\n", (int)kq.size());
    }
    // empty input leaves the outputs untouched; second call replaces them
    std::vector<cv::KeyPoint> keep = kps;
    ex(cv::Mat(), cv::Mat(), kps, desc);
    CHECK(kps.size() == keep.size());
    cv::Mat flat(H, W, CV_8UC1);
    std::memset(flat.data, 90, (size_t)W * H);
    ex(flat, cv::Mat(), kps, desc);
    CHECK(kps.empty() && desc.empty());
    orbo_destroy(o);
    std::printf("shim ok: %d keypoints, %d descriptor rows differ\n", on, desc_bad);
    return 0;
}

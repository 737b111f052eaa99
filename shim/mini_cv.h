// mini_cv.h — the few OpenCV types ORBextractor's interface mentions, for compile- and run-testing
// the shim class on machines without OpenCV. TEST SUPPORT ONLY: a real integration compiles
// shim/ORBextractor.{h,cc} against <opencv2/core.hpp> (leave ORBHIP_SHIM_MINI_CV undefined).
// Layouts match OpenCV where the shim relies on them: cv::KeyPoint is 28 bytes, cv::Mat exposes
// rows / cols / step / data and ROI views share the parent's buffer.
#pragma once
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5

typedef unsigned char uchar;

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;

struct Rect {
    int x, y, width, height;
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
};

struct Size {
    int width, height;
    Size(int w_, int h_) : width(w_), height(h_) {}
};

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar* data = nullptr;
    Mat() {}
    Mat(int r, int c, int t) { create(r, c, t); }
    Mat(Size s, int t) { create(s.height, s.width, t); }
    void create(int r, int c, int t) {
        const size_t esz = t == CV_32F ? 4 : 1;
        if (r == rows && c == cols && buf_ && step == (size_t)c * esz && t == type_) return;
        rows = r; cols = c; step = (size_t)c * esz; type_ = t;
        buf_ = std::shared_ptr<uchar>(new uchar[(size_t)r * step + 1], std::default_delete<uchar[]>());
        data = buf_.get();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; buf_.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + c * sizeof(T)); }
    template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + c * sizeof(T)); }
    Mat operator()(const Rect& r) const {
        Mat m;
        m.rows = r.height; m.cols = r.width; m.step = step; m.type_ = type_;
        m.data = data + (size_t)r.y * step + (size_t)r.x * (type_ == CV_32F ? 4 : 1);
        m.buf_ = buf_;
        return m;
    }
private:
    std::shared_ptr<uchar> buf_;
    int type_ = CV_8UC1;
};

class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
private:
    const Mat* m_;
};

class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int rows, int cols, int type) const { m_->create(rows, cols, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat* m_;
};

typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv

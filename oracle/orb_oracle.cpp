/*
 * orb_oracle.cpp — CPU restatement of the reference ORB extractor. TEST INFRASTRUCTURE ONLY
 * (parity checker + timed CPU baseline; never linked into or called from the HIP library).
 *
 * PARITY UNPINNED: the arithmetic of resize / FAST / GaussianBlur / fastAtan2 lives in OpenCV,
 * which the reference does not vendor and which is absent here; those routines are restated
 * from the published OpenCV 3.x generic C++ algorithms. The reference holds no tests or golden
 * vectors. What pins this file is the set of analytic known-answer tests in tests/.
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 * Build with -ffp-contract=off: float expressions must round exactly as written.
 */
#include "orb_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include <thread>
#include <utility>
#include <vector>

namespace {

const int kPatchSize = 31;      /* src/ORBextractor.cc:72 */
const int kHalfPatch = 15;      /* src/ORBextractor.cc:73 */
const int kEdge = 19;           /* src/ORBextractor.cc:74 */

const int kPattern[256 * 4] = { /* src/ORBextractor.cc:150-408 */
#include "../include/orb_pattern_31.inc"
};

/* cvRound: round half to even under the default rounding mode (OpenCV uses cvtsd2si / lrint). */
inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* BORDER_REFLECT_101: ... c b | a b c ... y z | y x ... */
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

struct Plane { /* a (w+2b)x(h+2b) buffer and the inner w x h view, as src/ORBextractor.cc:1113-1115 */
    int w = 0, h = 0, border = 0;
    size_t step = 0;
    std::vector<uint8_t> buf;
    void create(int w_, int h_, int border_) {
        w = w_; h = h_; border = border_;
        step = (size_t)(w + 2 * border);
        buf.assign(step * (size_t)(h + 2 * border), 0);
    }
    uint8_t* px() { return buf.data() + (size_t)border * step + border; }
    const uint8_t* px() const { return buf.data() + (size_t)border * step + border; }
};

/* ---------------------------------------------------------------------------------------------
 * cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (call site src/ORBextractor.cc:1120).
 * Fixed-point bilinear: 11-bit coefficients built from float weights, horizontal pass into int32,
 * vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.
 * ------------------------------------------------------------------------------------------- */
void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride,
                      uint8_t* dst, int dw, int dh, size_t dstride) {
    const int COEF = 2048;
    double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * (size_t)dw), beta(2 * (size_t)dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_short(cv_round((1.f - fx) * COEF));
        alpha[2 * dx + 1] = sat_short(cv_round(fx * COEF));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = sat_short(cv_round((1.f - fy) * COEF));
        beta[2 * dy + 1] = sat_short(cv_round(fy * COEF));
    }
    std::vector<int> row0(dw), row1(dw);
    auto hpass = [&](int sy, std::vector<int>& out) {
        sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        const uint8_t* S = src + (size_t)sy * sstride;
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            out[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
        }
        for (; dx < dw; dx++) out[dx] = S[xofs[dx]] * COEF;
    };
    int cached0 = INT32_MIN, cached1 = INT32_MIN;
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        int c0 = sy0 < 0 ? 0 : (sy0 >= sh ? sh - 1 : sy0);
        int c1 = sy1 < 0 ? 0 : (sy1 >= sh ? sh - 1 : sy1);
        if (c0 == cached1) { std::swap(row0, row1); std::swap(cached0, cached1); }
        if (c0 != cached0) { hpass(c0, row0); cached0 = c0; }
        if (c1 == cached0) { row1 = row0; cached1 = c1; }
        else if (c1 != cached1) { hpass(c1, row1); cached1 = c1; }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = sat_u8((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

/* cv::copyMakeBorder(..., BORDER_REFLECT_101) (call sites src/ORBextractor.cc:1122,1127). */
void border_reflect101(const uint8_t* src, int w, int h, size_t sstride,
                       uint8_t* dst, size_t dstride, int border) {
    for (int y = -border; y < h + border; y++) {
        const uint8_t* S = src + (size_t)reflect101(y, h) * sstride;
        uint8_t* D = dst + (size_t)(y + border) * dstride;
        for (int x = -border; x < w + border; x++) D[x + border] = S[reflect101(x, w)];
    }
}

/* ---------------------------------------------------------------------------------------------
 * cv::FAST(img, keypoints, threshold, true) — FAST-9 on the 16-pixel Bresenham circle of radius 3
 * with 3x3 non-maximum suppression on the corner score (call sites src/ORBextractor.cc:809,814).
 * Pixels closer than 3 to the border of `img` are never corners and score 0 in the NMS.
 * Output is in raster order; response = corner score.
 * ------------------------------------------------------------------------------------------- */
const int kCircle[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* Corner score: the largest t' such that the pixel is still a FAST-9 corner at threshold t',
 * floored by the threshold it was detected with (irrelevant for true corners). */
inline int corner_score(const uint8_t* p, const int* ofs, int threshold) {
    int d[25];
    int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[ofs[k % 16]];
    int a0 = threshold;
    for (int k = 0; k < 16; k++) { /* centre brighter than a whole 9-arc */
        int a = d[k];
        for (int j = 1; j < 9; j++) a = std::min(a, d[k + j]);
        a0 = std::max(a0, a);
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k++) { /* centre darker than a whole 9-arc */
        int b = d[k];
        for (int j = 1; j < 9; j++) b = std::max(b, d[k + j]);
        b0 = std::min(b0, b);
    }
    return -b0 - 1;
}

int fast9(const uint8_t* img, int w, int h, size_t stride, int threshold,
          std::vector<orbo_candidate>& out) {
    out.clear();
    if (w < 7 || h < 7) return 0;
    int ofs[16];
    for (int k = 0; k < 16; k++) ofs[k] = kCircle[k][0] + kCircle[k][1] * (int)stride;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<int> score((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++) {
        const uint8_t* row = img + (size_t)y * stride;
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = row + x;
            int v = p[0];
            int lo = v - threshold, hi = v + threshold;
            /* any 9-arc contains one pixel of every opposite pair: quick rejection */
            int cls = 3;
            for (int k = 0; k < 8 && cls; k++) {
                int a = p[ofs[k]], b = p[ofs[k + 8]];
                int c = (a < lo ? 1 : 0) | (a > hi ? 2 : 0) | (b < lo ? 1 : 0) | (b > hi ? 2 : 0);
                cls &= c;
            }
            if (!cls) continue;
            bool corner = false;
            if (cls & 1) {
                int run = 0;
                for (int k = 0; k < 25; k++) {
                    if (p[ofs[k % 16]] < lo) { if (++run > 8) { corner = true; break; } }
                    else run = 0;
                }
            }
            if (!corner && (cls & 2)) {
                int run = 0;
                for (int k = 0; k < 25; k++) {
                    if (p[ofs[k % 16]] > hi) { if (++run > 8) { corner = true; break; } }
                    else run = 0;
                }
            }
            if (corner) score[(size_t)y * w + x] = corner_score(p, ofs, threshold);
        }
    }
    for (int y = 3; y < h - 3; y++) {
        const int* s = &score[(size_t)y * w];
        for (int x = 3; x < w - 3; x++) {
            int sc = s[x];
            if (!sc) continue;
            if (sc > s[x - 1] && sc > s[x + 1] && sc > s[x - w - 1] && sc > s[x - w] && sc > s[x - w + 1] &&
                sc > s[x + w - 1] && sc > s[x + w] && sc > s[x + w + 1])
                out.push_back(orbo_candidate{x, y, sc});
        }
    }
    return (int)out.size();
}

/* ---------------------------------------------------------------------------------------------
 * cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1
 * (call site src/ORBextractor.cc:1085-1086).
 *
 * LEGACY (OpenCV 2.4 .. 3.4.1, what "tested with 2.4.11 and 3.2" in README.md:68 means):
 *   float kernel exp(-x^2/8)/sum -> *256 -> round = {18,34,49,55,49,34,18} (sum 257, not
 *   renormalised); exact int32 row pass; column pass scaled by 2^-16. The x86 column kernel
 *   works in float and converts with round-half-to-even for the first (w & ~3) pixels of a row;
 *   the scalar tail uses (v + 32768) >> 16 (round half up). Both are reproduced.
 * FIXEDPOINT (OpenCV >= 3.4.2): 8.8 fixed-point kernel that sums to exactly 256, 16.16
 *   accumulation, round half up. The tap values of that path are restated, not verified.
 * ------------------------------------------------------------------------------------------- */
void gaussian_taps(int variant, int taps[4]) { /* taps[0] = centre */
    double k[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; float t = (float)std::exp(-0.5 / 4.0 * x * x); k[i] = t; sum += t; }
    double scale = 1. / sum;
    float kf[7];
    for (int i = 0; i < 7; i++) kf[i] = (float)(k[i] * scale);
    int q[7], qs = 0;
    for (int i = 0; i < 7; i++) { q[i] = cv_round(kf[i] * 256.f); qs += q[i]; }
    if (variant == ORBO_BLUR_FIXEDPOINT) q[3] -= (qs - 256); /* fold the rounding excess into the centre tap */
    taps[0] = q[3]; taps[1] = q[2]; taps[2] = q[1]; taps[3] = q[0];
}

void gaussian_blur7(const uint8_t* src, int w, int h, size_t sstride,
                    uint8_t* dst, size_t dstride, int variant) {
    int t[4];
    gaussian_taps(variant, t);
    std::vector<int> rows((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        int* R = &rows[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            int acc = t[0] * S[x];
            for (int k = 1; k <= 3; k++) acc += t[k] * (S[reflect101(x - k, w)] + S[reflect101(x + k, w)]);
            R[x] = acc;
        }
    }
    const int wv = w & ~3;
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstride;
        const int* R0 = &rows[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            int acc = t[0] * R0[x];
            for (int k = 1; k <= 3; k++)
                acc += t[k] * (rows[(size_t)reflect101(y - k, h) * w + x] + rows[(size_t)reflect101(y + k, h) * w + x]);
            int q = acc >> 16, r = acc & 0xFFFF;
            if (r > 32768) q++;
            else if (r == 32768) {
                if (variant == ORBO_BLUR_LEGACY && x < wv) q += (q & 1); /* cvtps2dq: ties to even */
                else q++;
            }
            D[x] = sat_u8(q);
        }
    }
}

/* cv::fastAtan2 (call site src/ORBextractor.cc:103): 7th-order odd polynomial, degrees in [0,360). */
float fast_atan2(float y, float x) {
    const float s = (float)(180 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = (float)2.2204460492503131e-16;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ---------------------------------------------------------------------------------------------
 * Quadtree ("OctTree") distribution — ExtractorNode::DivideNode and ORBextractor::DistributeOctTree
 * (src/ORBextractor.cc:481-537, 539-763), kept as a std::list walk like the reference.
 * The reference orders equal-sized nodes in its final expansion rounds by heap address
 * (std::pair<int, ExtractorNode*> sort, :591,:684). ORBO_TIE_CREATION_ORDER replaces the address by
 * the node's creation sequence number (what a monotonic allocator would give);
 * ORBO_TIE_HEAP_ADDRESS keeps the literal, allocator-dependent comparison.
 * ------------------------------------------------------------------------------------------- */
struct Key { float x, y, response; };

struct Node {
    std::vector<Key> keys;
    int ulx = 0, uly = 0, urx = 0, bry = 0; /* UL=(ulx,uly) UR=(urx,uly) BL=(ulx,bry) BR=(urx,bry) */
    std::list<Node>::iterator lit;
    bool no_more = false;
    long seq = 0;
};

void divide_node(const Node& p, Node& n1, Node& n2, Node& n3, Node& n4) { /* :481-537 */
    const int halfX = (int)std::ceil(static_cast<float>(p.urx - p.ulx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.bry - p.uly) / 2);
    const int mx = p.ulx + halfX, my = p.uly + halfY;
    n1.ulx = p.ulx; n1.urx = mx;    n1.uly = p.uly; n1.bry = my;
    n2.ulx = mx;    n2.urx = p.urx; n2.uly = p.uly; n2.bry = my;
    n3.ulx = p.ulx; n3.urx = mx;    n3.uly = my;    n3.bry = p.bry;
    n4.ulx = mx;    n4.urx = p.urx; n4.uly = my;    n4.bry = p.bry;
    for (const Key& k : p.keys) {
        if (k.x < (float)mx) {
            if (k.y < (float)my) n1.keys.push_back(k); else n3.keys.push_back(k);
        } else if (k.y < (float)my) n2.keys.push_back(k);
        else n4.keys.push_back(k);
    }
    n1.no_more = n1.keys.size() == 1;
    n2.no_more = n2.keys.size() == 1;
    n3.no_more = n3.keys.size() == 1;
    n4.no_more = n4.keys.size() == 1;
}

typedef std::pair<int, Node*> SizeNode;

struct TieLess {
    int mode;
    bool operator()(const SizeNode& a, const SizeNode& b) const {
        if (a.first != b.first) return a.first < b.first;
        if (mode == ORBO_TIE_HEAP_ADDRESS) return std::less<Node*>()(a.second, b.second);
        return a.second->seq < b.second->seq;
    }
};

/* returns -3 when the reference would divide by zero (nIni == 0) */
int distribute_octtree(const std::vector<Key>& in, int minX, int maxX, int minY, int maxY, int N,
                       int tie_mode, std::vector<Key>& result) {
    result.clear();
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY)); /* :543 */
    if (nIni < 1) return -3;
    const float hX = static_cast<float>(maxX - minX) / nIni;                           /* :545 */
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {                                                   /* :552-563 */
        Node n;
        n.ulx = (int)(hX * static_cast<float>(i));
        n.urx = (int)(hX * static_cast<float>(i + 1));
        n.uly = 0;
        n.bry = maxY - minY;
        n.seq = seq++;
        nodes.push_back(n);
        ini[i] = &nodes.back();
    }
    for (const Key& k : in) {                                                          /* :566-570 */
        size_t idx = (size_t)(k.x / hX);
        if (idx >= ini.size()) return -3;
        ini[idx]->keys.push_back(k);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {                                /* :574-585 */
        if (it->keys.size() == 1) { it->no_more = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }
    bool finish = false;
    std::vector<SizeNode> pending;
    TieLess less{tie_mode};
    auto push_child = [&](Node& c, std::vector<SizeNode>& rec, int* to_expand) {
        if (c.keys.empty()) return;
        c.seq = seq++;
        nodes.push_front(c);
        if (c.keys.size() > 1) {
            if (to_expand) ++*to_expand;
            rec.push_back(std::make_pair((int)c.keys.size(), &nodes.front()));
            nodes.front().lit = nodes.begin();
        }
    };
    while (!finish) {                                                                  /* :594-739 */
        int prev = (int)nodes.size();
        int to_expand = 0;
        pending.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {                            /* :606-665 */
            if (it->no_more) { ++it; continue; }
            Node n1, n2, n3, n4;
            divide_node(*it, n1, n2, n3, n4);
            push_child(n1, pending, &to_expand);
            push_child(n2, pending, &to_expand);
            push_child(n3, pending, &to_expand);
            push_child(n4, pending, &to_expand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev) {                     /* :669-672 */
            finish = true;
        } else if ((int)nodes.size() + to_expand * 3 > N) {                            /* :673-738 */
            while (!finish) {
                prev = (int)nodes.size();
                std::vector<SizeNode> todo = pending;
                pending.clear();
                std::sort(todo.begin(), todo.end(), less);                            /* :684 */
                for (int j = (int)todo.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    divide_node(*todo[j].second, n1, n2, n3, n4);
                    push_child(n1, pending, nullptr);
                    push_child(n2, pending, nullptr);
                    push_child(n3, pending, nullptr);
                    push_child(n4, pending, nullptr);
                    nodes.erase(todo[j].second->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prev) finish = true;
            }
        }
    }
    result.reserve(nodes.size());
    for (const Node& n : nodes) {                                                      /* :744-760 */
        const Key* best = &n.keys[0];
        for (size_t k = 1; k < n.keys.size(); k++)
            if (n.keys[k].response > best->response) best = &n.keys[k];
        result.push_back(*best);
    }
    return 0;
}

} // namespace

/* ============================================================================================ */

struct orbo_extractor {
    int nfeatures, nlevels, ini_th, min_th, blur_variant, tie_mode, fma_mode;
    double scale_factor; /* float argument widened to double, include/ORBextractor.h:98 */
    std::vector<float> sf, inv_sf, sigma2, inv_sigma2;
    std::vector<int> per_level, umax;
    /* state of the last extract */
    std::vector<Plane> pyr, blurred;
    std::vector<std::vector<orbo_candidate>> cand;
    std::vector<std::vector<orbo_keypoint>> level_kp;
};

namespace {

/* IC_Angle (src/ORBextractor.cc:77-104) */
float ic_angle(const Plane& im, float px, float py, const std::vector<int>& umax) {
    int m_01 = 0, m_10 = 0;
    const int step = (int)im.step;
    const uint8_t* c = im.px() + (size_t)cv_round(py) * im.step + cv_round(px);
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * c[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = c[u + v * step], vm = c[u - v * step];
            v_sum += (vp - vm);
            m_10 += u * (vp + vm);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor (src/ORBextractor.cc:107-147) */
void orb_descriptor(const orbo_keypoint& kp, const Plane& im, int fma_mode, uint8_t* desc) {
    const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
    float angle = kp.angle * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t* c = im.px() + (size_t)cv_round(kp.y) * im.step + cv_round(kp.x);
    const int step = (int)im.step;
    auto val = [&](int idx) -> int {
        float x = (float)kPattern[2 * idx], y = (float)kPattern[2 * idx + 1];
        float ry, rx;
        if (fma_mode == ORBO_FMA_ON) { ry = fmaf(x, b, y * a); rx = fmaf(x, a, -(y * b)); }
        else { ry = x * b + y * a; rx = x * a - y * b; }
        return c[cv_round(ry) * step + cv_round(rx)];
    };
    for (int i = 0; i < 32; i++) {
        int v = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = val(16 * i + 2 * j), t1 = val(16 * i + 2 * j + 1);
            v |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)v;
    }
}

} // namespace

extern "C" {

orbo_extractor* orbo_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                            int min_th_fast, int blur_variant, int tie_mode, int fma_mode) {
    if (nfeatures < 1 || nlevels < 1 || nlevels > 32 || !(scale_factor > 1.0f)) return nullptr;
    orbo_extractor* e = new orbo_extractor;
    e->nfeatures = nfeatures; e->nlevels = nlevels; e->ini_th = ini_th_fast; e->min_th = min_th_fast;
    e->blur_variant = blur_variant; e->tie_mode = tie_mode; e->fma_mode = fma_mode;
    e->scale_factor = scale_factor;
    /* src/ORBextractor.cc:415-431 */
    e->sf.resize(nlevels); e->sigma2.resize(nlevels); e->inv_sf.resize(nlevels); e->inv_sigma2.resize(nlevels);
    e->sf[0] = 1.0f; e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->sf[i] = (float)(e->sf[i - 1] * e->scale_factor);
        e->sigma2[i] = e->sf[i] * e->sf[i];
    }
    for (int i = 0; i < nlevels; i++) {
        e->inv_sf[i] = 1.0f / e->sf[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    /* src/ORBextractor.cc:435-446 */
    e->per_level.resize(nlevels);
    float factor = (float)(1.0f / e->scale_factor);
    float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        e->per_level[l] = cv_round(desired);
        sum += e->per_level[l];
        desired *= factor;
    }
    e->per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
    /* src/ORBextractor.cc:454-469 */
    e->umax.assign(kHalfPatch + 2, 0);
    int v, v0, vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    e->umax.resize(kHalfPatch + 1);
    return e;
}

void orbo_destroy(orbo_extractor* e) { delete e; }

int orbo_extract(orbo_extractor* e, const uint8_t* img, int w, int h, size_t stride,
                 orbo_keypoint* kp_out, int cap, int* n_out, uint8_t* desc_out) {
    if (n_out) *n_out = 0;
    if (!img || w <= 0 || h <= 0) return -1;                                   /* :1046-1047 */
    const int nl = e->nlevels;
    /* ComputePyramid (:1107-1132) */
    e->pyr.assign(nl, Plane());
    for (int l = 0; l < nl; l++) {
        float scale = e->inv_sf[l];
        int lw = cv_round((float)w * scale), lh = cv_round((float)h * scale);
        if (lw < 1 || lh < 1) return -3;
        Plane& P = e->pyr[l];
        P.create(lw, lh, kEdge);
        if (l != 0) {
            const Plane& S = e->pyr[l - 1];
            std::vector<uint8_t> tmp((size_t)lw * lh);
            resize_linear_u8(S.px(), S.w, S.h, S.step, tmp.data(), lw, lh, (size_t)lw);
            border_reflect101(tmp.data(), lw, lh, (size_t)lw, P.buf.data(), P.step, kEdge);
        } else {
            border_reflect101(img, w, h, stride, P.buf.data(), P.step, kEdge);
        }
    }
    /* ComputeKeyPointsOctTree (:765-853) */
    e->cand.assign(nl, {});
    e->level_kp.assign(nl, {});
    const float W = 30;
    std::vector<orbo_candidate> cell;
    for (int l = 0; l < nl; l++) {
        const Plane& P = e->pyr[l];
        const int minBX = kEdge - 3, minBY = minBX;
        const int maxBX = P.w - kEdge + 3, maxBY = P.h - kEdge + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) continue; /* reference: no cell is visited, level stays empty */
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        std::vector<orbo_candidate>& all = e->cand[l];
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
                const uint8_t* sub = P.px() + (size_t)y0 * P.step + x0;
                fast9(sub, cw, ch, P.step, e->ini_th, cell);
                if (cell.empty()) fast9(sub, cw, ch, P.step, e->min_th, cell);
                for (const orbo_candidate& c : cell)
                    all.push_back(orbo_candidate{c.x + j * wCell, c.y + i * hCell, c.response});
            }
        }
        std::vector<Key> keys(all.size()), kept;
        for (size_t k = 0; k < all.size(); k++) keys[k] = Key{(float)all[k].x, (float)all[k].y, (float)all[k].response};
        int rc = distribute_octtree(keys, minBX, maxBX, minBY, maxBY, e->per_level[l], e->tie_mode, kept);
        if (rc) return rc;
        const int scaledPatch = (int)(kPatchSize * e->sf[l]);                   /* :837 */
        std::vector<orbo_keypoint>& out = e->level_kp[l];
        for (const Key& k : kept) {
            orbo_keypoint q;
            q.x = k.x + minBX; q.y = k.y + minBY;
            q.size = (float)scaledPatch; q.angle = -1.f; q.response = k.response;
            q.octave = l; q.class_id = -1;
            out.push_back(q);
        }
    }
    for (int l = 0; l < nl; l++)                                                /* :851-852 */
        for (orbo_keypoint& q : e->level_kp[l]) q.angle = ic_angle(e->pyr[l], q.x, q.y, e->umax);
    /* operator() tail (:1059-1104) */
    int total = 0;
    for (int l = 0; l < nl; l++) total += (int)e->level_kp[l].size();
    if (n_out) *n_out = total;
    e->blurred.assign(nl, Plane());
    if ((kp_out || desc_out) && total > cap) return -2;
    int offset = 0;
    for (int l = 0; l < nl; l++) {
        std::vector<orbo_keypoint>& kps = e->level_kp[l];
        if (kps.empty()) continue;
        const Plane& P = e->pyr[l];
        Plane& B = e->blurred[l];
        B.create(P.w, P.h, 0);
        gaussian_blur7(P.px(), P.w, P.h, P.step, B.px(), B.step, e->blur_variant);
        for (size_t k = 0; k < kps.size(); k++) {
            if (desc_out) orb_descriptor(kps[k], B, e->fma_mode, desc_out + (size_t)(offset + k) * 32);
            if (kp_out) {
                orbo_keypoint q = kps[k];
                if (l != 0) { float s = e->sf[l]; q.x *= s; q.y *= s; }
                kp_out[offset + k] = q;
            }
        }
        offset += (int)kps.size();
    }
    return 0;
}

int orbo_levels(const orbo_extractor* e) { return e->nlevels; }
float orbo_scale_factor(const orbo_extractor* e) { return (float)e->scale_factor; }
void orbo_table(const orbo_extractor* e, int which, float* out) {
    const std::vector<float>& t = which == 0 ? e->sf : which == 1 ? e->inv_sf : which == 2 ? e->sigma2 : e->inv_sigma2;
    std::memcpy(out, t.data(), t.size() * sizeof(float));
}
void orbo_features_per_level(const orbo_extractor* e, int* out) { std::memcpy(out, e->per_level.data(), e->per_level.size() * sizeof(int)); }
void orbo_umax(const orbo_extractor* e, int* out16) { std::memcpy(out16, e->umax.data(), 16 * sizeof(int)); }

int orbo_level_size(const orbo_extractor* e, int level, int* w, int* h) {
    if (level < 0 || level >= (int)e->pyr.size()) return -1;
    *w = e->pyr[level].w; *h = e->pyr[level].h;
    return 0;
}
int orbo_level_copy(const orbo_extractor* e, int level, uint8_t* dst, size_t dst_stride, int with_border) {
    if (level < 0 || level >= (int)e->pyr.size()) return -1;
    const Plane& P = e->pyr[level];
    if (with_border) {
        for (int y = 0; y < P.h + 2 * kEdge; y++) std::memcpy(dst + (size_t)y * dst_stride, P.buf.data() + (size_t)y * P.step, P.step);
    } else {
        for (int y = 0; y < P.h; y++) std::memcpy(dst + (size_t)y * dst_stride, P.px() + (size_t)y * P.step, P.w);
    }
    return 0;
}
int orbo_blurred_copy(const orbo_extractor* e, int level, uint8_t* dst, size_t dst_stride) {
    if (level < 0 || level >= (int)e->blurred.size()) return -1;
    const Plane& B = e->blurred[level];
    if (B.w == 0) return 0;
    for (int y = 0; y < B.h; y++) std::memcpy(dst + (size_t)y * dst_stride, B.px() + (size_t)y * B.step, B.w);
    return 1;
}
int orbo_level_candidates(const orbo_extractor* e, int level, orbo_candidate* out, int cap) {
    if (level < 0 || level >= (int)e->cand.size()) return -1;
    int n = (int)e->cand[level].size();
    if (out) std::memcpy(out, e->cand[level].data(), sizeof(orbo_candidate) * (size_t)std::min(n, cap));
    return n;
}
int orbo_level_keypoints(const orbo_extractor* e, int level, orbo_keypoint* out, int cap) {
    if (level < 0 || level >= (int)e->level_kp.size()) return -1;
    int n = (int)e->level_kp[level].size();
    if (out) std::memcpy(out, e->level_kp[level].data(), sizeof(orbo_keypoint) * (size_t)std::min(n, cap));
    return n;
}

void orbo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride,
                           uint8_t* dst, int dw, int dh, size_t dstride) {
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orbo_border_reflect101(const uint8_t* src, int w, int h, size_t sstride,
                            uint8_t* dst, size_t dstride, int border) {
    border_reflect101(src, w, h, sstride, dst, dstride, border);
}
int orbo_fast9(const uint8_t* img, int w, int h, size_t stride, int threshold, orbo_candidate* out, int cap) {
    std::vector<orbo_candidate> v;
    fast9(img, w, h, stride, threshold, v);
    if (out) std::memcpy(out, v.data(), sizeof(orbo_candidate) * std::min(v.size(), (size_t)std::max(cap, 0)));
    return (int)v.size();
}
void orbo_gaussian_blur7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride, int blur_variant) {
    gaussian_blur7(src, w, h, sstride, dst, dstride, blur_variant);
}
float orbo_fast_atan2(float y, float x) { return fast_atan2(y, x); }

static Plane plane_from(const uint8_t* img, int w, int h, size_t stride) {
    Plane P;
    P.create(w, h, 0);
    for (int y = 0; y < h; y++) std::memcpy(P.px() + (size_t)y * P.step, img + (size_t)y * stride, (size_t)w);
    return P;
}
float orbo_ic_angle(const uint8_t* img, int w, int h, size_t stride, int x, int y) {
    orbo_extractor* e = orbo_create(100, 1.2f, 1, 20, 7, 0, 0, 0);
    Plane P = plane_from(img, w, h, stride);
    float a = ic_angle(P, (float)x, (float)y, e->umax);
    orbo_destroy(e);
    return a;
}
void orbo_descriptor(const uint8_t* img, int w, int h, size_t stride, int x, int y, float angle_deg,
                     int fma_mode, uint8_t* out32) {
    Plane P = plane_from(img, w, h, stride);
    orbo_keypoint kp{(float)x, (float)y, 31.f, angle_deg, 0.f, 0, -1};
    orb_descriptor(kp, P, fma_mode, out32);
}

int orbo_distribute(const orbo_candidate* cand, int ncand, int min_x, int max_x, int min_y, int max_y,
                    int n_wanted, int tie_mode, orbo_candidate* out, int cap) {
    std::vector<Key> keys(ncand), kept;
    for (int i = 0; i < ncand; i++) keys[i] = Key{(float)cand[i].x, (float)cand[i].y, (float)cand[i].response};
    int rc = distribute_octtree(keys, min_x, max_x, min_y, max_y, n_wanted, tie_mode, kept);
    if (rc) return rc;
    for (size_t i = 0; i < kept.size() && (int)i < cap; i++)
        out[i] = orbo_candidate{(int)kept[i].x, (int)kept[i].y, (int)kept[i].response};
    return (int)kept.size();
}

double orbo_extract_batch_mt(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                             int min_th_fast, int blur_variant, const uint8_t* imgs, int nframes,
                             int w, int h, size_t row_stride, size_t frame_stride, int nthreads,
                             int* counts) {
    if (nthreads < 1) nthreads = 1;
    const int cap = nfeatures + 4 * nlevels + 64;
    std::vector<orbo_extractor*> ex(nthreads);
    for (auto& p : ex) p = orbo_create(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, blur_variant,
                                       ORBO_TIE_CREATION_ORDER, ORBO_FMA_OFF);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) {
        pool.emplace_back([&, t]() {
            std::vector<orbo_keypoint> kp(cap);
            std::vector<uint8_t> desc((size_t)cap * 32);
            for (int f = t; f < nframes; f += nthreads) {
                int n = 0;
                orbo_extract(ex[t], imgs + (size_t)f * frame_stride, w, h, row_stride, kp.data(), cap, &n, desc.data());
                if (counts) counts[f] = n;
            }
        });
    }
    for (auto& th : pool) th.join();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto p : ex) orbo_destroy(p);
    return dt;
}

/* Frame::ComputeStereoMatches (reference src/Frame.cc:481-653) on the keypoints, descriptors and
 * pyramids of a left and a right extractor (mvImagePyramid of both, src/Frame.cc:488,578,590,595).
 * Hamming distance: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1611-1627); thresholds
 * TH_HIGH = 100, TH_LOW = 50 (src/ORBmatcher.cc:38-39). uright / depth receive mvuRight / mvDepth
 * (-1 = no match). Returns the number of matches that survive the median test.
 * Where the reference would index out of bounds (a row band beyond the image, an empty match
 * list before the median) this restatement skips instead. */
int orbo_stereo_match(const orbo_extractor* L, const orbo_extractor* R,
                      const orbo_keypoint* kl, const uint8_t* dl, int nl,
                      const orbo_keypoint* kr, const uint8_t* dr, int nr,
                      float mb, float mbf, float* uright, float* depth) {
    if (!L || !R || L->pyr.empty() || R->pyr.empty()) return -1;
    for (int i = 0; i < nl; i++) { uright[i] = -1.0f; depth[i] = -1.0f; }                 /* :483-484 */
    const int thOrbDist = (100 + 50) / 2;                                                  /* :486 */
    const int nRows = L->pyr[0].h;                                                         /* :488 */
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < nr; iR++) {                                                      /* :498-508 */
        const float kpY = kr[iR].y;
        const float r = 2.0f * L->sf[kr[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back((size_t)iR);
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;                                    /* :511-513 */
    std::vector<std::pair<int, int>> vDistIdx;
    auto hamming = [](const uint8_t* a, const uint8_t* b) {
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t x, y;
            std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
            dist += __builtin_popcount(x ^ y);
        }
        return dist;
    };
    for (int iL = 0; iL < nl; iL++) {                                                      /* :519-639 */
        const orbo_keypoint& kpL = kl[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const size_t row = (size_t)vL;
        if (row >= (size_t)nRows) continue;
        const std::vector<size_t>& vCandidates = vRowIndices[row];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;
        size_t bestIdxR = 0;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {                               /* :543-564 */
            const size_t iR = vCandidates[iC];
            const orbo_keypoint& kpR = kr[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = hamming(dl + (size_t)iL * 32, dr + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {                                                        /* :567-638 */
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = L->inv_sf[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            const Plane& PL = L->pyr[kpL.octave];
            const Plane& PR = R->pyr[kpL.octave];
            const int yl0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w);
            float IL[11][11];
            for (int i = 0; i < 11; i++)
                for (int j = 0; j < 11; j++) IL[i][j] = (float)PL.px()[(ptrdiff_t)(yl0 + i) * (ptrdiff_t)PL.step + xl0 + j];
            const float cL = IL[w][w];
            for (int i = 0; i < 11; i++) for (int j = 0; j < 11; j++) IL[i][j] = IL[i][j] - cL;
            int bestDistS = 2147483647;
            int bestincR = 0;
            const int Lw = 5;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + Lw - w;
            const float endu = scaleduR0 + Lw + w + 1;
            if (iniu < 0 || endu >= PR.w) continue;                                        /* :590-591 */
            for (int incR = -Lw; incR <= +Lw; incR++) {                                    /* :593-607 */
                const int xr0 = (int)(scaleduR0 + incR - w);
                float IR[11][11];
                for (int i = 0; i < 11; i++)
                    for (int j = 0; j < 11; j++) IR[i][j] = (float)PR.px()[(ptrdiff_t)(yl0 + i) * (ptrdiff_t)PR.step + xr0 + j];
                const float cR = IR[w][w];
                double acc = 0;
                for (int i = 0; i < 11; i++)
                    for (int j = 0; j < 11; j++) acc += std::fabs((double)(IL[i][j] - (IR[i][j] - cR)));
                const float dist = (float)acc;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[Lw + incR] = dist;
            }
            if (bestincR == -Lw || bestincR == Lw) continue;                               /* :609-610 */
            const float dist1 = vDists[Lw + bestincR - 1];
            const float dist2 = vDists[Lw + bestincR];
            const float dist3 = vDists[Lw + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));  /* :617 */
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = L->sf[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR); /* :623 */
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {                                   /* :627-637 */
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                depth[iL] = mbf / disparity;
                uright[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0;
    std::sort(vDistIdx.begin(), vDistIdx.end());                                           /* :640-652 */
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        uright[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
        kept--;
    }
    return kept;
}

/* cv::cvtColor(src, dst, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) for 8-bit images, the
 * conversion the reference applies before extraction (src/Tracking.cc:166-191, 211-222, 249-262).
 * OpenCV 2.4 .. 3.x fixed point: yuv_shift = 14, R2Y = 4899, G2Y = 9617, B2Y = 1868, rounding term
 * 1 << 13 (RGB2Gray<uchar> in imgproc/src/color.cpp). OpenCV 4 uses 15-bit weights instead; the
 * reference is built against 2.4 / 3.x (CMakeLists.txt:31-37). */
void orbo_cvt_gray(const uint8_t* src, int w, int h, size_t sstride, int channels, int rgb,
                   uint8_t* dst, size_t dstride) {
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += channels) {
            const int r = rgb ? s[0] : s[2], g = s[1], b = rgb ? s[2] : s[0];
            d[x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
        }
    }
}

/* cv::undistortPoints(src, dst, K, distCoeffs, noArray(), K) as the reference calls it
 * (src/Frame.cc:438, 463): OpenCV's cvUndistortPoints (imgproc/src/undistort.cpp, 2.4 .. 3.x) for
 * CV_32FC2 points, K and distCoeffs given as CV_32F and widened to double, R = I, P = K, five
 * iterations. cam = {fx, fy, cx, cy, k1, k2, p1, p2, k3}. */
void orbo_undistort_points(const float* cam, const float* src_xy, int n, float* dst_xy) {
    double k[12] = {0};
    k[0] = cam[4]; k[1] = cam[5]; k[2] = cam[6]; k[3] = cam[7]; k[4] = cam[8];
    const double fx = cam[0], fy = cam[1], ifx = 1. / fx, ify = 1. / fy, cx = cam[2], cy = cam[3];
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};   /* P * R */
    for (int i = 0; i < n; i++) {
        double x = src_xy[2 * i], y = src_xy[2 * i + 1], x0, y0;
        x0 = x = (x - cx) * ifx;
        y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        dst_xy[2 * i] = (float)(xx * ww);
        dst_xy[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::ComputeImageBounds (src/Frame.cc:450-478): bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}. */
void orbo_image_bounds(const float* cam, int width, int height, float* bounds) {
    if (cam[4] != 0.0) {
        float mat[8] = {0.0f, 0.0f, (float)width, 0.0f, 0.0f, (float)height, (float)width, (float)height};
        orbo_undistort_points(cam, mat, 4, mat);
        bounds[0] = std::min(mat[0], mat[4]); bounds[1] = std::max(mat[2], mat[6]);
        bounds[2] = std::min(mat[1], mat[3]); bounds[3] = std::max(mat[5], mat[7]);
    } else {
        bounds[0] = 0.0f; bounds[1] = (float)width; bounds[2] = 0.0f; bounds[3] = (float)height;
    }
}

/* Frame::UndistortKeyPoints (src/Frame.cc:422-449) followed by Frame::AssignFeaturesToGrid (:255-270,
 * PosInGrid :403-411). kp_un receives mvKeysUn; the 64 x 48 lists of mGrid come back concatenated:
 * cell (x, y) is grid_idx[grid_start[x*48+y] .. grid_start[x*48+y+1]). */
void orbo_undistort_grid(const float* cam, const float* bounds, const orbo_keypoint* kp, int n,
                         orbo_keypoint* kp_un, int* grid_start, int* grid_idx) {
    const int COLS = 64, ROWS = 48;
    if (cam[4] == 0.0) {
        for (int i = 0; i < n; i++) kp_un[i] = kp[i];
    } else {
        std::vector<float> mat((size_t)2 * std::max(n, 1));
        for (int i = 0; i < n; i++) { mat[2 * i] = kp[i].x; mat[2 * i + 1] = kp[i].y; }
        orbo_undistort_points(cam, mat.data(), n, mat.data());
        for (int i = 0; i < n; i++) { kp_un[i] = kp[i]; kp_un[i].x = mat[2 * i]; kp_un[i].y = mat[2 * i + 1]; }
    }
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);      /* src/Frame.cc:128-129 */
    const float hInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
    std::vector<std::vector<size_t>> grid((size_t)COLS * ROWS);
    for (int i = 0; i < n; i++) {
        const int posX = (int)std::round((kp_un[i].x - mnMinX) * wInv);
        const int posY = (int)std::round((kp_un[i].y - mnMinY) * hInv);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        grid[(size_t)posX * ROWS + posY].push_back((size_t)i);
    }
    int pos = 0;
    for (int c = 0; c < COLS * ROWS; c++) {
        grid_start[c] = pos;
        for (size_t i : grid[c]) grid_idx[pos++] = (int)i;
    }
    grid_start[COLS * ROWS] = pos;
}

/* Frame::UnprojectStereo (src/Frame.cc:674-687) for n keypoints: kp_un = mvKeysUn, depth = mvDepth, pose_wc = mRwc
 * row-major (9) then mOw (3), cam = {fx, fy, cx, cy} (invfx = 1.0f / fx as the constructors compute it).
 * world[3 i ..] receives mRwc * x3Dc + mOw (the small-matrix product for CV_32F), valid[i] = 1, where
 * the depth is positive; zeros and 0 where the reference returns an empty Mat. */
void orbo_unproject_stereo(const orbo_keypoint* kp_un, const float* depth, int n, const float* pose_wc, const float* cam,
                           float* world, uint8_t* valid) {
    const float cx = cam[2], cy = cam[3];
    const float invfx = 1.0f / cam[0], invfy = 1.0f / cam[1];
    for (int i = 0; i < n; i++) {
        world[3 * i] = world[3 * i + 1] = world[3 * i + 2] = 0.0f; valid[i] = 0;
        const float z = depth[i];
        if (z > 0) {
            const float u = kp_un[i].x, v = kp_un[i].y;
            const float x = (u - cx) * z * invfx;
            const float y = (v - cy) * z * invfy;
            const float c[3] = {x, y, z};
            for (int r = 0; r < 3; r++) {
                const float t = pose_wc[3 * r] * c[0] + pose_wc[3 * r + 1] * c[1] + pose_wc[3 * r + 2] * c[2];
                world[3 * i + r] = (float)((double)t * 1.0 + (double)pose_wc[9 + r] * 1.0);
            }
            valid[i] = 1;
        }
    }
}

/* Frame::ComputeStereoFromRGBD (src/Frame.cc:655-672) behind the conversion of Tracking::GrabImageRGBD
 * (imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) when |factor - 1| > 1e-5 or the map is not CV_32F;
 * cvtScale to float multiplies in float). depth: height rows of `row_stride` bytes, unsigned 16-bit when
 * depth_u16, else float. kp = mvKeys, kp_un = mvKeysUn. */
void orbo_stereo_from_rgbd(const orbo_keypoint* kp, const orbo_keypoint* kp_un, int n, const void* depth, int depth_u16, int width,
                           int height, size_t row_stride, float mDepthMapFactor, float mbf, float* mvuRight, float* mvDepth) {
    const bool convert = (std::fabs(mDepthMapFactor - 1.0f) > 1e-5) || depth_u16;
    for (int i = 0; i < n; i++) {
        mvuRight[i] = -1; mvDepth[i] = -1;
        const float v = kp[i].y, u = kp[i].x;
        const int row = (int)v, col = (int)u;                  /* Mat::at<float>(int, int) called with floats */
        if (row < 0 || row >= height || col < 0 || col >= width) continue;   /* out of bounds in the reference */
        const uint8_t* r = static_cast<const uint8_t*>(depth) + (size_t)row * row_stride;
        float d;
        if (depth_u16) { uint16_t t; std::memcpy(&t, r + 2 * (size_t)col, 2); d = (float)t; }
        else std::memcpy(&d, r + 4 * (size_t)col, 4);
        if (convert) d = d * mDepthMapFactor;
        if (d > 0) {
            mvDepth[i] = d;
            mvuRight[i] = kp_un[i].x - mbf / d;
        }
    }
}

/* ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:402-517) with
 * Frame::GetFeaturesInArea (src/Frame.cc:351-401), Frame::AssignFeaturesToGrid / PosInGrid for F2's
 * grid (src/Frame.cc:255-270, 403-411), ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1611-1627)
 * and ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607). k1 / k2 are mvKeysUn of F1 / F2, bounds =
 * {mnMinX, mnMaxX, mnMinY, mnMaxY}; prev_matched (n1 x/y pairs) is vbPrevMatched, updated in place;
 * matches12 receives vnMatches12. Returns nmatches. */
int orbo_search_for_initialization(const orbo_keypoint* k1, const uint8_t* d1, int n1,
                                   const orbo_keypoint* k2, const uint8_t* d2, int n2, const float* bounds,
                                   float* prev_matched, int windowSize, float mfNNratio, int mbCheckOrientation,
                                   int* matches12) {
    const int COLS = 64, ROWS = 48, HISTO_LENGTH = 30, TH_LOW = 50;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
    const float hInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
    /* F2.mGrid */
    std::vector<std::vector<size_t>> mGrid((size_t)COLS * ROWS);
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k2[i].x - mnMinX) * wInv);
        const int posY = (int)std::round((k2[i].y - mnMinY) * hInv);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        mGrid[(size_t)posX * ROWS + posY].push_back((size_t)i);
    }
    auto GetFeaturesInArea = [&](float x, float y, float r, int minLevel, int maxLevel) {
        std::vector<size_t> vIndices;
        const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * wInv));
        if (nMinCellX >= COLS) return vIndices;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + r) * wInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * hInv));
        if (nMinCellY >= ROWS) return vIndices;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + r) * hInv));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[(size_t)ix * ROWS + iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const orbo_keypoint& kpUn = k2[vCell[j]];
                    if (bCheckLevels) {
                        if (kpUn.octave < minLevel) continue;
                        if (maxLevel >= 0 && kpUn.octave > maxLevel) continue;
                    }
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    };
    auto DescriptorDistance = [](const uint8_t* a, const uint8_t* b) {
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            unsigned int v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    };
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance((size_t)n2, 2147483647), vnMatches21((size_t)n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const orbo_keypoint kp1 = k1[i1];
        const int level1 = kp1.octave;
        if (level1 > 0) continue;
        std::vector<size_t> vIndices2 = GetFeaturesInArea(prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            const int dist = DescriptorDistance(d1 + (size_t)i1 * 32, d2 + i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (mbCheckOrientation) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;                      /* ComputeThreeMaxima */
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) { prev_matched[2 * i1] = k2[matches12[i1]].x; prev_matched[2 * i1 + 1] = k2[matches12[i1]].y; }
    return nmatches;
}

/* bForward / bBackward of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1330-1337): twc = -Rcw^T tcw (a cv::Mat product with a transposed operand:
 * accumulated in double, stored as float), tlc = Rlw twc + tlw (small-matrix path); forward when tlc.z > mb, backward when
 * -tlc.z > mb, neither for a monocular frame. T* = {R row-major (9), t (3)} of mTcw.
 * Returns 0 (neither), 1 (forward) or 2 (backward; the reference tests forward first). */
int orbo_projection_level_mode(const float* Tcw_cur, const float* Tcw_last, float mb, int bMono) {
    float twc[3];
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tcw_cur[3 * k + r] * (double)Tcw_cur[9 + k];
        twc[r] = (float)(-s);
    }
    /* no transposed operand here: the small-matrix path (float products summed in float, the
     * addend added in double), as for Rcw * x3Dw + tcw */
    const float rz = Tcw_last[6] * twc[0] + Tcw_last[7] * twc[1] + Tcw_last[8] * twc[2];
    const float tlc_z = (float)((double)rz + (double)Tcw_last[11]);
    if (bMono) return 0;
    if (tlc_z > mb) return 1;
    if (-tlc_z > mb) return 2;
    return 0;
}

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1324-1445; the matcher of Tracking::TrackWithMotionModel, src/Tracking.cc:1332,1337)
 * on plain arrays. Last frame, n1 keypoints: k_last = mvKeysUn (octave and angle are read),
 * world_last = GetWorldPos() of mvpMapPoints[i] (x, y, z), desc_last = its GetDescriptor(),
 * flags_last bit 0 = "mvpMapPoints[i] exists and mvbOutlier[i] is false", bit 1 = "its
 * Observations() > 0". Current frame, n2 keypoints: k_cur = mvKeysUn, desc_cur = mDescriptors,
 * uright_cur = mvuRight (NULL: all -1, a monocular frame); Tcw = {Rcw row-major, tcw};
 * cam = {fx, fy, cx, cy}; bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}; scale_factors = mvScaleFactors;
 * level_mode as orbo_projection_level_mode returns it. CurrentFrame.mvpMapPoints is all NULL on
 * entry (src/Tracking.cc:1323,1336); match_cur[i2] receives the index i of the last-frame keypoint
 * whose map point was assigned to current keypoint i2, or -1. Returns nmatches.
 * Rcw * x3Dw + tcw is OpenCV's small-matrix product for CV_32F (three float products summed in float,
 * the translation added in double, one rounding to float). The fork's bounding-box bookkeeping
 * (matches_in_box, points_in_box) does not influence the matching and is not restated. */
int orbo_search_by_projection(const orbo_keypoint* k_last, const float* world_last, const uint8_t* flags_last,
                              const uint8_t* desc_last, int n1, const orbo_keypoint* k_cur, const uint8_t* desc_cur,
                              const float* uright_cur, int n2, const float* Tcw, const float* cam, float mbf,
                              const float* bounds, const float* scale_factors, float th, int level_mode,
                              int mbCheckOrientation, int* match_cur) {
    const int COLS = 64, ROWS = 48, HISTO_LENGTH = 30, TH_HIGH = 100;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
    const float hInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    std::vector<std::vector<size_t>> mGrid((size_t)COLS * ROWS);
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k_cur[i].x - mnMinX) * wInv);
        const int posY = (int)std::round((k_cur[i].y - mnMinY) * hInv);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        mGrid[(size_t)posX * ROWS + posY].push_back((size_t)i);
    }
    auto GetFeaturesInArea = [&](float x, float y, float r, int minLevel, int maxLevel) {   /* src/Frame.cc:351-401 */
        std::vector<size_t> vIndices;
        const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * wInv));
        if (nMinCellX >= COLS) return vIndices;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + r) * wInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * hInv));
        if (nMinCellY >= ROWS) return vIndices;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + r) * hInv));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[(size_t)ix * ROWS + iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const orbo_keypoint& kpUn = k_cur[vCell[j]];
                    if (bCheckLevels) {
                        if (kpUn.octave < minLevel) continue;
                        if (maxLevel >= 0 && kpUn.octave > maxLevel) continue;
                    }
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    };
    auto DescriptorDistance = [](const uint8_t* a, const uint8_t* b) {   /* src/ORBmatcher.cc:1611-1627 */
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            unsigned int v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    };
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    /* CurrentFrame.mvpMapPoints: index of the last-frame keypoint whose point sits there, -1 = NULL */
    for (int i = 0; i < n2; i++) match_cur[i] = -1;
    for (int i = 0; i < n1; i++) {
        if (!(flags_last[i] & 1)) continue;                    /* pMP && !mvbOutlier[i] */
        const float* X = world_last + 3 * (size_t)i;
        float x3Dc[3];
        for (int r = 0; r < 3; r++) {
            const float t = Tcw[3 * r] * X[0] + Tcw[3 * r + 1] * X[1] + Tcw[3 * r + 2] * X[2];
            x3Dc[r] = (float)((double)t * 1.0 + (double)Tcw[9 + r] * 1.0);
        }
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (!(u == u) || !(v == v)) continue;                  /* a point in the camera centre: undefined in the reference */
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const int nLastOctave = k_last[i].octave;
        const float radius = th * scale_factors[nLastOctave];
        std::vector<size_t> vIndices2;
        if (level_mode == 1) vIndices2 = GetFeaturesInArea(u, v, radius, nLastOctave, -1);
        else if (level_mode == 2) vIndices2 = GetFeaturesInArea(u, v, radius, 0, nLastOctave);
        else vIndices2 = GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = desc_last + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (match_cur[i2] >= 0 && (flags_last[match_cur[i2]] & 2)) continue;   /* a point with observations sits there */
            const float ur2 = uright_cur ? uright_cur[i2] : -1.0f;
            if (ur2 > 0) {
                const float ur = u - mbf * invzc;
                const float er = std::fabs(ur - ur2);
                if (er > radius) continue;
            }
            const int dist = DescriptorDistance(dMP, desc_cur + 32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= TH_HIGH) {
            match_cur[bestIdx2] = i;
            nmatches++;
            if (mbCheckOrientation) {
                float rot = k_last[i].angle - k_cur[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;                      /* ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607) */
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx2 : rotHist[i]) { match_cur[idx2] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound,
 * const float th, const int ORBdist) (src/ORBmatcher.cc:1447-1564, relocalisation) on plain arrays.
 * Key frame, n1 keypoints: k_kf = pKF->mvKeysUn (angle), world_kf = GetWorldPos() of
 * GetMapPointMatches()[i], dist_range_kf = {mfMinDistance, mfMaxDistance} pairs, flags_kf bit 0 = "the
 * point exists, is not bad and is not in sAlreadyFound", desc_kf = GetDescriptor(). Current frame, n2
 * keypoints: k_cur = mvKeysUn, desc_cur, occupied_cur[i2] = "mvpMapPoints[i2] is not NULL on entry"
 * (NULL: none). scale_factor = mfScaleFactor, nlevels = mnScaleLevels. match_cur[i2] = index of the
 * key-frame keypoint whose point this call assigned to i2, else -1. Returns nmatches.
 * Ow = -Rcw.t() * tcw goes through the general CV_32F product (accumulated in double, stored as
 * float), Rcw * x3Dw + tcw through the small-matrix one, cv::norm of three floats in double. */
int orbo_search_by_projection_keyframe(const orbo_keypoint* k_kf, const float* world_kf, const float* dist_range_kf,
                                       const uint8_t* flags_kf, const uint8_t* desc_kf, int n1, const orbo_keypoint* k_cur,
                                       const uint8_t* desc_cur, const uint8_t* occupied_cur, int n2, const float* Tcw,
                                       const float* cam, const float* bounds, const float* scale_factors, float scale_factor,
                                       int nlevels, float th, int ORBdist, int mbCheckOrientation, int* match_cur) {
    const int COLS = 64, ROWS = 48, HISTO_LENGTH = 30;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
    const float hInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    const float mfLogScaleFactor = std::log(scale_factor);
    std::vector<std::vector<size_t>> mGrid((size_t)COLS * ROWS);
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k_cur[i].x - mnMinX) * wInv);
        const int posY = (int)std::round((k_cur[i].y - mnMinY) * hInv);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        mGrid[(size_t)posX * ROWS + posY].push_back((size_t)i);
    }
    auto GetFeaturesInArea = [&](float x, float y, float r, int minLevel, int maxLevel) {   /* src/Frame.cc:351-401 */
        std::vector<size_t> vIndices;
        const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * wInv));
        if (nMinCellX >= COLS) return vIndices;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + r) * wInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * hInv));
        if (nMinCellY >= ROWS) return vIndices;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + r) * hInv));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[(size_t)ix * ROWS + iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const orbo_keypoint& kpUn = k_cur[vCell[j]];
                    if (bCheckLevels) {
                        if (kpUn.octave < minLevel) continue;
                        if (maxLevel >= 0 && kpUn.octave > maxLevel) continue;
                    }
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    };
    auto DescriptorDistance = [](const uint8_t* a, const uint8_t* b) {   /* src/ORBmatcher.cc:1611-1627 */
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            unsigned int v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    };
    int nmatches = 0;
    float Ow[3];
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tcw[3 * k + r] * (double)Tcw[9 + k];
        Ow[r] = (float)(-s);
    }
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<char> holds((size_t)n2);                       /* CurrentFrame.mvpMapPoints[i2] != NULL */
    for (int i = 0; i < n2; i++) { holds[i] = occupied_cur && occupied_cur[i]; match_cur[i] = -1; }
    for (int i = 0; i < n1; i++) {
        if (!(flags_kf[i] & 1)) continue;
        const float* X = world_kf + 3 * (size_t)i;
        float x3Dc[3];
        for (int r = 0; r < 3; r++) {
            const float t = Tcw[3 * r] * X[0] + Tcw[3 * r + 1] * X[1] + Tcw[3 * r + 2] * X[2];
            x3Dc[r] = (float)((double)t * 1.0 + (double)Tcw[9 + r] * 1.0);
        }
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        const float u = fx * xc * invzc + cx;
        const float v = fy * yc * invzc + cy;
        if (!(u == u) || !(v == v)) continue;                  /* undefined in the reference */
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const float PO[3] = {X[0] - Ow[0], X[1] - Ow[1], X[2] - Ow[2]};
        double nrm = 0;
        for (int k = 0; k < 3; k++) { const double e = PO[k]; nrm += e * e; }
        const float dist3D = (float)std::sqrt(nrm);
        const float maxDistance = 1.2f * dist_range_kf[2 * i + 1];
        const float minDistance = 0.8f * dist_range_kf[2 * i];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float ratio = dist_range_kf[2 * i + 1] / dist3D;  /* MapPoint::PredictScale */
        const float c = std::ceil(std::log(ratio) / mfLogScaleFactor);
        int nPredictedLevel = (c >= -2147483648.0f && c < 2147483648.0f) ? (int)c : (-2147483647 - 1);
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= nlevels) nPredictedLevel = nlevels - 1;
        const float radius = th * scale_factors[nPredictedLevel];
        const std::vector<size_t> vIndices2 = GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = desc_kf + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (holds[i2]) continue;
            const int dist = DescriptorDistance(dMP, desc_cur + 32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= ORBdist) {
            holds[bestIdx2] = 1;
            match_cur[bestIdx2] = i;
            nmatches++;
            if (mbCheckOrientation) {
                float rot = k_kf[i].angle - k_cur[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;                      /* ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607) */
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx2 : rotHist[i]) { match_cur[idx2] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
 * vector<MapPoint*>& vpMatched, int th) (src/ORBmatcher.cc:287-400, loop closing) with
 * KeyFrame::GetFeaturesInArea and IsInImage (src/KeyFrame.cc:569-613) on plain arrays. Candidate map
 * points, n1: world = GetWorldPos(), normal = GetNormal(), dist_range = {mfMinDistance, mfMaxDistance},
 * flags bit 0 = "not bad and not in spAlreadyFound", desc = GetDescriptor(). Key frame, n2 keypoints:
 * k_kf = mvKeysUn, desc_kf, occupied_kf[idx] = "vpMatched[idx] is not NULL on entry" (NULL: none).
 * Scw: 12 floats, the upper 3 x 4 of the similarity as sR row-major (9) then t (3). match_kf[idx] =
 * index of the map point this call stored in vpMatched[idx], else -1. Returns nmatches.
 * Mat arithmetic as OpenCV performs it for CV_32F: row.dot(row) in double, Mat / scalar = every
 * element times the float value of 1 / scalar, Rcw * p + tcw through the small-matrix product,
 * -Rcw.t() * tcw through the general one (double accumulation), norm and dot of 3 elements in double. */
int orbo_search_by_projection_sim3(const float* world, const float* normal, const float* dist_range, const uint8_t* flags,
                                   const uint8_t* desc, int n1, const orbo_keypoint* k_kf, const uint8_t* desc_kf,
                                   const uint8_t* occupied_kf, int n2, const float* Scw, const float* cam, const float* bounds,
                                   const float* scale_factors, float scale_factor, int nlevels, int th, int* match_kf) {
    const int COLS = 64, ROWS = 48, TH_LOW = 50;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
    const float hInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    const float mfLogScaleFactor = std::log(scale_factor);
    /* decompose Scw */
    const float scw = (float)std::sqrt((double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2]);
    const float inv_scw = (float)(1.0 / scw);
    float Rcw[9], tcw[3], Ow[3];
    for (int i = 0; i < 9; i++) Rcw[i] = Scw[i] * inv_scw;
    for (int i = 0; i < 3; i++) tcw[i] = Scw[9 + i] * inv_scw;
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[3 * k + r] * (double)tcw[k];
        Ow[r] = (float)(-s);
    }
    std::vector<std::vector<size_t>> mGrid((size_t)COLS * ROWS);
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k_kf[i].x - mnMinX) * wInv);
        const int posY = (int)std::round((k_kf[i].y - mnMinY) * hInv);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        mGrid[(size_t)posX * ROWS + posY].push_back((size_t)i);
    }
    auto GetFeaturesInArea = [&](float x, float y, float r) {  /* src/KeyFrame.cc:569-608 */
        std::vector<size_t> vIndices;
        const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * wInv));
        if (nMinCellX >= COLS) return vIndices;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + r) * wInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * hInv));
        if (nMinCellY >= ROWS) return vIndices;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + r) * hInv));
        if (nMaxCellY < 0) return vIndices;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[(size_t)ix * ROWS + iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const orbo_keypoint& kpUn = k_kf[vCell[j]];
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    };
    auto DescriptorDistance = [](const uint8_t* a, const uint8_t* b) {   /* src/ORBmatcher.cc:1611-1627 */
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            unsigned int v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    };
    std::vector<char> holds((size_t)n2);                       /* vpMatched[idx] != NULL */
    for (int i = 0; i < n2; i++) { holds[i] = occupied_kf && occupied_kf[i]; match_kf[i] = -1; }
    int nmatches = 0;
    for (int iMP = 0; iMP < n1; iMP++) {
        if (!(flags[iMP] & 1)) continue;
        const float* p3Dw = world + 3 * (size_t)iMP;
        float p3Dc[3];
        for (int r = 0; r < 3; r++) {
            const float t = Rcw[3 * r] * p3Dw[0] + Rcw[3 * r + 1] * p3Dw[1] + Rcw[3 * r + 2] * p3Dw[2];
            p3Dc[r] = (float)((double)t * 1.0 + (double)tcw[r] * 1.0);
        }
        if (p3Dc[2] < 0.0) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz;
        const float y = p3Dc[1] * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!(u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY)) continue;   /* KeyFrame::IsInImage */
        const float maxDistance = 1.2f * dist_range[2 * iMP + 1];
        const float minDistance = 0.8f * dist_range[2 * iMP];
        const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
        double nrm = 0;
        for (int k = 0; k < 3; k++) { const double e = PO[k]; nrm += e * e; }
        const float dist = (float)std::sqrt(nrm);
        if (dist < minDistance || dist > maxDistance) continue;
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * normal[3 * (size_t)iMP + k];
        if (dot < 0.5 * dist) continue;
        const float ratio = dist_range[2 * iMP + 1] / dist;    /* MapPoint::PredictScale */
        const float c = std::ceil(std::log(ratio) / mfLogScaleFactor);
        int nPredictedLevel = (c >= -2147483648.0f && c < 2147483648.0f) ? (int)c : (-2147483647 - 1);
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= nlevels) nPredictedLevel = nlevels - 1;
        const float radius = th * scale_factors[nPredictedLevel];
        const std::vector<size_t> vIndices = GetFeaturesInArea(u, v, radius);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = desc + 32 * (size_t)iMP;
        int bestDist = 256, bestIdx = -1;
        for (size_t idx : vIndices) {
            if (holds[idx]) continue;
            const int kpLevel = k_kf[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int d = DescriptorDistance(dMP, desc_kf + 32 * idx);
            if (d < bestDist) { bestDist = d; bestIdx = (int)idx; }
        }
        if (bestDist <= TH_LOW) {
            holds[bestIdx] = 1;
            match_kf[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:156-285)
 * on plain arrays. node_kf / node_f: the vocabulary node of every feature (0xFFFFFFFF: none), from
 * which the DBoW2 feature vectors (std::map node -> indices in ascending order, as
 * DBoW2::FeatureVector::addFeature builds them while the features are transformed in index order) are
 * rebuilt. k_kf = pKF->mvKeysUn, k_f = F.mvKeys (angles), flags_kf bit 0 = "the key-frame feature has a
 * map point that is not bad". match_f[i2] = index of the key-frame feature whose point
 * vpMapPointMatches[i2] holds at the end, -1 = NULL. Returns nmatches. */
int orbo_search_by_bow(const orbo_keypoint* k_kf, const uint8_t* desc_kf, const uint32_t* node_kf, const uint8_t* flags_kf, int n1,
                       const orbo_keypoint* k_f, const uint8_t* desc_f, const uint32_t* node_f, int n2, float mfNNratio,
                       int mbCheckOrientation, int* match_f) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    std::map<uint32_t, std::vector<unsigned int>> vFeatVecKF, vFeatVecF;
    for (int i = 0; i < n1; i++) if (node_kf[i] != 0xFFFFFFFFu) vFeatVecKF[node_kf[i]].push_back((unsigned)i);
    for (int i = 0; i < n2; i++) if (node_f[i] != 0xFFFFFFFFu) vFeatVecF[node_f[i]].push_back((unsigned)i);
    auto DescriptorDistance = [](const uint8_t* a, const uint8_t* b) {   /* src/ORBmatcher.cc:1611-1627 */
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            unsigned int v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    };
    for (int i = 0; i < n2; i++) match_f[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    auto KFit = vFeatVecKF.begin(), KFend = vFeatVecKF.end();
    auto Fit = vFeatVecF.begin(), Fend = vFeatVecF.end();
    while (KFit != KFend && Fit != Fend) {
        if (KFit->first == Fit->first) {
            const std::vector<unsigned int>& vIndicesKF = KFit->second;
            const std::vector<unsigned int>& vIndicesF = Fit->second;
            for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
                const unsigned int realIdxKF = vIndicesKF[iKF];
                if (!(flags_kf[realIdxKF] & 1)) continue;      /* !pMP or pMP->isBad() */
                const uint8_t* dKF = desc_kf + 32 * (size_t)realIdxKF;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (size_t iF = 0; iF < vIndicesF.size(); iF++) {
                    const unsigned int realIdxF = vIndicesF[iF];
                    if (match_f[realIdxF] >= 0) continue;
                    const int dist = DescriptorDistance(dKF, desc_f + 32 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                        match_f[bestIdxF] = (int)realIdxKF;
                        if (mbCheckOrientation) {
                            float rot = k_kf[realIdxKF].angle - k_f[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            KFit++;
            Fit++;
        } else if (KFit->first < Fit->first) {
            KFit = vFeatVecKF.lower_bound(Fit->first);
        } else {
            Fit = vFeatVecF.lower_bound(KFit->first);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;                      /* ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607) */
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { match_f[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:519-652)
 * on plain arrays, as orbo_search_by_bow: both sides give mvKeysUn, descriptors, node ids and flags (bit 0 =
 * "has a map point that is not bad"). match12[i1] = feature of key frame 2 whose map point
 * vpMatches12[i1] holds at the end, -1 = NULL. Returns nmatches. */
int orbo_search_by_bow_keyframes(const orbo_keypoint* k1, const uint8_t* desc1, const uint32_t* node1, const uint8_t* flags1, int n1,
                                 const orbo_keypoint* k2, const uint8_t* desc2, const uint32_t* node2, const uint8_t* flags2, int n2,
                                 float mfNNratio, int mbCheckOrientation, int* match12) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    std::map<uint32_t, std::vector<unsigned int>> vFeatVec1, vFeatVec2;
    for (int i = 0; i < n1; i++) if (node1[i] != 0xFFFFFFFFu) vFeatVec1[node1[i]].push_back((unsigned)i);
    for (int i = 0; i < n2; i++) if (node2[i] != 0xFFFFFFFFu) vFeatVec2[node2[i]].push_back((unsigned)i);
    auto DescriptorDistance = [](const uint8_t* a, const uint8_t* b) {   /* src/ORBmatcher.cc:1611-1627 */
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            unsigned int v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    };
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<bool> vbMatched2((size_t)n2, false);
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    auto f1it = vFeatVec1.begin(), f1end = vFeatVec1.end();
    auto f2it = vFeatVec2.begin(), f2end = vFeatVec2.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
                const size_t idx1 = f1it->second[i1];
                if (!(flags1[idx1] & 1)) continue;             /* !pMP1 or bad */
                const uint8_t* d1 = desc1 + 32 * idx1;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
                    const size_t idx2 = f2it->second[i2];
                    if (vbMatched2[idx2] || !(flags2[idx2] & 1)) continue;
                    const int dist = DescriptorDistance(d1, desc2 + 32 * idx2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {
                    if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                        match12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (mbCheckOrientation) {
                            float rot = k1[idx1].angle - k2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back((int)idx1);
                        }
                        nmatches++;
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = vFeatVec1.lower_bound(f2it->first);
        } else {
            f2it = vFeatVec2.lower_bound(f1it->first);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;                      /* ComputeThreeMaxima (src/ORBmatcher.cc:1566-1607) */
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { match12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)
 * (src/ORBmatcher.cc:46-127; the matcher of Tracking::SearchLocalPoints) with RadiusByViewingCos
 * (:129-135) on plain arrays. mp[i] holds what the function reads from vpMapPoints[i]: mTrackProjX,
 * mTrackProjY, mTrackProjXR, mTrackViewCos, mnTrackScaleLevel (the fields Frame::isInFrustum,
 * src/Frame.cc:294-350, leaves there) and flags (bit 0: mbTrackInView && !isBad(), bit 1:
 * Observations() > 0); desc_mp its GetDescriptor(). Frame, n2 keypoints: k = mvKeysUn, desc =
 * mDescriptors, uright = mvuRight (NULL: all -1), occupied[idx] = "F.mvpMapPoints[idx] holds a point
 * with Observations() > 0" on entry (NULL: none does). match[idx] receives the index of the map
 * point assigned to F.mvpMapPoints[idx] (-1: none assigned here). Returns nmatches. */
int orbo_search_map_points(const orbo_track_point* mp, const uint8_t* desc_mp, int n1, const orbo_keypoint* k,
                           const uint8_t* desc, const float* uright, const uint8_t* occupied, int n2,
                           const float* bounds, const float* scale_factors, int nlevels, float th, float mfNNratio, int* match) {
    const int COLS = 64, ROWS = 48, TH_HIGH = 100;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / (mnMaxX - mnMinX);
    const float hInv = static_cast<float>(ROWS) / (mnMaxY - mnMinY);
    std::vector<std::vector<size_t>> mGrid((size_t)COLS * ROWS);
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k[i].x - mnMinX) * wInv);
        const int posY = (int)std::round((k[i].y - mnMinY) * hInv);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        mGrid[(size_t)posX * ROWS + posY].push_back((size_t)i);
    }
    auto GetFeaturesInArea = [&](float x, float y, float r, int minLevel, int maxLevel) {   /* src/Frame.cc:351-401 */
        std::vector<size_t> vIndices;
        const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * wInv));
        if (nMinCellX >= COLS) return vIndices;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + r) * wInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * hInv));
        if (nMinCellY >= ROWS) return vIndices;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + r) * hInv));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[(size_t)ix * ROWS + iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const orbo_keypoint& kpUn = k[vCell[j]];
                    if (bCheckLevels) {
                        if (kpUn.octave < minLevel) continue;
                        if (maxLevel >= 0 && kpUn.octave > maxLevel) continue;
                    }
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    };
    auto DescriptorDistance = [](const uint8_t* a, const uint8_t* b) {   /* src/ORBmatcher.cc:1611-1627 */
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            unsigned int v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    };
    auto RadiusByViewingCos = [](float viewCos) -> float { if (viewCos > 0.998) return 2.5; else return 4.0; };
    /* F.mvpMapPoints[idx]: -2 = a point from before the call with observations, -1 = none or one without, >= 0 = map point of this call */
    std::vector<int> holder((size_t)n2);
    for (int i = 0; i < n2; i++) { holder[i] = (occupied && occupied[i]) ? -2 : -1; match[i] = -1; }
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < n1; iMP++) {
        const orbo_track_point& pMP = mp[iMP];
        if (!(pMP.flags & 1)) continue;                        /* !mbTrackInView or isBad() */
        const int nPredictedLevel = pMP.level;
        if (nPredictedLevel < 0 || nPredictedLevel >= nlevels) continue;   /* outside mvScaleFactors: undefined in the reference */
        float r = RadiusByViewingCos(pMP.view_cos);
        if (bFactor) r *= th;
        const std::vector<size_t> vIndices = GetFeaturesInArea(pMP.proj_x, pMP.proj_y, r * scale_factors[nPredictedLevel],
                                                               nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        const uint8_t* MPdescriptor = desc_mp + 32 * (size_t)iMP;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t idx : vIndices) {
            if (holder[idx] == -2 || (holder[idx] >= 0 && (mp[holder[idx]].flags & 2))) continue;
            const float ur = uright ? uright[idx] : -1.0f;
            if (ur > 0) {
                const float er = std::fabs(pMP.proj_xr - ur);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = DescriptorDistance(MPdescriptor, desc + 32 * idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = k[idx].octave;
                bestIdx = (int)idx;
            } else if (dist < bestDist2) {
                bestLevel2 = k[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
            holder[bestIdx] = iMP;
            match[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

/* Frame::isInFrustum(MapPoint* pMP, float viewingCosLimit) (src/Frame.cc:294-350) for n map points, with
 * MapPoint::PredictScale (src/MapPoint.cc:401-417) and Get{Min,Max}DistanceInvariance (:372-382).
 * mp[i]: GetWorldPos(), GetNormal(), mfMinDistance, mfMaxDistance, flags (bit 0: the caller wants it
 * tested, bit 1: Observations() > 0, passed through). pose = mRcw row-major (9), mtcw (3), mOw (3);
 * cam = {fx, fy, cx, cy}; bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}; scale_factor = mfScaleFactor
 * (mfLogScaleFactor = log of it, the float overload), nlevels = mnScaleLevels. track[i] receives
 * mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos, mnTrackScaleLevel and flags bit 0 =
 * mbTrackInView. Returns the number of points in view.
 * cv::Mat arithmetic as OpenCV performs it for CV_32F: mRcw * P + mtcw through the small-matrix gemm
 * (float products summed in float, translation added in double), P - mOw per element in float,
 * cv::norm and Mat::dot of three elements accumulated in double. */
int orbo_is_in_frustum(const orbo_map_point* mp, int n, const float* pose, const float* cam, float mbf,
                       const float* bounds, float scale_factor, int nlevels, float viewingCosLimit, orbo_track_point* track) {
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float mfLogScaleFactor = std::log(scale_factor);
    int nvisible = 0;
    for (int i = 0; i < n; i++) {
        orbo_track_point& t = track[i];
        t.proj_x = t.proj_y = t.proj_xr = t.view_cos = 0.0f; t.level = 0; t.flags = mp[i].flags & 2;
        if (!(mp[i].flags & 1)) continue;
        const float* P = mp[i].pos;
        float Pc[3];
        for (int r = 0; r < 3; r++) {
            const float s0 = pose[3 * r] * P[0] + pose[3 * r + 1] * P[1] + pose[3 * r + 2] * P[2];
            Pc[r] = (float)((double)s0 * 1.0 + (double)pose[9 + r] * 1.0);
        }
        const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
        if (PcZ < 0.0f) continue;
        const float invz = 1.0f / PcZ;
        const float u = fx * PcX * invz + cx;
        const float v = fy * PcY * invz + cy;
        if (!(u == u) || !(v == v)) continue;                  /* a point in the camera centre: undefined in the reference */
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const float maxDistance = 1.2f * mp[i].max_distance;
        const float minDistance = 0.8f * mp[i].min_distance;
        const float PO[3] = {P[0] - pose[12], P[1] - pose[13], P[2] - pose[14]};
        double nrm = 0;
        for (int k = 0; k < 3; k++) { const double e = PO[k]; nrm += e * e; }
        const float dist = (float)std::sqrt(nrm);
        if (dist < minDistance || dist > maxDistance) continue;
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * mp[i].normal[k];
        const float viewCos = (float)(dot / dist);
        if (viewCos < viewingCosLimit) continue;
        /* MapPoint::PredictScale */
        const float ratio = mp[i].max_distance / dist;
        const float c = std::ceil(std::log(ratio) / mfLogScaleFactor);
        int nScale;
        if (c >= -2147483648.0f && c < 2147483648.0f) nScale = (int)c;
        else nScale = -2147483647 - 1;                         /* what the x86 conversion returns out of range */
        if (nScale < 0) nScale = 0;
        else if (nScale >= nlevels) nScale = nlevels - 1;
        t.flags |= 1;
        t.proj_x = u;
        t.proj_xr = u - mbf * invz;
        t.proj_y = v;
        t.level = nScale;
        t.view_cos = viewCos;
        nvisible++;
    }
    return nvisible;
}

/* cosf / sinf of the host C library, as the reference's computeOrbDescriptor calls them
 * (src/ORBextractor.cc:113-114), for the floats whose bit patterns are first_bits .. first_bits+count-1. */
void orbo_libm_sincos(uint32_t first_bits, uint32_t count, float* out_cos, float* out_sin) {
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t u = first_bits + i;
        float x;
        std::memcpy(&x, &u, sizeof x);
        volatile float arg = x;                                /* no compile-time folding, no sincosf pairing */
        out_cos[i] = cosf(arg);
        out_sin[i] = sinf(arg);
    }
}

/* logf of the host C library (what log() of a float resolves to in src/MapPoint.cc:409) for the floats
 * whose bit patterns are first_bits + i * stride. */
void orbo_libm_logf(uint32_t first_bits, uint32_t count, uint32_t stride, float* out) {
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t u = first_bits + i * stride;
        float x;
        std::memcpy(&x, &u, sizeof x);
        volatile float arg = x;
        out[i] = std::log(arg);
    }
}

/* cv::remap(src, dst, map1, map2, INTER_LINEAR, BORDER_CONSTANT, Scalar()) for an 8-bit single-channel
 * image and CV_32FC1 maps (the rectification of reference Examples/Stereo/stereo_euroc.cc:136-137),
 * OpenCV 2.4 .. 3.x imgproc/src/imgwarp.cpp: the maps become fixed point with INTER_BITS = 5
 * (cvRound(map * 32): integer part saturated to short, 5-bit fraction index into the table), the
 * table BilinearTab_i is built by initInterTab2D(INTER_LINEAR, true) exactly as below (15-bit weights,
 * saturate_cast<short>, the sum forced to 32768 by the fix-up loop, which for the 2x2 kernel also
 * looks at the following, still zero, table slots), remapBilinear applies it with
 * FixedPtCast<int, uchar, 15> and treats taps outside the source as 0. dst has the maps' size. */
void orbo_remap_linear(const uint8_t* src, int sw, int sh, size_t sstride, const float* map_x, const float* map_y,
                       int dw, int dh, uint8_t* dst, size_t dstride) {
    const int INTER_BITS = 5, INTER_TAB_SIZE = 1 << INTER_BITS, INTER_TAB_SIZE2 = INTER_TAB_SIZE * INTER_TAB_SIZE;
    const int INTER_REMAP_COEF_BITS = 15, INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS;
    const int ksize = 2;
    /* static, zero-initialised in OpenCV; a few slots of slack for the fix-up's look-ahead */
    std::vector<short> tabi((size_t)INTER_TAB_SIZE2 * ksize * ksize + 16, 0);
    std::vector<float> tab1((size_t)8 * INTER_TAB_SIZE);
    const float scale = 1.f / INTER_TAB_SIZE;
    for (int i = 0; i < INTER_TAB_SIZE; i++) {                 /* initInterTab1D, interpolateLinear */
        const float x = i * scale;
        tab1[i * ksize] = 1.f - x;
        tab1[i * ksize + 1] = x;
    }
    auto sat_short = [](float v) {
        int iv = (int)std::lrint(v);
        return (short)std::min(std::max(iv, -32768), 32767);
    };
    short* itab = tabi.data();
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++, itab += ksize * ksize) {
            int isum = 0;
            for (int k1 = 0; k1 < ksize; k1++) {
                const float vy = tab1[i * ksize + k1];
                for (int k2 = 0; k2 < ksize; k2++) {
                    const float v = vy * tab1[j * ksize + k2];
                    isum += itab[k1 * ksize + k2] = sat_short(v * INTER_REMAP_COEF_SCALE);
                }
            }
            if (isum != INTER_REMAP_COEF_SCALE) {
                const int diff = isum - INTER_REMAP_COEF_SCALE;
                const int ksize2 = ksize / 2;
                int Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2 + 2; k1++)
                    for (int k2 = ksize2; k2 < ksize2 + 2; k2++) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    const short* wtab = tabi.data();
    const unsigned width1 = (unsigned)std::max(sw - 1, 0), height1 = (unsigned)std::max(sh - 1, 0);
    for (int dy = 0; dy < dh; dy++) {
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            const int isx = (int)std::lrint(map_x[(size_t)dy * dw + dx] * INTER_TAB_SIZE);   /* cvRound */
            const int isy = (int)std::lrint(map_y[(size_t)dy * dw + dx] * INTER_TAB_SIZE);
            const int v = (isy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (isx & (INTER_TAB_SIZE - 1));
            const int sx = std::min(std::max(isx >> INTER_BITS, -32768), 32767);
            const int sy = std::min(std::max(isy >> INTER_BITS, -32768), 32767);
            const short* w = wtab + v * 4;
            int sum;
            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                const uint8_t* S = src + (size_t)sy * sstride + sx;
                sum = S[0] * w[0] + S[1] * w[1] + S[sstride] * w[2] + S[sstride + 1] * w[3];
            } else {
                if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) { D[dx] = 0; continue; }   /* BORDER_CONSTANT, value 0 */
                const int sx0 = sx, sx1 = sx + 1, sy0 = sy, sy1 = sy + 1;
                const int v0 = (sx0 >= 0 && sx0 < sw && sy0 >= 0 && sy0 < sh) ? src[(size_t)sy0 * sstride + sx0] : 0;
                const int v1 = (sx1 >= 0 && sx1 < sw && sy0 >= 0 && sy0 < sh) ? src[(size_t)sy0 * sstride + sx1] : 0;
                const int v2 = (sx0 >= 0 && sx0 < sw && sy1 >= 0 && sy1 < sh) ? src[(size_t)sy1 * sstride + sx0] : 0;
                const int v3 = (sx1 >= 0 && sx1 < sw && sy1 >= 0 && sy1 < sh) ? src[(size_t)sy1 * sstride + sx1] : 0;
                sum = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
            }
            const int r = (sum + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS;
            D[dx] = (uint8_t)std::min(std::max(r, 0), 255);
        }
    }
}

} // extern "C"

// Keypoint undistortion and the 64x48 feature grid of the reference's Frame
// (Frame::UndistortKeyPoints src/Frame.cc:422-449, Frame::AssignFeaturesToGrid :255-270,
// Frame::PosInGrid :403-411, Frame::ComputeImageBounds :450-478), on the keypoints an extraction left
// in device memory.
//
// cv::undistortPoints(pts, pts, K, distCoeffs, noArray(), K) as OpenCV 2.4 .. 3.x computes it: all
// in double, five fixed-point iterations of the inverse distortion, then re-projection with K.
#pragma once
#include <hip/hip_runtime.h>
#include "orb_device.h"

#define GRID_COLS 64          // FRAME_GRID_COLS (include/Frame.h:38)
#define GRID_ROWS 48          // FRAME_GRID_ROWS (include/Frame.h:37)
#define GRID_CELLS (GRID_COLS * GRID_ROWS)

struct CamDev { float fx, fy, cx, cy, k1, k2, p1, p2, k3; };   // mK and mDistCoef, CV_32F in the reference

// One point through cv::undistortPoints with R = I and P = K (pixel coordinates in and out).
__host__ __device__ inline void undistort_point(const CamDev& c, float& px, float& py) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    const double k0 = c.k1, k1 = c.k2, k2 = c.p1, k3 = c.p2, k4 = c.k3;
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = px, y = py;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        // the rational and thin-prism coefficients (k4..k6, s1..s4) are zero: their terms vanish exactly
        const double icdist = 1. / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    px = (float)(fx * x + cx);                                  // P = K, R = I: w = 1
    py = (float)(fy * y + cy);
}

// Cell of an undistorted keypoint (mGrid[x][y] <-> x * GRID_ROWS + y) or -1 outside the grid.
__device__ __forceinline__ int grid_cell(float x, float y, float min_x, float min_y, float inv_w, float inv_h) {
    const int posX = (int)roundf((x - min_x) * inv_w);
    const int posY = (int)roundf((y - min_y) * inv_h);
    if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return -1;
    return posX * GRID_ROWS + posY;
}

struct GridKp { float x, y, size, angle, response; int octave, class_id; };   // cv::KeyPoint record

// One workgroup per frame. kp_un = mvKeysUn; grid_start[f][c] .. grid_start[f][c + 1] is the range of
// cell c in grid_idx[f][], keypoint indices ascending inside a cell (the reference pushes them in
// index order).
__global__ __launch_bounds__(256) void k_undistort_grid(const GridKp* __restrict__ kp, const int* __restrict__ counts, int cap,
                                                        CamDev cam, float min_x, float min_y, float inv_w, float inv_h,
                                                        GridKp* __restrict__ kp_un, int* __restrict__ grid_start,
                                                        int* __restrict__ grid_idx) {
    __shared__ uint32_t s_cur[GRID_CELLS];
    __shared__ uint32_t s_start[GRID_CELLS + 1];
    __shared__ uint32_t s_tmp[4];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int n = min(counts[f], cap);
    const GridKp* K = kp + (size_t)f * cap;
    GridKp* U = kp_un + (size_t)f * cap;
    int* gs = grid_start + (size_t)f * (GRID_CELLS + 1);
    int* gi = grid_idx + (size_t)f * cap;
    for (int c = tid; c < GRID_CELLS; c += 256) s_cur[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        GridKp k = K[i];
        if (cam.k1 != 0.0f) undistort_point(cam, k.x, k.y);      // src/Frame.cc:424-428: only k1 is tested
        U[i] = k;
        const int c = grid_cell(k.x, k.y, min_x, min_y, inv_w, inv_h);
        if (c >= 0) atomicAdd(&s_cur[c], 1u);
    }
    __syncthreads();
    const uint32_t total = block_excl_scan<256>(s_cur, GRID_CELLS, s_tmp);
    for (int c = tid; c < GRID_CELLS; c += 256) { s_start[c] = s_cur[c]; gs[c] = (int)s_cur[c]; }
    if (tid == 0) { s_start[GRID_CELLS] = total; gs[GRID_CELLS] = (int)total; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const GridKp k = U[i];                                  // written by this same thread above
        const int c = grid_cell(k.x, k.y, min_x, min_y, inv_w, inv_h);
        if (c >= 0) gi[atomicAdd(&s_cur[c], 1u)] = i;
    }
    __syncthreads();
    // a cell holds a handful of indices: order them in place
    for (int c = tid; c < GRID_CELLS; c += 256) {
        const int lo = (int)s_start[c], hi = (int)s_start[c + 1];
        for (int a = lo + 1; a < hi; a++) {
            const int v = gi[a];
            int b = a - 1;
            while (b >= lo && gi[b] > v) { gi[b + 1] = gi[b]; b--; }
            gi[b + 1] = v;
        }
    }
}


// Frame::ComputeStereoFromRGBD (reference src/Frame.cc:655-672) with the depth scaling that precedes
// it in Tracking::GrabImageRGBD (src/Tracking.cc: imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor)
// when the factor differs from 1 by more than 1e-5 or the map is not CV_32F): the depth at the
// truncated position of the raw keypoint, mvuRight = undistorted x - mbf / depth where the depth is
// positive. depth_u16: the map holds unsigned 16-bit values, else floats; convert: apply the factor
// (a float multiplication, as cvtScale does for these types).
__global__ __launch_bounds__(256) void k_stereo_from_rgbd(const GridKp* __restrict__ kp, const GridKp* __restrict__ kp_un,
                                                          const int* __restrict__ counts, int cap, const uint8_t* __restrict__ depth,
                                                          size_t row_stride, size_t frame_stride, int width, int height,
                                                          int depth_u16, int convert, float factor, float mbf,
                                                          float* __restrict__ uright, float* __restrict__ out_depth) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const size_t o = (size_t)f * cap + i;
    float ur = -1.0f, dp = -1.0f;
    if (i < min(counts[f], cap)) {
        const int u = (int)kp[o].x, v = (int)kp[o].y;          // Mat::at<float>(float, float): the indices are truncated
        if (u >= 0 && u < width && v >= 0 && v < height) {      // (outside the map the reference reads out of bounds)
            const uint8_t* row = depth + (size_t)f * frame_stride + (size_t)v * row_stride;
            float d = depth_u16 ? (float)reinterpret_cast<const uint16_t*>(row)[u] : reinterpret_cast<const float*>(row)[u];
            if (convert) d = d * factor;
            if (d > 0) { dp = d; ur = kp_un[o].x - __fdiv_rn(mbf, d); }
        }
    }
    uright[o] = ur;
    out_depth[o] = dp;
}


// Frame::UnprojectStereo (reference src/Frame.cc:674-687) for every keypoint of a frame: the world position
// mRwc * ((u - cx) * z * invfx, (v - cy) * z * invfy, z) + mOw of a keypoint with a positive depth
// (OpenCV's small-matrix product: float products summed in float, mOw added in double). pose: per frame
// mRwc row-major (9) and mOw (3). valid[i] = 1 where a point was produced (the reference returns an
// empty Mat otherwise); world[i] is (0, 0, 0) there.
__global__ __launch_bounds__(256) void k_unproject_stereo(const GridKp* __restrict__ kp_un, const float* __restrict__ depth,
                                                          const int* __restrict__ counts, int cap, const float* __restrict__ pose,
                                                          float cx, float cy, float invfx, float invfy,
                                                          float* __restrict__ world, uint8_t* __restrict__ valid) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const size_t o = (size_t)f * cap + i;
    const float* T = pose + 12 * (size_t)f;
    float wx = 0.0f, wy = 0.0f, wz = 0.0f;
    uint8_t ok = 0;
    if (i < min(counts[f], cap)) {
        const float z = depth[o];
        if (z > 0) {
            const float x = (kp_un[o].x - cx) * z * invfx;
            const float y = (kp_un[o].y - cy) * z * invfy;
            wx = (float)((double)(T[0] * x + T[1] * y + T[2] * z) + (double)T[9]);
            wy = (float)((double)(T[3] * x + T[4] * y + T[5] * z) + (double)T[10]);
            wz = (float)((double)(T[6] * x + T[7] * y + T[8] * z) + (double)T[11]);
            ok = 1;
        }
    }
    world[3 * o] = wx; world[3 * o + 1] = wy; world[3 * o + 2] = wz;
    valid[o] = ok;
}

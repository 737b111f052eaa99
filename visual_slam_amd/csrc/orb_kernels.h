// HIP kernels of libORBhip (gfx950). Integer / stencil work: no MFMA on purpose.
//
// Pipeline per pass over F frames (all frames of a pass share one launch, blockIdx.z/y = frame):
//   k_level0     input image -> level 0 plane with REFLECT_101 border     (src/ORBextractor.cc:1127-1128)
//   k_resize     level l-1 -> level l, cv::resize 8U INTER_LINEAR + border (src/ORBextractor.cc:1120-1123)
//   k_fast       FAST-9 score, per-cell NMS, survivors into per-wave segments (orb_fast.h; src/ORBextractor.cc:789-829)
//   k_quadtree   DistributeOctTree, one workgroup per (level, frame)       (src/ORBextractor.cc:539-763)
//   k_describe   IC_Angle + 7x7 blur of the 45x45 patch + rBRIEF, one wave per keypoint
//                                                                          (src/ORBextractor.cc:77-147,1085-1101)
// Compiled with -ffp-contract=off: every float expression rounds exactly as written.
#pragma once
#include <hip/hip_runtime.h>
#include "orb_device.h"
#include "orb_fast.h"

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// ------------------------------------------------------------------------------------------------
// Level 0: copy of the input with the 19-px REFLECT_101 border. One thread writes 16 bytes of one
// row; groups that lie inside the image and are 16-byte aligned in the source move as one dwordx4.
// ------------------------------------------------------------------------------------------------
// k_level0_interior: groups [g_lo, g_hi) of every row lie inside the image and the source is 16-byte
// aligned. k_level0_edges: the remaining groups of a row (nb of them: those below skip_lo and from
// skip_hi on), one thread per (row, group), reflecting byte path.
#define L0_RPT 4                  // rows per thread of the interior copy

// The 16 bytes of group c16 of plane row r through the reflection, byte by byte.
__device__ __forceinline__ void level0_edge_group(const uint8_t* __restrict__ src_frame, size_t row_stride, uint8_t* __restrict__ plane,
                                                  const LevelGeom& g, int r, int c16) {
    const uint8_t* S = src_frame + (size_t)reflect101(r - ORB_OY, g.h) * row_stride;
    const int x0 = c16 * 16 - ORB_OX;
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        o[d] = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) o[d] |= (uint32_t)S[reflect101(x0 + 4 * d + b, g.w)] << (8 * b);
    }
    reinterpret_cast<uint4*>(plane + (size_t)r * g.pitch)[c16] = make_uint4(o[0], o[1], o[2], o[3]);
}

// Grid x: the blocks of 64 interior groups, then one more block that does the nb border groups of
// the block's 4 * L0_RPT rows (their latency-bound byte path runs beside the streaming copy).
// ALIGNED: every 16-byte source group is 16-byte aligned (base and strides are multiples of 16). Otherwise a
// group starts at any byte: it is read as five dwords from the 4-byte aligned address below it (four
// when it happens to be aligned) and shifted into place; the shift is the same for a whole row.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_level0_interior(const uint8_t* __restrict__ src, size_t row_stride,
                                                         size_t frame_stride, uint8_t* __restrict__ pyr,
                                                         LevelGeom g, int g_lo, int g_hi, int nb) {
    const int f = blockIdx.z;
    uint8_t* D = pyr + g.pyr_off + (size_t)f * g.plane_bytes;
    if (blockIdx.x == gridDim.x - 1) {
        const int rb = blockIdx.y * 4 * L0_RPT;
        for (int i = threadIdx.y * 64 + threadIdx.x; i < 4 * L0_RPT * nb; i += 256) {
            const int rr = i / nb;
            int c16 = i - rr * nb;
            if (c16 >= g_lo) c16 += g_hi - g_lo;
            if (rb + rr < g.rows) level0_edge_group(src + (size_t)f * frame_stride, row_stride, D, g, rb + rr, c16);
        }
        return;
    }
    const int c16 = g_lo + blockIdx.x * 64 + threadIdx.x;
    const int r0 = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * L0_RPT;
    if (c16 >= g_hi || r0 >= g.rows) return;
    const uint8_t* S = src + (size_t)f * frame_stride + (c16 * 16 - ORB_OX);
    typedef unsigned int nv4 __attribute__((ext_vector_type(4)));
    nv4 v[L0_RPT];
#pragma unroll
    for (int k = 0; k < L0_RPT; k++) {                                  // all loads first: a wave is short
        int sy = min(r0 + k, g.rows - 1) - ORB_OY;
        sy = sy < 0 ? -sy : (sy >= g.h ? 2 * (g.h - 1) - sy : sy);     // the border is narrower than the image
        const uint8_t* p = S + (size_t)sy * row_stride;
        if (ALIGNED) {
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const nv4*>(p));   // read once
        } else {
            const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3);   // wave-uniform: lanes differ by multiples of 16
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p - sh);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
            const uint32_t d4 = sh ? q[4] : 0u;                 // never past the row's last byte when the group is aligned
            v[k] = nv4{__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                       __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh)};
        }
    }
#pragma unroll
    for (int k = 0; k < L0_RPT; k++)
        if (r0 + k < g.rows) reinterpret_cast<nv4*>(D + (size_t)(r0 + k) * g.pitch)[c16] = v[k];
}

__global__ __launch_bounds__(256) void k_level0_edges(const uint8_t* __restrict__ src, size_t row_stride,
                                                      size_t frame_stride, uint8_t* __restrict__ pyr,
                                                      LevelGeom g, int nb, int skip_lo, int skip_hi) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    const int r = i / nb;
    if (r >= g.rows) return;
    int c16 = i - r * nb;
    if (c16 >= skip_lo) c16 += skip_hi - skip_lo;
    level0_edge_group(src + (size_t)f * frame_stride, row_stride, pyr + g.pyr_off + (size_t)f * g.plane_bytes, g, r, c16);
}

// Level 0 from an interleaved 3- or 4-channel 8-bit image: the cvtColor(..., CV_RGB2GRAY / CV_BGR2GRAY /
// CV_RGBA2GRAY / CV_BGRA2GRAY) that precedes extraction in the reference (src/Tracking.cc:166-191,
// 211-222, 249-262) fused into the border copy. 8-bit arithmetic of OpenCV 2.4 .. 3.x:
// gray = (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14. One thread per (row, 16-pixel group); groups
// inside the image with a 16-byte aligned source read CN dwordx4, the others go pixel by pixel
// through the reflection.
__device__ __forceinline__ uint32_t gray14(uint32_t c0, uint32_t c1, uint32_t c2, bool rgb) {
    const uint32_t r = rgb ? c0 : c2, b = rgb ? c2 : c0;
    return (b * 1868u + c1 * 9617u + r * 4899u + 8192u) >> 14;
}

template <int CN>
__global__ __launch_bounds__(256) void k_level0_color(const uint8_t* __restrict__ src, size_t row_stride, size_t frame_stride,
                                                      uint8_t* __restrict__ pyr, LevelGeom g, int n16, int rgb, int aligned16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    const int r = i / n16;
    if (r >= g.rows) return;
    const int c16 = i - r * n16;
    const uint8_t* S = src + (size_t)f * frame_stride + (size_t)reflect101(r - ORB_OY, g.h) * row_stride;
    const int x0 = c16 * 16 - ORB_OX;
    uint32_t o[4] = {0, 0, 0, 0};
    if (aligned16 && x0 >= 0 && x0 + 16 <= g.w) {
        uint32_t w[4 * CN];
#pragma unroll
        for (int k = 0; k < CN; k++) {
            const uint4 v = reinterpret_cast<const uint4*>(S + (size_t)x0 * CN)[k];
            w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int d = 0; d < 16; d++) {
            uint32_t c[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { const int byte = d * CN + k; c[k] = (w[byte >> 2] >> (8 * (byte & 3))) & 0xFF; }
            o[d >> 2] |= gray14(c[0], c[1], c[2], rgb != 0) << (8 * (d & 3));
        }
    } else {
#pragma unroll
        for (int d = 0; d < 16; d++) {
            const uint8_t* p = S + (size_t)reflect101(x0 + d, g.w) * CN;
            o[d >> 2] |= gray14(p[0], p[1], p[2], rgb != 0) << (8 * (d & 3));
        }
    }
    uint8_t* D = pyr + g.pyr_off + (size_t)f * g.plane_bytes + (size_t)r * g.pitch;
    reinterpret_cast<uint4*>(D)[c16] = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------
// Level 0 through a rectification: cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) of the stereo
// mains (reference Examples/Stereo/stereo_euroc.cc:136-137, maps from cv::initUndistortRectifyMap(...,
// CV_32F, M1, M2) at :97-98) fused into the level-0 kernel, border included (a border pixel is the
// rectified value at the reflected coordinate). OpenCV's arithmetic for CV_32FC1 maps and 8-bit
// images: position = cvRound(map * 32) (5 fractional bits), the four bilinear weights
// (32 - fy | fy) * (32 - fx | fx) * 32 (their table of 15-bit weights; its (0,0) entry
// {32767, 0, 0, 1} gives the same result as {32768, 0, 0, 0} on 8-bit data), result
// (sum + 2^14) >> 15, taps outside the image count as 0 (BORDER_CONSTANT, value 0).
// A thread owns 4 adjacent plane pixels; their source offsets and weights are computed once and
// applied to RM_FR frames, so the maps are read once per RM_FR frames. A tap pair of a row is one
// unaligned 16-bit load. A pair that straddles the image edge is moved inside and its weights are
// moved or cleared accordingly.
// ------------------------------------------------------------------------------------------------
#define RM_FR 8

struct __attribute__((packed, aligned(4))) Row12 { uint32_t a, b, c; };

typedef unsigned short v2us_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_level0_remap(const uint8_t* __restrict__ src, size_t row_stride, size_t frame_stride,
                                                      int frames, const float* __restrict__ map_x,
                                                      const float* __restrict__ map_y, uint8_t* __restrict__ pyr,
                                                      LevelGeom g, int ndw, int aligned4) {
    const int c4 = blockIdx.x * 64 + threadIdx.x;
    const int r = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int f0 = blockIdx.z * RM_FR;
    if (c4 >= ndw || r >= g.rows) return;
    const int y = reflect101(r - ORB_OY, g.h);
    uint32_t off[4], wa[4], wb[4];
    int cxs[4], cys[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = min(max(reflect101(4 * c4 + j - ORB_OX, g.w), 0), g.w - 1);   // columns of the row padding: anything inside
        const float mx = map_x[(size_t)y * g.w + x], my = map_y[(size_t)y * g.w + x];
        const int ix = __float2int_rn(mx * 32.0f), iy = __float2int_rn(my * 32.0f);     // cvRound
        const int fx = ix & 31, fy = iy & 31;
        const int sx = min(max(ix >> 5, -32768), 32767), sy = min(max(iy >> 5, -32768), 32767);   // saturate_cast<short>
        // weights of the taps (sx, sy), (sx+1, sy), (sx, sy+1), (sx+1, sy+1); a tap outside the image is 0
        uint32_t w00 = (uint32_t)((32 - fy) * (32 - fx) * 32), w01 = (uint32_t)((32 - fy) * fx * 32);
        uint32_t w10 = (uint32_t)(fy * (32 - fx) * 32), w11 = (uint32_t)(fy * fx * 32);
        const bool x0in = sx >= 0 && sx < g.w, x1in = sx + 1 >= 0 && sx + 1 < g.w;
        const bool y0in = sy >= 0 && sy < g.h, y1in = sy + 1 >= 0 && sy + 1 < g.h;
        if (!x0in) { w00 = 0; w10 = 0; }
        if (!x1in) { w01 = 0; w11 = 0; }
        if (!y0in) { w00 = 0; w01 = 0; }
        if (!y1in) { w10 = 0; w11 = 0; }
        // the loads read columns cx, cx+1 of rows cy, cy+1 with (cx, cy) inside [0, w-2] x [0, h-2]
        const int cx = min(max(sx, 0), g.w - 2), cy = min(max(sy, 0), g.h - 2);
        if (cx == sx + 1) { w00 = w01; w10 = w11; w01 = 0; w11 = 0; }        // sx = -1: its right tap is the left loaded byte
        else if (cx == sx - 1) { w01 = w00; w11 = w10; w00 = 0; w10 = 0; }   // sx = w-1: its left tap is the right loaded byte
        else if (cx != sx) { w00 = w01 = w10 = w11 = 0; }
        if (cy == sy + 1) { w00 = w10; w01 = w11; w10 = 0; w11 = 0; }
        else if (cy == sy - 1) { w10 = w00; w11 = w01; w00 = 0; w01 = 0; }
        else if (cy != sy) { w00 = w01 = w10 = w11 = 0; }
        off[j] = (uint32_t)cy * (uint32_t)row_stride + (uint32_t)cx;        // a frame is far below 4 GB
        cxs[j] = cx; cys[j] = cy;
        wa[j] = w00 | (w01 << 16);
        wb[j] = w10 | (w11 << 16);
    }
    const int nf = min(RM_FR, frames - f0);
    // Rectification maps move neighbours together: normally the taps of the 4 pixels lie within 8
    // columns of 3 consecutive source rows, which three 8-byte loads per frame cover (instead of
    // eight 2-byte ones). Pixel j then reads rows d, d+1 (d = 0 or 1) of the three, which is a
    // weight pair per row with zeros for the row it does not use.
    const int cx0 = min(min(cxs[0], cxs[1]), min(cxs[2], cxs[3])), cx1 = max(max(cxs[0], cxs[1]), max(cxs[2], cxs[3]));
    const int cy0 = min(min(cys[0], cys[1]), min(cys[2], cys[3])), cy1 = max(max(cys[0], cys[1]), max(cys[2], cys[3]));
    const int cxa = cx0 & ~3;                                  // dword-aligned loads: aligned4 says that the caller's rows are
    const bool compact = aligned4 && cx1 - cx0 <= 6 && cy1 - cy0 <= 1 && cxa + 12 <= g.w;
    if (__all(compact)) {
        uint32_t sel[4], w0[4], w1[4], w2[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t o = (uint32_t)(cxs[j] - cx0);      // 0 .. 6: bytes o, o+1 of the 8 loaded ones
            sel[j] = o | (0x0Cu << 8) | ((o + 1) << 16) | (0x0Cu << 24);
            const bool d = cys[j] != cy0;
            w0[j] = d ? 0u : wa[j];
            w1[j] = d ? wa[j] : wb[j];
            w2[j] = d ? wb[j] : 0u;
        }
        const uint32_t o0 = (uint32_t)cy0 * (uint32_t)row_stride + (uint32_t)cxa;
        const uint32_t o2 = (uint32_t)min(cy0 + 2, g.h - 1) * (uint32_t)row_stride + (uint32_t)cxa;   // row cy0+2 is only used when it exists
        const uint32_t sh = (uint32_t)(cx0 & 3);
        // all loads of the thread's frames first, then the arithmetic
#pragma unroll 1
        for (int k0 = 0; k0 < nf; k0 += RM_FR) {
            Row12 r0[RM_FR], r1[RM_FR], r2[RM_FR];
#pragma unroll
            for (int k = 0; k < RM_FR; k++) {
                if (k0 + k < nf) {
                    const uint8_t* S = src + (size_t)(f0 + k0 + k) * frame_stride;
                    r0[k] = *reinterpret_cast<const Row12*>(S + o0);
                    r1[k] = *reinterpret_cast<const Row12*>(S + o0 + (uint32_t)row_stride);
                    r2[k] = *reinterpret_cast<const Row12*>(S + o2);
                }
            }
#pragma unroll
            for (int k = 0; k < RM_FR; k++) {
                if (k0 + k < nf) {
                    uint32_t out = 0;
                    // byte 0 = column cx0
                    const uint32_t a0 = __builtin_amdgcn_alignbyte(r0[k].b, r0[k].a, sh), a1 = __builtin_amdgcn_alignbyte(r0[k].c, r0[k].b, sh);
                    const uint32_t b0 = __builtin_amdgcn_alignbyte(r1[k].b, r1[k].a, sh), b1 = __builtin_amdgcn_alignbyte(r1[k].c, r1[k].b, sh);
                    const uint32_t c0 = __builtin_amdgcn_alignbyte(r2[k].b, r2[k].a, sh), c1 = __builtin_amdgcn_alignbyte(r2[k].c, r2[k].b, sh);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t a = __builtin_amdgcn_perm(a1, a0, sel[j]);
                        const uint32_t b = __builtin_amdgcn_perm(b1, b0, sel[j]);
                        const uint32_t c = __builtin_amdgcn_perm(c1, c0, sel[j]);
                        uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, a), __builtin_bit_cast(v2us_t, w0[j]), 1u << 14, false);
                        v = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, b), __builtin_bit_cast(v2us_t, w1[j]), v, false);
                        v = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, c), __builtin_bit_cast(v2us_t, w2[j]), v, false);
                        out |= (v >> 15) << (8 * j);
                    }
                    uint8_t* D = pyr + g.pyr_off + (size_t)(f0 + k0 + k) * g.plane_bytes;
                    *reinterpret_cast<uint32_t*>(D + (size_t)r * g.pitch + 4 * c4) = out;
                }
            }
        }
        return;
    }
#pragma unroll 1
    for (int k0 = 0; k0 < nf; k0 += RM_FR) {
        uint16_t lo[RM_FR][4], hi[RM_FR][4];
#pragma unroll
        for (int k = 0; k < RM_FR; k++) {
            if (k0 + k < nf) {
                const uint8_t* S = src + (size_t)(f0 + k0 + k) * frame_stride;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    __builtin_memcpy(&lo[k][j], S + off[j], 2);
                    __builtin_memcpy(&hi[k][j], S + off[j] + (uint32_t)row_stride, 2);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RM_FR; k++) {
            if (k0 + k < nf) {
                uint32_t out = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t a = __builtin_amdgcn_perm(0, lo[k][j], 0x0C010C00u);   // byte0 | byte1 << 16
                    const uint32_t b = __builtin_amdgcn_perm(0, hi[k][j], 0x0C010C00u);
                    const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, a), __builtin_bit_cast(v2us_t, wa[j]),
                                       __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, b), __builtin_bit_cast(v2us_t, wb[j]), 1u << 14, false), false) >> 15;
                    out |= v << (8 * j);
                }
                uint8_t* D = pyr + g.pyr_off + (size_t)(f0 + k0 + k) * g.plane_bytes;
                *reinterpret_cast<uint32_t*>(D + (size_t)r * g.pitch + 4 * c4) = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Level l from level l-1: cv::resize(8UC1, INTER_LINEAR) fixed-point arithmetic. The tap tables are
// built on the host exactly as OpenCV builds xofs/ialpha and yofs/ibeta, and are indexed by DEVICE
// column / row. (They also cover the plane's border, where the reflection of copyMakeBorder,
// src/ORBextractor.cc:1120-1121, is folded in; the kernel does not produce the border: no kernel reads it,
// the descriptor kernel reflects the few patches that reach past a level's edge itself, and
// orbhip_pyramid_copy rebuilds it on the host.)
//
// A lane produces 4 adjacent pixels of RS_RPT consecutive rows. The source bytes of the 4 pixels
// lie inside 12 bytes starting at a 4-aligned offset (checked on the host), so one dwordx3 load
// per source row feeds them; v_perm_b32 pulls the two taps of a pixel into 16-bit halves and
// v_dot2_u32_u16 applies the two 11-bit weights. NARROW (every level of a pyramid whose steps
// shrink by less than 2: the host checks that the 4 pixels' taps span at most 8 bytes): two
// v_alignbyte_b32 first bring the leftmost source column to byte 0, so that every tap pair is one
// v_perm_b32 of the same two registers; otherwise each pixel selects its register pair first.
// Consecutive output rows usually share a source row; its horizontal pass is reused. All
// source rows of the RS_RPT output rows of a wave are requested before any arithmetic starts.
// ------------------------------------------------------------------------------------------------
#ifndef RS_RPT
#define RS_RPT 8    // rows per lane: 8 / 10 / 12 / 16 rows measure 1.47 / 1.41 / 1.41 / 1.45 ms per 512 1080p frames when the chain
                    // runs alone (fewer source rows read twice against fewer waves per SIMD: 72 / 88 / 104 / 136 registers);
                    // with 72 registers a wave of this kernel fits a SIMD beside the FAST waves of level 0, which run on a
                    // second stream while the chain works: the two together take 0.2 ms less than one after the other
#endif

typedef unsigned short v2us __attribute__((ext_vector_type(2)));

// src / dst: plane of frame 0 of the source / destination level (different levels never overlap,
// which lets the loads of all rows be issued before the first store).
//
// CLAMP (with NARROW): the source is an image read in place, whose rows end at byte `load_max + 12`
// with nothing readable behind the last one. A lane whose 12-byte window would reach past that loads the
// last 12 bytes of the row instead and shifts by up to 11 bytes.
template <bool NARROW, bool CLAMP>
__device__ __forceinline__ void resize_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                            const LevelGeom& gs, const LevelGeom& gd, const RsTab* __restrict__ tabs, int c4_end,
                                            int load_max) {
    static_assert(NARROW || !CLAMP, "the clamped window is built for the narrow variant");
    // (An XCD-aware tile order cuts the bytes the L2s fetch by a third here but measures 6 % slower,
    // with the vector work of either variant: in plain dispatch order the XCDs stream through the same
    // region together and the cache behind the L2s absorbs the rows they share. Plain order it is.)
    // Only the level itself is produced: plane columns ORB_OX .. ORB_OX + w - 1 (in groups of 4) and rows
    // ORB_OY .. ORB_OY + h - 1. Nothing on the device reads a plane's border columns and rows.
    const int bx = blockIdx.x, by = blockIdx.y, f = blockIdx.z;
    // (block columns are aligned to 256 bytes of the plane's rows; the 8 lanes left of the level in the first one idle)
    const int c4 = bx * 64 + threadIdx.x;
    const int row_end = ORB_OY + gd.h;
    // a wave = one value of threadIdx.y: rows, their tap entries and row addresses are wave-uniform
    const int r0 = ORB_OY + (by * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * RS_RPT;
    if (c4 < ORB_OX / 4 || c4 >= c4_end || r0 >= row_end) return;
    // x side: the same for every row
    const uint4* tx = reinterpret_cast<const uint4*>(tabs + gd.xtab_off + c4 * 4);
    const uint4 ta = tx[0], tb = tx[1];
    const uint32_t pos[4] = {ta.x, ta.z, tb.x, tb.z};          // s0 | s1 << 16
    const uint32_t wgt[4] = {ta.y, ta.w, tb.y, tb.w};          // a0 | a1 << 16
    const int s00 = pos[0] & 0xFFFF, s01 = pos[1] & 0xFFFF, s02 = pos[2] & 0xFFFF, s03 = pos[3] & 0xFFFF;
    const int smin = min(min(s00, s01), min(s02, s03));
    const int base = CLAMP ? min(smin & ~3, load_max) : smin & ~3;
    const uint32_t sh = (uint32_t)(smin - base);               // NARROW: the loaded bytes are shifted so that byte 0 is column smin (0..3; CLAMP: 0..11)
    const bool sh4 = CLAMP && sh >= 4, sh8 = CLAMP && sh >= 8;
    uint32_t sel[4];
    bool hi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int o = (int)(pos[i] & 0xFFFF) - (NARROW ? smin : base);   // NARROW: 0 .. 6, else 0 .. 10
        hi[i] = !NARROW && o >= 4;
        const uint32_t idx = (uint32_t)(hi[i] ? o - 4 : o);     // byte index inside the (chosen) dword pair
        sel[i] = idx | (0x0Cu << 8) | ((idx + 1) << 16) | (0x0Cu << 24);
    }
    // uniform plane pointers + 32-bit (row, lane) offsets: a plane is far below 4 GB
    const uint8_t* S = src + (size_t)f * gs.plane_bytes + (size_t)ORB_OY * gs.pitch + ORB_OX;
    const uint32_t lbase = (uint32_t)base;
    uint8_t* D = dst + (size_t)f * gd.plane_bytes;
    const RsTab* ytab = tabs + gd.ytab_off;
    // y taps of the wave's rows (scalar loads), then every source row it needs, then the arithmetic
    RsTab ty[RS_RPT];
#pragma unroll
    for (int k = 0; k < RS_RPT; k++) ty[k] = ytab[min(r0 + k, row_end - 1)];
    Row12 va[RS_RPT], vb[RS_RPT];
#pragma unroll
    for (int k = 0; k < RS_RPT; k++) {
        // consecutive output rows usually share a source row: the upper row of k is the lower row of k-1
        if (k == 0 || ty[k].s0 != ty[k - 1].s1)
            va[k] = *reinterpret_cast<const Row12*>(S + ((uint32_t)((int)ty[k].s0 * gs.pitch) + lbase));
        vb[k] = *reinterpret_cast<const Row12*>(S + ((uint32_t)((int)ty[k].s1 * gs.pitch) + lbase));
    }
    auto hpass = [&](const Row12& v, uint32_t (&h)[4]) {
        // CLAMP: whole dwords first (the bytes wanted start in dword sh / 4 of the window)
        uint32_t da = v.a, db = v.b, dc = v.c;
        if (CLAMP) {                                           // (two steps of one dword: stays in registers)
            da = sh4 ? db : da; db = sh4 ? dc : db; dc = sh4 ? 0u : dc;
            da = sh8 ? db : da; db = sh8 ? dc : db;
        }
        const uint32_t lo = NARROW ? __builtin_amdgcn_alignbyte(db, da, sh & 3) : 0u;
        const uint32_t up = NARROW ? __builtin_amdgcn_alignbyte(dc, db, sh & 3) : 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t p = NARROW ? __builtin_amdgcn_perm(up, lo, sel[i])
                                      : __builtin_amdgcn_perm(hi[i] ? v.c : v.b, hi[i] ? v.b : v.a, sel[i]);   // tap0 | tap1 << 16
            h[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us, p), __builtin_bit_cast(v2us, wgt[i]), 0u, false) >> 4;
        }
    };
    uint32_t hprev[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < RS_RPT; k++) {
        const int r = r0 + k;
        if (r >= row_end) break;
        uint32_t h0[4], h1[4];
        if (k > 0 && ty[k].s0 == ty[k - 1].s1) {
#pragma unroll
            for (int i = 0; i < 4; i++) h0[i] = hprev[i];
        } else {
            hpass(va[k], h0);
        }
        hpass(vb[k], h1);
        const uint32_t b0 = (uint32_t)ty[k].a0, b1 = (uint32_t)ty[k].a1;   // weights are never negative
        // 11-bit weight x 15-bit row value: the full-rate 24-bit multiply is exact.
        // ((p0 >> 16) + (p1 >> 16) + 2) >> 2 with the 2 added as 2 << 16 before the shift of p0; the sum of
        // the two upper halves is one add with 16-bit operand selects, written into one half of a register
        // per pixel pair, so that the final shift is one packed instruction per pair and the four result
        // bytes (each <= 255 by construction) are gathered by one byte permute
        uint32_t pair[2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            hprev[i] = h1[i];
            const uint32_t p0 = umul24(b0, h0[i]) + 0x20000u, p1 = umul24(b1, h1[i]);
            if (i & 1)
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"
                    : "+v"(pair[i >> 1]) : "v"(p0), "v"(p1));
            else
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1"
                    : "=v"(pair[i >> 1]) : "v"(p0), "v"(p1));
        }
        const v2us two = {2, 2};
        const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2us, pair[0]) >> two);
        const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2us, pair[1]) >> two);
        const uint32_t out = __builtin_amdgcn_perm(q23, q01, 0x06040200u);
        *reinterpret_cast<uint32_t*>(D + ((uint32_t)(r * gd.pitch) + 4u * (uint32_t)c4)) = out;
    }
}

// EDGE: the source is an image read in place (see resize_rows): the block columns from `first_clamp` on
// contain lanes whose window would pass the readable part of a row and run the clamping code, the
// others the plain code (a branch uniform over the workgroup).
template <bool NARROW, bool EDGE>
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                LevelGeom gs, LevelGeom gd, const RsTab* __restrict__ tabs, int c4_end,
                                                int first_clamp, int load_max) {
    static_assert(NARROW || !EDGE, "an image is read in place by the narrow variant only");
    if (EDGE && (int)blockIdx.x >= first_clamp) resize_rows<NARROW, NARROW && EDGE>(src, dst, gs, gd, tabs, c4_end, load_max);
    else resize_rows<NARROW, false>(src, dst, gs, gd, tabs, c4_end, load_max);
}

// ------------------------------------------------------------------------------------------------
// Quadtree distribution. One workgroup per (level, frame); the node list lives in LDS.
//
// The reference walks a std::list and copies key vectors; here every key carries the index of the
// list node that owns it ("label") and one round of the reference's loop is
//   node step : decide which nodes split, in which order (= creation order of their children),
//               build the new list = children in reverse creation order, then the unsplit nodes
//   key step  : relabel every key to its new node and count the keys per child quadrant
// Node counts and the "best response per node" reduction are order-independent, so the result
// does not depend on the order in which the candidates were gathered. Ties between equal
// responses go to the key that comes first in the reference's candidate order (cell row, cell
// column, raster), encoded as `ord`. Equal-size nodes in the final expansion rounds are taken in
// reverse creation order (the reference compares heap addresses there).
// ------------------------------------------------------------------------------------------------
// Threads per workgroup (QT_T): 256 where a level holds thousands of keys (the loops over keys scale with the threads),
// 128 where the lists are short (the node steps, a dozen barriers and three scans per round, cost more with every wave
// that joins them): 1080p / 2000 features 1.13 / 0.86 / 0.90 ms per 1024 frames at 128 / 256 / 512 threads, VGA / 1000
// features 0.20 / 0.24 / 0.40. The host picks by the quota of level 0.
#define QT_FRESH 0x80000000u

struct QtShared {
    uint32_t* box[2];     // [2*M] per buffer: x0 | x1<<16, y0 | y1<<16
    uint32_t* cnt[2];     // [M] key count | QT_FRESH
    uint32_t* cc[2];      // [4*M] child counts; after the node step of a round: child list positions
    uint32_t* surv;       // [M] new position of an unsplit node
    int* kord;            // [M] split order index, or -1
    uint32_t* ordarr;     // [M] node index by split order
    uint32_t* cbase;      // [M] children created before this split node
    // (the 64-bit sort keys of the size-ordered rounds and the best key per node of the last round live in the
    // child-count buffer of the round's new list, which nothing else uses while they are needed)
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) { return wave_scan_incl(v); }   // DPP, no LDS traffic

// In-place exclusive scan of a[0..n) by the whole workgroup; returns the total.
template <int QT_T>
__device__ uint32_t block_excl_scan(uint32_t* a, int n, uint32_t* s_tmp) {
    const int tid = threadIdx.x;
    const int chunk = (n + QT_T - 1) / QT_T;
    const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += a[i];
    const uint32_t inc = wave_incl_scan(s);
    if ((tid & 63) == 63) s_tmp[tid >> 6] = inc;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < QT_T / 64; w++) {
        if (w < (tid >> 6)) off += s_tmp[w];
        total += s_tmp[w];
    }
    uint32_t run = off + inc - s;
    for (int i = lo; i < hi; i++) { const uint32_t v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

__device__ __forceinline__ int qt_quadrant(uint32_t bx, uint32_t by, int x, int y) {
    const int x0 = bx & 0xFFFF, x1 = bx >> 16, y0 = by & 0xFFFF, y1 = by >> 16;
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);   // ceil(size/2)
    return (x < mx ? 0 : 1) | (y < my ? 0 : 2);
}

// The keys of a (frame, level) are three arrays: position x | y << 16 and FAST score, written once when the
// survivors are gathered, and the label word (quadrant | list node << 2, all ones = dropped) that every round
// of the distribution rewrites: 16 bits while the node arrays fit LDS (fewer than 2048 nodes), 32 bits for the
// levels that keep them in global memory. A round then moves 8 bytes per key (position read, label read
// and written), which is what bounds the kernel on large frames.
//
// Runs `body(a, b, k0)` over all keys of a thread (key k0 + u * QT_T: its position in a[u], its
// entry of the second array - label or score - in b[u]; slots past the end hold a dropped key at the
// window origin), QT_KU keys per trip. The keys of the next trip are requested before the body runs:
// the loops over keys are bound by memory latency.
#ifndef QT_KU
#define QT_KU 4   // (8 keys per trip need 104 registers per lane; 4 need 68, and the workgroups that fit beside each other
                  // hide more latency than the longer trips did: VGA-sized frames 30 % faster, 1080p unchanged)
#endif
template <int QT_T, typename S, typename F>
__device__ __forceinline__ void qt_for_keys(const uint32_t* xy, const S* second, uint32_t second_dead, int nkeys, int tid, F&& body) {
    const uint32_t xy_dead = 3u | (3u << 16);
    uint32_t a[QT_KU], b[QT_KU];
#pragma unroll
    for (int u = 0; u < QT_KU; u++) {
        const int k = tid + u * QT_T;
        a[u] = xy_dead; b[u] = second_dead;
        if (k < nkeys) { a[u] = xy[k]; b[u] = second[k]; }
    }
    for (int k0 = tid; k0 < nkeys; k0 += QT_KU * QT_T) {
        uint32_t na[QT_KU], nb[QT_KU];
#pragma unroll
        for (int u = 0; u < QT_KU; u++) {
            const int k = k0 + (QT_KU + u) * QT_T;
            na[u] = xy_dead; nb[u] = second_dead;
            if (k < nkeys) { na[u] = xy[k]; nb[u] = second[k]; }
        }
        body(a, b, k0);
#pragma unroll
        for (int u = 0; u < QT_KU; u++) { a[u] = na[u]; b[u] = nb[u]; }
    }
}

template <bool WIDE> struct QtLabel { typedef uint16_t type; };
template <> struct QtLabel<true> { typedef uint32_t type; };

// SPILL = false: the node arrays live in LDS (every level whose quota fits: 72 bytes per list node).
// SPILL = true: the levels whose node arrays do not fit (quotas beyond ~2000 features on one level; the
// reference has no such limit, src/ORBextractor.cc:539-763) keep them in a global scratch slice per
// (frame, level) instead; the code is the same, the workgroup's barriers order its own global accesses
// (all its waves share one CU and its cache). Each variant returns at once on the other's levels.
template <bool SPILL, int QT_T>
__global__ __launch_bounds__(QT_T) void k_quadtree(const PlanDev* __restrict__ plan,
                                                         const uint2* __restrict__ seg,
                                                         const int* __restrict__ seg_count,
                                                         uint32_t* __restrict__ keys_xy_all, uint8_t* __restrict__ keys_score_all,
                                                         uint32_t* __restrict__ keys_label_all,
                                                         int* __restrict__ key_count,
                                                         LevelKp* __restrict__ lvl_kp,
                                                         int* __restrict__ lvl_cnt,
                                                         int* __restrict__ status,
                                                         uint32_t* __restrict__ spill, unsigned long long spill_stride) {
    extern __shared__ __align__(16) uint32_t smem_lds[];
    __shared__ uint32_t s_tmp[QT_T / 64];
    __shared__ uint32_t s_red[4];
    __shared__ int s_nkeys;
    const int tid = threadIdx.x;
    // Grid (frames, levels): consecutive workgroups, which are dealt to the 8 XCDs in turn, hold the
    // same level of different frames (the levels differ a lot in cost), the large levels first.
    const int f = blockIdx.x, level = blockIdx.y;
    const int slot = plan->qt_spill_slot[level];
    if ((slot >= 0) != SPILL) return;                          // uniform over the workgroup
    uint32_t* const smem = SPILL ? spill + ((size_t)f * plan->qt_nspill + slot) * spill_stride : smem_lds;
    const LevelGeom g = plan->lv[level];
    const int M = g.node_cap;
    const int N = g.quota;
    QtShared q;
    {
        uint32_t* p = smem;
        q.box[0] = p; p += 2 * M; q.box[1] = p; p += 2 * M;
        q.cnt[0] = p; p += M; q.cnt[1] = p; p += M;
        q.cc[0] = p; p += 4 * M; q.cc[1] = p; p += 4 * M;
        q.surv = p; p += M;
        q.kord = reinterpret_cast<int*>(p); p += M;
        q.ordarr = p; p += M;
        q.cbase = p; p += M;
    }
    uint32_t* cellbits = smem + g.qt_words;        // one bit per FAST cell: it has a survivor >= iniTh
    typedef typename QtLabel<SPILL>::type Lab;                 // label word: quadrant | node << 2
    const uint32_t lab_dead = (uint32_t)(Lab)~(Lab)0;
    const size_t kbase = (size_t)f * plan->keys_per_frame + g.keys_off;
    uint32_t* kxy = keys_xy_all + kbase;
    uint8_t* ksc = keys_score_all + kbase;
    Lab* klab = reinterpret_cast<Lab*>(keys_label_all + kbase);   // (a 32-bit slot per key either way)
    const int ini_th = plan->ini_th;

    // ---- gather the FAST survivors of this level from the (tile, wave) segments into the key arrays
    // (any order: everything below is order-independent) and mark the cells that have a survivor
    // at the initial threshold (src/ORBextractor.cc:808-816: a cell is re-run at minThFAST only when
    // it found nothing at iniThFAST)
    const int ncells = g.ncols * g.nrows;
    for (int i = tid; i < (ncells + 31) / 32; i += QT_T) cellbits[i] = 0;
    const int nseg = g.fast_ntiles * 4;
    uint32_t* segbase = smem;                      // [nseg + 1], aliases the node arrays (not in use yet)
    {
        // step 1: segment sizes -> exclusive prefix sums = positions in `keys`
        const int* sc = seg_count + (size_t)f * plan->segc_per_frame + g.segc_off;
        for (int sidx = tid; sidx < nseg; sidx += QT_T) segbase[sidx] = (uint32_t)max(sc[sidx], 0);
        __syncthreads();
        const uint32_t total = block_excl_scan<QT_T>(segbase, nseg, s_tmp);
        if (tid == 0) { segbase[nseg] = total; s_nkeys = (int)total; }
        __syncthreads();
        // step 2: one thread per key position p: binary search for its segment (the last one with
        // segbase <= p; empty segments repeat a value), QT_KU positions per trip side by side so
        // that the LDS reads of the searches and then the global loads are in flight together
        const uint2* sbase = seg + (size_t)f * plan->seg_per_frame + g.seg_off;
        const int ncopy = min((int)total, g.keys_cap);
        const int nsteps = nseg > 1 ? 32 - __clz(nseg - 1) : 0;
        for (int p0 = tid; p0 < ncopy; p0 += QT_KU * QT_T) {
            int lo[QT_KU], hi[QT_KU];
#pragma unroll
            for (int u = 0; u < QT_KU; u++) { lo[u] = 0; hi[u] = nseg; }
            for (int st = 0; st < nsteps; st++) {
#pragma unroll
                for (int u = 0; u < QT_KU; u++) {
                    const int mid = (lo[u] + hi[u]) >> 1;
                    if ((int)segbase[mid] <= min(p0 + u * QT_T, ncopy - 1)) lo[u] = mid; else hi[u] = mid;
                }
            }
            uint2 e[QT_KU];
#pragma unroll
            for (int u = 0; u < QT_KU; u++) {
                const int p = min(p0 + u * QT_T, ncopy - 1);
                e[u] = sbase[(size_t)lo[u] * FT_SEGCAP + (p - (int)segbase[lo[u]])];
            }
#pragma unroll
            for (int u = 0; u < QT_KU; u++) {
                const int p = p0 + u * QT_T;
                if (p >= ncopy) continue;
                kxy[p] = e[u].x;
                ksc[p] = (uint8_t)e[u].y;
                if ((int)(e[u].y & 0xFF) >= ini_th) {
                    const int x = e[u].x & 0xFFFF, y = e[u].x >> 16;
                    const int c = __mul24((int)(umul24((uint32_t)(y - 3), g.hcell_m22) >> 22), g.ncols) +
                                  (int)(umul24((uint32_t)(x - 3), g.wcell_m22) >> 22);
                    atomicOr(&cellbits[c >> 5], 1u << (c & 31));
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_nkeys > g.keys_cap) atomicOr(status, 1);
        key_count[f * plan->nlevels + level] = min(s_nkeys, g.keys_cap);
    }
    const int nkeys = min(s_nkeys, g.keys_cap);

    // ---- roots (src/ORBextractor.cc:543-585)
    int L = g.n_ini;            // length of the node array (round 0 may contain empty roots)
    for (int i = tid; i < M; i += QT_T) {
        if (i < L) {
            const int x0 = (int)(g.hx * (float)i), x1 = (int)(g.hx * (float)(i + 1));
            q.box[0][2 * i] = (uint32_t)x0 | ((uint32_t)x1 << 16);
            q.box[0][2 * i + 1] = (uint32_t)(g.maxby - ORB_MINB) << 16;
        }
        q.cc[0][4 * i] = q.cc[0][4 * i + 1] = q.cc[0][4 * i + 2] = q.cc[0][4 * i + 3] = 0;
    }
    __syncthreads();
    qt_for_keys<QT_T>(kxy, ksc, 0u, nkeys, tid, [&](const uint32_t (&exy)[QT_KU], const uint32_t (&esc)[QT_KU], int k0) __attribute__((always_inline)) {
        int x[QT_KU], y[QT_KU], s[QT_KU], cell[QT_KU], root[QT_KU];
        uint32_t cb[QT_KU], b0[QT_KU], b1[QT_KU];
#pragma unroll
        for (int u = 0; u < QT_KU; u++) {
            x[u] = exy[u] & 0xFFFF; y[u] = exy[u] >> 16; s[u] = (int)esc[u];
            cell[u] = __mul24((int)(umul24((uint32_t)(y[u] - 3), g.hcell_m22) >> 22), g.ncols) + (int)(umul24((uint32_t)(x[u] - 3), g.wcell_m22) >> 22);
            cb[u] = cellbits[cell[u] >> 5];
            root[u] = min(max((int)__fdiv_rn((float)x[u], g.hx), 0), L - 1);
            b0[u] = q.box[0][2 * root[u]]; b1[u] = q.box[0][2 * root[u] + 1];
        }
#pragma unroll
        for (int u = 0; u < QT_KU; u++) {
            const int k = k0 + u * QT_T;
            if (k >= nkeys) continue;
            uint32_t lab = lab_dead;
            if (s[u] >= ini_th || ((cb[u] >> (cell[u] & 31)) & 1) == 0) {
                const uint32_t qd = qt_quadrant(b0[u], b1[u], x[u], y[u]);
                atomicAdd(&q.cc[0][4 * root[u] + qd], 1u);
                lab = qd | ((uint32_t)root[u] << 2);
            }
            klab[k] = (Lab)lab;
        }
    });
    __syncthreads();
    for (int i = tid; i < L; i += QT_T)
        q.cnt[0][i] = (q.cc[0][4 * i] + q.cc[0][4 * i + 1] + q.cc[0][4 * i + 2] + q.cc[0][4 * i + 3]) | QT_FRESH;
    __syncthreads();

    int cur = 0;
    const unsigned long long* best = nullptr;                  // best key per node of the final list
    int mode = 1;               // 1: expand every node with >1 key; 2: expand by size until N
    bool done = false;
    for (int round = 0; round < 64 && !done; round++) {
        const int nxt = cur ^ 1;
        // double-buffered node arrays of this round as plain pointers (indexing the pointer table
        // with a run-time index would put it into scratch memory)
        uint32_t* const cc_cur = cur ? q.cc[1] : q.cc[0];
        uint32_t* const cc_nxt = cur ? q.cc[0] : q.cc[1];
        uint32_t* const box_cur = cur ? q.box[1] : q.box[0];
        uint32_t* const box_nxt = cur ? q.box[0] : q.box[1];
        uint32_t* const cnt_cur = cur ? q.cnt[1] : q.cnt[0];
        uint32_t* const cnt_nxt = cur ? q.cnt[0] : q.cnt[1];
        // 64-bit keys of this round: the new list's child counts are not in use before step D clears them, and
        // not at all in the round that ends the loop
        unsigned long long* const key64 = reinterpret_cast<unsigned long long*>(cc_nxt);
        // ---- A. which nodes split, and in which order
        if (tid < 4) s_red[tid] = 0;
        __syncthreads();
        uint32_t prev_size_l = 0;
        for (int i = tid; i < L; i += QT_T) prev_size_l += (cnt_cur[i] & ~QT_FRESH) > 0;
        if (prev_size_l) atomicAdd(&s_red[0], prev_size_l);
        int nsplit;
        if (mode == 1) {
            for (int i = tid; i < L; i += QT_T) q.ordarr[i] = (cnt_cur[i] & ~QT_FRESH) > 1;
            __syncthreads();
            nsplit = (int)block_excl_scan<QT_T>(q.ordarr, L, s_tmp);
            for (int i = tid; i < L; i += QT_T)
                q.kord[i] = (cnt_cur[i] & ~QT_FRESH) > 1 ? (int)q.ordarr[i] : -1;
            __syncthreads();
        } else {
            // candidates: nodes created in the previous round that hold more than one key, largest
            // first; equal sizes: latest created first (= smaller list position)
            uint32_t ne_l = 0;
            for (int i = tid; i < L; i += QT_T) {
                const uint32_t c = cnt_cur[i];
                const bool e = (c & QT_FRESH) && (c & ~QT_FRESH) > 1;
                key64[i] = e ? (((unsigned long long)(c & ~QT_FRESH) << 32) | (0xFFFFFFFFu - (uint32_t)i)) : 0ull;
                ne_l += e;
            }
            if (ne_l) atomicAdd(&s_red[1], ne_l);
            __syncthreads();
            const int nE = (int)s_red[1];
            for (int i = tid; i < L; i += QT_T) {
                const unsigned long long k = key64[i];
                int rank = -1;
                if (k) {
                    rank = 0;
                    for (int j = 0; j < L; j++) rank += key64[j] > k;
                    q.ordarr[rank] = i;
                }
                q.kord[i] = rank;
            }
            __syncthreads();
            // growth of the list when the first r+1 candidates are split
            for (int r = tid; r < nE; r += QT_T) {
                const int i = q.ordarr[r];
                const uint32_t* c = &cc_cur[4 * i];
                q.cbase[r] = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0) - 1;
            }
            __syncthreads();
            block_excl_scan<QT_T>(q.cbase, nE, s_tmp);
            // inclusive value for r = exclusive[r] + own; count r with size still below N
            uint32_t below_l = 0;
            const int size0 = (int)s_red[0];
            for (int r = tid; r < nE; r += QT_T) {
                const int i = q.ordarr[r];
                const uint32_t* c = &cc_cur[4 * i];
                const int own = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0) - 1;
                below_l += (size0 + (int)q.cbase[r] + own) < N;
            }
            if (below_l) atomicAdd(&s_red[2], below_l);
            __syncthreads();
            nsplit = min((int)s_red[2] + 1, nE);
            for (int i = tid; i < L; i += QT_T)
                if (q.kord[i] >= nsplit) q.kord[i] = -1;
            __syncthreads();
        }
        const int prev_size = (int)s_red[0];
        // ---- B. children created before each split node (creation order = split order, n1..n4)
        for (int i = tid; i < L; i += QT_T) {
            const int k = q.kord[i];
            if (k >= 0) {
                q.ordarr[k] = i;
                const uint32_t* c = &cc_cur[4 * i];
                q.cbase[k] = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
            }
        }
        __syncthreads();
        const int C = (int)block_excl_scan<QT_T>(q.cbase, nsplit, s_tmp);
        // ---- C. unsplit, non-empty nodes keep their relative order behind the children
        for (int i = tid; i < L; i += QT_T)
            q.surv[i] = (q.kord[i] < 0 && (cnt_cur[i] & ~QT_FRESH) > 0) ? 1u : 0u;
        __syncthreads();
        const int S = (int)block_excl_scan<QT_T>(q.surv, L, s_tmp);
        const int Lnew = C + S;
        // ---- D. build the new list
        uint32_t expand_l = 0;
        for (int i = tid; i < L; i += QT_T) {
            const uint32_t c = cnt_cur[i] & ~QT_FRESH;
            const int k = q.kord[i];
            if (k >= 0) {
                const uint32_t bx = box_cur[2 * i], by = box_cur[2 * i + 1];
                const int x0 = bx & 0xFFFF, x1 = bx >> 16, y0 = by & 0xFFFF, y1 = by >> 16;
                const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                int created = (int)q.cbase[k];
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    const uint32_t n = cc_cur[4 * i + qd];
                    if (n == 0) continue;
                    const int pos = C - 1 - created++;
                    const int cx0 = (qd & 1) ? mx : x0, cx1 = (qd & 1) ? x1 : mx;
                    const int cy0 = (qd & 2) ? my : y0, cy1 = (qd & 2) ? y1 : my;
                    box_nxt[2 * pos] = (uint32_t)cx0 | ((uint32_t)cx1 << 16);
                    box_nxt[2 * pos + 1] = (uint32_t)cy0 | ((uint32_t)cy1 << 16);
                    cnt_nxt[pos] = n | QT_FRESH;
                    cc_cur[4 * i + qd] = (uint32_t)pos;
                    expand_l += n > 1;
                }
            } else if (c > 0) {
                const int pos = C + (int)q.surv[i];
                box_nxt[2 * pos] = box_cur[2 * i];
                box_nxt[2 * pos + 1] = box_cur[2 * i + 1];
                cnt_nxt[pos] = c;
                q.surv[i] = (uint32_t)pos;
            }
        }
        if (expand_l) atomicAdd(&s_red[3], expand_l);
        for (int i = tid; i < 4 * Lnew; i += QT_T) cc_nxt[i] = 0;
        __syncthreads();
        const int to_expand = (int)s_red[3];
        // ---- finish test (src/ORBextractor.cc:669-673, 734-735)
        if (Lnew >= N || Lnew == prev_size) done = true;
        else if (mode == 1 && Lnew + 3 * to_expand > N) mode = 2;
        // ---- E. keys follow their nodes
        if (!done) {
            qt_for_keys<QT_T>(kxy, klab, lab_dead, nkeys, tid, [&](const uint32_t (&exy)[QT_KU], const uint32_t (&elab)[QT_KU], int k0) __attribute__((always_inline)) {
                uint32_t li[QT_KU], nl[QT_KU], bx[QT_KU], by[QT_KU];
                int ko[QT_KU];
                bool alive[QT_KU];
#pragma unroll
                for (int u = 0; u < QT_KU; u++) {
                    alive[u] = elab[u] != lab_dead;
                    li[u] = alive[u] ? elab[u] >> 2 : 0u;
                    ko[u] = q.kord[li[u]];
                }
#pragma unroll
                for (int u = 0; u < QT_KU; u++) {
                    const uint32_t qd = elab[u] & 3;
                    nl[u] = *(ko[u] >= 0 ? &cc_cur[4 * li[u] + qd] : &q.surv[li[u]]);
                    if (!alive[u]) nl[u] = 0;
                }
#pragma unroll
                for (int u = 0; u < QT_KU; u++) { bx[u] = box_nxt[2 * nl[u]]; by[u] = box_nxt[2 * nl[u] + 1]; }
#pragma unroll
                for (int u = 0; u < QT_KU; u++) {
                    if (!alive[u]) continue;
                    const uint32_t nq = qt_quadrant(bx[u], by[u], exy[u] & 0xFFFF, exy[u] >> 16);
                    atomicAdd(&cc_nxt[4 * nl[u] + nq], 1u);
                    klab[k0 + u * QT_T] = (Lab)(nq | (nl[u] << 2));
                }
            });
        } else {
            // ---- best response per node, first in candidate order on ties (src/ORBextractor.cc:744-760)
            for (int i = tid; i < Lnew; i += QT_T) key64[i] = 0ull;
            best = key64;
            __syncthreads();
            qt_for_keys<QT_T>(kxy, klab, lab_dead, nkeys, tid, [&](const uint32_t (&exy)[QT_KU], const uint32_t (&elab)[QT_KU], int k0) __attribute__((always_inline)) {
                uint32_t li[QT_KU], nl[QT_KU], sc[QT_KU];
                int ko[QT_KU];
                bool alive[QT_KU];
#pragma unroll
                for (int u = 0; u < QT_KU; u++) {
                    alive[u] = elab[u] != lab_dead;            // (the slots past the end are dropped keys)
                    li[u] = alive[u] ? elab[u] >> 2 : 0u;
                    ko[u] = q.kord[li[u]];
                    sc[u] = alive[u] ? (uint32_t)ksc[k0 + u * QT_T] : 0u;
                }
#pragma unroll
                for (int u = 0; u < QT_KU; u++) {
                    const uint32_t qd = elab[u] & 3;
                    nl[u] = *(ko[u] >= 0 ? &cc_cur[4 * li[u] + qd] : &q.surv[li[u]]);
                }
#pragma unroll
                for (int u = 0; u < QT_KU; u++) {
                    if (!alive[u]) continue;
                    const int x = exy[u] & 0xFFFF, y = exy[u] >> 16;
                    const int cx = (int)(umul24((uint32_t)(x - 3), g.wcell_m22) >> 22);
                    const int cy = (int)(umul24((uint32_t)(y - 3), g.hcell_m22) >> 22);
                    // (every factor stays below 2^24 and every product below 2^32: full-rate 24-bit multiplies)
                    const uint32_t row_in_cells = umul24((uint32_t)(__mul24(cy, g.ncols) + cx), (uint32_t)g.hcell) +
                                                  (uint32_t)(y - 3 - __mul24(cy, g.hcell));
                    const uint32_t ord = umul24(row_in_cells, (uint32_t)g.wcell) + (uint32_t)(x - 3 - __mul24(cx, g.wcell));
                    atomicMax(&key64[nl[u]], ((unsigned long long)sc[u] << 32) | (0xFFFFFFFFu - ord));
                }
            });
        }
        __syncthreads();
        L = Lnew;
        cur = nxt;
    }
    if (!done || best == nullptr) {
        // the round limit ended the loop (a node is split down to single pixels in far fewer rounds than that): no
        // keypoints for this level and an error for the call, rather than a read through a pointer that was never set
        if (tid == 0) { atomicOr(status, 4); lvl_cnt[f * plan->nlevels + level] = 0; }
        return;
    }
    // ---- output in list order
    LevelKp* out = lvl_kp + (size_t)f * plan->kp_per_frame + g.kp_off;
    for (int i = tid; i < L; i += QT_T) {
        const unsigned long long b = best[i];
        const uint32_t ord = 0xFFFFFFFFu - (uint32_t)b;
        const int xin = ord % g.wcell;
        uint32_t t = ord / g.wcell;
        const int yin = t % g.hcell;
        t /= g.hcell;
        const int cx = t % g.ncols, cy = t / g.ncols;
        LevelKp kp;
        kp.x = (uint16_t)(cx * g.wcell + xin + 3 + ORB_MINB);
        kp.y = (uint16_t)(cy * g.hcell + yin + 3 + ORB_MINB);
        kp.score = (uint16_t)(b >> 32);
        kp.pad = 0;
        out[i] = kp;
    }
    if (tid == 0) lvl_cnt[f * plan->nlevels + level] = L;
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor, one wave per keypoint, four keypoints per workgroup.
// The reference blurs whole levels and then samples 512 points per keypoint; only the 39x39
// neighbourhood of a keypoint is ever sampled, so each wave stages the 45x45 raw patch in LDS,
// runs the horizontal 7-tap pass over it (45x39, kept as exact 16-bit sums) and evaluates the
// vertical pass only at the sampled positions. Bits are packed with wave ballots.
// ------------------------------------------------------------------------------------------------
#define DS_RAW_P 52               // raw patch pitch (13 dwords)
#define DS_RAW_BYTES (45 * DS_RAW_P)
#define DS_HP_P 46                // horizontal-pass pitch (u16 entries): the sums are kept column by column (45 rows of a
                                  // column one after the other), so that the seven sums under a test point are 14 adjacent bytes;
                                  // 23 dwords: neighbouring columns start in different banks
#define DS_HP_ENTRIES (40 * DS_HP_P + 8)   // 40 columns (the last one is padding) and the tail of a 16-byte read
struct __attribute__((packed, aligned(4))) HpWindow { uint32_t a, b, c, d; };   // eight sums read at dword alignment

__device__ __align__(16) const signed char d_pattern[1024] = {
#include "../../include/orb_pattern_31.inc"
};

typedef float v2f_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {   // cv::fastAtan2
    const float s = (float)(180 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = __fdiv_rn(ax, ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cosf / sinf of an angle in [0, 2*pi] exactly as glibc 2.28+ computes them (sysdeps/ieee754/flt-32/
// s_sinf.c, s_cosf.c, sincosf.h: argument reduction by multiples of pi/2 and two short polynomials,
// all in double, one rounding to float). The reference calls cosf / sinf (src/ORBextractor.cc:113-114)
// and a last-bit difference can flip the rounding of a rotated test point, so the library's
// arithmetic is restated operation by operation; a host program that ran this restatement against
// libm over every float in [2^-15, 2*pi] found no difference, with or without fused multiply-adds.
__device__ __forceinline__ void sincos_rounded(float y, float* c_out, float* s_out) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    double x = (double)y;
    int n = 0;
    if (!(fabsf(y) < 0x1.921FB6p-1f)) {                        // reduce_fast
        const double r = x * hpi_inv;
        n = ((int)r + 0x800000) >> 24;
        x = x - (double)n * hpi;
    }
    const double sgn = ((n + 1) & 2) ? -1.0 : 1.0;             // sign[n & 3] = {1, -1, -1, 1}
    const double xs = x * sgn, x2 = x * x;
    // sinf_poly, even quadrant
    const double x3 = xs * x2, sp1 = s2 + x2 * s3, x7 = x3 * x2, sp = xs + x3 * s1;
    const double sn = sp + x7 * sp1;
    // sinf_poly, odd quadrant (the table for quadrants 2 and 3 holds the negated coefficients)
    const double x4 = x2 * x2, cp2 = c3 + x2 * c4, cp1 = c0 + x2 * c1, x6 = x4 * x2, cp = cp1 + x4 * c2;
    const double cs0 = cp + x6 * cp2;
    const double cs = (n & 2) ? -cs0 : cs0;
    *s_out = (float)((n & 1) ? cs : sn);
    *c_out = (float)((n & 1) ? sn : cs);
}

// Diagnostic: sincos_rounded of the floats whose bit patterns are first_bits .. first_bits+count-1,
// so that a test can compare the device arithmetic with the host's cosf / sinf over a whole range.
__global__ __launch_bounds__(256) void k_debug_sincos(unsigned first_bits, unsigned count,
                                                      float* __restrict__ out_cos, float* __restrict__ out_sin) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    float c, s;
    sincos_rounded(__uint_as_float(first_bits + i), &c, &s);
    out_cos[i] = c;
    out_sin[i] = s;
}

__global__ __launch_bounds__(256) void k_describe(const PlanDev* __restrict__ plan,
                                                  const uint8_t* __restrict__ pyr,
                                                  const LevelKp* __restrict__ lvl_kp,
                                                  const int* __restrict__ lvl_cnt,
                                                  float* __restrict__ out_kp,   // 7 floats per keypoint
                                                  uint8_t* __restrict__ out_desc, int* __restrict__ out_count,
                                                  int cap, int frame0, int* __restrict__ status,
                                                  int frames, int ngroups, unsigned ngroups8_magic, Level0View v0) {
    __shared__ __align__(16) uint8_t s_raw[4][DS_RAW_BYTES];
    __shared__ __align__(16) uint16_t s_hp[4][DS_HP_ENTRIES];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // everything per keypoint is wave-uniform
    // One-dimensional grid: 8 consecutive workgroups (one per XCD, each with its own L2) take the
    // same group of 4 keypoints in 8 consecutive frames, then the next group of the same frames:
    // an XCD works through one frame at a time, whose pyramid then stays in its L2.
    const unsigned id = blockIdx.x;
    const unsigned fb = ngroups8_magic ? __umulhi(id, ngroups8_magic) : id;    // id / (8 * ngroups)
    const unsigned within = id - fb * (unsigned)(8 * ngroups);
    const int f = (int)(fb * 8 + (within & 7));
    if (f >= frames) return;
    const int idx = (int)(within >> 3) * 4 + wv;
    const int nl = plan->nlevels;
    // locate (level, j) of output keypoint idx: levels are concatenated in order (src/ORBextractor.cc:1076-1104)
    int level = -1, j = idx, total = 0;
    for (int l = 0; l < nl; l++) {
        const int c = lvl_cnt[f * nl + l];
        if (level < 0) {
            if (j < c) level = l; else j -= c;
        }
        total += c;
    }
    if (idx == 0 && lane == 0) {
        out_count[frame0 + f] = min(total, cap);
        if (total > cap) atomicOr(status, 2);
    }
    // Waves are independent from here on (each has its own LDS slices and there is no workgroup
    // barrier): LDS operations of one wave complete in issue order, so the stage -> read hand-offs
    // below need no barrier.
    if (level < 0 || idx >= cap) return;
    const LevelGeom g = plan->lv[level];
    const LevelKp kp = lvl_kp[(size_t)f * plan->kp_per_frame + g.kp_off + j];
    const int kx = kp.x, ky = kp.y;
    // pixel (0, 0) of the level: in its plane, or (level 0 read in place) in the caller's image
    const bool inplace = level == 0 && v0.org != nullptr;
    const int pitch = inplace ? v0.pitch : g.pitch;
    const uint8_t* P = inplace ? v0.org + (size_t)f * v0.frame_bytes
                               : pyr + g.pyr_off + (size_t)f * g.plane_bytes + (size_t)ORB_OY * g.pitch + ORB_OX;
    // ---- stage rows ky-22..ky+22, 13 dwords per row starting at column kx-22 (byte-aligned loads: staged
    // byte c of a row is column kx-22+c, whatever the alignment of that column in memory)
    const int col0 = kx - 22;                     // >= -3: keypoints are >= 19 px from the level edge
                                                  // (a plane has 32 border columns left of x = 0 and 19 rows above y = 0)
    uint8_t* raw = s_raw[wv];
    // disc weights of this lane's row for the orientation step below (loaded early)
    const uint4* dwp = reinterpret_cast<const uint4*>(plan->disc_w[lane < 31 ? (lane < 15 ? 15 - lane : lane - 15) : 0]);
    const uint4 dw0 = dwp[0], dw1 = dwp[1], dw2 = dwp[2], dw3 = dwp[3];
    // (16 lanes per row, 13 of them active, 4 rows per trip: addresses advance by a constant; rows
    // 0 and 44 are never read: the moments use rows 7..37 and the blur rows 1..43)
    const int sr = 1 + (lane >> 4), sc = lane & 15;
    uint8_t* dst = &raw[sr * DS_RAW_P + sc * 4];
    // Only the level's own pixels are stored: where a patch leaves the level (the blur of a keypoint's
    // outermost test points reaches up to three pixels past the edge, BORDER_REFLECT_101 of
    // src/ORBextractor.cc:1086) the missing rows and columns are rebuilt by reflection. Wave-uniform
    // branches; the columns beyond kx+-22 of a staged row are never used.
    const bool edge = !(kx >= 22 && kx + 22 <= g.w - 1 && ky >= 21 && ky + 22 <= g.h - 1);
    if (inplace && (edge || col0 + 52 > v0.xlim)) {
        // an image read in place has nothing readable around it: byte by byte, reflecting the coordinates
        if (sc < 13) {
#pragma unroll 1
            for (int p = 0; p < 11; p++) {
                const int r = 4 * p + sr;
                if (r >= 45) break;
                int y = ky - 22 + r;
                y = y < 0 ? -y : y >= g.h ? 2 * (g.h - 1) - y : y;
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    int x = col0 + 4 * sc + b;
                    x = x < 0 ? -x : x >= g.w ? 2 * (g.w - 1) - x : x;
                    v |= (uint32_t)P[y * pitch + x] << (8 * b);
                }
                *reinterpret_cast<uint32_t*>(dst + p * 4 * DS_RAW_P) = v;
            }
        }
    } else {
        int goff = (ky - 22 + sr) * pitch + col0 + sc * 4;     // a frame is far below 2 GB
        if (sc < 13) {
#pragma unroll
            for (int p = 0; p < 11; p++) {
                if (4 * p + sr < 45) {
                    uint32_t v;
                    __builtin_memcpy(&v, P + goff, 4);         // one dword load at any byte alignment
                    *reinterpret_cast<uint32_t*>(dst + p * 4 * DS_RAW_P) = v;
                }
                goff += 4 * pitch;
            }
        }
        if (edge) {
            // A plane has room around the level (32 columns left of x = 0, at least 19 right of w, 19 rows
            // above and below), so the loads above were in bounds, but nothing is stored there: the (at most
            // three) rows and columns of the patch outside the level are copied from their mirror images
            // inside the staged patch, rows first, so that the corners come out right. LDS operations of
            // a wave complete in order.
            uint32_t* raw32 = reinterpret_cast<uint32_t*>(raw);
            const int j = lane >> 4;                           // 0..3: which row from the top / bottom of the patch
            if (sc < 13) {
                const int rt = 1 + j, yt = ky - 22 + rt;       // mirror of row y < 0 is row -y: staged row rt - 2 * yt
                if (yt < 0) raw32[rt * (DS_RAW_P / 4) + sc] = raw32[(rt - 2 * yt) * (DS_RAW_P / 4) + sc];
                const int rb = 44 - j, yb = ky - 22 + rb;      // mirror of row y >= h is row 2 (h - 1) - y
                if (yb >= g.h) raw32[rb * (DS_RAW_P / 4) + sc] = raw32[(rb - 2 * (yb - (g.h - 1))) * (DS_RAW_P / 4) + sc];
            }
            for (int t = lane; t < 3 * 44; t += 64) {
                const int jj = (t >= 88) ? 2 : (t >= 44) ? 1 : 0;
                uint8_t* row = raw + (1 + t - 44 * jj) * DS_RAW_P;
                const int xl = -1 - jj, xr = g.w + jj;         // columns left / right of the level
                if (xl >= col0) row[xl - col0] = row[-xl - col0];
                if (xr - col0 < DS_RAW_P) row[xr - col0] = row[2 * (g.w - 1) - xr - col0];
            }
        }
    }
    // ---- IC_Angle (src/ORBextractor.cc:77-104): integer moments over the 749-px disc. Lane = one
    // column u of one half of the disc; the disc is symmetric, so column u spans |v| <= umax[|u|].
    int m10 = 0, m01 = 0;
    if (lane < 31) {
        // lane = row v of the disc: 32 bytes from u = -15 on, two byte dot products per dword
        // (sum of I and sum of (u + 15) * I over |u| <= umax[|v|])
        const int first = 7;                      // byte of u = -15 in the staged row
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(&raw[(lane + 7) * DS_RAW_P]) + (first >> 2);
        const uint32_t sh = first & 3;
        uint32_t d[9];
#pragma unroll
        for (int k = 0; k < 9; k++) d[k] = rw[k];
        const uint32_t w[16] = {dw0.x, dw0.y, dw0.z, dw0.w, dw1.x, dw1.y, dw1.z, dw1.w,
                                dw2.x, dw2.y, dw2.z, dw2.w, dw3.x, dw3.y, dw3.z, dw3.w};
        uint32_t s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t e = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
            s0 = __builtin_amdgcn_udot4(e, w[k], s0, false);
            s1 = __builtin_amdgcn_udot4(e, w[8 + k], s1, false);
        }
        m10 = (int)s1 - __mul24(15, (int)s0);     // s0 <= 31 * 255: 24-bit multiplies are exact
        m01 = __mul24(lane - 15, (int)s0);
    }
    m10 = __builtin_amdgcn_readlane((int)wave_scan_incl((uint32_t)m10), 63);
    m01 = __builtin_amdgcn_readlane((int)wave_scan_incl((uint32_t)m01), 63);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // ---- horizontal blur pass: hp[r][c] for rows ky-22..ky+22, columns kx-19..kx+19. A lane
    // produces 4 adjacent sums from 12 staged bytes (3 dwords) with v_dot4_u32_u8: the seven taps of sum q
    // lie at bytes q .. q+6, so every sum is two or three byte dot products of the same three dwords
    // against the taps shifted to where they belong (ten constant patterns; no byte shifting of the data).
    const int t0 = plan->blur_taps[0], t1 = plan->blur_taps[1], t2 = plan->blur_taps[2], t3 = plan->blur_taps[3];
    const uint32_t T[7] = {(uint32_t)t3, (uint32_t)t2, (uint32_t)t1, (uint32_t)t0, (uint32_t)t1, (uint32_t)t2, (uint32_t)t3};
    auto pat = [&](int k0) -> uint32_t {              // bytes = taps k0 .. k0+3 (zero outside 0..6)
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) if (k0 + i >= 0 && k0 + i < 7) v |= T[k0 + i] << (8 * i);
        return v;
    };
    uint16_t* hp = s_hp[wv];
    // (only the items a rotated test point can need: see PlanDev::hp_items)
    uint32_t items[ORB_HP_ITEMS / 64];
#pragma unroll
    for (int k = 0; k < ORB_HP_ITEMS / 64; k++) items[k] = plan->hp_items[64 * k + lane];
#pragma unroll
    for (int k = 0; k < ORB_HP_ITEMS / 64; k++) {
        // sums c..c+3 of rows r and r+1 (r even; column 39 is padding): the item holds both byte offsets
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(raw + (items[k] & 0xFFFF));
        uint32_t acc[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t d0 = rw[rr * (DS_RAW_P / 4)], d1 = rw[rr * (DS_RAW_P / 4) + 1], d2 = rw[rr * (DS_RAW_P / 4) + 2];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // sum q: bytes q+k of the twelve times tap k: dword j holds the taps 4j-q .. 4j-q+3
                uint32_t a = __builtin_amdgcn_udot4(d0, pat(-q), 0u, false);
                a = __builtin_amdgcn_udot4(d1, pat(4 - q), a, false);
                if (q >= 2) a = __builtin_amdgcn_udot4(d2, pat(8 - q), a, false);
                acc[rr][q] = a;
            }
        }
        uint32_t* o = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(hp) + (items[k] >> 16));   // column c, rows r and r+1
#pragma unroll
        for (int q = 0; q < 4; q++) o[q * (DS_HP_P / 2)] = acc[0][q] | (acc[1][q] << 16);
    }
    // ---- steered BRIEF (src/ORBextractor.cc:107-147)
    const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
    const float ang_rad = angle * factorPI;
    float a, b;                                   // cosf(ang_rad), sinf(ang_rad)
    sincos_rounded(ang_rad, &a, &b);
    const int ties_even = plan->blur_ties_even;
    const v2f_t ab = {a, b}, nba = {-b, a};
    // tap pairs for the dot products over the eight sums of a window (rows -3 .. +4; the last one gets weight 0)
    const v2us_t w01 = {(unsigned short)t3, (unsigned short)t2}, w23 = {(unsigned short)t1, (unsigned short)t0};
    const v2us_t w45 = {(unsigned short)t1, (unsigned short)t2}, w67 = {(unsigned short)t3, 0};
    // cvRound without conversions: adding 1.5 * 2^23 to a float of small magnitude rounds it to the nearest
    // integer, ties to even, and leaves 0x4B400000 + that integer in the bits of the sum. The two biased
    // integers go into the address as they are (the 24-bit multiply sees 0x400000 + ry); the biases are
    // taken off once, as a constant.
    const float round_magic = 12582912.0f;
    const int magic_bits = 0x4B400000;
    const int index_bias = 0x400000 * DS_HP_P + magic_bits;
    // the legacy column kernel rounds an exact half to even for x < (w & ~3): kx + rx below that limit
    const int tie_limit = ties_even ? (g.w & ~3) - kx + magic_bits : magic_bits - 0x10000;
    unsigned long long bits[4];
#pragma unroll
    for (int rnd = 0; rnd < 4; rnd++) {
        const int pair = rnd * 64 + lane;
        const int pk = *reinterpret_cast<const int*>(&d_pattern[4 * pair]);
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float px = (float)(signed char)(pk >> (16 * e));
            const float py = (float)(signed char)(pk >> (16 * e + 8));
            // (rx, ry) = round(px * a - py * b, px * b + py * a): two packed multiplies, each product
            // rounded on its own as in the scalar expression
            // (px * a - py * b as px * a + py * (-b): the same products and the same sum)
            const v2f_t pxv = v2f_t{px, px} * ab, pyv = v2f_t{py, py} * nba;
            const v2f_t rr = (pxv + pyv) + v2f_t{round_magic, round_magic};
            const int irx = __float_as_int(rr.x), iry = __float_as_int(rr.y);
            // the window starts at the top tap (column rx, row ry - 3): one 16-byte read, four dot products
            // of sum pairs against tap pairs, the rounding half already in the sum
            // (read from the dword the window starts in; a window that starts in its upper half is moved down
            // by two bytes)
            const int hidx = __mul24(irx, DS_HP_P) + iry - (index_bias - (19 * DS_HP_P + 19));
            const HpWindow hw = *reinterpret_cast<const HpWindow*>(hp + (hidx & ~1));
            const uint32_t sh = ((uint32_t)hidx & 1u) << 1;
            uint32_t m = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, __builtin_amdgcn_alignbyte(hw.b, hw.a, sh)), w01, 0x8000u, false);
            m = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, __builtin_amdgcn_alignbyte(hw.c, hw.b, sh)), w23, m, false);
            m = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, __builtin_amdgcn_alignbyte(hw.d, hw.c, sh)), w45, m, false);
            m = __builtin_amdgcn_udot2(__builtin_bit_cast(v2us_t, __builtin_amdgcn_alignbyte(0u, hw.d, sh)), w67, m, false);
            // round half up: m >> 16; an exact half (low 16 bits of m zero) that came out odd goes down to the
            // even value where the legacy rule applies
            const bool tie_odd = (m & 0x1FFFFu) == 0x10000u && irx < tie_limit;
            const uint32_t qv = (m >> 16) - (uint32_t)tie_odd;
            val[e] = (int)min(qv, 255u);
        }
        bits[rnd] = __ballot(val[0] < val[1]);
    }
    const size_t o = (size_t)(frame0 + f) * cap + idx;
    if (lane < 4) {
        const unsigned long long v = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long*>(out_desc + o * 32)[lane] = v;
    }
    if (lane == 0) {
        float* K = out_kp + o * 7;
        const float sc = g.scale;
        K[0] = level ? (float)kx * sc : (float)kx;
        K[1] = level ? (float)ky * sc : (float)ky;
        K[2] = g.patch_size;
        K[3] = angle;
        K[4] = (float)kp.score;
        reinterpret_cast<int*>(K)[5] = level;
        reinterpret_cast<int*>(K)[6] = -1;
    }
}

// Shared host/device definitions of libORBhip: level geometry, buffer layout, launch arguments.
//
// Device layout of one pyramid level of one frame ("plane"): rows of `pitch` bytes, (h + 38) rows.
// Level pixel (x, y) sits at byte (y + ORB_OY) * pitch + (x + ORB_OX); the 19-pixel REFLECT_101
// border the reference keeps around every level (src/ORBextractor.cc:1113-1128) therefore
// occupies columns ORB_OX-19 .. ORB_OX-1 and ORB_OX+w .. ORB_OX+w+18. ORB_OX = 32 keeps the first
// interior pixel 16-byte aligned; pitch is a multiple of 64. Planes are stored level-major:
// plane(level, frame) = pyr + lv.pyr_off + frame * lv.plane_bytes.
//
// FAST corners travel in two stages. k_fast writes them into fixed output segments, one per
// (tile, wave), with a count per segment (no atomics, any order). k_quadtree gathers the segments
// of its level into a compact key array and works on that.
#pragma once
#include <stdint.h>

#define ORB_OX 32
#define ORB_OY 19
#define ORB_EDGE 19           // EDGE_THRESHOLD (src/ORBextractor.cc:74)
#define ORB_MINB 16           // minBorderX/Y = EDGE_THRESHOLD-3 (src/ORBextractor.cc:773)
#define ORB_MAX_LEVELS 16
#define ORB_HP_ITEMS 192      // 3 trips of a wave

struct RsTab {                // one bilinear tap pair of cv::resize (8 bytes)
    uint16_t s0, s1;          // source indices (already clamped)
    int16_t a0, a1;           // 11-bit fixed-point weights
};

struct LevelGeom {
    int w, h;                 // level size
    int pitch, rows;          // plane row pitch (bytes) and row count (h + 38)
    unsigned long long plane_bytes;
    unsigned long long pyr_off;      // byte offset of plane(level, frame 0)
    // FAST cell grid (src/ORBextractor.cc:773-787)
    int maxbx, maxby;         // maxBorderX/Y = w-16, h-16
    int ncols, nrows, wcell, hcell;
    unsigned wcell_magic, hcell_magic;   // floor(2^32/d)+1: n/d == umulhi(n, magic) for n < 65536, 1 < d < 65536
    unsigned wcell_m22, hcell_m22;       // ceil(2^22/d): n/d == (n * m) >> 22 for n * d < 2^22, a 24-bit multiply
    // FAST output segments
    int ntx, nty;             // tile columns and tile rows of k_fast (248 x 32 pixels each)
    // Workgroups of k_fast per frame: fast_ntx_plain * nty plain tiles, then the packed tiles of the last
    // tile column (fast_bands tile rows each) when that column is narrow enough to hold two or more bands
    // of fast_lpb lanes in a wave (fast_bands == 1: no packing, fast_ntx_plain == ntx). 4 segments per workgroup.
    int fast_ntiles, fast_ntx_plain, fast_bands, fast_lpb;
    unsigned long long seg_off;      // entry offset of this level's first segment inside a frame's segment block
    int segc_off;             // offset of this level's first segment count inside a frame's count block
    // compact keys of the quadtree
    int keys_cap;             // capacity of the compact key array of this level
    unsigned long long keys_off;     // entry offset inside a frame's key block
    // quadtree (src/ORBextractor.cc:539-563)
    int n_ini; float hx;      // nIni, hX
    int quota;                // mnFeaturesPerLevel[level]
    int node_cap;             // max list length the distribution can reach
    int qt_words;             // LDS words of k_quadtree before its cell mask: max(18 * node_cap, segments + 1)
    int kp_off;               // offset of this level's slot in the per-frame level-keypoint block
    // output
    float scale;              // mvScaleFactor[level]
    float patch_size;         // (float)(int)(31 * mvScaleFactor[level])
    // resize tables
    int xtab_off, ytab_off;   // offsets (entries) into the RsTab buffer
    int rs_narrow;            // 1: the taps of every 4-pixel group of this level span at most 8 source bytes
};

struct PlanDev {
    int nlevels;
    int ini_th, min_th;
    unsigned long long seg_per_frame;    // entries of all segments of one frame
    int segc_per_frame;                  // segments of one frame
    unsigned long long keys_per_frame;   // sum over levels of keys_cap
    int kp_per_frame;                    // sum over levels of node_cap
    int blur_taps[4];                    // centre, +-1, +-2, +-3
    int blur_ties_even;                  // 1: legacy x86 column kernel (ties to even for x < (w & ~3))
    int umax[16];
    // Byte weights of the orientation disc, one row of 32 bytes per |v| (byte i <-> u = i - 15):
    // words 0..7 hold 1 where |u| <= umax[|v|], words 8..15 hold u + 15 there; 0 elsewhere.
    uint32_t disc_w[16][16];
    // Work items of the descriptor kernel's horizontal blur pass (4 adjacent sums of an even row of the
    // 45 x 40 grid of row sums and of the row below it per item), only where a rotated test point can
    // ever need them (the pattern lies inside a circle of radius sqrt(338)): byte offset of the item's
    // first pixel in the staged patch | byte offset of its first sum in the column-major array of sums
    // << 16; padded with repeats to ORB_HP_ITEMS.
    uint32_t hp_items[ORB_HP_ITEMS];
    // k_fast handles all levels in one launch per kind of tile: workgroups [fast_cum[k][l] * frames,
    // fast_cum[k][l+1] * frames) belong to level l (running sum of that kind's tiles per frame); magics for
    // the divisions by the tiles per frame and the plain tiles per row of a level (grid_magic)
    // k_quadtree: slot of a level's node arrays in the global spill buffer, or -1 when they fit in LDS
    int qt_spill_slot[ORB_MAX_LEVELS];
    int qt_nspill;                       // levels with a slot
    uint32_t fast_cum[2][ORB_MAX_LEVELS + 1];        // [0] plain tiles, [1] packed tiles (their own launch)
    uint32_t fast_tiles_magic[2][ORB_MAX_LEVELS], fast_ntx_magic[ORB_MAX_LEVELS];
    LevelGeom lv[ORB_MAX_LEVELS];
};

// Level 0 read in place. A gray input image already is level 0 (src/ORBextractor.cc:1107-1132 copies it into
// a bordered buffer, but nothing on the path reads that border except the blur of keypoints within three
// pixels of the detectable area's edge, which the descriptor kernel reflects itself). When the input
// allows it (4-byte aligned rows) the kernels read level 0 from the caller's buffer and the copy into a
// plane is skipped. org == nullptr: level 0 lives in its plane like the other levels.
struct Level0View {
    const uint8_t* org;               // pixel (0, 0) of frame 0
    unsigned long long frame_bytes;   // distance between frames
    int pitch;                        // distance between rows (multiple of 4)
    int xlim;                         // bytes of every row that may be read (multiple of 4, >= w)
};

struct LevelKp {              // keypoint of one level after the quadtree, level coordinates
    uint16_t x, y;
    uint16_t score;
    uint16_t pad;
};

// Division of a workgroup index by a grid extent d through a multiply: m = ceil(2^32 / d) (0 for
// d == 1: the quotient is n itself) gives n / d == umulhi(n, m) whenever n * d < 2^32; the host
// checks that bound for the grids it launches.
static inline unsigned grid_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

#ifdef __HIPCC__
// Unsigned 24-bit multiply (low 32 bits of the product). HIP declares __umul24 with an int result: a product
// of 2^31 or more would then be shifted arithmetically (the cell divisions below shift by 22).
//
// Contract of every 24-bit multiply in the kernels (umul24 / __mul24 read only the low 24 bits of their
// operands and return the low 32 bits of the product), with the bound that keeps each site inside it.
// The host refuses images beyond 16384 pixels a side and checks the size-dependent bounds in ensure_plan:
//   cell division   umul24(n, ceil(2^22 / d)) >> 22 == n / d   n < 16384 (a level coordinate), d = wcell or hcell >= 30,
//                   so the factor is below 2^18, the product below 2^32, and the quotient exact while n * d < 2^22
//                   (checked per level on the host); the remainder n - __mul24(q, d): q < 547, d < 16384
//   cell index      __mul24(cy, ncols): cy, ncols < 547
//   key order       umul24(cell, hcell) and umul24(cell * hcell + yin, wcell) in k_quadtree: cell * hcell + yin is below
//                   rows-of-the-level * ncols < 2^24 (checked per level on the host), the second product below w * h <= 2^28
//   LDS offsets     __mul24(row, FT_P): row < 48, FT_P = 260; __mul24(band, FT_ROWS), __mul24(band, 4 * lpb): band < 32;
//                   __mul24(irx, DS_HP_P) in k_describe: the low 24 bits of irx are 0x400000 + rx with |rx| <= 19 (see there)
//   band division   umul24(lane, ceil(2^16 / lpb)) >> 16: lane < 64, factor <= 2^16 / 3
//   resize          umul24(weight, row value): an 11-bit weight times a 15-bit value
//   moments         __mul24(15, s0), __mul24(lane - 15, s0): s0 <= 31 * 255
//   stereo windows  __mul24(y, pitch): y < 16384 + 38, pitch <= 16384 + 128, product below the plane size < 2^31 (checked on
//                   the host); __mul24(t, 745) >> 13 == t / 11 for t < 121
__device__ __forceinline__ uint32_t umul24(uint32_t a, uint32_t b) { return (uint32_t)__umul24(a, b); }

// Workgroups are dealt to the 8 XCDs (each with its own L2) round-robin in dispatch order, so
// tiles that are neighbours in the image would land on different L2s and each fetch the rows and
// cache lines they share from memory again. This remap gives every group of workgroups that
// share an XCD a contiguous part of a range of `n` consecutive workgroups instead (`i` = position
// inside the range; the XCD of a workgroup is its dispatch index modulo 8, so positions that are
// equal modulo 8 share one wherever the range starts). It is a bijection of
// the range for any size; a wrong guess about placement costs speed, never correctness.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned i, unsigned n) {
    const unsigned q = n >> 3, r = n & 7, xcd = i & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (i >> 3);
}
#endif

// FAST-9/16 corner detection for one pyramid level (src/ORBextractor.cc:789-829 and
// cv::FAST(img, kps, threshold, true) behind it).
//
// Score S(p) = max over the 16 nine-pixel arcs of min(centre - ring) (centre brighter) or
// min(ring - centre) (centre darker), minus 1. p is a FAST-9 corner at threshold t iff S(p) >= t and
// cv::FAST's cornerScore of a corner equals S(p) whatever threshold found it. In the reference
// every ~30-px cell is a separate cv::FAST call, first at iniThFAST and, when that yields nothing,
// at minThFAST. Because S does not depend on t this is equivalent to: p survives iff
// S(p) >= minTh and S(p) is strictly greater than the S of its 8-neighbours inside the same cell;
// a survivor is used iff S(p) >= iniTh or its cell has no survivor with S >= iniTh. This kernel
// emits all survivors; the quadtree kernel applies the last rule when it gathers them.
//
// Work split. A workgroup (4 waves) owns a tile of 248 columns x 32 rows of the detectable area.
// Scores are needed for tile rows -1..32 (34 rows); the waves sieve 9, 9, 8 and 8 of them.
//   sieve  (dense)  : every lane holds 4 adjacent pixels. A corner at threshold t needs, for each
//                     of the opposite ring pairs (N,S) and (E,W), one member that differs from the
//                     centre by more than t with the same sign. Two pixels at a time are tested
//                     with packed 16-bit ops (v_perm_b32, v_pk_sub/min/max_i16) straight from
//                     registers: the rows of a five-row step are loaded together (11
//                     coalesced dword loads in flight per lane, neighbour dwords by DPP wave
//                     shifts), the step is fully unrolled. Straight
//                     edges fail this test; the survivors of a step are compacted into a
//                     per-wave queue with one wave prefix sum (raw codes, decoded where they
//                     are consumed). The loaded rows are also copied into an image tile in LDS.
//   score  (sparse) : queued pixels are taken 128 at a time, two per lane in the halves of packed
//                     16-bit values (a remainder waits for the next step and the last one is
//                     taken one per lane if it is small): 16 ring bytes from the LDS image tile, exact score,
//                     written to a byte map of the tile in LDS; the position of a scored output
//                     pixel is also appended to a list of the tile (ballot + one LDS add per batch).
//   nms    (sparse) : after a barrier the waves share the list, one entry per lane:
//                     non-maximum suppression inside the cell (all eight neighbour reads in
//                     flight), then a plain store into the wave's own output segment
//                     (no atomics, no ordering between waves: the consumer does not depend on
//                     the order of the candidates). A tile with more scored pixels than the
//                     list holds (dense texture) finds them by scanning the map instead.
// Row bookkeeping is wave-uniform and runs on the scalar unit; tiles are taken in an XCD-aware
// order so that the rows and cache lines neighbouring tiles share are fetched from memory once.
#pragma once
#include <hip/hip_runtime.h>
#include "orb_device.h"

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s u2v(uint32_t u) { return __builtin_bit_cast(v2s, u); }
__device__ __forceinline__ uint32_t v2u(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ v2s pmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s pmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }

// Bytes O and O+1 of the 8-byte string {hi:lo}, zero-extended into the two halves of a dword.
template <int O>
__device__ __forceinline__ v2s pick2(uint32_t hi, uint32_t lo) {
    static_assert(O >= 0 && O + 1 <= 7, "byte offset");
    constexpr uint32_t sel = (uint32_t)O | (0x0Cu << 8) | ((uint32_t)(O + 1) << 16) | (0x0Cu << 24);
    return u2v(__builtin_amdgcn_perm(hi, lo, sel));
}

// Exact corner score of one pixel from its centre value and the 16 ring values (cv::FAST order).
// With d = v - p: max over arcs of min(d) = v - min over arcs of max(p), and min over arcs of max(d) =
// v - max over arcs of min(p), so the arcs are reduced on the ring values themselves and the centre
// enters once at the end.
__device__ __forceinline__ int fast_score16(int v, const int (&p)[16]) {
    int lo2[8], hi2[8];   // min/max of p[i], p[i+1] for odd i = 2j+1
#pragma unroll
    for (int j = 0; j < 8; j++) {
        lo2[j] = min(p[(2 * j + 1) & 15], p[(2 * j + 2) & 15]);
        hi2[j] = max(p[(2 * j + 1) & 15], p[(2 * j + 2) & 15]);
    }
    int lo4[8], hi4[8];   // over p[i..i+3]
#pragma unroll
    for (int j = 0; j < 8; j++) {
        lo4[j] = min(lo2[j], lo2[(j + 1) & 7]);
        hi4[j] = max(hi2[j], hi2[(j + 1) & 7]);
    }
    int amin = 512, bmax = -512;   // min over arcs of the arc's brightest pixel, max over arcs of its darkest
#pragma unroll
    for (int j = 0; j < 8; j++) {   // arcs [k..k+8] and [k+1..k+9], k = 2j, share p[k+1..k+8]
        const int k = 2 * j;
        const int lo8 = min(lo4[j], lo4[(j + 2) & 7]);
        const int hi8 = max(hi4[j], hi4[(j + 2) & 7]);
        amin = min(amin, max(hi8, min(p[k], p[(k + 9) & 15])));
        bmax = max(bmax, min(lo8, max(p[k], p[(k + 9) & 15])));
    }
    return max(v - amin, bmax - v) - 1;
}

// Three-input packed minimum / maximum of pixel values. gfx950 has no three-input packed integer minimum, but
// it has v_pk_minimum3_f16 / v_pk_maximum3_f16, and the values here are 0 .. 255 in 16-bit halves: read as half
// floats those bit patterns are zero or positive denormals, whose order is the order of the integers, and the
// result of a minimum or maximum is one of its inputs, bit for bit (kernels run with 16-bit denormals enabled,
// the compiler's default; nothing here is NaN, infinite or negative).
__device__ __forceinline__ v2s pmin3(v2s a, v2s b, v2s c) {
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(v2u(a)), "v"(v2u(b)), "v"(v2u(c)));
    return u2v(d);
}
__device__ __forceinline__ v2s pmax3(v2s a, v2s b, v2s c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(v2u(a)), "v"(v2u(b)), "v"(v2u(c)));
    return u2v(d);
}

// The same for two pixels at once: every value is a pair of 16-bit halves (packed min / max / sub). The eight
// ring pixels p[k+1..k+8] shared by the arcs [k..k+8] and [k+1..k+9] (k even) are the pairs (k+1,k+2) .. (k+7,k+8):
// three of them reduce in one three-input instruction, the fourth joins the reduction with the pixel that tells the
// two arcs apart (max(min(a, x), min(a, y)) = min(a, max(x, y))), and the eight results reduce three at a time.
__device__ __forceinline__ v2s fast_score16_pk(v2s v, const v2s (&p)[16]) {
    v2s lo2[8], hi2[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        lo2[j] = pmin(p[(2 * j + 1) & 15], p[(2 * j + 2) & 15]);
        hi2[j] = pmax(p[(2 * j + 1) & 15], p[(2 * j + 2) & 15]);
    }
    v2s at[8], bt[8];   // per k = 2j: the brightest pixel of the darker-centre arc pair's best arc, and the mirror image
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = 2 * j;
        const v2s lo6 = pmin3(lo2[j], lo2[(j + 1) & 7], lo2[(j + 2) & 7]);
        const v2s hi6 = pmax3(hi2[j], hi2[(j + 1) & 7], hi2[(j + 2) & 7]);
        at[j] = pmax3(hi6, hi2[(j + 3) & 7], pmin(p[k], p[(k + 9) & 15]));
        bt[j] = pmin3(lo6, lo2[(j + 3) & 7], pmax(p[k], p[(k + 9) & 15]));
    }
    const v2s amin = pmin(pmin3(pmin3(at[0], at[1], at[2]), pmin3(at[3], at[4], at[5]), at[6]), at[7]);
    const v2s bmax = pmax(pmax3(pmax3(bt[0], bt[1], bt[2]), pmax3(bt[3], bt[4], bt[5]), bt[6]), bt[7]);
    const v2s one = {1, 1};
    return pmax(v - amin, bmax - v) - one;
}

#define FT_COLS 248        // output columns per tile (62 lanes x 4; lanes 0 and 63 are halo)
#define FT_ROWS 32         // rows per tile
#define FT_GROUP 5         // rows per unrolled sieve step
#define FT_P 260           // LDS row pitch of map and image tile (tile column xl <-> level x xs - 4 + xl);
                           // 65 dwords: pixels of one column in consecutive rows fall into different banks
#define FT_MAP_ROWS 34     // tile rows -1 .. 32
#define FT_IMG_ROWS 40     // tile rows -4 .. 35
#define FT_QUEUE 256       // compaction window per wave
#define FT_SEGCAP 768      // output segment of one wave: a quarter of a tile's survivors (< 2000) plus a batch, or 8 rows x 248 columns
#ifndef FT_LIST
#define FT_LIST 832        // scored output pixels a tile can list (sized so that 7 workgroups fit a CU's LDS)
#endif

// Inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2,3
    return v;
}

// Queue entries are raw codes  tag << 11 | lane << 5 | b  (b = index of a set bit in the lane's
// word); they are decoded 64 at a time where they are consumed, which keeps the fill loop (one trip
// per set bit of the fullest lane) short. The set bits of the wave are numbered lane by lane.
// CHECKED: only entries whose slot lies in [lo, FT_QUEUE) are written (the caller makes several
// passes); otherwise the caller guarantees that all of them fit.
template <bool CHECKED>
__device__ __forceinline__ void fast_fill_queue(uint32_t bits, int idx, int lo, uint32_t code0, uint16_t* queue) {
    while (bits) {
        const uint32_t b = (uint32_t)__ffs(bits) - 1;
        bits &= bits - 1;
        if (!CHECKED || (idx >= lo && idx < FT_QUEUE)) queue[idx] = (uint16_t)(code0 | b);
        idx++;
    }
}

// Appends the tile positions (map row << 8 | tile column) of this batch's scored output pixels to the
// tile's list: `hit` lanes are numbered with a ballot, one lane reserves the range with an LDS add.
// Entries beyond the list's capacity are dropped; the counter keeps counting, which tells the
// consumer to scan the map instead.
__device__ __forceinline__ void fast_list_append(bool hit_a, uint32_t code_a, bool hit_b, uint32_t code_b, int lane,
                                                 uint16_t* list, int* nlist) {
    const unsigned long long ma = __ballot(hit_a), mb = __ballot(hit_b);
    if (ma | mb) {                                             // wave-uniform
        const int na = __popcll(ma);
        int base = 0;
        if (lane == 0) base = atomicAdd(nlist, na + __popcll(mb));
        base = __builtin_amdgcn_readfirstlane(base);
        if (hit_a) {
            const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0));
            if (slot < FT_LIST) list[slot] = (uint16_t)code_a;
        }
        if (hit_b) {
            const int slot = base + na + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0));
            if (slot < FT_LIST) list[slot] = (uint16_t)code_b;
        }
    }
}

// Lanes of a tile. A plain tile is one band of 64 lanes (62 with output columns between two halo lanes).
// PACKED: the last tile column of a level is often much narrower than 248 pixels; there the 64 lanes hold
// several bands of `lpb` lanes (output lanes between two halo lanes again), each band the same columns of
// another tile row (band b <-> tile row first + b, 32 level rows further down), so that one workgroup does
// the work of several nearly empty ones. LDS tile, map, queue and list are indexed by lane column and
// tile row exactly as in a plain tile; only the translation to level coordinates knows about bands.
struct FastBands {
    int lpb;                 // lanes per band
    int nbands;              // bands of this tile that exist (the last packed tile of a level may have fewer)
    unsigned magic;          // ceil(2^16 / lpb): lane / lpb == (lane * magic) >> 16 for lane < 64
};

// true for a map position that is an output pixel of the tile (map rows 1..32, tile columns 4..251, or
// the columns of a band's inner lanes); the rest of the map is the one-pixel halo that only serves as neighbours
template <bool PACKED>
__device__ __forceinline__ bool fast_is_output(int row, int xl, const FastBands& fb) {
    if (!PACKED) return (unsigned)(row - 1) < (unsigned)FT_ROWS && (unsigned)(xl - 4) < (unsigned)FT_COLS;
    const int ln = xl >> 2, li = ln - __mul24((int)(umul24((uint32_t)ln, fb.magic) >> 16), fb.lpb);
    return (unsigned)(row - 1) < (unsigned)FT_ROWS && (unsigned)(li - 1) < (unsigned)(fb.lpb - 2);
}

// Scores `n` (<= 64) queued pixels, one per lane, from the LDS image tile into the tile map.
// Queue entry: step << 11 | lane << 5 | b with b = 8 * pixel + 7 - row (the sieve's verdict word);
// map row = m0 + FT_GROUP * step + row, image tile row = map row + 3.
template <bool PACKED>
__device__ __forceinline__ void fast_score_queue(const uint16_t* queue, int n, int lane, int m0, const uint8_t* img,
                                                 int min_th, uint8_t* map, uint16_t* list, int* nlist, const FastBands& fb) {
    bool hit = false;
    uint32_t code = 0;
    if (lane < n) {
        const uint32_t e = queue[lane];
        const int xl = (int)((e >> 3) & 0xFC) + (int)((e >> 3) & 3);        // 4 * lane + pixel
        const int row = m0 + FT_GROUP * (int)(e >> 11) + (int)(~e & 7);
        const uint8_t* c = img + (__mul24(row, FT_P) + 3 * FT_P + xl);   // 24-bit multiplies run at full rate
        int p[16];
        p[0] = c[3 * FT_P];      p[1] = c[3 * FT_P + 1];   p[2] = c[2 * FT_P + 2];   p[3] = c[FT_P + 3];
        p[4] = c[3];             p[5] = c[-FT_P + 3];      p[6] = c[-2 * FT_P + 2];  p[7] = c[-3 * FT_P + 1];
        p[8] = c[-3 * FT_P];     p[9] = c[-3 * FT_P - 1];  p[10] = c[-2 * FT_P - 2]; p[11] = c[-FT_P - 3];
        p[12] = c[-3];           p[13] = c[FT_P - 3];      p[14] = c[2 * FT_P - 2];  p[15] = c[3 * FT_P - 1];
        const int s = fast_score16(c[0], p);
        if (s >= min_th) {
            map[__mul24(row, FT_P) + xl] = (uint8_t)s;
            hit = fast_is_output<PACKED>(row, xl, fb);
            code = (uint32_t)(row << 8 | xl);
        }
    }
    fast_list_append(hit, code, false, 0, lane, list, nlist);
}

// Scores `n` (64 < n <= 128) queued pixels, two per lane (entries lane and lane + 64), with packed
// 16-bit arithmetic. A lane without a second entry scores its first one twice.
template <bool PACKED>
__device__ __forceinline__ void fast_score_queue2(const uint16_t* queue, int n, int lane, int m0, const uint8_t* img,
                                                  int min_th, uint8_t* map, uint16_t* list, int* nlist, const FastBands& fb) {
    const bool two = lane + 64 < n;
    const uint32_t ea = queue[lane], eb = queue[two ? lane + 64 : lane];
    const int xa = (int)((ea >> 3) & 0xFC) + (int)((ea >> 3) & 3), xb = (int)((eb >> 3) & 0xFC) + (int)((eb >> 3) & 3);
    const int ra = m0 + FT_GROUP * (int)(ea >> 11) + (int)(~ea & 7), rb = m0 + FT_GROUP * (int)(eb >> 11) + (int)(~eb & 7);
    const int pa = __mul24(ra, FT_P) + xa, pb = __mul24(rb, FT_P) + xb;   // 24-bit multiplies run at full rate
    const uint8_t* ca = img + (pa + 3 * FT_P);
    const uint8_t* cb = img + (pb + 3 * FT_P);
    constexpr int ring[16] = {3 * FT_P, 3 * FT_P + 1, 2 * FT_P + 2, FT_P + 3, 3, -FT_P + 3, -2 * FT_P + 2, -3 * FT_P + 1,
                              -3 * FT_P, -3 * FT_P - 1, -2 * FT_P - 2, -FT_P - 3, -3, FT_P - 3, 2 * FT_P - 2, 3 * FT_P - 1};
    v2s p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) p[k] = v2s{(short)ca[ring[k]], (short)cb[ring[k]]};
    const v2s v = {(short)ca[0], (short)cb[0]};
    const v2s s = fast_score16_pk(v, p);
    const bool sa = s.x >= min_th, sb = two && s.y >= min_th;
    if (sa) map[pa] = (uint8_t)s.x;
    if (sb) map[pb] = (uint8_t)s.y;
    fast_list_append(sa && fast_is_output<PACKED>(ra, xa, fb), (uint32_t)(ra << 8 | xa), sb && fast_is_output<PACKED>(rb, xb, fb),
                     (uint32_t)(rb << 8 | xb), lane, list, nlist);
}

// One tile. xs / y0: level x of the tile's first output column, level y of its first row (of band 0).
// tile: index of the tile inside its level (its output segments are 4 * tile .. 4 * tile + 3).
template <bool PACKED>
__device__ __forceinline__ void fast_tile(const LevelGeom& g, const uint8_t* __restrict__ org, int pitch, int xlimit, int xs, int y0,
                                          int tile, const FastBands fb, int min_th, uint2* __restrict__ out, int* __restrict__ out_count,
                                          int* __restrict__ status, uint8_t* s_map, uint8_t* s_img, uint16_t (*s_queue)[FT_QUEUE],
                                          uint16_t* s_list, int* s_nlist_p) {
    int& s_nlist = *s_nlist_p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: row bookkeeping runs on the scalar unit
    // the lane's band and its place inside it
    int band = 0, li = lane;
    bool band_ok = true;
    if (PACKED) {
        band = (int)(umul24((uint32_t)lane, fb.magic) >> 16);
        li = lane - __mul24(band, fb.lpb);
        band_ok = band < fb.nbands;
    }
    const int last_li = PACKED ? fb.lpb - 1 : 63;
    const int x_hi = g.w - ORB_EDGE, y_hi = g.h - ORB_EDGE;    // detectable area [19, x_hi) x [19, y_hi)
    const int xg = xs - 4 + 4 * li;                            // level x of this lane's pixel 0 (multiple of 4, >= 12)
    const bool col_ok = band_ok && xg + 4 <= xlimit;           // the lane's dword exists
    // map row m <-> level row y0 - 1 + m (+ 32 * band); image tile row t <-> level row y0 - 4 + t (+ 32 * band)

    for (int i = tid; i < FT_MAP_ROWS * FT_P / 8; i += 256) reinterpret_cast<uint2*>(s_map)[i] = make_uint2(0, 0);
    if (tid == 0) s_nlist = 0;

    // ---- sieve: map rows m0 .. m0+nrows-1 of this wave (9, 9, 8, 8 rows), five rows per step
    const int m0 = wv < 2 ? 9 * wv : 2 + 8 * wv;
    const int nrows = wv < 2 ? 9 : 8;
    // bit i: pixel i is scored: inside the detectable columns and, for the halo lanes, only the
    // one column that neighbours an output column (tile columns 3 and 252, or the band's)
    uint32_t ok_bits = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int xl = 4 * li + i;
        ok_bits |= (uint32_t)(band_ok && xg + i >= ORB_EDGE && xg + i < x_hi && xl >= 3 && xl <= 4 * last_li) << i;
    }
    // the same as a mask over the sieve's verdict word (one byte per pixel)
    const uint32_t ok_mask = ((ok_bits & 1) ? 0xFFu : 0u) | ((ok_bits & 2) ? 0xFF00u : 0u) |
                             ((ok_bits & 4) ? 0xFF0000u : 0u) | ((ok_bits & 8) ? 0xFF000000u : 0u);
    __syncthreads();                                           // map is zero

    uint16_t* queue = s_queue[wv];
    const v2s thr = {(short)min_th, (short)min_th};
    const int y_last = g.h - 1;                                // last row of the level
    // The lane's dword of level row y (wave-uniform: scalar row address + per-lane column and band offset;
    // y >= 15). Rows past the level repeat its last row and lanes past its width read column 0: neither
    // feeds a scored pixel. PACKED (planes only): no clamp, a band's rows end at most 15 rows below the
    // level, inside the plane; lanes of a band that does not exist read band 0's rows.
    const uint32_t coff = col_ok ? (uint32_t)xg + (PACKED ? (uint32_t)(__mul24(band, FT_ROWS) * pitch) : 0u) : 0u;
    // Buffer loads: the row's offset is wave-uniform and travels in a scalar register, the lane's offset is a constant
    // register, so a load costs no vector add. (No range check is wanted from the descriptor: every offset is in
    // range by construction and far below its 2 GB of records.)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(org), 0, 0x7FFFFFFF, 0x00020000);
    auto load_row = [&](int y) -> uint32_t {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)coff, (PACKED ? y : min(y, y_last)) * pitch, 0);
    };
    int pending = 0;                                           // queued pixels not scored yet (wave-uniform, < 128 between steps)
#pragma unroll 1
    for (int grp = 0; grp < 2; grp++) {
        const int mg = m0 + FT_GROUP * grp;                    // first map row of this step
        const int rg = y0 - 1 + mg;                            // its level row
        const int nstep = min(FT_GROUP, nrows - FT_GROUP * grp);
        // one coalesced dword per lane and row; the dwords left and right of it come from the
        // neighbouring lanes (lane 0 / 63 get nothing useful there: their outer pixels are not scored)
        uint32_t ctr[FT_GROUP], lft[FT_GROUP], rgt[FT_GROUP], above[3], below[3];
#pragma unroll
        for (int j = 0; j < FT_GROUP; j++) ctr[j] = load_row(rg + j);
#pragma unroll
        for (int j = 0; j < FT_GROUP; j++) {
            lft[j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ctr[j], 0x138, 0xf, 0xf, true);   // wave_shr:1
            rgt[j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ctr[j], 0x130, 0xf, 0xf, true);   // wave_shl:1
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            above[j] = load_row(rg - 3 + j);
            below[j] = load_row(rg + FT_GROUP + j);
        }
        // copy what was loaded into the image tile (tile rows mg .. mg+10 = level rows rg-3 .. rg+7;
        // other steps and waves write identical bytes to the rows they share)
        {
            uint8_t* t = &s_img[mg * FT_P + 4 * lane];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                *reinterpret_cast<uint32_t*>(t + j * FT_P) = above[j];
                if (mg + 3 + FT_GROUP + j < FT_IMG_ROWS) *reinterpret_cast<uint32_t*>(t + (3 + FT_GROUP + j) * FT_P) = below[j];
            }
#pragma unroll
            for (int j = 0; j < FT_GROUP; j++) *reinterpret_cast<uint32_t*>(t + (3 + j) * FT_P) = ctr[j];
        }
        // unpack the own dword of every loaded row once: pixels 0,1 and 2,3 as 16-bit halves
        v2s ulo[FT_GROUP + 6], uhi[FT_GROUP + 6];              // index t <-> level row rg - 3 + t
#pragma unroll
        for (int t = 0; t < FT_GROUP + 6; t++) {
            const uint32_t v = t < 3 ? above[t < 3 ? t : 0] : t < 3 + FT_GROUP ? ctr[t >= 3 && t < 3 + FT_GROUP ? t - 3 : 0] : below[t >= 3 + FT_GROUP ? t - 3 - FT_GROUP : 0];
            ulo[t] = pick2<0>(v, v);
            uhi[t] = pick2<2>(v, v);
        }
        // With N,S,E,W the ring pixels at distance 3: the centre c can only be a corner at
        // threshold t if  c - max(min(N,S), min(E,W)) > t  (centre brighter than one of each
        // pair) or  min(max(N,S), max(E,W)) - c > t  (darker). The sign bit of
        // t - max(c - A, B - c) is the verdict; the bits of the 5 rows are collected in two
        // words: acc0 bit 15-j / 31-j <-> row j, pixel 0 / 1; acc1 likewise for pixels 2 / 3.
        uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
        for (int j = 0; j < FT_GROUP; j++) {
            const int r = rg + j;
            if (j >= nstep || (!PACKED && (r < ORB_EDGE || r >= y_hi))) continue;   // wave-uniform (PACKED: the rows of the bands differ, see rowmask)
            const v2s e0 = pick2<3>(rgt[j], ctr[j]), e1 = pick2<1>(rgt[j], rgt[j]);     // (+3, 0)
            const v2s w0 = pick2<1>(ctr[j], lft[j]), w1 = pick2<3>(ctr[j], lft[j]);     // (-3, 0)
            const v2s a0 = pmax(pmin(ulo[j + 6], ulo[j]), pmin(e0, w0)), b0 = pmin(pmax(ulo[j + 6], ulo[j]), pmax(e0, w0));
            const v2s a1 = pmax(pmin(uhi[j + 6], uhi[j]), pmin(e1, w1)), b1 = pmin(pmax(uhi[j + 6], uhi[j]), pmax(e1, w1));
            const v2s u0 = thr - pmax(ulo[j + 3] - a0, b0 - ulo[j + 3]);
            const v2s u1 = thr - pmax(uhi[j + 3] - a1, b1 - uhi[j + 3]);
            acc0 |= (v2u(u0) >> j) & (0x80008000u >> j);
            acc1 |= (v2u(u1) >> j) & (0x80008000u >> j);
        }
        // one word: byte i = pixel i, bit 7-j of the byte = row j
        uint32_t verdict = __builtin_amdgcn_perm(acc1, acc0, 0x07050301u) & ok_mask;
        if (PACKED) {
            // rows of this lane's band outside the detectable rows: step row j is level row rg + 32 * band + j
            const int rl = rg + __mul24(band, FT_ROWS);
            const int jlo = min(max(ORB_EDGE - rl, 0), 8), jhi = min(max(y_hi - rl, 0), 8);   // valid j in [jlo, jhi)
            verdict &= (((0xFFu >> jlo) & ~(0xFFu >> jhi)) & 0xFFu) * 0x01010101u;
        }
        // exact scores of the survivors (their rings lie in tile rows this wave has just written)
        // Only full batches of 128 are scored here; the remainder waits in the queue for the next step.
        if (__ballot(verdict != 0)) {
            const uint32_t cnt = __popc(verdict);
            const uint32_t incl = wave_scan_incl(cnt);
            const int total = __builtin_amdgcn_readlane((int)incl, 63);   // wave-uniform by construction: the queue bookkeeping runs on the scalar unit
            const int first = (int)(incl - cnt);
            const uint32_t code0 = ((uint32_t)grp << 11) | ((uint32_t)lane << 5);
            auto drain = [&](int n) {                          // n entries queued: score the full double batches
                const int nfull = n & ~127;
                for (int q = 0; q < nfull; q += 128) fast_score_queue2<PACKED>(queue + q, 128, lane, m0, s_img, min_th, s_map, s_list, &s_nlist, fb);
                pending = n - nfull;
                if (nfull) {                                   // move the remainder (< 128 entries) to the front
                    const uint16_t v0 = queue[nfull + lane], v1 = queue[min(nfull + 64 + lane, FT_QUEUE - 1)];
                    if (lane < pending) queue[lane] = v0;
                    if (lane + 64 < pending) queue[64 + lane] = v1;
                }
            };
            if (pending + total <= FT_QUEUE) {
                fast_fill_queue<false>(verdict, pending + first, 0, code0, queue);
                drain(pending + total);
            } else {
                for (int wb = 0; wb < total;) {
                    const int room = FT_QUEUE - pending;
                    fast_fill_queue<true>(verdict, first - wb + pending, pending, code0, queue);
                    drain(pending + min(room, total - wb));
                    wb += room;
                }
            }
        }
    }
    if (pending > 64) fast_score_queue2<PACKED>(queue, pending, lane, m0, s_img, min_th, s_map, s_list, &s_nlist, fb);
    else if (pending) fast_score_queue<PACKED>(queue, pending, lane, m0, s_img, min_th, s_map, s_list, &s_nlist, fb);
    __syncthreads();                                           // all scores of the tile are in the map

    // ---- non-maximum suppression inside the cell and emission
    out += (size_t)(tile * 4 + wv) * FT_SEGCAP;
    int nout = 0;
    // One scored output pixel per lane: tile column xl, map row m. Neighbours in another cell (or
    // outside the detectable area, where the map is 0) do not count.
    auto nms_emit = [&](bool have, int xl, int m) __attribute__((always_inline)) {
        bool keep = false;
        int x = 0, y = 0, s = 0;
        if (have) {
            const uint8_t* c = &s_map[__mul24(m, FT_P) + xl];
            s = c[0];
            x = xs - 4 + xl; y = y0 - 1 + m;
            if (PACKED) {
                const int bb = (int)(umul24((uint32_t)(xl >> 2), fb.magic) >> 16);   // the pixel's band
                x -= __mul24(bb, 4 * fb.lpb); y += __mul24(bb, FT_ROWS);
            }
            const int ux = x - ORB_EDGE, uy = y - ORB_EDGE;
            // position inside the cell: n / d == n * ceil(2^22 / d) >> 22 for n * d < 2^22 (full-rate multiplies)
            const int xin = ux - __mul24((int)(umul24((uint32_t)ux, g.wcell_m22) >> 22), g.wcell);
            const int yin = uy - __mul24((int)(umul24((uint32_t)uy, g.hcell_m22) >> 22), g.hcell);
            const bool l = xin > 0, r = xin < g.wcell - 1, u = yin > 0, d = yin < g.hcell - 1;
            // all eight reads are in range (the map has a one-pixel halo); issue them together
            const int nw_ = c[-FT_P - 1], n_ = c[-FT_P], ne_ = c[-FT_P + 1], w_ = c[-1], e_ = c[1];
            const int sw_ = c[FT_P - 1], s_ = c[FT_P], se_ = c[FT_P + 1];
            const int mu = max(max(l ? nw_ : 0, n_), r ? ne_ : 0);
            const int md = max(max(l ? sw_ : 0, s_), r ? se_ : 0);
            const int mc = max(l ? w_ : 0, r ? e_ : 0);
            keep = s > max(max(u ? mu : 0, d ? md : 0), mc);
        }
        const unsigned long long km = __ballot(keep);
        if (keep) {
            const int slot = nout + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0));
            if (slot < FT_SEGCAP) out[slot] = make_uint2((uint32_t)(x - ORB_MINB) | ((uint32_t)(y - ORB_MINB) << 16), (uint32_t)s);
        }
        nout += __popcll(km);
    };
    const int nlist = s_nlist;                                 // the same for the whole workgroup
    if (nlist <= FT_LIST) {
        // the listed pixels, 64 at a time, the waves taking turns
        for (int q = 64 * wv; q < nlist; q += 256) {
            const bool have = q + lane < nlist;
            const uint32_t e = have ? s_list[q + lane] : 0u;
            nms_emit(have, (int)(e & 0xFF), (int)(e >> 8));
        }
    } else {
        // dense texture: every wave finds the scored pixels of its 8 map rows 1+8*wv .. 8+8*wv in the
        // map itself (a scored pixel is a non-zero byte) and compacts them through its queue
        const int mh = 1 + 8 * wv;
        uint32_t bits = 0;
        if (band_ok && li >= 1 && li < last_li) {              // halo lanes hold neighbour columns only
            // rows outside the detectable area were never scored: their bytes are zero
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(&s_map[(mh + j) * FT_P + 4 * lane]);
                const uint32_t nz = (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;   // bit 7 of every non-zero byte
                bits |= nz >> (7 - j);                         // bit 8 * pixel + row
            }
        }
        if (__ballot(bits != 0)) {
            const uint32_t cnt = __popc(bits);
            const uint32_t incl = wave_scan_incl(cnt);
            const int total = __builtin_amdgcn_readlane((int)incl, 63);
            const int first = (int)(incl - cnt);
            for (int wb = 0; wb < total; wb += FT_QUEUE) {
                if (total <= FT_QUEUE) fast_fill_queue<false>(bits, first, 0, (uint32_t)lane << 5, queue);
                else fast_fill_queue<true>(bits, first - wb, 0, (uint32_t)lane << 5, queue);
                const int nw = min(FT_QUEUE, total - wb);
                for (int q = 0; q < nw; q += 64) {
                    const bool have = q + lane < nw;
                    const uint32_t e = have ? queue[q + lane] : 0u;       // lane << 5 | 8 * pixel + row
                    nms_emit(have, (int)((e >> 3) & 0xFC) + (int)((e >> 3) & 3), mh + (int)(e & 7));
                }
            }
        }
    }
    if (lane == 0) {
        if (nout > FT_SEGCAP) { atomicOr(status, 1); nout = FT_SEGCAP; }
        out_count[tile * 4 + wv] = nout;
    }
}

// Grid: one dimension, all plain (PACKED: all packed) tiles of all levels of all frames, level-major, then
// frame, tile row, tile column: the large levels start first. Plain and packed tiles are separate
// launches (one kernel holding both variants ran the plain one 7 % slower). Output: segment (tile, wave)
// of FT_SEGCAP entries {x | y << 16 (relative to the detection window origin (16,16)), score} and its
// entry count; the packed tiles of a level come behind its plain ones.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_fast(const uint8_t* __restrict__ pyr, const PlanDev* __restrict__ plan, int frames,
                                              uint2* __restrict__ seg, int* __restrict__ seg_count,
                                              int* __restrict__ status, unsigned id_base, Level0View v0) {
    __shared__ __align__(16) uint8_t s_map[FT_MAP_ROWS * FT_P];
    __shared__ __align__(16) uint8_t s_img[FT_IMG_ROWS * FT_P];
    __shared__ uint16_t s_queue[4][FT_QUEUE];
    __shared__ uint16_t s_list[FT_LIST];                       // scored output pixels of the tile
    __shared__ int s_nlist;
    // which tile: everything here is uniform over the workgroup (scalar unit)
    // The levels take consecutive ranges of the dispatch order; inside its level's range a workgroup
    // is remapped so that neighbouring tiles (shared halo rows / lines) share an L2 and every level
    // is spread over all XCDs (tiles of different levels do not cost the same).
    const unsigned id = id_base + blockIdx.x;                  // a launch may cover a sub-range of the levels
    const uint32_t* cum = plan->fast_cum[PACKED ? 1 : 0];
    int level = 0;
    while (level + 1 < plan->nlevels && id >= cum[level + 1] * (unsigned)frames) level++;
    const LevelGeom g = plan->lv[level];
    const int nplain = g.fast_ntx_plain * g.nty;
    const int ntiles = PACKED ? g.fast_ntiles - nplain : nplain;   // of this kind, per frame
    const unsigned rem = xcd_contiguous(id - cum[level] * (unsigned)frames, (unsigned)(ntiles * frames));
    const unsigned mt = plan->fast_tiles_magic[PACKED ? 1 : 0][level], mx = plan->fast_ntx_magic[level];
    const int f = (int)(mt ? __umulhi(rem, mt) : rem);
    const int t = (int)(rem - (unsigned)f * (unsigned)ntiles);
    // where the level's pixel (0, 0) of this frame is: in its plane, or (level 0 read in place) in the caller's image
    const bool inplace = level == 0 && v0.org != nullptr;
    const int pitch = inplace ? v0.pitch : g.pitch;
    const uint8_t* org = inplace ? v0.org + (size_t)f * v0.frame_bytes
                                 : pyr + g.pyr_off + (size_t)f * g.plane_bytes + (size_t)ORB_OY * g.pitch + ORB_OX;
    const int xlimit = inplace ? v0.xlim : g.pitch - ORB_OX;
    uint2* out = seg + (size_t)f * plan->seg_per_frame + g.seg_off;
    int* out_count = seg_count + (size_t)f * plan->segc_per_frame + g.segc_off;
    if (!PACKED) {
        const int by = (int)(mx ? __umulhi((unsigned)t, mx) : (unsigned)t);
        const int bx = t - by * g.fast_ntx_plain;
        fast_tile<false>(g, org, pitch, xlimit, ORB_MINB + FT_COLS * bx, ORB_EDGE + FT_ROWS * by, t, FastBands{64, 1, 0u},
                         plan->min_th, out, out_count, status, s_map, s_img, s_queue, s_list, &s_nlist);
    } else {
        const int first = t * g.fast_bands;                    // first tile row of this packed tile
        const FastBands fb = {g.fast_lpb, min(g.fast_bands, g.nty - first), (65536u + (unsigned)g.fast_lpb - 1u) / (unsigned)g.fast_lpb};
        fast_tile<true>(g, org, pitch, xlimit, ORB_MINB + FT_COLS * (g.ntx - 1), ORB_EDGE + FT_ROWS * first, nplain + t, fb,
                        plan->min_th, out, out_count, status, s_map, s_img, s_queue, s_list, &s_nlist);
    }
}

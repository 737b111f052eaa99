// FAST-9/16 corner detection for one pyramid level, streaming form (src/ORBextractor.cc:789-829 and
// cv::FAST(img, kps, threshold, true) behind it).
//
// A wave owns a strip of 62 four-pixel groups (248 output columns; lanes 0 and 63 only supply
// score neighbours) and walks down a band of 64 rows. Each lane keeps a 7-row x 12-byte window of
// the image in registers (one 12-byte load per lane per row), forms the 16 ring differences of two
// horizontally adjacent pixels at a time as packed 16-bit lanes (v_perm_b32 + v_pk_sub_i16) and
// evaluates the corner score with a packed min/max network (v_pk_min_i16 / v_pk_max_i16). No LDS is
// used for the image; LDS only stages the sparse output of a wave.
//
// Score S(p) = max over the 16 nine-pixel arcs of min(centre - ring) (centre brighter) or
// min(ring - centre) (centre darker), minus 1. p is a FAST-9 corner at threshold t iff S(p) >= t and
// cv::FAST's cornerScore of a corner equals S(p) whatever threshold found it. In the reference
// every ~30-px cell is a separate cv::FAST call, first at iniThFAST and, when that yields nothing,
// at minThFAST. Because S does not depend on t this is equivalent to: p survives iff
// S(p) >= minTh and S(p) is strictly greater than the S of its 8-neighbours inside the same cell;
// a survivor is used iff S(p) >= iniTh or its cell has no survivor with S >= iniTh. This kernel
// emits all survivors and counts the survivors >= iniTh per cell; the quadtree kernel applies the
// last rule when it reads the list.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "orb_device.h"

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s u2v(uint32_t u) { return __builtin_bit_cast(v2s, u); }
__device__ __forceinline__ uint32_t v2u(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ v2s pmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s pmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }

// Bresenham circle of radius 3, in cv::FAST's order.
constexpr int kRingDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
constexpr int kRingDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// Bytes O and O+1 of a 12-byte window row, zero-extended into the two halves of a dword.
template <int O>
__device__ __forceinline__ v2s pick2(const uint32_t (&row)[3]) {
    static_assert(O >= 0 && O + 1 <= 11, "window offset");
    if constexpr (O + 1 <= 7) {
        constexpr uint32_t sel = (uint32_t)O | (0x0Cu << 8) | ((uint32_t)(O + 1) << 16) | (0x0Cu << 24);
        return u2v(__builtin_amdgcn_perm(row[1], row[0], sel));
    } else {
        constexpr uint32_t sel = (uint32_t)(O - 4) | (0x0Cu << 8) | ((uint32_t)(O - 3) << 16) | (0x0Cu << 24);
        return u2v(__builtin_amdgcn_perm(row[2], row[1], sel));
    }
}

// d[k] = centre - ring[k] for the pixel pair PAIR (pixels 2*PAIR, 2*PAIR+1 of the lane's group).
template <int PAIR, size_t... K>
__device__ __forceinline__ void ring_diffs(const uint32_t (&W)[7][3], v2s (&d)[16], std::index_sequence<K...>) {
    const v2s c = pick2<4 + 2 * PAIR>(W[3]);
    ((d[K] = c - pick2<4 + 2 * PAIR + kRingDX[K]>(W[3 + kRingDY[K]])), ...);
}

// Necessary condition for a corner at threshold t: every 9-arc contains one pixel of each opposite
// pair, so all 8 pairs need a member with d > t (centre brighter) or all need one with d < -t.
// Returns, per half, max(min_j max(d_j,d_j+8), -max_j min(d_j,d_j+8)): the pixel can only be a
// corner at t if that value is > t.
__device__ __forceinline__ v2s fast_pretest(const v2s (&d)[16]) {
    v2s lo = pmax(d[0], d[8]), hi = pmin(d[0], d[8]);
#pragma unroll
    for (int j = 1; j < 8; j++) {
        lo = pmin(lo, pmax(d[j], d[j + 8]));
        hi = pmax(hi, pmin(d[j], d[j + 8]));
    }
    const v2s zero = {0, 0};
    return pmax(lo, zero - hi);
}

__device__ __forceinline__ v2s fast_score_packed(const v2s (&d)[16]) {
    v2s lo2[8], hi2[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {      // d[i], d[i+1] for odd i = 2j+1
        lo2[j] = pmin(d[(2 * j + 1) & 15], d[(2 * j + 2) & 15]);
        hi2[j] = pmax(d[(2 * j + 1) & 15], d[(2 * j + 2) & 15]);
    }
    v2s lo4[8], hi4[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        lo4[j] = pmin(lo2[j], lo2[(j + 1) & 7]);
        hi4[j] = pmax(hi2[j], hi2[(j + 1) & 7]);
    }
    v2s a0 = {-256, -256}, b0 = {256, 256};
#pragma unroll
    for (int j = 0; j < 8; j++) {      // arcs [k..k+8] and [k+1..k+9], k = 2j, share d[k+1..k+8]
        const int k = 2 * j;
        const v2s lo8 = pmin(lo4[j], lo4[(j + 2) & 7]);
        const v2s hi8 = pmax(hi4[j], hi4[(j + 2) & 7]);
        a0 = pmax(a0, pmin(lo8, pmax(d[k], d[(k + 9) & 15])));
        b0 = pmin(b0, pmax(hi8, pmin(d[k], d[(k + 9) & 15])));
    }
    const v2s zero = {0, 0}, one = {1, 1};
    return pmax(a0, zero - b0) - one;
}

#define F2_RB 64          // rows per band (one wave)
#define F2_COLS 248       // output columns per strip
#define F2_STAGE 512      // staged candidates per wave before a flush

struct __attribute__((packed, aligned(4))) Row12 { uint32_t a, b, c; };

__device__ __forceinline__ void fast_flush(uint2* stage, int n, uint2* out, int* counter, int cap, int* status, int lane) {
    if (n == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(counter, n);
    base = __shfl(base, 0, 64);
    for (int i = lane; i < n; i += 64) {
        if (base + i < cap) out[base + i] = stage[i];
        else atomicOr(status, 1);
    }
}

__global__ __launch_bounds__(256) void k_fast(const uint8_t* __restrict__ pyr, LevelGeom g, int level,
                                              int nlevels, int ini_th, int min_th, int cells_per_frame,
                                              unsigned long long cand_per_frame, int nbands,
                                              uint2* __restrict__ cand, int* __restrict__ cand_count,
                                              int* __restrict__ cell_ini, int* __restrict__ status) {
    __shared__ uint2 s_stage[4][F2_STAGE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int band = blockIdx.y * 4 + wv;
    const int f = blockIdx.z;
    if (band >= nbands) return;                      // the kernel has no workgroup barrier
    const int xg = ORB_MINB + F2_COLS * (int)blockIdx.x - 4 + 4 * lane;   // level x of this lane's pixel 0
    const int col = ORB_OX + xg;                     // device column, multiple of 4
    const bool col_ok = col + 8 <= g.pitch;          // the lane's 12-byte window [col-4, col+8) exists
    const int yb = ORB_EDGE + F2_RB * band;          // first row of the band
    const int x_hi = g.w - ORB_EDGE, y_hi = g.h - ORB_EDGE;
    const uint8_t* P = pyr + g.pyr_off + (size_t)f * g.plane_bytes + (size_t)ORB_OY * g.pitch + (col - 4);
    uint2* stage = s_stage[wv];
    uint2* out = cand + (size_t)f * cand_per_frame + g.cand_off;
    int* counter = cand_count + f * nlevels + level;
    int* cells = cell_ini + (size_t)f * cells_per_frame + g.cell_off;

    // per-pixel constants of this lane's four columns
    // bit i: pixel i lies in the detectable area / has a same-cell left / right neighbour
    uint32_t ok_bits = 0, left_bits = 0, right_bits = 0;
    const bool emit = lane >= 1 && lane <= 62;       // lanes 0 and 63 only provide neighbour scores
    int cellx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = xg + i;
        const bool ok = x >= ORB_EDGE && x < x_hi;
        const int ux = ok ? x - ORB_EDGE : 0;
        const int cx = ux / g.wcell;
        const int xin = ux - cx * g.wcell;
        cellx[i] = cx;
        ok_bits |= (uint32_t)ok << i;
        left_bits |= (uint32_t)(xin > 0) << i;
        right_bits |= (uint32_t)(xin < g.wcell - 1) << i;
    }

    auto load_row = [&](int y, uint32_t (&dst)[3]) {
        if (col_ok) {
            const Row12 v = *reinterpret_cast<const Row12*>(P + (ptrdiff_t)y * g.pitch);
            dst[0] = v.a; dst[1] = v.b; dst[2] = v.c;
        } else {
            dst[0] = dst[1] = dst[2] = 0;
        }
    };

    uint32_t W[7][3];
    const int r0 = yb - 1;                            // first score row (apron of the band)
    W[0][0] = W[0][1] = W[0][2] = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) load_row(r0 - 3 + j, W[j + 1]);
    uint32_t scl[3] = {0, 0, 0}, sc[3] = {0, 0, 0}, scr[3] = {0, 0, 0};   // score rows r-2, r-1, r
    int nstage = 0;
    const int r_end = min(yb + F2_RB, y_hi);          // last score row needed (apron below the band)
    for (int r = r0; r <= r_end; r++) {
#pragma unroll
        for (int j = 0; j < 6; j++) { W[j][0] = W[j + 1][0]; W[j][1] = W[j + 1][1]; W[j][2] = W[j + 1][2]; }
        load_row(r + 3, W[6]);
        // ---- scores of row r
        uint32_t snew = 0;
        if (r >= ORB_EDGE && r < y_hi) {
            v2s d0[16], d1[16];
            ring_diffs<0>(W, d0, std::make_index_sequence<16>());
            ring_diffs<1>(W, d1, std::make_index_sequence<16>());
            const v2s t0 = fast_pretest(d0), t1 = fast_pretest(d1);
            const int tmax = max(max((int)t0.x, (int)t0.y), max((int)t1.x, (int)t1.y));
            if (__ballot(tmax > min_th && ok_bits != 0)) {
                const v2s S0 = fast_score_packed(d0), S1 = fast_score_packed(d1);
                const int s0 = S0.x, s1 = S0.y, s2 = S1.x, s3 = S1.y;
                snew = ((s0 >= min_th && (ok_bits & 1)) ? (uint32_t)s0 : 0u) |
                       ((s1 >= min_th && (ok_bits & 2)) ? (uint32_t)s1 << 8 : 0u) |
                       ((s2 >= min_th && (ok_bits & 4)) ? (uint32_t)s2 << 16 : 0u) |
                       ((s3 >= min_th && (ok_bits & 8)) ? (uint32_t)s3 << 24 : 0u);
            }
        }
        scl[0] = scl[1]; scl[1] = scl[2]; sc[0] = sc[1]; sc[1] = sc[2]; scr[0] = scr[1]; scr[1] = scr[2];
        sc[2] = snew;
        if (__ballot(snew != 0)) {
            scl[2] = __shfl_up(snew, 1, 64) >> 24;    // rightmost score of the group to the left
            scr[2] = __shfl_down(snew, 1, 64) & 0xFF; // leftmost score of the group to the right
        } else {
            scl[2] = scr[2] = 0;
        }
        // ---- non-maximum suppression of row r-1 inside its cell, emission
        const int yn = r - 1;
        if (yn >= yb && yn < y_hi && __ballot(sc[1] != 0)) {
            const int uy = yn - ORB_EDGE;
            const int cy = uy / g.hcell, yin = uy - cy * g.hcell;
            const bool up = yin > 0, dn = yin < g.hcell - 1;
            // rows as byte strings: [left neighbour, own 4, right neighbour]
            const unsigned long long x0 = scl[0] | ((unsigned long long)sc[0] << 8) | ((unsigned long long)scr[0] << 40);
            const unsigned long long x1 = scl[1] | ((unsigned long long)sc[1] << 8) | ((unsigned long long)scr[1] << 40);
            const unsigned long long x2 = scl[2] | ((unsigned long long)sc[2] << 8) | ((unsigned long long)scr[2] << 40);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int s = (sc[1] >> (8 * i)) & 0xFF;
                if (!__ballot(s != 0)) continue;
                const bool l = (left_bits >> i) & 1, rr = (right_bits >> i) & 1;
                int m = 0;
                const int a0 = (int)(x0 >> (8 * i)) & 0xFF, a1 = (int)(x0 >> (8 * i + 8)) & 0xFF, a2 = (int)(x0 >> (8 * i + 16)) & 0xFF;
                const int b0 = (int)(x1 >> (8 * i)) & 0xFF, b2 = (int)(x1 >> (8 * i + 16)) & 0xFF;
                const int c0 = (int)(x2 >> (8 * i)) & 0xFF, c1 = (int)(x2 >> (8 * i + 8)) & 0xFF, c2 = (int)(x2 >> (8 * i + 16)) & 0xFF;
                if (up) { m = a1; if (l) m = max(m, a0); if (rr) m = max(m, a2); }
                if (l) m = max(m, b0);
                if (rr) m = max(m, b2);
                if (dn) { m = max(m, c1); if (l) m = max(m, c0); if (rr) m = max(m, c2); }
                const bool keep = emit && s != 0 && s > m;
                const unsigned long long km = __ballot(keep);
                if (!km) continue;
                if (keep) {
                    const int slot = nstage + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0));
                    stage[slot] = make_uint2((uint32_t)(xg + i - ORB_MINB) | ((uint32_t)(yn - ORB_MINB) << 16), (uint32_t)s);
                    if (s >= ini_th) atomicAdd(&cells[cy * g.ncols + cellx[i]], 1);
                }
                nstage += __popcll(km);
            }
            if (nstage > F2_STAGE - 256) {
                fast_flush(stage, nstage, out, counter, g.cand_cap, status, lane);
                nstage = 0;
            }
        }
    }
    fast_flush(stage, nstage, out, counter, g.cand_cap, status, lane);
}

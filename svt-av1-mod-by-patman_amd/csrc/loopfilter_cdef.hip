// loopfilter_cdef.hip — CDEF on gfx950: per-64x64 filter-block strength search and frame apply,
// plus the ABI-identical per-call entry points.  Replaces cdef_seg_search (cdef_process.c:106-349),
// svt_av1_cdef_frame (enc_cdef.c:284-610) and their leaves (cdef.c:150-307, enc_cdef.c:23-219).
//
// One workgroup owns one filter block of one plane: the (64+16) x (64+6) input tile is staged once in LDS as
// uint16 with CDEF_VERY_LARGE outside the picture, direction/variance of the 8x8 blocks are found once, and all
// strength candidates are evaluated from that tile — filtered samples never go to memory during the search, only
// the per-block distortion sums do (LDS atomics), so the search reads each picture sample exactly once from HBM.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_lf.h"
#include "common.hpp"

using namespace svthip;

namespace {

constexpr int BS = SVT_HIP_CDEF_BSTRIDE, VL = SVT_HIP_CDEF_VERY_LARGE, VB = 3, HB = 8;
constexpr int TILE_ROWS = 64 + 2 * VB;

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int msb(unsigned n) { return 31 - __builtin_clz(n); }

__device__ const int16_t DIRS_D[12][2] = {{1 * BS + 0, 2 * BS + 0},  {1 * BS + 0, 2 * BS - 1}, {-1 * BS + 1, -2 * BS + 2},
                                          {0 * BS + 1, -1 * BS + 2}, {0 * BS + 1, 0 * BS + 2}, {0 * BS + 1, 1 * BS + 2},
                                          {1 * BS + 1, 2 * BS + 2},  {1 * BS + 0, 2 * BS + 1}, {1 * BS + 0, 2 * BS + 0},
                                          {1 * BS + 0, 2 * BS - 1},  {-1 * BS + 1, -2 * BS + 2}, {0 * BS + 1, -1 * BS + 2}};

// the batched kernels keep their tile at a tighter pitch than the reference's CDEF_BSTRIDE (two copies of the tile have to fit
// LDS several times over): same direction table, other row pitch
constexpr int TS = 64 + 2 * HB + 8;  // 88 samples per tile row
__device__ const int16_t DIRS_T[12][2] = {{1 * TS + 0, 2 * TS + 0},  {1 * TS + 0, 2 * TS - 1}, {-1 * TS + 1, -2 * TS + 2},
                                          {0 * TS + 1, -1 * TS + 2}, {0 * TS + 1, 0 * TS + 2}, {0 * TS + 1, 1 * TS + 2},
                                          {1 * TS + 1, 2 * TS + 2},  {1 * TS + 0, 2 * TS + 1}, {1 * TS + 0, 2 * TS + 0},
                                          {1 * TS + 0, 2 * TS - 1},  {-1 * TS + 1, -2 * TS + 2}, {0 * TS + 1, -1 * TS + 2}};

__device__ __forceinline__ int32_t constrain(int32_t diff, int32_t threshold, int32_t damping) {
    if (!threshold)
        return 0;
    const int32_t shift = imax(0, damping - msb((unsigned)threshold));
    const int32_t ad    = diff < 0 ? -diff : diff;
    const int32_t m     = imin(ad, imax(0, threshold - (ad >> shift)));
    return diff < 0 ? -m : m;
}
__device__ __forceinline__ int adjust_strength(int strength, int var) {
    const int i = (var >> 6) ? imin(msb((unsigned)(var >> 6)), 12) : 0;
    return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

// One output sample of svt_cdef_filter_block_c (cdef.c:253-307); `p` points at the sample inside a stride-144 tile.
__device__ __forceinline__ int32_t cdef_pixel(const uint16_t *p, int pri_strength, int sec_strength, int dir, int pri_damping,
                                              int sec_damping, int coeff_shift) {
    const int     tsel = (pri_strength >> coeff_shift) & 1;
    const int32_t pt0 = tsel ? 3 : 4, pt1 = tsel ? 3 : 2;
    const int16_t x   = (int16_t)p[0];
    int16_t       sum = 0;
    int32_t       mx = x, mn = x;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int32_t pt = k ? pt1 : pt0, st = k ? 1 : 2;
        const int     o0 = DIRS_D[dir + 2][k], o1 = DIRS_D[dir + 4][k], o2 = DIRS_D[dir][k];
        const int16_t p0 = (int16_t)p[o0], p1 = (int16_t)p[-o0];
        sum = (int16_t)(sum + (int16_t)(pt * constrain(p0 - x, pri_strength, pri_damping)));
        sum = (int16_t)(sum + (int16_t)(pt * constrain(p1 - x, pri_strength, pri_damping)));
        if (p0 != VL) mx = imax(p0, mx);
        if (p1 != VL) mx = imax(p1, mx);
        mn = imin(p0, mn), mn = imin(p1, mn);
        const int16_t s0 = (int16_t)p[o1], s1 = (int16_t)p[-o1], s2 = (int16_t)p[o2], s3 = (int16_t)p[-o2];
        if (s0 != VL) mx = imax(s0, mx);
        if (s1 != VL) mx = imax(s1, mx);
        if (s2 != VL) mx = imax(s2, mx);
        if (s3 != VL) mx = imax(s3, mx);
        mn = imin(s0, mn), mn = imin(s1, mn), mn = imin(s2, mn), mn = imin(s3, mn);
        sum = (int16_t)(sum + (int16_t)(st * constrain(s0 - x, sec_strength, sec_damping)));
        sum = (int16_t)(sum + (int16_t)(st * constrain(s1 - x, sec_strength, sec_damping)));
        sum = (int16_t)(sum + (int16_t)(st * constrain(s2 - x, sec_strength, sec_damping)));
        sum = (int16_t)(sum + (int16_t)(st * constrain(s3 - x, sec_strength, sec_damping)));
    }
    int32_t y = (int32_t)x + ((8 + sum - (sum < 0)) >> 4);
    return y < mn ? mn : (y > mx ? mx : y);
}

// ---- two horizontally adjacent samples at once, in packed 16-bit arithmetic (v_pk_*_i16).  Every quantity of
// svt_cdef_filter_block_c is an int16 by construction (samples, CDEF_VERY_LARGE = 30000, differences, the int16 `sum`), so
// the packed evaluation is the reference's arithmetic lane by lane.  `p0` points at the even-indexed first sample of the
// pair in the tile, `p1` at the same position of the tile shifted left by one sample: a tap at an odd offset is read from
// the shifted copy, so that every tap pair is one aligned 32-bit LDS read.
typedef short          s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 splat2(int v) { return (s16x2)((short)v); }
__device__ __forceinline__ s16x2 ldpair(const uint16_t *p0, const uint16_t *p1, int o) {
    return __builtin_bit_cast(s16x2, *(const uint32_t *)((o & 1) ? p1 + (o - 1) : p0 + o));
}
// constrain() for a pair; threshold 0 gives 0 without a branch (min(|d|, max(0, 0 - (|d| >> shift))) = 0)
__device__ __forceinline__ s16x2 constrain2(s16x2 d, int threshold, int shift) {
    const s16x2 ad = __builtin_elementwise_max(d, -d);
    s16x2       t  = splat2(threshold) - (s16x2)((u16x2)ad >> (u16x2)((unsigned short)shift));
    t              = __builtin_elementwise_max(t, splat2(0));
    const s16x2 m  = __builtin_elementwise_min(ad, t);
    const s16x2 sg = d >> (s16x2)((short)15);
    return (m ^ sg) - sg;
}
__device__ __forceinline__ int constrain_shift(int threshold, int damping) { return threshold ? imax(0, damping - msb((unsigned)threshold)) : 0; }
__device__ __forceinline__ s16x2 cdef_pair(const uint16_t *p0, const uint16_t *p1, int pri_strength, int sec_strength, int dir, int pri_damping,
                                           int sec_damping, int coeff_shift) {
    const int   tsel = (pri_strength >> coeff_shift) & 1;
    const int   shp = constrain_shift(pri_strength, pri_damping), shs = constrain_shift(sec_strength, sec_damping);
    const s16x2 x = __builtin_bit_cast(s16x2, *(const uint32_t *)p0);
    s16x2       sum = splat2(0), mx = x, mn = x;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int pt = k ? (tsel ? 3 : 2) : (tsel ? 3 : 4), st = k ? 1 : 2;
        const int o0 = DIRS_T[dir + 2][k], o1 = DIRS_T[dir + 4][k], o2 = DIRS_T[dir][k];
        const s16x2 t[6] = {ldpair(p0, p1, o0), ldpair(p0, p1, -o0), ldpair(p0, p1, o1), ldpair(p0, p1, -o1), ldpair(p0, p1, o2), ldpair(p0, p1, -o2)};
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const s16x2 c = q < 2 ? constrain2(t[q] - x, pri_strength, shp) : constrain2(t[q] - x, sec_strength, shs);
            sum += splat2(q < 2 ? pt : st) * c;
            // CDEF_VERY_LARGE (bit 14 set, no sample has it) does not take part in the maximum: clear those lanes
            const s16x2 vl = -(s16x2)((u16x2)t[q] >> (u16x2)((unsigned short)14));
            mx = __builtin_elementwise_max(mx, t[q] & ~vl);
            mn = __builtin_elementwise_min(mn, t[q]);
        }
    }
    const s16x2 neg = (s16x2)((u16x2)sum >> (u16x2)((unsigned short)15));  // sum < 0
    s16x2       y   = x + ((splat2(8) + sum - neg) >> (s16x2)((short)4));
    y               = __builtin_elementwise_max(y, mn);
    return __builtin_elementwise_min(y, mx);
}

// ---- several strengths over one set of taps (the search) -------------------------------------------------------------
// Everything about a sample pair that does not depend on the strength is formed once: the twelve tap pairs, |tap - x|, the
// sign folded into the tap weight (unit weights for the primary taps, whose two weights depend on the strength), and the
// clamp range.  A strength then costs five packed instructions per tap: shift, subtract, max 0, min, multiply-add.
struct PairTaps {
    s16x2 x, mn, mx;
    s16x2 ad[12];  // |tap - x|;                index = 6 * distance + q, q < 2 primary, q >= 2 secondary
    s16x2 w[12];   // primary: +-1; secondary: +-{2, 1} by distance
};
__device__ __forceinline__ void pair_taps(PairTaps &T, const uint16_t *p0, const uint16_t *p1, int dir) {
    T.x  = __builtin_bit_cast(s16x2, *(const uint32_t *)p0);
    T.mx = T.x, T.mn = T.x;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int   o0 = DIRS_T[dir + 2][k], o1 = DIRS_T[dir + 4][k], o2 = DIRS_T[dir][k];
        const s16x2 t[6] = {ldpair(p0, p1, o0), ldpair(p0, p1, -o0), ldpair(p0, p1, o1), ldpair(p0, p1, -o1), ldpair(p0, p1, o2), ldpair(p0, p1, -o2)};
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const s16x2 d  = t[q] - T.x;
            const s16x2 sg = d >> (s16x2)((short)15);
            T.ad[6 * k + q] = __builtin_elementwise_max(d, -d);
            T.w[6 * k + q]  = q < 2 ? (sg | splat2(1)) : ((splat2(k ? 1 : 2) ^ sg) - sg);
            const s16x2 vl = -(s16x2)((u16x2)t[q] >> (u16x2)((unsigned short)14));  // CDEF_VERY_LARGE takes no part in the maximum
            T.mx = __builtin_elementwise_max(T.mx, t[q] & ~vl);
            T.mn = __builtin_elementwise_min(T.mn, t[q]);
        }
    }
}
// thr_p / shp / pt0 / pt1: primary threshold, its shift and the two primary tap weights (splat); thr_s / shs: secondary
__device__ __forceinline__ s16x2 pair_eval(const PairTaps &T, s16x2 thr_p, s16x2 shp, s16x2 pt0, s16x2 pt1, int thr_s, int shs) {
    s16x2 prim[2] = {splat2(0), splat2(0)}, sum = splat2(0);
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const s16x2 a = T.ad[6 * k + q];
            s16x2       t = q < 2 ? thr_p - (s16x2)((u16x2)a >> __builtin_bit_cast(u16x2, shp))
                                  : splat2(thr_s) - (s16x2)((u16x2)a >> (u16x2)((unsigned short)shs));
            t             = __builtin_elementwise_min(a, __builtin_elementwise_max(t, splat2(0)));
            if (q < 2)
                prim[k] += T.w[6 * k + q] * t;
            else
                sum += T.w[6 * k + q] * t;
        }
    sum += pt0 * prim[0] + pt1 * prim[1];
    const s16x2 neg = (s16x2)((u16x2)sum >> (u16x2)((unsigned short)15));  // sum < 0
    s16x2       y   = T.x + ((splat2(8) + sum - neg) >> (s16x2)((short)4));
    y               = __builtin_elementwise_max(y, T.mn);
    return __builtin_elementwise_min(y, T.mx);
}

// svt_aom_cdef_find_dir_c (cdef.c:150-210) for one 8x8 block, one thread.
__device__ int find_dir_block(const uint16_t *img, int stride, int32_t *var, int coeff_shift) {
    const int32_t div_table[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
    int32_t       cost[8], partial[8][15];
    for (int a = 0; a < 8; a++) {
        cost[a] = 0;
        for (int b = 0; b < 15; b++) partial[a][b] = 0;
    }
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const int32_t x = (img[i * stride + j] >> coeff_shift) - 128;
            partial[0][i + j] += x;
            partial[1][i + j / 2] += x;
            partial[2][i] += x;
            partial[3][3 + i - j / 2] += x;
            partial[4][7 + i - j] += x;
            partial[5][3 - i / 2 + j] += x;
            partial[6][j] += x;
            partial[7][i / 2 + j] += x;
        }
    for (int i = 0; i < 8; i++) {
        cost[2] += partial[2][i] * partial[2][i];
        cost[6] += partial[6][i] * partial[6][i];
    }
    cost[2] *= div_table[8];
    cost[6] *= div_table[8];
    for (int i = 0; i < 7; i++) {
        cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
        cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
    }
    cost[0] += partial[0][7] * partial[0][7] * div_table[8];
    cost[4] += partial[4][7] * partial[4][7] * div_table[8];
    for (int i = 1; i < 8; i += 2) {
        for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
        cost[i] *= div_table[8];
        for (int j = 0; j < 3; j++)
            cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
    }
    int32_t best = 0;
    int     dir  = 0;
    for (int i = 0; i < 8; i++)
        if (cost[i] > best)
            best = cost[i], dir = i;
    *var = (best - cost[(dir + 4) & 7]) >> 10;
    return dir;
}

// enc_cdef.c:23-48: the luma distortion of one 8xN block from its five sums (IEEE double, no contraction).
__device__ __forceinline__ uint64_t dist_8xn(uint64_t ss, uint64_t sd, uint64_t ss2, uint64_t sd2, uint64_t ssd, int coeff_shift) {
    const uint64_t svar = ss2 - ((ss * ss + 32) >> 6);
    const uint64_t dvar = sd2 - ((sd * sd + 32) >> 6);
    const double   num  = (double)(sd2 + ss2 - 2 * ssd) * .5 * (double)(svar + dvar + (uint64_t)(400 << 2 * coeff_shift));
    const double   den  = sqrt((double)(20000 << 4 * coeff_shift) + (double)svar * (double)dvar);
    return (uint64_t)floor(.5 + num / den);
}

__device__ __forceinline__ uint32_t load_px(const void *p, size_t idx, int is16) {
    return is16 ? ((const __attribute__((address_space(1))) uint16_t *)p)[idx] : ((const __attribute__((address_space(1))) uint8_t *)p)[idx];  // pictures are global memory: no flat loads
}

struct FbLds {
    __attribute__((aligned(4))) uint16_t tile[TILE_ROWS * TS];
    __attribute__((aligned(4))) uint16_t tile1[TILE_ROWS * TS];  // tile shifted left by one sample (cdef_pair)
    SvtHipCdefList dl[64];
    uint8_t        dir[64];
    int32_t        var[64];
    int            n;
};
struct SearchLds {  // search only: kept apart so that the apply kernel's LDS footprint stays small
    __attribute__((aligned(4))) uint16_t src[64 * 64];  // source samples of the filter block
    uint32_t           bsum[4][64][5];                  // per-(strength of the group, block) sums of the luma distortion
    unsigned long long total[4];                        // one per strength of the current group
};

// stage tile + dlist; returns cdef_count (uniform).  Contains barriers.
__device__ int stage_fb(FbLds &S, const SvtHipCdefPlane &pl, const uint8_t *filt, int fbx, int fby) {
    const int lw = (int)pl.width << pl.xdec, lh = (int)pl.height << pl.ydec;
    const int w8 = (lw + 7) / 8, h8 = (lh + 7) / 8;
    const int bw = 64 >> pl.xdec, bh = 64 >> pl.ydec;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int r = 0; r < 8 && fby * 8 + r < h8; r++)
            for (int c = 0; c < 8 && fbx * 8 + c < w8; c++)
                if (filt[(size_t)(fby * 8 + r) * w8 + fbx * 8 + c])
                    S.dl[n].by = (uint8_t)r, S.dl[n].bx = (uint8_t)c, n++;
        S.n = n;
    }
    for (int idx = threadIdx.x; idx < (bh + 2 * VB) * (bw + 2 * HB); idx += blockDim.x) {
        const int ry = idx / (bw + 2 * HB), rx = idx - ry * (bw + 2 * HB);
        const int py = fby * bh + ry - VB, px = fbx * bw + rx - HB;
        uint16_t  v  = VL;
        if (py >= 0 && px >= 0 && py < (int)pl.height && px < (int)pl.width)
            v = (uint16_t)load_px(pl.recon, (size_t)py * pl.recon_stride + px, pl.is_16bit);
        S.tile[ry * TS + rx] = v;
        if (ry * TS + rx)
            S.tile1[ry * TS + rx - 1] = v;
    }
    __syncthreads();
    return S.n;
}

__global__ __launch_bounds__(256, 4) void cdef_search_kernel(SvtHipCdefPlane pl, const uint8_t *__restrict__ filt,
                                                          SvtHipCdefSearchParams prm, uint64_t *__restrict__ mse,
                                                          uint8_t *__restrict__ gdir, int32_t *__restrict__ gvar, int nhfb) {
    __shared__ FbLds S;
    __shared__ SearchLds Q;
    const int fb = blockIdx.x, fbx = fb % nhfb, fby = fb / nhfb;
    const int n  = stage_fb(S, pl, filt, fbx, fby);
    if (n == 0)
        return;
    const uint16_t *in = S.tile + VB * TS + HB;
    const int       bwl = 3 - pl.xdec, bhl = 3 - pl.ydec, bw = 1 << bwl, bh = 1 << bhl;
    const int       bsize = pl.ydec ? (pl.xdec ? 0 : 2) : (pl.xdec ? 1 : 3);
    int             sub   = prm.subsampling_factor;
    sub                   = bsize == 3 ? imin(sub, 4) : (bsize == 0 ? imin(sub, 1) : imin(sub, 2));
    // direction / variance per 8x8 block
    if ((int)threadIdx.x < n) {
        const int by = S.dl[threadIdx.x].by, bx = S.dl[threadIdx.x].bx;
        if (pl.pli == 0) {
            int32_t   v;
            const int d = find_dir_block(in + 8 * by * TS + 8 * bx, TS, &v, prm.coeff_shift);
            S.dir[by * 8 + bx] = (uint8_t)d, S.var[by * 8 + bx] = v;
            gdir[(size_t)fb * 64 + by * 8 + bx] = (uint8_t)d, gvar[(size_t)fb * 64 + by * 8 + bx] = v;
        } else {
            S.dir[by * 8 + bx] = gdir[(size_t)fb * 64 + by * 8 + bx];
            S.var[by * 8 + bx] = gvar[(size_t)fb * 64 + by * 8 + bx];
        }
    }
    __syncthreads();
    const size_t soff  = (size_t)(fby * (64 >> pl.ydec)) * pl.source_stride + fbx * (64 >> pl.xdec);
    const int    rows  = bh / sub;  // filtered rows per block (power of two)
    const int    nitem = n * rows;  // work item = one filtered row of one block; the rows of a block sit in adjacent lanes
    const int    rl    = msb((unsigned)rows);
    // the source samples of the listed blocks, staged once for all strengths
    for (int idx = threadIdx.x; idx < (n << (bhl + bwl)); idx += 256) {
        const int bi = idx >> (bhl + bwl), r = (idx >> bwl) & (bh - 1), c = idx & (bw - 1);
        const int y = (S.dl[bi].by << bhl) + r, x = (S.dl[bi].bx << bwl) + c;
        Q.src[y * 64 + x] = (uint16_t)load_px(pl.source, soff + (size_t)y * pl.source_stride + x, pl.is_16bit);
    }
    __syncthreads();
    const uint16_t *in1 = S.tile1 + VB * TS + HB;
    const u16x2     ones = {1, 1};
    const bool luma8 = bsize == 3 && pl.pli == 0;
    const int  pd = prm.pri_damping + prm.coeff_shift - (pl.pli != 0), sd = prm.sec_damping + prm.coeff_shift - (pl.pli != 0);
    // Strengths are evaluated in groups of up to NG that share their taps: a zero primary strength filters along direction
    // 0 (cdef.c:380-388), every other one along the block's direction, so there are two classes.
    constexpr int NG = 4;
    for (int cls = 0; cls < 2; cls++) {
        int gi = 0;
        while (gi < prm.n_strengths) {
            int ids[NG], ng = 0;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                while (gi < prm.n_strengths && (prm.strengths[gi] < 0 || (prm.strengths[gi] / 4 == 0) != (cls == 0))) gi++;
                ids[g] = gi < prm.n_strengths ? gi++ : -1;
                ng += ids[g] >= 0;
            }
            if (ng == 0)
                break;
            int pri_s[NG], sec_s[NG], shs[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int st = ids[g] >= 0 ? prm.strengths[ids[g]] : 0;
                int       sec = st % 4;
                sec += sec == 3;
                pri_s[g] = (st / 4) << prm.coeff_shift, sec_s[g] = sec << prm.coeff_shift;
                shs[g]   = constrain_shift(sec_s[g], sd);
            }
            if (threadIdx.x < NG)
                Q.total[threadIdx.x] = 0;
            __syncthreads();
            unsigned long long acc[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = 0;
#pragma unroll 1
            for (int k = 0; k * 256 < nitem; k++) {
                const int  item = threadIdx.x + k * 256;
                const bool on   = item < nitem;
                const int  bi = on ? item >> rl : 0, row = ((on ? item : 0) & (rows - 1)) * sub;
                // luma: s_y, s_yy, s_yo per strength, s_o / s_oo once; otherwise s_yy = sum of e^2
                uint32_t s_y[NG], s_yy[NG], s_yo[NG], s_o = 0, s_oo = 0;
#pragma unroll
                for (int g = 0; g < NG; g++) s_y[g] = 0, s_yy[g] = 0, s_yo[g] = 0;
                if (on) {
                    const int by = S.dl[bi].by, bx = S.dl[bi].bx;
                    const int at = ((by << bhl) + row) * TS + (bx << bwl);
                    const uint16_t *sp = Q.src + ((by << bhl) + row) * 64 + (bx << bwl);
                    const int dd = cls ? S.dir[by * 8 + bx] : 0;
                    s16x2     thr_p[NG], shp[NG], pt0[NG], pt1[NG];
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        const int t    = pl.pli ? pri_s[g] : adjust_strength(pri_s[g], S.var[by * 8 + bx]);
                        const int tsel = (t >> prm.coeff_shift) & 1;
                        thr_p[g] = splat2(t), shp[g] = splat2(constrain_shift(t, pd));
                        pt0[g] = splat2(tsel ? 3 : 4), pt1[g] = splat2(tsel ? 3 : 2);
                    }
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {  // two samples per step
                        if (j >= bw)
                            break;
                        PairTaps T;
                        pair_taps(T, in + at + j, in1 + at + j, dd);
                        const u16x2 o = __builtin_bit_cast(u16x2, *(const uint32_t *)(sp + j));
                        if (luma8)
                            s_o = __builtin_amdgcn_udot2(o, ones, s_o, false), s_oo = __builtin_amdgcn_udot2(o, o, s_oo, false);
#pragma unroll
                        for (int g = 0; g < NG; g++) {
                            if (g >= ng)  // uniform
                                break;
                            uint32_t yb = __builtin_bit_cast(uint32_t, pair_eval(T, thr_p[g], shp[g], pt0[g], pt1[g], sec_s[g], shs[g]));
                            if (!pl.is_16bit)
                                yb &= 0x00ff00ffu;  // (uint8_t)(int16_t)y
                            const u16x2 y = __builtin_bit_cast(u16x2, yb);
                            if (luma8) {
                                s_y[g]  = __builtin_amdgcn_udot2(y, ones, s_y[g], false);
                                s_yy[g] = __builtin_amdgcn_udot2(y, y, s_yy[g], false);
                                s_yo[g] = __builtin_amdgcn_udot2(y, o, s_yo[g], false);
                            } else {
                                const s16x2 e = __builtin_bit_cast(s16x2, o) - __builtin_bit_cast(s16x2, y);
                                s_yy[g]       = (uint32_t)__builtin_amdgcn_sdot2(e, e, (int)s_yy[g], false);
                            }
                        }
                    }
                }
                // sum over the rows of the block: `rows` adjacent lanes (groups are aligned: 256 % rows == 0)
                for (int off = rows >> 1; off > 0; off >>= 1) {
                    if (luma8)
                        s_o += __shfl_down(s_o, off, 64), s_oo += __shfl_down(s_oo, off, 64);
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        if (g >= ng)
                            break;
                        s_yy[g] += __shfl_down(s_yy[g], off, 64);
                        if (luma8)
                            s_y[g] += __shfl_down(s_y[g], off, 64), s_yo[g] += __shfl_down(s_yo[g], off, 64);
                    }
                }
                if (on && row == 0) {
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        if (g >= ng)
                            break;
                        if (luma8) {  // the five sums of this block; the double-precision distortion is evaluated block-parallel below
                            uint32_t *bs = Q.bsum[g][bi];
                            bs[0] = s_y[g], bs[1] = s_o, bs[2] = s_yy[g], bs[3] = s_oo, bs[4] = s_yo[g];
                        } else {
                            acc[g] += (unsigned long long)s_yy[g];
                        }
                    }
                }
            }
            if (luma8) {  // one lane per (strength, 8x8 block) instead of one in eight lanes of every wave
                __syncthreads();
                for (int e = threadIdx.x; e < ng * n; e += 256) {
                    const int       g = e / n, bi = e - g * n;
                    const uint32_t *bs = Q.bsum[g][bi];
                    atomicAdd(&Q.total[g], (unsigned long long)dist_8xn(bs[0], bs[1], bs[2], bs[3], bs[4], prm.coeff_shift));
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (g >= ng)
                        break;
                    unsigned long long a = acc[g];
                    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
                    if ((threadIdx.x & 63) == 0)
                        atomicAdd(&Q.total[g], a);
                }
            }
            __syncthreads();
            if ((int)threadIdx.x < ng) {
                int id = ids[0];
#pragma unroll
                for (int g = 1; g < NG; g++) id = (int)threadIdx.x == g ? ids[g] : id;
                mse[(size_t)fb * prm.n_strengths + id] = ((uint64_t)Q.total[threadIdx.x] >> 2 * prm.coeff_shift) * (uint64_t)sub;
            }
            __syncthreads();  // totals read before the next group clears them
        }
    }
}

__device__ __forceinline__ void cdef_apply_fb(const SvtHipCdefPlane &pl, const uint8_t *__restrict__ filt,
                                              const uint8_t *__restrict__ fb_strength, int damping, int coeff_shift,
                                              const uint8_t *__restrict__ gdir, const int32_t *__restrict__ gvar, int nhfb, int fb) {
    __shared__ FbLds S;
    __shared__ uint8_t on[64];
    const int fbx = fb % nhfb, fby = fb / nhfb;
    const int n  = stage_fb(S, pl, filt, fbx, fby);
    int       pri = fb_strength[fb] / 4, sec = fb_strength[fb] % 4;
    sec += sec == 3;
    const bool active = (pri || sec) && n > 0;
    if (threadIdx.x < 64) {  // direction / variance of the 64 blocks once, not per sample pair behind the previous pair's stores
        on[threadIdx.x]    = 0;
        S.dir[threadIdx.x] = gdir[(size_t)fb * 64 + threadIdx.x];
        S.var[threadIdx.x] = gvar[(size_t)fb * 64 + threadIdx.x];
    }
    __syncthreads();
    if (active && (int)threadIdx.x < n)
        on[S.dl[threadIdx.x].by * 8 + S.dl[threadIdx.x].bx] = 1;
    __syncthreads();
    const uint16_t *in = S.tile + VB * TS + HB;
    const int       bwl = 3 - pl.xdec, bhl = 3 - pl.ydec;
    const int       fw = 64 >> pl.xdec, fh = 64 >> pl.ydec;
    const int       pri_s = pri << coeff_shift, sec_s = sec << coeff_shift, dmp = damping + coeff_shift - (pl.pli != 0);
    const uint16_t *in1 = S.tile1 + VB * TS + HB;
    for (int idx = threadIdx.x; idx < (fw >> 1) * fh; idx += blockDim.x) {  // two samples per step
        const int y = idx / (fw >> 1), x = 2 * (idx - y * (fw >> 1));
        const int py = fby * fh + y, px = fbx * fw + x;
        if (py >= (int)pl.height || px >= (int)pl.width)
            continue;
        const int by = y >> bhl, bx = x >> bwl;
        uint32_t  v  = *(const uint32_t *)(in + y * TS + x);
        if (on[by * 8 + bx]) {
            const int t = pl.pli ? pri_s : adjust_strength(pri_s, S.var[by * 8 + bx]);
            v = __builtin_bit_cast(uint32_t, cdef_pair(in + y * TS + x, in1 + y * TS + x, t, sec_s, pri_s ? S.dir[by * 8 + bx] : 0, dmp, dmp, coeff_shift));
        }
        const bool two = px + 1 < (int)pl.width;
        if (pl.is_16bit) {  // one store per pair wherever the pair is naturally aligned
            uint16_t *o = (uint16_t *)pl.source + (size_t)py * pl.source_stride + px;
            if (two && ((uintptr_t)o & 3) == 0) {
                *(uint32_t *)o = v;
            } else {
                o[0] = (uint16_t)v;
                if (two)
                    o[1] = (uint16_t)(v >> 16);
            }
        } else {
            uint8_t *o = (uint8_t *)pl.source + (size_t)py * pl.source_stride + px;
            if (two && ((uintptr_t)o & 1) == 0) {
                *(uint16_t *)o = (uint16_t)((v & 0xffu) | ((v >> 8) & 0xff00u));
            } else {
                o[0] = (uint8_t)v;
                if (two)
                    o[1] = (uint8_t)(v >> 16);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cdef_apply_kernel(SvtHipCdefPlane pl, const uint8_t *__restrict__ filt,
                                                         const uint8_t *__restrict__ fb_strength, int damping, int coeff_shift,
                                                         const uint8_t *__restrict__ gdir, const int32_t *__restrict__ gvar, int nhfb) {
    cdef_apply_fb(pl, filt, fb_strength, damping, coeff_shift, gdir, gvar, nhfb, blockIdx.x);
}
// all planes of a picture in one launch (blockIdx.y = plane): a plane's 2040 filter blocks of a 4K picture are 1.3 rounds
// of the 1536 workgroups the chip holds, three planes are 4.0
struct ApplyFrame {
    SvtHipCdefPlane pl[3];
    const uint8_t  *strength[3];
};
__global__ __launch_bounds__(256) void cdef_apply_frame_kernel(ApplyFrame f, const uint8_t *__restrict__ filt, int damping, int coeff_shift,
                                                               const uint8_t *__restrict__ gdir, const int32_t *__restrict__ gvar, int nhfb) {
    const int p = blockIdx.y;
    cdef_apply_fb(p == 0 ? f.pl[0] : (p == 1 ? f.pl[1] : f.pl[2]), filt, p == 0 ? f.strength[0] : (p == 1 ? f.strength[1] : f.strength[2]), damping,
                  coeff_shift, gdir, gvar, nhfb, blockIdx.x);
}

// ---- Tier A kernels ----
__global__ void find_dir_kernel(const uint16_t *img1, const uint16_t *img2, int stride, int coeff_shift, int32_t *out /* dir1,var1,dir2,var2 */) {
    if (threadIdx.x < 2) {
        const uint16_t *img = threadIdx.x ? img2 : img1;
        if (img) {
            int32_t v;
            out[2 * threadIdx.x]     = find_dir_block(img, stride, &v, coeff_shift);
            out[2 * threadIdx.x + 1] = v;
        }
    }
}
__global__ void filter_block_kernel(uint8_t *dst8, uint16_t *dst16, int dstride, const uint16_t *in, int pri, int sec, int dir, int pd,
                                    int sd, int bsize, int coeff_shift, int sub) {
    const int bh = 4 << (bsize == 3 || bsize == 1), bw = 4 << (bsize == 3 || bsize == 2);
    const int i = threadIdx.x / 8, j = threadIdx.x & 7;
    if (i >= bh || j >= bw || (i % sub))
        return;
    const int32_t y = cdef_pixel(in + i * BS + j, pri, sec, dir, pd, sd, coeff_shift);
    if (dst8)
        dst8[i * dstride + j] = (uint8_t)(int16_t)y;
    else
        dst16[i * dstride + j] = (uint16_t)(int16_t)y;
}
__global__ __launch_bounds__(256) void dist_kernel(const void *dst, int dstride, const void *src, const SvtHipCdefList *dlist, int n,
                                                   int bsize, int coeff_shift, int pli, int sub, int is16, uint64_t *out) {
    __shared__ unsigned long long total;
    if (threadIdx.x == 0)
        total = 0;
    __syncthreads();
    const int          bw = 4 << (bsize == 3 || bsize == 2), bh = 4 << (bsize == 3 || bsize == 1);
    const int          wl = bw == 8 ? 3 : 2, hl = bh == 8 ? 3 : 2;
    unsigned long long acc = 0;
    for (int bi = threadIdx.x; bi < n; bi += blockDim.x) {
        const int    by = dlist[bi].by, bx = dlist[bi].bx;
        const size_t so = (size_t)bi << (wl + hl), dof = (size_t)(by << hl) * dstride + (bx << wl);
        if (bsize == 3 && pli == 0) {
            uint64_t ss = 0, sd = 0, ss2 = 0, sd2 = 0, ssd = 0;
            for (int i = 0; i < 8; i += sub)
                for (int j = 0; j < 8; j++) {
                    const int32_t s = (int32_t)load_px(src, so + 8 * i + j, is16), d = (int32_t)load_px(dst, dof + (size_t)i * dstride + j, is16);
                    ss += s, sd += d, ss2 += (uint64_t)(s * s), sd2 += (uint64_t)(d * d), ssd += (uint64_t)(s * d);
                }
            acc += dist_8xn(ss, sd, ss2, sd2, ssd, coeff_shift);
        } else {
            for (int i = 0; i < bh; i += sub)
                for (int j = 0; j < bw; j++) {
                    const int32_t e = (int32_t)load_px(dst, dof + (size_t)i * dstride + j, is16) - (int32_t)load_px(src, so + bw * i + j, is16);
                    acc += (unsigned long long)(int64_t)(e * e);
                }
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&total, acc);
    __syncthreads();
    if (threadIdx.x == 0)
        *out = (uint64_t)total >> 2 * coeff_shift;
}

// xdec != ydec (4:2:2 / 4:4:0) is refused: the reference encoder only accepts 4:2:0 (enc_settings.c:447) and its filter_fb
// remaps chroma directions in place for those layouts (cdef.c:389-396), which no reachable reference path exercises.
bool plane_ok(const SvtHipCdefPlane *p) {
    return p && p->recon && p->source && p->width && p->height && p->xdec <= 1 && p->ydec == p->xdec && (p->pli == 0 ? p->xdec == 0 : p->pli <= 2);
}
// svt_search_one_dual (enc_cdef.c:627-686).  mse: [2][sb_count][n] dense (luma, chroma), n = end_gi.  Each thread owns
// (luma strength j, chroma strength k) pairs and walks the filter blocks in order, so every total is the same 64-bit sum as
// the reference's; the winner is the smallest (total, pair number) — the first strict minimum of the reference's raster scan.
__global__ __launch_bounds__(256) void search_one_dual_kernel(const uint64_t *__restrict__ mse, const int32_t *__restrict__ lev, int nb,
                                                              int sb_count, int n, int start_gi, uint64_t *__restrict__ out) {
    __shared__ uint64_t s_tot[256];
    __shared__ uint32_t s_id[256];
    const uint64_t     *m0 = mse, *m1 = mse + (size_t)sb_count * n;
    const int           span = n - start_gi;
    uint64_t            best = 1ull << 63;
    uint32_t            best_id = 0xffffffffu;
    for (int pair = threadIdx.x; pair < span * span; pair += 256) {
        const int j = start_gi + pair / span, k = start_gi + pair % span;
        uint64_t  tot = 0;
        for (int i = 0; i < sb_count; i++) {
            uint64_t cur = 1ull << 63;
            for (int gi = 0; gi < nb; gi++) {
                const uint64_t c = m0[(size_t)i * n + lev[gi]] + m1[(size_t)i * n + lev[8 + gi]];
                cur              = c < cur ? c : cur;
            }
            const uint64_t c = m0[(size_t)i * n + j] + m1[(size_t)i * n + k];
            tot += c < cur ? c : cur;
        }
        if (tot < best)  // pairs of one thread are visited in increasing order
            best = tot, best_id = (uint32_t)pair;
    }
    s_tot[threadIdx.x] = best, s_id[threadIdx.x] = best_id;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const uint64_t t = s_tot[threadIdx.x + off];
            const uint32_t d = s_id[threadIdx.x + off];
            if (t < s_tot[threadIdx.x] || (t == s_tot[threadIdx.x] && d < s_id[threadIdx.x]))
                s_tot[threadIdx.x] = t, s_id[threadIdx.x] = d;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[0] = s_tot[0], out[1] = s_id[0];
}

[[noreturn]] void fatal(const char *what) { svthip::tier_a_throw("%s: %s", what, svt_hip_last_error()); }
inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// ------------------------------------------------------------------------------------------------ Tier B
extern "C" int32_t svt_hip_cdef_search_plane(const SvtHipCdefPlane *plane, const uint8_t *d_filt8x8, const SvtHipCdefSearchParams *prm,
                                             uint64_t *d_mse, uint8_t *d_dir, int32_t *d_var, void *stream) {
    if (!plane_ok(plane) || !d_filt8x8 || !prm || !d_mse || !d_dir || !d_var || prm->n_strengths < 1 ||
        prm->n_strengths > SVT_HIP_CDEF_MAX_STRENGTHS || prm->subsampling_factor < 1) {
        set_error("svt_hip_cdef_search_plane: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const int lw = (int)plane->width << plane->xdec, lh = (int)plane->height << plane->ydec;
    const int nhfb = (lw + 63) / 64, nvfb = (lh + 63) / 64;
    hipLaunchKernelGGL(cdef_search_kernel, dim3(nhfb * nvfb), dim3(256), 0, resolve_stream(stream), *plane, d_filt8x8, *prm, d_mse, d_dir,
                       d_var, nhfb);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_cdef_apply_plane(const SvtHipCdefPlane *plane, const uint8_t *d_filt8x8, const uint8_t *d_fb_strength,
                                            int32_t damping, int32_t coeff_shift, const uint8_t *d_dir, const int32_t *d_var,
                                            void *stream) {
    if (!plane_ok(plane) || !d_filt8x8 || !d_fb_strength || !d_dir || !d_var || plane->recon == plane->source) {
        set_error("svt_hip_cdef_apply_plane: bad argument (input and output planes must differ)");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const int lw = (int)plane->width << plane->xdec, lh = (int)plane->height << plane->ydec;
    const int nhfb = (lw + 63) / 64, nvfb = (lh + 63) / 64;
    hipLaunchKernelGGL(cdef_apply_kernel, dim3(nhfb * nvfb), dim3(256), 0, resolve_stream(stream), *plane, d_filt8x8, d_fb_strength,
                       damping, coeff_shift, d_dir, d_var, nhfb);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------ Tier A
extern "C" int32_t svt_hip_cdef_apply_frame(const SvtHipCdefPlane *planes, uint32_t n_planes, const uint8_t *d_filt8x8,
                                            const uint8_t *const *fb_strength_dptrs, int32_t damping, int32_t coeff_shift, const uint8_t *d_dir,
                                            const int32_t *d_var, void *stream) {
    if (!planes || n_planes == 0 || n_planes > 3 || !d_filt8x8 || !fb_strength_dptrs || !d_dir || !d_var) {
        set_error("svt_hip_cdef_apply_frame: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    ApplyFrame f{};
    for (uint32_t p = 0; p < n_planes; p++) {
        const int lw = (int)planes[p].width << planes[p].xdec, lh = (int)planes[p].height << planes[p].ydec;
        const int lw0 = (int)planes[0].width << planes[0].xdec, lh0 = (int)planes[0].height << planes[0].ydec;
        if (!plane_ok(&planes[p]) || !fb_strength_dptrs[p] || planes[p].recon == planes[p].source || (lw + 63) / 64 != (lw0 + 63) / 64 ||
            (lh + 63) / 64 != (lh0 + 63) / 64) {
            set_error("svt_hip_cdef_apply_frame: plane %u: bad plane (input and output must differ; all planes must cover the same filter blocks)", p);
            return SVT_HIP_ERR_BAD_PARAMETER;
        }
        f.pl[p] = planes[p], f.strength[p] = fb_strength_dptrs[p];
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const int lw = (int)planes[0].width << planes[0].xdec, lh = (int)planes[0].height << planes[0].ydec;
    const int nhfb = (lw + 63) / 64, nvfb = (lh + 63) / 64;
    hipLaunchKernelGGL(cdef_apply_frame_kernel, dim3(nhfb * nvfb, n_planes), dim3(256), 0, resolve_stream(stream), f, d_filt8x8, damping,
                       coeff_shift, d_dir, d_var, nhfb);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

static void svt_aom_cdef_find_dir_dual_hip_impl(const uint16_t *img1, const uint16_t *img2, int stride, int32_t *var1, int32_t *var2, int32_t coeff_shift, uint8_t *out1, uint8_t *out2);
extern "C" void svt_aom_cdef_find_dir_dual_hip(const uint16_t *img1, const uint16_t *img2, int stride, int32_t *var1, int32_t *var2, int32_t coeff_shift, uint8_t *out1, uint8_t *out2) { TIER_A_CALL(svt_aom_cdef_find_dir_dual, svt_aom_cdef_find_dir_dual_hip_impl(img1, img2, stride, var1, var2, coeff_shift, out1, out2), (img1, img2, stride, var1, var2, coeff_shift, out1, out2)); }
static void svt_aom_cdef_find_dir_dual_hip_impl(const uint16_t *img1, const uint16_t *img2, int stride, int32_t *var1, int32_t *var2, int32_t coeff_shift, uint8_t *out1, uint8_t *out2) {
    if (!ensure_init())
        fatal("cdef_find_dir");
    hipStream_t  st   = resolve_stream(nullptr);
    Scratch     &sc   = tls_scratch();
    const size_t span = ((size_t)7 * stride + 8) * 2, o2 = up256(span + 16), ores = 2 * o2;
    uint8_t     *d = sc.device(ores + 256), *h = sc.host(ores + 256);
    memcpy(h, img1, span);
    if (img2)
        memcpy(h + o2, img2, span);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, ores, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(find_dir_kernel, dim3(1), dim3(64), 0, st, (const uint16_t *)d, img2 ? (const uint16_t *)(d + o2) : nullptr, stride,
                       coeff_shift, (int32_t *)(d + ores));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + ores, d + ores, 16, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    const int32_t *r = (const int32_t *)(h + ores);
    *out1 = (uint8_t)r[0], *var1 = r[1];
    if (img2)
        *out2 = (uint8_t)r[2], *var2 = r[3];
}
static uint8_t svt_aom_cdef_find_dir_hip_impl(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift);
extern "C" uint8_t svt_aom_cdef_find_dir_hip(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift) { TIER_A_CALL(svt_aom_cdef_find_dir, svt_aom_cdef_find_dir_hip_impl(img, stride, var, coeff_shift), (img, stride, var, coeff_shift)); }
static uint8_t svt_aom_cdef_find_dir_hip_impl(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift) {
    uint8_t d = 0;
    svt_aom_cdef_find_dir_dual_hip(img, nullptr, stride, var, nullptr, coeff_shift, &d, nullptr);
    return d;
}

static void svt_cdef_filter_block_hip_impl(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in, int32_t pri_strength, int32_t sec_strength, int32_t dir, int32_t pri_damping, int32_t sec_damping, int32_t bsize, int32_t coeff_shift, uint8_t subsampling_factor);
extern "C" void svt_cdef_filter_block_hip(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in, int32_t pri_strength, int32_t sec_strength, int32_t dir, int32_t pri_damping, int32_t sec_damping, int32_t bsize, int32_t coeff_shift, uint8_t subsampling_factor) { TIER_A_CALL(svt_cdef_filter_block, svt_cdef_filter_block_hip_impl(dst8, dst16, dstride, in, pri_strength, sec_strength, dir, pri_damping, sec_damping, bsize, coeff_shift, subsampling_factor), (dst8, dst16, dstride, in, pri_strength, sec_strength, dir, pri_damping, sec_damping, bsize, coeff_shift, subsampling_factor)); }
static void svt_cdef_filter_block_hip_impl(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in, int32_t pri_strength, int32_t sec_strength, int32_t dir, int32_t pri_damping, int32_t sec_damping, int32_t bsize, int32_t coeff_shift, uint8_t subsampling_factor) {
    if (!ensure_init())
        fatal("cdef_filter_block");
    const int    bh = 4 << (bsize == 3 || bsize == 1), bw = 4 << (bsize == 3 || bsize == 2);
    const size_t before = 2 * BS + 2, after = (size_t)(bh - 1 + 2) * BS + bw + 2;  // taps reach +-(2 rows, 2 columns)
    const size_t in_bytes = (before + after) * 2, px = dst16 ? 2 : 1, out_bytes = ((size_t)(bh - 1) * dstride + bw) * px;
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    const size_t oo = up256(in_bytes + 16);
    uint8_t     *d = sc.device(oo + out_bytes + 256), *h = sc.host(oo + out_bytes + 256);
    memcpy(h, in - before, in_bytes);
    memcpy(h + oo, dst16 ? (const void *)dst16 : (const void *)dst8, out_bytes);  // rows skipped by sub-sampling keep their content
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, oo + out_bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(filter_block_kernel, dim3(1), dim3(64), 0, st, dst16 ? nullptr : d + oo, dst16 ? (uint16_t *)(d + oo) : nullptr,
                       dstride, (const uint16_t *)d + before, pri_strength, sec_strength, dir, pri_damping, sec_damping, bsize, coeff_shift,
                       (int)subsampling_factor);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + oo, d + oo, out_bytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    for (int r = 0; r < bh; r++)
        memcpy((uint8_t *)(dst16 ? (void *)dst16 : (void *)dst8) + (size_t)r * dstride * px, h + oo + (size_t)r * dstride * px, bw * px);
}

// Pure 8-bit -> 16-bit widening of a host rectangle: no arithmetic, evaluated on the calling thread
// (the device-resident equivalent is the tile staging of the kernels above).
static void svt_aom_copy_rect8_8bit_to_16bit_hip_impl(uint16_t *dst, int32_t dstride, const uint8_t *src, int32_t sstride, int32_t v, int32_t hh);
extern "C" void svt_aom_copy_rect8_8bit_to_16bit_hip(uint16_t *dst, int32_t dstride, const uint8_t *src, int32_t sstride, int32_t v, int32_t hh) { TIER_A_CALL(svt_aom_copy_rect8_8bit_to_16bit, svt_aom_copy_rect8_8bit_to_16bit_hip_impl(dst, dstride, src, sstride, v, hh), (dst, dstride, src, sstride, v, hh)); }
static void svt_aom_copy_rect8_8bit_to_16bit_hip_impl(uint16_t *dst, int32_t dstride, const uint8_t *src, int32_t sstride, int32_t v, int32_t hh) {
    for (int32_t i = 0; i < v; i++)
        for (int32_t j = 0; j < hh; j++) dst[i * dstride + j] = src[i * sstride + j];
}

static uint64_t dist_tier_a(const void *dst, int32_t dstride, const void *src, const SvtHipCdefList *dlist, int32_t n, int32_t bsize,
                            int32_t coeff_shift, int32_t pli, uint8_t sub, int is16) {
    if (n <= 0)
        return 0;
    if (!ensure_init())
        fatal("compute_cdef_dist");
    const int    bw = 4 << (bsize == 3 || bsize == 2), bh = 4 << (bsize == 3 || bsize == 1), px = is16 ? 2 : 1;
    int          maxy = 0, maxx = 0;
    for (int i = 0; i < n; i++) maxy = dlist[i].by > maxy ? dlist[i].by : maxy, maxx = dlist[i].bx > maxx ? dlist[i].bx : maxx;
    const size_t dst_bytes = ((size_t)((maxy + 1) * bh - 1) * dstride + (size_t)(maxx + 1) * bw) * px;
    const size_t src_bytes = (size_t)n * bw * bh * px, dl_bytes = (size_t)n * sizeof(SvtHipCdefList);
    const size_t o_src = up256(dst_bytes + 16), o_dl = o_src + up256(src_bytes + 16), o_res = o_dl + up256(dl_bytes + 16);
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    uint8_t     *d = sc.device(o_res + 256), *h = sc.host(o_res + 256);
    memcpy(h, dst, dst_bytes), memcpy(h + o_src, src, src_bytes), memcpy(h + o_dl, dlist, dl_bytes);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, o_res, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dist_kernel, dim3(1), dim3(256), 0, st, (const void *)d, dstride, (const void *)(d + o_src),
                       (const SvtHipCdefList *)(d + o_dl), n, bsize, coeff_shift, pli, (int)sub, is16, (uint64_t *)(d + o_res));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + o_res, d + o_res, 8, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    return *(const uint64_t *)(h + o_res);
}
static uint64_t svt_compute_cdef_dist_16bit_hip_impl(const uint16_t *dst, int32_t dstride, const uint16_t *src, const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli, uint8_t sub);
extern "C" uint64_t svt_compute_cdef_dist_16bit_hip(const uint16_t *dst, int32_t dstride, const uint16_t *src, const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli, uint8_t sub) { TIER_A_CALL(svt_compute_cdef_dist_16bit, svt_compute_cdef_dist_16bit_hip_impl(dst, dstride, src, dlist, cdef_count, bsize, coeff_shift, pli, sub), (dst, dstride, src, dlist, cdef_count, bsize, coeff_shift, pli, sub)); }
static uint64_t svt_compute_cdef_dist_16bit_hip_impl(const uint16_t *dst, int32_t dstride, const uint16_t *src, const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli, uint8_t sub) {
    return dist_tier_a(dst, dstride, src, dlist, cdef_count, bsize, coeff_shift, pli, sub, 1);
}
static uint64_t svt_compute_cdef_dist_8bit_hip_impl(const uint8_t *dst8, int32_t dstride, const uint8_t *src8, const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli, uint8_t sub);
extern "C" uint64_t svt_compute_cdef_dist_8bit_hip(const uint8_t *dst8, int32_t dstride, const uint8_t *src8, const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli, uint8_t sub) { TIER_A_CALL(svt_compute_cdef_dist_8bit, svt_compute_cdef_dist_8bit_hip_impl(dst8, dstride, src8, dlist, cdef_count, bsize, coeff_shift, pli, sub), (dst8, dstride, src8, dlist, cdef_count, bsize, coeff_shift, pli, sub)); }
static uint64_t svt_compute_cdef_dist_8bit_hip_impl(const uint8_t *dst8, int32_t dstride, const uint8_t *src8, const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli, uint8_t sub) {
    return dist_tier_a(dst8, dstride, src8, dlist, cdef_count, bsize, coeff_shift, pli, sub, 0);
}

// svt_search_one_dual (aom_dsp_rtcd.h:239): mse[0][i] / mse[1][i] are the luma / chroma tables of filter block i (TOTAL_STRENGTHS
// entries each in the reference; entries below end_gi are read).  lev0 / lev1 hold nb_strengths chosen pairs and receive one more.
static uint64_t svt_search_one_dual_hip_impl(int *lev0, int *lev1, int nb_strengths, uint64_t **mse[2], int sb_count, int start_gi, int end_gi);
extern "C" uint64_t svt_search_one_dual_hip(int *lev0, int *lev1, int nb_strengths, uint64_t **mse[2], int sb_count, int start_gi, int end_gi) { TIER_A_CALL(svt_search_one_dual, svt_search_one_dual_hip_impl(lev0, lev1, nb_strengths, mse, sb_count, start_gi, end_gi), (lev0, lev1, nb_strengths, mse, sb_count, start_gi, end_gi)); }
static uint64_t svt_search_one_dual_hip_impl(int *lev0, int *lev1, int nb_strengths, uint64_t **mse[2], int sb_count, int start_gi, int end_gi) {
    if (nb_strengths < 0 || nb_strengths >= 8 || start_gi < 0 || end_gi > 64 || sb_count < 0) {
        set_error("svt_search_one_dual: nb_strengths %d / strength range %d..%d outside CDEF's limits", nb_strengths, start_gi, end_gi);
        fatal("svt_search_one_dual");
    }
    if (end_gi <= start_gi) {  // nothing to search: the reference leaves (0, 0) and 1 << 63
        lev0[nb_strengths] = lev1[nb_strengths] = 0;
        return 1ull << 63;
    }
    if (!ensure_init())
        fatal("svt_search_one_dual");
    const size_t n = (size_t)end_gi, tab = up256((size_t)2 * sb_count * n * 8 + 8), o_lev = tab, o_res = o_lev + 256;
    Scratch     &sc = tls_scratch();
    uint8_t     *h = sc.host(o_res + 256), *d = sc.device(o_res + 256);
    uint64_t    *m = (uint64_t *)h;
    for (int p = 0; p < 2; p++)
        for (int i = 0; i < sb_count; i++) memcpy(m + ((size_t)p * sb_count + i) * n, mse[p][i], n * 8);
    int32_t *lv = (int32_t *)(h + o_lev);
    for (int g = 0; g < 8; g++) lv[g] = g < nb_strengths ? lev0[g] : 0, lv[8 + g] = g < nb_strengths ? lev1[g] : 0;
    for (int g = 0; g < nb_strengths; g++)
        if (lev0[g] < 0 || lev0[g] >= end_gi || lev1[g] < 0 || lev1[g] >= end_gi) {
            set_error("svt_search_one_dual: chosen strength outside the searched range");
            fatal("svt_search_one_dual");
        }
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, o_res, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(search_one_dual_kernel, dim3(1), dim3(256), 0, st, (const uint64_t *)d, (const int32_t *)(d + o_lev), nb_strengths,
                       sb_count, end_gi, start_gi, (uint64_t *)(d + o_res));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + o_res, d + o_res, 16, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    const uint64_t *res = (const uint64_t *)(h + o_res);
    const int       span = end_gi - start_gi;
    const uint32_t  id   = (uint32_t)res[1];
    lev0[nb_strengths] = id == 0xffffffffu ? 0 : start_gi + (int)id / span;
    lev1[nb_strengths] = id == 0xffffffffu ? 0 : start_gi + (int)id % span;
    return res[0];
}

SVT_HIP_MODULE_WARMUP(loopfilter_cdef)

// tf_picture.hip — the temporal filter's block loop for one centre picture on gfx950 (the motion half).
// Replaces produce_temporally_filtered_pic (temporal_filtering.c:2752-3308): per reference picture ME_MCTF motion estimation
// (me_frame.hip), then — this file — tf_64x64 / tf_32x32 / tf_16x16_sub_pel_search (:1531-2104), tf_use_64x64_pred (:2646), the
// 64-vs-32 and 32-vs-16 decisions (:3118-3245, derive_tf_32x32_block_split_flag :236-285), the descriptors of the final
// sharp-filter predictions (tf_64x64 / tf_32x32_inter_prediction :2226-2576, run by inter_convolve.hip),
// convert_64x64_info_to_32x32_info (:2661-2728) and the SvtHipTfBlock records the accumulate stage (tf_filter.hip) consumes.
//
// tf_refine_kernel: one workgroup per (reference picture, 64x64 block).  The source block lives in LDS for the whole search; each
// stage (the 64x64 block, the four 32x32, the four 16x16 of a 32x32) stages one reference window per block (block + 10 samples each
// side: the 7/8-sample drift of the three refinement rounds, the 8-tap margin and the doubled row step of the sub-sampled centre
// position) and evaluates the candidates of a refinement round of all its blocks side by side, each by the lanes of one wave —
// see "the sub-pel searches" below.
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/svt_hip_inter.h"
#include "../../include/svt_hip_tf.h"
#include "common.hpp"
#include "convolve_device.hpp"

using namespace svthip;

namespace {

// the AV1 interpolation kernels (inter_prediction.c:223-300): regular, sharp, bilinear — 16 phases x 8 taps
__device__ const int16_t TF_KERNELS[4][16][8] = {
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 2, -6, 126, 8, -2, 0, 0}, {0, 2, -10, 122, 18, -4, 0, 0}, {0, 2, -12, 116, 28, -8, 2, 0},
     {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -14, 102, 48, -12, 2, 0}, {0, 2, -16, 94, 58, -12, 2, 0}, {0, 2, -14, 84, 66, -12, 2, 0},
     {0, 2, -14, 76, 76, -14, 2, 0}, {0, 2, -12, 66, 84, -14, 2, 0}, {0, 2, -12, 58, 94, -16, 2, 0}, {0, 2, -12, 48, 102, -14, 2, 0},
     {0, 2, -10, 38, 110, -14, 2, 0}, {0, 2, -8, 28, 116, -12, 2, 0}, {0, 0, -4, 18, 122, -10, 2, 0}, {0, 0, -2, 8, 126, -6, 2, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {-2, 2, -6, 126, 8, -2, 2, 0}, {-2, 6, -12, 124, 16, -6, 4, -2}, {-2, 8, -18, 120, 26, -10, 6, -2},
     {-4, 10, -22, 116, 38, -14, 6, -2}, {-4, 10, -22, 108, 48, -18, 8, -2}, {-4, 10, -24, 100, 60, -20, 8, -2},
     {-4, 10, -24, 90, 70, -22, 10, -2}, {-4, 12, -24, 80, 80, -24, 12, -4}, {-2, 10, -22, 70, 90, -24, 10, -4},
     {-2, 8, -20, 60, 100, -24, 10, -4}, {-2, 8, -18, 48, 108, -22, 10, -4}, {-2, 6, -14, 38, 116, -22, 10, -4},
     {-2, 6, -10, 26, 120, -18, 8, -2}, {-2, 4, -6, 16, 124, -12, 6, -2}, {0, 2, -2, 8, 126, -6, 2, -2}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, 0, 120, 8, 0, 0, 0}, {0, 0, 0, 112, 16, 0, 0, 0}, {0, 0, 0, 104, 24, 0, 0, 0},
     {0, 0, 0, 96, 32, 0, 0, 0}, {0, 0, 0, 88, 40, 0, 0, 0}, {0, 0, 0, 80, 48, 0, 0, 0}, {0, 0, 0, 72, 56, 0, 0, 0},
     {0, 0, 0, 64, 64, 0, 0, 0}, {0, 0, 0, 56, 72, 0, 0, 0}, {0, 0, 0, 48, 80, 0, 0, 0}, {0, 0, 0, 40, 88, 0, 0, 0},
     {0, 0, 0, 32, 96, 0, 0, 0}, {0, 0, 0, 24, 104, 0, 0, 0}, {0, 0, 0, 16, 112, 0, 0, 0}, {0, 0, 0, 8, 120, 0, 0, 0}},
    // sub_pel_filters_4: what blocks of width <= 4 get for the regular and the sharp filter (inter_prediction.h:137-145)
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, -4, 126, 8, -2, 0, 0}, {0, 0, -8, 122, 18, -4, 0, 0}, {0, 0, -10, 116, 28, -6, 0, 0},
     {0, 0, -12, 110, 38, -8, 0, 0}, {0, 0, -12, 102, 48, -10, 0, 0}, {0, 0, -14, 94, 58, -10, 0, 0}, {0, 0, -12, 84, 66, -10, 0, 0},
     {0, 0, -12, 76, 76, -12, 0, 0}, {0, 0, -10, 66, 84, -12, 0, 0}, {0, 0, -10, 58, 94, -14, 0, 0}, {0, 0, -10, 48, 102, -12, 0, 0},
     {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -6, 28, 116, -10, 0, 0}, {0, 0, -4, 18, 122, -8, 0, 0}, {0, 0, -2, 8, 126, -4, 0, 0}}};
enum { K_REGULAR = 0, K_SHARP = 1, K_BILINEAR = 2, K_REGULAR4 = 3 };
constexpr int WIN_MARGIN = 10;
constexpr int DESC_PER_B64 = 48;

struct PicPlanes {  // one picture of the window; every pointer at sample (0,0) of its plane
    const uint8_t  *y8, *c8[2];
    const uint16_t *y16, *c16[2];
    uint32_t        stride, stride_c;
};
struct RefineArgs {  // uniform for the call
    SvtHipTfCtrls ctrls;
    PicPlanes     centre;
    uint32_t      decay[3];
    uint32_t      tf_me_exit_th;
    int32_t       mi_rows, mi_cols;
    uint32_t      nb, bw;
    int64_t       idx_min, idx_max;  // first / last sample of a padded luma plane relative to its sample (0,0)
    uint16_t      mv_dist_th;
    uint8_t       bit_depth, chroma;
    uint32_t     *accum;  // [nb][3][4096]
    uint16_t     *count;  // [nb][3][4096]
};
struct RefineRef {  // per reference picture
    PicPlanes                   pic;
    const uint32_t             *best_mv, *best_sad;  // [nb][2][4][85]
    const SvtHipMeSearchResult *sr;                  // [nb][2][4]
    uint8_t                    *pred;                // [nb][3][4096] samples of the filter's depth
    SvtHipTfB64State           *state;               // [nb]
    SvtHipConvolveDesc         *desc;                // [nb][48]
    SvtHipTfBlock              *blocks;              // [nb][4]
};

__device__ __forceinline__ int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int32_t rnd(int32_t v, int n) { return (v + ((1 << n) >> 1)) >> n; }

// compute_subpel_params + clamp_mv_to_umv_border_sb (enc_inter_prediction.c:28-48, 3166-3177) for a bw x bh block of a plane
// with sub-sampling ss that belongs to the luma block (lx, ly, lsize): 1/16-sample column / row
__device__ __forceinline__ void clamp_mv(int32_t mi_rows, int32_t mi_cols, int bw, int bh, int mvx, int mvy, int ss, int lx, int ly, int lsize,
                                         int &col, int &row) {
    const int     mirow = ly >> 2, micol = lx >> 2, bmi = lsize >> 2;
    const int32_t to_top = -((mirow * 4) * 8), to_bottom = ((mi_rows - bmi - mirow) * 4) * 8;
    const int32_t to_left = -((micol * 4) * 8), to_right = ((mi_cols - bmi - micol) * 4) * 8;
    const int32_t spel_left = (4 + bw) << 4, spel_right = spel_left - 16, spel_top = (4 + bh) << 4, spel_bottom = spel_top - 16;
    const int     m = 1 << (1 - ss);
    col = (int16_t)clampi((int16_t)(mvx * m), to_left * m - spel_left, to_right * m + spel_right);
    row = (int16_t)clampi((int16_t)(mvy * m), to_top * m - spel_top, to_bottom * m + spel_bottom);
}

constexpr int WIN_ELEMS = 4 * (32 + 2 * WIN_MARGIN) * (32 + 2 * WIN_MARGIN);  // four 32x32 windows >= one 64x64 window, four 16x16 windows
constexpr int MAX_ITEMS = 32;                                                  // four blocks x eight candidates of a round
struct Blk {  // one square block of a search stage
    uint64_t best;      // best distortion so far
    int32_t  mvx, mvy;  // its vector (1/8 sample) = the centre of the next round
    int16_t  lx, ly;    // block origin inside the 64x64 block
    int16_t  wx, wy;    // window origin relative to the 64x64 block's origin
    uint32_t done;      // best == 0 or below the early-exit threshold: every later candidate is skipped
};
struct Item {  // one candidate position of a round
    int16_t px, py;  // block sample (0,0) of the candidate in window coordinates
    int16_t mvx, mvy;
    uint8_t sx, sy, blk, on;
};
// S16 = false (8-bit searches): the windows and the source block are BYTES, the window stored as sample - 128 so that the horizontal
// filter is two v_dot4_i32_i8 over eight samples (the taps fit an int8 except the unit tap 128 of phase 0: its row is stored as zeros
// and an integer-phase lane adds 128 * sample itself); half the LDS of the 16-bit layout: a sixth workgroup per CU.
template <bool S16>
struct LdsT {
    typedef typename std::conditional<S16, uint16_t, uint8_t>::type Pix;
    alignas(4) Pix src[64 * 64];
    alignas(4) Pix win[WIN_ELEMS];
    alignas(16) int16_t taps[2][16][8];  // regular, bilinear
    alignas(8) int8_t   taps8[2][16][8]; // the same as int8, phase 0 zeroed (S16 = false)
    Item     item[MAX_ITEMS];
    uint64_t dist[MAX_ITEMS];
    Blk      blk[4];
};

struct SearchCtx {
    const RefineArgs *a;
    const void       *ref0;  // search-depth luma of the reference picture
    uint32_t          ref_stride;
    int               ox, oy, is16, bd;
};

template <bool S16>
__device__ __forceinline__ uint32_t ldg(const void *p, ptrdiff_t i) {
    if (S16)
        return ((const __attribute__((address_space(1))) uint16_t *)p)[i];
    return ((const __attribute__((address_space(1))) uint8_t *)p)[i];
}

// ---- the sub-pel searches (tf_64x64 / tf_32x32 / tf_16x16_sub_pel_search -> tf_subpel_search, temporal_filtering.c:1531-2104)
//
// One search = a centre position and up to three refinement rounds (half, quarter, eighth of a sample) of 4 or 8 candidate vectors
// around the best one so far.  The reference walks the candidates one after the other and skips a candidate when the best distortion
// so far is zero or below the early-exit threshold; a candidate's distortion itself does not depend on its predecessors.  So all
// candidates of a round — of ALL blocks of the stage (one 64x64, four 32x32, the four 16x16 of a 32x32) — are evaluated side by
// side as `items`, and a replay applies the reference's skip / take rules to the distortions in the reference's order (both skip
// tests are monotone in the best distortion, so a block that meets one at the start of a round is left out for good).
//
// An item is evaluated by W lanes of ONE wave (W = block width: a wave carries 64 / W items), without a workgroup barrier: a lane
// owns a column of the block and streams down the rows of the staged reference window; the horizontal filter of a row is NT / 2
// v_dot2_i32_i16 over the NT samples of the row (aligned dword reads, funnel-shifted by the lane's parity), its result enters a register ring of row PAIRS, and the vertical
// filter of an output row is NT / 2 v_dot2_i32_i16 over that ring — no intermediate plane in LDS.  The 2-D form is used for every
// position of a mixed wave: with the unit kernel of phase 0 on one axis it returns exactly what svt_av1_(highbd_)convolve_x_sr /
// _y_sr / the plain copy return (round_0 = 3, round_1 = 11: the offsets cancel and rnd(rnd(s, 3), 4), rnd(s, 7) fall out); waves
// whose items all lack one axis take the cheaper 1-D / copy loops.
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int32_t dot2(uint32_t a, uint32_t b, int32_t c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b), c, false);
}
// NT consecutive 16-bit samples of L.win from element e on, as NT / 2 packed pairs.  An LDS access that is not naturally aligned
// is executed one lane per cycle on gfx950 (tools/ubench/lds_unaligned.hip: 65 cycles per wave-instruction whatever its width,
// against 5 - 7 aligned), so the lane reads the aligned dwords around its samples and funnel-shifts them by its parity.
template <int NT>
__device__ __forceinline__ void lds_pairs(const uint16_t *win, int e, uint32_t sh, uint32_t (&v)[NT / 2]) {
    const uint32_t *q = (const uint32_t *)win + (e >> 1);
    uint32_t        d[NT / 2 + 1];
#pragma unroll
    for (int k = 0; k <= NT / 2; k++) d[k] = q[k];
#pragma unroll
    for (int k = 0; k < NT / 2; k++) v[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);
}
// 8-bit layout: sum of f[k] * x[k] over the eight samples from byte index e on (x stored as x - 128), + init.  f8 = the phase's int8 taps
// (zeros for phase 0), k_unit = 128 on lanes whose phase is 0 (the unit tap multiplies the centre sample, byte 3), else 0.
__device__ __forceinline__ int32_t hsum8_bytes(const uint8_t *win, int e, uint32_t shb, uint2 f8, int32_t k_unit, int32_t init) {
    const uint32_t *q  = (const uint32_t *)win + (e >> 2);
    const uint32_t  d0 = q[0], d1 = q[1], d2 = q[2];
    const uint32_t  lo = __builtin_amdgcn_alignbyte(d1, d0, shb), hi = __builtin_amdgcn_alignbyte(d2, d1, shb);
    int32_t acc = __builtin_amdgcn_sdot4((int)lo, (int)f8.x, init + 16384, false);  // sum f = 128: + 128 * 128 undoes the - 128 of the samples
    acc         = __builtin_amdgcn_sdot4((int)hi, (int)f8.y, acc, false);
    return acc + k_unit * (int32_t)__builtin_amdgcn_sbfe((int)lo, 24, 8);
}
// the NT non-zero taps of a phase as packed pairs in tap order (bilinear: taps 3 and 4 straddle two dwords of the 8-tap row)
template <int NT>
__device__ __forceinline__ void tap_pairs(const int16_t *row8, uint32_t (&f)[NT / 2]) {
    uint32_t d[4];
    __builtin_memcpy(d, row8, 16);
    if (NT == 8) {
#pragma unroll
        for (int i = 0; i < NT / 2; i++) f[i] = d[i];
    } else {
        f[0] = (d[1] >> 16) | (d[2] << 16);
    }
}

template <int W>
struct Geo {
    static constexpr int WD = W + 2 * WIN_MARGIN;      // window width = height = pitch
    static constexpr int WDP = WD > 64 ? 128 : 64;      // the staging loop's row length (a power of two)
    static constexpr int G = 64 / W;                    // items per wave
    static constexpr int LB = W == 64 ? 6 : (W == 32 ? 5 : (W == 16 ? 4 : 3));
};

// evaluate the item of this lane's group (valid == false: an idle group); returns the distortion on every lane of the group
template <bool S16, int W, int NT>
__device__ uint64_t eval_item(const LdsT<S16> &L, const Item &it, bool valid, int kidx, int bd, bool centre, int sss) {
    constexpr int WD = Geo<W>::WD, LB = Geo<W>::LB, K0 = (8 - NT) / 2;
    const int     lane = threadIdx.x & 63, c = lane & (W - 1);
    const int     vshift = sss, rstep = centre ? 1 << sss : 1, n_out = W >> vshift, hi = (1 << bd) - 1;
    const bool    has_h = __ballot(valid && it.sx) != 0, has_v = __ballot(valid && it.sy) != 0;  // uniform over the wave
    const Blk    &B = L.blk[it.blk];
    const int win0 = it.blk * (WD * WD) + it.py * WD + it.px + c;  // block sample (c, 0) of the candidate in L.win
    const int src0 = B.ly * 64 + B.lx + c;
    uint32_t fx[NT / 2], fy[NT / 2];
    tap_pairs<NT>(L.taps[kidx][it.sx], fx);
    tap_pairs<NT>(L.taps[kidx][it.sy], fy);
    int32_t  sum = 0;
    uint32_t sse = 0;
    const uint32_t psh = ((win0 - 3 + K0) & 1) * 16;  // the pitch is even: the parity of a lane's first sample is the same in every row
    // 8-bit layout: the eight bytes from sample x - 3 on; the pitch is a multiple of 4, so the byte phase is the same in every row
    const uint32_t shb = (uint32_t)(win0 - 3) & 3u;
    uint2          f8  = {0u, 0u};
    int32_t        k_unit = 0;
    if (!S16) {
        __builtin_memcpy(&f8, L.taps8[kidx][it.sx], 8);
        k_unit = it.sx ? 0 : 128;
    }
    auto raw = [&](int idx) -> int32_t { return S16 ? (int32_t)L.win[idx] : (int32_t)(L.win[idx] ^ 0x80); };
    if (!has_v) {
        // rows o = i << vshift straight from the window: copy or horizontal filter
        int wo = win0 - 3 + K0, so = src0;
        for (int i = 0; i < n_out; i++, wo += WD << vshift, so += 64 << vshift) {
            int32_t p;
            if (has_h) {
                int32_t acc;
                if constexpr (S16) {
                    uint32_t x[NT / 2];
                    lds_pairs<NT>((const uint16_t *)L.win, wo, psh, x);
                    acc = 4;
#pragma unroll
                    for (int k = 0; k < NT / 2; k++) acc = dot2(x[k], fx[k], acc);
                } else {
                    acc = hsum8_bytes((const uint8_t *)L.win, wo - K0, shb, f8, k_unit, 4);
                }
                p = ((acc >> 3) + 8) >> 4;
                p = p < 0 ? 0 : (p > hi ? hi : p);
            } else {
                p = raw(wo + 3 - K0);
            }
            const int32_t d = p - (int32_t)L.src[so];
            sum += d, sse += (uint32_t)(d * d);
        }
    } else {
        // stream of window rows py + (K0 - 3 + j) * rstep; output row o = (j - NT + 1) * rstep once NT rows are in the ring
        const int       n = centre ? n_out + NT - 1 : ((n_out - 1) << vshift) + NT, emit_mask = centre ? 0 : (1 << vshift) - 1;
        int             wo = win0 + (K0 - 3) * rstep * WD - 3 + K0, so = src0 - (NT - 1) * rstep * 64;
        const int32_t   c0 = (1 << (bd + 6)) + 4;             // horizontal: offset + rounding of >> 3
        const int32_t   c1 = 1024 - (1 << (bd + 10));         // vertical: rounding of >> 11 and the offsets of both passes
        uint32_t        ring[NT], prev = 0;
#pragma unroll
        for (int u = 0; u < NT; u++) ring[u] = 0;
        for (int jb = 0; jb < n; jb += NT) {
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const int j = jb + u;
                if (j < n) {
                    uint32_t h;
                    if (has_h) {
                        int32_t acc;
                        if constexpr (S16) {
                            uint32_t x[NT / 2];
                            lds_pairs<NT>((const uint16_t *)L.win, wo, psh, x);
                            acc = c0;
#pragma unroll
                            for (int k = 0; k < NT / 2; k++) acc = dot2(x[k], fx[k], acc);
                        } else {
                            acc = hsum8_bytes((const uint8_t *)L.win, wo - K0, shb, f8, k_unit, c0);
                        }
                        h = (uint32_t)(acc >> 3);
                    } else {
                        h = (uint32_t)raw(wo + 3 - K0);
                    }
                    ring[u] = (h << 16) | prev, prev = h;
                    if (j >= NT - 1 && ((j - (NT - 1)) & emit_mask) == 0) {
                        int32_t acc = has_h ? c1 : 64;
#pragma unroll
                        for (int k = 0; k < NT / 2; k++) acc = dot2(ring[(u + 2 + 2 * k) % NT], fy[k], acc);
                        int32_t p = has_h ? acc >> 11 : acc >> 7;
                        p = p < 0 ? 0 : (p > hi ? hi : p);
                        const int32_t d = p - (int32_t)L.src[so];
                        sum += d, sse += (uint32_t)(d * d);
                    }
                    wo += rstep * WD, so += rstep * 64;
                }
            }
        }
    }
    // sums over the W lanes of the group (a lane holds at most 64 samples of |difference| < 2^10: 2^26 per lane; a 64x64 block of a
    // 10-bit picture can pass 2^32 in total)
    uint64_t tsse = sse;
    int32_t  tsum = sum;
    if (S16 && W == 64) {
#pragma unroll
        for (int off = W / 2; off > 0; off >>= 1) tsum += __shfl_xor(tsum, off, 64), tsse += __shfl_xor(tsse, off, 64);
    } else {
        uint32_t e = sse;
#pragma unroll
        for (int off = W / 2; off > 0; off >>= 1) tsum += __shfl_xor(tsum, off, 64), e += __shfl_xor(e, off, 64);
        tsse = e;
    }
    const int ln = 2 * LB - vshift;  // n = W * (W >> vshift) samples: a power of two, and the squares are not negative: / n is a shift
    uint64_t  var;
    if (!S16) {  // svt_aom_variance*_c
        var = (uint32_t)((uint32_t)tsse - (uint32_t)(((int64_t)tsum * tsum) >> ln));
    } else {  // svt_aom_highbd_10_variance*_c
        const uint32_t e = (uint32_t)((tsse + 8) >> 4);
        const int32_t  m = (int32_t)(((int64_t)tsum + 2) >> 2);
        const int64_t  v = (int64_t)e - (((int64_t)m * m) >> ln);
        var = v >= 0 ? (uint32_t)v : 0;
    }
    return var << vshift;
}

// tf_subpel_search of the nblk (1 or 4) square blocks of width W described by L.blk[]: in  .best / .mvx / .mvy / .lx / .ly,
// out .best / .mvx / .mvy.  Called by the whole workgroup; the results are visible to every lane on return.
template <bool S16, int W, int NT>
__device__ void search_blocks(LdsT<S16> &L, const SearchCtx &s, int nblk, int kernel) {
    constexpr int WD = Geo<W>::WD, WDP = Geo<W>::WDP, G = Geo<W>::G;
    const SvtHipTfCtrls &c = s.a->ctrls;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lbn = nblk == 4 ? 2 : 0;
    const int kidx = kernel == K_BILINEAR ? 1 : 0;
    __syncthreads();  // L.blk[] is set up; the previous search is done with the windows
    if (tid < nblk) {  // window around the (clamped) starting vector
        Blk &B = L.blk[tid];
        int  col0, row0;
        clamp_mv(s.a->mi_rows, s.a->mi_cols, W, W, B.mvx, B.mvy, 0, s.ox + B.lx, s.oy + B.ly, W, col0, row0);
        B.wx = (int16_t)(B.lx + (col0 >> 4) - WIN_MARGIN), B.wy = (int16_t)(B.ly + (row0 >> 4) - WIN_MARGIN), B.done = 0;
    }
    __syncthreads();
    {
        // the margin rows / columns nobody reads may lie outside the padded plane: keep the address inside it
        const int32_t lo = (int32_t)s.a->idx_min, hi = (int32_t)s.a->idx_max;
        for (int i = tid; i < nblk * WD * WDP; i += 256) {
            const int cc = i & (WDP - 1), rr = i / WDP, b = rr / WD, r = rr - b * WD;
            if (cc < WD) {
                const Blk &B = L.blk[b];
                int32_t idx = (s.oy + B.wy + r) * (int32_t)s.ref_stride + s.ox + B.wx + cc;
                idx         = idx < lo ? lo : (idx > hi ? hi : idx);
                const uint32_t v = ldg<S16>(s.ref0, (ptrdiff_t)idx);
                L.win[b * (WD * WD) + r * WD + cc] = (typename LdsT<S16>::Pix)(S16 ? v : (v ^ 0x80u));
            }
        }
    }
    for (int round = -1; round < 3; round++) {
        const int mode = round <= 0 ? c.half_pel_mode : (round == 1 ? c.quarter_pel_mode : c.eight_pel_mode);
        if (round >= 0 && !mode)
            continue;
        const int step = round < 0 ? 0 : (4 >> round);
        // the candidates in the reference's order (i = x outer, j = y inner over -1 .. 1; mode >= 2: svt_check_position drops the diagonals)
        const int nc = round < 0 ? 1 : (mode >= 2 ? 4 : 8), nitems = nc << lbn;
        if (tid < nitems) {
            const int  k = tid >> lbn, b = tid & (nblk - 1), q = round < 0 ? 4 : (mode >= 2 ? 2 * k + 1 : (k < 4 ? k : k + 1));
            const int  xd = (q / 3 - 1) * step, yd = (q % 3 - 1) * step;
            const Blk &B = L.blk[b];
            Item       it;
            it.mvx = (int16_t)(B.mvx + xd), it.mvy = (int16_t)(B.mvy + yd);
            int col, row;
            clamp_mv(s.a->mi_rows, s.a->mi_cols, W, W, it.mvx, it.mvy, 0, s.ox + B.lx, s.oy + B.ly, W, col, row);
            it.px = (int16_t)(B.lx + (col >> 4) - B.wx), it.py = (int16_t)(B.ly + (row >> 4) - B.wy);
            it.sx = (uint8_t)(col & 15), it.sy = (uint8_t)(row & 15), it.blk = (uint8_t)b, it.on = !B.done;
            L.item[tid] = it;
        }
        __syncthreads();  // items (and, in the first round, the windows)
        for (int base = wave * G; base < nitems; base += 4 * G) {
            const int  idx = base + lane / W;
            const bool valid = idx < nitems && L.item[idx < nitems ? idx : 0].on;
            if (__ballot(valid) == 0)
                continue;
            const Item     it = L.item[valid ? idx : 0];
            const uint64_t d  = eval_item<S16, W, NT>(L, it, valid, kidx, s.bd, round < 0, c.sub_sampling_shift);
            if (valid && (lane & (W - 1)) == 0)
                L.dist[idx] = d;
        }
        __syncthreads();
        if (tid < nblk) {  // the reference's walk over the candidates of this round
            Blk           &B = L.blk[tid];
            uint64_t       best = B.best;
            int            bx = B.mvx, by = B.mvy;
            const uint64_t bound = c.subpel_early_exit_th ? (((uint64_t)(W * W) * c.subpel_early_exit_th) << s.is16) : 0;
            if (!B.done) {
                for (int k = 0; k < nc; k++) {
                    const int t = (k << lbn) + tid;
                    if (best == 0 || best < bound)
                        continue;
                    const uint64_t d = L.dist[t];
                    if (d < best)
                        best = d, bx = L.item[t].mvx, by = L.item[t].mvy;
                }
            }
            B.best = best, B.mvx = bx, B.mvy = by, B.done = best == 0 || best < bound;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int mvx_of(uint32_t mv) { return (int16_t)(mv & 0xffff); }
__device__ __forceinline__ int mvy_of(uint32_t mv) { return (int16_t)(mv >> 16); }

// the static part of one SvtHipTfBlock (32x32 block q of b64 b)
__device__ void fill_block(const RefineArgs &a, uint32_t b, int q, const uint8_t *pred, SvtHipTfBlock &t) {
    const int is16 = a.bit_depth > 8, lx = (q & 1) * 32, ly = (q >> 1) * 32;
    const int ox = (int)(b % a.bw) * 64, oy = (int)(b / a.bw) * 64;
    memset(&t, 0, sizeof(t));
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const int       ss = p ? 1 : 0, ps = p ? 32 : 64;
        const size_t    stride = p ? a.centre.stride_c : a.centre.stride;
        const size_t    org = (size_t)((oy + ly) >> ss) * stride + ((ox + lx) >> ss);
        const uint8_t  *base = is16 ? (const uint8_t *)(p ? a.centre.c16[p - 1] : a.centre.y16) : (p ? a.centre.c8[p - 1] : a.centre.y8);
        const size_t    po = (size_t)(ly >> ss) * ps + (lx >> ss);
        t.src[p]   = base + (org << is16);
        t.pred[p]  = pred + ((((size_t)b * 3 + p) * 4096 + po) << is16);
        t.accum[p] = a.accum + ((size_t)b * 3 + p) * 4096 + po;
        t.count[p] = a.count + ((size_t)b * 3 + p) * 4096 + po;
        t.src_stride[p] = (uint32_t)stride, t.pred_stride[p] = (uint32_t)ps;
        t.decay_factor_fp16[p] = a.decay[p];
    }
    t.mv_dist_th = a.mv_dist_th, t.chroma = a.chroma, t.ss_x = t.ss_y = 1;
    t.is_16bit = (uint8_t)is16, t.bit_depth = a.bit_depth, t.zz_based = a.ctrls.use_zz_based_filter;
}

// E8: with the 8x8 stage of tf level 1 (its own instance: the extra search keeps 26 more registers live across the whole kernel)
template <bool S16, bool E8>
__global__ __launch_bounds__(256) void tf_refine_kernel(RefineArgs a, const RefineRef *__restrict__ refs, uint32_t *__restrict__ tot) {
    __shared__ LdsT<S16>        L;
    // the block's state record; without the 8x8 stage only the part in front of err8 lives in LDS (the whole record cost 0.12 ms per picture)
    constexpr int ST_BYTES = E8 ? (int)sizeof(SvtHipTfB64State) : (int)offsetof(SvtHipTfB64State, err8);
    static_assert(ST_BYTES % 8 == 0, "state record prefix");
    __shared__ uint64_t st_mem[ST_BYTES / 8];
    SvtHipTfB64State   &st = *reinterpret_cast<SvtHipTfB64State *>(st_mem);
    const RefineRef    &R = refs[blockIdx.y];  // a by-value copy went to scratch memory: its plane arrays are indexed at run time
    const uint32_t      b = blockIdx.x;
    const int           tid = threadIdx.x, ox = (int)(b % a.bw) * 64, oy = (int)(b / a.bw) * 64;
    const SvtHipTfCtrls &c = a.ctrls;
    SearchCtx s;
    s.a = &a, s.ox = ox, s.oy = oy, s.is16 = S16, s.bd = S16 ? a.bit_depth : 8;
    s.ref0 = S16 ? (const void *)R.pic.y16 : (const void *)R.pic.y8, s.ref_stride = R.pic.stride;
    const void *src0 = S16 ? (const void *)a.centre.y16 : (const void *)a.centre.y8;
    for (int i = tid; i < 64 * 64; i += 256) L.src[i] = (typename LdsT<S16>::Pix)ldg<S16>(src0, (ptrdiff_t)(oy + (i >> 6)) * a.centre.stride + ox + (i & 63));
    for (int i = tid; i < ST_BYTES / 4; i += 256) ((uint32_t *)&st)[i] = 0;
    L.taps[tid >> 7][(tid >> 3) & 15][tid & 7] = TF_KERNELS[(tid >> 7) ? K_BILINEAR : K_REGULAR][(tid >> 3) & 15][tid & 7];
    L.taps8[tid >> 7][(tid >> 3) & 15][tid & 7] = ((tid >> 3) & 15) ? (int8_t)TF_KERNELS[(tid >> 7) ? K_BILINEAR : K_REGULAR][(tid >> 3) & 15][tid & 7] : (int8_t)0;
    const SvtHipMeSearchResult sr = R.sr[(size_t)b * 8];
    const uint32_t *best_mv = R.best_mv + (size_t)b * 8 * 85, *best_sad = R.best_sad + (size_t)b * 8 * 85;
    if (tid == 0 && tot)
        atomicAdd(&tot[abs((int)sr.hme_sc_x) > abs((int)sr.hme_sc_y) ? 0 : 1], 1u);
    // svt_aom_motion_estimation_b64 leaves after HME when the HME distortion is below tf_me_exit_th (motion_estimation.c:3179)
    const int use64_th = sr.hme_sad < a.tf_me_exit_th ? 255 : c.use_pred_64x64_only_th;
    // ---- tf_64x64_sub_pel_search
    if (tid == 0) {
        Blk &B = L.blk[0];
        B.best = 0x7fffffff, B.lx = B.ly = 0;
        B.mvx = (int16_t)((use64_th == 255 ? sr.hme_sc_x : mvx_of(best_mv[0])) << 3);
        B.mvy = (int16_t)((use64_th == 255 ? sr.hme_sc_y : mvy_of(best_mv[0])) << 3);
    }
    if (c.use_2tap)
        search_blocks<S16, 64, 2>(L, s, 1, K_BILINEAR);
    else
        search_blocks<S16, 64, 8>(L, s, 1, K_REGULAR);
    const uint64_t err64 = L.blk[0].best;
    const int      mv64x = L.blk[0].mvx, mv64y = L.blk[0].mvy;
    bool use64 = false;
    if (use64_th) {
        if (use64_th == 255) {
            use64 = true;
        } else {  // tf_use_64x64_pred
            uint32_t d32 = 0;
            for (int i = 0; i < 4; i++) d32 += best_sad[1 + i];
            const int64_t x = best_sad[0] > 1 ? best_sad[0] : 1, y = d32 > 1 ? d32 : 1;
            use64 = ((x - y) * 100) / y < use64_th;
        }
    }
    if (!use64) {  // ---- tf_32x32_sub_pel_search: the four blocks side by side
        __syncthreads();  // everybody has read the 64x64 result
        if (tid < 4) {
            Blk &B = L.blk[tid];
            B.best = 0x7fffffff, B.lx = (int16_t)((tid & 1) * 32), B.ly = (int16_t)((tid >> 1) * 32);
            B.mvx = (int16_t)(mvx_of(best_mv[1 + tid]) << 3), B.mvy = (int16_t)(mvy_of(best_mv[1 + tid]) << 3);
        }
        if (c.use_2tap)
            search_blocks<S16, 32, 2>(L, s, 4, K_BILINEAR);
        else
            search_blocks<S16, 32, 8>(L, s, 4, K_REGULAR);
        uint64_t sum32 = 0;
        for (int i = 0; i < 4; i++) sum32 += L.blk[i].best;
        if (tid < 4)
            st.err32[tid] = L.blk[tid].best, st.mv32_x[tid] = (int16_t)L.blk[tid].mvx, st.mv32_y[tid] = (int16_t)L.blk[tid].mvy;
        if (err64 * 14 < sum32 * 16 && err64 < (1u << 18))
            use64 = true;
        __syncthreads();  // st.err32 -> every lane; L.blk[] may be rewritten
    }
    if (!use64) {
        for (int i = 0; i < 4; i++) {
            const uint64_t e32 = st.err32[i];
            if (e32 < c.pred_error_32x32_th)
                continue;  // split flag stays 0
            // tf_16x16_sub_pel_search (always the regular 8-tap kernel) of the four 16x16 blocks, derive_tf_32x32_block_split_flag without 8x8
            if (tid < 4) {
                Blk &B = L.blk[tid];
                B.best = 0x7fffffff, B.lx = (int16_t)((i & 1) * 32 + (tid & 1) * 16), B.ly = (int16_t)((i >> 1) * 32 + (tid >> 1) * 16);
                B.mvx = (int16_t)(mvx_of(best_mv[5 + i * 4 + tid]) << 3), B.mvy = (int16_t)(mvy_of(best_mv[5 + i * 4 + tid]) << 3);
            }
            search_blocks<S16, 16, 8>(L, s, 4, K_REGULAR);
            int64_t sum16 = 0;
            for (int k = 0; k < 4; k++) sum16 += (int)L.blk[k].best;
            if (tid < 4)
                st.err16[i * 4 + tid] = L.blk[tid].best, st.mv16_x[i * 4 + tid] = (int16_t)L.blk[tid].mvx, st.mv16_y[i * 4 + tid] = (int16_t)L.blk[tid].mvy;
            if (!E8 || !c.enable_8x8_pred) {
                if (tid == 0)
                    st.split32[i] = !((int)e32 * 14 < (int)sum16 * 16);
                __syncthreads();  // L.blk[] is rewritten by the next 32x32 block
                continue;
            }
            // tf_8x8_sub_pel_search (temporal_filtering.c:2106-2224): the four 8x8 blocks of each 16x16 block side by side, eight candidates
            // per wave; then derive_tf_32x32_block_split_flag with the 16x16 -> 8x8 decisions (:236-285)
            for (int k = 0; k < 4; k++) {
                __syncthreads();  // L.blk[] has been read
                if (tid < 4) {
                    Blk      &B = L.blk[tid];
                    const int idx = i * 16 + k * 4 + tid;  // idx_32x32_to_idx_8x8 / tab8x8: z-order, the ME's 8x8 vectors in the same order
                    B.best = 0x7fffffff, B.lx = (int16_t)((i & 1) * 32 + (k & 1) * 16 + (tid & 1) * 8), B.ly = (int16_t)((i >> 1) * 32 + (k >> 1) * 16 + (tid >> 1) * 8);
                    B.mvx = (int16_t)(mvx_of(best_mv[21 + idx]) << 3), B.mvy = (int16_t)(mvy_of(best_mv[21 + idx]) << 3);
                }
                if constexpr (E8)
                    search_blocks<S16, 8, 8>(L, s, 4, K_REGULAR);
                if (tid < 4) {
                    const int idx = i * 16 + k * 4 + tid;
                    st.err8[idx] = L.blk[tid].best, st.mv8_x[idx] = (int16_t)L.blk[tid].mvx, st.mv8_y[idx] = (int16_t)L.blk[tid].mvy;
                }
            }
            __syncthreads();
            if (tid == 0) {
                int sum = 0;
                for (int k = 0; k < 4; k++) {
                    const int q = i * 4 + k;
                    int       sub = (int)st.err16[q], e8 = 0;
                    for (int e = 0; e < 4; e++) e8 += (int)st.err8[q * 4 + e];
                    if (sub * 8 < e8 * 16) {
                        st.split16[q] = 0;
                    } else {
                        st.split16[q] = 1, st.err16[q] = (uint64_t)e8, sub = e8;
                    }
                    sum += sub;
                }
                st.split32[i] = !((int)e32 * 14 < sum * 16);
            }
            __syncthreads();  // L.blk[] is rewritten by the next 32x32 block
        }
    }
    __syncthreads();
    if (tid == 0)
        st.err64 = err64, st.mv64_x = (int16_t)mv64x, st.mv64_y = (int16_t)mv64y, st.use_64x64 = use64;
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(SvtHipTfB64State) / 4); i += 256) ((uint32_t *)&R.state[b])[i] = i < ST_BYTES / 4 ? ((const uint32_t *)&st)[i] : 0u;
    // ---- descriptors of the final predictions (sharp kernel, the filter's bit depth): 16 luma + 16 + 16 chroma slots
    if (tid < DESC_PER_B64) {
        const int plane = tid / 16, idx = tid % 16, q = idx >> 2, k = idx & 3;
        const int is16 = a.bit_depth > 8;
        int       lx, ly, bsz, mx, my;
        bool      on;
        if (st.use_64x64) {
            on = idx == 0, lx = ly = 0, bsz = 64, mx = st.mv64_x, my = st.mv64_y;
        } else if (st.split32[q]) {
            on = !(E8 && st.split16[idx]) /* else four 8x8 blocks: tf_predict8_kernel */, lx = (q & 1) * 32 + (k & 1) * 16, ly = (q >> 1) * 32 + (k >> 1) * 16, bsz = 16, mx = st.mv16_x[idx], my = st.mv16_y[idx];
        } else {
            on = k == 0, lx = (q & 1) * 32, ly = (q >> 1) * 32, bsz = 32, mx = st.mv32_x[q], my = st.mv32_y[q];
        }
        on = on && (plane == 0 || a.chroma);
        SvtHipConvolveDesc d;
        memset(&d, 0, sizeof(d));
        if (on) {
            const int ss = plane ? 1 : 0, bw = bsz >> ss;
            int       col, row;
            clamp_mv(a.mi_rows, a.mi_cols, bw, bw, mx, my, ss, ox + lx, oy + ly, bsz, col, row);
            const int pre_x = plane ? (((ox + lx) >> 3) << 3) / 2 : ox + lx, pre_y = plane ? (((oy + ly) >> 3) << 3) / 2 : oy + ly;
            const int dx = plane ? ((lx >> 3) << 3) / 2 : lx, dy = plane ? ((ly >> 3) << 3) / 2 : ly, ps = plane ? 32 : 64;
            const uint32_t stride = plane ? R.pic.stride_c : R.pic.stride;
            const uint8_t *base = is16 ? (const uint8_t *)(plane ? R.pic.c16[plane - 1] : R.pic.y16) : (plane ? R.pic.c8[plane - 1] : R.pic.y8);
            const int sx = col & 15, sy = row & 15;
            d.src        = base + ((((ptrdiff_t)(pre_y + (row >> 4))) * (ptrdiff_t)stride + pre_x + (col >> 4)) * (is16 ? 2 : 1));
            d.dst        = R.pred + ((((size_t)b * 3 + plane) * 4096 + (size_t)dy * ps + dx) << is16);
            d.src_stride = stride, d.dst_stride = (uint32_t)ps, d.w = d.h = (uint16_t)bw;
            for (int t = 0; t < 8; t++) d.filter_x[t] = TF_KERNELS[K_SHARP][sx][t], d.filter_y[t] = TF_KERNELS[K_SHARP][sy][t];
            d.taps_x = sx ? 8 : 0, d.taps_y = sy ? 8 : 0, d.round_0 = 3, d.round_1 = 11;
            d.bit_depth = a.bit_depth, d.is_16bit = (uint8_t)is16;
        }
        R.desc[(size_t)b * DESC_PER_B64 + tid] = d;
    }
}

// the final predictions of one (reference picture, 64x64 block): up to 48 descriptors in fixed slots.  The blocks wider than 16
// samples (at most four per 64x64 block: one 64x64 + its two chroma blocks, or four 32x32) one after the other by a whole workgroup
// through the tile function of inter_convolve.hip; the small ones (16x16 luma, 16x16 / 8x8 chroma: up to 48) by four one-wave
// workgroups with a 2 KB tile each, so that a 64x64 block's slots run side by side instead of one after the other.  Both kernels
// pick their slots from one look at all 48 widths: walking the slots with one dependent descriptor load each cost ~2 us per slot
// (1.39 -> 0.45 ms at 4K for both together).
// (Handing the fixed-slot arrays to svt_hip_convolve_batch works too, but three quarters of its 256-thread, 19 KB workgroups
// would find an empty slot: 2.3 ms per reference picture at 4K.)
constexpr int SMALL_T = 16, SMALL_P = SMALL_T + 8;
__global__ __launch_bounds__(256) void tf_predict_kernel(const RefineRef *__restrict__ refs) {
    __shared__ alignas(4) uint16_t in[(conv::TILE + 7) * conv::IP + conv::CONV_IN_SLACK];
    __shared__ int16_t  im[(conv::TILE + 7) * conv::TILE];
    const SvtHipConvolveDesc *descs = refs[blockIdx.y].desc + (size_t)blockIdx.x * DESC_PER_B64;
    // which slots are this kernel's: every wave looks at all 48 widths at once (walking the slots one dependent descriptor load
    // after the other cost ~2 us per slot, 100 us per workgroup)
    const int lane = threadIdx.x & 63;
    uint64_t  todo = __ballot(lane < DESC_PER_B64 && descs[lane < DESC_PER_B64 ? lane : 0].w > SMALL_T);
    for (; todo; todo &= todo - 1) {
        const SvtHipConvolveDesc d = descs[__builtin_ctzll(todo)];
        __syncthreads();  // the previous descriptor's readers are done with the LDS buffers
        conv::convolve_tile(d, 0, in, im);
    }
}
constexpr int SMALL_SPLIT = 4;  // one-wave workgroups per 64x64 block: its small slots are dealt round-robin to them
__global__ __launch_bounds__(64) void tf_predict_small_kernel(const RefineRef *__restrict__ refs) {
    __shared__ alignas(4) uint16_t in[(SMALL_T + 7) * SMALL_P + conv::CONV_IN_SLACK];
    __shared__ int16_t  im[(SMALL_T + 7) * SMALL_T];
    const SvtHipConvolveDesc *descs = refs[blockIdx.y].desc + (size_t)(blockIdx.x / SMALL_SPLIT) * DESC_PER_B64;
    const int      lane = threadIdx.x, part = blockIdx.x % SMALL_SPLIT;
    const uint32_t w    = lane < DESC_PER_B64 ? descs[lane].w : 0;
    int            k    = 0;
    for (uint64_t todo = __ballot(w != 0 && w <= SMALL_T); todo; todo &= todo - 1, k++) {
        if (k % SMALL_SPLIT != part)
            continue;
        const SvtHipConvolveDesc d = descs[__builtin_ctzll(todo)];
        __syncthreads();
        conv::convolve_tile_t<64, SMALL_T, SMALL_P>(d, 0, in, im);
    }
}

// tf_32x32_inter_prediction for the 16x16 blocks that were split into 8x8 (temporal_filtering.c:2384-2445, enable_8x8_pred): 8x8 luma
// with the sharp 8-tap kernels, 4x4 chroma with the 4-tap kernels of narrow blocks.  One workgroup per (reference picture, 64x64 block);
// a wave per 8x8 block: lane = luma sample, lanes 0 .. 15 / 16 .. 31 also one Cb / Cr sample.  Each lane filters its eight rows itself
// (2-D form with the unit kernel where an axis has no fraction: equal to the 1-D functions, see eval_item) — tf level 1 only.
__device__ __forceinline__ int32_t predict_sample(const void *plane, uint32_t stride, int is16, int bd, int px, int py, int sx, int sy, int kernel) {
    int32_t v = (1 << (bd + 11)) + 1024 - (((1 << bd) + (1 << (bd - 1))) << 11);
#pragma unroll
    for (int t = 0; t < 8; t++) {
        int32_t h = (1 << (bd + 6)) + 4;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const ptrdiff_t i = (ptrdiff_t)(py - 3 + t) * stride + px - 3 + u;
            h += (int32_t)TF_KERNELS[kernel][sx][u] * (int32_t)(is16 ? ldg<true>(plane, i) : ldg<false>(plane, i));
        }
        v += (int32_t)TF_KERNELS[kernel][sy][t] * (int32_t)(int16_t)(h >> 3);
    }
    v >>= 11;
    const int32_t hi = (1 << bd) - 1;
    return v < 0 ? 0 : (v > hi ? hi : v);
}
__global__ __launch_bounds__(256) void tf_predict8_kernel(RefineArgs a, const RefineRef *__restrict__ refs) {
    const RefineRef        &R = refs[blockIdx.y];
    const uint32_t          b = blockIdx.x;
    const SvtHipTfB64State &st = R.state[b];
    if (st.use_64x64)
        return;
    const int tid = threadIdx.x, lane = tid & 63, e = tid >> 6, ox = (int)(b % a.bw) * 64, oy = (int)(b / a.bw) * 64, is16 = a.bit_depth > 8;
    for (int q16 = 0; q16 < 16; q16++) {
        if (!st.split32[q16 >> 2] || !st.split16[q16])
            continue;
        const int i = q16 >> 2, k = q16 & 3, idx = q16 * 4 + e;
        const int lx = (i & 1) * 32 + (k & 1) * 16 + (e & 1) * 8, ly = (i >> 1) * 32 + (k >> 1) * 16 + (e >> 1) * 8, mx = st.mv8_x[idx], my = st.mv8_y[idx];
        {
            int col, row;
            clamp_mv(a.mi_rows, a.mi_cols, 8, 8, mx, my, 0, ox + lx, oy + ly, 8, col, row);
            const int   r = lane >> 3, cc = lane & 7;
            const void *pl = is16 ? (const void *)R.pic.y16 : (const void *)R.pic.y8;
            const int32_t v = predict_sample(pl, R.pic.stride, is16, a.bit_depth, ox + lx + (col >> 4) + cc, oy + ly + (row >> 4) + r, col & 15, row & 15, K_SHARP);
            const size_t  o = ((size_t)b * 3) * 4096 + (size_t)(ly + r) * 64 + lx + cc;
            if (is16)
                ((uint16_t *)R.pred)[o] = (uint16_t)v;
            else
                R.pred[o] = (uint8_t)v;
        }
        if (a.chroma && lane < 32) {
            const int plane = 1 + (lane >> 4), r = (lane >> 2) & 3, cc = lane & 3;
            int       col, row;
            clamp_mv(a.mi_rows, a.mi_cols, 4, 4, mx, my, 1, ox + lx, oy + ly, 8, col, row);
            const int   pre_x = (((ox + lx) >> 3) << 3) / 2, pre_y = (((oy + ly) >> 3) << 3) / 2, dx = ((lx >> 3) << 3) / 2, dy = ((ly >> 3) << 3) / 2;
            const void *pl = is16 ? (const void *)R.pic.c16[plane - 1] : (const void *)R.pic.c8[plane - 1];
            const int32_t v = predict_sample(pl, R.pic.stride_c, is16, a.bit_depth, pre_x + (col >> 4) + cc, pre_y + (row >> 4) + r, col & 15, row & 15, K_REGULAR4);
            const size_t  o = ((size_t)b * 3 + plane) * 4096 + (size_t)(dy + r) * 32 + dx + cc;
            if (is16)
                ((uint16_t *)R.pred)[o] = (uint16_t)v;
            else
                R.pred[o] = (uint8_t)v;
        }
    }
}

// produce_temporally_filtered_pic_ld (temporal_filtering.c:3533-3560): no motion search — tf_64x64_mv = 0 and tf_64x64_inter_prediction
// of the co-located block, which at a zero vector is a copy of the reference picture's samples (luma 64x64, chroma 32x32 each, the depth
// the filter works on).  Writes the prediction and the state record (use_64x64 = 1, vector 0); tf_blocks_kernel then measures the four
// 32x32 variances exactly as for a searched 64x64 block.
__global__ __launch_bounds__(256) void tf_low_delay_kernel(RefineArgs a, const RefineRef *__restrict__ refs) {
    const RefineRef &R = refs[blockIdx.y];
    const uint32_t   b = blockIdx.x;
    const int        tid = threadIdx.x, ox = (int)(b % a.bw) * 64, oy = (int)(b / a.bw) * 64, is16 = a.bit_depth > 8;
    for (int i = tid; i < (int)(sizeof(SvtHipTfB64State) / 4); i += 256) ((uint32_t *)&R.state[b])[i] = 0;
    __syncthreads();
    if (tid == 0)
        R.state[b].use_64x64 = 1;
    for (int p = 0; p < (a.chroma ? 3 : 1); p++) {
        const int       ss = p ? 1 : 0, n = 64 >> ss;
        const uint32_t  stride = p ? R.pic.stride_c : R.pic.stride;
        const void     *src = is16 ? (const void *)(p ? R.pic.c16[p - 1] : R.pic.y16) : (const void *)(p ? R.pic.c8[p - 1] : R.pic.y8);
        uint8_t        *dst = R.pred + ((((size_t)b * 3 + p) * 4096) << is16);
        for (int i = tid; i < n * n; i += 256) {
            const int      r = i / n, c = i - r * n;
            const uint32_t v = is16 ? ldg<true>(src, (ptrdiff_t)((oy >> ss) + r) * stride + (ox >> ss) + c) : ldg<false>(src, (ptrdiff_t)((oy >> ss) + r) * stride + (ox >> ss) + c);
            if (is16)
                ((uint16_t *)dst)[r * n + c] = (uint16_t)v;
            else
                dst[r * n + c] = (uint8_t)v;
        }
    }
}

// after the predictions: convert_64x64_info_to_32x32_info for the blocks predicted as one 64x64, and the four SvtHipTfBlock
// records of every block.  blockIdx.y == n_refs writes the reference-independent records used by the central / normalise
// launches (and their SvtHipTfOut).
__global__ __launch_bounds__(256) void tf_blocks_kernel(RefineArgs a, const RefineRef *__restrict__ refs, uint32_t n_refs,
                                                        SvtHipTfBlock *__restrict__ static_blocks, SvtHipTfOut *__restrict__ outs) {
    __shared__ int64_t  rs[4][4];
    __shared__ uint64_t re[4][4];
    const uint32_t b = blockIdx.x;
    const int      tid = threadIdx.x;
    if (blockIdx.y == n_refs) {
        if (tid < 4) {
            SvtHipTfBlock t;
            fill_block(a, b, tid, refs[0].pred, t);
            static_blocks[(size_t)b * 4 + tid] = t;
            SvtHipTfOut o;
            memset(&o, 0, sizeof(o));
#pragma unroll
            for (int p = 0; p < 3; p++) o.dst[p] = (void *)t.src[p], o.dst_stride[p] = t.src_stride[p];
            outs[(size_t)b * 4 + tid] = o;
        }
        return;
    }
    const RefineRef  &R = refs[blockIdx.y];
    SvtHipTfB64State *st = &R.state[b];
    const int is16 = a.bit_depth > 8, ox = (int)(b % a.bw) * 64, oy = (int)(b / a.bw) * 64;
    if (st->use_64x64) {  // uniform over the workgroup
        const int      sh = a.ctrls.sub_sampling_shift;
        const uint8_t *pred = R.pred + (((size_t)b * 3) * 4096 << is16);
        const void    *src0 = is16 ? (const void *)a.centre.y16 : (const void *)a.centre.y8;
        int64_t        sum[4] = {0, 0, 0, 0};
        uint64_t       sse[4] = {0, 0, 0, 0};
        for (int i = tid; i < (64 >> sh) * 64; i += 256) {
            const int r = (i >> 6) << sh, cc = i & 63, q = (r >> 5) * 2 + (cc >> 5);
            const int32_t p = is16 ? ((const uint16_t *)pred)[r * 64 + cc] : pred[r * 64 + cc];
            const int32_t v = is16 ? ldg<true>(src0, (ptrdiff_t)(oy + r) * a.centre.stride + ox + cc) : ldg<false>(src0, (ptrdiff_t)(oy + r) * a.centre.stride + ox + cc);
            const int32_t d = p - v;
#pragma unroll
            for (int k = 0; k < 4; k++) sum[k] += k == q ? d : 0, sse[k] += k == q ? (uint32_t)(d * d) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum[k] += __shfl_xor(sum[k], off, 64), sse[k] += __shfl_xor(sse[k], off, 64);
            if ((tid & 63) == 0)
                rs[tid >> 6][k] = sum[k], re[tid >> 6][k] = sse[k];
        }
        __syncthreads();
        if (tid < 4) {
            const int64_t  tsum = rs[0][tid] + rs[1][tid] + rs[2][tid] + rs[3][tid];
            const uint64_t tsse = re[0][tid] + re[1][tid] + re[2][tid] + re[3][tid];
            const int32_t  n = 32 * (32 >> sh);
            uint64_t       var;
            if (!is16) {
                const int32_t s32 = (int32_t)tsum;
                var = (uint32_t)((uint32_t)tsse - (uint32_t)(((int64_t)s32 * s32) / n));
            } else {
                const uint32_t e = (uint32_t)((tsse + 8) >> 4);
                const int32_t  m = (int32_t)((tsum + 2) >> 2);
                const int64_t  v = (int64_t)e - (((int64_t)m * m) / n);
                var = v >= 0 ? (uint32_t)v : 0;
            }
            st->err32[tid] = var << sh, st->mv32_x[tid] = st->mv64_x, st->mv32_y[tid] = st->mv64_y, st->split32[tid] = 0;
        }
        __syncthreads();
    }
    if (tid < 4) {
        SvtHipTfBlock t;
        fill_block(a, b, tid, R.pred, t);
        t.split = st->split32[tid];
        if (t.split) {
#pragma unroll
            for (int k = 0; k < 4; k++) t.block_error[k] = st->err16[tid * 4 + k], t.mv_x[k] = st->mv16_x[tid * 4 + k], t.mv_y[k] = st->mv16_y[tid * 4 + k];
        } else {
            t.block_error[0] = st->err32[tid], t.mv_x[0] = st->mv32_x[tid], t.mv_y[0] = st->mv32_y[tid];
        }
        R.blocks[(size_t)b * 4 + tid] = t;
    }
}

struct Layout {  // workspace, in bytes from its start (all 256-byte aligned)
    size_t best_sad, best_mv, sr, me_scratch, states, desc, blocks, static_blocks, outs, pred, accum, count, refs, total;
    size_t per_ref_best, per_ref_sr, per_ref_state, per_ref_desc, per_ref_blocks, per_ref_pred;
};
size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
Layout layout(uint32_t width, uint32_t height, uint32_t n_refs, int px) {
    Layout       l;
    const size_t nb = (size_t)((width + 63) / 64) * ((height + 63) / 64), n = n_refs ? n_refs : 1;
    size_t       o = 0;
    l.per_ref_best = al(nb * 2 * 4 * 85 * 4), l.per_ref_sr = al(nb * 8 * sizeof(SvtHipMeSearchResult));
    l.per_ref_state = al(nb * sizeof(SvtHipTfB64State)), l.per_ref_desc = al(nb * DESC_PER_B64 * sizeof(SvtHipConvolveDesc));
    l.per_ref_blocks = al(nb * 4 * sizeof(SvtHipTfBlock)), l.per_ref_pred = al(nb * 3 * 4096 * px);
    l.best_sad = o, o += l.per_ref_best * n;
    l.best_mv = o, o += l.per_ref_best * n;
    l.sr = o, o += l.per_ref_sr * n;
    l.me_scratch = o, o += al(nb * 85 * 16 * 4);  // arrays the ME_MCTF mode never writes (all point here)
    l.states = o, o += l.per_ref_state * n;
    l.desc = o, o += l.per_ref_desc * n;
    l.blocks = o, o += l.per_ref_blocks * n;
    l.static_blocks = o, o += l.per_ref_blocks;
    l.outs = o, o += al(nb * 4 * sizeof(SvtHipTfOut));
    l.pred = o, o += l.per_ref_pred * n;
    l.accum = o, o += al(nb * 3 * 4096 * 4);
    l.count = o, o += al(nb * 3 * 4096 * 2);
    l.refs = o, o += al(SVT_HIP_TF_MAX_REFS * sizeof(RefineRef));
    l.total = o;
    return l;
}

PicPlanes planes_of(const SvtHipTfPic &p) {
    PicPlanes            q;
    const SvtHipPlane8  &f = p.pyr.full;
    const size_t         yo = (size_t)f.org_y * f.stride + f.org_x, co = (size_t)(f.org_y / 2) * p.chroma8_stride + f.org_x / 2;
    q.y8 = f.buf + yo, q.stride = f.stride, q.stride_c = p.chroma8_stride;
    for (int i = 0; i < 2; i++) q.c8[i] = p.chroma8[i] ? p.chroma8[i] + co : nullptr, q.c16[i] = p.hbd[i + 1] ? p.hbd[i + 1] + co : nullptr;
    q.y16 = p.hbd[0] ? p.hbd[0] + yo : nullptr;
    return q;
}

}  // namespace

extern "C" uint64_t svt_hip_tf_workspace_bytes(uint32_t width, uint32_t height, uint32_t n_refs) { return layout(width, height, n_refs, 2).total; }

extern "C" uint64_t svt_hip_tf_workspace_state_offset(uint32_t width, uint32_t height, uint32_t n_refs, uint32_t ref) {
    const Layout l = layout(width, height, n_refs, 2);
    return l.states + l.per_ref_state * ref;
}

extern "C" int32_t svt_hip_tf_filter_picture(const SvtHipTfPictureJob *job, void *stream) {
    auto bad = [](const char *m) {
        set_error("svt_hip_tf_filter_picture: %s", m);
        return (int32_t)SVT_HIP_ERR_BAD_PARAMETER;
    };
    if (!job)
        return bad("NULL job");
    const SvtHipPlane8 &f = job->centre.pyr.full;
    if (job->n_refs == 0 || job->n_refs > SVT_HIP_TF_MAX_REFS)
        return bad("n_refs must be 1 .. SVT_HIP_TF_MAX_REFS");
    if (job->bit_depth != 8 && job->bit_depth != 10)
        return bad("bit_depth must be 8 or 10");
    if (job->ctrls.enable_8x8_pred > 1 || job->ctrls.low_delay > 1)
        return bad("enable_8x8_pred / low_delay are 0 or 1");
    if (job->ctrls.sub_sampling_shift > 1 || job->ctrls.use_2tap > 1)
        return bad("sub_sampling_shift / use_2tap out of range");
    if (!f.buf || f.width < 64 || f.height < 64 || f.org_x < 68 || f.org_y < 68 || (f.org_x & 1) || (f.org_y & 1))
        return bad("centre picture: needs >= 64 x 64 samples and an even padding of >= 68 samples (the reference's 64 + 4)");
    // every 64x64 block, the vectors the clamp allows around it and the 8-tap margin must stay inside the padded planes
    const uint32_t cover_w = (f.width + 63) / 64 * 64 + 15, cover_h = (f.height + 63) / 64 * 64 + 15;
    if (f.width + f.org_x < cover_w || f.height + f.org_y < cover_h || f.stride < f.width + 2u * f.org_x)
        return bad("centre picture: padding does not cover the 64-aligned area + 15 samples");
    const bool is16 = job->bit_depth > 8, chroma = job->chroma != 0;
    auto pic_ok = [&](const SvtHipTfPic &p) {
        const SvtHipPlane8 &g = p.pyr.full;
        if (!g.buf || g.width != f.width || g.height != f.height || g.org_x != f.org_x || g.org_y != f.org_y || g.stride != f.stride)
            return false;
        if (p.chroma8_stride != job->centre.chroma8_stride)
            return false;
        if (chroma && !is16 && (!p.chroma8[0] || !p.chroma8[1]))
            return false;
        if (is16 && (!p.hbd[0] || (chroma && (!p.hbd[1] || !p.hbd[2]))))
            return false;
        return true;
    };
    if (!pic_ok(job->centre))
        return bad("centre picture: planes missing");
    if (chroma && job->centre.chroma8_stride * 2 != f.stride)
        return bad("chroma stride must be half the luma stride (apply_filtering_central, temporal_filtering.c:356)");
    for (uint32_t r = 0; r < job->n_refs; r++)
        if (!pic_ok(job->ref[r]))
            return bad("reference picture: geometry differs from the centre picture or planes missing");
    const Layout l = layout(f.width, f.height, job->n_refs, 2);
    if (!job->workspace || job->workspace_bytes < layout(f.width, f.height, job->n_refs, 2).total)
        return bad("workspace too small (svt_hip_tf_workspace_bytes)");
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t    st = resolve_stream(stream);
    uint8_t       *ws = (uint8_t *)job->workspace;
    const uint32_t bw = (f.width + 63) / 64, nb = bw * ((f.height + 63) / 64);

    // ---- ME_MCTF of the centre picture against every reference picture: one launch
    SvtHipMeFrameJob *mj = new SvtHipMeFrameJob[job->n_refs];
    RefineRef         refs[SVT_HIP_TF_MAX_REFS];
    for (uint32_t r = 0; r < job->n_refs; r++) {
        memset(&mj[r], 0, sizeof(mj[r]));
        mj[r].prm = job->me;
        mj[r].prm.me_mctf = 1, mj[r].prm.num_of_list_to_search = 1;
        mj[r].prm.num_of_ref_pic_to_search[0] = 1, mj[r].prm.num_of_ref_pic_to_search[1] = 0;
        mj[r].prm.picture_number = job->centre.picture_number, mj[r].prm.ref_picture_number[0][0] = job->ref[r].picture_number;
        mj[r].src = job->centre.pyr, mj[r].ref[0][0] = job->ref[r].pyr;
        SvtHipMeFrameOut &o = mj[r].out;
        o.best_sad = (uint32_t *)(ws + l.best_sad + l.per_ref_best * r), o.best_mv = (uint32_t *)(ws + l.best_mv + l.per_ref_best * r);
        o.search_results = (SvtHipMeSearchResult *)(ws + l.sr + l.per_ref_sr * r);
        o.me_mv_array = (uint32_t *)(ws + l.me_scratch), o.me_candidate_array = ws + l.me_scratch, o.total_me_candidate_index = ws + l.me_scratch;
        o.me_64x64_distortion = o.me_32x32_distortion = o.me_16x16_distortion = o.me_8x8_distortion = o.me_8x8_cost_variance = o.rc_me_distortion =
            (uint32_t *)(ws + l.me_scratch);
        refs[r].pic = planes_of(job->ref[r]);
        refs[r].best_mv = o.best_mv, refs[r].best_sad = o.best_sad, refs[r].sr = o.search_results;
        refs[r].pred = ws + l.pred + l.per_ref_pred * r;
        refs[r].state = (SvtHipTfB64State *)(ws + l.states + l.per_ref_state * r);
        refs[r].desc = (SvtHipConvolveDesc *)(ws + l.desc + l.per_ref_desc * r);
        refs[r].blocks = (SvtHipTfBlock *)(ws + l.blocks + l.per_ref_blocks * r);
    }
    const bool low_delay = job->ctrls.low_delay != 0;
    int32_t    rc = low_delay ? (int32_t)SVT_HIP_OK : svt_hip_me_frames(mj, job->n_refs, st);  // the low-delay variant has no motion search
    delete[] mj;
    if (rc != SVT_HIP_OK)
        return rc;
    RefineRef *d_refs = (RefineRef *)stage_descriptors(refs, sizeof(RefineRef) * job->n_refs, st);
    if (!d_refs)
        return SVT_HIP_ERR_RUNTIME;
    RefineArgs a;
    memset(&a, 0, sizeof(a));
    a.ctrls = job->ctrls, a.centre = planes_of(job->centre);
    for (int p = 0; p < 3; p++) a.decay[p] = job->decay_factor_fp16[p];
    a.tf_me_exit_th = job->me.tf_me_exit_th, a.mi_rows = (int32_t)job->mi_rows, a.mi_cols = (int32_t)job->mi_cols;
    a.idx_min = -((int64_t)f.org_y * f.stride + f.org_x), a.idx_max = ((int64_t)f.height + f.org_y) * f.stride - f.org_x - 1;
    a.nb = nb, a.bw = bw, a.mv_dist_th = job->mv_dist_th, a.bit_depth = job->bit_depth, a.chroma = chroma;
    a.accum = (uint32_t *)(ws + l.accum), a.count = (uint16_t *)(ws + l.count);
    SvtHipTfBlock *static_blocks = (SvtHipTfBlock *)(ws + l.static_blocks);
    SvtHipTfOut   *outs = (SvtHipTfOut *)(ws + l.outs);
    const bool     s16 = is16 && !job->ctrls.use_8bit_subpel;
    if (low_delay) {
        hipLaunchKernelGGL(tf_low_delay_kernel, dim3(nb, job->n_refs), dim3(256), 0, st, a, d_refs);
    } else {
        const dim3 grid(nb, job->n_refs);
        if (job->ctrls.enable_8x8_pred) {
            if (s16)
                hipLaunchKernelGGL((tf_refine_kernel<true, true>), grid, dim3(256), 0, st, a, d_refs, job->tot_blks);
            else
                hipLaunchKernelGGL((tf_refine_kernel<false, true>), grid, dim3(256), 0, st, a, d_refs, job->tot_blks);
        } else if (s16) {
            hipLaunchKernelGGL((tf_refine_kernel<true, false>), grid, dim3(256), 0, st, a, d_refs, job->tot_blks);
        } else {
            hipLaunchKernelGGL((tf_refine_kernel<false, false>), grid, dim3(256), 0, st, a, d_refs, job->tot_blks);
        }
        SVT_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(tf_predict_kernel, dim3(nb, job->n_refs), dim3(256), 0, st, d_refs);
        hipLaunchKernelGGL(tf_predict_small_kernel, dim3(nb * SMALL_SPLIT, job->n_refs), dim3(64), 0, st, d_refs);
        if (job->ctrls.enable_8x8_pred)
            hipLaunchKernelGGL(tf_predict8_kernel, dim3(nb, job->n_refs), dim3(256), 0, st, a, d_refs);
    }
    SVT_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(tf_blocks_kernel, dim3(nb, job->n_refs + 1), dim3(256), 0, st, a, d_refs, job->n_refs, static_blocks, outs);
    SVT_HIP_CHECK(hipGetLastError());
    stage_commit(st);
    {  // central + the accumulation over every reference + normalise: one launch, accumulators in registers
        const SvtHipTfBlock *lists[SVT_HIP_TF_MAX_REFS];
        for (uint32_t r = 0; r < job->n_refs; r++) lists[r] = refs[r].blocks;
        rc = svt_hip_tf_filter_blocks(lists, job->n_refs, static_blocks, outs, nb * 4, 1, 1, st);
    }
    return rc;
}

SVT_HIP_MODULE_WARMUP(tf_picture)

/*
 * oracle/src/orc_convolve.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the single-reference inter-prediction interpolation:
 *   svt_av1_convolve_2d_sr_c / _x_sr_c / _y_sr_c / _2d_copy_sr_c            (inter_prediction.c:311-417)
 *   svt_av1_highbd_convolve_2d_sr_c / _x_sr_c / _y_sr_c / _2d_copy_sr_c     (inter_prediction.c:670-789)
 * Pinned against the real functions through oracle/_ref (tests/test_convolve_oracle.py).
 */
#include <stdint.h>
#include <stddef.h>

#include "orc_lf.h"

#define FILTER_BITS 7
#define RND(v, n) (((v) + ((1 << (n)) >> 1)) >> (n))

static inline int32_t px(const void *p, ptrdiff_t idx, int is16) {
    return is16 ? ((const uint16_t *)p)[idx] : ((const uint8_t *)p)[idx];
}
static inline void put(void *p, ptrdiff_t idx, int is16, int32_t v, int bd) {
    const int32_t hi = (1 << bd) - 1;
    v                = v < 0 ? 0 : (v > hi ? hi : v);
    if (is16)
        ((uint16_t *)p)[idx] = (uint16_t)v;
    else
        ((uint8_t *)p)[idx] = (uint8_t)v;
}

/* fx / fy: the kernel of the block's sub-pel phase (taps_x / taps_y coefficients); taps == 0 selects the variant the
 * reference dispatches to when that direction has no sub-pel offset (x_sr / y_sr / 2d_copy_sr). */
void orc_convolve_sr(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, int32_t w, int32_t h, const int16_t *fx,
                     int32_t taps_x, const int16_t *fy, int32_t taps_y, int32_t round_0, int32_t round_1, int32_t bd, int32_t is16) {
    if (!taps_x && !taps_y) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) put(dst, (ptrdiff_t)y * dst_stride + x, is16, px(src, (ptrdiff_t)y * src_stride + x, is16), 16);
        return;
    }
    if (!taps_y) { /* x_sr */
        const int fo = taps_x / 2 - 1, bits = FILTER_BITS - round_0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t res = 0;
                for (int k = 0; k < taps_x; k++) res += fx[k] * px(src, (ptrdiff_t)y * src_stride + x - fo + k, is16);
                res = RND(res, round_0);
                put(dst, (ptrdiff_t)y * dst_stride + x, is16, RND(res, bits), bd);
            }
        return;
    }
    if (!taps_x) { /* y_sr */
        const int fo = taps_y / 2 - 1;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t res = 0;
                for (int k = 0; k < taps_y; k++) res += fy[k] * px(src, (ptrdiff_t)(y - fo + k) * src_stride + x, is16);
                put(dst, (ptrdiff_t)y * dst_stride + x, is16, RND(res, FILTER_BITS), bd);
            }
        return;
    }
    /* 2d_sr */
    static int16_t im[(128 + 7) * 128];
    const int      fo_v = taps_y / 2 - 1, fo_h = taps_x / 2 - 1, im_h = h + taps_y - 1;
    const int      bits = 2 * FILTER_BITS - round_0 - round_1, offset_bits = bd + 2 * FILTER_BITS - round_0;
    for (int y = 0; y < im_h; y++)
        for (int x = 0; x < w; x++) {
            int32_t sum = 1 << (bd + FILTER_BITS - 1);
            for (int k = 0; k < taps_x; k++) sum += fx[k] * px(src, (ptrdiff_t)(y - fo_v) * src_stride + x - fo_h + k, is16);
            im[y * w + x] = (int16_t)(uint16_t)RND(sum, round_0);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t sum = 1 << offset_bits;
            for (int k = 0; k < taps_y; k++) sum += fy[k] * im[(y + k) * w + x];
            int32_t res = RND(sum, round_1) - ((1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1)));
            if (!is16)
                res = (int16_t)res; /* the 8-bit function narrows to int16 first (:343-345) */
            put(dst, (ptrdiff_t)y * dst_stride + x, is16, RND(res, bits), bd);
        }
}

/* Compound ("jnt") family: svt_av1_jnt_convolve_{2d,x,y,2d_copy}_c (inter_prediction.c:494-668) and the highbd set
 * (inter_prediction.c:852-1035).  mode 1: first prediction, the offset intermediate goes to the ConvBufType buffer `cbuf`;
 * mode 2: second prediction averaged with cbuf (do_average); mode 3: distance-weighted average (use_jnt_comp_avg, weights
 * fwd / bck, DIST_PRECISION_BITS = 4); the averaged pixel goes to dst. */
void orc_convolve_jnt(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, int32_t w, int32_t h, const int16_t *fx,
                      int32_t taps_x, const int16_t *fy, int32_t taps_y, int32_t round_0, int32_t round_1, int32_t bd, int32_t is16,
                      uint16_t *cbuf, int32_t cbuf_stride, int32_t mode, int32_t fwd, int32_t bck) {
    static int16_t im[(128 + 7) * 128];
    const int      offset_bits = bd + 2 * FILTER_BITS - round_0, round_bits = 2 * FILTER_BITS - round_0 - round_1;
    const int32_t  round_offset = (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
    const int      fo_v = taps_y ? taps_y / 2 - 1 : 0, fo_h = taps_x ? taps_x / 2 - 1 : 0;
    if (taps_x && taps_y) {
        const int im_h = h + taps_y - 1;
        for (int y = 0; y < im_h; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 1 << (bd + FILTER_BITS - 1);
                for (int k = 0; k < taps_x; k++) sum += fx[k] * px(src, (ptrdiff_t)(y - fo_v) * src_stride + x - fo_h + k, is16);
                im[y * w + x] = (int16_t)(uint16_t)RND(sum, round_0);
            }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t res;
            if (taps_x && taps_y) {
                int32_t sum = 1 << offset_bits;
                for (int k = 0; k < taps_y; k++) sum += fy[k] * im[(y + k) * w + x];
                res = (uint16_t)RND(sum, round_1); /* ConvBufType res */
            } else if (taps_y) {
                const int bits = FILTER_BITS - round_0;
                res            = 0;
                for (int k = 0; k < taps_y; k++) res += fy[k] * px(src, (ptrdiff_t)(y - fo_v + k) * src_stride + x, is16);
                res *= (1 << bits);
                res = RND(res, round_1) + round_offset;
            } else if (taps_x) {
                const int bits = FILTER_BITS - round_1;
                res            = 0;
                for (int k = 0; k < taps_x; k++) res += fx[k] * px(src, (ptrdiff_t)y * src_stride + x - fo_h + k, is16);
                res = (1 << bits) * RND(res, round_0);
                res += round_offset;
            } else {
                uint16_t r16 = (uint16_t)(px(src, (ptrdiff_t)y * src_stride + x, is16) << round_bits);
                r16          = (uint16_t)(r16 + (uint16_t)round_offset);
                res          = r16;
            }
            if (mode == 1) {
                cbuf[(ptrdiff_t)y * cbuf_stride + x] = (uint16_t)res;
            } else {
                int32_t tmp = cbuf[(ptrdiff_t)y * cbuf_stride + x];
                tmp         = mode == 3 ? (tmp * fwd + res * bck) >> 4 : (tmp + res) >> 1;
                tmp -= round_offset;
                put(dst, (ptrdiff_t)y * dst_stride + x, is16, RND(tmp, round_bits), bd);
            }
        }
}

/*
 * oracle/src/orc_wiener.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the Wiener-restoration kernels:
 *   svt_av1_compute_stats_c / _highbd_c      (restoration_pick.c:671-757, find_average restoration_pick.h:24-44)
 *   svt_av1_wiener_convolve_add_src_c / svt_av1_highbd_wiener_convolve_add_src_c (convolve.c:57-200)
 * Pinned against the real functions through oracle/_ref (tests/test_wiener_oracle.py).
 */
#include <stdlib.h>
#include <string.h>

#include "orc_lf.h"

static inline int32_t px(const void *p, ptrdiff_t idx, int is16) {
    return is16 ? ((const uint16_t *)p)[idx] : ((const uint8_t *)p)[idx];
}

/* M[win2], H[win2][win2]; y index = (column offset) * win + (row offset), as in the reference's loops */
void orc_wiener_compute_stats(int32_t wiener_win, const void *dgd, const void *src, int32_t h_start, int32_t h_end, int32_t v_start,
                              int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H, int32_t is16,
                              int32_t bit_depth) {
    const int32_t win2 = wiener_win * wiener_win, half = wiener_win >> 1;
    uint64_t      sum  = 0;
    for (int i = v_start; i < v_end; i++)
        for (int j = h_start; j < h_end; j++) sum += (uint64_t)px(dgd, (ptrdiff_t)i * dgd_stride + j, is16);
    const int32_t avg = (int32_t)(sum / (uint64_t)((v_end - v_start) * (h_end - h_start)));
    int32_t       y[49];
    memset(M, 0, sizeof(*M) * win2);
    memset(H, 0, sizeof(*H) * win2 * win2);
    for (int i = v_start; i < v_end; i++)
        for (int j = h_start; j < h_end; j++) {
            const int32_t x = px(src, (ptrdiff_t)i * src_stride + j, is16) - avg;
            int           idx = 0;
            for (int k = -half; k <= half; k++)
                for (int l = -half; l <= half; l++) y[idx++] = px(dgd, (ptrdiff_t)(i + l) * dgd_stride + (j + k), is16) - avg;
            for (int k = 0; k < win2; k++) {
                M[k] += (int64_t)y[k] * x;
                for (int l = k; l < win2; l++) H[k * win2 + l] += (int64_t)y[k] * y[l];
            }
        }
    /* highbd only: scale back to the 8-bit range (truncating division, :749-756); 8-bit path has no divider */
    const int64_t div = is16 ? (bit_depth == 12 ? 16 : (bit_depth == 10 ? 4 : 1)) : 1;
    for (int k = 0; k < win2; k++) {
        M[k] /= div;
        H[k * win2 + k] /= div;
        for (int l = k + 1; l < win2; l++) {
            H[k * win2 + l] /= div;
            H[l * win2 + k] = H[k * win2 + l];
        }
    }
}

/* Separable 7-tap (8 coefficients, the last one zero) filter around the source sample (convolve.c:57-147, 150-200).
 * w, h <= 128.  round_0 / round_1 as get_conv_params_wiener(bd) (convolve.h:70-86). */
void orc_wiener_convolve_add_src(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, const int16_t *fx, const int16_t *fy,
                                 int32_t w, int32_t h, int32_t round_0, int32_t round_1, int32_t bd, int32_t is16) {
    const int32_t ih    = h + 7; /* rows -3 .. h+3 (the 8th tap is zero) */
    uint16_t     *temp  = calloc((size_t)(ih + 1) * w, sizeof(uint16_t));
    const int32_t limit = (1 << (bd + 1 + 7 - round_0)) - 1;
    for (int y = 0; y < ih; y++)
        for (int x = 0; x < w; x++) {
            int32_t sum = (px(src, (ptrdiff_t)(y - 3) * src_stride + x, is16) << 7) + (1 << (bd + 7 - 1));
            for (int k = 0; k < 8; k++) sum += px(src, (ptrdiff_t)(y - 3) * src_stride + x - 3 + k, is16) * fx[k];
            int32_t v = (sum + ((1 << round_0) >> 1)) >> round_0;
            temp[y * w + x] = (uint16_t)(v < 0 ? 0 : (v > limit ? limit : v));
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t sum = ((int32_t)temp[(y + 3) * w + x] << 7) - (1 << (bd + round_1 - 1));
            for (int k = 0; k < 8; k++) sum += (int32_t)temp[(y + k) * w + x] * fy[k];
            int32_t v = (sum + ((1 << round_1) >> 1)) >> round_1;
            const int32_t hi = (1 << bd) - 1;
            v = v < 0 ? 0 : (v > hi ? hi : v);
            if (is16)
                ((uint16_t *)dst)[(ptrdiff_t)y * dst_stride + x] = (uint16_t)v;
            else
                ((uint8_t *)dst)[(ptrdiff_t)y * dst_stride + x] = (uint8_t)v;
        }
    free(temp);
}

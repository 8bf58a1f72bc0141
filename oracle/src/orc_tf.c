/* oracle/src/orc_tf.c — TEST INFRASTRUCTURE: CPU restatement of the temporal filter's accumulate / normalise stage
 * (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *   orc_tf_accumulate   follows svt_av1_apply_temporal_filter_planewise_medium_c / _hbd_c and their *_partial_c helpers
 *                       (Source/Lib/Codec/temporal_filtering.c:999-1330), with sqrt_fast (:686-714), the exp(-x/16) table
 *                       (:674-684) and calculate_squared_errors_sum[_highbd] (:746-772)
 *   orc_tf_central      follows svt_aom_apply_filtering_central_c / _highbd_c (:349-420)
 *   orc_tf_normalise    follows svt_aom_get_final_filtered_pixels_c (:2578-2650)
 * Pinned against the reference through oracle/ref_harness.c::ref_tf_block_accumulate (tests/test_tf_oracle.py) and
 * tests/golden/tf.npz. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/svt_hip_tf.h"

#define ORC_API __attribute__((visibility("default")))
#define TF_WEIGHT_SCALE 1000
#define TF_BALANCE 5

static uint32_t px(const void *p, size_t i, int is16) { return is16 ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i]; }

static int ilog2(uint32_t x) { /* svt_aom_log2f_32: index of the highest set bit */
    int l = 0;
    while (x >>= 1) l++;
    return l;
}
static uint32_t sqrt_fast(uint32_t x) {
    /* (uint32_t)(sqrt(i) * 65536) for i = 0..15, the reference's sqrt_array_fp16 */
    static const uint32_t tab[16] = {0,      65536,  92681,  113511, 131072, 146542, 160529, 173391,
                                     185363, 196608, 207243, 217358, 227023, 236293, 245213, 253819};
    if (x > 15) {
        const int log2_half = ilog2(x) >> 1, mul2 = log2_half << 1;
        return tab[x >> (mul2 - 2)] >> (17 - log2_half);
    }
    return tab[x] >> 16;
}
/* the reference's expf_tab_fp16: 65536 * exp(-i / 16) truncated, i = 0 .. 112 */
static uint32_t exp_fp16(uint32_t i) {
    static uint32_t tab[113];
    if (!tab[0])
        for (int k = 0; k < 113; k++) tab[k] = (uint32_t)(65536.0 * exp(-k / 16.0));
    return tab[i];
}

static void plane(const SvtHipTfBlock *b, int pl, uint32_t bw, uint32_t bh, uint32_t luma_err[4]) {
    const int      is16 = b->is_16bit, shift = is16 ? (b->bit_depth - 8) * 2 : 0;
    const uint32_t dist_th = (uint32_t)((b->mv_dist_th << 16) / 10) > (1u << 16) ? (uint32_t)((b->mv_dist_th << 16) / 10) : (1u << 16);
    uint32_t       decay = b->decay_factor_fp16[pl], d_factor[4], blk_err[4], win[4];
    for (int i = 0; i < 4; i++) {
        const int     k = b->split ? i : 0;
        const int32_t col = b->mv_x[k], row = b->mv_y[k];
        const uint32_t dist = sqrt_fast(((uint32_t)(col * col + row * row)) << 8);
        const uint32_t df = (dist << 12) / (dist_th >> 8);
        d_factor[i]       = df > (1u << 8) ? df : (1u << 8);
        if (b->split)
            blk_err[i] = (uint32_t)(is16 ? b->block_error[i] >> 4 : b->block_error[i]);
        else
            blk_err[i] = (uint32_t)(b->block_error[0] >> (is16 ? 6 : 2));
    }
    if (!b->split)
        decay <<= 1;
    const uint32_t hw = bw >> 1, hh = bh >> 1;
    for (int q = 0; q < 4; q++) {
        const uint32_t x0 = (q & 1) * hw, y0 = (q >> 1) * hh;
        uint32_t       sum = 0;
        for (uint32_t i = 0; i < hh; i++)
            for (uint32_t j = 0; j < hw; j++) {
                const int32_t d = (int32_t)px(b->src[pl], (size_t)(y0 + i) * b->src_stride[pl] + x0 + j, is16) -
                    (int32_t)px(b->pred[pl], (size_t)(y0 + i) * b->pred_stride[pl] + x0 + j, is16);
                sum += (uint32_t)(is16 ? d * d : (int32_t)((int16_t)d * (int16_t)d));
            }
        sum >>= shift;
        win[q] = (((sum << 4) / hw) << 4) / hh;
        if (pl)
            win[q] = (win[q] * 5 + luma_err[q]) / 6;
        else
            luma_err[q] = win[q];
    }
    for (int q = 0; q < 4; q++) {
        const uint32_t combined = (win[q] * TF_BALANCE + blk_err[q]) / (TF_BALANCE + 1);
        const uint64_t avg_err  = (uint64_t)((combined >> 3) * (d_factor[q] >> 3)); /* 32-bit product, as the reference */
        const uint32_t den = (decay >> 10) > 1 ? (decay >> 10) : 1;
        uint32_t       sd  = (uint32_t)(avg_err / den);
        sd                 = sd < 7 * 16 ? sd : 7 * 16;
        const uint32_t w   = (exp_fp16(sd) * TF_WEIGHT_SCALE) >> 16;
        const uint32_t x0 = (q & 1) * bw / 2, y0 = (q >> 1) * bh / 2;
        for (uint32_t i = 0; i < bh / 2; i++)
            for (uint32_t j = 0; j < bw / 2; j++) {
                const size_t k = (size_t)(i + y0) * b->pred_stride[pl] + j + x0;
                b->count[pl][k] = (uint16_t)(b->count[pl][k] + w);
                b->accum[pl][k] += w * px(b->pred[pl], k, is16);
            }
    }
}

/* svt_av1_apply_zz_based_temporal_filter_planewise_medium_partial_c / _hbd (temporal_filtering.c:789-835, 890-940) */
static void plane_zz(const SvtHipTfBlock *b, int pl, uint32_t bw, uint32_t bh) {
    const int      is16 = b->is_16bit;
    const uint32_t decay = b->decay_factor_fp16[pl];
    for (int q = 0; q < 4; q++) {
        const uint32_t blk_err = b->split ? (uint32_t)(is16 ? b->block_error[q] >> 4 : b->block_error[q])
                                          : (uint32_t)(b->block_error[0] >> (is16 ? 6 : 2));
        const uint32_t avg_err = blk_err << 2, den = (decay >> 10) > 1 ? (decay >> 10) : 1;
        uint32_t       sd = avg_err / den;
        sd                = sd < 7 * 16 ? sd : 7 * 16;
        const uint32_t w  = (exp_fp16(sd) * TF_WEIGHT_SCALE) >> 17;
        const uint32_t x0 = (q & 1) * bw / 2, y0 = (q >> 1) * bh / 2;
        for (uint32_t i = 0; i < bh / 2; i++)
            for (uint32_t j = 0; j < bw / 2; j++) {
                const size_t k = (size_t)(i + y0) * b->pred_stride[pl] + j + x0;
                b->count[pl][k] = (uint16_t)(b->count[pl][k] + w);
                b->accum[pl][k] += w * px(b->pred[pl], k, is16);
            }
    }
}

ORC_API void orc_tf_accumulate(const SvtHipTfBlock *b) {
    uint32_t luma_err[4];
    if (b->zz_based) {
        plane_zz(b, 0, 32, 32);
        if (b->chroma)
            plane_zz(b, 1, 32u >> b->ss_x, 32u >> b->ss_y), plane_zz(b, 2, 32u >> b->ss_x, 32u >> b->ss_y);
        return;
    }
    plane(b, 0, 32, 32, luma_err);
    if (b->chroma) {
        plane(b, 1, 32u >> b->ss_x, 32u >> b->ss_y, luma_err);
        plane(b, 2, 32u >> b->ss_x, 32u >> b->ss_y, luma_err);
    }
}

ORC_API void orc_tf_central(const SvtHipTfBlock *b) {
    for (int pl = 0; pl < (b->chroma ? 3 : 1); pl++) {
        const uint32_t bw = pl ? 32u >> b->ss_x : 32, bh = pl ? 32u >> b->ss_y : 32;
        for (uint32_t i = 0; i < bh; i++)
            for (uint32_t j = 0; j < bw; j++) {
                const size_t k = (size_t)i * b->pred_stride[pl] + j;
                b->accum[pl][k] = TF_WEIGHT_SCALE * px(b->src[pl], (size_t)i * b->src_stride[pl] + j, b->is_16bit);
                b->count[pl][k] = TF_WEIGHT_SCALE;
            }
    }
}

ORC_API void orc_tf_normalise(const SvtHipTfBlock *b, const SvtHipTfOut *o) {
    for (int pl = 0; pl < (b->chroma ? 3 : 1); pl++) {
        const uint32_t bw = pl ? 32u >> b->ss_x : 32, bh = pl ? 32u >> b->ss_y : 32;
        for (uint32_t i = 0; i < bh; i++)
            for (uint32_t j = 0; j < bw; j++) {
                const size_t   k = (size_t)i * b->pred_stride[pl] + j;
                const uint32_t v = (b->accum[pl][k] + (b->count[pl][k] >> 1)) / b->count[pl][k];
                if (b->is_16bit)
                    ((uint16_t *)o->dst[pl])[(size_t)i * o->dst_stride[pl] + j] = (uint16_t)v;
                else
                    ((uint8_t *)o->dst[pl])[(size_t)i * o->dst_stride[pl] + j] = (uint8_t)v;
            }
    }
}

/* svt_estimate_noise_fp16_c / svt_estimate_noise_highbd_fp16_c (temporal_filtering.c:3668-3736); stride in samples,
 * bd is 8 for 8-bit planes (no rounding shift). */
ORC_API int32_t orc_estimate_noise(const void *src, int width, int height, int stride, int is16, int bd) {
    int64_t   sum = 0, num = 0;
    const int sh = is16 ? bd - 8 : 0, rnd = sh ? 1 << (sh - 1) : 0;
#define PX(r, c) (is16 ? (int)((const uint16_t *)src)[(size_t)(r) * stride + (c)] : (int)((const uint8_t *)src)[(size_t)(r) * stride + (c)])
    for (int i = 1; i < height - 1; i++)
        for (int j = 1; j < width - 1; j++) {
            const int gx = (PX(i - 1, j - 1) - PX(i - 1, j + 1)) + (PX(i + 1, j - 1) - PX(i + 1, j + 1)) + 2 * (PX(i, j - 1) - PX(i, j + 1));
            const int gy = (PX(i - 1, j - 1) - PX(i + 1, j - 1)) + (PX(i - 1, j + 1) - PX(i + 1, j + 1)) + 2 * (PX(i - 1, j) - PX(i + 1, j));
            const int ga = (abs(gx) + abs(gy) + rnd) >> sh;
            if (ga < 50) {
                const int v = 4 * PX(i, j) - 2 * (PX(i, j - 1) + PX(i, j + 1) + PX(i - 1, j) + PX(i + 1, j)) +
                    (PX(i - 1, j - 1) + PX(i - 1, j + 1) + PX(i + 1, j - 1) + PX(i + 1, j + 1));
                sum += (abs(v) + rnd) >> sh;
                num++;
            }
        }
#undef PX
    if (num < 16)
        return -65536;
    return (int32_t)((sum * 82137) / (6 * num));
}

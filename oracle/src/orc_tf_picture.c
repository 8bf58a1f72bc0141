/*
 * oracle/src/orc_tf_picture.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the temporal filter's block loop for one centre picture (the motion half):
 *   produce_temporally_filtered_pic                         temporal_filtering.c:2752-3308
 *   svt_check_position / tf_subpel_search                   :1531-1761
 *   tf_64x64 / tf_32x32 / tf_16x16_sub_pel_search           :1763-2104
 *   derive_tf_32x32_block_split_flag                        :236-285
 *   tf_use_64x64_pred, convert_64x64_info_to_32x32_info     :2646-2728
 *   tf_64x64 / tf_32x32_inter_prediction                    :2226-2576   (svt_aom_inter_prediction, enc_inter_prediction.c:4070)
 *   compute_subpel_params / clamp_mv_to_umv_border_sb       enc_inter_prediction.c:28-48, 3126-3178
 *   svt_aom_variance{64,32,16}x*_c, svt_aom_highbd_10_variance*_c   C_DEFAULT/variance.c:257-306, svt_psnr.c:139-176
 * built on the already pinned pieces: orc_me_frame_range (ME_MCTF), orc_convolve_sr, orc_tf_central / _accumulate / _normalise.
 * Pinned against the REAL produce_temporally_filtered_pic through oracle/ref_harness_tfme.c (tests/test_tf_picture_oracle.py)
 * and against tests/golden/tf_picture.npz.  Every pointer inside the job is a HOST pointer here; `workspace` is not used.
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/svt_hip_tf.h"
#include "orc.h"

void orc_convolve_sr(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, int32_t w, int32_t h, const int16_t *fx,
                     int32_t taps_x, const int16_t *fy, int32_t taps_y, int32_t round_0, int32_t round_1, int32_t bd, int32_t is16);
ORC_API void orc_tf_accumulate(const SvtHipTfBlock *b);
ORC_API void orc_tf_central(const SvtHipTfBlock *b);
ORC_API void orc_tf_normalise(const SvtHipTfBlock *b, const SvtHipTfOut *o);

/* the AV1 interpolation kernels (inter_prediction.c:223-300): regular, sharp, bilinear — 16 phases x 8 taps */
ORC_API const int16_t orc_interp_kernels[4][16][8] = {
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
    /* sub_pel_filters_4 (inter_prediction.c:239-254): what av1_get_interp_filter_params_with_block_size gives blocks of width <= 4 for the
     * regular AND the sharp filter (inter_prediction.h:137-145) */
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, -4, 126, 8, -2, 0, 0}, {0, 0, -8, 122, 18, -4, 0, 0}, {0, 0, -10, 116, 28, -6, 0, 0},
     {0, 0, -12, 110, 38, -8, 0, 0}, {0, 0, -12, 102, 48, -10, 0, 0}, {0, 0, -14, 94, 58, -10, 0, 0}, {0, 0, -12, 84, 66, -10, 0, 0},
     {0, 0, -12, 76, 76, -12, 0, 0}, {0, 0, -10, 66, 84, -12, 0, 0}, {0, 0, -10, 58, 94, -14, 0, 0}, {0, 0, -10, 48, 102, -12, 0, 0},
     {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -6, 28, 116, -10, 0, 0}, {0, 0, -4, 18, 122, -8, 0, 0}, {0, 0, -2, 8, 126, -4, 0, 0}}};
enum { K_REGULAR = 0, K_SHARP = 1, K_BILINEAR = 2, K_REGULAR4 = 3 };

typedef struct Geo { /* what the reference keeps in MacroBlockD / Av1Common for the motion-vector clamp */
    int32_t mi_rows, mi_cols;
} Geo;

static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* tf_inter_predictor / svt_aom_enc_make_inter_predictor, single reference, identity scale: the block of bw x bh samples of
 * the plane with sub-sampling `ss` whose origin in that plane is (pre_x, pre_y); (lx, ly, lsize) = the LUMA block it belongs to
 * (mb_to_*_edge); ref0 = sample (0,0) of the reference plane */
static void predict(const Geo *g, const void *ref0, int32_t ref_stride, int is16, int bd, void *dst, int32_t dst_stride, int pre_x,
                    int pre_y, int bw, int bh, int mvx, int mvy, int ss, int lx, int ly, int lsize, int kernel, int row_shift) {
    const int mirow = ly >> 2, micol = lx >> 2, bmi = lsize >> 2;
    const int32_t to_top = -((mirow * 4) * 8), to_bottom = ((g->mi_rows - bmi - mirow) * 4) * 8;
    const int32_t to_left = -((micol * 4) * 8), to_right = ((g->mi_cols - bmi - micol) * 4) * 8;
    const int32_t spel_left = (4 + bw) << 4, spel_right = spel_left - 16, spel_top = (4 + bh) << 4, spel_bottom = spel_top - 16;
    const int     m   = 1 << (1 - ss);
    int16_t       col = (int16_t)(mvx * m), row = (int16_t)(mvy * m);
    col = (int16_t)clampi(col, to_left * m - spel_left, to_right * m + spel_right);
    row = (int16_t)clampi(row, to_top * m - spel_top, to_bottom * m + spel_bottom);
    const int sx = col & 15, sy = row & 15;
    const int pos_x = pre_x + (col >> 4), pos_y = pre_y + (row >> 4);
    int       r0 = 3, r1 = 11;
    if (bd + 7 - r0 + 2 > 16)
        r1 -= bd + 7 - r0 + 2 - 16, r0 += bd + 7 - 3 + 2 - 16;
    const uint8_t *src = (const uint8_t *)ref0 + (((ptrdiff_t)pos_y * ref_stride + pos_x) << is16);
    orc_convolve_sr(src, ref_stride << row_shift, dst, dst_stride << row_shift, bw, bh >> row_shift, orc_interp_kernels[kernel][sx], sx ? 8 : 0,
                    orc_interp_kernels[kernel][sy], sy ? 8 : 0, r0, r1, bd, is16);
}

/* fn_ptr->vf / vf_hbd_10 of a w x h block */
static uint64_t variance(const void *a, int32_t a_stride, const void *b, int32_t b_stride, int w, int h, int is16) {
    if (!is16) {
        const uint8_t *pa = a, *pb = b;
        int            sum = 0;
        uint32_t       sse = 0;
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                const int d = pa[(ptrdiff_t)i * a_stride + j] - pb[(ptrdiff_t)i * b_stride + j];
                sum += d, sse += (uint32_t)(d * d);
            }
        return sse - (uint32_t)(((int64_t)sum * sum) / (w * h));
    }
    const uint16_t *pa = a, *pb = b;
    int64_t         tsum = 0;
    uint64_t        tsse = 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int d = pa[(ptrdiff_t)i * a_stride + j] - pb[(ptrdiff_t)i * b_stride + j];
            tsum += d, tsse += (uint32_t)(d * d);
        }
    const uint32_t sse = (uint32_t)((tsse + 8) >> 4);
    const int      sum = (int)((tsum + 2) >> 2);
    const int64_t  var = (int64_t)sse - (((int64_t)sum * sum) / (w * h));
    return var >= 0 ? (uint32_t)var : 0;
}

typedef struct Search { /* one (reference picture, b64) */
    const SvtHipTfPictureJob *job;
    Geo         g;
    const void *src0, *ref0; /* sample (0,0) of the planes the SEARCH runs on */
    int32_t     src_stride, ref_stride;
    int         is16, bd;    /* of the search */
    int         ox, oy;      /* block origin */
    void       *tmp;         /* 64 x 64 prediction scratch of the search depth */
} Search;

/* tf_subpel_search: (lx, ly) local origin of the bsize block inside the b64 */
static void subpel_search(const Search *s, int bsize, int lx, int ly, int kernel, uint64_t *best_dist, int16_t *best_x, int16_t *best_y) {
    const SvtHipTfCtrls *c = &s->job->ctrls;
    const int modes[3] = {c->half_pel_mode, c->quarter_pel_mode, c->eight_pel_mode}, steps[3] = {4, 2, 1};
    for (int round = -1; round < 3; round++) {
        if (round >= 0 && !modes[round])
            continue;
        const int     mode = round < 0 ? c->half_pel_mode : modes[round], step = round < 0 ? 0 : steps[round];
        const int16_t cx = *best_x, cy = *best_y;
        for (int i = -1; i <= 1; i++)
            for (int j = -1; j <= 1; j++) {
                const int xd = i * step, yd = j * step;
                if (round < 0 ? (i || j) : (!i && !j))
                    continue;
                /* svt_check_position */
                if (mode >= 2 && xd != 0 && yd != 0)
                    continue;
                if (*best_dist == 0)
                    continue;
                if (c->subpel_early_exit_th && *best_dist < (((uint64_t)(bsize * bsize) * c->subpel_early_exit_th) << s->is16))
                    continue;
                const int16_t mvx = (int16_t)(cx + xd), mvy = (int16_t)(cy + yd);
                const int     shift = (xd == 0 && yd == 0) ? c->sub_sampling_shift : 0;
                uint8_t      *pred = (uint8_t *)s->tmp + (((size_t)ly * 64 + lx) << s->is16);
                predict(&s->g, s->ref0, s->ref_stride, s->is16, s->bd, pred, 64, s->ox + lx, s->oy + ly, bsize, bsize, mvx, mvy, 0, s->ox + lx,
                        s->oy + ly, bsize, kernel, shift);
                const uint8_t *src = (const uint8_t *)s->src0 + (((ptrdiff_t)(s->oy + ly) * s->src_stride + s->ox + lx) << s->is16);
                /* the reference always passes the full-height size's sub-sampled variant (64x32 ...) with doubled strides */
                const uint64_t d = variance(pred, 64 << c->sub_sampling_shift, src, s->src_stride << c->sub_sampling_shift, bsize,
                                            bsize >> c->sub_sampling_shift, s->is16)
                    << c->sub_sampling_shift;
                if (d < *best_dist)
                    *best_dist = d, *best_x = mvx, *best_y = mvy;
            }
    }
}

static inline int16_t mv_x_of(uint32_t mv) { return (int16_t)(mv & 0xffff); }
static inline int16_t mv_y_of(uint32_t mv) { return (int16_t)(mv >> 16); }

/* the final prediction of one square luma block + its chroma into the 64 x 64 (32 x 32 chroma) prediction buffers */
static void final_prediction(const SvtHipTfPictureJob *job, const Geo *g, const SvtHipTfPic *ref, void *const pred[3], int ox, int oy, int lx,
                             int ly, int bsize, int mvx, int mvy) {
    const int is16 = job->bit_depth > 8;
    const SvtHipPlane8 *f = &ref->pyr.full;
    const void *y0 = is16 ? (const void *)(ref->hbd[0] + (size_t)f->org_y * f->stride + f->org_x) : (const void *)(f->buf + (size_t)f->org_y * f->stride + f->org_x);
    predict(g, y0, (int32_t)f->stride, is16, job->bit_depth, (uint8_t *)pred[0] + (((size_t)ly * 64 + lx) << is16), 64, ox + lx, oy + ly, bsize, bsize,
            mvx, mvy, 0, ox + lx, oy + ly, bsize, K_SHARP, 0);
    if (!job->chroma)
        return;
    for (int p = 1; p < 3; p++) {
        const size_t corg = (size_t)(f->org_y / 2) * ref->chroma8_stride + f->org_x / 2;
        const void  *c0 = is16 ? (const void *)(ref->hbd[p] + corg) : (const void *)(ref->chroma8[p - 1] + corg);
        const int    px = (((ox + lx) >> 3) << 3) / 2, py = (((oy + ly) >> 3) << 3) / 2; /* pu_origin_*_chroma */
        const int    dx = ((lx >> 3) << 3) / 2, dy = ((ly >> 3) << 3) / 2;
        predict(g, c0, (int32_t)ref->chroma8_stride, is16, job->bit_depth, (uint8_t *)pred[p] + (((size_t)dy * 32 + dx) << is16), 32, px, py, bsize / 2,
                bsize / 2, mvx, mvy, 1, ox + lx, oy + ly, bsize, bsize / 2 <= 4 ? K_REGULAR4 : K_SHARP, 0);
    }
}

/* the motion refinement of one b64 against one reference picture; best_mv / best_sad / sr: the ME_MCTF results of the block */
static void refine_b64(const SvtHipTfPictureJob *job, const SvtHipTfPic *ref, int ox, int oy, const uint32_t *best_mv, const uint32_t *best_sad,
                       const SvtHipMeSearchResult *sr, void *const pred[3], void *tmp, SvtHipTfB64State *st) {
    const SvtHipTfCtrls *c = &job->ctrls;
    const int full16 = job->bit_depth > 8, s16 = full16 && !c->use_8bit_subpel;
    Search    s;
    s.job = job, s.g.mi_rows = (int32_t)job->mi_rows, s.g.mi_cols = (int32_t)job->mi_cols;
    const SvtHipPlane8 *cf = &job->centre.pyr.full, *rf = &ref->pyr.full;
    s.src0 = s16 ? (const void *)(job->centre.hbd[0] + (size_t)cf->org_y * cf->stride + cf->org_x) : (const void *)(cf->buf + (size_t)cf->org_y * cf->stride + cf->org_x);
    s.ref0 = s16 ? (const void *)(ref->hbd[0] + (size_t)rf->org_y * rf->stride + rf->org_x) : (const void *)(rf->buf + (size_t)rf->org_y * rf->stride + rf->org_x);
    s.src_stride = (int32_t)cf->stride, s.ref_stride = (int32_t)rf->stride;
    s.is16 = s16, s.bd = s16 ? job->bit_depth : 8, s.ox = ox, s.oy = oy, s.tmp = tmp;
    memset(st, 0, sizeof(*st));
    int use64 = 0;
    if (c->low_delay) { /* produce_temporally_filtered_pic_ld (temporal_filtering.c:3533-3620): tf_64x64_mv = 0, tf_64x64_inter_prediction,
                         * the four 32x32 variances, no split */
        st->use_64x64 = 1, use64 = 1;
        goto low_delay;
    }
    /* svt_aom_motion_estimation_b64 leaves after HME when the HME distortion is below tf_me_exit_th (motion_estimation.c:3179) */
    const int use64_th = (sr->hme_sad < job->me.tf_me_exit_th) ? 255 : c->use_pred_64x64_only_th;
    const int k6432 = c->use_2tap ? K_BILINEAR : K_REGULAR;
    /* tf_64x64_sub_pel_search */
    st->err64  = 0x7fffffff;
    st->mv64_x = (int16_t)((use64_th == 255 ? sr->hme_sc_x : mv_x_of(best_mv[0])) << 3);
    st->mv64_y = (int16_t)((use64_th == 255 ? sr->hme_sc_y : mv_y_of(best_mv[0])) << 3);
    subpel_search(&s, 64, 0, 0, k6432, &st->err64, &st->mv64_x, &st->mv64_y);
    if (use64_th) {
        if (use64_th == 255)
            use64 = 1;
        else { /* tf_use_64x64_pred */
            uint32_t d32 = 0;
            for (int i = 0; i < 4; i++) d32 += best_sad[1 + i];
            const int64_t a = best_sad[0] > 1 ? best_sad[0] : 1, b = d32 > 1 ? d32 : 1;
            use64 = ((a - b) * 100) / b < use64_th;
        }
    }
    if (!use64) {
        uint64_t sum32 = 0;
        for (int i = 0; i < 4; i++) {
            st->err32[i]  = 0x7fffffff;
            st->mv32_x[i] = (int16_t)(mv_x_of(best_mv[1 + i]) << 3), st->mv32_y[i] = (int16_t)(mv_y_of(best_mv[1 + i]) << 3);
            subpel_search(&s, 32, (i & 1) * 32, (i >> 1) * 32, k6432, &st->err32[i], &st->mv32_x[i], &st->mv32_y[i]);
            sum32 += st->err32[i];
        }
        if (st->err64 * 14 < sum32 * 16 && st->err64 < (1 << 18))
            use64 = 1;
    }
    st->use_64x64 = (uint8_t)use64;
low_delay:
    if (use64) {
        final_prediction(job, &s.g, ref, pred, ox, oy, 0, 0, 64, st->mv64_x, st->mv64_y);
        /* convert_64x64_info_to_32x32_info: measured on the pictures the FILTER works on */
        const void *src_full = full16 ? (const void *)(job->centre.hbd[0] + (size_t)cf->org_y * cf->stride + cf->org_x) : (const void *)(cf->buf + (size_t)cf->org_y * cf->stride + cf->org_x);
        for (int i = 0; i < 4; i++) {
            st->mv32_x[i] = st->mv64_x, st->mv32_y[i] = st->mv64_y, st->split32[i] = 0;
            const int      lx = (i & 1) * 32, ly = (i >> 1) * 32, sh = c->sub_sampling_shift;
            const uint8_t *pp = (const uint8_t *)pred[0] + (((size_t)ly * 64 + lx) << full16);
            const uint8_t *sp = (const uint8_t *)src_full + (((ptrdiff_t)(oy + ly) * cf->stride + ox + lx) << full16);
            st->err32[i]      = variance(pp, 64 << sh, sp, (int32_t)cf->stride << sh, 32, 32 >> sh, full16) << sh;
        }
        return;
    }
    for (int i = 0; i < 4; i++) {
        const int lx = (i & 1) * 32, ly = (i >> 1) * 32;
        if (st->err32[i] < c->pred_error_32x32_th) {
            st->split32[i] = 0;
        } else {
            /* tf_16x16_sub_pel_search (always the regular 8-tap kernel), tf_8x8_sub_pel_search (:2106-2224) with enable_8x8_pred, then
             * derive_tf_32x32_block_split_flag (:236-285) */
            for (int k = 0; k < 4; k++) {
                const int q = i * 4 + k;
                st->err16[q]  = 0x7fffffff;
                st->mv16_x[q] = (int16_t)(mv_x_of(best_mv[5 + q]) << 3), st->mv16_y[q] = (int16_t)(mv_y_of(best_mv[5 + q]) << 3);
                subpel_search(&s, 16, lx + (k & 1) * 16, ly + (k >> 1) * 16, K_REGULAR, &st->err16[q], &st->mv16_x[q], &st->mv16_y[q]);
            }
            if (c->enable_8x8_pred)
                for (int k = 0; k < 4; k++)
                    for (int e = 0; e < 4; e++) {
                        /* idx_32x32_to_idx_8x8 / subblock_xy_8x8 / tab8x8: the 8x8 blocks in z-order, their vectors in the same order */
                        const int idx = i * 16 + k * 4 + e, lx8 = lx + (k & 1) * 16 + (e & 1) * 8, ly8 = ly + (k >> 1) * 16 + (e >> 1) * 8;
                        st->err8[idx]  = 0x7fffffff;
                        st->mv8_x[idx] = (int16_t)(mv_x_of(best_mv[21 + idx]) << 3), st->mv8_y[idx] = (int16_t)(mv_y_of(best_mv[21 + idx]) << 3);
                        subpel_search(&s, 8, lx8, ly8, K_REGULAR, &st->err8[idx], &st->mv8_x[idx], &st->mv8_y[idx]);
                    }
            int sum16 = 0;
            for (int k = 0; k < 4; k++) {
                const int q = i * 4 + k;
                int       sub = (int)st->err16[q];
                if (c->enable_8x8_pred) {
                    int e8 = 0;
                    for (int e = 0; e < 4; e++) e8 += (int)st->err8[q * 4 + e];
                    if (sub * 8 < e8 * 16) {
                        st->split16[q] = 0;
                    } else {
                        st->split16[q] = 1, st->err16[q] = (uint64_t)e8, sub = e8;
                    }
                }
                sum16 += sub;
            }
            st->split32[i] = !((int)st->err32[i] * 14 < sum16 * 16);
        }
        /* tf_32x32_inter_prediction */
        if (st->split32[i])
            for (int k = 0; k < 4; k++) {
                if (st->split16[i * 4 + k]) {
                    for (int e = 0; e < 4; e++)
                        final_prediction(job, &s.g, ref, pred, ox, oy, lx + (k & 1) * 16 + (e & 1) * 8, ly + (k >> 1) * 16 + (e >> 1) * 8, 8,
                                         st->mv8_x[i * 16 + k * 4 + e], st->mv8_y[i * 16 + k * 4 + e]);
                } else {
                    final_prediction(job, &s.g, ref, pred, ox, oy, lx + (k & 1) * 16, ly + (k >> 1) * 16, 16, st->mv16_x[i * 4 + k], st->mv16_y[i * 4 + k]);
                }
            }
        else
            final_prediction(job, &s.g, ref, pred, ox, oy, lx, ly, 32, st->mv32_x[i], st->mv32_y[i]);
    }
}

/* one SvtHipTfBlock per 32x32 of the b64 at (ox, oy): the static fields */
static void tf_block_of(const SvtHipTfPictureJob *job, int ox, int oy, int q, void *const pred[3], uint32_t *const accum[3], uint16_t *const count[3],
                        SvtHipTfBlock *b) {
    const int is16 = job->bit_depth > 8, lx = (q & 1) * 32, ly = (q >> 1) * 32;
    const SvtHipPlane8 *cf = &job->centre.pyr.full;
    memset(b, 0, sizeof(*b));
    for (int p = 0; p < 3; p++) {
        const int    ss = p ? 1 : 0, ps = p ? 32 : 64;
        const size_t stride = p ? job->centre.chroma8_stride : cf->stride;
        const size_t org = (size_t)(cf->org_y >> ss) * stride + (cf->org_x >> ss) + (size_t)((oy + ly) >> ss) * stride + ((ox + lx) >> ss);
        const uint8_t *base = is16 ? (const uint8_t *)job->centre.hbd[p] : (p ? job->centre.chroma8[p - 1] : cf->buf);
        b->src[p]           = base + (org << is16);
        const size_t po     = (size_t)(ly >> ss) * ps + (lx >> ss);
        b->pred[p]          = (const uint8_t *)pred[p] + (po << is16);
        b->accum[p] = accum[p] + po, b->count[p] = count[p] + po;
        b->src_stride[p] = (uint32_t)stride, b->pred_stride[p] = (uint32_t)ps;
        b->decay_factor_fp16[p] = job->decay_factor_fp16[p];
    }
    b->mv_dist_th = job->mv_dist_th, b->chroma = job->chroma, b->ss_x = b->ss_y = 1;
    b->is_16bit = (uint8_t)is16, b->bit_depth = job->bit_depth, b->zz_based = job->ctrls.use_zz_based_filter;
}

/* states: [n_refs][n_b64] or NULL; tot[2] += horizontal / vertical block counts (motion_estimation.c:2539-2544) */
ORC_API int32_t orc_tf_filter_picture(const SvtHipTfPictureJob *job, SvtHipTfB64State *states, uint32_t *tot) {
    if (job->n_refs > SVT_HIP_TF_MAX_REFS || (job->bit_depth != 8 && job->bit_depth != 10))
        return -1;
    const SvtHipPlane8 *cf = &job->centre.pyr.full;
    const uint32_t W = cf->width, H = cf->height, bw = (W + 63) / 64, bh = (H + 63) / 64, nb = bw * bh;
    const int      is16 = job->bit_depth > 8;
    /* ME_MCTF against every reference picture */
    uint32_t             *best_sad = calloc((size_t)nb * 2 * 4 * 85, 4), *best_mv = calloc((size_t)nb * 2 * 4 * 85, 4);
    SvtHipMeSearchResult *sr       = calloc((size_t)nb * 2 * 4, sizeof(*sr));
    uint32_t             *accum    = malloc((size_t)nb * 3 * 4096 * 4);
    uint16_t             *count    = malloc((size_t)nb * 3 * 4096 * 2);
    uint8_t              *pred     = malloc((size_t)3 * 4096 * 2), *tmp = malloc((size_t)4096 * 2);
    SvtHipMeFrameJob     *mj       = calloc(1, sizeof(*mj));
    if (!best_sad || !best_mv || !sr || !accum || !count || !pred || !tmp || !mj)
        return -2;
    memset(accum, 0, (size_t)nb * 3 * 4096 * 4), memset(count, 0, (size_t)nb * 3 * 4096 * 2);
    void *const predp[3] = {pred, pred + (4096 << is16), pred + (8192 << is16)};
    SvtHipTfBlock blk;
    for (uint32_t b = 0; b < nb; b++) {
        uint32_t *const ac[3] = {accum + (size_t)b * 3 * 4096, accum + (size_t)b * 3 * 4096 + 4096, accum + (size_t)b * 3 * 4096 + 8192};
        uint16_t *const cn[3] = {count + (size_t)b * 3 * 4096, count + (size_t)b * 3 * 4096 + 4096, count + (size_t)b * 3 * 4096 + 8192};
        for (int q = 0; q < 4; q++) {
            tf_block_of(job, (int)(b % bw) * 64, (int)(b / bw) * 64, q, predp, ac, cn, &blk);
            orc_tf_central(&blk);
        }
    }
    for (uint32_t r = 0; r < job->n_refs; r++) {
        mj->prm = job->me;
        mj->prm.me_mctf = 1, mj->prm.num_of_list_to_search = 1, mj->prm.num_of_ref_pic_to_search[0] = 1, mj->prm.num_of_ref_pic_to_search[1] = 0;
        mj->prm.picture_number = job->centre.picture_number, mj->prm.ref_picture_number[0][0] = job->ref[r].picture_number;
        mj->src = job->centre.pyr, mj->ref[0][0] = job->ref[r].pyr;
        mj->out.best_sad = best_sad, mj->out.best_mv = best_mv, mj->out.search_results = sr;
        int32_t rc = job->ctrls.low_delay ? 0 : orc_me_frame_range(mj, 0, nb); /* the low-delay variant has no motion search */
        if (rc)
            return rc;
        for (uint32_t b = 0; b < nb; b++) {
            const int ox = (int)(b % bw) * 64, oy = (int)(b / bw) * 64;
            const SvtHipMeSearchResult *s0 = sr + (size_t)b * 8;
            if (tot && !job->ctrls.low_delay)
                tot[abs(s0->hme_sc_x) > abs(s0->hme_sc_y) ? 0 : 1]++;
            SvtHipTfB64State st;
            refine_b64(job, &job->ref[r], ox, oy, best_mv + (size_t)b * 8 * 85, best_sad + (size_t)b * 8 * 85, s0, predp, tmp, &st);
            if (states)
                states[(size_t)r * nb + b] = st;
            uint32_t *const ac[3] = {accum + (size_t)b * 3 * 4096, accum + (size_t)b * 3 * 4096 + 4096, accum + (size_t)b * 3 * 4096 + 8192};
            uint16_t *const cn[3] = {count + (size_t)b * 3 * 4096, count + (size_t)b * 3 * 4096 + 4096, count + (size_t)b * 3 * 4096 + 8192};
            for (int q = 0; q < 4; q++) {
                tf_block_of(job, ox, oy, q, predp, ac, cn, &blk);
                blk.split = st.split32[q];
                if (blk.split)
                    for (int k = 0; k < 4; k++) blk.block_error[k] = st.err16[q * 4 + k], blk.mv_x[k] = st.mv16_x[q * 4 + k], blk.mv_y[k] = st.mv16_y[q * 4 + k];
                else
                    blk.block_error[0] = st.err32[q], blk.mv_x[0] = st.mv32_x[q], blk.mv_y[0] = st.mv32_y[q];
                orc_tf_accumulate(&blk);
            }
        }
    }
    /* get_final_filtered_pixels: the centre picture is overwritten */
    for (uint32_t b = 0; b < nb; b++) {
        uint32_t *const ac[3] = {accum + (size_t)b * 3 * 4096, accum + (size_t)b * 3 * 4096 + 4096, accum + (size_t)b * 3 * 4096 + 8192};
        uint16_t *const cn[3] = {count + (size_t)b * 3 * 4096, count + (size_t)b * 3 * 4096 + 4096, count + (size_t)b * 3 * 4096 + 8192};
        for (int q = 0; q < 4; q++) {
            tf_block_of(job, (int)(b % bw) * 64, (int)(b / bw) * 64, q, predp, ac, cn, &blk);
            SvtHipTfOut o;
            memset(&o, 0, sizeof(o));
            for (int p = 0; p < 3; p++) o.dst[p] = (void *)blk.src[p], o.dst_stride[p] = blk.src_stride[p];
            orc_tf_normalise(&blk, &o);
        }
    }
    free(best_sad), free(best_mv), free(sr), free(accum), free(count), free(pred), free(tmp), free(mj);
    return 0;
}

ORC_API uint32_t orc_sizeof_tf_picture_job(void) { return (uint32_t)sizeof(SvtHipTfPictureJob); }
ORC_API uint32_t orc_sizeof_tf_b64_state(void) { return (uint32_t)sizeof(SvtHipTfB64State); }
ORC_API uint32_t orc_sizeof_tf_ctrls(void) { return (uint32_t)sizeof(SvtHipTfCtrls); }

/*
 * oracle/src/orc_tpl.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the TPL dispenser of one picture for the configuration of include/svt_hip_tpl.h
 * (16x16 blocks, or 32x32 blocks with the transform on every 4th row = tpl level 5; DC intra prediction, SAD source search,
 * full-pel vectors, no rate estimate):
 *   tpl_mc_flow_dispenser_sb_generic                                   src_ops_process.c:519-1207
 *   get_neighbor_samples_dc                                            :360-373
 *   svt_aom_update_neighbor_samples_array_open_loop_mb / _mb_recon     enc_intra_prediction.c:1127-1300
 *   svt_aom_intra_prediction_open_loop_mb (DC_PRED) + dc predictors    intra_prediction.c:1023-1073, 2579-2601
 *   get_quantize_error, result_model_store                             src_ops_process.c:225-249, 266-340
 * built on the pinned pieces orc_nxm_sad, orc_fwd_txfm2d, orc_quantize_fp, orc_inv_txfm2d_add_8bit.
 * Pinned against the REAL function through oracle/ref_harness_tpl.c (tests/test_tpl_oracle.py) and tests/golden/tpl_frame.npz.
 * Every pointer inside the job is a HOST pointer here; `workspace` is not used.  Blocks run in the reference's order (64x64 blocks
 * in raster order, 16x16 / 32x32 blocks in z-order inside).
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/svt_hip_tpl.h"
#include "orc.h"
#include "orc_txfm.h"

#define TPL_PAD 32
#define NEWMV_MODE 16

/* both neighbour-array builders: pic0 = sample (0,0) of the plane, (x, y) block origin, width / height of the picture.
 * above[-1 .. 2*bw-1], left[-1 .. 2*bh-1] */
static void neighbours(uint8_t *above_ref, uint8_t *left_ref, const uint8_t *pic0, uint32_t stride, uint32_t x, uint32_t y, uint32_t bw, uint32_t bh,
                       uint32_t width, uint32_t height) {
    const uint32_t n = bw << 1, m = bh << 1;
    const uint8_t *src = pic0 + (size_t)y * stride + x;
    memset(above_ref, 127, n + 1), memset(left_ref, 129, m + 1);
    uint8_t *a = above_ref, *l = left_ref;
    if (x != 0 && y != 0)
        *a = *l = src[-(ptrdiff_t)stride - 1];
    else
        *a = *l = 128;
    a++, l++;
    uint32_t count = n; /* the left column is counted with the WIDTH of the neighbourhood, as in the reference */
    if (x != 0) {
        const uint8_t *rp = src - 1;
        if (y == 0)
            l[-1] = *rp;
        count = (y + count > height) ? count - (y + count - height) : count;
        for (uint32_t i = 0; i < count; i++, rp += stride) *l++ = *rp;
        l += n - count;
        for (uint32_t i = 0; i < bh; i++) l[-(ptrdiff_t)bh + i] = l[-(ptrdiff_t)bh - 1]; /* unknown bottom-left <- sample (-1, bh - 1) */
    } else if (y != 0) {
        count = (y + count > height) ? count - (y + count - height) : count;
        memset(l - 1, src[-(ptrdiff_t)stride], count + 1);
        a[-1] = src[-(ptrdiff_t)stride];
    } else
        l += count;
    count = n;
    if (y != 0) {
        count = (x + count > width) ? count - (x + count - width) : count;
        memcpy(a, src - stride, count);
        if (x != 0)
            for (uint32_t i = 0; i < bw; i++) a[bw + i] = a[bw - 1]; /* unknown top-right <- sample (bw - 1, -1) */
    } else if (x != 0) {
        count = (x + count > width) ? count - (x + count - width) : count;
        memset(a - 1, *(l - count), count + 1);
    }
}

/* svt_aom_dc_pred[x > 0][y > 0][TX_16X16 / TX_32X32] */
static uint8_t dc_value(const uint8_t *above, const uint8_t *left, uint32_t x, uint32_t y, int bs) {
    int32_t sa = 0, sl = 0;
    for (int i = 0; i < bs; i++) sa += above[i], sl += left[i];
    if (x > 0 && y > 0)
        return (uint8_t)((sa + sl + bs) / (2 * bs));
    if (x > 0)
        return (uint8_t)((sl + bs / 2) / bs);
    if (y > 0)
        return (uint8_t)((sa + bs / 2) / bs);
    return 128;
}

static void dc_predict(const uint8_t *pic0, uint32_t stride, uint32_t x, uint32_t y, uint32_t width, uint32_t height, uint8_t *dst, uint32_t dst_stride,
                       uint32_t bs) {
    uint8_t above_data[8 + 128 + 8], left_data[8 + 128 + 8];
    uint8_t *above = above_data + 8, *left = left_data + 8;
    const int inside = x + bs <= width && y + bs <= height;
    if (x > 0 && y > 0 && inside) { /* get_neighbor_samples_dc */
        const uint8_t *src = pic0 + (size_t)y * stride + x;
        memcpy(above, src - stride, bs);
        for (uint32_t i = 0; i < bs; i++) left[i] = src[(ptrdiff_t)i * stride - 1];
    } else {
        neighbours(above - 1, left - 1, pic0, stride, x, y, bs, bs, width, height);
    }
    const uint8_t v = dc_value(above, left, x, y, (int)bs);
    for (uint32_t r = 0; r < bs; r++) memset(dst + (size_t)r * dst_stride, v, bs);
}

/* subtract -> svt_av1_wht_fwd_txfm (DCT_DCT bs x (bs >> sub) on every (1 << sub)-th row, pf_shape) -> get_quantize_error; dqcoeff out */
static int64_t quantize_error(const SvtHipTplFrameJob *job, const uint8_t *src, uint32_t src_stride, const uint8_t *pred, uint32_t pred_stride,
                              int32_t *dqcoeff, uint16_t *eob, int bs, int sub) {
    int16_t   diff[1024], scan[1024];
    int32_t   coeff[1024], qcoeff[1024];
    const int th = bs >> sub, n = bs * th;
    for (int i = 0; i < n; i++) scan[i] = (int16_t)i; /* the scan only orders the end-of-block position, of which only "non-zero" is used */
    orc_subtract_block(th, bs, diff, bs, src, (ptrdiff_t)src_stride << sub, pred, (ptrdiff_t)pred_stride << sub);
    memset(coeff, 0, sizeof(coeff));
    orc_fwd_txfm2d(diff, coeff, (uint32_t)bs, bs, th, 0, 8, job->pf_shape);
    orc_quantize_fp(coeff, n, job->round_fp, job->quant_fp, qcoeff, dqcoeff, job->dequant, eob, scan, NULL, NULL, 0);
    int64_t err = 0;
    for (int i = 0; i < n; i++) {
        const int64_t d = (int64_t)coeff[i] - dqcoeff[i];
        err += d * d;
    }
    err >>= (bs == 32 && th == 32) ? 0 : 2; /* shift = tx_size == TX_32X32 ? 0 : 2 (:229) */
    return err > 1 ? err : 1;
}

static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

/* ---- tpl level 3: tpl_subpel_search (src_ops_process.c:418-517) = svt_av1_find_best_sub_pixel_tree_pruned (mcomp.c:609-695) with
 * allow_hp 0, forced_stop QUARTER_PEL (two rounds: hstep 4, 2), iters_per_step 2, skip_diag_refinement 4 (org_error 0: neither the
 * diagonal nor the second level is ever tried), MV_COST_NONE, abs_th_mult / pred_variance_th 0, round_dev_th MAX: per round the four
 * cardinal neighbours of the round's start vector through svt_check_better_fast -> vfp->svf = svt_aom_sub_pixel_varianceWxH_c
 * (variance.c:28-68, 303-318: 2-tap bilinear, first pass to 16 bit, second pass to 8 bit). */
static const uint8_t BIL_2T[8][2] = {{128, 0}, {112, 16}, {96, 32}, {80, 48}, {64, 64}, {48, 80}, {32, 96}, {16, 112}};
static uint32_t subpel_variance(const uint8_t *a, ptrdiff_t a_stride, int xo, int yo, const uint8_t *b, ptrdiff_t b_stride, int n) {
    uint16_t f[33 * 32];
    for (int i = 0; i <= n; i++)
        for (int j = 0; j < n; j++)
            f[i * n + j] = (uint16_t)(((int)a[i * a_stride + j] * BIL_2T[xo][0] + (int)a[i * a_stride + j + 1] * BIL_2T[xo][1] + 64) >> 7);
    int      sum = 0;
    uint32_t sse = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const int p = (uint8_t)(((int)f[i * n + j] * BIL_2T[yo][0] + (int)f[(i + 1) * n + j] * BIL_2T[yo][1] + 64) >> 7);
            const int d = p - b[i * b_stride + j];
            sum += d, sse += (uint32_t)(d * d);
        }
    return sse - (uint32_t)(((int64_t)sum * sum) / (n * n));
}
static inline int imax_(int a, int b) { return a > b ? a : b; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
/* mx / my: in the clipped full-pel candidate (1/8 units, multiple of 8), out the refined vector */
static void tpl_subpel_search(const uint8_t *src, ptrdiff_t ss, const uint8_t *ref0 /* sample (x, y) of the reference */, ptrdiff_t rs, int x, int y,
                              int bs, int mi_rows, int mi_cols, int16_t *mx, int16_t *my) {
    /* MvLimits of the block (:438-449), svt_av1_set_mv_search_range and svt_av1_set_subpel_mv_search_range around a zero ref_mv */
    const int mi_row = y >> 2, mi_col = x >> 2, mi = bs >> 2;
    int row_min = -(((mi_row + mi) * 4) + 4), col_min = -(((mi_col + mi) * 4) + 4);
    int row_max = (mi_rows - mi_row) * 4 + 4, col_max = (mi_cols - mi_col) * 4 + 4;
    col_min = imax_(col_min, imax_(-1023, (-16384 >> 3) + 1)), row_min = imax_(row_min, imax_(-1023, (-16384 >> 3) + 1));
    col_max = imin_(col_max, imin_(1023, (16384 >> 3) - 1)), row_max = imin_(row_max, imin_(1023, (16384 >> 3) - 1));
    const int sc_min = imax_(-16384 + 1, imax_(col_min * 8, -1023 * 8)), sc_max = imin_(16384 - 1, imin_(col_max * 8, 1023 * 8));
    const int sr_min = imax_(-16384 + 1, imax_(row_min * 8, -1023 * 8)), sr_max = imin_(16384 - 1, imin_(row_max * 8, 1023 * 8));
    int best_r = (*my >> 3) * 8, best_c = (*mx >> 3) * 8; /* best_mv >> 3, back to 1/8 units */
    unsigned besterr;
    {
        const uint8_t *a = ref0 + (ptrdiff_t)(best_r >> 3) * rs + (best_c >> 3);
        int      sum = 0;
        uint32_t sse = 0;
        for (int i = 0; i < bs; i++)
            for (int j = 0; j < bs; j++) {
                const int d = a[i * rs + j] - src[i * ss + j];
                sum += d, sse += (uint32_t)(d * d);
            }
        besterr = sse - (uint32_t)(((int64_t)sum * sum) / (bs * bs));
    }
    int start_r = best_r, start_c = best_c;
    for (int iter = 0, hstep = 4; iter < 2; iter++, hstep >>= 1) {
        const int cr[4] = {start_r, start_r, start_r - hstep, start_r + hstep}, cc[4] = {start_c - hstep, start_c + hstep, start_c, start_c};
        for (int k = 0; k < 4; k++) { /* left, right, up, down */
            if (cc[k] < sc_min || cc[k] > sc_max || cr[k] < sr_min || cr[k] > sr_max)
                continue;
            const uint8_t *a = ref0 + (ptrdiff_t)(cr[k] >> 3) * rs + (cc[k] >> 3);
            const unsigned cost = subpel_variance(a, rs, cc[k] & 7, cr[k] & 7, src, ss, bs);
            if (cost < besterr)
                besterr = cost, best_r = cr[k], best_c = cc[k];
        }
        start_r = best_r, start_c = best_c;
    }
    *mx = (int16_t)best_c, *my = (int16_t)best_r;
}
/* svt_aom_enc_make_inter_predictor of the luma block (:814-850): regular 8-tap kernels, vector clamped as
 * clamp_mv_to_umv_border_sb does with the xd of init_xd_tpl; plane0 = sample (0, 0) of the reference plane */
extern const int16_t orc_interp_kernels[4][16][8];
static void tpl_predict(const uint8_t *plane0, ptrdiff_t stride, int x, int y, int bs, int mvx, int mvy, int mi_rows, int mi_cols, uint8_t *dst,
                        ptrdiff_t dst_stride) {
    const int     mirow = y >> 2, micol = x >> 2, bmi = bs >> 2;
    const int32_t to_top = -((mirow * 4) * 8), to_bottom = ((mi_rows - bmi - mirow) * 4) * 8;
    const int32_t to_left = -((micol * 4) * 8), to_right = ((mi_cols - bmi - micol) * 4) * 8;
    const int32_t spel_left = (4 + bs) << 4, spel_right = spel_left - 16, spel_top = (4 + bs) << 4, spel_bottom = spel_top - 16;
    int col = (int16_t)(mvx * 2), row = (int16_t)(mvy * 2);
    col = imax_(to_left * 2 - spel_left, imin_(col, to_right * 2 + spel_right)), row = imax_(to_top * 2 - spel_top, imin_(row, to_bottom * 2 + spel_bottom));
    const int sx = col & 15, sy = row & 15;
    const uint8_t *src = plane0 + (ptrdiff_t)(y + (row >> 4)) * stride + x + (col >> 4);
    orc_convolve_sr(src, (int32_t)stride, dst, (int32_t)dst_stride, bs, bs, orc_interp_kernels[0][sx], sx ? 8 : 0, orc_interp_kernels[0][sy], sy ? 8 : 0, 3, 11, 8,
                    0);
}

ORC_API int32_t orc_tpl_dispenser_frame(const SvtHipTplFrameJob *job) {
    const uint32_t W = job->src.width, H = job->src.height, aw = (W + 7) & ~7u, ah = (H + 7) & ~7u;
    const uint32_t bw64 = (aw + 63) / 64, bh64 = (ah + 63) / 64, a16 = (aw + 15) >> 4, rows16 = (ah + 15) >> 4;
    const int      mi_rows = (int)(ah >> 2), mi_cols = (int)(aw >> 2); /* Av1Common of the picture */
    const uint8_t *src0 = job->src.buf + (size_t)job->src.org_y * job->src.stride + job->src.org_x;
    uint8_t       *rec0 = job->recon.buf + (size_t)job->recon.org_y * job->recon.stride + job->recon.org_x;
    const uint32_t ss = job->src.stride, rs = job->recon.stride;
    const int      sub = job->subsample_tx;
    if (job->synth_blk_size != 16 && job->synth_blk_size != 8 && !(job->synth_blk_size == 32 && job->blk_size == 32))
        return -1;
    if ((job->blk_size != 0 && job->blk_size != 16 && job->blk_size != 32) || (sub != 0 && sub != 2) || (job->blk_size != 32 && sub != 0))
        return -1;
    if (job->quarter_pel && job->blk_size == 32)
        return -1; /* set_tpl_params: QUARTER_PEL only with dispenser_search_level 0 */
    for (uint32_t sb = 0; sb < bw64 * bh64; sb++) {
        /* the caller dispenses an incomplete 64x64 block (right / bottom picture edge) with 16x16 blocks whatever the level
         * (svt_aom_tpl_disp_kernel, :2043-2051); the transform sub-sampling stays: TX_16X4 there */
        const int      full = aw - (sb % bw64) * 64 >= 64 && ah - (sb / bw64) * 64 >= 64;
        const uint32_t bs = (job->blk_size == 32 && full) ? 32 : 16; /* size_array[dispenser_search_level] */
        const uint32_t nz = bs == 32 ? 4 : 16;
        for (uint32_t z = 0; z < nz; z++) {
            /* z-order inside the b64 (tpl_blk_idx_tab[0][blk_start .. blk_end]) */
            const uint32_t bx = bs == 32 ? (z & 1) : ((z & 1) | ((z >> 2) & 1) << 1), by = bs == 32 ? (z >> 1) : (((z >> 1) & 1) | ((z >> 3) & 1) << 1);
            const uint32_t x = (sb % bw64) * 64 + bx * bs, y = (sb / bw64) * 64 + by * bs;
            if (x + (bs >> 1) > W || y + (bs >> 1) > H) /* at least half of the block inside */
                continue;
            const uint8_t *src = src0 + (size_t)y * ss + x;
            uint8_t       *dst = rec0 + (size_t)y * rs + x;
            SvtHipTplSrcStats *sst = &job->src_stats[(size_t)(y >> 4) * a16 + (x >> 4)];
            SvtHipTplStats     st;
            memset(&st, 0, sizeof(st));
            int64_t  recon_error = 1;
            uint64_t best_ref_poc = 0;
            int32_t  best_rf_idx = -1;
            int16_t  mv_row = 0, mv_col = 0;
            uint8_t  best_mode = 0;
            uint8_t  pred[1024], comp[1024];
            int32_t  dq[1024];
            uint16_t eob;
            if (!job->src_data_ready) {
                int64_t best_inter = INT64_MAX, best_intra = INT64_MAX;
                if (!job->disable_intra_pred) {
                    dc_predict(src0, ss, x, y, W, H, pred, bs, bs);
                    best_intra = orc_nxm_sad(src, ss, pred, bs, bs, bs);
                }
                uint32_t me_off = bs == 32 ? 1 + by * 2 + bx : 5 + by * 4 + bx; /* tpl_blk_idx_tab[1] */
                if (!job->enable_me_16x16)
                    me_off = (me_off - 1) / 4;
                const size_t   pu = (size_t)sb * job->stored_pus + me_off;
                const uint8_t *cands = job->me_candidate_array + pu * job->max_cand;
                const uint32_t n_cand = job->i_slice ? 0 : job->total_me_candidate_index[pu];
                for (uint32_t ci = 0; ci < n_cand; ci++) {
                    const uint32_t dir = cands[ci] & 3;
                    if (dir > 1)
                        continue;
                    const uint32_t ri = dir == 0 ? (cands[ci] >> 2) & 3 : (cands[ci] >> 4) & 3;
                    const SvtHipTplRef *rf = &job->ref[dir][ri];
                    if (!rf->usable)
                        continue;
                    const uint32_t mv = job->me_mv_array[pu * job->max_refs + (dir ? job->max_l0 : 0) + ri];
                    int16_t mx = (int16_t)((int16_t)(mv & 0xffff) << 3), my = (int16_t)((int16_t)(mv >> 16) << 3);
                    if ((int)x + (mx >> 3) < -TPL_PAD)
                        mx = (int16_t)((-TPL_PAD - (int)x) << 3);
                    if ((int)x + (int)bs + (mx >> 3) > TPL_PAD + (int)rf->max_width - 1)
                        mx = (int16_t)(((TPL_PAD + (int)rf->max_width - 1) - ((int)x + (int)bs)) << 3);
                    if ((int)y + (my >> 3) < -TPL_PAD)
                        my = (int16_t)((-TPL_PAD - (int)y) << 3);
                    if ((int)y + (int)bs + (my >> 3) > TPL_PAD + (int)rf->max_height - 1)
                        my = (int16_t)(((TPL_PAD + (int)rf->max_height - 1) - ((int)y + (int)bs)) << 3);
                    const uint8_t *rp = rf->src + ((ptrdiff_t)y + my / 8) * (ptrdiff_t)rf->src_stride + (ptrdiff_t)x + mx / 8;
                    ptrdiff_t      rps = rf->src_stride;
                    if (job->quarter_pel) {
                        tpl_subpel_search(src, ss, rf->src + (ptrdiff_t)y * rf->src_stride + x, rf->src_stride, (int)x, (int)y, (int)bs, mi_rows, mi_cols, &mx, &my);
                        rp = rf->src + ((ptrdiff_t)y + my / 8) * (ptrdiff_t)rf->src_stride + (ptrdiff_t)x + mx / 8;
                        if ((mx & 7) || (my & 7)) {
                            tpl_predict(rf->src, rf->src_stride, (int)x, (int)y, (int)bs, mx, my, mi_rows, mi_cols, comp, bs);
                            rp = comp, rps = bs;
                        }
                    }
                    const int64_t  cost = orc_nxm_sad(src, ss, rp, (uint32_t)rps, bs, bs);
                    if (cost < best_inter)
                        best_inter = cost, best_ref_poc = rf->picture_number, best_rf_idx = (int32_t)(dir * 4 + ri), mv_row = my, mv_col = mx;
                }
                if (best_inter < best_intra)
                    best_mode = NEWMV_MODE;
                if (best_mode == NEWMV_MODE) {
                    const SvtHipTplRef *rf = &job->ref[best_rf_idx < 4 ? 0 : 1][best_rf_idx & 3];
                    const uint8_t      *rp = rf->src + ((ptrdiff_t)y + (mv_row >> 3)) * (ptrdiff_t)rf->src_stride + (ptrdiff_t)x + (mv_col >> 3);
                    uint32_t            rps = rf->src_stride;
                    if ((mv_col & 7) || (mv_row & 7)) {
                        tpl_predict(rf->src, rf->src_stride, (int)x, (int)y, (int)bs, mv_col, mv_row, mi_rows, mi_cols, comp, bs);
                        rp = comp, rps = bs;
                    }
                    recon_error = quantize_error(job, src, ss, rp, rps, dq, &eob, (int)bs, sub);
                    st.srcrf_rate = 0, st.srcrf_dist = (recon_error << 4) << sub;
                }
                if (job->store_src_stats) {
                    memset(sst, 0, sizeof(*sst));
                    sst->srcrf_dist = st.srcrf_dist, sst->srcrf_rate = st.srcrf_rate, sst->mv_row = mv_row, sst->mv_col = mv_col;
                    sst->best_rf_idx = best_rf_idx, sst->ref_frame_poc = best_ref_poc, sst->best_mode = best_mode, sst->best_intra_mode = 0;
                }
            } else {
                st.srcrf_dist = sst->srcrf_dist, st.srcrf_rate = sst->srcrf_rate, mv_row = sst->mv_row, mv_col = sst->mv_col;
                best_rf_idx = sst->best_rf_idx, best_ref_poc = sst->ref_frame_poc, best_mode = sst->best_mode;
            }
            /* reconstruction path */
            if (best_mode == NEWMV_MODE) {
                const SvtHipTplRef *rf = &job->ref[best_rf_idx < 4 ? 0 : 1][best_rf_idx & 3];
                const uint8_t      *rp = rf->recon + ((ptrdiff_t)y + (mv_row >> 3)) * (ptrdiff_t)rf->recon_stride + (ptrdiff_t)x + (mv_col >> 3);
                if ((mv_col & 7) || (mv_row & 7))
                    tpl_predict(rf->recon, rf->recon_stride, (int)x, (int)y, (int)bs, mv_col, mv_row, mi_rows, mi_cols, dst, rs);
                else
                    for (uint32_t r = 0; r < bs; r++) memcpy(dst + (size_t)r * rs, rp + (ptrdiff_t)r * rf->recon_stride, bs);
            } else {
                dc_predict(rec0, rs, x, y, W, H, dst, rs, bs);
            }
            recon_error = quantize_error(job, src, ss, dst, rs, dq, &eob, (int)bs, sub);
            if (!job->disable_intra_pred || job->is_ref)
                if (eob) {
                    orc_inv_txfm2d_add_8bit(dq, dst, (int32_t)(rs << sub), dst, (int32_t)(rs << sub), (int)bs, (int)(bs >> sub), 0);
                    /* the rows the sub-sampled transform left out repeat the row above them (:1162-1180) */
                    for (uint32_t i = 0; sub && i < bs; i += 1u << sub)
                        for (uint32_t k = 1; k < (1u << sub); k++) memcpy(dst + (size_t)(i + k) * rs, dst + (size_t)i * rs, bs);
                }
            st.recrf_dist = (recon_error << 4) << sub, st.recrf_rate = 0;
            if (best_mode != NEWMV_MODE)
                st.srcrf_dist = (recon_error << 4) << sub, st.srcrf_rate = 0;
            st.recrf_dist = max64(st.srcrf_dist, st.recrf_dist), st.recrf_rate = max64(st.srcrf_rate, st.recrf_rate);
            if (!job->tpl_i_slice && best_rf_idx != -1)
                st.mv_row = mv_row, st.mv_col = mv_col, st.ref_frame_poc = best_ref_poc;
            /* result_model_store: the block's statistics on the synthesizer's grid, normalised to the grid's cell size; cells
             * beyond the grid (a 32x32 block half outside the picture) are dropped — the reference writes them past the row end */
            st.srcrf_dist = max64(1, st.srcrf_dist), st.recrf_dist = max64(1, st.recrf_dist);
            st.srcrf_rate = max64(1, st.srcrf_rate), st.recrf_rate = max64(1, st.recrf_rate);
            const uint32_t cell = job->synth_blk_size, per = bs / cell; /* cells per block side: 1, 2 or 4; 0: a 16x16 block on the 32x32 grid overwrites its cell */
            const uint32_t gstride = cell == 32 ? (aw + 31) / 32 : (cell == 16 ? a16 : a16 << 1);
            const uint32_t grows = cell == 32 ? (ah + 31) / 32 : (cell == 16 ? rows16 : rows16 << 1);
            if (per > 1) {
                const int64_t div = (int64_t)per * per;
                st.srcrf_dist = max64(1, st.srcrf_dist / div), st.recrf_dist = max64(1, st.recrf_dist / div);
                st.srcrf_rate = max64(1, st.srcrf_rate / div), st.recrf_rate = max64(1, st.recrf_rate / div);
            }
            for (uint32_t cy = 0; cy < (per ? per : 1); cy++)
                for (uint32_t cx = 0; cx < (per ? per : 1); cx++) {
                    const uint32_t gx = x / cell + cx, gy = y / cell + cy;
                    if (gx < gstride && gy < grows)
                        job->stats[(size_t)gy * gstride + gx] = st;
                }
        }
    }
    return 0;
}

ORC_API uint32_t orc_sizeof_tpl_job(void) { return (uint32_t)sizeof(SvtHipTplFrameJob); }
ORC_API uint32_t orc_sizeof_tpl_stats(void) { return (uint32_t)sizeof(SvtHipTplStats); }
ORC_API uint32_t orc_sizeof_tpl_src_stats(void) { return (uint32_t)sizeof(SvtHipTplSrcStats); }

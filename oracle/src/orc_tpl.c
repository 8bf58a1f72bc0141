/*
 * oracle/src/orc_tpl.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the TPL dispenser of one picture for the configuration of include/svt_hip_tpl.h
 * (16x16 blocks, DC intra prediction, SAD source search, full-pel vectors, no transform sub-sampling, no rate estimate):
 *   tpl_mc_flow_dispenser_sb_generic                                   src_ops_process.c:519-1207
 *   get_neighbor_samples_dc                                            :360-373
 *   svt_aom_update_neighbor_samples_array_open_loop_mb / _mb_recon     enc_intra_prediction.c:1127-1300
 *   svt_aom_intra_prediction_open_loop_mb (DC_PRED) + dc predictors    intra_prediction.c:1023-1073, 2579-2601
 *   get_quantize_error, result_model_store                             src_ops_process.c:225-249, 266-340
 * built on the pinned pieces orc_nxm_sad, orc_fwd_txfm2d, orc_quantize_fp, orc_inv_txfm2d_add_8bit.
 * Pinned against the REAL function through oracle/ref_harness_tpl.c (tests/test_tpl_oracle.py) and tests/golden/tpl_frame.npz.
 * Every pointer inside the job is a HOST pointer here; `workspace` is not used.  Blocks run in the reference's order (64x64 blocks
 * in raster order, 16x16 blocks in z-order inside).
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

/* svt_aom_dc_pred[x > 0][y > 0][TX_16X16] */
static uint8_t dc_value(const uint8_t *above, const uint8_t *left, uint32_t x, uint32_t y) {
    int32_t sa = 0, sl = 0;
    for (int i = 0; i < 16; i++) sa += above[i], sl += left[i];
    if (x > 0 && y > 0)
        return (uint8_t)((sa + sl + 16) / 32);
    if (x > 0)
        return (uint8_t)((sl + 8) / 16);
    if (y > 0)
        return (uint8_t)((sa + 8) / 16);
    return 128;
}

static void dc_predict(const uint8_t *pic0, uint32_t stride, uint32_t x, uint32_t y, uint32_t width, uint32_t height, uint8_t *dst, uint32_t dst_stride) {
    uint8_t above_data[8 + 64 + 8], left_data[8 + 64 + 8];
    uint8_t *above = above_data + 8, *left = left_data + 8;
    const int inside = x + 16 <= width && y + 16 <= height;
    if (x > 0 && y > 0 && inside) { /* get_neighbor_samples_dc */
        const uint8_t *src = pic0 + (size_t)y * stride + x;
        memcpy(above, src - stride, 16);
        for (int i = 0; i < 16; i++) left[i] = src[(ptrdiff_t)i * stride - 1];
    } else {
        neighbours(above - 1, left - 1, pic0, stride, x, y, 16, 16, width, height);
    }
    const uint8_t v = dc_value(above, left, x, y);
    for (int r = 0; r < 16; r++) memset(dst + (size_t)r * dst_stride, v, 16);
}

/* subtract -> svt_av1_wht_fwd_txfm (DCT_DCT 16x16, pf_shape) -> get_quantize_error; dqcoeff out */
static int64_t quantize_error(const SvtHipTplFrameJob *job, const uint8_t *src, uint32_t src_stride, const uint8_t *pred, uint32_t pred_stride,
                              int32_t *dqcoeff, uint16_t *eob) {
    static const int16_t identity_scan[256] = {0};
    int16_t diff[256], scan[256];
    int32_t coeff[256], qcoeff[256];
    (void)identity_scan;
    for (int i = 0; i < 256; i++) scan[i] = (int16_t)i; /* the scan only orders the end-of-block position, which nothing here depends on */
    orc_subtract_block(16, 16, diff, 16, src, src_stride, pred, pred_stride);
    memset(coeff, 0, sizeof(coeff));
    orc_fwd_txfm2d(diff, coeff, 16, 16, 16, 0, 8, job->pf_shape);
    orc_quantize_fp(coeff, 256, job->round_fp, job->quant_fp, qcoeff, dqcoeff, job->dequant, eob, scan, NULL, NULL, 0);
    int64_t err = 0;
    for (int i = 0; i < 256; i++) {
        const int64_t d = (int64_t)coeff[i] - dqcoeff[i];
        err += d * d;
    }
    err >>= 2;
    return err > 1 ? err : 1;
}

static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

ORC_API int32_t orc_tpl_dispenser_frame(const SvtHipTplFrameJob *job) {
    const uint32_t W = job->src.width, H = job->src.height, aw = (W + 7) & ~7u, ah = (H + 7) & ~7u;
    const uint32_t bw64 = (aw + 63) / 64, bh64 = (ah + 63) / 64, a16 = (aw + 15) >> 4;
    const uint8_t *src0 = job->src.buf + (size_t)job->src.org_y * job->src.stride + job->src.org_x;
    uint8_t       *rec0 = job->recon.buf + (size_t)job->recon.org_y * job->recon.stride + job->recon.org_x;
    const uint32_t ss = job->src.stride, rs = job->recon.stride;
    if (job->synth_blk_size != 16 && job->synth_blk_size != 8)
        return -1;
    for (uint32_t sb = 0; sb < bw64 * bh64; sb++)
        for (uint32_t z = 0; z < 16; z++) {
            const uint32_t bx = (z & 1) | ((z >> 2) & 1) << 1, by = ((z >> 1) & 1) | ((z >> 3) & 1) << 1; /* z-order inside the b64 */
            const uint32_t x = (sb % bw64) * 64 + bx * 16, y = (sb / bw64) * 64 + by * 16;
            if (x + 8 > W || y + 8 > H) /* at least half of the block inside */
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
            uint8_t  pred[256];
            int32_t  dq[256];
            uint16_t eob;
            if (!job->src_data_ready) {
                int64_t best_inter = INT64_MAX, best_intra = INT64_MAX;
                if (!job->disable_intra_pred) {
                    dc_predict(src0, ss, x, y, W, H, pred, 16);
                    best_intra = orc_nxm_sad(src, ss, pred, 16, 16, 16);
                }
                uint32_t me_off = 5 + by * 4 + bx; /* tpl_blk_idx_tab[1] */
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
                    if ((int)x + 16 + (mx >> 3) > TPL_PAD + (int)rf->max_width - 1)
                        mx = (int16_t)(((TPL_PAD + (int)rf->max_width - 1) - ((int)x + 16)) << 3);
                    if ((int)y + (my >> 3) < -TPL_PAD)
                        my = (int16_t)((-TPL_PAD - (int)y) << 3);
                    if ((int)y + 16 + (my >> 3) > TPL_PAD + (int)rf->max_height - 1)
                        my = (int16_t)(((TPL_PAD + (int)rf->max_height - 1) - ((int)y + 16)) << 3);
                    const uint8_t *rp = rf->src + ((ptrdiff_t)y + my / 8) * (ptrdiff_t)rf->src_stride + (ptrdiff_t)x + mx / 8;
                    const int64_t  cost = orc_nxm_sad(src, ss, rp, rf->src_stride, 16, 16);
                    if (cost < best_inter)
                        best_inter = cost, best_ref_poc = rf->picture_number, best_rf_idx = (int32_t)(dir * 4 + ri), mv_row = my, mv_col = mx;
                }
                if (best_inter < best_intra)
                    best_mode = NEWMV_MODE;
                if (best_mode == NEWMV_MODE) {
                    const SvtHipTplRef *rf = &job->ref[best_rf_idx < 4 ? 0 : 1][best_rf_idx & 3];
                    const uint8_t      *rp = rf->src + ((ptrdiff_t)y + (mv_row >> 3)) * (ptrdiff_t)rf->src_stride + (ptrdiff_t)x + (mv_col >> 3);
                    recon_error = quantize_error(job, src, ss, rp, rf->src_stride, dq, &eob);
                    st.srcrf_rate = 0, st.srcrf_dist = recon_error << 4;
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
                for (int r = 0; r < 16; r++) memcpy(dst + (size_t)r * rs, rp + (ptrdiff_t)r * rf->recon_stride, 16);
            } else {
                dc_predict(rec0, rs, x, y, W, H, dst, rs);
            }
            recon_error = quantize_error(job, src, ss, dst, rs, dq, &eob);
            if (!job->disable_intra_pred || job->is_ref)
                if (eob)
                    orc_inv_txfm2d_add_8bit(dq, dst, (int32_t)rs, dst, (int32_t)rs, 16, 16, 0);
            st.recrf_dist = recon_error << 4, st.recrf_rate = 0;
            if (best_mode != NEWMV_MODE)
                st.srcrf_dist = recon_error << 4, st.srcrf_rate = 0;
            st.recrf_dist = max64(st.srcrf_dist, st.recrf_dist), st.recrf_rate = max64(st.srcrf_rate, st.recrf_rate);
            if (!job->tpl_i_slice && best_rf_idx != -1)
                st.mv_row = mv_row, st.mv_col = mv_col, st.ref_frame_poc = best_ref_poc;
            /* result_model_store */
            st.srcrf_dist = max64(1, st.srcrf_dist), st.recrf_dist = max64(1, st.recrf_dist);
            st.srcrf_rate = max64(1, st.srcrf_rate), st.recrf_rate = max64(1, st.recrf_rate);
            if (job->synth_blk_size == 16) {
                job->stats[(size_t)(y >> 4) * a16 + (x >> 4)] = st;
            } else {
                const uint32_t stride = a16 << 1;
                st.srcrf_dist = max64(1, st.srcrf_dist / 4), st.recrf_dist = max64(1, st.recrf_dist / 4);
                st.srcrf_rate = max64(1, st.srcrf_rate / 4), st.recrf_rate = max64(1, st.recrf_rate / 4);
                SvtHipTplStats *d = &job->stats[(size_t)(y >> 3) * stride + (x >> 3)];
                d[0] = d[1] = d[stride] = d[stride + 1] = st;
            }
        }
    return 0;
}

ORC_API uint32_t orc_sizeof_tpl_job(void) { return (uint32_t)sizeof(SvtHipTplFrameJob); }
ORC_API uint32_t orc_sizeof_tpl_stats(void) { return (uint32_t)sizeof(SvtHipTplStats); }
ORC_API uint32_t orc_sizeof_tpl_src_stats(void) { return (uint32_t)sizeof(SvtHipTplSrcStats); }

/*
 * oracle/ref_harness_tpl.c — TEST INFRASTRUCTURE.  Compiled ONLY into oracle/_ref/libsvtref.so.
 * tpl_mc_flow_dispenser_sb_generic (src_ops_process.c:519-1207) is `static` in the reference.  To pin the oracle's
 * restatement of the TPL dispenser against the REAL code, this translation unit compiles the reference's src_ops_process.c in
 * place (by #include from where it lies, nothing is copied) and exposes one entry point that builds the control structures
 * around the flat SvtHipTplFrameJob of include/svt_hip_tpl.h (host pointers) and runs the real function over every 64x64
 * block of the picture in raster order; oracle/Makefile makes every other symbol of this object local.
 */
#include "src_ops_process.c"

#include "../include/svt_hip_tpl.h"

void ref_init(void);
void svt_aom_init_intra_dc_predictors_c_internal(void);
void svt_aom_init_intra_predictors_internal(void);
void svt_aom_asm_set_convolve_asm_table(void);
void svt_aom_asm_set_convolve_hbd_asm_table(void);
void init_fn_ptr(void);

static void plane_desc(EbPictureBufferDesc *d, const uint8_t *sample0, uint32_t stride, uint16_t org_x, uint16_t org_y, uint16_t w, uint16_t h) {
    memset(d, 0, sizeof(*d));
    d->buffer_y = (uint8_t *)sample0 - ((size_t)org_y * stride + org_x);
    d->stride_y = (uint16_t)stride, d->org_x = org_x, d->org_y = org_y, d->width = w, d->height = h;
    d->max_width = w, d->max_height = h, d->bit_depth = EB_EIGHT_BIT;
}

/* quants_8bit / deq_8bit at qindex: round_fp[2], quant_fp[2], dequant[2] */
__attribute__((visibility("default"))) void ref_tpl_quant(int qindex, int16_t *out) {
    Quants   *q = calloc(1, sizeof(*q));
    Dequants *d = calloc(1, sizeof(*d));
    svt_av1_build_quantizer(EB_EIGHT_BIT, 0, 0, 0, 0, 0, q, d);
    out[0] = q->y_round_fp[qindex][0], out[1] = q->y_round_fp[qindex][1], out[2] = q->y_quant_fp[qindex][0], out[3] = q->y_quant_fp[qindex][1];
    out[4] = d->y_dequant_qtx[qindex][0], out[5] = d->y_dequant_qtx[qindex][1];
    free(q), free(d);
}

__attribute__((visibility("default"))) int ref_tpl_dispenser_frame(const SvtHipTplFrameJob *job, int qindex) {
    ref_init();
    static int tables_done;
    if (!tables_done) { /* one-time table set-up of svt_av1_enc_init (enc_handle.c:1478-1491) */
        svt_aom_init_intra_dc_predictors_c_internal();
        svt_aom_init_intra_predictors_internal();
        svt_aom_asm_set_convolve_asm_table(); /* tpl level 3: svt_aom_enc_make_inter_predictor, svt_aom_mefn_ptr */
        svt_aom_asm_set_convolve_hbd_asm_table();
        init_fn_ptr();
        tables_done = 1;
    }
    const uint32_t W = job->src.width, H = job->src.height;
    const uint32_t aw = (W + 7) & ~7u, ah = (H + 7) & ~7u, bw64 = (aw + 63) / 64, bh64 = (ah + 63) / 64, nb = bw64 * bh64;
    SequenceControlSet      *scs  = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs  = calloc(1, sizeof(*pcs)), *base = calloc(1, sizeof(*base));
    EncodeContext           *enc  = calloc(1, sizeof(*enc));
    MotionEstimationData    *med  = calloc(1, sizeof(*med));
    Av1Common               *cm   = calloc(1, sizeof(*cm));
    B64Geom                 *geom = calloc(nb, sizeof(*geom));
    MeSbResults             *res  = calloc(nb, sizeof(*res));
    if (!scs || !pcs || !base || !enc || !med || !cm || !geom || !res)
        return -1;
    EbPictureBufferDesc src, recon, ref_src[2][4], ref_rec[2][4];
    plane_desc(&src, job->src.buf + (size_t)job->src.org_y * job->src.stride + job->src.org_x, job->src.stride, job->src.org_x, job->src.org_y, (uint16_t)W, (uint16_t)H);
    plane_desc(&recon, job->recon.buf + (size_t)job->recon.org_y * job->recon.stride + job->recon.org_x, job->recon.stride, job->recon.org_x,
               job->recon.org_y, (uint16_t)W, (uint16_t)H);
    scs->enc_ctx = enc, scs->b64_geom = geom, scs->in_loop_ois = 1, scs->tpl_lad_mg = job->store_src_stats ? 1 : 0;
    scs->max_input_luma_width = (uint16_t)aw, scs->max_input_luma_height = (uint16_t)ah;
    svt_av1_build_quantizer(EB_EIGHT_BIT, 0, 0, 0, 0, 0, &enc->quants_8bit, &enc->deq_8bit);
    /* the job's quantiser scalars must be the reference's own tables at this qindex */
    if (enc->quants_8bit.y_round_fp[qindex][0] != job->round_fp[0] || enc->quants_8bit.y_round_fp[qindex][1] != job->round_fp[1] ||
        enc->quants_8bit.y_quant_fp[qindex][0] != job->quant_fp[0] || enc->quants_8bit.y_quant_fp[qindex][1] != job->quant_fp[1] ||
        enc->deq_8bit.y_dequant_qtx[qindex][0] != job->dequant[0] || enc->deq_8bit.y_dequant_qtx[qindex][1] != job->dequant[1])
        return -2;
    for (uint32_t i = 0; i < nb; i++) {
        geom[i].horizontal_index = (uint8_t)(i % bw64), geom[i].vertical_index = (uint8_t)(i / bw64);
        geom[i].org_x = (uint16_t)((i % bw64) * 64), geom[i].org_y = (uint16_t)((i / bw64) * 64);
        geom[i].width = (uint8_t)(aw - geom[i].org_x < 64 ? aw - geom[i].org_x : 64), geom[i].height = (uint8_t)(ah - geom[i].org_y < 64 ? ah - geom[i].org_y : 64);
    }
    cm->mi_rows = (int32_t)(ah >> 2), cm->mi_cols = (int32_t)(aw >> 2);
    pcs->scs = scs, pcs->av1_cm = cm, pcs->enhanced_pic = &src, pcs->pa_me_data = med;
    pcs->aligned_width = (uint16_t)aw, pcs->aligned_height = (uint16_t)ah;
    pcs->slice_type = job->i_slice ? I_SLICE : B_SLICE;
    pcs->enable_me_16x16 = job->enable_me_16x16;
    pcs->tpl_src_data_ready = job->src_data_ready;
    /* disable_intra_pred = disable_intra_pred_nref && temporal_layer_index == hierarchical_levels */
    pcs->temporal_layer_index = 3, pcs->hierarchical_levels = 3;
    TplControls *c = &pcs->tpl_ctrls;
    c->enable = 1, c->compute_rate = 0, c->disable_intra_pred_nref = job->disable_intra_pred, c->intra_mode_end = DC_PRED;
    c->pf_shape = (EB_TRANS_COEFF_SHAPE)job->pf_shape, c->use_sad_in_src_search = 1, c->dispenser_search_level = job->blk_size == 32 ? 1 : 0, c->subsample_tx = job->subsample_tx;
    c->synth_blk_size = job->synth_blk_size, c->subpel_depth = job->quarter_pel ? QUARTER_PEL : FULL_PEL, c->subpel_diag_refinement = 4; /* set_tpl_params levels 3 - 5 */
    scs->static_config.qp = 35, pcs->update_type = SVT_AV1_ARF_UPDATE; /* feed the (unused: MV_COST_NONE) rdmult of tpl_subpel_search */
    svt_av1_setup_scale_factors_for_frame(&scs->sf_identity, (int)W, (int)H, (int)W, (int)H);
    pcs->tpl_data.base_pcs = base, pcs->tpl_data.is_ref = job->is_ref;
    pcs->tpl_data.tpl_slice_type = job->tpl_i_slice ? I_SLICE : B_SLICE;
    enc->mc_flow_rec_picture_buffer[0] = &recon;
    enc->poc_map_idx[0]                = ~(uint64_t)0;
    int slot = 1;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            const SvtHipTplRef *f = &job->ref[l][r];
            pcs->tpl_data.ref_tpl_group_idx[l][r] = -1;
            if (!f->src)
                continue;
            plane_desc(&ref_src[l][r], f->src, f->src_stride, job->src.org_x, job->src.org_y, (uint16_t)W, (uint16_t)H);
            ref_src[l][r].max_width = f->max_width, ref_src[l][r].max_height = f->max_height;
            pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_ptr    = &ref_src[l][r];
            pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_number = f->picture_number;
            if (f->recon != f->src) {
                plane_desc(&ref_rec[l][r], f->recon, f->recon_stride, job->recon.org_x, job->recon.org_y, (uint16_t)W, (uint16_t)H);
                pcs->tpl_data.ref_in_slide_window[l][r] = 1;
                enc->poc_map_idx[slot] = f->picture_number, enc->mc_flow_rec_picture_buffer[slot] = &ref_rec[l][r];
                pcs->tpl_data.ref_tpl_group_idx[l][r] = slot;
                base->tpl_valid_pic[slot]             = f->usable;
                slot++;
            } else if (!f->usable) {
                pcs->tpl_data.ref_tpl_group_idx[l][r] = slot;
                base->tpl_valid_pic[slot++]           = 0;
            }
        }
    med->max_cand = job->max_cand, med->max_refs = job->max_refs, med->max_l0 = job->max_l0;
    med->me_results = calloc(nb, sizeof(MeSbResults *));
    for (uint32_t i = 0; i < nb; i++) {
        med->me_results[i]              = &res[i];
        res[i].me_mv_array              = (MvCandidate *)(job->me_mv_array + (size_t)i * job->stored_pus * job->max_refs);
        res[i].me_candidate_array       = (MeCandidate *)(job->me_candidate_array + (size_t)i * job->stored_pus * job->max_cand);
        res[i].total_me_candidate_index = (uint8_t *)job->total_me_candidate_index + (size_t)i * job->stored_pus;
    }
    const uint32_t a16 = (aw + 15) >> 4, rows16 = (ah + 15) >> 4;
    const uint32_t gstride = job->synth_blk_size == 32 ? (aw + 31) / 32 : (job->synth_blk_size == 16 ? a16 : a16 << 1);
    const uint32_t grows   = job->synth_blk_size == 32 ? (ah + 31) / 32 : (job->synth_blk_size == 16 ? rows16 : rows16 << 1);
    const size_t   n_stats = (size_t)gstride * grows;
    /* a 32x32 block half outside the picture stores cells past the last row of the grid (result_model_store has no bound):
     * room behind the array for them; they are not part of the result */
    const size_t   n_alloc = n_stats + (size_t)gstride * 4 + 8;
    TplSrcStats   *ss = calloc((size_t)a16 * rows16, sizeof(*ss));
    TplStats      *ts = calloc(n_alloc, sizeof(*ts));
    med->tpl_src_stats_buffer = ss;
    med->tpl_stats            = calloc(n_alloc, sizeof(TplStats *));
    for (size_t i = 0; i < n_alloc; i++) med->tpl_stats[i] = &ts[i];
    for (size_t i = 0; i < (size_t)a16 * rows16; i++) { /* in: previously computed source-based data (src_data_ready) */
        const SvtHipTplSrcStats *q = &job->src_stats[i];
        ss[i].srcrf_dist = q->srcrf_dist, ss[i].srcrf_rate = q->srcrf_rate, ss[i].ref_frame_poc = q->ref_frame_poc;
        ss[i].mv.row = q->mv_row, ss[i].mv.col = q->mv_col, ss[i].best_mode = q->best_mode, ss[i].best_rf_idx = q->best_rf_idx;
        ss[i].best_intra_mode = q->best_intra_mode;
    }
    for (uint32_t i = 0; i < nb; i++) /* as svt_aom_tpl_disp_kernel calls it (:2043-2051): an incomplete 64x64 block is always dispensed at level 0 */
        tpl_mc_flow_dispenser_sb_generic(enc, scs, pcs, 0, i, qindex, (geom[i].width == 64 && geom[i].height == 64 && job->blk_size == 32) ? 1 : 0);
    for (size_t i = 0; i < (size_t)a16 * rows16; i++) {
        SvtHipTplSrcStats *q = &job->src_stats[i];
        memset(q, 0, sizeof(*q));
        q->srcrf_dist = ss[i].srcrf_dist, q->srcrf_rate = ss[i].srcrf_rate, q->ref_frame_poc = ss[i].ref_frame_poc;
        q->mv_row = ss[i].mv.row, q->mv_col = ss[i].mv.col, q->best_mode = ss[i].best_mode, q->best_rf_idx = ss[i].best_rf_idx;
        q->best_intra_mode = (uint8_t)ss[i].best_intra_mode;
    }
    for (size_t i = 0; i < n_stats; i++) {
        SvtHipTplStats *q = &job->stats[i];
        memset(q, 0, sizeof(*q));
        q->srcrf_dist = ts[i].srcrf_dist, q->recrf_dist = ts[i].recrf_dist, q->srcrf_rate = ts[i].srcrf_rate, q->recrf_rate = ts[i].recrf_rate;
        q->mc_dep_rate = ts[i].mc_dep_rate, q->mc_dep_dist = ts[i].mc_dep_dist;
        q->mv_row = ts[i].mv.row, q->mv_col = ts[i].mv.col, q->ref_frame_poc = ts[i].ref_frame_poc;
    }
    free(ss), free(ts), free(med->tpl_stats), free(med->me_results), free(res), free(geom), free(cm), free(med), free(enc), free(base), free(pcs), free(scs);
    return 0;
}

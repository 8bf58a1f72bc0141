/*
 * oracle/ref_harness.c — TEST INFRASTRUCTURE.  Compiled ONLY into oracle/_ref/libsvtref.so (this
 * container, where /root/reference exists).  It is our own glue around the REAL reference functions:
 * it builds the pcs / scs / MeContext objects the reference expects from the flat structs of
 * include/svt_hip_me.h and calls the reference code unchanged.  No reference source is copied here.
 */
#include <stdlib.h>
#include <string.h>

#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"
#include "definitions.h"
#include "enc_mode_config.h"
#include "me_context.h"
#include "motion_estimation.h"
#include "pcs.h"
#include "pic_analysis_process.h"
#include "sequence_control_set.h"

#include "../include/svt_hip_me.h"

#define REF_API __attribute__((visibility("default")))

EbErrorType b64_geom_init_pcs(SequenceControlSet *scs, PictureParentControlSet *pcs);
void        svt_aom_gathering_picture_statistics(SequenceControlSet *scs, PictureParentControlSet *pcs,
                                                 EbPictureBufferDesc *input_padded_pic,
                                                 EbPictureBufferDesc *sixteenth_decimated_picture_ptr);
void svt_aom_downsample_filtering_input_picture(PictureParentControlSet *pcs, EbPictureBufferDesc *input_padded_pic,
                                                EbPictureBufferDesc *quarter_picture_ptr,
                                                EbPictureBufferDesc *sixteenth_picture_ptr);

static int g_init_done;

/* enc_handle.c:1475-1476: fill the function-pointer tables (C-only build: every pointer = *_c). */
REF_API void ref_init(void) {
    if (g_init_done)
        return;
    svt_aom_setup_common_rtcd_internal(0);
    svt_aom_setup_rtcd_internal(0);
    g_init_done = 1;
}

static void to_desc(EbPictureBufferDesc *d, const SvtHipPlane8 *p) {
    memset(d, 0, sizeof(*d));
    d->buffer_y = p->buf;
    d->stride_y = (uint16_t)p->stride;
    d->org_x    = p->org_x;
    d->org_y    = p->org_y;
    d->width    = p->width;
    d->height   = p->height;
    d->max_width = p->width, d->max_height = p->height;
    d->bit_depth = EB_EIGHT_BIT;
}

/* Preset -> ME parameters through the reference's own svt_aom_sig_deriv_me (enc_mode_config.c:684). */
REF_API int ref_derive_me_params(int enc_mode, int width, int height, int qp, int hierarchical_levels,
                                 int temporal_layer_index, int sc_class1, int frame_rate_q16, int enable_hme_l0,
                                 int enable_hme_l1, int enable_hme_l2, SvtHipMeParams *out) {
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    MeContext               *me  = calloc(1, sizeof(*me));
    if (!scs || !pcs || !me)
        return -1;
    pcs->scs                       = scs;
    pcs->enc_mode                  = (EncMode)enc_mode;
    pcs->sc_class1                 = (uint8_t)sc_class1;
    pcs->hierarchical_levels       = (uint8_t)hierarchical_levels;
    pcs->temporal_layer_index      = (uint8_t)temporal_layer_index;
    pcs->enable_hme_flag           = (enable_hme_l0 || enable_hme_l1 || enable_hme_l2);
    pcs->enable_hme_level0_flag    = (Bool)enable_hme_l0;
    pcs->enable_hme_level1_flag    = (Bool)enable_hme_l1;
    pcs->enable_hme_level2_flag    = (Bool)enable_hme_l2;
    scs->static_config.qp          = (uint32_t)qp;
    scs->static_config.pred_structure = SVT_AV1_PRED_RANDOM_ACCESS;
    scs->frame_rate                = (uint32_t)frame_rate_q16;
    svt_aom_derive_input_resolution(&scs->input_resolution, (uint32_t)(width * height));
    svt_aom_sig_deriv_me(scs, pcs, me);

    memset(out, 0, sizeof(*out));
    out->hme_search_method      = me->hme_search_method == FULL_SAD_SEARCH;
    out->me_search_method       = me->me_search_method == FULL_SAD_SEARCH;
    out->enable_hme_flag        = me->enable_hme_flag;
    out->enable_hme_level0_flag = me->enable_hme_level0_flag;
    out->enable_hme_level1_flag = me->enable_hme_level1_flag;
    out->enable_hme_level2_flag = me->enable_hme_level2_flag;
    out->num_hme_sa_w           = (uint8_t)me->num_hme_sa_w;
    out->num_hme_sa_h           = (uint8_t)me->num_hme_sa_h;
#define SA(dst, srcv) (dst).width = (srcv).width, (dst).height = (srcv).height
    SA(out->hme_l0_sa_min, me->hme_l0_sa.sa_min);
    SA(out->hme_l0_sa_max, me->hme_l0_sa.sa_max);
    SA(out->hme_l1_sa, me->hme_l1_sa);
    SA(out->hme_l2_sa, me->hme_l2_sa);
    SA(out->me_sa_min, me->me_sa.sa_min);
    SA(out->me_sa_max, me->me_sa.sa_max);
    out->prehme_enable           = me->prehme_ctrl.enable;
    out->prehme_skip_search_line = me->prehme_ctrl.skip_search_line;
    out->prehme_l1_early_exit    = me->prehme_ctrl.l1_early_exit;
    for (int i = 0; i < 2; i++) {
        SA(out->prehme_sa_min[i], me->prehme_ctrl.prehme_sa_cfg[i].sa_min);
        SA(out->prehme_sa_max[i], me->prehme_ctrl.prehme_sa_cfg[i].sa_max);
    }
    out->enable_me_hme_ref_pruning = me->me_hme_prune_ctrls.enable_me_hme_ref_pruning;
    out->prune_ref_if_hme_sad_dev_bigger_than_th = me->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th;
    out->prune_ref_if_me_sad_dev_bigger_than_th  = me->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th;
    out->zz_sad_th    = me->me_hme_prune_ctrls.zz_sad_th;
    out->zz_sad_pct   = me->me_hme_prune_ctrls.zz_sad_pct;
    out->phme_sad_th  = me->me_hme_prune_ctrls.phme_sad_th;
    out->phme_sad_pct = me->me_hme_prune_ctrls.phme_sad_pct;
    out->enable_me_sr_adjustment              = me->me_sr_adjustment_ctrls.enable_me_sr_adjustment;
    out->distance_based_hme_resizing          = me->me_sr_adjustment_ctrls.distance_based_hme_resizing;
    out->reduce_me_sr_based_on_mv_length_th   = me->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th;
    out->stationary_hme_sad_abs_th            = me->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th;
    out->stationary_me_sr_divisor             = me->me_sr_adjustment_ctrls.stationary_me_sr_divisor;
    out->reduce_me_sr_based_on_hme_sad_abs_th = me->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th;
    out->me_sr_divisor_for_low_hme_sad        = me->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad;
    out->me_8x8_var_enabled = me->me_8x8_var_ctrls.enabled;
    out->me_sr_div4_th      = me->me_8x8_var_ctrls.me_sr_div4_th;
    out->me_sr_div2_th      = me->me_8x8_var_ctrls.me_sr_div2_th;
    out->me_sr_mult2_th     = me->me_8x8_var_ctrls.me_sr_mult2_th;
    out->mv_sa_adj_enabled  = me->mv_based_sa_adj.enabled;
    out->mv_sa_adj_nearest_ref_only = me->mv_based_sa_adj.nearest_ref_only;
    out->mv_sa_adj_mv_size_th       = me->mv_based_sa_adj.mv_size_th;
    out->mv_sa_adj_sa_multiplier    = me->mv_based_sa_adj.sa_multiplier;
    out->reduce_hme_l0_sr_th_min    = me->reduce_hme_l0_sr_th_min;
    out->reduce_hme_l0_sr_th_max    = me->reduce_hme_l0_sr_th_max;
    out->me_early_exit_th           = me->me_early_exit_th;
    out->me_safe_limit_zz_th        = me->me_safe_limit_zz_th;
    out->prev_me_stage_based_exit_th = me->prev_me_stage_based_exit_th;
    out->prune_me_candidates_th     = me->prune_me_candidates_th;
    out->use_best_unipred_cand_only = me->use_best_unipred_cand_only;
    out->enable_me_8x8   = svt_aom_get_enable_me_8x8((EncMode)enc_mode, false, scs->input_resolution);
    out->enable_me_16x16 = svt_aom_get_enable_me_16x16((EncMode)enc_mode);
    out->max_number_of_pus_per_sb = SQUARE_PU_COUNT;
    out->input_resolution_le_480p = scs->input_resolution <= INPUT_SIZE_480p_RANGE;
    out->temporal_layer_index     = (uint8_t)temporal_layer_index;
    out->hierarchical_levels      = (uint8_t)hierarchical_levels;
    free(me), free(pcs), free(scs);
    return 0;
}

void ref_params_to_ctx(MeContext *me, const SvtHipMeParams *p);
static void params_to_ctx(MeContext *me, const SvtHipMeParams *p) {
    me->hme_search_method      = p->hme_search_method ? FULL_SAD_SEARCH : SUB_SAD_SEARCH;
    me->me_search_method       = p->me_search_method ? FULL_SAD_SEARCH : SUB_SAD_SEARCH;
    me->enable_hme_flag        = p->enable_hme_flag;
    me->enable_hme_level0_flag = p->enable_hme_level0_flag;
    me->enable_hme_level1_flag = p->enable_hme_level1_flag;
    me->enable_hme_level2_flag = p->enable_hme_level2_flag;
    me->num_hme_sa_w           = p->num_hme_sa_w;
    me->num_hme_sa_h           = p->num_hme_sa_h;
    SA(me->hme_l0_sa.sa_min, p->hme_l0_sa_min);
    SA(me->hme_l0_sa.sa_max, p->hme_l0_sa_max);
    SA(me->hme_l1_sa, p->hme_l1_sa);
    SA(me->hme_l2_sa, p->hme_l2_sa);
    SA(me->me_sa.sa_min, p->me_sa_min);
    SA(me->me_sa.sa_max, p->me_sa_max);
    me->prehme_ctrl.enable           = p->prehme_enable;
    me->prehme_ctrl.skip_search_line = p->prehme_skip_search_line;
    me->prehme_ctrl.l1_early_exit    = p->prehme_l1_early_exit;
    for (int i = 0; i < 2; i++) {
        SA(me->prehme_ctrl.prehme_sa_cfg[i].sa_min, p->prehme_sa_min[i]);
        SA(me->prehme_ctrl.prehme_sa_cfg[i].sa_max, p->prehme_sa_max[i]);
    }
    me->me_hme_prune_ctrls.enable_me_hme_ref_pruning               = p->enable_me_hme_ref_pruning;
    me->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th = p->prune_ref_if_hme_sad_dev_bigger_than_th;
    me->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th  = p->prune_ref_if_me_sad_dev_bigger_than_th;
    me->me_hme_prune_ctrls.zz_sad_th    = p->zz_sad_th;
    me->me_hme_prune_ctrls.zz_sad_pct   = p->zz_sad_pct;
    me->me_hme_prune_ctrls.phme_sad_th  = p->phme_sad_th;
    me->me_hme_prune_ctrls.phme_sad_pct = p->phme_sad_pct;
    me->me_sr_adjustment_ctrls.enable_me_sr_adjustment              = p->enable_me_sr_adjustment;
    me->me_sr_adjustment_ctrls.distance_based_hme_resizing          = p->distance_based_hme_resizing;
    me->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th   = p->reduce_me_sr_based_on_mv_length_th;
    me->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th            = p->stationary_hme_sad_abs_th;
    me->me_sr_adjustment_ctrls.stationary_me_sr_divisor             = p->stationary_me_sr_divisor;
    me->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th = p->reduce_me_sr_based_on_hme_sad_abs_th;
    me->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad        = p->me_sr_divisor_for_low_hme_sad;
    me->me_8x8_var_ctrls.enabled        = p->me_8x8_var_enabled;
    me->me_8x8_var_ctrls.me_sr_div4_th  = p->me_sr_div4_th;
    me->me_8x8_var_ctrls.me_sr_div2_th  = p->me_sr_div2_th;
    me->me_8x8_var_ctrls.me_sr_mult2_th = p->me_sr_mult2_th;
    me->mv_based_sa_adj.enabled          = p->mv_sa_adj_enabled;
    me->mv_based_sa_adj.nearest_ref_only = p->mv_sa_adj_nearest_ref_only;
    me->mv_based_sa_adj.mv_size_th       = p->mv_sa_adj_mv_size_th;
    me->mv_based_sa_adj.sa_multiplier    = p->mv_sa_adj_sa_multiplier;
    me->reduce_hme_l0_sr_th_min          = p->reduce_hme_l0_sr_th_min;
    me->reduce_hme_l0_sr_th_max          = p->reduce_hme_l0_sr_th_max;
    me->me_early_exit_th                 = p->me_early_exit_th;
    me->me_safe_limit_zz_th              = p->me_safe_limit_zz_th;
    me->prev_me_stage_based_exit_th      = p->prev_me_stage_based_exit_th;
    me->prune_me_candidates_th           = p->prune_me_candidates_th;
    me->use_best_unipred_cand_only       = p->use_best_unipred_cand_only;
    me->me_type                          = p->me_mctf ? ME_MCTF : ME_OPEN_LOOP;
    me->tf_me_exit_th                    = p->tf_me_exit_th;
    me->num_of_list_to_search            = p->num_of_list_to_search;
    me->num_of_ref_pic_to_search[0]      = p->num_of_ref_pic_to_search[0];
    me->num_of_ref_pic_to_search[1]      = p->num_of_ref_pic_to_search[1];
    me->temporal_layer_index             = p->temporal_layer_index;
    me->is_ref                           = p->is_ref;
}

/* shared with ref_harness_tfme.c */
void ref_params_to_ctx(MeContext *me, const SvtHipMeParams *p) { params_to_ctx(me, p); }
void ref_b64_geom_init_pcs(SequenceControlSet *scs, PictureParentControlSet *pcs) { b64_geom_init_pcs(scs, pcs); }

/* The b64 loop of me_process.c:174-290 around the real svt_aom_motion_estimation_b64.  Every pointer in
 * `job` is a host pointer.  Outputs follow include/svt_hip_me.h (SvtHipMeFrameOut). */
REF_API int ref_me_frame(const SvtHipMeFrameJob *job, uint32_t first_b64, uint32_t count) {
    ref_init();
    const SvtHipMeParams    *p   = &job->prm;
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    MeContext               *me  = calloc(1, sizeof(*me));
    MotionEstimationData    *med = calloc(1, sizeof(*med));
    if (!scs || !pcs || !me || !med)
        return -1;
    EbPictureBufferDesc src[3], ref[2][4][3];
    to_desc(&src[0], &job->src.full), to_desc(&src[1], &job->src.quarter), to_desc(&src[2], &job->src.sixteenth);
    scs->b64_size                  = 64;
    scs->mrp_ctrls.only_l_bwd      = p->only_l_bwd;
    scs->input_resolution          = p->input_resolution_le_480p ? INPUT_SIZE_480p_RANGE : INPUT_SIZE_1080p_RANGE;
    pcs->scs                       = scs;
    pcs->aligned_width             = (uint16_t)((src[0].width + 7) & ~7);
    pcs->aligned_height            = (uint16_t)((src[0].height + 7) & ~7);
    pcs->picture_number            = p->picture_number;
    pcs->hierarchical_levels       = p->hierarchical_levels;
    pcs->temporal_layer_index      = p->temporal_layer_index;
    pcs->similar_brightness_refs   = p->similar_brightness_refs;
    pcs->enable_me_8x8             = p->enable_me_8x8;
    pcs->enable_me_16x16           = p->enable_me_16x16;
    pcs->max_number_of_pus_per_sb  = p->max_number_of_pus_per_sb;
    pcs->gm_ctrls.enabled          = 0;
    b64_geom_init_pcs(scs, pcs);
    const uint32_t bw = (pcs->aligned_width + 63) / 64, bh = (pcs->aligned_height + 63) / 64, nb = bw * bh;
    const uint32_t stored = svt_hip_me_stored_pus(p);
    pcs->b64_total_count  = (uint16_t)nb;
    pcs->pa_me_data       = med;
    med->max_cand = p->max_cand, med->max_refs = p->max_refs, med->max_l0 = p->max_l0;
    med->me_results = calloc(nb, sizeof(MeSbResults *));
    MeSbResults *res = calloc(nb, sizeof(MeSbResults));
    for (uint32_t i = 0; i < nb; i++) {
        med->me_results[i]               = &res[i];
        res[i].me_mv_array               = (MvCandidate *)(job->out.me_mv_array + (size_t)i * stored * p->max_refs);
        res[i].me_candidate_array        = (MeCandidate *)(job->out.me_candidate_array + (size_t)i * stored * p->max_cand);
        res[i].total_me_candidate_index  = job->out.total_me_candidate_index + (size_t)i * stored;
    }
    pcs->me_64x64_distortion         = job->out.me_64x64_distortion;
    pcs->me_32x32_distortion         = job->out.me_32x32_distortion;
    pcs->me_16x16_distortion         = job->out.me_16x16_distortion;
    pcs->me_8x8_distortion           = job->out.me_8x8_distortion;
    pcs->me_8x8_cost_variance        = job->out.me_8x8_cost_variance;
    pcs->rc_me_distortion            = job->out.rc_me_distortion;
    pcs->stationary_block_present_sb = calloc(nb, 1);
    pcs->rc_me_allow_gm              = calloc(nb, 1);

    params_to_ctx(me, p);
    for (int l = 0; l < p->num_of_list_to_search; l++)
        for (int r = 0; r < p->num_of_ref_pic_to_search[l]; r++) {
            to_desc(&ref[l][r][0], &job->ref[l][r].full);
            to_desc(&ref[l][r][1], &job->ref[l][r].quarter);
            to_desc(&ref[l][r][2], &job->ref[l][r].sixteenth);
            me->me_ds_ref_array[l][r].picture_ptr           = &ref[l][r][0];
            me->me_ds_ref_array[l][r].quarter_picture_ptr   = &ref[l][r][1];
            me->me_ds_ref_array[l][r].sixteenth_picture_ptr = &ref[l][r][2];
            me->me_ds_ref_array[l][r].picture_number        = p->ref_picture_number[l][r];
        }
    for (uint32_t i = first_b64; i < first_b64 + count && i < nb; i++) {
        const uint32_t ox = (i % bw) * 64, oy = (i / bw) * 64;
        /* me_process.c:183-214 */
        me->b64_src_ptr    = &src[0].buffer_y[(src[0].org_y + oy) * src[0].stride_y + src[0].org_x + ox];
        me->b64_src_stride = src[0].stride_y;
        me->quarter_b64_buffer =
            &src[1].buffer_y[(src[1].org_y + (oy >> 1)) * src[1].stride_y + src[1].org_x + (ox >> 1)];
        me->quarter_b64_buffer_stride = src[1].stride_y;
        me->sixteenth_b64_buffer =
            &src[2].buffer_y[(src[2].org_y + (oy >> 2)) * src[2].stride_y + src[2].org_x + (ox >> 2)];
        me->sixteenth_b64_buffer_stride = src[2].stride_y;
        /* the reference leaves p_sb_best_sad of unsearched references stale; define it as zero */
        memset(me->p_sb_best_sad, 0, sizeof(me->p_sb_best_sad));
        svt_aom_motion_estimation_b64(pcs, i, ox, oy, me, &src[0]);
        memcpy(job->out.best_sad + (size_t)i * 2 * 4 * 85, me->p_sb_best_sad, sizeof(me->p_sb_best_sad));
        memcpy(job->out.best_mv + (size_t)i * 2 * 4 * 85, me->p_sb_best_mv, sizeof(me->p_sb_best_mv));
        for (int l = 0; l < 2; l++)
            for (int r = 0; r < 4; r++) {
                SvtHipMeSearchResult *o = job->out.search_results + ((size_t)i * 2 + l) * 4 + r;
                memset(o, 0, sizeof(*o));
                o->hme_sad  = me->search_results[l][r].hme_sad;
                o->hme_sc_x = me->search_results[l][r].hme_sc_x;
                o->hme_sc_y = me->search_results[l][r].hme_sc_y;
                o->do_ref   = me->search_results[l][r].do_ref;
            }
    }
    free(pcs->stationary_block_present_sb), free(pcs->rc_me_allow_gm);
    free(pcs->b64_geom), free(res), free(med->me_results), free(med), free(me), free(pcs), free(scs);
    return 0;
}

/* Pyramid + variance through the reference drivers (pic_analysis_process.c:1922-1979, 1533-1553). */
REF_API int ref_pyramid_frame(const SvtHipPlane8 *full, const SvtHipPlane8 *quarter, const SvtHipPlane8 *sixteenth,
                              int hme_level1_enabled) {
    ref_init();
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    EbPictureBufferDesc      d[3];
    to_desc(&d[0], full), to_desc(&d[1], quarter), to_desc(&d[2], sixteenth);
    pcs->enable_hme_flag        = 1;
    pcs->enable_hme_level0_flag = 1;
    pcs->enable_hme_level1_flag = (Bool)hme_level1_enabled;
    svt_aom_downsample_filtering_input_picture(pcs, &d[0], &d[1], &d[2]);
    free(pcs);
    return 0;
}

REF_API int ref_variance_frame(const SvtHipPlane8 *full, uint16_t *variance, int full_precision) {
    ref_init();
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    EbPictureBufferDesc      d;
    to_desc(&d, full);
    scs->b64_size                                  = 64;
    scs->calculate_variance                        = 1;
    scs->calc_hist                                 = 0;
    scs->block_mean_calc_prec                      = full_precision ? BLOCK_MEAN_PREC_FULL : BLOCK_MEAN_PREC_SUB;
    scs->static_config.enable_adaptive_quantization = 1;
    pcs->scs                                       = scs;
    pcs->aligned_width                             = (uint16_t)((d.width + 7) & ~7);
    pcs->aligned_height                            = (uint16_t)((d.height + 7) & ~7);
    b64_geom_init_pcs(scs, pcs);
    const uint32_t nb    = ((pcs->aligned_width + 63) / 64) * ((pcs->aligned_height + 63) / 64);
    pcs->b64_total_count = (uint16_t)nb;
    pcs->variance        = calloc(nb, sizeof(uint16_t *));
    for (uint32_t i = 0; i < nb; i++) pcs->variance[i] = variance + (size_t)85 * i;
    svt_aom_gathering_picture_statistics(scs, pcs, &d, NULL);
    free(pcs->variance), free(pcs->b64_geom), free(pcs), free(scs);
    return 0;
}

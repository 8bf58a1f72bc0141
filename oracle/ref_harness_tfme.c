/*
 * oracle/ref_harness_tfme.c — TEST INFRASTRUCTURE.  Compiled ONLY into oracle/_ref/libsvtref.so.
 * produce_temporally_filtered_pic (temporal_filtering.c:2752-3308) and the sub-pel searches / predictions / split decision it
 * drives are `static` in the reference.  To pin the oracle's restatement of that block loop against the REAL code, this
 * translation unit compiles the reference's temporal_filtering.c in place (by #include from where it lies, nothing is
 * copied) and exposes one entry point that builds the control structures around flat test inputs (SvtHipTfPictureJob of
 * include/svt_hip_tf.h, with host pointers) and runs the real function over the whole picture; oracle/Makefile makes
 * every other symbol of this object local so that it does not clash with temporal_filtering.o of the reference archive.
 */
#include "temporal_filtering.c"

#include "../include/svt_hip_tf.h"

void ref_init(void);
void init_fn_ptr(void);
void svt_aom_asm_set_convolve_asm_table(void);
void svt_aom_asm_set_convolve_hbd_asm_table(void);
void ref_params_to_ctx(MeContext *me, const SvtHipMeParams *p);
void ref_b64_geom_init_pcs(SequenceControlSet *scs, PictureParentControlSet *pcs);

typedef struct TfPicBuild {
    PictureParentControlSet pcs;
    EbPaReferenceObject     pa;
    EbObjectWrapper         wrap;
    EbPictureBufferDesc     full, quarter, sixteenth, enhanced;
} TfPicBuild;

static void plane_desc(EbPictureBufferDesc *d, const SvtHipPlane8 *p) {
    memset(d, 0, sizeof(*d));
    d->buffer_y = p->buf, d->stride_y = (uint16_t)p->stride;
    d->org_x = p->org_x, d->org_y = p->org_y, d->width = p->width, d->height = p->height;
    d->max_width = p->width, d->max_height = p->height, d->bit_depth = EB_EIGHT_BIT;
}

static void build_pic(TfPicBuild *b, const SvtHipTfPic *pic, SequenceControlSet *scs, const SvtHipTfPictureJob *job, Av1Common *cm) {
    memset(b, 0, sizeof(*b));
    plane_desc(&b->full, &pic->pyr.full), plane_desc(&b->quarter, &pic->pyr.quarter), plane_desc(&b->sixteenth, &pic->pyr.sixteenth);
    b->enhanced            = b->full;
    b->enhanced.buffer_cb  = pic->chroma8[0], b->enhanced.buffer_cr = pic->chroma8[1];
    b->enhanced.stride_cb  = b->enhanced.stride_cr = (uint16_t)pic->chroma8_stride;
    b->enhanced.stride_bit_inc_cb = b->enhanced.stride_bit_inc_cr = (uint16_t)pic->chroma8_stride;
    b->enhanced.stride_bit_inc_y  = b->enhanced.stride_y;
    b->pa.input_padded_pic                  = &b->full;
    b->pa.quarter_downsampled_picture_ptr   = &b->quarter;
    b->pa.sixteenth_downsampled_picture_ptr = &b->sixteenth;
    b->wrap.object_ptr                      = &b->pa;
    PictureParentControlSet *pcs = &b->pcs;
    pcs->scs                 = scs;
    pcs->pa_ref_pic_wrapper  = &b->wrap;
    pcs->enhanced_pic        = &b->enhanced;
    pcs->aligned_width       = (uint16_t)((b->full.width + 7) & ~7);
    pcs->aligned_height      = (uint16_t)((b->full.height + 7) & ~7);
    pcs->av1_cm              = cm;
    pcs->hierarchical_levels = job->me.hierarchical_levels;
    pcs->temporal_layer_index = job->me.temporal_layer_index;
    pcs->is_ref              = job->me.is_ref;
    pcs->slice_type          = B_SLICE;
    pcs->tf_segments_column_count = pcs->tf_segments_row_count = 1;
    pcs->similar_brightness_refs  = job->me.similar_brightness_refs;
    pcs->enable_me_8x8 = job->me.enable_me_8x8, pcs->enable_me_16x16 = job->me.enable_me_16x16;
    pcs->max_number_of_pus_per_sb = job->me.max_number_of_pus_per_sb;
    for (int p = 0; p < 3; p++) pcs->altref_buffer_highbd[p] = pic->hbd[p];
    TfControls *c = &pcs->tf_ctrls;
    c->enabled = 1;
    c->half_pel_mode = job->ctrls.half_pel_mode, c->quarter_pel_mode = job->ctrls.quarter_pel_mode, c->eight_pel_mode = job->ctrls.eight_pel_mode;
    c->use_2tap = job->ctrls.use_2tap, c->sub_sampling_shift = job->ctrls.sub_sampling_shift;
    c->use_pred_64x64_only_th = job->ctrls.use_pred_64x64_only_th, c->subpel_early_exit_th = job->ctrls.subpel_early_exit_th;
    c->use_8bit_subpel = job->ctrls.use_8bit_subpel, c->use_zz_based_filter = job->ctrls.use_zz_based_filter;
    c->enable_8x8_pred = job->ctrls.enable_8x8_pred, c->pred_error_32x32_th = job->ctrls.pred_error_32x32_th;
    c->me_exit_th = job->me.tf_me_exit_th, c->ref_frame_factor = 1, c->chroma_lvl = job->chroma ? 1 : 0;
}

/* The real produce_temporally_filtered_pic over one picture (one segment).  noise_log1p_fp16 / qp feed the reference's own
 * decay-factor computation; the factors it derived are returned in decay_out so that the restatement and the GPU path can be
 * given the same numbers.  tot[2] = tf_tot_horz_blks, tf_tot_vert_blks. */
__attribute__((visibility("default"))) int ref_tf_picture(const SvtHipTfPictureJob *job, const int32_t *noise_log1p_fp16, int qp,
                                                           uint32_t *decay_out, uint32_t *tot) {
    ref_init();
    static int tables_done;
    if (!tables_done) { /* the one-time table set-up of svt_av1_enc_init (enc_handle.c:1478-1491) */
        svt_aom_asm_set_convolve_asm_table();
        svt_aom_asm_set_convolve_hbd_asm_table();
        svt_aom_build_blk_geom(GEOM_0);
        init_fn_ptr();
        tables_done = 1;
    }
    const int n = (int)job->n_refs + 1;
    if (n > ALTREF_MAX_NFRAMES)
        return -1;
    SequenceControlSet *scs = calloc(1, sizeof(*scs));
    TfPicBuild         *pb  = calloc((size_t)n, sizeof(*pb));
    MeContext          *me  = calloc(1, sizeof(*me));
    Av1Common          *cm  = calloc(1, sizeof(*cm));
    if (!scs || !pb || !me || !cm)
        return -2;
    scs->b64_size = 64, scs->super_block_size = 64;
    scs->seq_header.sb_size = BLOCK_64X64;
    scs->mrp_ctrls.only_l_bwd = job->me.only_l_bwd;
    scs->input_resolution     = job->me.input_resolution_le_480p ? INPUT_SIZE_480p_RANGE : INPUT_SIZE_1080p_RANGE;
    scs->subsampling_x = scs->subsampling_y = 1;
    scs->static_config.encoder_bit_depth = job->bit_depth;
    scs->static_config.qp                = (uint32_t)qp;
    scs->picture_analysis_number_of_regions_per_width = scs->picture_analysis_number_of_regions_per_height = 4;
    cm->mi_rows = (int32_t)job->mi_rows, cm->mi_cols = (int32_t)job->mi_cols;
    svt_av1_setup_scale_factors_for_frame(&scs->sf_identity, job->centre.pyr.full.width, job->centre.pyr.full.height, job->centre.pyr.full.width,
                                          job->centre.pyr.full.height);
    PictureParentControlSet *list[ALTREF_MAX_NFRAMES];
    EbPictureBufferDesc     *pics[ALTREF_MAX_NFRAMES];
    build_pic(&pb[0], &job->centre, scs, job, cm);
    for (int i = 1; i < n; i++) build_pic(&pb[i], &job->ref[i - 1], scs, job, cm);
    for (int i = 0; i < n; i++) list[i] = &pb[i].pcs, pics[i] = &pb[i].enhanced;
    PictureParentControlSet *centre = list[0];
    centre->past_altref_nframes = 0, centre->future_altref_nframes = (uint8_t)(n - 1);
    centre->picture_number      = job->centre.picture_number;
    ref_b64_geom_init_pcs(scs, centre);
    centre->b64_total_count = (uint16_t)(((centre->aligned_width + 63) / 64) * ((centre->aligned_height + 63) / 64));
    centre->stationary_block_present_sb = calloc(centre->b64_total_count, 1);
    centre->rc_me_allow_gm              = calloc(centre->b64_total_count, 1);
    for (int i = 1; i < n; i++) pb[i].pa.picture_number = job->ref[i - 1].picture_number;

    ref_params_to_ctx(me, &job->me);
    me->hme_l0_sa_default_tf = me->hme_l0_sa;
    me->tf_ctrls             = centre->tf_ctrls;
    me->tf_chroma            = job->chroma;
    me->tf_mv_dist_th        = job->mv_dist_th;
    MotionEstimationContext_t mectx;
    memset(&mectx, 0, sizeof(mectx));
    mectx.me_ctx = me;

    /* svt_av1_init_temporal_filtering picks the variant by the prediction structure (temporal_filtering.c:4046-4062) */
    const EbErrorType rc = job->ctrls.low_delay ? produce_temporally_filtered_pic_ld(list, pics, 0, &mectx, noise_log1p_fp16, 0, job->bit_depth > 8)
                                                : produce_temporally_filtered_pic(list, pics, 0, &mectx, noise_log1p_fp16, 0, job->bit_depth > 8);
    for (int p = 0; p < 3; p++) decay_out[p] = me->tf_decay_factor_fp16[p];
    tot[0] = me->tf_tot_horz_blks, tot[1] = me->tf_tot_vert_blks;
    free(centre->stationary_block_present_sb), free(centre->rc_me_allow_gm), free(centre->b64_geom);
    free(cm), free(me), free(pb), free(scs);
    return rc == EB_ErrorNone ? 0 : -3;
}

/*
 * oracle/ref_harness_tf.c — TEST INFRASTRUCTURE.  Compiled ONLY into oracle/_ref/libsvtref.so.  Our own glue around
 * the REAL temporal-filter functions of the reference: it fills a real MeContext from the flat SvtHipTfBlock of
 * include/svt_hip_tf.h (this is the field map of INTEGRATION.md, run on the real structure) and calls the reference's
 * RTCD pointers unchanged.  No reference source is copied here.
 */
#include <stdlib.h>
#include <string.h>

#include "aom_dsp_rtcd.h"
#include "definitions.h"
#include "me_context.h"

#include "../include/svt_hip_tf.h"

#define REF_API __attribute__((visibility("default")))
void ref_init(void);

static MeContext *ctx_of(const SvtHipTfBlock *b) {
    MeContext *me = (MeContext *)calloc(1, sizeof(MeContext));
    if (!me)
        return NULL;
    me->tf_block_col = 0, me->tf_block_row = 0; /* idx_32x32 = 0 */
    me->tf_32x32_block_split_flag[0] = b->split;
    for (int i = 0; i < 4; i++) {
        me->tf_16x16_mv_x[i] = b->mv_x[i], me->tf_16x16_mv_y[i] = b->mv_y[i];
        me->tf_16x16_block_error[i] = b->block_error[i];
    }
    me->tf_32x32_mv_x[0] = b->mv_x[0], me->tf_32x32_mv_y[0] = b->mv_y[0];
    me->tf_32x32_block_error[0] = b->block_error[0];
    for (int p = 0; p < 3; p++) me->tf_decay_factor_fp16[p] = b->decay_factor_fp16[p];
    me->tf_chroma     = b->chroma;
    me->tf_mv_dist_th = b->mv_dist_th;
    return me;
}

/* svt_av1_apply_temporal_filter_planewise_medium[_hbd] on one block; host pointers in the struct */
REF_API int ref_tf_block_accumulate(const SvtHipTfBlock *b) {
    ref_init();
    if (b->src_stride[1] != b->src_stride[2] || b->pred_stride[1] != b->pred_stride[2])
        return -1;
    MeContext *me = ctx_of(b);
    if (!me)
        return -2;
    if (b->zz_based && b->is_16bit)
        svt_av1_apply_zz_based_temporal_filter_planewise_medium_hbd(me, b->pred[0], (int)b->pred_stride[0], b->pred[1], b->pred[2],
                                                                    (int)b->pred_stride[1], 32, 32, b->ss_x, b->ss_y, b->accum[0], b->count[0],
                                                                    b->accum[1], b->count[1], b->accum[2], b->count[2], b->bit_depth);
    else if (b->zz_based)
        svt_av1_apply_zz_based_temporal_filter_planewise_medium(me, b->pred[0], (int)b->pred_stride[0], b->pred[1], b->pred[2],
                                                                (int)b->pred_stride[1], 32, 32, b->ss_x, b->ss_y, b->accum[0], b->count[0],
                                                                b->accum[1], b->count[1], b->accum[2], b->count[2]);
    else if (b->is_16bit)
        svt_av1_apply_temporal_filter_planewise_medium_hbd(me, b->src[0], (int)b->src_stride[0], b->pred[0], (int)b->pred_stride[0], b->src[1],
                                                           b->src[2], (int)b->src_stride[1], b->pred[1], b->pred[2], (int)b->pred_stride[1], 32,
                                                           32, b->ss_x, b->ss_y, b->accum[0], b->count[0], b->accum[1], b->count[1],
                                                           b->accum[2], b->count[2], b->bit_depth);
    else
        svt_av1_apply_temporal_filter_planewise_medium(me, b->src[0], (int)b->src_stride[0], b->pred[0], (int)b->pred_stride[0], b->src[1],
                                                       b->src[2], (int)b->src_stride[1], b->pred[1], b->pred[2], (int)b->pred_stride[1], 32, 32,
                                                       b->ss_x, b->ss_y, b->accum[0], b->count[0], b->accum[1], b->count[1], b->accum[2],
                                                       b->count[2]);
    free(me);
    return 0;
}

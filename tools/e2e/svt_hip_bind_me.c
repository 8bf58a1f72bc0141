/*
 * svt_hip_bind_me.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 2b of INTEGRATION.md):
 * open-loop motion estimation of a whole picture through the BATCHED entry point svt_hip_me_frames instead of one
 * svt_aom_motion_estimation_b64 call per 64x64 block (Source/Lib/Codec/me_process.c:174-290).
 *
 * The patch puts ONE call in front of the reference's per-block call:
 *     if (svt_hip_bind_me_b64(pcs, b64_index, me_ctx, input_padded_pic, quarter_picture_ptr, sixteenth_picture_ptr)) <their call>;
 * The first block of a picture that arrives here (from whichever ME thread / segment) runs the whole picture on the GPU:
 * MeContext -> SvtHipMeParams (the field map of INTEGRATION.md), the source and reference pyramids taken from the device-resident
 * picture mirrors (svt_hip_bind_dev.h: a pyramid crosses PCIe once, however many pictures reference it), one launch, results
 * downloaded; every block — that one and all later ones, on any thread — then only copies its own results into the picture's
 * MeSbResults / distortion arrays.  Threads that arrive while the picture is being computed wait on a condition variable.
 * Active with `--asm hip` and SVTAV1_HIP_TIERB_ME=1 for pictures whose ME needs nothing but what the batched call produces
 * (global motion off: presets >= M3; no super-res / resize); everything else returns 1 and the reference's own call runs.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "definitions.h"
#include "me_context.h"
#include "pcs.h"
#include "sequence_control_set.h"

#include "svt_hip.h"
#include "svt_hip_me.h"
#include "svt_hip_bind.h"
#include "svt_hip_bind_dev.h"

static int32_t (*p_me_frames)(const SvtHipMeFrameJob *, uint32_t, void *);
static int           g_active;
static unsigned long g_pictures, g_blocks; /* statistics, printed at exit */

static void report(void) {
    fprintf(stderr, "svt_hip_bind_me: %lu pictures / %lu blocks of open-loop ME through svt_hip_me_frames\n", g_pictures, g_blocks);
}

void svt_hip_bind_me_setup(void *(*sym)(const char *)) {
    p_me_frames = (int32_t(*)(const SvtHipMeFrameJob *, uint32_t, void *))sym("svt_hip_me_frames");
    g_active    = hd_env_on("SVTAV1_HIP_TIERB_ME") && g_hd.ok && p_me_frames;
    if (g_active)
        atexit(report);
}

/* ---- one picture in flight: what the first block downloads and every block copies its share of ----------------------- */
typedef struct PicResults {
    uint32_t  stored, max_refs, max_cand;
    uint32_t *mv;      /* [nb][stored * max_refs] */
    uint8_t  *cand;    /* [nb][stored * max_cand] */
    uint8_t  *cnt;     /* [nb][stored] */
    uint32_t *dist[6]; /* 64, 32, 16, 8, 8x8 cost variance, rc */
} PicResults;
static HdOnceTable g_tab;

static void results_free(void *p) {
    PicResults *e = (PicResults *)p;
    free(e->mv), free(e->cand), free(e->cnt);
    for (int i = 0; i < 6; i++) free(e->dist[i]);
    free(e);
}

#define SA(dst, srcv) (dst).width = (uint16_t)(srcv).width, (dst).height = (uint16_t)(srcv).height
void svt_hip_bind_me_params(SvtHipMeParams *out, const PictureParentControlSet *pcs, const MeContext *me); /* also used by svt_hip_bind_tf.c */
void svt_hip_bind_me_params(SvtHipMeParams *out, const PictureParentControlSet *pcs, const MeContext *me) {
    const SequenceControlSet *scs = pcs->scs;
    memset(out, 0, sizeof(*out));
    out->hme_search_method      = me->hme_search_method == FULL_SAD_SEARCH;
    out->me_search_method       = me->me_search_method == FULL_SAD_SEARCH;
    out->enable_hme_flag        = me->enable_hme_flag;
    out->enable_hme_level0_flag = me->enable_hme_level0_flag;
    out->enable_hme_level1_flag = me->enable_hme_level1_flag;
    out->enable_hme_level2_flag = me->enable_hme_level2_flag;
    out->num_hme_sa_w = (uint8_t)me->num_hme_sa_w, out->num_hme_sa_h = (uint8_t)me->num_hme_sa_h;
    SA(out->hme_l0_sa_min, me->hme_l0_sa.sa_min), SA(out->hme_l0_sa_max, me->hme_l0_sa.sa_max);
    SA(out->hme_l1_sa, me->hme_l1_sa), SA(out->hme_l2_sa, me->hme_l2_sa);
    SA(out->me_sa_min, me->me_sa.sa_min), SA(out->me_sa_max, me->me_sa.sa_max);
    out->prehme_enable           = me->prehme_ctrl.enable;
    out->prehme_skip_search_line = me->prehme_ctrl.skip_search_line;
    out->prehme_l1_early_exit    = me->prehme_ctrl.l1_early_exit;
    for (int i = 0; i < 2; i++) {
        SA(out->prehme_sa_min[i], me->prehme_ctrl.prehme_sa_cfg[i].sa_min);
        SA(out->prehme_sa_max[i], me->prehme_ctrl.prehme_sa_cfg[i].sa_max);
    }
    out->enable_me_hme_ref_pruning               = me->me_hme_prune_ctrls.enable_me_hme_ref_pruning;
    out->prune_ref_if_hme_sad_dev_bigger_than_th = me->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th;
    out->prune_ref_if_me_sad_dev_bigger_than_th  = me->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th;
    out->zz_sad_th = me->me_hme_prune_ctrls.zz_sad_th, out->zz_sad_pct = me->me_hme_prune_ctrls.zz_sad_pct;
    out->phme_sad_th = me->me_hme_prune_ctrls.phme_sad_th, out->phme_sad_pct = me->me_hme_prune_ctrls.phme_sad_pct;
    out->enable_me_sr_adjustment              = me->me_sr_adjustment_ctrls.enable_me_sr_adjustment;
    out->distance_based_hme_resizing          = me->me_sr_adjustment_ctrls.distance_based_hme_resizing;
    out->reduce_me_sr_based_on_mv_length_th   = me->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th;
    out->stationary_hme_sad_abs_th            = me->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th;
    out->stationary_me_sr_divisor             = me->me_sr_adjustment_ctrls.stationary_me_sr_divisor;
    out->reduce_me_sr_based_on_hme_sad_abs_th = me->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th;
    out->me_sr_divisor_for_low_hme_sad        = me->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad;
    out->me_8x8_var_enabled = me->me_8x8_var_ctrls.enabled;
    out->me_sr_div4_th = me->me_8x8_var_ctrls.me_sr_div4_th, out->me_sr_div2_th = me->me_8x8_var_ctrls.me_sr_div2_th;
    out->me_sr_mult2_th             = me->me_8x8_var_ctrls.me_sr_mult2_th;
    out->mv_sa_adj_enabled          = me->mv_based_sa_adj.enabled;
    out->mv_sa_adj_nearest_ref_only = me->mv_based_sa_adj.nearest_ref_only;
    out->mv_sa_adj_mv_size_th       = me->mv_based_sa_adj.mv_size_th;
    out->mv_sa_adj_sa_multiplier    = me->mv_based_sa_adj.sa_multiplier;
    out->reduce_hme_l0_sr_th_min = me->reduce_hme_l0_sr_th_min, out->reduce_hme_l0_sr_th_max = me->reduce_hme_l0_sr_th_max;
    out->me_early_exit_th            = me->me_early_exit_th;
    out->me_safe_limit_zz_th         = me->me_safe_limit_zz_th;
    out->prev_me_stage_based_exit_th = me->prev_me_stage_based_exit_th;
    out->prune_me_candidates_th      = me->prune_me_candidates_th;
    out->use_best_unipred_cand_only  = me->use_best_unipred_cand_only;
    out->num_of_list_to_search       = me->num_of_list_to_search;
    out->num_of_ref_pic_to_search[0] = me->num_of_ref_pic_to_search[0];
    out->num_of_ref_pic_to_search[1] = me->num_of_list_to_search > 1 ? me->num_of_ref_pic_to_search[1] : 0;
    out->temporal_layer_index = me->temporal_layer_index, out->is_ref = me->is_ref;
    out->hierarchical_levels     = pcs->hierarchical_levels;
    out->similar_brightness_refs = pcs->similar_brightness_refs;
    out->enable_me_8x8 = pcs->enable_me_8x8, out->enable_me_16x16 = pcs->enable_me_16x16;
    out->max_number_of_pus_per_sb = pcs->max_number_of_pus_per_sb;
    if (pcs->pa_me_data) /* not attached yet when the temporal filter runs (svt_hip_bind_tf.c): ME_MCTF does not store candidates */
        out->max_cand = pcs->pa_me_data->max_cand, out->max_refs = pcs->pa_me_data->max_refs, out->max_l0 = pcs->pa_me_data->max_l0;
    out->only_l_bwd               = scs->mrp_ctrls.only_l_bwd;
    out->input_resolution_le_480p = scs->input_resolution <= INPUT_SIZE_480p_RANGE;
    out->picture_number           = pcs->picture_number;
    for (int l = 0; l < out->num_of_list_to_search; l++)
        for (int r = 0; r < out->num_of_ref_pic_to_search[l]; r++) out->ref_picture_number[l][r] = me->me_ds_ref_array[l][r].picture_number;
}


static size_t plane_bytes(const EbPictureBufferDesc *d) { return (size_t)d->stride_y * (d->height + 2u * d->org_y); }
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

/* the device mirror of one luma plane; `pinned` collects the host addresses to unpin after the launch */
static int get_plane(SvtHipPlane8 *p, const EbPictureBufferDesc *d, uint64_t picture_number, const void **pinned, int *n_pinned) {
    p->stride = d->stride_y, p->org_x = d->org_x, p->org_y = d->org_y, p->width = d->width, p->height = d->height;
    p->buf = hd_mirror_get(d->buffer_y, plane_bytes(d), HD_TAG(picture_number, HD_ST_FILTERED));
    if (!p->buf)
        return -1;
    pinned[(*n_pinned)++] = d->buffer_y;
    return 0;
}

static PicResults *compute_picture(PictureParentControlSet *pcs, MeContext *me, EbPictureBufferDesc *full, EbPictureBufferDesc *quarter,
                                   EbPictureBufferDesc *sixteenth) {
    SvtHipMeFrameJob *job = (SvtHipMeFrameJob *)calloc(1, sizeof(*job));
    PicResults       *e   = (PicResults *)calloc(1, sizeof(*e));
    if (!job || !e) {
        free(job), free(e);
        return NULL;
    }
    svt_hip_bind_me_params(&job->prm, pcs, me);
    const SvtHipMeParams *p = &job->prm;
    const uint32_t nb = pcs->b64_total_count, stored = svt_hip_me_stored_pus(p);
    e->stored = stored, e->max_refs = p->max_refs, e->max_cand = p->max_cand;
    const size_t n_mv = (size_t)nb * stored * p->max_refs * 4, n_cand = (size_t)nb * stored * p->max_cand, n_cnt = (size_t)nb * stored;
    const size_t n_best = (size_t)nb * 2 * 4 * 85 * 4, n_sr = (size_t)nb * 8 * sizeof(SvtHipMeSearchResult), n_d = (size_t)nb * 4;
    const void  *pinned[3 + 3 * SVT_HIP_ME_MAX_LIST * SVT_HIP_ME_MAX_REF];
    int          n_pinned = 0;
    int          rc = get_plane(&job->src.full, full, pcs->picture_number, pinned, &n_pinned) |
        get_plane(&job->src.quarter, quarter, pcs->picture_number, pinned, &n_pinned) |
        get_plane(&job->src.sixteenth, sixteenth, pcs->picture_number, pinned, &n_pinned);
    for (int l = 0; rc == 0 && l < p->num_of_list_to_search; l++)
        for (int r = 0; rc == 0 && r < p->num_of_ref_pic_to_search[l]; r++) {
            const EbDownScaledBufDescPtrArray *a = &me->me_ds_ref_array[l][r];
            rc = get_plane(&job->ref[l][r].full, a->picture_ptr, a->picture_number, pinned, &n_pinned) |
                get_plane(&job->ref[l][r].quarter, a->quarter_picture_ptr, a->picture_number, pinned, &n_pinned) |
                get_plane(&job->ref[l][r].sixteenth, a->sixteenth_picture_ptr, a->picture_number, pinned, &n_pinned);
        }
    uint8_t *dev = rc == 0 ? hd_alloc(al256(n_mv) + al256(n_cand) + al256(n_cnt) + 2 * al256(n_best) + al256(n_sr) + 6 * al256(n_d)) : NULL;
    if (rc == 0 && !dev)
        rc = -1;
    uint8_t *d_mv = dev, *d_cand = d_mv + al256(n_mv), *d_cnt = d_cand + al256(n_cand), *d_bs = d_cnt + al256(n_cnt);
    uint8_t *d_bm = d_bs + al256(n_best), *d_sr = d_bm + al256(n_best), *d_dist = d_sr + al256(n_sr);
    if (rc == 0) {
        SvtHipMeFrameOut *o = &job->out;
        o->me_mv_array = (uint32_t *)d_mv, o->me_candidate_array = d_cand, o->total_me_candidate_index = d_cnt;
        o->best_sad = (uint32_t *)d_bs, o->best_mv = (uint32_t *)d_bm, o->search_results = (SvtHipMeSearchResult *)d_sr;
        uint32_t **dd[6] = {&o->me_64x64_distortion, &o->me_32x32_distortion, &o->me_16x16_distortion, &o->me_8x8_distortion,
                            &o->me_8x8_cost_variance, &o->rc_me_distortion};
        for (int i = 0; i < 6; i++) *dd[i] = (uint32_t *)(d_dist + i * al256(n_d));
        rc = p_me_frames(job, 1, NULL);
    }
    if (rc == 0) {
        e->mv = (uint32_t *)malloc(n_mv), e->cand = (uint8_t *)malloc(n_cand), e->cnt = (uint8_t *)malloc(n_cnt);
        for (int i = 0; i < 6; i++) e->dist[i] = (uint32_t *)malloc(n_d);
        rc = hd_download(e->mv, d_mv, n_mv) | hd_download(e->cand, d_cand, n_cand) | hd_download(e->cnt, d_cnt, n_cnt);
        for (int i = 0; i < 6; i++) rc |= hd_download(e->dist[i], d_dist + i * al256(n_d), n_d);
        rc |= hd_sync();
    } else {
        hd_sync();
    }
    for (int i = 0; i < n_pinned; i++) hd_mirror_unpin(pinned[i]);
    if (rc != 0)
        fprintf(stderr, "svt_hip_bind_me: picture %llu falls back to the CPU search (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
    hd_free(dev);
    free(job);
    if (rc != 0) {
        results_free(e);
        return NULL;
    }
    hd_count_picture();
    return e;
}

/* Returns 0 when block b64_index of the picture has been filled in from the GPU results, 1 when the caller must run the
 * reference's svt_aom_motion_estimation_b64 itself. */
int svt_hip_bind_me_b64(PictureParentControlSet *pcs, uint32_t b64_index, MeContext *me, EbPictureBufferDesc *full, EbPictureBufferDesc *quarter,
                        EbPictureBufferDesc *sixteenth) {
    if (!g_active || me->me_type != ME_OPEN_LOOP || pcs->gm_ctrls.enabled || pcs->frame_superres_enabled || pcs->frame_resize_enabled)
        return 1;
    int     first;
    HdOnce *once = hd_once_enter(&g_tab, pcs, pcs->picture_number, pcs->b64_total_count, &first);
    if (!once)
        return 1;
    if (first) {
        const uint64_t t0 = hd_now_ns();
        PicResults    *r  = compute_picture(pcs, me, full, quarter, sixteenth);
        hd_timer_add("me_picture", hd_now_ns() - t0);
        if (r)
            __atomic_add_fetch(&g_pictures, 1, __ATOMIC_RELAXED);
        hd_once_done(once, r != NULL, r);
    }
    const int ok = hd_once_ok(once);
    if (ok) {
        const PicResults *e   = (const PicResults *)hd_once_payload(once);
        MeSbResults      *res = pcs->pa_me_data->me_results[b64_index];
        memcpy(res->me_mv_array, e->mv + (size_t)b64_index * e->stored * e->max_refs, (size_t)e->stored * e->max_refs * 4);
        memcpy(res->me_candidate_array, e->cand + (size_t)b64_index * e->stored * e->max_cand, (size_t)e->stored * e->max_cand);
        memcpy(res->total_me_candidate_index, e->cnt + (size_t)b64_index * e->stored, e->stored);
        pcs->me_64x64_distortion[b64_index]  = e->dist[0][b64_index];
        pcs->me_32x32_distortion[b64_index]  = e->dist[1][b64_index];
        pcs->me_16x16_distortion[b64_index]  = e->dist[2][b64_index];
        pcs->me_8x8_distortion[b64_index]    = e->dist[3][b64_index];
        pcs->me_8x8_cost_variance[b64_index] = e->dist[4][b64_index];
        pcs->rc_me_distortion[b64_index]     = e->dist[5][b64_index];
        /* the tail of svt_aom_motion_estimation_b64 with global motion off (motion_estimation.c:3213-3215) */
        pcs->stationary_block_present_sb[b64_index] = 0;
        pcs->rc_me_allow_gm[b64_index]              = 0;
        __atomic_add_fetch(&g_blocks, 1, __ATOMIC_RELAXED);
    }
    hd_once_release(&g_tab, once, results_free);
    return ok ? 0 : 1;
}

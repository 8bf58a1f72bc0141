/*
 * svt_hip_bind_me.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 2b of INTEGRATION.md):
 * open-loop motion estimation of a whole picture through the BATCHED entry point svt_hip_me_frames instead of one
 * svt_aom_motion_estimation_b64 call per 64x64 block (Source/Lib/Codec/me_process.c:174-290).
 *
 * The patch puts ONE call in front of the reference's per-block call:
 *     if (svt_hip_bind_me_b64(pcs, b64_index, me_ctx, input_padded_pic, quarter_picture_ptr, sixteenth_picture_ptr)) <their call>;
 * The first block of a picture that arrives here (from whichever ME thread / segment) runs the whole picture on the GPU:
 * MeContext -> SvtHipMeParams (the field map of INTEGRATION.md), the source and reference pyramids uploaded, one launch, results
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

/* entry points of libsvtav1_hip.so, resolved by svt_hip_bind_install() (svt_hip_bind.c) */
typedef struct HipApi {
    int32_t (*malloc_)(void **, size_t);
    int32_t (*free_)(void *);
    int32_t (*upload)(void *, const void *, size_t, void *);
    int32_t (*download)(void *, const void *, size_t, void *);
    int32_t (*sync)(void *);
    int32_t (*me_frames)(const SvtHipMeFrameJob *, uint32_t, void *);
    const char *(*last_error)(void);
} HipApi;
static HipApi g_api;
static int    g_active;
static unsigned long g_pictures, g_blocks; /* statistics, printed at exit */

static void report(void) {
    fprintf(stderr, "svt_hip_bind_me: %lu pictures / %lu blocks of open-loop ME through svt_hip_me_frames\n", g_pictures, g_blocks);
}

void svt_hip_bind_me_setup(void *(*sym)(const char *)) {
    g_api.malloc_    = (int32_t(*)(void **, size_t))sym("svt_hip_malloc");
    g_api.free_      = (int32_t(*)(void *))sym("svt_hip_free");
    g_api.upload     = (int32_t(*)(void *, const void *, size_t, void *))sym("svt_hip_upload");
    g_api.download   = (int32_t(*)(void *, const void *, size_t, void *))sym("svt_hip_download");
    g_api.sync       = (int32_t(*)(void *))sym("svt_hip_stream_sync");
    g_api.me_frames  = (int32_t(*)(const SvtHipMeFrameJob *, uint32_t, void *))sym("svt_hip_me_frames");
    g_api.last_error = (const char *(*)(void))sym("svt_hip_last_error");
    const char *env  = getenv("SVTAV1_HIP_TIERB_ME");
    g_active = env && atoi(env) && g_api.malloc_ && g_api.free_ && g_api.upload && g_api.download && g_api.sync && g_api.me_frames;
    if (g_active)
        atexit(report);
}

/* ---- one picture in flight ------------------------------------------------------------------------------------------- */
typedef struct PicEntry {
    PictureParentControlSet *pcs;
    uint64_t                 picture_number;
    int                      state; /* 0 free, 1 being computed, 2 ready, 3 failed (blocks fall back to the reference's call) */
    uint32_t                 consumed, total, stored, max_refs, max_cand;
    uint32_t                *mv;    /* [nb][stored * max_refs] */
    uint8_t                 *cand;  /* [nb][stored * max_cand] */
    uint8_t                 *cnt;   /* [nb][stored] */
    uint32_t                *dist[6]; /* 64, 32, 16, 8, 8x8 cost variance, rc */
} PicEntry;
#define N_ENTRIES 64
static PicEntry        g_tab[N_ENTRIES];
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t  g_cv = PTHREAD_COND_INITIALIZER;

static void entry_release(PicEntry *e) {
    free(e->mv), free(e->cand), free(e->cnt);
    for (int i = 0; i < 6; i++) free(e->dist[i]);
    memset(e, 0, sizeof(*e));
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

/* device copy of one luma plane at dev + *off */
static int put_plane(SvtHipPlane8 *p, const EbPictureBufferDesc *d, uint8_t *dev, size_t *off) {
    p->buf = dev + *off, p->stride = d->stride_y, p->org_x = d->org_x, p->org_y = d->org_y, p->width = d->width, p->height = d->height;
    const size_t n = plane_bytes(d);
    *off += al256(n + 64);
    return g_api.upload(p->buf, d->buffer_y, n, NULL);
}

static int compute_picture(PicEntry *e, PictureParentControlSet *pcs, MeContext *me, EbPictureBufferDesc *full, EbPictureBufferDesc *quarter,
                           EbPictureBufferDesc *sixteenth) {
    SvtHipMeFrameJob *job = (SvtHipMeFrameJob *)calloc(1, sizeof(*job));
    if (!job)
        return -1;
    svt_hip_bind_me_params(&job->prm, pcs, me);
    const SvtHipMeParams *p = &job->prm;
    const uint32_t nb = pcs->b64_total_count, stored = svt_hip_me_stored_pus(p);
    e->total = nb, e->stored = stored, e->max_refs = p->max_refs, e->max_cand = p->max_cand;
    const size_t n_mv = (size_t)nb * stored * p->max_refs * 4, n_cand = (size_t)nb * stored * p->max_cand, n_cnt = (size_t)nb * stored;
    const size_t n_best = (size_t)nb * 2 * 4 * 85 * 4, n_sr = (size_t)nb * 8 * sizeof(SvtHipMeSearchResult), n_d = (size_t)nb * 4;
    size_t       need = 0;
    const EbPictureBufferDesc *planes[3] = {full, quarter, sixteenth};
    for (int k = 0; k < 3; k++) need += al256(plane_bytes(planes[k]) + 64);
    for (int l = 0; l < p->num_of_list_to_search; l++)
        for (int r = 0; r < p->num_of_ref_pic_to_search[l]; r++)
            need += al256(plane_bytes(me->me_ds_ref_array[l][r].picture_ptr) + 64) + al256(plane_bytes(me->me_ds_ref_array[l][r].quarter_picture_ptr) + 64) +
                al256(plane_bytes(me->me_ds_ref_array[l][r].sixteenth_picture_ptr) + 64);
    need += al256(n_mv) + al256(n_cand) + al256(n_cnt) + 2 * al256(n_best) + al256(n_sr) + 6 * al256(n_d);
    uint8_t *dev = NULL;
    int      rc  = g_api.malloc_((void **)&dev, need);
    size_t   off = 0;
    if (rc == 0)
        rc = put_plane(&job->src.full, full, dev, &off) | put_plane(&job->src.quarter, quarter, dev, &off) | put_plane(&job->src.sixteenth, sixteenth, dev, &off);
    for (int l = 0; rc == 0 && l < p->num_of_list_to_search; l++)
        for (int r = 0; rc == 0 && r < p->num_of_ref_pic_to_search[l]; r++)
            rc = put_plane(&job->ref[l][r].full, me->me_ds_ref_array[l][r].picture_ptr, dev, &off) |
                put_plane(&job->ref[l][r].quarter, me->me_ds_ref_array[l][r].quarter_picture_ptr, dev, &off) |
                put_plane(&job->ref[l][r].sixteenth, me->me_ds_ref_array[l][r].sixteenth_picture_ptr, dev, &off);
    uint8_t *d_mv = dev + off, *d_cand = d_mv + al256(n_mv), *d_cnt = d_cand + al256(n_cand), *d_bs = d_cnt + al256(n_cnt);
    uint8_t *d_bm = d_bs + al256(n_best), *d_sr = d_bm + al256(n_best), *d_dist = d_sr + al256(n_sr);
    if (rc == 0) {
        SvtHipMeFrameOut *o = &job->out;
        o->me_mv_array = (uint32_t *)d_mv, o->me_candidate_array = d_cand, o->total_me_candidate_index = d_cnt;
        o->best_sad = (uint32_t *)d_bs, o->best_mv = (uint32_t *)d_bm, o->search_results = (SvtHipMeSearchResult *)d_sr;
        uint32_t **dd[6] = {&o->me_64x64_distortion, &o->me_32x32_distortion, &o->me_16x16_distortion, &o->me_8x8_distortion,
                            &o->me_8x8_cost_variance, &o->rc_me_distortion};
        for (int i = 0; i < 6; i++) *dd[i] = (uint32_t *)(d_dist + i * al256(n_d));
        rc = g_api.me_frames(job, 1, NULL);
    }
    if (rc == 0) {
        e->mv = (uint32_t *)malloc(n_mv), e->cand = (uint8_t *)malloc(n_cand), e->cnt = (uint8_t *)malloc(n_cnt);
        for (int i = 0; i < 6; i++) e->dist[i] = (uint32_t *)malloc(n_d);
        rc = g_api.download(e->mv, d_mv, n_mv, NULL) | g_api.download(e->cand, d_cand, n_cand, NULL) | g_api.download(e->cnt, d_cnt, n_cnt, NULL);
        for (int i = 0; i < 6; i++) rc |= g_api.download(e->dist[i], d_dist + i * al256(n_d), n_d, NULL);
        rc |= g_api.sync(NULL);
    }
    if (rc != 0)
        fprintf(stderr, "svt_hip_bind_me: picture %llu falls back to the CPU search (%s)\n", (unsigned long long)pcs->picture_number,
                g_api.last_error ? g_api.last_error() : "?");
    if (dev)
        g_api.free_(dev);
    free(job);
    return rc;
}

/* Returns 0 when block b64_index of the picture has been filled in from the GPU results, 1 when the caller must run the
 * reference's svt_aom_motion_estimation_b64 itself. */
int svt_hip_bind_me_b64(PictureParentControlSet *pcs, uint32_t b64_index, MeContext *me, EbPictureBufferDesc *full, EbPictureBufferDesc *quarter,
                        EbPictureBufferDesc *sixteenth) {
    if (!g_active || me->me_type != ME_OPEN_LOOP || pcs->gm_ctrls.enabled || pcs->frame_superres_enabled || pcs->frame_resize_enabled)
        return 1;
    pthread_mutex_lock(&g_mu);
    PicEntry *e = NULL, *fr = NULL;
    for (int i = 0; i < N_ENTRIES; i++) {
        if (g_tab[i].state && g_tab[i].pcs == pcs && g_tab[i].picture_number == pcs->picture_number)
            e = &g_tab[i];
        else if (!g_tab[i].state && !fr)
            fr = &g_tab[i];
    }
    if (!e) {
        if (!fr) { /* table full: cannot happen with the reference's look-ahead depth; let the CPU do this block */
            pthread_mutex_unlock(&g_mu);
            return 1;
        }
        e = fr;
        e->pcs = pcs, e->picture_number = pcs->picture_number, e->state = 1, e->consumed = 0, e->total = pcs->b64_total_count;
        pthread_mutex_unlock(&g_mu);
        const int rc = compute_picture(e, pcs, me, full, quarter, sixteenth);
        pthread_mutex_lock(&g_mu);
        e->state = rc == 0 ? 2 : 3;
        g_pictures += rc == 0;
        pthread_cond_broadcast(&g_cv);
    }
    while (e->state == 1) pthread_cond_wait(&g_cv, &g_mu);
    const int ok = e->state == 2;
    if (ok) {
        MeSbResults *res = pcs->pa_me_data->me_results[b64_index];
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
        g_blocks++;
    }
    if (++e->consumed >= e->total)
        entry_release(e);
    pthread_mutex_unlock(&g_mu);
    return ok ? 0 : 1;
}

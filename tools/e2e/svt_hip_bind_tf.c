/*
 * svt_hip_bind_tf.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 6b of INTEGRATION.md):
 * the block loop of produce_temporally_filtered_pic (Source/Lib/Codec/temporal_filtering.c:2752-3308) through the BATCHED entry
 * point svt_hip_tf_filter_picture.  The patch puts `if (svt_hip_bind_tf_picture(...))` in front of the reference's block loop:
 * the first temporal-filter segment of a picture that arrives here runs the WHOLE picture on the GPU (window pictures uploaded,
 * one call, the filtered centre picture downloaded into the planes the reference's loop would have written); the other
 * segments of that picture wait for it and return.  What stays in the reference: which pictures are in the window and the
 * outlier tests (re-evaluated here exactly as at :3002-3030, they are scalar), the decay factors (computed by the reference
 * right before the hook), 10-bit packing before / unpacking after, padding + decimation of the filtered picture.
 * Active with `--asm hip` and SVTAV1_HIP_TIERB_TF=1; a picture this path does not cover (8x8 prediction, sub-64 pictures ...)
 * returns 1 and the reference's own loop runs.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "definitions.h"
#include "me_context.h"
#include "pcs.h"
#include "reference_object.h"
#include "sequence_control_set.h"

#include "svt_hip.h"
#include "svt_hip_tf.h"
#include "svt_hip_bind.h"

typedef struct TfApi {
    int32_t (*malloc_)(void **, size_t);
    int32_t (*free_)(void *);
    int32_t (*upload)(void *, const void *, size_t, void *);
    int32_t (*download)(void *, const void *, size_t, void *);
    int32_t (*memset_)(void *, int32_t, size_t, void *);
    int32_t (*sync)(void *);
    int32_t (*tf_picture)(const SvtHipTfPictureJob *, void *);
    uint64_t (*tf_ws_bytes)(uint32_t, uint32_t, uint32_t);
    const char *(*last_error)(void);
} TfApi;
static TfApi         g_api;
static int           g_active;
static unsigned long g_pictures;

void svt_hip_bind_me_params(SvtHipMeParams *out, const PictureParentControlSet *pcs, const MeContext *me); /* svt_hip_bind_me.c */

static void report(void) { fprintf(stderr, "svt_hip_bind_tf: %lu pictures through svt_hip_tf_filter_picture\n", g_pictures); }

void svt_hip_bind_tf_setup(void *(*sym)(const char *)) {
    g_api.malloc_     = (int32_t(*)(void **, size_t))sym("svt_hip_malloc");
    g_api.free_       = (int32_t(*)(void *))sym("svt_hip_free");
    g_api.upload      = (int32_t(*)(void *, const void *, size_t, void *))sym("svt_hip_upload");
    g_api.download    = (int32_t(*)(void *, const void *, size_t, void *))sym("svt_hip_download");
    g_api.memset_     = (int32_t(*)(void *, int32_t, size_t, void *))sym("svt_hip_memset");
    g_api.sync        = (int32_t(*)(void *))sym("svt_hip_stream_sync");
    g_api.tf_picture  = (int32_t(*)(const SvtHipTfPictureJob *, void *))sym("svt_hip_tf_filter_picture");
    g_api.tf_ws_bytes = (uint64_t(*)(uint32_t, uint32_t, uint32_t))sym("svt_hip_tf_workspace_bytes");
    g_api.last_error  = (const char *(*)(void))sym("svt_hip_last_error");
    const char *env   = getenv("SVTAV1_HIP_TIERB_TF");
    g_active = env && atoi(env) && g_api.malloc_ && g_api.free_ && g_api.upload && g_api.download && g_api.memset_ && g_api.sync &&
        g_api.tf_picture && g_api.tf_ws_bytes;
    if (g_active)
        atexit(report);
}

/* ---- pictures in flight: the first segment computes, the others wait ---------------------------------------------------- */
typedef struct TfEntry {
    PictureParentControlSet *pcs;
    uint64_t                 picture_number;
    int                      state; /* 0 free, 1 being computed, 2 done on the GPU, 3 not covered / failed */
    int                      seen, total;
    uint32_t                 tot[2];
} TfEntry;
#define N_TF 16
static TfEntry         g_tab[N_TF];
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t  g_cv = PTHREAD_COND_INITIALIZER;

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

typedef struct DevPic {
    SvtHipTfPic pic;
    uint8_t    *d_luma8, *d_c8[2], *d_hbd[3];
    size_t      n_luma8, n_c8, n_hbd_y, n_hbd_c;
} DevPic;

/* device copies of one picture of the window: 8-bit luma pyramid (pa reference object), 8-bit chroma, 16-bit planes */
static int put_picture(DevPic *dp, PictureParentControlSet *pcs, EbPictureBufferDesc *pic, int is_highbd, int chroma, uint8_t *dev, size_t *off) {
    EbPaReferenceObject *pa = (EbPaReferenceObject *)pcs->pa_ref_pic_wrapper->object_ptr;
    EbPictureBufferDesc *pl[3] = {pic, pa->quarter_downsampled_picture_ptr, pa->sixteenth_downsampled_picture_ptr};
    SvtHipPlane8        *dst[3] = {&dp->pic.pyr.full, &dp->pic.pyr.quarter, &dp->pic.pyr.sixteenth};
    int                  rc = 0;
    memset(dp, 0, sizeof(*dp));
    for (int k = 0; k < 3; k++) {
        const size_t n = (size_t)pl[k]->stride_y * (pl[k]->height + 2u * pl[k]->org_y);
        dst[k]->buf = dev + *off, dst[k]->stride = pl[k]->stride_y, dst[k]->org_x = pl[k]->org_x, dst[k]->org_y = pl[k]->org_y;
        dst[k]->width = pl[k]->width, dst[k]->height = pl[k]->height;
        rc |= g_api.upload(dst[k]->buf, pl[k]->buffer_y, n, NULL);
        if (k == 0)
            dp->d_luma8 = dst[k]->buf, dp->n_luma8 = n;
        *off += al256(n + 64);
    }
    dp->pic.chroma8_stride = pic->stride_cb;
    dp->n_c8               = (size_t)pic->stride_cb * ((pic->height + 2u * pic->org_y) >> 1);
    if (chroma && !is_highbd)
        for (int c = 0; c < 2; c++) {
            dp->d_c8[c] = dp->pic.chroma8[c] = dev + *off;
            rc |= g_api.upload(dp->d_c8[c], c ? pic->buffer_cr : pic->buffer_cb, dp->n_c8, NULL);
            *off += al256(dp->n_c8 + 64);
        }
    if (is_highbd) {
        dp->n_hbd_y = (size_t)pic->stride_y * (pic->height + 2u * pic->org_y) * 2, dp->n_hbd_c = dp->n_c8 * 2;
        for (int c = 0; c < (chroma ? 3 : 1); c++) {
            const size_t n = c ? dp->n_hbd_c : dp->n_hbd_y;
            dp->d_hbd[c]   = dev + *off;
            dp->pic.hbd[c] = (uint16_t *)dp->d_hbd[c];
            rc |= g_api.upload(dp->d_hbd[c], pcs->altref_buffer_highbd[c], n, NULL);
            *off += al256(n + 64);
        }
    }
    dp->pic.picture_number = pcs->picture_number;
    return rc;
}

static size_t picture_bytes(PictureParentControlSet *pcs, EbPictureBufferDesc *pic) {
    EbPaReferenceObject *pa = (EbPaReferenceObject *)pcs->pa_ref_pic_wrapper->object_ptr;
    EbPictureBufferDesc *pl[3] = {pic, pa->quarter_downsampled_picture_ptr, pa->sixteenth_downsampled_picture_ptr};
    size_t               n = 0;
    for (int k = 0; k < 3; k++) n += al256((size_t)pl[k]->stride_y * (pl[k]->height + 2u * pl[k]->org_y) + 64);
    const size_t c8 = (size_t)pic->stride_cb * ((pic->height + 2u * pic->org_y) >> 1);
    return n + 2 * al256(c8 + 64) + al256((size_t)pic->stride_y * (pic->height + 2u * pic->org_y) * 2 + 64) + 2 * al256(c8 * 2 + 64);
}

static int run_picture(PictureParentControlSet **pcs_list, EbPictureBufferDesc **pics, int index_center, MeContext *ctx, int is_highbd, uint32_t tot[2]) {
    PictureParentControlSet *centre = pcs_list[index_center];
    SequenceControlSet      *scs    = centre->scs;
    EbPictureBufferDesc     *cpic   = pics[index_center];
    const TfControls        *tc     = &ctx->tf_ctrls;
    if (tc->enable_8x8_pred || scs->subsampling_x != 1 || scs->subsampling_y != 1 || cpic->width < 64 || cpic->height < 64 || cpic->org_x < 68 ||
        cpic->org_y < 68 || cpic->stride_cb * 2 != cpic->stride_y || (is_highbd && scs->static_config.encoder_bit_depth != 10))
        return 1;
    SvtHipTfPictureJob *job = (SvtHipTfPictureJob *)calloc(1, sizeof(*job));
    if (!job)
        return 1;
    /* the pictures filtered against, in the reference's order, with its outlier tests (temporal_filtering.c:2990-3030) */
    int       idx[ALTREF_MAX_NFRAMES], n = 0;
    const int start[3] = {0, centre->past_altref_nframes, centre->past_altref_nframes + 1};
    const int end[3]   = {centre->past_altref_nframes - 1, centre->past_altref_nframes, centre->past_altref_nframes + centre->future_altref_nframes};
    for (int seg = 0; seg < 3; seg++)
        for (int fi = start[seg]; fi <= end[seg]; fi += tc->ref_frame_factor) {
            if (fi == index_center)
                continue;
            const uint32_t low_ahd_err = centre->aligned_width * centre->aligned_height;
            const uint8_t  th          = (centre->slice_type == I_SLICE) ? 20 : 40;
            if (pcs_list[fi]->tf_ahd_error_to_central > low_ahd_err &&
                ((int)(((int)pcs_list[fi]->tf_ahd_error_to_central - (int)centre->tf_avg_ahd_error) * 100)) > (th * (int)centre->tf_avg_ahd_error))
                continue;
            uint32_t bright = 0;
            for (uint32_t w = 0; w < scs->picture_analysis_number_of_regions_per_width; w++)
                for (uint32_t h = 0; h < scs->picture_analysis_number_of_regions_per_height; h++)
                    if (abs((int)pcs_list[fi]->average_intensity_per_region[w][h] - (int)centre->average_intensity_per_region[w][h]) > 2 &&
                        pcs_list[fi]->avg_luma != centre->tf_avg_luma)
                        bright++;
            if (bright >= ((14 * scs->picture_analysis_number_of_regions_per_width * scs->picture_analysis_number_of_regions_per_height) / 16))
                continue;
            idx[n++] = fi;
        }
    if (n == 0 || n > SVT_HIP_TF_MAX_REFS) {
        free(job);
        return 1; /* nothing to filter against: the reference's loop does central + normalise, which leaves the picture as it is */
    }
    /* ME parameters: the MeContext as svt_aom_sig_deriv_me_tf left it + what create_me_context_and_picture_control and the frame loop set */
    svt_hip_bind_me_params(&job->me, centre, ctx);
    job->me.me_mctf = 1, job->me.hme_search_method = 1, job->me.tf_me_exit_th = (uint16_t)tc->me_exit_th;
    job->me.hme_l0_sa_min.width = ctx->hme_l0_sa_default_tf.sa_min.width, job->me.hme_l0_sa_min.height = ctx->hme_l0_sa_default_tf.sa_min.height;
    job->me.hme_l0_sa_max.width = ctx->hme_l0_sa_default_tf.sa_max.width, job->me.hme_l0_sa_max.height = ctx->hme_l0_sa_default_tf.sa_max.height;
    job->me.num_of_list_to_search = 1, job->me.num_of_ref_pic_to_search[0] = 1, job->me.num_of_ref_pic_to_search[1] = 0;
    job->me.temporal_layer_index = centre->temporal_layer_index, job->me.is_ref = centre->is_ref;
    if (!job->me.max_refs)
        job->me.max_refs = 1;
    if (!job->me.max_cand)
        job->me.max_cand = 1;
    job->ctrls.half_pel_mode = tc->half_pel_mode, job->ctrls.quarter_pel_mode = tc->quarter_pel_mode, job->ctrls.eight_pel_mode = tc->eight_pel_mode;
    job->ctrls.use_2tap = tc->use_2tap, job->ctrls.sub_sampling_shift = tc->sub_sampling_shift;
    job->ctrls.use_pred_64x64_only_th = tc->use_pred_64x64_only_th, job->ctrls.subpel_early_exit_th = tc->subpel_early_exit_th;
    job->ctrls.use_8bit_subpel = tc->use_8bit_subpel, job->ctrls.use_zz_based_filter = tc->use_zz_based_filter;
    job->ctrls.pred_error_32x32_th = tc->pred_error_32x32_th;
    for (int p = 0; p < 3; p++) job->decay_factor_fp16[p] = ctx->tf_decay_factor_fp16[p];
    job->mv_dist_th = ctx->tf_mv_dist_th, job->chroma = ctx->tf_chroma, job->bit_depth = is_highbd ? 10 : 8;
    job->mi_rows = (uint32_t)centre->av1_cm->mi_rows, job->mi_cols = (uint32_t)centre->av1_cm->mi_cols, job->n_refs = (uint32_t)n;

    const uint64_t wsb = g_api.tf_ws_bytes(cpic->width, cpic->height, (uint32_t)n);
    size_t         need = al256(wsb) + 256;
    need += picture_bytes(centre, cpic);
    for (int k = 0; k < n; k++) need += picture_bytes(pcs_list[idx[k]], pics[idx[k]]);
    uint8_t *dev = NULL;
    int      rc  = g_api.malloc_((void **)&dev, need);
    size_t   off = 0;
    DevPic   dc, dr;
    if (rc == 0)
        rc = put_picture(&dc, centre, cpic, is_highbd, job->chroma, dev, &off);
    job->centre = dc.pic;
    for (int k = 0; rc == 0 && k < n; k++) {
        rc = put_picture(&dr, pcs_list[idx[k]], pics[idx[k]], is_highbd, job->chroma, dev, &off);
        job->ref[k] = dr.pic;
    }
    if (rc == 0) {
        job->workspace = dev + off, job->workspace_bytes = wsb, off += al256(wsb);
        job->tot_blks = (uint32_t *)(dev + off);
        rc = g_api.memset_(job->tot_blks, 0, 8, NULL);
    }
    if (rc == 0)
        rc = g_api.tf_picture(job, NULL);
    if (rc == 0) { /* the filtered centre picture back into the planes the reference's loop writes */
        if (!is_highbd) {
            rc = g_api.download(cpic->buffer_y, dc.d_luma8, dc.n_luma8, NULL);
            if (job->chroma)
                rc |= g_api.download(cpic->buffer_cb, dc.d_c8[0], dc.n_c8, NULL) | g_api.download(cpic->buffer_cr, dc.d_c8[1], dc.n_c8, NULL);
        } else {
            for (int c = 0; c < (job->chroma ? 3 : 1); c++) rc |= g_api.download(centre->altref_buffer_highbd[c], dc.d_hbd[c], c ? dc.n_hbd_c : dc.n_hbd_y, NULL);
        }
        rc |= g_api.download(tot, job->tot_blks, 8, NULL);
        rc |= g_api.sync(NULL);
    }
    if (rc != 0)
        fprintf(stderr, "svt_hip_bind_tf: picture %llu stays on the CPU (%s)\n", (unsigned long long)centre->picture_number,
                g_api.last_error ? g_api.last_error() : "?");
    if (dev)
        g_api.free_(dev);
    free(job);
    return rc != 0;
}

/* Returns 0 when the picture has been filtered on the GPU (the caller skips its block loop), 1 when the caller must run it. */
int svt_hip_bind_tf_picture(PictureParentControlSet **pcs_list, EbPictureBufferDesc **pics, int index_center, MeContext *ctx, int is_highbd) {
    if (!g_active)
        return 1;
    PictureParentControlSet *centre = pcs_list[index_center];
    pthread_mutex_lock(&g_mu);
    TfEntry *e = NULL, *fr = NULL;
    for (int i = 0; i < N_TF; i++) {
        if (g_tab[i].state && g_tab[i].pcs == centre && g_tab[i].picture_number == centre->picture_number)
            e = &g_tab[i];
        else if (!g_tab[i].state && !fr)
            fr = &g_tab[i];
    }
    if (!e) {
        if (!fr) {
            pthread_mutex_unlock(&g_mu);
            return 1;
        }
        e = fr;
        e->pcs = centre, e->picture_number = centre->picture_number, e->state = 1, e->seen = 0, e->total = centre->tf_segments_total_count;
        pthread_mutex_unlock(&g_mu);
        uint32_t  tot[2] = {0, 0};
        const int rc = run_picture(pcs_list, pics, index_center, ctx, is_highbd, tot);
        pthread_mutex_lock(&g_mu);
        e->state = rc == 0 ? 2 : 3;
        if (rc == 0) {
            /* tf_tot_*_blks of the whole picture go to this segment's context (the caller adds every segment's into the pcs) */
            ctx->tf_tot_horz_blks += tot[0], ctx->tf_tot_vert_blks += tot[1];
            g_pictures++;
        }
        pthread_cond_broadcast(&g_cv);
    }
    while (e->state == 1) pthread_cond_wait(&g_cv, &g_mu);
    const int on_gpu = e->state == 2;
    if (++e->seen >= e->total)
        memset(e, 0, sizeof(*e));
    pthread_mutex_unlock(&g_mu);
    return on_gpu ? 0 : 1;
}

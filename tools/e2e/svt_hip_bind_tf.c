/*
 * svt_hip_bind_tf.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 6b of INTEGRATION.md):
 * the block loop of produce_temporally_filtered_pic (Source/Lib/Codec/temporal_filtering.c:2752-3308) through the BATCHED entry
 * point svt_hip_tf_filter_picture.  The patch puts `if (svt_hip_bind_tf_picture(...))` in front of the reference's block loop:
 * the first temporal-filter segment of a picture that arrives here runs the WHOLE picture on the GPU (window pictures from the
 * device-resident picture mirrors of svt_hip_bind_dev.h, one call, the filtered centre picture downloaded into host staging and
 * copied into the planes the reference's loop would have written only once everything has succeeded — a failure half-way leaves
 * the picture untouched for the reference's own loop); the other segments of that picture wait for it and return.  What stays in the reference: which pictures are in the window and the
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
#include "svt_hip_bind_dev.h"

static int32_t (*p_tf_picture)(const SvtHipTfPictureJob *, void *);
static uint64_t (*p_tf_ws_bytes)(uint32_t, uint32_t, uint32_t);
static int           g_active;
static unsigned long g_pictures, g_pictures_ld;
static __thread int  t_low_delay; /* the variant the calling thread is in (svt_hip_bind_tf_picture_ld) */

void svt_hip_bind_me_params(SvtHipMeParams *out, const PictureParentControlSet *pcs, const MeContext *me); /* svt_hip_bind_me.c */

static void report(void) {
    fprintf(stderr, "svt_hip_bind_tf: %lu pictures through svt_hip_tf_filter_picture (%lu of them the low-delay variant)\n", g_pictures, g_pictures_ld);
}

void svt_hip_bind_tf_setup(void *(*sym)(const char *)) {
    p_tf_picture  = (int32_t(*)(const SvtHipTfPictureJob *, void *))sym("svt_hip_tf_filter_picture");
    p_tf_ws_bytes = (uint64_t(*)(uint32_t, uint32_t, uint32_t))sym("svt_hip_tf_workspace_bytes");
    g_active      = hd_env_on("SVTAV1_HIP_TIERB_TF") && g_hd.ok && p_tf_picture && p_tf_ws_bytes;
    if (g_active)
        atexit(report);
}

static HdOnceTable g_tab;
static size_t      al256(size_t v) { return (v + 255) & ~(size_t)255; }

/* device copies of one picture of the window */
typedef struct DevPic {
    SvtHipTfPic pic;
    uint8_t    *d_luma8, *d_c8[2], *d_hbd[3]; /* d_hbd: scratch owned by this struct; the 8-bit planes are mirrors */
    size_t      n_luma8, n_c8, n_hbd_y, n_hbd_c;
    const void *pinned[5];
    int         n_pinned;
} DevPic;

static uint8_t *mirror(DevPic *dp, const void *host, size_t n, uint64_t picture_number) {
    uint8_t *d = hd_mirror_get(host, n, HD_TAG(picture_number, HD_ST_FILTERED));
    if (d)
        dp->pinned[dp->n_pinned++] = host;
    return d;
}

/* 8-bit luma pyramid (the picture + its pa reference object's decimations) and 8-bit chroma from the mirrors -- "FILTERED" stands
 * for "what the buffer holds until this picture's own temporal filter rewrites it" (release_picture_mirrors); the packed 16-bit
 * planes are per-call temporaries of the reference (altref_buffer_highbd) and go through scratch memory. */
static int put_picture(DevPic *dp, PictureParentControlSet *pcs, EbPictureBufferDesc *pic, int is_highbd, int chroma, int scratch_8bit) {
    EbPaReferenceObject *pa = (EbPaReferenceObject *)pcs->pa_ref_pic_wrapper->object_ptr;
    EbPictureBufferDesc *pl[3] = {pic, pa->quarter_downsampled_picture_ptr, pa->sixteenth_downsampled_picture_ptr};
    SvtHipPlane8        *dst[3] = {&dp->pic.pyr.full, &dp->pic.pyr.quarter, &dp->pic.pyr.sixteenth};
    memset(dp, 0, sizeof(*dp));
    for (int k = 0; k < 3; k++) {
        const size_t n = (size_t)pl[k]->stride_y * (pl[k]->height + 2u * pl[k]->org_y);
        dst[k]->stride = pl[k]->stride_y, dst[k]->org_x = pl[k]->org_x, dst[k]->org_y = pl[k]->org_y;
        dst[k]->width = pl[k]->width, dst[k]->height = pl[k]->height;
        if (k == 0)
            dp->n_luma8 = n;
        if (k == 0 && scratch_8bit) /* the centre picture of an 8-bit filter: the caller supplies a scratch copy (filtered in place) */
            continue;
        if (!(dst[k]->buf = mirror(dp, pl[k]->buffer_y, n, pcs->picture_number)))
            return -1;
        if (k == 0)
            dp->d_luma8 = dst[k]->buf;
    }
    dp->pic.chroma8_stride = pic->stride_cb;
    dp->n_c8               = (size_t)pic->stride_cb * ((pic->height + 2u * pic->org_y) >> 1);
    if (chroma && !is_highbd && !scratch_8bit)
        for (int c = 0; c < 2; c++)
            if (!(dp->d_c8[c] = dp->pic.chroma8[c] = mirror(dp, c ? pic->buffer_cr : pic->buffer_cb, dp->n_c8, pcs->picture_number)))
                return -1;
    if (is_highbd) {
        dp->n_hbd_y = (size_t)pic->stride_y * (pic->height + 2u * pic->org_y) * 2, dp->n_hbd_c = dp->n_c8 * 2;
        for (int c = 0; c < (chroma ? 3 : 1); c++) {
            const size_t n = c ? dp->n_hbd_c : dp->n_hbd_y;
            if (!(dp->d_hbd[c] = hd_alloc(n + 256)))
                return -1;
            dp->pic.hbd[c] = (uint16_t *)dp->d_hbd[c];
            if (hd_upload(dp->d_hbd[c], pcs->altref_buffer_highbd[c], n) != 0)
                return -1;
        }
    }
    dp->pic.picture_number = pcs->picture_number;
    return 0;
}
static void release_picture(DevPic *dp) {
    for (int i = 0; i < dp->n_pinned; i++) hd_mirror_unpin(dp->pinned[i]);
    for (int c = 0; c < 3; c++) hd_free(dp->d_hbd[c]);
    dp->n_pinned = 0;
}

/* The temporal filter is about to rewrite (or has rewritten) the centre picture: its source planes and, right behind the block
 * loop, the pa reference object's padded copy and decimations (temporal_filtering.c: pad + decimate of the filtered picture). */
static void drop_picture_mirrors(PictureParentControlSet *pcs, EbPictureBufferDesc *pic) {
    EbPaReferenceObject *pa = (EbPaReferenceObject *)pcs->pa_ref_pic_wrapper->object_ptr;
    hd_mirror_drop(pic->buffer_y), hd_mirror_drop(pic->buffer_cb), hd_mirror_drop(pic->buffer_cr);
    if (pa->input_padded_pic)
        hd_mirror_drop(pa->input_padded_pic->buffer_y);
    hd_mirror_drop(pa->quarter_downsampled_picture_ptr->buffer_y), hd_mirror_drop(pa->sixteenth_downsampled_picture_ptr->buffer_y);
}

static int run_picture(PictureParentControlSet **pcs_list, EbPictureBufferDesc **pics, int index_center, MeContext *ctx, int is_highbd, uint32_t tot[2]) {
    PictureParentControlSet *centre = pcs_list[index_center];
    SequenceControlSet      *scs    = centre->scs;
    EbPictureBufferDesc     *cpic   = pics[index_center];
    const TfControls        *tc     = &ctx->tf_ctrls;
    if ((tc->enable_8x8_pred && !centre->enable_me_8x8) /* tf_8x8_sub_pel_search starts from the ME's 8x8 vectors */ || scs->subsampling_x != 1 || scs->subsampling_y != 1 || cpic->width < 64 || cpic->height < 64 || cpic->org_x < 68 ||
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
    job->ctrls.low_delay           = (uint8_t)t_low_delay;
    job->ctrls.enable_8x8_pred     = tc->enable_8x8_pred ? 1 : 0;
    for (int p = 0; p < 3; p++) job->decay_factor_fp16[p] = ctx->tf_decay_factor_fp16[p];
    job->mv_dist_th = ctx->tf_mv_dist_th, job->chroma = ctx->tf_chroma, job->bit_depth = is_highbd ? 10 : 8;
    job->mi_rows = (uint32_t)centre->av1_cm->mi_rows, job->mi_cols = (uint32_t)centre->av1_cm->mi_cols, job->n_refs = (uint32_t)n;

    const uint64_t wsb = p_tf_ws_bytes(cpic->width, cpic->height, (uint32_t)n);
    uint8_t       *ws  = hd_alloc(al256(wsb) + 256);
    int            rc  = ws ? 0 : -1;
    DevPic         dc, dr[SVT_HIP_TF_MAX_REFS];
    int            n_dr = 0;
    memset(&dc, 0, sizeof(dc));
    if (rc == 0)
        rc = put_picture(&dc, centre, cpic, is_highbd, job->chroma, !is_highbd);
    job->centre = dc.pic;
    for (int k = 0; rc == 0 && k < n; k++) {
        rc = put_picture(&dr[k], pcs_list[idx[k]], pics[idx[k]], is_highbd, job->chroma, 0);
        n_dr++;
        job->ref[k] = dr[k].pic;
    }
    /* the kernel filters the centre picture IN PLACE on the device: it must not do that to the cached mirror (a later window may
     * ask for the unfiltered picture again if this call fails) -- the centre's 8-bit planes are copied into scratch first */
    uint8_t *d_out[3] = {NULL, NULL, NULL};
    size_t   n_out[3] = {0, 0, 0};
    uint8_t *h_out[3] = {NULL, NULL, NULL};
    if (rc == 0) {
        const int np = job->chroma ? 3 : 1;
        for (int c = 0; c < np; c++) n_out[c] = is_highbd ? (c ? dc.n_hbd_c : dc.n_hbd_y) : (c ? dc.n_c8 : dc.n_luma8);
        if (!is_highbd) {
            /* fresh scratch copies of the centre planes (uploaded from the host: the same bytes the mirror holds) */
            const uint8_t *hsrc[3] = {cpic->buffer_y, cpic->buffer_cb, cpic->buffer_cr};
            for (int c = 0; rc == 0 && c < np; c++) {
                if (!(d_out[c] = hd_alloc(n_out[c] + 256)))
                    rc = -1;
                else
                    rc = hd_upload(d_out[c], hsrc[c], n_out[c]);
            }
            if (rc == 0) {
                job->centre.pyr.full.buf = d_out[0];
                if (job->chroma)
                    job->centre.chroma8[0] = d_out[1], job->centre.chroma8[1] = d_out[2];
            }
        } else {
            for (int c = 0; c < np; c++) d_out[c] = NULL; /* the 16-bit planes already are scratch (dc.d_hbd) */
        }
    }
    uint32_t *d_tot = NULL;
    if (rc == 0) {
        job->workspace = ws, job->workspace_bytes = wsb;
        d_tot = (uint32_t *)(ws + al256(wsb));
        job->tot_blks = d_tot;
        rc = g_hd.memset_(d_tot, 0, 8, NULL);
    }
    if (rc == 0)
        rc = p_tf_picture(job, NULL);
    if (rc == 0) { /* the filtered centre picture into host staging; into the encoder's planes only when all of it has arrived */
        const int np = job->chroma ? 3 : 1;
        for (int c = 0; rc == 0 && c < np; c++) {
            h_out[c] = (uint8_t *)hd_host_alloc(n_out[c]);
            rc = h_out[c] ? hd_download(h_out[c], is_highbd ? dc.d_hbd[c] : d_out[c], n_out[c]) : -1;
        }
        if (rc == 0)
            rc = hd_download(tot, d_tot, 8);
        rc |= hd_sync();
        if (rc == 0) {
            uint8_t *hdst[3] = {cpic->buffer_y, cpic->buffer_cb, cpic->buffer_cr};
            for (int c = 0; c < np; c++) memcpy(is_highbd ? (uint8_t *)centre->altref_buffer_highbd[c] : hdst[c], h_out[c], n_out[c]);
        }
    } else {
        hd_sync();
    }
    release_picture(&dc);
    for (int k = 0; k < n_dr; k++) release_picture(&dr[k]);
    for (int c = 0; c < 3; c++) hd_free(d_out[c]), hd_host_free(h_out[c]);
    hd_free(ws);
    if (rc != 0)
        fprintf(stderr, "svt_hip_bind_tf: picture %llu stays on the CPU (%s)\n", (unsigned long long)centre->picture_number, hd_error());
    else
        hd_count_picture();
    free(job);
    return rc != 0;
}

/* Returns 0 when the picture has been filtered on the GPU (the caller skips its block loop), 1 when the caller must run it. */
static int tf_picture(PictureParentControlSet **pcs_list, EbPictureBufferDesc **pics, int index_center, MeContext *ctx, int is_highbd);
int svt_hip_bind_tf_picture(PictureParentControlSet **pcs_list, EbPictureBufferDesc **pics, int index_center, MeContext *ctx, int is_highbd) {
    t_low_delay = 0;
    return tf_picture(pcs_list, pics, index_center, ctx, is_highbd);
}
/* the same in front of the block loop of produce_temporally_filtered_pic_ld (pred_structure LOW_DELAY_B): co-located predictions, no search */
int svt_hip_bind_tf_picture_ld(PictureParentControlSet **pcs_list, EbPictureBufferDesc **pics, int index_center, MeContext *ctx, int is_highbd) {
    static int traced = -1; /* SVTAV1_E2E_TRACE_TF_LD=1: says once that the encoder reached this variant (tests, also without a GPU) */
    if (traced < 0)
        traced = getenv("SVTAV1_E2E_TRACE_TF_LD") != NULL;
    if (traced == 1 && __atomic_exchange_n(&traced, 2, __ATOMIC_RELAXED) == 1)
        fprintf(stderr, "svt_hip_bind_tf: produce_temporally_filtered_pic_ld reached (picture %llu)\n", (unsigned long long)pcs_list[index_center]->picture_number);
    t_low_delay = 1;
    const int rc = tf_picture(pcs_list, pics, index_center, ctx, is_highbd);
    t_low_delay  = 0;
    return rc;
}
static int tf_picture(PictureParentControlSet **pcs_list, EbPictureBufferDesc **pics, int index_center, MeContext *ctx, int is_highbd) {
    PictureParentControlSet *centre = pcs_list[index_center];
    if (!g_active) {
        if (g_hd.ok)
            drop_picture_mirrors(centre, pics[index_center]); /* other hooks may hold mirrors of the picture this loop rewrites */
        return 1;
    }
    int     first;
    HdOnce *once = hd_once_enter(&g_tab, centre, centre->picture_number, centre->tf_segments_total_count, &first);
    if (!once) {
        drop_picture_mirrors(centre, pics[index_center]);
        return 1;
    }
    if (first) {
        uint32_t  tot[2] = {0, 0};
        const uint64_t t0 = hd_now_ns();
        const int      rc = run_picture(pcs_list, pics, index_center, ctx, is_highbd, tot);
        hd_timer_add("tf_picture", hd_now_ns() - t0);
        /* either way the centre picture changes now: here (GPU) or in the reference's loop right behind this call (CPU) */
        drop_picture_mirrors(centre, pics[index_center]);
        if (rc == 0) {
            /* tf_tot_*_blks of the whole picture go to this segment's context (the caller adds every segment's into the pcs) */
            ctx->tf_tot_horz_blks += tot[0], ctx->tf_tot_vert_blks += tot[1];
            __atomic_add_fetch(&g_pictures, 1, __ATOMIC_RELAXED);
            if (t_low_delay)
                __atomic_add_fetch(&g_pictures_ld, 1, __ATOMIC_RELAXED);
        }
        hd_once_done(once, rc == 0, NULL);
    }
    const int on_gpu = hd_once_ok(once);
    hd_once_release(&g_tab, once, NULL);
    return on_gpu ? 0 : 1;
}

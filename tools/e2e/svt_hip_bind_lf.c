/*
 * svt_hip_bind_lf.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 4 of INTEGRATION.md): the
 * in-loop filters of a whole picture through the batched entry points of include/svt_hip_lf.h, called from the reference's own
 * kernels (dlf_process.c, cdef_process.c, rest_process.c and the functions they call):
 *
 *   svt_hip_bind_dlf_frame      in front of the SB loop of svt_av1_loop_filter_frame (deblocking_filter.c:624-653): covers the final
 *                               frame filter of dlf_process.c:106 AND every trial of the level search (try_filter_frame, :872)
 *                               -> svt_hip_loop_filter_frame.  With SB-based deblocking (presets M6+, coding_loop.c:2260-2281) the
 *                               per-SB call is skipped when svt_hip_bind_dlf_deferred() and dlf_process.c makes ONE frame call
 *                               instead (the frame order is equivalent to the SB schedule: tests/test_gpu_lf.py proves it against
 *                               the real function), so the deferral is correct whether that call then runs on the GPU or not.
 *   svt_hip_bind_cdef_seg       in front of cdef_seg_search (cdef_process.c:106-349): the first segment of a picture searches ALL
 *                               filter blocks of all planes -> svt_hip_cdef_search_plane x 3; results into pcs->mse_seg /
 *                               cdef_dir_data / skip_cdef_seg exactly where the reference's loop stores them.
 *   svt_hip_bind_cdef_frame     in front of svt_av1_cdef_frame (enc_cdef.c:284-610) -> svt_hip_cdef_apply_frame.
 *   svt_hip_bind_wiener_stats   in front of svt_av1_compute_stats(_highbd) in search_wiener_seg (restoration_pick.c:1322-1346):
 *                               the first unit of a plane that asks computes M / H of ALL units of the plane ->
 *                               svt_hip_wiener_stats; every unit then copies its own.
 *   svt_hip_bind_lr_frame       in front of svt_av1_loop_restoration_filter_frame (restoration.c:1179-1248) ->
 *                               svt_hip_restoration_filter_frame.
 *
 * Every hook returns 0 when the GPU did the work and 1 when the caller must run the reference's own code (feature off,
 * configuration not covered, any failure: results reach the encoder's buffers only after everything has succeeded).
 * Pictures travel through the device-resident mirrors of svt_hip_bind_dev.h: one upload per stage input (reconstruction before
 * deblocking / after deblocking / after CDEF, the source picture), shared by the calls of that stage (all trials of the
 * deblocking level search, CDEF search + apply, Wiener statistics + final restoration).
 * Active with `--asm hip` and SVTAV1_HIP_TIERB_DLF / _CDEF / _LR = 1.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "definitions.h"
#include "av1_common.h"
#include "cdef.h"
#include "deblocking_common.h"
#include "deblocking_filter.h"
#include "pcs.h"
#include "restoration.h"
#include "sequence_control_set.h"
#include "utility.h"

#include "svt_hip.h"
#include "svt_hip_lf.h"
#include "svt_hip_bind.h"
#include "svt_hip_bind_dev.h"

void svt_aom_get_recon_pic(PictureControlSet *pcs, EbPictureBufferDesc **recon_ptr, Bool is_highbd);

static int32_t (*p_lf_frame)(const SvtHipLfFrame *, void *);
static int32_t (*p_cdef_search)(const SvtHipCdefPlane *, const uint8_t *, const SvtHipCdefSearchParams *, uint64_t *, uint8_t *, int32_t *, void *);
static int32_t (*p_cdef_apply)(const SvtHipCdefPlane *, uint32_t, const uint8_t *, const uint8_t *const *, int32_t, int32_t, const uint8_t *,
                               const int32_t *, void *);
static int32_t (*p_wiener_stats)(const SvtHipWienerUnit *, uint32_t, int32_t, int32_t, int32_t, int64_t *, int64_t *, void *);
static int32_t (*p_lr_frame)(const SvtHipLrPlane *, uint32_t, void *);
static int32_t (*p_copy)(void *, const void *, size_t, void *);
static int32_t (*p_plane_sse)(const void *, uint32_t, const void *, uint32_t, uint32_t, uint32_t, int32_t, uint64_t *, void *);
static int32_t (*p_download_2d)(void *, size_t, const void *, size_t, size_t, size_t, void *);
static int g_dlf, g_cdef, g_lr;
static unsigned long g_n_dlf, g_n_dlf_trials, g_n_cdef_search, g_n_cdef_apply, g_n_wiener, g_n_lr;

static void report(void) {
    fprintf(stderr,
            "svt_hip_bind_lf: %lu frame deblocking calls, %lu CDEF searches, %lu CDEF applications, %lu Wiener statistics planes, %lu restoration "
            "frames on the GPU; %lu trials of the deblocking level search filtered and measured on the device\n",
            g_n_dlf, g_n_cdef_search, g_n_cdef_apply, g_n_wiener, g_n_lr, g_n_dlf_trials);
}

void svt_hip_bind_lf_setup(void *(*sym)(const char *)) {
    p_lf_frame     = (int32_t(*)(const SvtHipLfFrame *, void *))sym("svt_hip_loop_filter_frame");
    p_cdef_search  = (int32_t(*)(const SvtHipCdefPlane *, const uint8_t *, const SvtHipCdefSearchParams *, uint64_t *, uint8_t *, int32_t *,
                                void *))sym("svt_hip_cdef_search_plane");
    p_cdef_apply   = (int32_t(*)(const SvtHipCdefPlane *, uint32_t, const uint8_t *, const uint8_t *const *, int32_t, int32_t, const uint8_t *,
                               const int32_t *, void *))sym("svt_hip_cdef_apply_frame");
    p_wiener_stats = (int32_t(*)(const SvtHipWienerUnit *, uint32_t, int32_t, int32_t, int32_t, int64_t *, int64_t *, void *))sym("svt_hip_wiener_stats");
    p_lr_frame     = (int32_t(*)(const SvtHipLrPlane *, uint32_t, void *))sym("svt_hip_restoration_filter_frame");
    p_copy         = (int32_t(*)(void *, const void *, size_t, void *))sym("svt_hip_copy");
    p_plane_sse    = (int32_t(*)(const void *, uint32_t, const void *, uint32_t, uint32_t, uint32_t, int32_t, uint64_t *, void *))sym("svt_hip_plane_sse");
    p_download_2d  = (int32_t(*)(void *, size_t, const void *, size_t, size_t, size_t, void *))sym("svt_hip_download_2d");
    const int base = g_hd.ok && p_copy && p_download_2d;
    g_dlf          = base && hd_env_on("SVTAV1_HIP_TIERB_DLF") && p_lf_frame;
    g_cdef         = base && hd_env_on("SVTAV1_HIP_TIERB_CDEF") && p_cdef_search && p_cdef_apply;
    g_lr           = base && hd_env_on("SVTAV1_HIP_TIERB_LR") && p_wiener_stats && p_lr_frame;
    if (g_dlf || g_cdef || g_lr)
        atexit(report);
}

/* ---- picture planes --------------------------------------------------------------------------------------------------- */
typedef struct PlaneRef {
    uint8_t *host;   /* start of the allocation (padding included) */
    size_t   bytes;  /* of the allocation */
    size_t   origin; /* byte offset of sample (0, 0) */
    uint32_t stride; /* samples */
} PlaneRef;

static void picture_planes(const EbPictureBufferDesc *pic, int is16, PlaneRef out[3]) {
    out[0].host = pic->buffer_y, out[0].bytes = (size_t)pic->luma_size << is16, out[0].stride = pic->stride_y;
    out[0].origin = ((size_t)pic->org_y * pic->stride_y + pic->org_x) << is16;
    out[1].host = pic->buffer_cb, out[1].bytes = (size_t)pic->chroma_size << is16, out[1].stride = pic->stride_cb;
    out[1].origin = ((size_t)(pic->org_y >> 1) * pic->stride_cb + (pic->org_x >> 1)) << is16;
    out[2].host = pic->buffer_cr, out[2].bytes = (size_t)pic->chroma_size << is16, out[2].stride = pic->stride_cr;
    out[2].origin = ((size_t)(pic->org_y >> 1) * pic->stride_cr + (pic->org_x >> 1)) << is16;
}

static int covered(const PictureControlSet *pcs) {
    const SequenceControlSet *scs = pcs->scs;
    return scs->subsampling_x == 1 && scs->subsampling_y == 1 && scs->sb_size == 64 && !pcs->ppcs->frame_superres_enabled &&
        !pcs->ppcs->frame_resize_enabled && scs->static_config.resize_mode == RESIZE_NONE &&
        (scs->static_config.encoder_bit_depth == 8 || scs->static_config.encoder_bit_depth == 10);
}

/* ============================================================ deblocking ================================================== */
/* SVTAV1_HIP_DLF_DEFER_TEST=1 defers also without a device (the frame call then runs the reference's own loops): the CPU test of
 * the claim that one frame call behind EncDec equals the per-SB calls inside it (tests/test_e2e.py) */
int svt_hip_bind_dlf_deferred(void) {
    static int forced = -1;
    if (forced < 0)
        forced = hd_env_on("SVTAV1_HIP_DLF_DEFER_TEST");
    return g_dlf || forced;
}

/* the mode-info records of a picture, gathered once per picture (the level search calls the frame filter up to ~20 times) */
typedef struct MiGather {
    const void *grid; /* pcs->mi_grid_base */
    uint64_t    picture_number;
    SvtHipLfMi *mi;
    size_t      n;
} MiGather;
#define N_GATHER 8
static MiGather        g_gather[N_GATHER];
static unsigned        g_gather_next;
static pthread_mutex_t g_gather_mu = PTHREAD_MUTEX_INITIALIZER;

/* one 8-byte record per 4x4 from the fields set_lpf_parameters reads (deblocking_filter.c:162-282) */
static const SvtHipLfMi *gather_mi(PictureControlSet *pcs, size_t *bytes) {
    const Av1Common *cm = pcs->ppcs->av1_cm;
    const size_t     n  = (size_t)cm->mi_rows * pcs->mi_stride;
    pthread_mutex_lock(&g_gather_mu);
    for (int i = 0; i < N_GATHER; i++)
        if (g_gather[i].mi && g_gather[i].grid == pcs->mi_grid_base && g_gather[i].picture_number == pcs->picture_number && g_gather[i].n == n) {
            *bytes = n * sizeof(SvtHipLfMi);
            pthread_mutex_unlock(&g_gather_mu);
            return g_gather[i].mi;
        }
    MiGather *g = &g_gather[g_gather_next++ % N_GATHER];
    if (g->mi)
        hd_mirror_drop(g->mi);
    SvtHipLfMi *mi = (SvtHipLfMi *)realloc(g->mi, n * sizeof(*mi));
    if (!mi) {
        free(g->mi), g->mi = NULL;
        pthread_mutex_unlock(&g_gather_mu);
        return NULL;
    }
    for (int32_t r = 0; r < cm->mi_rows; r++)
        for (int32_t c = 0; c < pcs->mi_stride; c++) {
            SvtHipLfMi     *o = &mi[(size_t)r * pcs->mi_stride + c];
            const ModeInfo *p = pcs->mi_grid_base[(size_t)r * pcs->mi_stride + c];
            memset(o, 0, sizeof(*o));
            if (!p || c >= cm->mi_cols)
                continue;
            const BlockModeInfoEnc *b          = &p->mbmi.block_mi;
            const int               skip_inter = b->skip && is_inter_block_no_intrabc(b->ref_frame[0]);
            o->bsize                           = (uint8_t)b->bsize;
            o->tx_size_y                       = (uint8_t)tx_depth_to_tx_size[skip_inter ? 0 : b->tx_depth][b->bsize];
            o->tx_size_uv                      = (uint8_t)av1_get_max_uv_txsize(b->bsize, 1, 1);
            o->skip_inter                      = (uint8_t)skip_inter;
            o->segment_id                      = b->segment_id;
            o->ref_frame0                      = (uint8_t)b->ref_frame[0];
            o->mode_lf                         = (uint8_t)mode_lf_lut[b->mode];
        }
    g->mi = mi, g->grid = pcs->mi_grid_base, g->picture_number = pcs->picture_number, g->n = n;
    *bytes = n * sizeof(SvtHipLfMi);
    pthread_mutex_unlock(&g_gather_mu);
    return mi;
}

static int dlf_frame_impl(EbPictureBufferDesc *frame_buffer, PictureControlSet *pcs, int32_t plane_start, int32_t plane_end, int64_t *trial_sse);
int svt_hip_bind_dlf_frame(EbPictureBufferDesc *frame_buffer, PictureControlSet *pcs, int32_t plane_start, int32_t plane_end) {
    if (!g_dlf)
        return 1;
    const uint64_t t0 = hd_now_ns();
    const int      rc = dlf_frame_impl(frame_buffer, pcs, plane_start, plane_end, NULL);
    hd_timer_add("dlf_frame", hd_now_ns() - t0);
    return rc;
}
/* One trial of the deblocking level search (try_filter_frame, deblocking_filter.c:842-882) entirely on the device: the plane is filtered in a
 * scratch copy of the mirror of the un-filtered reconstruction and compared with the source picture's mirror there (picture_sse_calculations,
 * :716-834); 8 bytes come back instead of the filtered plane, and the encoder's buffer is never touched (the reference filters it, measures
 * and restores it).  Returns 0 with *filt_err set, 1 when the caller must run the reference's sequence. */
int svt_hip_bind_dlf_try(EbPictureBufferDesc *frame_buffer, PictureControlSet *pcs, int32_t plane, int64_t *filt_err) {
    if (!g_dlf || !p_plane_sse || plane < 0 || plane > 2)
        return 1;
    const uint64_t t0 = hd_now_ns();
    const int      rc = dlf_frame_impl(frame_buffer, pcs, plane, plane + 1, filt_err);
    hd_timer_add("dlf_trial", hd_now_ns() - t0);
    return rc;
}
static int dlf_frame_impl(EbPictureBufferDesc *frame_buffer, PictureControlSet *pcs, int32_t plane_start, int32_t plane_end, int64_t *trial_sse) {
    if (!g_dlf || !covered(pcs) || pcs->ppcs->frm_hdr.delta_lf_params.delta_lf_present || plane_start < 0 || plane_end > 3 || plane_start >= plane_end)
        return 1;
    SequenceControlSet      *scs  = pcs->scs;
    PictureParentControlSet *ppcs = pcs->ppcs;
    const Av1Common         *cm   = ppcs->av1_cm;
    const int                is16 = scs->is_16bit_pipeline;
    /* what svt_av1_loop_filter_frame does first (:640) */
    svt_av1_loop_filter_frame_init(&ppcs->frm_hdr, &ppcs->lf_info, plane_start, plane_end);
    PlaneRef pl[3];
    picture_planes(frame_buffer, is16, pl);
    size_t            mi_bytes = 0;
    const SvtHipLfMi *h_mi     = gather_mi(pcs, &mi_bytes);
    if (!h_mi)
        return 1;
    SvtHipLfFrame f;
    memset(&f, 0, sizeof(f));
    int      rc = 0, mi_pinned = 0;
    uint8_t *d_mi = hd_mirror_get(h_mi, mi_bytes, HD_TAG(pcs->picture_number, HD_ST_MI_LF));
    if (!d_mi)
        rc = -1;
    else
        mi_pinned = 1;
    /* the planes to filter: a scratch copy of the mirror of the picture as EncDec left it (the level search filters that same
     * picture again and again: the reference restores it after every trial, the mirror never changed) */
    uint8_t *d_work[3] = {NULL, NULL, NULL};
    for (int p = plane_start; rc == 0 && p < plane_end; p++) {
        uint8_t *d_src = hd_mirror_get(pl[p].host, pl[p].bytes, HD_TAG(pcs->picture_number, HD_ST_RECON));
        if (!d_src) {
            rc = -1;
            break;
        }
        d_work[p] = hd_alloc(pl[p].bytes + 256);
        rc        = d_work[p] ? p_copy(d_work[p], d_src, pl[p].bytes, NULL) : -1;
        if (rc == 0)
            rc = hd_sync(); /* the copy has read the mirror before it is unpinned */
        hd_mirror_unpin(pl[p].host);
    }
    if (rc == 0) {
        for (int p = 0; p < 3; p++) f.plane[p] = d_work[p] ? d_work[p] + pl[p].origin : NULL, f.stride[p] = pl[p].stride;
        _Static_assert(sizeof(f.lvl) == sizeof(ppcs->lf_info.lvl), "LoopFilterInfoN.lvl layout");
        f.width = frame_buffer->width, f.height = frame_buffer->height;
        f.mi = (const SvtHipLfMi *)d_mi, f.mi_stride = (uint32_t)pcs->mi_stride, f.mi_rows = (uint32_t)cm->mi_rows, f.mi_cols = (uint32_t)cm->mi_cols;
        memcpy(f.lvl, ppcs->lf_info.lvl, sizeof(f.lvl));
        const struct LoopFilter *lf = &ppcs->frm_hdr.loop_filter_params;
        f.filter_level[0] = (uint8_t)lf->filter_level[0], f.filter_level[1] = (uint8_t)lf->filter_level[1];
        f.filter_level_u = (uint8_t)lf->filter_level_u, f.filter_level_v = (uint8_t)lf->filter_level_v;
        f.sharpness_level = (uint8_t)lf->sharpness_level;
        f.bit_depth = (uint8_t)scs->static_config.encoder_bit_depth, f.is_16bit = (uint8_t)is16;
        f.plane_start = (uint8_t)plane_start, f.plane_end = (uint8_t)plane_end;
        rc = p_lf_frame(&f, NULL);
    }
    uint8_t *h_out[3] = {NULL, NULL, NULL};
    if (trial_sse) {
        /* picture_sse_calculations: against the source picture over the aligned size (8 bit) / the picture size (16 bit) */
        const int            p     = plane_start, ss = p ? 1 : 0;
        EbPictureBufferDesc *input = is16 ? pcs->input_frame16bit : ppcs->enhanced_pic;
        PlaneRef             sp[3];
        picture_planes(input, is16, sp);
        const uint32_t w = is16 ? (uint32_t)(input->width + ss) >> ss : (uint32_t)ppcs->aligned_width >> ss;
        const uint32_t h = is16 ? (uint32_t)(input->height + ss) >> ss : (uint32_t)ppcs->aligned_height >> ss;
        uint8_t       *d_src = rc == 0 ? hd_mirror_get(sp[p].host, sp[p].bytes, HD_TAG(pcs->picture_number, is16 ? HD_ST_SOURCE16 : HD_ST_FILTERED)) : NULL;
        uint8_t       *d_sum = rc == 0 ? hd_alloc(256) : NULL;
        uint64_t       sum   = 0;
        if (rc == 0 && (!d_src || !d_sum))
            rc = -1;
        if (rc == 0)
            rc = p_plane_sse(d_src + sp[p].origin, sp[p].stride, d_work[p] + pl[p].origin, pl[p].stride, w, h, is16, (uint64_t *)d_sum, NULL);
        if (rc == 0)
            rc = hd_download(&sum, d_sum, 8);
        rc |= hd_sync();
        if (d_src)
            hd_mirror_unpin(sp[p].host);
        hd_free(d_sum);
        if (rc == 0)
            *trial_sse = (int64_t)sum;
    } else {
    for (int p = plane_start; rc == 0 && p < plane_end; p++) {
        h_out[p] = (uint8_t *)hd_host_alloc(pl[p].bytes);
        rc       = h_out[p] ? hd_download(h_out[p], d_work[p], pl[p].bytes) : -1;
    }
    rc |= hd_sync();
    if (rc == 0)
        for (int p = plane_start; p < plane_end; p++) memcpy(pl[p].host, h_out[p], pl[p].bytes);
    /* the picture's final deblocking (all planes; a trial filters one): the filtered planes become the mirrors of the host planes they were
     * just downloaded into, so that the CDEF search finds its input resident instead of uploading it again */
    if (rc == 0 && plane_start == 0 && plane_end == 3)
        for (int p = 0; p < 3; p++) {
            uint8_t *d_new = hd_mirror_new(pl[p].host, pl[p].bytes, HD_TAG(pcs->picture_number, HD_ST_DEBLOCKED));
            if (!d_new)
                continue;
            if (p_copy(d_new, d_work[p], pl[p].bytes, NULL) != 0 || hd_sync() != 0)
                hd_mirror_drop(pl[p].host);
            hd_mirror_unpin(pl[p].host);
        }
    }
    if (mi_pinned)
        hd_mirror_unpin(h_mi);
    for (int p = 0; p < 3; p++) hd_free(d_work[p]), hd_host_free(h_out[p]);
    if (rc != 0) {
        fprintf(stderr, "svt_hip_bind_lf: deblocking of picture %llu stays on the CPU (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
        return 1;
    }
    __atomic_add_fetch(trial_sse ? &g_n_dlf_trials : &g_n_dlf, 1, __ATOMIC_RELAXED);
    return 0;
}

/* ================================================================ CDEF ==================================================== */
static HdOnceTable g_cdef_tab;

/* filt8x8[r][c] != 0: the 8x8 luma block (r, c) is filtered = not all four of its 4x4 units are skipped (svt_sb_compute_cdef_list,
 * enc_cdef.c:238-276) */
static uint8_t *skip_bitmap(const PictureControlSet *pcs, uint32_t *w8, uint32_t *h8) {
    const Av1Common *cm = pcs->ppcs->av1_cm;
    *w8 = (uint32_t)(cm->mi_cols + 1) >> 1, *h8 = (uint32_t)(cm->mi_rows + 1) >> 1;
    uint8_t *m = (uint8_t *)calloc((size_t)*w8 * *h8, 1);
    if (!m)
        return NULL;
    ModeInfo **grid = pcs->mi_grid_base;
    const int  ms   = pcs->mi_stride;
    for (int32_t r = 0; r + 1 < cm->mi_rows + (cm->mi_rows & 1); r += 2)
        for (int32_t c = 0; c + 1 < cm->mi_cols + (cm->mi_cols & 1); c += 2) {
            const int r1 = r + 1 < cm->mi_rows ? r + 1 : r, c1 = c + 1 < cm->mi_cols ? c + 1 : c;
            m[(size_t)(r >> 1) * *w8 + (c >> 1)] = !grid[r * ms + c]->mbmi.block_mi.skip || !grid[r * ms + c1]->mbmi.block_mi.skip ||
                !grid[r1 * ms + c]->mbmi.block_mi.skip || !grid[r1 * ms + c1]->mbmi.block_mi.skip;
        }
    return m;
}

typedef struct CdefPlanes {
    SvtHipCdefPlane pl[3];
    const void     *pinned[6];
    int             n_pinned;
} CdefPlanes;

/* recon (after deblocking) and the second plane of every SvtHipCdefPlane: the source picture (search) or NULL (apply: set by the caller) */
static int cdef_planes(CdefPlanes *cp, PictureControlSet *pcs, int with_source) {
    const SequenceControlSet *scs  = pcs->scs;
    const Av1Common          *cm   = pcs->ppcs->av1_cm;
    const int                 is16 = scs->is_16bit_pipeline;
    EbPictureBufferDesc      *recon, *input = is16 ? pcs->input_frame16bit : pcs->ppcs->enhanced_pic;
    svt_aom_get_recon_pic(pcs, &recon, is16);
    PlaneRef rp[3], sp[3];
    picture_planes(recon, is16, rp), picture_planes(input, is16, sp);
    memset(cp, 0, sizeof(*cp));
    for (int p = 0; p < 3; p++) {
        SvtHipCdefPlane *o = &cp->pl[p];
        uint8_t         *d = hd_mirror_get(rp[p].host, rp[p].bytes, HD_TAG(pcs->picture_number, HD_ST_DEBLOCKED));
        if (!d)
            return -1;
        cp->pinned[cp->n_pinned++] = rp[p].host;
        o->recon = d + rp[p].origin, o->recon_stride = rp[p].stride;
        if (with_source) {
            uint8_t *s = hd_mirror_get(sp[p].host, sp[p].bytes, HD_TAG(pcs->picture_number, is16 ? HD_ST_SOURCE16 : HD_ST_FILTERED));
            if (!s)
                return -1;
            cp->pinned[cp->n_pinned++] = sp[p].host;
            o->source = s + sp[p].origin, o->source_stride = sp[p].stride;
        }
        o->width = (uint32_t)(cm->mi_cols * 4) >> (p ? 1 : 0), o->height = (uint32_t)(cm->mi_rows * 4) >> (p ? 1 : 0);
        o->is_16bit = (uint8_t)is16, o->xdec = o->ydec = p ? 1 : 0, o->pli = (uint8_t)p;
    }
    return 0;
}
static void cdef_planes_release(CdefPlanes *cp) {
    for (int i = 0; i < cp->n_pinned; i++) hd_mirror_unpin(cp->pinned[i]);
    cp->n_pinned = 0;
}

#define CDEF_DEFAULT_MSE_UV ((uint64_t)1040400 * 64) /* default_mse_uv * 64, cdef_process.c:78, :251 */

static int cdef_search_picture(PictureControlSet *pcs, SequenceControlSet *scs) {
    PictureParentControlSet *ppcs = pcs->ppcs;
    const Av1Common         *cm   = ppcs->av1_cm;
    const CdefControls      *cc   = &ppcs->cdef_ctrls;
    const int                n1 = cc->first_pass_fs_num, n2 = cc->default_second_pass_fs_num, n = n1 + n2;
    if (n < 1 || n > SVT_HIP_CDEF_MAX_STRENGTHS)
        return 1;
    const int32_t nvfb = (cm->mi_rows + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nhfb = (cm->mi_cols + MI_SIZE_64X64 - 1) / MI_SIZE_64X64;
    const size_t  nfb  = (size_t)nvfb * nhfb;
    uint32_t      w8, h8;
    uint8_t      *filt = skip_bitmap(pcs, &w8, &h8);
    if (!filt)
        return 1;
    SvtHipCdefSearchParams prm[2]; /* luma, chroma */
    memset(prm, 0, sizeof(prm));
    for (int k = 0; k < 2; k++) {
        prm[k].n_strengths = n;
        for (int gi = 0; gi < n; gi++) {
            const int first = gi < n1;
            const int fs    = first ? cc->default_first_pass_fs[gi] : cc->default_second_pass_fs[gi - n1];
            const int uv    = first ? cc->default_first_pass_fs_uv[gi] : cc->default_second_pass_fs_uv[gi - n1];
            prm[k].strengths[gi] = (int8_t)((k && uv == -1) ? -1 : fs);
        }
        prm[k].pri_damping = prm[k].sec_damping = 3 + (ppcs->frm_hdr.quantization_params.base_q_idx >> 6);
        prm[k].coeff_shift        = AOMMAX(scs->static_config.encoder_bit_depth - 8, 0);
        prm[k].subsampling_factor = cc->subsampling_factor;
    }
    CdefPlanes cp;
    int        rc = cdef_planes(&cp, pcs, 1);
    const size_t n_mse = nfb * n * sizeof(uint64_t), n_dir = nfb * 64, n_var = nfb * 64 * sizeof(int32_t), n_filt = (size_t)w8 * h8;
    uint8_t   *dev = rc == 0 ? hd_alloc(3 * n_mse + n_dir + n_var + n_filt + 1024) : NULL;
    if (rc == 0 && !dev)
        rc = -1;
    uint8_t  *d_mse = dev, *d_dir = dev ? dev + 3 * n_mse : NULL, *d_var = dev ? d_dir + ((n_dir + 255) & ~(size_t)255) : NULL;
    uint8_t  *d_filt = dev ? d_var + ((n_var + 255) & ~(size_t)255) : NULL;
    uint64_t *h_mse = (uint64_t *)malloc(3 * n_mse);
    uint8_t  *h_dir = (uint8_t *)malloc(n_dir);
    int32_t  *h_var = (int32_t *)malloc(n_var);
    if (!h_mse || !h_dir || !h_var)
        rc = -1;
    if (rc == 0)
        rc = hd_upload(d_filt, filt, n_filt) | g_hd.memset_(d_mse, 0, 3 * n_mse, NULL);
    for (int p = 0; rc == 0 && p < 3; p++) /* luma first: it writes the directions the chroma planes read */
        rc = p_cdef_search(&cp.pl[p], d_filt, &prm[p ? 1 : 0], (uint64_t *)(d_mse + p * n_mse), d_dir, (int32_t *)d_var, NULL);
    if (rc == 0)
        rc = hd_download(h_mse, d_mse, 3 * n_mse) | hd_download(h_dir, d_dir, n_dir) | hd_download(h_var, d_var, n_var);
    rc |= hd_sync();
    cdef_planes_release(&cp);
    if (rc == 0) {
        for (int32_t fbr = 0; fbr < nvfb; fbr++)
            for (int32_t fbc = 0; fbc < nhfb; fbc++) {
                const size_t fb = (size_t)fbr * nhfb + fbc;
                /* cdef_count == 0 <=> no 8x8 of the filter block is filtered (:197-202) */
                int any = 0;
                for (uint32_t r = fbr * 8u; r < fbr * 8u + 8 && r < h8 && !any; r++)
                    for (uint32_t c = fbc * 8u; c < fbc * 8u + 8 && c < w8; c++)
                        if (filt[(size_t)r * w8 + c]) {
                            any = 1;
                            break;
                        }
                pcs->skip_cdef_seg[fb] = !any;
                if (!any)
                    continue;
                memcpy(pcs->cdef_dir_data[fb].dir, h_dir + fb * 64, 64);
                memcpy(pcs->cdef_dir_data[fb].var, h_var + fb * 64, 64 * sizeof(int32_t));
                for (int gi = 0; gi < n; gi++) {
                    pcs->mse_seg[0][fb][gi] = h_mse[fb * n + gi];
                    pcs->mse_seg[1][fb][gi] = prm[1].strengths[gi] == -1 ? CDEF_DEFAULT_MSE_UV : h_mse[(nfb + fb) * n + gi] + h_mse[(2 * nfb + fb) * n + gi];
                }
            }
    }
    hd_free(dev);
    free(filt), free(h_mse), free(h_dir), free(h_var);
    if (rc != 0) {
        fprintf(stderr, "svt_hip_bind_lf: CDEF search of picture %llu stays on the CPU (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
        return 1;
    }
    __atomic_add_fetch(&g_n_cdef_search, 1, __ATOMIC_RELAXED);
    return 0;
}

int svt_hip_bind_cdef_seg(PictureControlSet *pcs, SequenceControlSet *scs, uint32_t segment_index) {
    (void)segment_index;
    if (!g_cdef || !covered(pcs))
        return 1;
    int     first;
    HdOnce *once = hd_once_enter(&g_cdef_tab, pcs, pcs->picture_number, pcs->cdef_segments_total_count, &first);
    if (!once)
        return 1;
    if (first) {
        const uint64_t t0 = hd_now_ns();
        const int      ok = cdef_search_picture(pcs, scs) == 0;
        hd_timer_add("cdef_search", hd_now_ns() - t0);
        hd_once_done(once, ok, NULL);
    }
    const int ok = hd_once_ok(once);
    hd_once_release(&g_cdef_tab, once, NULL);
    return ok ? 0 : 1;
}

static int cdef_apply_impl(SequenceControlSet *scs, PictureControlSet *pcs);
int svt_hip_bind_cdef_frame(SequenceControlSet *scs, PictureControlSet *pcs) {
    if (!g_cdef)
        return 1;
    const uint64_t t0 = hd_now_ns();
    const int      rc = cdef_apply_impl(scs, pcs);
    hd_timer_add("cdef_apply", hd_now_ns() - t0);
    return rc;
}
static int cdef_apply_impl(SequenceControlSet *scs, PictureControlSet *pcs) {
    if (!g_cdef || !covered(pcs) || pcs->ppcs->cdef_ctrls.use_reference_cdef_fs)
        return 1;
    PictureParentControlSet *ppcs    = pcs->ppcs;
    const Av1Common         *cm      = ppcs->av1_cm;
    const FrameHeader       *frm_hdr = &ppcs->frm_hdr;
    const int                is16    = scs->is_16bit_pipeline;
    const int32_t nvfb = (cm->mi_rows + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nhfb = (cm->mi_cols + MI_SIZE_64X64 - 1) / MI_SIZE_64X64;
    const size_t  nfb  = (size_t)nvfb * nhfb;
    uint32_t      w8, h8;
    uint8_t      *filt = skip_bitmap(pcs, &w8, &h8);
    uint8_t      *fbs  = (uint8_t *)calloc(2 * nfb, 1);
    uint8_t      *h_dir = (uint8_t *)calloc(nfb, 64);
    int32_t      *h_var = (int32_t *)calloc(nfb, 64 * sizeof(int32_t));
    int           rc = (filt && fbs && h_dir && h_var) ? 0 : -1;
    for (int32_t fbr = 0; rc == 0 && fbr < nvfb; fbr++)
        for (int32_t fbc = 0; fbc < nhfb; fbc++) {
            const size_t  fb  = (size_t)fbr * nhfb + fbc;
            const int32_t idx = pcs->mi_grid_base[MI_SIZE_64X64 * fbr * cm->mi_stride + MI_SIZE_64X64 * fbc]->mbmi.cdef_strength;
            if (idx < 0 || idx >= CDEF_MAX_STRENGTHS) {
                rc = -1;
                break;
            }
            fbs[fb] = (uint8_t)frm_hdr->cdef_params.cdef_y_strength[idx], fbs[nfb + fb] = (uint8_t)frm_hdr->cdef_params.cdef_uv_strength[idx];
            /* all four strengths zero: the reference leaves the filter block alone (:398-404) = not filtered for us */
            if (fbs[fb] == 0 && fbs[nfb + fb] == 0)
                for (uint32_t r = fbr * 8u; r < fbr * 8u + 8 && r < h8; r++)
                    for (uint32_t c = fbc * 8u; c < fbc * 8u + 8 && c < w8; c++) filt[(size_t)r * w8 + c] = 0;
            memcpy(h_dir + fb * 64, pcs->cdef_dir_data[fb].dir, 64);
            memcpy(h_var + fb * 64, pcs->cdef_dir_data[fb].var, 64 * sizeof(int32_t));
        }
    CdefPlanes cp;
    memset(&cp, 0, sizeof(cp));
    if (rc == 0)
        rc = cdef_planes(&cp, pcs, 0);
    EbPictureBufferDesc *recon;
    svt_aom_get_recon_pic(pcs, &recon, is16);
    PlaneRef rp[3];
    picture_planes(recon, is16, rp);
    const size_t n_dir = nfb * 64, n_var = nfb * 64 * sizeof(int32_t), n_filt = (size_t)w8 * h8;
    uint8_t     *dev   = rc == 0 ? hd_alloc(n_dir + n_var + n_filt + 2 * nfb + 2048) : NULL;
    if (rc == 0 && !dev)
        rc = -1;
    uint8_t *d_dir = dev, *d_var = dev ? dev + ((n_dir + 255) & ~(size_t)255) : NULL, *d_filt = dev ? d_var + ((n_var + 255) & ~(size_t)255) : NULL;
    uint8_t *d_fbs = dev ? d_filt + ((n_filt + 255) & ~(size_t)255) : NULL;
    uint8_t *d_out[3] = {NULL, NULL, NULL}, *h_out[3] = {NULL, NULL, NULL};
    if (rc == 0)
        rc = hd_upload(d_dir, h_dir, n_dir) | hd_upload(d_var, h_var, n_var) | hd_upload(d_filt, filt, n_filt) | hd_upload(d_fbs, fbs, 2 * nfb);
    for (int p = 0; rc == 0 && p < 3; p++) {
        /* the output plane starts as a copy of the input: the kernel writes the picture area, the padding stays what it was */
        d_out[p] = hd_alloc(rp[p].bytes + 256);
        rc       = d_out[p] ? p_copy(d_out[p], (const uint8_t *)cp.pl[p].recon - rp[p].origin, rp[p].bytes, NULL) : -1;
        cp.pl[p].source = d_out[p] ? d_out[p] + rp[p].origin : NULL, cp.pl[p].source_stride = rp[p].stride;
    }
    if (rc == 0) {
        const uint8_t *strength[3] = {d_fbs, d_fbs + nfb, d_fbs + nfb};
        rc = p_cdef_apply(cp.pl, 3, d_filt, strength, frm_hdr->cdef_params.cdef_damping, AOMMAX(scs->static_config.encoder_bit_depth - 8, 0), d_dir,
                          (const int32_t *)d_var, NULL);
    }
    for (int p = 0; rc == 0 && p < 3; p++) {
        h_out[p] = (uint8_t *)hd_host_alloc(rp[p].bytes);
        rc       = h_out[p] ? hd_download(h_out[p], d_out[p], rp[p].bytes) : -1;
    }
    rc |= hd_sync();
    cdef_planes_release(&cp);
    if (rc == 0)
        for (int p = 0; p < 3; p++) memcpy(rp[p].host, h_out[p], rp[p].bytes);
    for (int p = 0; p < 3; p++) hd_free(d_out[p]), hd_host_free(h_out[p]);
    hd_free(dev);
    free(filt), free(fbs), free(h_dir), free(h_var);
    if (rc != 0) {
        fprintf(stderr, "svt_hip_bind_lf: CDEF of picture %llu stays on the CPU (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
        return 1;
    }
    __atomic_add_fetch(&g_n_cdef_apply, 1, __ATOMIC_RELAXED);
    return 0;
}

/* ============================================================ restoration ================================================= */
/* the limits of every restoration unit of a plane in the order of foreach_rest_unit_in_tile (restoration.c:1250-1293; one tile) */
static int unit_limits(int W, int H, int unit_size, int ss_y, int horz_units, int vert_units, RestorationTileLimits *out) {
    const int ext = unit_size * 3 / 2, voff = RESTORATION_UNIT_OFFSET >> ss_y;
    int       y0 = 0, i = 0;
    while (y0 < H) {
        const int rem_h = H - y0, h = rem_h < ext ? rem_h : unit_size;
        int       x0 = 0, j = 0;
        if (i >= vert_units)
            return -1;
        while (x0 < W) {
            const int rem_w = W - x0, w = rem_w < ext ? rem_w : unit_size;
            if (j >= horz_units)
                return -1;
            RestorationTileLimits *l = &out[i * horz_units + j];
            l->v_start = y0 - voff > 0 ? y0 - voff : 0, l->v_end = y0 + h < H ? y0 + h - voff : y0 + h;
            l->h_start = x0, l->h_end = x0 + w;
            x0 += w, j++;
        }
        y0 += h, i++;
    }
    return 0;
}

typedef struct WienerPlaneStats {
    int                    n_units, win;
    int64_t               *M, *H; /* [n][49], [n][49 * 49] */
    RestorationTileLimits *lim;   /* [n]: the unit geometry the statistics were computed for */
} WienerPlaneStats;
static void wiener_stats_free(WienerPlaneStats *s) {
    if (s)
        free(s->M), free(s->H), free(s->lim), free(s);
}
/* statistics of the planes in flight: (pcs, picture number, plane) -> WienerPlaneStats.  Units that reuse the previous picture's
 * coefficients never ask (search_wiener_seg :1312-1318), so an entry cannot count its callers: it is replaced when its slot is
 * needed again (oldest first); readers copy their unit's share under the lock. */
typedef struct WienerSlot {
    const void       *pcs;
    uint64_t          picture_number;
    int               plane, state; /* 0 free, 1 being computed, 2 ready, 3 failed */
    uint64_t          stamp;
    WienerPlaneStats *st;
} WienerSlot;
#define N_WIENER 24
static WienerSlot      g_wiener[N_WIENER];
static uint64_t        g_wiener_clock;
static pthread_mutex_t g_wiener_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t  g_wiener_cv = PTHREAD_COND_INITIALIZER;

/* plane sample (0,0) pointers as the search sees them (init_rsc_seg: dgd = org_fts, src = cpi_source) against our PlaneRefs */
static WienerPlaneStats *wiener_plane(PictureControlSet *pcs, int plane, int wiener_win, const uint8_t *dgd, const uint8_t *src, int dgd_stride,
                                      int src_stride, int highbd, int bit_depth) {
    const SequenceControlSet *scs  = pcs->scs;
    const Av1Common          *cm   = pcs->ppcs->av1_cm;
    const int                 is16 = scs->is_16bit_pipeline;
    const RestorationInfo    *rsi  = &pcs->rst_info[plane];
    if (scs->use_boundaries_in_rest_search || highbd != is16)
        return NULL;
    EbPictureBufferDesc *recon, *input = is16 ? pcs->input_frame16bit : pcs->ppcs->enhanced_unscaled_pic;
    svt_aom_get_recon_pic(pcs, &recon, is16);
    PlaneRef rp[3], sp[3];
    picture_planes(recon, is16, rp), picture_planes(input, is16, sp);
    const uint8_t *dgd_raw = highbd ? (const uint8_t *)CONVERT_TO_SHORTPTR(dgd) : dgd, *src_raw = highbd ? (const uint8_t *)CONVERT_TO_SHORTPTR(src) : src;
    if (dgd_raw != rp[plane].host + rp[plane].origin || src_raw != sp[plane].host + sp[plane].origin || (uint32_t)dgd_stride != rp[plane].stride ||
        (uint32_t)src_stride != sp[plane].stride)
        return NULL; /* not the buffers this hook knows how to mirror (scaled source, a private copy of the reconstruction) */
    const int ss = plane > 0, W = (cm->frm_size.superres_upscaled_width + ss) >> ss, H = (cm->frm_size.frame_height + ss) >> ss;
    const int n  = rsi->horz_units_per_tile * rsi->vert_units_per_tile;
    if (n < 1)
        return NULL;
    RestorationTileLimits *lim = (RestorationTileLimits *)calloc(n, sizeof(*lim));
    SvtHipWienerUnit      *u   = (SvtHipWienerUnit *)calloc(n, sizeof(*u));
    WienerPlaneStats      *st  = (WienerPlaneStats *)calloc(1, sizeof(*st));
    int rc = (lim && u && st && unit_limits(W, H, rsi->restoration_unit_size, ss, rsi->horz_units_per_tile, rsi->vert_units_per_tile, lim) == 0) ? 0 : -1;
    uint8_t *d_dgd = NULL, *d_src = NULL, *dev = NULL;
    /* the reconstruction after CDEF with the borders restoration_seg_search extended (svt_extend_frame, restoration_pick.c:1511: the
     * statistics read 3 samples beyond the picture) and the source as cdef_process.c left it */
    if (rc == 0 && !(d_dgd = hd_mirror_get(rp[plane].host, rp[plane].bytes, HD_TAG(pcs->picture_number, HD_ST_CDEF_EXT))))
        rc = -1;
    if (rc == 0 && !(d_src = hd_mirror_get(sp[plane].host, sp[plane].bytes, HD_TAG(pcs->picture_number, is16 ? HD_ST_SOURCE16_LR : HD_ST_FILTERED))))
        rc = -1;
    const size_t n_m = (size_t)n * 49 * 8, n_h = (size_t)n * 49 * 49 * 8;
    if (rc == 0 && !(dev = hd_alloc(n_m + n_h + 512)))
        rc = -1;
    if (rc == 0) {
        for (int i = 0; i < n; i++) {
            u[i].dgd = d_dgd + rp[plane].origin, u[i].src = d_src + sp[plane].origin, u[i].dgd_stride = rp[plane].stride, u[i].src_stride = sp[plane].stride;
            u[i].h_start = lim[i].h_start, u[i].h_end = lim[i].h_end, u[i].v_start = lim[i].v_start, u[i].v_end = lim[i].v_end;
        }
        st->n_units = n, st->win = wiener_win;
        st->M = (int64_t *)malloc(n_m), st->H = (int64_t *)malloc(n_h);
        rc = (st->M && st->H) ? p_wiener_stats(u, (uint32_t)n, wiener_win, is16, bit_depth, (int64_t *)dev, (int64_t *)(dev + ((n_m + 255) & ~(size_t)255)), NULL) : -1;
    }
    if (rc == 0)
        rc = hd_download(st->M, dev, n_m) | hd_download(st->H, dev + ((n_m + 255) & ~(size_t)255), n_h);
    rc |= hd_sync();
    if (d_dgd)
        hd_mirror_unpin(rp[plane].host);
    if (d_src)
        hd_mirror_unpin(sp[plane].host);
    hd_free(dev);
    free(u);
    if (rc == 0) {
        st->lim = lim;
        __atomic_add_fetch(&g_n_wiener, 1, __ATOMIC_RELAXED);
        return st;
    }
    fprintf(stderr, "svt_hip_bind_lf: Wiener statistics of picture %llu plane %d stay on the CPU (%s)\n", (unsigned long long)pcs->picture_number, plane,
            hd_error());
    free(lim);
    if (st)
        st->lim = NULL;
    wiener_stats_free(st);
    return NULL;
}

int svt_hip_bind_wiener_stats(PictureControlSet *pcs, int plane, int rest_unit_idx, int wiener_win, const uint8_t *dgd, const uint8_t *src,
                              const RestorationTileLimits *limits, int dgd_stride, int src_stride, int highbd, int bit_depth, int64_t *M, int64_t *H) {
    if (!g_lr || !covered(pcs) || plane < 0 || plane > 2)
        return 1;
    const RestorationInfo *rsi = &pcs->rst_info[plane];
    const int              n   = rsi->horz_units_per_tile * rsi->vert_units_per_tile;
    if (rest_unit_idx < 0 || rest_unit_idx >= n)
        return 1;
    pthread_mutex_lock(&g_wiener_mu);
    WienerSlot *e = NULL;
    for (;;) {
        e = NULL;
        for (int i = 0; i < N_WIENER; i++)
            if (g_wiener[i].state && g_wiener[i].pcs == pcs && g_wiener[i].picture_number == pcs->picture_number && g_wiener[i].plane == plane)
                e = &g_wiener[i];
        if (!e || e->state != 1)
            break;
        pthread_cond_wait(&g_wiener_cv, &g_wiener_mu);
    }
    if (!e) { /* first unit of this plane: take the free or the oldest settled slot and compute the whole plane */
        for (int i = 0; i < N_WIENER && (!e || e->state != 0); i++) {
            WienerSlot *c = &g_wiener[i];
            if (c->state == 1)
                continue;
            if (c->state == 0 || !e || c->stamp < e->stamp)
                e = c;
        }
        if (!e) {
            pthread_mutex_unlock(&g_wiener_mu);
            return 1;
        }
        wiener_stats_free(e->st);
        e->st = NULL, e->pcs = pcs, e->picture_number = pcs->picture_number, e->plane = plane, e->state = 1, e->stamp = ++g_wiener_clock;
        pthread_mutex_unlock(&g_wiener_mu);
        const uint64_t    t0 = hd_now_ns();
        WienerPlaneStats *st = wiener_plane(pcs, plane, wiener_win, dgd, src, dgd_stride, src_stride, highbd, bit_depth);
        hd_timer_add("wiener_plane", hd_now_ns() - t0);
        pthread_mutex_lock(&g_wiener_mu);
        e->st = st, e->state = st ? 2 : 3;
        pthread_cond_broadcast(&g_wiener_cv);
    }
    int rc = 1;
    if (e->state == 2) {
        const WienerPlaneStats *st = e->st;
        /* only for the unit geometry this hook derived itself; anything else the reference computes */
        if (st->win == wiener_win && rest_unit_idx < st->n_units && memcmp(&st->lim[rest_unit_idx], limits, sizeof(*limits)) == 0) {
            memcpy(M, st->M + (size_t)rest_unit_idx * 49, 49 * 8);
            memcpy(H, st->H + (size_t)rest_unit_idx * 49 * 49, 49 * 49 * 8);
            rc = 0;
        }
    }
    pthread_mutex_unlock(&g_wiener_mu);
    return rc;
}

static int lr_frame_impl(Yv12BufferConfig *frame, Av1Common *cm, int32_t optimized_lr);
int svt_hip_bind_lr_frame(Yv12BufferConfig *frame, Av1Common *cm, int32_t optimized_lr) {
    if (!g_lr)
        return 1;
    const uint64_t t0 = hd_now_ns();
    const int      rc = lr_frame_impl(frame, cm, optimized_lr);
    hd_timer_add("lr_frame", hd_now_ns() - t0);
    return rc;
}
static int lr_frame_impl(Yv12BufferConfig *frame, Av1Common *cm, int32_t optimized_lr) {
    PictureControlSet *pcs = cm->child_pcs;
    if (!g_lr || !pcs || !covered(pcs))
        return 1;
    const SequenceControlSet *scs  = pcs->scs;
    const int                 is16 = scs->is_16bit_pipeline;
    if (cm->use_highbitdepth != is16)
        return 1;
    EbPictureBufferDesc *recon;
    svt_aom_get_recon_pic(pcs, &recon, is16);
    PlaneRef rp[3];
    picture_planes(recon, is16, rp);
    SvtHipLrPlane pl[3];
    int           plane_of[3], n_pl = 0, rc = 0;
    uint8_t      *d_out[3] = {NULL, NULL, NULL}, *d_units[3] = {NULL, NULL, NULL}, *d_bnd[3] = {NULL, NULL, NULL}, *h_out[3] = {NULL, NULL, NULL};
    const void   *pinned[3];
    int           n_pinned = 0;
    memset(pl, 0, sizeof(pl));
    for (int plane = 0; rc == 0 && plane < 3; plane++) {
        RestorationInfo *rsi = &pcs->rst_info[plane];
        rsi->optimized_lr    = optimized_lr; /* what the reference's loop sets for every plane (:1207) */
        if (rsi->frame_restoration_type == RESTORE_NONE)
            continue;
        const int      is_uv = plane > 0;
        const uint8_t *buf   = is16 ? (const uint8_t *)CONVERT_TO_SHORTPTR(frame->buffers[plane]) : frame->buffers[plane];
        if (buf != rp[plane].host + rp[plane].origin || (uint32_t)frame->strides[is_uv] != rp[plane].stride) {
            rc = -1; /* not the picture's reconstruction buffer */
            break;
        }
        SvtHipLrPlane *o = &pl[n_pl];
        const int      n = rsi->horz_units_per_tile * rsi->vert_units_per_tile;
        uint8_t       *d = hd_mirror_get(rp[plane].host, rp[plane].bytes, HD_TAG(pcs->picture_number, HD_ST_CDEF_EXT));
        if (!d) {
            rc = -1;
            break;
        }
        pinned[n_pinned++] = rp[plane].host;
        SvtHipLrUnit *hu   = (SvtHipLrUnit *)calloc(n, sizeof(*hu));
        d_out[n_pl]        = hd_alloc(rp[plane].bytes + 256);
        d_units[n_pl]      = hd_alloc((size_t)n * sizeof(*hu) + 256);
        if (!hu || !d_out[n_pl] || !d_units[n_pl]) {
            free(hu);
            rc = -1;
            break;
        }
        for (int i = 0; i < n; i++) {
            const RestorationUnitInfo *ui = &rsi->unit_info[i];
            hu[i].restoration_type       = (uint8_t)ui->restoration_type;
            hu[i].ep = (uint8_t)ui->sgrproj_info.ep, hu[i].xqd[0] = ui->sgrproj_info.xqd[0], hu[i].xqd[1] = ui->sgrproj_info.xqd[1];
            memcpy(hu[i].hfilter, ui->wiener_info.hfilter, sizeof(hu[i].hfilter));
            memcpy(hu[i].vfilter, ui->wiener_info.vfilter, sizeof(hu[i].vfilter));
        }
        rc = hd_upload(d_units[n_pl], hu, (size_t)n * sizeof(*hu));
        free(hu);
        if (rc == 0 && !optimized_lr) {
            const size_t nb = (size_t)rsi->boundaries.stripe_boundary_size;
            d_bnd[n_pl]     = hd_alloc(2 * nb + 512);
            rc = d_bnd[n_pl] ? (hd_upload(d_bnd[n_pl], rsi->boundaries.stripe_boundary_above, nb) |
                                hd_upload(d_bnd[n_pl] + ((nb + 255) & ~(size_t)255), rsi->boundaries.stripe_boundary_below, nb))
                             : -1;
            o->boundary_above = d_bnd[n_pl], o->boundary_below = d_bnd[n_pl] ? d_bnd[n_pl] + ((nb + 255) & ~(size_t)255) : NULL;
            o->boundary_stride = (uint32_t)rsi->boundaries.stripe_boundary_stride;
        }
        o->src = d + rp[plane].origin, o->dst = d_out[n_pl] + rp[plane].origin, o->src_stride = o->dst_stride = rp[plane].stride;
        o->width = (uint32_t)frame->crop_widths[is_uv], o->height = (uint32_t)frame->crop_heights[is_uv];
        o->ss_x = o->ss_y = (uint8_t)is_uv, o->is_16bit = (uint8_t)is16, o->bit_depth = (uint8_t)cm->bit_depth;
        o->unit_size = (uint32_t)rsi->restoration_unit_size, o->horz_units = (uint32_t)rsi->horz_units_per_tile, o->vert_units = (uint32_t)rsi->vert_units_per_tile;
        o->units = (const SvtHipLrUnit *)d_units[n_pl], o->optimized_lr = (uint32_t)optimized_lr;
        plane_of[n_pl++] = plane;
    }
    if (rc == 0 && n_pl)
        rc = p_lr_frame(pl, (uint32_t)n_pl, NULL);
    /* only the picture area comes back (copy_funs[plane](dst, frame) copies the cropped plane, :1243) */
    for (int k = 0; rc == 0 && k < n_pl; k++) {
        const size_t row = (size_t)pl[k].width << is16;
        h_out[k]         = (uint8_t *)hd_host_alloc(row * pl[k].height);
        rc = h_out[k] ? p_download_2d(h_out[k], row, pl[k].dst, (size_t)pl[k].dst_stride << is16, row, pl[k].height, NULL) : -1;
    }
    rc |= hd_sync();
    for (int i = 0; i < n_pinned; i++) hd_mirror_unpin(pinned[i]);
    if (rc == 0)
        for (int k = 0; k < n_pl; k++) {
            const PlaneRef *r   = &rp[plane_of[k]];
            const size_t    row = (size_t)pl[k].width << is16;
            for (uint32_t y = 0; y < pl[k].height; y++) memcpy(r->host + r->origin + (((size_t)y * r->stride) << is16), h_out[k] + y * row, row);
        }
    for (int k = 0; k < 3; k++) hd_free(d_out[k]), hd_free(d_units[k]), hd_free(d_bnd[k]), hd_host_free(h_out[k]);
    if (rc != 0) {
        fprintf(stderr, "svt_hip_bind_lf: restoration of picture %llu stays on the CPU (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
        return 1;
    }
    __atomic_add_fetch(&g_n_lr, 1, __ATOMIC_RELAXED);
    return 0;
}

/*
 * svt_hip_bind_tpl.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 3c of INTEGRATION.md):
 * the TPL dispenser of a whole picture through svt_hip_tpl_dispenser_frame instead of one tpl_mc_flow_dispenser_sb_generic call
 * per 64x64 block (Source/Lib/Codec/src_ops_process.c:519-1207, called from svt_aom_tpl_disp_kernel :1964).
 * The patch puts `if (svt_hip_bind_tpl_sb(pcs, frame_idx, sb_index, qIndex))` in front of the reference's per-block call: the
 * first block of a picture that arrives runs the whole picture on the GPU (the source, the references' source and TPL
 * reconstruction pictures from the device-resident mirrors of svt_hip_bind_dev.h, the picture's ME results uploaded; the TPL
 * reconstruction, TplStats and TplSrcStats downloaded into host staging, the kernel's status word checked, and only then stored
 * where the reference's loop stores them, result_model_store's grids included); the other blocks wait and return.
 * Covered: the configuration of include/svt_hip_tpl.h (the tpl level of presets M7 ... M9); anything else returns 1 for every block
 * of the picture and the reference's own loop runs.  Active with `--asm hip` and SVTAV1_HIP_TIERB_TPL=1.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "definitions.h"
#include "encode_context.h"
#include "pcs.h"
#include "sequence_control_set.h"

#include "svt_hip.h"
#include "svt_hip_tpl.h"
#include "svt_hip_bind.h"
#include "svt_hip_bind_dev.h"

static int32_t (*p_tpl_frame)(const SvtHipTplFrameJob *, void *);
static uint64_t (*p_ws_bytes)(uint32_t, uint32_t);
static uint64_t (*p_status_offset)(uint32_t, uint32_t);
static int           g_active;
static unsigned long g_pictures;

static void report(void) { fprintf(stderr, "svt_hip_bind_tpl: %lu pictures through svt_hip_tpl_dispenser_frame\n", g_pictures); }

void svt_hip_bind_tpl_setup(void *(*sym)(const char *)) {
    p_tpl_frame     = (int32_t(*)(const SvtHipTplFrameJob *, void *))sym("svt_hip_tpl_dispenser_frame");
    p_ws_bytes      = (uint64_t(*)(uint32_t, uint32_t))sym("svt_hip_tpl_workspace_bytes");
    p_status_offset = (uint64_t(*)(uint32_t, uint32_t))sym("svt_hip_tpl_status_offset");
    g_active        = hd_env_on("SVTAV1_HIP_TIERB_TPL") && g_hd.ok && p_tpl_frame && p_ws_bytes && p_status_offset;
    if (g_active)
        atexit(report);
}

static HdOnceTable g_tab;

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t luma_bytes(const EbPictureBufferDesc *d) { return (size_t)d->stride_y * (d->height + 2u * d->org_y); }
static int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

static int run_picture(PictureParentControlSet *pcs, int32_t frame_idx, int32_t qindex) {
    SequenceControlSet *scs = pcs->scs;
    EncodeContext      *enc = scs->enc_ctx;
    const TplControls  *tc  = &pcs->tpl_ctrls;
    EbPictureBufferDesc *src = pcs->enhanced_pic, *recon = enc->mc_flow_rec_picture_buffer[frame_idx];
    /* tpl level 4 (16x16 blocks), level 3 (the same with the quarter-pel refinement of tpl_subpel_search) or level 5 (32x32 blocks, TX_32X8
     * on every 4th row) */
    const int lvl5 = tc->dispenser_search_level == 1 && tc->subsample_tx == 2;
    const int qpel = tc->subpel_depth == QUARTER_PEL;
    /* (the clamp of svt_aom_enc_make_inter_predictor keeps a compensated block within 16 + 4 samples of the picture and the taps add three:
     * the 32 samples of padding the full-pel path needs cover it) */
    if (qpel && (lvl5 || tc->subpel_diag_refinement < 4 ||
                 pcs->av1_cm->mi_rows != (int32_t)(pcs->aligned_height >> 2) || pcs->av1_cm->mi_cols != (int32_t)(pcs->aligned_width >> 2))) {
        static int said;
        if (!__atomic_exchange_n(&said, 1, __ATOMIC_RELAXED))
            fprintf(stderr, "svt_hip_bind_tpl: level 3 declined: lvl5 %d diag %d mi %d x %d vs %d x %d\n", lvl5, tc->subpel_diag_refinement, pcs->av1_cm->mi_rows,
                    pcs->av1_cm->mi_cols, pcs->aligned_height >> 2, pcs->aligned_width >> 2);
        return 1;
    }
    if (!(lvl5 || (tc->dispenser_search_level == 0 && tc->subsample_tx == 0)) || tc->intra_mode_end != DC_PRED || !tc->use_sad_in_src_search ||
        (tc->subpel_depth != FULL_PEL && !qpel) || tc->compute_rate || !scs->in_loop_ois || src->org_x < 32 || src->org_y < 32 || recon->org_x < 32 ||
        recon->org_y < 32 || (tc->synth_blk_size != 8 && tc->synth_blk_size != 16 && tc->synth_blk_size != 32) ||
        scs->static_config.tile_rows || scs->static_config.tile_columns) {
        static int said2;
        if (!__atomic_exchange_n(&said2, 1, __ATOMIC_RELAXED) && getenv("SVTAV1_E2E_TRACE_TPL"))
            fprintf(stderr, "svt_hip_bind_tpl: declined: search level %d subsample_tx %d intra_mode_end %d sad %d subpel %d rate %d ois %d synth %d\n",
                    tc->dispenser_search_level, tc->subsample_tx, tc->intra_mode_end, tc->use_sad_in_src_search, tc->subpel_depth, tc->compute_rate, scs->in_loop_ois,
                    tc->synth_blk_size);
        return 1;
    }
    SvtHipTplFrameJob *job = (SvtHipTplFrameJob *)calloc(1, sizeof(*job));
    if (!job)
        return 1;
    const uint32_t W = src->width, H = src->height, aw = pcs->aligned_width, ah = pcs->aligned_height;
    const uint32_t a16 = (aw + 15) >> 4, rows16 = (ah + 15) >> 4, nb = pcs->b64_total_count;
    if (lvl5 && tc->synth_blk_size == 8) { /* (no preset: the synthesizer grid is 16 or 32, initial_rc_process.c:198-232) */
        free(job);
        return 1;
    }
    MotionEstimationData *med = pcs->pa_me_data;
    const uint32_t stored = pcs->enable_me_16x16 ? (pcs->enable_me_8x8 ? 85u : 21u) : 5u;
    const size_t   n_mv = (size_t)nb * stored * med->max_refs * 4, n_cand = (size_t)nb * stored * med->max_cand, n_cnt = (size_t)nb * stored;
    const size_t   n_stats = (size_t)a16 * rows16 * sizeof(SvtHipTplStats), n_sst = (size_t)a16 * rows16 * sizeof(SvtHipTplSrcStats);
    const uint64_t wsb = p_ws_bytes(W, H);
    EbPictureBufferDesc *rsrc[2][4] = {{0}}, *rrec[2][4] = {{0}};
    uint64_t             rrec_poc[2][4] = {{0}};
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            EbPictureBufferDesc *p = (EbPictureBufferDesc *)pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_ptr;
            if (!p || !p->buffer_y)
                continue;
            rsrc[l][r] = p;
            if (pcs->tpl_data.ref_in_slide_window[l][r]) {
                const uint64_t poc = pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_number;
                uint32_t       k = 0;
                while (k < MAX_TPL_LA_SW && enc->poc_map_idx[k] != poc) k++;
                if (k < MAX_TPL_LA_SW && enc->mc_flow_rec_picture_buffer[k])
                    rrec[l][r] = enc->mc_flow_rec_picture_buffer[k], rrec_poc[l][r] = poc;
            }
        }
    /* host-side staging of the scattered inputs: the ME results live in one MeSbResults per block */
    uint8_t *h_mv = (uint8_t *)malloc(n_mv), *h_cand = (uint8_t *)malloc(n_cand), *h_cnt = (uint8_t *)malloc(n_cnt);
    SvtHipTplSrcStats *h_sst = (SvtHipTplSrcStats *)calloc((size_t)a16 * rows16, sizeof(*h_sst));
    SvtHipTplStats    *h_st  = (SvtHipTplStats *)calloc((size_t)a16 * rows16, sizeof(*h_st));
    for (uint32_t i = 0; i < nb; i++) {
        const MeSbResults *res = med->me_results[i];
        memcpy(h_mv + (size_t)i * stored * med->max_refs * 4, res->me_mv_array, (size_t)stored * med->max_refs * 4);
        memcpy(h_cand + (size_t)i * stored * med->max_cand, res->me_candidate_array, (size_t)stored * med->max_cand);
        memcpy(h_cnt + (size_t)i * stored, res->total_me_candidate_index, stored);
    }
    for (size_t i = 0; i < (size_t)a16 * rows16; i++) {
        const TplSrcStats *q = &med->tpl_src_stats_buffer[i];
        h_sst[i].srcrf_dist = q->srcrf_dist, h_sst[i].srcrf_rate = q->srcrf_rate, h_sst[i].ref_frame_poc = q->ref_frame_poc;
        h_sst[i].mv_row = q->mv.row, h_sst[i].mv_col = q->mv.col, h_sst[i].best_rf_idx = q->best_rf_idx, h_sst[i].best_mode = q->best_mode;
        h_sst[i].best_intra_mode = (uint8_t)q->best_intra_mode;
    }
    /* device side: the picture's own scratch (ME results, statistics, reconstruction, workspace) in one allocation; the source
     * picture, the references' source pictures and their TPL reconstructions from the device-resident mirrors */
    const size_t n_rec = luma_bytes(recon);
    const size_t need  = al256(n_rec + 64) + al256(n_mv + 64) + al256(n_cand + 64) + al256(n_cnt + 64) + al256(n_stats) + al256(n_sst + 64) + al256(wsb);
    uint8_t     *dev   = hd_alloc(need);
    int          rc    = dev ? 0 : -1;
    size_t       off   = 0;
    const void  *pinned[1 + 2 * 2 * 4];
    int          n_pinned = 0;
#define PUT(dst, hostp, n) (dst = dev + off, off += al256((n) + 64), hd_upload(dst, hostp, n))
#define MIRROR(dst, hostp, n, tag) ((dst = hd_mirror_get(hostp, n, tag)) ? (pinned[n_pinned++] = (hostp), 0) : -1)
    uint8_t *d_src = NULL, *d_rec = NULL, *d_mv = NULL, *d_cand = NULL, *d_cnt = NULL, *d_st = NULL, *d_sst = NULL, *d_p = NULL;
    if (rc == 0)
        rc = MIRROR(d_src, src->buffer_y, luma_bytes(src), HD_TAG(pcs->picture_number, HD_ST_FILTERED));
    if (rc == 0) /* the reconstruction buffer keeps what the kernel does not write (its padding) */
        rc = PUT(d_rec, recon->buffer_y, n_rec) | PUT(d_mv, h_mv, n_mv) | PUT(d_cand, h_cand, n_cand) | PUT(d_cnt, h_cnt, n_cnt) | PUT(d_sst, h_sst, n_sst);
    if (rc == 0) {
        d_st = dev + off, off += al256(n_stats);
        rc   = g_hd.memset_(d_st, 0, n_stats, NULL);
    }
    job->src.buf = d_src, job->src.stride = src->stride_y, job->src.org_x = src->org_x, job->src.org_y = src->org_y, job->src.width = (uint16_t)W,
    job->src.height = (uint16_t)H;
    job->recon = job->src, job->recon.buf = d_rec, job->recon.stride = recon->stride_y, job->recon.org_x = recon->org_x, job->recon.org_y = recon->org_y;
    for (int l = 0; rc == 0 && l < 2; l++)
        for (int r = 0; rc == 0 && r < 4; r++) {
            if (!rsrc[l][r])
                continue;
            SvtHipTplRef        *f = &job->ref[l][r];
            EbPictureBufferDesc *p = rsrc[l][r];
            const uint64_t       poc = pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_number;
            rc                       = MIRROR(d_p, p->buffer_y, luma_bytes(p), HD_TAG(poc, HD_ST_FILTERED));
            if (rc != 0)
                break;
            f->src = d_p + (size_t)p->org_y * p->stride_y + p->org_x, f->src_stride = p->stride_y;
            f->recon = f->src, f->recon_stride = f->src_stride;
            if (rrec[l][r]) {
                EbPictureBufferDesc *q = rrec[l][r];
                rc                     = MIRROR(d_p, q->buffer_y, luma_bytes(q), HD_TAG(rrec_poc[l][r], HD_ST_TPL_RECON));
                if (rc != 0)
                    break;
                f->recon = d_p + (size_t)q->org_y * q->stride_y + q->org_x, f->recon_stride = q->stride_y;
            }
            f->picture_number = poc;
            f->max_width = p->max_width, f->max_height = p->max_height;
            const int32_t grp = pcs->tpl_data.ref_tpl_group_idx[l][r];
            f->usable = !(grp > 0 && pcs->tpl_data.base_pcs->tpl_valid_pic[grp] == 0);
        }
    uint8_t *h_rec = NULL;
    if (rc == 0) {
        job->me_mv_array = (const uint32_t *)d_mv, job->me_candidate_array = d_cand, job->total_me_candidate_index = d_cnt;
        job->max_cand = med->max_cand, job->max_refs = med->max_refs, job->max_l0 = med->max_l0;
        job->enable_me_16x16 = pcs->enable_me_16x16, job->stored_pus = (uint8_t)stored;
        job->pf_shape = (uint8_t)tc->pf_shape;
        job->disable_intra_pred = tc->disable_intra_pred_nref && (pcs->temporal_layer_index == pcs->hierarchical_levels);
        job->is_ref = pcs->tpl_data.is_ref, job->i_slice = pcs->slice_type == I_SLICE, job->tpl_i_slice = pcs->tpl_data.tpl_slice_type == I_SLICE;
        job->src_data_ready = pcs->tpl_src_data_ready, job->store_src_stats = scs->tpl_lad_mg > 0;
        /* level 4: one cell per 16x16 block, the synthesizer's grid is filled below; level 5: the library writes that grid itself */
        job->synth_blk_size = lvl5 ? tc->synth_blk_size : 16, job->blk_size = lvl5 ? 32 : 16, job->subsample_tx = lvl5 ? 2 : 0;
        job->quarter_pel = (uint8_t)qpel;
        for (int i = 0; i < 2; i++) {
            job->round_fp[i] = enc->quants_8bit.y_round_fp[qindex][i], job->quant_fp[i] = enc->quants_8bit.y_quant_fp[qindex][i];
            job->dequant[i] = enc->deq_8bit.y_dequant_qtx[qindex][i];
        }
        job->stats = (SvtHipTplStats *)d_st, job->src_stats = (SvtHipTplSrcStats *)d_sst;
        job->workspace = dev + off, job->workspace_bytes = wsb;
        rc = p_tpl_frame(job, NULL);
    }
    if (rc == 0) {
        /* everything comes back into host staging first; the status word says whether a dependency wait inside the kernel ran into
         * its bound (the kernel then went on with unsynchronised neighbours: the results are unusable and the CPU loop runs) */
        uint32_t status = 1;
        h_rec = (uint8_t *)hd_host_alloc(n_rec);
        rc = h_rec ? (hd_download(h_rec, d_rec, n_rec) | hd_download(h_st, d_st, n_stats) | hd_download(h_sst, d_sst, n_sst) |
                      hd_download(&status, (uint8_t *)job->workspace + p_status_offset(W, H), 4) | hd_sync())
                   : -1;
        if (rc == 0 && status != 0) {
            fprintf(stderr, "svt_hip_bind_tpl: picture %llu: a dependency wait of the kernel timed out\n", (unsigned long long)pcs->picture_number);
            rc = -1;
        }
        if (rc == 0)
            memcpy(recon->buffer_y, h_rec, n_rec);
    } else {
        hd_sync();
    }
    for (int i = 0; i < n_pinned; i++) hd_mirror_unpin(pinned[i]);
#undef PUT
#undef MIRROR
    if (rc == 0) {
        /* result_model_store (src_ops_process.c:266-340) from the one-cell-per-block grid, and the source-based statistics, block by
         * block in the reference's order (64x64 blocks raster, blocks in z-order inside: with 16x16 blocks a 32x32 synthesizer cell
         * keeps its last block) */
        const uint32_t bw64 = (aw + 63) / 64, s32 = (aw + 31) / 32;
        for (uint32_t sb = 0; sb < nb; sb++) {
            /* level 5 dispenses complete 64x64 blocks as 32x32 blocks and incomplete ones as 16x16 blocks (:2043-2051) */
            const B64Geom *g  = &scs->b64_geom[sb];
            const uint32_t bs = (lvl5 && g->width == 64 && g->height == 64) ? 32 : 16, nz = bs == 32 ? 4 : 16;
            for (uint32_t z = 0; z < nz; z++) {
                const uint32_t bx = bs == 32 ? (z & 1) : ((z & 1) | ((z >> 2) & 1) << 1), by = bs == 32 ? (z >> 1) : (((z >> 1) & 1) | ((z >> 3) & 1) << 1);
                const uint32_t x = (sb % bw64) * 64 + bx * bs, y = (sb / bw64) * 64 + by * bs;
                if (x + (bs >> 1) > W || y + (bs >> 1) > H)
                    continue;
                if (lvl5) { /* the cells this block stores, as they are in the library's grid */
                    const uint32_t cell = tc->synth_blk_size, stride = cell == 32 ? s32 : a16, per = bs / cell ? bs / cell : 1;
                    for (uint32_t cy = 0; cy < per; cy++)
                        for (uint32_t cx = 0; cx < per; cx++) {
                            const size_t          idx = (size_t)(y / cell + cy) * stride + x / cell + cx;
                            const SvtHipTplStats *s   = &h_st[idx];
                            TplStats             *t   = med->tpl_stats[idx];
                            memset(t, 0, sizeof(*t));
                            t->srcrf_dist = s->srcrf_dist, t->recrf_dist = s->recrf_dist, t->srcrf_rate = s->srcrf_rate, t->recrf_rate = s->recrf_rate;
                            t->mv.row = s->mv_row, t->mv.col = s->mv_col, t->ref_frame_poc = s->ref_frame_poc;
                        }
                } else {
                const SvtHipTplStats *s = &h_st[(size_t)(y >> 4) * a16 + (x >> 4)];
                TplStats              t;
                memset(&t, 0, sizeof(t));
                t.srcrf_dist = s->srcrf_dist, t.recrf_dist = s->recrf_dist, t.srcrf_rate = s->srcrf_rate, t.recrf_rate = s->recrf_rate;
                t.mv.row = s->mv_row, t.mv.col = s->mv_col, t.ref_frame_poc = s->ref_frame_poc;
                if (tc->synth_blk_size == 32) {
                    *med->tpl_stats[(size_t)(y >> 5) * s32 + (x >> 5)] = t;
                } else if (tc->synth_blk_size == 16) {
                    *med->tpl_stats[(size_t)(y >> 4) * a16 + (x >> 4)] = t;
                } else {
                    const uint32_t stride = a16 << 1;
                    t.srcrf_dist = max64(1, t.srcrf_dist / 4), t.recrf_dist = max64(1, t.recrf_dist / 4);
                    t.srcrf_rate = max64(1, t.srcrf_rate / 4), t.recrf_rate = max64(1, t.recrf_rate / 4);
                    TplStats **d = &med->tpl_stats[(size_t)(y >> 3) * stride + (x >> 3)];
                    *d[0] = t, *d[1] = t, *d[stride] = t, *d[stride + 1] = t;
                }
                }
                if (!job->src_data_ready && job->store_src_stats) {
                    const SvtHipTplSrcStats *q = &h_sst[(size_t)(y >> 4) * a16 + (x >> 4)];
                    TplSrcStats             *o = &med->tpl_src_stats_buffer[(size_t)(y >> 4) * a16 + (x >> 4)];
                    o->srcrf_dist = q->srcrf_dist, o->srcrf_rate = q->srcrf_rate, o->ref_frame_poc = q->ref_frame_poc;
                    o->mv.row = q->mv_row, o->mv.col = q->mv_col, o->best_rf_idx = q->best_rf_idx, o->best_mode = q->best_mode;
                    o->best_intra_mode = (PredictionMode)q->best_intra_mode;
                }
            }
        }
    }
    if (rc != 0)
        fprintf(stderr, "svt_hip_bind_tpl: picture %llu stays on the CPU (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
    else
        hd_count_picture();
    hd_free(dev);
    hd_host_free(h_rec), free(h_mv), free(h_cand), free(h_cnt), free(h_sst), free(h_st), free(job);
    return rc != 0;
}

/* Returns 0 when the picture's dispenser ran on the GPU (the caller skips its per-block call), 1 when the caller must run it. */
int svt_hip_bind_tpl_sb(PictureParentControlSet *pcs, int32_t frame_idx, uint32_t sb_index, int32_t qindex) {
    (void)sb_index;
    if (!g_hd.ok)
        return 1;
    /* this picture's TPL reconstruction is rewritten now (GPU or CPU; a picture is dispensed again when it belongs to the next TPL
     * group too): the first block forgets the mirror other pictures' dispensers used */
    int     first;
    HdOnce *once = hd_once_enter(&g_tab, pcs, ((uint64_t)pcs->picture_number << 8) | (uint32_t)(frame_idx & 0xff), pcs->b64_total_count, &first);
    if (!once)
        return 1;
    if (first) {
        EbPictureBufferDesc *recon = pcs->scs->enc_ctx->mc_flow_rec_picture_buffer[frame_idx];
        if (recon)
            hd_mirror_drop(recon->buffer_y);
        const uint64_t t0 = hd_now_ns();
        const int      rc = g_active ? run_picture(pcs, frame_idx, qindex) : 1;
        if (g_active)
            hd_timer_add("tpl_picture", hd_now_ns() - t0);
        if (rc == 0)
            __atomic_add_fetch(&g_pictures, 1, __ATOMIC_RELAXED);
        hd_once_done(once, rc == 0, NULL);
    }
    const int on_gpu = hd_once_ok(once);
    hd_once_release(&g_tab, once, NULL);
    return on_gpu ? 0 : 1;
}

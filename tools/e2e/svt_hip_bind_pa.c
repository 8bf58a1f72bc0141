/*
 * svt_hip_bind_pa.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 2a of INTEGRATION.md): the
 * picture-analysis kernel's per-picture luma work through the batched entry points of include/svt_hip_me.h:
 *
 *   svt_hip_bind_pa_pyramid    in front of the body of svt_aom_downsample_filtering_input_picture (pic_analysis_process.c:1922-1979,
 *                              called at :2126) -> svt_hip_pyramid_frame: 1/4 and 1/16 decimation with their padding, one call.
 *   svt_hip_bind_pa_variance   in front of the b64 loop of compute_picture_spatial_statistics (:1532-1553, called through
 *                              svt_aom_gathering_picture_statistics at :2137) -> svt_hip_variance_frame: the 85 block variances of
 *                              every 64x64 block.
 *
 * The picture is uploaded ONCE here, into the device-resident mirrors of svt_hip_bind_dev.h; the two decimated planes are born
 * on the device and become the mirrors of the host planes they are downloaded into — the open-loop ME of this picture and of
 * every picture that references it, the temporal filter and the TPL dispenser then find all of it resident (no further upload).
 * Returns 0 when the GPU did the work, 1 when the caller must run the reference's code.  Active with `--asm hip` and
 * SVTAV1_HIP_TIERB_PA=1.
 */
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

static int32_t (*p_pyramid)(const SvtHipPlane8 *, const SvtHipPlane8 *, const SvtHipPlane8 *, int32_t, void *);
static int32_t (*p_variance)(const SvtHipPlane8 *, uint16_t *, uint64_t *, int32_t, void *);
static int           g_active;
static unsigned long g_n_pyr, g_n_var;

static void report(void) {
    fprintf(stderr, "svt_hip_bind_pa: %lu pyramids, %lu variance maps through svt_hip_pyramid_frame / svt_hip_variance_frame\n", g_n_pyr, g_n_var);
}

void svt_hip_bind_pa_setup(void *(*sym)(const char *)) {
    p_pyramid  = (int32_t(*)(const SvtHipPlane8 *, const SvtHipPlane8 *, const SvtHipPlane8 *, int32_t, void *))sym("svt_hip_pyramid_frame");
    p_variance = (int32_t(*)(const SvtHipPlane8 *, uint16_t *, uint64_t *, int32_t, void *))sym("svt_hip_variance_frame");
    g_active   = hd_env_on("SVTAV1_HIP_TIERB_PA") && g_hd.ok && p_pyramid && p_variance;
    if (g_active)
        atexit(report);
}

static size_t plane_bytes(const EbPictureBufferDesc *d) { return (size_t)d->stride_y * (d->height + 2u * d->org_y); }
static void   plane_of(SvtHipPlane8 *p, const EbPictureBufferDesc *d, uint8_t *dev) {
    p->buf = dev, p->stride = d->stride_y, p->org_x = d->org_x, p->org_y = d->org_y, p->width = d->width, p->height = d->height;
}
/* paddings the library asks for (include/svt_hip_me.h: >= 64 / 32 / 16, the decimated planes exactly half / a quarter) */
static int geometry_ok(const EbPictureBufferDesc *f, const EbPictureBufferDesc *q, const EbPictureBufferDesc *s) {
    return f->org_x >= 64 && f->org_y >= 64 && q->org_x >= 32 && q->org_y >= 32 && s->org_x >= 16 && s->org_y >= 16 && q->org_x == q->org_y &&
        s->org_x == s->org_y /* the reference's destination offset uses org_x twice (:1934, :1953) */ && q->width == f->width >> 1 &&
        q->height == f->height >> 1 && s->width == f->width >> 2 && s->height == f->height >> 2;
}

static int pa_pyramid_impl(PictureParentControlSet *pcs, EbPictureBufferDesc *full, EbPictureBufferDesc *quarter, EbPictureBufferDesc *sixteenth);
int svt_hip_bind_pa_pyramid(PictureParentControlSet *pcs, EbPictureBufferDesc *full, EbPictureBufferDesc *quarter, EbPictureBufferDesc *sixteenth) {
    if (!g_active)
        return 1;
    const uint64_t t0 = hd_now_ns();
    const int      rc = pa_pyramid_impl(pcs, full, quarter, sixteenth);
    hd_timer_add("pa_pyramid", hd_now_ns() - t0);
    return rc;
}
static int pa_pyramid_impl(PictureParentControlSet *pcs, EbPictureBufferDesc *full, EbPictureBufferDesc *quarter, EbPictureBufferDesc *sixteenth) {
    if (!g_active || !(pcs->enable_hme_flag || pcs->tf_enable_hme_flag) || !(pcs->enable_hme_level0_flag || pcs->tf_enable_hme_level0_flag) ||
        !geometry_ok(full, quarter, sixteenth))
        return 1;
    const int      level1 = pcs->enable_hme_level1_flag || pcs->tf_enable_hme_level1_flag;
    const uint64_t tag    = HD_TAG(pcs->picture_number, HD_ST_FILTERED);
    const size_t   nq = plane_bytes(quarter), ns = plane_bytes(sixteenth);
    const uint64_t t_a = hd_now_ns();
    uint8_t       *d_f = hd_mirror_get(full->buffer_y, plane_bytes(full), tag);
    if (!d_f)
        return 1;
    hd_timer_add("pa_pyramid.1_source_mirror", hd_now_ns() - t_a);
    const uint64_t t_b = hd_now_ns();
    /* the decimated planes are produced on the device: their buffers become the mirrors of the host planes */
    uint8_t *d_q = level1 ? hd_mirror_new(quarter->buffer_y, nq, tag) : hd_alloc(nq + 256);
    uint8_t *d_s = hd_mirror_new(sixteenth->buffer_y, ns, tag);
    uint8_t *h_q = level1 ? (uint8_t *)hd_host_alloc(nq) : NULL, *h_s = (uint8_t *)hd_host_alloc(ns);
    int      rc  = (d_q && d_s && h_s && (h_q || !level1)) ? 0 : -1;
    hd_timer_add("pa_pyramid.2_alloc", hd_now_ns() - t_b);
    const uint64_t t_c = hd_now_ns();
    if (rc == 0) {
        SvtHipPlane8 pf, pq, ps;
        plane_of(&pf, full, d_f), plane_of(&pq, quarter, d_q), plane_of(&ps, sixteenth, d_s);
        /* a plane keeps the bytes the kernel does not write (row tails behind the padding): start from the host's */
        rc = (level1 ? hd_upload(d_q, quarter->buffer_y, nq) : 0) | hd_upload(d_s, sixteenth->buffer_y, ns);
        if (rc == 0)
            rc = p_pyramid(&pf, &pq, &ps, level1, NULL);
        if (rc == 0)
            rc = (level1 ? hd_download(h_q, d_q, nq) : 0) | hd_download(h_s, d_s, ns);
    }
    rc |= hd_sync();
    hd_timer_add("pa_pyramid.3_upload_kernel_download", hd_now_ns() - t_c);
    const uint64_t t_d = hd_now_ns();
    if (rc == 0) {
        if (level1)
            memcpy(quarter->buffer_y, h_q, nq);
        memcpy(sixteenth->buffer_y, h_s, ns);
    }
    hd_timer_add("pa_pyramid.4_copy_out", hd_now_ns() - t_d);
    hd_mirror_unpin(full->buffer_y);
    if (level1 && d_q) {
        if (rc != 0)
            hd_mirror_drop(quarter->buffer_y);
        hd_mirror_unpin(quarter->buffer_y);
    } else if (!level1) {
        hd_free(d_q);
    }
    if (d_s) {
        if (rc != 0)
            hd_mirror_drop(sixteenth->buffer_y);
        hd_mirror_unpin(sixteenth->buffer_y);
    }
    hd_host_free(h_q), hd_host_free(h_s);
    if (rc != 0) {
        fprintf(stderr, "svt_hip_bind_pa: pyramid of picture %llu stays on the CPU (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
        return 1;
    }
    __atomic_add_fetch(&g_n_pyr, 1, __ATOMIC_RELAXED);
    hd_count_picture();
    return 0;
}

static int pa_variance_impl(SequenceControlSet *scs, PictureParentControlSet *pcs, EbPictureBufferDesc *full);
int svt_hip_bind_pa_variance(SequenceControlSet *scs, PictureParentControlSet *pcs, EbPictureBufferDesc *full) {
    if (!g_active)
        return 1;
    const uint64_t t0 = hd_now_ns();
    const int      rc = pa_variance_impl(scs, pcs, full);
    hd_timer_add("pa_variance", hd_now_ns() - t0);
    return rc;
}
static int pa_variance_impl(SequenceControlSet *scs, PictureParentControlSet *pcs, EbPictureBufferDesc *full) {
    if (!g_active || full->org_x < 64 || full->org_y < 64)
        return 1;
    const uint32_t nb = pcs->b64_total_count;
    if (nb != ((uint32_t)(pcs->aligned_width + 63) / 64) * ((uint32_t)(pcs->aligned_height + 63) / 64))
        return 1;
    uint8_t *d_f = hd_mirror_get(full->buffer_y, plane_bytes(full), HD_TAG(pcs->picture_number, HD_ST_FILTERED));
    if (!d_f)
        return 1;
    const size_t n   = (size_t)nb * 85 * sizeof(uint16_t);
    uint8_t     *dev = hd_alloc(n + 256);
    uint16_t    *h   = (uint16_t *)malloc(n);
    int          rc  = (dev && h) ? 0 : -1;
    if (rc == 0) {
        SvtHipPlane8 pf;
        plane_of(&pf, full, d_f);
        pf.width = pcs->aligned_width, pf.height = pcs->aligned_height; /* the b64 grid of the reference's loop (b64_geom) */
        rc = p_variance(&pf, (uint16_t *)dev, NULL, scs->block_mean_calc_prec == BLOCK_MEAN_PREC_FULL, NULL);
    }
    if (rc == 0)
        rc = hd_download(h, dev, n);
    rc |= hd_sync();
    hd_mirror_unpin(full->buffer_y);
    if (rc == 0) {
        /* what compute_block_mean_compute_variance stores (:1111-1380): all 85 with adaptive quantisation 1 or variance_octile, else
         * only the 64x64 entry; then the picture average (:1547-1550) */
        const int all       = scs->static_config.enable_adaptive_quantization == 1 || scs->static_config.variance_octile;
        uint64_t  pic_total = 0;
        for (uint32_t b = 0; b < nb; b++) {
            if (all)
                memcpy(pcs->variance[b], h + (size_t)b * 85, 85 * sizeof(uint16_t));
            else
                pcs->variance[b][ME_TIER_ZERO_PU_64x64] = h[(size_t)b * 85 + ME_TIER_ZERO_PU_64x64];
            pic_total += pcs->variance[b][RASTER_SCAN_CU_INDEX_64x64];
        }
        pcs->pic_avg_variance = (uint16_t)(pic_total / nb);
    }
    hd_free(dev), free(h);
    if (rc != 0) {
        fprintf(stderr, "svt_hip_bind_pa: variance of picture %llu stays on the CPU (%s)\n", (unsigned long long)pcs->picture_number, hd_error());
        return 1;
    }
    __atomic_add_fetch(&g_n_var, 1, __ATOMIC_RELAXED);
    return 0;
}

/*
 * oracle/ref_harness_lr.c — TEST INFRASTRUCTURE.  Compiled ONLY into oracle/_ref/libsvtref.so.  Our glue around the REAL
 * svt_av1_loop_restoration_filter_unit (restoration.c:1067) and svt_extend_frame (:197): builds the reference's structures
 * (RestorationUnitInfo, RestorationStripeBoundaries, RestorationLineBuffers, limits, tile rectangle) from the flat
 * SvtHipLrPlane of include/svt_hip_lf.h and walks the restoration units in the order of foreach_rest_unit_in_tile (:1250).
 * No reference source is copied here.
 */
#include <stdlib.h>
#include <string.h>

#include "definitions.h"
#include "restoration.h"

#include "../include/svt_hip_lf.h"

#define REF_API __attribute__((visibility("default")))

void svt_extend_frame(uint8_t *data, int32_t width, int32_t height, int32_t stride, int32_t border_horz, int32_t border_vert,
                      int32_t highbd);
void svt_av1_loop_restoration_filter_unit(uint8_t need_bounadaries, const RestorationTileLimits *limits, const RestorationUnitInfo *rui,
                                          const RestorationStripeBoundaries *rsb, RestorationLineBuffers *rlbs,
                                          const Av1PixelRect *tile_rect, int32_t tile_stripe0, int32_t ss_x, int32_t ss_y, int32_t highbd,
                                          int32_t bit_depth, uint8_t *data8, int32_t stride, uint8_t *dst8, int32_t dst_stride,
                                          int32_t *tmpbuf, int32_t optimized_lr);
void ref_init(void);

REF_API int ref_restoration_filter_frame(const SvtHipLrPlane *planes, uint32_t n_planes) {
    ref_init();
    int32_t                *tmpbuf = malloc(RESTORATION_TMPBUF_SIZE);
    RestorationLineBuffers *rlbs   = malloc(sizeof(*rlbs));
    for (uint32_t p = 0; p < n_planes; p++) {
        const SvtHipLrPlane *pl = &planes[p];
        const int W = (int)pl->width, H = (int)pl->height, hbd = pl->is_16bit, PAD = 16;
        /* a private, extended copy of the CDEF output (the reference extends the frame in place, :1219-1225) */
        const int  stride = W + 2 * PAD;
        uint8_t   *copy   = calloc((size_t)(H + 2 * PAD) * stride, 1 << hbd);
        uint8_t   *org    = copy + (((size_t)PAD * stride + PAD) << hbd);
        for (int y = 0; y < H; y++)
            memcpy(org + (((size_t)y * stride) << hbd), (const uint8_t *)pl->src + (((size_t)y * pl->src_stride) << hbd), (size_t)W << hbd);
        uint8_t *data8 = hbd ? CONVERT_TO_BYTEPTR(org) : org;
        uint8_t *dst8  = hbd ? CONVERT_TO_BYTEPTR(pl->dst) : (uint8_t *)pl->dst;
        svt_extend_frame(data8, W, H, stride, RESTORATION_BORDER, RESTORATION_BORDER, hbd);
        RestorationStripeBoundaries rsb;
        memset(&rsb, 0, sizeof(rsb));
        rsb.stripe_boundary_above  = (uint8_t *)pl->boundary_above;
        rsb.stripe_boundary_below  = (uint8_t *)pl->boundary_below;
        rsb.stripe_boundary_stride = (int32_t)pl->boundary_stride;
        const Av1PixelRect tile = {0, 0, W, H};
        const int          us = (int)pl->unit_size, ext = us * 3 / 2, voff = RESTORATION_UNIT_OFFSET >> pl->ss_y;
        int                y0 = 0, i = 0;
        while (y0 < H) { /* foreach_rest_unit_in_tile (:1260-1293) */
            const int             rem_h = H - y0, h = rem_h < ext ? rem_h : us;
            RestorationTileLimits lim;
            lim.v_start = y0 - voff > 0 ? y0 - voff : 0;
            lim.v_end   = y0 + h < H ? y0 + h - voff : y0 + h;
            int x0 = 0, j = 0;
            while (x0 < W) {
                const int rem_w = W - x0, w = rem_w < ext ? rem_w : us;
                lim.h_start = x0, lim.h_end = x0 + w;
                const SvtHipLrUnit *u = &pl->units[i * pl->horz_units + j];
                RestorationUnitInfo rui;
                memset(&rui, 0, sizeof(rui));
                rui.restoration_type = (RestorationType)u->restoration_type;
                memcpy(rui.wiener_info.hfilter, u->hfilter, sizeof(u->hfilter));
                memcpy(rui.wiener_info.vfilter, u->vfilter, sizeof(u->vfilter));
                rui.sgrproj_info.ep     = u->ep;
                rui.sgrproj_info.xqd[0] = u->xqd[0], rui.sgrproj_info.xqd[1] = u->xqd[1];
                svt_av1_loop_restoration_filter_unit(1, &lim, &rui, &rsb, rlbs, &tile, 0, pl->ss_x, pl->ss_y, hbd, pl->bit_depth, data8, stride,
                                                     dst8, (int32_t)pl->dst_stride, tmpbuf, (int32_t)pl->optimized_lr);
                x0 += w, j++;
            }
            y0 += h, i++;
        }
        free(copy);
    }
    free(tmpbuf), free(rlbs);
    return 0;
}

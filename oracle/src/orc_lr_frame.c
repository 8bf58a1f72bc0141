/* oracle/src/orc_lr_frame.c — TEST INFRASTRUCTURE.  CPU restatement of the frame-level loop restoration:
 * svt_av1_loop_restoration_filter_frame (restoration.c:1179-1248) -> foreach_rest_unit_in_tile (:1250-1294) ->
 * svt_av1_loop_restoration_filter_unit (:1067-1147) with its processing-stripe boundary rules
 * (svt_aom_get_stripe_boundary_info :257-276, svt_aom_setup_processing_stripe_boundary :288-384).  Instead of patching the
 * boundary rows into the picture and back, every stripe's input is assembled in a private tile; the stripe filters are
 * the leaf functions restated in orc_wiener.c / orc_sgr.c.  Pinned against the real functions by tests/test_lr_frame_oracle.py. */
#include <stdlib.h>
#include <string.h>

#include "orc_lf.h"

#define RU_OFFSET 8  /* RESTORATION_UNIT_OFFSET */
#define PROC_UNIT 64 /* RESTORATION_PROC_UNIT_SIZE */
#define BORDER 3     /* RESTORATION_BORDER */

static int32_t px(const void *p, ptrdiff_t i, int is16) { return is16 ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i]; }

/* get_conv_params_wiener (restoration.c:49-73) */
static void wiener_rounds(int bd, int *r0, int *r1) {
    *r0 = 3, *r1 = 11;
    const int rng = bd + 7 - *r0 + 2;
    if (rng > 16)
        *r0 += rng - 16, *r1 -= rng - 16;
}

/* sample (x, y) as the stripe [ys, ys + h) of the plane sees it */
static int32_t stripe_px(const SvtHipLrPlane *pl, int x, int y, int ys, int h, int stripe, int first, int last) {
    const int W = (int)pl->width, H = (int)pl->height, is16 = pl->is_16bit;
    const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    if (y < ys && !first) { /* copy_above */
        const int i = y - ys;  /* -3 .. -1 */
        if (!pl->optimized_lr) {
            const int row = 2 * stripe + (i + 2 > 0 ? i + 2 : 0);
            return px(pl->boundary_above, (ptrdiff_t)row * pl->boundary_stride + x + SVT_HIP_LR_EXTRA_HORZ, is16); /* the saved line as it is: its own extension (svt_aom_extend_lines) */
        }
        return px(pl->src, (ptrdiff_t)(i == -3 ? ys - 2 : y) * pl->src_stride + xc, is16);
    }
    if (y >= ys + h && !last) { /* copy_below */
        const int i = y - (ys + h); /* 0 .. 2 */
        if (!pl->optimized_lr) {
            const int row = 2 * stripe + (i < 1 ? i : 1);
            return px(pl->boundary_below, (ptrdiff_t)row * pl->boundary_stride + x + SVT_HIP_LR_EXTRA_HORZ, is16);
        }
        const int yb = i == 2 ? ys + h + 1 : y; /* below the last picture row: its replica (svt_extend_frame) */
        return px(pl->src, (ptrdiff_t)(yb < H ? yb : H - 1) * pl->src_stride + xc, is16);
    }
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y); /* svt_extend_frame: the picture's own edge rows */
    return px(pl->src, (ptrdiff_t)yc * pl->src_stride + xc, is16);
}

ORC_API void orc_restoration_filter_frame(const SvtHipLrPlane *planes, uint32_t n_planes) {
    for (uint32_t p = 0; p < n_planes; p++) {
        const SvtHipLrPlane *pl = &planes[p];
        const int W = (int)pl->width, H = (int)pl->height, is16 = pl->is_16bit, bd = pl->bit_depth;
        const int full = PROC_UNIT >> pl->ss_y, off = RU_OFFSET >> pl->ss_y, pw = PROC_UNIT >> pl->ss_x, us = (int)pl->unit_size;
        const int TP = pw + 2 * BORDER + 2;
        void     *tile = calloc((size_t)(full + 2 * BORDER) * TP, 2), *out = calloc((size_t)full * pw, 2);
        int       r0, r1;
        wiener_rounds(bd, &r0, &r1);
        int stripe = 0;
        for (int ys = 0; ys < H; stripe++) {
            const int first = ys == 0;
            const int nominal = full - (first ? off : 0);
            const int h = nominal < H - ys ? nominal : H - ys;
            const int last = ys + nominal >= H; /* last_stripe_in_tile (:271) */
            /* the restoration unit row this stripe belongs to: unit i covers rows [i*us - off, (i+1)*us - off), the last one to H */
            int ur = (ys + off) / us;
            ur     = ur > (int)pl->vert_units - 1 ? (int)pl->vert_units - 1 : ur;
            for (int x0 = 0; x0 < W; x0 += pw) {
                const int w = pw < W - x0 ? pw : W - x0;
                int       uc = x0 / us;
                uc           = uc > (int)pl->horz_units - 1 ? (int)pl->horz_units - 1 : uc;
                const SvtHipLrUnit *u = &pl->units[ur * pl->horz_units + uc];
                for (int r = -BORDER; r < h + BORDER; r++)
                    for (int c = -BORDER; c < w + BORDER; c++) {
                        const int32_t v = stripe_px(pl, x0 + c, ys + r, ys, h, stripe, first, last);
                        if (is16)
                            ((uint16_t *)tile)[(r + BORDER) * TP + c + BORDER] = (uint16_t)v;
                        else
                            ((uint8_t *)tile)[(r + BORDER) * TP + c + BORDER] = (uint8_t)v;
                    }
                const void *t0 = (const uint8_t *)tile + (((size_t)BORDER * TP + BORDER) << is16);
                uint8_t    *d  = (uint8_t *)pl->dst + (((size_t)ys * pl->dst_stride + x0) << is16);
                if (u->restoration_type == 1)
                    orc_wiener_convolve_add_src(t0, TP, d, (int32_t)pl->dst_stride, u->hfilter, u->vfilter, w, h, r0, r1, bd, is16);
                else if (u->restoration_type == 2)
                    orc_apply_selfguided_restoration(t0, w, h, TP, u->ep, u->xqd, d, (int32_t)pl->dst_stride, bd, is16);
                else
                    for (int r = 0; r < h; r++)
                        memcpy(d + (((size_t)r * pl->dst_stride) << is16), (const uint8_t *)t0 + (((size_t)r * TP) << is16), (size_t)w << is16);
            }
            ys += h;
        }
        free(tile), free(out);
    }
}

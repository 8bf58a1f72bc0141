/*
 * oracle/src/orc_me.c — TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the reference's per-64x64 open-loop motion estimation
 * (svt_aom_motion_estimation_b64, Source/Lib/Codec/motion_estimation.c:3146-3223, and the b64 loop
 * of Source/Lib/Codec/me_process.c:174-290), ME_OPEN_LOOP only.  Pinned bit-exactly against the real
 * function through oracle/_ref (tests/test_oracle_vs_ref.py).  Integer types (int16 wrap, uint32 wrap)
 * follow the reference expression by expression; comments give the reference lines.
 *
 * Not supported (the product rejects them with SVT_HIP_ERR_BAD_PARAMETER as well):
 *   restricted_motion_vector (tile-restricted search, motion_estimation.c:1445-1477),
 *   global-motion detection (perform_gm_detection, :2908-3031; off for enc_mode > M2),
 *   ME_MCTF (temporal filtering caller).
 */
#include "orc.h"

#include <stdlib.h>
#include <string.h>

#define NL 2
#define NR 4
#define MAX_SAD_VALUE (128 * 128 * 255) /* motion_estimation.h:85 */
#define MAX_U32 0xFFFFFFFFu

#define MINV(a, b) ((a) < (b) ? (a) : (b))
#define MAXV(a, b) ((a) > (b) ? (a) : (b))
#define ABSV(a) ((a) < 0 ? -(a) : (a))

typedef struct {
    uint16_t sa_w, sa_h;
    int16_t  col, row;
    uint64_t sad;
    uint8_t  valid;
} PreHme;

typedef struct {
    const SvtHipMeParams *p;
    const SvtHipPyramid8 *src;
    const SvtHipPyramid8 (*ref)[NR];
    /* per b64 */
    uint32_t org_x, org_y, b64_w, b64_h;
    const uint8_t *src_full, *src_q, *src_s;
    SvtHipMeSearchResult sr[NL][NR];
    uint32_t reduce_div[NL][NR];
    uint32_t zz_sad[NL][NR];
    PreHme   ph[NL][NR][2];
    uint8_t  performed_phme[NL][NR][2];
    int16_t  l0x[NL][NR][2][2], l0y[NL][NR][2][2], l1x[NL][NR][2][2], l1y[NL][NR][2][2], l2x[NL][NR][2][2],
        l2y[NL][NR][2][2];
    uint64_t l0s[NL][NR][2][2], l1s[NL][NR][2][2], l2s[NL][NR][2][2];
    SvtHipSearchArea l0_min, l0_max; /* me_ctx->hme_l0_sa, modified in place per reference */
    uint32_t best_sad[NL][NR][85], best_mv[NL][NR][85];
    uint8_t  searched[NL][NR];
    uint32_t me_distortion[85];
} Ctx;

/* motion_estimation.c:1239-1243 */
static uint16_t scaled_dist(uint16_t dist) { return (uint16_t)(((dist * 5) / 8) + ((dist % 8) == 0 ? 0 : 1)); }

/* get_me_reference's *dist (motion_estimation.c:1232-1234) */
static uint16_t pic_dist(const Ctx *c, int li, int ri) {
    int64_t d = (int64_t)c->p->picture_number - (int64_t)c->p->ref_picture_number[li][ri];
    return (uint16_t)(int16_t)ABSV(d);
}

static inline const uint8_t *plane_at(const SvtHipPlane8 *pl, int x, int y) {
    return pl->buf + (ptrdiff_t)(pl->org_y + y) * pl->stride + pl->org_x + x;
}

/* Common body of hme_level_0/1/2 and prehme_core: one svt_sad_loop_kernel call with the
 * SUB_SAD/FULL_SAD argument rewrite (motion_estimation.c:891-917, 993-1019, 1087-1110, 1708-1733). */
static void hme_search(const Ctx *c, const uint8_t *src, uint32_t src_stride, const SvtHipPlane8 *rp, int x_tl,
                       int y_tl, uint32_t bw, uint32_t bh, int16_t sa_w, int16_t sa_h, uint8_t skip,
                       uint64_t *sad, int16_t *mx, int16_t *my) {
    const int      full = c->p->hme_search_method == 1; /* FULL_SAD_SEARCH */
    const uint8_t *r    = rp->buf + (ptrdiff_t)y_tl * rp->stride + x_tl;
    orc_sad_loop_kernel(src, full ? src_stride : src_stride * 2, r, full ? rp->stride : rp->stride * 2,
                        full ? bh : bh >> 1, bw, sad, mx, my, rp->stride, skip, sa_w, sa_h);
    if (!full)
        *sad *= 2;
}

/* motion_estimation.c:1638-1736 */
static void prehme_core(Ctx *c, int16_t org_x, int16_t org_y, uint32_t bw, uint32_t bh, const SvtHipPlane8 *rp,
                        PreHme *d) {
    int16_t pad_w = (int16_t)rp->org_x - 1, pad_h = (int16_t)rp->org_y - 1;
    int16_t sa_w = (int16_t)d->sa_w, sa_h = (int16_t)d->sa_h;
    int16_t ox = -(int16_t)(sa_w >> 1), oy = -(int16_t)(sa_h >> 1);
    int16_t W = (int16_t)rp->width, H = (int16_t)rp->height;

    ox   = ((org_x + ox) < -pad_w) ? (int16_t)(-pad_w - org_x) : ox;
    sa_w = ((org_x + ox) < -pad_w) ? (int16_t)(sa_w - (-pad_w - (org_x + ox))) : sa_w;
    ox   = ((org_x + ox) > W - 1) ? (int16_t)(ox - ((org_x + ox) - (W - 1))) : ox;
    sa_w = ((org_x + ox + sa_w) > W) ? (int16_t)MAXV(1, sa_w - ((org_x + ox + sa_w) - W)) : sa_w;
    oy   = ((org_y + oy) < -pad_h) ? (int16_t)(-pad_h - org_y) : oy;
    sa_h = ((org_y + oy) < -pad_h) ? (int16_t)(sa_h - (-pad_h - (org_y + oy))) : sa_h;
    oy   = ((org_y + oy) > H - 1) ? (int16_t)(oy - ((org_y + oy) - (H - 1))) : oy;
    sa_h = ((org_y + oy + sa_h) > H) ? (int16_t)MAXV(1, sa_h - ((org_y + oy + sa_h) - H)) : sa_h;

    int16_t x_tl = (int16_t)(((int16_t)rp->org_x + org_x) + ox);
    int16_t y_tl = (int16_t)(((int16_t)rp->org_y + org_y) + oy);
    hme_search(c, c->src_s, c->src->sixteenth.stride, rp, x_tl, y_tl, bw, bh, sa_w, sa_h,
               c->p->prehme_skip_search_line, &d->sad, &d->col, &d->row);
    d->col = (int16_t)(d->col + ox);
    d->col = (int16_t)(d->col * 4);
    d->row = (int16_t)(d->row + oy);
    d->row = (int16_t)(d->row * 4);
    d->valid = 1;
}

/* motion_estimation.c:1763-1789 */
static int prehme_early_exit(Ctx *c, int li, int ri, int si) {
    const SvtHipMeParams *p = c->p;
    PreHme               *d = &c->ph[li][ri][si];
    if (p->me_early_exit_th && c->zz_sad[li][ri] < p->me_early_exit_th) {
        d->col = d->row = 0;
        d->sad          = 0;
        d->valid        = 1;
        return 1;
    }
    if (p->prehme_l1_early_exit) {
        const PreHme *o = &c->ph[0][ri][si];
        if (li == 1 && o->valid && ((o->sad < (32 * 32)) || ((ABSV(o->col) < 16) && (ABSV(o->row) < 16)))) {
            d->col   = (int16_t)-o->col;
            d->row   = (int16_t)-o->row;
            d->sad   = o->sad;
            d->valid = 1;
            return 1;
        }
    }
    return 0;
}

/* motion_estimation.c:1792-1866 */
static void prehme_b64(Ctx *c) {
    const SvtHipMeParams *p        = c->p;
    uint32_t              best_sad = MAX_U32;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
            if (p->temporal_layer_index > 0 || li == 0) {
                uint32_t factor = scaled_dist(pic_dist(c, li, ri));
                for (int si = 0; si < 2; si++) {
                    if (prehme_early_exit(c, li, ri, si))
                        continue;
                    PreHme *d = &c->ph[li][ri][si];
                    if (!c->sr[li][ri].do_ref) {
                        d->col = d->row = 0;
                        d->sad          = MAX_U32;
                        continue;
                    }
                    d->sa_w = (uint16_t)MINV((uint32_t)(p->prehme_sa_min[si].width * factor),
                                             (uint32_t)p->prehme_sa_max[si].width);
                    d->sa_h = (uint16_t)MINV((uint32_t)(p->prehme_sa_min[si].height * factor),
                                             (uint32_t)p->prehme_sa_max[si].height);
                    prehme_core(c, (int16_t)(((int16_t)c->org_x) >> 2), (int16_t)(((int16_t)c->org_y) >> 2),
                                c->b64_w >> 2, c->b64_h >> 2, &c->ref[li][ri].sixteenth, d);
                    c->performed_phme[li][ri][si] = 1;
                }
                uint32_t min_sad = (uint32_t)MINV(c->ph[li][ri][0].sad, c->ph[li][ri][1].sad);
                best_sad         = MINV(best_sad, min_sad);
            } else {
                for (int si = 0; si < 2; si++) {
                    c->ph[1][ri][si].col = (int16_t)-c->ph[0][ri][si].col;
                    c->ph[1][ri][si].row = (int16_t)-c->ph[0][ri][si].row;
                    c->ph[1][ri][si].sad = c->ph[0][ri][si].sad;
                }
            }
        }
    if (p->temporal_layer_index > 0 && best_sad < p->phme_sad_th) {
        for (int li = 0; li < p->num_of_list_to_search; ++li)
            for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
                if (!c->sr[li][ri].do_ref || ri == 0)
                    continue;
                const uint32_t th  = p->phme_sad_pct;
                uint32_t       sad = (uint32_t)MINV(c->ph[li][ri][0].sad, c->ph[li][ri][1].sad);
                if ((uint32_t)((sad - best_sad) * 100u) > (uint32_t)(th * best_sad)) /* uint32 wrap, :1860 */
                    c->sr[li][ri].do_ref = 0;
            }
    }
}

/* motion_estimation.c:1870-1937 */
static void get_hme_l0_search_area(Ctx *c, int li, int ri, uint16_t dist, int16_t *sa_w, int16_t *sa_h) {
    const SvtHipMeParams *p = c->p;
    if (p->enable_me_sr_adjustment && p->distance_based_hme_resizing) {
        uint8_t is_hor = 1, is_ver = 1, is_still = 0;
        if (p->reduce_hme_l0_sr_th_min && p->reduce_hme_l0_sr_th_max && (li || ri)) {
            int16_t mvx = c->l0x[0][0][0][0], mvy = c->l0y[0][0][0][0];
            is_ver   = (ABSV(mvx) < p->reduce_hme_l0_sr_th_min) && (ABSV(mvy) > p->reduce_hme_l0_sr_th_max);
            is_hor   = (ABSV(mvx) > p->reduce_hme_l0_sr_th_max) && (ABSV(mvy) < p->reduce_hme_l0_sr_th_min);
            is_still = (ABSV(mvx) < (p->reduce_hme_l0_sr_th_min * 3)) && (ABSV(mvy) < (p->reduce_hme_l0_sr_th_min * 3));
        }
        uint8_t xo = 1, yo = 1;
        if (!is_ver) yo = 2;
        if (!is_hor) xo = 2;
        if (p->enable_me_sr_adjustment == 2 && is_still) xo = yo = 4;
        c->l0_min.width  = (uint16_t)(c->l0_min.width / (xo + ri));
        c->l0_min.height = (uint16_t)(c->l0_min.height / (yo + ri));
        c->l0_max.width  = (uint16_t)(c->l0_max.width / (xo + ri));
        c->l0_max.height = (uint16_t)(c->l0_max.height / (yo + ri));
    }
    int32_t factor = scaled_dist(dist);
    int16_t w      = (int16_t)(c->l0_min.width / p->num_hme_sa_w);
    w = (int16_t)MINV((((w * factor) + 15) & ~0x0F), (((c->l0_max.width / p->num_hme_sa_w) + 15) & ~0x0F));
    int16_t h = (int16_t)(c->l0_min.height / p->num_hme_sa_h);
    h         = (int16_t)MINV((h * factor), (c->l0_max.height / p->num_hme_sa_h));
    *sa_w = w;
    *sa_h = h;
}

/* The window clamp shared by hme_level_0/1/2 (motion_estimation.c:837-888, 940-990, 1041-1084).
 * Note the left/top branches update the origin first, so the width/height correction is a no-op. */
static void hme_clamp(int16_t org_x, int16_t org_y, int16_t pad_w, int16_t pad_h, int16_t W, int16_t H,
                      int16_t *pox, int16_t *poy, int16_t *pw, int16_t *ph) {
    int16_t ox = *pox, oy = *poy, sa_w = *pw, sa_h = *ph;
    if ((org_x + ox) < -pad_w) {
        ox   = (int16_t)(-pad_w - org_x);
        sa_w = (int16_t)(sa_w - (-pad_w - (org_x + ox)));
    }
    if ((org_x + ox) > W - 1)
        ox = (int16_t)(ox - ((org_x + ox) - (W - 1)));
    if ((org_x + ox + sa_w) > W)
        sa_w = (int16_t)MAXV(1, sa_w - ((org_x + ox + sa_w) - W));
    sa_w = (sa_w < 8) ? sa_w : (int16_t)(sa_w & ~0x07);
    if ((org_y + oy) < -pad_h) {
        oy   = (int16_t)(-pad_h - org_y);
        sa_h = (int16_t)(sa_h - (-pad_h - (org_y + oy)));
    }
    if ((org_y + oy) > H - 1)
        oy = (int16_t)(oy - ((org_y + oy) - (H - 1)));
    if ((org_y + oy + sa_h) > H)
        sa_h = (int16_t)MAXV(1, sa_h - ((org_y + oy + sa_h) - H));
    *pox = ox, *poy = oy, *pw = sa_w, *ph = sa_h;
}

/* motion_estimation.c:820-920 */
static void hme_level_0(Ctx *c, int16_t org_x, int16_t org_y, uint32_t bw, uint32_t bh, int16_t sa_w,
                        int16_t sa_h, const SvtHipPlane8 *rp, uint32_t sr_w, uint32_t sr_h, uint64_t *sad,
                        int16_t *mx, int16_t *my) {
    sa_w        = (int16_t)((sa_w + 7) & ~0x07);
    int16_t xd  = (int16_t)(sa_w * sr_w);
    int16_t yd  = (int16_t)(sa_h * sr_h);
    int16_t ox  = (int16_t)(-(int16_t)((sa_w * c->p->num_hme_sa_w) >> 1) + xd);
    int16_t oy  = (int16_t)(-(int16_t)((sa_h * c->p->num_hme_sa_h) >> 1) + yd);
    hme_clamp(org_x, org_y, (int16_t)(rp->org_x - 1), (int16_t)(rp->org_y - 1), (int16_t)rp->width,
              (int16_t)rp->height, &ox, &oy, &sa_w, &sa_h);
    int16_t x_tl = (int16_t)(((int16_t)rp->org_x + org_x) + ox);
    int16_t y_tl = (int16_t)(((int16_t)rp->org_y + org_y) + oy);
    hme_search(c, c->src_s, c->src->sixteenth.stride, rp, x_tl, y_tl, bw, bh, sa_w, sa_h, 0, sad, mx, my);
    *mx = (int16_t)(*mx + ox);
    *mx = (int16_t)(*mx * 4);
    *my = (int16_t)(*my + oy);
    *my = (int16_t)(*my * 4);
}

/* motion_estimation.c:923-1022 */
static void hme_level_1(Ctx *c, int16_t org_x, int16_t org_y, uint32_t bw, uint32_t bh, const SvtHipPlane8 *rp,
                        int16_t sa_w, int16_t sa_h, int16_t cx, int16_t cy, uint64_t *sad, int16_t *mx,
                        int16_t *my) {
    sa_w       = (int16_t)((sa_w + 7) & ~0x07);
    int16_t ox = (int16_t)(-(sa_w >> 1) + cx);
    int16_t oy = (int16_t)(-(sa_h >> 1) + cy);
    hme_clamp(org_x, org_y, (int16_t)(rp->org_x - 1), (int16_t)(rp->org_y - 1), (int16_t)rp->width,
              (int16_t)rp->height, &ox, &oy, &sa_w, &sa_h);
    int16_t x_tl = (int16_t)(((int16_t)rp->org_x + org_x) + ox);
    int16_t y_tl = (int16_t)(((int16_t)rp->org_y + org_y) + oy);
    hme_search(c, c->src_q, c->src->quarter.stride, rp, x_tl, y_tl, bw, bh, sa_w, sa_h, 0, sad, mx, my);
    *mx = (int16_t)(*mx + ox);
    *mx = (int16_t)(*mx * 2);
    *my = (int16_t)(*my + oy);
    *my = (int16_t)(*my * 2);
}

/* motion_estimation.c:1025-1113 (pad is BLOCK_SIZE_64-1 here, not the plane's padding) */
static void hme_level_2(Ctx *c, int16_t org_x, int16_t org_y, uint32_t bw, uint32_t bh, const SvtHipPlane8 *rp,
                        int16_t sa_w, int16_t sa_h, int16_t cx, int16_t cy, uint64_t *sad, int16_t *mx,
                        int16_t *my) {
    sa_w       = (int16_t)((sa_w + 7) & ~0x07);
    int16_t ox = (int16_t)(-(sa_w >> 1) + cx);
    int16_t oy = (int16_t)(-(sa_h >> 1) + cy);
    hme_clamp(org_x, org_y, 63, 63, (int16_t)rp->width, (int16_t)rp->height, &ox, &oy, &sa_w, &sa_h);
    int16_t x_tl = (int16_t)(((int16_t)rp->org_x + org_x) + ox);
    int16_t y_tl = (int16_t)(((int16_t)rp->org_y + org_y) + oy);
    hme_search(c, c->src_full, c->src->full.stride, rp, x_tl, y_tl, bw, bh, sa_w, sa_h, 0, sad, mx, my);
    *mx = (int16_t)(*mx + ox);
    *my = (int16_t)(*my + oy);
}

static void set_quadrants(int16_t x[2][2], int16_t y[2][2], uint64_t s[2][2], int16_t vx, int16_t vy,
                          uint64_t vs) {
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) x[a][b] = vx, y[a][b] = vy, s[a][b] = vs;
}

/* motion_estimation.c:1976-2106 */
static void hme_level0_b64(Ctx *c) {
    const SvtHipMeParams  *p        = c->p;
    const SvtHipSearchArea base_min = c->l0_min, base_max = c->l0_max;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
            if (p->me_early_exit_th && c->zz_sad[li][ri] < (p->me_early_exit_th >> 2)) {
                set_quadrants(c->l0x[li][ri], c->l0y[li][ri], c->l0s[li][ri], 0, 0, 0);
                continue;
            }
            if (p->prev_me_stage_based_exit_th) {
                int si = c->ph[li][ri][0].sad <= c->ph[li][ri][1].sad ? 0 : 1;
                if (c->performed_phme[li][ri][si] &&
                    c->ph[li][ri][si].sad < (p->prev_me_stage_based_exit_th >> 4)) {
                    set_quadrants(c->l0x[li][ri], c->l0y[li][ri], c->l0s[li][ri], c->ph[li][ri][si].col,
                                  c->ph[li][ri][si].row, c->ph[li][ri][si].sad);
                    continue;
                }
            }
            if (!c->sr[li][ri].do_ref) {
                set_quadrants(c->l0x[li][ri], c->l0y[li][ri], c->l0s[li][ri], 0, 0, MAX_U32);
                continue;
            }
            if (p->temporal_layer_index > 0 || li == 0) {
                int16_t sa_w = 0, sa_h = 0;
                get_hme_l0_search_area(c, li, ri, pic_dist(c, li, ri), &sa_w, &sa_h);
                for (uint32_t sh = 0; sh < p->num_hme_sa_h; sh++)
                    for (uint32_t sw = 0; sw < p->num_hme_sa_w; sw++)
                        hme_level_0(c, (int16_t)(((int16_t)c->org_x) >> 2), (int16_t)(((int16_t)c->org_y) >> 2),
                                    c->b64_w >> 2, c->b64_h >> 2, sa_w, sa_h, &c->ref[li][ri].sixteenth, sw, sh,
                                    &c->l0s[li][ri][sw][sh], &c->l0x[li][ri][sw][sh], &c->l0y[li][ri][sw][sh]);
                if (p->enable_me_sr_adjustment && p->distance_based_hme_resizing) {
                    c->l0_min = base_min;
                    c->l0_max = base_max;
                }
                if (p->prehme_enable) {
                    /* get_worst_quadrant, :1942-1971 (last compare does not update max_sad) */
                    uint8_t  bw = 0, bh = 0;
                    uint64_t mx = 0;
                    if (c->l0s[li][ri][0][0] > mx) mx = c->l0s[li][ri][0][0], bw = 0, bh = 0;
                    if (c->l0s[li][ri][1][0] > mx) mx = c->l0s[li][ri][1][0], bw = 1, bh = 0;
                    if (c->l0s[li][ri][0][1] > mx) mx = c->l0s[li][ri][0][1], bw = 0, bh = 1;
                    if (c->l0s[li][ri][1][1] > mx) bw = 1, bh = 1;
                    int si = c->ph[li][ri][0].sad <= c->ph[li][ri][1].sad ? 0 : 1;
                    if (c->ph[li][ri][si].sad < c->l0s[li][ri][bw][bh]) {
                        c->l0s[li][ri][bw][bh] = c->ph[li][ri][si].sad;
                        c->l0x[li][ri][bw][bh] = c->ph[li][ri][si].col;
                        c->l0y[li][ri][bw][bh] = c->ph[li][ri][si].row;
                    }
                }
            }
        }
}

/* motion_estimation.c:2111-2192 */
static void hme_level1_b64(Ctx *c) {
    const SvtHipMeParams *p = c->p;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
            if (!(p->temporal_layer_index > 0 || li == 0))
                continue;
            if (p->me_early_exit_th && c->zz_sad[li][ri] < (p->me_early_exit_th >> 2)) {
                set_quadrants(c->l1x[li][ri], c->l1y[li][ri], c->l1s[li][ri], 0, 0, 0);
                continue;
            }
            if (!c->sr[li][ri].do_ref) {
                set_quadrants(c->l1x[li][ri], c->l1y[li][ri], c->l1s[li][ri], 0, 0, MAX_U32);
                continue;
            }
            for (uint32_t sh = 0; sh < p->num_hme_sa_h; sh++)
                for (uint32_t sw = 0; sw < p->num_hme_sa_w; sw++) {
                    if (p->prev_me_stage_based_exit_th &&
                        c->l0s[li][ri][sw][sh] < (p->prev_me_stage_based_exit_th >> 5)) {
                        c->l1x[li][ri][sw][sh] = c->l0x[li][ri][sw][sh];
                        c->l1y[li][ri][sw][sh] = c->l0y[li][ri][sw][sh];
                        c->l1s[li][ri][sw][sh] = c->l0s[li][ri][sw][sh];
                        continue;
                    }
                    hme_level_1(c, (int16_t)(((int16_t)c->org_x) >> 1), (int16_t)(((int16_t)c->org_y) >> 1),
                                c->b64_w >> 1, c->b64_h >> 1, &c->ref[li][ri].quarter, (int16_t)p->hme_l1_sa.width,
                                (int16_t)p->hme_l1_sa.height, (int16_t)(c->l0x[li][ri][sw][sh] >> 1),
                                (int16_t)(c->l0y[li][ri][sw][sh] >> 1), &c->l1s[li][ri][sw][sh],
                                &c->l1x[li][ri][sw][sh], &c->l1y[li][ri][sw][sh]);
                }
        }
}

/* motion_estimation.c:2197-2247 */
static void hme_level2_b64(Ctx *c) {
    const SvtHipMeParams *p = c->p;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
            if (!(p->temporal_layer_index > 0 || li == 0))
                continue;
            for (uint32_t sh = 0; sh < p->num_hme_sa_h; sh++)
                for (uint32_t sw = 0; sw < p->num_hme_sa_w; sw++) {
                    if (p->prev_me_stage_based_exit_th &&
                        c->l1s[li][ri][sw][sh] < (p->prev_me_stage_based_exit_th >> 2)) {
                        c->l2x[li][ri][sw][sh] = c->l1x[li][ri][sw][sh];
                        c->l2y[li][ri][sw][sh] = c->l1y[li][ri][sw][sh];
                        c->l2s[li][ri][sw][sh] = c->l1s[li][ri][sw][sh];
                        continue;
                    }
                    hme_level_2(c, (int16_t)c->org_x, (int16_t)c->org_y, c->b64_w, c->b64_h, &c->ref[li][ri].full,
                                (int16_t)p->hme_l2_sa.width, (int16_t)p->hme_l2_sa.height, c->l1x[li][ri][sw][sh],
                                c->l1y[li][ri][sw][sh], &c->l2s[li][ri][sw][sh], &c->l2x[li][ri][sw][sh],
                                &c->l2y[li][ri][sw][sh]);
                }
        }
}

/* Pick the best quadrant in the order (0,0),(1,0),(0,1),(1,1) with strict <
 * (the while loops of motion_estimation.c:2296-2331 etc.). */
static void best_quadrant(int16_t x[2][2], int16_t y[2][2], uint64_t s[2][2], int16_t *bx, int16_t *by,
                          uint64_t *bs) {
    static const int order[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}};
    *bx = x[0][0], *by = y[0][0], *bs = s[0][0];
    for (int k = 1; k < 4; k++) {
        int w = order[k][0], h = order[k][1];
        if (s[w][h] < *bs)
            *bx = x[w][h], *by = y[w][h], *bs = s[w][h];
    }
}

/* motion_estimation.c:2252-2450.  xc/yc/sad deliberately carry over between references. */
static void set_final_search_centre(Ctx *c) {
    const SvtHipMeParams *p  = c->p;
    int16_t               hx = 0, hy = 0, xc = 0, yc = 0;
    uint64_t              hsad = 0;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
            if (p->temporal_layer_index > 0 || li == 0) {
                if (p->enable_hme_flag) {
                    if (p->enable_hme_level0_flag && !p->enable_hme_level1_flag && !p->enable_hme_level2_flag)
                        best_quadrant(c->l0x[li][ri], c->l0y[li][ri], c->l0s[li][ri], &hx, &hy, &hsad);
                    if (p->enable_hme_level1_flag && !p->enable_hme_level2_flag)
                        best_quadrant(c->l1x[li][ri], c->l1y[li][ri], c->l1s[li][ri], &hx, &hy, &hsad);
                    if (p->enable_hme_level2_flag)
                        best_quadrant(c->l2x[li][ri], c->l2y[li][ri], c->l2s[li][ri], &hx, &hy, &hsad);
                    xc = hx;
                    yc = hy;
                }
            } else {
                xc = 0;
                yc = 0;
            }
            c->sr[li][ri].hme_sc_x = xc;
            c->sr[li][ri].hme_sc_y = yc;
            c->sr[li][ri].hme_sad  = hsad;
        }
}

/* motion_estimation.c:1737-1759 + 2452-2507 */
static void init_zz_sad(Ctx *c) {
    const SvtHipMeParams *p    = c->p;
    uint32_t              best = MAX_U32;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri)
            if (p->temporal_layer_index > 0 || li == 0) {
                const SvtHipPlane8 *rp = &c->ref[li][ri].full;
                uint32_t z = orc_nxm_sad(c->src_full, c->src->full.stride << 1,
                                         plane_at(rp, (int16_t)c->org_x, (int16_t)c->org_y), rp->stride << 1,
                                         c->b64_h >> 1, c->b64_w);
                z <<= 1;
                z                 = (z * 64 * 64) / (c->b64_w * c->b64_h);
                c->zz_sad[li][ri] = z;
                best              = MINV(best, z);
            }
    if (p->temporal_layer_index > 0 && best < p->zz_sad_th) {
        for (int li = 0; li < p->num_of_list_to_search; ++li)
            for (int ri = 1; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
                const uint32_t pct = p->zz_sad_pct;
                if ((uint32_t)((c->zz_sad[li][ri] - best) * 100u) > (uint32_t)(pct * best))
                    c->sr[li][ri].do_ref = 0;
            }
    }
    if (p->me_safe_limit_zz_th) {
        int lim = p->hierarchical_levels > 0 && p->num_of_list_to_search == 2 &&
            p->temporal_layer_index >= p->hierarchical_levels && p->similar_brightness_refs &&
            c->zz_sad[0][0] < p->me_safe_limit_zz_th && c->zz_sad[1][0] < p->me_safe_limit_zz_th;
        if (lim)
            for (int li = 0; li < p->num_of_list_to_search; ++li)
                for (int ri = 1; ri < p->num_of_ref_pic_to_search[li]; ++ri) c->sr[li][ri].do_ref = 0;
    }
}

/* motion_estimation.c:2547-2588 */
static void hme_prune_ref_and_adjust_sr(Ctx *c) {
    const SvtHipMeParams *p  = c->p;
    uint16_t              th = p->prune_ref_if_hme_sad_dev_bigger_than_th;
    if (p->enable_me_hme_ref_pruning && th != (uint16_t)~0) {
        uint64_t best = ~(uint64_t)0;
        for (int i = 0; i < NL; i++)
            for (int j = 0; j < NR; j++)
                if (c->sr[i][j].hme_sad < best) best = c->sr[i][j].hme_sad;
        for (int i = 0; i < NL; i++)
            for (int j = 1; j < NR; j++)
                if ((c->sr[i][j].hme_sad - best) * 100 > (th * best)) c->sr[i][j].do_ref = 0;
    }
    if (p->enable_me_sr_adjustment) {
        for (int i = 0; i < NL; i++)
            for (int j = 0; j < NR; j++) {
                if (ABSV(c->sr[i][j].hme_sc_x) <= p->reduce_me_sr_based_on_mv_length_th &&
                    ABSV(c->sr[i][j].hme_sc_y) <= p->reduce_me_sr_based_on_mv_length_th &&
                    c->sr[i][j].hme_sad < p->stationary_hme_sad_abs_th)
                    c->reduce_div[i][j] = p->stationary_me_sr_divisor;
                else if (c->sr[i][j].hme_sad < p->reduce_me_sr_based_on_hme_sad_abs_th)
                    c->reduce_div[i][j] = p->me_sr_divisor_for_low_hme_sad;
            }
    }
}

/* motion_estimation.c:1139-1206 */
static uint32_t check_00_center(Ctx *c, const SvtHipPlane8 *rp, int16_t *xc, int16_t *yc, uint32_t zz_sad) {
    const SvtHipMeParams *p = c->p;
    int16_t org_x = (int16_t)c->org_x, org_y = (int16_t)c->org_y, pad = 63;
    int16_t W = (int16_t)rp->width, H = (int16_t)rp->height;
    uint32_t zero_sad;
    if (p->me_early_exit_th)
        zero_sad = zz_sad;
    else
        zero_sad = orc_nxm_sad(c->src_full, c->src->full.stride << 1, plane_at(rp, org_x, org_y), rp->stride << 1,
                               c->b64_h >> 1, c->b64_w);
    zero_sad <<= 1;
    *xc = ((org_x + *xc) < -pad) ? (int16_t)(-pad - org_x) : *xc;
    *xc = ((org_x + *xc) > W - 1) ? (int16_t)(*xc - ((org_x + *xc) - (W - 1))) : *xc;
    *yc = ((org_y + *yc) < -pad) ? (int16_t)(-pad - org_y) : *yc;
    *yc = ((org_y + *yc) > H - 1) ? (int16_t)(*yc - ((org_y + *yc) - (H - 1))) : *yc;
    uint64_t zero_cost = (uint64_t)(zero_sad << 8);
    uint32_t hme_sad   = orc_nxm_sad(c->src_full, c->src->full.stride << 1,
                                     plane_at(rp, org_x + *xc, org_y + *yc), rp->stride << 1, c->b64_h >> 1,
                                     c->b64_w);
    hme_sad <<= 1;
    uint64_t hme_cost = (uint64_t)(hme_sad << 8);
    uint64_t cost     = MINV(zero_cost, hme_cost);
    *xc = (cost == zero_cost) ? 0 : *xc;
    *yc = (cost == zero_cost) ? 0 : *yc;
    return hme_sad;
}

/* open_loop_me_fullpel_search_sblock, motion_estimation.c:781-817 with the two point functions
 * (:429-472, :476-776): raster over the window, groups of 8 then single positions. */
static void fullpel_search(Ctx *c, int li, int ri, const uint8_t *win, uint32_t stride, int16_t x0, int16_t y0,
                           uint32_t w, uint32_t h) {
    const uint8_t sub = c->p->me_search_method == 0; /* SUB_SAD_SEARCH */
    uint32_t     *bs  = c->best_sad[li][ri], *bm = c->best_mv[li][ri];
    uint32_t      e16[16][8], e32[4][8], s16[16], s8[64], s32[4];
    const uint32_t w8 = w - (w & 7);
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w8; x += 8) {
            uint32_t mv = ((uint32_t)(uint16_t)((int32_t)y + y0) << 16) | (uint16_t)((int32_t)x + x0);
            orc_ext_all_sad_calculation_8x8_16x16(c->src_full, c->src->full.stride, win + (size_t)y * stride + x,
                                                  stride, mv, bs + 21, bs + 5, bm + 21, bm + 5, e16, NULL, sub);
            orc_ext_eight_sad_calculation_32x32_64x64(e16, bs + 1, bs, bm + 1, bm, mv, e32);
        }
        for (uint32_t x = w8; x < w; x++) {
            uint32_t       mv = ((uint32_t)(uint16_t)((int32_t)y + y0) << 16) | (uint16_t)((int32_t)x + x0);
            const uint8_t *r  = win + (size_t)y * stride + x;
            for (int by = 0; by < 4; by++)
                for (int bx = 0; bx < 4; bx++) {
                    uint32_t z = orc_z16[4 * by + bx];
                    orc_ext_sad_calculation_8x8_16x16(c->src_full + (size_t)16 * by * c->src->full.stride + 16 * bx,
                                                      c->src->full.stride, r + (size_t)16 * by * stride + 16 * bx,
                                                      stride, bs + 21 + 4 * z, bs + 5 + z, bm + 21 + 4 * z,
                                                      bm + 5 + z, mv, &s16[z], &s8[4 * z], sub);
                }
            orc_ext_sad_calculation_32x32_64x64(s16, bs + 1, bs, bm + 1, bm, mv, s32);
        }
    }
}

/* motion_estimation.c:1249-1586 */
static void integer_search_b64(Ctx *c) {
    const SvtHipMeParams *p = c->p;
    const int16_t W = (int16_t)((c->src->full.width + 7) & ~7), H = (int16_t)((c->src->full.height + 7) & ~7);
    const int16_t pad = 63, org_x = (int16_t)c->org_x, org_y = (int16_t)c->org_y;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
            const SvtHipPlane8 *rp   = &c->ref[li][ri].full;
            uint16_t            dist = pic_dist(c, li, ri);
            if (c->sr[li][ri].do_ref == 0)
                continue;
            int16_t xc = c->sr[li][ri].hme_sc_x, yc = c->sr[li][ri].hme_sc_y;
            int16_t sw = (int16_t)p->me_sa_min.width, sh = (int16_t)p->me_sa_min.height;
            if (!p->me_mctf) /* :1302 */
                dist = scaled_dist(dist);
            sw         = (int16_t)MINV((sw * dist), p->me_sa_max.width);
            sh         = (int16_t)MINV((sh * dist), p->me_sa_max.height);
            if (p->mv_sa_adj_enabled && (!p->mv_sa_adj_nearest_ref_only || ri == 0)) {
                if (ABSV(xc) > p->mv_sa_adj_mv_size_th) sw = (int16_t)(sw * p->mv_sa_adj_sa_multiplier);
                if (ABSV(yc) > p->mv_sa_adj_mv_size_th) sh = (int16_t)(sh * p->mv_sa_adj_sa_multiplier);
            }
            sw = (int16_t)((MAXV(1u, ((uint32_t)sw / c->reduce_div[li][ri])) + 7) & ~0x07u);
            sh = (int16_t)MAXV(3u, ((uint32_t)sh / c->reduce_div[li][ri]));
            const int16_t sh0 = sh, sw0 = sw;
            uint64_t      best_hme_sad = ~(uint64_t)0;
            if (p->me_early_exit_th) {
                if (c->zz_sad[li][ri] < (p->me_early_exit_th / 6))
                    sw = sh = 1;
            } else {
                uint8_t accurate = 1;
                if ((xc != 0 || yc != 0) && p->is_ref) {
                    best_hme_sad = check_00_center(c, rp, &xc, &yc, c->zz_sad[li][ri]);
                    if (xc == 0 && yc == 0)
                        accurate = 0;
                }
                if (p->enable_me_sr_adjustment == 2) {
                    if ((accurate && (best_hme_sad < (24 * 24))) || (p->is_ref && c->sr[li][ri].hme_sad < (24 * 24)))
                        sh = (int16_t)(sh / 2);
                    if ((li || ri) && c->best_sad[0][0][0] < 5000 && sh == sh0 && sw == sw0) {
                        sh = (int16_t)(sh >> 1);
                        sw = (int16_t)(sw >> 1);
                    }
                }
            }
            for (int i = 0; i < 85; i++) c->best_sad[li][ri][i] = MAX_SAD_VALUE; /* :1368 (21*4+1 entries) */
            c->searched[li][ri] = 1;

            if (p->me_8x8_var_enabled && (sw * sh > 24)) { /* :1393-1441 */
                const uint8_t *win = plane_at(rp, org_x + xc, org_y + yc);
                fullpel_search(c, li, ri, win, rp->stride, xc, yc, 1, 1);
                const uint32_t mean = c->best_sad[li][ri][0] / 64;
                uint32_t       ssq  = 0;
                for (int i = 0; i < 64; i++) {
                    const int32_t d = (int32_t)c->best_sad[li][ri][21 + i] - (int32_t)mean;
                    ssq += (uint32_t)(d * d);
                }
                uint32_t var = ssq / 64;
                if (var > p->me_sr_mult2_th) {
                    sw = (int16_t)((MAXV(1, sw * 3 / 2) + 7) & ~0x7);
                    sh = (int16_t)MAXV(1, sh * 3 / 2);
                }
                if (var < p->me_sr_div4_th) {
                    sw = (int16_t)((MAXV(1, sw >> 2) + 7) & ~0x7);
                    sh = (int16_t)MAXV(1, sh >> 2);
                    sh = (int16_t)MAXV(3, sh);
                } else if (var < p->me_sr_div2_th) {
                    sw = (int16_t)((MINV(sw, sw >> 1) + 7) & ~0x7);
                    sh = (int16_t)MINV(sh, sh >> 1);
                    sh = (int16_t)MAXV(3, sh);
                }
            }
            int16_t ox = (int16_t)(xc - (sw >> 1)), oy = (int16_t)(yc - (sh >> 1));
            /* :1478-1506, :1539-1561 */
            ox = ((org_x + ox) < -pad) ? (int16_t)(-pad - org_x) : ox;
            sw = ((org_x + ox) < -pad) ? (int16_t)(sw - (-pad - (org_x + ox))) : sw;
            ox = ((org_x + ox) > W - 1) ? (int16_t)(ox - ((org_x + ox) - (W - 1))) : ox;
            sw = ((org_x + ox + sw) > W) ? (int16_t)MAXV(1, sw - ((org_x + ox + sw) - W)) : sw;
            sw = (sw < 8) ? sw : (int16_t)(sw & ~0x07);
            oy = ((org_y + oy) < -pad) ? (int16_t)(-pad - org_y) : oy;
            sh = ((org_y + oy) < -pad) ? (int16_t)(sh - (-pad - (org_y + oy))) : sh;
            oy = ((org_y + oy) > H - 1) ? (int16_t)(oy - ((org_y + oy) - (H - 1))) : oy;
            sh = ((org_y + oy + sh) > H) ? (int16_t)MAXV(1, sh - ((org_y + oy + sh) - H)) : sh;

            const uint8_t *win = plane_at(rp, org_x + ox, org_y + oy);
            fullpel_search(c, li, ri, win, rp->stride, ox, oy, (uint32_t)sw, (uint32_t)sh);
        }
}

static const uint8_t tab8x8[64] = {0,  1,  4,  5,  16, 17, 20, 21, 2,  3,  6,  7,  18, 19, 22, 23,
                                   8,  9,  12, 13, 24, 25, 28, 29, 10, 11, 14, 15, 26, 27, 30, 31,
                                   32, 33, 36, 37, 48, 49, 52, 53, 34, 35, 38, 39, 50, 51, 54, 55,
                                   40, 41, 44, 45, 56, 57, 60, 61, 42, 43, 46, 47, 58, 59, 62, 63};

/* motion_estimation.c:1592-1635 */
static void me_prune_ref(Ctx *c) {
    const SvtHipMeParams *p = c->p;
    for (int li = 0; li < p->num_of_list_to_search; ++li)
        for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ++ri) {
            c->sr[li][ri].hme_sad = 0;
            if (c->sr[li][ri].do_ref == 0) {
                c->sr[li][ri].hme_sad = (uint64_t)(MAX_SAD_VALUE * 64);
                continue;
            }
            for (int k = 0; k < 64; k++) c->sr[li][ri].hme_sad += c->best_sad[li][ri][21 + tab8x8[k]];
        }
    uint16_t th = p->prune_ref_if_me_sad_dev_bigger_than_th;
    if (p->enable_me_hme_ref_pruning && th != (uint16_t)~0) {
        uint64_t best = ~(uint64_t)0;
        for (int i = 0; i < NL; i++)
            for (int j = 0; j < NR; j++)
                if (c->sr[i][j].hme_sad < best) best = c->sr[i][j].hme_sad;
        for (int i = 0; i < NL; i++)
            for (int j = 1; j < NR; j++)
                if ((c->sr[i][j].hme_sad - best) * 100 > (th * best)) c->sr[i][j].do_ref = 0;
    }
}

static const uint8_t z_to_raster[85] = {
    0,  1,  2,  3,  4,  5,  6,  9,  10, 7,  8,  11, 12, 13, 14, 17, 18, 15, 16, 19, 20, 21, 22, 29, 30, 23, 24, 31, 32,
    37, 38, 45, 46, 39, 40, 47, 48, 25, 26, 33, 34, 27, 28, 35, 36, 41, 42, 49, 50, 43, 44, 51, 52, 53, 54, 61, 62, 55,
    56, 63, 64, 69, 70, 77, 78, 71, 72, 79, 80, 57, 58, 65, 66, 59, 60, 67, 68, 73, 74, 81, 82, 75, 76, 83, 84};

static inline uint8_t pack_cand(uint32_t direction, uint32_t l0, uint32_t l1, uint32_t r0, uint32_t r1) {
    /* MeCandidate bit-fields (me_sb_results.h:28-34), LSB first; 1-bit fields keep bit 0 only */
    return (uint8_t)((direction & 3) | ((l0 & 3) << 2) | ((l1 & 3) << 4) | ((r0 & 1) << 6) | ((r1 & 1) << 7));
}

typedef struct {
    uint32_t *mv;
    uint8_t  *cand, *total;
} SbOut;

static int use_me_pu(const SvtHipMeParams *p, uint32_t n) {
    return p->enable_me_16x16 ? (p->enable_me_8x8 || n < 21) : (n < 5);
}

/* motion_estimation.c:2716-2767 */
static void cand_single_ref(Ctx *c, SbOut *o) {
    const SvtHipMeParams *p = c->p;
    uint8_t do_ref = c->sr[0][0].do_ref;
    memset(o->total, 1, svt_hip_me_stored_pus(p));
    for (uint32_t n = 0; n < p->max_number_of_pus_per_sb; ++n) {
        const uint8_t pu     = z_to_raster[n];
        c->me_distortion[pu] = c->best_sad[0][0][n];
        if (!do_ref)
            continue;
        if (use_me_pu(p, n)) {
            o->cand[pu * p->max_cand] = pack_cand(0, 0, 0, 0, 0);
            o->mv[pu * p->max_refs]   = c->best_mv[0][0][n];
        }
    }
}

/* motion_estimation.c:2602-2715 */
static void cand_mrp_off(Ctx *c, SbOut *o, uint32_t nlist) {
    const SvtHipMeParams *p      = c->p;
    uint8_t               org[2] = {c->sr[0][0].do_ref, (uint8_t)((nlist == 1) ? 0 : c->sr[1][0].do_ref)};
    if (nlist < 2 || !c->sr[1][0].do_ref)
        nlist = 1;
    const uint32_t prune_th = (org[0] && org[1]) ? (uint32_t)p->prune_me_candidates_th : 0;
    memset(o->total, 1, svt_hip_me_stored_pus(p));
    for (uint32_t n = 0; n < p->max_number_of_pus_per_sb; ++n) {
        const uint8_t pu  = z_to_raster[n];
        uint8_t       off = 0;
        const int     use = use_me_pu(p, n);
        uint8_t      *ca  = use ? &o->cand[pu * p->max_cand] : NULL;
        uint8_t       dr[2] = {org[0], org[1]};
        const uint32_t best = (org[0] && org[1]) ? MINV(c->best_sad[0][0][n], c->best_sad[1][0][n])
            : org[0]                             ? c->best_sad[0][0][n]
                                                 : c->best_sad[1][0][n];
        c->me_distortion[pu] = best;
        int8_t min_list      = -1;
        if (p->use_best_unipred_cand_only && dr[0] && dr[1])
            min_list = c->best_sad[0][0][n] < c->best_sad[1][0][n] ? 0 : 1;
        for (uint32_t li = 0; li < nlist && (use || off == 0); ++li) {
            if (dr[li] == 0)
                continue;
            if (prune_th > 0) {
                uint32_t d = (c->best_sad[li][0][n] - best) * 100u;
                if (d > (uint32_t)(best * prune_th)) {
                    dr[li] = 0;
                    continue;
                }
            }
            if (min_list != -1 && min_list != (int)li) {
                if (use)
                    o->mv[pu * p->max_refs + (li ? p->max_l0 : 0)] = c->best_mv[li][0][n];
                continue;
            }
            if (use) {
                ca[off] = pack_cand(li, 0, 0, li == 0 ? li : 24, li == 1 ? li : 24);
                o->mv[pu * p->max_refs + (li ? p->max_l0 : 0)] = c->best_mv[li][0][n];
            }
            off++;
        }
        if (dr[0] && dr[1] && use) {
            ca[off]       = pack_cand(2 /*BI_PRED*/, 0, 0, 0, 1);
            o->total[pu] = (uint8_t)(off + 1);
        }
    }
}

/* motion_estimation.c:2768-2905 */
static void cand_general(Ctx *c, SbOut *o, uint32_t nlist) {
    const SvtHipMeParams *p = c->p;
    for (uint32_t n = 0; n < p->max_number_of_pus_per_sb; ++n) {
        const uint8_t pu  = (n > 4) ? z_to_raster[n] : (uint8_t)n;
        uint8_t       off = 0;
        const int     use = use_me_pu(p, n);
        uint8_t      *ca  = use ? &o->cand[pu * p->max_cand] : NULL;
        uint8_t       dr[NL][NR];
        memset(dr, 0, sizeof(dr));
        const uint32_t prune_th = (uint32_t)p->prune_me_candidates_th;
        uint32_t       best     = MAX_U32;
        for (uint32_t li = 0; li < nlist; li++)
            for (uint32_t ri = 0; ri < p->num_of_ref_pic_to_search[li]; ri++) {
                dr[li][ri] = c->sr[li][ri].do_ref;
                if (!dr[li][ri])
                    continue;
                best = c->best_sad[li][ri][n] < best ? c->best_sad[li][ri][n] : best;
            }
        c->me_distortion[pu] = best;
        for (uint32_t li = 0; li < nlist && (use || off == 0); ++li)
            for (uint32_t ri = 0; ri < p->num_of_ref_pic_to_search[li] && (use || off == 0); ++ri) {
                if (!dr[li][ri])
                    continue;
                if (prune_th > 0) {
                    uint32_t d = (c->best_sad[li][ri][n] - best) * 100u;
                    if (d > (uint32_t)(best * prune_th)) {
                        dr[li][ri] = 0;
                        continue;
                    }
                }
                if (use) {
                    ca[off] = pack_cand(li, ri, ri, li == 0 ? li : 24, li == 1 ? li : 24);
                    o->mv[pu * p->max_refs + (li ? p->max_l0 : 0) + ri] = c->best_mv[li][ri][n];
                }
                off++;
            }
        if (nlist == 2 && use) {
            for (uint32_t a = 0; a < p->num_of_ref_pic_to_search[0]; a++)
                for (uint32_t b = 0; b < p->num_of_ref_pic_to_search[1]; b++) {
                    if (p->only_l_bwd && (a > 0 || b > 0))
                        continue;
                    if (dr[0][a] && dr[1][b])
                        ca[off++] = pack_cand(2, a, b, 0, 1);
                }
            if (!p->only_l_bwd) {
                for (uint32_t a = 1; a < p->num_of_ref_pic_to_search[0]; a++)
                    if (dr[0][0] && dr[0][a])
                        ca[off++] = pack_cand(2, 0, a, 0, 0);
                if (p->num_of_ref_pic_to_search[1] == 3 && dr[1][0] && dr[1][2])
                    ca[off++] = pack_cand(2, 0, 2, 1, 1);
            }
        }
        if (use)
            o->total[pu] = off;
    }
}

/* init_me_hme_data, motion_estimation.c:3080-3140 */
static void init_b64(Ctx *c) {
    memset(c->l0x, 0, sizeof(c->l0x)), memset(c->l0y, 0, sizeof(c->l0y));
    memset(c->l1x, 0, sizeof(c->l1x)), memset(c->l1y, 0, sizeof(c->l1y));
    memset(c->l2x, 0, sizeof(c->l2x)), memset(c->l2y, 0, sizeof(c->l2y));
    memset(c->best_mv, 0, sizeof(c->best_mv));
    /* not reset by the reference (stale from the previous b64) but never read before written;
     * zeroed here so that "not searched" is a defined output */
    memset(c->best_sad, 0, sizeof(c->best_sad));
    memset(c->searched, 0, sizeof(c->searched));
    memset(c->l0s, 0, sizeof(c->l0s)), memset(c->l1s, 0, sizeof(c->l1s)), memset(c->l2s, 0, sizeof(c->l2s));
    for (int i = 0; i < NL; i++)
        for (int j = 0; j < NR; j++) {
            memset(&c->sr[i][j], 0, sizeof(c->sr[i][j]));
            c->sr[i][j].do_ref  = 1;
            c->sr[i][j].hme_sad = MAX_U32;
            c->reduce_div[i][j] = 1;
            c->zz_sad[i][j]     = ~0u;
            c->ph[i][j][0].valid = c->ph[i][j][1].valid = 0;
        }
    memset(c->performed_phme, 0, sizeof(c->performed_phme));
}

static void me_b64(Ctx *c, const SvtHipMeFrameJob *job, uint32_t b64_index, uint32_t bx, uint32_t by) {
    const SvtHipMeParams *p = c->p;
    c->org_x = bx * 64, c->org_y = by * 64;
    const uint32_t aw = (c->src->full.width + 7) & ~7u, ah = (c->src->full.height + 7) & ~7u;
    c->b64_w     = (aw - c->org_x) < 64 ? aw - c->org_x : 64;
    c->b64_h     = (ah - c->org_y) < 64 ? ah - c->org_y : 64;
    c->src_full  = plane_at(&c->src->full, (int)c->org_x, (int)c->org_y);
    c->src_q     = plane_at(&c->src->quarter, (int)(c->org_x >> 1), (int)(c->org_y >> 1));
    c->src_s     = plane_at(&c->src->sixteenth, (int)(c->org_x >> 2), (int)(c->org_y >> 2));
    const int prune_ref = p->enable_hme_flag && !p->me_mctf; /* :3173 */
    init_b64(c);
    /* hme_b64, :2511-2545 */
    if (p->me_early_exit_th || p->me_safe_limit_zz_th)
        init_zz_sad(c);
    if (p->prehme_enable)
        prehme_b64(c);
    if (p->enable_hme_flag) {
        if (p->enable_hme_level0_flag) hme_level0_b64(c);
        if (p->enable_hme_level1_flag) hme_level1_b64(c);
        if (p->enable_hme_level2_flag) hme_level2_b64(c);
    }
    set_final_search_centre(c);
    const SvtHipMeFrameOut *out = &job->out;
    /* ME_MCTF: a block whose first reference already matches well keeps its HME vector (:3179-3183) */
    const int tf_exit = p->me_mctf && c->sr[0][0].hme_sad < p->tf_me_exit_th;
    if (!tf_exit) {
        if (prune_ref)
            hme_prune_ref_and_adjust_sr(c);
        integer_search_b64(c);
        if (prune_ref && p->enable_me_hme_ref_pruning)
            me_prune_ref(c);
    }
    if (p->me_mctf) { /* no candidate lists, no distortion statistics (:3196) */
        memcpy(out->best_sad + (size_t)b64_index * NL * NR * 85, c->best_sad, sizeof(c->best_sad));
        memcpy(out->best_mv + (size_t)b64_index * NL * NR * 85, c->best_mv, sizeof(c->best_mv));
        memcpy(out->search_results + (size_t)b64_index * NL * NR, c->sr, sizeof(c->sr));
        return;
    }

    const uint32_t          stored = svt_hip_me_stored_pus(p);
    SbOut o = {out->me_mv_array + (size_t)b64_index * stored * p->max_refs,
               out->me_candidate_array + (size_t)b64_index * stored * p->max_cand,
               out->total_me_candidate_index + (size_t)b64_index * stored};
    if (p->num_of_ref_pic_to_search[0] == 1 && p->num_of_ref_pic_to_search[1] == 0)
        cand_single_ref(c, &o);
    else if (p->num_of_ref_pic_to_search[0] == 1 && p->num_of_ref_pic_to_search[1] == 1)
        cand_mrp_off(c, &o, p->num_of_list_to_search);
    else
        cand_general(c, &o, p->num_of_list_to_search);

    /* compute_distortion, :3034-3077 */
    uint32_t d64 = c->me_distortion[0], d32 = 0, d16 = 0, d8 = 0;
    for (int i = 0; i < 4; i++) d32 += c->me_distortion[1 + i];
    for (int i = 0; i < 16; i++) d16 += c->me_distortion[5 + i];
    for (int i = 0; i < 64; i++) d8 += c->me_distortion[21 + i];
    uint64_t mean = d8 / 64, ssq = 0;
    for (int i = 0; i < 64; i++) {
        const int64_t d = (int64_t)c->me_distortion[21 + i] - (int64_t)mean;
        ssq += (uint64_t)(d * d);
    }
    /* b64_geom width/height (pcs.c b64 geometry): the part of the block inside the (8-aligned) picture */
    const uint32_t pix = c->b64_w * c->b64_h;
    out->me_8x8_cost_variance[b64_index] = (uint32_t)(ssq / 64);
    out->rc_me_distortion[b64_index]     = p->input_resolution_le_480p ? d8 : d16;
    out->me_64x64_distortion[b64_index]  = (d64 * 4096u) / pix;
    out->me_32x32_distortion[b64_index]  = (d32 * 4096u) / pix;
    out->me_16x16_distortion[b64_index]  = (d16 * 4096u) / pix;
    out->me_8x8_distortion[b64_index]    = (d8 * 4096u) / pix;

    memcpy(out->best_sad + (size_t)b64_index * NL * NR * 85, c->best_sad, sizeof(c->best_sad));
    memcpy(out->best_mv + (size_t)b64_index * NL * NR * 85, c->best_mv, sizeof(c->best_mv));
    memcpy(out->search_results + (size_t)b64_index * NL * NR, c->sr, sizeof(c->sr));
}

int32_t orc_me_frame_range(const SvtHipMeFrameJob *job, uint32_t first, uint32_t count) {
    Ctx *c = (Ctx *)calloc(1, sizeof(Ctx));
    if (!c)
        return -1;
    c->p   = &job->prm;
    c->src = &job->src;
    c->ref = job->ref;
    c->l0_min = job->prm.hme_l0_sa_min, c->l0_max = job->prm.hme_l0_sa_max;
    const uint32_t aw = (job->src.full.width + 7) & ~7u, ah = (job->src.full.height + 7) & ~7u;
    const uint32_t bw = (aw + 63) / 64, bh = (ah + 63) / 64;
    for (uint32_t i = first; i < first + count && i < bw * bh; i++) me_b64(c, job, i, i % bw, i / bw);
    free(c);
    return 0;
}

int32_t orc_me_frames(const SvtHipMeFrameJob *jobs, uint32_t n_jobs) {
    for (uint32_t j = 0; j < n_jobs; j++) {
        int32_t rc = orc_me_frame_range(&jobs[j], 0, ~0u - 1);
        if (rc)
            return rc;
    }
    return 0;
}

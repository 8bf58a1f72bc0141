/*
 * oracle/src/orc_dlf.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the reference's deblocking filter: the sixteen edge filters
 * (deblocking_common.c:141-865), the per-edge decision set_lpf_parameters (deblocking_filter.c:162-282) and the
 * superblock schedule of svt_av1_loop_filter_frame / svt_aom_loop_filter_sb (deblocking_filter.c:547-653).
 * Pinned against the real functions through oracle/_ref (tests/test_lf_oracle.py).
 *
 * One arithmetic core serves 8-bit and high-bit-depth samples: with bd == 8 the highbd formulas reduce to the
 * 8-bit ones (shift 0, clamp to [-128,127]), which the pinning tests confirm per function.
 */
#include <stdlib.h>
#include <string.h>

#include "orc_lf.h"

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int sclamp(int t, int bd) { /* signed_char_clamp / signed_char_clamp_high (:26-35) */
    const int h = bd == 10 ? 512 : (bd == 12 ? 2048 : 128);
    return clampi(t, -h, h - 1);
}

/* filter4 / highbd_filter4 (:214-241, :426-457) on v[-2..1] around the edge (v points at q0) */
static void filter4(int mask, int thresh, int *p1, int *p0, int *q0, int *q1, int bd) {
    const int shift = bd - 8, off = 0x80 << shift, t16 = thresh << shift;
    const int ps1 = *p1 - off, ps0 = *p0 - off, qs0 = *q0 - off, qs1 = *q1 - off;
    const int hev = iabs(*p1 - *p0) > t16 || iabs(*q1 - *q0) > t16;
    int       filter = hev ? sclamp(ps1 - qs1, bd) : 0;
    filter           = mask ? sclamp(filter + 3 * (qs0 - ps0), bd) : 0;
    const int f1 = sclamp(filter + 4, bd) >> 3, f2 = sclamp(filter + 3, bd) >> 3;
    *q0    = sclamp(qs0 - f1, bd) + off;
    *p0    = sclamp(ps0 + f2, bd) + off;
    filter = hev ? 0 : ((f1 + 1) >> 1);
    *q1    = sclamp(qs1 - filter, bd) + off;
    *p1    = sclamp(ps1 + filter, bd) + off;
}

/* One sample position across an edge: s[k] for k = -7..6 with s[0] = q0.  len in {4,6,8,14}. */
static void filter_line(int *s, int len, int blimit, int limit, int thresh, int bd) {
    const int shift = bd - 8, l16 = limit << shift, b16 = blimit << shift, one = 1 << shift;
#define P(k) s[-1 - (k)]
#define Q(k) s[(k)]
    int mask = !(iabs(P(1) - P(0)) > l16 || iabs(Q(1) - Q(0)) > l16 || iabs(P(0) - Q(0)) * 2 + iabs(P(1) - Q(1)) / 2 > b16);
    if (len == 4) {
        filter4(mask, thresh, &P(1), &P(0), &Q(0), &Q(1), bd);
        return;
    }
    if (len == 6) { /* filter_mask3_chroma, flat_mask3_chroma, filter6 */
        mask = mask && !(iabs(P(2) - P(1)) > l16 || iabs(Q(2) - Q(1)) > l16);
        const int flat = !(iabs(P(1) - P(0)) > one || iabs(Q(1) - Q(0)) > one || iabs(P(2) - P(0)) > one || iabs(Q(2) - Q(0)) > one);
        if (flat && mask) {
            const int p2 = P(2), p1 = P(1), p0 = P(0), q0 = Q(0), q1 = Q(1), q2 = Q(2);
            P(1) = (p2 * 3 + p1 * 2 + p0 * 2 + q0 + 4) >> 3;
            P(0) = (p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + 4) >> 3;
            Q(0) = (p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + 4) >> 3;
            Q(1) = (p0 + q0 * 2 + q1 * 2 + q2 * 3 + 4) >> 3;
        } else
            filter4(mask, thresh, &P(1), &P(0), &Q(0), &Q(1), bd);
        return;
    }
    /* filter_mask, flat_mask4 */
    mask = mask && !(iabs(P(3) - P(2)) > l16 || iabs(P(2) - P(1)) > l16 || iabs(Q(2) - Q(1)) > l16 || iabs(Q(3) - Q(2)) > l16);
    const int flat = !(iabs(P(1) - P(0)) > one || iabs(Q(1) - Q(0)) > one || iabs(P(2) - P(0)) > one || iabs(Q(2) - Q(0)) > one ||
                       iabs(P(3) - P(0)) > one || iabs(Q(3) - Q(0)) > one);
    const int p3 = P(3), p2 = P(2), p1 = P(1), p0 = P(0), q0 = Q(0), q1 = Q(1), q2 = Q(2), q3 = Q(3);
    if (len == 14) {
        const int p6 = P(6), p5 = P(5), p4 = P(4), q4 = Q(4), q5 = Q(5), q6 = Q(6);
        const int flat2 = !(iabs(p4 - p0) > one || iabs(q4 - q0) > one || iabs(p5 - p0) > one || iabs(q5 - q0) > one ||
                            iabs(p6 - p0) > one || iabs(q6 - q0) > one);
        if (flat2 && flat && mask) {
            P(5) = (p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0 + 8) >> 4;
            P(4) = (p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1 + 8) >> 4;
            P(3) = (p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2 + 8) >> 4;
            P(2) = (p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3 + 8) >> 4;
            P(1) = (p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4 + 8) >> 4;
            P(0) = (p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5 + 8) >> 4;
            Q(0) = (p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6 + 8) >> 4;
            Q(1) = (p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2 + 8) >> 4;
            Q(2) = (p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3 + 8) >> 4;
            Q(3) = (p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4 + 8) >> 4;
            Q(4) = (p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5 + 8) >> 4;
            Q(5) = (p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7 + 8) >> 4;
            return;
        }
    }
    if (flat && mask) {
        P(2) = (p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3;
        P(1) = (p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3;
        P(0) = (p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3;
        Q(0) = (p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3;
        Q(1) = (p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3 + 4) >> 3;
        Q(2) = (p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3 + 4) >> 3;
    } else
        filter4(mask, thresh, &P(1), &P(0), &Q(0), &Q(1), bd);
#undef P
#undef Q
}

/* svt_aom_[highbd_]lpf_{vertical,horizontal}_{4,6,8,14}_c: 4 sample positions along the edge.
 * `s` = first q0 sample; `vertical` = vertical EDGE (taps run along the row). */
void orc_lpf(void *s, int32_t pitch, int blimit, int limit, int thresh, int bd, int is16, int len, int vertical) {
    const int       reach = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
    const ptrdiff_t tap = vertical ? 1 : pitch, along = vertical ? pitch : 1;
    for (int i = 0; i < 4; i++) {
        int buf[14];
        for (int k = -reach; k < reach; k++) {
            const ptrdiff_t o = i * along + k * tap;
            buf[7 + k]        = is16 ? ((uint16_t *)s)[o] : ((uint8_t *)s)[o];
        }
        filter_line(buf + 7, len, blimit, limit, thresh, bd);
        for (int k = -reach; k < reach; k++) {
            const ptrdiff_t o = i * along + k * tap;
            if (is16)
                ((uint16_t *)s)[o] = (uint16_t)buf[7 + k];
            else
                ((uint8_t *)s)[o] = (uint8_t)buf[7 + k];
        }
    }
}

/* ---- enum geometry (definitions.h TxSize / BlockSize orders), log2 of the dimension in samples ---- */
static const uint8_t tx_w_log2[19]  = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
static const uint8_t tx_h_log2[19]  = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
static const uint8_t blk_w_log2[22] = {2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 2, 4, 3, 5, 4, 6};
static const uint8_t blk_h_log2[22] = {2, 3, 2, 3, 4, 3, 4, 5, 4, 5, 6, 5, 6, 7, 6, 7, 4, 2, 5, 3, 6, 4};

/* svt_aom_update_sharpness (deblocking_common.c:582-600) + hev (deblocking_filter.c:47) for one level */
void orc_lf_thresholds(int level, int sharpness, int *lim, int *mblim, int *hev_thr) {
    int inside = level >> ((sharpness > 0) + (sharpness > 4));
    if (sharpness > 0 && inside > 9 - sharpness)
        inside = 9 - sharpness;
    if (inside < 1)
        inside = 1;
    *lim = inside, *mblim = 2 * (level + 2) + inside, *hev_thr = level >> 4;
}

/* set_lpf_parameters (deblocking_filter.c:162-282).  dir 0 = VERT_EDGE.  Returns log2 of the transform dimension
 * across the edge (the loop advance), *len = filter_length, *level = lfthr index. */
static int lpf_params(const SvtHipLfFrame *f, int dir, uint32_t x, uint32_t y, int plane, int *len, int *level) {
    const int ss = plane > 0;
    *len         = 0;
    if ((f->width >> ss) <= x || (f->height >> ss) <= y)
        return 2; /* TX_4X4 */
    const int         mi_row = ss | ((y << ss) >> 2), mi_col = ss | ((x << ss) >> 2);
    const SvtHipLfMi *mi   = f->mi + (size_t)mi_row * f->mi_stride + mi_col;
    const int         tsz  = plane ? mi->tx_size_uv : mi->tx_size_y;
    const int         ts   = dir == 0 ? tx_w_log2[tsz] : tx_h_log2[tsz];
    const uint32_t    coord = dir == 0 ? x : y;
    if (coord & ((1u << ts) - 1))
        return ts;
    const int curr_level = f->lvl[plane][mi->segment_id][dir][mi->ref_frame0][mi->mode_lf];
    int       lv         = curr_level;
    if (coord) {
        const SvtHipLfMi *pv     = dir == 0 ? mi - (1 << ss) : mi - ((size_t)f->mi_stride << ss);
        const int         ptsz   = plane ? pv->tx_size_uv : pv->tx_size_y;
        const int         pv_ts  = dir == 0 ? tx_w_log2[ptsz] : tx_h_log2[ptsz];
        const int         pv_lvl = f->lvl[plane][pv->segment_id][dir][pv->ref_frame0][pv->mode_lf];
        int               bdim   = dir == 0 ? blk_w_log2[mi->bsize] : blk_h_log2[mi->bsize];
        if (ss)
            bdim = bdim - 1 < 2 ? 2 : bdim - 1; /* ss_size_lookup[bsize][1][1] (utility.h:86-110) */
        const int pu_edge = !(coord & ((1u << bdim) - 1));
        if ((curr_level || pv_lvl) && (!pv->skip_inter || !mi->skip_inter || pu_edge)) {
            const int min_ts = ts < pv_ts ? ts : pv_ts;
            *len             = min_ts <= 2 ? 4 : (plane ? 6 : (min_ts == 3 ? 8 : 14));
            lv               = curr_level ? curr_level : pv_lvl;
        }
    }
    *level = lv;
    return ts;
}

static void *sample_ptr(const SvtHipLfFrame *f, int plane, uint32_t x, uint32_t y) {
    return (uint8_t *)f->plane[plane] + (((size_t)y * f->stride[plane] + x) << f->is_16bit);
}

/* svt_av1_filter_block_plane_vert / _horz (:287-542) for the superblock at (mi_row, mi_col) */
static void filter_sb_dir(const SvtHipLfFrame *f, int plane, int dir, uint32_t mi_row, uint32_t mi_col, int sb_mi) {
    const int ss = plane > 0, range = sb_mi >> ss;
    for (int a = 0; a < range; a++)       /* rows for vertical edges, columns for horizontal edges */
        for (int b = 0; b < range;) {     /* position along the filtering direction */
            const uint32_t x = ((mi_col * 4) >> ss) + (dir == 0 ? b : a) * 4, y = ((mi_row * 4) >> ss) + (dir == 0 ? a : b) * 4;
            int            len, level = 0;
            const int      ts = lpf_params(f, dir, x, y, plane, &len, &level);
            if (len) {
                int lim, mblim, hev;
                orc_lf_thresholds(level, f->sharpness_level, &lim, &mblim, &hev);
                orc_lpf(sample_ptr(f, plane, x, y), (int32_t)f->stride[plane], mblim, lim, hev, f->bit_depth, f->is_16bit, len, dir == 0);
            }
            b += 1 << (ts - 2);
        }
}

/* svt_av1_loop_filter_frame (:624-653) with svt_aom_loop_filter_sb's combine_vert_horz_lf schedule (:580-603) */
void orc_loop_filter_frame(const SvtHipLfFrame *f, int sb_size) {
    const int      sb_mi = sb_size >> 2;
    const uint32_t sbw = (f->mi_cols * 4 + sb_size - 1) / sb_size, sbh = (f->mi_rows * 4 + sb_size - 1) / sb_size;
    for (uint32_t sy = 0; sy < sbh; sy++)
        for (uint32_t sx = 0; sx < sbw; sx++)
            for (int plane = f->plane_start; plane < f->plane_end; plane++) {
                if (plane == 0 && !f->filter_level[0] && !f->filter_level[1])
                    break;
                if ((plane == 1 && !f->filter_level_u) || (plane == 2 && !f->filter_level_v))
                    continue;
                const uint32_t mi_row = sy * sb_mi, mi_col = sx * sb_mi;
                filter_sb_dir(f, plane, 0, mi_row, mi_col, sb_mi);
                if (sx > 0)
                    filter_sb_dir(f, plane, 1, mi_row, mi_col - sb_mi, sb_mi);
                if (sx == sbw - 1)
                    filter_sb_dir(f, plane, 1, mi_row, mi_col, sb_mi);
            }
}

size_t orc_sizeof_lf_frame(void) { return sizeof(SvtHipLfFrame); }

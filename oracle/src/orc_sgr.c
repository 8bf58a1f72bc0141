/*
 * oracle/src/orc_sgr.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the reference's self-guided restoration path:
 *   svt_av1_selfguided_restoration_c + the two internal filters + boxsum  (restoration.c:468-955)
 *   svt_apply_selfguided_restoration_c, svt_decode_xq                     (restoration.c:634-645, 957-992)
 *   svt_av1_{lowbd,highbd}_pixel_proj_error_c                              (restoration_pick.c:167-303)
 *   svt_get_proj_subspace_c, encode_xq, finer_search_pixel_proj_error,
 *   apply_sgr, search_selfguided_restoration                               (restoration_pick.c:320-652)
 * Pinned against the real functions through oracle/_ref (tests/test_sgr_oracle.py).
 *
 * Samples are addressed generically (`is16` selects uint8 / uint16); the reference's CONVERT_TO_SHORTPTR
 * pointer encoding is not reproduced — callers pass real pointers.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc_lf.h"

#define RST_BITS 4
#define PRJ_BITS 7
#define SGR_BITS 8
#define MTABLE_BITS 20
#define RECIP_BITS 12
#define RND(v, n) (((v) + ((1 << (n)) >> 1)) >> (n))

/* svt_aom_eb_sgr_params (restoration.c:85-103): the AV1 specification's Sgr_Params table */
const int32_t orc_sgr_params[16][4] = {/* r0, r1, s0, s1 */
                                       {2, 1, 140, 3236}, {2, 1, 112, 2158}, {2, 1, 93, 1618}, {2, 1, 80, 1438},
                                       {2, 1, 70, 1295},  {2, 1, 58, 1177},  {2, 1, 47, 1079}, {2, 1, 37, 996},
                                       {2, 1, 30, 925},   {2, 1, 25, 863},   {0, 1, -1, 2589}, {0, 1, -1, 1618},
                                       {0, 1, -1, 1177},  {0, 1, -1, 925},   {2, 0, 56, -1},   {2, 0, 22, -1}};

static inline int32_t px(const void *p, ptrdiff_t idx, int is16) {
    return is16 ? ((const uint16_t *)p)[idx] : ((const uint8_t *)p)[idx];
}
/* x / (x + 1) in Q8 with 0 -> 1 (svt_aom_eb_x_by_xplus1, restoration.c:647-662): round(256 x / (x + 1)) */
static inline int32_t x_by_xplus1(uint32_t z) {
    if (z == 0)
        return 1;
    if (z >= 255)
        return 256;
    return (int32_t)((256 * z + (z + 1) / 2) / (z + 1));
}
/* svt_aom_eb_one_by_x[n - 1] = round(2^12 / n) (restoration.c:664-667) */
static inline uint32_t one_by_x(int n) { return (uint32_t)((4096 + n / 2) / n); }

/* A/B of one position: box sums over (2r+1)^2 around (i, j) of the processing unit (restoration.c:709-770) */
static void ab_at(const void *dgd, ptrdiff_t stride, int is16, int i, int j, int r, uint32_t s, int bit_depth, int32_t *Aout,
                  int32_t *Bout) {
    uint32_t sum = 0, ssq = 0;
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) {
            const uint32_t v = (uint32_t)px(dgd, (ptrdiff_t)(i + dy) * stride + j + dx, is16);
            sum += v, ssq += v * v;
        }
    const uint32_t n = (uint32_t)((2 * r + 1) * (2 * r + 1));
    const uint32_t a = RND(ssq, 2 * (bit_depth - 8)), b = RND(sum, bit_depth - 8);
    const uint32_t p = (a * n < b * b) ? 0 : a * n - b * b;
    const uint32_t z = (p * s + (1u << (MTABLE_BITS - 1))) >> MTABLE_BITS; /* uint32 arithmetic as in the reference */
    const int32_t  A = x_by_xplus1(z > 255 ? 255 : z);
    *Aout            = A;
    *Bout            = (int32_t)(((uint32_t)(256 - A) * sum * one_by_x((int)n) + (1u << (RECIP_BITS - 1))) >> RECIP_BITS);
}

/* svt_av1_selfguided_restoration_c for one processing unit (w, h <= 64); dgd needs a 3-sample border */
void orc_selfguided_restoration(const void *dgd, int32_t width, int32_t height, int32_t stride, int32_t *flt0, int32_t *flt1,
                                int32_t flt_stride, int32_t ep, int32_t bit_depth, int32_t is16) {
    const int32_t *prm = orc_sgr_params[ep];
    const int      W2  = width + 2;
    int32_t       *A   = malloc(sizeof(int32_t) * (size_t)W2 * (height + 2)), *B = malloc(sizeof(int32_t) * (size_t)W2 * (height + 2));
#define AT(M, i, j) M[((i) + 1) * W2 + (j) + 1]
    if (prm[0] > 0) { /* selfguided_restoration_fast_internal (r = 2): A/B on rows -1, 1, 3, ... only */
        for (int i = -1; i < height + 1; i += 2)
            for (int j = -1; j < width + 1; j++) ab_at(dgd, stride, is16, i, j, prm[0], (uint32_t)prm[2], bit_depth, &AT(A, i, j), &AT(B, i, j));
        for (int i = 0; i < height; i++)
            for (int j = 0; j < width; j++) {
                int32_t a, b, nb;
                if (!(i & 1)) {
                    nb = 5;
                    a  = (AT(A, i - 1, j) + AT(A, i + 1, j)) * 6 + (AT(A, i - 1, j - 1) + AT(A, i + 1, j - 1) + AT(A, i - 1, j + 1) + AT(A, i + 1, j + 1)) * 5;
                    b  = (AT(B, i - 1, j) + AT(B, i + 1, j)) * 6 + (AT(B, i - 1, j - 1) + AT(B, i + 1, j - 1) + AT(B, i - 1, j + 1) + AT(B, i + 1, j + 1)) * 5;
                } else {
                    nb = 4;
                    a  = AT(A, i, j) * 6 + (AT(A, i, j - 1) + AT(A, i, j + 1)) * 5;
                    b  = AT(B, i, j) * 6 + (AT(B, i, j - 1) + AT(B, i, j + 1)) * 5;
                }
                const int32_t v             = a * px(dgd, (ptrdiff_t)i * stride + j, is16) + b;
                flt0[i * flt_stride + j] = RND(v, SGR_BITS + nb - RST_BITS);
            }
    }
    if (prm[1] > 0) { /* selfguided_restoration_internal (r = 1) */
        for (int i = -1; i < height + 1; i++)
            for (int j = -1; j < width + 1; j++) ab_at(dgd, stride, is16, i, j, prm[1], (uint32_t)prm[3], bit_depth, &AT(A, i, j), &AT(B, i, j));
        for (int i = 0; i < height; i++)
            for (int j = 0; j < width; j++) {
                const int32_t a = (AT(A, i, j) + AT(A, i, j - 1) + AT(A, i, j + 1) + AT(A, i - 1, j) + AT(A, i + 1, j)) * 4 +
                    (AT(A, i - 1, j - 1) + AT(A, i + 1, j - 1) + AT(A, i - 1, j + 1) + AT(A, i + 1, j + 1)) * 3;
                const int32_t b = (AT(B, i, j) + AT(B, i, j - 1) + AT(B, i, j + 1) + AT(B, i - 1, j) + AT(B, i + 1, j)) * 4 +
                    (AT(B, i - 1, j - 1) + AT(B, i + 1, j - 1) + AT(B, i - 1, j + 1) + AT(B, i + 1, j + 1)) * 3;
                const int32_t v             = a * px(dgd, (ptrdiff_t)i * stride + j, is16) + b;
                flt1[i * flt_stride + j] = RND(v, SGR_BITS + 5 - RST_BITS);
            }
    }
#undef AT
    free(A), free(B);
}

/* svt_decode_xq (restoration.c:634-645) */
void orc_sgr_decode_xq(const int32_t *xqd, int32_t *xq, int32_t ep) {
    const int32_t *prm = orc_sgr_params[ep];
    if (prm[0] == 0)
        xq[0] = 0, xq[1] = (1 << PRJ_BITS) - xqd[1];
    else if (prm[1] == 0)
        xq[0] = xqd[0], xq[1] = 0;
    else
        xq[0] = xqd[0], xq[1] = (1 << PRJ_BITS) - xq[0] - xqd[1];
}

/* svt_apply_selfguided_restoration_c (restoration.c:957-992); w, h <= 64 (one processing unit) */
void orc_apply_selfguided_restoration(const void *dat, int32_t width, int32_t height, int32_t stride, int32_t ep, const int32_t *xqd,
                                      void *dst, int32_t dst_stride, int32_t bit_depth, int32_t is16) {
    int32_t *flt0 = malloc(sizeof(int32_t) * (size_t)width * height * 2), *flt1 = flt0 + (size_t)width * height;
    orc_selfguided_restoration(dat, width, height, stride, flt0, flt1, width, ep, bit_depth, is16);
    const int32_t *prm = orc_sgr_params[ep];
    int32_t        xq[2];
    orc_sgr_decode_xq(xqd, xq, ep);
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const int32_t u = px(dat, (ptrdiff_t)i * stride + j, is16) << RST_BITS;
            int32_t       v = u << PRJ_BITS;
            if (prm[0] > 0)
                v += xq[0] * (flt0[i * width + j] - u);
            if (prm[1] > 0)
                v += xq[1] * (flt1[i * width + j] - u);
            const int16_t w   = (int16_t)RND(v, PRJ_BITS + RST_BITS);
            const int32_t hi  = (1 << bit_depth) - 1;
            const int32_t out = w < 0 ? 0 : (w > hi ? hi : w);
            if (is16)
                ((uint16_t *)dst)[(ptrdiff_t)i * dst_stride + j] = (uint16_t)out;
            else
                ((uint8_t *)dst)[(ptrdiff_t)i * dst_stride + j] = (uint8_t)out;
        }
    free(flt0);
}

/* svt_av1_lowbd/highbd_pixel_proj_error_c (restoration_pick.c:167-303): the two are the same arithmetic */
int64_t orc_sgr_pixel_proj_error(const void *src, int32_t width, int32_t height, int32_t src_stride, const void *dat, int32_t dat_stride,
                                 const int32_t *flt0, int32_t flt0_stride, const int32_t *flt1, int32_t flt1_stride, const int32_t *xq,
                                 int32_t ep, int32_t is16) {
    const int32_t *prm = orc_sgr_params[ep];
    int64_t        err = 0;
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const int32_t d = px(dat, (ptrdiff_t)i * dat_stride + j, is16), s = px(src, (ptrdiff_t)i * src_stride + j, is16);
            const int32_t u = d << RST_BITS;
            int32_t       v = 1 << (RST_BITS + PRJ_BITS - 1);
            if (prm[0] > 0)
                v += xq[0] * (flt0[i * flt0_stride + j] - u);
            if (prm[1] > 0)
                v += xq[1] * (flt1[i * flt1_stride + j] - u);
            const int32_t e = (prm[0] > 0 || prm[1] > 0) ? (v >> (RST_BITS + PRJ_BITS)) + d - s : d - s;
            err += (int64_t)e * e;
        }
    return err;
}

/* The five second-moment sums of svt_get_proj_subspace_c (restoration_pick.c:437-470); every term is an integer
 * below 2^32 and there are at most 2^18 of them, so the reference's double accumulation is exact. */
void orc_sgr_proj_sums(const void *src, int32_t width, int32_t height, int32_t src_stride, const void *dat, int32_t dat_stride,
                       const int32_t *flt0, int32_t flt0_stride, const int32_t *flt1, int32_t flt1_stride, int32_t ep, int32_t is16,
                       int64_t sums[5]) {
    const int32_t *prm = orc_sgr_params[ep];
    memset(sums, 0, 5 * sizeof(int64_t));
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const int64_t u = px(dat, (ptrdiff_t)i * dat_stride + j, is16) << RST_BITS;
            const int64_t s = (px(src, (ptrdiff_t)i * src_stride + j, is16) << RST_BITS) - u;
            const int64_t f1 = prm[0] > 0 ? flt0[i * flt0_stride + j] - u : 0, f2 = prm[1] > 0 ? flt1[i * flt1_stride + j] - u : 0;
            sums[0] += f1 * f1, sums[1] += f2 * f2, sums[2] += f1 * f2, sums[3] += f1 * s, sums[4] += f2 * s;
        }
}

/* the closed-form part of svt_get_proj_subspace_c (restoration_pick.c:471-506) */
void orc_sgr_solve_subspace(const int64_t sums[5], int32_t size, int32_t ep, int32_t *xq) {
    const int32_t *prm = orc_sgr_params[ep];
    double         H00 = (double)sums[0], H11 = (double)sums[1], H01 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    xq[0] = xq[1] = 0;
    H00 /= size, H01 /= size, H11 /= size, C0 /= size, C1 /= size;
    const double H10 = H01;
    if (prm[0] == 0) {
        if (H11 < 1e-8)
            return;
        xq[1] = (int32_t)rint(C1 / H11 * (1 << PRJ_BITS));
    } else if (prm[1] == 0) {
        if (H00 < 1e-8)
            return;
        xq[0] = (int32_t)rint(C0 / H00 * (1 << PRJ_BITS));
    } else {
        const double det = H00 * H11 - H01 * H10;
        if (det < 1e-8)
            return;
        const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
        xq[0] = (int32_t)rint(x0 * (1 << PRJ_BITS)), xq[1] = (int32_t)rint(x1 * (1 << PRJ_BITS));
    }
}

void orc_get_proj_subspace(const void *src, int32_t width, int32_t height, int32_t src_stride, const void *dat, int32_t dat_stride,
                           int32_t is16, const int32_t *flt0, int32_t flt0_stride, const int32_t *flt1, int32_t flt1_stride, int32_t *xq,
                           int32_t ep) {
    int64_t sums[5];
    orc_sgr_proj_sums(src, width, height, src_stride, dat, dat_stride, flt0, flt0_stride, flt1, flt1_stride, ep, is16, sums);
    orc_sgr_solve_subspace(sums, width * height, ep, xq);
}

static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
#define PRJ_MIN0 (-(1 << PRJ_BITS) * 3 / 4)
#define PRJ_MAX0 (PRJ_MIN0 + (1 << PRJ_BITS) - 1)
#define PRJ_MIN1 (-(1 << PRJ_BITS) / 4)
#define PRJ_MAX1 (PRJ_MIN1 + (1 << PRJ_BITS) - 1)

/* encode_xq (restoration_pick.c:508-520) */
void orc_sgr_encode_xq(const int32_t *xq, int32_t *xqd, int32_t ep) {
    const int32_t *prm = orc_sgr_params[ep];
    if (prm[0] == 0) {
        xqd[0] = 0;
        xqd[1] = clampi((1 << PRJ_BITS) - xq[1], PRJ_MIN1, PRJ_MAX1);
    } else if (prm[1] == 0) {
        xqd[0] = clampi(xq[0], PRJ_MIN0, PRJ_MAX0);
        xqd[1] = clampi((1 << PRJ_BITS) - xqd[0], PRJ_MIN1, PRJ_MAX1);
    } else {
        xqd[0] = clampi(xq[0], PRJ_MIN0, PRJ_MAX0);
        xqd[1] = clampi((1 << PRJ_BITS) - xqd[0] - xq[1], PRJ_MIN1, PRJ_MAX1);
    }
}

typedef struct ErrCtx {
    const void    *src, *dat;
    int32_t        width, height, src_stride, dat_stride, is16, flt_stride, ep;
    const int32_t *flt0, *flt1;
} ErrCtx;
static int64_t err_of(const ErrCtx *c, const int32_t *xqd) {
    int32_t xq[2];
    orc_sgr_decode_xq(xqd, xq, c->ep);
    return orc_sgr_pixel_proj_error(c->src, c->width, c->height, c->src_stride, c->dat, c->dat_stride, c->flt0, c->flt_stride, c->flt1,
                                    c->flt_stride, xq, c->ep, c->is16);
}
/* finer_search_pixel_proj_error (restoration_pick.c:320-411) */
static int64_t finer_search(const ErrCtx *c, int32_t start_step, int32_t *xqd, int do_refine) {
    int64_t err = err_of(c, xqd), err2;
    if (!do_refine)
        return err;
    const int32_t *prm       = orc_sgr_params[c->ep];
    const int32_t  tap_min[] = {PRJ_MIN0, PRJ_MIN1}, tap_max[] = {PRJ_MAX0, PRJ_MAX1};
    for (int32_t s = start_step; s >= 1; s >>= 1)
        for (int32_t p = 0; p < 2; ++p) {
            if ((prm[0] == 0 && p == 0) || (prm[1] == 0 && p == 1))
                continue;
            int32_t skip = 0;
            do {
                if (xqd[p] - s >= tap_min[p]) {
                    xqd[p] -= s;
                    err2 = err_of(c, xqd);
                    if (err2 > err)
                        xqd[p] += s;
                    else {
                        err = err2, skip = 1;
                        if (s == start_step)
                            continue;
                    }
                }
                break;
            } while (1);
            if (skip)
                break;
            do {
                if (xqd[p] + s <= tap_max[p]) {
                    xqd[p] += s;
                    err2 = err_of(c, xqd);
                    if (err2 > err)
                        xqd[p] -= s;
                    else {
                        err = err2;
                        if (s == start_step)
                            continue;
                    }
                }
                break;
            } while (1);
        }
    return err;
}

/* apply_sgr (restoration_pick.c:523-548): the filter over a restoration unit in processing units */
void orc_sgr_filter_unit(const void *dat, int32_t width, int32_t height, int32_t dat_stride, int32_t is16, int32_t bit_depth, int32_t pu_w,
                         int32_t pu_h, int32_t ep, int32_t *flt0, int32_t *flt1, int32_t flt_stride) {
    for (int i = 0; i < height; i += pu_h)
        for (int j = 0; j < width; j += pu_w) {
            const int   h = pu_h < height - i ? pu_h : height - i, w = pu_w < width - j ? pu_w : width - j;
            const void *d = (const uint8_t *)dat + (((ptrdiff_t)i * dat_stride + j) << is16);
            orc_selfguided_restoration(d, w, h, dat_stride, flt0 + i * flt_stride + j, flt1 + i * flt_stride + j, flt_stride, ep, bit_depth, is16);
        }
}

/* search_selfguided_restoration (restoration_pick.c:550-652) over eps start_ep, start_ep + ep_inc, ... < end_ep.
 * out = {ep, xqd0, xqd1}; returns the best error. */
int64_t orc_sgr_search_unit(const void *dat, int32_t width, int32_t height, int32_t dat_stride, const void *src, int32_t src_stride,
                            int32_t is16, int32_t bit_depth, int32_t pu_w, int32_t pu_h, int32_t start_ep, int32_t end_ep, int32_t ep_inc,
                            int32_t do_refine, int32_t out[3]) {
    const int32_t flt_stride = ((width + 7) & ~7) + 8;
    int32_t      *flt0 = malloc(sizeof(int32_t) * (size_t)flt_stride * height * 2), *flt1 = flt0 + (size_t)flt_stride * height;
    int64_t       besterr = -1;
    out[0] = out[1] = out[2] = 0;
    for (int ep = start_ep; ep < end_ep; ep += ep_inc) {
        int32_t exq[2], exqd[2];
        orc_sgr_filter_unit(dat, width, height, dat_stride, is16, bit_depth, pu_w, pu_h, ep, flt0, flt1, flt_stride);
        orc_get_proj_subspace(src, width, height, src_stride, dat, dat_stride, is16, flt0, flt_stride, flt1, flt_stride, exq, ep);
        orc_sgr_encode_xq(exq, exqd, ep);
        const ErrCtx  c   = {src, dat, width, height, src_stride, dat_stride, is16, flt_stride, ep, flt0, flt1};
        const int64_t err = finer_search(&c, 2, exqd, do_refine);
        if (besterr == -1 || err < besterr)
            besterr = err, out[0] = ep, out[1] = exqd[0], out[2] = exqd[1];
    }
    free(flt0);
    return besterr;
}

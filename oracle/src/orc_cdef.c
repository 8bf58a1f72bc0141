/*
 * oracle/src/orc_cdef.c — TEST INFRASTRUCTURE, not product code.
 * CPU restatement of the reference's CDEF kernels and per-filter-block search/apply.
 * Pinned against the real functions through oracle/_ref (tests/test_lf_oracle.py).
 */
#include "orc_lf.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BS 144        /* CDEF_BSTRIDE */
#define VL 0x7F7F     /* CDEF_VERY_LARGE */
#define VB 3
#define HB 8

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int msb(unsigned n) {
    int l = 0;
    while (n >>= 1) l++;
    return l;
}

/* cdef.c:150-210 (svt_aom_cdef_find_dir_c) */
uint8_t orc_cdef_find_dir(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift) {
    static const int32_t div_table[] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
    int32_t cost[8] = {0}, partial[8][15];
    memset(partial, 0, sizeof(partial));
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const int32_t x = (img[i * stride + j] >> coeff_shift) - 128;
            partial[0][i + j] += x;
            partial[1][i + j / 2] += x;
            partial[2][i] += x;
            partial[3][3 + i - j / 2] += x;
            partial[4][7 + i - j] += x;
            partial[5][3 - i / 2 + j] += x;
            partial[6][j] += x;
            partial[7][i / 2 + j] += x;
        }
    for (int i = 0; i < 8; i++) {
        cost[2] += partial[2][i] * partial[2][i];
        cost[6] += partial[6][i] * partial[6][i];
    }
    cost[2] *= div_table[8];
    cost[6] *= div_table[8];
    for (int i = 0; i < 7; i++) {
        cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
        cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
    }
    cost[0] += partial[0][7] * partial[0][7] * div_table[8];
    cost[4] += partial[4][7] * partial[4][7] * div_table[8];
    for (int i = 1; i < 8; i += 2) {
        for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
        cost[i] *= div_table[8];
        for (int j = 0; j < 3; j++)
            cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
    }
    int32_t best = 0;
    uint8_t dir  = 0;
    for (int i = 0; i < 8; i++)
        if (cost[i] > best)
            best = cost[i], dir = (uint8_t)i;
    *var = (best - cost[(dir + 4) & 7]) >> 10;
    return dir;
}

/* cdef.c:83-90 */
static inline int32_t constrain(int32_t diff, int32_t threshold, int32_t damping) {
    if (!threshold)
        return 0;
    const int32_t shift = imax(0, damping - msb((unsigned)threshold));
    const int32_t ad    = abs(diff);
    return (diff < 0 ? -1 : 1) * imin(ad, imax(0, threshold - (ad >> shift)));
}
/* Cdef_Directions with the +-2 padding (cdef.c:99-122), offsets for stride BS */
static const int DIRS[12][2] = {{1 * BS + 0, 2 * BS + 0},  {1 * BS + 0, 2 * BS - 1}, {-1 * BS + 1, -2 * BS + 2},
                                {0 * BS + 1, -1 * BS + 2}, {0 * BS + 1, 0 * BS + 2}, {0 * BS + 1, 1 * BS + 2},
                                {1 * BS + 1, 2 * BS + 2},  {1 * BS + 0, 2 * BS + 1}, {1 * BS + 0, 2 * BS + 0},
                                {1 * BS + 0, 2 * BS - 1},  {-1 * BS + 1, -2 * BS + 2}, {0 * BS + 1, -1 * BS + 2}};

/* cdef.c:253-307 (svt_cdef_filter_block_c); bsize: 0 4x4, 1 4x8, 2 8x4, 3 8x8 (BlockSize enum values) */
void orc_cdef_filter_block(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in, int32_t pri_strength,
                           int32_t sec_strength, int32_t dir, int32_t pri_damping, int32_t sec_damping, int32_t bsize,
                           int32_t coeff_shift, uint8_t subsampling_factor) {
    static const int32_t pri_taps_t[2][2] = {{4, 2}, {3, 3}}, sec_taps_t[2][2] = {{2, 1}, {2, 1}};
    const int32_t *pri_taps = pri_taps_t[(pri_strength >> coeff_shift) & 1];
    const int32_t *sec_taps = sec_taps_t[(pri_strength >> coeff_shift) & 1];
    const int (*D)[2] = DIRS + 2;
    const int bh = 4 << (bsize == 3 || bsize == 1), bw = 4 << (bsize == 3 || bsize == 2);
    for (int i = 0; i < bh; i += subsampling_factor)
        for (int j = 0; j < bw; j++) {
            int16_t       sum = 0;
            const int16_t x   = (int16_t)in[i * BS + j];
            int32_t       mx = x, mn = x;
            for (int k = 0; k < 2; k++) {
                const int16_t p0 = (int16_t)in[i * BS + j + D[dir][k]], p1 = (int16_t)in[i * BS + j - D[dir][k]];
                sum = (int16_t)(sum + (int16_t)(pri_taps[k] * constrain(p0 - x, pri_strength, pri_damping)));
                sum = (int16_t)(sum + (int16_t)(pri_taps[k] * constrain(p1 - x, pri_strength, pri_damping)));
                if (p0 != VL) mx = imax(p0, mx);
                if (p1 != VL) mx = imax(p1, mx);
                mn = imin(p0, mn), mn = imin(p1, mn);
                const int16_t s0 = (int16_t)in[i * BS + j + D[dir + 2][k]], s1 = (int16_t)in[i * BS + j - D[dir + 2][k]];
                const int16_t s2 = (int16_t)in[i * BS + j + D[dir - 2][k]], s3 = (int16_t)in[i * BS + j - D[dir - 2][k]];
                if (s0 != VL) mx = imax(s0, mx);
                if (s1 != VL) mx = imax(s1, mx);
                if (s2 != VL) mx = imax(s2, mx);
                if (s3 != VL) mx = imax(s3, mx);
                mn = imin(s0, mn), mn = imin(s1, mn), mn = imin(s2, mn), mn = imin(s3, mn);
                sum = (int16_t)(sum + (int16_t)(sec_taps[k] * constrain(s0 - x, sec_strength, sec_damping)));
                sum = (int16_t)(sum + (int16_t)(sec_taps[k] * constrain(s1 - x, sec_strength, sec_damping)));
                sum = (int16_t)(sum + (int16_t)(sec_taps[k] * constrain(s2 - x, sec_strength, sec_damping)));
                sum = (int16_t)(sum + (int16_t)(sec_taps[k] * constrain(s3 - x, sec_strength, sec_damping)));
            }
            int32_t y = (int16_t)x + ((8 + sum - (sum < 0)) >> 4);
            y         = y < mn ? mn : (y > mx ? mx : y);
            if (dst8)
                dst8[i * dstride + j] = (uint8_t)(int16_t)y;
            else
                dst16[i * dstride + j] = (uint16_t)(int16_t)y;
        }
}

/* enc_cdef.c:23-48 / 76-101: luma distortion of one 8xN block (double formula evaluated without contraction) */
static uint64_t dist_8xn(uint64_t sum_s, uint64_t sum_d, uint64_t sum_s2, uint64_t sum_d2, uint64_t sum_sd, int coeff_shift) {
    const uint64_t svar = sum_s2 - ((sum_s * sum_s + 32) >> 6);
    const uint64_t dvar = sum_d2 - ((sum_d * sum_d + 32) >> 6);
    return (uint64_t)floor(.5 + (sum_d2 + sum_s2 - 2 * sum_sd) * .5 * (svar + dvar + (400 << 2 * coeff_shift)) /
                                    (sqrt((20000 << 4 * coeff_shift) + svar * (double)dvar)));
}

/* enc_cdef.c:129-219: src = packed filtered blocks, dst = strided picture; `is16` selects the sample type */
uint64_t orc_compute_cdef_dist(const void *dst, int32_t dstride, const void *src, const SvtHipCdefList *dlist,
                               int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli,
                               uint8_t subsampling_factor, int is16) {
    const int bw = 4 << (bsize == 3 || bsize == 2), bh = 4 << (bsize == 3 || bsize == 1);
    const int wl = bw == 8 ? 3 : 2, hl = bh == 8 ? 3 : 2;
    uint64_t  sum = 0;
#define PIX(p, idx) (is16 ? (int32_t)((const uint16_t *)(p))[idx] : (int32_t)((const uint8_t *)(p))[idx])
    for (int bi = 0; bi < cdef_count; bi++) {
        const int    by = dlist[bi].by, bx = dlist[bi].bx;
        const size_t so = (size_t)bi << (wl + hl), dof = (size_t)(by << hl) * dstride + (bx << wl);
        if (bsize == 3 && pli == 0) {
            uint64_t ss = 0, sd = 0, ss2 = 0, sd2 = 0, ssd = 0;
            for (int i = 0; i < 8; i += subsampling_factor)
                for (int j = 0; j < 8; j++) {
                    const int32_t s = PIX(src, so + 8 * i + j), d = PIX(dst, dof + (size_t)i * dstride + j);
                    ss += (uint64_t)s, sd += (uint64_t)d;
                    ss2 += (uint64_t)(int64_t)(s * s), sd2 += (uint64_t)(int64_t)(d * d), ssd += (uint64_t)(int64_t)(s * d);
                }
            sum += dist_8xn(ss, sd, ss2, sd2, ssd, coeff_shift);
        } else {
            for (int i = 0; i < bh; i += subsampling_factor)
                for (int j = 0; j < bw; j++) {
                    const int32_t e = PIX(dst, dof + (size_t)i * dstride + j) - PIX(src, so + bw * i + j);
                    sum += (uint64_t)(int64_t)(e * e);
                }
        }
    }
    return sum >> 2 * coeff_shift;
}

/* ---- per-filter-block tile (cdef_process.c:204-221): recon samples with VERY_LARGE outside the picture ---- */
static void build_tile(uint16_t *inbuf, const SvtHipCdefPlane *pl, int fbx, int fby) {
    const int bw = 64 >> pl->xdec, bh = 64 >> pl->ydec; /* filter block size in this plane */
    for (int i = 0; i < BS * (64 + 2 * VB); i++) inbuf[i] = VL;
    uint16_t *in = inbuf + VB * BS + HB;
    for (int y = -VB; y < bh + VB; y++)
        for (int x = -HB; x < bw + HB; x++) {
            const int py = fby * bh + y, px = fbx * bw + x;
            if (py < 0 || px < 0 || py >= (int)pl->height || px >= (int)pl->width)
                continue;
            in[y * BS + x] = pl->is_16bit ? ((const uint16_t *)pl->recon)[(size_t)py * pl->recon_stride + px]
                                          : ((const uint8_t *)pl->recon)[(size_t)py * pl->recon_stride + px];
        }
}
static int build_dlist(SvtHipCdefList *dl, const uint8_t *filt, int w8, int h8, int fbx, int fby) {
    int n = 0;
    for (int r = 0; r < 8 && fby * 8 + r < h8; r++)
        for (int c = 0; c < 8 && fbx * 8 + c < w8; c++)
            if (filt[(size_t)(fby * 8 + r) * w8 + fbx * 8 + c])
                dl[n].by = (uint8_t)r, dl[n].bx = (uint8_t)c, n++;
    return n;
}
static inline int adjust_strength(int strength, int var) { /* cdef.c:130-134 */
    const int i = (var >> 6) ? imin(msb((unsigned)(var >> 6)), 12) : 0;
    return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

/* cdef_seg_search (cdef_process.c:106-349) for one plane: mse[fb][gi] = curr_mse * subsampling_factor.
 * luma (pli 0) also produces dir/var [fb][8][8]; chroma reads them. */
void orc_cdef_search_plane(const SvtHipCdefPlane *pl, const uint8_t *filt8x8, const SvtHipCdefSearchParams *prm,
                           uint64_t *mse, uint8_t *dir, int32_t *var) {
    const int lw = (int)pl->width << pl->xdec, lh = (int)pl->height << pl->ydec; /* luma size */
    const int w8 = (lw + 7) / 8, h8 = (lh + 7) / 8, nhfb = (lw + 63) / 64, nvfb = (lh + 63) / 64;
    const int bsize = pl->ydec ? (pl->xdec ? 0 : 2) : (pl->xdec ? 1 : 3);
    const int bwl = 3 - pl->xdec, bhl = 3 - pl->ydec;
    int       sub = prm->subsampling_factor;
    sub           = bsize == 3 ? imin(sub, 4) : (bsize == 0 ? imin(sub, 1) : imin(sub, 2));
    uint16_t       *inbuf = malloc(sizeof(uint16_t) * BS * (64 + 2 * VB)), *tmp = malloc(sizeof(uint16_t) * 64 * 64);
    SvtHipCdefList  dl[64];
    for (int fby = 0; fby < nvfb; fby++)
        for (int fbx = 0; fbx < nhfb; fbx++) {
            const int fb = fby * nhfb + fbx;
            const int n  = build_dlist(dl, filt8x8, w8, h8, fbx, fby);
            if (n == 0)
                continue; /* skip_cdef_seg: mse untouched */
            build_tile(inbuf, pl, fbx, fby);
            const uint16_t *in = inbuf + VB * BS + HB;
            uint8_t        *fd = dir + (size_t)fb * 64;
            int32_t        *fv = var + (size_t)fb * 64;
            if (pl->pli == 0)
                for (int bi = 0; bi < n; bi++)
                    fd[dl[bi].by * 8 + dl[bi].bx] = orc_cdef_find_dir(in + 8 * dl[bi].by * BS + 8 * dl[bi].bx, BS,
                                                                      &fv[dl[bi].by * 8 + dl[bi].bx], prm->coeff_shift);
            for (int gi = 0; gi < prm->n_strengths; gi++) {
                if (prm->strengths[gi] < 0)
                    continue;
                int pri = prm->strengths[gi] / 4, sec = prm->strengths[gi] % 4;
                sec += sec == 3;
                const int pri_s = pri << prm->coeff_shift, sec_s = sec << prm->coeff_shift;
                const int pd = prm->pri_damping + prm->coeff_shift - (pl->pli != 0);
                const int sd = prm->sec_damping + prm->coeff_shift - (pl->pli != 0);
                for (int bi = 0; bi < n; bi++) {
                    const int       by = dl[bi].by, bx = dl[bi].bx;
                    const uint16_t *bin = in + ((by * BS) << bhl) + (bx << bwl);
                    if (pri_s == 0 && sec_s == 0) { /* cdef.c:355-381: plain copy of the (sub-sampled) rows */
                        for (int iy = 0; iy < (1 << bhl); iy += sub)
                            for (int ix = 0; ix < (1 << bwl); ix++) {
                                if (pl->is_16bit)
                                    tmp[(bi << (bwl + bhl)) + (iy << bwl) + ix] = bin[iy * BS + ix];
                                else
                                    ((uint8_t *)tmp)[(bi << (bwl + bhl)) + (iy << bwl) + ix] = (uint8_t)bin[iy * BS + ix];
                            }
                        continue;
                    }
                    const int t = pl->pli ? pri_s : adjust_strength(pri_s, fv[by * 8 + bx]);
                    orc_cdef_filter_block(pl->is_16bit ? NULL : (uint8_t *)tmp + (bi << (bwl + bhl)),
                                          pl->is_16bit ? tmp + (bi << (bwl + bhl)) : NULL, 1 << bwl, bin, t, sec_s,
                                          pri_s ? fd[by * 8 + bx] : 0, pd, sd, bsize, prm->coeff_shift, (uint8_t)sub);
                }
                const size_t soff = (size_t)(fby * (64 >> pl->ydec)) * pl->source_stride + fbx * (64 >> pl->xdec);
                const void  *srcp = pl->is_16bit ? (const void *)((const uint16_t *)pl->source + soff)
                                                 : (const void *)((const uint8_t *)pl->source + soff);
                const uint64_t m = orc_compute_cdef_dist(srcp, (int32_t)pl->source_stride, tmp, dl, n, bsize, prm->coeff_shift,
                                                         pl->pli, (uint8_t)sub, pl->is_16bit);
                mse[(size_t)fb * prm->n_strengths + gi] = m * (uint64_t)sub;
            }
        }
    free(inbuf), free(tmp);
}

/* svt_av1_cdef_frame (enc_cdef.c:284-610) for one plane, out of place: `source` is the OUTPUT plane.
 * fb_strength[fb] = pri*4+sec (sec 3 -> 4 inside).  The line/column buffers of the reference only preserve
 * unfiltered neighbours, i.e. every tap reads the pre-CDEF picture — which is what reading `recon` does here. */
void orc_cdef_apply_plane(const SvtHipCdefPlane *pl, const uint8_t *filt8x8, const uint8_t *fb_strength, int damping,
                          int coeff_shift, const uint8_t *dir, const int32_t *var) {
    const int lw = (int)pl->width << pl->xdec, lh = (int)pl->height << pl->ydec;
    const int w8 = (lw + 7) / 8, h8 = (lh + 7) / 8, nhfb = (lw + 63) / 64, nvfb = (lh + 63) / 64;
    const int bsize = pl->ydec ? (pl->xdec ? 0 : 2) : (pl->xdec ? 1 : 3);
    const int bwl = 3 - pl->xdec, bhl = 3 - pl->ydec;
    uint16_t      *inbuf = malloc(sizeof(uint16_t) * BS * (64 + 2 * VB));
    SvtHipCdefList dl[64];
    /* copy-through first */
    for (uint32_t y = 0; y < pl->height; y++)
        for (uint32_t x = 0; x < pl->width; x++) {
            if (pl->is_16bit)
                ((uint16_t *)pl->source)[(size_t)y * pl->source_stride + x] = ((const uint16_t *)pl->recon)[(size_t)y * pl->recon_stride + x];
            else
                ((uint8_t *)pl->source)[(size_t)y * pl->source_stride + x] = ((const uint8_t *)pl->recon)[(size_t)y * pl->recon_stride + x];
        }
    for (int fby = 0; fby < nvfb; fby++)
        for (int fbx = 0; fbx < nhfb; fbx++) {
            const int fb = fby * nhfb + fbx;
            int       pri = fb_strength[fb] / 4, sec = fb_strength[fb] % 4;
            sec += sec == 3;
            if (pri == 0 && sec == 0)
                continue;
            const int n = build_dlist(dl, filt8x8, w8, h8, fbx, fby);
            if (n == 0)
                continue;
            build_tile(inbuf, pl, fbx, fby);
            const uint16_t *in    = inbuf + VB * BS + HB;
            const int       pri_s = pri << coeff_shift, sec_s = sec << coeff_shift;
            const int       dmp   = damping + coeff_shift - (pl->pli != 0);
            for (int bi = 0; bi < n; bi++) {
                const int    by = dl[bi].by, bx = dl[bi].bx;
                const int    t  = pl->pli ? pri_s : adjust_strength(pri_s, var[(size_t)fb * 64 + by * 8 + bx]);
                const size_t o  = (size_t)(fby * (64 >> pl->ydec) + (by << bhl)) * pl->source_stride + fbx * (64 >> pl->xdec) + (bx << bwl);
                orc_cdef_filter_block(pl->is_16bit ? NULL : (uint8_t *)pl->source + o, pl->is_16bit ? (uint16_t *)pl->source + o : NULL,
                                      (int32_t)pl->source_stride, in + ((by * BS) << bhl) + (bx << bwl), t, sec_s,
                                      pri_s ? dir[(size_t)fb * 64 + by * 8 + bx] : 0, dmp, dmp, bsize, coeff_shift, 1);
            }
        }
    free(inbuf);
}

/*
 * oracle/src/orc_sad.c — TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C restatement of the reference's SAD / pyramid / variance leaf kernels, used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for the HIP
 * path.  Pinned against the real reference C functions (oracle/_ref/libsvtref.so, built from
 * /root/reference by oracle/Makefile) in tests/test_oracle_vs_ref.py and against the golden
 * vectors in tests/golden/.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include "orc.h"

#include <stdlib.h>
#include <string.h>

static inline uint32_t absdiff_u8(uint8_t a, uint8_t b) { return a > b ? (uint32_t)(a - b) : (uint32_t)(b - a); }

/* Source/Lib/C_DEFAULT/compute_sad_c.c:20-36 (svt_fast_loop_nxm_sad_kernel) and :209-212
 * (svt_nxm_sad_kernel_helper_c, the C entry of the svt_nxm_sad_kernel pointer). */
uint32_t orc_nxm_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                     uint32_t height, uint32_t width) {
    uint32_t acc = 0;
    for (uint32_t r = 0; r < height; r++)
        for (uint32_t c = 0; c < width; c++)
            acc += absdiff_u8(src[(size_t)r * src_stride + c], ref[(size_t)r * ref_stride + c]);
    return acc;
}

/* Source/Lib/C_DEFAULT/compute_sad_c.c:58-101 (svt_sad_loop_kernel_c).
 * First minimum in raster order wins (strict <); best starts at 0xffffff and x/y are left
 * untouched when nothing beats it; rows with even index are skipped (not searched, but ref still
 * advances) when skip_search_line && width==16 && height<=16. */
void orc_sad_loop_kernel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                         uint32_t block_height, uint32_t block_width, uint64_t *best_sad,
                         int16_t *x_search_center, int16_t *y_search_center, uint32_t src_stride_raw,
                         uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height) {
    const int skip_even = skip_search_line && block_width == 16 && block_height <= 16;
    uint64_t  best      = 0xffffff;
    for (int sy = 0; sy < search_area_height; sy++) {
        const uint8_t *row = ref + (size_t)sy * src_stride_raw;
        if (skip_even && (sy & 1) == 0)
            continue;
        for (int sx = 0; sx < search_area_width; sx++) {
            uint32_t sad = orc_nxm_sad(src, src_stride, row + sx, ref_stride, block_height, block_width);
            if (sad < best) {
                best             = sad;
                *x_search_center = (int16_t)sx;
                *y_search_center = (int16_t)sy;
            }
        }
    }
    *best_sad = best;
}

/* 8x8 SAD, full (motion_estimation.c:69-91) or on rows 0,2,4,6 doubled (sub_sad; the callers pass
 * 2*stride to svt_aom_compute8x4_sad_kernel_c and shift the result left by 1,
 * motion_estimation.c:105-125, 224-281). */
static uint32_t sad8x8(const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs, int sub_sad) {
    if (sub_sad)
        return orc_nxm_sad(src, 2 * ss, ref, 2 * rs, 4, 8) << 1;
    return orc_nxm_sad(src, ss, ref, rs, 8, 8);
}

static inline uint32_t pack_mv(int16_t x, int16_t y) { return ((uint32_t)(uint16_t)y << 16) | (uint16_t)x; }

/* motion_estimation.c:98-164 (svt_ext_sad_calculation_8x8_16x16_c): one 16x16 block, one position. */
void orc_ext_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                       uint32_t ref_stride, uint32_t *p_best_sad_8x8,
                                       uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                       uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16,
                                       uint32_t *p_sad8x8, uint8_t sub_sad) {
    uint32_t total = 0;
    for (int q = 0; q < 4; q++) {
        const size_t so = (size_t)(q >> 1) * 8 * src_stride + (q & 1) * 8;
        const size_t ro = (size_t)(q >> 1) * 8 * ref_stride + (q & 1) * 8;
        p_sad8x8[q]     = sad8x8(src + so, src_stride, ref + ro, ref_stride, sub_sad);
        if (p_sad8x8[q] < p_best_sad_8x8[q]) {
            p_best_sad_8x8[q] = p_sad8x8[q];
            p_best_mv8x8[q]   = mv;
        }
        total += p_sad8x8[q];
    }
    if (total < p_best_sad_16x16[0]) {
        p_best_sad_16x16[0] = total;
        p_best_mv16x16[0]   = mv;
    }
    *p_sad16x16 = total;
}

/* motion_estimation.c:171-205 (svt_ext_sad_calculation_32x32_64x64_c). */
void orc_ext_sad_calculation_32x32_64x64(const uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32,
                                         uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                         uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32) {
    uint32_t total = 0;
    for (int k = 0; k < 4; k++) {
        uint32_t s = p_sad16x16[4 * k] + p_sad16x16[4 * k + 1] + p_sad16x16[4 * k + 2] + p_sad16x16[4 * k + 3];
        p_sad32x32[k] = s;
        if (s < p_best_sad_32x32[k]) {
            p_best_sad_32x32[k] = s;
            p_best_mv32x32[k]   = mv;
        }
        total += s;
    }
    if (total < p_best_sad_64x64[0]) {
        p_best_sad_64x64[0] = total;
        p_best_mv64x64[0]   = mv;
    }
}

/* z-order slot of the 16x16 block at raster (x,y): motion_estimation.c:341. */
const uint8_t orc_z16[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};

/* motion_estimation.c:335-362 + :210-333 (svt_ext_all_sad_calculation_8x8_16x16_c): the 64x64
 * block at 8 consecutive x positions.  8x8 children of 16x16 slot z live at 4z..4z+3. */
void orc_ext_all_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                           uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8,
                                           uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                           uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8],
                                           uint32_t p_eight_sad8x8[64][8], uint8_t sub_sad) {
    (void)p_eight_sad8x8; /* never written by the reference either (:218) */
    const int16_t mvx = (int16_t)(mv & 0xffff), mvy = (int16_t)(mv >> 16);
    for (int by = 0; by < 4; by++)
        for (int bx = 0; bx < 4; bx++) {
            const uint32_t z  = orc_z16[4 * by + bx];
            const uint8_t *s  = src + (size_t)16 * by * src_stride + 16 * bx;
            const uint8_t *rf = ref + (size_t)16 * by * ref_stride + 16 * bx;
            for (int p = 0; p < 8; p++) {
                const uint32_t cand  = pack_mv((int16_t)(mvx + (int16_t)p), mvy);
                uint32_t       total = 0;
                for (int q = 0; q < 4; q++) {
                    const size_t so = (size_t)(q >> 1) * 8 * src_stride + (q & 1) * 8;
                    const size_t ro = (size_t)(q >> 1) * 8 * ref_stride + (q & 1) * 8 + p;
                    uint32_t     v  = sad8x8(s + so, src_stride, rf + ro, ref_stride, sub_sad);
                    if (v < p_best_sad_8x8[4 * z + q]) {
                        p_best_sad_8x8[4 * z + q] = v;
                        p_best_mv8x8[4 * z + q]   = cand;
                    }
                    total += v;
                }
                p_eight_sad16x16[z][p] = total;
                if (total < p_best_sad_16x16[z]) {
                    p_best_sad_16x16[z] = total;
                    p_best_mv16x16[z]   = cand;
                }
            }
        }
}

/* motion_estimation.c:369-425 (svt_ext_eight_sad_calculation_32x32_64x64_c). */
void orc_ext_eight_sad_calculation_32x32_64x64(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32,
                                               uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                               uint32_t *p_best_mv64x64, uint32_t mv,
                                               uint32_t p_sad32x32[4][8]) {
    const int16_t mvx = (int16_t)(mv & 0xffff), mvy = (int16_t)(mv >> 16);
    for (int p = 0; p < 8; p++) {
        const uint32_t cand  = pack_mv((int16_t)(mvx + (int16_t)p), mvy);
        uint32_t       total = 0;
        for (int k = 0; k < 4; k++) {
            uint32_t s = p_sad16x16[4 * k][p] + p_sad16x16[4 * k + 1][p] + p_sad16x16[4 * k + 2][p] +
                p_sad16x16[4 * k + 3][p];
            p_sad32x32[k][p] = s;
            if (s < p_best_sad_32x32[k]) {
                p_best_sad_32x32[k] = s;
                p_best_mv32x32[k]   = cand;
            }
            total += s;
        }
        if (total < p_best_sad_64x64[0]) {
            p_best_sad_64x64[0] = total;
            p_best_mv64x64[0]   = cand;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Pyramid                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Source/Lib/Codec/pic_analysis_process.c:131-161 (svt_aom_downsample_2d_c): output (i,j) is the
 * rounded mean of the 2x2 input samples whose bottom-right corner is
 * (step/2 + i*step, step/2 + j*step). */
void orc_downsample_2d(const uint8_t *in, uint32_t in_stride, uint32_t in_w, uint32_t in_h, uint8_t *out,
                       uint32_t out_stride, uint32_t step) {
    const uint32_t half = step >> 1;
    uint32_t       oy   = 0;
    for (uint32_t y = half; y < in_h; y += step, oy++) {
        uint32_t ox = 0;
        for (uint32_t x = half; x < in_w; x += step, ox++) {
            const uint8_t *p = in + (size_t)y * in_stride + x;
            uint32_t       s = p[-(ptrdiff_t)in_stride - 1] + p[-(ptrdiff_t)in_stride] + p[-1] + p[0];
            out[(size_t)oy * out_stride + ox] = (uint8_t)((s + 2) >> 2);
        }
    }
}

/* Source/Lib/Codec/pic_operators.c:338-383 (svt_aom_generate_padding): replicate left/right
 * columns over the picture rows, then copy whole padded rows up and down. */
void orc_generate_padding(uint8_t *buf, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_w,
                          uint32_t pad_h) {
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *row = buf + (size_t)(pad_h + y) * stride + pad_w;
        memset(row - pad_w, row[0], pad_w);
        memset(row + w, row[w - 1], pad_w);
    }
    uint8_t *top = buf + (size_t)pad_h * stride;
    uint8_t *bot = buf + (size_t)(pad_h + h - 1) * stride;
    for (uint32_t k = 1; k <= pad_h; k++) {
        memcpy(top - (size_t)k * stride, top, stride);
        memcpy(bot + (size_t)k * stride, bot, stride);
    }
}

/* Source/Lib/Codec/pic_analysis_process.c:1922-1979 (svt_aom_downsample_filtering_input_picture).
 * NB the destination offset uses org_x for BOTH axes (:1934,:1953) — kept. */
void orc_pyramid_frame(const SvtHipPlane8 *full, const SvtHipPlane8 *quarter, const SvtHipPlane8 *sixteenth,
                       int hme_level1_enabled) {
    const uint8_t *fsrc = full->buf + full->org_x + (size_t)full->org_y * full->stride;
    if (hme_level1_enabled) {
        orc_downsample_2d(fsrc, full->stride, full->width, full->height,
                          quarter->buf + quarter->org_x + (size_t)quarter->org_x * quarter->stride,
                          quarter->stride, 2);
        orc_generate_padding(quarter->buf, quarter->stride, quarter->width, quarter->height, quarter->org_x,
                             quarter->org_y);
        orc_downsample_2d(quarter->buf + quarter->org_x + (size_t)quarter->org_y * quarter->stride,
                          quarter->stride, quarter->width, quarter->height,
                          sixteenth->buf + sixteenth->org_x + (size_t)sixteenth->org_x * sixteenth->stride,
                          sixteenth->stride, 2);
    } else {
        orc_downsample_2d(fsrc, full->stride, full->width, full->height,
                          sixteenth->buf + sixteenth->org_x + (size_t)sixteenth->org_x * sixteenth->stride,
                          sixteenth->stride, 4);
    }
    orc_generate_padding(sixteenth->buf, sixteenth->stride, sixteenth->width, sixteenth->height,
                         sixteenth->org_x, sixteenth->org_y);
}

/* ------------------------------------------------------------------------------------------ */
/* Block mean / variance                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* pic_analysis_process.c:234-251 (svt_compute_sub_mean_8x8_c): rows 0,2,4,6; sum<<3. */
uint64_t orc_compute_sub_mean_8x8(const uint8_t *in, uint16_t stride) {
    uint64_t s = 0;
    for (int r = 0; r < 8; r += 2)
        for (int c = 0; c < 8; c++) s += in[(size_t)r * stride + c];
    return s << 3;
}

/* pic_analysis_process.c:253-272 (svt_aom_compute_sub_mean_squared_values_c) for 8x8: sum<<11. */
static uint64_t sub_mean_sq_8x8(const uint8_t *in, uint32_t stride) {
    uint64_t s = 0;
    for (int r = 0; r < 8; r += 2)
        for (int c = 0; c < 8; c++) {
            uint32_t v = in[(size_t)r * stride + c];
            s += v * v;
        }
    return s << 11;
}

/* pic_analysis_process.c:191-208 (svt_compute_mean_c): (sum << 8) / (w*h). */
uint64_t orc_compute_mean(const uint8_t *in, uint32_t stride, uint32_t w, uint32_t h) {
    uint64_t s = 0;
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++) s += in[(size_t)r * stride + c];
    return (s << 8) / (w * h);
}

/* pic_analysis_process.c:213-232 (svt_compute_mean_squared_values_c): (sum_sq << 16) / (w*h). */
uint64_t orc_compute_mean_squared_values(const uint8_t *in, uint32_t stride, uint32_t w, uint32_t h) {
    uint64_t s = 0;
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++) {
            uint32_t v = in[(size_t)r * stride + c];
            s += v * v;
        }
    return (s << 16) / (w * h);
}

/* pic_analysis_process.c:274-301 (svt_compute_interm_var_four8x8_c): four 8x8 side by side. */
void orc_compute_interm_var_four8x8(const uint8_t *in, uint16_t stride, uint64_t *mean, uint64_t *mean_sq) {
    for (int k = 0; k < 4; k++) {
        mean[k]    = orc_compute_sub_mean_8x8(in + 8 * k, stride);
        mean_sq[k] = sub_mean_sq_8x8(in + 8 * k, stride);
    }
}

/* pic_analysis_process.c:307-1382 (compute_block_mean_compute_variance) for one 64x64 block.
 * 8x8 means in raster order; each parent = (sum of its 4 children) >> 2; variance of a block =
 * (mean_sq - mean*mean) >> 16, truncated to uint16.  Output order = EbMeTierZeroPu raster:
 * [0] 64x64, [1..4] 32x32, [5..20] 16x16, [21..84] 8x8. */
void orc_block_mean_variance_b64(const uint8_t *blk, uint32_t stride, int full_precision, uint16_t var[85],
                                 uint64_t mean[85]) {
    uint64_t m8[64], q8[64], m16[16], q16[16], m32[4], q32[4], m64, q64;
    for (int by = 0; by < 8; by++)
        for (int bx = 0; bx < 8; bx++) {
            const uint8_t *p = blk + (size_t)8 * by * stride + 8 * bx;
            if (full_precision) {
                m8[8 * by + bx] = orc_compute_mean(p, stride, 8, 8);
                q8[8 * by + bx] = orc_compute_mean_squared_values(p, stride, 8, 8);
            } else {
                m8[8 * by + bx] = orc_compute_sub_mean_8x8(p, (uint16_t)stride);
                q8[8 * by + bx] = sub_mean_sq_8x8(p, stride);
            }
        }
    for (int by = 0; by < 4; by++)
        for (int bx = 0; bx < 4; bx++) {
            int a = 16 * by + 2 * bx;
            m16[4 * by + bx] = (m8[a] + m8[a + 1] + m8[a + 8] + m8[a + 9]) >> 2;
            q16[4 * by + bx] = (q8[a] + q8[a + 1] + q8[a + 8] + q8[a + 9]) >> 2;
        }
    for (int by = 0; by < 2; by++)
        for (int bx = 0; bx < 2; bx++) {
            int a = 8 * by + 2 * bx;
            m32[2 * by + bx] = (m16[a] + m16[a + 1] + m16[a + 4] + m16[a + 5]) >> 2;
            q32[2 * by + bx] = (q16[a] + q16[a + 1] + q16[a + 4] + q16[a + 5]) >> 2;
        }
    m64 = (m32[0] + m32[1] + m32[2] + m32[3]) >> 2;
    q64 = (q32[0] + q32[1] + q32[2] + q32[3]) >> 2;

    mean[0] = m64;
    var[0]  = (uint16_t)((q64 - m64 * m64) >> 16);
    for (int i = 0; i < 4; i++) {
        mean[1 + i] = m32[i];
        var[1 + i]  = (uint16_t)((q32[i] - m32[i] * m32[i]) >> 16);
    }
    for (int i = 0; i < 16; i++) {
        mean[5 + i] = m16[i];
        var[5 + i]  = (uint16_t)((q16[i] - m16[i] * m16[i]) >> 16);
    }
    for (int i = 0; i < 64; i++) {
        mean[21 + i] = m8[i];
        var[21 + i]  = (uint16_t)((q8[i] - m8[i] * m8[i]) >> 16);
    }
}

/* pic_analysis_process.c:1533-1553 (compute_picture_spatial_statistics): every b64 of the frame.
 * Blocks on the right / bottom edge read into the padding exactly like the reference. */
void orc_variance_frame(const SvtHipPlane8 *full, uint16_t *variance, uint64_t *mean, int full_precision) {
    const uint32_t bw = (full->width + 63) / 64, bh = (full->height + 63) / 64;
    for (uint32_t by = 0; by < bh; by++)
        for (uint32_t bx = 0; bx < bw; bx++) {
            const uint8_t *p = full->buf + (size_t)(full->org_y + 64 * by) * full->stride + full->org_x + 64 * bx;
            uint64_t       m[85];
            orc_block_mean_variance_b64(p, full->stride, full_precision, variance + (size_t)85 * (by * bw + bx), m);
            if (mean)
                memcpy(mean + (size_t)85 * (by * bw + bx), m, sizeof(m));
        }
}

/* layout checks for the ctypes mirrors (tests/test_abi.py) */
ORC_API uint32_t orc_sizeof_me_params(void) { return (uint32_t)sizeof(SvtHipMeParams); }
ORC_API uint32_t orc_sizeof_me_job(void) { return (uint32_t)sizeof(SvtHipMeFrameJob); }
ORC_API uint32_t orc_sizeof_plane(void) { return (uint32_t)sizeof(SvtHipPlane8); }

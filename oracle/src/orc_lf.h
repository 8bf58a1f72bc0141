/* oracle/src/orc_lf.h — TEST INFRASTRUCTURE (CPU restatement of the in-loop filter part of the hot path). */
#ifndef ORC_LF_H
#define ORC_LF_H
#include <stddef.h>
#include <stdint.h>

#include "../../include/svt_hip_lf.h"
#ifdef __cplusplus
extern "C" {
#endif
#define ORC_API __attribute__((visibility("default")))

ORC_API uint8_t  orc_cdef_find_dir(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift);
ORC_API void     orc_cdef_filter_block(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in, int32_t pri_strength,
                                       int32_t sec_strength, int32_t dir, int32_t pri_damping, int32_t sec_damping, int32_t bsize,
                                       int32_t coeff_shift, uint8_t subsampling_factor);
ORC_API uint64_t orc_compute_cdef_dist(const void *dst, int32_t dstride, const void *src, const SvtHipCdefList *dlist,
                                       int32_t cdef_count, int32_t bsize, int32_t coeff_shift, int32_t pli,
                                       uint8_t subsampling_factor, int is16);
ORC_API void     orc_cdef_search_plane(const SvtHipCdefPlane *pl, const uint8_t *filt8x8, const SvtHipCdefSearchParams *prm,
                                       uint64_t *mse, uint8_t *dir, int32_t *var);
ORC_API void     orc_cdef_apply_plane(const SvtHipCdefPlane *pl, const uint8_t *filt8x8, const uint8_t *fb_strength, int damping,
                                      int coeff_shift, const uint8_t *dir, const int32_t *var);
/* deblocking (orc_dlf.c) */
ORC_API void   orc_lpf(void *s, int32_t pitch, int blimit, int limit, int thresh, int bd, int is16, int len, int vertical);
ORC_API void   orc_lf_thresholds(int level, int sharpness, int *lim, int *mblim, int *hev_thr);
ORC_API void   orc_loop_filter_frame(const SvtHipLfFrame *f, int sb_size); /* f->plane / f->mi are HOST pointers here */
ORC_API size_t orc_sizeof_lf_frame(void);
/* self-guided restoration (orc_sgr.c) */
ORC_API extern const int32_t orc_sgr_params[16][4];
ORC_API void    orc_selfguided_restoration(const void *dgd, int32_t width, int32_t height, int32_t stride, int32_t *flt0, int32_t *flt1,
                                           int32_t flt_stride, int32_t ep, int32_t bit_depth, int32_t is16);
ORC_API void    orc_sgr_decode_xq(const int32_t *xqd, int32_t *xq, int32_t ep);
ORC_API void    orc_sgr_encode_xq(const int32_t *xq, int32_t *xqd, int32_t ep);
ORC_API void    orc_apply_selfguided_restoration(const void *dat, int32_t width, int32_t height, int32_t stride, int32_t ep,
                                                 const int32_t *xqd, void *dst, int32_t dst_stride, int32_t bit_depth, int32_t is16);
ORC_API int64_t orc_sgr_pixel_proj_error(const void *src, int32_t width, int32_t height, int32_t src_stride, const void *dat,
                                         int32_t dat_stride, const int32_t *flt0, int32_t flt0_stride, const int32_t *flt1,
                                         int32_t flt1_stride, const int32_t *xq, int32_t ep, int32_t is16);
ORC_API void    orc_sgr_proj_sums(const void *src, int32_t width, int32_t height, int32_t src_stride, const void *dat, int32_t dat_stride,
                                  const int32_t *flt0, int32_t flt0_stride, const int32_t *flt1, int32_t flt1_stride, int32_t ep,
                                  int32_t is16, int64_t sums[5]);
ORC_API void    orc_sgr_solve_subspace(const int64_t sums[5], int32_t size, int32_t ep, int32_t *xq);
ORC_API void    orc_get_proj_subspace(const void *src, int32_t width, int32_t height, int32_t src_stride, const void *dat,
                                      int32_t dat_stride, int32_t is16, const int32_t *flt0, int32_t flt0_stride, const int32_t *flt1,
                                      int32_t flt1_stride, int32_t *xq, int32_t ep);
ORC_API void    orc_sgr_filter_unit(const void *dat, int32_t width, int32_t height, int32_t dat_stride, int32_t is16, int32_t bit_depth,
                                    int32_t pu_w, int32_t pu_h, int32_t ep, int32_t *flt0, int32_t *flt1, int32_t flt_stride);
ORC_API int64_t orc_sgr_search_unit(const void *dat, int32_t width, int32_t height, int32_t dat_stride, const void *src,
                                    int32_t src_stride, int32_t is16, int32_t bit_depth, int32_t pu_w, int32_t pu_h, int32_t start_ep,
                                    int32_t end_ep, int32_t ep_inc, int32_t do_refine, int32_t out[3]);
/* Wiener restoration (orc_wiener.c) */
ORC_API void orc_wiener_compute_stats(int32_t wiener_win, const void *dgd, const void *src, int32_t h_start, int32_t h_end,
                                      int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H,
                                      int32_t is16, int32_t bit_depth);
ORC_API void orc_wiener_convolve_add_src(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, const int16_t *fx,
                                         const int16_t *fy, int32_t w, int32_t h, int32_t round_0, int32_t round_1, int32_t bd,
                                         int32_t is16);
/* inter-prediction interpolation, single reference (orc_convolve.c) */
ORC_API void orc_convolve_sr(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, int32_t w, int32_t h,
                             const int16_t *fx, int32_t taps_x, const int16_t *fy, int32_t taps_y, int32_t round_0,
                             int32_t round_1, int32_t bd, int32_t is16);
ORC_API void orc_convolve_jnt(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, int32_t w, int32_t h, const int16_t *fx,
                              int32_t taps_x, const int16_t *fy, int32_t taps_y, int32_t round_0, int32_t round_1, int32_t bd,
                              int32_t is16, uint16_t *cbuf, int32_t cbuf_stride, int32_t mode, int32_t fwd, int32_t bck);
#ifdef __cplusplus
}
#endif
#endif

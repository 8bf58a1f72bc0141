/*
 * svt_hip_lf.h — C-ABI for the in-loop filters: CDEF, deblocking, self-guided restoration
 * (SURVEY.md §8 rows a9–a11).
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   Source/Lib/Codec/common_dsp_rtcd.c:796-804   svt_aom_cdef_find_dir, svt_aom_cdef_find_dir_dual,
 *                                                svt_cdef_filter_block, svt_aom_copy_rect8_8bit_to_16bit
 *   Source/Lib/Codec/aom_dsp_rtcd.c:207-208      svt_compute_cdef_dist_16bit, svt_compute_cdef_dist_8bit
 *   Source/Lib/Codec/cdef_process.c:106-349      cdef_seg_search            (Tier B: svt_hip_cdef_search_plane)
 *   Source/Lib/Codec/enc_cdef.c:284-610          svt_av1_cdef_frame         (Tier B: svt_hip_cdef_apply_plane)
 */
#ifndef SVT_HIP_LF_H
#define SVT_HIP_LF_H

#include <stddef.h>

#include "svt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_HIP_CDEF_BSTRIDE 144      /* CDEF_BSTRIDE, cdef.h:35 */
#define SVT_HIP_CDEF_VERY_LARGE 0x7F7F /* CDEF_VERY_LARGE, cdef.h:38 */

typedef struct SvtHipCdefList { /* CdefList, definitions.h:255-259 */
    uint8_t by, bx;
} SvtHipCdefList;

/* ---------------------------------------------------------------------------------------------
 * Tier A (host pointers, RTCD signatures; BlockSize values: BLOCK_4X4=0, 4X8=1, 8X4=2, 8X8=3)
 * ------------------------------------------------------------------------------------------- */
SVT_HIP_API uint8_t svt_aom_cdef_find_dir_hip(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift);
SVT_HIP_API void    svt_aom_cdef_find_dir_dual_hip(const uint16_t *img1, const uint16_t *img2, int stride, int32_t *var1,
                                                   int32_t *var2, int32_t coeff_shift, uint8_t *out1, uint8_t *out2);
SVT_HIP_API void    svt_cdef_filter_block_hip(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in,
                                              int32_t pri_strength, int32_t sec_strength, int32_t dir, int32_t pri_damping,
                                              int32_t sec_damping, int32_t bsize, int32_t coeff_shift,
                                              uint8_t subsampling_factor);
SVT_HIP_API void    svt_aom_copy_rect8_8bit_to_16bit_hip(uint16_t *dst, int32_t dstride, const uint8_t *src, int32_t sstride,
                                                         int32_t v, int32_t h);
SVT_HIP_API uint64_t svt_compute_cdef_dist_16bit_hip(const uint16_t *dst, int32_t dstride, const uint16_t *src,
                                                     const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize,
                                                     int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor);
SVT_HIP_API uint64_t svt_compute_cdef_dist_8bit_hip(const uint8_t *dst8, int32_t dstride, const uint8_t *src8,
                                                    const SvtHipCdefList *dlist, int32_t cdef_count, int32_t bsize,
                                                    int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor);

/* svt_search_one_dual (aom_dsp_rtcd.h:239; enc_cdef.c:627-686): one greedy step of the joint luma / chroma strength
 * search over the per-filter-block tables mse[0][i][strength], mse[1][i][strength]. */
SVT_HIP_API uint64_t svt_search_one_dual_hip(int *lev0, int *lev1, int nb_strengths, uint64_t **mse[2], int sb_count,
                                             int start_gi, int end_gi);

/* ---------------------------------------------------------------------------------------------
 * Tier B — one picture plane per call, device pointers.
 *   recon / source : top-left sample of the picture area (no padding needed; samples outside the
 *                    8-aligned picture are CDEF_VERY_LARGE as in the reference)
 *   filt8x8        : uint8 [ceil(h8)][ceil(w8)] in units of 8x8 LUMA blocks, non-zero = block is filtered
 *                    (what svt_sb_compute_cdef_list derives from the skip flags, enc_cdef.c:238-276)
 * ------------------------------------------------------------------------------------------- */
typedef struct SvtHipCdefPlane {
    void    *recon;       /* uint8 or uint16 samples */
    void    *source;      /* search: the original picture; apply: the OUTPUT plane (must not alias recon) */
    uint32_t recon_stride, source_stride; /* in samples */
    uint32_t width, height;               /* plane size in samples (luma: 8-aligned picture size; chroma: half) */
    uint8_t  is_16bit, xdec, ydec, pli;   /* pli: 0 luma, 1/2 chroma; xdec == ydec (4:2:0 or 4:4:4) — the reference
                                           * encoder accepts 4:2:0 only (enc_settings.c:447); others: BAD_PARAMETER */
} SvtHipCdefPlane;

#define SVT_HIP_CDEF_MAX_STRENGTHS 64
typedef struct SvtHipCdefSearchParams {
    int32_t n_strengths;
    int8_t  strengths[SVT_HIP_CDEF_MAX_STRENGTHS]; /* pri*4+sec as in default_first/second_pass_fs; -1 = not tested */
    int32_t pri_damping, sec_damping;              /* 3 + (base_q_idx >> 6), cdef_process.c:139-140 */
    int32_t coeff_shift;                           /* bit_depth - 8 */
    int32_t subsampling_factor;                    /* cdef_ctrls->subsampling_factor (capped per block size inside) */
} SvtHipCdefSearchParams;

/* mse[fb][gi] (uint64, fb raster over 64x64 filter blocks) = curr_mse * subsampling_factor of
 * cdef_process.c:283-286 for this plane; dir / var: [fb][8][8] luma direction / variance (uint8 / int32),
 * written when pli == 0 and read when pli != 0 (run the luma plane first). */
SVT_HIP_API int32_t svt_hip_cdef_search_plane(const SvtHipCdefPlane *plane, const uint8_t *d_filt8x8,
                                              const SvtHipCdefSearchParams *prm, uint64_t *d_mse, uint8_t *d_dir,
                                              int32_t *d_var, void *stream);

/* Apply: strength per filter block (uint8 [n_fb], pri*4+sec with sec already in {0,1,2,3}->{0,1,2,4} mapping
 * applied inside); blocks that are not filtered are copied through. */
SVT_HIP_API int32_t svt_hip_cdef_apply_plane(const SvtHipCdefPlane *plane, const uint8_t *d_filt8x8,
                                             const uint8_t *d_fb_strength, int32_t damping, int32_t coeff_shift,
                                             const uint8_t *d_dir, const int32_t *d_var, void *stream);

/* The same for up to three planes of one picture in ONE launch (svt_av1_cdef_frame, enc_cdef.c:284-610, walks the planes
 * of a filter block together): planes[p] / fb_strength_dptrs[p] per plane (luma strength index for plane 0, chroma for 1, 2);
 * all planes must cover the same 64x64 (luma) filter-block grid.  `planes` and `fb_strength_dptrs` are HOST arrays of
 * n_planes entries (the latter holds DEVICE pointers, one strength array per plane); everything else is device memory. */
SVT_HIP_API int32_t svt_hip_cdef_apply_frame(const SvtHipCdefPlane *planes, uint32_t n_planes, const uint8_t *d_filt8x8,
                                             const uint8_t *const *fb_strength_dptrs, int32_t damping, int32_t coeff_shift,
                                             const uint8_t *d_dir, const int32_t *d_var, void *stream);

/* =============================================================================================
 * Deblocking
 *   Source/Lib/Codec/common_dsp_rtcd.h:1043-1074  svt_aom_lpf_{horizontal,vertical}_{4,6,8,14} and the highbd set
 *   Source/Lib/Codec/deblocking_filter.c:162-282  set_lpf_parameters
 *   Source/Lib/Codec/deblocking_filter.c:287-653  svt_av1_filter_block_plane_vert/horz, svt_aom_loop_filter_sb,
 *                                                 svt_av1_loop_filter_frame   (Tier B: svt_hip_loop_filter_frame)
 *   Source/Lib/Codec/deblocking_common.c:582-600  svt_aom_update_sharpness (lim / mblim; hev_thr = level >> 4,
 *                                                 deblocking_filter.c:47)
 * ============================================================================================= */

/* Tier A: RTCD signatures, host pointers.  `s` points at the first q0 sample of a 4-sample edge segment;
 * the call reads/writes up to 7 samples on either side of the edge (what the reference touches). */
#define SVT_HIP_DECL_LPF(dir, n)                                                                                              \
    SVT_HIP_API void svt_aom_lpf_##dir##_##n##_hip(uint8_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit,    \
                                                   const uint8_t *thresh);                                                    \
    SVT_HIP_API void svt_aom_highbd_lpf_##dir##_##n##_hip(uint16_t *s, int32_t pitch, const uint8_t *blimit,                  \
                                                          const uint8_t *limit, const uint8_t *thresh, int32_t bd);
SVT_HIP_DECL_LPF(horizontal, 4)
SVT_HIP_DECL_LPF(horizontal, 6)
SVT_HIP_DECL_LPF(horizontal, 8)
SVT_HIP_DECL_LPF(horizontal, 14)
SVT_HIP_DECL_LPF(vertical, 4)
SVT_HIP_DECL_LPF(vertical, 6)
SVT_HIP_DECL_LPF(vertical, 8)
SVT_HIP_DECL_LPF(vertical, 14)
#undef SVT_HIP_DECL_LPF

/* One record per 4x4 luma mode-info unit: the fields set_lpf_parameters reads through pcs->mi_grid_base,
 * gathered into a flat array (INTEGRATION.md shows the gather loop).  Enum values are the reference's
 * BlockSize / TxSize (definitions.h). */
typedef struct SvtHipLfMi {
    uint8_t bsize;      /* mbmi->block_mi.bsize */
    uint8_t tx_size_y;  /* tx_depth_to_tx_size[skip_inter ? 0 : tx_depth][bsize]   (get_transform_size, :145-150) */
    uint8_t tx_size_uv; /* av1_get_max_uv_txsize(bsize, 1, 1) */
    uint8_t skip_inter; /* block_mi.skip && is_inter_block_no_intrabc(ref_frame[0]) */
    uint8_t segment_id; /* block_mi.segment_id */
    uint8_t ref_frame0; /* block_mi.ref_frame[0] (INTRA_FRAME = 0) */
    uint8_t mode_lf;    /* mode_lf_lut[block_mi.mode] (deblocking_common.h:33-37) */
    uint8_t reserved;
} SvtHipLfMi;

typedef struct SvtHipLfFrame {
    void             *plane[3];  /* device; top-left picture sample of Y, Cb, Cr (4:2:0).  Filtered IN PLACE.  The buffer
                                  * must be padded like the reference's recon pictures: whole 4x4 units are filtered and
                                  * taps reach 7 samples past an edge, also below / right of the picture. */
    uint32_t          stride[3]; /* in samples */
    uint32_t          width, height; /* unpadded luma size (plane_ptr->dst.width/height, :90-96); chroma is >> 1 */
    const SvtHipLfMi *mi;        /* device; [mi_rows][mi_stride] */
    uint32_t          mi_stride, mi_rows, mi_cols; /* mi_cols = aligned_width >> 2, mi_rows = aligned_height >> 2 */
    uint8_t           lvl[3][8][2][8][2]; /* LoopFilterInfoN.lvl after svt_av1_loop_filter_frame_init */
    uint8_t           filter_level[2], filter_level_u, filter_level_v; /* frm_hdr.loop_filter_params: plane on/off */
    uint8_t           sharpness_level;
    uint8_t           bit_depth; /* static_config.encoder_bit_depth: 8 / 10 */
    uint8_t           is_16bit;  /* samples are uint16 (is_16bit_pipeline or bit_depth > 8) */
    uint8_t           plane_start, plane_end; /* as svt_av1_loop_filter_frame(.., plane_start, plane_end) */
    uint8_t           reserved[3];
} SvtHipLfFrame;

/* Whole-frame deblocking: all vertical edges, then all horizontal edges, per plane (the order
 * svt_aom_loop_filter_sb's combine_vert_horz_lf schedule is equivalent to).  delta_lf is not supported —
 * the reference never enables it (resource_coordination_process.c:410-412). */
SVT_HIP_API int32_t svt_hip_loop_filter_frame(const SvtHipLfFrame *frame, void *stream);

/* =============================================================================================
 * Self-guided restoration
 *   Source/Lib/Codec/common_dsp_rtcd.h:181,185   svt_apply_selfguided_restoration, svt_av1_selfguided_restoration
 *   Source/Lib/Codec/aom_dsp_rtcd.h:79,81,212    svt_av1_{lowbd,highbd}_pixel_proj_error, svt_get_proj_subspace
 *   Source/Lib/Codec/restoration_pick.c:523-652  apply_sgr, search_selfguided_restoration (Tier B: svt_hip_sgr_*)
 * ============================================================================================= */
typedef struct SvtHipSgrParams { /* SgrParamsType, definitions.h:1758-1761 */
    int32_t r[2], s[2];
} SvtHipSgrParams;

/* Tier A: RTCD signatures.  For highbd != 0 the uint8_t pointers carry the reference's CONVERT_TO_BYTEPTR encoding
 * (address >> 1 of a uint16 buffer, definitions.h:953-954), exactly as the callers pass them. */
SVT_HIP_API void    svt_av1_selfguided_restoration_hip(const uint8_t *dgd8, int32_t width, int32_t height, int32_t dgd_stride,
                                                       int32_t *flt0, int32_t *flt1, int32_t flt_stride, int32_t sgr_params_idx,
                                                       int32_t bit_depth, int32_t highbd);
SVT_HIP_API void    svt_apply_selfguided_restoration_hip(const uint8_t *dat, int32_t width, int32_t height, int32_t stride, int32_t eps,
                                                         const int32_t *xqd, uint8_t *dst, int32_t dst_stride, int32_t *tmpbuf,
                                                         int32_t bit_depth, int32_t highbd);
SVT_HIP_API int64_t svt_av1_lowbd_pixel_proj_error_hip(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride,
                                                       const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride,
                                                       int32_t *flt1, int32_t flt1_stride, int32_t xq[2],
                                                       const SvtHipSgrParams *params);
SVT_HIP_API int64_t svt_av1_highbd_pixel_proj_error_hip(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride,
                                                        const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride,
                                                        int32_t *flt1, int32_t flt1_stride, int32_t xq[2],
                                                        const SvtHipSgrParams *params);
SVT_HIP_API void    svt_get_proj_subspace_hip(const uint8_t *src8, int width, int height, int src_stride, const uint8_t *dat8,
                                              int dat_stride, int use_highbitdepth, int32_t *flt0, int flt0_stride, int32_t *flt1,
                                              int flt1_stride, int *xq, const SvtHipSgrParams *params);

/* Tier B: one restoration unit (<= 384 x 384 samples for the search; filter / apply take any region up to a whole
 * plane whose origin lies on the processing-unit grid), device pointers.  `dat` = the degraded picture (after
 * deblocking + CDEF) at the unit's top-left sample, readable 3 samples beyond the unit on every side (the extended
 * frame the reference searches on); `src` = the original. */
typedef struct SvtHipSgrUnit {
    const void *dat, *src;
    uint32_t    dat_stride, src_stride; /* in samples */
    uint32_t    width, height;
    uint8_t     is_16bit, bit_depth;
    uint8_t     pu_w, pu_h; /* processing-unit size: 64 (luma) or 32 (4:2:0 chroma), restoration_pick.c:561-562 */
} SvtHipSgrUnit;

/* apply_sgr (restoration_pick.c:523-548): flt0 / flt1 = int32 [height][flt_stride] */
SVT_HIP_API int32_t svt_hip_sgr_filter_unit(const SvtHipSgrUnit *unit, int32_t ep, int32_t *d_flt0, int32_t *d_flt1,
                                            uint32_t flt_stride, void *stream);
/* svt_apply_selfguided_restoration over the unit, filter and projection fused (no flt arrays); d_dst has the
 * sample type of `dat`.  Stripe-boundary handling of the final loop-restoration pass stays with the caller. */
SVT_HIP_API int32_t svt_hip_sgr_apply_unit(const SvtHipSgrUnit *unit, int32_t ep, const int32_t xqd[2], void *d_dst,
                                           uint32_t dst_stride, void *stream);
/* search_selfguided_restoration (restoration_pick.c:550-652) for ep = start_ep, start_ep + ep_inc, ... < end_ep:
 * filter, projection (get_proj_subspace + encode_xq) and the finer search all run on the device; the call returns
 * after reading back out = {ep, xqd[0], xqd[1]} and the winning error.  d_work: svt_hip_sgr_search_work_bytes(). */
SVT_HIP_API size_t  svt_hip_sgr_search_work_bytes(uint32_t width, uint32_t height, int32_t n_ep);
SVT_HIP_API int32_t svt_hip_sgr_search_unit(const SvtHipSgrUnit *unit, int32_t start_ep, int32_t end_ep, int32_t ep_inc,
                                            int32_t do_refine, void *d_work, int32_t out[3], int64_t *best_err, void *stream);

/* =============================================================================================
 * Wiener restoration (what runs at presets M4-M8)
 *   Source/Lib/Codec/aom_dsp_rtcd.h:66,68        svt_av1_compute_stats, svt_av1_compute_stats_highbd
 *   Source/Lib/Codec/common_dsp_rtcd.h:177,179   svt_av1_wiener_convolve_add_src, svt_av1_highbd_wiener_convolve_add_src
 *   callers: search_wiener (restoration_pick.c:1296-1430), wiener_filter_stripe (restoration.c:432-458)
 * ============================================================================================= */
typedef struct SvtHipConvolveParams { /* ConvolveParams, definitions.h:580-593 (only round_0 / round_1 are read) */
    int32_t ref, do_average;
    void   *dst;
    int32_t dst_stride, round_0, round_1, plane, is_compound, use_jnt_comp_avg, fwd_offset, bck_offset, use_dist_wtd_comp_avg;
} SvtHipConvolveParams;

/* Tier A: RTCD signatures (highbd: CONVERT_TO_BYTEPTR-encoded pointers; bit_depth = EbBitDepth value 8 / 10 / 12). */
SVT_HIP_API void svt_av1_compute_stats_hip(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end,
                                           int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H);
SVT_HIP_API void svt_av1_compute_stats_highbd_hip(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start,
                                                  int32_t h_end, int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride,
                                                  int64_t *M, int64_t *H, int32_t bit_depth);
SVT_HIP_API void svt_av1_wiener_convolve_add_src_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride,
                                                     const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h,
                                                     const SvtHipConvolveParams *conv_params);
SVT_HIP_API void svt_av1_highbd_wiener_convolve_add_src_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst,
                                                            ptrdiff_t dst_stride, const int16_t *filter_x, const int16_t *filter_y,
                                                            int32_t w, int32_t h, const SvtHipConvolveParams *conv_params, int32_t bd);

/* Tier B: statistics of many restoration units in one launch chain.  dgd / src point at sample (0,0) of the planes the
 * limits refer to (device memory; dgd readable wiener_win/2 samples beyond every limit); M = int64 [n][49],
 * H = int64 [n][49*49] (device), laid out as the reference's (only the first win^2 rows / columns are used). */
typedef struct SvtHipWienerUnit {
    const void *dgd, *src;
    uint32_t    dgd_stride, src_stride; /* in samples */
    int32_t     h_start, h_end, v_start, v_end; /* RestorationTileLimits */
} SvtHipWienerUnit;
SVT_HIP_API int32_t svt_hip_wiener_stats(const SvtHipWienerUnit *units /* host */, uint32_t n_units, int32_t wiener_win, int32_t is_16bit,
                                         int32_t bit_depth, int64_t *d_M, int64_t *d_H, void *stream);
/* The separable filter over a w x h region (any size; the reference's callers tile it in <= 64 x 64 processing units,
 * which gives the same samples).  Output must not alias the input. */
SVT_HIP_API int32_t svt_hip_wiener_convolve(const void *d_src, uint32_t src_stride, void *d_dst, uint32_t dst_stride, uint32_t w,
                                            uint32_t h, const int16_t filter_x[8], const int16_t filter_y[8], int32_t is_16bit,
                                            int32_t bit_depth, void *stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Frame-level loop restoration: svt_av1_loop_restoration_filter_frame (restoration.c:1179-1248) = for every restoration
 * unit svt_av1_loop_restoration_filter_unit (:1067-1147) = for every 64-row processing stripe of the unit
 * {svt_aom_setup_processing_stripe_boundary (:288-384) -> stripe filter (Wiener :437-466, :1017-1038 or self-guided
 * :994-1015, :1040-1062) -> svt_aom_restore_processing_stripe_boundary (:386-435)}.
 *
 * The reference patches the three rows above / below a stripe INTO the picture, filters, and patches them back.  On the
 * device the same rows are selected while the stripe's tile is assembled in LDS (nothing is ever written into the source
 * plane), one workgroup per (64 >> ss_x)-wide processing unit of a stripe, all planes in one launch:
 *   rows above the first stripe / below the last stripe / left and right of the picture: edge replication (svt_extend_frame);
 *   other stripes, optimized_lr == 0: two saved DEBLOCKED rows from stripe_boundary_above / _below, used 0,0,1 / 0,1,1
 *                                     (svt_av1_loop_restoration_save_boundary_lines wrote them; row 2*stripe + k, element j is
 *                                     picture column j - SVT_HIP_LR_EXTRA_HORZ);
 *   optimized_lr == 1: the picture's own rows, the outermost of the three duplicated from its neighbour.
 * src and dst must be different planes; RESTORE_NONE units are copied. */
#define SVT_HIP_LR_EXTRA_HORZ 4 /* RESTORATION_EXTRA_HORZ (restoration.h) */
typedef struct SvtHipLrUnit {   /* RestorationUnitInfo (restoration.h:176-181) */
    uint8_t restoration_type;   /* 0 RESTORE_NONE, 1 RESTORE_WIENER, 2 RESTORE_SGRPROJ */
    uint8_t ep;                 /* sgrproj_info.ep */
    int16_t pad_;
    int32_t xqd[2];             /* sgrproj_info.xqd */
    int16_t hfilter[8], vfilter[8]; /* wiener_info (InterpKernel: 7 taps + 0, centre tap stored minus 128) */
} SvtHipLrUnit;
typedef struct SvtHipLrPlane {
    const void *src;            /* device: CDEF output, pointer to sample (0, 0) */
    void       *dst;            /* device: restored plane, sample (0, 0) */
    uint32_t    src_stride, dst_stride; /* samples */
    uint32_t    width, height;  /* plane size (cropped) */
    uint8_t     ss_x, ss_y, is_16bit, bit_depth;
    uint32_t    unit_size;      /* rsi->restoration_unit_size of this plane */
    uint32_t    horz_units, vert_units; /* rsi->horz_units_per_tile, vert_units_per_tile */
    const SvtHipLrUnit *units;  /* device [vert_units][horz_units] */
    const void *boundary_above, *boundary_below; /* device: rsb->stripe_boundary_above / _below (NULL when optimized_lr) */
    uint32_t    boundary_stride; /* samples */
    uint32_t    optimized_lr;
} SvtHipLrPlane;
SVT_HIP_API int32_t svt_hip_restoration_filter_frame(const SvtHipLrPlane *planes, uint32_t n_planes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_LF_H */

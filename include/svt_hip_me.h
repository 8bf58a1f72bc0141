/*
 * svt_hip_me.h — C-ABI for the open-loop motion-estimation + pyramid/variance part of the hot
 * path (SURVEY.md §8 rows a1–a5).
 *
 * Reference interfaces replaced (all paths relative to /root/reference):
 *   Source/Lib/Codec/aom_dsp_rtcd.h:776      svt_sad_loop_kernel
 *   Source/Lib/Codec/aom_dsp_rtcd.h:838      downsample_2d
 *   Source/Lib/Codec/aom_dsp_rtcd.h:839,845  svt_ext_sad_calculation_8x8_16x16 / _32x32_64x64
 *   Source/Lib/Codec/aom_dsp_rtcd.h:850,851  svt_ext_all_sad_calculation_8x8_16x16 /
 *                                            svt_ext_eight_sad_calculation_32x32_64x64
 *   Source/Lib/Codec/aom_dsp_rtcd.h:852,854  svt_initialize_buffer_32bits / svt_nxm_sad_kernel
 *   Source/Lib/Codec/aom_dsp_rtcd.h:853,861,866  svt_nxm_sad_kernel_sub_sampled / sad_16b_kernel /
 *                                            svt_pme_sad_loop_kernel
 *   Source/Lib/Codec/aom_dsp_rtcd.h:856-860  svt_compute_mean_8x8, _mean_square_values_8x8,
 *                                            _sub_mean_8x8, svt_compute_interm_var_four8x8
 *   Source/Lib/Codec/motion_estimation.c:3146   svt_aom_motion_estimation_b64   (Tier B: whole frame)
 *   Source/Lib/Codec/me_process.c:174-290       the b64 loop of the ME kernel   (Tier B: whole frame)
 *   Source/Lib/Codec/pic_analysis_process.c:1922-1979  svt_aom_downsample_filtering_input_picture
 *   Source/Lib/Codec/pic_analysis_process.c:1533-1553  compute_picture_spatial_statistics
 */
#ifndef SVT_HIP_ME_H
#define SVT_HIP_ME_H

#include "svt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Tier A — ABI-identical per-call entry points (host pointers).  `Bool` of the reference is
 * uint8_t (Source/Lib/Codec/definitions.h).
 * ---------------------------------------------------------------------------------------- */
SVT_HIP_API void svt_sad_loop_kernel_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref,
                                         uint32_t ref_stride, uint32_t block_height,
                                         uint32_t block_width, uint64_t *best_sad,
                                         int16_t *x_search_center, int16_t *y_search_center,
                                         uint32_t src_stride_raw, uint8_t skip_search_line,
                                         int16_t search_area_width, int16_t search_area_height);

SVT_HIP_API uint32_t svt_nxm_sad_kernel_hip(const uint8_t *src, uint32_t src_stride,
                                            const uint8_t *ref, uint32_t ref_stride,
                                            uint32_t height, uint32_t width);

/* svt_nxm_sad_kernel_sub_sampled (aom_dsp_rtcd.h:853): the generic-C row of the reference's table binds this pointer to
 * the plain N x M SAD (aom_dsp_rtcd.c:1213), which is what this computes. */
SVT_HIP_API uint32_t svt_nxm_sad_kernel_sub_sampled_hip(const uint8_t *src, uint32_t src_stride,
                                                        const uint8_t *ref, uint32_t ref_stride,
                                                        uint32_t height, uint32_t width);
/* sad_16b_kernel (aom_dsp_rtcd.h:861; the C function is svt_aom_sad_16b_kernel_c, compute_sad_c.c:39).
 * svt_hip_rtcd_lookup("sad_16b_kernel") resolves to this export. */
SVT_HIP_API uint32_t svt_aom_sad_16b_kernel_hip(uint16_t *src, uint32_t src_stride, uint16_t *ref,
                                                uint32_t ref_stride, uint32_t height, uint32_t width);
/* svt_initialize_buffer_32bits (aom_dsp_rtcd.h:852): count128 * 4 + count32 words set to `value`. */
SVT_HIP_API void svt_initialize_buffer_32bits_hip(uint32_t *pointer, uint32_t count128, uint32_t count32,
                                                  uint32_t value);

/* Mirrors of MV (block_structures.h:26-29) and of struct svt_mv_cost_param (mcomp.h:37-49): same field order, types
 * and padding, so that a `const struct svt_mv_cost_param *` can be passed as is.  mv_cost_type holds MV_COST_TYPE
 * (mcomp.h:29-36: ENTROPY 0, L1_LOWRES 1, L1_MIDRES 2, L1_HDRES 3, OPT 4, NONE 5); mvcost[0] / mvcost[1] point at the
 * centre of the row / column cost tables (valid index range -(1 << 14) .. (1 << 14)). */
typedef struct SvtHipMv {
    int16_t row, col;
} SvtHipMv;
typedef struct SvtHipMvCostParam {
    const SvtHipMv *ref_mv;
    SvtHipMv        full_ref_mv;
    uint8_t         mv_cost_type;
    const int      *mvjcost;
    const int      *mvcost[2];
    int             error_per_bit;
    int             early_exit_th;
    int             sad_per_bit;
} SvtHipMvCostParam;
/* svt_pme_sad_loop_kernel (aom_dsp_rtcd.h:866; product_coding_loop.c:1781-1828): SAD + motion-vector cost over the
 * sparse search grid (eight consecutive columns, then a jump of search_step; rows search_step apart); best_cost /
 * best_mvx / best_mvy are updated only by a strictly smaller cost, the first position in scan order winning a tie. */
SVT_HIP_API void svt_pme_sad_loop_kernel_hip(const SvtHipMvCostParam *mv_cost_params, uint8_t *src,
                                             uint32_t src_stride, uint8_t *ref, uint32_t ref_stride,
                                             uint32_t block_height, uint32_t block_width, uint32_t *best_cost,
                                             int16_t *best_mvx, int16_t *best_mvy,
                                             int16_t search_position_start_x, int16_t search_position_start_y,
                                             int16_t search_area_width, int16_t search_area_height,
                                             int16_t search_step, int16_t mvx, int16_t mvy);

SVT_HIP_API void svt_ext_all_sad_calculation_8x8_16x16_hip(
    uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t mv,
    uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
    uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8], uint32_t p_eight_sad8x8[64][8],
    uint8_t sub_sad);

SVT_HIP_API void svt_ext_eight_sad_calculation_32x32_64x64_hip(
    uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64,
    uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]);

SVT_HIP_API void svt_ext_sad_calculation_8x8_16x16_hip(
    uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride,
    uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
    uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16, uint32_t *p_sad8x8,
    uint8_t sub_sad);

SVT_HIP_API void svt_ext_sad_calculation_32x32_64x64_hip(
    uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64,
    uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32);

SVT_HIP_API void svt_aom_downsample_2d_hip(uint8_t *input_samples, uint32_t input_stride,
                                           uint32_t input_area_width, uint32_t input_area_height,
                                           uint8_t *decim_samples, uint32_t decim_stride,
                                           uint32_t decim_step);

SVT_HIP_API void svt_compute_interm_var_four8x8_hip(uint8_t *input_samples, uint16_t input_stride,
                                                    uint64_t *mean_of8x8_blocks,
                                                    uint64_t *mean_of_squared8x8_blocks);
SVT_HIP_API uint64_t svt_compute_sub_mean_8x8_hip(uint8_t *input_samples, uint16_t input_stride);
SVT_HIP_API uint64_t svt_compute_mean_8x8_hip(uint8_t *input_samples, uint32_t input_stride,
                                              uint32_t input_area_width, uint32_t input_area_height);
SVT_HIP_API uint64_t svt_compute_mean_square_values_8x8_hip(uint8_t *input_samples,
                                                            uint32_t input_stride,
                                                            uint32_t input_area_width,
                                                            uint32_t input_area_height);

/* ------------------------------------------------------------------------------------------
 * Tier B — batched search: many svt_sad_loop_kernel calls in one launch.  Offsets index into one
 * device arena `base` (so a descriptor is position-independent).  Results: best_sad (u64),
 * x, y (i16) per descriptor, exactly what the per-call pointer would have produced.
 * ---------------------------------------------------------------------------------------- */
typedef struct SvtHipSadLoopDesc {
    uint64_t src_off, ref_off;       /* byte offsets of src block / search-window origin in `base` */
    uint32_t src_stride, ref_stride; /* as passed to svt_sad_loop_kernel (already x2 when sub-sampled) */
    uint32_t src_stride_raw;         /* ref advance per search row */
    uint16_t block_width, block_height;
    int16_t  search_area_width, search_area_height;
    uint8_t  skip_search_line;
    uint8_t  pad_[7];
} SvtHipSadLoopDesc;

typedef struct SvtHipSadLoopResult {
    uint64_t best_sad;
    int16_t  x, y;
    uint32_t pad_;
} SvtHipSadLoopResult;

SVT_HIP_API int32_t svt_hip_sad_loop_batch(const uint8_t *d_base, const SvtHipSadLoopDesc *d_desc,
                                           SvtHipSadLoopResult *d_result, uint32_t n_desc,
                                           void *stream);

/* ------------------------------------------------------------------------------------------
 * Tier B — pyramid + variance for one picture (replaces pic_analysis_process.c:2126,2137).
 * `full` is the padded input luma; quarter / sixteenth are written including their padding
 * (svt_aom_generate_padding, pic_operators.c:338-383).  `variance` receives 85 uint16 per b64 in
 * EbMeTierZeroPu raster order (me_context.h:52-138) and `mean` 85 uint64 Q8 block means
 * (the reference's pcs->variance[b64][85], pic_analysis_process.c:1111-1380; y_mean is not kept
 * by the reference but costs nothing here).  All pointers are device pointers.
 * ---------------------------------------------------------------------------------------- */
SVT_HIP_API int32_t svt_hip_pyramid_frame(const SvtHipPlane8 *full, const SvtHipPlane8 *quarter,
                                          const SvtHipPlane8 *sixteenth, int32_t hme_level1_enabled,
                                          void *stream);
SVT_HIP_API int32_t svt_hip_variance_frame(const SvtHipPlane8 *full, uint16_t *d_variance,
                                           uint64_t *d_mean, int32_t full_precision, void *stream);
SVT_HIP_API int32_t svt_hip_pad_plane(const SvtHipPlane8 *plane, void *stream);

/* ------------------------------------------------------------------------------------------
 * Tier B — open-loop ME of whole pictures (replaces the loop me_process.c:174-290 calling
 * svt_aom_motion_estimation_b64).
 *
 * SvtHipMeParams is a flat copy of every MeContext / pcs / scs field that
 * svt_aom_motion_estimation_b64 reads (me_context.h:366-509); the encoder fills it right after
 * svt_aom_sig_deriv_me (enc_mode_config.c:684-823) — see INTEGRATION.md for the field map.
 * ---------------------------------------------------------------------------------------- */
#define SVT_HIP_ME_MAX_LIST   2  /* MAX_NUM_OF_REF_PIC_LIST, definitions.h:2349 */
#define SVT_HIP_ME_MAX_REF    4  /* MAX_REF_IDX == REF_LIST_MAX_DEPTH, definitions.h:2364 */
#define SVT_HIP_ME_SQUARE_PUS 85 /* SQUARE_PU_COUNT, me_sb_results.h:23 */

typedef struct SvtHipSearchArea { uint16_t width, height; } SvtHipSearchArea;

typedef struct SvtHipMeParams {
    /* search methods: 0 = SUB_SAD_SEARCH, 1 = FULL_SAD_SEARCH (definitions.h:2079-2080) */
    uint8_t hme_search_method, me_search_method;
    uint8_t enable_hme_flag, enable_hme_level0_flag, enable_hme_level1_flag, enable_hme_level2_flag;
    uint8_t num_hme_sa_w, num_hme_sa_h; /* must be 2x2 (get_worst_quadrant, motion_estimation.c:1942) */
    SvtHipSearchArea hme_l0_sa_min, hme_l0_sa_max, hme_l1_sa, hme_l2_sa;
    SvtHipSearchArea me_sa_min, me_sa_max;
    /* PreHmeCtrls (me_context.h) */
    uint8_t prehme_enable, prehme_skip_search_line, prehme_l1_early_exit;
    uint8_t me_mctf; /* 1: me_ctx->me_type == ME_MCTF (the temporal filter's call of svt_aom_motion_estimation_b64,
                      * temporal_filtering.c:3075): picture distance not scaled in the full-pel search area
                      * (motion_estimation.c:1302), no HME / ME reference pruning (:3173), early exit on tf_me_exit_th
                      * (:3179-3183), no candidate lists and no distortion statistics (:3196); 0: ME_OPEN_LOOP */
    SvtHipSearchArea prehme_sa_min[2], prehme_sa_max[2];
    /* MeHmeRefPruneCtrls */
    uint8_t  enable_me_hme_ref_pruning, pad1_;
    uint16_t prune_ref_if_hme_sad_dev_bigger_than_th, prune_ref_if_me_sad_dev_bigger_than_th;
    uint16_t zz_sad_pct, phme_sad_pct;
    uint16_t tf_me_exit_th; /* me_ctx->tf_me_exit_th (me_mctf only): a b64 whose search_results[0][0].hme_sad is below it skips
                             * the full-pel search; best_sad / best_mv stay zero, the caller reads hme_sad from search_results */
    uint32_t zz_sad_th, phme_sad_th;
    /* MeSrCtrls */
    uint8_t  enable_me_sr_adjustment, distance_based_hme_resizing;
    uint16_t reduce_me_sr_based_on_mv_length_th, stationary_hme_sad_abs_th, stationary_me_sr_divisor;
    uint16_t reduce_me_sr_based_on_hme_sad_abs_th, me_sr_divisor_for_low_hme_sad;
    /* Me8x8VarCtrls */
    uint8_t  me_8x8_var_enabled, pad3_[3];
    uint32_t me_sr_div4_th, me_sr_div2_th, me_sr_mult2_th;
    /* MvBasedSearchAdj */
    uint8_t  mv_sa_adj_enabled, mv_sa_adj_nearest_ref_only;
    uint16_t mv_sa_adj_mv_size_th, mv_sa_adj_sa_multiplier;
    uint8_t  reduce_hme_l0_sr_th_min, reduce_hme_l0_sr_th_max;
    uint32_t me_early_exit_th, me_safe_limit_zz_th, prev_me_stage_based_exit_th;
    int32_t  prune_me_candidates_th;
    uint8_t  use_best_unipred_cand_only;
    /* per picture (me_process.c:218-227, pcs fields read inside ME) */
    uint8_t  num_of_list_to_search;
    uint8_t  num_of_ref_pic_to_search[SVT_HIP_ME_MAX_LIST];
    uint8_t  temporal_layer_index, is_ref, hierarchical_levels, similar_brightness_refs;
    uint8_t  enable_me_8x8, enable_me_16x16, max_number_of_pus_per_sb /* 85 */;
    uint8_t  max_cand, max_refs, max_l0;  /* pcs->pa_me_data (pcs.h:493-495) */
    uint8_t  only_l_bwd;                  /* scs->mrp_ctrls.only_l_bwd */
    uint8_t  input_resolution_le_480p;    /* scs->input_resolution <= INPUT_SIZE_480p_RANGE */
    uint8_t  pad4_[2];
    uint64_t picture_number;
    uint64_t ref_picture_number[SVT_HIP_ME_MAX_LIST][SVT_HIP_ME_MAX_REF];
} SvtHipMeParams;

/* Number of PUs whose candidates are stored (size of the per-b64 me_results arrays,
 * pcs.c:108-117): 85, 21 (no 8x8) or 5 (no 16x16). */
static inline uint32_t svt_hip_me_stored_pus(const SvtHipMeParams *p) {
    return p->enable_me_16x16 ? (p->enable_me_8x8 ? 85u : 21u) : 5u;
}

/* Memory contract of the picture planes (svt_hip_analysis_frames, svt_hip_pyramid_frame, svt_hip_variance_frame,
 * svt_hip_me_frames[_dev]): a plane occupies exactly stride * (height + 2 * org_y) bytes from `buf` (the layout of
 * EbPictureBufferDesc, pic_buffer_desc.h:34-75) and NOTHING outside those bytes is ever read or written — a plane may end
 * at the last byte of an allocation.  This holds because svt_hip_me_validate_jobs enforces the reference's paddings
 * (org_x, org_y >= 64 / 32 / 16 for the full / quarter / sixteenth plane; enc_handle.c:1276,1292,1308 allocate 68 / 32 /
 * 16), and with those the search-area clamps of the reference (motion_estimation.c:837-888, 1442-1561) keep every
 * search window at least one whole row above the plane's last row: the window stagers fetch 16-byte chunks and may
 * read up to 31 bytes past the last byte a window row needs, which then are the first bytes of the following row.
 * tests/test_gpu_me.py::test_me_frame_exact_size_planes pins this (planes packed back to back, results unchanged).
 *
 * svt_hip_sad_loop_batch (arbitrary windows in a caller-built arena) cannot make that argument: the arena must be
 * readable for 64 bytes past the last byte addressed by any descriptor (window or source block). */
typedef struct SvtHipPyramid8 {
    SvtHipPlane8 full, quarter, sixteenth; /* input_padded_pic, quarter_/sixteenth_downsampled_picture_ptr */
} SvtHipPyramid8;

/* Batched picture analysis: svt_hip_pyramid_frame + svt_hip_variance_frame for n pictures in three launches
 * (the picture-analysis kernel's per-picture work, pic_analysis_process.c:2126,2137, for a whole mini-GOP at once).
 * `jobs` is a HOST array; the planes and output arrays it names are device memory. */
typedef struct SvtHipAnalysisJob {
    SvtHipPyramid8 pyr;
    uint16_t      *variance; /* [n_b64][85] */
    uint64_t      *mean;     /* [n_b64][85] or NULL */
} SvtHipAnalysisJob;
SVT_HIP_API int32_t svt_hip_analysis_frames(const SvtHipAnalysisJob *jobs, uint32_t n_jobs, int32_t hme_level1_enabled,
                                            int32_t full_precision, void *stream);

/* Per-reference HME / pruning state kept by the reference in MeContext::search_results. */
typedef struct SvtHipMeSearchResult {
    uint64_t hme_sad;
    int16_t  hme_sc_x, hme_sc_y;
    uint8_t  do_ref;
    uint8_t  pad_[3];
} SvtHipMeSearchResult;

/* Output arrays for one picture; every pointer is a device pointer sized for n_b64 blocks in
 * raster order (b64_index = x_b64 + y_b64 * pic_width_in_b64, me_process.c:177).
 *   best_sad / best_mv : MeContext::p_sb_best_sad / p_sb_best_mv [list][ref][85] in the search
 *                        kernels' order (0 = 64x64, 1-4 32x32, 5-20 16x16 z-order, 21-84 8x8
 *                        z-order).  Zero for references that were not searched.
 *   me_mv_array        : MeSbResults::me_mv_array      [stored_pus * max_refs]  (uint32 as_int)
 *   me_candidate_array : MeSbResults::me_candidate_array [stored_pus * max_cand] (1 byte each:
 *                        direction | ref_idx_l0<<2 | ref_idx_l1<<4 | ref0_list<<6 | ref1_list<<7,
 *                        the bit-field layout of MeCandidate, me_sb_results.h:28-34)
 *   total_me_candidate_index : [stored_pus]
 *   Entries the reference does not write are left untouched (caller pre-fills). */
typedef struct SvtHipMeFrameOut {
    uint32_t *best_sad;                 /* [n_b64][2][4][85] */
    uint32_t *best_mv;                  /* [n_b64][2][4][85] */
    SvtHipMeSearchResult *search_results; /* [n_b64][2][4] */
    uint32_t *me_mv_array;              /* [n_b64][stored_pus*max_refs] */
    uint8_t  *me_candidate_array;       /* [n_b64][stored_pus*max_cand] */
    uint8_t  *total_me_candidate_index; /* [n_b64][stored_pus] */
    uint32_t *me_64x64_distortion, *me_32x32_distortion, *me_16x16_distortion, *me_8x8_distortion;
    uint32_t *me_8x8_cost_variance;     /* [n_b64] each (pcs->me_*_distortion, motion_estimation.c:3065-3076) */
    uint32_t *rc_me_distortion;         /* [n_b64] */
} SvtHipMeFrameOut;

typedef struct SvtHipMeFrameJob {
    SvtHipMeParams   prm;
    SvtHipPyramid8   src;                                         /* picture being analysed */
    SvtHipPyramid8   ref[SVT_HIP_ME_MAX_LIST][SVT_HIP_ME_MAX_REF]; /* its references */
    SvtHipMeFrameOut out;
} SvtHipMeFrameJob;

/* Bytes of each output array for a picture of width x height luma samples. */
SVT_HIP_API uint32_t svt_hip_me_b64_count(uint32_t width, uint32_t height);

/* Open-loop ME for `n_jobs` pictures in one launch chain.  `jobs` is a HOST array (it is copied
 * to the device on `stream`); all planes/outputs inside are device pointers.  Asynchronous. */
SVT_HIP_API int32_t svt_hip_me_frames(const SvtHipMeFrameJob *jobs, uint32_t n_jobs, void *stream);

/* Same, with the job array already resident on the device (`d_jobs`): exactly one kernel launch, nothing
 * else is enqueued — the form to capture in a hipGraph or to bracket with events.  `max_b64` = largest
 * svt_hip_me_b64_count() over the jobs.  The jobs must have been validated once through
 * svt_hip_me_validate_jobs (host copy). */
SVT_HIP_API int32_t svt_hip_me_validate_jobs(const SvtHipMeFrameJob *jobs, uint32_t n_jobs, uint32_t *max_b64);
SVT_HIP_API int32_t svt_hip_me_frames_dev(const SvtHipMeFrameJob *d_jobs, uint32_t n_jobs, uint32_t max_b64,
                                          void *stream);

/* Installs the Tier A functions into a table of the reference's RTCD pointers.  `table` holds the
 * ADDRESSES of the encoder's pointers (e.g. &svt_sad_loop_kernel) in the order of
 * SvtHipRtcdSlot; NULL entries are skipped.  See INTEGRATION.md. */
typedef enum SvtHipRtcdSlot {
    SVT_HIP_SLOT_SAD_LOOP_KERNEL = 0,
    SVT_HIP_SLOT_NXM_SAD_KERNEL,
    SVT_HIP_SLOT_EXT_ALL_SAD_8X8_16X16,
    SVT_HIP_SLOT_EXT_EIGHT_SAD_32X32_64X64,
    SVT_HIP_SLOT_EXT_SAD_8X8_16X16,
    SVT_HIP_SLOT_EXT_SAD_32X32_64X64,
    SVT_HIP_SLOT_DOWNSAMPLE_2D,
    SVT_HIP_SLOT_COMPUTE_INTERM_VAR_FOUR8X8,
    SVT_HIP_SLOT_COMPUTE_SUB_MEAN_8X8,
    SVT_HIP_SLOT_COMPUTE_MEAN_8X8,
    SVT_HIP_SLOT_COMPUTE_MEAN_SQUARE_VALUES_8X8,
    SVT_HIP_SLOT_ME_COUNT
} SvtHipRtcdSlot;
SVT_HIP_API int32_t svt_hip_install_rtcd_me(void **table, uint32_t n_slots);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_ME_H */

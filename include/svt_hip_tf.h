/*
 * svt_hip_tf.h — C-ABI for the temporal filter's accumulate / normalise stage.
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   Source/Lib/Codec/aom_dsp_rtcd.h:795-835        svt_av1_apply_[zz_based_]temporal_filter_planewise_medium(_hbd),
 *                                                  apply_filtering_central(_highbd), get_final_filtered_pixels
 *   Source/Lib/Codec/temporal_filtering.c:349-420, 789-997, 999-1330, 2578-2650   their C implementations
 *   caller: tf_16x16 / tf_32x32 loop of produce_temporally_filtered_pic (temporal_filtering.c:3075-3460)
 *
 * The reference's leaves take `struct MeContext *` and read ten of its fields; the control structure does not cross
 * this boundary, so there is no ABI-identical (Tier A) form — the fields travel in SvtHipTfBlock (the field map is in
 * INTEGRATION.md, and oracle/ref_harness.c::ref_tf_block_accumulate fills a real MeContext from the same struct to pin
 * the oracle).  The motion search of the temporal filter is the SAD path of svt_hip_me.h (svt_hip_sad_loop_batch), its
 * sub-pel interpolation the convolve path of svt_hip_inter.h.
 */
#ifndef SVT_HIP_TF_H
#define SVT_HIP_TF_H

#include "svt_hip.h"
#include "svt_hip_me.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One 32x32 luma block (tf_block_col / tf_block_row of a 64x64 block) of the picture being filtered against the
 * motion-compensated prediction from ONE reference picture.  Plane order Y, U, V.  All pointers device memory, at the
 * block's origin; uint8 samples, or uint16 when is_16bit. */
typedef struct SvtHipTfBlock {
    const void *src[3];         /* picture being filtered */
    const void *pred[3];        /* prediction; accum / count use the SAME row pitch as pred (temporal_filtering.c:1104-1110) */
    uint32_t   *accum[3];
    uint16_t   *count[3];
    uint32_t    src_stride[3], pred_stride[3]; /* in samples */
    uint32_t    decay_factor_fp16[3];          /* me_ctx->tf_decay_factor_fp16 */
    uint64_t    block_error[4];                /* tf_16x16_block_error[idx*4 + i] when split, else [0] = tf_32x32_block_error[idx] */
    int16_t     mv_x[4], mv_y[4];              /* tf_16x16_mv_x/y[idx*4 + i] when split, else [0] = tf_32x32_mv_x/y[idx] */
    uint16_t    mv_dist_th;                    /* me_ctx->tf_mv_dist_th */
    uint8_t     split;                         /* tf_32x32_block_split_flag[idx] */
    uint8_t     chroma;                        /* me_ctx->tf_chroma */
    uint8_t     ss_x, ss_y;
    uint8_t     is_16bit, bit_depth;           /* 8 / 10 / 12: svt_av1_apply_temporal_filter_planewise_medium_hbd's encoder_bit_depth */
    uint8_t     zz_based;                      /* 1: svt_av1_apply_zz_based_temporal_filter_planewise_medium[_hbd] (the weight comes from
                                                * the block error alone; src, mv_*, mv_dist_th are not read) */
    uint8_t     pad_[7];
} SvtHipTfBlock;

/* accum += w * pred, count += w with the per-quadrant weights of the planewise "medium" filter
 * (svt_av1_apply_temporal_filter_planewise_medium[_hbd]).  Blocks of one call must not share accum / count samples:
 * launch once per reference picture. */
SVT_HIP_API int32_t svt_hip_tf_accumulate_batch(const SvtHipTfBlock *d_blocks, uint32_t n_blocks, void *stream);

/* The centre picture's own contribution (apply_filtering_central[_highbd], temporal_filtering.c:349-420):
 * accum = 1000 * src, count = 1000 over the same 32x32 (+ chroma) areas; uses src / accum / count / strides / chroma / ss. */
SVT_HIP_API int32_t svt_hip_tf_central_batch(const SvtHipTfBlock *d_blocks, uint32_t n_blocks, void *stream);

/* get_final_filtered_pixels (temporal_filtering.c:2578-2650): dst = (accum + count / 2) / count over the same areas;
 * `pred` of each block is ignored, the result is written to dst[plane] (stride dst_stride[plane]) at the block origin. */
typedef struct SvtHipTfOut {
    void    *dst[3];
    uint32_t dst_stride[3];
    uint32_t pad_;
} SvtHipTfOut;
SVT_HIP_API int32_t svt_hip_tf_normalise_batch(const SvtHipTfBlock *d_blocks, const SvtHipTfOut *d_out, uint32_t n_blocks, void *stream);

/* The three steps above for a whole window in ONE launch: central, accumulate over the n_refs reference pictures (d_ref_blocks: a HOST
 * array of n_refs device arrays of n_blocks records, block i of every array describing the same 32x32 block; only pred / pred_stride /
 * decay_factor_fp16 / block_error / mv_* / mv_dist_th / split / zz_based are read from them), normalise into d_out.  The source, the
 * chroma flag / bit depth come from d_static_blocks[i]; ss_x / ss_y (0 or 1) are the chroma sub-sampling of EVERY block of the call (the
 * records' own ss_x / ss_y are not read).  accum[] / count[] are neither read nor written: the accumulators
 * stay in registers (temporal_filtering.c:3253-3301 per block: apply_filtering_central, the filter per reference, the normalisation).
 * d_out may alias the source planes (the temporal filter works in place): a block is read before it is written, by the same workgroup. */
SVT_HIP_API int32_t svt_hip_tf_filter_blocks(const SvtHipTfBlock *const *d_ref_blocks, uint32_t n_refs, const SvtHipTfBlock *d_static_blocks,
                                             const SvtHipTfOut *d_out, uint32_t n_blocks, uint32_t ss_x, uint32_t ss_y, void *stream);

/* ---- whole-picture driver: produce_temporally_filtered_pic (temporal_filtering.c:2752-3308) -------------------------------
 * The block loop of the reference, for ONE centre picture against its window of reference pictures, device-resident:
 *   per reference picture  svt_aom_motion_estimation_b64 in ME_MCTF mode (the b64 kernel of svt_hip_me.h)
 *                          -> tf_64x64 / tf_32x32 / tf_16x16_sub_pel_search, tf_use_64x64_pred, the 64x64-vs-32x32 and the
 *                             32x32-vs-16x16 decisions (derive_tf_32x32_block_split_flag)        (:1531-2105, 236-285, 2646, 3085-3245)
 *                          -> tf_64x64 / tf_32x32_inter_prediction (sharp 8-tap, luma + chroma)     (:2226-2576)
 *                          -> convert_64x64_info_to_32x32_info                                       (:2661-2728)
 *                          -> apply_filtering_block_plane_wise                                       (:1382-1524)
 *   apply_filtering_central before, get_final_filtered_pixels after; the centre picture is filtered IN PLACE.
 * What stays with the caller (scalar control logic): which pictures enter the window (the ahd-error / brightness outlier
 * tests, ref_frame_factor), tf_decay_factor_fp16 (noise levels, qp), tf_chroma, tf_mv_dist_th; packing 10-bit pictures into
 * 16-bit planes before and unpacking / re-decimating the filtered centre picture after (svt_hip_pyramid_frame).
 * 8x8 prediction (TfControls::enable_8x8_pred, tf level 1 = presets <= M2): tf_8x8_sub_pel_search (:2106-2224) behind the 16x16
 * searches, the 16x16 -> 8x8 decisions of derive_tf_32x32_block_split_flag (:236-285), 8x8 luma / 4x4 chroma predictions (:2384-2445;
 * the chroma blocks with the 4-tap kernels of narrow blocks); the ME parameters must have enable_me_8x8 set. */
#define SVT_HIP_TF_MAX_REFS 32 /* ALTREF_MAX_NFRAMES - 1 */

typedef struct SvtHipTfCtrls {      /* the TfControls fields (definitions.h:120-215) the block loop reads */
    uint8_t  half_pel_mode, quarter_pel_mode, eight_pel_mode; /* 0 off, 1 all eight neighbours, >= 2 horizontal / vertical only */
    uint8_t  use_2tap;               /* bilinear instead of regular 8-tap for the 64x64 and 32x32 searches */
    uint8_t  sub_sampling_shift;     /* the centre position of a search is measured on every 2nd row */
    uint8_t  use_pred_64x64_only_th; /* 0 off, 255 always 64x64, else tf_use_64x64_pred's deviation threshold */
    uint8_t  subpel_early_exit_th;
    uint8_t  use_8bit_subpel;        /* bit depth > 8: the sub-pel searches run on the 8-bit planes */
    uint8_t  use_zz_based_filter, enable_8x8_pred;
    uint8_t  low_delay;              /* 1: produce_temporally_filtered_pic_ld (temporal_filtering.c:3310-3660, pred_structure LOW_DELAY_B): no
                                      * motion search — every 64x64 block is predicted from the co-located block (vector 0) and filtered as
                                      * four un-split 32x32 blocks whose error is their variance against the source; the ME parameters, the
                                      * sub-pel controls and tot_blks are not used */
    uint8_t  pad_[5];
    uint64_t pred_error_32x32_th;
} SvtHipTfCtrls;

/* One picture of the window.  Every plane follows the EbPictureBufferDesc convention of SvtHipPlane8: sample (x, y) =
 * buf[(org_y + y) * stride + org_x + x]; chroma planes have half the luma origin (4:2:0). */
typedef struct SvtHipTfPic {
    SvtHipPyramid8 pyr;        /* 8-bit luma: padded input picture, 1/4 and 1/16 versions (EbPaReferenceObject) */
    uint8_t       *chroma8[2]; /* 8-bit Cb / Cr; origin (pyr.full.org_x / 2, pyr.full.org_y / 2) */
    uint32_t       chroma8_stride, pad_;
    uint16_t      *hbd[3];     /* bit depth > 8: 16-bit Y / Cb / Cr (altref_buffer_highbd) with the geometry and strides of the
                                * 8-bit planes; NULL otherwise */
    uint64_t       picture_number; /* replaces me.picture_number (centre) / me.ref_picture_number[0][0] (reference picture) */
} SvtHipTfPic;

typedef struct SvtHipTfPictureJob {
    SvtHipMeParams me;         /* as for svt_hip_me_frames (hme_l0_sa = MeContext::hme_l0_sa_default_tf); me_mctf, the single list /
                                * reference and tf_me_exit_th are taken from this struct as given */
    SvtHipTfCtrls  ctrls;
    uint32_t       decay_factor_fp16[3]; /* MeContext::tf_decay_factor_fp16 */
    uint16_t       mv_dist_th;           /* MeContext::tf_mv_dist_th */
    uint8_t        chroma;               /* MeContext::tf_chroma */
    uint8_t        bit_depth;            /* 8 or 10 */
    uint32_t       mi_rows, mi_cols;     /* Av1Common (4x4 units): the motion-vector clamp of the predictions */
    uint32_t       n_refs;               /* pictures filtered against, at most SVT_HIP_TF_MAX_REFS */
    SvtHipTfPic    centre;
    SvtHipTfPic    ref[SVT_HIP_TF_MAX_REFS];
    void          *workspace;            /* device, svt_hip_tf_workspace_bytes() */
    uint64_t       workspace_bytes;
    uint32_t      *tot_blks;             /* device [2] or NULL: += MeContext::tf_tot_horz_blks, tf_tot_vert_blks */
} SvtHipTfPictureJob;

SVT_HIP_API uint64_t svt_hip_tf_workspace_bytes(uint32_t width, uint32_t height, uint32_t n_refs);
/* byte offset inside the workspace of SvtHipTfB64State[n_b64] of reference picture `ref` (valid after the call has run) */
SVT_HIP_API uint64_t svt_hip_tf_workspace_state_offset(uint32_t width, uint32_t height, uint32_t n_refs, uint32_t ref);
/* `job` is a HOST struct, every pointer inside is device memory.  Asynchronous on `stream`. */
SVT_HIP_API int32_t svt_hip_tf_filter_picture(const SvtHipTfPictureJob *job, void *stream);

/* The per-64x64-block state of the motion refinement, as the reference keeps it in MeContext (tf_64x64_*, tf_32x32_*,
 * tf_16x16_*): one record per (reference picture, b64) in the workspace; exposed for tests and for callers that run the
 * stages themselves.  16x16 entries are in the order of tf_16x16_mv_x (idx_32x32 * 4 + idx_16x16). */
typedef struct SvtHipTfB64State {
    uint64_t err64, err32[4], err16[16];
    int16_t  mv64_x, mv64_y, mv32_x[4], mv32_y[4], mv16_x[16], mv16_y[16];
    uint8_t  split32[4];       /* tf_32x32_block_split_flag */
    uint8_t  use_64x64;        /* the block is predicted as one 64x64 (convert_64x64_info_to_32x32_info applies) */
    uint8_t  pad_[3];
    /* enable_8x8_pred (tf level 1): tf_8x8_block_error / tf_8x8_mv_x / _y in the reference's order (idx_32x32 * 16 + idx_16x16 * 4 +
     * idx_8x8) and tf_16x16_block_split_flag[idx_32x32][idx_16x16]; err16 of a split 16x16 block is the sum of its four 8x8 errors */
    uint64_t err8[64];
    int16_t  mv8_x[64], mv8_y[64];
    uint8_t  split16[16];
} SvtHipTfB64State;

/* ---- noise estimate (svt_estimate_noise_fp16 / svt_estimate_noise_highbd_fp16, aom_dsp_rtcd.h:874-877;
 * temporal_filtering.c:3668-3736): over the interior of one plane, the mean absolute Laplacian of the samples whose
 * Sobel gradient magnitude is below EDGE_THRESHOLD, scaled by sqrt(pi/2)/6, in 16.16 fixed point; -65536 when fewer
 * than SMOOTH_THRESHOLD samples qualify.  Callers: pd_process.c:2990-3041 (per plane, before the temporal filter),
 * md_config_process.c:499. */
typedef struct SvtHipTfNoise {
    uint64_t sum;        /* sum of |Laplacian| (rounded down to 8-bit scale for deeper samples) over the smooth samples */
    uint64_t num;        /* number of smooth samples */
    int32_t  noise_fp16; /* the reference's return value */
    int32_t  pad_;
} SvtHipTfNoise;
/* Tier B: device plane in, device record out (zeroed by the call); stride in samples. */
SVT_HIP_API int32_t svt_hip_tf_estimate_noise(const void *d_src, uint32_t width, uint32_t height, uint32_t stride,
                                              int32_t is_16bit, int32_t bit_depth, SvtHipTfNoise *d_out, void *stream);
/* Tier A: the RTCD signatures (host pointers). */
SVT_HIP_API int32_t svt_estimate_noise_fp16_hip(const uint8_t *src, uint16_t width, uint16_t height, uint16_t stride_y);
SVT_HIP_API int32_t svt_estimate_noise_highbd_fp16_hip(const uint16_t *src, int width, int height, int stride, int bd);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_TF_H */

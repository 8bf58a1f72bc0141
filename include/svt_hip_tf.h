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

/*
 * svt_hip_inter.h — C-ABI for inter-prediction interpolation: the single-reference and the
 * compound ("jnt") families.
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   Source/Lib/Codec/common_dsp_rtcd.h:185-221   svt_av1_convolve_{2d_sr,x_sr,y_sr,2d_copy_sr},
 *                                                svt_av1_jnt_convolve_{2d,x,y,2d_copy} and the highbd sets
 *   Source/Lib/Codec/inter_prediction.c:311-668, 670-1035   their C implementations
 *   callers: svt_aom_inter_predictor / highbd_inter_predictor via svt_aom_convolve[subpel_x != 0][subpel_y != 0][is_compound]
 *            (inter_prediction.c:1036-1062)
 * Scaled references, OBMC and masked compounds are not covered.
 */
#ifndef SVT_HIP_INTER_H
#define SVT_HIP_INTER_H

#include "svt_hip_lf.h" /* SvtHipConvolveParams */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SvtHipInterpFilterParams { /* InterpFilterParams, definitions.h:750-755 */
    const int16_t *filter_ptr;           /* [subpel_shifts][taps] */
    uint16_t       taps, subpel_shifts;
    int32_t        interp_filter;
} SvtHipInterpFilterParams;

/* Tier A: RTCD signatures, host pointers (the highbd functions take real uint16 pointers, as in the reference). */
#define SVT_HIP_DECL_CONV(mode)                                                                                                  \
    SVT_HIP_API void svt_av1_convolve_##mode##_hip(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride,      \
                                                   int32_t w, int32_t h, SvtHipInterpFilterParams *filter_params_x,              \
                                                   SvtHipInterpFilterParams *filter_params_y, const int32_t subpel_x_q4,         \
                                                   const int32_t subpel_y_q4, SvtHipConvolveParams *conv_params);               \
    SVT_HIP_API void svt_av1_highbd_convolve_##mode##_hip(const uint16_t *src, int32_t src_stride, uint16_t *dst,               \
                                                          int32_t dst_stride, int32_t w, int32_t h,                             \
                                                          const SvtHipInterpFilterParams *filter_params_x,                      \
                                                          const SvtHipInterpFilterParams *filter_params_y,                      \
                                                          const int32_t subpel_x_q4, const int32_t subpel_y_q4,                 \
                                                          SvtHipConvolveParams *conv_params, int32_t bd);
SVT_HIP_DECL_CONV(2d_sr)
SVT_HIP_DECL_CONV(x_sr)
SVT_HIP_DECL_CONV(y_sr)
SVT_HIP_DECL_CONV(2d_copy_sr)
#undef SVT_HIP_DECL_CONV
/* compound: conv_params->dst / dst_stride is the ConvBufType (uint16) buffer; do_average == 0 stores the offset
 * intermediate there, do_average == 1 reads it, averages (use_jnt_comp_avg: fwd_offset / bck_offset weights) and writes
 * pixels to dst */
#define SVT_HIP_DECL_JNT(mode)                                                                                                   \
    SVT_HIP_API void svt_av1_jnt_convolve_##mode##_hip(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride,  \
                                                       int32_t w, int32_t h, SvtHipInterpFilterParams *filter_params_x,          \
                                                       SvtHipInterpFilterParams *filter_params_y, const int32_t subpel_x_q4,     \
                                                       const int32_t subpel_y_q4, SvtHipConvolveParams *conv_params);           \
    SVT_HIP_API void svt_av1_highbd_jnt_convolve_##mode##_hip(const uint16_t *src, int32_t src_stride, uint16_t *dst,           \
                                                              int32_t dst_stride, int32_t w, int32_t h,                         \
                                                              const SvtHipInterpFilterParams *filter_params_x,                  \
                                                              const SvtHipInterpFilterParams *filter_params_y,                  \
                                                              const int32_t subpel_x_q4, const int32_t subpel_y_q4,             \
                                                              SvtHipConvolveParams *conv_params, int32_t bd);
SVT_HIP_DECL_JNT(2d)
SVT_HIP_DECL_JNT(x)
SVT_HIP_DECL_JNT(y)
SVT_HIP_DECL_JNT(2d_copy)
#undef SVT_HIP_DECL_JNT

/* Tier B: one descriptor per predicted block, all pointers device memory.  taps_x / taps_y == 0 selects what the
 * reference dispatches to when that direction has no sub-pel offset (x_sr, y_sr, 2d_copy_sr). */
typedef struct SvtHipConvolveDesc {
    const void *src;        /* sample the block's (0,0) maps to in the reference plane; taps/2-1 samples are read to the
                             * left / above and taps/2 to the right / below */
    void       *dst;
    uint32_t    src_stride, dst_stride; /* in samples */
    uint16_t    w, h;                   /* 2 .. 128; a descriptor with w == 0 or h == 0 is skipped */
    int16_t     filter_x[8], filter_y[8]; /* the kernels of this block's sub-pel phases (av1_get_interp_filter_subpel_kernel) */
    uint8_t     taps_x, taps_y;         /* 0, or an even number <= 8 */
    uint8_t     round_0, round_1;       /* ConvolveParams of get_conv_params (convolve.h:40-68) */
    uint8_t     bit_depth, is_16bit;
    uint8_t     compound;               /* 0 single reference; 1 first prediction of a compound: the offset intermediate goes to
                                         * cbuf; 2 second prediction, plain average with cbuf -> dst; 3 second prediction,
                                         * distance-weighted average (use_jnt_comp_avg) */
    uint8_t     fwd_offset, bck_offset; /* compound 3: weights of cbuf and of this prediction (sum 16, DIST_PRECISION_BITS 4) */
    uint8_t     pad_[3];
    uint16_t   *cbuf;                   /* ConvBufType [h][cbuf_stride] (compound != 0); a compound 2 / 3 descriptor must be
                                         * launched AFTER the call that ran the compound 1 descriptor filling its cbuf */
    uint32_t    cbuf_stride, pad2_;
} SvtHipConvolveDesc;
SVT_HIP_API int32_t svt_hip_convolve_batch(const SvtHipConvolveDesc *d_desc, uint32_t n, void *stream);
/* the same entry point under its first name */
SVT_HIP_API int32_t svt_hip_convolve_sr_batch(const SvtHipConvolveDesc *d_desc, uint32_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_INTER_H */

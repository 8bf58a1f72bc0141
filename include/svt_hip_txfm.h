/*
 * svt_hip_txfm.h — C-ABI for forward / inverse 2-D transforms and quantisation (SURVEY.md §8 rows a6–a8).
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   Source/Lib/Codec/aom_dsp_rtcd.h:85-205     svt_av1_fwd_txfm2d_{WxH}[_N2|_N4]           (57 pointers)
 *   Source/Lib/Codec/aom_dsp_rtcd.h:214-237    svt_handle_transform{16x64,32x64,64x16,64x32,64x64}[_N2_N4]
 *   Source/Lib/Codec/aom_dsp_rtcd.h:244-260    svt_aom_quantize_b, svt_aom_highbd_quantize_b, svt_av1_quantize_b_qm,
 *                                              svt_av1_highbd_quantize_b_qm, svt_av1_quantize_fp[_32x32|_64x64|_qm],
 *                                              svt_av1_highbd_quantize_fp[_qm]
 *   Source/Lib/Codec/common_dsp_rtcd.c:482-500 svt_av1_inv_txfm2d_add_{WxH}                 (19 pointers)
 *   Source/Lib/Codec/full_loop.c:1462-1686     svt_aom_quantize_inv_quantize (the per-TB driver: Tier B batch)
 *   Source/Lib/Codec/transforms.c:3100-3154    svt_aom_estimate_transform     (the per-TB driver: Tier B batch)
 *
 * TxType is the reference's enum (definitions.h:981-998, 0 = DCT_DCT ... 15 = H_FLIPADST); TranLow is int32_t;
 * QmVal is uint8_t.
 */
#ifndef SVT_HIP_TXFM_H
#define SVT_HIP_TXFM_H

#include "svt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Tier A — ABI-identical per-call entry points (host pointers).
 * ------------------------------------------------------------------------------------------- */
#define SVT_HIP_FWD_DECL(W, H)                                                                                     \
    SVT_HIP_API void svt_av1_fwd_txfm2d_##W##x##H##_hip(int16_t *input, int32_t *output, uint32_t input_stride,    \
                                                        int32_t transform_type, uint8_t bit_depth);               \
    SVT_HIP_API void svt_av1_fwd_txfm2d_##W##x##H##_N2_hip(int16_t *input, int32_t *output, uint32_t input_stride, \
                                                           int32_t transform_type, uint8_t bit_depth);            \
    SVT_HIP_API void svt_av1_fwd_txfm2d_##W##x##H##_N4_hip(int16_t *input, int32_t *output, uint32_t input_stride, \
                                                           int32_t transform_type, uint8_t bit_depth);
SVT_HIP_FWD_DECL(4, 4) SVT_HIP_FWD_DECL(8, 8) SVT_HIP_FWD_DECL(16, 16) SVT_HIP_FWD_DECL(32, 32) SVT_HIP_FWD_DECL(64, 64)
SVT_HIP_FWD_DECL(4, 8) SVT_HIP_FWD_DECL(8, 4) SVT_HIP_FWD_DECL(8, 16) SVT_HIP_FWD_DECL(16, 8) SVT_HIP_FWD_DECL(16, 32)
SVT_HIP_FWD_DECL(32, 16) SVT_HIP_FWD_DECL(32, 64) SVT_HIP_FWD_DECL(64, 32) SVT_HIP_FWD_DECL(4, 16) SVT_HIP_FWD_DECL(16, 4)
SVT_HIP_FWD_DECL(8, 32) SVT_HIP_FWD_DECL(32, 8) SVT_HIP_FWD_DECL(16, 64) SVT_HIP_FWD_DECL(64, 16)

/* svt_handle_transformWxH / _N2_N4: energy of the discarded 64-point area + repack to 32-wide */
SVT_HIP_API uint64_t svt_handle_transform16x64_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform32x64_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform64x16_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform64x32_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform64x64_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform16x64_N2_N4_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform32x64_N2_N4_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform64x16_N2_N4_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform64x32_N2_N4_hip(int32_t *output);
SVT_HIP_API uint64_t svt_handle_transform64x64_N2_N4_hip(int32_t *output);

/* inverse + add; the three signature flavours of common_dsp_rtcd.h (square / 4xN / the rest) */
#define SVT_HIP_INV_DECL_SQ(W, H)                                                                                    \
    SVT_HIP_API void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, \
                                                            uint16_t *output_w, int32_t stride_w, int32_t tx_type,  \
                                                            int32_t bd);
#define SVT_HIP_INV_DECL_TS(W, H)                                                                                    \
    SVT_HIP_API void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, \
                                                            uint16_t *output_w, int32_t stride_w, int32_t tx_type,  \
                                                            int32_t tx_size, int32_t bd);
#define SVT_HIP_INV_DECL_EOB(W, H)                                                                                   \
    SVT_HIP_API void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, \
                                                            uint16_t *output_w, int32_t stride_w, int32_t tx_type,  \
                                                            int32_t tx_size, int32_t eob, int32_t bd);
SVT_HIP_INV_DECL_SQ(4, 4) SVT_HIP_INV_DECL_SQ(8, 8) SVT_HIP_INV_DECL_SQ(16, 16) SVT_HIP_INV_DECL_SQ(32, 32)
SVT_HIP_INV_DECL_SQ(64, 64) SVT_HIP_INV_DECL_TS(4, 8) SVT_HIP_INV_DECL_TS(8, 4) SVT_HIP_INV_DECL_TS(4, 16)
SVT_HIP_INV_DECL_TS(16, 4) SVT_HIP_INV_DECL_EOB(8, 16) SVT_HIP_INV_DECL_EOB(16, 8) SVT_HIP_INV_DECL_EOB(16, 32)
SVT_HIP_INV_DECL_EOB(32, 16) SVT_HIP_INV_DECL_EOB(32, 64) SVT_HIP_INV_DECL_EOB(64, 32) SVT_HIP_INV_DECL_EOB(8, 32)
SVT_HIP_INV_DECL_EOB(32, 8) SVT_HIP_INV_DECL_EOB(16, 64) SVT_HIP_INV_DECL_EOB(64, 16)

/* quantizers */
#define SVT_HIP_QARGS                                                                                           \
    const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr,            \
        const int16_t *quant_ptr, const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr,   \
        const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, const int16_t *iscan
SVT_HIP_API void svt_aom_quantize_b_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int32_t log_scale);
SVT_HIP_API void svt_av1_quantize_b_qm_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int32_t log_scale);
SVT_HIP_API void svt_aom_highbd_quantize_b_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int32_t log_scale);
SVT_HIP_API void svt_av1_highbd_quantize_b_qm_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int32_t log_scale);
SVT_HIP_API void svt_av1_quantize_fp_hip(SVT_HIP_QARGS);
SVT_HIP_API void svt_av1_quantize_fp_32x32_hip(SVT_HIP_QARGS);
SVT_HIP_API void svt_av1_quantize_fp_64x64_hip(SVT_HIP_QARGS);
SVT_HIP_API void svt_av1_quantize_fp_qm_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale);
SVT_HIP_API void svt_av1_highbd_quantize_fp_hip(SVT_HIP_QARGS, int16_t log_scale);
SVT_HIP_API void svt_av1_highbd_quantize_fp_qm_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale);

/* Residual producer and transform-domain cost of the TPL dispenser / mode decision: svt_aom_subtract_block,
 * svt_aom_highbd_subtract_block (common_dsp_rtcd.h:234-237; src8 / pred8 of the highbd form are uint16 planes) and
 * svt_aom_satd (aom_dsp_rtcd.h:206-207). */
/* Mirror of TxfmParam (definitions.h:1051-1063): TxType, TxSize and TxSetType are one-byte (packed) enums there. */
typedef struct SvtHipTxfmParam {
    uint8_t tx_type;
    uint8_t tx_size;
    int32_t lossless;
    int32_t bd;
    int32_t is_hbd;
    uint8_t tx_set_type;
    int32_t eob;
} SvtHipTxfmParam;
/* svt_av1_inv_txfm_add (common_dsp_rtcd.h:150; inv_transforms.c:3177-3193): 8-bit prediction + inverse transform ->
 * 8-bit reconstruction, size and type taken from txfm_param.  bd must be 8 and lossless 0 (all the reference passes). */
SVT_HIP_API void svt_av1_inv_txfm_add_hip(const int32_t *dqcoeff, uint8_t *dst_r, int32_t stride_r, uint8_t *dst_w,
                                          int32_t stride_w, const SvtHipTxfmParam *txfm_param);
/* svt_residual_kernel8bit / 16bit (common_dsp_rtcd.h:163,174), svt_spatial_full_distortion_kernel (:171),
 * svt_full_distortion_kernel16_bits (:173: byte pointers that hold 16-bit samples; offsets and strides in samples) */
SVT_HIP_API void svt_residual_kernel8bit_hip(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride,
                                             int16_t *residual, uint32_t residual_stride, uint32_t area_width,
                                             uint32_t area_height);
SVT_HIP_API void svt_residual_kernel16bit_hip(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride,
                                              int16_t *residual, uint32_t residual_stride, uint32_t area_width,
                                              uint32_t area_height);
SVT_HIP_API uint64_t svt_spatial_full_distortion_kernel_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride,
                                                            uint8_t *recon, int32_t recon_offset, uint32_t recon_stride,
                                                            uint32_t area_width, uint32_t area_height);
SVT_HIP_API uint64_t svt_full_distortion_kernel16_bits_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride,
                                                           uint8_t *pred, int32_t pred_offset, uint32_t pred_stride,
                                                           uint32_t area_width, uint32_t area_height);
/* Tier B form of the two distortion leaves: the sum of squared differences of two DEVICE planes (strides in samples) into *d_out (device,
 * zeroed by the call).  Caller: picture_sse_calculations (deblocking_filter.c:716-834) after every trial of the deblocking level search
 * (try_filter_frame :842-882) — the filtered trial picture then never leaves the device, 8 bytes come back. */
SVT_HIP_API int32_t svt_hip_plane_sse(const void *d_a, uint32_t a_stride, const void *d_b, uint32_t b_stride, uint32_t width, uint32_t height, int32_t is_16bit,
                                      uint64_t *d_out, void *stream);
SVT_HIP_API void svt_aom_subtract_block_hip(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr,
                                            ptrdiff_t src_stride, const uint8_t *pred_ptr, ptrdiff_t pred_stride);
SVT_HIP_API void svt_aom_highbd_subtract_block_hip(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride,
                                                   const uint8_t *src_ptr, ptrdiff_t src_stride, const uint8_t *pred_ptr,
                                                   ptrdiff_t pred_stride, int bd);
SVT_HIP_API int svt_aom_satd_hip(const int32_t *coeff, int length);
/* Transform-domain distortion of the full loop: svt_full_distortion_kernel32_bits and its cbf-zero form
 * (common_dsp_rtcd.h:166-167, pic_operators.c:150-221); distortion_result = {residual, prediction}. */
SVT_HIP_API void svt_full_distortion_kernel32_bits_hip(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff,
                                                       uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                                       uint32_t area_width, uint32_t area_height);
SVT_HIP_API void svt_full_distortion_kernel_cbf_zero32_bits_hip(int32_t *coeff, uint32_t coeff_stride, uint64_t distortion_result[2],
                                                                uint32_t area_width, uint32_t area_height);

/* ---------------------------------------------------------------------------------------------
 * Tier B — batched, fused transform block processing on device-resident data:
 *   residual --fwd txfm--> coeff --[64-pt: energy + repack]--> quantize --> qcoeff/dqcoeff/eob
 *            --[optional]--> inverse txfm + prediction --> reconstruction
 * One call = n blocks of ONE size (w x h); type / shape / quantizer parameters vary per block.
 * All offsets are BYTE offsets into one device arena `d_base`; SVT_HIP_NO_OFFSET disables an output.
 * ------------------------------------------------------------------------------------------- */
#define SVT_HIP_NO_OFFSET (~(uint64_t)0)

enum { /* SvtHipTxfmDesc::quant_mode */
    SVT_HIP_QUANT_NONE = 0,
    SVT_HIP_QUANT_B,        /* svt_aom_quantize_b          (full_loop.c:25-75)   */
    SVT_HIP_QUANT_B_HBD,    /* svt_aom_highbd_quantize_b   (full_loop.c:145-194) */
    SVT_HIP_QUANT_FP,       /* svt_av1_quantize_fp*        (full_loop.c:278-338) */
    SVT_HIP_QUANT_FP_HBD    /* svt_av1_highbd_quantize_fp* (full_loop.c:383-449) */
};
enum { /* SvtHipTxfmDesc::flags */
    SVT_HIP_TX_FWD     = 1, /* run the forward transform from `residual_off` */
    SVT_HIP_TX_INV     = 2, /* run the inverse transform + add (needs dqcoeff: computed here or read from dqcoeff_off) */
    SVT_HIP_TX_PIXEL16 = 4, /* pred / recon are uint16 planes (else uint8, bit_depth must be 8) */
    SVT_HIP_TX_FULLCOEFF = 8, /* coeff_off receives the complete [h][w] array even for 64-point sizes (no repack) */
    /* The residual is formed on the fly, svt_aom_subtract_block / svt_aom_highbd_subtract_block semantics
     * (inter_prediction.c:35-60): residual_off / residual_stride address the SOURCE pixels, pred_off / pred_stride the
     * prediction (uint8, or uint16 with SVT_HIP_TX_PIXEL16); strides in pixels. */
    SVT_HIP_TX_SRC_PRED = 16,
    /* result.satd = svt_aom_satd (common_dsp_rtcd.c:71-78) over the retained coefficients of the forward transform.
     * FWD | SRC_PRED | SATD with DCT_DCT is the TPL dispenser's block cost (src_ops_process.c:734-748, 861-873): pass the
     * sub-sampled transform size as w x h and the strides pre-shifted by subsample_tx exactly as the reference does, and
     * shift the result left by subsample_tx.  (For sizes with a 64-point side the retained 32x32 block is summed once;
     * the reference's length = 64*64 walk also re-reads the stale rows 16..31 of the un-repacked array.) */
    SVT_HIP_TX_SATD = 32
};

typedef struct SvtHipTxfmDesc {
    uint64_t residual_off;            /* int16 [h][residual_stride] */
    uint64_t coeff_off;               /* int32 out: [h][w], or [min(h,32)][min(w,32)] repacked for 64-point sizes */
    uint64_t qcoeff_off, dqcoeff_off; /* int32 [n], n = min(w,32)*min(h,32) */
    uint64_t pred_off, recon_off;     /* pixel planes (SVT_HIP_TX_INV) */
    uint64_t iscan_off;               /* int16 iscan[n] (position of raster index in scan order) */
    uint64_t qm_off, iqm_off;         /* uint8 [n] quantisation matrices or SVT_HIP_NO_OFFSET */
    uint32_t residual_stride;         /* in int16 units */
    uint32_t pred_stride, recon_stride; /* in pixels */
    int16_t  zbin[2], round[2], quant[2], quant_shift[2], dequant[2]; /* [0] DC, [1] AC */
    uint8_t  tx_type, shape /* 0 full, 1 N2, 2 N4 */, bit_depth, quant_mode, log_scale, flags;
    uint8_t  dist_w, dist_h; /* svt_hip_txfm_distortion_batch: cropped_tx_width / cropped_tx_height of the caller
                              * (0 = min(w,32) / min(h,32)) */
} SvtHipTxfmDesc;

typedef struct SvtHipTxfmResult {
    uint64_t three_quad_energy; /* svt_handle_transformWxH return value (0 for sizes without a 64-point side) */
    uint16_t eob;
    uint16_t pad_;
    uint32_t satd; /* SVT_HIP_TX_SATD, else 0 */
} SvtHipTxfmResult;

SVT_HIP_API int32_t svt_hip_txfm_quant_batch(uint8_t *d_base, const SvtHipTxfmDesc *d_desc, SvtHipTxfmResult *d_result,
                                             uint32_t n_blocks, uint32_t w, uint32_t h, void *stream);

/* Transform-domain distortion of the same blocks, what svt_aom_full_loop_core reads right after the quantiser
 * (svt_aom_picture_full_distortion32_bits_single, pic_operators.c:150-234): d_distortion[i] = {DIST_CALC_RESIDUAL =
 * sum (coeff - dqcoeff)^2, DIST_CALC_PREDICTION = sum coeff^2} over the dist_w x dist_h top-left area of the retained
 * coefficient block of descriptor i ({0, 0} when it has no coeff_off / dqcoeff_off).  Run it after
 * svt_hip_txfm_quant_batch on the same stream with the same descriptors (that call writes the coefficient arrays). */
SVT_HIP_API int32_t svt_hip_txfm_distortion_batch(const uint8_t *d_base, const SvtHipTxfmDesc *d_desc, uint64_t (*d_distortion)[2],
                                                  uint32_t n_blocks, uint32_t w, uint32_t h, void *stream);

/* Stand-alone batched quantiser over device coefficient arrays (same descriptor; coeff_off is the INPUT). */
SVT_HIP_API int32_t svt_hip_quantize_batch(uint8_t *d_base, const SvtHipTxfmDesc *d_desc, SvtHipTxfmResult *d_result,
                                           uint32_t n_blocks, uint32_t n_coeffs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_TXFM_H */

/* oracle/src/orc_txfm.h — TEST INFRASTRUCTURE (CPU restatement of the transform / quantize part of the hot path). */
#ifndef ORC_TXFM_H
#define ORC_TXFM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define ORC_API __attribute__((visibility("default")))

ORC_API const int32_t *orc_cospi(int bit);
ORC_API int      orc_txfm_valid(int w, int h, int tx_type);
ORC_API void     orc_fwd_txfm2d(const int16_t *input, int32_t *output, uint32_t stride, int w, int h, int tx_type,
                                int bd, int shape);
ORC_API uint64_t orc_handle_transform64(int32_t *output, int w, int h);
ORC_API void     orc_inv_txfm2d_add(const int32_t *input, const uint16_t *pred, int32_t stride_r, uint16_t *recon,
                                    int32_t stride_w, int w, int h, int tx_type, int bd);
ORC_API void     orc_inv_txfm2d_add_8bit(const int32_t *input, const uint8_t *pred, int32_t stride_r, uint8_t *recon,
                                         int32_t stride_w, int w, int h, int tx_type);
ORC_API void orc_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                            const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant,
                            uint16_t *eob_ptr, const int16_t *scan, const uint8_t *qm, const uint8_t *iqm, int log_scale);
ORC_API void orc_highbd_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                                   const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                                   const int16_t *dequant, uint16_t *eob_ptr, const int16_t *scan, const uint8_t *qm,
                                   const uint8_t *iqm, int log_scale);
ORC_API void orc_quantize_fp(const int32_t *coeff, intptr_t n, const int16_t *round, const int16_t *quant, int32_t *qcoeff,
                             int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_ptr, const int16_t *scan,
                             const uint8_t *qm, const uint8_t *iqm, int log_scale);
ORC_API void orc_highbd_quantize_fp(const int32_t *coeff, intptr_t n, const int16_t *round, const int16_t *quant,
                                    int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_ptr,
                                    const int16_t *scan, const uint8_t *qm, const uint8_t *iqm, int log_scale);
/* orc_residual.c */
ORC_API void orc_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint8_t *src, ptrdiff_t src_stride,
                                const uint8_t *pred, ptrdiff_t pred_stride);
ORC_API void orc_highbd_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint16_t *src,
                                       ptrdiff_t src_stride, const uint16_t *pred, ptrdiff_t pred_stride);
ORC_API int     orc_satd(const int32_t *coeff, int length);
ORC_API void    orc_full_distortion32(const int32_t *coeff, uint32_t coeff_stride, const int32_t *recon, uint32_t recon_stride,
                                      uint64_t out[2], uint32_t w, uint32_t h);
ORC_API int64_t orc_tpl_block_cost(const uint8_t *src, int src_stride, const uint8_t *pred, int pred_stride, int size,
                                   int subsample_tx, int pf_shape);
#ifdef __cplusplus
}
#endif
#endif

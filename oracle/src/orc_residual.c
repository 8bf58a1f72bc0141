/* oracle/src/orc_residual.c — TEST INFRASTRUCTURE: CPU restatement of the residual producer and the transform-domain cost
 * around the forward transform (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 *   orc_subtract_block / orc_highbd_subtract_block  follow svt_aom_subtract_block_c / svt_aom_highbd_subtract_block_c
 *                                                   (Source/Lib/Codec/inter_prediction.c:35-60)
 *   orc_satd                                        follows svt_aom_satd_c (Source/Lib/Codec/common_dsp_rtcd.c:71-78)
 *   orc_tpl_block_cost                              follows the TPL dispenser's transform-domain cost
 *                                                   (Source/Lib/Codec/src_ops_process.c:734-748, 861-873):
 *                                                   subtract -> svt_av1_wht_fwd_txfm (transforms.c:3569-3584: a DCT_DCT
 *                                                   forward transform of the sub-sampled size) -> satd << subsample_tx
 * Pinned against the reference by tests/test_residual_oracle.py (oracle/_ref) and tests/golden/tpl_cost.npz. */
#include <stdlib.h>
#include <string.h>

#include "orc_txfm.h"

ORC_API void orc_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint8_t *src, ptrdiff_t src_stride,
                                const uint8_t *pred, ptrdiff_t pred_stride) {
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) diff[r * diff_stride + c] = (int16_t)(src[r * src_stride + c] - pred[r * pred_stride + c]);
}

ORC_API void orc_highbd_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint16_t *src,
                                       ptrdiff_t src_stride, const uint16_t *pred, ptrdiff_t pred_stride) {
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) diff[r * diff_stride + c] = (int16_t)(src[r * src_stride + c] - pred[r * pred_stride + c]);
}

ORC_API int orc_satd(const int32_t *coeff, int length) {
    unsigned acc = 0; /* the reference accumulates in int; unsigned gives the same bits without overflow UB */
    for (int i = 0; i < length; i++) acc += (unsigned)abs(coeff[i]);
    return (int)acc;
}

/* One size x size TPL block: `size` is 8/16/32, subsample_tx 0..2 (every 2^s-th row is used and the transform is
 * size x (size >> s)), pf_shape 0 full / 1 N2 / 2 N4.  src/pred strides in pixels, NOT pre-shifted. */
ORC_API int64_t orc_tpl_block_cost(const uint8_t *src, int src_stride, const uint8_t *pred, int pred_stride, int size,
                                   int subsample_tx, int pf_shape) {
    int16_t diff[64 * 64];
    int32_t coeff[64 * 64];
    const int rows = size >> subsample_tx;
    memset(diff, 0, sizeof(diff));
    /* diff rows land every (size << s) elements, as in the reference's call */
    orc_subtract_block(rows, size, diff, (ptrdiff_t)size << subsample_tx, src, (ptrdiff_t)src_stride << subsample_tx, pred,
                       (ptrdiff_t)pred_stride << subsample_tx);
    memset(coeff, 0, sizeof(coeff));
    orc_fwd_txfm2d(diff, coeff, (uint32_t)(size << subsample_tx), size, rows, 0 /* DCT_DCT */, 8, pf_shape);
    return (int64_t)orc_satd(coeff, (size * size) >> subsample_tx) << subsample_tx;
}

/* svt_full_distortion_kernel32_bits_c / svt_full_distortion_kernel_cbf_zero32_bits_c (Source/Lib/Codec/pic_operators.c:150-221):
 * out[0] = sum (coeff - recon)^2 (recon == NULL: sum coeff^2), out[1] = sum coeff^2 over a w x h area */
ORC_API void orc_full_distortion32(const int32_t *coeff, uint32_t coeff_stride, const int32_t *recon, uint32_t recon_stride,
                                   uint64_t out[2], uint32_t w, uint32_t h) {
    uint64_t res = 0, prd = 0;
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++) {
            const int64_t v = coeff[(size_t)r * coeff_stride + c], e = v - (recon ? (int64_t)recon[(size_t)r * recon_stride + c] : 0);
            res += (uint64_t)(e * e), prd += (uint64_t)(v * v);
        }
    out[0] = res, out[1] = prd;
}

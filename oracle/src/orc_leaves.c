/* oracle/src/orc_leaves.c — TEST INFRASTRUCTURE: CPU restatement of the remaining per-call leaves of the reference's RTCD
 * tables on the hot path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 *   orc_sad_16b               follows svt_aom_sad_16b_kernel_c          (Source/Lib/C_DEFAULT/compute_sad_c.c:39-56)
 *   orc_initialize_buffer32   follows svt_initialize_buffer_32bits_c    (Source/Lib/Codec/me_sad_calculation.c:14-17)
 *   orc_residual8 / 16        follow  svt_residual_kernel8bit_c / 16bit (Source/Lib/Codec/pic_operators.c:101-143)
 *   orc_spatial_sse8          follows svt_spatial_full_distortion_kernel_c (Source/Lib/C_DEFAULT/picture_operators_c.c:62-78)
 *   orc_spatial_sse16         follows svt_full_distortion_kernel16_bits_c  (Source/Lib/Codec/pic_operators.c:174-196)
 *   orc_pme_sad_loop          follows svt_pme_sad_loop_kernel_c         (Source/Lib/Codec/product_coding_loop.c:1781-1828)
 *                             with svt_mv_err_cost (mcomp.c:44-68) and svt_mv_cost (mcomp.h:136-139)
 *   orc_search_one_dual       follows svt_search_one_dual_c             (Source/Lib/Codec/enc_cdef.c:627-686), the per-block
 *                             strength tables passed as one dense array
 * Pinned against the reference by tests/test_leaves_oracle.py (oracle/_ref) and tests/golden/leaves.npz. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

ORC_API uint32_t orc_sad_16b(const uint16_t *src, uint32_t src_stride, const uint16_t *ref, uint32_t ref_stride, uint32_t height,
                             uint32_t width) {
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const int d = (int)src[(size_t)y * src_stride + x] - (int)ref[(size_t)y * ref_stride + x];
            sad += (uint32_t)(d < 0 ? -d : d);
        }
    return sad;
}

ORC_API void orc_initialize_buffer32(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value) {
    for (uint32_t i = 0; i < count128 * 4 + count32; i++) pointer[i] = value;
}

ORC_API void orc_residual8(const uint8_t *input, uint32_t input_stride, const uint8_t *pred, uint32_t pred_stride, int16_t *residual,
                           uint32_t residual_stride, uint32_t w, uint32_t h) {
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++)
            residual[(size_t)r * residual_stride + c] = (int16_t)((int16_t)input[(size_t)r * input_stride + c] - (int16_t)pred[(size_t)r * pred_stride + c]);
}
ORC_API void orc_residual16(const uint16_t *input, uint32_t input_stride, const uint16_t *pred, uint32_t pred_stride, int16_t *residual,
                            uint32_t residual_stride, uint32_t w, uint32_t h) {
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++)
            residual[(size_t)r * residual_stride + c] = (int16_t)((int16_t)input[(size_t)r * input_stride + c] - (int16_t)pred[(size_t)r * pred_stride + c]);
}

ORC_API uint64_t orc_spatial_sse8(const uint8_t *input, uint32_t input_offset, uint32_t input_stride, const uint8_t *recon,
                                  int32_t recon_offset, uint32_t recon_stride, uint32_t w, uint32_t h) {
    uint64_t acc = 0;
    input += input_offset, recon += recon_offset;
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++) {
            const int64_t d = (int64_t)input[(size_t)r * input_stride + c] - recon[(size_t)r * recon_stride + c];
            acc += (uint64_t)(d * d);
        }
    return acc;
}
/* offsets and strides count 16-bit samples; the byte pointers are reinterpreted (pic_operators.c:180-183) */
ORC_API uint64_t orc_spatial_sse16(const uint8_t *input, uint32_t input_offset, uint32_t input_stride, const uint8_t *pred,
                                   int32_t pred_offset, uint32_t pred_stride, uint32_t w, uint32_t h) {
    const uint16_t *a = (const uint16_t *)input + input_offset, *b = (const uint16_t *)pred + pred_offset;
    uint64_t        acc = 0;
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++) {
            const int64_t d = (int64_t)a[(size_t)r * input_stride + c] - b[(size_t)r * pred_stride + c];
            acc += (uint64_t)(d * d);
        }
    return acc;
}

/* ---- svt_mv_err_cost (mcomp.c:44-68); MV_COST_TYPE values: mcomp.h:29-36 */
enum { COST_ENTROPY, COST_L1_LOWRES, COST_L1_MIDRES, COST_L1_HDRES, COST_OPT, COST_NONE };
#define ORC_MV_UPP (1 << 14)
#define ORC_MV_LOW (-(1 << 14))
static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static int mv_err_cost(int16_t row, int16_t col, int16_t ref_row, int16_t ref_col, const int *mvjcost, const int *mvcost0,
                       const int *mvcost1, int error_per_bit, int type) {
    const int16_t dr = (int16_t)(row - ref_row), dc = (int16_t)(col - ref_col);          /* MV diff: int16 fields */
    const int16_t ar = (int16_t)abs(dr), ac = (int16_t)abs(dc);
    switch (type) {
    case COST_ENTROPY: {
        const int joint = dr == 0 ? (dc == 0 ? 0 : 1) : (dc == 0 ? 2 : 3);             /* rd_cost.c:55-60 */
        const int rate  = mvjcost[joint] + mvcost0[clip3(ORC_MV_LOW, ORC_MV_UPP, dr)] + mvcost1[clip3(ORC_MV_LOW, ORC_MV_UPP, dc)];
        return (int)(((int64_t)rate * error_per_bit + ((int64_t)1 << 13)) >> 14);      /* RDDIV 7 + PROB_COST 9 - EPB 6 + 4 */
    }
    case COST_L1_LOWRES: return (2 * (ar + ac)) >> 3;
    case COST_L1_MIDRES: return 0;
    case COST_L1_HDRES: return (ar + ac) >> 3;
    case COST_OPT: return (int)(((int64_t)((ar + ac) << 8) * error_per_bit + ((int64_t)1 << 13)) >> 14);
    default: return 0;
    }
}

/* mvcost0 / mvcost1 point at the CENTRE of the reference's component cost tables (index range MV_LOW..MV_UPP) */
ORC_API void orc_pme_sad_loop(int16_t ref_mv_row, int16_t ref_mv_col, int mv_cost_type, const int *mvjcost, const int *mvcost0,
                              const int *mvcost1, int error_per_bit, const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                              uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx,
                              int16_t *best_mvy, int16_t start_x, int16_t start_y, int16_t sa_w, int16_t sa_h, int16_t search_step,
                              int16_t mvx, int16_t mvy) {
    int16_t col_num = 0, step_x = 1;
    for (int16_t ys = 0; ys < sa_h; ys = (int16_t)(ys + search_step)) {
        for (int16_t xs = 0; xs < sa_w; xs = (int16_t)(xs + step_x)) {
            if ((sa_w - xs) < 8 && col_num == 0)
                continue;
            if (col_num == 7)
                col_num = 0, step_x = search_step;
            else
                col_num++, step_x = 1;
            uint32_t cost = 0;
            for (uint32_t y = 0; y < block_height; y++)
                for (uint32_t x = 0; x < block_width; x++) {
                    const int d = (int)src[(size_t)y * src_stride + x] - (int)ref[xs + (size_t)y * ref_stride + x];
                    cost += (uint32_t)(d < 0 ? -d : d);
                }
            const uint32_t px = (uint32_t)(start_x + xs), py = (uint32_t)(start_y + ys);
            const int16_t  col = (int16_t)(mvx + px * 8), row = (int16_t)(mvy + py * 8);
            cost += (uint32_t)mv_err_cost(row, col, ref_mv_row, ref_mv_col, mvjcost, mvcost0, mvcost1, error_per_bit, mv_cost_type);
            if (cost < *best_cost)
                *best_mvx = col, *best_mvy = row, *best_cost = cost;
        }
        ref += (ptrdiff_t)search_step * ref_stride;
    }
}

/* mse: [2][sb_count][stride] (luma table, then chroma table), `stride` >= end_gi entries per filter block */
ORC_API uint64_t orc_search_one_dual(int *lev0, int *lev1, int nb_strengths, const uint64_t *mse, int sb_count, int stride, int start_gi,
                                     int end_gi) {
    uint64_t       *tot = (uint64_t *)calloc((size_t)end_gi * end_gi + 1, sizeof(uint64_t));
    const uint64_t *m0 = mse, *m1 = mse + (size_t)sb_count * stride;
    for (int i = 0; i < sb_count; i++) {
        uint64_t best_mse = (uint64_t)1 << 63;
        for (int gi = 0; gi < nb_strengths; gi++) {
            const uint64_t curr = m0[(size_t)i * stride + lev0[gi]] + m1[(size_t)i * stride + lev1[gi]];
            if (curr < best_mse)
                best_mse = curr;
        }
        for (int j = start_gi; j < end_gi; j++)
            for (int k = start_gi; k < end_gi; k++) {
                const uint64_t curr = m0[(size_t)i * stride + j] + m1[(size_t)i * stride + k];
                tot[j * end_gi + k] += curr < best_mse ? curr : best_mse;
            }
    }
    uint64_t best_tot = (uint64_t)1 << 63;
    int      id0 = 0, id1 = 0;
    for (int j = start_gi; j < end_gi; j++)
        for (int k = start_gi; k < end_gi; k++)
            if (tot[j * end_gi + k] < best_tot)
                best_tot = tot[j * end_gi + k], id0 = j, id1 = k;
    lev0[nb_strengths] = id0, lev1[nb_strengths] = id1;
    free(tot);
    return best_tot;
}

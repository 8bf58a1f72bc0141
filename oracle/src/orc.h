/* oracle/src/orc.h — TEST INFRASTRUCTURE (CPU restatement of the reference hot path). */
#ifndef ORC_H
#define ORC_H

#include <stddef.h>
#include <stdint.h>

#include "../../include/svt_hip_me.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_API __attribute__((visibility("default")))

/* orc_sad.c */
ORC_API uint32_t orc_nxm_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                             uint32_t height, uint32_t width);
ORC_API void     orc_sad_loop_kernel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                     uint32_t ref_stride, uint32_t block_height, uint32_t block_width,
                                     uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center,
                                     uint32_t src_stride_raw, uint8_t skip_search_line,
                                     int16_t search_area_width, int16_t search_area_height);
ORC_API void orc_ext_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                               uint32_t ref_stride, uint32_t *p_best_sad_8x8,
                                               uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                               uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16,
                                               uint32_t *p_sad8x8, uint8_t sub_sad);
ORC_API void orc_ext_sad_calculation_32x32_64x64(const uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32,
                                                 uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                                 uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32);
ORC_API void orc_ext_all_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                                   uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8,
                                                   uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                                   uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8],
                                                   uint32_t p_eight_sad8x8[64][8], uint8_t sub_sad);
ORC_API void orc_ext_eight_sad_calculation_32x32_64x64(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32,
                                                       uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                                       uint32_t *p_best_mv64x64, uint32_t mv,
                                                       uint32_t p_sad32x32[4][8]);
ORC_API void orc_downsample_2d(const uint8_t *in, uint32_t in_stride, uint32_t in_w, uint32_t in_h, uint8_t *out,
                               uint32_t out_stride, uint32_t step);
ORC_API void orc_generate_padding(uint8_t *buf, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_w,
                                  uint32_t pad_h);
ORC_API void orc_pyramid_frame(const SvtHipPlane8 *full, const SvtHipPlane8 *quarter,
                               const SvtHipPlane8 *sixteenth, int hme_level1_enabled);
ORC_API uint64_t orc_compute_sub_mean_8x8(const uint8_t *in, uint16_t stride);
ORC_API uint64_t orc_compute_mean(const uint8_t *in, uint32_t stride, uint32_t w, uint32_t h);
ORC_API uint64_t orc_compute_mean_squared_values(const uint8_t *in, uint32_t stride, uint32_t w, uint32_t h);
ORC_API void     orc_compute_interm_var_four8x8(const uint8_t *in, uint16_t stride, uint64_t *mean,
                                                uint64_t *mean_sq);
ORC_API void     orc_block_mean_variance_b64(const uint8_t *blk, uint32_t stride, int full_precision,
                                             uint16_t var[85], uint64_t mean[85]);
ORC_API void     orc_variance_frame(const SvtHipPlane8 *full, uint16_t *variance, uint64_t *mean,
                                    int full_precision);
extern const uint8_t orc_z16[16];

/* orc_me.c — whole-picture open-loop ME; every pointer inside `jobs` is a HOST pointer here. */
ORC_API int32_t orc_me_frames(const SvtHipMeFrameJob *jobs, uint32_t n_jobs);
/* restrict to a b64 range [first, first+count) of job 0 (bounded cpu_baseline samples) */
ORC_API int32_t orc_me_frame_range(const SvtHipMeFrameJob *job, uint32_t first_b64, uint32_t count);

#ifdef __cplusplus
}
#endif
#endif

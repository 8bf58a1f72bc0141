/*
 * svt_hip_bind.h — what the reference encoder includes to get `--asm hip` (see tools/reference_hip.patch).
 *
 * EB_CPU_FLAGS_HIP is a new bit of EbCpuFlags (Source/API/EbSvtAv1.h:421-456: x86 uses bits 0-15, bit 63 is
 * EB_CPU_FLAGS_INVALID and EB_CPU_FLAGS_ALL = bits 0-61, what `--asm max` sets).  Bit 62 is therefore neither part of
 * "max" nor "invalid": only `--asm hip` selects it, and every SIMD bit stays clear, so svt_aom_setup_rtcd_internal
 * (aom_dsp_rtcd.c:187) first installs the C functions and svt_hip_bind_install() then swaps in the HIP leaves.
 */
#ifndef SVT_HIP_BIND_H
#define SVT_HIP_BIND_H

#include <stdint.h>

#define EB_CPU_FLAGS_HIP (1ULL << 62)

/* Loads libsvtav1_hip.so (env SVTAV1_HIP_LIB, else next to the executable, else the loader's search path),
 * initialises device `SVTAV1_HIP_DEVICE` (default 0) and assigns every Tier A export into the encoder's RTCD
 * pointers.  Returns the number of pointers installed, or -1 with the CPU pointers untouched (library or device
 * missing): the caller logs and carries on with its C kernels (SURVEY.md 8b "never abort").
 * Env SVTAV1_HIP_ONLY / SVTAV1_HIP_SKIP: comma-separated name prefixes to restrict the set (bisecting aid). */
int svt_hip_bind_install(char *msg, unsigned msg_len);

/* Step 2b (svt_hip_bind_me.c): whole-picture open-loop ME through svt_hip_me_frames.  Called from the b64 loop of
 * me_process.c in front of svt_aom_motion_estimation_b64; returns 0 when the block's results have been filled in from the
 * GPU, 1 when the caller must run the reference's own call (feature off, or a picture the batched path does not cover). */
struct PictureParentControlSet;
struct MeContext;
struct EbPictureBufferDesc;
int  svt_hip_bind_me_b64(struct PictureParentControlSet *pcs, uint32_t b64_index, struct MeContext *me_ctx, struct EbPictureBufferDesc *full,
                         struct EbPictureBufferDesc *quarter, struct EbPictureBufferDesc *sixteenth);
void svt_hip_bind_me_setup(void *(*sym)(const char *));

/* Step 6b (svt_hip_bind_tf.c): the block loop of produce_temporally_filtered_pic — _ld: of produce_temporally_filtered_pic_ld — through
 * svt_hip_tf_filter_picture.  Returns 0 when the picture has been filtered on the GPU (the caller skips its loop), 1 when the caller must
 * run its own loop. */
int  svt_hip_bind_tf_picture(struct PictureParentControlSet **pcs_list, struct EbPictureBufferDesc **pics, int index_center,
                             struct MeContext *ctx, int is_highbd);
int  svt_hip_bind_tf_picture_ld(struct PictureParentControlSet **pcs_list, struct EbPictureBufferDesc **pics, int index_center,
                             struct MeContext *ctx, int is_highbd);
void svt_hip_bind_tf_setup(void *(*sym)(const char *));

/* Step 3c (svt_hip_bind_tpl.c): the TPL dispenser of a picture through svt_hip_tpl_dispenser_frame.  Called in front of
 * tpl_mc_flow_dispenser_sb_generic in svt_aom_tpl_disp_kernel; 0 = the picture ran on the GPU, 1 = the caller runs its own call. */
int  svt_hip_bind_tpl_sb(struct PictureParentControlSet *pcs, int32_t frame_idx, uint32_t sb_index, int32_t qindex);
void svt_hip_bind_tpl_setup(void *(*sym)(const char *));


/* Step 2a (svt_hip_bind_pa.c): pyramid and block variances of the picture-analysis kernel through svt_hip_pyramid_frame /
 * svt_hip_variance_frame.  0 = done on the GPU, 1 = the caller runs its own code. */
struct SequenceControlSet;
int  svt_hip_bind_pa_pyramid(struct PictureParentControlSet *pcs, struct EbPictureBufferDesc *full, struct EbPictureBufferDesc *quarter,
                             struct EbPictureBufferDesc *sixteenth);
int  svt_hip_bind_pa_variance(struct SequenceControlSet *scs, struct PictureParentControlSet *pcs, struct EbPictureBufferDesc *full);
void svt_hip_bind_pa_setup(void *(*sym)(const char *));

/* Step 4 (svt_hip_bind_lf.c): the in-loop filters of a whole picture.  0 = done on the GPU, 1 = the caller runs its own code.
 * svt_hip_bind_dlf_deferred() != 0: SB-based deblocking is left out of the EncDec loop (coding_loop.c) and dlf_process.c makes one
 * frame call instead. */
struct PictureControlSet;
struct Av1Common;
struct Yv12BufferConfig;
struct RestorationTileLimits;
int  svt_hip_bind_dlf_deferred(void);
int  svt_hip_bind_dlf_frame(struct EbPictureBufferDesc *frame_buffer, struct PictureControlSet *pcs, int32_t plane_start, int32_t plane_end);
/* one trial of the level search (try_filter_frame): filter + picture_sse_calculations on the device; 0 = *filt_err is set */
int  svt_hip_bind_dlf_try(struct EbPictureBufferDesc *frame_buffer, struct PictureControlSet *pcs, int32_t plane, int64_t *filt_err);
int  svt_hip_bind_cdef_seg(struct PictureControlSet *pcs, struct SequenceControlSet *scs, uint32_t segment_index);
int  svt_hip_bind_cdef_frame(struct SequenceControlSet *scs, struct PictureControlSet *pcs);
int  svt_hip_bind_wiener_stats(struct PictureControlSet *pcs, int plane, int rest_unit_idx, int wiener_win, const uint8_t *dgd, const uint8_t *src,
                               const struct RestorationTileLimits *limits, int dgd_stride, int src_stride, int highbd, int bit_depth, int64_t *M,
                               int64_t *H);
int  svt_hip_bind_lr_frame(struct Yv12BufferConfig *frame, struct Av1Common *cm, int32_t optimized_lr);
void svt_hip_bind_lf_setup(void *(*sym)(const char *));

/* Step 3a (svt_hip_bind_txt.c): the forward transforms of every transform type tx_type_search can reach for one transform block through
 * ONE svt_hip_txfm_quant_batch call.  prepare returns NULL when declined; take returns 0 when it filled the coefficients in. */
void *svt_hip_bind_txt_prepare(const int16_t *residual, uint32_t stride, int tx_size, int bit_depth, int pf_shape, uint32_t type_mask);
int   svt_hip_bind_txt_take(void *cache, int tx_type, int32_t *coeff, uint64_t *three_quad_energy);
void  svt_hip_bind_txt_setup(void *(*sym)(const char *));

#endif

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

/* Step 6b (svt_hip_bind_tf.c): the block loop of produce_temporally_filtered_pic through svt_hip_tf_filter_picture.  Returns 0
 * when the picture has been filtered on the GPU (the caller skips its loop), 1 when the caller must run its own loop. */
int  svt_hip_bind_tf_picture(struct PictureParentControlSet **pcs_list, struct EbPictureBufferDesc **pics, int index_center,
                             struct MeContext *ctx, int is_highbd);
void svt_hip_bind_tf_setup(void *(*sym)(const char *));

/* Step 3c (svt_hip_bind_tpl.c): the TPL dispenser of a picture through svt_hip_tpl_dispenser_frame.  Called in front of
 * tpl_mc_flow_dispenser_sb_generic in svt_aom_tpl_disp_kernel; 0 = the picture ran on the GPU, 1 = the caller runs its own call. */
int  svt_hip_bind_tpl_sb(struct PictureParentControlSet *pcs, int32_t frame_idx, uint32_t sb_index, int32_t qindex);
void svt_hip_bind_tpl_setup(void *(*sym)(const char *));

#endif

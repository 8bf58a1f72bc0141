/*
 * svt_hip_tpl.h — C-ABI for the TPL dispenser of one picture.
 *
 * Reference interface replaced (paths relative to /root/reference):
 *   Source/Lib/Codec/src_ops_process.c:519-1207   tpl_mc_flow_dispenser_sb_generic, called per 64x64 block from
 *                                                 tpl_mc_flow_dispenser (:1348-1410) / svt_aom_tpl_disp_kernel (:1964)
 * for the configurations the reference runs at presets M5 and faster (tpl levels 3, 4 and 5 of set_tpl_params,
 * initial_rc_process.c:284-296, 345-382): 16x16 blocks (dispenser_search_level 0)
 * or 32x32 blocks whose transform runs on every 4th row (level 5: dispenser_search_level 1, subsample_tx 2, TX_32X8), DC intra
 * prediction only (intra_mode_end == DC_PRED), SAD in the source-based search, full-pel vectors straight from the open-loop
 * ME results — or, level 3 (quarter_pel), refined to a quarter sample and compensated with the 8-tap kernels —, no rate estimate
 * (compute_rate 0), any coefficient shape (pf_shape).  Per block:
 *   source-based path  DC prediction from the SOURCE neighbours and its sub-sampled SAD; every single-reference ME candidate,
 *                      vector clipped to the TPL padding, sub-sampled SAD against the reference's source picture; for an
 *                      inter winner residual -> DCT 16x16 (32x8) -> svt_av1_quantize_fp -> svt_av1_block_error  (srcrf_dist)
 *   reconstruction     inter: the block of the reference's RECONSTRUCTION at that vector; intra: DC prediction from the
 *                      reconstructed neighbours of this picture; residual -> DCT -> quantise -> error (recrf_dist) ->
 *                      inverse transform + reconstruction into the TPL reconstruction picture (sub-sampled: the rows left out
 *                      repeat the row above them)
 *   result_model_store TplStats on the synthesizer's 32x32 / 16x16 / 8x8 grid (cells beyond the grid are dropped), TplSrcStats
 * The intra blocks depend on the reconstruction of their left / top / top-left neighbours: the kernel runs the source-based
 * path of every block in parallel and orders only the reconstruction of intra blocks behind their neighbours' (flags in
 * device memory; block indices are handed out by a ticket counter, so no dispatch order is assumed).  The caller orders pictures (a reference picture's reconstruction must be complete before this call).
 * A block hands its reconstruction to the other compute units by storing its samples once more with device scope (written through the
 * XCD's L2), waiting for those stores and then setting its flags; this needs 4-byte aligned reconstruction rows (stride and sample
 * (0,0) address multiples of 4), otherwise — or with SVTAV1_HIP_TPL_FENCE set in the environment — a device-scope release fence per
 * block (an L2 write-back, about three times slower) does it.
 * Not provided: the other intra modes and the SATD source search (tpl levels 1, 2), 64x64 dispenser blocks, subsample_tx 1, the
 * rate estimate.
 */
#ifndef SVT_HIP_TPL_H
#define SVT_HIP_TPL_H

#include "svt_hip.h"
#include "svt_hip_me.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SvtHipTplStats { /* TplStats, coding_unit.h:312-321 */
    int64_t  srcrf_dist, recrf_dist, srcrf_rate, recrf_rate, mc_dep_rate, mc_dep_dist;
    int16_t  mv_row, mv_col;
    uint32_t pad_;
    uint64_t ref_frame_poc;
} SvtHipTplStats;

typedef struct SvtHipTplSrcStats { /* TplSrcStats, coding_unit.h:323-331 */
    int64_t  srcrf_dist, srcrf_rate;
    uint64_t ref_frame_poc;
    int16_t  mv_row, mv_col;
    int32_t  best_rf_idx;
    uint8_t  best_mode;       /* PredictionMode: 0 DC_PRED, 16 NEWMV */
    uint8_t  best_intra_mode; /* 0 */
    uint8_t  pad_[6];
} SvtHipTplSrcStats;

typedef struct SvtHipTplRef { /* one entry of pcs->tpl_data.tpl_ref_ds_ptr_array[list][ref] */
    const uint8_t *src;       /* luma of the reference's SOURCE picture, sample (0,0) (picture_ptr) */
    const uint8_t *recon;     /* what the reconstruction path predicts from, sample (0,0): mc_flow_rec_picture_buffer of
                               * that picture when ref_in_slide_window, else the same plane as src */
    uint32_t       src_stride, recon_stride;
    uint64_t       picture_number;
    uint16_t       max_width, max_height; /* EbPictureBufferDesc::max_width / max_height of picture_ptr (the vector clip) */
    uint8_t        usable;    /* 0: candidates of this reference are skipped (inside the window without valid TPL data, :780-782) */
    uint8_t        pad_[3];
} SvtHipTplRef;

typedef struct SvtHipTplFrameJob {
    SvtHipPlane8 src;    /* pcs->enhanced_pic luma */
    SvtHipPlane8 recon;  /* enc_ctx->mc_flow_rec_picture_buffer[frame_idx] luma: written */
    SvtHipTplRef ref[SVT_HIP_ME_MAX_LIST][SVT_HIP_ME_MAX_REF];
    /* the open-loop ME results of this picture (SvtHipMeFrameOut of svt_hip_me_frames) */
    const uint32_t *me_mv_array;
    const uint8_t  *me_candidate_array, *total_me_candidate_index;
    uint8_t  max_cand, max_refs, max_l0, enable_me_16x16; /* MotionEstimationData / pcs */
    uint8_t  stored_pus;         /* PUs per b64 in the three arrays (svt_hip_me_stored_pus) */
    uint8_t  pf_shape;           /* 0 default, 1 N2, 2 N4 (EB_TRANS_COEFF_SHAPE) */
    uint8_t  disable_intra_pred; /* tpl_ctrls.disable_intra_pred_nref && temporal_layer_index == hierarchical_levels */
    uint8_t  is_ref;             /* pcs->tpl_data.is_ref */
    uint8_t  i_slice;            /* pcs->slice_type == I_SLICE: no ME candidates */
    uint8_t  tpl_i_slice;        /* pcs->tpl_data.tpl_slice_type == I_SLICE */
    uint8_t  src_data_ready;     /* pcs->tpl_src_data_ready: the source-based results are READ from src_stats */
    uint8_t  store_src_stats;    /* scs->tpl_lad_mg > 0 */
    uint8_t  synth_blk_size;     /* 16 or 8 (tpl_ctrls.synth_blk_size; 32 too with 32x32 blocks): grid of `stats` */
    uint8_t  blk_size;           /* 0 / 16: 16x16 blocks (dispenser_search_level 0); 32: 32x32 blocks (level 1) */
    uint8_t  subsample_tx;       /* tpl_ctrls.subsample_tx: 0 with 16x16 blocks; 0 or 2 (transform TX_32X8 on every 4th row) with 32x32 */
    uint8_t  publish_fence;      /* how a block hands its reconstruction to the blocks that wait for it (other CUs / XCDs).
                                  * 0 (default): every lane re-stores its samples with agent scope (global_store ... sc1: written
                                  * through this XCD's L2 to the device's coherence point), waits for the acknowledgement and
                                  * lane 0 sets the done-flags; used when the reconstruction rows are 4-byte aligned.
                                  * 1: one agent-scope RELEASE fence per block (buffer_wbl2: writes back the whole L2, ~3x slower);
                                  * also what unaligned rows get.  Consumers poll the flag and then issue an agent-scope ACQUIRE
                                  * fence (buffer_inv sc1) in both cases, so a line of the reconstruction that their XCD's L2 cached
                                  * BEFORE the neighbour was published is dropped.  Mode 0 relies on the gfx942 / gfx950 memory
                                  * model as LLVM's AMDGPUUsage documents it (agent-scope atomic store = sc1 write-through;
                                  * agent-scope acquire = buffer_inv sc1); it is not a release / acquire pair of the C++ model. */
    /* quants_8bit / deq_8bit of the picture's qindex: [0] DC, [1] AC */
    int16_t  round_fp[2], quant_fp[2], dequant[2];
    uint8_t  quarter_pel; /* tpl_ctrls.subpel_depth == QUARTER_PEL (tpl level 3 = presets M5 / M6; 16x16 blocks only): every candidate
                           * vector is refined by tpl_subpel_search (src_ops_process.c:418-517: svt_av1_find_best_sub_pixel_tree_pruned
                           * with two rounds — half, quarter — of the four cardinal neighbours, bilinear sub-pixel variance, no vector
                           * cost, no diagonal), and a fractional vector is compensated with the regular 8-tap kernels for the SAD,
                           * the source-based residual and the reconstruction.  mi_rows / mi_cols of Av1Common are taken as the
                           * picture size rounded up to 8, in 4x4 units */
    uint8_t  pad2_;
    SvtHipTplStats    *stats;     /* [rows][stride]: stride = (aligned_width + 15) / 16 for synth_blk_size 16, twice that for 8, (aligned_width + 31) / 32 for 32 */
    SvtHipTplSrcStats *src_stats; /* [..][(aligned_width + 15) >> 4] */
    void              *workspace; /* device scratch of svt_hip_tpl_workspace_bytes(): the done-flags of the blocks, one status word
                                   * (first uint32 behind the flags: non-zero if a dependency wait ran into its bound — the
                                   * results are then unreliable; never observed) */
    uint64_t           workspace_bytes;
} SvtHipTplFrameJob;

SVT_HIP_API uint64_t svt_hip_tpl_workspace_bytes(uint32_t width, uint32_t height);
/* byte offset of the status word inside the workspace */
SVT_HIP_API uint64_t svt_hip_tpl_status_offset(uint32_t width, uint32_t height);

/* `job` is a HOST struct, every pointer inside is device memory.  Asynchronous on `stream`. */
SVT_HIP_API int32_t svt_hip_tpl_dispenser_frame(const SvtHipTplFrameJob *job, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_TPL_H */

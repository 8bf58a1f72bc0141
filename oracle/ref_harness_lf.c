/*
 * oracle/ref_harness_lf.c — TEST INFRASTRUCTURE.  Glue compiled against the reference's own headers and linked
 * into oracle/_ref/libsvtref.so: builds the reference's control structures around flat test inputs and calls the
 * REAL svt_av1_loop_filter_init / svt_av1_loop_filter_frame (deblocking_filter.c:35, :624).  No reference code
 * is restated here; ref_lf_gather() is the gather loop INTEGRATION.md shows for mi_grid_base -> SvtHipLfMi.
 */
#include <stdlib.h>
#include <string.h>

#include "common_dsp_rtcd.h"
#include "deblocking_common.h"
#include "deblocking_filter.h"
#include "definitions.h"
#include "pcs.h"
#include "sequence_control_set.h"
#include "utility.h"

#include "../include/svt_hip_lf.h"

#define REF_API __attribute__((visibility("default")))

/* Raw per-mi mode info as the encoder holds it (one byte array per field, [mi_rows * mi_stride]). */
typedef struct RefLfModeInfo {
    const uint8_t *bsize, *tx_depth, *skip, *ref_frame0, *mode, *segment_id;
} RefLfModeInfo;

/* Frame-header side of the loop filter (FrameHeader.loop_filter_params + the LF segmentation features). */
typedef struct RefLfHeader {
    int32_t filter_level[2], filter_level_u, filter_level_v, sharpness_level;
    uint8_t mode_ref_delta_enabled;
    int8_t  ref_deltas[8], mode_deltas[2];
    uint8_t segmentation_enabled;
    int16_t seg_lf_data[8][4];    /* SEG_LVL_ALT_LF_Y_V, _Y_H, _U, _V */
    uint8_t seg_lf_enabled[8][4];
} RefLfHeader;

REF_API void ref_lf_gather(uint32_t n, const RefLfModeInfo *m, SvtHipLfMi *out) {
    for (uint32_t i = 0; i < n; i++) {
        const int skip_inter = m->skip[i] && is_inter_block_no_intrabc((MvReferenceFrame)m->ref_frame0[i]);
        out[i].bsize         = m->bsize[i];
        out[i].tx_size_y     = (uint8_t)tx_depth_to_tx_size[skip_inter ? 0 : m->tx_depth[i]][m->bsize[i]];
        out[i].tx_size_uv    = (uint8_t)av1_get_max_uv_txsize((BlockSize)m->bsize[i], 1, 1);
        out[i].skip_inter    = (uint8_t)skip_inter;
        out[i].segment_id    = m->segment_id[i];
        out[i].ref_frame0    = m->ref_frame0[i];
        out[i].mode_lf       = (uint8_t)mode_lf_lut[m->mode[i]];
        out[i].reserved      = 0;
    }
}

/* Runs the reference's frame deblocking in place on f->plane[] (HOST pointers here).  f->mi is ignored (the real
 * ModeInfo grid is rebuilt from `m`); f->lvl receives the level table the reference derived from `hdr`. */
REF_API int ref_loop_filter_frame(SvtHipLfFrame *f, const RefLfModeInfo *m, const RefLfHeader *hdr, int sb_size) {
    SequenceControlSet      *scs  = calloc(1, sizeof(*scs));
    PictureParentControlSet *ppcs = calloc(1, sizeof(*ppcs));
    PictureControlSet       *pcs  = calloc(1, sizeof(*pcs));
    EbPictureBufferDesc     *pic  = calloc(1, sizeof(*pic));
    const size_t             nmi  = (size_t)f->mi_rows * f->mi_stride;
    ModeInfo                *mis  = calloc(nmi, sizeof(ModeInfo));
    ModeInfo               **grid = calloc(nmi, sizeof(ModeInfo *));
    if (!scs || !ppcs || !pcs || !pic || !mis || !grid)
        return -1;
    for (size_t i = 0; i < nmi; i++) {
        BlockModeInfoEnc *b = &mis[i].mbmi.block_mi;
        b->bsize            = (BlockSize)m->bsize[i];
        b->tx_depth         = m->tx_depth[i];
        b->skip             = m->skip[i] & 1;
        b->ref_frame[0]     = (MvReferenceFrame)m->ref_frame0[i];
        b->mode             = (PredictionMode)m->mode[i];
        b->segment_id       = m->segment_id[i];
        grid[i]             = &mis[i];
    }
    scs->seq_header.sb_size             = sb_size == 128 ? BLOCK_128X128 : BLOCK_64X64;
    scs->sb_size                        = (uint16_t)sb_size;
    scs->is_16bit_pipeline              = f->is_16bit;
    scs->static_config.encoder_bit_depth = f->bit_depth;
    scs->max_input_luma_width           = (uint16_t)(f->mi_cols * 4);
    scs->max_input_luma_height          = (uint16_t)(f->mi_rows * 4);
    scs->max_input_pad_right            = (uint16_t)(f->mi_cols * 4 - f->width);
    scs->max_input_pad_bottom           = (uint16_t)(f->mi_rows * 4 - f->height);
    pcs->scs = scs, pcs->ppcs = ppcs, ppcs->scs = scs;
    pcs->mi_grid_base = grid, pcs->mi_stride = (int32_t)f->mi_stride;
    ppcs->aligned_width = (uint16_t)(f->mi_cols * 4), ppcs->aligned_height = (uint16_t)(f->mi_rows * 4);
    struct LoopFilter *lf = &ppcs->frm_hdr.loop_filter_params;
    lf->filter_level[0] = hdr->filter_level[0], lf->filter_level[1] = hdr->filter_level[1];
    lf->filter_level_u = hdr->filter_level_u, lf->filter_level_v = hdr->filter_level_v;
    lf->sharpness_level = hdr->sharpness_level, lf->mode_ref_delta_enabled = hdr->mode_ref_delta_enabled;
    memcpy(lf->ref_deltas, hdr->ref_deltas, 8), memcpy(lf->mode_deltas, hdr->mode_deltas, 2);
    SegmentationParams *sp    = &ppcs->frm_hdr.segmentation_params;
    sp->segmentation_enabled  = hdr->segmentation_enabled;
    static const int feat[4]  = {SEG_LVL_ALT_LF_Y_V, SEG_LVL_ALT_LF_Y_H, SEG_LVL_ALT_LF_U, SEG_LVL_ALT_LF_V};
    for (int s = 0; s < 8; s++)
        for (int k = 0; k < 4; k++)
            sp->feature_data[s][feat[k]] = hdr->seg_lf_data[s][k], sp->feature_enabled[s][feat[k]] = hdr->seg_lf_enabled[s][k];
    pic->buffer_y = f->plane[0], pic->buffer_cb = f->plane[1], pic->buffer_cr = f->plane[2];
    pic->stride_y = (uint16_t)f->stride[0], pic->stride_cb = (uint16_t)f->stride[1], pic->stride_cr = (uint16_t)f->stride[2];
    pic->org_x = pic->org_y = 0;
    pic->bit_depth = f->bit_depth > 8 ? EB_TEN_BIT : EB_EIGHT_BIT;
    pic->width = (uint16_t)f->width, pic->height = (uint16_t)f->height;
    svt_av1_loop_filter_init(pcs);                                         /* dlf_process.c:103 */
    svt_av1_loop_filter_frame(pic, pcs, f->plane_start, f->plane_end);     /* dlf_process.c:106 */
    memcpy(f->lvl, ppcs->lf_info.lvl, sizeof(f->lvl));
    free(grid), free(mis), free(pic), free(pcs), free(ppcs), free(scs);
    return 0;
}

REF_API size_t ref_sizeof_lf_lvl(void) { return sizeof(((LoopFilterInfoN *)0)->lvl); }

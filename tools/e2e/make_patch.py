#!/usr/bin/env python3
"""Generates tools/reference_hip.patch (and the per-file pieces the e2e build streams through `patch -o -`).

The patch is what a maintainer of the reference applies to get `--asm hip`: it touches two
files of the reference, Source/Lib/Globals/enc_settings.c (the --asm keyword) and Source/Lib/Globals/enc_handle.c (keep
the new flag through the CPU-capability mask, install the HIP leaves right behind the RTCD set-up calls), and relies on
tools/e2e/svt_hip_bind.{c,h} of this repository.  The edits are made in memory on the files where they lie under
/root/reference; nothing of the reference is copied to disk, only the unified diff (a few context lines) is written.
"""
import difflib
import os
import sys

REF = os.environ.get("REF", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def edit(text, old, new, count=1):
    assert text.count(old) == count, f"expected {count} x {old!r}, found {text.count(old)}"
    return text.replace(old, new)


def patch_enc_settings(t):
    t = edit(t, '#include "EbVersion.h"\n', '#include "EbVersion.h"\n#include "svt_hip_bind.h"\n')
    t = edit(t, '        {"c", 0},\n        {"0", 0},\n',
             '        {"c", 0},\n        {"0", 0},\n        {"hip", EB_CPU_FLAGS_HIP}, /* C table + HIP leaves (libsvtav1_hip) */\n')
    return t


def patch_enc_handle(t):
    t = edit(t, '#include "EbVersion.h"\n', '#include "EbVersion.h"\n#include "svt_hip_bind.h"\n')
    t = edit(t, "        scs->static_config.use_cpu_flags &= cpu_flags_to_use;\n",
             "        scs->static_config.use_cpu_flags &= cpu_flags_to_use | EB_CPU_FLAGS_HIP;\n")
    t = edit(t, "        scs->static_config.use_cpu_flags &= 0;\n",
             "        scs->static_config.use_cpu_flags &= EB_CPU_FLAGS_HIP;\n")
    old = ("    svt_aom_setup_rtcd_internal(enc_handle_ptr->scs_instance_array[0]->scs->static_config.use_cpu_flags);\n")
    new = old + (
        "    if (enc_handle_ptr->scs_instance_array[0]->scs->static_config.use_cpu_flags & EB_CPU_FLAGS_HIP) {\n"
        "        char hip_msg[256];\n"
        "        if (svt_hip_bind_install(hip_msg, sizeof(hip_msg)) < 0)\n"
        "            SVT_WARN(\"HIP hot path unavailable (%s); keeping the CPU kernels\\n\", hip_msg);\n"
        "        else\n"
        "            SVT_INFO(\"HIP hot path: %s\\n\", hip_msg);\n"
        "    }\n")
    t = edit(t, old, new)
    return t


def patch_me_process(t):
    """Step 2b: the batched open-loop ME in front of the per-block call (the reference's loop, counters and mutex stay)."""
    t = edit(t, '#include "me_process.h"\n', '#include "me_process.h"\n#include "svt_hip_bind.h"\n')
    old = ("                            svt_aom_motion_estimation_b64(pcs,\n"
           "                                b64_index,\n")
    new = ("                            if (svt_hip_bind_me_b64(pcs, b64_index, me_context_ptr->me_ctx, input_padded_pic,\n"
           "                                                    quarter_picture_ptr, sixteenth_picture_ptr))\n" + old)
    return edit(t, old, new)


def patch_temporal_filtering(t):
    """Step 6b: the whole-picture temporal filter in front of the block loops of produce_temporally_filtered_pic and of its low-delay
    variant (two identical loop heads, in this order)."""
    t = edit(t, '#include "temporal_filtering.h"\n', '#include "temporal_filtering.h"\n#include "svt_hip_bind.h"\n')
    old = "    for (uint32_t blk_row = y_b64_start_idx; blk_row < y_b64_end_idx; blk_row++) {\n"
    assert t.count(old) == 2
    i = t.index(old)
    j = t.index(old, i + 1)  # produce_temporally_filtered_pic_ld: its decay factors are computed above the loop as well
    t = t[:j] + "    if (svt_hip_bind_tf_picture_ld(pcs_list, list_input_picture_ptr, index_center, ctx, is_highbd))\n" + t[j:]
    return t[:i] + "    if (svt_hip_bind_tf_picture(pcs_list, list_input_picture_ptr, index_center, ctx, is_highbd))\n" + t[i:]


def patch_src_ops_process(t):
    """Step 3c: the whole-picture TPL dispenser in front of the per-block call of the segment loop (no-tiles path)."""
    t = edit(t, '#include "src_ops_process.h"\n', '#include "src_ops_process.h"\n#include "svt_hip_bind.h"\n')
    old = ("                        tpl_mc_flow_dispenser_sb_generic(pcs->scs->enc_ctx,\n"
           "                                                         scs,\n")
    new = ("                        if (svt_hip_bind_tpl_sb(pcs, frame_idx, context_ptr->sb_index, in_results_ptr->qIndex))\n" + old)
    return edit(t, old, new)


def patch_deblocking_filter(t):
    """Step 4a: the whole-picture deblocking in front of the SB loop of svt_av1_loop_filter_frame, and the device-side trial of the
    level search in try_filter_frame (a trial the glue declines still reaches the first hook through the reference's own sequence)."""
    t = edit(t, '#include "deblocking_filter.h"\n', '#include "deblocking_filter.h"\n#include "svt_hip_bind.h"\n')
    old = ("    uint32_t picture_height_in_sb = (pcs->ppcs->aligned_height + scs->sb_size - 1) / scs->sb_size;\n\n"
           "    svt_av1_loop_filter_frame_init(&pcs->ppcs->frm_hdr, &pcs->ppcs->lf_info, plane_start, plane_end);\n")
    new = ("    uint32_t picture_height_in_sb = (pcs->ppcs->aligned_height + scs->sb_size - 1) / scs->sb_size;\n\n"
           "    if (svt_hip_bind_dlf_frame(frame_buffer, pcs, plane_start, plane_end) == 0)\n"
           "        return;\n"
           "    svt_av1_loop_filter_frame_init(&pcs->ppcs->frm_hdr, &pcs->ppcs->lf_info, plane_start, plane_end);\n")
    t = edit(t, old, new)
    # a trial of the level search (try_filter_frame): filter + picture_sse_calculations on the device, nothing to restore
    old = ("    svt_av1_loop_filter_frame(recon_buffer, pcs, plane, plane + 1);\n\n"
           "    filt_err = picture_sse_calculations(pcs, recon_buffer, plane);\n")
    new = ("    if (svt_hip_bind_dlf_try(recon_buffer, pcs, plane, &filt_err) == 0)\n"
           "        return filt_err;\n" + old)
    return edit(t, old, new)


def patch_coding_loop(t):
    """Step 4a, SB-based deblocking (presets M6 and faster): the per-SB filter call is left out when the frame call of
    dlf_process.c takes over (svt_hip_bind_dlf_deferred)."""
    t = edit(t, '#include "coding_loop.h"\n', '#include "coding_loop.h"\n#include "svt_hip_bind.h"\n')
    old = "            svt_aom_loop_filter_sb(recon_buffer, pcs, sb_org_y >> 2, sb_org_x >> 2, 0, 3, last_col);\n"
    new = "            if (!svt_hip_bind_dlf_deferred())\n    " + old
    return edit(t, old, new)


def patch_dlf_process(t):
    """Step 4a: ... and the frame call that replaces the per-SB calls (levels were picked from Q at the first SB, coding_loop.c:2265-2270)."""
    t = edit(t, '#include "dlf_process.h"\n', '#include "dlf_process.h"\n#include "svt_hip_bind.h"\n')
    old = "            svt_av1_loop_filter_frame(recon_buffer, pcs, 0, 3);\n        }\n"
    new = (old[:-len("        }\n")] +
           "        } else if (dlf_enable_flag && tg_count == 1 && svt_hip_bind_dlf_deferred() &&\n"
           "                   (pcs->ppcs->frm_hdr.loop_filter_params.filter_level[0] || pcs->ppcs->frm_hdr.loop_filter_params.filter_level[1])) {\n"
           "            EbPictureBufferDesc *recon_buffer;\n"
           "            svt_aom_get_recon_pic(pcs, &recon_buffer, is_16bit);\n"
           "            svt_av1_loop_filter_frame(recon_buffer, pcs, 0, 3);\n"
           "        }\n")
    return edit(t, old, new)


def patch_cdef_process(t):
    """Step 4b: the whole-picture CDEF search in front of the filter-block loop of cdef_seg_search."""
    t = edit(t, '#include "cdef_process.h"\n', '#include "cdef_process.h"\n#include "svt_hip_bind.h"\n')
    old = ("static void cdef_seg_search(PictureControlSet *pcs, SequenceControlSet *scs, uint32_t segment_index) {\n"
           "    struct PictureParentControlSet *ppcs     = pcs->ppcs;\n")
    new = ("static void cdef_seg_search(PictureControlSet *pcs, SequenceControlSet *scs, uint32_t segment_index) {\n"
           "    if (svt_hip_bind_cdef_seg(pcs, scs, segment_index) == 0)\n"
           "        return;\n"
           "    struct PictureParentControlSet *ppcs     = pcs->ppcs;\n")
    return edit(t, old, new)


def patch_enc_cdef(t):
    """Step 4b: the whole-picture CDEF application in front of svt_av1_cdef_frame's filter-block loop."""
    t = edit(t, '#include "enc_cdef.h"\n', '#include "enc_cdef.h"\n#include "svt_hip_bind.h"\n')
    old = ("void svt_av1_cdef_frame(SequenceControlSet *scs, PictureControlSet *pcs) {\n"
           "    struct PictureParentControlSet *ppcs     = pcs->ppcs;\n")
    new = ("void svt_av1_cdef_frame(SequenceControlSet *scs, PictureControlSet *pcs) {\n"
           "    if (svt_hip_bind_cdef_frame(scs, pcs) == 0)\n"
           "        return;\n"
           "    struct PictureParentControlSet *ppcs     = pcs->ppcs;\n")
    return edit(t, old, new)


def patch_restoration(t):
    """Step 4e: the whole-picture restoration in front of svt_av1_loop_restoration_filter_frame's plane loop."""
    t = edit(t, '#include "restoration.h"\n', '#include "restoration.h"\n#include "svt_hip_bind.h"\n')
    old = ("    static const CopyFun copy_funs[3] = {\n"
           "        svt_aom_yv12_copy_y_c, svt_aom_yv12_copy_u_c, svt_aom_yv12_copy_v_c}; //CHKN SSE\n\n")
    new = old + ("    if (svt_hip_bind_lr_frame(frame, cm, optimized_lr) == 0)\n"
                 "        return;\n")
    return edit(t, old, new)


def patch_restoration_pick(t):
    """Step 4d: the Wiener statistics of a whole plane in front of svt_av1_compute_stats(_highbd) in search_wiener_seg."""
    t = edit(t, '#include "restoration_pick.h"\n', '#include "restoration_pick.h"\n#include "svt_hip_bind.h"\n')
    old = ("        int32_t              vfilterd[WIENER_WIN], hfilterd[WIENER_WIN];\n\n"
           "        if (cm->use_highbitdepth)\n"
           "            svt_av1_compute_stats_highbd(wiener_win,\n")
    new = ("        int32_t              vfilterd[WIENER_WIN], hfilterd[WIENER_WIN];\n\n"
           "        if (svt_hip_bind_wiener_stats(cm->child_pcs, rsc->plane, rest_unit_idx, wiener_win, rsc->dgd_buffer, rsc->src_buffer, limits,\n"
           "                                      rsc->dgd_stride, rsc->src_stride, cm->use_highbitdepth, cm->bit_depth, M, H) == 0)\n"
           "            ;\n"
           "        else if (cm->use_highbitdepth)\n"
           "            svt_av1_compute_stats_highbd(wiener_win,\n")
    assert t.count(old) >= 1
    i = t.index(old, t.index("static void search_wiener_seg("))
    return t[:i] + new + t[i + len(old):]


def patch_pic_analysis_process(t):
    """Step 2a: pyramid and block variances of a whole picture in front of the reference's own loops."""
    t = edit(t, '#include "pic_analysis_process.h"\n', '#include "pic_analysis_process.h"\n#include "svt_hip_bind.h"\n')
    old = ("    // Downsample input picture for HME L0 and L1\n"
           "    if (pcs->enable_hme_flag || pcs->tf_enable_hme_flag) {\n")
    new = ("    if (svt_hip_bind_pa_pyramid(pcs, input_padded_pic, quarter_picture_ptr, sixteenth_picture_ptr) == 0)\n"
           "        return;\n" + old)
    t = edit(t, old, new)
    old = ("    uint16_t b64_total_count  = pcs->b64_total_count;\n\n"
           "    for (uint16_t b64_idx = 0; b64_idx < b64_total_count; ++b64_idx) {\n"
           "        B64Geom *b64_geom = &pcs->b64_geom[b64_idx];\n")
    new = ("    uint16_t b64_total_count  = pcs->b64_total_count;\n\n"
           "    if (svt_hip_bind_pa_variance(scs, pcs, input_padded_pic) == 0)\n"
           "        return;\n"
           "    for (uint16_t b64_idx = 0; b64_idx < b64_total_count; ++b64_idx) {\n"
           "        B64Geom *b64_geom = &pcs->b64_geom[b64_idx];\n")
    return edit(t, old, new)


def patch_product_coding_loop(t):
    """Step 3a: the forward transforms of all transform types tx_type_search can reach for one transform block, in ONE batch in front of
    the loop; inside the loop the reference's svt_aom_estimate_transform call runs only for a type the batch does not hold."""
    t = edit(t, '#include "full_loop.h"\n', '#include "full_loop.h"\n#include "svt_hip_bind.h"\n')
    old = ("    int          tx_type_tot_group     = get_tx_type_group(ctx, cand_bf, only_dct_dct);\n"
           "    for (int tx_type_group_idx = 0; tx_type_group_idx < tx_type_tot_group; ++tx_type_group_idx) {\n"
           "        uint32_t best_tx_non_coeff = 64 * 64;\n")
    new = ("    int          tx_type_tot_group     = get_tx_type_group(ctx, cand_bf, only_dct_dct);\n"
           "    void        *hip_txt               = NULL;\n"
           "    if (!tx_search_skip_flag) { /* the types the loop below can reach (its static filters) */\n"
           "        uint32_t hip_mask = 0;\n"
           "        for (int g = 0; g < tx_type_tot_group; ++g)\n"
           "            for (int i = 0; i < TX_TYPES; ++i) {\n"
           "                const int tt = pcs->ppcs->sc_class1 ? tx_type_group_sc[g][i] : tx_type_group[g][i];\n"
           "                if (tt == INVALID_TX_TYPE)\n"
           "                    break;\n"
           "                if (tt != DCT_DCT && (only_dct_dct || av1_ext_tx_used[tx_set_type][tt] == 0))\n"
           "                    continue;\n"
           "                hip_mask |= 1u << tt;\n"
           "            }\n"
           "        hip_txt = svt_hip_bind_txt_prepare(&(((int16_t *)cand_bf->residual->buffer_y)[ctx->blk_geom->tx_org_x[is_inter][ctx->tx_depth][ctx->txb_itr] +\n"
           "                                               ctx->blk_geom->tx_org_y[is_inter][ctx->tx_depth][ctx->txb_itr] * cand_bf->residual->stride_y]),\n"
           "                                           cand_bf->residual->stride_y, tx_size, ctx->hbd_md ? EB_TEN_BIT : EB_EIGHT_BIT, pf_shape, hip_mask);\n"
           "    }\n"
           "    for (int tx_type_group_idx = 0; tx_type_group_idx < tx_type_tot_group; ++tx_type_group_idx) {\n"
           "        uint32_t best_tx_non_coeff = 64 * 64;\n")
    t = edit(t, old, new)
    old = ("                // Y: T Q i_q\n"
           "                svt_aom_estimate_transform(&(((int16_t *)cand_bf->residual->buffer_y)[txb_origin_index]),\n")
    new = ("                // Y: T Q i_q\n"
           "                if (svt_hip_bind_txt_take(hip_txt, tx_type, &(((int32_t *)ctx->tx_coeffs->buffer_y)[ctx->txb_1d_offset]), &ctx->three_quad_energy))\n"
           "                svt_aom_estimate_transform(&(((int16_t *)cand_bf->residual->buffer_y)[txb_origin_index]),\n")
    assert t.count(old) >= 1
    i = t.index(old, t.index("static void tx_type_search("))
    return t[:i] + new + t[i + len(old):]


def main():
    pieces = []
    for rel, fn in (("Source/Lib/Globals/enc_settings.c", patch_enc_settings),
                    ("Source/Lib/Globals/enc_handle.c", patch_enc_handle),
                    ("Source/Lib/Codec/me_process.c", patch_me_process),
                    ("Source/Lib/Codec/temporal_filtering.c", patch_temporal_filtering),
                    ("Source/Lib/Codec/src_ops_process.c", patch_src_ops_process),
                    ("Source/Lib/Codec/pic_analysis_process.c", patch_pic_analysis_process),
                    ("Source/Lib/Codec/product_coding_loop.c", patch_product_coding_loop),
                    ("Source/Lib/Codec/deblocking_filter.c", patch_deblocking_filter),
                    ("Source/Lib/Codec/coding_loop.c", patch_coding_loop),
                    ("Source/Lib/Codec/dlf_process.c", patch_dlf_process),
                    ("Source/Lib/Codec/cdef_process.c", patch_cdef_process),
                    ("Source/Lib/Codec/enc_cdef.c", patch_enc_cdef),
                    ("Source/Lib/Codec/restoration_pick.c", patch_restoration_pick),
                    ("Source/Lib/Codec/restoration.c", patch_restoration)):
        with open(os.path.join(REF, rel), encoding="utf-8", errors="surrogateescape") as f:
            a = f.read()
        b = fn(a)
        d = "".join(difflib.unified_diff(a.splitlines(True), b.splitlines(True), "a/" + rel, "b/" + rel, n=3))
        with open(os.path.join(ROOT, "tools", "e2e", os.path.basename(rel) + ".patch"), "w", encoding="utf-8",
                  errors="surrogateescape") as f:
            f.write(d)
        pieces.append(d)
    head = ("# tools/reference_hip.patch — `--asm hip` for the reference encoder (SVT-AV1 v2.2.0, Patman86 fork).\n"
            "# Apply with `patch -p1` at the root of a checkout of the reference, add tools/e2e/svt_hip_bind.c of the svtav1-hip\n"
            "# repository to the library's sources and `include/` + `tools/e2e/` to its include path, link with -ldl.\n"
            "# Generated by tools/e2e/make_patch.py; `make -C oracle e2e` compiles the patched files from a pipe.\n")
    with open(os.path.join(ROOT, "tools", "reference_hip.patch"), "w", encoding="utf-8", errors="surrogateescape") as f:
        f.write(head + "".join(pieces))
    print("wrote tools/reference_hip.patch", sum(p.count("\n") for p in pieces), "lines")


if __name__ == "__main__":
    sys.exit(main())

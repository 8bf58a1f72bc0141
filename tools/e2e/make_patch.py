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
    """Step 6b: the whole-picture temporal filter in front of the block loop of produce_temporally_filtered_pic (the first of the
    two identical loop heads: the second belongs to the low-delay variant)."""
    t = edit(t, '#include "temporal_filtering.h"\n', '#include "temporal_filtering.h"\n#include "svt_hip_bind.h"\n')
    old = "    for (uint32_t blk_row = y_b64_start_idx; blk_row < y_b64_end_idx; blk_row++) {\n"
    assert t.count(old) == 2
    i = t.index(old)
    return t[:i] + "    if (svt_hip_bind_tf_picture(pcs_list, list_input_picture_ptr, index_center, ctx, is_highbd))\n" + t[i:]


def patch_src_ops_process(t):
    """Step 3c: the whole-picture TPL dispenser in front of the per-block call of the segment loop (no-tiles path)."""
    t = edit(t, '#include "src_ops_process.h"\n', '#include "src_ops_process.h"\n#include "svt_hip_bind.h"\n')
    old = ("                        tpl_mc_flow_dispenser_sb_generic(pcs->scs->enc_ctx,\n"
           "                                                         scs,\n")
    new = ("                        if (svt_hip_bind_tpl_sb(pcs, frame_idx, context_ptr->sb_index, in_results_ptr->qIndex))\n" + old)
    return edit(t, old, new)


def main():
    pieces = []
    for rel, fn in (("Source/Lib/Globals/enc_settings.c", patch_enc_settings),
                    ("Source/Lib/Globals/enc_handle.c", patch_enc_handle),
                    ("Source/Lib/Codec/me_process.c", patch_me_process),
                    ("Source/Lib/Codec/temporal_filtering.c", patch_temporal_filtering),
                    ("Source/Lib/Codec/src_ops_process.c", patch_src_ops_process)):
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

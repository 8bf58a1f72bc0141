"""End-to-end bitstream gate (.gitlab/workflows/linux/.gitlab-ci.yml:338-366 is the reference's own form of it):
the reference encoder, built from its own sources with tools/reference_hip.patch (`make -C oracle e2e`), must write the
SAME .ivf and reconstruction with `--asm hip` (C table + every Tier A HIP leaf installed by svt_hip_install_rtcd) as with
`--asm c`.  The expected md5s (tests/golden/e2e_md5.json) were written by the C path in the build container.

Skipped where the encoder binary is absent (it can only be built where /root/reference exists; the built binary travels
with the repository snapshot to the GPU box)."""
import os
import re
import tempfile

import pytest

import e2e_cases as E

needs_app = pytest.mark.skipif(not E.have_app(), reason="oracle/_ref/e2e/SvtAv1EncApp not built (make -C oracle ref e2e)")


@needs_app
@pytest.mark.parametrize("case", ["p12_8bit", "p8_10bit"])
def test_c_path_reproduces_golden(case):
    """CPU: pins the committed md5s to the reference's C path (and the clip generator to its recipe)."""
    with tempfile.TemporaryDirectory() as d:
        md5, _ = E.encode(case, d, "c")
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


@needs_app
def test_asm_hip_without_device_keeps_cpu_kernels():
    """CPU: `--asm hip` on a machine with no gfx950 device installs nothing, warns, and encodes with the C kernels
    (never abort).  Only meaningful where no GPU is visible."""
    from svtav1_hip import abi
    if abi.load().svt_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode("p12_8bit", d, "hip")
    assert "HIP hot path unavailable" in log
    g = E.golden()["p12_8bit"]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case", list(E.CASES))
def test_asm_hip_bitstream_md5(hip, case):
    """GPU: every RTCD pointer that has a HIP leaf is swapped, the encoder runs, the bitstream is unchanged."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip")
    E.assert_hip_ran_clean(log)
    m = re.search(r"HIP hot path: (\d+) of (\d+) RTCD pointers", log)
    assert m, "the HIP leaves were not installed:\n" + log[-2000:]
    assert int(m.group(1)) == int(m.group(2)) >= 170
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the HIP leaves installed\n{log[-1500:]}"


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p12_8bit", 1), ("p8_8bit", 4), ("p8_10bit", 2)])
def test_batched_me_bitstream_md5(hip, case, lp):
    """GPU, Tier B inside the real encoder: with SVTAV1_HIP_TIERB_ME=1 the b64 loop of me_process.c hands
    every picture's open-loop ME to ONE svt_hip_me_frames call (tools/e2e/svt_hip_bind_me.c) instead of calling
    svt_aom_motion_estimation_b64 per block; several ME threads (--lp) share the picture's results.  Same bitstream."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra={"SVTAV1_HIP_TIERB_ME": "1"})
    E.assert_hip_ran_clean(log)
    m = re.search(r"svt_hip_bind_me: (\d+) pictures / (\d+) blocks", log)
    assert m, "the batched ME path did not run:\n" + log[-2000:]
    w, h, n, bd, preset = E.CASES[case]
    assert int(m.group(1)) >= n - 2 and int(m.group(2)) == int(m.group(1)) * ((w + 63) // 64) * ((h + 63) // 64)
    assert "falls back to the CPU search" not in log
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched ME\n{log[-1500:]}"


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p8_8bit", 1), ("p8_8bit", 4), ("p8_10bit", 2), ("p12_8bit", 2)])
def test_batched_tf_bitstream_md5(hip, case, lp):
    """GPU, Tier B inside the real encoder: with SVTAV1_HIP_TIERB_TF=1 produce_temporally_filtered_pic hands the
    whole picture to ONE svt_hip_tf_filter_picture call (tools/e2e/svt_hip_bind_tf.c) instead of running its block loop; together
    with the batched open-loop ME.  Same bitstream."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra={"SVTAV1_HIP_TIERB_TF": "1", "SVTAV1_HIP_TIERB_ME": "1"})
    E.assert_hip_ran_clean(log)
    m = re.search(r"svt_hip_bind_tf: (\d+) pictures", log)
    assert m and int(m.group(1)) >= 1, "the batched temporal filter did not run:\n" + log[-2000:]
    assert "stays on the CPU" not in log
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched temporal filter\n{log[-1500:]}"


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p5_8bit_lf", 3), ("p6_10bit_lf", 4)])
def test_batched_tpl_level3_bitstream_md5(hip, case, lp):
    """GPU: presets M5 / M6 (BASELINE configs[3]'s preset) run tpl level 3 — every candidate vector refined to a quarter sample by
    tpl_subpel_search, fractional vectors compensated with the 8-tap kernels — through svt_hip_tpl_dispenser_frame (quarter_pel = 1),
    with the batched picture analysis, ME and temporal filter.  Same bitstream."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra={"SVTAV1_HIP_TIERB_TPL": "1", "SVTAV1_HIP_TIERB_TF": "1", "SVTAV1_HIP_TIERB_ME": "1",
                                                               "SVTAV1_HIP_TIERB_PA": "1", "SVTAV1_HIP_ONLY": "__none__"})
    E.assert_hip_ran_clean(log)
    m = re.search(r"svt_hip_bind_tpl: (\d+) pictures", log)
    assert m and int(m.group(1)) >= 1, "the batched TPL dispenser did not run:\n" + log[-2000:]
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched TPL dispenser (level 3)\n{log[-1500:]}"


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p8_8bit_ld", 4), ("p8_10bit_ld", 4)])
def test_batched_tf_low_delay_bitstream_md5(hip, case, lp):
    """GPU: the low-delay prediction structure (`--pred-struct 1`) runs produce_temporally_filtered_pic_ld — co-located predictions, no
    motion search — which the patch hands to svt_hip_tf_filter_picture with ctrls.low_delay = 1 (tools/e2e/svt_hip_bind_tf.c); every
    other batched open-loop stage on as well.  Same bitstream."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra={"SVTAV1_HIP_TIERB_TF": "1", "SVTAV1_HIP_TIERB_ME": "1", "SVTAV1_HIP_TIERB_PA": "1",
                                                               "SVTAV1_HIP_TIERB_TPL": "1", "SVTAV1_HIP_ONLY": "__none__"})
    E.assert_hip_ran_clean(log)
    m = re.search(r"svt_hip_bind_tf: (\d+) pictures through svt_hip_tf_filter_picture \((\d+) of them the low-delay variant\)", log)
    assert m and int(m.group(2)) >= 1, "the low-delay temporal filter did not run on the GPU:\n" + log[-2000:]
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched low-delay temporal filter\n{log[-1500:]}"


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p8_8bit", 1), ("p8_8bit", 4), ("p8_10bit", 2), ("p12_8bit", 2)])
def test_batched_tpl_bitstream_md5(hip, case, lp):
    """GPU, Tier B inside the real encoder: with SVTAV1_HIP_TIERB_TPL=1 svt_aom_tpl_disp_kernel hands every
    picture's TPL dispenser to ONE svt_hip_tpl_dispenser_frame call (tools/e2e/svt_hip_bind_tpl.c); here together with the batched
    ME and the batched temporal filter: all three whole-picture entry points inside the running encoder.  Same bitstream."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra={"SVTAV1_HIP_TIERB_TPL": "1", "SVTAV1_HIP_TIERB_TF": "1", "SVTAV1_HIP_TIERB_ME": "1"})
    E.assert_hip_ran_clean(log)
    m = re.search(r"svt_hip_bind_tpl: (\d+) pictures", log)
    assert m, log[-2000:]
    # presets M7 ... M9 run tpl level 4 (16x16 blocks), M10 and faster level 5 (32x32 blocks, sub-sampled transform): both covered
    assert int(m.group(1)) >= 1, "the batched TPL dispenser did not run:\n" + log[-2000:]
    assert "stays on the CPU" not in log
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched TPL dispenser\n{log[-1500:]}"


@needs_app
@pytest.mark.parametrize("case", list(E.TIER_B_CASES))
def test_c_path_reproduces_golden_ragged(case):
    """CPU: the larger ragged clips' md5s are the reference's C path."""
    with tempfile.TemporaryDirectory() as d:
        md5, _ = E.encode(case, d, "c", lp=4)
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p8_8bit_ragged", 4), ("p10_10bit_ragged", 3)])
def test_batched_paths_ragged_clip(hip, case, lp):
    """GPU: ME, temporal filter and TPL dispenser through their whole-picture entry points (no Tier A leaves) on clips whose sizes
    are multiples of neither 64 nor 16, 10 - 17 frames, several threads.  Same bitstream as the C-only encode."""
    env = {"SVTAV1_HIP_TIERB_TPL": "1", "SVTAV1_HIP_TIERB_TF": "1", "SVTAV1_HIP_TIERB_ME": "1", "SVTAV1_HIP_ONLY": "__none__"}
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra=env)
    E.assert_hip_ran_clean(log)
    for what in ("me", "tf", "tpl"):
        m = re.search(rf"svt_hip_bind_{what}: (\d+) pictures", log)
        assert m and int(m.group(1)) >= 1, f"{what}: the batched path did not run\n" + log[-1500:]
    assert "stays on the CPU" not in log and "falls back" not in log
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs\n{log[-1500:]}"


@needs_app
@pytest.mark.parametrize("case", list(E.TF8_CASES))
def test_c_path_reproduces_golden_tf_8x8(case):
    """CPU: the goldens of the preset-2 clips (tf level 1: the temporal filter predicts with 8x8 blocks)."""
    with tempfile.TemporaryDirectory() as d:
        md5, _ = E.encode(case, d, "c", lp=4)
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p2_8bit_tf8", 4), ("p2_10bit_tf8", 3)])
def test_batched_tf_8x8_bitstream_md5(hip, case, lp):
    """GPU: preset 2 = tf level 1 = enable_8x8_pred: tf_8x8_sub_pel_search, the 16x16 -> 8x8 split decisions and the 8x8 luma / 4x4 chroma
    predictions inside svt_hip_tf_filter_picture, with the batched picture analysis and ME.  Same bitstream."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra={"SVTAV1_HIP_TIERB_TF": "1", "SVTAV1_HIP_TIERB_ME": "1", "SVTAV1_HIP_TIERB_PA": "1",
                                                               "SVTAV1_HIP_ONLY": "__none__"})
    E.assert_hip_ran_clean(log)
    m = re.search(r"svt_hip_bind_tf: (\d+) pictures", log)
    assert m and int(m.group(1)) >= 1, "the batched temporal filter did not run:\n" + log[-2000:]
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched temporal filter (8x8 prediction)\n{log[-1500:]}"


@needs_app
@pytest.mark.parametrize("case", list(E.LD_CASES))
def test_c_path_reproduces_golden_low_delay(case):
    """CPU: the low-delay goldens (`--pred-struct 1`, 720p: the smallest size at which the reference filters in that mode), and the
    clips do reach produce_temporally_filtered_pic_ld."""
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "c", lp=4, env_extra={"SVTAV1_E2E_TRACE_TF_LD": "1"})
    assert "produce_temporally_filtered_pic_ld reached" in log
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


@needs_app
@pytest.mark.parametrize("case", list(E.LF_CASES))
def test_c_path_reproduces_golden_lf(case):
    """CPU: the md5s of the in-loop-filter clips are the reference's C path."""
    with tempfile.TemporaryDirectory() as d:
        md5, _ = E.encode(case, d, "c", lp=2)
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


@needs_app
@pytest.mark.parametrize("case", ["p8_8bit", "p8_10bit", "p6_10bit_lf", "p10_10bit_ragged"])
def test_deferred_sb_deblocking_equals_the_sb_schedule(case):
    """CPU: the patch moves SB-based deblocking (presets M6 and faster, coding_loop.c:2260-2281) out of the EncDec loop into ONE
    svt_av1_loop_filter_frame call in dlf_process.c whenever the GPU hook is active.  SVTAV1_HIP_DLF_DEFER_TEST=1 makes that move
    with the reference's own C loops doing the frame call: the bitstream must not change (several EncDec threads)."""
    with tempfile.TemporaryDirectory() as d:
        md5, _ = E.encode(case, d, "c", lp=4, env_extra={"SVTAV1_HIP_DLF_DEFER_TEST": "1"})
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


LF_ENV = {"SVTAV1_HIP_TIERB_PA": "1", "SVTAV1_HIP_TIERB_DLF": "1", "SVTAV1_HIP_TIERB_CDEF": "1", "SVTAV1_HIP_TIERB_LR": "1"}


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp,wiener", [("p12_8bit", 2, False), ("p8_8bit", 4, True), ("p8_10bit", 2, True), ("p5_8bit_lf", 3, True),
                                            ("p6_10bit_lf", 4, True), ("p3_8bit_lf", 2, None), ("p8_8bit_ragged", 4, True)])
def test_batched_in_loop_filters_bitstream_md5(hip, case, lp, wiener):
    """GPU, Tier B inside the real encoder (INTEGRATION.md steps 2a and 4, row h of the coverage table): pyramid + block variances
    of the picture-analysis kernel, frame deblocking (the level search's trials included; SB-based deblocking deferred to one frame
    call), CDEF search and application, Wiener statistics and the final restoration pass through their whole-picture entry points
    (tools/e2e/svt_hip_bind_pa.c, svt_hip_bind_lf.c), no Tier A leaves.  Same bitstream and reconstruction as the C-only encode."""
    env = dict(LF_ENV, SVTAV1_HIP_ONLY="__none__")
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra=env)
    E.assert_hip_ran_clean(log)
    assert "stays on the CPU" not in log, log[-2000:]
    m = re.search(r"svt_hip_bind_lf: (\d+) frame deblocking calls, (\d+) CDEF searches, (\d+) CDEF applications, (\d+) Wiener statistics planes, "
                  r"(\d+) restoration", log)
    assert m, log[-2000:]
    dlf, cdef_s, cdef_a, wn, lr = map(int, m.groups())
    assert dlf >= 1 and cdef_s >= 1, (case, m.groups())
    if wiener:
        assert wn >= 1 and lr >= 1, (case, m.groups())
    if case in ("p5_8bit_lf", "p3_8bit_lf", "p6_10bit_lf"):
        # presets <= M5 (and base pictures at M6) search the deblocking level: every trial is filtered AND measured on the device
        # (svt_hip_bind_dlf_try: svt_hip_loop_filter_frame on a scratch copy + svt_hip_plane_sse against the source mirror)
        t = re.search(r"(\d+) trials of the deblocking level search filtered and measured on the device", log)
        assert t and int(t.group(1)) >= 3, (case, log[-1500:])
    m = re.search(r"svt_hip_bind_pa: (\d+) pyramids, (\d+) variance maps", log)
    w, h, n, bd, preset = E.ALL_CASES[case]
    # every picture once from the picture-analysis kernel, temporally filtered pictures again when they are re-decimated
    # (temporal_filtering.c pads and decimates the filtered picture); the variance map only where the preset computes it
    assert m and int(m.group(1)) >= n and int(m.group(2)) in (0, n), log[-2000:]
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched in-loop filters {m.groups()}\n{log[-1500:]}"


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p8_10bit", 3), ("p6_10bit_lf", 4), ("p10_10bit_ragged", 3)])
def test_every_batched_path_together(hip, case, lp):
    """GPU: picture analysis, open-loop ME, temporal filter, TPL dispenser, the transform-type search's batches, deblocking, CDEF,
    restoration — every batched entry point that is wired, together, sharing the device-resident picture mirrors (each hit compared with
    the host buffer)."""
    env = dict(LF_ENV, SVTAV1_HIP_TIERB_ME="1", SVTAV1_HIP_TIERB_TF="1", SVTAV1_HIP_TIERB_TPL="1", SVTAV1_HIP_TIERB_TXT="1", SVTAV1_HIP_ONLY="__none__")
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra=env)
    E.assert_hip_ran_clean(log)
    assert "stays on the CPU" not in log and "falls back" not in log, log[-2000:]
    m = re.search(r"mirrors: (\d+) hits", log)
    assert m and int(m.group(1)) > 0, log[-1500:]
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs\n{log[-1500:]}"


@needs_app
@pytest.mark.parametrize("case", ["p8_8bit", "p8_10bit", "p5_8bit_lf"])
def test_x86_intrinsics_table_reproduces_golden(case):
    """CPU: bench.py's encoder-level CPU baseline is the patched encoder with the reference's own x86 C-intrinsics ladder installed over
    the C table (SVTAV1_E2E_SIMD=1, tools/e2e/svt_hip_bind_simd.c + oracle/build_simd.py; the build container has no NASM for the real
    `--asm avx2` build).  The reference pins SIMD == C, so the bitstream must be the C one — which also catches a pointer the table
    leaves NULL (round 3: svt_cdef_filter_block_8xn_16 is assigned outside the SET_ ladders, common_dsp_rtcd.c:800)."""
    if " avx2 " not in open("/proc/cpuinfo").read().replace("\n", " "):
        pytest.skip("host CPU without AVX2")
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=3, env_extra={"SVTAV1_E2E_SIMD": "1"})
    assert "x86 intrinsics kernels" in log
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}


@needs_app
@pytest.mark.gpu
@pytest.mark.parametrize("case,lp", [("p12_8bit", 1), ("p8_8bit", 2), ("p8_10bit", 2), ("p5_8bit_lf", 3), ("p3_8bit_lf", 2)])
def test_batched_txt_bitstream_md5(hip, case, lp):
    """GPU, the transform batches inside the real encoder (the last stage of row h): tx_type_search hands the
    forward transforms of every transform type it can reach for a transform block to ONE svt_hip_txfm_quant_batch call
    (tools/e2e/svt_hip_bind_txt.c) and takes the coefficients from it instead of calling svt_aom_estimate_transform per type.
    Presets 3 and 5 search up to 16 types per block, 8 and 12 mostly DCT_DCT alone.  Same bitstream."""
    env = {"SVTAV1_HIP_TIERB_TXT": "1", "SVTAV1_HIP_ONLY": "__none__"}
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=lp, env_extra=env)
    E.assert_hip_ran_clean(log)
    assert "stays on the CPU" not in log, log[-1500:]
    m = re.search(r"svt_hip_bind_txt: (\d+) transform blocks / (\d+) forward transforms through svt_hip_txfm_quant_batch, (\d+) of them used", log)
    assert m and int(m.group(1)) > 0 and int(m.group(3)) > 0, log[-1500:]
    if case in ("p5_8bit_lf", "p3_8bit_lf"):
        assert int(m.group(2)) > int(m.group(1)), "no block searched more than one transform type"
    g = E.golden()[case]
    assert md5 == {"ivf": g["ivf"], "recon": g["recon"]}, f"{case}: bitstream differs with the batched transform-type search {m.groups()}\n{log[-1500:]}"

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

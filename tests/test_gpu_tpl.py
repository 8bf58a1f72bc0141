"""GPU parity: svt_hip_tpl_dispenser_frame (one launch, intra blocks ordered behind their neighbours by flags) against the
oracle's restatement of tpl_mc_flow_dispenser_sb_generic and the golden results of the reference, bit-exact: the TPL
reconstruction picture, TplStats and TplSrcStats."""
import ctypes as C
import os

import numpy as np
import pytest

import tpl_cases as T
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tpl_frame.npz")


@pytest.fixture(scope="module", autouse=True)
def quant():
    T.load_quant(np.load(GOLD))


class DevMap:
    """Uploads host arrays on first use and hands out their device addresses (keyed by the host address)."""

    def __init__(self, hip):
        self.hip, self.m = hip, {}

    def __call__(self, arr):
        key = arr.ctypes.data
        if key not in self.m:
            b = device.DeviceBuffer(self.hip, arr.nbytes)
            b.upload(arr)
            self.m[key] = (b, arr)
        return self.m[key][0].ptr

    def download(self, arr):
        b, a = self.m[arr.ctypes.data]
        return b.download(a.dtype, a.shape)


def run_gpu(hip, scene):
    name, kind, w, h, qindex, opt = scene.case
    dm = DevMap(hip)
    job = scene.job(dm)
    hip.svt_hip_tpl_workspace_bytes.restype = C.c_uint64
    hip.svt_hip_tpl_status_offset.restype = C.c_uint64
    wsb = hip.svt_hip_tpl_workspace_bytes(w, h)
    ws = device.DeviceBuffer(hip, wsb)
    ws.fill(0xCD)
    job.workspace, job.workspace_bytes = ws.ptr, wsb
    device.check(hip, hip.svt_hip_tpl_dispenser_frame(C.byref(job), None), "svt_hip_tpl_dispenser_frame")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    off = hip.svt_hip_tpl_status_offset(w, h)
    status = ws.download(np.uint8, (wsb,))[off:off + 4].view(np.uint32)[0]
    assert status == 0, "a dependency wait ran into its bound"
    return {"recon": dm.download(scene.out.buf), "stats": dm.download(scene.stats).view(np.uint8),
            "src_stats": dm.download(scene.src_stats).view(np.uint8)}


@pytest.mark.parametrize("case", T.CASES, ids=lambda c: c[0])
def test_dispenser_frame(hip, orc, case):
    gold = np.load(GOLD)
    a, b = T.TplScene(orc, case), T.TplScene(orc, case)
    if case[5]["src_data_ready"]:
        T.prime_second_pass(orc, a), T.prime_second_pass(orc, b)
    got = run_gpu(hip, a)
    assert orc.orc_tpl_dispenser_frame(C.byref(b.job())) == 0
    for k, v in b.results().items():
        assert np.array_equal(got[k], v), (case[0], k, int((got[k] != v).sum()))
        assert np.array_equal(got[k], gold[f"{case[0]}_{k}"]), (case[0], k, "golden")


def test_all_intra_wavefront_1080p(hip, orc):
    """Every block intra (I slice): the longest dependency chains the flags have to carry, on a 1920x1080 picture."""
    case = ("islice_1080p", "pan", 1920, 1080, 120, dict(T.BASE, i_slice=1, tpl_i_slice=1))
    a, b = T.TplScene(orc, case), T.TplScene(orc, case)
    got = run_gpu(hip, a)
    assert orc.orc_tpl_dispenser_frame(C.byref(b.job())) == 0
    for k, v in b.results().items():
        assert np.array_equal(got[k], v), (k, int((got[k] != v).sum()))


def test_dispenser_frame_4k(hip, orc):
    """BASELINE size: a 3840x2160 picture with 2+2 references (preset-8 tpl level), GPU == oracle."""
    case = ("pan_4k", "pan", 3840, 2160, 120, dict(T.BASE))
    a, b = T.TplScene(orc, case, key="m8_4k_tl2"), T.TplScene(orc, case, key="m8_4k_tl2")
    got = run_gpu(hip, a)
    assert orc.orc_tpl_dispenser_frame(C.byref(b.job())) == 0
    for k, v in b.results().items():
        assert np.array_equal(got[k], v), (k, int((got[k] != v).sum()))
    modes = b.src_stats["best_mode"]
    assert (modes == 16).sum() > 1000


def test_level5_all_intra_and_inter_1080p(hip, orc):
    """tpl level 5 (32x32 blocks, TX_32X8 on every 4th row, 32x32 synthesizer grid as the reference picks above 720p) at 1920x1080:
    an I slice (every block waits for its neighbours) and a B picture."""
    for name, opt in (("l5_islice_1080p", dict(T.BASE, i_slice=1, tpl_i_slice=1, synth_blk_size=32, **T.L5)),
                      ("l5_pan_1080p", dict(T.BASE, synth_blk_size=32, **T.L5))):
        case = (name, "pan", 1920, 1080, 120, opt)
        a, b = T.TplScene(orc, case, key="m8_1080p_tl2"), T.TplScene(orc, case, key="m8_1080p_tl2")
        assert orc.orc_tpl_dispenser_frame(C.byref(a.job())) == 0
        got = run_gpu(hip, b)
        for k, v in a.results().items():
            assert np.array_equal(got[k], v), (name, k, int((got[k] != v).sum()))
        if not opt["i_slice"]:
            assert (a.src_stats["best_mode"] == 16).any() and (a.src_stats["best_mode"] == 0).any()


def test_fence_publish_path(hip, orc):
    """A reconstruction plane whose rows are not 4-byte aligned cannot be published by write-through dword stores: the kernel then
    falls back to one agent-scope release fence per block.  job.publish_fence forces that path on an aligned plane."""
    for case in (T.CASES[3], [c for c in T.CASES if c[0] == "l5_pan_half_column"][0]):
        case = case[:5] + (dict(case[5], publish_fence=1),)
        a, b = T.TplScene(orc, case), T.TplScene(orc, case)
        got = run_gpu(hip, a)
        assert orc.orc_tpl_dispenser_frame(C.byref(b.job())) == 0
        for k, v in b.results().items():
            assert np.array_equal(got[k], v), (case[0], k)


@pytest.mark.parametrize("level,fence", [(4, 0), (5, 0), (4, 1), (3, 0)])
def test_stale_line_pattern_4k(hip, orc, level, fence):
    """The hand-over the write-through publication has to get right: block columns ALTERNATE between inter blocks (static
    stripes: they start at once and read nothing of their neighbours) and intra blocks (fresh noise per picture: they wait for the
    left / top / top-left neighbours and read their reconstructed border samples).  An inter block to the right of an intra block
    shares 128-byte row lines with it and is reconstructed long before it, so the line is in some XCD's L2 in its old state when the
    intra block next to it -- on another XCD -- publishes; the intra block one further right must still see the published samples.
    3840x2160, tpl level 4 (16x16 blocks, 16-sample stripes), level 3 (the same with the quarter-pel refinement) and level 5 (32x32 blocks,
    32-sample stripes), both publication modes."""
    bs = 32 if level == 5 else 16
    opt = dict(T.BASE, publish_fence=fence, **(T.L5 if level == 5 else (T.L3 if level == 3 else {})))
    if level == 5:
        opt["synth_blk_size"] = 32
    case = (f"stripes{bs}_4k", f"stripes{bs}", 3840, 2160, 120, opt)
    a, b = T.TplScene(orc, case, key="m8_4k_tl2"), T.TplScene(orc, case, key="m8_4k_tl2")
    assert orc.orc_tpl_dispenser_frame(C.byref(b.job())) == 0
    step = bs // 16                                    # a 32x32 block stores its source statistics in its top-left 16x16 cell
    modes = b.src_stats["best_mode"].reshape(-1, 3840 // 16)[::step, ::step]
    cols = np.arange(3840 // bs) & 1
    # the pattern is what the test is about: noisy columns intra (DC_PRED = 0), static columns inter (NEWMV = 16)
    # (with the quarter-pel refinement of level 3 a seventh of the noisy blocks find a smoothed inter prediction cheaper than DC)
    assert (modes[:, cols == 1] == 0).mean() > (0.8 if level == 3 else 0.95) and (modes[:, cols == 0] == 16).mean() > 0.95
    got = run_gpu(hip, a)
    for k, v in b.results().items():
        assert np.array_equal(got[k], v), (case[0], k, int((got[k] != v).sum()))


def test_argument_checks(hip, orc):
    s = T.TplScene(orc, T.CASES[0])
    dm = DevMap(hip)
    job = s.job(dm)
    assert hip.svt_hip_tpl_dispenser_frame(C.byref(job), None) == abi.SVT_HIP_ERR_BAD_PARAMETER  # no workspace
    assert b"workspace" in hip.svt_hip_last_error()
    assert hip.svt_hip_tpl_dispenser_frame(None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    hip.svt_hip_tpl_workspace_bytes.restype = C.c_uint64
    wsb = hip.svt_hip_tpl_workspace_bytes(s.case[2], s.case[3])
    ws = device.DeviceBuffer(hip, wsb)
    job.workspace, job.workspace_bytes, job.quarter_pel, job.blk_size, job.subsample_tx = ws.ptr, wsb, 1, 32, 2
    assert hip.svt_hip_tpl_dispenser_frame(C.byref(job), None) == abi.SVT_HIP_ERR_BAD_PARAMETER  # quarter-pel comes with 16x16 blocks only
    assert b"quarter_pel" in hip.svt_hip_last_error()

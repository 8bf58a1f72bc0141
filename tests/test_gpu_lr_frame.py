"""GPU parity: svt_hip_restoration_filter_frame (all planes of a picture in one launch) against the oracle's restatement of
svt_av1_loop_restoration_filter_frame and against the fixture the real reference functions produced."""
import ctypes as C
import os

import numpy as np
import pytest

import lr_cases as R
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_frame.npz")


def run_gpu(hip, case):
    keep, dsts, units_ptr = [], [], []
    ptrs = {}

    def up(a):
        b = device.DeviceBuffer(hip, a.nbytes)
        b.upload(a)
        keep.append(b)
        ptrs[id(a)] = b.ptr
        return b.ptr
    for c in case:
        for k in ("src", "above", "below", "units"):
            up(c[k])
        d = device.DeviceBuffer(hip, (c["h"] + 2) * (c["w"] + R.DST_EXTRA) * c["src"].itemsize)
        d.fill(0)
        dsts.append(d)
    arr, _ = R.lr_planes(case, ptr_of=lambda a: ptrs[id(a)], dsts=[d.ptr for d in dsts])
    device.check(hip, hip.svt_hip_restoration_filter_frame(arr, C.c_uint32(len(case)), None), "restoration_filter_frame")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    return [d.download(c["src"].dtype, (c["h"] + 2, c["w"] + R.DST_EXTRA))[:c["h"], :c["w"]] for d, c in zip(dsts, case)]


@pytest.mark.parametrize("name", list(R.CASES))
def test_frame_vs_oracle(hip, orc, name):
    case = R.make_case(name)
    arr, outs = R.lr_planes(case)
    orc.orc_restoration_filter_frame(arr, C.c_uint32(len(case)))
    got = run_gpu(hip, case)
    for p, (a, o, c) in enumerate(zip(got, outs, case)):
        want = o[:c["h"], :c["w"]]
        assert np.array_equal(a, want), (name, p, np.argwhere(a != want)[:5])


def test_frame_golden(hip):
    g = np.load(GOLD)
    for name in R.CASES:
        for p, a in enumerate(run_gpu(hip, R.make_case(name))):
            assert np.array_equal(a, g[f"{name}_p{p}"]), (name, p)


def test_frame_4k10(hip, orc):
    """BASELINE.json configs[3] size: 3840 x 2160 10-bit, all three planes, 256 / 128 restoration units, saved boundaries."""
    case = R.make_case(None, dims=(3840, 2160, 10, 1, 256, 0, 9))
    arr, outs = R.lr_planes(case)
    orc.orc_restoration_filter_frame(arr, C.c_uint32(len(case)))
    for p, (a, o, c) in enumerate(zip(run_gpu(hip, case), outs, case)):
        want = o[:c["h"], :c["w"]]
        assert np.array_equal(a, want), (p, np.argwhere(a != want)[:5])


def test_bad_arguments(hip):
    assert hip.svt_hip_restoration_filter_frame(None, C.c_uint32(1), None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    pl = (abi.LrPlane * 1)()
    assert hip.svt_hip_restoration_filter_frame(pl, C.c_uint32(1), None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_restoration_filter_frame(pl, C.c_uint32(4), None) == abi.SVT_HIP_ERR_BAD_PARAMETER

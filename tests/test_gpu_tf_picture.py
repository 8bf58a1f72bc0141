"""GPU parity: svt_hip_tf_filter_picture (ME_MCTF -> sub-pel refinement -> 64/32/16 decisions -> predictions -> accumulate ->
normalise, all device-resident) against the oracle's restatement of produce_temporally_filtered_pic and against the golden
pictures the reference itself produced, bit-exact: the filtered picture, the per-block refinement state and the counters."""
import ctypes as C
import os

import numpy as np
import pytest

import tf_picture_cases as tpc
from svtav1_hip import abi, device, frames

pytestmark = pytest.mark.gpu
V = C.c_void_p
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_picture.npz")


class DevWindow:
    """Device copies of every plane of a window + the pointer sets tf_picture_cases.make_job takes."""

    def __init__(self, hip, pics):
        self.hip, self.pics, self.ptrs, self.bufs = hip, pics, [], []
        for p in pics:
            pyr = device.DevicePyramid(hip, p.pyr)
            c8 = [device.DevicePlane(hip, c) for c in p.c8]
            hbd = []
            if p.hbd is not None:
                for a in p.hbd:
                    b = device.DeviceBuffer(hip, a.nbytes)
                    b.upload(a)
                    hbd.append(b)
            self.bufs.append((pyr, c8, hbd))
            self.ptrs.append({"pyr": pyr.desc(), "c8": [c.buf.ptr for c in c8], "hbd": [b.ptr for b in hbd]})

    def centre_arrays(self):
        pyr, c8, hbd = self.bufs[0]
        out = {"y8": pyr.full.download(), "cb8": c8[0].download(), "cr8": c8[1].download()}
        if hbd:
            for k, b, a in zip(("y16", "cb16", "cr16"), hbd, self.pics[0].hbd):
                out[k] = b.download(np.uint16, a.shape)
        return out


def run_gpu(hip, pics, case, decay):
    name, kind, w, h, n_refs, bd, key, ctl = case
    dev = DevWindow(hip, pics)
    job = tpc.make_job(pics, w, h, bd, key, ctl, decay, dev.ptrs)
    hip.svt_hip_tf_workspace_bytes.restype = C.c_uint64
    hip.svt_hip_tf_workspace_state_offset.restype = C.c_uint64
    wsb = hip.svt_hip_tf_workspace_bytes(w, h, n_refs)
    ws = device.DeviceBuffer(hip, wsb)
    tot = device.DeviceBuffer(hip, 8)
    tot.fill(0)
    job.workspace, job.workspace_bytes, job.tot_blks = ws.ptr, wsb, tot.ptr
    device.check(hip, hip.svt_hip_tf_filter_picture(C.byref(job), None), "svt_hip_tf_filter_picture")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    nb = frames.b64_count(w, h)
    raw = ws.download(np.uint8, (wsb,))
    states = []
    for r in range(n_refs):
        off = hip.svt_hip_tf_workspace_state_offset(w, h, n_refs, r)
        states.append(raw[off:off + nb * C.sizeof(abi.TfB64State)].reshape(nb, -1))
    return dev.centre_arrays(), np.concatenate(states), tuple(int(x) for x in tot.download(np.uint32, (2,)))


@pytest.mark.parametrize("case", tpc.CASES, ids=lambda c: c[0])
def test_filter_picture(hip, orc, case):
    gold = np.load(GOLD)
    name = case[0]
    decay = tuple(int(x) for x in gold[f"{name}_decay"])
    got, states, tot = run_gpu(hip, tpc.case_window(orc, case), case, decay)
    pics = tpc.case_window(orc, case)
    ostates, otot = tpc.run_oracle(orc, pics, case, decay)
    assert tot == otot == tuple(int(x) for x in gold[f"{name}_tot"])
    ost = tpc.states_to_array(ostates)
    bad = np.argwhere((ost != states).any(axis=1))
    assert len(bad) == 0, (name, "state of (ref, b64) entries", bad[:8].ravel().tolist())
    for k, v in pics[0].arrays().items():
        assert np.array_equal(got[k], v), (name, k, int((got[k] != v).sum()))
        assert np.array_equal(got[k], gold[f"{name}_{k}"]), (name, k, "golden")


def test_argument_checks(hip, orc):
    case = tpc.CASES[0]
    name, kind, w, h, n_refs, bd, key, ctl = case
    pics = tpc.case_window(orc, case)
    dev = DevWindow(hip, pics)
    job = tpc.make_job(pics, w, h, bd, key, ctl, (1, 1, 1), dev.ptrs)
    assert hip.svt_hip_tf_filter_picture(C.byref(job), None) == abi.SVT_HIP_ERR_BAD_PARAMETER  # no workspace
    hip.svt_hip_tf_workspace_bytes.restype = C.c_uint64
    wsb = hip.svt_hip_tf_workspace_bytes(w, h, n_refs)
    ws = device.DeviceBuffer(hip, wsb)
    job.workspace, job.workspace_bytes = ws.ptr, wsb
    job.ctrls.enable_8x8_pred = 2   # a flag: 0 or 1
    assert hip.svt_hip_tf_filter_picture(C.byref(job), None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert b"8x8" in hip.svt_hip_last_error()
    job.ctrls.enable_8x8_pred, job.n_refs = 0, 0
    assert hip.svt_hip_tf_filter_picture(C.byref(job), None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    job.n_refs, job.bit_depth = n_refs, 12
    assert hip.svt_hip_tf_filter_picture(C.byref(job), None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_tf_filter_picture(None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER


def test_filter_picture_4k(hip, orc):
    """BASELINE size: one 3840x2160 8-bit picture against one neighbour (tf level 6 controls, luma + chroma), GPU == oracle on
    every plane and every (b64) refinement state; and the size-independent property that a window of identical pictures
    leaves the centre picture unchanged (every prediction is the picture itself, (sum w * x + sum w / 2) / sum w == x)."""
    case = ("pan_4k", "pan", 3840, 2160, 1, 8, "m8_4k_tl2", tpc.LVL6)
    decay = (2247286, 6156426, 6156426)
    got, states, tot = run_gpu(hip, tpc.case_window(orc, case), case, decay)
    pics = tpc.case_window(orc, case)
    ostates, otot = tpc.run_oracle(orc, pics, case, decay)
    assert tot == otot
    assert np.array_equal(tpc.states_to_array(ostates), states)
    for k, v in pics[0].arrays().items():
        assert np.array_equal(got[k], v), (k, int((got[k] != v).sum()))
    # identical pictures
    same = tpc.case_window(orc, case)
    for k, v in same[1].arrays().items():
        v[...] = same[0].arrays()[k]
    d = same[1].pyr.desc()
    orc.orc_pyramid_frame(C.byref(d.full), C.byref(d.quarter), C.byref(d.sixteenth), 1)
    before = {k: v.copy() for k, v in same[0].arrays().items()}
    got, _, _ = run_gpu(hip, same, case, decay)
    for k, v in before.items():
        assert np.array_equal(got[k], v), k


@pytest.mark.parametrize("case", [
    ("blocks_8x8_4k", "blocks", 3840, 2160, 1, 8, "m8_4k_tl2", dict(tpc.LVL1, enable_8x8_pred=1)),
    ("fastpan_ld_4k_10bit", "fastpan", 3840, 2160, 2, 10, "m8_4k_tl2", dict(tpc.LVL8, low_delay=1, chroma=1)),
], ids=lambda c: c[0])
def test_filter_picture_variants_4k(hip, orc, case):
    """BASELINE size for the variants added in round 3: tf level 1 with 8x8 prediction (tens of thousands of 16x16 blocks split into 8x8) and
    the low-delay filter on a 10-bit picture with two neighbours — GPU == oracle on every plane and every refinement state."""
    decay = (2247286, 6156426, 6156426)
    got, states, tot = run_gpu(hip, tpc.case_window(orc, case), case, decay)
    pics = tpc.case_window(orc, case)
    ostates, otot = tpc.run_oracle(orc, pics, case, decay)
    assert tot == otot
    assert np.array_equal(tpc.states_to_array(ostates), states)
    if case[7].get("enable_8x8_pred"):
        assert sum(sum(s.split16) for s in ostates) > 1000     # the 8x8 path is what this case is about
    for k, v in pics[0].arrays().items():
        assert np.array_equal(got[k], v), (k, int((got[k] != v).sum()))

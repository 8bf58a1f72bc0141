"""GPU parity: self-guided restoration of libsvtav1_hip (through the C-ABI) against the oracle and the golden fixture."""
import ctypes as C
import os

import numpy as np
import pytest

import sgr_cases as G
from lf_cases import P, V
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu
I64 = C.c_int64


def enc(addr, is16):
    return V(addr >> 1) if is16 else V(addr)     # CONVERT_TO_BYTEPTR, as the reference's callers pass uint16 buffers


def prm(ep):
    r0, r1, s0, s1 = abi.SGR_PARAMS[ep]
    return abi.SgrParams((C.c_int32 * 2)(r0, r1), (C.c_int32 * 2)(s0, s1))


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (8, 1)])
def test_tier_a_filter_apply(hip, orc, bd, is16):
    rng = np.random.default_rng(bd + is16)
    for trial in range(32):
        ep = trial % 16
        w, h = ((64, 64), (64, 56), (32, 32), (40, 17), (8, 8), (61, 64), (136, 72))[trial % 7]
        dat, src = G.sgr_plane(rng, w, h, bd, is16, (0, 0, 1, 2)[trial % 4])
        fs = w + 3
        f0a, f1a = np.full((h, fs), 7777, np.int32), np.full((h, fs), 7777, np.int32)
        f0b, f1b = f0a.copy(), f1a.copy()
        orc.orc_sgr_filter_unit(V(G.at(dat)), w, h, dat.shape[1], is16, bd, 64, 64, ep, P(f0a), P(f1a), fs)
        hip.svt_av1_selfguided_restoration_hip(enc(G.at(dat), is16), w, h, dat.shape[1], P(f0b), P(f1b), fs, ep, bd, is16)
        r0, r1 = abi.SGR_PARAMS[ep][:2]
        assert (r0 == 0 or np.array_equal(f0a, f0b)) and (r1 == 0 or np.array_equal(f1a, f1b)), (trial, ep, w, h)
        if w <= 64 and h <= 64:
            xqd = np.array([int(rng.integers(-96, 32)), int(rng.integers(-32, 96))], np.int32)
            o1, o2 = np.zeros((h, w + 5), dat.dtype), np.zeros((h, w + 5), dat.dtype)
            orc.orc_apply_selfguided_restoration(V(G.at(dat)), w, h, dat.shape[1], ep, P(xqd), P(o1), w + 5, bd, is16)
            hip.svt_apply_selfguided_restoration_hip(enc(G.at(dat), is16), w, h, dat.shape[1], ep, P(xqd), enc(o2.ctypes.data, is16), w + 5,
                                                     None, bd, is16)
            assert np.array_equal(o1, o2), (trial, ep)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1)])
def test_tier_a_projection(hip, orc, bd, is16):
    rng = np.random.default_rng(40 + bd)
    orc.orc_sgr_pixel_proj_error.restype = I64
    fn = hip.svt_av1_highbd_pixel_proj_error_hip if is16 else hip.svt_av1_lowbd_pixel_proj_error_hip
    fn.restype = I64
    for trial in range(32):
        ep = trial % 16
        w, h = ((96, 80), (64, 64), (33, 47))[trial % 3]
        dat, src = G.sgr_plane(rng, w, h, bd, is16, (0, 0, 2, 1)[trial % 4])
        fs = ((w + 7) & ~7) + 8
        f0, f1 = np.zeros((h, fs), np.int32), np.zeros((h, fs), np.int32)
        orc.orc_sgr_filter_unit(V(G.at(dat)), w, h, dat.shape[1], is16, bd, 64, 64, ep, P(f0), P(f1), fs)
        xq = np.array([int(rng.integers(-100, 100)), int(rng.integers(-100, 100))], np.int32)
        p = prm(ep)
        a = orc.orc_sgr_pixel_proj_error(V(G.at(src)), w, h, src.shape[1], V(G.at(dat)), dat.shape[1], P(f0), fs, P(f1), fs, P(xq), ep, is16)
        b = fn(enc(G.at(src), is16), w, h, src.shape[1], enc(G.at(dat), is16), dat.shape[1], P(f0), fs, P(f1), fs, P(xq), C.byref(p))
        assert a == b, (trial, a, b)
        x1, x2 = np.zeros(2, np.int32), np.zeros(2, np.int32)
        orc.orc_get_proj_subspace(V(G.at(src)), w, h, src.shape[1], V(G.at(dat)), dat.shape[1], is16, P(f0), fs, P(f1), fs, P(x1), ep)
        hip.svt_get_proj_subspace_hip(enc(G.at(src), is16), w, h, src.shape[1], enc(G.at(dat), is16), dat.shape[1], is16, P(f0), fs, P(f1),
                                      fs, P(x2), C.byref(p))
        assert np.array_equal(x1, x2), (trial, x1, x2)


def gpu_unit(hip, dat, src, w, h, bd, is16, pu):
    d_dat, d_src = device.DeviceBuffer(hip, dat.nbytes), device.DeviceBuffer(hip, src.nbytes)
    d_dat.upload(dat), d_src.upload(src)
    off = (G.B * dat.shape[1] + G.B) * dat.itemsize
    return abi.SgrUnit(d_dat.ptr + off, d_src.ptr + off, dat.shape[1], src.shape[1], w, h, is16, bd, pu, pu), (d_dat, d_src)


@pytest.mark.parametrize("case", range(8))
def test_tier_b_search_filter_apply(hip, orc, case):
    rng = np.random.default_rng(200 + case)
    w, h = ((96, 80), (136, 72), (64, 64), (200, 120), (56, 40), (384, 96), (256, 256), (328, 200))[case]
    bd, is16 = ((8, 0), (10, 1), (8, 1))[case % 3]
    pu = 64 if case % 2 == 0 else 32
    start, end, inc, refine = ((0, 16, 1, 1), (0, 16, 2, 1), (10, 16, 1, 0), (0, 8, 3, 1), (14, 16, 1, 1), (0, 16, 4, 1), (0, 16, 1, 1),
                               (3, 4, 1, 1))[case]
    dat, src = G.sgr_plane(rng, w, h, bd, is16, (0, 0, 2)[case % 3])
    o1 = np.zeros(3, np.int32)
    orc.orc_sgr_search_unit.restype = I64
    e1 = orc.orc_sgr_search_unit(V(G.at(dat)), w, h, dat.shape[1], V(G.at(src)), src.shape[1], is16, bd, pu, pu, start, end, inc, refine, P(o1))
    unit, keep = gpu_unit(hip, dat, src, w, h, bd, is16, pu)
    hip.svt_hip_sgr_search_work_bytes.restype = C.c_size_t
    n_ep = (end - start + inc - 1) // inc
    work = device.DeviceBuffer(hip, hip.svt_hip_sgr_search_work_bytes(w, h, n_ep))
    o2, e2 = np.zeros(3, np.int32), I64(0)
    device.check(hip, hip.svt_hip_sgr_search_unit(C.byref(unit), start, end, inc, refine, V(work.ptr), P(o2), C.byref(e2), None), "sgr_search")
    assert np.array_equal(o1, o2) and e1 == e2.value, (o1, o2, e1, e2.value)
    # filter + fused apply of the winner over the whole unit
    ep = int(o1[0])
    fs = w + 5
    f0, f1 = np.zeros((h, fs), np.int32), np.zeros((h, fs), np.int32)
    orc.orc_sgr_filter_unit(V(G.at(dat)), w, h, dat.shape[1], is16, bd, pu, pu, ep, P(f0), P(f1), fs)
    d0, d1 = device.DeviceBuffer(hip, f0.nbytes), device.DeviceBuffer(hip, f1.nbytes)
    d0.fill(0), d1.fill(0)
    device.check(hip, hip.svt_hip_sgr_filter_unit(C.byref(unit), ep, V(d0.ptr), V(d1.ptr), fs, None), "sgr_filter")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    r0, r1 = abi.SGR_PARAMS[ep][:2]
    assert r0 == 0 or np.array_equal(d0.download(np.int32, f0.shape), f0)
    assert r1 == 0 or np.array_equal(d1.download(np.int32, f1.shape), f1)
    xqd = np.array([o1[1], o1[2]], np.int32)
    want = np.zeros((h, w), dat.dtype)
    for i in range(0, h, pu):
        for j in range(0, w, pu):
            ph, pw = min(pu, h - i), min(pu, w - j)
            orc.orc_apply_selfguided_restoration(V(G.at(dat) + (i * dat.shape[1] + j) * dat.itemsize), pw, ph, dat.shape[1], ep, P(xqd),
                                                 V(want.ctypes.data + (i * w + j) * dat.itemsize), w, bd, is16)
    d_out = device.DeviceBuffer(hip, want.nbytes)
    device.check(hip, hip.svt_hip_sgr_apply_unit(C.byref(unit), ep, P(xqd), V(d_out.ptr), w, None), "sgr_apply")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    assert np.array_equal(d_out.download(dat.dtype, want.shape), want)


def test_tier_b_golden(hip):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sgr.npz"))
    hip.svt_hip_sgr_search_work_bytes.restype = C.c_size_t
    for key, w, h, bd, is16, kind, pu, (s0, s1, inc, refine), seed in G.GOLDEN_SGR:
        dat, src, best = g[key + "_dat"].copy(), g[key + "_src"].copy(), g[key + "_best"]
        unit, keep = gpu_unit(hip, dat, src, w, h, bd, is16, pu)
        work = device.DeviceBuffer(hip, hip.svt_hip_sgr_search_work_bytes(w, h, (s1 - s0 + inc - 1) // inc))
        out = np.zeros(3, np.int32)
        device.check(hip, hip.svt_hip_sgr_search_unit(C.byref(unit), s0, s1, inc, refine, V(work.ptr), P(out), None, None), "sgr_search")
        assert np.array_equal(out, best), key
        pw, ph = min(pu, w), min(pu, h)
        f0, f1 = np.zeros((ph, pw), np.int32), np.zeros((ph, pw), np.int32)
        e = (lambda a: V(a >> 1)) if is16 else V
        hip.svt_av1_selfguided_restoration_hip(e(G.at(dat)), pw, ph, dat.shape[1], P(f0), P(f1), pw, int(best[0]), bd, is16)
        r0, r1 = abi.SGR_PARAMS[int(best[0])][:2]
        assert (r0 == 0 or np.array_equal(f0, g[key + "_flt0"])) and (r1 == 0 or np.array_equal(f1, g[key + "_flt1"])), key
        rec = np.zeros((ph, pw), dat.dtype)
        xqd = np.array([best[1], best[2]], np.int32)
        hip.svt_apply_selfguided_restoration_hip(e(G.at(dat)), pw, ph, dat.shape[1], int(best[0]), P(xqd), e(rec.ctypes.data), pw, None, bd, is16)
        assert np.array_equal(rec, g[key + "_rec"]), key


def test_tier_b_bad_arguments(hip):
    u = abi.SgrUnit()
    out = np.zeros(3, np.int32)
    assert hip.svt_hip_sgr_search_unit(C.byref(u), 0, 16, 1, 1, None, P(out), None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_sgr_filter_unit(C.byref(u), 0, None, None, 0, None) == abi.SVT_HIP_ERR_BAD_PARAMETER

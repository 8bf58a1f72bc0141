"""GPU parity at the size BASELINE.json configs[3] is quoted on — one 3840 x 2160 10-bit picture — for the restoration
stages (the deblocking and CDEF cases of that size live in test_gpu_lf.py as one more parameter of the frame tests):
self-guided filter + fused apply over the whole luma plane in one launch, Wiener statistics of every 256 x 256 restoration
unit and the Wiener filter over the whole plane.  Bit-exact against the oracle on every sample / every unit."""
import ctypes as C

import numpy as np
import pytest

import sgr_cases as G
from lf_cases import P, V
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu
W, H, BD = 3840, 2160, 10


@pytest.fixture(scope="module")
def plane4k():
    rng = np.random.default_rng(4000)
    return G.sgr_plane(rng, W, H, BD, 1, 0)


def test_sgr_filter_and_apply_4k10(hip, orc, plane4k):
    dat, src = plane4k
    d_dat, d_src = device.DeviceBuffer(hip, dat.nbytes), device.DeviceBuffer(hip, src.nbytes)
    d_dat.upload(dat), d_src.upload(src)
    off = (G.B * dat.shape[1] + G.B) * dat.itemsize
    unit = abi.SgrUnit(d_dat.ptr + off, d_src.ptr + off, dat.shape[1], src.shape[1], W, H, 1, BD, 64, 64)
    for ep in (3, 12):                       # both radii (r0 = 2, r1 = 1) / r1 only (abi.SGR_PARAMS)
        fs = W
        f0, f1 = np.zeros((H, fs), np.int32), np.zeros((H, fs), np.int32)
        orc.orc_sgr_filter_unit(V(G.at(dat)), W, H, dat.shape[1], 1, BD, 64, 64, ep, P(f0), P(f1), fs)
        d0, d1 = device.DeviceBuffer(hip, f0.nbytes), device.DeviceBuffer(hip, f1.nbytes)
        d0.fill(0), d1.fill(0)
        device.check(hip, hip.svt_hip_sgr_filter_unit(C.byref(unit), ep, V(d0.ptr), V(d1.ptr), fs, None), "sgr_filter")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
        r0, r1 = abi.SGR_PARAMS[ep][:2]
        if r0:
            assert np.array_equal(d0.download(np.int32, f0.shape), f0), ep
        if r1:
            assert np.array_equal(d1.download(np.int32, f1.shape), f1), ep
        del d0, d1
        xqd = np.array([-20, 40], np.int32)
        want = np.zeros((H, W), dat.dtype)
        for i in range(0, H, 64):
            for j in range(0, W, 64):
                ph, pw = min(64, H - i), min(64, W - j)
                orc.orc_apply_selfguided_restoration(V(G.at(dat) + (i * dat.shape[1] + j) * dat.itemsize), pw, ph, dat.shape[1], ep, P(xqd),
                                                     V(want.ctypes.data + (i * W + j) * dat.itemsize), W, BD, 1)
        d_out = device.DeviceBuffer(hip, want.nbytes)
        device.check(hip, hip.svt_hip_sgr_apply_unit(C.byref(unit), ep, P(xqd), V(d_out.ptr), W, None), "sgr_apply")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
        got = d_out.download(dat.dtype, want.shape)
        assert np.array_equal(got, want), (ep, np.argwhere(got != want)[:5])
        assert (want != dat[G.B:G.B + H, G.B:G.B + W]).any()


def test_wiener_stats_and_filter_4k10(hip, orc, plane4k):
    dat, src = plane4k
    rng = np.random.default_rng(4001)
    d_dat, d_src = device.DeviceBuffer(hip, dat.nbytes), device.DeviceBuffer(hip, src.nbytes)
    d_dat.upload(dat), d_src.upload(src)
    off = (G.B * dat.shape[1] + G.B) * dat.itemsize
    # 256 x 256 restoration units; the last row / column absorbs a remainder below half a unit (restoration.c:93-97)
    xs = [x for x in range(0, W - 127, 256)]
    ys = [y for y in range(0, H - 127, 256)]
    limits = [(x, W if W - x < 384 else x + 256, y, H if H - y < 384 else y + 256) for y in ys for x in xs]
    units = (abi.WienerUnit * len(limits))()
    for i, (hs, he, vs, ve) in enumerate(limits):
        units[i] = abi.WienerUnit(d_dat.ptr + off, d_src.ptr + off, dat.shape[1], src.shape[1], hs, he, vs, ve)
    dM, dH = device.DeviceBuffer(hip, len(limits) * 49 * 8), device.DeviceBuffer(hip, len(limits) * 49 * 49 * 8)
    device.check(hip, hip.svt_hip_wiener_stats(units, len(limits), 7, 1, BD, V(dM.ptr), V(dH.ptr), None), "wiener_stats")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    Mg, Hg = dM.download(np.int64, (len(limits), 49)), dH.download(np.int64, (len(limits), 49 * 49))
    for i in range(len(limits)):
        hs, he, vs, ve = limits[i]
        M, Hh = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
        orc.orc_wiener_compute_stats(7, V(G.at(dat)), V(G.at(src)), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M), P(Hh), 1, BD)
        assert np.array_equal(M, Mg[i]) and np.array_equal(Hh, Hg[i]), (i, limits[i])
    fx, kx = G.wiener_filter(rng)
    fy, ky = G.wiener_filter(rng)
    r0, r1 = G.wiener_rounds(BD)
    want = np.zeros((H, W), dat.dtype)
    for y in range(0, H, 64):
        for x in range(0, W, 64):
            ph, pw = min(64, H - y), min(64, W - x)
            orc.orc_wiener_convolve_add_src(V(G.at(dat) + (y * dat.shape[1] + x) * dat.itemsize), dat.shape[1],
                                            V(want.ctypes.data + (y * W + x) * dat.itemsize), W, P(fx), P(fy), pw, ph, r0, r1, BD, 1)
    d_out = device.DeviceBuffer(hip, want.nbytes)
    device.check(hip, hip.svt_hip_wiener_convolve(V(d_dat.ptr + off), dat.shape[1], V(d_out.ptr), W, W, H, P(fx), P(fy), 1, BD, None), "wiener_convolve")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    got = d_out.download(dat.dtype, want.shape)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]

"""GPU parity: Wiener restoration of libsvtav1_hip (through the C-ABI) against the oracle and the golden fixture."""
import ctypes as C
import os

import numpy as np
import pytest

import sgr_cases as G
from lf_cases import P, V
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu


def enc(addr, is16):
    return V(addr >> 1) if is16 else V(addr)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (8, 1), (12, 1)])
def test_tier_a_compute_stats(hip, orc, bd, is16):
    rng = np.random.default_rng(90 + bd + is16)
    for trial in range(10):
        win = (7, 5)[trial % 2]
        w, h = ((64, 48), (100, 37), (33, 64), (130, 70), (16, 16))[trial % 5]
        dat, src = G.sgr_plane(rng, w + 8, h + 8, bd, is16, (0, 2, 1)[trial % 3])
        hs, he, vs, ve = 3, 3 + w, 2, 2 + h
        M1, H1 = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
        M2, H2 = M1.copy(), H1.copy()
        orc.orc_wiener_compute_stats(win, V(G.at(dat)), V(G.at(src)), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M1), P(H1), is16, bd)
        if is16:
            hip.svt_av1_compute_stats_highbd_hip(win, enc(G.at(dat), 1), enc(G.at(src), 1), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M2),
                                                 P(H2), bd)
        else:
            hip.svt_av1_compute_stats_hip(win, V(G.at(dat)), V(G.at(src)), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M2), P(H2))
        w2 = win * win
        assert np.array_equal(M1[:w2], M2[:w2]) and np.array_equal(H1[:w2 * w2], H2[:w2 * w2]), (trial, win)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (12, 1)])
def test_tier_a_convolve(hip, orc, bd, is16):
    rng = np.random.default_rng(95 + bd)
    r0, r1 = G.wiener_rounds(bd)
    cp = abi.ConvolveParams(round_0=r0, round_1=r1)
    for trial in range(18):
        w, h = ((64, 64), (64, 56), (32, 32), (40, 17), (8, 8), (128, 64))[trial % 6]
        dat, _ = G.sgr_plane(rng, w, h, bd, is16, (0, 2, 1)[trial % 3])
        fx, keepx = G.wiener_filter(rng)
        fy, keepy = G.wiener_filter(rng)
        o1, o2 = np.zeros((h, w + 3), dat.dtype), np.zeros((h, w + 3), dat.dtype)
        orc.orc_wiener_convolve_add_src(V(G.at(dat)), dat.shape[1], P(o1), w + 3, P(fx), P(fy), w, h, r0, r1, bd, is16)
        if is16:
            hip.svt_av1_highbd_wiener_convolve_add_src_hip(enc(G.at(dat), 1), C.c_ssize_t(dat.shape[1]), enc(o2.ctypes.data, 1),
                                                           C.c_ssize_t(w + 3), P(fx), P(fy), w, h, C.byref(cp), bd)
        else:
            hip.svt_av1_wiener_convolve_add_src_hip(V(G.at(dat)), C.c_ssize_t(dat.shape[1]), P(o2), C.c_ssize_t(w + 3), P(fx), P(fy), w, h,
                                                    C.byref(cp))
        assert np.array_equal(o1, o2), (trial, w, h)


@pytest.mark.parametrize("bd,is16,win", [(8, 0, 7), (10, 1, 7), (10, 1, 5), (8, 0, 5)])
def test_tier_b_units_of_a_plane(hip, orc, bd, is16, win):
    """All restoration units of a plane in one call (units of different sizes), then the filter over the whole plane."""
    rng = np.random.default_rng(300 + bd + win)
    W, H = 520, 300
    dat, src = G.sgr_plane(rng, W, H, bd, is16, 0)
    d_dat, d_src = device.DeviceBuffer(hip, dat.nbytes), device.DeviceBuffer(hip, src.nbytes)
    d_dat.upload(dat), d_src.upload(src)
    off = (G.B * dat.shape[1] + G.B) * dat.itemsize
    limits = [(x, min(x + 256, W), y, min(y + 128, H)) for y in range(0, H, 128) for x in range(0, W, 256)]
    units = (abi.WienerUnit * len(limits))()
    for i, (hs, he, vs, ve) in enumerate(limits):
        units[i] = abi.WienerUnit(d_dat.ptr + off, d_src.ptr + off, dat.shape[1], src.shape[1], hs, he, vs, ve)
    dM, dH = device.DeviceBuffer(hip, len(limits) * 49 * 8), device.DeviceBuffer(hip, len(limits) * 49 * 49 * 8)
    device.check(hip, hip.svt_hip_wiener_stats(units, len(limits), win, is16, bd, V(dM.ptr), V(dH.ptr), None), "wiener_stats")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    Mg, Hg = dM.download(np.int64, (len(limits), 49)), dH.download(np.int64, (len(limits), 49 * 49))
    w2 = win * win
    for i, (hs, he, vs, ve) in enumerate(limits):
        M, Hh = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
        orc.orc_wiener_compute_stats(win, V(G.at(dat)), V(G.at(src)), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M), P(Hh), is16, bd)
        assert np.array_equal(M[:w2], Mg[i, :w2]) and np.array_equal(Hh[:w2 * w2], Hg[i, :w2 * w2]), i
    fx, kx = G.wiener_filter(rng)
    fy, ky = G.wiener_filter(rng)
    r0, r1 = G.wiener_rounds(bd)
    want = np.zeros((H, W), dat.dtype)
    for y in range(0, H, 64):
        for x in range(0, W, 64):
            ph, pw = min(64, H - y), min(64, W - x)
            orc.orc_wiener_convolve_add_src(V(G.at(dat) + (y * dat.shape[1] + x) * dat.itemsize), dat.shape[1],
                                            V(want.ctypes.data + (y * W + x) * dat.itemsize), W, P(fx), P(fy), pw, ph, r0, r1, bd, is16)
    d_out = device.DeviceBuffer(hip, want.nbytes)
    device.check(hip, hip.svt_hip_wiener_convolve(V(d_dat.ptr + off), dat.shape[1], V(d_out.ptr), W, W, H, P(fx), P(fy), is16, bd, None), "wiener_convolve")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    assert np.array_equal(d_out.download(dat.dtype, want.shape), want)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (12, 1)])
def test_stats_extreme_values(hip, orc, bd, is16):
    """The statistics are an int8 Gram matrix of digit planes on the matrix cores: the ends of the sample range (largest digits of either
    sign), alternating extremes and a unit much larger than a workgroup's share of it, against the oracle."""
    rng = np.random.default_rng(17 + bd)
    top, dt = (1 << bd) - 1, (np.uint16 if is16 else np.uint8)
    W, H = 330, 290
    yy, xx = np.mgrid[0:H + 2 * G.B, 0:W + 2 * G.B]
    patterns = [np.zeros_like(xx), np.full_like(xx, top), ((xx + yy) & 1) * top, ((xx // 3 + yy // 5) & 1) * top,
                rng.choice([0, 1, top - 1, top], size=xx.shape)]
    for pi, pat in enumerate(patterns):
        for win in (7, 5):
            dat = np.ascontiguousarray(pat.astype(dt))
            src = np.ascontiguousarray(patterns[(pi + 2) % len(patterns)].astype(dt))
            M1, H1 = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
            M2, H2 = M1.copy(), H1.copy()
            orc.orc_wiener_compute_stats(win, V(G.at(dat)), V(G.at(src)), 0, W, 0, H, dat.shape[1], src.shape[1], P(M1), P(H1), is16, bd)
            if is16:
                hip.svt_av1_compute_stats_highbd_hip(win, enc(G.at(dat), 1), enc(G.at(src), 1), 0, W, 0, H, dat.shape[1], src.shape[1], P(M2), P(H2), bd)
            else:
                hip.svt_av1_compute_stats_hip(win, V(G.at(dat)), V(G.at(src)), 0, W, 0, H, dat.shape[1], src.shape[1], P(M2), P(H2))
            w2 = win * win
            assert np.array_equal(M1[:w2], M2[:w2]) and np.array_equal(H1[:w2 * w2], H2[:w2 * w2]), (pi, win)


def test_golden(hip):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wiener.npz"))
    for key, bd, is16, win, w, h, seed in G.GOLDEN_WIENER:
        dat, src = g[key + "_dat"].copy(), g[key + "_src"].copy()
        M, H = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
        if is16:
            hip.svt_av1_compute_stats_highbd_hip(win, enc(G.at(dat), 1), enc(G.at(src), 1), 1, 1 + w, 2, 2 + h, dat.shape[1], src.shape[1], P(M),
                                                 P(H), bd)
        else:
            hip.svt_av1_compute_stats_hip(win, V(G.at(dat)), V(G.at(src)), 1, 1 + w, 2, 2 + h, dat.shape[1], src.shape[1], P(M), P(H))
        w2 = win * win
        assert np.array_equal(M[:w2], g[key + "_M"][:w2]) and np.array_equal(H[:w2 * w2], g[key + "_H"][:w2 * w2]), key
        fx, fy = g[key + "_fx"].copy(), g[key + "_fy"].copy()
        cw, ch = min(w, 128), min(h, 128)
        out = np.zeros((ch, cw), dat.dtype)
        cp = abi.ConvolveParams()
        if is16:
            hip.svt_av1_highbd_wiener_convolve_add_src_hip(enc(G.at(dat), 1), C.c_ssize_t(dat.shape[1]), enc(out.ctypes.data, 1),
                                                           C.c_ssize_t(cw), P(fx), P(fy), cw, ch, C.byref(cp), bd)
        else:
            hip.svt_av1_wiener_convolve_add_src_hip(V(G.at(dat)), C.c_ssize_t(dat.shape[1]), P(out), C.c_ssize_t(cw), P(fx), P(fy), cw, ch,
                                                    C.byref(cp))
        assert np.array_equal(out, g[key + "_out"]), key


def test_bad_arguments(hip):
    assert hip.svt_hip_wiener_stats(None, 0, 7, 0, 8, None, None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    f = np.zeros(8, np.int16)
    assert hip.svt_hip_wiener_convolve(None, 0, None, 0, 0, 0, P(f), P(f), 0, 8, None) == abi.SVT_HIP_ERR_BAD_PARAMETER

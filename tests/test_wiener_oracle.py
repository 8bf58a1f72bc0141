"""CPU: the Wiener-restoration oracle against the REAL reference functions (oracle/_ref RTCD pointers)."""
import ctypes as C

import numpy as np
import pytest

import lf_cases as L
import sgr_cases as G
from lf_cases import P, V


class ConvolveParams(C.Structure):      # definitions.h:580-593
    _fields_ = [("ref", C.c_int32), ("do_average", C.c_int32), ("dst", C.c_void_p), ("dst_stride", C.c_int32), ("round_0", C.c_int32),
                ("round_1", C.c_int32), ("plane", C.c_int32), ("is_compound", C.c_int32), ("use_jnt_comp_avg", C.c_int32),
                ("fwd_offset", C.c_int32), ("bck_offset", C.c_int32), ("use_dist_wtd_comp_avg", C.c_int32)]


from sgr_cases import wiener_filter, wiener_rounds  # noqa: E402


def enc(addr, is16):
    return V(addr >> 1) if is16 else V(addr)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (8, 1)])
def test_compute_stats(orc, ref, bd, is16):
    rng = np.random.default_rng(90 + bd + is16)
    f8 = L.rtcd(ref, "svt_av1_compute_stats", None, C.c_int32, V, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, V, V)
    f16 = L.rtcd(ref, "svt_av1_compute_stats_highbd", None, C.c_int32, V, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                 C.c_int32, V, V, C.c_int32)
    for trial in range(10):
        win = (7, 5)[trial % 2]
        w, h = ((64, 48), (100, 37), (33, 64), (130, 70), (16, 16))[trial % 5]
        dat, src = G.sgr_plane(rng, w + 8, h + 8, bd, is16, (0, 2, 1)[trial % 3])
        hs, he, vs, ve = 3, 3 + w, 2, 2 + h                  # a unit strictly inside the plane (taps reach 3 samples out)
        M1, H1 = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
        M2, H2 = M1.copy(), H1.copy()
        if is16:
            f16(win, enc(G.at(dat), 1), enc(G.at(src), 1), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M1), P(H1), bd)
        else:
            f8(win, V(G.at(dat)), V(G.at(src)), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M1), P(H1))
        orc.orc_wiener_compute_stats(win, V(G.at(dat)), V(G.at(src)), hs, he, vs, ve, dat.shape[1], src.shape[1], P(M2), P(H2), is16, bd)
        assert np.array_equal(M1, M2) and np.array_equal(H1, H2), (trial, win)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (12, 1)])
def test_convolve_add_src(orc, ref, bd, is16):
    rng = np.random.default_rng(95 + bd)
    f8 = L.rtcd(ref, "svt_av1_wiener_convolve_add_src", None, V, C.c_ssize_t, V, C.c_ssize_t, V, V, C.c_int32, C.c_int32, V)
    f16 = L.rtcd(ref, "svt_av1_highbd_wiener_convolve_add_src", None, V, C.c_ssize_t, V, C.c_ssize_t, V, V, C.c_int32, C.c_int32, V, C.c_int32)
    r0, r1 = wiener_rounds(bd)
    cp = ConvolveParams(round_0=r0, round_1=r1)
    for trial in range(24):
        w, h = ((64, 64), (64, 56), (32, 32), (40, 17), (8, 8), (128, 64))[trial % 6]
        dat, _ = G.sgr_plane(rng, w, h, bd, is16, (0, 2, 1)[trial % 3])
        fx, keepx = wiener_filter(rng)
        fy, keepy = wiener_filter(rng)
        o1, o2 = np.zeros((h, w + 3), dat.dtype), np.zeros((h, w + 3), dat.dtype)
        if is16:
            f16(enc(G.at(dat), 1), dat.shape[1], enc(o1.ctypes.data, 1), w + 3, P(fx), P(fy), w, h, C.byref(cp), bd)
        else:
            f8(V(G.at(dat)), dat.shape[1], P(o1), w + 3, P(fx), P(fy), w, h, C.byref(cp))
        orc.orc_wiener_convolve_add_src(V(G.at(dat)), dat.shape[1], P(o2), w + 3, P(fx), P(fy), w, h, r0, r1, bd, is16)
        assert np.array_equal(o1, o2), (trial, w, h)


def test_wiener_oracle_vs_golden(orc):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wiener.npz"))
    for key, bd, is16, win, w, h, seed in G.GOLDEN_WIENER:
        dat, src = g[key + "_dat"].copy(), g[key + "_src"].copy()
        M, H = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
        orc.orc_wiener_compute_stats(win, V(G.at(dat)), V(G.at(src)), 1, 1 + w, 2, 2 + h, dat.shape[1], src.shape[1], P(M), P(H), is16, bd)
        assert np.array_equal(M, g[key + "_M"]) and np.array_equal(H, g[key + "_H"]), key
        fx, fy = g[key + "_fx"].copy(), g[key + "_fy"].copy()
        r0, r1 = wiener_rounds(bd)
        out = np.zeros((h, w), dat.dtype)
        orc.orc_wiener_convolve_add_src(V(G.at(dat)), dat.shape[1], P(out), w, P(fx), P(fy), min(w, 128), min(h, 128), r0, r1, bd, is16)
        assert np.array_equal(out[:min(h, 128), :min(w, 128)], g[key + "_out"]), key

"""CPU: the self-guided restoration oracle against the REAL reference functions (oracle/_ref RTCD pointers)."""
import ctypes as C

import numpy as np
import pytest

import lf_cases as L
from lf_cases import P, V

I64 = C.c_int64


class SgrParams(C.Structure):           # SgrParamsType (definitions.h:1758-1761)
    _fields_ = [("r", C.c_int32 * 2), ("s", C.c_int32 * 2)]


from sgr_cases import sgr_plane as _sgr_plane, at as _at, B as _B  # noqa: E402


def sgr_plane(rng, w, h, bd, is16, kind):
    dat, src = _sgr_plane(rng, w, h, bd, is16, kind)
    return dat, src, _B


def at(a, B):
    return _at(a)


def refptr(addr, is16):
    return V(addr >> 1) if is16 else V(addr)     # CONVERT_TO_BYTEPTR for uint16 buffers


def test_tables(orc, ref):
    prm = (SgrParams * 16).in_dll(ref, "svt_aom_eb_sgr_params")
    mine = (C.c_int32 * 64).in_dll(orc, "orc_sgr_params")
    for ep in range(16):
        assert [prm[ep].r[0], prm[ep].r[1], prm[ep].s[0], prm[ep].s[1]] == list(mine[4 * ep:4 * ep + 4])
    # the two LUTs are closed forms in the oracle: checked through the filter on flat / extreme inputs below and here
    xb = (C.c_int32 * 256).in_dll(ref, "svt_aom_eb_x_by_xplus1")
    ob = (C.c_int32 * 25).in_dll(ref, "svt_aom_eb_one_by_x")
    assert list(xb) == [1] + [min(256, (256 * z + (z + 1) // 2) // (z + 1)) if z < 255 else 256 for z in range(1, 256)]
    assert list(ob) == [(4096 + n // 2) // n for n in range(1, 26)]


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (8, 1)])
def test_selfguided_filter_and_apply(orc, ref, bd, is16):
    rng = np.random.default_rng(bd + is16)
    filt = L.rtcd(ref, "svt_av1_selfguided_restoration", None, V, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, C.c_int32, C.c_int32,
                  C.c_int32)
    appl = L.rtcd(ref, "svt_apply_selfguided_restoration", None, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, V,
                  C.c_int32, C.c_int32)
    tmp = np.zeros(2 * 161 * 1024, np.int32)          # SGRPROJ_TMPBUF_SIZE worth of int32 (restoration.h:84-87) is 2 * UNITPELS_MAX
    for trial in range(48):
        ep = trial % 16
        w, h = ((64, 64), (64, 56), (32, 32), (40, 17), (8, 8), (61, 64))[trial % 6]
        dat, src, B = sgr_plane(rng, w, h, bd, is16, (0, 0, 1, 2)[trial % 4])
        fs = w + 3
        f0a, f1a = np.full((h, fs), 7777, np.int32), np.full((h, fs), 7777, np.int32)
        f0b, f1b = f0a.copy(), f1a.copy()
        filt(refptr(at(dat, B), is16), w, h, dat.shape[1], P(f0a), P(f1a), fs, ep, bd, is16)
        orc.orc_selfguided_restoration(V(at(dat, B)), w, h, dat.shape[1], P(f0b), P(f1b), fs, ep, bd, is16)
        assert np.array_equal(f0a, f0b) and np.array_equal(f1a, f1b), (trial, ep, w, h)
        xqd = np.array([int(rng.integers(-96, 32)), int(rng.integers(-32, 96))], np.int32)
        o1, o2 = np.zeros((h, w + 5), dat.dtype), np.zeros((h, w + 5), dat.dtype)
        appl(refptr(at(dat, B), is16), w, h, dat.shape[1], ep, P(xqd), refptr(o1.ctypes.data, is16), w + 5, P(tmp), bd, is16)
        orc.orc_apply_selfguided_restoration(V(at(dat, B)), w, h, dat.shape[1], ep, P(xqd), P(o2), w + 5, bd, is16)
        assert np.array_equal(o1, o2), (trial, ep)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1)])
def test_proj_error_and_subspace(orc, ref, bd, is16):
    rng = np.random.default_rng(40 + bd)
    err = L.rtcd(ref, "svt_av1_highbd_pixel_proj_error" if is16 else "svt_av1_lowbd_pixel_proj_error", I64, V, C.c_int32, C.c_int32,
                 C.c_int32, V, C.c_int32, V, C.c_int32, V, C.c_int32, V, V)
    sub = L.rtcd(ref, "svt_get_proj_subspace", None, V, C.c_int, C.c_int, C.c_int, V, C.c_int, C.c_int, V, C.c_int, V, C.c_int, V, V)
    prm = (SgrParams * 16).in_dll(ref, "svt_aom_eb_sgr_params")
    orc.orc_sgr_pixel_proj_error.restype = I64
    for trial in range(48):
        ep = trial % 16
        w, h = ((96, 80), (64, 64), (33, 47))[trial % 3]
        dat, src, B = sgr_plane(rng, w, h, bd, is16, (0, 0, 2, 1)[trial % 4])
        fs = ((w + 7) & ~7) + 8
        f0, f1 = np.zeros((h, fs), np.int32), np.zeros((h, fs), np.int32)
        orc.orc_sgr_filter_unit(V(at(dat, B)), w, h, dat.shape[1], is16, bd, 64, 64, ep, P(f0), P(f1), fs)
        xq = np.array([int(rng.integers(-100, 100)), int(rng.integers(-100, 100))], np.int32)
        a = err(refptr(at(src, B), is16), w, h, src.shape[1], refptr(at(dat, B), is16), dat.shape[1], P(f0), fs, P(f1), fs, P(xq),
                C.addressof(prm[ep]))
        b = orc.orc_sgr_pixel_proj_error(V(at(src, B)), w, h, src.shape[1], V(at(dat, B)), dat.shape[1], P(f0), fs, P(f1), fs, P(xq), ep,
                                         is16)
        assert a == b, (trial, a, b)
        x1, x2 = np.zeros(2, np.int32), np.zeros(2, np.int32)
        sub(refptr(at(src, B), is16), w, h, src.shape[1], refptr(at(dat, B), is16), dat.shape[1], is16, P(f0), fs, P(f1), fs, P(x1),
            C.addressof(prm[ep]))
        orc.orc_get_proj_subspace(V(at(src, B)), w, h, src.shape[1], V(at(dat, B)), dat.shape[1], is16, P(f0), fs, P(f1), fs, P(x2), ep)
        assert np.array_equal(x1, x2), (trial, x1, x2)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1)])
def test_search_unit_vs_real_driver(orc, ref, bd, is16):
    """orc_sgr_search_unit against the reference's (static) search_selfguided_restoration, reached through
    oracle/ref_harness_sgr.c which compiles restoration_pick.c in place."""
    rng = np.random.default_rng(70 + bd)
    orc.orc_sgr_search_unit.restype = I64
    for trial in range(6):
        w, h = ((96, 80), (136, 72), (64, 64), (200, 120), (56, 40), (384, 96))[trial]
        dat, src, B = sgr_plane(rng, w, h, bd, is16, (0, 0, 2)[trial % 3])
        pu = 64 if trial % 2 == 0 else 32
        start, end, inc, refine = ((0, 16, 1, 1), (0, 16, 2, 1), (10, 16, 1, 0), (0, 8, 3, 1), (14, 16, 1, 1), (0, 16, 4, 1))[trial]
        o1, o2 = np.zeros(3, np.int32), np.zeros(3, np.int32)
        assert ref.ref_sgr_search_unit(refptr(at(dat, B), is16), w, h, dat.shape[1], refptr(at(src, B), is16), src.shape[1], is16, bd,
                                       pu, pu, start, end, inc, refine, P(o1)) == 0
        orc.orc_sgr_search_unit(V(at(dat, B)), w, h, dat.shape[1], V(at(src, B)), src.shape[1], is16, bd, pu, pu, start, end, inc, refine,
                                P(o2))
        assert np.array_equal(o1, o2), (trial, o1, o2)


def test_sgr_oracle_vs_golden(orc):
    """No reference needed: tests/golden/sgr.npz (search driver, filter and apply by the reference) pins the oracle."""
    import os
    import sgr_cases as G
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sgr.npz"))
    orc.orc_sgr_search_unit.restype = I64
    for key, w, h, bd, is16, kind, pu, (s0, s1, inc, refine), seed in G.GOLDEN_SGR:
        dat, src, best = g[key + "_dat"].copy(), g[key + "_src"].copy(), g[key + "_best"]
        out = np.zeros(3, np.int32)
        orc.orc_sgr_search_unit(V(G.at(dat)), w, h, dat.shape[1], V(G.at(src)), src.shape[1], is16, bd, pu, pu, s0, s1, inc, refine, P(out))
        assert np.array_equal(out, best), key
        pw, ph = min(pu, w), min(pu, h)
        f0, f1 = np.zeros((ph, pw), np.int32), np.zeros((ph, pw), np.int32)
        orc.orc_selfguided_restoration(V(G.at(dat)), pw, ph, dat.shape[1], P(f0), P(f1), pw, int(best[0]), bd, is16)
        assert np.array_equal(f0, g[key + "_flt0"]) and np.array_equal(f1, g[key + "_flt1"]), key
        rec = np.zeros((ph, pw), dat.dtype)
        xqd = np.array([best[1], best[2]], np.int32)
        orc.orc_apply_selfguided_restoration(V(G.at(dat)), pw, ph, dat.shape[1], int(best[0]), P(xqd), P(rec), pw, bd, is16)
        assert np.array_equal(rec, g[key + "_rec"]), key

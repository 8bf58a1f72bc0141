"""CPU: the single-reference interpolation oracle against the REAL reference functions (oracle/_ref RTCD pointers)."""
import ctypes as C

import numpy as np
import pytest

import conv_cases as K
import lf_cases as L
from lf_cases import P, V
from test_wiener_oracle import ConvolveParams


def test_kernel_tables_match_reference(ref):
    for name, tab in K.TABLES.items():
        got = np.frombuffer((C.c_int16 * 128).in_dll(ref, name), np.int16).reshape(16, 8)
        assert np.array_equal(got, np.array(tab, np.int16)), name


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (12, 1)])
def test_convolve_sr(orc, ref, bd, is16):
    rng = np.random.default_rng(120 + bd)
    r0, r1 = K.conv_rounds(bd)
    sig8 = (V, C.c_int32, V, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, C.c_int32, V)
    fns = {}
    for mode in ("2d_sr", "x_sr", "y_sr", "2d_copy_sr"):
        name = f"svt_av1_highbd_convolve_{mode}" if is16 else f"svt_av1_convolve_{mode}"
        fns[mode] = L.rtcd(ref, name, None, *(sig8 + ((C.c_int32,) if is16 else ())))
    tabs = {n: K.kernel_table(n) for n in K.TABLES}
    for trial in range(66):
        w, h = K.SIZES[trial % len(K.SIZES)]
        name = list(K.TABLES)[trial % 3]
        tab = tabs[name][0]
        sx, sy = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        mode = ("2d_sr", "x_sr", "y_sr", "2d_copy_sr")[trial % 4]
        plane, at = K.ref_plane(rng, w, h, bd, is16, (0, 2, 1)[trial % 3])
        fp = K.InterpFilterParams(tab.ctypes.data, 8, 16, trial % 3)
        cp = ConvolveParams(round_0=r0, round_1=r1)
        o1, o2 = np.zeros((h, w + 3), plane.dtype), np.zeros((h, w + 3), plane.dtype)
        args = [V(at), plane.shape[1], P(o1), w + 3, w, h, C.byref(fp), C.byref(fp), sx, sy, C.byref(cp)] + ([bd] if is16 else [])
        fns[mode](*args)
        tx = 8 if mode in ("2d_sr", "x_sr") else 0
        ty = 8 if mode in ("2d_sr", "y_sr") else 0
        orc.orc_convolve_sr(V(at), plane.shape[1], P(o2), w + 3, w, h, V(tab[sx].ctypes.data), tx, V(tab[sy].ctypes.data), ty, r0, r1, bd, is16)
        assert np.array_equal(o1, o2), (trial, mode, w, h, sx, sy)


def test_convolve_oracle_vs_golden(orc):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convolve.npz"))
    for i in range(int(g["n"])):
        bd, is16, w, h, mode, ti, sx, sy = (int(v) for v in g[f"c{i}_meta"])
        plane = g[f"c{i}_plane"].copy()
        at = plane.ctypes.data + (8 * plane.shape[1] + 8) * plane.itemsize
        tab = np.array(K.TABLES[list(K.TABLES)[ti]], np.int16)
        r0, r1 = K.conv_rounds(bd)
        o = np.zeros((h, w), plane.dtype)
        tx, ty = (8 if mode in (0, 1) else 0), (8 if mode in (0, 2) else 0)
        orc.orc_convolve_sr(V(at), plane.shape[1], P(o), w, w, h, V(tab[sx].ctypes.data), tx, V(tab[sy].ctypes.data), ty, r0, r1, bd, is16)
        assert np.array_equal(o, g[f"c{i}_out"]), i


def run_orc_jnt(orc, c, bd, is16):
    """both predictions of one compound block through the oracle: returns (conv buffer after the first, pixels after the second)"""
    w, h, mode, ti, sx, sy, p0, a0, p1, a1, avg, fwd, bck = c
    tab = np.array(K.TABLES[list(K.TABLES)[ti]], np.int16)
    r0, r1 = K.conv_rounds_compound(bd)
    tx, ty = (8 if mode in (0, 1) else 0), (8 if mode in (0, 2) else 0)
    cb = np.zeros((h, w + 5), np.uint16)
    out = np.zeros((h, w + 3), p0.dtype)
    orc.orc_convolve_jnt(V(a0), p0.shape[1], P(out), w + 3, w, h, V(tab[sx].ctypes.data), tx, V(tab[sy].ctypes.data), ty, r0, r1, bd, is16,
                         P(cb), w + 5, 1, 0, 0)
    first = cb.copy()
    orc.orc_convolve_jnt(V(a1), p1.shape[1], P(out), w + 3, w, h, V(tab[sy].ctypes.data), tx, V(tab[sx].ctypes.data), ty, r0, r1, bd, is16,
                         P(cb), w + 5, avg, fwd, bck)
    return first, out


def run_fn_jnt(fns, c, bd, is16, CP, FP):
    """the same through functions with the reference's signatures (the reference's own, or the HIP Tier A leaves)"""
    w, h, mode, ti, sx, sy, p0, a0, p1, a1, avg, fwd, bck = c
    tab = K.kernel_table(list(K.TABLES)[ti])[0]
    r0, r1 = K.conv_rounds_compound(bd)
    cb = np.zeros((h, w + 5), np.uint16)
    out = np.zeros((h, w + 3), p0.dtype)
    fp = FP(tab.ctypes.data, 8, 16, ti)
    cp = CP(do_average=0, dst=cb.ctypes.data, dst_stride=w + 5, round_0=r0, round_1=r1, is_compound=1)
    tail = [bd] if is16 else []
    fns[mode](*([V(a0), p0.shape[1], P(out), w + 3, w, h, C.byref(fp), C.byref(fp), sx, sy, C.byref(cp)] + tail))
    first = cb.copy()
    cp.do_average, cp.use_jnt_comp_avg, cp.fwd_offset, cp.bck_offset = 1, int(avg == 3), fwd, bck
    fns[mode](*([V(a1), p1.shape[1], P(out), w + 3, w, h, C.byref(fp), C.byref(fp), sy, sx, C.byref(cp)] + tail))
    return first, out


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (12, 1)])
def test_convolve_jnt(orc, ref, bd, is16):
    sig8 = (V, C.c_int32, V, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, C.c_int32, V)
    fns = [L.rtcd(ref, (f"svt_av1_highbd_jnt_convolve_{m}" if is16 else f"svt_av1_jnt_convolve_{m}"), None,
                  *(sig8 + ((C.c_int32,) if is16 else ()))) for m in K.JNT_MODES]
    for i, c in enumerate(K.jnt_cases(bd, is16)):
        f1, o1 = run_fn_jnt(fns, c, bd, is16, ConvolveParams, K.InterpFilterParams)
        f2, o2 = run_orc_jnt(orc, c, bd, is16)
        assert np.array_equal(f1, f2) and np.array_equal(o1, o2), (i, c[:6])


def test_convolve_jnt_oracle_vs_golden(orc):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convolve_jnt.npz"))
    k = 0
    for bd, is16 in ((8, 0), (10, 1)):
        for c in K.jnt_cases(bd, is16, n=16, seed=1):
            f, o = run_orc_jnt(orc, c, bd, is16)
            assert np.array_equal(f[:, :c[0]], g[f"first{k}"]) and np.array_equal(o[:, :c[0]], g[f"out{k}"]), k
            k += 1

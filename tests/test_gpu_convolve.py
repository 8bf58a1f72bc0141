"""GPU parity: single-reference inter-prediction interpolation (through the C-ABI) against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import conv_cases as K
from lf_cases import P, V
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu
MODES = ("2d_sr", "x_sr", "y_sr", "2d_copy_sr")


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (12, 1)])
def test_tier_a(hip, orc, bd, is16):
    rng = np.random.default_rng(120 + bd)
    r0, r1 = K.conv_rounds(bd)
    tabs = {n: K.kernel_table(n) for n in K.TABLES}
    for trial in range(44):
        w, h = K.SIZES[trial % len(K.SIZES)]
        tab = tabs[list(K.TABLES)[trial % 3]][0]
        sx, sy = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        mode = MODES[trial % 4]
        plane, at = K.ref_plane(rng, w, h, bd, is16, (0, 2, 1)[trial % 3])
        fp = abi.InterpFilterParams(tab.ctypes.data, 8, 16, trial % 3)
        cp = abi.ConvolveParams(round_0=r0, round_1=r1)
        o1, o2 = np.zeros((h, w + 3), plane.dtype), np.zeros((h, w + 3), plane.dtype)
        tx = 8 if mode in ("2d_sr", "x_sr") else 0
        ty = 8 if mode in ("2d_sr", "y_sr") else 0
        orc.orc_convolve_sr(V(at), plane.shape[1], P(o1), w + 3, w, h, V(tab[sx].ctypes.data), tx, V(tab[sy].ctypes.data), ty, r0, r1, bd, is16)
        fn = getattr(hip, f"svt_av1_highbd_convolve_{mode}_hip" if is16 else f"svt_av1_convolve_{mode}_hip")
        fn(*([V(at), plane.shape[1], P(o2), w + 3, w, h, C.byref(fp), C.byref(fp), sx, sy, C.byref(cp)] + ([bd] if is16 else [])))
        assert np.array_equal(o1, o2), (trial, mode, w, h, sx, sy)


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1)])
def test_tier_b_batch(hip, orc, bd, is16):
    """A whole picture's worth of blocks (random sizes, phases, filters, modes) from one reference plane in one launch."""
    rng = np.random.default_rng(500 + bd)
    W, H = 640, 384
    plane, at0 = K.ref_plane(rng, W, H, bd, is16, 0)
    d_ref = device.DeviceBuffer(hip, plane.nbytes)
    d_ref.upload(plane)
    off0 = at0 - plane.ctypes.data
    out = np.zeros((H, W), plane.dtype)
    d_out = device.DeviceBuffer(hip, out.nbytes)
    d_out.fill(0)
    r0, r1 = K.conv_rounds(bd)
    tabs = [np.array(K.TABLES[n], np.int16) for n in K.TABLES]
    descs, want = [], np.zeros_like(out)
    for by in range(0, H, 128):
        for bx in range(0, W, 128):
            bs = int(rng.choice([16, 32, 64, 128]))
            for y in range(by, by + 128, bs):
                for x in range(bx, bx + 128, bs):
                    w, h = bs, bs
                    if rng.random() < 0.3 and bs > 16:
                        h = bs // 2        # a rectangular block; the lower half is predicted separately
                    for (yy, hh) in ((y, h),) + (((y + h, bs - h),) if h != bs else ()):
                        mvx, mvy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))      # whole-sample part, stays inside the border
                        sx, sy = int(rng.integers(0, 16)), int(rng.integers(0, 16))
                        mode = int(rng.integers(0, 4))
                        tx = 8 if mode in (0, 1) else 0
                        ty = 8 if mode in (0, 2) else 0
                        t = tabs[int(rng.integers(0, 3))]
                        so = ((yy + mvy) * plane.shape[1] + x + mvx) * plane.itemsize
                        d = abi.ConvolveDesc(d_ref.ptr + off0 + so, d_out.ptr + (yy * W + x) * plane.itemsize, plane.shape[1], W, w, hh,
                                             (C.c_int16 * 8)(*t[sx]), (C.c_int16 * 8)(*t[sy]), tx, ty, r0, r1, bd, is16)
                        descs.append(d)
                        orc.orc_convolve_sr(V(at0 + so), plane.shape[1], V(want.ctypes.data + (yy * W + x) * plane.itemsize), W, w, hh,
                                            V(t[sx].ctypes.data), tx, V(t[sy].ctypes.data), ty, r0, r1, bd, is16)
    arr = (abi.ConvolveDesc * len(descs))(*descs)
    d_desc = device.DeviceBuffer(hip, C.sizeof(arr))
    d_desc.upload(np.frombuffer(arr, np.uint8))
    device.check(hip, hip.svt_hip_convolve_sr_batch(V(d_desc.ptr), len(descs), None), "convolve_sr_batch")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    assert np.array_equal(d_out.download(plane.dtype, out.shape), want)
    assert hip.svt_hip_convolve_sr_batch(None, 0, None) == abi.SVT_HIP_ERR_BAD_PARAMETER


def test_golden(hip):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convolve.npz"))
    tabs = {n: K.kernel_table(n) for n in K.TABLES}
    for i in range(int(g["n"])):
        bd, is16, w, h, mode, ti, sx, sy = (int(v) for v in g[f"c{i}_meta"])
        plane = g[f"c{i}_plane"].copy()
        at = plane.ctypes.data + (8 * plane.shape[1] + 8) * plane.itemsize
        tab = tabs[list(K.TABLES)[ti]][0]
        r0, r1 = K.conv_rounds(bd)
        fp = abi.InterpFilterParams(tab.ctypes.data, 8, 16, ti)
        cp = abi.ConvolveParams(round_0=r0, round_1=r1)
        o = np.zeros((h, w), plane.dtype)
        fn = getattr(hip, f"svt_av1_highbd_convolve_{MODES[mode]}_hip" if is16 else f"svt_av1_convolve_{MODES[mode]}_hip")
        fn(*([V(at), plane.shape[1], P(o), w, w, h, C.byref(fp), C.byref(fp), sx, sy, C.byref(cp)] + ([bd] if is16 else [])))
        assert np.array_equal(o, g[f"c{i}_out"]), i


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1), (12, 1)])
def test_tier_a_compound(hip, orc, bd, is16):
    """svt_av1_(highbd_)jnt_convolve_{2d,x,y,2d_copy}_hip: first prediction into the ConvBufType buffer, second prediction
    averaged (plain / distance-weighted) into pixels — both steps against the oracle."""
    import test_convolve_oracle as T
    fns = [getattr(hip, f"svt_av1_highbd_jnt_convolve_{m}_hip" if is16 else f"svt_av1_jnt_convolve_{m}_hip") for m in K.JNT_MODES]
    for i, c in enumerate(K.jnt_cases(bd, is16)):
        f1, o1 = T.run_fn_jnt(fns, c, bd, is16, abi.ConvolveParams, abi.InterpFilterParams)
        f2, o2 = T.run_orc_jnt(orc, c, bd, is16)
        assert np.array_equal(f1, f2) and np.array_equal(o1, o2), (i, c[:6])


def test_compound_golden(hip):
    import test_convolve_oracle as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convolve_jnt.npz"))
    k = 0
    for bd, is16 in ((8, 0), (10, 1)):
        fns = [getattr(hip, f"svt_av1_highbd_jnt_convolve_{m}_hip" if is16 else f"svt_av1_jnt_convolve_{m}_hip") for m in K.JNT_MODES]
        for c in K.jnt_cases(bd, is16, n=16, seed=1):
            f, o = T.run_fn_jnt(fns, c, bd, is16, abi.ConvolveParams, abi.InterpFilterParams)
            assert np.array_equal(f[:, :c[0]], g[f"first{k}"]) and np.array_equal(o[:, :c[0]], g[f"out{k}"]), k
            k += 1


@pytest.mark.parametrize("bd,is16", [(8, 0), (10, 1)])
def test_tier_b_compound_batch(hip, orc, bd, is16):
    """Compound prediction of a tiled picture: one launch for every block's first reference (into a picture-sized
    ConvBufType plane), one for the second reference with the average, both from device-resident reference planes."""
    rng = np.random.default_rng(640 + bd)
    W, H = 512, 256
    p0, a0 = K.ref_plane(rng, W, H, bd, is16, 0)
    p1, a1 = K.ref_plane(rng, W, H, bd, is16, 2)
    d0, d1 = device.DeviceBuffer(hip, p0.nbytes), device.DeviceBuffer(hip, p1.nbytes)
    d0.upload(p0), d1.upload(p1)
    o0, o1 = a0 - p0.ctypes.data, a1 - p1.ctypes.data
    d_out, d_cb = device.DeviceBuffer(hip, W * H * p0.itemsize), device.DeviceBuffer(hip, W * H * 2)
    d_out.fill(0), d_cb.fill(0)
    r0, r1 = K.conv_rounds_compound(bd)
    tabs = [np.array(K.TABLES[n], np.int16) for n in K.TABLES]
    first, second = [], []
    want, cb = np.zeros((H, W), p0.dtype), np.zeros((H, W), np.uint16)
    for y in range(0, H, 64):
        for x in range(0, W, 64):
            bs = int(rng.choice([8, 16, 32, 64]))
            for yy in range(y, y + 64, bs):
                for xx in range(x, x + 64, bs):
                    avg = int(rng.choice([2, 3]))
                    fwd, bck = K.DIST_WEIGHTS[int(rng.integers(0, len(K.DIST_WEIGHTS)))]
                    for ref_i, (dev, off, plane, at, lst) in enumerate(((d0, o0, p0, a0, first), (d1, o1, p1, a1, second))):
                        mvx, mvy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
                        sx, sy = int(rng.integers(0, 16)), int(rng.integers(0, 16))
                        mode = int(rng.integers(0, 4))
                        tx, ty = (8 if mode in (0, 1) else 0), (8 if mode in (0, 2) else 0)
                        t = tabs[int(rng.integers(0, 3))]
                        so = ((yy + mvy) * plane.shape[1] + xx + mvx) * plane.itemsize
                        comp = 1 if ref_i == 0 else avg
                        lst.append(abi.ConvolveDesc(dev.ptr + off + so, d_out.ptr + (yy * W + xx) * plane.itemsize, plane.shape[1], W, bs, bs,
                                                    (C.c_int16 * 8)(*t[sx]), (C.c_int16 * 8)(*t[sy]), tx, ty, r0, r1, bd, is16, comp, fwd, bck,
                                                    (C.c_uint8 * 3)(), d_cb.ptr + (yy * W + xx) * 2, W, 0))
                        orc.orc_convolve_jnt(V(at + so), plane.shape[1], V(want.ctypes.data + (yy * W + xx) * plane.itemsize), W, bs, bs,
                                             V(t[sx].ctypes.data), tx, V(t[sy].ctypes.data), ty, r0, r1, bd, is16,
                                             V(cb.ctypes.data + (yy * W + xx) * 2), W, comp, fwd, bck)
    for lst in (first, second):
        arr = (abi.ConvolveDesc * len(lst))(*lst)
        d_desc = device.DeviceBuffer(hip, C.sizeof(arr))
        d_desc.upload(np.frombuffer(arr, np.uint8))
        device.check(hip, hip.svt_hip_convolve_batch(V(d_desc.ptr), len(lst), None), "convolve_batch")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    assert np.array_equal(d_cb.download(np.uint16, (H, W)), cb)
    assert np.array_equal(d_out.download(p0.dtype, (H, W)), want)

"""GPU parity: CDEF of libsvtav1_hip (through the C-ABI) against the oracle, bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import lf_cases as L
from lf_cases import BS, HB, P, V, VB, VL
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cdef.npz")


def test_tier_a_find_dir(hip, orc):
    rng = np.random.default_rng(1)
    for trial in range(60):
        bd = (8, 10, 12)[trial % 3]
        img = rng.integers(0, 1 << bd, size=(16, 24)).astype(np.uint16)
        if trial % 7 == 0:
            img[:] = np.arange(24, dtype=np.uint16) * ((1 << bd) // 32)
        v1, v2, v3, v4 = (C.c_int32(-1) for _ in range(4))
        d1 = orc.orc_cdef_find_dir(P(img), 24, C.byref(v1), bd - 8)
        d2 = hip.svt_aom_cdef_find_dir_hip(P(img), 24, C.byref(v2), bd - 8)
        assert (d1, v1.value) == (d2, v2.value)
        o1, o2 = C.c_uint8(9), C.c_uint8(9)
        hip.svt_aom_cdef_find_dir_dual_hip(P(img), V(img.ctypes.data + 2 * 8 * 24), 24, C.byref(v3), C.byref(v4), bd - 8,
                                           C.byref(o1), C.byref(o2))
        d5 = orc.orc_cdef_find_dir(V(img.ctypes.data + 2 * 8 * 24), 24, C.byref(v2), bd - 8)
        assert (o1.value, v3.value, o2.value, v4.value) == (d1, v1.value, d5, v2.value)


def test_tier_a_filter_block(hip, orc):
    rng = np.random.default_rng(2)
    for trial in range(160):
        bd = (8, 10)[trial % 2]
        cs = bd - 8
        tile = L.cdef_tile(rng, bd, edge=trial % 16)
        if trial % 5 == 0:
            tile = np.where(tile == VL, VL, (tile.astype(np.int32) // 64 + (1 << (bd - 1)))).astype(np.uint16)
        bsize = trial % 4
        by, bx = (0, 0) if trial % 9 == 0 else (7, 7) if trial % 9 == 1 else (int(rng.integers(0, 8)), int(rng.integers(0, 8)))
        bw, bh = 4 << (bsize in (2, 3)), 4 << (bsize in (1, 3))
        off = (VB + by * bh) * BS + HB + bx * bw
        pri, sec = int(rng.integers(0, 16)) << cs, int(rng.choice([0, 1, 2, 4])) << cs
        d = int(rng.integers(0, 8))
        damp = int(rng.integers(3, 7)) + cs
        sub = (1, 2, 4)[trial % 3] if bsize == 3 else (1, 2)[trial % 2] if bsize in (1, 2) else 1
        for is16 in (0, 1):
            o1 = np.full((8, 16), 0xAAAA if is16 else 0xAA, np.uint16 if is16 else np.uint8)
            o2 = o1.copy()
            inp = tile.ctypes.data + 2 * off
            orc.orc_cdef_filter_block(None if is16 else P(o1), P(o1) if is16 else None, 16, V(inp), pri, sec, d, damp, damp, bsize,
                                      cs, C.c_uint8(sub))
            hip.svt_cdef_filter_block_hip(None if is16 else P(o2), P(o2) if is16 else None, 16, V(inp), pri, sec, d, damp, damp,
                                          bsize, cs, C.c_uint8(sub))
            assert np.array_equal(o1, o2), (trial, is16)


def test_tier_a_dist_and_copy(hip, orc):
    rng = np.random.default_rng(3)
    orc.orc_compute_cdef_dist.restype = C.c_uint64
    hip.svt_compute_cdef_dist_16bit_hip.restype = C.c_uint64
    hip.svt_compute_cdef_dist_8bit_hip.restype = C.c_uint64
    for trial in range(80):
        is16 = trial % 2
        bd = 10 if is16 and trial % 4 == 1 else 8
        cs = bd - 8
        bsize = trial % 4
        pli = 0 if (bsize == 3 and trial % 3) else 1
        n = int(rng.integers(1, 65))
        dl = (abi.CdefList * 64)()
        for i, c in enumerate(sorted(rng.permutation(64)[:n])):
            dl[i].by, dl[i].bx = c // 8, c % 8
        dt = np.uint16 if is16 else np.uint8
        pic = rng.integers(0, 1 << bd, size=(64, 80)).astype(dt)
        packed = np.clip(pic[:64, :64].astype(np.int32) + rng.integers(-9, 10, size=(64, 64)), 0, (1 << bd) - 1).astype(dt).reshape(-1)
        sub = (1, 2, 4)[trial % 3] if bsize == 3 else 1
        a = orc.orc_compute_cdef_dist(P(pic), 80, P(packed), C.byref(dl), n, bsize, cs, pli, C.c_uint8(sub), is16)
        fn = hip.svt_compute_cdef_dist_16bit_hip if is16 else hip.svt_compute_cdef_dist_8bit_hip
        b = fn(P(pic), 80, P(packed), C.byref(dl), n, bsize, cs, pli, C.c_uint8(sub))
        assert a == b, (trial, a, b)
    src = rng.integers(0, 256, size=(9, 40)).astype(np.uint8)
    dst = np.zeros((9, 48), np.uint16)
    hip.svt_aom_copy_rect8_8bit_to_16bit_hip(P(dst), 48, P(src), 40, 9, 33)
    assert np.array_equal(dst[:, :33], src[:, :33]) and not dst[:, 33:].any()


def run_plane_gpu(hip, recon, source, w, h, is16, xdec, ydec, pli, filt, prm, fbs, damping, cs, ddir, dvar, n_fb):
    """search + apply for one plane on the GPU; returns (mse, applied plane)."""
    d_recon, d_src, d_out = (device.DeviceBuffer(hip, recon.nbytes) for _ in range(3))
    d_recon.upload(recon), d_src.upload(source)
    d_out.fill(0)
    d_filt, d_fbs = device.DeviceBuffer(hip, filt.nbytes), device.DeviceBuffer(hip, fbs.nbytes)
    d_filt.upload(filt), d_fbs.upload(fbs)
    d_mse = device.DeviceBuffer(hip, n_fb * prm.n_strengths * 8)
    d_mse.upload(np.full(n_fb * prm.n_strengths, 0xABCD, np.uint64))
    stride = recon.shape[1]
    pl = abi.CdefPlane(d_recon.ptr, d_src.ptr, stride, stride, w, h, is16, xdec, ydec, pli)
    device.check(hip, hip.svt_hip_cdef_search_plane(C.byref(pl), V(d_filt.ptr), C.byref(prm), V(d_mse.ptr), V(ddir.ptr), V(dvar.ptr), None),
                 "cdef_search")
    pl2 = abi.CdefPlane(d_recon.ptr, d_out.ptr, stride, stride, w, h, is16, xdec, ydec, pli)
    device.check(hip, hip.svt_hip_cdef_apply_plane(C.byref(pl2), V(d_filt.ptr), V(d_fbs.ptr), damping, cs, V(ddir.ptr), V(dvar.ptr), None),
                 "cdef_apply")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    return d_mse.download(np.uint64, (n_fb, prm.n_strengths)), d_out.download(recon.dtype, recon.shape)


def run_plane_orc(orc, recon, source, w, h, is16, xdec, ydec, pli, filt, prm, fbs, damping, cs, ldir, lvar, n_fb):
    stride = recon.shape[1]
    mse = np.full((n_fb, prm.n_strengths), 0xABCD, np.uint64)
    out = np.zeros_like(recon)
    pl = abi.CdefPlane(recon.ctypes.data, source.ctypes.data, stride, stride, w, h, is16, xdec, ydec, pli)
    orc.orc_cdef_search_plane(C.byref(pl), P(filt), C.byref(prm), P(mse), P(ldir), P(lvar))
    pl2 = abi.CdefPlane(recon.ctypes.data, out.ctypes.data, stride, stride, w, h, is16, xdec, ydec, pli)
    orc.orc_cdef_apply_plane(C.byref(pl2), P(filt), P(fbs), damping, cs, P(ldir), P(lvar))
    return mse, out


@pytest.mark.parametrize("lw,lh,bd,is16,sub,fmt", [
    (200, 136, 8, 0, 2, 420), (200, 136, 10, 1, 1, 420), (64, 64, 8, 0, 4, 420), (328, 72, 8, 1, 1, 444), (136, 200, 10, 1, 2, 420),
    (1920, 1080, 8, 0, 1, 420)])
def test_tier_b_picture(hip, orc, lw, lh, bd, is16, sub, fmt):
    """Whole-picture CDEF search + apply, luma then both chroma planes, against the oracle."""
    rng = np.random.default_rng(lw * 7 + lh + bd + sub)
    lw8, lh8 = (lw + 7) // 8 * 8, (lh + 7) // 8 * 8          # CDEF works on the 8-aligned picture
    w8, h8, nhfb, nvfb = lw8 // 8, lh8 // 8, (lw8 + 63) // 64, (lh8 + 63) // 64
    n_fb, cs = nhfb * nvfb, bd - 8
    dt = np.uint16 if is16 else np.uint8
    filt = (rng.random((h8, w8)) < 0.75).astype(np.uint8)
    if nhfb > 1:
        filt[0:8, 8:16] = 0
    strengths = [0, 5, 18, 35, 63, -1, 12, 1, 2, 3, 60]
    damping = 3 + int(rng.integers(0, 4))
    prm = L.search_params(strengths, damping, cs, sub)
    fbs = rng.choice(np.array([s for s in strengths if s >= 0], np.uint8), size=n_fb).astype(np.uint8)
    ldir_o, lvar_o = np.zeros((n_fb, 64), np.uint8), np.zeros((n_fb, 64), np.int32)
    ddir, dvar = device.DeviceBuffer(hip, n_fb * 64), device.DeviceBuffer(hip, n_fb * 64 * 4)
    ddir.fill(0), dvar.fill(0)
    for pli in range(3):
        xdec = int(pli > 0 and fmt != 444)
        ydec = int(pli > 0 and fmt == 420)
        w, h = lw8 >> xdec, lh8 >> ydec
        recon = L.smooth_plane(rng, w + 11, h, bd).astype(dt)
        source = np.clip(recon.astype(np.int32) + rng.integers(-6, 7, size=recon.shape), 0, (1 << bd) - 1).astype(dt)
        args = (recon, source, w, h, is16, xdec, ydec, pli, filt, prm, fbs, damping, cs)
        m1, o1 = run_plane_orc(orc, *args, ldir_o, lvar_o, n_fb)
        m2, o2 = run_plane_gpu(hip, *args, ddir, dvar, n_fb)
        assert np.array_equal(m1, m2), (pli, np.argwhere(m1 != m2)[:5])
        assert np.array_equal(o1[:, :w], o2[:, :w]), (pli, np.argwhere(o1[:, :w] != o2[:, :w])[:5])
        if pli == 0:
            assert np.array_equal(ddir.download(np.uint8, (n_fb, 64)), ldir_o)
            assert np.array_equal(dvar.download(np.int32, (n_fb, 64)), lvar_o)


def test_tier_b_golden(hip):
    """Fixture produced by the reference's own svt_cdef_filter_fb + svt_compute_cdef_dist (tests/golden/make_golden_lf.py)."""
    g = np.load(GOLD)
    for key in sorted(k[:-6] for k in g.files if k.endswith("_recon")):
        recon, source, filt = g[key + "_recon"], g[key + "_source"], g[key + "_filt"]
        meta = g[key + "_meta"]     # w, h, is16, xdec, ydec, pli, damping, cs, sub
        w, h, is16, xdec, ydec, pli, damping, cs, sub = (int(v) for v in meta)
        strengths = [int(s) for s in g[key + "_strengths"]]
        prm = L.search_params(strengths, damping, cs, sub)
        n_fb = g[key + "_mse"].shape[0]
        ddir, dvar = device.DeviceBuffer(hip, n_fb * 64), device.DeviceBuffer(hip, n_fb * 64 * 4)
        ddir.upload(g[key + "_dir"]), dvar.upload(g[key + "_var"])
        fbs = g[key + "_fbs"]
        m, o = run_plane_gpu(hip, recon, source, w, h, is16, xdec, ydec, pli, filt, prm, fbs, damping, cs, ddir, dvar, n_fb)
        assert np.array_equal(m, g[key + "_mse"]), key
        assert np.array_equal(o[:, :w], g[key + "_applied"][:, :w]), key
        if pli == 0:
            assert np.array_equal(ddir.download(np.uint8, (n_fb, 64)), g[key + "_dir"])


def test_tier_b_bad_arguments(hip):
    prm = L.search_params([0, 1], 3, 0, 1)
    pl = abi.CdefPlane(0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    assert hip.svt_hip_cdef_search_plane(C.byref(pl), None, C.byref(prm), None, None, None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_cdef_apply_plane(C.byref(pl), None, None, 3, 0, None, None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER

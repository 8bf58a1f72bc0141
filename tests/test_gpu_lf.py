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
    (1920, 1080, 8, 0, 1, 420),
    (3840, 2160, 10, 1, 1, 420)])      # BASELINE.json configs[3] size: 60 x 34 filter blocks, u16 strides > 4096 samples
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
    planes = []
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
        planes.append((recon, w, h, xdec, ydec))
    # the three planes in ONE launch (svt_hip_cdef_apply_frame), luma and chroma with their own strength per filter block
    fbs_uv = rng.choice(np.array([s for s in strengths if s >= 0], np.uint8), size=n_fb).astype(np.uint8)
    d_filt, d_fbs, d_fbs_uv = (device.DeviceBuffer(hip, a.nbytes) for a in (filt, fbs, fbs_uv))
    d_filt.upload(filt), d_fbs.upload(fbs), d_fbs_uv.upload(fbs_uv)
    arr, bufs, exp = (abi.CdefPlane * 3)(), [], []
    for pli, (recon, w, h, xdec, ydec) in enumerate(planes):
        d_in, d_out = device.DeviceBuffer(hip, recon.nbytes), device.DeviceBuffer(hip, recon.nbytes)
        d_in.upload(recon), d_out.fill(0)
        arr[pli] = abi.CdefPlane(d_in.ptr, d_out.ptr, recon.shape[1], recon.shape[1], w, h, is16, xdec, ydec, pli)
        bufs.append((d_in, d_out))
        out = np.zeros_like(recon)
        plo = abi.CdefPlane(recon.ctypes.data, out.ctypes.data, recon.shape[1], recon.shape[1], w, h, is16, xdec, ydec, pli)
        orc.orc_cdef_apply_plane(C.byref(plo), P(filt), P(fbs if pli == 0 else fbs_uv), damping, cs, P(ldir_o), P(lvar_o))
        exp.append(out)
    st = (C.c_void_p * 3)(d_fbs.ptr, d_fbs_uv.ptr, d_fbs_uv.ptr)
    device.check(hip, hip.svt_hip_cdef_apply_frame(arr, C.c_uint32(3), V(d_filt.ptr), st, damping, cs, V(ddir.ptr), V(dvar.ptr), None), "cdef_apply_frame")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    for pli, (recon, w, h, xdec, ydec) in enumerate(planes):
        got = bufs[pli][1].download(recon.dtype, recon.shape)
        assert np.array_equal(got[:, :w], exp[pli][:, :w]), ("frame", pli)


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
    assert hip.svt_hip_cdef_apply_frame(C.byref(pl), C.c_uint32(4), None, None, 3, 0, None, None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER


# ------------------------------------------------------------------------------------------------ deblocking
@pytest.mark.parametrize("d,n", [(d, n) for d in ("horizontal", "vertical") for n in (4, 6, 8, 14)])
def test_tier_a_lpf(hip, orc, d, n):
    import test_lf_oracle as TL
    rng = np.random.default_rng(50 + n + (d == "vertical"))
    for trial in range(48):
        bd = (8, 10, 8)[trial % 3]
        is16 = int(trial % 3 != 0)
        a = TL.lpf_block(rng, bd, trial)
        if d == "horizontal":
            a = a.T
        a = np.ascontiguousarray(a).astype(np.uint16 if is16 else np.uint8)
        level, sharp = int(rng.integers(0, 64)), int(rng.integers(0, 8))
        lim, mblim, hev = C.c_int(), C.c_int(), C.c_int()
        orc.orc_lf_thresholds(level, sharp, C.byref(lim), C.byref(mblim), C.byref(hev))
        th = [np.full(16, v.value, np.uint8) for v in (mblim, lim, hev)]
        b = a.copy()
        off = (8 * 16 + 4) if d == "horizontal" else (4 * 16 + 8)
        orc.orc_lpf(V(a.ctypes.data + off * a.itemsize), 16, mblim.value, lim.value, hev.value, bd, is16, n, int(d == "vertical"))
        if is16:
            getattr(hip, f"svt_aom_highbd_lpf_{d}_{n}_hip")(V(b.ctypes.data + 2 * off), 16, P(th[0]), P(th[1]), P(th[2]), bd)
        else:
            getattr(hip, f"svt_aom_lpf_{d}_{n}_hip")(V(b.ctypes.data + off), 16, P(th[0]), P(th[1]), P(th[2]))
        assert np.array_equal(a, b), (d, n, trial)


def gpu_deblock(hip, planes, w, h, flat, mi_stride, mi_rows, mi_cols, hdr, bd, is16, lvl, ps=0, pe=3):
    bufs = [device.DeviceBuffer(hip, p.nbytes) for p in planes]
    for b, p in zip(bufs, planes):
        b.upload(p)
    d_mi = device.DeviceBuffer(hip, flat.nbytes)
    d_mi.upload(flat.view(np.uint8))
    dev_planes = [(b.ptr + (L.PAD * p.shape[1] + L.PAD) * p.itemsize, p.shape[1]) for b, p in zip(bufs, planes)]
    f = L.lf_frame(dev_planes, w, h, d_mi.ptr, mi_stride, mi_rows, mi_cols, hdr, bd, is16, ps, pe, lvl)
    device.check(hip, hip.svt_hip_loop_filter_frame(C.byref(f), None), "loop_filter_frame")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    return [b.download(p.dtype, p.shape) for b, p in zip(bufs, planes)]


class OrcLvl:
    """Level table for GPU-box tests (no reference there): filter_level per plane/direction, segment/ref/mode deltas
    drawn at random — any table is a valid input, the reference's derivation of it is control-plane code."""

    @staticmethod
    def make(rng, hdr):
        lvl = rng.integers(0, 64, size=(3, 8, 2, 8, 2)).astype(np.uint8)
        lvl[:, :, :, :, :] = np.where(rng.random(lvl.shape) < 0.15, 0, lvl)
        return lvl.reshape(-1)


@pytest.mark.parametrize("variant", list(range(8)) + [10])
def test_tier_b_deblock_frame(hip, orc, variant):
    rng = np.random.default_rng(300 + variant)
    # variant 10: the configs[3] size, 3840 x 2160 10-bit (variant % 3 == 1), all three planes
    w, h = {**dict(enumerate(((200, 136), (328, 184), (64, 64), (136, 264), (196, 134), (322, 182), (1920, 1080), (130, 258)))),
            10: (3840, 2160)}[variant]
    bd, is16 = ((8, 0), (10, 1), (8, 1))[variant % 3]
    mi_cols, mi_rows = (w + 7) // 8 * 2, (h + 7) // 8 * 2
    mi_stride = mi_cols + 3
    minfo = L.random_mode_info(rng, mi_rows, mi_cols, mi_stride, sb=128 if variant == 3 else 64)
    hdr = L.lf_header(rng, variant)
    lvl = OrcLvl.make(rng, hdr)
    # flat SvtHipLfMi records without the reference: the same table lookups, restated in the test
    flat = np.zeros((mi_rows, mi_stride), abi.LF_MI_DTYPE)
    skip_inter = (minfo["skip"] != 0) & (minfo["ref_frame0"] > 0)
    flat["bsize"], flat["skip_inter"], flat["segment_id"], flat["ref_frame0"] = minfo["bsize"], skip_inter, minfo["segment_id"], minfo["ref_frame0"]
    flat["mode_lf"] = np.isin(minfo["mode"], (13, 14, 16, 17, 18, 19, 20, 21, 22, 24))
    flat["tx_size_y"] = rng.integers(0, 19, size=flat.shape)   # any TxSize: the kernel only needs its two dimensions
    flat["tx_size_uv"] = rng.integers(0, 19, size=flat.shape)
    # keep transform sizes constant inside a block and no larger than it (what a real partition guarantees)
    bw = np.array([4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 4, 16, 8, 32, 16, 64])
    bh = np.array([4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 64, 32, 64, 128, 64, 128, 16, 4, 32, 8, 64, 16])
    txw = np.array([4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64])
    txh = np.array([4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16])
    sq = {4: 0, 8: 1, 16: 2, 32: 3, 64: 4}
    for r in range(mi_rows):
        for c in range(mi_cols):
            b = int(minfo["bsize"][r, c])
            d = int(minfo["tx_depth"][r, c])
            side = max(4, min(bw[b], bh[b], 64) >> d)
            flat["tx_size_y"][r, c] = sq[side]
            flat["tx_size_uv"][r, c] = sq[max(4, min(bw[b] // 2, bh[b] // 2, 32))]
    planes = L.lf_planes(rng, mi_cols * 4, mi_rows * 4, bd, is16)
    p_orc = [p.copy() for p in planes]
    ps, pe = (0, 3) if variant != 4 else (1, 3)
    f = L.lf_frame(p_orc, w, h, flat.ctypes.data, mi_stride, mi_rows, mi_cols, hdr, bd, is16, ps, pe, lvl)
    orc.orc_loop_filter_frame(C.byref(f), 64)
    got = gpu_deblock(hip, planes, w, h, flat, mi_stride, mi_rows, mi_cols, hdr, bd, is16, lvl, ps, pe)
    changed = 0
    for a, b, o in zip(p_orc, got, planes):
        assert np.array_equal(a, b), np.argwhere(a != b)[:5]
        changed += int((a != o).sum())
    assert (changed > 0) == (variant != 7)


def test_tier_b_deblock_golden(hip):
    for key, w, h, bd, is16, mi_cols, mi_rows, mi_stride, flat, lvl, hdr, planes, want in L.golden_dlf_cases():
        got = gpu_deblock(hip, planes, w, h, flat, mi_stride, mi_rows, mi_cols, hdr, bd, is16, lvl)
        for a, b in zip(got, want):
            assert np.array_equal(a, b), key


def test_tier_b_deblock_bad_arguments(hip):
    f = abi.LfFrame()
    assert hip.svt_hip_loop_filter_frame(C.byref(f), None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_loop_filter_frame(None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER

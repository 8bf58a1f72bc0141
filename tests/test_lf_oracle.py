"""CPU: the in-loop filter oracle against the REAL reference functions (oracle/_ref RTCD pointers / _c symbols)."""
import ctypes as C

import numpy as np
import pytest

import lf_cases as L
from lf_cases import BS, HB, P, V, VB, VL
from svtav1_hip import abi


def rtcd(ref, name, restype, *argtypes):
    p = C.c_void_p.in_dll(ref, name).value
    assert p, name
    return C.CFUNCTYPE(restype, *argtypes)(p)


def test_cdef_find_dir(orc, ref):
    rng = np.random.default_rng(1)
    fn = rtcd(ref, "svt_aom_cdef_find_dir", C.c_uint8, V, C.c_int32, V, C.c_int32)
    for trial in range(300):
        bd = (8, 10, 12)[trial % 3]
        img = rng.integers(0, 1 << bd, size=(8, 24)).astype(np.uint16)
        if trial % 7 == 0:
            img[:] = np.arange(24, dtype=np.uint16) * ((1 << bd) // 32)   # pure gradient
        if trial % 11 == 0:
            img[:] = 7
        v1, v2 = C.c_int32(-1), C.c_int32(-1)
        d1 = fn(img.ctypes.data, 24, C.addressof(v1), bd - 8)
        d2 = orc.orc_cdef_find_dir(P(img), 24, C.byref(v2), bd - 8)
        assert (d1, v1.value) == (d2, v2.value)


def test_cdef_filter_block(orc, ref):
    rng = np.random.default_rng(2)
    fn = rtcd(ref, "svt_cdef_filter_block", None, V, V, C.c_int32, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
              C.c_int32, C.c_int32, C.c_uint8)
    for trial in range(600):
        bd = (8, 10)[trial % 2]
        cs = bd - 8
        tile = L.cdef_tile(rng, bd, edge=trial % 16)
        if trial % 5 == 0:   # low-contrast content: the constrain() ramps matter
            tile = np.where(tile == VL, VL, (tile.astype(np.int32) // 64 + (1 << (bd - 1)))).astype(np.uint16)
        bsize = trial % 4
        by, bx = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        bw, bh = 4 << (bsize in (2, 3)), 4 << (bsize in (1, 3))
        off = (VB + by * bh) * BS + HB + bx * bw
        pri, sec = int(rng.integers(0, 16)) << cs, int(rng.choice([0, 1, 2, 4])) << cs
        d = int(rng.integers(0, 8))
        damp = int(rng.integers(3, 7)) + cs
        sub = (1, 2, 4)[trial % 3] if bsize == 3 else (1, 2)[trial % 2] if bsize in (1, 2) else 1
        for is16 in (0, 1):
            o1 = np.full((8, 16), 0xAAAA if is16 else 0xAA, np.uint16 if is16 else np.uint8)
            o2 = o1.copy()
            flat = tile.reshape(-1)
            inp = flat.ctypes.data + 2 * off
            fn(None if is16 else o1.ctypes.data, o1.ctypes.data if is16 else None, 16, inp, pri, sec, d, damp, damp, bsize, cs, sub)
            orc.orc_cdef_filter_block(None if is16 else P(o2), P(o2) if is16 else None, 16, V(inp), pri, sec, d, damp, damp, bsize,
                                      cs, C.c_uint8(sub))
            assert np.array_equal(o1, o2), (trial, is16)


def test_cdef_dist(orc, ref):
    rng = np.random.default_rng(3)
    f16 = rtcd(ref, "svt_compute_cdef_dist_16bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    f8 = rtcd(ref, "svt_compute_cdef_dist_8bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    orc.orc_compute_cdef_dist.restype = C.c_uint64
    for trial in range(200):
        is16 = trial % 2
        bd = 10 if is16 and trial % 4 == 1 else 8
        cs = bd - 8
        bsize, pli = trial % 4, int(trial % 3 == 0 and trial % 4 == 3) * 0 + (0 if trial % 4 == 3 and trial % 3 else 1)
        n = int(rng.integers(1, 65))
        dl = (abi.CdefList * 64)()
        cells = rng.permutation(64)[:n]
        for i, c in enumerate(sorted(cells)):
            dl[i].by, dl[i].bx = c // 8, c % 8
        dt = np.uint16 if is16 else np.uint8
        pic = rng.integers(0, 1 << bd, size=(64, 80)).astype(dt)
        packed = np.clip(pic[:64, :64].astype(np.int32) + rng.integers(-9, 10, size=(64, 64)), 0, (1 << bd) - 1).astype(dt).reshape(-1)
        sub = (1, 2, 4)[trial % 3] if bsize == 3 else 1
        a = (f16 if is16 else f8)(pic.ctypes.data, 80, packed.ctypes.data, C.addressof(dl), n, bsize, cs, pli, sub)
        b = orc.orc_compute_cdef_dist(P(pic), 80, P(packed), C.byref(dl), n, bsize, cs, pli, C.c_uint8(sub), is16)
        assert a == b, (trial, a, b)


@pytest.mark.parametrize("is16,bd,pli", [(0, 8, 0), (1, 10, 0), (0, 8, 1), (1, 10, 2)])
def test_cdef_search_and_apply_vs_filter_fb(orc, ref, is16, bd, pli):
    """Per filter block, orc_cdef_search_plane / orc_cdef_apply_plane must equal the reference's svt_cdef_filter_fb
    (+ svt_compute_cdef_dist) run on a tile built the way cdef_seg_search / svt_av1_cdef_frame build it."""
    rng = np.random.default_rng(10 + pli + bd)
    xdec = ydec = int(pli != 0)
    lw, lh = 200, 136                           # luma size: partial filter blocks at right / bottom
    w, h = lw >> xdec, lh >> ydec
    cs = bd - 8
    dt = np.uint16 if is16 else np.uint8
    recon = L.smooth_plane(rng, w + 11, h, bd).astype(dt)
    source = np.clip(recon.astype(np.int32) + rng.integers(-6, 7, size=recon.shape), 0, (1 << bd) - 1).astype(dt)
    w8, h8 = (lw + 7) // 8, (lh + 7) // 8
    filt = (rng.random((h8, w8)) < 0.7).astype(np.uint8)
    filt[0:8, 8:16] = 0                          # one filter block entirely skipped
    nhfb, nvfb = (lw + 63) // 64, (lh + 63) // 64
    strengths = [0, 5, 18, 35, 63, -1, 12]
    damping, sub = 5, 2
    prm = L.search_params(strengths, damping, cs, sub)
    pl = abi.CdefPlane(recon.ctypes.data, source.ctypes.data, w + 11, w + 11, w, h, is16, xdec, ydec, pli)
    mse = np.full((nhfb * nvfb, len(strengths)), 0xABCD, np.uint64)
    # luma direction data: chroma needs the luma pass first (run it on a luma plane of the same picture)
    ldir, lvar = np.zeros((nhfb * nvfb, 64), np.uint8), np.zeros((nhfb * nvfb, 64), np.int32)
    lrecon = L.smooth_plane(np.random.default_rng(99), lw + 5, lh, bd).astype(dt)
    lpl = abi.CdefPlane(lrecon.ctypes.data, lrecon.ctypes.data, lw + 5, lw + 5, lw, lh, is16, 0, 0, 0)
    lmse = np.zeros((nhfb * nvfb, len(strengths)), np.uint64)
    orc.orc_cdef_search_plane(C.byref(lpl if pli else pl), P(filt), C.byref(prm), P(lmse if pli else mse), P(ldir), P(lvar))
    if pli:
        orc.orc_cdef_search_plane(C.byref(pl), P(filt), C.byref(prm), P(mse), P(ldir), P(lvar))
    # ---- reference, filter block by filter block
    ffb = rtcd(ref, "svt_cdef_filter_fb", None, V, V, C.c_int32, V, C.c_int32, C.c_int32, V, V, V, C.c_int32, V, C.c_int32,
               C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint8) if False else ref.svt_cdef_filter_fb
    f16 = rtcd(ref, "svt_compute_cdef_dist_16bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    f8 = rtcd(ref, "svt_compute_cdef_dist_8bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    bsz = 0 if pli else 3
    eff_sub = min(sub, 4) if bsz == 3 else 1
    bw, bh = 64 >> xdec, 64 >> ydec
    out_apply = recon.copy()
    want_apply = recon.copy()
    fbs = np.array([strengths[(i * 3 + 1) % 5] for i in range(nhfb * nvfb)], np.uint8)
    orc.orc_cdef_apply_plane(C.byref(abi.CdefPlane(recon.ctypes.data, out_apply.ctypes.data, w + 11, w + 11, w, h, is16, xdec, ydec, pli)),
                             P(filt), P(fbs), damping, cs, P(ldir), P(lvar))
    for fby in range(nvfb):
        for fbx in range(nhfb):
            fb = fby * nhfb + fbx
            dl = (abi.CdefList * 64)()
            n = 0
            for r in range(8):
                for c in range(8):
                    if fby * 8 + r < h8 and fbx * 8 + c < w8 and filt[fby * 8 + r, fbx * 8 + c]:
                        dl[n].by, dl[n].bx = r, c
                        n += 1
            if n == 0:
                assert (mse[fb] == 0xABCD).all()
                continue
            tile = np.full((64 + 2 * VB, BS), VL, np.uint16)
            for y in range(-VB, bh + VB):
                py = fby * bh + y
                if 0 <= py < h:
                    x0, x1 = max(fbx * bw - HB, 0), min(fbx * bw + bw + HB, w)
                    tile[VB + y, HB + x0 - fbx * bw:HB + x1 - fbx * bw] = recon[py, x0:x1]
            inp = tile.ctypes.data + 2 * (VB * BS + HB)
            d = np.ascontiguousarray(ldir[fb].reshape(8, 8))
            d16 = np.zeros((16, 16), np.uint8)
            d16[:8, :8] = d
            v16 = np.zeros((16, 16), np.int32)
            v16[:8, :8] = lvar[fb].reshape(8, 8)
            for gi, s in enumerate(strengths):
                if s < 0:
                    assert mse[fb, gi] == 0xABCD
                    continue
                pri, sec = s // 4, s % 4
                tmp = np.zeros(64 * 64, np.uint16)
                dirinit = C.c_int32(1)
                ref.svt_cdef_filter_fb(None if is16 else P(tmp), P(tmp) if is16 else None, 0, V(inp), xdec, ydec, P(d16), C.byref(dirinit),
                                       P(v16), pli, C.byref(dl), n, pri, sec + (sec == 3), damping, damping, cs, C.c_uint8(eff_sub))
                soff = (fby * bh) * (w + 11) + fbx * bw
                m = (f16 if is16 else f8)(source.ctypes.data + soff * source.itemsize, w + 11, tmp.ctypes.data, C.addressof(dl), n, bsz, cs, pli, eff_sub)
                assert mse[fb, gi] == m * eff_sub, (fb, gi, mse[fb, gi], m)
            # apply
            s = int(fbs[fb])
            pri, sec = s // 4, s % 4
            if pri or sec:
                dirinit = C.c_int32(1)
                dst = want_apply.ctypes.data + ((fby * bh) * (w + 11) + fbx * bw) * recon.itemsize
                ref.svt_cdef_filter_fb(None if is16 else V(dst), V(dst) if is16 else None, w + 11, V(inp), xdec, ydec, P(d16), C.byref(dirinit),
                                       P(v16), pli, C.byref(dl), n, pri, sec + (sec == 3), damping, damping, cs, C.c_uint8(1))
    assert np.array_equal(out_apply[:, :w], want_apply[:, :w])


def test_cdef_oracle_vs_golden(orc):
    """No reference needed: the committed fixture (made by the reference, tests/golden/make_golden_lf.py) pins the oracle."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cdef.npz"))
    keys = sorted(k[:-6] for k in g.files if k.endswith("_recon"))
    assert len(keys) == 3 * len(L.GOLDEN_CDEF)
    for key in keys:
        recon, source, filt = g[key + "_recon"].copy(), g[key + "_source"].copy(), g[key + "_filt"].copy()
        w, h, is16, xdec, ydec, pli, damping, cs, sub = (int(v) for v in g[key + "_meta"])
        prm = L.search_params([int(s) for s in g[key + "_strengths"]], damping, cs, sub)
        n_fb = g[key + "_mse"].shape[0]
        ldir, lvar = g[key + "_dir"].copy(), g[key + "_var"].copy()
        if pli == 0:
            ldir[:], lvar[:] = 0, 0
        mse = np.full((n_fb, prm.n_strengths), 0xABCD, np.uint64)
        out = np.zeros_like(recon)
        fbs = g[key + "_fbs"].copy()
        st = recon.shape[1]
        orc.orc_cdef_search_plane(C.byref(abi.CdefPlane(recon.ctypes.data, source.ctypes.data, st, st, w, h, is16, xdec, ydec, pli)),
                                  P(filt), C.byref(prm), P(mse), P(ldir), P(lvar))
        orc.orc_cdef_apply_plane(C.byref(abi.CdefPlane(recon.ctypes.data, out.ctypes.data, st, st, w, h, is16, xdec, ydec, pli)),
                                 P(filt), P(fbs), damping, cs, P(ldir), P(lvar))
        assert np.array_equal(mse, g[key + "_mse"]), key
        assert np.array_equal(out[:, :w], g[key + "_applied"][:, :w]), key
        assert np.array_equal(ldir, g[key + "_dir"]) and np.array_equal(lvar, g[key + "_var"]), key


# ------------------------------------------------------------------------------------------------ deblocking
LPF = [(d, n) for d in ("horizontal", "vertical") for n in (4, 6, 8, 14)]


def lpf_block(rng, bd, trial):
    """A 16x16 neighbourhood around an edge: flat, stepped or noisy so all three filter branches are hit."""
    base = int(rng.integers(16 << (bd - 8), 240 << (bd - 8)))
    a = np.full((16, 16), base, np.int32)
    kind = trial % 4
    if kind == 0:
        a += rng.integers(-1, 2, size=a.shape) * (1 << (bd - 8))
        a[:, 8:] += int(rng.integers(-3, 4)) << (bd - 8)
    elif kind == 1:
        a[:, 8:] += int(rng.integers(-12, 13)) << (bd - 8)
        a += rng.integers(-2, 3, size=a.shape) << (bd - 8)
    elif kind == 2:
        a += rng.integers(-30, 31, size=a.shape) << (bd - 8)
    else:
        a = rng.integers(0, 1 << bd, size=a.shape)
    return np.clip(a, 0, (1 << bd) - 1)


@pytest.mark.parametrize("d,n", LPF)
def test_lpf_leaves(orc, ref, d, n):
    rng = np.random.default_rng(n + (d == "vertical"))
    f8 = L.rtcd(ref, f"svt_aom_lpf_{d}_{n}", None, V, C.c_int32, V, V, V)
    f16 = L.rtcd(ref, f"svt_aom_highbd_lpf_{d}_{n}", None, V, C.c_int32, V, V, V, C.c_int32)
    for trial in range(400):
        bd = (8, 10, 8)[trial % 3]
        is16 = int(trial % 3 != 0)
        a = lpf_block(rng, bd, trial)
        if d == "horizontal":
            a = a.T
        a = np.ascontiguousarray(a).astype(np.uint16 if is16 else np.uint8)
        level, sharp = int(rng.integers(0, 64)), int(rng.integers(0, 8))
        lim, mblim, hev = C.c_int(), C.c_int(), C.c_int()
        orc.orc_lf_thresholds(level, sharp, C.byref(lim), C.byref(mblim), C.byref(hev))
        th = [np.full(16, v.value, np.uint8) for v in (mblim, lim, hev)]
        b = a.copy()
        off = (8 * 16 + 4) if d == "horizontal" else (4 * 16 + 8)
        if is16:
            f16(a.ctypes.data + 2 * off, 16, P(th[0]), P(th[1]), P(th[2]), bd)
        else:
            f8(a.ctypes.data + off, 16, P(th[0]), P(th[1]), P(th[2]))
        orc.orc_lpf(V(b.ctypes.data + off * b.itemsize), 16, mblim.value, lim.value, hev.value, bd, is16, n, int(d == "vertical"))
        assert np.array_equal(a, b), (d, n, trial)


def test_lf_thresholds(orc, ref):
    """orc_lf_thresholds against the lfthr table svt_av1_loop_filter_init builds (read back through a frame run)."""
    # indirectly covered by test_loop_filter_frame (every level/sharpness pair used there goes through the real table);
    # here: the closed form for all 64 x 8 combinations against the published AV1 formula values at the corners.
    lim, mblim, hev = C.c_int(), C.c_int(), C.c_int()
    for level, sharp, want in ((0, 0, (1, 5, 0)), (63, 0, (63, 193, 3)), (63, 7, (2, 132, 3)), (32, 4, (5, 73, 2)), (10, 5, (2, 26, 0))):
        orc.orc_lf_thresholds(level, sharp, C.byref(lim), C.byref(mblim), C.byref(hev))
        assert (lim.value, mblim.value, hev.value) == want


@pytest.mark.parametrize("variant", range(8))
def test_loop_filter_frame(orc, ref, variant):
    """Whole-frame deblocking: oracle vs the REAL svt_av1_loop_filter_frame on random partitions."""
    rng = np.random.default_rng(100 + variant)
    w, h = ((200, 136), (328, 184), (64, 64), (136, 264), (196, 134), (322, 182), (264, 200), (130, 258))[variant]
    bd, is16 = ((8, 0), (10, 1), (8, 1))[variant % 3]
    sb = 128 if variant == 6 else 64
    mi_cols, mi_rows = (w + 7) // 8 * 2, (h + 7) // 8 * 2
    mi_stride = mi_cols + 3
    minfo = L.random_mode_info(rng, mi_rows, mi_cols, mi_stride, sb=sb)
    hdr = L.lf_header(rng, variant)
    planes = L.lf_planes(rng, mi_cols * 4, mi_rows * 4, bd, is16)
    p_ref = [p.copy() for p in planes]
    p_orc = [p.copy() for p in planes]
    ps, pe = (0, 3) if variant != 4 else (1, 3)
    flat, lvl = L.ref_deblock(ref, p_ref, w, h, minfo, mi_stride, mi_rows, mi_cols, hdr, bd, is16, sb, ps, pe)
    f = L.lf_frame(p_orc, w, h, flat.ctypes.data, mi_stride, mi_rows, mi_cols, hdr, bd, is16, ps, pe, lvl)
    orc.orc_loop_filter_frame(C.byref(f), sb)
    changed = 0
    for a, b, o in zip(p_ref, p_orc, planes):
        assert np.array_equal(a, b), np.argwhere(a != b)[:5]
        changed += int((a != o).sum())
    assert (changed > 0) == (variant != 7)


def test_loop_filter_oracle_vs_golden(orc):
    """No reference needed: tests/golden/dlf.npz (made by the real svt_av1_loop_filter_frame) pins the oracle."""
    n = 0
    for key, w, h, bd, is16, mi_cols, mi_rows, mi_stride, flat, lvl, hdr, planes, want in L.golden_dlf_cases():
        f = L.lf_frame(planes, w, h, flat.ctypes.data, mi_stride, mi_rows, mi_cols, hdr, bd, is16, 0, 3, lvl)
        orc.orc_loop_filter_frame(C.byref(f), 64)      # the schedule's SB size does not change the result
        for a, b in zip(planes, want):
            assert np.array_equal(a, b), key
        n += 1
    assert n == len(L.GOLDEN_DLF)

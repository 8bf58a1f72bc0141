"""Shared case builders for the in-loop filter parity tests (test infrastructure)."""
import ctypes as C

import numpy as np

from svtav1_hip import abi

BS, VL, VB, HB = 144, 0x7F7F, 3, 8
V = C.c_void_p


def P(a):
    return V(a.ctypes.data)


def cdef_tile(rng, bd, edge=0):
    """A CDEF input tile (u16, stride 144, 64+6 rows) as cdef_seg_search builds it; `edge` bit-mask puts
    CDEF_VERY_LARGE on the top/left/bottom/right borders like at picture boundaries."""
    t = rng.integers(0, 1 << bd, size=(64 + 2 * VB, BS)).astype(np.uint16)
    if edge & 1:
        t[:VB, :] = VL
    if edge & 2:
        t[:, :HB] = VL
    if edge & 4:
        t[VB + 64:, :] = VL
    if edge & 8:
        t[:, HB + 64:] = VL
    return t


def smooth_plane(rng, w, h, bd):
    """Picture-like content (edges + gradients + noise) so that CDEF directions and clamps are exercised."""
    y, x = np.mgrid[0:h, 0:w]
    img = 0.5 * (1 << bd) + 0.3 * (1 << bd) * np.sin(x / 9.0 + y / 17.0) + 0.15 * (1 << bd) * ((x // 16 + y // 24) % 2)
    img += rng.normal(0, (1 << bd) / 64.0, size=(h, w))
    return np.clip(np.rint(img), 0, (1 << bd) - 1)


def cdef_plane(recon, source, pli, xdec, ydec, is16):
    return abi.CdefPlane(recon.ctypes.data, source.ctypes.data, recon.strides[0] // recon.itemsize,
                         source.strides[0] // source.itemsize, recon.shape[1] if False else 0, 0, is16, xdec, ydec, pli)


def search_params(strengths, damping, coeff_shift, sub):
    p = abi.CdefSearchParams()
    p.n_strengths = len(strengths)
    for i, s in enumerate(strengths):
        p.strengths[i] = s
    p.pri_damping = p.sec_damping = damping
    p.coeff_shift, p.subsampling_factor = coeff_shift, sub
    return p


def rtcd(ref, name, restype, *argtypes):
    p = C.c_void_p.in_dll(ref, name).value
    assert p, name
    return C.CFUNCTYPE(restype, *argtypes)(p)


def ref_cdef_plane(ref, recon, source, w, h, is16, xdec, ydec, pli, filt, strengths, fbs, damping, cs, sub, ldir, lvar):
    """One plane through the REFERENCE's own per-filter-block functions (svt_aom_cdef_find_dir, svt_cdef_filter_fb,
    svt_compute_cdef_dist_*), tiles built as cdef_seg_search (cdef_process.c:204-221) builds them.
    Luma fills ldir/lvar [n_fb][64]; chroma reads them.  Returns (mse[n_fb][n_strengths], applied plane)."""
    lw, lh = w << xdec, h << ydec
    w8, h8, nhfb, nvfb = (lw + 7) // 8, (lh + 7) // 8, (lw + 63) // 64, (lh + 63) // 64
    stride = recon.shape[1]
    fdir = rtcd(ref, "svt_aom_cdef_find_dir", C.c_uint8, V, C.c_int32, V, C.c_int32)
    f16 = rtcd(ref, "svt_compute_cdef_dist_16bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    f8 = rtcd(ref, "svt_compute_cdef_dist_8bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    bsz = 3 if (xdec, ydec) == (0, 0) else 0 if (xdec, ydec) == (1, 1) else 1 if xdec else 2
    eff_sub = min(sub, 4) if bsz == 3 else 1 if bsz == 0 else min(sub, 2)
    bw, bh = 64 >> xdec, 64 >> ydec
    mse = np.full((nhfb * nvfb, len(strengths)), 0xABCD, np.uint64)
    applied = np.zeros_like(recon)
    applied[:, :w] = recon[:, :w]
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
                continue
            tile = np.full((64 + 2 * VB, BS), VL, np.uint16)
            for y in range(-VB, bh + VB):
                py = fby * bh + y
                if 0 <= py < h:
                    x0, x1 = max(fbx * bw - HB, 0), min(fbx * bw + bw + HB, w)
                    tile[VB + y, HB + x0 - fbx * bw:HB + x1 - fbx * bw] = recon[py, x0:x1]
            inp = tile.ctypes.data + 2 * (VB * BS + HB)
            if pli == 0:
                for i in range(n):
                    v = C.c_int32(0)
                    ldir[fb, dl[i].by * 8 + dl[i].bx] = fdir(inp + 2 * (8 * dl[i].by * BS + 8 * dl[i].bx), BS, C.addressof(v), cs)
                    lvar[fb, dl[i].by * 8 + dl[i].bx] = v.value
            d16, v16 = np.zeros((16, 16), np.uint8), np.zeros((16, 16), np.int32)
            d16[:8, :8], v16[:8, :8] = ldir[fb].reshape(8, 8), lvar[fb].reshape(8, 8)
            soff = (fby * bh) * stride + fbx * bw
            for gi, s in enumerate(strengths):
                if s < 0:
                    continue
                pri, sec = s // 4, s % 4
                tmp = np.zeros(64 * 64, np.uint16)
                dirinit = C.c_int32(1)
                ref.svt_cdef_filter_fb(None if is16 else P(tmp), P(tmp) if is16 else None, 0, V(inp), xdec, ydec, P(d16), C.byref(dirinit),
                                       P(v16), pli, C.byref(dl), n, pri, sec + (sec == 3), damping, damping, cs, C.c_uint8(eff_sub))
                m = (f16 if is16 else f8)(source.ctypes.data + soff * source.itemsize, stride, tmp.ctypes.data, C.addressof(dl), n, bsz, cs,
                                          pli, eff_sub)
                mse[fb, gi] = m * eff_sub
            s = int(fbs[fb])
            pri, sec = s // 4, s % 4
            if pri or sec:
                dirinit = C.c_int32(1)
                dst = applied.ctypes.data + soff * recon.itemsize
                ref.svt_cdef_filter_fb(None if is16 else V(dst), V(dst) if is16 else None, stride, V(inp), xdec, ydec, P(d16),
                                       C.byref(dirinit), P(v16), pli, C.byref(dl), n, pri, sec + (sec == 3), damping, damping, cs,
                                       C.c_uint8(1))
    return mse, applied


GOLDEN_CDEF = [  # key, luma w, luma h, bd, is16, fmt, sub, seed
    ("a_420_8", 200, 136, 8, 0, 420, 2, 5), ("b_420_10", 136, 72, 10, 1, 420, 1, 6), ("c_444_8in16", 72, 136, 8, 1, 444, 4, 7),
    ("d_420_10_tall", 72, 200, 10, 1, 420, 4, 8)]


def golden_cdef_inputs(lw, lh, bd, is16, fmt, sub, seed):
    """Seeded inputs of one golden picture: yields per plane (pli, xdec, ydec, w, h, recon, source) + shared parameters."""
    rng = np.random.default_rng(seed)
    w8, h8, nhfb, nvfb = lw // 8, lh // 8, (lw + 63) // 64, (lh + 63) // 64
    filt = (rng.random((h8, w8)) < 0.75).astype(np.uint8)
    strengths = [0, 5, 18, 35, 63, -1, 12, 2]
    damping = 3 + seed % 4
    fbs = rng.choice(np.array([s for s in strengths if s >= 0], np.uint8), size=nhfb * nvfb).astype(np.uint8)
    dt = np.uint16 if is16 else np.uint8
    planes = []
    for pli in range(3):
        xdec, ydec = int(pli > 0 and fmt != 444), int(pli > 0 and fmt == 420)
        w, h = lw >> xdec, lh >> ydec
        recon = smooth_plane(rng, w + 11, h, bd).astype(dt)
        source = np.clip(recon.astype(np.int32) + rng.integers(-6, 7, size=recon.shape), 0, (1 << bd) - 1).astype(dt)
        planes.append((pli, xdec, ydec, w, h, recon, source))
    return filt, strengths, damping, fbs, planes

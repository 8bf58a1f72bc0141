#!/usr/bin/env python3
"""Golden vectors for CDEF, produced by the REAL reference functions (oracle/_ref: svt_aom_cdef_find_dir,
svt_cdef_filter_fb, svt_compute_cdef_dist_*) on seeded pictures (tests/lf_cases.py::golden_cdef_inputs).
Inputs are stored too so that the GPU test needs nothing but this file."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lf_cases as L  # noqa: E402
import pyorc  # noqa: E402

ref = pyorc.ref()
store = {}
for key, lw, lh, bd, is16, fmt, sub, seed in L.GOLDEN_CDEF:
    filt, strengths, damping, fbs, planes = L.golden_cdef_inputs(lw, lh, bd, is16, fmt, sub, seed)
    n_fb = ((lw + 63) // 64) * ((lh + 63) // 64)
    ldir, lvar = np.zeros((n_fb, 64), np.uint8), np.zeros((n_fb, 64), np.int32)
    for pli, xdec, ydec, w, h, recon, source in planes:
        mse, applied = L.ref_cdef_plane(ref, recon, source, w, h, is16, xdec, ydec, pli, filt, strengths, fbs, damping, bd - 8, sub,
                                        ldir, lvar)
        k = f"{key}_p{pli}"
        store[k + "_recon"], store[k + "_source"], store[k + "_filt"] = recon, source, filt
        store[k + "_meta"] = np.array([w, h, is16, xdec, ydec, pli, damping, bd - 8, sub], np.int32)
        store[k + "_strengths"] = np.array(strengths, np.int8)
        store[k + "_fbs"], store[k + "_mse"], store[k + "_applied"] = fbs, mse, applied
        store[k + "_dir"], store[k + "_var"] = ldir.copy(), lvar.copy()
np.savez_compressed(os.path.join(HERE, "cdef.npz"), **store)
print("cdef.npz:", len(store), "arrays", os.path.getsize(os.path.join(HERE, "cdef.npz")), "bytes")

# ---- deblocking: the REAL svt_av1_loop_filter_frame on seeded pictures / partitions (lf_cases.golden_dlf_inputs)
store = {}
for key, w, h, bd, is16, variant, sb, seed in L.GOLDEN_DLF:
    mi_cols, mi_rows, mi_stride, minfo, hdr, planes = L.golden_dlf_inputs(w, h, bd, is16, variant, sb, seed)
    out = [p.copy() for p in planes]
    flat, lvl = L.ref_deblock(ref, out, w, h, minfo, mi_stride, mi_rows, mi_cols, hdr, bd, is16, sb)
    store[key + "_mi"], store[key + "_lvl"] = flat.view(np.uint8).reshape(mi_rows, mi_stride, 8), lvl
    store[key + "_meta"] = np.array([w, h, bd, is16, mi_cols, mi_rows, mi_stride, hdr.filter_level[0], hdr.filter_level[1],
                                     hdr.filter_level_u, hdr.filter_level_v, hdr.sharpness_level], np.int32)
    for i in range(3):
        store[f"{key}_in{i}"], store[f"{key}_out{i}"] = planes[i], out[i]
np.savez_compressed(os.path.join(HERE, "dlf.npz"), **store)
print("dlf.npz:", len(store), "arrays", os.path.getsize(os.path.join(HERE, "dlf.npz")), "bytes")

# ---- self-guided restoration: the REAL search driver (through oracle/ref_harness_sgr.c), filter and apply
import ctypes as C  # noqa: E402
import sgr_cases as G  # noqa: E402
from lf_cases import P, V  # noqa: E402

store = {}
filt = L.rtcd(ref, "svt_av1_selfguided_restoration", None, V, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32)
appl = L.rtcd(ref, "svt_apply_selfguided_restoration", None, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, V, C.c_int32, C.c_int32)
tmp = np.zeros(2 * 161 * 1024, np.int32)
for key, w, h, bd, is16, kind, pu, (s0, s1, inc, refine), seed in G.GOLDEN_SGR:
    rng = np.random.default_rng(seed)
    dat, src = G.sgr_plane(rng, w, h, bd, is16, kind)
    enc = (lambda a: V(a >> 1)) if is16 else V
    out = np.zeros(3, np.int32)
    assert ref.ref_sgr_search_unit(enc(G.at(dat)), w, h, dat.shape[1], enc(G.at(src)), src.shape[1], is16, bd, pu, pu, s0, s1, inc, refine, P(out)) == 0
    # filter + apply of the winning candidate on the first processing unit
    pw, ph = min(pu, w), min(pu, h)
    f0, f1 = np.zeros((ph, pw), np.int32), np.zeros((ph, pw), np.int32)
    filt(enc(G.at(dat)), pw, ph, dat.shape[1], P(f0), P(f1), pw, int(out[0]), bd, is16)
    rec = np.zeros((ph, pw), dat.dtype)
    xqd = np.array([out[1], out[2]], np.int32)
    appl(enc(G.at(dat)), pw, ph, dat.shape[1], int(out[0]), P(xqd), enc(rec.ctypes.data), pw, P(tmp), bd, is16)
    store[key + "_dat"], store[key + "_src"], store[key + "_best"] = dat, src, out
    store[key + "_flt0"], store[key + "_flt1"], store[key + "_rec"] = f0, f1, rec
np.savez_compressed(os.path.join(HERE, "sgr.npz"), **store)
print("sgr.npz:", len(store), "arrays", os.path.getsize(os.path.join(HERE, "sgr.npz")), "bytes")

# ---- Wiener restoration: svt_av1_compute_stats(_highbd) and svt_av1_(highbd_)wiener_convolve_add_src of the reference
import test_wiener_oracle as TW  # noqa: E402

store = {}
f8 = L.rtcd(ref, "svt_av1_compute_stats", None, C.c_int32, V, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, V, V)
f16 = L.rtcd(ref, "svt_av1_compute_stats_highbd", None, C.c_int32, V, V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32)
c8 = L.rtcd(ref, "svt_av1_wiener_convolve_add_src", None, V, C.c_ssize_t, V, C.c_ssize_t, V, V, C.c_int32, C.c_int32, V)
c16 = L.rtcd(ref, "svt_av1_highbd_wiener_convolve_add_src", None, V, C.c_ssize_t, V, C.c_ssize_t, V, V, C.c_int32, C.c_int32, V, C.c_int32)
for key, bd, is16, win, w, h, seed in G.GOLDEN_WIENER:
    rng = np.random.default_rng(seed)
    dat, src = G.sgr_plane(rng, w + 4, h + 4, bd, is16, 0)
    enc = (lambda a: V(a >> 1)) if is16 else V
    M, H = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
    if is16:
        f16(win, enc(G.at(dat)), enc(G.at(src)), 1, 1 + w, 2, 2 + h, dat.shape[1], src.shape[1], P(M), P(H), bd)
    else:
        f8(win, enc(G.at(dat)), enc(G.at(src)), 1, 1 + w, 2, 2 + h, dat.shape[1], src.shape[1], P(M), P(H))
    fx, kx = G.wiener_filter(rng)
    fy, ky = G.wiener_filter(rng)
    r0, r1 = G.wiener_rounds(bd)
    cp = TW.ConvolveParams(round_0=r0, round_1=r1)
    cw, ch = min(w, 128), min(h, 128)
    out = np.zeros((ch, cw), dat.dtype)
    if is16:
        c16(enc(G.at(dat)), dat.shape[1], enc(out.ctypes.data), cw, P(fx), P(fy), cw, ch, C.byref(cp), bd)
    else:
        c8(enc(G.at(dat)), dat.shape[1], P(out), cw, P(fx), P(fy), cw, ch, C.byref(cp))
    store[key + "_dat"], store[key + "_src"], store[key + "_M"], store[key + "_H"] = dat, src, M, H
    store[key + "_fx"], store[key + "_fy"], store[key + "_out"] = np.array(fx), np.array(fy), out
np.savez_compressed(os.path.join(HERE, "wiener.npz"), **store)
print("wiener.npz:", len(store), "arrays", os.path.getsize(os.path.join(HERE, "wiener.npz")), "bytes")

# ---- single-reference interpolation: svt_av1_(highbd_)convolve_{2d,x,y,2d_copy}_sr of the reference
import conv_cases as K  # noqa: E402

store, n = {}, 0
rng = np.random.default_rng(77)
tabs = {nm: K.kernel_table(nm) for nm in K.TABLES}
sig = (V, C.c_int32, V, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, C.c_int32, V)
MODES = ("2d_sr", "x_sr", "y_sr", "2d_copy_sr")
for bd, is16 in ((8, 0), (10, 1)):
    r0, r1 = K.conv_rounds(bd)
    for k in range(8):
        w, h = K.SIZES[(k * 3) % len(K.SIZES)]
        mode, ti = k % 4, k % 3
        sx, sy = int(rng.integers(1, 16)), int(rng.integers(1, 16))
        plane, at = K.ref_plane(rng, w, h, bd, is16, 0)
        tab = tabs[list(K.TABLES)[ti]][0]
        fp = K.InterpFilterParams(tab.ctypes.data, 8, 16, ti)
        cp = TW.ConvolveParams(round_0=r0, round_1=r1)
        o = np.zeros((h, w), plane.dtype)
        fn = L.rtcd(ref, (f"svt_av1_highbd_convolve_{MODES[mode]}" if is16 else f"svt_av1_convolve_{MODES[mode]}"), None,
                    *(sig + ((C.c_int32,) if is16 else ())))
        fn(*([V(at), plane.shape[1], P(o), w, w, h, C.byref(fp), C.byref(fp), sx, sy, C.byref(cp)] + ([bd] if is16 else [])))
        store[f"c{n}_meta"], store[f"c{n}_plane"], store[f"c{n}_out"] = np.array([bd, is16, w, h, mode, ti, sx, sy], np.int32), plane, o
        n += 1
store["n"] = np.array(n)
np.savez_compressed(os.path.join(HERE, "convolve.npz"), **store)
print("convolve.npz:", n, "cases", os.path.getsize(os.path.join(HERE, "convolve.npz")), "bytes")

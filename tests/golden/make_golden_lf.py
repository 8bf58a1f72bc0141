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

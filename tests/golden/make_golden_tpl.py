#!/usr/bin/env python3
"""Golden vectors for the residual producer and the TPL block cost, produced by the REAL reference functions
(oracle/_ref).  Inputs are seeded (tests/test_residual_oracle.py::subtract_cases / tpl_cases); only expected outputs are
stored."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyorc  # noqa: E402
import test_residual_oracle as TR  # noqa: E402
from tx_cases import V  # noqa: E402

ref = pyorc.ref()
PD = C.c_ssize_t
sub = TR.rtcd(ref, "svt_aom_subtract_block", None, C.c_int, C.c_int, V, PD, V, PD, V, PD)
hsub = TR.rtcd(ref, "svt_aom_highbd_subtract_block", None, C.c_int, C.c_int, V, PD, V, PD, V, PD, C.c_int)
store = {}
for i, (rows, cols, ds, s, p, hbd) in enumerate(TR.subtract_cases()):
    d = np.zeros((rows, ds), np.int16)
    if hbd:
        hsub(rows, cols, d.ctypes.data, ds, s.ctypes.data, s.shape[1], p.ctypes.data, p.shape[1], 10)
    else:
        sub(rows, cols, d.ctypes.data, ds, s.ctypes.data, s.shape[1], p.ctypes.data, p.shape[1])
    store[f"sub{i}"] = d[:, :cols].copy()
store["tpl_cost"] = np.array([TR.ref_tpl_cost(ref, *c) for c in TR.tpl_cases()], np.int64)
np.savez_compressed(os.path.join(HERE, "tpl_cost.npz"), **store)
print("tpl_cost.npz:", len(store), "arrays")

#!/usr/bin/env python3
"""Golden vectors for the transforms, produced by the REAL reference functions (oracle/_ref RTCD pointers).
Inputs are seeded (tests/test_txfm_oracle.py::golden_cases); only expected outputs are stored."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyorc  # noqa: E402
import test_txfm_oracle as TT  # noqa: E402
from tx_cases import V  # noqa: E402

ref, orc = pyorc.ref(), pyorc.oracle()
store = {}
for i, (w, h, tt, bd, res, pred) in enumerate(TT.golden_cases()):
    if not orc.orc_txfm_valid(w, h, tt):
        continue
    co = np.zeros(w * h, np.int32)
    TT.rtcd(ref, f"svt_av1_fwd_txfm2d_{w}x{h}", None, V, V, C.c_uint32, C.c_int, C.c_uint8)(res.ctypes.data, co.ctypes.data, w + 3, tt, bd)
    store[f"fwd{i}"] = co
    ci = co.reshape(h, w)[:min(h, 32), :min(w, 32)].copy().reshape(-1)
    rec = np.zeros((h, w + 7), np.uint16)
    TT.ref_inverse(ref, w, h, ci, pred, w + 5, rec, w + 7, tt, bd)
    store[f"inv{i}"] = rec
np.savez_compressed(os.path.join(HERE, "txfm.npz"), **store)
print("txfm.npz:", len(store), "arrays")

#!/usr/bin/env python3
"""Golden vectors for the compound interpolation family, produced by the REAL reference functions (oracle/_ref RTCD
pointers svt_av1_(highbd_)jnt_convolve_*).  Inputs are seeded (tests/conv_cases.py::jnt_cases); only the expected
conv-buffer contents after the first prediction and the pixels after the second are stored."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyorc  # noqa: E402
import conv_cases as K  # noqa: E402
import lf_cases as L  # noqa: E402
import test_convolve_oracle as T  # noqa: E402
from lf_cases import V  # noqa: E402

ref = pyorc.ref()
store, k = {}, 0
sig8 = (V, C.c_int32, V, C.c_int32, C.c_int32, C.c_int32, V, V, C.c_int32, C.c_int32, V)
for bd, is16 in ((8, 0), (10, 1)):
    fns = [L.rtcd(ref, (f"svt_av1_highbd_jnt_convolve_{m}" if is16 else f"svt_av1_jnt_convolve_{m}"), None,
                  *(sig8 + ((C.c_int32,) if is16 else ()))) for m in K.JNT_MODES]
    for c in K.jnt_cases(bd, is16, n=16, seed=1):
        f, o = T.run_fn_jnt(fns, c, bd, is16, T.ConvolveParams, K.InterpFilterParams)
        store[f"first{k}"], store[f"out{k}"] = f[:, :c[0]].copy(), o[:, :c[0]].copy()
        k += 1
np.savez_compressed(os.path.join(HERE, "convolve_jnt.npz"), **store)
print("convolve_jnt.npz:", k, "cases", os.path.getsize(os.path.join(HERE, "convolve_jnt.npz")), "bytes")

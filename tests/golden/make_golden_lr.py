#!/usr/bin/env python3
"""Generates tests/golden/lr_frame.npz: outputs of the REAL svt_av1_loop_restoration_filter_unit + svt_extend_frame
(through oracle/ref_harness_lr.c) for the cases of tests/lr_cases.py.  Needs oracle/_ref/libsvtref.so (this container)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import lr_cases as R  # noqa: E402
import pyorc  # noqa: E402

ref = pyorc.ref()
out = {}
for name in R.CASES:
    case = R.make_case(name)
    arr, outs = R.lr_planes(case)
    assert ref.ref_restoration_filter_frame(arr, C.c_uint32(len(case))) == 0
    for p, (o, c) in enumerate(zip(outs, case)):
        out[f"{name}_p{p}"] = o[:c["h"], :c["w"]].copy()
np.savez_compressed(os.path.join(HERE, "lr_frame.npz"), **out)
print("wrote", len(out), "planes")

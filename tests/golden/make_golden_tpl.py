#!/usr/bin/env python3
"""Generates tests/golden/tpl_frame.npz: TPL reconstruction pictures, TplStats and TplSrcStats the REAL
tpl_mc_flow_dispenser_sb_generic of the reference (oracle/ref_harness_tpl.c::ref_tpl_dispenser_frame) produces for
tests/tpl_cases.py::CASES, plus the reference's 8-bit quantiser scalars at the qindex values used.  Needs oracle/_ref."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyorc  # noqa: E402
import tpl_cases as T  # noqa: E402

ref, orc = pyorc.ref(), pyorc.oracle()
out = {}
for case in T.CASES:
    q = (C.c_int16 * 6)()
    ref.ref_tpl_quant(case[4], q)
    T.QUANT[case[4]] = list(q)
    out[f"quant_{case[4]}"] = np.array(list(q), np.int16)
for case in T.CASES:
    s = T.TplScene(orc, case)
    if case[5]["src_data_ready"]:
        T.prime_second_pass(orc, s)
    assert ref.ref_tpl_dispenser_frame(C.byref(s.job()), case[4]) == 0
    for k, v in s.results().items():
        out[f"{case[0]}_{k}"] = v.copy()
np.savez_compressed(os.path.join(HERE, "tpl_frame.npz"), **out)
print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "tpl_frame.npz")) // 1024, "KiB")

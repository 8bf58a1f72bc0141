#!/usr/bin/env python3
"""Generates tests/golden/tf_picture.npz: the centre pictures the REAL produce_temporally_filtered_pic of the reference
(oracle/ref_harness_tfme.c::ref_tf_picture) produces for tests/tf_picture_cases.py::CASES, with the decay factors it derived and
its horizontal / vertical block counters.  Needs oracle/_ref/libsvtref.so.  Only the interior the filter can touch is stored
(the 64-aligned area), as uint8 / uint16 arrays."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyorc  # noqa: E402
import tf_picture_cases as tpc  # noqa: E402

ref, orc = pyorc.ref(), pyorc.oracle()
out = {}
for case in tpc.CASES:
    name = case[0]
    pics = tpc.case_window(orc, case)
    decay, tot = tpc.run_reference(ref, pics, case)
    out[f"{name}_decay"] = np.array(decay, np.uint32)
    out[f"{name}_tot"] = np.array(tot, np.uint32)
    for k, v in pics[0].arrays().items():
        out[f"{name}_{k}"] = v
np.savez_compressed(os.path.join(HERE, "tf_picture.npz"), **out)
print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "tf_picture.npz")) // 1024, "KiB")

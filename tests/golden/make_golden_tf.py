#!/usr/bin/env python3
"""Golden vectors for the temporal filter's accumulate stage, produced by the REAL reference functions through
oracle/ref_harness_tf.c.  Inputs are seeded (tests/tf_cases.py::block_case); only the accumulator / counter planes after
the call are stored."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyorc  # noqa: E402
import tf_cases as F  # noqa: E402

ref = pyorc.ref()
store, k = {}, 0
for bd in (8, 10):
    for trial in range(8):
        b, a = F.block_case(trial, bd, seed=3)
        assert ref.ref_tf_block_accumulate(C.byref(b)) == 0
        for pl in range(3):
            n = a[pl][0].shape[0]          # the block region; everything beyond it must stay untouched (checked against the input)
            store[f"acc{k}_{pl}"], store[f"cnt{k}_{pl}"] = a[pl][2][:, :n].copy(), a[pl][3][:, :n].copy()
        k += 1
np.savez_compressed(os.path.join(HERE, "tf.npz"), **store)
print("tf.npz:", k, "cases", os.path.getsize(os.path.join(HERE, "tf.npz")), "bytes")

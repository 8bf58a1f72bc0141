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

# noise estimate: the reference's dispatch pointers on the seeded planes of tf_cases.noise_cases(); only the results are stored
from test_txfm_oracle import rtcd  # noqa: E402
V = C.c_void_p
lo = rtcd(ref, "svt_estimate_noise_fp16", C.c_int32, V, C.c_uint16, C.c_uint16, C.c_uint16)
hi = rtcd(ref, "svt_estimate_noise_highbd_fp16", C.c_int32, V, C.c_int, C.c_int, C.c_int, C.c_int)
noise = [lo(img.ctypes.data, w, h, stride) if bd == 8 else hi(img.ctypes.data, w, h, stride, bd) for img, w, h, stride, bd in F.noise_cases()]
np.savez_compressed(os.path.join(HERE, "tf_noise.npz"), noise=np.array(noise, np.int64))
print("tf_noise.npz:", len(noise), "values", noise[:8])

#!/usr/bin/env python3
"""Tuning aid: milliseconds per svt_hip_wiener_stats call on the bench's 4K 10-bit luma plane for the library named by SVTAV1_HIP_LIB."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-mod-by-patman_amd"))


def main():
    import torch
    import bench
    from benchlib import lf_inputs as LB
    from svtav1_hip import abi
    lib = abi.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    assert lib.svt_hip_init(0) == 0
    inp = LB.build(lib, dev, np.random.default_rng(11), 3840, 2160, 10, torch)
    stream = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(stream.cuda_stream)
    ms = [round(bench.timed_launches(stream, 10, 2, lambda: LB.run_wiener_stats(lib, inp, sp)), 4) for _ in range(3)]
    print(os.path.basename(os.environ.get("SVTAV1_HIP_LIB", "libsvtav1_hip.so")), inp["n_wiener"], "units", ms, flush=True)


if __name__ == "__main__":
    main()

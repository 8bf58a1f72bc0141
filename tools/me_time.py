#!/usr/bin/env python3
"""Tuning aid: milliseconds per me_b64_kernel launch of the headline workload (16 x 4K x 5 references) for the library named by
SVTAV1_HIP_LIB (default: the in-tree build); several builds are compared by running this once per build in ONE gpurun call."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-mod-by-patman_amd"))


def main():
    import torch
    import bench
    from svtav1_hip import abi
    lib = abi.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    assert lib.svt_hip_init(0) == 0
    mw = bench.MeWorkload(lib, dev, 3840, 2160, 16, "m8_4k_tl2", (-1, -2, -3), (1, 2), seed=7)
    stream = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(stream.cuda_stream)
    mw.analysis(sp)
    ms = [round(bench.timed_launches(stream, 5, 2, lambda: mw.me(sp)), 4) for _ in range(4)]
    print(os.path.basename(os.environ.get("SVTAV1_HIP_LIB", "libsvtav1_hip.so")), ms, flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Tuning aid: time of me_b64_kernel truncated behind each stage (needs an ABLATE=1 build of the library, picked up through
SVTAV1_HIP_LIB).  Differences between consecutive rows = what each stage costs with the real overlap between workgroups."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-mod-by-patman_amd"))
STAGES = ["init+sources", "zz_sad", "pre-HME", "HME L0", "HME L1 (+L2)", "centre/prune", "full-pel", "me_prune", "all"]


def main():
    import torch
    import bench
    from svtav1_hip import abi
    lib = abi.load()
    assert hasattr(lib, "svt_hip_debug_me_stop"), "library was not built with ABLATE=1"
    big = "--1080" not in sys.argv
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    assert lib.svt_hip_init(0) == 0
    mw = bench.MeWorkload(lib, dev, 3840 if big else 1920, 2160 if big else 1080, 16, "m8_4k_tl2" if big else "m8_1080p_tl2",
                          (-1, -2, -3) if big else (-1, -2), (1, 2), seed=7)
    stream = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(stream.cuda_stream)
    mw.analysis(sp)
    out, prev = {}, 0.0
    for k in list(range(8)) + [99]:
        assert lib.svt_hip_debug_me_stop(k) == 0
        ms = bench.timed_launches(stream, 10, 2, lambda: mw.me(sp))
        out[STAGES[min(k, 8)]] = {"cumulative_ms": round(ms, 4), "stage_ms": round(ms - prev, 4)}
        prev = ms
    print(json.dumps({"me_ablation": out, "width": mw.W}))


if __name__ == "__main__":
    main()

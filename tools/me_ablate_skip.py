#!/usr/bin/env python3
"""Tuning aid (ABLATE=1 library through SVTAV1_HIP_LIB): me_b64_kernel truncated behind each search stage, with the window
staging loads and / or the SAD arithmetic of wg_multi_search left out.  3 launches per (stop, mask) in a fixed order, so that
a rocprofv3 --pmc run over this script can be decoded by dispatch order (tools/me_ablate_skip_pmc.sh)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-mod-by-patman_amd"))
STOPS = [int(x) for x in os.environ.get("ME_ABLATE_STOPS", "1,2,3,4,99").split(",")]
MASKS = [int(x) for x in os.environ.get("ME_ABLATE_MASKS", "0,1,2,3,7").split(",")]


def main():
    import torch
    import bench
    from svtav1_hip import abi
    lib = abi.load()
    assert hasattr(lib, "svt_hip_debug_me_stop"), "library was not built with ABLATE=1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    assert lib.svt_hip_init(0) == 0
    mw = bench.MeWorkload(lib, dev, 3840, 2160, 16, "m8_4k_tl2", (-1, -2, -3), (1, 2), seed=7)
    stream = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(stream.cuda_stream)
    mw.analysis(sp)
    out = {}
    for stop in STOPS:
        for mask in MASKS:
            assert lib.svt_hip_debug_me_stop(stop | (mask << 8)) == 0
            ms = bench.timed_launches(stream, 2, 1, lambda: mw.me(sp))
            out[f"stop{stop}_mask{mask}"] = round(ms, 4)
    print(json.dumps({"me_ablation_skip_ms": out}))


if __name__ == "__main__":
    main()

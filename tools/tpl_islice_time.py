#!/usr/bin/env python3
"""Tuning aid: wall time of svt_hip_tpl_dispenser_frame on an all-intra 1920x1080 picture (every block waits for its neighbours:
the longest dependency chains), tpl level 4 and level 5.  Usage (GPU box, repo root): python tools/tpl_islice_time.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "svt-av1-mod-by-patman_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import pyorc  # noqa: E402
import tpl_cases as T  # noqa: E402
from svtav1_hip import abi, device  # noqa: E402
from test_gpu_tpl import DevMap  # noqa: E402

hip, orc = abi.load(), pyorc.oracle()
assert hip.svt_hip_init(0) == 0
T.load_quant(np.load(os.path.join(ROOT, "tests", "golden", "tpl_frame.npz")))
for name, opt in (("level 4", dict(T.BASE, i_slice=1, tpl_i_slice=1)), ("level 5", dict(T.BASE, i_slice=1, tpl_i_slice=1, synth_blk_size=32, **T.L5))):
    s = T.TplScene(orc, ("islice", "pan", 1920, 1080, 120, opt), key="m8_1080p_tl2")
    dm = DevMap(hip)
    job = s.job(dm)
    hip.svt_hip_tpl_workspace_bytes.restype = C.c_uint64
    wsb = hip.svt_hip_tpl_workspace_bytes(1920, 1080)
    ws = device.DeviceBuffer(hip, wsb)
    job.workspace, job.workspace_bytes = ws.ptr, wsb
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        device.check(hip, hip.svt_hip_tpl_dispenser_frame(C.byref(job), None), "tpl")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
        ts.append((time.perf_counter() - t) * 1e3)
    print(name, "all-intra 1080p:", " ".join(f"{t:.3f}" for t in ts), "ms (call + sync)")

#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the REAL reference C path.

Runs only where oracle/_ref/libsvtref.so exists (the build container: `make -C oracle ref` compiles it from
/root/reference).  The fixtures are data only — inputs (seeded) and the outputs the reference functions
produced — and are committed so that the oracle stays pinned on machines without the reference.

    python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import pyorc  # noqa: E402
from svtav1_hip import abi, frames  # noqa: E402
import me_cases  # noqa: E402

ref = pyorc.ref()
u8p = C.POINTER(C.c_uint8)


def ptr(a, off=0):
    return C.cast(a.ctypes.data + off, u8p)


def gen_me_params():
    out = {}
    res = {"360p": (640, 360), "720p": (1280, 720), "1080p": (1920, 1080), "4k": (3840, 2160)}
    for mode in (0, 2, 4, 6, 8, 10, 12):
        for rname, (w, h) in res.items():
            for tl in (0, 2):
                hl = 5 if mode <= 8 else 4
                # HME level 2 is on up to M6 (enc_mode_config.c:1803-1809)
                l2 = 1 if mode <= 6 else 0
                p = abi.MeParams()
                rc = ref.ref_derive_me_params(mode, w, h, 35, hl, tl, 0, 30 << 16, 1, 1, l2, C.byref(p))
                assert rc == 0
                out[f"m{mode}_{rname}_tl{tl}"] = p.to_dict()
    with open(os.path.join(HERE, "me_params.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("me_params.json:", len(out), "sets")


def gen_sad_loop():
    """Expected outputs of svt_sad_loop_kernel_c for me_cases.iter_sad_loop_cases() (inputs are seeded)."""
    cases = []
    for prm, src, refw in me_cases.iter_sad_loop_cases():
        best, x, y = me_cases.call_sad_loop(ref.svt_sad_loop_kernel_c, prm, src, refw)
        cases.append(dict(prm, best=best, x=x, y=y))
    with open(os.path.join(HERE, "sad_loop_cases.json"), "w") as f:
        json.dump(cases, f)
    print("sad_loop_cases.json:", len(cases))


def gen_me_frames():
    """Whole-picture open-loop ME of small clips through the real svt_aom_motion_estimation_b64."""
    store = {}
    scen = [
        ("pan", 320, 192, "m8_360p_tl2", 2, [1, 0], [3, 4]),
        ("noise", 256, 200, "m8_360p_tl2", 2, [1, 0], [3, 4]),
        ("static", 320, 192, "m8_360p_tl0", 4, [3, 2, 1], []),
        ("fastpan", 328, 200, "m4_360p_tl2", 2, [0], [4]),
        ("blocks", 384, 256, "m0_360p_tl2", 1, [0], [2]),
        ("noise", 256, 136, "m12_360p_tl2", 2, [1], [3]),
    ]
    for i, (kind, w, h, key, cur, l0, l1) in enumerate(scen):
        clip = me_cases.make_clip(kind, w, h, 5, seed=100 + i)
        pyrs = me_cases.build_pyramids(pyorc.oracle(), clip)
        prm = me_cases.scenario_params(key, cur, l0, l1)
        arrs = me_cases.run_cpu(ref.ref_me_frame, prm, pyrs, cur, l0, l1, w, h)
        for k, v in arrs.items():
            store[f"s{i}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "me_frames.npz"), **store)
    with open(os.path.join(HERE, "me_frames.json"), "w") as f:
        json.dump([dict(kind=k, w=w, h=h, key=key, cur=cur, l0=l0, l1=l1, seed=100 + i)
                   for i, (k, w, h, key, cur, l0, l1) in enumerate(scen)], f)
    print("me_frames.npz:", len(scen), "scenarios")


def gen_pyramid_variance():
    clip = me_cases.make_clip("pan", 200, 136, 1, seed=5)
    p = frames.HostPyramid(clip[0])
    d = p.desc()
    ref.ref_pyramid_frame(C.byref(d.full), C.byref(d.quarter), C.byref(d.sixteenth), 1)
    nb = frames.b64_count(200, 136)
    var_sub = np.zeros((nb, 85), np.uint16)
    var_full = np.zeros((nb, 85), np.uint16)
    ref.ref_variance_frame(C.byref(d.full), var_sub.ctypes.data_as(C.c_void_p), 0)
    ref.ref_variance_frame(C.byref(d.full), var_full.ctypes.data_as(C.c_void_p), 1)
    p2 = frames.HostPyramid(clip[0])
    d2 = p2.desc()
    ref.ref_pyramid_frame(C.byref(d2.full), C.byref(d2.quarter), C.byref(d2.sixteenth), 0)
    np.savez_compressed(os.path.join(HERE, "pyramid_variance.npz"), quarter=p.quarter.buf, sixteenth=p.sixteenth.buf,
                        sixteenth_step4=p2.sixteenth.buf, var_sub=var_sub, var_full=var_full)
    print("pyramid_variance.npz")


if __name__ == "__main__":
    gen_me_params()
    gen_sad_loop()
    gen_me_frames()
    gen_pyramid_variance()

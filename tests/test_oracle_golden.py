"""CPU: the oracle (our C restatement) against the committed golden vectors produced by the real reference."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import me_cases
from svtav1_hip import frames

G = me_cases.GOLDEN


def test_sad_loop_golden(orc):
    with open(os.path.join(G, "sad_loop_cases.json")) as f:
        expected = json.load(f)
    n = 0
    for exp, (prm, src, refw) in zip(expected, me_cases.iter_sad_loop_cases()):
        assert {k: exp[k] for k in prm} == prm
        got = me_cases.call_sad_loop(orc.orc_sad_loop_kernel, prm, src, refw)
        assert got == (exp["best"], exp["x"], exp["y"]), prm
        n += 1
    assert n == len(expected) == 280


def test_me_frames_golden(orc):
    with open(os.path.join(G, "me_frames.json")) as f:
        scen = json.load(f)
    gold = np.load(os.path.join(G, "me_frames.npz"))
    for i, s in enumerate(scen):
        clip = me_cases.make_clip(s["kind"], s["w"], s["h"], 5, seed=s["seed"])
        pyrs = me_cases.build_pyramids(orc, clip)
        prm = me_cases.scenario_params(s["key"], s["cur"], s["l0"], s["l1"])
        got = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, s["cur"], s["l0"], s["l1"], s["w"], s["h"])
        me_cases.assert_same({k: gold[f"s{i}_{k}"] for k in got}, got, f"scenario {i} {s}")


def test_pyramid_variance_golden(orc):
    gold = np.load(os.path.join(G, "pyramid_variance.npz"))
    clip = me_cases.make_clip("pan", 200, 136, 1, seed=5)
    for l1, key in ((1, "sixteenth"), (0, "sixteenth_step4")):
        p = frames.HostPyramid(clip[0])
        d = p.desc()
        orc.orc_pyramid_frame(C.byref(d.full), C.byref(d.quarter), C.byref(d.sixteenth), l1)
        if l1:
            assert np.array_equal(p.quarter.buf, gold["quarter"])
        assert np.array_equal(p.sixteenth.buf, gold[key])
    nb = frames.b64_count(200, 136)
    for fp, key in ((0, "var_sub"), (1, "var_full")):
        var = np.zeros((nb, 85), np.uint16)
        orc.orc_variance_frame(C.byref(d.full), var.ctypes.data_as(C.c_void_p), None, fp)
        assert np.array_equal(var, gold[key])


def test_me_mctf_golden(orc):
    """ME_MCTF mode against the outputs of the real svt_aom_motion_estimation_b64 (tests/golden/make_golden_mctf.py)."""
    gold = np.load(os.path.join(G, "me_mctf.npz"))
    for i, (kind, w, h, key, cur, refpoc, th, seed) in enumerate(me_cases.MCTF_SCENARIOS):
        clip = me_cases.make_clip(kind, w, h, 5, seed=seed)
        pyrs = me_cases.build_pyramids(orc, clip)
        got = me_cases.run_cpu(orc.orc_me_frame_range, me_cases.mctf_params(key, cur, refpoc, th), pyrs, cur, [refpoc], [], w, h)
        for k in ("best_sad", "best_mv", "search_results"):
            assert np.array_equal(got[k], gold[f"s{i}_{k}"]), (i, k)

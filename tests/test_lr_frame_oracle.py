"""CPU: the oracle's frame-level loop restoration (oracle/src/orc_lr_frame.c) against the REAL
svt_av1_loop_restoration_filter_unit + svt_extend_frame run over the same planes (oracle/ref_harness_lr.c), and against the
committed fixture those produced (tests/golden/lr_frame.npz) where the reference build is absent."""
import ctypes as C
import os

import numpy as np
import pytest

import lr_cases as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_frame.npz")


def run(fn, case):
    arr, outs = R.lr_planes(case)
    assert fn(arr, C.c_uint32(len(case))) in (0, None)
    return [o[:c["h"], :c["w"]].copy() for o, c in zip(outs, case)]


@pytest.mark.parametrize("name", list(R.CASES))
def test_oracle_equals_reference(orc, ref, name):
    case = R.make_case(name)
    want = run(ref.ref_restoration_filter_frame, case)
    got = run(orc.orc_restoration_filter_frame, case)
    for p, (a, b, c) in enumerate(zip(want, got, case)):
        assert np.array_equal(a, b), (name, p, np.argwhere(a != b)[:5])
        types = set(int(u["restoration_type"]) for u in c["units"])
        assert (a != c["src"]).any() or types == {0}


def test_oracle_equals_golden(orc):
    g = np.load(GOLD)
    for name in R.CASES:
        got = run(orc.orc_restoration_filter_frame, R.make_case(name))
        for p, a in enumerate(got):
            assert np.array_equal(a, g[f"{name}_p{p}"]), (name, p)

"""CPU: the oracle's restatement of the temporal filter's block loop (oracle/src/orc_tf_picture.c) against the REAL
produce_temporally_filtered_pic of the reference (oracle/ref_harness_tfme.c compiles temporal_filtering.c in place), and
against the committed golden pictures the reference produced (tests/golden/make_golden_tf_picture.py)."""
import os

import numpy as np
import pytest

import pyorc
import tf_picture_cases as tpc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_picture.npz")


@pytest.fixture(scope="module")
def orc():
    return pyorc.oracle()


@pytest.mark.parametrize("case", tpc.CASES, ids=lambda c: c[0])
def test_oracle_vs_reference(orc, case):
    ref = pyorc.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    a = tpc.case_window(orc, case)
    b = tpc.case_window(orc, case)
    before = {k: v.copy() for k, v in a[0].arrays().items()}
    decay, tot_ref = tpc.run_reference(ref, a, case)
    states, tot = tpc.run_oracle(orc, b, case, decay)
    assert tot == tot_ref
    changed = 0
    for k, v in a[0].arrays().items():
        assert np.array_equal(v, b[0].arrays()[k]), (case[0], k, int((v != b[0].arrays()[k]).sum()))
        changed += int((v != before[k]).sum())
    assert changed > 0  # the filter did something
    for pa, pb in zip(a[1:], b[1:]):  # reference pictures are inputs only
        for k, v in pa.arrays().items():
            assert np.array_equal(v, pb.arrays()[k])


@pytest.mark.parametrize("case", tpc.CASES, ids=lambda c: c[0])
def test_oracle_vs_golden(orc, case):
    gold = np.load(GOLDEN)
    name = case[0]
    pics = tpc.case_window(orc, case)
    decay = tuple(int(x) for x in gold[f"{name}_decay"])
    states, tot = tpc.run_oracle(orc, pics, case, decay)
    assert tot == tuple(int(x) for x in gold[f"{name}_tot"])
    for k, v in pics[0].arrays().items():
        if f"{name}_{k}" in gold:
            assert np.array_equal(v, gold[f"{name}_{k}"]), (name, k)

"""CPU: the oracle's restatement of the TPL dispenser (oracle/src/orc_tpl.c) against the REAL tpl_mc_flow_dispenser_sb_generic
(oracle/ref_harness_tpl.c compiles src_ops_process.c in place) and against the golden results the reference produced."""
import ctypes as C
import os

import numpy as np
import pytest

import pyorc
import tpl_cases as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tpl_frame.npz")


@pytest.fixture(scope="module")
def orc():
    return pyorc.oracle()


@pytest.fixture(scope="module", autouse=True)
def quant():
    T.load_quant(np.load(GOLDEN))


@pytest.mark.parametrize("case", T.CASES, ids=lambda c: c[0])
def test_oracle_vs_reference(orc, case):
    ref = pyorc.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    q = (C.c_int16 * 6)()
    ref.ref_tpl_quant(case[4], q)
    assert list(q) == T.QUANT[case[4]]
    a, b = T.TplScene(orc, case), T.TplScene(orc, case)
    if case[5]["src_data_ready"]:
        T.prime_second_pass(orc, a), T.prime_second_pass(orc, b)
    before = a.out.buf.copy()
    assert ref.ref_tpl_dispenser_frame(C.byref(a.job()), case[4]) == 0
    assert orc.orc_tpl_dispenser_frame(C.byref(b.job())) == 0
    for k, v in a.results().items():
        assert np.array_equal(v, b.results()[k]), (case[0], k, int((v != b.results()[k]).sum()))
    assert (a.out.buf != before).any()
    modes = a.src_stats["best_mode"]
    print(case[0], "NEWMV", int((modes == 16).sum()), "DC", int((modes == 0).sum()))


@pytest.mark.parametrize("case", T.CASES, ids=lambda c: c[0])
def test_oracle_vs_golden(orc, case):
    gold = np.load(GOLDEN)
    s = T.TplScene(orc, case)
    if case[5]["src_data_ready"]:
        T.prime_second_pass(orc, s)
    assert orc.orc_tpl_dispenser_frame(C.byref(s.job())) == 0
    for k, v in s.results().items():
        assert np.array_equal(v, gold[f"{case[0]}_{k}"]), (case[0], k)

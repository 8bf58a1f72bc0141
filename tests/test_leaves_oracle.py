"""CPU: the remaining per-call RTCD leaves — oracle (oracle/src/orc_leaves.c, orc_inv_txfm2d_add_8bit) against the REAL
reference through its own dispatch pointers (oracle/_ref: sad_16b_kernel, svt_initialize_buffer_32bits, svt_residual_kernel8bit /
16bit, svt_spatial_full_distortion_kernel, svt_full_distortion_kernel16_bits, svt_pme_sad_loop_kernel, svt_search_one_dual,
svt_av1_inv_txfm_add, svt_nxm_sad_kernel_sub_sampled) and against tests/golden/leaves.npz (tests/golden/make_golden_leaves.py)."""
import ctypes as C
import os

import numpy as np

import leaf_cases as L
from svtav1_hip import abi
from test_txfm_oracle import rtcd
from tx_cases import P, V

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "leaves.npz")
U32 = C.c_uint32


def orc_residual(orc, h, w, rs, a, b, hbd):
    out = np.full((h, rs), -9, np.int16)
    (orc.orc_residual16 if hbd else orc.orc_residual8)(P(a), U32(a.shape[1]), P(b), U32(b.shape[1]), P(out), U32(rs), U32(w), U32(h))
    return out


def orc_sse(orc, h, w, o0, o1, a, b, hbd):
    fn = orc.orc_spatial_sse16 if hbd else orc.orc_spatial_sse8
    fn.restype = C.c_uint64
    return int(fn(P(a), U32(o0), U32(a.shape[1]), P(b), C.c_int32(o1), U32(b.shape[1]), U32(w), U32(h)))


def orc_inv8(orc, w, h, tt, co, pred, rs):
    rec = np.full((h, rs), 7, np.uint8)
    orc.orc_inv_txfm2d_add_8bit(P(co), P(pred), C.c_int32(pred.shape[1]), P(rec), C.c_int32(rs), w, h, tt)
    return rec


def all_expected(orc):
    """name -> expected value of every golden-pinned case, computed by `orc` (an object with the oracle's entry points)."""
    out = {}
    orc.orc_sad_16b.restype = C.c_uint32
    out["sad16"] = np.array([orc.orc_sad_16b(P(s), U32(s.shape[1]), P(r), U32(r.shape[1]), U32(h), U32(w)) for h, w, s, r in L.sad16_cases()],
                            np.uint64)
    out["sse"] = np.array([orc_sse(orc, *c) for c in L.sse_cases()], np.uint64)
    out["pme"] = np.array([L.run_pme_orc(orc, c) for c in L.pme_cases()], np.int64)
    out["dual"] = np.array([[t, *l0, *l1] for t, l0, l1 in (L.run_dual_orc(orc, c) for c in L.dual_cases())], np.uint64)
    for i, (h, w, rs, a, b, hbd) in enumerate(L.residual_cases()):
        out[f"res{i}"] = orc_residual(orc, h, w, rs, a, b, hbd)[:, :w].copy()
    for i, (ti, w, h, tt, co, pred, rs) in enumerate(L.inv8_cases(orc)):
        out[f"inv{i}"] = orc_inv8(orc, w, h, tt, co, pred, rs)[:, :w].copy()
    return out


def test_oracle_matches_golden(orc):
    gold = np.load(GOLD)
    exp = all_expected(orc)
    assert set(gold.files) == set(exp)
    for k, v in exp.items():
        assert np.array_equal(gold[k], v), k


def test_leaves_vs_reference(orc, ref):
    sad16 = rtcd(ref, "sad_16b_kernel", U32, V, U32, V, U32, U32, U32)
    orc.orc_sad_16b.restype = C.c_uint32
    for h, w, s, r in L.sad16_cases():
        assert sad16(s.ctypes.data, s.shape[1], r.ctypes.data, r.shape[1], h, w) == orc.orc_sad_16b(P(s), U32(s.shape[1]), P(r), U32(r.shape[1]), U32(h), U32(w))
    # the C row of the table binds the sub-sampled pointer to the plain SAD (aom_dsp_rtcd.c:1213)
    nxm, sub = rtcd(ref, "svt_nxm_sad_kernel", U32, V, U32, V, U32, U32, U32), rtcd(ref, "svt_nxm_sad_kernel_sub_sampled", U32, V, U32, V, U32, U32, U32)
    rng = np.random.default_rng(1)
    a, b = rng.integers(0, 256, size=(64, 70)).astype(np.uint8), rng.integers(0, 256, size=(64, 80)).astype(np.uint8)
    for h, w in ((64, 64), (16, 8), (32, 32)):
        assert nxm(a.ctypes.data, 70, b.ctypes.data, 80, h, w) == sub(a.ctypes.data, 70, b.ctypes.data, 80, h, w) == orc.orc_nxm_sad(P(a), U32(70), P(b), U32(80), U32(h), U32(w))
    fill = rtcd(ref, "svt_initialize_buffer_32bits", None, V, U32, U32, U32)
    for c128, c32, val in ((21, 1, 0xFFFFFFFF), (0, 3, 7), (5, 0, 0x01020304), (0, 0, 9)):
        x, y = np.full(c128 * 4 + c32 + 3, 5, np.uint32), np.full(c128 * 4 + c32 + 3, 5, np.uint32)
        fill(x.ctypes.data, c128, c32, val)
        orc.orc_initialize_buffer32(P(y), U32(c128), U32(c32), U32(val))
        assert np.array_equal(x, y)
    res8 = rtcd(ref, "svt_residual_kernel8bit", None, V, U32, V, U32, V, U32, U32, U32)
    res16 = rtcd(ref, "svt_residual_kernel16bit", None, V, U32, V, U32, V, U32, U32, U32)
    for h, w, rs, a, b, hbd in L.residual_cases():
        out = np.full((h, rs), -9, np.int16)
        (res16 if hbd else res8)(a.ctypes.data, a.shape[1], b.ctypes.data, b.shape[1], out.ctypes.data, rs, w, h)
        assert np.array_equal(out, orc_residual(orc, h, w, rs, a, b, hbd)), (h, w, hbd)
    sse8 = rtcd(ref, "svt_spatial_full_distortion_kernel", C.c_uint64, V, U32, U32, V, C.c_int32, U32, U32, U32)
    sse16 = rtcd(ref, "svt_full_distortion_kernel16_bits", C.c_uint64, V, U32, U32, V, C.c_int32, U32, U32, U32)
    for h, w, o0, o1, a, b, hbd in L.sse_cases():
        assert (sse16 if hbd else sse8)(a.ctypes.data, o0, a.shape[1], b.ctypes.data, o1, b.shape[1], w, h) == orc_sse(orc, h, w, o0, o1, a, b, hbd)


def test_pme_and_dual_vs_reference(orc, ref):
    pme = rtcd(ref, "svt_pme_sad_loop_kernel", None, *L.PME_ARGS)
    moved = 0
    for c in L.pme_cases():
        got, exp = L.run_pme(pme, c), L.run_pme_orc(orc, c)
        assert got == exp, (c.bw, c.bh, c.saw, c.sah, c.step, c.type)
        moved += got[0] != c.best
    assert moved > 20          # most cases update the best; the rest exercise "nothing beats the incoming cost"
    dual = rtcd(ref, "svt_search_one_dual", C.c_uint64, V, V, C.c_int, V, C.c_int, C.c_int, C.c_int)
    for case in L.dual_cases():
        t0, a0, a1 = L.run_dual(dual, case)
        t1, b0, b1 = L.run_dual_orc(orc, case)
        assert t0 == t1 and np.array_equal(a0, b0) and np.array_equal(a1, b1), case[2:]


def test_inv_txfm_add_8bit_vs_reference(orc, ref):
    inv = rtcd(ref, "svt_av1_inv_txfm_add", None, V, V, C.c_int32, V, C.c_int32, V)
    assert C.sizeof(abi.TxfmParam) == 24 and abi.TxfmParam.eob.offset == 20 and abi.TxfmParam.tx_set_type.offset == 16
    assert C.sizeof(abi.MvCostParam) == 56 and abi.MvCostParam.mvjcost.offset == 16 and abi.MvCostParam.error_per_bit.offset == 40
    n = 0
    for ti, w, h, tt, co, pred, rs in L.inv8_cases(orc):
        prm = abi.TxfmParam(tx_type=tt, tx_size=ti, lossless=0, bd=8, is_hbd=1, tx_set_type=0, eob=w * h)
        rec = np.full((h, rs), 7, np.uint8)
        # the reference reads min(w,32) x min(h,32) coefficients of a 32-wide packed buffer for 64-point sizes (:2567-2580)
        inv(co.ctypes.data, pred.ctypes.data, pred.shape[1], rec.ctypes.data, rs, C.byref(prm))
        assert np.array_equal(rec, orc_inv8(orc, w, h, tt, co, pred, rs)), (w, h, tt)
        n += 1
    assert n > 60

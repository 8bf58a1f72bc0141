"""CPU: the transform / quantiser oracle against the REAL reference (RTCD pointers of oracle/_ref) and against the
committed golden vectors (tests/golden/txfm.npz, produced by tests/golden/make_golden_txfm.py from the reference)."""
import ctypes as C
import os

import numpy as np
import pytest

import tx_cases as T
from tx_cases import P, V

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "txfm.npz")


def rtcd(ref, name, restype, *argtypes):
    p = C.c_void_p.in_dll(ref, name).value
    assert p, name
    return C.CFUNCTYPE(restype, *argtypes)(p)


def ref_inverse(ref, w, h, co, pred, ps, rec, rs, tt, bd):
    name = f"svt_av1_inv_txfm2d_add_{w}x{h}"
    if w == h:
        rtcd(ref, name, None, V, V, C.c_int32, V, C.c_int32, C.c_int, C.c_int32)(co.ctypes.data, pred.ctypes.data, ps, rec.ctypes.data, rs, tt, bd)
    elif (w, h) in ((4, 8), (8, 4), (4, 16), (16, 4)):
        rtcd(ref, name, None, V, V, C.c_int32, V, C.c_int32, C.c_int, C.c_int, C.c_int32)(co.ctypes.data, pred.ctypes.data, ps, rec.ctypes.data, rs, tt, 0, bd)
    else:
        rtcd(ref, name, None, V, V, C.c_int32, V, C.c_int32, C.c_int, C.c_int, C.c_int32, C.c_int32)(
            co.ctypes.data, pred.ctypes.data, ps, rec.ctypes.data, rs, tt, 0, len(co), bd)


def test_cospi_table(orc, ref):
    cos = np.ctypeslib.as_array((C.c_int32 * (7 * 64)).in_dll(ref, "svt_aom_eb_av1_cospi_arr_data")).reshape(7, 64)
    orc.orc_cospi.restype = C.POINTER(C.c_int32)
    for b in range(10, 14):
        assert np.array_equal(np.ctypeslib.as_array(orc.orc_cospi(b), shape=(64,)), cos[b - 10])


@pytest.mark.parametrize("w,h", T.SIZES)
def test_fwd_inv_vs_reference(orc, ref, w, h):
    rng = np.random.default_rng(w * 100 + h)
    FW = (None, V, V, C.c_uint32, C.c_int, C.c_uint8)
    for tt in range(16):
        if not orc.orc_txfm_valid(w, h, tt):
            continue
        for bd in (8, 10):
            for trial in range(3):
                res = T.residual(rng, w, h, bd, trial)
                for shape, suf in ((0, ""), (1, "_N2"), (2, "_N4")):
                    o1, o2 = np.zeros(w * h, np.int32), np.zeros(w * h, np.int32)
                    rtcd(ref, f"svt_av1_fwd_txfm2d_{w}x{h}{suf}", *FW)(res.ctypes.data, o1.ctypes.data, w + 3, tt, bd)
                    orc.orc_fwd_txfm2d(P(res), P(o2), C.c_uint32(w + 3), w, h, tt, bd, shape)
                    assert np.array_equal(o1, o2), (w, h, tt, bd, trial, shape)
                co = T.coeffs_for_inverse(rng, orc, w, h, tt, bd, trial)
                pred = rng.integers(0, 1 << bd, size=(h, w + 5)).astype(np.uint16)
                r1, r2 = np.zeros((h, w + 7), np.uint16), np.zeros((h, w + 7), np.uint16)
                ref_inverse(ref, w, h, co, pred, w + 5, r1, w + 7, tt, bd)
                orc.orc_inv_txfm2d_add(P(co), P(pred), w + 5, P(r2), w + 7, w, h, tt, bd)
                assert np.array_equal(r1, r2), (w, h, tt, bd, trial)
    if max(w, h) == 64:
        orc.orc_handle_transform64.restype = C.c_uint64
        for suf, en in (("", 1), ("_N2_N4", 0)):
            co = rng.integers(-100000, 100000, size=w * h).astype(np.int32)
            c2 = co.copy()
            e1 = rtcd(ref, f"svt_handle_transform{w}x{h}{suf}", C.c_uint64, V)(co.ctypes.data)
            e2 = orc.orc_handle_transform64(P(c2), w, h) if en else (orc.orc_handle_transform64(P(c2), w, h) * 0)
            kw, kh = min(w, 32), min(h, 32)
            assert e1 == e2 and np.array_equal(co[:kw * kh], c2[:kw * kh])


def test_inverse_8bit_entry(orc, ref):
    """svt_av1_inv_txfm_add_c: the 8-bit pixel path (widen, hbd inverse with bd=8, narrow)."""
    rng = np.random.default_rng(5)
    for (w, h) in ((8, 8), (16, 32), (64, 64), (4, 16)):
        co = T.coeffs_for_inverse(rng, orc, w, h, 0, 8, 0)
        pred = rng.integers(0, 256, size=(h, w + 5)).astype(np.uint8)
        p16 = pred.astype(np.uint16)
        r16 = np.zeros((h, w + 5), np.uint16)
        ref_inverse(ref, w, h, co, p16, w + 5, r16, w + 5, 0, 8)
        r8 = np.zeros((h, w + 7), np.uint8)
        orc.orc_inv_txfm2d_add_8bit(P(co), P(pred), w + 5, P(r8), w + 7, w, h, 0)
        assert np.array_equal(r8[:, :w], r16[:, :w].astype(np.uint8))


def test_quantizers_vs_reference(orc, ref):
    rng = np.random.default_rng(2)
    SIG = (V, C.c_ssize_t, V, V, V, V, V, V, V, V, V, V)
    qb = rtcd(ref, "svt_aom_quantize_b", None, *SIG, V, V, C.c_int32)
    hqb = rtcd(ref, "svt_aom_highbd_quantize_b", None, *SIG, V, V, C.c_int32)
    fpq = rtcd(ref, "svt_av1_quantize_fp_qm", None, *SIG, V, V, C.c_int16)
    hfpq = rtcd(ref, "svt_av1_highbd_quantize_fp_qm", None, *SIG, V, V, C.c_int16)
    fps = [rtcd(ref, n, None, *SIG) for n in ("svt_av1_quantize_fp", "svt_av1_quantize_fp_32x32", "svt_av1_quantize_fp_64x64")]
    hfp = rtcd(ref, "svt_av1_highbd_quantize_fp", None, *SIG, C.c_int16)
    for trial in range(300):
        c = T.quant_case(rng, trial)
        n, t, ls = c["n"], c["t"], c["ls"]
        A = lambda a: a.ctypes.data
        qm = A(c["qm"]) if c["qm"] is not None else None
        iqm = A(c["iqm"]) if c["iqm"] is not None else None

        def run_ref(fn, rnd, qnt, *tail):
            qc, dq, eob = np.full(n, 7, np.int32), np.full(n, 7, np.int32), C.c_uint16(9999)
            fn(A(c["coeff"]), n, A(t["zbin"]), A(rnd), A(qnt), A(t["qshift"]), A(qc), A(dq), A(t["dequant"]), C.addressof(eob),
               A(c["scan"]), A(c["iscan"]), *tail)
            return qc, dq, eob.value
        pairs = [(run_ref(qb, t["round"], t["quant"], qm, iqm, ls), T.orc_quant(orc, 1, c)),
                 (run_ref(hqb, t["round"], t["quant"], qm, iqm, ls), T.orc_quant(orc, 2, c)),
                 (run_ref(fpq, t["round_fp"], t["quant_fp"], qm, iqm, ls), T.orc_quant(orc, 3, c)),
                 (run_ref(hfpq, t["round_fp"], t["quant_fp"], qm, iqm, ls), T.orc_quant(orc, 4, c))]
        if c["qm"] is None:
            pairs += [(run_ref(fps[ls], t["round_fp"], t["quant_fp"]), T.orc_quant(orc, 3, c)),
                      (run_ref(hfp, t["round_fp"], t["quant_fp"], ls), T.orc_quant(orc, 4, c))]
        for k, (a, b) in enumerate(pairs):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (trial, k)


def golden_cases():
    """Seeded subset used by the committed golden file: every size, a DCT / ADST-or-flip / identity type, bd 8 & 10."""
    rng = np.random.default_rng(99)
    for (w, h) in T.SIZES:
        for tt in (0, 9, 6 if max(w, h) <= 16 else (10 if (w, h) == (32, 32) else 0)):
            for bd in (8, 10):
                yield w, h, tt, bd, T.residual(rng, w, h, bd, 0), rng.integers(0, 1 << bd, size=(h, w + 5)).astype(np.uint16)


def test_fwd_inv_golden(orc):
    gold = np.load(GOLD)
    for i, (w, h, tt, bd, res, pred) in enumerate(golden_cases()):
        if not orc.orc_txfm_valid(w, h, tt):
            continue
        co = np.zeros(w * h, np.int32)
        orc.orc_fwd_txfm2d(P(res), P(co), C.c_uint32(w + 3), w, h, tt, bd, 0)
        assert np.array_equal(co, gold[f"fwd{i}"]), (w, h, tt, bd)
        ci = co.reshape(h, w)[:min(h, 32), :min(w, 32)].copy().reshape(-1)
        rec = np.zeros((h, w + 7), np.uint16)
        orc.orc_inv_txfm2d_add(P(ci), P(pred), w + 5, P(rec), w + 7, w, h, tt, bd)
        assert np.array_equal(rec, gold[f"inv{i}"]), (w, h, tt, bd)

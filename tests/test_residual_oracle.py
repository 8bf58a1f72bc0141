"""CPU: residual producer + transform-domain cost oracle (oracle/src/orc_residual.c) against the REAL reference
(svt_aom_subtract_block / svt_aom_highbd_subtract_block / svt_aom_satd RTCD slots and svt_av1_wht_fwd_txfm of oracle/_ref)
and against the committed golden vectors (tests/golden/tpl_cost.npz, written by tests/golden/make_golden_tpl.py)."""
import ctypes as C
import os

import numpy as np

from tx_cases import P, V
from test_txfm_oracle import rtcd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tpl_cost.npz")
PD = C.c_ssize_t
# TxSize values (definitions.h): the TPL dispenser's transform per (size, subsample_tx) — src_ops_process.c:380-382
TX = {(16, 0): 2, (16, 1): 8, (16, 2): 14, (32, 0): 3, (32, 1): 10, (32, 2): 16}


def subtract_cases():
    rng = np.random.default_rng(515)
    for k in range(24):
        rows, cols = int(rng.choice([4, 8, 16, 32, 64])), int(rng.choice([4, 8, 16, 32, 64]))
        ss, ps, ds = cols + int(rng.integers(0, 9)), cols + int(rng.integers(0, 9)), cols + int(rng.integers(0, 9))
        hbd = k % 2 == 1
        hi = 1 << (10 if hbd and k % 4 == 1 else (16 if hbd else 8))
        dt = np.uint16 if hbd else np.uint8
        yield rows, cols, ds, rng.integers(0, hi, size=(rows, ss)).astype(dt), rng.integers(0, hi, size=(rows, ps)).astype(dt), hbd


def tpl_cases():
    rng = np.random.default_rng(616)
    for size in (16, 32):
        for ss in (0, 1, 2):
            for pf in (0, 1, 2):
                for amp in (255, 40, 6):
                    stride_s, stride_p = size + 16, size + 5
                    src = rng.integers(0, 256, size=(size, stride_s)).astype(np.uint8)
                    pred = np.clip(src[:, :size].astype(np.int32) + rng.integers(-amp, amp + 1, size=(size, size)), 0, 255).astype(np.uint8)
                    pp = np.zeros((size, stride_p), np.uint8)
                    pp[:, :size] = pred
                    yield size, ss, pf, src, pp


def orc_subtract(orc, rows, cols, ds, s, p, hbd):
    d = np.full((rows, ds), -7, np.int16)
    fn = orc.orc_highbd_subtract_block if hbd else orc.orc_subtract_block
    fn(rows, cols, P(d), PD(ds), P(s), PD(s.shape[1]), P(p), PD(p.shape[1]))
    return d


def test_subtract_and_satd_vs_reference(orc, ref):
    sub = rtcd(ref, "svt_aom_subtract_block", None, C.c_int, C.c_int, V, PD, V, PD, V, PD)
    hsub = rtcd(ref, "svt_aom_highbd_subtract_block", None, C.c_int, C.c_int, V, PD, V, PD, V, PD, C.c_int)
    satd = rtcd(ref, "svt_aom_satd", C.c_int, V, C.c_int)
    orc.orc_satd.restype = C.c_int
    for rows, cols, ds, s, p, hbd in subtract_cases():
        d1 = np.full((rows, ds), -7, np.int16)
        if hbd:
            hsub(rows, cols, d1.ctypes.data, ds, s.ctypes.data, s.shape[1], p.ctypes.data, p.shape[1], 10)
        else:
            sub(rows, cols, d1.ctypes.data, ds, s.ctypes.data, s.shape[1], p.ctypes.data, p.shape[1])
        assert np.array_equal(d1, orc_subtract(orc, rows, cols, ds, s, p, hbd)), (rows, cols, hbd)
    rng = np.random.default_rng(3)
    for n in (16, 64, 256, 1024, 100):
        for mag in (5, 32640, 1 << 20):
            co = rng.integers(-mag, mag + 1, size=n).astype(np.int32)
            assert satd(co.ctypes.data, n) == orc.orc_satd(P(co), n)


def ref_tpl_cost(ref, size, ss, pf, src, pred):
    """src_ops_process.c:734-748 with the reference's own functions."""
    sub = rtcd(ref, "svt_aom_subtract_block", None, C.c_int, C.c_int, V, PD, V, PD, V, PD)
    satd = rtcd(ref, "svt_aom_satd", C.c_int, V, C.c_int)
    wht = ref.svt_av1_wht_fwd_txfm
    wht.restype, wht.argtypes = None, [V, C.c_int, V, C.c_int, C.c_int, C.c_int, C.c_int]
    diff, coeff = np.zeros(64 * 64, np.int16), np.zeros(64 * 64, np.int32)
    sub(size >> ss, size, diff.ctypes.data, size << ss, src.ctypes.data, src.shape[1] << ss, pred.ctypes.data, pred.shape[1] << ss)
    wht(diff.ctypes.data, size << ss, coeff.ctypes.data, TX[(size, ss)], pf, 8, 0)
    return satd(coeff.ctypes.data, (size * size) >> ss) << ss


def orc_tpl_cost(orc, size, ss, pf, src, pred):
    orc.orc_tpl_block_cost.restype = C.c_int64
    return orc.orc_tpl_block_cost(P(src), src.shape[1], P(pred), pred.shape[1], size, ss, pf)


def test_tpl_block_cost_vs_reference(orc, ref):
    for size, ss, pf, src, pred in tpl_cases():
        assert ref_tpl_cost(ref, size, ss, pf, src, pred) == orc_tpl_cost(orc, size, ss, pf, src, pred), (size, ss, pf)


def test_golden(orc):
    gold = np.load(GOLD)
    for i, (rows, cols, ds, s, p, hbd) in enumerate(subtract_cases()):
        assert np.array_equal(gold[f"sub{i}"], orc_subtract(orc, rows, cols, ds, s, p, hbd)[:, :cols]), i
    costs = [orc_tpl_cost(orc, *c) for c in tpl_cases()]
    assert np.array_equal(gold["tpl_cost"], np.array(costs, np.int64))


def distortion_cases():
    rng = np.random.default_rng(717)
    for k in range(20):
        w, h = int(rng.choice([4, 8, 16, 32, 11])), int(rng.choice([4, 8, 16, 32, 7]))
        cs, rs = w + int(rng.integers(0, 5)), w + int(rng.integers(0, 5))
        mag = int(rng.choice([40, 3000, 1 << 17, 1 << 24]))
        co = rng.integers(-mag, mag + 1, size=(h, cs)).astype(np.int32)
        rc = (co[:, :w] + rng.integers(-mag // 8 - 1, mag // 8 + 2, size=(h, w))).astype(np.int32)
        rr = np.zeros((h, rs), np.int32)
        rr[:, :w] = rc
        yield w, h, co, rr


def test_full_distortion_vs_reference(orc, ref):
    f32 = rtcd(ref, "svt_full_distortion_kernel32_bits", None, V, C.c_uint32, V, C.c_uint32, V, C.c_uint32, C.c_uint32)
    fz = rtcd(ref, "svt_full_distortion_kernel_cbf_zero32_bits", None, V, C.c_uint32, V, C.c_uint32, C.c_uint32)
    for w, h, co, rr in distortion_cases():
        a, b = np.zeros(2, np.uint64), np.zeros(2, np.uint64)
        f32(co.ctypes.data, co.shape[1], rr.ctypes.data, rr.shape[1], a.ctypes.data, w, h)
        orc.orc_full_distortion32(P(co), co.shape[1], P(rr), rr.shape[1], P(b), w, h)
        assert np.array_equal(a, b), (w, h)
        fz(co.ctypes.data, co.shape[1], a.ctypes.data, w, h)
        orc.orc_full_distortion32(P(co), co.shape[1], None, 0, P(b), w, h)
        assert np.array_equal(a, b), (w, h, "cbf0")

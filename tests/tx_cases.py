"""Shared case generators for the transform / quantiser parity tests (test infrastructure)."""
import ctypes as C

import numpy as np

SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64),
         (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
V = C.c_void_p


def P(a):
    return V(a.ctypes.data)


def residual(rng, w, h, bd, trial, pad=3):
    """Recipe of test/FwdTxfm2dAsmTest.cc: values in +-(2^bd - 1), plus the all-max / alternating extremes."""
    lim = (1 << bd) - 1
    if trial == 0:
        return rng.integers(-lim, lim + 1, size=(h, w + pad)).astype(np.int16)
    if trial == 1:
        return np.full((h, w + pad), lim, np.int16)
    return ((rng.integers(0, 2, size=(h, w + pad)) * 2 - 1) * lim).astype(np.int16)


def coeffs_for_inverse(rng, orc, w, h, tt, bd, trial):
    iw, ih = min(w, 32), min(h, 32)
    if trial == 0:  # a real forward transform output
        res = residual(rng, w, h, bd, 0)
        co = np.zeros(w * h, np.int32)
        orc.orc_fwd_txfm2d(P(res), P(co), C.c_uint32(w + 3), w, h, tt, bd, 0)
        return co.reshape(h, w)[:ih, :iw].copy().reshape(-1)
    if trial == 1:
        return rng.integers(-(1 << (bd + 7)), 1 << (bd + 7), size=iw * ih).astype(np.int32)
    co = rng.integers(-(1 << (bd + 9)), 1 << (bd + 9), size=iw * ih).astype(np.int32)  # out of range: clamps
    co[rng.integers(0, iw * ih, size=iw * ih // 2)] = 0
    return co


def quant_tables(rng, bd):
    """Quantiser tables with the structure of svt_av1_build_quantizer (md_config_process.c:83-144)."""
    q = int(rng.integers(4, 1337 if bd == 8 else 5347))
    dequant = np.array([q, min(32767, int(q * 1.3) + 1)] + [0] * 6, np.int16)

    def inv(d):
        l = int(d).bit_length() - 1
        m = 1 + (1 << (16 + l)) // int(d)
        return np.int16(m - (1 << 16)), np.int16(1 << (16 - l))
    quant, qshift = np.zeros(8, np.int16), np.zeros(8, np.int16)
    for i in range(2):
        quant[i], qshift[i] = inv(dequant[i])
    t = dict(dequant=dequant, quant=quant, qshift=qshift,
             zbin=np.array([(int(d) * 84 + 64) >> 7 for d in dequant[:2]] + [0] * 6, np.int16),
             round=np.array([(int(d) * 48) >> 7 for d in dequant[:2]] + [0] * 6, np.int16),
             round_fp=np.array([(int(d) * 64) >> 7 for d in dequant[:2]] + [0] * 6, np.int16),
             quant_fp=np.array([min(32767, (1 << 16) // int(d)) for d in dequant[:2]] + [0] * 6, np.int16))
    return t


def quant_case(rng, trial):
    n = int(rng.choice([16, 64, 256, 1024]))
    ls = int(rng.integers(0, 3))
    bd = int(rng.choice([8, 10]))
    mag = int(rng.choice([50, 2000, 1 << (bd + 7), 1 << 20]))
    coeff = rng.integers(-mag, mag + 1, size=n).astype(np.int32)
    coeff[rng.random(n) < 0.5] = 0
    scan = rng.permutation(n).astype(np.int16) if trial % 3 else np.arange(n, dtype=np.int16)
    iscan = np.empty(n, np.int16)
    iscan[scan] = np.arange(n)
    use_qm = trial % 4 == 1
    qm = rng.integers(16, 64, size=n).astype(np.uint8) if use_qm else None
    iqm = rng.integers(16, 64, size=n).astype(np.uint8) if use_qm else None
    return dict(n=n, ls=ls, bd=bd, coeff=coeff, scan=scan, iscan=iscan, qm=qm, iqm=iqm, t=quant_tables(rng, bd))


def orc_quant(orc, mode, c):
    """mode: 1 quantize_b, 2 highbd_quantize_b, 3 quantize_fp, 4 highbd_quantize_fp -> (qcoeff, dqcoeff, eob)"""
    n, t = c["n"], c["t"]
    qc, dq, eob = np.full(n, 7, np.int32), np.full(n, 7, np.int32), C.c_uint16(9999)
    qm = P(c["qm"]) if c["qm"] is not None else None
    iqm = P(c["iqm"]) if c["iqm"] is not None else None
    if mode in (1, 2):
        fn = orc.orc_quantize_b if mode == 1 else orc.orc_highbd_quantize_b
        fn(P(c["coeff"]), C.c_ssize_t(n), P(t["zbin"]), P(t["round"]), P(t["quant"]), P(t["qshift"]), P(qc), P(dq),
           P(t["dequant"]), C.byref(eob), P(c["scan"]), qm, iqm, c["ls"])
    else:
        fn = orc.orc_quantize_fp if mode == 3 else orc.orc_highbd_quantize_fp
        fn(P(c["coeff"]), C.c_ssize_t(n), P(t["round_fp"]), P(t["quant_fp"]), P(qc), P(dq), P(t["dequant"]), C.byref(eob),
           P(c["scan"]), qm, iqm, c["ls"])
    return qc, dq, eob.value

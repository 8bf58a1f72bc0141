"""GPU parity: the remaining per-call (Tier A) RTCD leaves through the C-ABI against the oracle and the golden vectors written by
the real reference (tests/golden/leaves.npz) — bit-exact.  svt_nxm_sad_kernel_sub_sampled_hip, svt_aom_sad_16b_kernel_hip,
svt_initialize_buffer_32bits_hip, svt_residual_kernel8bit/16bit_hip, svt_spatial_full_distortion_kernel_hip,
svt_full_distortion_kernel16_bits_hip, svt_pme_sad_loop_kernel_hip, svt_search_one_dual_hip, svt_av1_inv_txfm_add_hip."""
import ctypes as C

import numpy as np
import pytest

import leaf_cases as L
from svtav1_hip import abi
from test_leaves_oracle import GOLD, U32, orc_inv8, orc_residual, orc_sse
from tx_cases import P, V

pytestmark = pytest.mark.gpu


def test_sad_fill_residual_distortion(hip, orc):
    gold = np.load(GOLD)
    hip.svt_aom_sad_16b_kernel_hip.restype = U32
    hip.svt_nxm_sad_kernel_sub_sampled_hip.restype = U32
    orc.orc_sad_16b.restype = U32
    for i, (h, w, s, r) in enumerate(L.sad16_cases()):
        got = hip.svt_aom_sad_16b_kernel_hip(P(s), U32(s.shape[1]), P(r), U32(r.shape[1]), U32(h), U32(w))
        assert got == orc.orc_sad_16b(P(s), U32(s.shape[1]), P(r), U32(r.shape[1]), U32(h), U32(w)) == int(gold["sad16"][i])
    assert hip.svt_aom_sad_16b_kernel_hip(P(s), U32(9), P(r), U32(9), U32(0), U32(4)) == 0
    rng = np.random.default_rng(1)
    a, b = rng.integers(0, 256, size=(64, 70)).astype(np.uint8), rng.integers(0, 256, size=(64, 80)).astype(np.uint8)
    for h, w in ((64, 64), (16, 8), (32, 32), (5, 3)):
        assert hip.svt_nxm_sad_kernel_sub_sampled_hip(P(a), U32(70), P(b), U32(80), U32(h), U32(w)) == orc.orc_nxm_sad(P(a), U32(70), P(b), U32(80), U32(h), U32(w))
    for c128, c32, val in ((21, 1, 0xFFFFFFFF), (0, 3, 7), (5, 0, 0x01020304), (0, 0, 9), (300, 2, 0xDEADBEEF)):
        x, y = np.full(c128 * 4 + c32 + 3, 5, np.uint32), np.full(c128 * 4 + c32 + 3, 5, np.uint32)
        hip.svt_initialize_buffer_32bits_hip(P(x), U32(c128), U32(c32), U32(val))
        orc.orc_initialize_buffer32(P(y), U32(c128), U32(c32), U32(val))
        assert np.array_equal(x, y)         # incl. the untouched tail
    for i, (h, w, rs, a, b, hbd) in enumerate(L.residual_cases()):
        out = np.full((h, rs), -9, np.int16)
        (hip.svt_residual_kernel16bit_hip if hbd else hip.svt_residual_kernel8bit_hip)(P(a), U32(a.shape[1]), P(b), U32(b.shape[1]), P(out),
                                                                                      U32(rs), U32(w), U32(h))
        assert np.array_equal(out, orc_residual(orc, h, w, rs, a, b, hbd)), (h, w, hbd)
        assert np.array_equal(out[:, :w], gold[f"res{i}"])
    hip.svt_spatial_full_distortion_kernel_hip.restype = C.c_uint64
    hip.svt_full_distortion_kernel16_bits_hip.restype = C.c_uint64
    for i, (h, w, o0, o1, a, b, hbd) in enumerate(L.sse_cases()):
        fn = hip.svt_full_distortion_kernel16_bits_hip if hbd else hip.svt_spatial_full_distortion_kernel_hip
        got = fn(P(a), U32(o0), U32(a.shape[1]), P(b), C.c_int32(o1), U32(b.shape[1]), U32(w), U32(h))
        assert got == orc_sse(orc, h, w, o0, o1, a, b, hbd) == int(gold["sse"][i]), (h, w, hbd)


def test_pme_sad_loop(hip, orc):
    gold = np.load(GOLD)["pme"]
    fn = hip.svt_pme_sad_loop_kernel_hip
    fn.restype, fn.argtypes = None, list(L.PME_ARGS)
    for i, c in enumerate(L.pme_cases()):
        got = L.run_pme(fn, c)
        assert got == L.run_pme_orc(orc, c) == tuple(int(v) for v in gold[i]), (i, c.bw, c.bh, c.saw, c.sah, c.step, c.type)


def test_search_one_dual(hip, orc):
    gold = np.load(GOLD)["dual"]
    fn = hip.svt_search_one_dual_hip
    fn.restype, fn.argtypes = C.c_uint64, [V, V, C.c_int, V, C.c_int, C.c_int, C.c_int]
    for i, case in enumerate(L.dual_cases()):
        t0, a0, a1 = L.run_dual(fn, case)
        t1, b0, b1 = L.run_dual_orc(orc, case)
        assert t0 == t1 == int(gold[i][0]) and np.array_equal(a0, b0) and np.array_equal(a1, b1), case[2:]
        assert np.array_equal(np.concatenate([a0, a1]).astype(np.uint64), gold[i][1:])


def test_inv_txfm_add_8bit_entry(hip, orc):
    gold = np.load(GOLD)
    for i, (ti, w, h, tt, co, pred, rs) in enumerate(L.inv8_cases(orc)):
        prm = abi.TxfmParam(tx_type=tt, tx_size=ti, lossless=0, bd=8, is_hbd=1, tx_set_type=0, eob=w * h)
        rec = np.full((h, rs), 7, np.uint8)
        hip.svt_av1_inv_txfm_add_hip(P(co), P(pred), C.c_int32(pred.shape[1]), P(rec), C.c_int32(rs), C.byref(prm))
        assert np.array_equal(rec, orc_inv8(orc, w, h, tt, co, pred, rs)), (w, h, tt)   # incl. untouched padding columns
        assert np.array_equal(rec[:, :w], gold[f"inv{i}"])
    # in place (read and write pointers equal), as the TPL dispenser calls it (src_ops_process.c:1150-1159)
    ti, w, h, tt, co, pred, rs = next(iter(L.inv8_cases(orc)))
    buf = pred.copy()
    prm = abi.TxfmParam(tx_type=tt, tx_size=ti, lossless=0, bd=8, is_hbd=1, tx_set_type=0, eob=w * h)
    hip.svt_av1_inv_txfm_add_hip(P(co), P(buf), C.c_int32(buf.shape[1]), P(buf), C.c_int32(buf.shape[1]), C.byref(prm))
    assert np.array_equal(buf[:, :w], gold["inv0"]) and np.array_equal(buf[:, w:], pred[:, w:])


def test_rtcd_lookup_covers_the_new_leaves(hip):
    hip.svt_hip_rtcd_lookup.restype, hip.svt_hip_rtcd_lookup.argtypes = C.c_void_p, [C.c_char_p]
    for name in ("svt_nxm_sad_kernel_sub_sampled", "sad_16b_kernel", "svt_initialize_buffer_32bits", "svt_pme_sad_loop_kernel", "downsample_2d",
                 "svt_residual_kernel8bit", "svt_residual_kernel16bit", "svt_spatial_full_distortion_kernel", "svt_full_distortion_kernel16_bits",
                 "svt_search_one_dual", "svt_av1_inv_txfm_add"):
        assert hip.svt_hip_rtcd_lookup(name.encode()), name

"""GPU parity: the residual producer and transform-domain cost around the forward transform (the TPL dispenser's use of
the transform path, src_ops_process.c:734-748 / 861-873 / 1132-1188), through the C-ABI, against the oracle — bit-exact.
Tier A: svt_aom_subtract_block_hip / svt_aom_highbd_subtract_block_hip / svt_aom_satd_hip.  Tier B:
svt_hip_txfm_quant_batch with SVT_HIP_TX_SRC_PRED (residual formed in the kernel) and SVT_HIP_TX_SATD."""
import ctypes as C
import os

import numpy as np
import pytest

import tx_cases as T
from svtav1_hip import abi, device
from test_gpu_txfm import ArenaBuilder
from test_residual_oracle import GOLD, PD, distortion_cases, orc_subtract, orc_tpl_cost, subtract_cases, tpl_cases
from tx_cases import P, V

pytestmark = pytest.mark.gpu


def test_tier_a_subtract_and_satd(hip, orc):
    gold = np.load(GOLD)
    for i, (rows, cols, ds, s, p, hbd) in enumerate(subtract_cases()):
        d = np.full((rows, ds), -7, np.int16)
        if hbd:
            hip.svt_aom_highbd_subtract_block_hip(rows, cols, P(d), PD(ds), P(s), PD(s.shape[1]), P(p), PD(p.shape[1]), 10)
        else:
            hip.svt_aom_subtract_block_hip(rows, cols, P(d), PD(ds), P(s), PD(s.shape[1]), P(p), PD(p.shape[1]))
        assert np.array_equal(d, orc_subtract(orc, rows, cols, ds, s, p, hbd)), (rows, cols, hbd)   # incl. untouched padding
        assert np.array_equal(d[:, :cols], gold[f"sub{i}"]), i
    rng = np.random.default_rng(3)
    hip.svt_aom_satd_hip.restype = C.c_int
    orc.orc_satd.restype = C.c_int
    for n in (16, 64, 256, 1024, 100, 1):
        for mag in (5, 32640, 1 << 20):
            co = rng.integers(-mag, mag + 1, size=n).astype(np.int32)
            assert hip.svt_aom_satd_hip(P(co), n) == orc.orc_satd(P(co), n)
    assert hip.svt_aom_satd_hip(P(np.zeros(4, np.int32)), 0) == 0


def test_tier_a_full_distortion(hip, orc):
    for w, h, co, rr in distortion_cases():
        a, b = np.zeros(2, np.uint64), np.zeros(2, np.uint64)
        hip.svt_full_distortion_kernel32_bits_hip(P(co), co.shape[1], P(rr), rr.shape[1], P(a), w, h)
        orc.orc_full_distortion32(P(co), co.shape[1], P(rr), rr.shape[1], P(b), w, h)
        assert np.array_equal(a, b), (w, h)
        hip.svt_full_distortion_kernel_cbf_zero32_bits_hip(P(co), co.shape[1], P(a), w, h)
        orc.orc_full_distortion32(P(co), co.shape[1], None, 0, P(b), w, h)
        assert np.array_equal(a, b), (w, h, "cbf0")


def run_batch(hip, ab, descs, w, h):
    arena = ab.build()
    darena = device.DeviceBuffer(hip, arena.nbytes + 256)
    darena.upload(arena)
    darr = (abi.TxfmDesc * len(descs))(*descs)
    ddesc = device.DeviceBuffer(hip, C.sizeof(darr))
    ddesc.upload(np.frombuffer(darr, dtype=np.uint8))
    dres = device.DeviceBuffer(hip, abi.TXFM_RESULT_BYTES * len(descs))
    device.check(hip, hip.svt_hip_txfm_quant_batch(V(darena.ptr), V(ddesc.ptr), V(dres.ptr), C.c_uint32(len(descs)), C.c_uint32(w),
                                                   C.c_uint32(h), None), "svt_hip_txfm_quant_batch")
    return darena.download(np.uint8, (arena.nbytes,)), dres.download(np.uint8, (len(descs), abi.TXFM_RESULT_BYTES))


def blank_desc():
    d = abi.TxfmDesc()
    for f in ("residual_off", "coeff_off", "qcoeff_off", "dqcoeff_off", "pred_off", "recon_off", "iscan_off", "qm_off", "iqm_off"):
        setattr(d, f, abi.NO_OFFSET)
    return d


@pytest.mark.parametrize("size,ss", [(16, 0), (16, 1), (16, 2), (32, 0), (32, 1), (32, 2)])
def test_tier_b_tpl_block_cost(hip, orc, size, ss):
    """One launch = the transform-domain cost of many TPL blocks: flags FWD | SRC_PRED | SATD, DCT_DCT, the sub-sampled
    transform size as w x h and the strides pre-shifted as the reference's call does."""
    cases = [(k, c) for k, c in enumerate(tpl_cases()) if c[0] == size and c[1] == ss] * 9   # 81 blocks
    gold = np.load(GOLD)["tpl_cost"]
    ab, descs = ArenaBuilder(), []
    for _, (_, _, pf, src, pred) in cases:
        d = blank_desc()
        d.residual_off, d.residual_stride = ab.add(src), src.shape[1] << ss
        d.pred_off, d.pred_stride = ab.add(pred), pred.shape[1] << ss
        d.tx_type, d.shape, d.bit_depth, d.quant_mode = 0, pf, 8, abi.QUANT_NONE
        d.flags = abi.TX_FWD | abi.TX_SRC_PRED | abi.TX_SATD
        descs.append(d)
    _, res = run_batch(hip, ab, descs, size, size >> ss)
    for i, (k, (_, _, pf, src, pred)) in enumerate(cases):
        got = int(res[i, 12:16].view(np.uint32)[0]) << ss
        assert got == orc_tpl_cost(orc, size, ss, pf, src, pred), (size, ss, pf, i)
        assert got == int(gold[k])


@pytest.mark.parametrize("bd", [8, 10])
def test_tier_b_source_minus_prediction_to_recon(hip, orc, bd):
    """The whole TPL reconstruction step in one launch (src_ops_process.c:1132-1160): residual = source - prediction inside
    the kernel, forward DCT, quantise, inverse, add the prediction."""
    rng = np.random.default_rng(900 + bd)
    w = h = 16
    n = w * h
    iscan = np.arange(n, dtype=np.int16)
    ab, descs, expect = ArenaBuilder(), [], []
    iscan_off = ab.add(iscan)
    pix16 = bd == 10
    dt = np.uint16 if pix16 else np.uint8
    for i in range(120):
        tq = T.quant_tables(rng, bd)
        src = rng.integers(0, 1 << bd, size=(h, w + 9)).astype(dt)
        amp = (3, 20, 200)[i % 3]
        pred = np.clip(src[:, :w].astype(np.int32) + rng.integers(-amp, amp + 1, size=(h, w)), 0, (1 << bd) - 1).astype(dt)
        pp = np.zeros((h, w + 2), dt)
        pp[:, :w] = pred
        tt = (0, 1, 4, 9)[i % 4]
        d = blank_desc()
        d.residual_off, d.residual_stride = ab.add(src), w + 9
        d.pred_off, d.pred_stride = ab.add(pp), w + 2
        d.qcoeff_off, d.dqcoeff_off = ab.add(nbytes=n * 4), ab.add(nbytes=n * 4)
        d.recon_off, d.recon_stride = ab.add(nbytes=h * (w + 4) * (2 if pix16 else 1)), w + 4
        d.iscan_off = iscan_off
        mode = abi.QUANT_B_HBD if pix16 else abi.QUANT_B
        for k in range(2):
            d.zbin[k], d.round[k], d.quant[k] = int(tq["zbin"][k]), int(tq["round"][k]), int(tq["quant"][k])
            d.quant_shift[k], d.dequant[k] = int(tq["qshift"][k]), int(tq["dequant"][k])
        d.tx_type, d.shape, d.bit_depth, d.quant_mode, d.log_scale = tt, 0, bd, mode, 0
        d.flags = abi.TX_FWD | abi.TX_INV | abi.TX_SRC_PRED | abi.TX_SATD | (abi.TX_PIXEL16 if pix16 else 0)
        descs.append(d)
        # oracle: subtract -> forward -> satd / quantise -> inverse + prediction
        diff = np.zeros((h, w), np.int16)
        orc_fn = orc.orc_highbd_subtract_block if pix16 else orc.orc_subtract_block
        orc_fn(h, w, P(diff), PD(w), P(src), PD(w + 9), P(pp), PD(w + 2))
        co = np.zeros(n, np.int32)
        orc.orc_fwd_txfm2d(P(diff), P(co), C.c_uint32(w), w, h, tt, bd, 0)
        orc.orc_satd.restype = C.c_int
        satd = orc.orc_satd(P(co), n)
        qc, dq, eob = T.orc_quant(orc, 2 if pix16 else 1, dict(n=n, ls=0, coeff=co, scan=iscan, iscan=iscan, qm=None, iqm=None, t=tq))
        rec = np.zeros((h, w + 4), np.uint16)
        orc.orc_inv_txfm2d_add(P(dq), P(pp.astype(np.uint16)), w + 2, P(rec), w + 4, w, h, tt, bd)
        expect.append((qc, dq, eob, satd, rec, d))
    out, res = run_batch(hip, ab, descs, w, h)
    for i, (qc, dq, eob, satd, rec, d) in enumerate(expect):
        g = lambda off, cnt, t: out[off:off + cnt * np.dtype(t).itemsize].view(t)
        assert np.array_equal(g(d.qcoeff_off, n, np.int32), qc), ("qcoeff", i)
        assert np.array_equal(g(d.dqcoeff_off, n, np.int32), dq), ("dqcoeff", i)
        assert int(res[i, 8:10].view(np.uint16)[0]) == eob and int(res[i, 12:16].view(np.uint32)[0]) == satd, ("eob/satd", i)
        got = g(d.recon_off, h * (w + 4), dt).reshape(h, w + 4)
        assert np.array_equal(got[:, :w], rec[:, :w].astype(dt)), ("recon", i)


def test_plane_sse_device_planes(hip):
    """svt_hip_plane_sse (Tier B form of svt_spatial_full_distortion_kernel / svt_full_distortion_kernel16_bits: two DEVICE planes, the sum
    into device memory) against numpy; what picture_sse_calculations asks for after a trial of the deblocking level search."""
    rng = np.random.default_rng(17)
    for dt, is16, hi in ((np.uint8, 0, 256), (np.uint16, 1, 1024)):
        for (w, h, sa, sb) in ((64, 48, 80, 72), (1, 1, 8, 8), (1923, 1081, 2048, 1984), (200, 3, 208, 256)):
            a = rng.integers(0, hi, size=(h, sa), dtype=dt)
            b = rng.integers(0, hi, size=(h, sb), dtype=dt)
            da, db, dout = device.DeviceBuffer(hip, a.nbytes), device.DeviceBuffer(hip, b.nbytes), device.DeviceBuffer(hip, 8)
            da.upload(a), db.upload(b), dout.fill(0xAB)
            device.check(hip, hip.svt_hip_plane_sse(C.c_void_p(da.ptr), sa, C.c_void_p(db.ptr), sb, w, h, is16, C.c_void_p(dout.ptr), None), "svt_hip_plane_sse")
            device.check(hip, hip.svt_hip_stream_sync(None), "sync")
            got = int(dout.download(np.uint64, (1,))[0])
            want = int(((a[:, :w].astype(np.int64) - b[:, :w].astype(np.int64)) ** 2).sum())
            assert got == want, (dt, w, h)
    assert hip.svt_hip_plane_sse(None, 8, None, 8, 4, 4, 0, None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER

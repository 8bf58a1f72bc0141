"""GPU parity: transforms + quantisers of libsvtav1_hip (through the C-ABI) against the oracle, bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import tx_cases as T
from svtav1_hip import abi, device
from tx_cases import P, V

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "txfm.npz")


def hip_inverse(hip, w, h, co, pred, ps, rec, rs, tt, bd):
    fn = getattr(hip, f"svt_av1_inv_txfm2d_add_{w}x{h}_hip")
    if w == h:
        fn(P(co), P(pred), ps, P(rec), rs, tt, bd)
    elif (w, h) in ((4, 8), (8, 4), (4, 16), (16, 4)):
        fn(P(co), P(pred), ps, P(rec), rs, tt, 0, bd)
    else:
        fn(P(co), P(pred), ps, P(rec), rs, tt, 0, len(co), bd)


@pytest.mark.parametrize("w,h", T.SIZES)
def test_tier_a_fwd_inv(hip, orc, w, h):
    rng = np.random.default_rng(w * 100 + h)
    for tt in range(16):
        if not orc.orc_txfm_valid(w, h, tt):
            continue
        for bd in (8, 10):
            trial = (tt + bd) % 3
            res = T.residual(rng, w, h, bd, trial)
            for shape, suf in ((0, ""), (1, "_N2"), (2, "_N4")):
                o1, o2 = np.zeros(w * h, np.int32), np.full(w * h, 5, np.int32)
                orc.orc_fwd_txfm2d(P(res), P(o1), C.c_uint32(w + 3), w, h, tt, bd, shape)
                getattr(hip, f"svt_av1_fwd_txfm2d_{w}x{h}{suf}_hip")(P(res), P(o2), C.c_uint32(w + 3), tt, C.c_uint8(bd))
                assert np.array_equal(o1, o2), (w, h, tt, bd, trial, shape)
            for itrial in range(3):
                co = T.coeffs_for_inverse(rng, orc, w, h, tt, bd, itrial)
                pred = rng.integers(0, 1 << bd, size=(h, w + 5)).astype(np.uint16)
                r1, r2 = np.zeros((h, w + 7), np.uint16), np.zeros((h, w + 7), np.uint16)
                orc.orc_inv_txfm2d_add(P(co), P(pred), w + 5, P(r1), w + 7, w, h, tt, bd)
                hip_inverse(hip, w, h, co, pred, w + 5, r2, w + 7, tt, bd)
                assert np.array_equal(r1, r2), (w, h, tt, bd, itrial)
    if max(w, h) == 64:
        orc.orc_handle_transform64.restype = C.c_uint64
        for suf, en in (("", 1), ("_N2_N4", 0)):
            fn = getattr(hip, f"svt_handle_transform{w}x{h}{suf}_hip")
            fn.restype = C.c_uint64
            co = rng.integers(-100000, 100000, size=w * h).astype(np.int32)
            c2 = co.copy()
            e1 = orc.orc_handle_transform64(P(co), w, h) * en
            e2 = fn(P(c2))
            kw, kh = min(w, 32), min(h, 32)
            assert e1 == e2 and np.array_equal(co[:kw * kh], c2[:kw * kh])


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 32), (64, 16)])
def test_tier_a_fast_path_thresholds(hip, orc, w, h):
    """The forward kernels switch to 24-bit multiplies when the largest input magnitude of a wave is below a proven limit
    (txfm_device.hpp, FWD_FAST_LIMIT).  Sweep the residual magnitude across every such limit, with the sign patterns that
    maximise the intermediate values, so that both arithmetic paths and their hand-over are compared with the oracle."""
    rng = np.random.default_rng(4242 + w * 64 + h)
    for tt in (0, 1, 4, 9, 10, 11):   # DCT_DCT, ADST_DCT, FLIPADST_DCT, IDTX, V_DCT, H_DCT
        if not orc.orc_txfm_valid(w, h, tt):
            continue
        for mag in (60, 250, 700, 1023, 1500, 2300, 4100, 4200, 6000, 9000, 14000, 23000, 32767):
            for pat in range(3):
                if pat == 0:
                    res = rng.integers(-mag, mag + 1, size=(h, w + 3))
                elif pat == 1:
                    res = np.full((h, w + 3), mag)
                else:
                    res = (rng.integers(0, 2, size=(h, w + 3)) * 2 - 1) * mag
                res = res.astype(np.int16)
                o1, o2 = np.zeros(w * h, np.int32), np.full(w * h, 5, np.int32)
                orc.orc_fwd_txfm2d(P(res), P(o1), C.c_uint32(w + 3), w, h, tt, 10, 0)
                getattr(hip, f"svt_av1_fwd_txfm2d_{w}x{h}_hip")(P(res), P(o2), C.c_uint32(w + 3), tt, C.c_uint8(10))
                assert np.array_equal(o1, o2), (w, h, tt, mag, pat)


def test_tier_a_golden(hip):
    """HIP == committed outputs of the reference's own transform functions."""
    import test_txfm_oracle as TT
    gold = np.load(GOLD)
    for i, (w, h, tt, bd, res, pred) in enumerate(TT.golden_cases()):
        if f"fwd{i}" not in gold:
            continue
        co = np.zeros(w * h, np.int32)
        getattr(hip, f"svt_av1_fwd_txfm2d_{w}x{h}_hip")(P(res), P(co), C.c_uint32(w + 3), tt, C.c_uint8(bd))
        assert np.array_equal(co, gold[f"fwd{i}"]), (w, h, tt, bd)
        ci = co.reshape(h, w)[:min(h, 32), :min(w, 32)].copy().reshape(-1)
        rec = np.zeros((h, w + 7), np.uint16)
        hip_inverse(hip, w, h, ci, pred, w + 5, rec, w + 7, tt, bd)
        assert np.array_equal(rec, gold[f"inv{i}"]), (w, h, tt, bd)


def test_tier_a_quantizers(hip, orc):
    rng = np.random.default_rng(2)
    for trial in range(120):
        c = T.quant_case(rng, trial)
        n, t, ls = c["n"], c["t"], c["ls"]
        qm = P(c["qm"]) if c["qm"] is not None else None
        iqm = P(c["iqm"]) if c["iqm"] is not None else None

        def run(fn, rnd, qnt, *tail):
            qc, dq, eob = np.full(n, 7, np.int32), np.full(n, 7, np.int32), C.c_uint16(9999)
            fn(P(c["coeff"]), C.c_ssize_t(n), P(t["zbin"]), P(rnd), P(qnt), P(t["qshift"]), P(qc), P(dq), P(t["dequant"]),
               C.byref(eob), P(c["scan"]), P(c["iscan"]), *tail)
            return qc, dq, eob.value
        pairs = [(run(hip.svt_aom_quantize_b_hip, t["round"], t["quant"], qm, iqm, C.c_int32(ls)), T.orc_quant(orc, 1, c)),
                 (run(hip.svt_av1_quantize_b_qm_hip, t["round"], t["quant"], qm, iqm, C.c_int32(ls)), T.orc_quant(orc, 1, c)),
                 (run(hip.svt_aom_highbd_quantize_b_hip, t["round"], t["quant"], qm, iqm, C.c_int32(ls)), T.orc_quant(orc, 2, c)),
                 (run(hip.svt_av1_highbd_quantize_b_qm_hip, t["round"], t["quant"], qm, iqm, C.c_int32(ls)), T.orc_quant(orc, 2, c)),
                 (run(hip.svt_av1_quantize_fp_qm_hip, t["round_fp"], t["quant_fp"], qm, iqm, C.c_int16(ls)), T.orc_quant(orc, 3, c)),
                 (run(hip.svt_av1_highbd_quantize_fp_qm_hip, t["round_fp"], t["quant_fp"], qm, iqm, C.c_int16(ls)), T.orc_quant(orc, 4, c))]
        if c["qm"] is None:
            f = (hip.svt_av1_quantize_fp_hip, hip.svt_av1_quantize_fp_32x32_hip, hip.svt_av1_quantize_fp_64x64_hip)[ls]
            pairs += [(run(f, t["round_fp"], t["quant_fp"]), T.orc_quant(orc, 3, c)),
                      (run(hip.svt_av1_highbd_quantize_fp_hip, t["round_fp"], t["quant_fp"], C.c_int16(ls)), T.orc_quant(orc, 4, c))]
        for k, (a, b) in enumerate(pairs):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (trial, k)


class ArenaBuilder:
    def __init__(self):
        self.chunks, self.size = [], 0

    def add(self, arr=None, nbytes=None):
        off = self.size
        nbytes = arr.nbytes if arr is not None else nbytes
        self.chunks.append((off, None if arr is None else np.ascontiguousarray(arr).view(np.uint8).reshape(-1)))
        self.size += (nbytes + 255) // 256 * 256 + 256
        return off

    def build(self):
        buf = np.zeros(self.size, np.uint8)
        for off, a in self.chunks:
            if a is not None:
                buf[off:off + a.size] = a
        return buf


@pytest.mark.parametrize("w,h", T.SIZES)
def test_tier_b_fused_batch(hip, orc, w, h):
    """residual -> fwd -> [energy/repack] -> quantise -> inverse -> recon for hundreds of blocks in one launch,
    every stage compared with the oracle pipeline."""
    rng = np.random.default_rng(1000 + w * 100 + h)
    n_tb = 300 if w * h <= 1024 else 60
    iw, ih = min(w, 32), min(h, 32)
    n = iw * ih
    ls = 2 if max(w, h) == 64 and (w * h) > 1024 else (1 if w * h > 256 and max(w, h) >= 32 and min(w, h) >= 16 else 0)
    types = [tt for tt in range(16) if orc.orc_txfm_valid(w, h, tt)]
    scan = rng.permutation(n).astype(np.int16)
    iscan = np.empty(n, np.int16)
    iscan[scan] = np.arange(n)
    ab = ArenaBuilder()
    iscan_off = ab.add(iscan)
    descs, expect = [], []
    for i in range(n_tb):
        bd = 8 if i % 3 == 0 else 10
        pix16 = bd == 10 or i % 2 == 0
        tt = types[i % len(types)]
        shape = (0, 0, 1, 2)[i % 4]
        mode = 1 + (i % 4)
        tq = T.quant_tables(rng, bd)
        res = T.residual(rng, w, h, bd, 0, pad=5) // (1 + (i % 5))
        res = res.astype(np.int16)
        pred16 = rng.integers(0, 1 << bd, size=(h, w + 2)).astype(np.uint16)
        d = abi.TxfmDesc()
        d.residual_off, d.residual_stride = ab.add(res), w + 5
        d.coeff_off = ab.add(nbytes=n * 4) if i % 2 else abi.NO_OFFSET
        d.qcoeff_off, d.dqcoeff_off = ab.add(nbytes=n * 4), ab.add(nbytes=n * 4)
        d.pred_off = ab.add(pred16 if pix16 else pred16.astype(np.uint8))
        d.recon_off = ab.add(nbytes=h * (w + 4) * (2 if pix16 else 1))
        d.pred_stride, d.recon_stride = w + 2, w + 4
        d.iscan_off, d.qm_off, d.iqm_off = iscan_off, abi.NO_OFFSET, abi.NO_OFFSET
        rnd, qnt = (tq["round"], tq["quant"]) if mode <= 2 else (tq["round_fp"], tq["quant_fp"])
        for k in range(2):
            d.zbin[k], d.round[k], d.quant[k] = int(tq["zbin"][k]), int(rnd[k]), int(qnt[k])
            d.quant_shift[k], d.dequant[k] = int(tq["qshift"][k]), int(tq["dequant"][k])
        d.tx_type, d.shape, d.bit_depth, d.quant_mode, d.log_scale = tt, shape, bd, mode, ls
        d.flags = abi.TX_FWD | abi.TX_INV | (abi.TX_PIXEL16 if pix16 else 0)
        if i % 6 == 1:      # a block cut by the picture edge: the caller's cropped_tx_width / cropped_tx_height
            d.dist_w, d.dist_h = max(1, iw - 3), max(1, ih // 2)
        descs.append(d)
        # oracle pipeline
        co = np.zeros(w * h, np.int32)
        orc.orc_fwd_txfm2d(P(res), P(co), C.c_uint32(w + 5), w, h, tt, bd, shape)
        energy = 0
        if max(w, h) == 64:
            orc.orc_handle_transform64.restype = C.c_uint64
            energy = orc.orc_handle_transform64(P(co), w, h)
        co = co[:n].copy()
        qc, dq, eob = T.orc_quant(orc, mode, dict(n=n, ls=ls, coeff=co, scan=scan, iscan=iscan, qm=None, iqm=None, t=tq))
        rec = np.zeros((h, w + 4), np.uint16)
        orc.orc_inv_txfm2d_add(P(dq), P(pred16), w + 2, P(rec), w + 4, w, h, tt, bd)
        expect.append((co, qc, dq, eob, energy, rec if pix16 else rec.astype(np.uint8), pix16, d))
    arena = ab.build()
    darena = device.DeviceBuffer(hip, arena.nbytes + 256)
    darena.upload(arena)
    darr = (abi.TxfmDesc * n_tb)(*descs)
    ddesc = device.DeviceBuffer(hip, C.sizeof(darr))
    ddesc.upload(np.frombuffer(darr, dtype=np.uint8))
    dres = device.DeviceBuffer(hip, abi.TXFM_RESULT_BYTES * n_tb)
    device.check(hip, hip.svt_hip_txfm_quant_batch(V(darena.ptr), V(ddesc.ptr), V(dres.ptr), C.c_uint32(n_tb), C.c_uint32(w),
                                                   C.c_uint32(h), None), "svt_hip_txfm_quant_batch")
    ddist = device.DeviceBuffer(hip, 16 * n_tb)
    device.check(hip, hip.svt_hip_txfm_distortion_batch(V(darena.ptr), V(ddesc.ptr), V(ddist.ptr), C.c_uint32(n_tb), C.c_uint32(w),
                                                        C.c_uint32(h), None), "svt_hip_txfm_distortion_batch")
    out = darena.download(np.uint8, (arena.nbytes,))
    res_raw = dres.download(np.uint8, (n_tb, abi.TXFM_RESULT_BYTES))
    dist = ddist.download(np.uint64, (n_tb, 2))
    for i, (co, qc, dq, eob, energy, rec, pix16, d) in enumerate(expect):
        g = lambda off, cnt, dt: out[off:off + cnt * np.dtype(dt).itemsize].view(dt)
        if d.coeff_off != abi.NO_OFFSET:
            assert np.array_equal(g(d.coeff_off, n, np.int32), co), ("coeff", i)
        assert np.array_equal(g(d.qcoeff_off, n, np.int32), qc), ("qcoeff", i)
        assert np.array_equal(g(d.dqcoeff_off, n, np.int32), dq), ("dqcoeff", i)
        assert int(res_raw[i, 8:10].view(np.uint16)[0]) == eob, ("eob", i)
        assert int(res_raw[i, :8].view(np.uint64)[0]) == energy, ("energy", i)
        got = g(d.recon_off, h * (w + 4), np.uint16 if pix16 else np.uint8).reshape(h, w + 4)
        assert np.array_equal(got[:, :w], rec[:, :w]), ("recon", i)
        want = np.zeros(2, np.uint64)
        if d.coeff_off != abi.NO_OFFSET:   # svt_aom_picture_full_distortion32_bits_single over the (cropped) area
            orc.orc_full_distortion32(P(co), iw, P(dq), iw, P(want), d.dist_w or iw, d.dist_h or ih)
        assert np.array_equal(dist[i], want), ("distortion", i)


def test_quantize_batch(hip, orc):
    rng = np.random.default_rng(77)
    n, n_tb = 1024, 40
    ab = ArenaBuilder()
    descs, expect = [], []
    for i in range(n_tb):
        c = T.quant_case(rng, i)
        c["n"] = n
        c["coeff"] = rng.integers(-3000, 3000, size=n).astype(np.int32)
        c["scan"] = rng.permutation(n).astype(np.int16)
        c["iscan"] = np.empty(n, np.int16)
        c["iscan"][c["scan"]] = np.arange(n)
        c["qm"] = c["iqm"] = None
        mode = 1 + i % 4
        t = c["t"]
        d = abi.TxfmDesc()
        d.coeff_off, d.iscan_off = ab.add(c["coeff"]), ab.add(c["iscan"])
        d.qcoeff_off, d.dqcoeff_off = ab.add(nbytes=n * 4), ab.add(nbytes=n * 4)
        d.qm_off = d.iqm_off = abi.NO_OFFSET
        rnd, qnt = (t["round"], t["quant"]) if mode <= 2 else (t["round_fp"], t["quant_fp"])
        for k in range(2):
            d.zbin[k], d.round[k], d.quant[k] = int(t["zbin"][k]), int(rnd[k]), int(qnt[k])
            d.quant_shift[k], d.dequant[k] = int(t["qshift"][k]), int(t["dequant"][k])
        d.quant_mode, d.log_scale = mode, c["ls"]
        descs.append(d)
        expect.append(T.orc_quant(orc, mode, c))
    arena = ab.build()
    darena = device.DeviceBuffer(hip, arena.nbytes + 256)
    darena.upload(arena)
    darr = (abi.TxfmDesc * n_tb)(*descs)
    ddesc = device.DeviceBuffer(hip, C.sizeof(darr))
    ddesc.upload(np.frombuffer(darr, dtype=np.uint8))
    dres = device.DeviceBuffer(hip, abi.TXFM_RESULT_BYTES * n_tb)
    device.check(hip, hip.svt_hip_quantize_batch(V(darena.ptr), V(ddesc.ptr), V(dres.ptr), C.c_uint32(n_tb), C.c_uint32(n), None),
                 "svt_hip_quantize_batch")
    out = darena.download(np.uint8, (arena.nbytes,))
    res_raw = dres.download(np.uint8, (n_tb, abi.TXFM_RESULT_BYTES))
    for i, (qc, dq, eob) in enumerate(expect):
        d = descs[i]
        assert np.array_equal(out[d.qcoeff_off:d.qcoeff_off + 4 * n].view(np.int32), qc)
        assert np.array_equal(out[d.dqcoeff_off:d.dqcoeff_off + 4 * n].view(np.int32), dq)
        assert int(res_raw[i, 8:10].view(np.uint16)[0]) == eob


@pytest.mark.parametrize("w,h", [(8, 8), (16, 16), (32, 32), (64, 64)])
def test_tier_b_4k_frame_properties(hip, orc, w, h):
    """BASELINE.json configs[2] size: every luma transform block of one 3840x2160 10-bit picture through the fused kernel in
    one launch.  Checked through (a) a random sample of blocks against the oracle pipeline, (b) order independence: the same
    blocks launched in a permuted descriptor order give byte-identical coefficient, eob and reconstruction data."""
    W4, H4, bd = 3840, 2160, 10
    rng = np.random.default_rng(2160 + w)
    resid = (rng.integers(-400, 401, size=(H4, W4)) // rng.integers(1, 9, size=(H4, W4))).astype(np.int16)
    pred = rng.integers(0, 1 << bd, size=(H4, W4), dtype=np.uint16)
    bw, bh = W4 // w, H4 // h
    nblk = bw * bh
    iw, ih = min(w, 32), min(h, 32)
    n = iw * ih
    off_res, off_pred = 0, W4 * H4 * 2
    off_rec, off_q = off_pred + W4 * H4 * 2, off_pred + 2 * W4 * H4 * 2
    off_dq = off_q + nblk * n * 4
    off_iscan = off_dq + nblk * n * 4
    total = off_iscan + n * 2 + 512
    scan = rng.permutation(n).astype(np.int16)
    iscan = np.empty(n, np.int16)
    iscan[scan] = np.arange(n)
    tq = T.quant_tables(rng, bd)
    ls = 2 if w == 64 else (1 if w == 32 else 0)
    types = [tt for tt in range(16) if orc.orc_txfm_valid(w, h, tt)]
    i = np.arange(nblk, dtype=np.uint64)
    pix = (i // bw * h) * W4 + (i % bw) * w
    descs = np.zeros(nblk, dtype=np.dtype(abi.TxfmDesc))
    descs["residual_off"], descs["residual_stride"] = off_res + pix * 2, W4
    descs["coeff_off"] = abi.NO_OFFSET
    descs["qcoeff_off"], descs["dqcoeff_off"] = off_q + i * (n * 4), off_dq + i * (n * 4)
    descs["pred_off"], descs["recon_off"], descs["pred_stride"], descs["recon_stride"] = off_pred + pix * 2, off_rec + pix * 2, W4, W4
    descs["iscan_off"], descs["qm_off"], descs["iqm_off"] = off_iscan, abi.NO_OFFSET, abi.NO_OFFSET
    for k in range(2):
        descs["zbin"][:, k], descs["round"][:, k], descs["quant"][:, k] = int(tq["zbin"][k]), int(tq["round"][k]), int(tq["quant"][k])
        descs["quant_shift"][:, k], descs["dequant"][:, k] = int(tq["qshift"][k]), int(tq["dequant"][k])
    descs["tx_type"] = np.array(types, np.uint8)[(i * 7 + i // bw) % len(types)]
    descs["shape"], descs["bit_depth"], descs["quant_mode"], descs["log_scale"] = 0, bd, abi.QUANT_B_HBD, ls
    descs["flags"] = abi.TX_FWD | abi.TX_INV | abi.TX_PIXEL16 | abi.TX_SATD
    darena = device.DeviceBuffer(hip, total)
    ddesc = device.DeviceBuffer(hip, descs.nbytes)
    dres = device.DeviceBuffer(hip, abi.TXFM_RESULT_BYTES * nblk)

    def run(order):
        darena.fill(0)
        for off, a in ((off_res, resid), (off_pred, pred), (off_iscan, iscan)):
            device.check(hip, hip.svt_hip_upload(V(darena.ptr + off), P(a), C.c_size_t(a.nbytes), None), "svt_hip_upload")
        device.check(hip, hip.svt_hip_stream_sync(None), "svt_hip_stream_sync")
        ddesc.upload(np.ascontiguousarray(descs[order]).view(np.uint8))
        device.check(hip, hip.svt_hip_txfm_quant_batch(V(darena.ptr), V(ddesc.ptr), V(dres.ptr), C.c_uint32(nblk), C.c_uint32(w),
                                                       C.c_uint32(h), None), "svt_hip_txfm_quant_batch")
        out = darena.download(np.uint8, (total,))
        res = dres.download(np.uint8, (nblk, abi.TXFM_RESULT_BYTES))
        back = np.empty_like(res)
        back[order] = res                      # result i belongs to descriptor order[i]
        return out, back

    ident = np.arange(nblk)
    out, res = run(ident)
    out2, res2 = run(rng.permutation(nblk))
    assert np.array_equal(out[off_rec:off_iscan], out2[off_rec:off_iscan]) and np.array_equal(res, res2)
    rec_plane = out[off_rec:off_rec + W4 * H4 * 2].view(np.uint16).reshape(H4, W4)
    orc.orc_satd.restype = C.c_int
    for b in rng.choice(nblk, size=60, replace=False):
        by, bx = (b // bw) * h, (b % bw) * w
        tt = int(descs["tx_type"][b])
        r = np.ascontiguousarray(resid[by:by + h, bx:bx + w])
        co = np.zeros(w * h, np.int32)
        orc.orc_fwd_txfm2d(P(r), P(co), C.c_uint32(w), w, h, tt, bd, 0)
        if max(w, h) == 64:
            orc.orc_handle_transform64.restype = C.c_uint64
            orc.orc_handle_transform64(P(co), w, h)
        co = co[:n].copy()
        qc, dq, eob = T.orc_quant(orc, 2, dict(n=n, ls=ls, coeff=co, scan=scan, iscan=iscan, qm=None, iqm=None, t=tq))
        p = np.ascontiguousarray(pred[by:by + h, bx:bx + w])
        rec = np.zeros((h, w), np.uint16)
        orc.orc_inv_txfm2d_add(P(dq), P(p), w, P(rec), w, w, h, tt, bd)
        assert np.array_equal(out[off_q + b * n * 4:off_q + (b + 1) * n * 4].view(np.int32), qc), ("qcoeff", b)
        assert np.array_equal(out[off_dq + b * n * 4:off_dq + (b + 1) * n * 4].view(np.int32), dq), ("dqcoeff", b)
        assert int(res[b, 8:10].view(np.uint16)[0]) == eob and int(res[b, 12:16].view(np.uint32)[0]) == orc.orc_satd(P(co), n), ("eob/satd", b)
        assert np.array_equal(rec_plane[by:by + h, bx:bx + w], rec), ("recon", b)

"""GPU parity: the temporal filter's accumulate / central / normalise stage (through the C-ABI) against the oracle and the
golden vectors, bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import tf_cases as F
from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu
V = C.c_void_p
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf.npz")


class DevBlocks:
    """Device copies of a list of host block cases + the descriptor array pointing at them."""

    def __init__(self, hip, cases):
        self.hip, self.cases, self.bufs, descs = hip, cases, [], []
        for b, a in cases:
            d = abi.TfBlock.from_buffer_copy(b)
            for pl in range(3):
                row = []
                for arr in a[pl]:
                    db = device.DeviceBuffer(hip, arr.nbytes)
                    db.upload(arr)
                    row.append(db)
                self.bufs.append(row)
                d.src[pl], d.pred[pl], d.accum[pl], d.count[pl] = (x.ptr for x in row)
            descs.append(d)
        arr = (abi.TfBlock * len(descs))(*descs)
        self.ddesc = device.DeviceBuffer(hip, C.sizeof(arr))
        self.ddesc.upload(np.frombuffer(arr, np.uint8))
        self.n = len(descs)

    def planes(self, i, pl):
        acc, cnt = self.bufs[3 * i + pl][2], self.bufs[3 * i + pl][3]
        ref = self.cases[i][1][pl]
        return acc.download(np.uint32, ref[2].shape), cnt.download(np.uint16, ref[3].shape)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_accumulate_batch(hip, orc, bd):
    cases = [F.block_case(t, bd) for t in range(40)]
    dev = DevBlocks(hip, cases)
    device.check(hip, hip.svt_hip_tf_accumulate_batch(V(dev.ddesc.ptr), dev.n, None), "svt_hip_tf_accumulate_batch")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    for i, (b, a) in enumerate(cases):
        orc.orc_tf_accumulate(C.byref(b))
        for pl in range(3):
            acc, cnt = dev.planes(i, pl)
            assert np.array_equal(acc, a[pl][2]) and np.array_equal(cnt, a[pl][3]), (i, pl)
    assert hip.svt_hip_tf_accumulate_batch(None, 0, None) == abi.SVT_HIP_ERR_BAD_PARAMETER


def test_golden(hip):
    g = np.load(GOLD)
    k = 0
    for bd in (8, 10):
        cases = [F.block_case(t, bd, seed=3) for t in range(8)]
        dev = DevBlocks(hip, cases)
        device.check(hip, hip.svt_hip_tf_accumulate_batch(V(dev.ddesc.ptr), dev.n, None), "svt_hip_tf_accumulate_batch")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
        for i in range(len(cases)):
            for pl in range(3):
                acc, cnt = dev.planes(i, pl)
                n = cases[i][1][pl][0].shape[0]
                assert np.array_equal(acc[:, :n], g[f"acc{k}_{pl}"]) and np.array_equal(cnt[:, :n], g[f"cnt{k}_{pl}"]), (k, pl)
            k += 1


@pytest.mark.parametrize("bd", [8, 10])
def test_central_accumulate_normalise_chain(hip, orc, bd):
    """The per-block pipeline of produce_temporally_filtered_pic: the centre picture's own weight, three reference
    pictures accumulated one launch each, then the normalisation — against the same chain through the oracle."""
    n_blk, n_ref = 24, 3
    base = [F.block_case(t, bd, seed=11) for t in range(n_blk)]
    dev = DevBlocks(hip, base)
    device.check(hip, hip.svt_hip_tf_central_batch(V(dev.ddesc.ptr), dev.n, None), "svt_hip_tf_central_batch")
    for b, _ in base:
        orc.orc_tf_central(C.byref(b))
    for r in range(n_ref):
        preds = [F.block_case(t, bd, seed=20 + r) for t in range(n_blk)]     # another reference: new predictions / MVs / errors
        keep = []
        for i, ((b, a), (pb, pa)) in enumerate(zip(base, preds)):
            for pl in range(3):
                a[pl][1][:] = pa[pl][1]                                       # host prediction for the oracle
                dev.bufs[3 * i + pl][1].upload(pa[pl][1])
            for f in ("split", "mv_dist_th"):
                setattr(b, f, getattr(pb, f))
            for k in range(4):
                b.mv_x[k], b.mv_y[k], b.block_error[k] = pb.mv_x[k], pb.mv_y[k], pb.block_error[k]
            keep.append(pa)
        descs = []
        for i, (b, a) in enumerate(base):
            d = abi.TfBlock.from_buffer_copy(b)
            for pl in range(3):
                d.src[pl], d.pred[pl], d.accum[pl], d.count[pl] = (x.ptr for x in dev.bufs[3 * i + pl])
            descs.append(d)
        arr = (abi.TfBlock * n_blk)(*descs)
        dev.ddesc.upload(np.frombuffer(arr, np.uint8))
        device.check(hip, hip.svt_hip_tf_accumulate_batch(V(dev.ddesc.ptr), n_blk, None), "svt_hip_tf_accumulate_batch")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
        for b, _ in base:
            orc.orc_tf_accumulate(C.byref(b))
    dt = np.uint16 if bd > 8 else np.uint8
    outs_h, outs_d, odesc, odesc_h = [], [], [], []
    for i in range(n_blk):
        oh, od = abi.TfOut(), abi.TfOut()
        for pl in range(3):
            n = 32 if pl == 0 else 16
            hbuf = np.zeros((n, n + 6), dt)
            dbuf = device.DeviceBuffer(hip, hbuf.nbytes)
            dbuf.fill(0)
            outs_h.append(hbuf), outs_d.append(dbuf)
            oh.dst[pl], od.dst[pl], oh.dst_stride[pl], od.dst_stride[pl] = hbuf.ctypes.data, dbuf.ptr, n + 6, n + 6
        odesc.append(od), odesc_h.append(oh)
    oarr = (abi.TfOut * n_blk)(*odesc)
    dout = device.DeviceBuffer(hip, C.sizeof(oarr))
    dout.upload(np.frombuffer(oarr, np.uint8))
    device.check(hip, hip.svt_hip_tf_normalise_batch(V(dev.ddesc.ptr), V(dout.ptr), n_blk, None), "svt_hip_tf_normalise_batch")
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    for i, (b, a) in enumerate(base):
        orc.orc_tf_normalise(C.byref(b), C.byref(odesc_h[i]))
        for pl in range(3 if b.chroma else 1):
            acc, cnt = dev.planes(i, pl)
            assert np.array_equal(acc, a[pl][2]) and np.array_equal(cnt, a[pl][3]), ("accum", i, pl)
            got = outs_d[3 * i + pl].download(dt, outs_h[3 * i + pl].shape)
            assert np.array_equal(got, outs_h[3 * i + pl]), ("pixels", i, pl)


@pytest.mark.parametrize("bd,ss", [(8, 1), (10, 1), (8, 0), (10, 0)])
def test_filter_blocks_one_launch(hip, orc, bd, ss):
    """svt_hip_tf_filter_blocks — central + every reference + normalise in one launch, accumulators in registers — writes the pixels the
    three-step chain of the oracle writes (zero, one and four reference pictures)."""
    n_blk = 24
    dt = np.uint16 if bd > 8 else np.uint8
    for n_ref in (0, 1, 4):
        base = [F.block_case(t, bd, seed=31, ss=ss) for t in range(n_blk)]
        dev = DevBlocks(hip, base)                                              # static records: source, geometry
        for b, _ in base:
            orc.orc_tf_central(C.byref(b))
        lists, keep = [], []
        for r in range(n_ref):
            preds = [F.block_case(t, bd, seed=40 + r, ss=ss) for t in range(n_blk)]
            descs = []
            for i, ((b, a), (pb, pa)) in enumerate(zip(base, preds)):
                for pl in range(3):
                    a[pl][1][:] = pa[pl][1]
                for f in ("split", "mv_dist_th"):
                    setattr(b, f, getattr(pb, f))
                for k in range(4):
                    b.mv_x[k], b.mv_y[k], b.block_error[k] = pb.mv_x[k], pb.mv_y[k], pb.block_error[k]
                orc.orc_tf_accumulate(C.byref(b))
                d = abi.TfBlock.from_buffer_copy(b)
                for pl in range(3):
                    pbuf = device.DeviceBuffer(hip, pa[pl][1].nbytes)
                    pbuf.upload(pa[pl][1])
                    keep.append(pbuf)
                    d.src[pl], d.pred[pl], d.accum[pl], d.count[pl] = dev.bufs[3 * i + pl][0].ptr, pbuf.ptr, None, None
                descs.append(d)
            arr = (abi.TfBlock * n_blk)(*descs)
            dl = device.DeviceBuffer(hip, C.sizeof(arr))
            dl.upload(np.frombuffer(arr, np.uint8))
            lists.append(dl)
        outs_h, outs_d, odesc, odesc_h = [], [], [], []
        for i in range(n_blk):
            oh, od = abi.TfOut(), abi.TfOut()
            for pl in range(3):
                n = 32 if pl == 0 else 32 >> ss
                hbuf = np.zeros((n, n + 6), dt)
                dbuf = device.DeviceBuffer(hip, hbuf.nbytes)
                dbuf.fill(0)
                outs_h.append(hbuf), outs_d.append(dbuf)
                oh.dst[pl], od.dst[pl], oh.dst_stride[pl], od.dst_stride[pl] = hbuf.ctypes.data, dbuf.ptr, n + 6, n + 6
            odesc.append(od), odesc_h.append(oh)
        oarr = (abi.TfOut * n_blk)(*odesc)
        dout = device.DeviceBuffer(hip, C.sizeof(oarr))
        dout.upload(np.frombuffer(oarr, np.uint8))
        ptrs = (V * max(n_ref, 1))(*[V(d.ptr) for d in lists])
        device.check(hip, hip.svt_hip_tf_filter_blocks(ptrs, n_ref, V(dev.ddesc.ptr), V(dout.ptr), n_blk, ss, ss, None), "svt_hip_tf_filter_blocks")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
        for i, (b, a) in enumerate(base):
            orc.orc_tf_normalise(C.byref(b), C.byref(odesc_h[i]))
            for pl in range(3 if b.chroma else 1):
                got = outs_d[3 * i + pl].download(dt, outs_h[3 * i + pl].shape)
                assert np.array_equal(got, outs_h[3 * i + pl]), ("pixels", n_ref, i, pl)
    assert hip.svt_hip_tf_filter_blocks(None, 1, V(dev.ddesc.ptr), V(dout.ptr), n_blk, ss, ss, None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_tf_filter_blocks(ptrs, 33, V(dev.ddesc.ptr), V(dout.ptr), n_blk, ss, ss, None) == abi.SVT_HIP_ERR_BAD_PARAMETER


def test_noise_estimate(hip, orc):
    """svt_estimate_noise_fp16_hip / svt_estimate_noise_highbd_fp16_hip (Tier A) and svt_hip_tf_estimate_noise (Tier B, device
    plane) against the oracle and the values the real reference produced (golden/tf_noise.npz) — bit-exact."""
    from test_tf_oracle import GOLD_NOISE, orc_noise
    gold = np.load(GOLD_NOISE)["noise"]
    hip.svt_estimate_noise_fp16_hip.restype = C.c_int32
    hip.svt_estimate_noise_highbd_fp16_hip.restype = C.c_int32
    rec = np.dtype([("sum", "<u8"), ("num", "<u8"), ("noise", "<i4"), ("pad", "<i4")])
    for i, (img, w, h, stride, bd) in enumerate(F.noise_cases()):
        exp = orc_noise(orc, img, w, h, stride, bd)
        assert exp == int(gold[i])
        if bd == 8:
            got = hip.svt_estimate_noise_fp16_hip(C.c_void_p(img.ctypes.data), C.c_uint16(w), C.c_uint16(h), C.c_uint16(stride))
        else:
            got = hip.svt_estimate_noise_highbd_fp16_hip(C.c_void_p(img.ctypes.data), w, h, stride, bd)
        assert got == exp, (w, h, bd)
        dsrc, dout = device.DeviceBuffer(hip, img.nbytes), device.DeviceBuffer(hip, rec.itemsize)
        dsrc.upload(img.view(np.uint8).reshape(-1))
        device.check(hip, hip.svt_hip_tf_estimate_noise(C.c_void_p(dsrc.ptr), C.c_uint32(w), C.c_uint32(h), C.c_uint32(stride), int(bd > 8), bd,
                                                        C.c_void_p(dout.ptr), None), "svt_hip_tf_estimate_noise")
        out = dout.download(np.uint8, rec.itemsize).view(rec)[0]
        assert int(out["noise"]) == exp and (int(out["num"]) >= 16) == (exp != -65536), (w, h, bd)

"""CPU: the synthetic inputs and byte counts bench.py uses are what it says they are, and the CPU-baseline harness computes
the same results with the reference's C table and with its x86 intrinsics table (so the two baselines time the same work)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

from benchlib.scan import zigzag_scan  # noqa: E402


def test_zigzag_is_the_reference_default_scan(ref):
    """benchlib.scan.zigzag_scan == av1_scan_orders[tx_size][DCT_DCT] (coefficients.h:2197)."""
    for txs, (w, h) in {0: (4, 4), 1: (8, 8), 2: (16, 16), 3: (32, 32), 4: (32, 32)}.items():
        n = w * h
        s, i = np.zeros(n, np.int16), np.zeros(n, np.int16)
        assert ref.ref_scan_table(txs, 0, s.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), n) == 0
        zs, zi = zigzag_scan(w, h)
        assert np.array_equal(s, zs) and np.array_equal(i, zi)


def test_zigzag_is_a_permutation():
    for n in (4, 8, 16, 32):
        s, i = zigzag_scan(n, n)
        assert sorted(s.tolist()) == list(range(n * n)) and np.array_equal(s[i], np.arange(n * n))
        assert s[:4].tolist() == [0, 1, n, 2 * n]


def test_txfm_tiling_covers_every_sample_once():
    import bench
    W, H = 3840, 2160
    cover = [np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)]
    for (p, x0, y0, rw, rh, w, h) in bench.txfm_tiling(W, H):
        assert rw % w == 0 and rh % h == 0
        cover[p][y0:y0 + rh, x0:x0 + rw] += 1
    assert all((c == 1).all() for c in cover)
    assert sum(c.size for c in cover) == 12441600          # 12.44 M coefficients per 4K 4:2:0 picture


def test_algorithmic_bytes():
    import bench
    assert bench.algorithmic_bytes_txfm(16, 16, 1) == 14 * 256          # 10 N + 2 d N at d = 2
    assert bench.algorithmic_bytes_txfm(64, 64, 1) == 8 * 4096          # a 64-point side keeps 32 coefficients
    assert int(bench.algorithmic_bytes_me(3840, 2160, 5)) == int(1.3125 * 3840 * 2160 * 6 + 2040 * 5 * 680)


def test_cpu_txfm_baseline_c_equals_simd(ref):
    """The reference's drivers give identical reconstructions with the C table and with the intrinsics table."""
    import bench
    if not bench.cpu_has_avx2():
        pytest.skip("no AVX2 on this CPU")
    ref.ref_set_simd.restype = C.c_int
    ref.ref_txfm_pass.restype = C.c_uint64
    rng = np.random.default_rng(2)
    out = {}
    for lvl in (0, 1):
        assert ref.ref_set_simd(lvl) == (0 if lvl == 0 else ref.ref_set_simd(1))
        recs = []
        for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64)):
            r = np.random.default_rng(w)
            res = r.integers(-700, 701, size=(128, 256), dtype=np.int16)
            pred = r.integers(0, 1024, size=(128, 256), dtype=np.uint16)
            rec = np.zeros_like(pred)
            t = bench.RefTxfmPass()
            t.residual, t.pred, t.recon, t.stride = res.ctypes.data, pred.ctypes.data, rec.ctypes.data, 256
            t.x0, t.y0, t.rw, t.rh, t.w, t.h = 0, 0, 256, 128, w, h
            t.tx_size, t.bit_depth, t.log_scale = bench.TX_SIZE_ENUM[(w, h)], 10, 2 if w == 64 else (1 if w == 32 else 0)
            t.tx_type[0], t.tx_type[1] = 0, (1 if w <= 16 else 0)
            for k, v in bench.QUANT.items():
                getattr(t, k)[0], getattr(t, k)[1] = v
            sc, isc = zigzag_scan(min(w, 32), min(h, 32))
            t.scan, t.iscan = sc.ctypes.data, isc.ctypes.data
            chk = ref.ref_txfm_pass(C.byref(t))
            recs.append((chk, rec))
        out[lvl] = recs
    ref.ref_set_simd(0)
    for (c0, r0), (c1, r1) in zip(out[0], out[1]):
        assert c0 == c1 and np.array_equal(r0, r1) and r0.any()

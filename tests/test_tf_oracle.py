"""CPU: the temporal-filter oracle (oracle/src/orc_tf.c) against the REAL reference — svt_av1_apply_temporal_filter_planewise_medium
and its hbd form reached through oracle/ref_harness_tf.c, which fills a real MeContext from the same flat block — and against
the committed golden vectors (tests/golden/tf.npz, written by tests/golden/make_golden_tf.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import tf_cases as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf.npz")
GOLD_NOISE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_noise.npz")    # tests/golden/make_golden_tf.py


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_accumulate_vs_reference(orc, ref, bd):
    for trial in range(40):
        b1, a1 = F.block_case(trial, bd)
        b2, a2 = F.block_case(trial, bd)
        assert ref.ref_tf_block_accumulate(C.byref(b1)) == 0
        orc.orc_tf_accumulate(C.byref(b2))
        for pl in range(3):
            assert np.array_equal(a1[pl][2], a2[pl][2]) and np.array_equal(a1[pl][3], a2[pl][3]), (trial, pl)


def test_golden(orc):
    g = np.load(GOLD)
    k = 0
    for bd in (8, 10):
        for trial in range(8):
            b, a = F.block_case(trial, bd, seed=3)
            _, a0 = F.block_case(trial, bd, seed=3)
            orc.orc_tf_accumulate(C.byref(b))
            for pl in range(3):
                n = a[pl][0].shape[0]
                assert np.array_equal(a[pl][2][:, :n], g[f"acc{k}_{pl}"]) and np.array_equal(a[pl][3][:, :n], g[f"cnt{k}_{pl}"]), (k, pl)
                assert np.array_equal(a[pl][2][:, n:], a0[pl][2][:, n:]) and np.array_equal(a[pl][3][:, n:], a0[pl][3][:, n:])
            k += 1


def orc_noise(orc, img, w, h, stride, bd):
    orc.orc_estimate_noise.restype = C.c_int32
    return orc.orc_estimate_noise(C.c_void_p(img.ctypes.data), w, h, stride, int(bd > 8), bd)


def test_noise_estimate_vs_reference(orc, ref):
    """svt_estimate_noise_fp16 / svt_estimate_noise_highbd_fp16 through the reference's own dispatch pointers."""
    from test_txfm_oracle import rtcd
    V = C.c_void_p
    lo = rtcd(ref, "svt_estimate_noise_fp16", C.c_int32, V, C.c_uint16, C.c_uint16, C.c_uint16)
    hi = rtcd(ref, "svt_estimate_noise_highbd_fp16", C.c_int32, V, C.c_int, C.c_int, C.c_int, C.c_int)
    seen = set()
    for img, w, h, stride, bd in F.noise_cases():
        exp = lo(img.ctypes.data, w, h, stride) if bd == 8 else hi(img.ctypes.data, w, h, stride, bd)
        assert orc_noise(orc, img, w, h, stride, bd) == exp, (w, h, bd)
        seen.add(-1 if exp == -65536 else (0 if exp == 0 else 1))
    assert seen == {-1, 0, 1}       # unreliable, exactly zero and ordinary estimates all occur


def test_noise_estimate_golden(orc):
    gold = np.load(GOLD_NOISE)["noise"]
    got = [orc_noise(orc, *c) for c in F.noise_cases()]
    assert np.array_equal(np.array(got, np.int64), gold)

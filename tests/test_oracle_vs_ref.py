"""CPU: the oracle against the REAL reference C functions (oracle/_ref/libsvtref.so), on seeded inputs.

Skipped where the reference build is absent.  This is what pins the oracle; tests/test_oracle_golden.py
keeps the pin on machines without /root/reference.
"""
import ctypes as C
import zlib

import numpy as np
import pytest

import me_cases
from svtav1_hip import frames

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)


def P(a, off=0):
    return C.cast(int(a.ctypes.data) + int(off), u8p)


def test_sad_loop(orc, ref):
    for prm, src, refw in me_cases.iter_sad_loop_cases():
        assert me_cases.call_sad_loop(orc.orc_sad_loop_kernel, prm, src, refw) == \
            me_cases.call_sad_loop(ref.svt_sad_loop_kernel_c, prm, src, refw), prm


def test_sad_loop_subsampled_call_shape(orc, ref):
    """The HME call shape: src/ref stride doubled, half the rows, raw stride = plane stride."""
    rng = np.random.default_rng(3)
    for (bw, bh, sw, sh) in ((16, 16, 48, 24), (32, 32, 8, 3), (64, 64, 8, 3), (10, 14, 16, 9)):
        stride = 200
        src = rng.integers(0, 256, size=(bh, stride), dtype=np.uint8)
        refw = rng.integers(0, 256, size=(bh + sh + 2, stride), dtype=np.uint8)
        out = []
        for fn in (orc.orc_sad_loop_kernel, ref.svt_sad_loop_kernel_c):
            best, x, y = C.c_uint64(), C.c_int16(-1), C.c_int16(-1)
            fn(P(src), 2 * stride, P(refw), 2 * stride, bh >> 1, bw, C.byref(best), C.byref(x), C.byref(y), stride, 0, sw, sh)
            out.append((best.value, x.value, y.value))
        assert out[0] == out[1]


def test_ext_sad_pyramid(orc, ref):
    rng = np.random.default_rng(11)
    for sub in (0, 1):
        for trial in range(6):
            stride = 64 + 8 + 17
            src = rng.integers(0, 256, size=(64, stride), dtype=np.uint8)
            refw = rng.integers(0, 256, size=(64, stride), dtype=np.uint8)
            if trial == 4:
                refw[...] = src  # all-zero SADs: tie-breaking on the first position
            res = []
            for all_fn, e32_fn in ((orc.orc_ext_all_sad_calculation_8x8_16x16, orc.orc_ext_eight_sad_calculation_32x32_64x64),
                                   (ref.svt_ext_all_sad_calculation_8x8_16x16_c, ref.svt_ext_eight_sad_calculation_32x32_64x64_c)):
                b8 = np.full(64, 100 * 64 if trial != 5 else 1, np.uint32)
                b16 = np.full(16, 400 * 64 if trial != 5 else 1, np.uint32)
                b32 = np.full(4, 0xffffff, np.uint32)
                b64_ = np.full(1, 0xffffff, np.uint32)
                m8, m16, m32, m64 = (np.zeros(n, np.uint32) for n in (64, 16, 4, 1))
                e16 = np.zeros((16, 8), np.uint32)
                e32 = np.zeros((4, 8), np.uint32)
                mv = (0xfffd << 16) | 0x0005
                all_fn(P(src), stride, P(refw), stride, C.c_uint32(mv), b8.ctypes.data_as(u32p), b16.ctypes.data_as(u32p),
                       m8.ctypes.data_as(u32p), m16.ctypes.data_as(u32p), e16.ctypes.data_as(C.c_void_p), None, C.c_uint8(sub))
                e32_fn(e16.ctypes.data_as(C.c_void_p), b32.ctypes.data_as(u32p), b64_.ctypes.data_as(u32p),
                       m32.ctypes.data_as(u32p), m64.ctypes.data_as(u32p), C.c_uint32(mv), e32.ctypes.data_as(C.c_void_p))
                res.append((b8, b16, b32, b64_, m8, m16, m32, m64, e16, e32))
            for a, b in zip(*res):
                assert np.array_equal(a, b)
            # single-position variants
            res = []
            for f16, f32 in ((orc.orc_ext_sad_calculation_8x8_16x16, orc.orc_ext_sad_calculation_32x32_64x64),
                             (ref.svt_ext_sad_calculation_8x8_16x16_c, ref.svt_ext_sad_calculation_32x32_64x64_c)):
                b8 = np.full(4, 5000, np.uint32)
                b16 = np.full(1, 20000, np.uint32)
                m8, m16 = np.zeros(4, np.uint32), np.zeros(1, np.uint32)
                s16, s8 = np.zeros(16, np.uint32), np.zeros(4, np.uint32)
                f16(P(src), stride, P(refw, 3), stride, b8.ctypes.data_as(u32p), b16.ctypes.data_as(u32p), m8.ctypes.data_as(u32p),
                    m16.ctypes.data_as(u32p), C.c_uint32(7), s16.ctypes.data_as(u32p), s8.ctypes.data_as(u32p), C.c_uint8(sub))
                s16[:] = rng.integers(0, 4000, 16) if f16 is orc.orc_ext_sad_calculation_8x8_16x16 else res[0][4]
                keep = s16.copy()
                b32, b64_ = np.full(4, 9000, np.uint32), np.full(1, 30000, np.uint32)
                m32, m64, s32 = np.zeros(4, np.uint32), np.zeros(1, np.uint32), np.zeros(4, np.uint32)
                f32(s16.ctypes.data_as(u32p), b32.ctypes.data_as(u32p), b64_.ctypes.data_as(u32p), m32.ctypes.data_as(u32p),
                    m64.ctypes.data_as(u32p), C.c_uint32(9), s32.ctypes.data_as(u32p))
                res.append((b8, b16, m8, m16, keep, s8, b32, b64_, m32, m64, s32))
            for a, b in zip(*res):
                assert np.array_equal(a, b)


def test_downsample_and_stats(orc, ref):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(70, 150), dtype=np.uint8)
    for step in (2, 4):
        o1 = np.zeros((40, 80), np.uint8)
        o2 = np.zeros((40, 80), np.uint8)
        orc.orc_downsample_2d(P(img), 150, 140, 66, P(o1), 80, step)
        ref.svt_aom_downsample_2d_c(P(img), 150, 140, 66, P(o2), 80, step)
        assert np.array_equal(o1, o2) and o1.any()
    m1, q1, m2, q2 = (np.zeros(4, np.uint64) for _ in range(4))
    orc.orc_compute_interm_var_four8x8(P(img, 7), C.c_uint16(150), m1.ctypes.data_as(C.c_void_p), q1.ctypes.data_as(C.c_void_p))
    ref.svt_compute_interm_var_four8x8_c(P(img, 7), C.c_uint16(150), m2.ctypes.data_as(C.c_void_p), q2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(m1, m2) and np.array_equal(q1, q2)
    assert orc.orc_compute_sub_mean_8x8(P(img, 3), C.c_uint16(150)) == ref.svt_compute_sub_mean_8x8_c(P(img, 3), C.c_uint16(150))
    assert orc.orc_compute_mean(P(img), 150, 8, 8) == ref.svt_compute_mean_c(P(img), 150, 8, 8)
    assert orc.orc_compute_mean_squared_values(P(img), 150, 8, 8) == ref.svt_compute_mean_squared_values_c(P(img), 150, 8, 8)


@pytest.mark.parametrize("w,h", [(200, 136), (1920 // 4, 1080 // 4 + 2)])
def test_pyramid_variance_frame(orc, ref, w, h):
    clip = me_cases.make_clip("pan", w, h, 1, seed=9)
    for l1 in (1, 0):
        a, b = frames.HostPyramid(clip[0]), frames.HostPyramid(clip[0])
        da, db = a.desc(), b.desc()
        orc.orc_pyramid_frame(C.byref(da.full), C.byref(da.quarter), C.byref(da.sixteenth), l1)
        ref.ref_pyramid_frame(C.byref(db.full), C.byref(db.quarter), C.byref(db.sixteenth), l1)
        if l1:
            assert np.array_equal(a.quarter.buf, b.quarter.buf)
        assert np.array_equal(a.sixteenth.buf, b.sixteenth.buf)
    nb = frames.b64_count(w, h)
    for fp in (0, 1):
        v1, v2 = np.zeros((nb, 85), np.uint16), np.zeros((nb, 85), np.uint16)
        orc.orc_variance_frame(C.byref(da.full), v1.ctypes.data_as(C.c_void_p), None, fp)
        ref.ref_variance_frame(C.byref(da.full), v2.ctypes.data_as(C.c_void_p), fp)
        assert np.array_equal(v1, v2)


ME_SCENARIOS = [
    # kind, w, h, params key, cur, list0, list1, tl, is_ref
    ("pan", 640, 360, "m8_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("noise", 328, 264, "m8_360p_tl2", 2, [1, 0], [3, 4], 2, 0),
    ("noise", 256, 192, "m6_360p_tl2", 2, [1, 0], [3], 2, 1),
    ("static", 320, 192, "m8_360p_tl0", 4, [3, 2, 1], [], 0, 1),
    ("static", 320, 200, "m8_360p_tl0", 2, [1, 0], [3, 4], 0, 1),   # base-layer B: list1 takes the no-HME path
    ("flat", 256, 128, "m8_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("fastpan", 712, 472, "m4_360p_tl2", 2, [0], [4], 2, 1),
    ("fastpan", 640, 360, "m2_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("blocks", 512, 384, "m0_360p_tl2", 1, [0], [2], 2, 1),
    ("blocks", 512, 384, "m6_360p_tl2", 2, [0, 1], [4, 3], 2, 0),
    ("noise", 264, 136, "m12_360p_tl2", 2, [1], [3], 2, 1),
    ("pan", 200, 136, "m10_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("blocks", 1000, 600, "m8_720p_tl2", 2, [1, 0, 0], [3, 4], 3, 1),
    ("pan", 960, 544, "m8_1080p_tl2", 3, [2, 1, 0], [4, 4, 4], 2, 1),  # 3+3 refs: (BWD,ALT) bipred branch
]


@pytest.mark.parametrize("sc", ME_SCENARIOS, ids=lambda s: f"{s[0]}-{s[1]}x{s[2]}-{s[3]}")
def test_me_frame(orc, ref, sc):
    kind, w, h, key, cur, l0, l1, tl, is_ref = sc
    clip = me_cases.make_clip(kind, w, h, 5, seed=zlib.crc32(repr(sc[:4]).encode()) % 1000)
    pyrs = me_cases.build_pyramids(orc, clip)
    prm = me_cases.scenario_params(key, cur, l0, l1, tl, is_ref)
    a = me_cases.run_cpu(ref.ref_me_frame, prm, pyrs, cur, l0, l1, w, h)
    b = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h)
    me_cases.assert_same(a, b, str(sc))


@pytest.mark.parametrize("sc", me_cases.MCTF_SCENARIOS, ids=lambda s: f"{s[0]}-{s[1]}x{s[2]}-th{s[6]}")
def test_me_frame_mctf(orc, ref, sc):
    """me_type == ME_MCTF: unscaled distance in the full-pel area, no pruning, tf_me_exit_th early exit, no candidates."""
    kind, w, h, key, cur, refpoc, th, seed = sc
    clip = me_cases.make_clip(kind, w, h, 5, seed=seed)
    pyrs = me_cases.build_pyramids(orc, clip)
    prm = me_cases.mctf_params(key, cur, refpoc, th)
    a = me_cases.run_cpu(ref.ref_me_frame, prm, pyrs, cur, [refpoc], [], w, h)
    b = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, [refpoc], [], w, h)
    me_cases.assert_same(a, b, str(sc))
    searched = (a["best_sad"][:, 0, 0, 0] != 0).sum()
    assert (th == 0) <= (searched == len(a["best_sad"]))  # hme_sad < th exits: never at th 0, noise blocks stay above 65535
    assert (a["me_64x64_distortion"].view(np.uint8) == 0xA5).all()      # statistics / candidates are not produced in this mode


def test_me_frame_param_variants(orc, ref):
    """Branches no preset reaches at qp 35: FULL_SAD search, pre-HME l1 early exit + skip lines, sr_adjustment 2,
    MV-based SA growth, stage-based exits, unipred-only, only_l_bwd."""
    w, h, cur, l0, l1 = 384, 256, 2, [1, 0], [3, 4]
    clip = me_cases.make_clip("blocks", w, h, 5, seed=77)
    pyrs = me_cases.build_pyramids(orc, clip)
    variants = [
        dict(hme_search_method=1, me_search_method=1),
        dict(prehme_l1_early_exit=1, prehme_skip_search_line=1),
        dict(enable_me_sr_adjustment=2, me_early_exit_th=0),
        dict(me_early_exit_th=0, mv_sa_adj_enabled=1, mv_sa_adj_mv_size_th=3, mv_sa_adj_sa_multiplier=2),
        dict(prev_me_stage_based_exit_th=64 * 64 * 4),
        dict(reduce_hme_l0_sr_th_min=8, reduce_hme_l0_sr_th_max=100),
        dict(only_l_bwd=1, prune_me_candidates_th=0),
        dict(me_safe_limit_zz_th=200000, similar_brightness_refs=1, hierarchical_levels=2),
        dict(enable_hme_level1_flag=0, prehme_enable=0),
        dict(enable_hme_flag=0, enable_hme_level0_flag=0, enable_hme_level1_flag=0, prehme_enable=0),
        dict(me_8x8_var_enabled=0),
        dict(enable_me_8x8=0),
    ]
    for v in variants:
        prm = me_cases.scenario_params("m6_360p_tl2", cur, l0, l1, 2, 1)
        for k, val in v.items():
            setattr(prm, k, val)
        a = me_cases.run_cpu(ref.ref_me_frame, prm, pyrs, cur, l0, l1, w, h)
        b = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h)
        me_cases.assert_same(a, b, str(v))
    # single-list / single-ref candidate builders
    for (ll0, ll1, extra) in (([1], [], {}), ([1], [3], dict(use_best_unipred_cand_only=1))):
        prm = me_cases.scenario_params("m8_360p_tl2", cur, ll0, ll1, 2, 1)
        for k, val in extra.items():
            setattr(prm, k, val)
        a = me_cases.run_cpu(ref.ref_me_frame, prm, pyrs, cur, ll0, ll1, w, h)
        b = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, ll0, ll1, w, h)
        me_cases.assert_same(a, b, str((ll0, ll1, extra)))

"""GPU parity: the HIP path (through the C-ABI of libsvtav1_hip.so) against the oracle, bit-exact."""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

import me_cases
from svtav1_hip import abi, device, frames

pytestmark = pytest.mark.gpu

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)


def P(a, off=0):
    return C.cast(int(a.ctypes.data) + int(off), u8p)


# ----------------------------------------------------------------------------- Tier A (per-call, host pointers)
def test_tier_a_sad_loop_golden(hip):
    """Every SadTest-style case: HIP == golden vectors produced by the reference's svt_sad_loop_kernel_c."""
    with open(os.path.join(me_cases.GOLDEN, "sad_loop_cases.json")) as f:
        expected = json.load(f)
    for exp, (prm, src, refw) in zip(expected, me_cases.iter_sad_loop_cases()):
        got = me_cases.call_sad_loop(hip.svt_sad_loop_kernel_hip, prm, src, refw)
        assert got == (exp["best"], exp["x"], exp["y"]), prm


def test_tier_a_sad_loop_hme_shapes(hip, orc):
    rng = np.random.default_rng(3)
    shapes = [(16, 16, 48, 24), (16, 16, 8, 400), (16, 16, 384, 3), (32, 32, 8, 3), (64, 64, 8, 3), (10, 14, 16, 9),
              (16, 12, 96, 96), (4, 4, 3, 2), (128, 128, 64, 64), (20, 6, 7, 5)]
    for (bw, bh, sw, sh) in shapes:
        stride = max(200, bw + sw + 8)
        src = rng.integers(0, 256, size=(bh, stride), dtype=np.uint8)
        refw = rng.integers(0, 256, size=(bh + sh + 2, stride), dtype=np.uint8)
        for sub in (0, 1):
            out = []
            for fn in (orc.orc_sad_loop_kernel, hip.svt_sad_loop_kernel_hip):
                best, x, y = C.c_uint64(), C.c_int16(-1), C.c_int16(-1)
                m = 2 if sub else 1
                fn(P(src), C.c_uint32(m * stride), P(refw), C.c_uint32(m * stride), C.c_uint32(bh // m), C.c_uint32(bw),
                   C.byref(best), C.byref(x), C.byref(y), C.c_uint32(stride), C.c_uint8(0), C.c_int16(sw), C.c_int16(sh))
                out.append((best.value, x.value, y.value))
            assert out[0] == out[1], (bw, bh, sw, sh, sub)


def test_tier_a_edge_cases(hip, orc):
    """Empty search areas and the 'nothing searched' case must leave x/y untouched, like the C function."""
    src = np.zeros((16, 64), np.uint8)
    refw = np.full((40, 64), 9, np.uint8)
    for (sw, sh, skip) in ((0, 4, 0), (4, 0, 0), (4, 1, 1), (-3, 2, 0)):
        out = []
        for fn in (orc.orc_sad_loop_kernel, hip.svt_sad_loop_kernel_hip):
            best, x, y = C.c_uint64(5), C.c_int16(-9), C.c_int16(-9)
            fn(P(src), C.c_uint32(64), P(refw), C.c_uint32(64), C.c_uint32(16), C.c_uint32(16), C.byref(best), C.byref(x),
               C.byref(y), C.c_uint32(64), C.c_uint8(skip), C.c_int16(sw), C.c_int16(sh))
            out.append((best.value, x.value, y.value))
        assert out[0] == out[1] == (0xffffff, -9, -9)


def test_tier_a_nxm_and_ext(hip, orc):
    rng = np.random.default_rng(11)
    stride = 64 + 8 + 17
    for trial in range(4):
        src = rng.integers(0, 256, size=(64, stride), dtype=np.uint8)
        refw = rng.integers(0, 256, size=(64, stride), dtype=np.uint8)
        if trial == 3:
            refw[...] = src
        for (h, w) in ((64, 64), (32, 64), (7, 5), (1, 1)):
            assert hip.svt_nxm_sad_kernel_hip(P(src), stride, P(refw), stride, h, w) == orc.orc_nxm_sad(P(src), stride, P(refw), stride, h, w)
        for sub in (0, 1):
            res = []
            for all_fn, e32_fn in ((orc.orc_ext_all_sad_calculation_8x8_16x16, orc.orc_ext_eight_sad_calculation_32x32_64x64),
                                   (hip.svt_ext_all_sad_calculation_8x8_16x16_hip, hip.svt_ext_eight_sad_calculation_32x32_64x64_hip)):
                b8, b16 = np.full(64, 6400, np.uint32), np.full(16, 25600, np.uint32)
                b32, b64_ = np.full(4, 0xffffff, np.uint32), np.full(1, 0xffffff, np.uint32)
                m8, m16, m32, m64 = (np.zeros(n, np.uint32) for n in (64, 16, 4, 1))
                e16, e32 = np.zeros((16, 8), np.uint32), np.zeros((4, 8), np.uint32)
                mv = (0xfffd << 16) | 0x0005
                all_fn(P(src), C.c_uint32(stride), P(refw), C.c_uint32(stride), C.c_uint32(mv), b8.ctypes.data_as(u32p),
                       b16.ctypes.data_as(u32p), m8.ctypes.data_as(u32p), m16.ctypes.data_as(u32p),
                       e16.ctypes.data_as(C.c_void_p), None, C.c_uint8(sub))
                e32_fn(e16.ctypes.data_as(C.c_void_p), b32.ctypes.data_as(u32p), b64_.ctypes.data_as(u32p),
                       m32.ctypes.data_as(u32p), m64.ctypes.data_as(u32p), C.c_uint32(mv), e32.ctypes.data_as(C.c_void_p))
                res.append((b8, b16, b32, b64_, m8, m16, m32, m64, e16, e32))
            for a, b in zip(*res):
                assert np.array_equal(a, b)
            res = []
            for f16 in (orc.orc_ext_sad_calculation_8x8_16x16, hip.svt_ext_sad_calculation_8x8_16x16_hip):
                b8, b16 = np.full(4, 5000, np.uint32), np.full(1, 20000, np.uint32)
                m8, m16 = np.zeros(4, np.uint32), np.zeros(1, np.uint32)
                s16, s8 = np.zeros(1, np.uint32), np.zeros(4, np.uint32)
                f16(P(src), C.c_uint32(stride), P(refw, 3), C.c_uint32(stride), b8.ctypes.data_as(u32p), b16.ctypes.data_as(u32p),
                    m8.ctypes.data_as(u32p), m16.ctypes.data_as(u32p), C.c_uint32(7), s16.ctypes.data_as(u32p),
                    s8.ctypes.data_as(u32p), C.c_uint8(sub))
                res.append((b8, b16, m8, m16, s16, s8))
            for a, b in zip(*res):
                assert np.array_equal(a, b)


def test_tier_a_downsample_and_stats(hip, orc):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(70, 150), dtype=np.uint8)
    for step in (2, 4):
        o1, o2 = np.zeros((40, 80), np.uint8), np.zeros((40, 80), np.uint8)
        orc.orc_downsample_2d(P(img), 150, 140, 66, P(o1), 80, step)
        hip.svt_aom_downsample_2d_hip(P(img), C.c_uint32(150), C.c_uint32(140), C.c_uint32(66), P(o2), C.c_uint32(80), C.c_uint32(step))
        assert np.array_equal(o1, o2) and o1.any()
    m1, q1, m2, q2 = (np.zeros(4, np.uint64) for _ in range(4))
    orc.orc_compute_interm_var_four8x8(P(img, 7), C.c_uint16(150), m1.ctypes.data_as(C.c_void_p), q1.ctypes.data_as(C.c_void_p))
    hip.svt_compute_interm_var_four8x8_hip(P(img, 7), C.c_uint16(150), m2.ctypes.data_as(C.c_void_p), q2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(m1, m2) and np.array_equal(q1, q2)
    assert hip.svt_compute_sub_mean_8x8_hip(P(img, 3), C.c_uint16(150)) == orc.orc_compute_sub_mean_8x8(P(img, 3), C.c_uint16(150))
    assert hip.svt_compute_mean_8x8_hip(P(img), 150, 8, 8) == orc.orc_compute_mean(P(img), 150, 8, 8)
    assert hip.svt_compute_mean_square_values_8x8_hip(P(img), 150, 8, 8) == orc.orc_compute_mean_squared_values(P(img), 150, 8, 8)


# ----------------------------------------------------------------------------- Tier B
def test_sad_loop_batch(hip, orc):
    """Many descriptors over one device arena == per-descriptor oracle results."""
    rng = np.random.default_rng(21)
    stride, rows = 512, 300
    arena = rng.integers(0, 256, size=(rows, stride), dtype=np.uint8)
    arena[100:140, 100:200] = 0  # flat areas: ties
    descs, expect = [], []
    for i in range(300):
        bw, bh = [(16, 8), (32, 16), (64, 32), (16, 16), (8, 8), (12, 6)][i % 6]
        sw, sh = int(rng.integers(1, 64)), int(rng.integers(1, 40))
        k = int(rng.integers(1, 3))
        sy, sx = int(rng.integers(0, rows - bh * k)), int(rng.integers(0, stride - bw))
        ry, rx = int(rng.integers(0, rows - (sh + bh * k))), int(rng.integers(0, stride - (sw + bw)))
        skip = int(bw == 16 and bh <= 16 and i % 5 == 0)
        d = abi.SadLoopDesc(sy * stride + sx, ry * stride + rx, k * stride, k * stride, stride, bw, bh, sw, sh, skip)
        descs.append(d)
        best, x, y = C.c_uint64(), C.c_int16(0x7fff), C.c_int16(0x7fff)
        orc.orc_sad_loop_kernel(P(arena, d.src_off), C.c_uint32(k * stride), P(arena, d.ref_off), C.c_uint32(k * stride),
                                C.c_uint32(bh), C.c_uint32(bw), C.byref(best), C.byref(x), C.byref(y), C.c_uint32(stride),
                                C.c_uint8(skip), C.c_int16(sw), C.c_int16(sh))
        expect.append((best.value, x.value, y.value))
    darena = device.DeviceBuffer(hip, arena.nbytes)
    darena.upload(arena)
    darr = (abi.SadLoopDesc * len(descs))(*descs)
    ddesc = device.DeviceBuffer(hip, C.sizeof(darr))
    ddesc.upload(np.frombuffer(darr, dtype=np.uint8))
    dres = device.DeviceBuffer(hip, 16 * len(descs))
    device.check(hip, hip.svt_hip_sad_loop_batch(C.c_void_p(darena.ptr), C.c_void_p(ddesc.ptr), C.c_void_p(dres.ptr),
                                                 C.c_uint32(len(descs)), None), "svt_hip_sad_loop_batch")
    raw = dres.download(np.uint8, (len(descs), 16))
    got = [(int(r[:8].view(np.uint64)[0]), int(r[8:10].view(np.int16)[0]), int(r[10:12].view(np.int16)[0])) for r in raw]
    assert got == expect


@pytest.mark.parametrize("w,h", [(200, 136), (648, 360), (1920, 1080)])
def test_pyramid_and_variance_frame(hip, orc, w, h):
    clip = me_cases.make_clip("pan", w, h, 1, seed=9)
    for l1 in (1, 0):
        hp = frames.HostPyramid(clip[0])
        d = hp.desc()
        orc.orc_pyramid_frame(C.byref(d.full), C.byref(d.quarter), C.byref(d.sixteenth), l1)
        hz = frames.HostPyramid(clip[0])  # decimated planes start as zeros on the device
        dp = device.DevicePyramid(hip, hz)
        dd = dp.desc()
        device.check(hip, hip.svt_hip_pyramid_frame(C.byref(dd.full), C.byref(dd.quarter), C.byref(dd.sixteenth), l1, None),
                     "svt_hip_pyramid_frame")
        hip.svt_hip_stream_sync(None)
        if l1:
            assert np.array_equal(dp.quarter.download(), hp.quarter.buf)
        assert np.array_equal(dp.sixteenth.download(), hp.sixteenth.buf)
    nb = frames.b64_count(w, h)
    for fp in (0, 1):
        v1, m1 = np.zeros((nb, 85), np.uint16), np.zeros((nb, 85), np.uint64)
        orc.orc_variance_frame(C.byref(d.full), v1.ctypes.data_as(C.c_void_p), m1.ctypes.data_as(C.c_void_p), fp)
        dv, dm = device.DeviceBuffer(hip, v1.nbytes), device.DeviceBuffer(hip, m1.nbytes)
        device.check(hip, hip.svt_hip_variance_frame(C.byref(dd.full), C.c_void_p(dv.ptr), C.c_void_p(dm.ptr), fp, None),
                     "svt_hip_variance_frame")
        assert np.array_equal(dv.download(np.uint16, v1.shape), v1)
        assert np.array_equal(dm.download(np.uint64, m1.shape), m1)
    # svt_hip_pad_plane == edge replication
    hq = frames.HostPlane(100, 40, 16, clip[0][:40, :100])
    dq = device.DevicePlane(hip, hq)
    dsc = dq.desc()
    device.check(hip, hip.svt_hip_pad_plane(C.byref(dsc), None), "svt_hip_pad_plane")
    hq.pad_edges()
    assert np.array_equal(dq.download(), hq.buf)


@pytest.mark.parametrize("l1,fp", [(1, 0), (0, 1)])
def test_analysis_frames_batch(hip, orc, l1, fp):
    """svt_hip_analysis_frames: pyramids + variances of several pictures of different sizes in three launches."""
    sizes = [(200, 136), (648, 360), (328, 200), (200, 136)]
    hosts, devs, outs, jobs = [], [], [], (abi.AnalysisJob * len(sizes))()
    for i, (w, h) in enumerate(sizes):
        clip = me_cases.make_clip("pan", w, h, 1, seed=20 + i)
        hp = frames.HostPyramid(clip[0])
        d = hp.desc()
        orc.orc_pyramid_frame(C.byref(d.full), C.byref(d.quarter), C.byref(d.sixteenth), l1)
        nb = frames.b64_count(w, h)
        v1, m1 = np.zeros((nb, 85), np.uint16), np.zeros((nb, 85), np.uint64)
        orc.orc_variance_frame(C.byref(d.full), v1.ctypes.data_as(C.c_void_p), m1.ctypes.data_as(C.c_void_p), fp)
        dp = device.DevicePyramid(hip, frames.HostPyramid(clip[0]))
        dv, dm = device.DeviceBuffer(hip, v1.nbytes), device.DeviceBuffer(hip, m1.nbytes)
        jobs[i] = abi.AnalysisJob(dp.desc(), dv.ptr, dm.ptr if i % 2 == 0 else None)
        hosts.append((hp, v1, m1)), devs.append(dp), outs.append((dv, dm))
    device.check(hip, hip.svt_hip_analysis_frames(jobs, len(sizes), l1, fp, None), "svt_hip_analysis_frames")
    hip.svt_hip_stream_sync(None)
    for i, ((hp, v1, m1), dp, (dv, dm)) in enumerate(zip(hosts, devs, outs)):
        if l1:
            assert np.array_equal(dp.quarter.download(), hp.quarter.buf), i
        assert np.array_equal(dp.sixteenth.download(), hp.sixteenth.buf), i
        assert np.array_equal(dv.download(np.uint16, v1.shape), v1), i
        if i % 2 == 0:
            assert np.array_equal(dm.download(np.uint64, m1.shape), m1), i
    assert hip.svt_hip_analysis_frames(None, 0, 1, 0, None) == abi.SVT_HIP_ERR_BAD_PARAMETER


def run_hip_me(hip, prm, pyrs, cur, l0, l1, w, h, n_copies=1):
    nb = frames.b64_count(w, h)
    dpyr = {i: device.DevicePyramid(hip, pyrs[i]) for i in set([cur] + l0 + l1)}
    outs, jobs = [], []
    for _ in range(n_copies):
        o = device.DeviceMeOut(hip, prm, nb)
        job = abi.MeFrameJob()
        job.prm = prm
        job.src = dpyr[cur].desc()
        for r, poc in enumerate(l0):
            job.ref[0][r] = dpyr[poc].desc()
        for r, poc in enumerate(l1):
            job.ref[1][r] = dpyr[poc].desc()
        job.out = o.desc()
        outs.append(o)
        jobs.append(job)
    device.me_frames(hip, jobs)
    return [o.download() for o in outs]


ME_SCENARIOS = [
    ("pan", 640, 360, "m8_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("noise", 328, 264, "m8_360p_tl2", 2, [1, 0], [3, 4], 2, 0),
    ("noise", 256, 192, "m6_360p_tl2", 2, [1, 0], [3], 2, 1),
    ("static", 320, 192, "m8_360p_tl0", 4, [3, 2, 1], [], 0, 1),
    ("static", 320, 200, "m8_360p_tl0", 2, [1, 0], [3, 4], 0, 1),
    ("flat", 256, 128, "m8_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("fastpan", 712, 472, "m4_360p_tl2", 2, [0], [4], 2, 1),
    ("fastpan", 640, 360, "m2_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("blocks", 512, 384, "m0_360p_tl2", 1, [0], [2], 2, 1),
    ("blocks", 512, 384, "m6_360p_tl2", 2, [0, 1], [4, 3], 2, 0),
    ("noise", 264, 136, "m12_360p_tl2", 2, [1], [3], 2, 1),
    ("pan", 200, 136, "m10_360p_tl2", 2, [1, 0], [3, 4], 2, 1),
    ("blocks", 1000, 600, "m8_720p_tl2", 2, [1, 0, 0], [3, 4], 3, 1),
    ("pan", 960, 544, "m8_1080p_tl2", 3, [2, 1, 0], [4, 4, 4], 2, 1),
    ("pan", 1920, 1080, "m8_1080p_tl2", 2, [1, 0], [3, 4], 2, 1),     # BASELINE.json configs[1] size
    ("blocks", 1920, 1080, "m8_1080p_tl2", 2, [1, 0], [3, 4], 2, 1),
]


@pytest.mark.parametrize("sc", ME_SCENARIOS, ids=lambda s: f"{s[0]}-{s[1]}x{s[2]}-{s[3]}")
def test_me_frame(hip, orc, sc):
    kind, w, h, key, cur, l0, l1, tl, is_ref = sc
    clip = me_cases.make_clip(kind, w, h, 5, seed=zlib.crc32(repr(sc[:4]).encode()) % 1000)
    pyrs = me_cases.build_pyramids(orc, clip)
    prm = me_cases.scenario_params(key, cur, l0, l1, tl, is_ref)
    want = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h)
    got = run_hip_me(hip, prm, pyrs, cur, l0, l1, w, h)[0]
    me_cases.assert_same(want, got, str(sc))


@pytest.mark.parametrize("sc", me_cases.MCTF_SCENARIOS, ids=lambda s: f"{s[0]}-{s[1]}x{s[2]}-th{s[6]}")
def test_me_frame_mctf(hip, orc, sc):
    """ME_MCTF mode of the b64 kernel (the temporal filter's motion search) against the oracle and against
    the outputs of the real svt_aom_motion_estimation_b64 with me_type = ME_MCTF (tests/golden/me_mctf.npz)."""
    kind, w, h, key, cur, refpoc, th, seed = sc
    clip = me_cases.make_clip(kind, w, h, 5, seed=seed)
    pyrs = me_cases.build_pyramids(orc, clip)
    prm = me_cases.mctf_params(key, cur, refpoc, th)
    want = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, [refpoc], [], w, h)
    got = run_hip_me(hip, prm, pyrs, cur, [refpoc], [], w, h)[0]
    me_cases.assert_same(want, got, str(sc))
    gold = np.load(os.path.join(me_cases.GOLDEN, "me_mctf.npz"))
    i = me_cases.MCTF_SCENARIOS.index(sc)
    for k in ("best_sad", "best_mv", "search_results"):
        assert np.array_equal(got[k], gold[f"s{i}_{k}"]), k


def test_me_frame_golden(hip, orc):
    """HIP == committed outputs of the reference's svt_aom_motion_estimation_b64 (tests/golden/me_frames.npz)."""
    with open(os.path.join(me_cases.GOLDEN, "me_frames.json")) as f:
        scen = json.load(f)
    gold = np.load(os.path.join(me_cases.GOLDEN, "me_frames.npz"))
    for i, s in enumerate(scen):
        clip = me_cases.make_clip(s["kind"], s["w"], s["h"], 5, seed=s["seed"])
        pyrs = me_cases.build_pyramids(orc, clip)
        prm = me_cases.scenario_params(s["key"], s["cur"], s["l0"], s["l1"])
        got = run_hip_me(hip, prm, pyrs, s["cur"], s["l0"], s["l1"], s["w"], s["h"])[0]
        me_cases.assert_same({k: gold[f"s{i}_{k}"] for k in got}, got, f"golden scenario {i}")


def test_me_frame_param_variants(hip, orc):
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
        want = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h)
        got = run_hip_me(hip, prm, pyrs, cur, l0, l1, w, h)[0]
        me_cases.assert_same(want, got, str(v))
    for (ll0, ll1, extra) in (([1], [], {}), ([1], [3], dict(use_best_unipred_cand_only=1))):
        prm = me_cases.scenario_params("m8_360p_tl2", cur, ll0, ll1, 2, 1)
        for k, val in extra.items():
            setattr(prm, k, val)
        want = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, ll0, ll1, w, h)
        got = run_hip_me(hip, prm, pyrs, cur, ll0, ll1, w, h)[0]
        me_cases.assert_same(want, got, str((ll0, ll1, extra)))


def test_me_frames_batch_is_deterministic(hip, orc):
    """Several pictures in one launch: every copy of the same job gives identical results (idempotence),
    and they equal the single-job result; also at the full 4K size of BASELINE.json configs[2..4]
    (size-independent property: SAD of the winning vector recomputed by the oracle's leaf SAD)."""
    w, h, cur, l0, l1 = 3840, 2160, 1, [0], [2]
    clip = me_cases.make_clip("pan", w, h, 3, seed=4)
    pyrs = me_cases.build_pyramids(orc, clip)
    prm = me_cases.scenario_params("m8_4k_tl2", cur, l0, l1, 2, 1)
    res = run_hip_me(hip, prm, pyrs, cur, l0, l1, w, h, n_copies=3)
    for r in res[1:]:
        me_cases.assert_same(res[0], r, "batch copies")
    # spot-check 64x64 winners against an independent SAD (oracle leaf kernel) on 40 random b64
    rng = np.random.default_rng(0)
    nb = frames.b64_count(w, h)
    bw = (w + 63) // 64
    src, rf = pyrs[cur].full, pyrs[l0[0]].full
    for b in rng.integers(0, nb, 40):
        sad = int(res[0]["best_sad"][b, 0, 0, 0])
        mv = int(res[0]["best_mv"][b, 0, 0, 0])
        mx, my = np.int16(mv & 0xffff), np.int16(mv >> 16)
        ox, oy = (b % bw) * 64, (b // bw) * 64
        s_off = (src.pad + oy) * src.stride + src.pad + ox
        r_off = (rf.pad + oy + int(my)) * rf.stride + rf.pad + ox + int(mx)
        want = orc.orc_nxm_sad(P(src.buf, s_off), 2 * src.stride, P(rf.buf, r_off), 2 * rf.stride, 32, 64) * 2
        assert sad == want
    # and the whole 4K frame against the oracle
    want = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h)
    me_cases.assert_same(want, res[0], "4K frame")


# ------------------------------------------------------------------------------------------------ memory contract
def test_me_frame_exact_size_planes(hip, orc):
    """include/svt_hip_me.h "Memory contract": planes of EXACTLY stride * (height + 2 * org_y) bytes, packed back to back in
    one allocation (so that any byte a kernel reads past a plane belongs to the next plane, and past the last one to a
    sentinel area filled first with 0x00 and then with 0xFF), give the oracle's results both times."""
    w, h, cur, l0, l1 = 456, 264, 2, [1, 0], [3, 4]
    clip = me_cases.make_clip("blocks", w, h, 5, seed=21)
    pyrs = me_cases.build_pyramids(orc, clip)
    prm = me_cases.scenario_params("m8_360p_tl2", cur, l0, l1, 2, 1)
    want = me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h)
    planes = [pl for p in pyrs for pl in p.planes()]
    total = sum(pl.nbytes for pl in planes)
    SENT = 4096
    big = device.DeviceBuffer(hip, total + SENT)
    nb = frames.b64_count(w, h)
    for sentinel in (0x00, 0xFF):
        host = np.full(total + SENT, sentinel, np.uint8)
        descs, off = [], 0
        for pl in planes:
            host[off:off + pl.nbytes] = pl.buf.reshape(-1)
            descs.append(pl.desc(big.ptr + off))
            off += pl.nbytes
        big.upload(host)
        pyr_desc = [abi.Pyramid8(*descs[3 * i:3 * i + 3]) for i in range(len(pyrs))]
        out = device.DeviceMeOut(hip, prm, nb)
        job = abi.MeFrameJob()
        job.prm, job.src, job.out = prm, pyr_desc[cur], out.desc()
        for r, poc in enumerate(l0):
            job.ref[0][r] = pyr_desc[poc]
        for r, poc in enumerate(l1):
            job.ref[1][r] = pyr_desc[poc]
        device.me_frames(hip, [job])
        me_cases.assert_same(want, out.download(), f"exact-size planes, sentinel {sentinel:#x}")


def test_me_frames_descriptor_lifetime(hip, orc):
    """svt_hip_me_frames copies the HOST job array before it returns: six asynchronous calls in a row, each host array
    overwritten right after its call, no synchronisation in between (the regression test for the round-1 abort:
    DESIGN.md section 8)."""
    w, h, cur, l0, l1 = 320, 200, 2, [1, 0], [3, 4]
    clip = me_cases.make_clip("pan", w, h, 5, seed=5)
    pyrs = me_cases.build_pyramids(orc, clip)
    nb = frames.b64_count(w, h)
    dpyr = {i: device.DevicePyramid(hip, pyrs[i]) for i in range(5)}
    keys = ["m8_360p_tl2", "m4_360p_tl2", "m12_360p_tl2", "m8_360p_tl0", "m6_360p_tl2", "m10_360p_tl2"]
    outs, wants = [], []
    for key in keys:
        prm = me_cases.scenario_params(key, cur, l0, l1)
        wants.append(me_cases.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h))
        o = device.DeviceMeOut(hip, prm, nb)
        job = abi.MeFrameJob()
        job.prm, job.src, job.out = prm, dpyr[cur].desc(), o.desc()
        for r, poc in enumerate(l0):
            job.ref[0][r] = dpyr[poc].desc()
        for r, poc in enumerate(l1):
            job.ref[1][r] = dpyr[poc].desc()
        arr = (abi.MeFrameJob * 1)(job)
        device.check(hip, hip.svt_hip_me_frames(arr, C.c_uint32(1), None), "svt_hip_me_frames")
        C.memset(arr, 0xFF, C.sizeof(arr))          # the caller's array is dead the moment the call returns
        outs.append(o)
    device.check(hip, hip.svt_hip_stream_sync(None), "sync")
    for key, o, want in zip(keys, outs, wants):
        me_cases.assert_same(want, o.download(), key)

"""Shared scenario builders for the ME / pyramid parity tests (test infrastructure)."""
import ctypes as C
import json
import os

import numpy as np

from svtav1_hip import abi, frames

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_params(key):
    """ME parameter sets derived by the REFERENCE's svt_aom_sig_deriv_me (tests/golden/make_golden.py)."""
    with open(os.path.join(GOLDEN, "me_params.json")) as f:
        return abi.MeParams.from_dict(json.load(f)[key])


def make_clip(kind, width, height, n, seed=1):
    rng = np.random.default_rng(seed)
    if kind == "pan":
        return frames.synthetic_clip(width, height, n, seed=seed)
    if kind == "noise":  # every reference equally bad: nothing gets pruned by SAD deviation
        return [rng.integers(0, 256, size=(height, width), dtype=np.uint8) for _ in range(n)]
    if kind == "static":  # zero motion + sensor noise: early-exit paths
        base = frames.synthetic_clip(width, height, 1, seed=seed)[0].astype(np.int16)
        return [np.clip(base + rng.integers(-2, 3, size=base.shape), 0, 255).astype(np.uint8) for _ in range(n)]
    if kind == "flat":  # constant picture: every SAD ties -> pure tie-breaking test
        return [np.full((height, width), 77, dtype=np.uint8) for _ in range(n)]
    if kind == "fastpan":  # 11 px/frame horizontally, -5 vertically: large vectors, edge clamps
        big = frames.synthetic_clip(width + 16 * n, height + 8 * n, 1, seed=seed)[0]
        out = []
        for i in range(n):
            oy, ox = 5 * (n - 1 - i), 11 * i
            f = big[oy:oy + height, ox:ox + width].astype(np.int16) + rng.integers(-1, 2, size=(height, width))
            out.append(np.clip(f, 0, 255).astype(np.uint8))
        return out
    if kind == "blocks":  # independent motion per 128x128 region
        big = frames.synthetic_clip(width + 64, height + 64, 1, seed=seed)[0]
        out = []
        for i in range(n):
            f = np.empty((height, width), np.uint8)
            for by in range(0, height, 128):
                for bx in range(0, width, 128):
                    k = (by // 128 * 7 + bx // 128 * 3) % 5
                    dx, dy = (k - 2) * i, ((k * 2) % 5 - 2) * i
                    h, w = min(128, height - by), min(128, width - bx)
                    f[by:by + h, bx:bx + w] = big[32 + by + dy:32 + by + dy + h, 32 + bx + dx:32 + bx + dx + w]
            out.append(f)
        return out
    if kind.startswith("stripes"):  # vertical stripes alternating static content (inter prediction wins) and fresh noise per picture (intra wins)
        sw = int(kind[7:])
        base = frames.synthetic_clip(width, height, 1, seed=seed)[0]
        noisy = ((np.arange(width) // sw) & 1).astype(bool)
        out = []
        for _ in range(n):
            f = base.copy()
            f[:, noisy] = rng.integers(0, 256, size=(height, int(noisy.sum())), dtype=np.uint8)
            out.append(f)
        return out
    raise ValueError(kind)


def build_pyramids(orc, clip, hme_level1=1):
    """Host pyramids with the decimations produced by the ORACLE's pyramid (test input preparation)."""
    pyrs = []
    for f in clip:
        p = frames.HostPyramid(f)
        d = p.desc()
        orc.orc_pyramid_frame(C.byref(d.full), C.byref(d.quarter), C.byref(d.sixteenth), hme_level1)
        pyrs.append(p)
    return pyrs


def host_job(prm, pyrs, cur, l0, l1, out_desc):
    job = abi.MeFrameJob()
    job.prm = prm
    job.src = pyrs[cur].desc()
    for r, poc in enumerate(l0):
        job.ref[0][r] = pyrs[poc].desc()
    for r, poc in enumerate(l1):
        job.ref[1][r] = pyrs[poc].desc()
    job.out = out_desc
    return job


def scenario_params(key, cur, l0, l1, tl=None, is_ref=1):
    prm = load_params(key)
    frames.set_refs(prm, cur, l0, l1)
    if tl is not None:
        prm.temporal_layer_index = tl
    prm.is_ref = is_ref
    return prm


def run_cpu(fn, prm, pyrs, cur, l0, l1, width, height):
    """fn = orc.orc_me_frame_range or ref.ref_me_frame; returns dict of numpy output arrays."""
    nb = frames.b64_count(width, height)
    arrs, out = frames.alloc_me_out_host(prm, nb)
    job = host_job(prm, pyrs, cur, l0, l1, out)
    rc = fn(C.byref(job), 0, nb)
    assert rc == 0
    return arrs


def assert_same(a, b, what=""):
    for k in a:
        if not np.array_equal(a[k], b[k]):
            diff = np.argwhere(a[k] != b[k])
            raise AssertionError(f"{what}: {k} differs at {len(diff)} entries, first {diff[0].tolist()}: "
                                 f"{a[k][tuple(diff[0])]} vs {b[k][tuple(diff[0])]}")


def iter_sad_loop_cases():
    """Seeded inputs following the recipes of the reference's test/SadTest.cc (REF_MAX / SRC_MAX / RANDOM /
    UNALIGN patterns, block-size and search-area lists).  Yields (params dict, src, ref window)."""
    rng = np.random.default_rng(13596)
    sizes = [(16, 16), (16, 8), (32, 16), (64, 32), (8, 8), (24, 24), (48, 32), (64, 64), (12, 7), (128, 128)]
    areas = [(8, 3), (16, 5), (1, 1), (24, 17), (64, 9), (13, 6), (8, 40)]
    for (bw, bh) in sizes:
        for (sw, sh) in areas:
            for pattern in ("random", "ref_max", "src_max", "unalign"):
                stride = 127 if pattern == "unalign" else bw + sw + 16
                stride = max(stride, bw + sw)
                src = rng.integers(0, 256, size=(bh, stride), dtype=np.uint8)
                refw = rng.integers(0, 256, size=(bh + sh + 1, stride), dtype=np.uint8)
                if pattern == "ref_max":
                    refw[...] = 255
                    src[...] = 0
                if pattern == "src_max":
                    src[...] = 255
                    refw[...] = 0
                skip = int(bw == 16 and bh <= 16 and sh > 2 and pattern == "random")
                yield dict(bw=bw, bh=bh, sw=sw, sh=sh, stride=stride, skip=skip, pattern=pattern), src, refw


def call_sad_loop(fn, prm, src, refw):
    """fn has the RTCD signature of svt_sad_loop_kernel (aom_dsp_rtcd.h:776)."""
    u8p = C.POINTER(C.c_uint8)
    best = C.c_uint64(0)
    x, y = C.c_int16(-7), C.c_int16(-7)
    fn(C.cast(src.ctypes.data, u8p), C.c_uint32(prm["stride"]), C.cast(refw.ctypes.data, u8p), C.c_uint32(prm["stride"]),
       C.c_uint32(prm["bh"]), C.c_uint32(prm["bw"]), C.byref(best), C.byref(x), C.byref(y), C.c_uint32(prm["stride"]),
       C.c_uint8(prm["skip"]), C.c_int16(prm["sw"]), C.c_int16(prm["sh"]))
    return int(best.value), int(x.value), int(y.value)


# ME_MCTF mode (the temporal filter's call of svt_aom_motion_estimation_b64, temporal_filtering.c:3075): one list, one
# reference = the neighbouring picture being aligned to the central one.  (kind, w, h, params key, central, reference,
# tf_me_exit_th, seed): thresholds from "never" over "some blocks" to "every block" exits behind HME.
MCTF_SCENARIOS = [
    ("pan", 328, 200, "m8_360p_tl0", 2, 1, 0, 61), ("static", 320, 192, "m8_360p_tl0", 2, 3, 1500, 62),
    ("blocks", 456, 264, "m6_360p_tl0", 2, 0, 4000, 63), ("noise", 200, 136, "m4_360p_tl0", 2, 4, 65535, 64),
    ("fastpan", 392, 232, "m8_360p_tl0", 1, 3, 900, 65),
]


def mctf_params(key, cur, refpoc, exit_th):
    prm = scenario_params(key, cur, [refpoc], [], 0, 1)
    prm.me_mctf, prm.tf_me_exit_th = 1, exit_th
    return prm

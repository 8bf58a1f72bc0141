#!/usr/bin/env python3
"""One-off randomised parity sweep (GPU box) over the round-3 additions: TPL level 3, temporal filter with 8x8 prediction / low delay,
the open-loop ME with random presets / reference counts / thresholds (the all-reference centre probes and the lane-parallel rules).
    python tools/stress_r3.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "svt-av1-mod-by-patman_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import pyorc  # noqa: E402
import tf_picture_cases as TP  # noqa: E402
import tpl_cases as T  # noqa: E402
import test_gpu_tf_picture as GTF  # noqa: E402
import test_gpu_tpl as GTPL  # noqa: E402
import me_cases as MC  # noqa: E402
import test_gpu_me as GME  # noqa: E402
from svtav1_hip import abi  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    hip = abi.load()
    assert hip.svt_hip_init(0) == 0
    orc = pyorc.oracle()
    T.load_quant(np.load(os.path.join(ROOT, "tests", "golden", "tpl_frame.npz")))
    rng = np.random.default_rng(2026)
    kinds = ("pan", "blocks", "fastpan", "noise", "static")
    bad = 0
    for i in range(n):
        w, h = int(rng.integers(9, 40)) * 8, int(rng.integers(9, 30)) * 8
        q = int(rng.choice(list(T.QUANT)))
        opt = dict(T.BASE, pf_shape=int(rng.integers(0, 3)), synth_blk_size=int(rng.choice([8, 16])), disable_intra_pred=int(rng.integers(0, 2)),
                   is_ref=int(rng.integers(0, 2)), src_data_ready=int(rng.integers(0, 4) == 0), **T.L3)
        case = (f"stress_l3_{i}", str(rng.choice(kinds)), w, h, q, opt)
        a, b = T.TplScene(orc, case), T.TplScene(orc, case)
        if opt["src_data_ready"]:
            T.prime_second_pass(orc, a), T.prime_second_pass(orc, b)
        got = GTPL.run_gpu(hip, a)
        assert orc.orc_tpl_dispenser_frame(C.byref(b.job())) == 0
        ok = all(np.array_equal(got[k], v) for k, v in b.results().items())
        frac = int((((b.src_stats["mv_row"] & 7) != 0) | ((b.src_stats["mv_col"] & 7) != 0)).sum())
        print(f"tpl3 {i:2d} {case[1]:8s} {w}x{h} q{q} {opt['pf_shape']} fractional {frac:3d} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    tkinds = ("pan", "blocks", "fastpan", "static", "subpel16", "subpel32")
    for i in range(n):
        w, h = int(rng.integers(4, 14)) * 16, int(rng.integers(4, 10)) * 16
        bd = int(rng.choice([8, 10]))
        base = dict(rng.choice([TP.LVL1, TP.LVL6, TP.LVL8]))
        ctl = dict(base, chroma=int(rng.integers(0, 2)), sub_sampling_shift=int(rng.integers(0, 2)))
        if rng.integers(0, 3) == 0:
            ctl["low_delay"] = 1
        else:
            ctl["enable_8x8_pred"] = 1
            ctl["pred_error_32x32_th"] = int(rng.choice([0, 20 * 32 * 32]))
        if bd == 8:
            ctl["use_8bit_subpel"] = 1
        case = (f"stress_tf_{i}", str(rng.choice(tkinds)), w, h, int(rng.integers(1, 4)), bd, "m6_360p_tl0", ctl)
        decay = (2247286, 6156426, 6156426)
        try:
            got, states, tot = GTF.run_gpu(hip, TP.case_window(orc, case), case, decay)
        except Exception as e:  # a size the call refuses (padding of the test window)
            print(f"tf   {i:2d} {w}x{h} refused: {str(e)[-80:]}", flush=True)
            continue
        pics = TP.case_window(orc, case)
        ostates, otot = TP.run_oracle(orc, pics, case, decay)
        ok = tot == otot and np.array_equal(TP.states_to_array(ostates), states) and all(np.array_equal(got[k], v) for k, v in pics[0].arrays().items())
        s16 = sum(sum(s.split16) for s in ostates)
        print(f"tf   {i:2d} {case[1]:8s} {w}x{h} {bd}-bit refs {case[4]} ld {ctl.get('low_delay', 0)} sss {ctl['sub_sampling_shift']} split16 {s16:4d} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    import json
    with open(os.path.join(ROOT, "tests", "golden", "me_params.json")) as f:
        keys = sorted(json.load(f))
    for i in range(n):
        w, h = int(rng.integers(17, 100)) * 8, int(rng.integers(9, 60)) * 8
        key = str(rng.choice(keys))
        tl = int(rng.choice([0, 2]))
        n0, n1 = int(rng.integers(1, 5)), int(rng.integers(0, 4))
        cur = 3
        l0 = [int(x) for x in rng.choice([2, 1, 0], size=n0)]
        l1 = [int(x) for x in rng.choice([4, 5, 6], size=n1)]
        kind = str(rng.choice(("pan", "blocks", "fastpan", "noise", "static", "flat")))
        clip = MC.make_clip(kind, w, h, 7, seed=int(rng.integers(0, 1000)))
        pyrs = MC.build_pyramids(orc, clip)
        prm = MC.scenario_params(key, cur, l0, l1, tl, int(rng.integers(0, 2)))
        tweak = {}
        if rng.integers(0, 2):
            tweak.update(me_sr_div4_th=int(rng.choice([0, 2000, 80000, 10 ** 6])), me_sr_div2_th=int(rng.choice([0, 20000, 150000, 10 ** 7])),
                         me_sr_mult2_th=int(rng.choice([0, 500, 2 ** 32 - 1])), me_8x8_var_enabled=1)
        if rng.integers(0, 3) == 0:
            tweak.update(me_early_exit_th=int(rng.choice([0, 6000, 32768, 400000])))
        if rng.integers(0, 3) == 0:
            tweak.update(me_search_method=int(rng.integers(0, 2)), hme_search_method=int(rng.integers(0, 2)))
        if rng.integers(0, 4) == 0:
            tweak.update(prune_ref_if_hme_sad_dev_bigger_than_th=int(rng.choice([5, 30, 65535])), prune_ref_if_me_sad_dev_bigger_than_th=int(rng.choice([5, 60, 65535])),
                         zz_sad_th=int(rng.choice([0, 10 ** 6])), zz_sad_pct=int(rng.choice([5, 50])), phme_sad_th=int(rng.choice([0, 10 ** 6])), phme_sad_pct=int(rng.choice([5, 50])))
        for k, v in tweak.items():
            setattr(prm, k, v)
        want = MC.run_cpu(orc.orc_me_frame_range, prm, pyrs, cur, l0, l1, w, h)
        got = GME.run_hip_me(hip, prm, pyrs, cur, l0, l1, w, h)[0]
        ok = all(np.array_equal(want[k], got[k]) for k in want)
        print(f"me   {i:2d} {kind:8s} {w}x{h} {key} tl{tl} refs {n0}+{n1} {tweak} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    # Wiener statistics (the int8 MFMA Gram-matrix kernel): random unit sizes, offsets, bit depths, window sizes, several units per call
    import sgr_cases as G
    from lf_cases import V
    from svtav1_hip import device
    for i in range(n):
        bd, is16 = [(8, 0), (8, 1), (10, 1), (12, 1)][int(rng.integers(0, 4))]
        win = int(rng.choice([5, 7]))
        W, H = int(rng.integers(20, 700)), int(rng.integers(20, 500))
        dat, src = G.sgr_plane(rng, W, H, bd, is16, int(rng.integers(0, 3)))
        d_dat, d_src = device.DeviceBuffer(hip, dat.nbytes), device.DeviceBuffer(hip, src.nbytes)
        d_dat.upload(dat), d_src.upload(src)
        off = (G.B * dat.shape[1] + G.B) * dat.itemsize
        uw, uh = int(rng.integers(8, max(9, W))), int(rng.integers(8, max(9, H)))
        limits = [(x, min(x + uw, W), y, min(y + uh, H)) for y in range(0, H, uh) for x in range(0, W, uw)][:40]
        units = (abi.WienerUnit * len(limits))()
        for j, (hs, he, vs, ve) in enumerate(limits):
            units[j] = abi.WienerUnit(d_dat.ptr + off, d_src.ptr + off, dat.shape[1], src.shape[1], hs, he, vs, ve)
        dM, dH = device.DeviceBuffer(hip, len(limits) * 49 * 8), device.DeviceBuffer(hip, len(limits) * 49 * 49 * 8)
        device.check(hip, hip.svt_hip_wiener_stats(units, len(limits), win, is16, bd, V(dM.ptr), V(dH.ptr), None), "wiener_stats")
        device.check(hip, hip.svt_hip_stream_sync(None), "sync")
        Mg, Hg = dM.download(np.int64, (len(limits), 49)), dH.download(np.int64, (len(limits), 49 * 49))
        w2, ok = win * win, True
        for j, (hs, he, vs, ve) in enumerate(limits):
            M, Hh = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
            orc.orc_wiener_compute_stats(win, V(G.at(dat)), V(G.at(src)), hs, he, vs, ve, dat.shape[1], src.shape[1], M.ctypes.data_as(C.c_void_p),
                                         Hh.ctypes.data_as(C.c_void_p), is16, bd)
            ok = ok and np.array_equal(M[:w2], Mg[j, :w2]) and np.array_equal(Hh[:w2 * w2], Hg[j, :w2 * w2])
        print(f"wien {i:2d} {W}x{H} units {uw}x{uh} x{len(limits)} {bd}-bit is16 {is16} win {win} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Shared builders for the whole-picture temporal-filter tests (test infrastructure): windows of pictures with luma pyramids,
4:2:0 chroma and optional 10-bit planes, and the SvtHipTfPictureJob around them."""
import ctypes as C

import numpy as np

import me_cases
from svtav1_hip import abi, frames

# (name, clip kind, width, height, refs, bit depth, ME parameter key, controls)
# controls follow the reference's tf levels (enc_handle.c:2697-3300): level 6/8-like (bilinear, 8-bit sub-pel, h/v half-pel),
# level 1-like without 8x8 (all rounds, eighth-pel, regular filters), sub-sampled distortion + 64x64-only decision, early exits.
LVL6 = dict(half_pel_mode=2, quarter_pel_mode=1, eight_pel_mode=0, use_2tap=1, sub_sampling_shift=0, use_pred_64x64_only_th=0,
            subpel_early_exit_th=1, use_8bit_subpel=1, pred_error_32x32_th=20 * 32 * 32, me_exit_th=0, chroma=1)
LVL1 = dict(half_pel_mode=1, quarter_pel_mode=1, eight_pel_mode=1, use_2tap=0, sub_sampling_shift=0, use_pred_64x64_only_th=0,
            subpel_early_exit_th=0, use_8bit_subpel=0, pred_error_32x32_th=0, me_exit_th=0, chroma=1)
LVL8 = dict(half_pel_mode=2, quarter_pel_mode=1, eight_pel_mode=0, use_2tap=1, sub_sampling_shift=1, use_pred_64x64_only_th=35,
            subpel_early_exit_th=4, use_8bit_subpel=1, pred_error_32x32_th=(1 << 64) - 1, me_exit_th=16 * 16, chroma=0)
ZZ = dict(LVL6, use_zz_based_filter=1, me_exit_th=1500)
CASES = [
    ("pan_lvl6_8bit", "pan", 192, 128, 2, 8, "m8_360p_tl0", LVL6),
    ("blocks_lvl1_8bit", "blocks", 256, 128, 2, 8, "m6_360p_tl0", LVL1),
    ("fastpan_lvl8_8bit", "fastpan", 208, 144, 3, 8, "m8_360p_tl0", LVL8),
    ("static_zz_8bit", "static", 128, 128, 1, 8, "m8_360p_tl0", ZZ),
    ("pan_lvl6_10bit", "pan", 192, 128, 2, 10, "m8_360p_tl0", LVL6),
    ("blocks_lvl1_10bit", "blocks", 128, 192, 1, 10, "m6_360p_tl0", LVL1),
    ("subpel16_lvl6_8bit", "subpel16", 192, 128, 2, 8, "m8_360p_tl0", LVL6),
    ("subpel32_lvl1_8bit", "subpel32", 128, 128, 2, 8, "m6_360p_tl0", dict(LVL1, pred_error_32x32_th=20 * 32 * 32)),
    ("subpel64_lvl6_10bit", "subpel64", 192, 128, 2, 10, "m8_360p_tl0", LVL6),
    ("subpel16_lvl1_10bit", "subpel16", 128, 128, 1, 10, "m6_360p_tl0", LVL1),
    ("static_lvl8_10bit", "static", 128, 64, 2, 10, "m8_360p_tl0", dict(LVL8, chroma=1)),
    ("fastpan_lvl8b_10bit", "fastpan", 144, 80, 2, 10, "m8_360p_tl0", dict(LVL8, use_pred_64x64_only_th=0, me_exit_th=0, use_8bit_subpel=0, chroma=1)),
    # produce_temporally_filtered_pic_ld (pred_structure LOW_DELAY_B): co-located prediction, no search
    ("pan_ld_8bit", "pan", 192, 128, 2, 8, "m8_360p_tl0", dict(LVL6, low_delay=1)),
    ("static_ld_sub_10bit", "static", 144, 80, 3, 10, "m8_360p_tl0", dict(LVL8, low_delay=1, chroma=1)),
    ("blocks_ld_10bit", "blocks", 128, 128, 1, 10, "m6_360p_tl0", dict(LVL1, low_delay=1)),
    # tf level 1 (presets <= M2): enable_8x8_pred — tf_8x8_sub_pel_search, 16x16 -> 8x8 split flags, 8x8 luma / 4x4 chroma predictions
    ("blocks_lvl1_8x8_8bit", "blocks", 256, 128, 2, 8, "m6_360p_tl0", dict(LVL1, enable_8x8_pred=1)),
    ("subpel16_8x8_10bit", "subpel16", 128, 128, 1, 10, "m6_360p_tl0", dict(LVL1, enable_8x8_pred=1)),
    ("fastpan_8x8_sub_8bit", "fastpan", 144, 80, 2, 8, "m6_360p_tl0", dict(LVL1, enable_8x8_pred=1, sub_sampling_shift=1, subpel_early_exit_th=1, use_2tap=1)),
    ("blocks_8x8_10bit_sub8", "blocks", 256, 128, 2, 10, "m6_360p_tl0", dict(LVL1, enable_8x8_pred=1, use_8bit_subpel=1, eight_pel_mode=0)),
]
NOISE_LOG1P_FP16 = (3 << 16) // 4, (1 << 16) // 2, (1 << 16) // 2
QP = 35


class WindowPic:
    """One picture: 8-bit luma pyramid, 8-bit chroma planes, 10-bit planes with the same geometry when bit_depth > 8."""

    def __init__(self, orc, luma10_or_8, cb, cr, bd, poc):
        self.bd, self.poc = bd, poc
        sh = bd - 8
        self.pyr = frames.HostPyramid((luma10_or_8 >> sh).astype(np.uint8))
        d = self.pyr.desc()
        orc.orc_pyramid_frame(C.byref(d.full), C.byref(d.quarter), C.byref(d.sixteenth), 1)
        h, w = luma10_or_8.shape
        self.c8 = [frames.HostPlane(w // 2, h // 2, frames.FULL_PAD // 2, (p >> sh).astype(np.uint8)) for p in (cb, cr)]
        for p in self.c8:
            p.pad_edges()
        self.hbd = None
        if bd > 8:
            self.hbd = []
            for plane, src in ((self.pyr.full, luma10_or_8), (self.c8[0], cb), (self.c8[1], cr)):
                a = np.zeros(plane.buf.shape, np.uint16)
                p = plane.pad
                a[p:p + plane.height, p:p + plane.width] = src
                a[p:p + plane.height, :p] = a[p:p + plane.height, p:p + 1]
                a[p:p + plane.height, p + plane.width:] = a[p:p + plane.height, p + plane.width - 1:p + plane.width]
                a[:p, :] = a[p:p + 1, :]
                a[p + plane.height:, :] = a[p + plane.height - 1:p + plane.height, :]
                self.hbd.append(a)

    def desc(self, ptrs=None):
        """ptrs: optional dict of device addresses {"pyr": Pyramid8, "c8": [..], "hbd": [..]}"""
        t = abi.TfPic()
        t.pyr = self.pyr.desc() if ptrs is None else ptrs["pyr"]
        for i in range(2):
            t.chroma8[i] = self.c8[i].buf.ctypes.data if ptrs is None else ptrs["c8"][i]
        t.chroma8_stride = self.c8[0].stride
        for i in range(3):
            t.hbd[i] = (self.hbd[i].ctypes.data if ptrs is None else ptrs["hbd"][i]) if self.hbd is not None else None
        t.picture_number = self.poc
        return t

    def arrays(self):
        """name -> numpy array of every buffer the filter may write (centre picture)"""
        out = {"y8": self.pyr.full.buf, "cb8": self.c8[0].buf, "cr8": self.c8[1].buf}
        if self.hbd is not None:
            out.update(y16=self.hbd[0], cb16=self.hbd[1], cr16=self.hbd[2])
        return out


def subpel_clip(w, h, n, seed, region):
    """Quarter-sample motion that differs per `region` x `region` area (box-filtered from a 4x finer picture) + sensor noise:
    sub-pel vectors win, and areas with several motions make the 64 / 32 / 16 decisions go every way."""
    rng = np.random.default_rng(seed)
    m = 48
    big = np.kron(rng.integers(0, 256, size=((h + 2 * m) // 4 + 2, (w + 2 * m) // 4 + 2)).astype(np.float32), np.ones((16, 16), np.float32))
    for _ in range(4):
        big = (big + np.roll(big, 3, 0) + np.roll(big, -3, 0) + np.roll(big, 3, 1) + np.roll(big, -3, 1)) / 5
    out = []
    for i in range(n):
        f = np.empty((h, w), np.float32)
        for by in range(0, h, region):
            for bx in range(0, w, region):
                k = (by // region * 5 + bx // region * 3) % 7
                dx, dy = (k - 3) * 3 * i, ((k * 2) % 7 - 3) * 5 * i   # quarter samples
                hh, ww = min(region, h - by), min(region, w - bx)
                y0, x0 = 4 * (m + by) + dy, 4 * (m + bx) + dx
                blk = big[y0:y0 + 4 * hh, x0:x0 + 4 * ww]
                f[by:by + hh, bx:bx + ww] = blk.reshape(hh, 4, ww, 4).mean(axis=(1, 3))
        f += rng.normal(0.0, 1.5, size=f.shape)
        out.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
    return out


def make_window(orc, kind, w, h, n_refs, bd, seed):
    n = n_refs + 1
    rng = np.random.default_rng(seed)
    if kind.startswith("subpel"):
        luma = subpel_clip(w, h, n, seed, int(kind[6:]))
        chroma_src = subpel_clip(w, h, n, seed + 100, int(kind[6:]))
    else:
        luma = me_cases.make_clip(kind, w, h, n, seed=seed)
        chroma_src = me_cases.make_clip(kind, w, h, n, seed=seed + 100)
    order = [n // 2] + [i for i in range(n) if i != n // 2]  # centre first, then its neighbours
    pics = []
    for i in order:
        y = luma[i].astype(np.uint16)
        c = chroma_src[i].astype(np.uint16)
        cb = (c[0::2, 0::2] + c[1::2, 0::2] + c[0::2, 1::2] + c[1::2, 1::2] + 2) >> 2
        cr = 255 - ((y[0::2, 0::2] + y[1::2, 1::2] + 1) >> 1)
        if bd > 8:
            y = (y << 2) | rng.integers(0, 4, size=y.shape, dtype=np.uint16)
            cb = (cb << 2) | rng.integers(0, 4, size=cb.shape, dtype=np.uint16)
            cr = (cr << 2) | rng.integers(0, 4, size=cr.shape, dtype=np.uint16)
        pics.append(WindowPic(orc, y, cb, cr, bd, 16 + i))
    return pics


def make_job(pics, w, h, bd, key, ctl, decay=(0, 0, 0), ptrs=None):
    job = abi.TfPictureJob()
    prm = me_cases.scenario_params(key, 1, [0], [], 0, 1)
    prm.me_mctf, prm.tf_me_exit_th, prm.hme_search_method = 1, ctl["me_exit_th"], 1
    job.me = prm
    for k in ("half_pel_mode", "quarter_pel_mode", "eight_pel_mode", "use_2tap", "sub_sampling_shift", "use_pred_64x64_only_th",
              "subpel_early_exit_th", "use_8bit_subpel", "pred_error_32x32_th"):
        setattr(job.ctrls, k, ctl[k])
    job.ctrls.use_zz_based_filter, job.ctrls.low_delay = ctl.get("use_zz_based_filter", 0), ctl.get("low_delay", 0)
    job.ctrls.enable_8x8_pred = ctl.get("enable_8x8_pred", 0)
    for i in range(3):
        job.decay_factor_fp16[i] = decay[i]
    job.mv_dist_th = min(450, max(64, min(h, w) - 150))
    job.chroma, job.bit_depth = ctl["chroma"], bd
    job.mi_rows, job.mi_cols = ((h + 7) // 8) * 2, ((w + 7) // 8) * 2
    job.n_refs = len(pics) - 1
    job.centre = pics[0].desc(None if ptrs is None else ptrs[0])
    for i, p in enumerate(pics[1:]):
        job.ref[i] = p.desc(None if ptrs is None else ptrs[i + 1])
    return job


def case_window(orc, case):
    name, kind, w, h, n_refs, bd, key, ctl = case
    return make_window(orc, kind, w, h, n_refs, bd, seed=sum(map(ord, name)))


def run_reference(ref, pics, case):
    """-> (decay factors, tf_tot counters); the centre picture of `pics` is filtered in place"""
    name, kind, w, h, n_refs, bd, key, ctl = case
    job = make_job(pics, w, h, bd, key, ctl)
    noise = (C.c_int32 * 3)(*NOISE_LOG1P_FP16)
    decay, tot = (C.c_uint32 * 3)(), (C.c_uint32 * 2)()
    assert ref.ref_tf_picture(C.byref(job), noise, QP, decay, tot) == 0
    return tuple(decay), tuple(tot)


def run_oracle(orc, pics, case, decay):
    name, kind, w, h, n_refs, bd, key, ctl = case
    job = make_job(pics, w, h, bd, key, ctl, decay)
    nb = frames.b64_count(w, h)
    states = (abi.TfB64State * (nb * n_refs))()
    tot = (C.c_uint32 * 2)()
    assert orc.orc_tf_filter_picture(C.byref(job), states, tot) == 0
    return states, tuple(tot)


def states_to_array(states):
    return np.frombuffer(bytes(states), dtype=np.uint8).reshape(len(states), C.sizeof(abi.TfB64State)).copy()

"""Shared builders for the TPL dispenser tests (test infrastructure): a picture with its open-loop ME results, reference
pictures with separate 'reconstructions', and the SvtHipTplFrameJob around them."""
import ctypes as C

import numpy as np

import me_cases
from svtav1_hip import abi, frames

# (name, clip kind, width, height, qindex, options)
BASE = dict(pf_shape=2, disable_intra_pred=0, is_ref=1, i_slice=0, tpl_i_slice=0, src_data_ready=0, store_src_stats=1, synth_blk_size=16)
L5 = dict(blk_size=32, subsample_tx=2)
L3 = dict(quarter_pel=1)
CASES = [
    ("pan_n4", "pan", 192, 128, 120, dict(BASE)),
    ("blocks_full", "blocks", 256, 192, 60, dict(BASE, pf_shape=0)),
    ("fastpan_n2_ragged", "fastpan", 200, 136, 180, dict(BASE, pf_shape=1, synth_blk_size=8)),
    ("noise_intra", "noise", 136, 72, 40, dict(BASE)),
    ("static_nointra_nonref", "static", 128, 128, 100, dict(BASE, disable_intra_pred=1, is_ref=0)),
    ("pan_islice", "pan", 144, 96, 90, dict(BASE, i_slice=1, tpl_i_slice=1)),
    ("blocks_nointra_ref", "blocks", 192, 192, 20, dict(BASE, disable_intra_pred=1, is_ref=1, pf_shape=1)),
    ("pan_second_pass", "pan", 192, 128, 120, dict(BASE, src_data_ready=1)),
    # tpl level 5 (presets M10 and faster): 32x32 blocks, transform TX_32X8 on every 4th row
    ("l5_pan_n4", "pan", 192, 128, 120, dict(BASE, **L5)),
    ("l5_blocks_full_synth32", "blocks", 256, 192, 60, dict(BASE, pf_shape=0, synth_blk_size=32, **L5)),
    ("l5_fastpan_n2_ragged_synth8", "fastpan", 200, 136, 180, dict(BASE, pf_shape=1, synth_blk_size=8, **L5)),
    ("l5_noise_intra_halfrow", "noise", 136, 112, 40, dict(BASE, **L5)),
    ("l5_static_nointra_nonref", "static", 128, 128, 100, dict(BASE, disable_intra_pred=1, is_ref=0, **L5)),
    ("l5_pan_islice", "pan", 160, 96, 90, dict(BASE, i_slice=1, tpl_i_slice=1, **L5)),
    ("l5_pan_second_pass", "pan", 192, 128, 120, dict(BASE, src_data_ready=1, **L5)),
    ("l5_pan_half_column", "pan", 376, 216, 120, dict(BASE, **L5)),   # the last 32x32 column / row is 24 samples inside the picture
    # tpl level 3 (presets M5 / M6): quarter-pel refinement of every candidate (tpl_subpel_search), 8-tap compensation
    ("l3_pan_n4", "pan", 192, 128, 120, dict(BASE, **L3)),
    ("l3_blocks_full", "blocks", 256, 192, 60, dict(BASE, pf_shape=0, **L3)),
    ("l3_fastpan_n2_ragged", "fastpan", 200, 136, 180, dict(BASE, pf_shape=1, synth_blk_size=8, **L3)),
    ("l3_noise_intra", "noise", 136, 72, 40, dict(BASE, **L3)),
    ("l3_static_nointra_nonref", "static", 128, 128, 100, dict(BASE, disable_intra_pred=1, is_ref=0, **L3)),
    ("l3_fastpan_second_pass", "fastpan", 200, 136, 180, dict(BASE, pf_shape=1, src_data_ready=1, **L3)),
]
# round_fp[2], quant_fp[2], dequant[2] of the reference's 8-bit tables (svt_av1_build_quantizer) at the qindex values above,
# read from the reference by tests/golden/make_golden_tpl.py and checked against it in test_tpl_oracle.py
QUANT = {}


def stats_cells(w, h, synth):
    aw, ah = (w + 7) // 8 * 8, (h + 7) // 8 * 8
    if synth == 32:
        return ((aw + 31) // 32) * ((ah + 31) // 32)
    g = 1 if synth == 16 else 2
    return ((aw + 15) >> 4) * g * ((ah + 15) >> 4) * g


class TplScene:
    def __init__(self, orc, case, key="m8_360p_tl0"):
        name, kind, w, h, qindex, opt = case
        self.case, self.w, self.h = case, w, h
        seed = sum(map(ord, name))
        rng = np.random.default_rng(seed)
        clip = me_cases.make_clip(kind, w, h, 5, seed=seed)
        self.pyrs = me_cases.build_pyramids(orc, clip)
        self.cur, l0, l1 = 2, [1, 0], [3, 4]
        self.prm = me_cases.scenario_params(key, self.cur, l0, l1, 0, 1)
        self.me = me_cases.run_cpu(orc.orc_me_frame_range, self.prm, self.pyrs, self.cur, l0, l1, w, h)
        self.ref_pocs = [l0, l1]
        # 'reconstructions' of the reference pictures that are inside the sliding window: the source + coding noise
        self.recon = {}
        for poc in (1, 3, 4):
            p = frames.HostPlane(w, h, frames.FULL_PAD, np.clip(clip[poc].astype(np.int16) + rng.integers(-3, 4, size=(h, w)), 0, 255).astype(np.uint8))
            p.pad_edges()
            self.recon[poc] = p
        self.unusable = {4}
        # this picture's TPL reconstruction buffer holds stale data before the call
        self.out = frames.HostPlane(w, h, frames.FULL_PAD)
        self.out.buf[...] = rng.integers(0, 256, size=self.out.buf.shape, dtype=np.uint8)
        a16, rows16 = ((w + 7) // 8 * 8 + 15) >> 4, ((h + 7) // 8 * 8 + 15) >> 4
        self.stats = np.zeros(stats_cells(w, h, opt["synth_blk_size"]), dtype=np.dtype(abi.TplStats))
        self.src_stats = np.zeros(a16 * rows16, dtype=np.dtype(abi.TplSrcStats))

    def job(self, ptr=None):
        """ptr: optional function host array / plane -> device address"""
        name, kind, w, h, qindex, opt = self.case
        at = (lambda a: a.ctypes.data) if ptr is None else ptr
        j = abi.TplFrameJob()

        def plane(hp):
            return abi.Plane8(at(hp.buf), hp.stride, hp.pad, hp.pad, hp.width, hp.height)

        def sample0(hp):
            return at(hp.buf) + hp.pad * hp.stride + hp.pad
        j.src, j.recon = plane(self.pyrs[self.cur].full), plane(self.out)
        for l in range(2):
            for r, poc in enumerate(self.ref_pocs[l]):
                f = j.ref[l][r]
                sp = self.pyrs[poc].full
                f.src, f.src_stride = sample0(sp), sp.stride
                rp = self.recon.get(poc, sp)
                f.recon, f.recon_stride = sample0(rp), rp.stride
                f.picture_number, f.max_width, f.max_height = 100 + poc, w, h
                f.usable = 0 if poc in self.unusable else 1
        j.me_mv_array, j.me_candidate_array = at(self.me["me_mv_array"]), at(self.me["me_candidate_array"])
        j.total_me_candidate_index = at(self.me["total_me_candidate_index"])
        j.max_cand, j.max_refs, j.max_l0 = self.prm.max_cand, self.prm.max_refs, self.prm.max_l0
        j.enable_me_16x16, j.stored_pus = self.prm.enable_me_16x16, self.prm.stored_pus()
        for k, v in opt.items():
            setattr(j, k, v)
        q = QUANT[qindex]
        for i in range(2):
            j.round_fp[i], j.quant_fp[i], j.dequant[i] = q[i], q[2 + i], q[4 + i]
        j.stats, j.src_stats = at(self.stats), at(self.src_stats)
        return j

    def results(self):
        return {"recon": self.out.buf, "stats": self.stats.view(np.uint8), "src_stats": self.src_stats.view(np.uint8)}


def prime_second_pass(orc, scene):
    """src_data_ready: the source-based statistics come from an earlier pass over the same picture"""
    first = TplScene(orc, (scene.case[0],) + scene.case[1:5] + (dict(scene.case[5], src_data_ready=0),))
    assert orc.orc_tpl_dispenser_frame(C.byref(first.job())) == 0
    scene.src_stats[...] = first.src_stats


def load_quant(gold):
    for k in gold.files:
        if k.startswith("quant_"):
            QUANT[int(k[6:])] = [int(x) for x in gold[k]]

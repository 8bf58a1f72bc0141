"""ctypes mirror of include/svt_hip.h and include/svt_hip_me.h (the C-ABI of libsvtav1_hip).

This module only describes the binary interface; it contains no compute and no CPU fallback.
`load()` raises if the HIP library has not been built (product path must fail loudly).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.path.join(PKG_ROOT, "csrc", "libsvtav1_hip.so")

ME_MAX_LIST, ME_MAX_REF, ME_SQUARE_PUS = 2, 4, 85


class Plane8(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("stride", C.c_uint32), ("org_x", C.c_uint16), ("org_y", C.c_uint16),
                ("width", C.c_uint16), ("height", C.c_uint16)]


class SearchArea(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16)]


class MeParams(C.Structure):
    _fields_ = [
        ("hme_search_method", C.c_uint8), ("me_search_method", C.c_uint8),
        ("enable_hme_flag", C.c_uint8), ("enable_hme_level0_flag", C.c_uint8),
        ("enable_hme_level1_flag", C.c_uint8), ("enable_hme_level2_flag", C.c_uint8),
        ("num_hme_sa_w", C.c_uint8), ("num_hme_sa_h", C.c_uint8),
        ("hme_l0_sa_min", SearchArea), ("hme_l0_sa_max", SearchArea), ("hme_l1_sa", SearchArea),
        ("hme_l2_sa", SearchArea), ("me_sa_min", SearchArea), ("me_sa_max", SearchArea),
        ("prehme_enable", C.c_uint8), ("prehme_skip_search_line", C.c_uint8),
        ("prehme_l1_early_exit", C.c_uint8), ("me_mctf", C.c_uint8),
        ("prehme_sa_min", SearchArea * 2), ("prehme_sa_max", SearchArea * 2),
        ("enable_me_hme_ref_pruning", C.c_uint8), ("pad1_", C.c_uint8),
        ("prune_ref_if_hme_sad_dev_bigger_than_th", C.c_uint16),
        ("prune_ref_if_me_sad_dev_bigger_than_th", C.c_uint16),
        ("zz_sad_pct", C.c_uint16), ("phme_sad_pct", C.c_uint16), ("tf_me_exit_th", C.c_uint16),
        ("zz_sad_th", C.c_uint32), ("phme_sad_th", C.c_uint32),
        ("enable_me_sr_adjustment", C.c_uint8), ("distance_based_hme_resizing", C.c_uint8),
        ("reduce_me_sr_based_on_mv_length_th", C.c_uint16), ("stationary_hme_sad_abs_th", C.c_uint16),
        ("stationary_me_sr_divisor", C.c_uint16), ("reduce_me_sr_based_on_hme_sad_abs_th", C.c_uint16),
        ("me_sr_divisor_for_low_hme_sad", C.c_uint16),
        ("me_8x8_var_enabled", C.c_uint8), ("pad3_", C.c_uint8 * 3),
        ("me_sr_div4_th", C.c_uint32), ("me_sr_div2_th", C.c_uint32), ("me_sr_mult2_th", C.c_uint32),
        ("mv_sa_adj_enabled", C.c_uint8), ("mv_sa_adj_nearest_ref_only", C.c_uint8),
        ("mv_sa_adj_mv_size_th", C.c_uint16), ("mv_sa_adj_sa_multiplier", C.c_uint16),
        ("reduce_hme_l0_sr_th_min", C.c_uint8), ("reduce_hme_l0_sr_th_max", C.c_uint8),
        ("me_early_exit_th", C.c_uint32), ("me_safe_limit_zz_th", C.c_uint32),
        ("prev_me_stage_based_exit_th", C.c_uint32), ("prune_me_candidates_th", C.c_int32),
        ("use_best_unipred_cand_only", C.c_uint8),
        ("num_of_list_to_search", C.c_uint8), ("num_of_ref_pic_to_search", C.c_uint8 * 2),
        ("temporal_layer_index", C.c_uint8), ("is_ref", C.c_uint8), ("hierarchical_levels", C.c_uint8),
        ("similar_brightness_refs", C.c_uint8),
        ("enable_me_8x8", C.c_uint8), ("enable_me_16x16", C.c_uint8), ("max_number_of_pus_per_sb", C.c_uint8),
        ("max_cand", C.c_uint8), ("max_refs", C.c_uint8), ("max_l0", C.c_uint8),
        ("only_l_bwd", C.c_uint8), ("input_resolution_le_480p", C.c_uint8), ("pad4_", C.c_uint8 * 2),
        ("picture_number", C.c_uint64), ("ref_picture_number", (C.c_uint64 * 4) * 2),
    ]

    def stored_pus(self):
        return (85 if self.enable_me_8x8 else 21) if self.enable_me_16x16 else 5

    def to_dict(self):
        def conv(v):
            if isinstance(v, SearchArea):
                return [v.width, v.height]
            if hasattr(v, "__len__"):
                return [conv(x) for x in v]
            return int(v)
        return {n: conv(getattr(self, n)) for n, _ in self._fields_ if not n.startswith("pad")}

    @classmethod
    def from_dict(cls, d):
        p = cls()
        for n, t in cls._fields_:
            if n.startswith("pad") or n not in d:
                continue
            v = d[n]
            if t is SearchArea:
                setattr(p, n, SearchArea(*v))
            elif n in ("prehme_sa_min", "prehme_sa_max"):
                for i in range(2):
                    getattr(p, n)[i] = SearchArea(*v[i])
            elif n == "ref_picture_number":
                for l in range(2):
                    for r in range(4):
                        p.ref_picture_number[l][r] = v[l][r]
            elif n == "num_of_ref_pic_to_search":
                p.num_of_ref_pic_to_search[0], p.num_of_ref_pic_to_search[1] = v
            else:
                setattr(p, n, v)
        return p


class Pyramid8(C.Structure):
    _fields_ = [("full", Plane8), ("quarter", Plane8), ("sixteenth", Plane8)]


class MeSearchResult(C.Structure):
    _fields_ = [("hme_sad", C.c_uint64), ("hme_sc_x", C.c_int16), ("hme_sc_y", C.c_int16),
                ("do_ref", C.c_uint8), ("pad_", C.c_uint8 * 3)]


class MeFrameOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "best_sad", "best_mv", "search_results", "me_mv_array", "me_candidate_array",
        "total_me_candidate_index", "me_64x64_distortion", "me_32x32_distortion", "me_16x16_distortion",
        "me_8x8_distortion", "me_8x8_cost_variance", "rc_me_distortion")]


class MeFrameJob(C.Structure):
    _fields_ = [("prm", MeParams), ("src", Pyramid8), ("ref", (Pyramid8 * 4) * 2), ("out", MeFrameOut)]


class SadLoopDesc(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("ref_off", C.c_uint64), ("src_stride", C.c_uint32),
                ("ref_stride", C.c_uint32), ("src_stride_raw", C.c_uint32), ("block_width", C.c_uint16),
                ("block_height", C.c_uint16), ("search_area_width", C.c_int16),
                ("search_area_height", C.c_int16), ("skip_search_line", C.c_uint8), ("pad_", C.c_uint8 * 7)]


class SadLoopResult(C.Structure):
    _fields_ = [("best_sad", C.c_uint64), ("x", C.c_int16), ("y", C.c_int16), ("pad_", C.c_uint32)]


_lib = None


def load():
    """Load libsvtav1_hip.so (built by __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is None:
        # tuning aid: SVTAV1_HIP_LIB names another build of the SAME library (kernel variants side by side in one GPU call)
        global LIB_PATH
        LIB_PATH = os.environ.get("SVTAV1_HIP_LIB", LIB_PATH)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP path has no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.svt_hip_last_error.restype = C.c_char_p
        _lib.svt_hip_version.restype = C.c_char_p
        _lib.svt_nxm_sad_kernel_hip.restype = C.c_uint32
        for n in ("svt_compute_sub_mean_8x8_hip", "svt_compute_mean_8x8_hip",
                  "svt_compute_mean_square_values_8x8_hip"):
            getattr(_lib, n).restype = C.c_uint64
    return _lib


NO_OFFSET = 0xFFFFFFFFFFFFFFFF
# SvtHipStatus (include/svt_hip.h:34-39), as the signed int32 values a ctypes call returns
SVT_HIP_OK = 0
SVT_HIP_ERR_NO_DEVICE = 0x80001000 - (1 << 32)
SVT_HIP_ERR_BAD_PARAMETER = 0x80001005 - (1 << 32)
SVT_HIP_ERR_RUNTIME = 0x80001001 - (1 << 32)
QUANT_NONE, QUANT_B, QUANT_B_HBD, QUANT_FP, QUANT_FP_HBD = range(5)
TX_FWD, TX_INV, TX_PIXEL16, TX_FULLCOEFF, TX_SRC_PRED, TX_SATD = 1, 2, 4, 8, 16, 32
TXFM_RESULT_BYTES = 16


class TxfmDesc(C.Structure):
    _fields_ = [("residual_off", C.c_uint64), ("coeff_off", C.c_uint64), ("qcoeff_off", C.c_uint64),
                ("dqcoeff_off", C.c_uint64), ("pred_off", C.c_uint64), ("recon_off", C.c_uint64),
                ("iscan_off", C.c_uint64), ("qm_off", C.c_uint64), ("iqm_off", C.c_uint64),
                ("residual_stride", C.c_uint32), ("pred_stride", C.c_uint32), ("recon_stride", C.c_uint32),
                ("zbin", C.c_int16 * 2), ("round", C.c_int16 * 2), ("quant", C.c_int16 * 2),
                ("quant_shift", C.c_int16 * 2), ("dequant", C.c_int16 * 2),
                ("tx_type", C.c_uint8), ("shape", C.c_uint8), ("bit_depth", C.c_uint8), ("quant_mode", C.c_uint8),
                ("log_scale", C.c_uint8), ("flags", C.c_uint8), ("dist_w", C.c_uint8), ("dist_h", C.c_uint8)]


class TxfmResult(C.Structure):
    _fields_ = [("three_quad_energy", C.c_uint64), ("eob", C.c_uint16), ("pad_", C.c_uint16), ("satd", C.c_uint32)]


class CdefList(C.Structure):
    _fields_ = [("by", C.c_uint8), ("bx", C.c_uint8)]


class CdefPlane(C.Structure):
    _fields_ = [("recon", C.c_void_p), ("source", C.c_void_p), ("recon_stride", C.c_uint32), ("source_stride", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("is_16bit", C.c_uint8), ("xdec", C.c_uint8),
                ("ydec", C.c_uint8), ("pli", C.c_uint8)]


class CdefSearchParams(C.Structure):
    _fields_ = [("n_strengths", C.c_int32), ("strengths", C.c_int8 * 64), ("pri_damping", C.c_int32),
                ("sec_damping", C.c_int32), ("coeff_shift", C.c_int32), ("subsampling_factor", C.c_int32)]


class LfMi(C.Structure):          # SvtHipLfMi (include/svt_hip_lf.h)
    _fields_ = [(n, C.c_uint8) for n in ("bsize", "tx_size_y", "tx_size_uv", "skip_inter", "segment_id", "ref_frame0", "mode_lf",
                                         "reserved")]


class LfFrame(C.Structure):       # SvtHipLfFrame
    _fields_ = [("plane", C.c_void_p * 3), ("stride", C.c_uint32 * 3), ("width", C.c_uint32), ("height", C.c_uint32),
                ("mi", C.c_void_p), ("mi_stride", C.c_uint32), ("mi_rows", C.c_uint32), ("mi_cols", C.c_uint32),
                ("lvl", C.c_uint8 * 768), ("filter_level", C.c_uint8 * 2), ("filter_level_u", C.c_uint8),
                ("filter_level_v", C.c_uint8), ("sharpness_level", C.c_uint8), ("bit_depth", C.c_uint8), ("is_16bit", C.c_uint8),
                ("plane_start", C.c_uint8), ("plane_end", C.c_uint8), ("reserved", C.c_uint8 * 3)]


LF_MI_DTYPE = [(n, "u1") for n in ("bsize", "tx_size_y", "tx_size_uv", "skip_inter", "segment_id", "ref_frame0", "mode_lf", "reserved")]


class SgrParams(C.Structure):     # SvtHipSgrParams == SgrParamsType
    _fields_ = [("r", C.c_int32 * 2), ("s", C.c_int32 * 2)]


class SgrUnit(C.Structure):       # SvtHipSgrUnit
    _fields_ = [("dat", C.c_void_p), ("src", C.c_void_p), ("dat_stride", C.c_uint32), ("src_stride", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("is_16bit", C.c_uint8), ("bit_depth", C.c_uint8),
                ("pu_w", C.c_uint8), ("pu_h", C.c_uint8)]


SGR_PARAMS = [(2, 1, 140, 3236), (2, 1, 112, 2158), (2, 1, 93, 1618), (2, 1, 80, 1438), (2, 1, 70, 1295), (2, 1, 58, 1177),
              (2, 1, 47, 1079), (2, 1, 37, 996), (2, 1, 30, 925), (2, 1, 25, 863), (0, 1, -1, 2589), (0, 1, -1, 1618),
              (0, 1, -1, 1177), (0, 1, -1, 925), (2, 0, 56, -1), (2, 0, 22, -1)]   # r0, r1, s0, s1 (AV1 Sgr_Params)


class AnalysisJob(C.Structure):   # SvtHipAnalysisJob
    _fields_ = [("pyr", Pyramid8), ("variance", C.c_void_p), ("mean", C.c_void_p)]


class WienerUnit(C.Structure):    # SvtHipWienerUnit
    _fields_ = [("dgd", C.c_void_p), ("src", C.c_void_p), ("dgd_stride", C.c_uint32), ("src_stride", C.c_uint32),
                ("h_start", C.c_int32), ("h_end", C.c_int32), ("v_start", C.c_int32), ("v_end", C.c_int32)]


LR_EXTRA_HORZ = 4                   # SVT_HIP_LR_EXTRA_HORZ


class LrUnit(C.Structure):          # SvtHipLrUnit
    _fields_ = [("restoration_type", C.c_uint8), ("ep", C.c_uint8), ("pad_", C.c_int16), ("xqd", C.c_int32 * 2),
                ("hfilter", C.c_int16 * 8), ("vfilter", C.c_int16 * 8)]


import numpy as np  # noqa: E402

LR_UNIT_DTYPE = np.dtype([("restoration_type", np.uint8), ("ep", np.uint8), ("pad_", np.int16), ("xqd", np.int32, 2),
                          ("hfilter", np.int16, 8), ("vfilter", np.int16, 8)])
assert LR_UNIT_DTYPE.itemsize == C.sizeof(LrUnit) == 44


class LrPlane(C.Structure):         # SvtHipLrPlane
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_stride", C.c_uint32), ("dst_stride", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("ss_x", C.c_uint8), ("ss_y", C.c_uint8), ("is_16bit", C.c_uint8),
                ("bit_depth", C.c_uint8), ("unit_size", C.c_uint32), ("horz_units", C.c_uint32), ("vert_units", C.c_uint32),
                ("units", C.c_void_p), ("boundary_above", C.c_void_p), ("boundary_below", C.c_void_p),
                ("boundary_stride", C.c_uint32), ("optimized_lr", C.c_uint32)]


class ConvolveParams(C.Structure):   # SvtHipConvolveParams == ConvolveParams (definitions.h:580-593)
    _fields_ = [("ref", C.c_int32), ("do_average", C.c_int32), ("dst", C.c_void_p), ("dst_stride", C.c_int32), ("round_0", C.c_int32),
                ("round_1", C.c_int32), ("plane", C.c_int32), ("is_compound", C.c_int32), ("use_jnt_comp_avg", C.c_int32),
                ("fwd_offset", C.c_int32), ("bck_offset", C.c_int32), ("use_dist_wtd_comp_avg", C.c_int32)]


class InterpFilterParams(C.Structure):   # SvtHipInterpFilterParams == InterpFilterParams (definitions.h:750-755)
    _fields_ = [("filter_ptr", C.c_void_p), ("taps", C.c_uint16), ("subpel_shifts", C.c_uint16), ("interp_filter", C.c_int32)]


class ConvolveDesc(C.Structure):         # SvtHipConvolveDesc (include/svt_hip_inter.h)
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_stride", C.c_uint32), ("dst_stride", C.c_uint32), ("w", C.c_uint16),
                ("h", C.c_uint16), ("filter_x", C.c_int16 * 8), ("filter_y", C.c_int16 * 8), ("taps_x", C.c_uint8), ("taps_y", C.c_uint8),
                ("round_0", C.c_uint8), ("round_1", C.c_uint8), ("bit_depth", C.c_uint8), ("is_16bit", C.c_uint8), ("compound", C.c_uint8),
                ("fwd_offset", C.c_uint8), ("bck_offset", C.c_uint8), ("pad_", C.c_uint8 * 3), ("cbuf", C.c_void_p),
                ("cbuf_stride", C.c_uint32), ("pad2_", C.c_uint32)]


class TfBlock(C.Structure):              # SvtHipTfBlock (include/svt_hip_tf.h)
    _fields_ = [("src", C.c_void_p * 3), ("pred", C.c_void_p * 3), ("accum", C.c_void_p * 3), ("count", C.c_void_p * 3),
                ("src_stride", C.c_uint32 * 3), ("pred_stride", C.c_uint32 * 3), ("decay_factor_fp16", C.c_uint32 * 3),
                ("block_error", C.c_uint64 * 4), ("mv_x", C.c_int16 * 4), ("mv_y", C.c_int16 * 4), ("mv_dist_th", C.c_uint16),
                ("split", C.c_uint8), ("chroma", C.c_uint8), ("ss_x", C.c_uint8), ("ss_y", C.c_uint8), ("is_16bit", C.c_uint8),
                ("bit_depth", C.c_uint8), ("zz_based", C.c_uint8), ("pad_", C.c_uint8 * 7)]


class TfOut(C.Structure):                # SvtHipTfOut
    _fields_ = [("dst", C.c_void_p * 3), ("dst_stride", C.c_uint32 * 3), ("pad_", C.c_uint32)]


TF_MAX_REFS = 32


class TfCtrls(C.Structure):              # SvtHipTfCtrls
    _fields_ = [("half_pel_mode", C.c_uint8), ("quarter_pel_mode", C.c_uint8), ("eight_pel_mode", C.c_uint8), ("use_2tap", C.c_uint8),
                ("sub_sampling_shift", C.c_uint8), ("use_pred_64x64_only_th", C.c_uint8), ("subpel_early_exit_th", C.c_uint8),
                ("use_8bit_subpel", C.c_uint8), ("use_zz_based_filter", C.c_uint8), ("enable_8x8_pred", C.c_uint8), ("low_delay", C.c_uint8),
                ("pad_", C.c_uint8 * 5),
                ("pred_error_32x32_th", C.c_uint64)]


class TfPic(C.Structure):                # SvtHipTfPic
    _fields_ = [("pyr", Pyramid8), ("chroma8", C.c_void_p * 2), ("chroma8_stride", C.c_uint32), ("pad_", C.c_uint32), ("hbd", C.c_void_p * 3),
                ("picture_number", C.c_uint64)]


class TfPictureJob(C.Structure):         # SvtHipTfPictureJob
    _fields_ = [("me", MeParams), ("ctrls", TfCtrls), ("decay_factor_fp16", C.c_uint32 * 3), ("mv_dist_th", C.c_uint16), ("chroma", C.c_uint8),
                ("bit_depth", C.c_uint8), ("mi_rows", C.c_uint32), ("mi_cols", C.c_uint32), ("n_refs", C.c_uint32), ("centre", TfPic),
                ("ref", TfPic * TF_MAX_REFS), ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64), ("tot_blks", C.c_void_p)]


class TfB64State(C.Structure):           # SvtHipTfB64State
    _fields_ = [("err64", C.c_uint64), ("err32", C.c_uint64 * 4), ("err16", C.c_uint64 * 16), ("mv64_x", C.c_int16), ("mv64_y", C.c_int16),
                ("mv32_x", C.c_int16 * 4), ("mv32_y", C.c_int16 * 4), ("mv16_x", C.c_int16 * 16), ("mv16_y", C.c_int16 * 16),
                ("split32", C.c_uint8 * 4), ("use_64x64", C.c_uint8), ("pad_", C.c_uint8 * 3), ("err8", C.c_uint64 * 64),
                ("mv8_x", C.c_int16 * 64), ("mv8_y", C.c_int16 * 64), ("split16", C.c_uint8 * 16)]


class TplStats(C.Structure):             # SvtHipTplStats (include/svt_hip_tpl.h)
    _fields_ = [("srcrf_dist", C.c_int64), ("recrf_dist", C.c_int64), ("srcrf_rate", C.c_int64), ("recrf_rate", C.c_int64),
                ("mc_dep_rate", C.c_int64), ("mc_dep_dist", C.c_int64), ("mv_row", C.c_int16), ("mv_col", C.c_int16), ("pad_", C.c_uint32),
                ("ref_frame_poc", C.c_uint64)]


class TplSrcStats(C.Structure):          # SvtHipTplSrcStats
    _fields_ = [("srcrf_dist", C.c_int64), ("srcrf_rate", C.c_int64), ("ref_frame_poc", C.c_uint64), ("mv_row", C.c_int16), ("mv_col", C.c_int16),
                ("best_rf_idx", C.c_int32), ("best_mode", C.c_uint8), ("best_intra_mode", C.c_uint8), ("pad_", C.c_uint8 * 6)]


class TplRef(C.Structure):               # SvtHipTplRef
    _fields_ = [("src", C.c_void_p), ("recon", C.c_void_p), ("src_stride", C.c_uint32), ("recon_stride", C.c_uint32), ("picture_number", C.c_uint64),
                ("max_width", C.c_uint16), ("max_height", C.c_uint16), ("usable", C.c_uint8), ("pad_", C.c_uint8 * 3)]


class TplFrameJob(C.Structure):          # SvtHipTplFrameJob
    _fields_ = [("src", Plane8), ("recon", Plane8), ("ref", (TplRef * 4) * 2), ("me_mv_array", C.c_void_p), ("me_candidate_array", C.c_void_p),
                ("total_me_candidate_index", C.c_void_p), ("max_cand", C.c_uint8), ("max_refs", C.c_uint8), ("max_l0", C.c_uint8),
                ("enable_me_16x16", C.c_uint8), ("stored_pus", C.c_uint8), ("pf_shape", C.c_uint8), ("disable_intra_pred", C.c_uint8),
                ("is_ref", C.c_uint8), ("i_slice", C.c_uint8), ("tpl_i_slice", C.c_uint8), ("src_data_ready", C.c_uint8),
                ("store_src_stats", C.c_uint8), ("synth_blk_size", C.c_uint8), ("blk_size", C.c_uint8), ("subsample_tx", C.c_uint8), ("publish_fence", C.c_uint8),
                ("round_fp", C.c_int16 * 2),
                ("quant_fp", C.c_int16 * 2), ("dequant", C.c_int16 * 2), ("quarter_pel", C.c_uint8), ("pad2_", C.c_uint8), ("stats", C.c_void_p), ("src_stats", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64)]


class Mv(C.Structure):                   # SvtHipMv == MV (block_structures.h:26-29)
    _fields_ = [("row", C.c_int16), ("col", C.c_int16)]


class MvCostParam(C.Structure):          # SvtHipMvCostParam == struct svt_mv_cost_param (mcomp.h:37-49)
    _fields_ = [("ref_mv", C.POINTER(Mv)), ("full_ref_mv", Mv), ("mv_cost_type", C.c_uint8), ("mvjcost", C.c_void_p),
                ("mvcost", C.c_void_p * 2), ("error_per_bit", C.c_int), ("early_exit_th", C.c_int), ("sad_per_bit", C.c_int)]


class TxfmParam(C.Structure):            # SvtHipTxfmParam == TxfmParam (definitions.h:1051-1063)
    _fields_ = [("tx_type", C.c_uint8), ("tx_size", C.c_uint8), ("lossless", C.c_int32), ("bd", C.c_int32), ("is_hbd", C.c_int32),
                ("tx_set_type", C.c_uint8), ("eob", C.c_int32)]

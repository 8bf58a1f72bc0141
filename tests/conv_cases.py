"""Shared helpers for the inter-prediction interpolation tests (test infrastructure)."""
import ctypes as C

import numpy as np

V = C.c_void_p


class InterpFilterParams(C.Structure):     # definitions.h:750-755 (== SvtHipInterpFilterParams)
    _fields_ = [("filter_ptr", C.c_void_p), ("taps", C.c_uint16), ("subpel_shifts", C.c_uint16), ("interp_filter", C.c_int32)]


# The AV1 interpolation kernels (inter_prediction.c:223-300 holds the same tables): 16 phases x 8 taps.  Tests read the
# reference's own copies when oracle/_ref is present and check this restatement against them (test_convolve_oracle.py).
REGULAR = [[0, 0, 0, 128, 0, 0, 0, 0], [0, 2, -6, 126, 8, -2, 0, 0], [0, 2, -10, 122, 18, -4, 0, 0], [0, 2, -12, 116, 28, -8, 2, 0],
           [0, 2, -14, 110, 38, -10, 2, 0], [0, 2, -14, 102, 48, -12, 2, 0], [0, 2, -16, 94, 58, -12, 2, 0], [0, 2, -14, 84, 66, -12, 2, 0],
           [0, 2, -14, 76, 76, -14, 2, 0], [0, 2, -12, 66, 84, -14, 2, 0], [0, 2, -12, 58, 94, -16, 2, 0], [0, 2, -12, 48, 102, -14, 2, 0],
           [0, 2, -10, 38, 110, -14, 2, 0], [0, 2, -8, 28, 116, -12, 2, 0], [0, 0, -4, 18, 122, -10, 2, 0], [0, 0, -2, 8, 126, -6, 2, 0]]
SHARP = [[0, 0, 0, 128, 0, 0, 0, 0], [-2, 2, -6, 126, 8, -2, 2, 0], [-2, 6, -12, 124, 16, -6, 4, -2], [-2, 8, -18, 120, 26, -10, 6, -2],
         [-4, 10, -22, 116, 38, -14, 6, -2], [-4, 10, -22, 108, 48, -18, 8, -2], [-4, 10, -24, 100, 60, -20, 8, -2],
         [-4, 10, -24, 90, 70, -22, 10, -2], [-4, 12, -24, 80, 80, -24, 12, -4], [-2, 10, -22, 70, 90, -24, 10, -4],
         [-2, 8, -20, 60, 100, -24, 10, -4], [-2, 8, -18, 48, 108, -22, 10, -4], [-2, 6, -14, 38, 116, -22, 10, -4],
         [-2, 6, -10, 26, 120, -18, 8, -2], [-2, 4, -6, 16, 124, -12, 6, -2], [0, 2, -2, 8, 126, -6, 2, -2]]
BILINEAR = [[0, 0, 0, 128 - 8 * i, 8 * i, 0, 0, 0] for i in range(16)]
TABLES = {"sub_pel_filters_8": REGULAR, "sub_pel_filters_8sharp": SHARP, "bilinear_filters": BILINEAR}


def kernel_table(name):
    """256-byte aligned int16 [16][8] copy of a kernel table (what InterpFilterParams.filter_ptr points at)."""
    buf = np.zeros(16 * 8 + 128, np.int16)
    off = (-buf.ctypes.data % 256) // 2
    t = buf[off:off + 128].reshape(16, 8)
    t[:] = np.array(TABLES[name], np.int16)
    return t, buf


def conv_rounds(bd):
    """get_conv_params (convolve.h:40-68), non-compound"""
    r0, r1 = 3, 11
    rng = bd + 7 - r0 + 2
    if rng > 16:
        r0, r1 = r0 + rng - 16, r1 - (rng - 16)
    return r0, r1


def ref_plane(rng, w, h, bd, is16, kind):
    B = 8
    dt = np.uint16 if is16 else np.uint8
    if kind == 2:
        a = rng.integers(0, 1 << bd, size=(h + 2 * B, w + 2 * B))
    elif kind == 1:
        a = np.where(rng.random((h + 2 * B, w + 2 * B)) < 0.5, 0, (1 << bd) - 1)
    else:
        yy, xx = np.mgrid[0:h + 2 * B, 0:w + 2 * B]
        a = (1 << bd) * (0.5 + 0.4 * np.sin(xx / 5.0) * np.cos(yy / 7.0)) + rng.integers(-4, 5, size=xx.shape)
    a = np.clip(np.rint(a), 0, (1 << bd) - 1).astype(dt)
    return a, a.ctypes.data + (B * a.shape[1] + B) * a.itemsize


SIZES = [(4, 4), (8, 8), (16, 8), (8, 16), (32, 32), (64, 64), (128, 64), (64, 128), (128, 128), (4, 16), (16, 64)]


def conv_rounds_compound(bd):
    """get_conv_params_no_round with is_compound = 1 (convolve.h:39-63)"""
    r0, r1 = 3, 7
    rng = bd + 7 - r0 + 2
    if rng > 16:
        r0 += rng - 16
    return r0, r1


JNT_MODES = ("2d", "x", "y", "2d_copy")
# quant_dist_lookup_table weights (fwd, bck) the reference uses for distance-weighted compounds (inter_prediction.c: the
# pairs sum to 16)
DIST_WEIGHTS = [(9, 7), (11, 5), (12, 4), (13, 3), (7, 9), (5, 11), (4, 12), (3, 13), (8, 8)]


def jnt_cases(bd, is16, n=40, seed=0):
    """(w, h, mode, table index, sx, sy, plane0, at0, plane1, at1, averaging mode 2|3, fwd, bck)"""
    rng = np.random.default_rng(7000 + bd + seed)
    for trial in range(n):
        w, h = SIZES[trial % len(SIZES)]
        ti = trial % 3
        sx, sy = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        mode = trial % 4
        p0, a0 = ref_plane(rng, w, h, bd, is16, (0, 2, 1)[trial % 3])
        p1, a1 = ref_plane(rng, w, h, bd, is16, (2, 0, 1)[trial % 3])
        avg = 3 if trial % 2 else 2
        fwd, bck = DIST_WEIGHTS[trial % len(DIST_WEIGHTS)]
        yield w, h, mode, ti, sx, sy, p0, a0, p1, a1, avg, fwd, bck

"""Shared inputs for the self-guided restoration tests (test infrastructure)."""
import numpy as np

import lf_cases as L

B = 8   # border kept around every test plane (the filter reads 3)


def sgr_plane(rng, w, h, bd, is16, kind):
    """Degraded picture + original: smooth content with noise (kind 0), flat (1), random (2)."""
    dt = np.uint16 if is16 else np.uint8
    if kind == 1:
        dat = np.full((h + 2 * B, w + 2 * B), int(rng.integers(0, 1 << bd)))
    elif kind == 2:
        dat = rng.integers(0, 1 << bd, size=(h + 2 * B, w + 2 * B))
    else:
        dat = L.smooth_plane(rng, w + 2 * B, h + 2 * B, bd) + rng.integers(-3, 4, size=(h + 2 * B, w + 2 * B)) * (1 << (bd - 8))
    dat = np.clip(dat, 0, (1 << bd) - 1).astype(dt)
    src = np.clip(dat.astype(np.int32) + rng.integers(-5, 6, size=dat.shape) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(dt)
    return dat, src


def at(a):
    return a.ctypes.data + (B * a.shape[1] + B) * a.itemsize


GOLDEN_SGR = [  # key, w, h, bd, is16, kind, pu, (start, end, inc, refine), seed
    ("a_8", 136, 72, 8, 0, 0, 64, (0, 16, 1, 1), 31), ("b_10", 96, 80, 10, 1, 0, 64, (0, 16, 3, 1), 32),
    ("c_8_chroma", 72, 40, 8, 0, 0, 32, (10, 16, 1, 0), 33), ("d_10_noise", 64, 64, 10, 1, 2, 64, (0, 16, 5, 1), 34)]


# ---- Wiener restoration
def wiener_rounds(bd):
    """get_conv_params_wiener (convolve.h:70-86)"""
    r0, r1 = 3, 11
    rng = bd + 7 - r0 + 2
    if rng > 16:
        r0, r1 = r0 + rng - 16, r1 - (rng - 16)
    return r0, r1


def wiener_filter(rng):
    """A symmetric 7-tap Wiener kernel as the encoder codes it: taps sum to 128, the centre tap is stored minus 128,
    8th coefficient zero; 256-byte aligned like the reference's kernel tables (convolve.c:45-54)."""
    t0, t1, t2 = int(rng.integers(-5, 11)), int(rng.integers(-23, 9)), int(rng.integers(-17, 47))
    c = 128 - 2 * (t0 + t1 + t2)
    buf = np.zeros(8 + 128, np.int16)
    off = (-buf.ctypes.data % 256) // 2
    f = buf[off:off + 8]
    f[:] = [t0, t1, t2, c - 128, t2, t1, t0, 0]
    return f, buf


GOLDEN_WIENER = [("a_8_win7", 8, 0, 7, 100, 72, 41), ("b_10_win5", 10, 1, 5, 72, 40, 42), ("c_10_win7", 10, 1, 7, 136, 64, 43)]

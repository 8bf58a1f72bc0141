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

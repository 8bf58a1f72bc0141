"""Shared inputs of the frame-level loop-restoration tests (test infrastructure): CDEF-output planes, the saved stripe
boundary rows, a random restoration-unit table (RESTORE_NONE / WIENER / SGRPROJ) and the flat SvtHipLrPlane records."""
import ctypes as C

import numpy as np

import lf_cases as L
import sgr_cases as G
from svtav1_hip import abi

# name: width, height (luma), bit depth, is16, luma restoration-unit size, optimized_lr, seed
CASES = {
    "a_8bit": (200, 136, 8, 0, 64, 0, 1),
    "b_10bit": (328, 200, 10, 1, 128, 0, 2),
    "c_opt_10bit": (264, 152, 10, 1, 64, 1, 3),
    "d_8in16": (136, 264, 8, 1, 64, 0, 4),
    "e_one_unit": (72, 40, 10, 1, 256, 0, 5),
    "f_opt_h57": (136, 121, 8, 0, 64, 1, 6),    # H = 64 + 57: the rows below the second stripe run past the picture
}


DST_EXTRA = 40


def units_in(size, unit):
    return max((size + (unit >> 1)) // unit, 1)             # count_units_in_tile (restoration.c:124-126)


def make_case(name, planes=(0, 1, 2), dims=None):
    w, h, bd, is16, us, opt, seed = dims if dims else CASES[name]
    rng = np.random.default_rng(1000 + seed)
    dt = np.uint16 if is16 else np.uint8
    out = []
    for p in planes:
        ss = int(p > 0)
        pw, ph, unit = w >> ss, h >> ss, max(us >> ss, 32)
        src = np.clip(L.smooth_plane(rng, pw, ph, bd) + rng.integers(-3, 4, size=(ph, pw)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(dt)
        n_stripes = (ph + (8 >> ss) + (64 >> ss) - 1) // (64 >> ss)
        bstride = pw + 2 * abi.LR_EXTRA_HORZ + 8
        above = rng.integers(0, 1 << bd, size=(2 * n_stripes, bstride)).astype(dt)
        below = rng.integers(0, 1 << bd, size=(2 * n_stripes, bstride)).astype(dt)
        hu, vu = units_in(pw, unit), units_in(ph, unit)
        units = np.zeros(vu * hu, abi.LR_UNIT_DTYPE)
        for u in units:
            u["restoration_type"] = int(rng.integers(0, 3))
            u["ep"] = int(rng.integers(0, 16))
            r0, r1 = abi.SGR_PARAMS[int(u["ep"])][:2]
            u["xqd"] = (int(rng.integers(-96, 32)) if r0 else 0, int(rng.integers(-32, 96)) if r1 else 0)
            if r0 == 0:                                        # encode_xq keeps xqd[0] = 0 / clamps xqd[1] (restoration_pick.c)
                u["xqd"] = (0, int(np.clip(128 - 0 - int(rng.integers(33, 128)), -32, 95)))
            u["hfilter"] = G.wiener_filter(rng)[0]
            u["vfilter"] = G.wiener_filter(rng)[0]
        out.append(dict(src=src, above=above, below=below, units=units, w=pw, h=ph, ss=ss, bd=bd, is16=is16, unit=unit, hu=hu, vu=vu,
                        opt=opt, bstride=bstride))
    return out


def lr_planes(case, ptr_of=lambda a: a.ctypes.data, dsts=None, units_ptr=None):
    """-> (ctypes array of SvtHipLrPlane, list of dst arrays).  ptr_of maps a host array to the address to put in the record."""
    arr = (abi.LrPlane * len(case))()
    outs = []
    for i, c in enumerate(case):
        dst = np.zeros((c["h"] + 2, c["w"] + DST_EXTRA), c["src"].dtype) if dsts is None else dsts[i]   # the reference's Wiener stripe filter rounds the width of a unit's last column of processing units up to 16 (restoration.c:444): room to the right
        outs.append(dst)
        arr[i] = abi.LrPlane(ptr_of(c["src"]), ptr_of(dst) if dsts is None else dsts[i], c["w"], c["w"] + DST_EXTRA, c["w"], c["h"], c["ss"], c["ss"], c["is16"],
                             c["bd"], c["unit"], c["hu"], c["vu"], ptr_of(c["units"]) if units_ptr is None else units_ptr[i],
                             0 if c["opt"] else ptr_of(c["above"]), 0 if c["opt"] else ptr_of(c["below"]), c["bstride"], c["opt"])
    return arr, outs

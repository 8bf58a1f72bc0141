"""Shared case generators for the remaining per-call RTCD leaves (test infrastructure): 16-bit SAD, buffer fill, residual,
spatial distortion, the PME SAD + motion-vector-cost search, the CDEF strength-pair step and the 8-bit inverse transform entry."""
import ctypes as C

import numpy as np

from svtav1_hip import abi
from tx_cases import P, SIZES, V

MV_TABLE = 1 << 14   # MV_UPP (cabac_context_model.h:524): component cost tables are indexed -MV_TABLE .. MV_TABLE


def sad16_cases():
    rng = np.random.default_rng(901)
    for k in range(16):
        h, w = int(rng.choice([4, 8, 16, 32, 64, 128])), int(rng.choice([4, 8, 16, 32, 64, 128]))
        ss, rs = w + int(rng.integers(0, 19)), w + int(rng.integers(0, 19))
        hi = 1 << (10 if k % 3 else 16)
        yield h, w, rng.integers(0, hi, size=(h, ss)).astype(np.uint16), rng.integers(0, hi, size=(h, rs)).astype(np.uint16)


def residual_cases():
    rng = np.random.default_rng(902)
    for k in range(16):
        h, w = int(rng.choice([4, 8, 16, 32, 64, 128])), int(rng.choice([4, 8, 16, 32, 64, 128]))
        s0, s1, s2 = (w + int(rng.integers(0, 11)) for _ in range(3))
        hbd = k % 2 == 1
        hi, dt = ((1 << (16 if k % 4 == 3 else 10)), np.uint16) if hbd else (256, np.uint8)
        yield h, w, s2, rng.integers(0, hi, size=(h, s0)).astype(dt), rng.integers(0, hi, size=(h, s1)).astype(dt), hbd


def sse_cases():
    rng = np.random.default_rng(903)
    for k in range(16):
        h, w = int(rng.choice([4, 8, 16, 32, 64, 128])), int(rng.choice([4, 8, 16, 32, 64, 128]))
        s0, s1 = w + int(rng.integers(0, 40)), w + int(rng.integers(0, 40))
        o0, o1 = int(rng.integers(0, 3 * s0)), int(rng.integers(0, 3 * s1))
        hbd = k % 2 == 1
        hi, dt = ((1 << (16 if k % 4 == 3 else 10)), np.uint16) if hbd else (256, np.uint8)
        yield h, w, o0, o1, rng.integers(0, hi, size=(h + 4, s0)).astype(dt), rng.integers(0, hi, size=(h + 4, s1)).astype(dt), hbd


class PmeCase:
    pass


def pme_cases():
    """Blocks and search grids of md_full_pel_search (product_coding_loop.c:1830-1990): widths that are and are not
    multiples of 8, sparse steps, every cost type, flat areas (ties), an incoming best that nothing beats."""
    rng = np.random.default_rng(904)
    tab = [rng.integers(0, 1 << 14, size=2 * MV_TABLE + 1).astype(np.int32) for _ in range(2)]
    joint = rng.integers(0, 4000, size=4).astype(np.int32)
    k = 0
    for bw, bh in ((8, 8), (16, 16), (32, 16), (64, 64), (16, 64), (128, 128), (4, 8)):
        for saw, sah, step in ((8, 1, 1), (16, 5, 1), (31, 7, 2), (24, 9, 4), (40, 17, 8), (7, 3, 1), (13, 4, 3)):
            c = PmeCase()
            c.bw, c.bh, c.saw, c.sah, c.step = bw, bh, saw, sah, step
            c.ss, c.rs = bw + int(rng.integers(0, 9)), bw + saw + int(rng.integers(0, 9))
            c.src = rng.integers(0, 256, size=(bh, c.ss)).astype(np.uint8)
            c.ref = rng.integers(0, 256, size=(bh + sah, c.rs)).astype(np.uint8)
            if k % 5 == 1:      # flat: every position has the same SAD, the cost term / scan order decides
                c.src[:], c.ref[:] = 100, 90
            if k % 5 == 2:      # smooth: neighbouring positions nearly equal
                c.ref[:] = (np.add.outer(np.arange(bh + sah), np.arange(c.rs)) // 3 % 256).astype(np.uint8)
                c.src[:] = c.ref[sah // 2:sah // 2 + bh, saw // 2:saw // 2 + c.ss] if saw // 2 + c.ss <= c.rs else c.src
            c.type = k % 6
            c.epb = int(rng.integers(1, 3000))
            c.start_x, c.start_y = int(rng.integers(-40, 8)), int(rng.integers(-40, 8))
            c.mvx, c.mvy = int(rng.integers(-2000, 2000)), int(rng.integers(-2000, 2000))
            c.ref_mv = (int(rng.integers(-2000, 2000)), int(rng.integers(-2000, 2000)))    # (row, col)
            if k % 7 == 3:      # the search crosses the reference vector: zero differences hit the joint classes
                c.ref_mv = (c.mvy + 8 * (c.start_y + (sah // 2) // step * step), c.mvx + 8 * (c.start_x + 2))
            if k % 11 == 5:     # far apart: differences clamp at the table ends / wrap in 16 bits
                c.mvx, c.ref_mv = 30000, (-30000, -30000)
            c.best = (0xFFFFFFFF, bw * bh * 90, 0x7FFFFFFF, 5)[k % 4]
            c.tab, c.joint = tab, joint
            k += 1
            yield c


def pme_param(c):
    """(struct, keep-alive objects)"""
    ref_mv = abi.Mv(c.ref_mv[0], c.ref_mv[1])
    p = abi.MvCostParam()
    p.ref_mv = C.pointer(ref_mv)
    p.mv_cost_type = c.type
    p.mvjcost = c.joint.ctypes.data
    p.mvcost[0] = c.tab[0].ctypes.data + 4 * MV_TABLE
    p.mvcost[1] = c.tab[1].ctypes.data + 4 * MV_TABLE
    p.error_per_bit = c.epb
    return p, ref_mv


PME_ARGS = (V, V, C.c_uint32, V, C.c_uint32, C.c_uint32, C.c_uint32, V, V, V) + (C.c_int16,) * 7


def run_pme(fn, c):
    """fn has the RTCD signature (the reference pointer or the HIP export)."""
    p, keep = pme_param(c)
    best = np.array([c.best], np.uint32)
    mv = np.array([-77, 99], np.int16)
    fn(C.byref(p), P(c.src), c.ss, P(c.ref), c.rs, c.bh, c.bw, P(best), V(mv.ctypes.data), V(mv.ctypes.data + 2), c.start_x, c.start_y,
       c.saw, c.sah, c.step, c.mvx, c.mvy)
    del keep
    return int(best[0]), int(mv[0]), int(mv[1])


def run_pme_orc(orc, c):
    best = np.array([c.best], np.uint32)
    mv = np.array([-77, 99], np.int16)
    orc.orc_pme_sad_loop(C.c_int16(c.ref_mv[0]), C.c_int16(c.ref_mv[1]), c.type, P(c.joint), V(c.tab[0].ctypes.data + 4 * MV_TABLE),
                         V(c.tab[1].ctypes.data + 4 * MV_TABLE), c.epb, P(c.src), C.c_uint32(c.ss), P(c.ref), C.c_uint32(c.rs),
                         C.c_uint32(c.bh), C.c_uint32(c.bw), P(best), V(mv.ctypes.data), V(mv.ctypes.data + 2), C.c_int16(c.start_x),
                         C.c_int16(c.start_y), C.c_int16(c.saw), C.c_int16(c.sah), C.c_int16(c.step), C.c_int16(c.mvx), C.c_int16(c.mvy))
    return int(best[0]), int(mv[0]), int(mv[1])


def dual_cases():
    """(lev0, lev1, nb, mse[2][sb][64], start_gi, end_gi): totals of finish_cdef_search (enc_cdef.c:728-) incl. ties and huge values."""
    rng = np.random.default_rng(905)
    for k in range(18):
        sb = int(rng.choice([1, 7, 60, 510, 2040]))
        end_gi = int(rng.choice([4, 9, 16, 64]))
        start_gi = 0 if k % 3 else int(rng.integers(0, end_gi))
        nb = k % 8
        mse = rng.integers(0, 1 << int(rng.choice([8, 24, 40])), size=(2, sb, 64)).astype(np.uint64)
        if k % 4 == 1:
            mse[:] = mse // np.uint64(1 << 6) * np.uint64(1 << 6)     # many equal sums
        if k % 6 == 2:
            mse[:] = 12345                                          # all equal: the first pair wins
        if k % 9 == 4:
            mse[:, :, :] = np.uint64(1) << np.uint64(62)            # totals pass 1 << 63 and wrap
        lev0 = np.concatenate([rng.integers(0, end_gi, size=nb), np.zeros(8 - nb)]).astype(np.int32)
        lev1 = np.concatenate([rng.integers(0, end_gi, size=nb), np.zeros(8 - nb)]).astype(np.int32)
        yield lev0, lev1, nb, mse, start_gi, end_gi


def run_dual(fn, case):
    """fn with the RTCD signature: mse passed as uint64_t **mse[2]."""
    lev0, lev1, nb, mse, start_gi, end_gi = case
    l0, l1 = lev0.copy(), lev1.copy()
    sb = mse.shape[1]
    rows = [(C.c_void_p * sb)(*[mse[p, i].ctypes.data for i in range(sb)]) for p in range(2)]
    top = (C.c_void_p * 2)(C.addressof(rows[0]), C.addressof(rows[1]))
    tot = fn(P(l0), P(l1), nb, top, sb, start_gi, end_gi)
    return int(tot), l0, l1


def run_dual_orc(orc, case):
    lev0, lev1, nb, mse, start_gi, end_gi = case
    l0, l1 = lev0.copy(), lev1.copy()
    orc.orc_search_one_dual.restype = C.c_uint64
    tot = orc.orc_search_one_dual(P(l0), P(l1), nb, P(mse), mse.shape[1], 64, start_gi, end_gi)
    return int(tot), l0, l1


def inv8_cases(orc):
    """(tx_size index, w, h, tx_type, coefficients, prediction, strides)"""
    import tx_cases as T
    rng = np.random.default_rng(906)
    for ti, (w, h) in enumerate(SIZES):
        # the dispatcher admits DCT_DCT / IDTX only for 32x32 and DCT_DCT only for 64x64 (inv_transforms.c:2868-2891)
        cand = (0,) if max(w, h) == 64 else ((0, 9) if max(w, h) == 32 else (0, 1, 9, 10, 15))
        types = [t for t in cand if orc.orc_txfm_valid(w, h, t)]
        for tt in types:
            for trial in (0, 2):
                co = T.coeffs_for_inverse(rng, orc, w, h, tt, 8, trial)
                ps, rs = w + int(rng.integers(0, 9)), w + int(rng.integers(0, 9))
                yield ti, w, h, tt, co, rng.integers(0, 256, size=(h, ps)).astype(np.uint8), rs

"""Shared case builders for the in-loop filter parity tests (test infrastructure)."""
import ctypes as C

import numpy as np

from svtav1_hip import abi

BS, VL, VB, HB = 144, 0x7F7F, 3, 8
V = C.c_void_p


def P(a):
    return V(a.ctypes.data)


def cdef_tile(rng, bd, edge=0):
    """A CDEF input tile (u16, stride 144, 64+6 rows) as cdef_seg_search builds it; `edge` bit-mask puts
    CDEF_VERY_LARGE on the top/left/bottom/right borders like at picture boundaries."""
    t = rng.integers(0, 1 << bd, size=(64 + 2 * VB, BS)).astype(np.uint16)
    if edge & 1:
        t[:VB, :] = VL
    if edge & 2:
        t[:, :HB] = VL
    if edge & 4:
        t[VB + 64:, :] = VL
    if edge & 8:
        t[:, HB + 64:] = VL
    return t


def smooth_plane(rng, w, h, bd):
    """Picture-like content (edges + gradients + noise) so that CDEF directions and clamps are exercised."""
    y, x = np.mgrid[0:h, 0:w]
    img = 0.5 * (1 << bd) + 0.3 * (1 << bd) * np.sin(x / 9.0 + y / 17.0) + 0.15 * (1 << bd) * ((x // 16 + y // 24) % 2)
    img += rng.normal(0, (1 << bd) / 64.0, size=(h, w))
    return np.clip(np.rint(img), 0, (1 << bd) - 1)


def cdef_plane(recon, source, pli, xdec, ydec, is16):
    return abi.CdefPlane(recon.ctypes.data, source.ctypes.data, recon.strides[0] // recon.itemsize,
                         source.strides[0] // source.itemsize, recon.shape[1] if False else 0, 0, is16, xdec, ydec, pli)


def search_params(strengths, damping, coeff_shift, sub):
    p = abi.CdefSearchParams()
    p.n_strengths = len(strengths)
    for i, s in enumerate(strengths):
        p.strengths[i] = s
    p.pri_damping = p.sec_damping = damping
    p.coeff_shift, p.subsampling_factor = coeff_shift, sub
    return p


def rtcd(ref, name, restype, *argtypes):
    p = C.c_void_p.in_dll(ref, name).value
    assert p, name
    return C.CFUNCTYPE(restype, *argtypes)(p)


def ref_cdef_plane(ref, recon, source, w, h, is16, xdec, ydec, pli, filt, strengths, fbs, damping, cs, sub, ldir, lvar):
    """One plane through the REFERENCE's own per-filter-block functions (svt_aom_cdef_find_dir, svt_cdef_filter_fb,
    svt_compute_cdef_dist_*), tiles built as cdef_seg_search (cdef_process.c:204-221) builds them.
    Luma fills ldir/lvar [n_fb][64]; chroma reads them.  Returns (mse[n_fb][n_strengths], applied plane)."""
    lw, lh = w << xdec, h << ydec
    w8, h8, nhfb, nvfb = (lw + 7) // 8, (lh + 7) // 8, (lw + 63) // 64, (lh + 63) // 64
    stride = recon.shape[1]
    fdir = rtcd(ref, "svt_aom_cdef_find_dir", C.c_uint8, V, C.c_int32, V, C.c_int32)
    f16 = rtcd(ref, "svt_compute_cdef_dist_16bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    f8 = rtcd(ref, "svt_compute_cdef_dist_8bit", C.c_uint64, V, C.c_int32, V, V, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8)
    bsz = 3 if (xdec, ydec) == (0, 0) else 0 if (xdec, ydec) == (1, 1) else 1 if xdec else 2
    eff_sub = min(sub, 4) if bsz == 3 else 1 if bsz == 0 else min(sub, 2)
    bw, bh = 64 >> xdec, 64 >> ydec
    mse = np.full((nhfb * nvfb, len(strengths)), 0xABCD, np.uint64)
    applied = np.zeros_like(recon)
    applied[:, :w] = recon[:, :w]
    for fby in range(nvfb):
        for fbx in range(nhfb):
            fb = fby * nhfb + fbx
            dl = (abi.CdefList * 64)()
            n = 0
            for r in range(8):
                for c in range(8):
                    if fby * 8 + r < h8 and fbx * 8 + c < w8 and filt[fby * 8 + r, fbx * 8 + c]:
                        dl[n].by, dl[n].bx = r, c
                        n += 1
            if n == 0:
                continue
            tile = np.full((64 + 2 * VB, BS), VL, np.uint16)
            for y in range(-VB, bh + VB):
                py = fby * bh + y
                if 0 <= py < h:
                    x0, x1 = max(fbx * bw - HB, 0), min(fbx * bw + bw + HB, w)
                    tile[VB + y, HB + x0 - fbx * bw:HB + x1 - fbx * bw] = recon[py, x0:x1]
            inp = tile.ctypes.data + 2 * (VB * BS + HB)
            if pli == 0:
                for i in range(n):
                    v = C.c_int32(0)
                    ldir[fb, dl[i].by * 8 + dl[i].bx] = fdir(inp + 2 * (8 * dl[i].by * BS + 8 * dl[i].bx), BS, C.addressof(v), cs)
                    lvar[fb, dl[i].by * 8 + dl[i].bx] = v.value
            d16, v16 = np.zeros((16, 16), np.uint8), np.zeros((16, 16), np.int32)
            d16[:8, :8], v16[:8, :8] = ldir[fb].reshape(8, 8), lvar[fb].reshape(8, 8)
            soff = (fby * bh) * stride + fbx * bw
            for gi, s in enumerate(strengths):
                if s < 0:
                    continue
                pri, sec = s // 4, s % 4
                tmp = np.zeros(64 * 64, np.uint16)
                dirinit = C.c_int32(1)
                ref.svt_cdef_filter_fb(None if is16 else P(tmp), P(tmp) if is16 else None, 0, V(inp), xdec, ydec, P(d16), C.byref(dirinit),
                                       P(v16), pli, C.byref(dl), n, pri, sec + (sec == 3), damping, damping, cs, C.c_uint8(eff_sub))
                m = (f16 if is16 else f8)(source.ctypes.data + soff * source.itemsize, stride, tmp.ctypes.data, C.addressof(dl), n, bsz, cs,
                                          pli, eff_sub)
                mse[fb, gi] = m * eff_sub
            s = int(fbs[fb])
            pri, sec = s // 4, s % 4
            if pri or sec:
                dirinit = C.c_int32(1)
                dst = applied.ctypes.data + soff * recon.itemsize
                ref.svt_cdef_filter_fb(None if is16 else V(dst), V(dst) if is16 else None, stride, V(inp), xdec, ydec, P(d16),
                                       C.byref(dirinit), P(v16), pli, C.byref(dl), n, pri, sec + (sec == 3), damping, damping, cs,
                                       C.c_uint8(1))
    return mse, applied


GOLDEN_CDEF = [  # key, luma w, luma h, bd, is16, fmt, sub, seed
    ("a_420_8", 200, 136, 8, 0, 420, 2, 5), ("b_420_10", 136, 72, 10, 1, 420, 1, 6), ("c_444_8in16", 72, 136, 8, 1, 444, 4, 7),
    ("d_420_10_tall", 72, 200, 10, 1, 420, 4, 8)]


def golden_cdef_inputs(lw, lh, bd, is16, fmt, sub, seed):
    """Seeded inputs of one golden picture: yields per plane (pli, xdec, ydec, w, h, recon, source) + shared parameters."""
    rng = np.random.default_rng(seed)
    w8, h8, nhfb, nvfb = lw // 8, lh // 8, (lw + 63) // 64, (lh + 63) // 64
    filt = (rng.random((h8, w8)) < 0.75).astype(np.uint8)
    strengths = [0, 5, 18, 35, 63, -1, 12, 2]
    damping = 3 + seed % 4
    fbs = rng.choice(np.array([s for s in strengths if s >= 0], np.uint8), size=nhfb * nvfb).astype(np.uint8)
    dt = np.uint16 if is16 else np.uint8
    planes = []
    for pli in range(3):
        xdec, ydec = int(pli > 0 and fmt != 444), int(pli > 0 and fmt == 420)
        w, h = lw >> xdec, lh >> ydec
        recon = smooth_plane(rng, w + 11, h, bd).astype(dt)
        source = np.clip(recon.astype(np.int32) + rng.integers(-6, 7, size=recon.shape), 0, (1 << bd) - 1).astype(dt)
        planes.append((pli, xdec, ydec, w, h, recon, source))
    return filt, strengths, damping, fbs, planes


# ------------------------------------------------------------------------------------------------ deblocking
# BlockSize enum (definitions.h): value by (width, height)
BSIZE = {(4, 4): 0, (4, 8): 1, (8, 4): 2, (8, 8): 3, (8, 16): 4, (16, 8): 5, (16, 16): 6, (16, 32): 7, (32, 16): 8, (32, 32): 9,
         (32, 64): 10, (64, 32): 11, (64, 64): 12, (64, 128): 13, (128, 64): 14, (128, 128): 15, (4, 16): 16, (16, 4): 17,
         (8, 32): 18, (32, 8): 19, (16, 64): 20, (64, 16): 21}


class RefLfModeInfo(C.Structure):   # oracle/ref_harness_lf.c
    _fields_ = [(n, C.c_void_p) for n in ("bsize", "tx_depth", "skip", "ref_frame0", "mode", "segment_id")]


class RefLfHeader(C.Structure):
    _fields_ = [("filter_level", C.c_int32 * 2), ("filter_level_u", C.c_int32), ("filter_level_v", C.c_int32),
                ("sharpness_level", C.c_int32), ("mode_ref_delta_enabled", C.c_uint8), ("ref_deltas", C.c_int8 * 8),
                ("mode_deltas", C.c_int8 * 2), ("segmentation_enabled", C.c_uint8), ("seg_lf_data", (C.c_int16 * 4) * 8),
                ("seg_lf_enabled", (C.c_uint8 * 4) * 8)]


def random_mode_info(rng, mi_rows, mi_cols, mi_stride, sb=64, p_skip=0.45, p_intra=0.3, max_depth=2):
    """A random but structurally valid block partition of the picture: per-4x4 arrays of the fields
    set_lpf_parameters reads (bsize, tx_depth, skip, ref_frame[0], mode, segment_id)."""
    f = {k: np.zeros((mi_rows, mi_stride), np.uint8) for k in ("bsize", "tx_depth", "skip", "ref_frame0", "mode", "segment_id")}

    def leaf(x, y, w, h):
        if x >= mi_cols * 4 or y >= mi_rows * 4:
            return
        intra = rng.random() < p_intra
        vals = dict(bsize=BSIZE[(w, h)], tx_depth=int(rng.integers(0, max_depth + 1)), skip=int(rng.random() < p_skip),
                    ref_frame0=0 if intra else int(rng.integers(1, 8)), mode=int(rng.integers(0, 13)) if intra else int(rng.integers(13, 25)),
                    segment_id=int(rng.integers(0, 8)))
        for k, v in vals.items():
            f[k][y // 4:min((y + h) // 4, mi_rows), x // 4:min((x + w) // 4, mi_cols)] = v

    def part(x, y, s):
        if x >= mi_cols * 4 or y >= mi_rows * 4:
            return
        kinds = ["none", "split", "horz", "vert"] if s > 4 else ["none"]
        if 16 <= s <= 64:
            kinds += ["horz4", "vert4"]
        if s >= 16:
            kinds += ["horz_a", "horz_b", "vert_a", "vert_b"]
        w = [3 if k == "split" and s > 16 else 1 for k in kinds]
        k = kinds[int(rng.choice(len(kinds), p=np.array(w) / sum(w)))]
        hs = s // 2
        if k == "none":
            leaf(x, y, s, s)
        elif k == "split":
            for dy in (0, hs):
                for dx in (0, hs):
                    part(x + dx, y + dy, hs)
        elif k == "horz":
            leaf(x, y, s, hs), leaf(x, y + hs, s, hs)
        elif k == "vert":
            leaf(x, y, hs, s), leaf(x + hs, y, hs, s)
        elif k == "horz4":
            for i in range(4):
                leaf(x, y + i * s // 4, s, s // 4)
        elif k == "vert4":
            for i in range(4):
                leaf(x + i * s // 4, y, s // 4, s)
        elif k == "horz_a":
            leaf(x, y, hs, hs), leaf(x + hs, y, hs, hs), leaf(x, y + hs, s, hs)
        elif k == "horz_b":
            leaf(x, y, s, hs), leaf(x, y + hs, hs, hs), leaf(x + hs, y + hs, hs, hs)
        elif k == "vert_a":
            leaf(x, y, hs, hs), leaf(x, y + hs, hs, hs), leaf(x + hs, y, hs, s)
        elif k == "vert_b":
            leaf(x, y, hs, s), leaf(x + hs, y, hs, hs), leaf(x + hs, y + hs, hs, hs)

    for y in range(0, mi_rows * 4, sb):
        for x in range(0, mi_cols * 4, sb):
            part(x, y, sb)
    return f


def lf_header(rng, variant):
    """Frame-header loop-filter parameters; `variant` walks through the branches of svt_av1_loop_filter_frame_init."""
    h = RefLfHeader()
    h.filter_level[0], h.filter_level[1] = int(rng.integers(1, 64)), int(rng.integers(1, 64))
    h.filter_level_u, h.filter_level_v = int(rng.integers(1, 64)), int(rng.integers(1, 64))
    h.sharpness_level = (0, 3, 7, 5)[variant % 4]
    if variant % 3 == 1:
        h.mode_ref_delta_enabled = 1
        for i, v in enumerate((1, 0, 0, 0, -1, 0, -1, -1)):      # the AV1 default ref deltas
            h.ref_deltas[i] = v + int(rng.integers(-2, 3))
        h.mode_deltas[0], h.mode_deltas[1] = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
    if variant % 3 == 2:
        h.segmentation_enabled = 1
        for s in range(8):
            for k in range(4):
                h.seg_lf_enabled[s][k] = int(rng.random() < 0.6)
                h.seg_lf_data[s][k] = int(rng.integers(-40, 41))
    if variant == 5:
        h.filter_level_u = 0                                      # plane switched off
    if variant == 7:
        h.filter_level[0] = h.filter_level[1] = 0                 # luma off => everything off (the `break`, :570-572)
    return h


PAD = 32


def lf_planes(rng, w, h, bd, is16):
    """Three padded 4:2:0 planes with blocky content (so that every filter and both flat branches fire)."""
    dt = np.uint16 if is16 else np.uint8
    out = []
    for pl in range(3):
        pw, ph = (w >> (pl > 0)) + 2 * PAD, (h >> (pl > 0)) + 2 * PAD
        yy, xx = np.mgrid[0:ph, 0:pw]
        img = (1 << bd) * (0.5 + 0.25 * np.sin(xx / 23.0) * np.cos(yy / 31.0))
        img += (1 << (bd - 8)) * rng.integers(-6, 7, size=(ph // 8 + 1, pw // 8 + 1)).repeat(8, 0).repeat(8, 1)[:ph, :pw]   # block steps
        img += (1 << (bd - 8)) * rng.integers(-40, 41, size=(ph // 32 + 1, pw // 32 + 1)).repeat(32, 0).repeat(32, 1)[:ph, :pw] * (rng.random() < 0.5)
        img += rng.integers(-1, 2, size=(ph, pw)) * (1 << (bd - 8)) * (rng.random((ph, pw)) < 0.3)
        out.append(np.clip(np.rint(img), 0, (1 << bd) - 1).astype(dt))
    return out


def lf_frame(planes, w, h, mi_ptr, mi_stride, mi_rows, mi_cols, hdr, bd, is16, plane_start=0, plane_end=3, lvl=None):
    f = abi.LfFrame()
    for i, p in enumerate(planes):
        if isinstance(p, np.ndarray):
            f.plane[i] = p.ctypes.data + (PAD * p.shape[1] + PAD) * p.itemsize
            f.stride[i] = p.shape[1]
        else:
            f.plane[i], f.stride[i] = p
    f.width, f.height, f.mi, f.mi_stride, f.mi_rows, f.mi_cols = w, h, mi_ptr, mi_stride, mi_rows, mi_cols
    f.filter_level[0], f.filter_level[1] = hdr.filter_level[0], hdr.filter_level[1]
    f.filter_level_u, f.filter_level_v, f.sharpness_level = hdr.filter_level_u, hdr.filter_level_v, hdr.sharpness_level
    f.bit_depth, f.is_16bit, f.plane_start, f.plane_end = bd, is16, plane_start, plane_end
    if lvl is not None:
        C.memmove(f.lvl, lvl.ctypes.data, 768)
    return f


def ref_deblock(ref, planes, w, h, minfo, mi_stride, mi_rows, mi_cols, hdr, bd, is16, sb_size=64, plane_start=0, plane_end=3):
    """Run the REAL svt_av1_loop_filter_frame in place on `planes`; returns (flat SvtHipLfMi array, lvl table)."""
    m = RefLfModeInfo(*[minfo[k].ctypes.data for k in ("bsize", "tx_depth", "skip", "ref_frame0", "mode", "segment_id")])
    flat = np.zeros((mi_rows, mi_stride), abi.LF_MI_DTYPE)
    ref.ref_lf_gather(mi_rows * mi_stride, C.byref(m), P(flat))
    f = lf_frame(planes, w, h, None, mi_stride, mi_rows, mi_cols, hdr, bd, is16, plane_start, plane_end)
    assert ref.ref_loop_filter_frame(C.byref(f), C.byref(m), C.byref(hdr), sb_size) == 0
    return flat, np.frombuffer(bytes(f.lvl), np.uint8).copy()


GOLDEN_DLF = [  # key, w, h, bd, is16, header variant, sb, seed
    ("a_8bit", 200, 136, 8, 0, 1, 64, 21), ("b_10bit_seg", 136, 72, 10, 1, 2, 64, 22), ("c_8in16_sb128", 264, 136, 8, 1, 0, 128, 23)]


def golden_dlf_inputs(w, h, bd, is16, variant, sb, seed):
    rng = np.random.default_rng(seed)
    mi_cols, mi_rows = (w + 7) // 8 * 2, (h + 7) // 8 * 2
    mi_stride = mi_cols + 1
    minfo = random_mode_info(rng, mi_rows, mi_cols, mi_stride, sb=sb)
    hdr = lf_header(rng, variant)
    planes = lf_planes(rng, mi_cols * 4, mi_rows * 4, bd, is16)
    return mi_cols, mi_rows, mi_stride, minfo, hdr, planes


class GoldHdr:
    """The header fields lf_frame() needs, rebuilt from a golden fixture's meta row."""

    def __init__(self, meta):
        self.filter_level = [int(meta[7]), int(meta[8])]
        self.filter_level_u, self.filter_level_v, self.sharpness_level = int(meta[9]), int(meta[10]), int(meta[11])


def golden_dlf_cases():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dlf.npz"))
    for key in sorted(k[:-5] for k in g.files if k.endswith("_meta")):
        meta = g[key + "_meta"]
        w, h, bd, is16, mi_cols, mi_rows, mi_stride = (int(v) for v in meta[:7])
        flat = np.ascontiguousarray(g[key + "_mi"]).view(abi.LF_MI_DTYPE).reshape(mi_rows, mi_stride)
        yield (key, w, h, bd, is16, mi_cols, mi_rows, mi_stride, flat, g[key + "_lvl"].copy(), GoldHdr(meta),
               [g[f"{key}_in{i}"].copy() for i in range(3)], [g[f"{key}_out{i}"] for i in range(3)])

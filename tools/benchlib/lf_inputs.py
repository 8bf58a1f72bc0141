"""Synthetic device-resident inputs and launch helpers for `bench.py --workload lf` (in-loop filters, 4K 10-bit)."""
import ctypes as C

import numpy as np

from svtav1_hip import abi

PAD = 32


def _plane(rng, w, h, bd, torch, dev):
    yy, xx = np.mgrid[0:h + 2 * PAD, 0:w + 2 * PAD]
    img = (1 << bd) * (0.5 + 0.25 * np.sin(xx / 23.0) * np.cos(yy / 31.0))
    img += (1 << (bd - 8)) * rng.integers(-6, 7, size=((h + 2 * PAD) // 8 + 1, (w + 2 * PAD) // 8 + 1)).repeat(8, 0).repeat(8, 1)[:h + 2 * PAD, :w + 2 * PAD]
    a = np.clip(np.rint(img), 0, (1 << bd) - 1).astype(np.uint16)
    return torch.from_numpy(a.view(np.int16)).to(dev)


def build(lib, dev, rng, W, H, bd, torch):
    inp = {"W": W, "H": H, "bd": bd, "keep": []}
    planes, srcs, outs = [], [], []
    for pl in range(3):
        w, h = W >> (pl > 0), H >> (pl > 0)
        planes.append(_plane(rng, w, h, bd, torch, dev))
        srcs.append(_plane(rng, w, h, bd, torch, dev))
        outs.append(torch.zeros_like(planes[-1]))
    inp["planes"], inp["srcs"], inp["outs"] = planes, srcs, outs

    def ptr(t, w):
        return t.data_ptr() + (PAD * (w + 2 * PAD) + PAD) * 2
    inp["ptr"] = ptr
    # ---- deblocking: random partition into 8..64 squares, 8-byte records
    mi_cols, mi_rows = W // 4, H // 4
    mi = np.zeros((mi_rows, mi_cols), abi.LF_MI_DTYPE)
    bs_enum = {8: 3, 16: 6, 32: 9, 64: 12}
    tx_enum = {4: 0, 8: 1, 16: 2, 32: 3, 64: 4}
    for y in range(0, H, 64):
        for x in range(0, W, 64):
            s = int(rng.choice([8, 16, 32, 64], p=[0.2, 0.35, 0.3, 0.15]))
            for yy in range(y, min(y + 64, H), s):
                for xx in range(x, min(x + 64, W), s):
                    blk = mi[yy // 4:(yy + s) // 4, xx // 4:(xx + s) // 4]
                    blk["bsize"], blk["tx_size_y"], blk["tx_size_uv"] = bs_enum[s], tx_enum[min(s, 64) >> int(rng.integers(0, 2))], tx_enum[max(4, min(s // 2, 32))]
                    blk["skip_inter"], blk["ref_frame0"], blk["mode_lf"] = int(rng.random() < 0.4), int(rng.integers(0, 8)), int(rng.integers(0, 2))
    inp["d_mi"] = torch.from_numpy(mi.view(np.uint8).reshape(-1).copy()).to(dev)
    f = abi.LfFrame()
    for i in range(3):
        w = W >> (i > 0)
        f.plane[i], f.stride[i] = ptr(planes[i], w), w + 2 * PAD
    f.width, f.height, f.mi, f.mi_stride, f.mi_rows, f.mi_cols = W, H, inp["d_mi"].data_ptr(), mi_cols, mi_rows, mi_cols
    lvl = rng.integers(8, 48, size=768).astype(np.uint8)
    C.memmove(f.lvl, lvl.ctypes.data, 768)
    f.filter_level[0] = f.filter_level[1] = 24
    f.filter_level_u = f.filter_level_v = 16
    f.sharpness_level, f.bit_depth, f.is_16bit, f.plane_start, f.plane_end = 0, bd, 1, 0, 3
    inp["lf_frame"] = f
    # ---- CDEF
    w8, h8 = W // 8, (H + 7) // 8
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    inp["n_fb"] = nhfb * nvfb
    filt = (rng.random((h8, w8)) < 0.8).astype(np.uint8)
    inp["d_filt"] = torch.from_numpy(filt).to(dev)
    prm = abi.CdefSearchParams()
    strengths = [0, 4, 8, 17, 25, 34, 44, 63]
    prm.n_strengths = len(strengths)
    for i, s in enumerate(strengths):
        prm.strengths[i] = s
    prm.pri_damping = prm.sec_damping = 5
    prm.coeff_shift, prm.subsampling_factor = bd - 8, 1
    inp["cdef_prm"] = prm
    inp["d_mse"] = [torch.zeros(inp["n_fb"] * 8 * 8, dtype=torch.uint8, device=dev) for _ in range(3)]
    inp["d_dir"] = torch.zeros(inp["n_fb"] * 64, dtype=torch.uint8, device=dev)
    inp["d_var"] = torch.zeros(inp["n_fb"] * 64 * 4, dtype=torch.uint8, device=dev)
    inp["d_fbs"] = torch.from_numpy(rng.choice(np.array(strengths[1:], np.uint8), size=inp["n_fb"]).astype(np.uint8)).to(dev)
    # CDEF works on the 8-aligned picture: 2160 -> 2160 (already a multiple of 8)
    inp["cdef_planes_search"], inp["cdef_planes_apply"] = [], []
    for pli in range(3):
        w, h = W >> (pli > 0), H >> (pli > 0)
        dec = int(pli > 0)
        inp["cdef_planes_search"].append(abi.CdefPlane(ptr(planes[pli], w), ptr(srcs[pli], w), w + 2 * PAD, w + 2 * PAD, w, h, 1, dec, dec, pli))
        inp["cdef_planes_apply"].append(abi.CdefPlane(ptr(planes[pli], w), ptr(outs[pli], w), w + 2 * PAD, w + 2 * PAD, w, h, 1, dec, dec, pli))
    # ---- self-guided: luma, 256x256 restoration units (the last row / column of units absorbs the remainder)
    units = []
    for y in range(0, H, 256):
        for x in range(0, W, 256):
            uw, uh = min(256, W - x), min(256, H - y)
            off = (y * (W + 2 * PAD) + x) * 2
            units.append((abi.SgrUnit(ptr(planes[0], W) + off, ptr(srcs[0], W) + off, W + 2 * PAD, W + 2 * PAD, uw, uh, 1, bd, 64, 64), y, x))
    inp["sgr_units"] = units
    inp["sgr_plane"] = abi.SgrUnit(ptr(planes[0], W), ptr(srcs[0], W), W + 2 * PAD, W + 2 * PAD, W, H, 1, bd, 64, 64)
    inp["d_flt0"] = torch.zeros(H * W, dtype=torch.int32, device=dev)
    inp["d_flt1"] = torch.zeros(H * W, dtype=torch.int32, device=dev)
    inp["xqd"] = (C.c_int32 * 2)(-20, 40)
    # ---- Wiener: every 256x256 luma restoration unit of the picture (the last row / column absorbs the remainder)
    lim = [(x, W if W - x < 384 else x + 256, y, H if H - y < 384 else y + 256) for y in range(0, H - 127, 256) for x in range(0, W - 127, 256)]
    wu = (abi.WienerUnit * len(lim))()
    for i, (hs, he, vs, ve) in enumerate(lim):
        wu[i] = abi.WienerUnit(ptr(planes[0], W), ptr(srcs[0], W), W + 2 * PAD, W + 2 * PAD, hs, he, vs, ve)
    inp["wiener_units"], inp["n_wiener"] = wu, len(lim)
    inp["d_wM"] = torch.zeros(len(lim) * 49, dtype=torch.int64, device=dev)
    inp["d_wH"] = torch.zeros(len(lim) * 49 * 49, dtype=torch.int64, device=dev)
    inp["wfx"] = (C.c_int16 * 8)(3, -12, 30, 86 - 128, 30, -12, 3, 0)
    inp["wfy"] = (C.c_int16 * 8)(2, -9, 25, 92 - 128, 25, -9, 2, 0)
    return inp


def _chk(lib, rc):
    assert rc == 0, lib.svt_hip_last_error().decode()


def run_deblock(lib, inp, sp):
    _chk(lib, lib.svt_hip_loop_filter_frame(C.byref(inp["lf_frame"]), sp))


def run_cdef_search(lib, inp, sp):
    for pli in range(3):
        _chk(lib, lib.svt_hip_cdef_search_plane(C.byref(inp["cdef_planes_search"][pli]), C.c_void_p(inp["d_filt"].data_ptr()),
                                                C.byref(inp["cdef_prm"]), C.c_void_p(inp["d_mse"][pli].data_ptr()),
                                                C.c_void_p(inp["d_dir"].data_ptr()), C.c_void_p(inp["d_var"].data_ptr()), sp))


def run_cdef_apply(lib, inp, sp):
    """all three planes in one launch (svt_hip_cdef_apply_frame)"""
    if "cdef_apply_frame" not in inp:
        inp["cdef_apply_frame"] = ((abi.CdefPlane * 3)(*inp["cdef_planes_apply"]), (C.c_void_p * 3)(*[inp["d_fbs"].data_ptr()] * 3))
    planes, strengths = inp["cdef_apply_frame"]
    _chk(lib, lib.svt_hip_cdef_apply_frame(planes, C.c_uint32(3), C.c_void_p(inp["d_filt"].data_ptr()), strengths, 5, inp["bd"] - 8,
                                           C.c_void_p(inp["d_dir"].data_ptr()), C.c_void_p(inp["d_var"].data_ptr()), sp))


def run_sgr_filter(lib, inp, sp):
    """whole luma plane in one launch (restoration units are multiples of the 64x64 processing unit)"""
    _chk(lib, lib.svt_hip_sgr_filter_unit(C.byref(inp["sgr_plane"]), 3, C.c_void_p(inp["d_flt0"].data_ptr()), C.c_void_p(inp["d_flt1"].data_ptr()),
                                          inp["W"], sp))


def run_sgr_apply(lib, inp, sp):
    W = inp["W"]
    out = inp["outs"][0]
    off = (PAD * (W + 2 * PAD) + PAD) * 2
    _chk(lib, lib.svt_hip_sgr_apply_unit(C.byref(inp["sgr_plane"]), 3, inp["xqd"], C.c_void_p(out.data_ptr() + off), W + 2 * PAD, sp))


def run_wiener_stats(lib, inp, sp):
    _chk(lib, lib.svt_hip_wiener_stats(inp["wiener_units"], inp["n_wiener"], 7, 1, inp["bd"], C.c_void_p(inp["d_wM"].data_ptr()),
                                       C.c_void_p(inp["d_wH"].data_ptr()), sp))


def run_wiener_convolve(lib, inp, sp):
    W, H = inp["W"], inp["H"]
    out = inp["outs"][0]
    off = (PAD * (W + 2 * PAD) + PAD) * 2
    _chk(lib, lib.svt_hip_wiener_convolve(C.c_void_p(inp["ptr"](inp["planes"][0], W)), W + 2 * PAD, C.c_void_p(out.data_ptr() + off),
                                          W + 2 * PAD, W, H, inp["wfx"], inp["wfy"], 1, inp["bd"], sp))


def run_tf_noise(lib, inp, sp):
    """The temporal filter's noise estimate over the luma plane (svt_hip_tf_estimate_noise)."""
    W, H = inp["W"], inp["H"]
    if "d_noise" not in inp:
        import torch
        inp["d_noise"] = torch.zeros(24, dtype=torch.uint8, device=inp["planes"][0].device)
    _chk(lib, lib.svt_hip_tf_estimate_noise(C.c_void_p(inp["ptr"](inp["planes"][0], W)), C.c_uint32(W), C.c_uint32(H), C.c_uint32(W + 2 * PAD), 1,
                                            inp["bd"], C.c_void_p(inp["d_noise"].data_ptr()), sp))

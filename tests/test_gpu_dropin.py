"""GPU: the drop-in itself.  The Tier A functions are installed INTO THE REAL REFERENCE's RTCD pointers
(oracle/_ref/libsvtref.so, built from the reference sources by oracle/Makefile) with svt_hip_install_rtcd, and the
reference's own drivers — svt_aom_motion_estimation_b64, svt_av1_loop_filter_frame, svt_cdef_filter_fb,
svt_aom_estimate_transform, svt_aom_inv_transform_recon — are run twice: with their C leaves and with the HIP leaves.
Outputs must be identical.  This is the test the reference's maintainers would run after applying INTEGRATION.md."""
import ctypes as C

import numpy as np
import pytest

import lf_cases as L
import me_cases as M
import tx_cases as T
from lf_cases import BS, HB, P, V, VB, VL
from svtav1_hip import abi
from test_abi import declared_symbols

pytestmark = pytest.mark.gpu


class Binding(C.Structure):
    _fields_ = [("name", C.c_char_p), ("slot", C.c_void_p)]


ALIAS = {"svt_aom_downsample_2d": "downsample_2d", "svt_aom_sad_16b_kernel": "sad_16b_kernel"}     # export stem -> the reference's pointer variable (aom_dsp_rtcd.h:838, :861)


def tier_a_pointer_names():
    stems = [s[:-4] for s in declared_symbols() if s.endswith("_hip") and not s.startswith("svt_hip_")]
    return [ALIAS.get(n, n) for n in stems]


class DropIn:
    def __init__(self, hip, ref):
        self.hip, self.ref = hip, ref
        self.names = []
        for n in tier_a_pointer_names():
            try:
                C.c_void_p.in_dll(ref, n)
                self.names.append(n)
            except ValueError:
                pass                                    # not an RTCD pointer of the reference (none expected)
        self.saved = {n: C.c_void_p.in_dll(ref, n).value for n in self.names}

    def install(self):
        tab = (Binding * len(self.names))()
        for i, n in enumerate(self.names):
            tab[i].name, tab[i].slot = n.encode(), C.addressof(C.c_void_p.in_dll(self.ref, n))
        done = C.c_uint32(0)
        rc = self.hip.svt_hip_install_rtcd(tab, len(self.names), C.byref(done))
        assert rc == 0 and done.value == len(self.names)
        back = {v: k for k, v in ALIAS.items()}
        for n in self.names:
            assert C.c_void_p.in_dll(self.ref, n).value == C.cast(getattr(self.hip, back.get(n, n) + "_hip"), C.c_void_p).value

    def restore(self):
        for n, v in self.saved.items():
            C.c_void_p.in_dll(self.ref, n).value = v


@pytest.fixture()
def dropin(hip, ref):
    d = DropIn(hip, ref)
    assert hip.svt_hip_debug_tier_a_broken(0) == 0, "the fail-over latch was already set when the test started"
    yield d
    d.restore()
    # A HIP error inside a leaf puts the C pointers back and finishes the call with them (common.hpp TIER_A_CALL): the "hip"
    # phase of a test would then compare C with C and pass.  Every test that installed the leaves must leave the latch clear
    # (test_device_failure_restores_cpu_kernels un-latches it itself before it returns).
    assert hip.svt_hip_debug_tier_a_broken(0) == 0, "a Tier A leaf failed over to the CPU function during this test"


def test_every_tier_a_export_has_a_reference_pointer(dropin):
    missing = [n for n in tier_a_pointer_names() if n not in dropin.names]
    assert not missing, f"exports without a matching RTCD pointer in the reference: {missing}"
    assert len(dropin.names) >= 120


def test_motion_estimation_b64_with_hip_leaves(dropin, ref, orc):
    W, H = 192, 128
    clip = M.make_clip("blocks", W, H, 5, seed=9)
    pyrs = M.build_pyramids(orc, clip)
    results = []
    for phase in ("c", "hip"):
        if phase == "hip":
            dropin.install()
        out = []
        for key in ("m8_360p_tl2", "m4_360p_tl0"):
            prm = M.scenario_params(key, 2, [1, 0], [3, 4])
            out.append(M.run_cpu(ref.ref_me_frame, prm, pyrs, 2, [1, 0], [3, 4], W, H))
        results.append(out)
    dropin.restore()
    for a, b in zip(*results):
        M.assert_same(a, b, "reference ME driver: C leaves vs HIP leaves")


def test_loop_filter_frame_with_hip_leaves(dropin, ref):
    rng = np.random.default_rng(77)
    got = []
    for phase in ("c", "hip"):
        if phase == "hip":
            dropin.install()
        out = []
        for variant, (w, h) in ((0, (136, 72)), (1, (72, 136))):
            r = np.random.default_rng(500 + variant)
            bd, is16 = ((8, 0), (10, 1))[variant]
            mi_cols, mi_rows = w // 4, h // 4
            minfo = L.random_mode_info(r, mi_rows, mi_cols, mi_cols + 1)
            hdr = L.lf_header(r, variant)
            planes = L.lf_planes(r, w, h, bd, is16)
            L.ref_deblock(ref, planes, w, h, minfo, mi_cols + 1, mi_rows, mi_cols, hdr, bd, is16)
            out.append(planes)
        got.append(out)
    dropin.restore()
    for a, b in zip(*got):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_cdef_filter_fb_with_hip_leaves(dropin, ref):
    got = []
    for phase in ("c", "hip"):
        if phase == "hip":
            dropin.install()
        rng = np.random.default_rng(5)
        out = []
        for is16, bd, pli in ((0, 8, 0), (1, 10, 0), (1, 10, 1)):
            tile = L.cdef_tile(rng, bd, edge=3)
            tile[VB:VB + 64, HB:HB + 64] = L.smooth_plane(rng, 64, 64, bd).astype(np.uint16)
            dl = (abi.CdefList * 64)()
            n = 0
            for r in range(8):
                for c in range(8):
                    if (r * 3 + c) % 4:
                        dl[n].by, dl[n].bx = r, c
                        n += 1
            d16, v16 = np.zeros((16, 16), np.uint8), np.zeros((16, 16), np.int32)
            dirinit = C.c_int32(0)
            dst = np.zeros((64, 64), np.uint16 if is16 else np.uint8)
            xdec = int(pli > 0)
            if pli:                              # chroma needs luma directions: take them from a luma call
                tmp = np.zeros(64 * 64, np.uint16)
                ref.svt_cdef_filter_fb(None, P(tmp), 0, V(tile.ctypes.data + 2 * (VB * BS + HB)), 0, 0, P(d16), C.byref(dirinit), P(v16), 0,
                                       C.byref(dl), n, 5, 2, 5, 5, bd - 8, C.c_uint8(1))
            ref.svt_cdef_filter_fb(None if is16 else P(dst), P(dst) if is16 else None, 64, V(tile.ctypes.data + 2 * (VB * BS + HB)), xdec,
                                   xdec, P(d16), C.byref(dirinit), P(v16), pli, C.byref(dl), n, 9, 4, 5, 5, bd - 8, C.c_uint8(1))
            out.append((dst, d16.copy(), v16.copy()))
        got.append(out)
    dropin.restore()
    for a, b in zip(*got):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_transform_drivers_with_hip_leaves(dropin, ref, orc):
    """svt_aom_estimate_transform (transforms.c:3040-3156) and svt_aom_inv_transform_recon (inv_transforms.c:3147-3175)."""
    TXS = {(4, 4): 0, (8, 8): 1, (16, 16): 2, (32, 32): 3, (64, 64): 4, (4, 8): 5, (8, 4): 6, (8, 16): 7, (16, 8): 8, (16, 32): 9,
           (32, 16): 10, (32, 64): 11, (64, 32): 12, (4, 16): 13, (16, 4): 14, (8, 32): 15, (32, 8): 16, (16, 64): 17, (64, 16): 18}
    got = []
    for phase in ("c", "hip"):
        if phase == "hip":
            dropin.install()
        rng = np.random.default_rng(11)
        out = []
        for (w, h), txs in TXS.items():
            for tt in (0, 1, 9, 10):
                if not orc.orc_txfm_valid(w, h, tt) or (max(w, h) == 64 and tt != 0) or (max(w, h) == 32 and tt not in (0, 9)):
                    # the production dispatcher asserts on anything else (inv_transforms.c:2870-2880)
                    continue
                for bd, shape in ((8, 0), (10, 1), (10, 2)):
                    res = T.residual(rng, w, h, bd, 1)
                    coeff = np.zeros(w * h, np.int32)
                    energy = C.c_uint64(0)
                    rc = ref.svt_aom_estimate_transform(P(res), w + 3, P(coeff), w, txs, C.byref(energy), bd, tt, 0, shape)
                    assert rc == 0
                    pred = rng.integers(0, 1 << bd, size=(h, w + 5)).astype(np.uint16)
                    rec = np.zeros((h, w + 7), np.uint16)
                    eob = min(w, 32) * min(h, 32)
                    rc = ref.svt_aom_inv_transform_recon(P(coeff), V(pred.ctypes.data >> 1), w + 5, V(rec.ctypes.data >> 1), w + 7, txs,
                                                         bd, tt, 0, eob, 0)
                    assert rc == 0
                    out.append((coeff, energy.value, rec))
        got.append(out)
    dropin.restore()
    assert len(got[0]) > 100
    for a, b in zip(*got):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])


def test_sgr_search_with_hip_leaves(dropin, ref):
    """search_selfguided_restoration (restoration_pick.c:550-652, reached through oracle/ref_harness_sgr.c) with its five
    RTCD leaves — filter, apply, both projection errors, subspace — replaced by the HIP functions."""
    import sgr_cases as G
    got = []
    for phase in ("c", "hip"):
        if phase == "hip":
            dropin.install()
        out = []
        for bd, is16, (w, h), pu in ((8, 0, (96, 80), 64), (10, 1, (72, 40), 32)):
            rng = np.random.default_rng(600 + bd)
            dat, src = G.sgr_plane(rng, w, h, bd, is16, 0)
            e = (lambda a: V(a >> 1)) if is16 else V
            o = np.zeros(3, np.int32)
            assert ref.ref_sgr_search_unit(e(G.at(dat)), w, h, dat.shape[1], e(G.at(src)), src.shape[1], is16, bd, pu, pu, 0, 16, 3, 1, P(o)) == 0
            out.append(o)
        got.append(out)
    dropin.restore()
    for a, b in zip(*got):
        assert np.array_equal(a, b)


def test_tpl_block_cost_with_hip_leaves(dropin, ref):
    """The TPL dispenser's transform-domain cost (src_ops_process.c:734-748): svt_aom_subtract_block ->
    svt_av1_wht_fwd_txfm (the reference's own dispatcher, which lands in the forward-transform pointers) -> svt_aom_satd,
    first with the C leaves, then with the HIP ones installed."""
    import test_residual_oracle as TR
    got = []
    for phase in ("c", "hip"):
        if phase == "hip":
            dropin.install()
        got.append([TR.ref_tpl_cost(ref, *c) for c in TR.tpl_cases()])
    dropin.restore()
    assert len(got[0]) == 54 and got[0] == got[1]


def test_device_failure_restores_cpu_kernels(dropin, ref, hip):
    """SURVEY 8b "never abort": a HIP failure inside a Tier A leaf (injected through the library's test hook) must (1) complete
    that very call with the encoder's own CPU function, (2) put every RTCD pointer back to what svt_hip_install_rtcd had
    replaced, (3) refuse to be installed again."""
    sig = C.CFUNCTYPE(None, M.u8p if hasattr(M, "u8p") else C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32,
                      C.POINTER(C.c_uint64), C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.c_uint32, C.c_uint8, C.c_int16, C.c_int16)
    cases = list(M.iter_sad_loop_cases())[:6]
    want = [M.call_sad_loop(ref.svt_sad_loop_kernel_c, prm, src, refw) for prm, src, refw in cases]
    dropin.install()
    try:
        slot = C.c_void_p.in_dll(ref, "svt_sad_loop_kernel")
        assert slot.value != dropin.saved["svt_sad_loop_kernel"]
        got = [M.call_sad_loop(sig(slot.value), *cases[0])]               # healthy: the HIP leaf
        hip.svt_hip_debug_inject_failure(0)                               # the next device check inside a leaf fails
        got.append(M.call_sad_loop(sig(slot.value), *cases[1]))           # enters the HIP leaf, which fails and falls back
        assert hip.svt_hip_debug_tier_a_broken(0) == 1
        for n, v in dropin.saved.items():                                 # every pointer is the CPU function again
            assert C.c_void_p.in_dll(ref, n).value == v, n
        got += [M.call_sad_loop(sig(C.c_void_p.in_dll(ref, "svt_sad_loop_kernel").value), *c) for c in cases[2:]]
        assert got == want
        tab = (Binding * 1)()
        tab[0].name, tab[0].slot = b"svt_sad_loop_kernel", C.addressof(C.c_void_p.in_dll(ref, "svt_sad_loop_kernel"))
        assert hip.svt_hip_install_rtcd(tab, 1, None) == abi.SVT_HIP_ERR_RUNTIME
        assert b"disabled" in hip.svt_hip_last_error()
    finally:
        hip.svt_hip_debug_inject_failure(-1)
        hip.svt_hip_debug_tier_a_broken(1)                                # un-latch for the tests that follow
        dropin.restore()

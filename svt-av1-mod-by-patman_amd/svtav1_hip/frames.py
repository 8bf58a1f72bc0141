"""Host-side picture containers and the synthetic clip generator (SURVEY.md §8d recipe).

Padded 8-bit luma planes laid out like the reference's EbPictureBufferDesc
(Source/Lib/Codec/pic_buffer_desc.h:34-75): buffer = (height + 2*org_y) rows of `stride` bytes,
sample (x, y) at buf[(org_y + y) * stride + org_x + x].  Padding sizes follow
Source/Lib/Globals/enc_handle.c:1276,1292,1308 (64+4 full, 32 quarter, 16 sixteenth).
"""
import ctypes as C

import numpy as np

from . import abi

FULL_PAD, QUARTER_PAD, SIXTEENTH_PAD = 68, 32, 16


class HostPlane:
    """A padded u8 plane in host memory (numpy) + its Plane8 descriptor."""

    def __init__(self, width, height, pad, data=None):
        self.width, self.height, self.pad = int(width), int(height), int(pad)
        self.stride = self.width + 2 * self.pad
        self.rows = self.height + 2 * self.pad
        self.buf = np.zeros((self.rows, self.stride), dtype=np.uint8)
        if data is not None:
            self.interior[...] = data

    @property
    def interior(self):
        p = self.pad
        return self.buf[p:p + self.height, p:p + self.width]

    def pad_edges(self):
        """Edge replication over the whole padding (what svt_aom_generate_padding produces)."""
        p, h, w = self.pad, self.height, self.width
        self.buf[p:p + h, :p] = self.buf[p:p + h, p:p + 1]
        self.buf[p:p + h, p + w:] = self.buf[p:p + h, p + w - 1:p + w]
        self.buf[:p, :] = self.buf[p:p + 1, :]
        self.buf[p + h:, :] = self.buf[p + h - 1:p + h, :]

    def desc(self, base_ptr=None):
        ptr = self.buf.ctypes.data if base_ptr is None else base_ptr
        return abi.Plane8(ptr, self.stride, self.pad, self.pad, self.width, self.height)

    @property
    def nbytes(self):
        return self.buf.size


class HostPyramid:
    """full + 1/4 + 1/16 planes of one picture (input_padded_pic and the two decimations)."""

    def __init__(self, luma):
        h, w = luma.shape
        self.full = HostPlane(w, h, FULL_PAD, luma)
        self.full.pad_edges()
        self.quarter = HostPlane(w >> 1, h >> 1, QUARTER_PAD)
        self.sixteenth = HostPlane(w >> 2, h >> 2, SIXTEENTH_PAD)

    def desc(self):
        return abi.Pyramid8(self.full.desc(), self.quarter.desc(), self.sixteenth.desc())

    def planes(self):
        return (self.full, self.quarter, self.sixteenth)


def synthetic_clip(width, height, n_frames, seed=7):
    """Panning low-pass noise + N(0,2) sensor noise, 8-bit luma."""
    rng = np.random.default_rng(seed)
    margin = 16 + 2 * n_frames + 16
    bh, bw = (height + margin + 7) // 8 + 2, (width + 2 * margin + 7) // 8 + 2
    base = np.kron(rng.integers(0, 256, size=(bh, bw)).astype(np.float32), np.ones((8, 8), np.float32))
    for _ in range(3):
        base = (base + np.roll(base, 1, 0) + np.roll(base, -1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 1)) / 5
    frames = []
    for i in range(n_frames):
        oy, ox = 10 + i, 10 + 2 * i
        f = base[oy:oy + height, ox:ox + width] + rng.normal(0.0, 2.0, size=(height, width)).astype(np.float32)
        frames.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
    return frames


def me_out_shapes(prm, n_b64):
    """name -> (numpy dtype, shape) of every SvtHipMeFrameOut array for one picture."""
    sp = prm.stored_pus()
    return {
        "best_sad": (np.uint32, (n_b64, 2, 4, 85)),
        "best_mv": (np.uint32, (n_b64, 2, 4, 85)),
        "search_results": (np.uint8, (n_b64, 2, 4, C.sizeof(abi.MeSearchResult))),
        "me_mv_array": (np.uint32, (n_b64, sp * prm.max_refs)),
        "me_candidate_array": (np.uint8, (n_b64, sp * prm.max_cand)),
        "total_me_candidate_index": (np.uint8, (n_b64, sp)),
        "me_64x64_distortion": (np.uint32, (n_b64,)),
        "me_32x32_distortion": (np.uint32, (n_b64,)),
        "me_16x16_distortion": (np.uint32, (n_b64,)),
        "me_8x8_distortion": (np.uint32, (n_b64,)),
        "me_8x8_cost_variance": (np.uint32, (n_b64,)),
        "rc_me_distortion": (np.uint32, (n_b64,)),
    }


def b64_count(width, height):
    aw, ah = (width + 7) & ~7, (height + 7) & ~7
    return ((aw + 63) // 64) * ((ah + 63) // 64)


def alloc_me_out_host(prm, n_b64, fill=0xA5):
    """Host output arrays pre-filled (entries the reference never writes keep the fill)."""
    arrs = {}
    for name, (dt, shape) in me_out_shapes(prm, n_b64).items():
        a = np.empty(shape, dtype=dt)
        a.view(np.uint8)[...] = fill
        arrs[name] = a
    out = abi.MeFrameOut(**{k: v.ctypes.data for k, v in arrs.items()})
    return arrs, out


def set_refs(prm, cur_poc, ref_pocs_l0, ref_pocs_l1):
    """Fill the per-picture list/ref fields of MeParams like me_process.c:218-227 + pcs.c:91-96."""
    n0, n1 = len(ref_pocs_l0), len(ref_pocs_l1)
    prm.num_of_list_to_search = 2 if n1 else 1
    prm.num_of_ref_pic_to_search[0], prm.num_of_ref_pic_to_search[1] = n0, n1
    prm.picture_number = cur_poc
    for r, poc in enumerate(ref_pocs_l0):
        prm.ref_picture_number[0][r] = poc
    for r, poc in enumerate(ref_pocs_l1):
        prm.ref_picture_number[1][r] = poc
    prm.max_refs = n0 + n1
    prm.max_l0 = n0
    prm.max_cand = n0 + n1 + n0 * n1 + (n0 - 1) + (1 if n1 == 3 else 0)
    return prm

"""Device-side plumbing over the C-ABI (no compute here): device buffers, picture upload, job building.

Everything goes through libsvtav1_hip's own allocator / copy entry points (include/svt_hip.h), so this
module works without torch; bench.py may instead hand in torch-owned device pointers.
"""
import ctypes as C

import numpy as np

from . import abi, frames


class HipError(RuntimeError):
    pass


def check(lib, rc, what):
    if rc != 0:
        raise HipError(f"{what} failed (0x{rc & 0xffffffff:08x}): {lib.svt_hip_last_error().decode()}")


class DeviceBuffer:
    def __init__(self, lib, nbytes):
        self.lib, self.nbytes = lib, int(nbytes)
        p = C.c_void_p()
        check(lib, lib.svt_hip_malloc(C.byref(p), C.c_size_t(self.nbytes)), "svt_hip_malloc")
        self.ptr = p.value

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.lib, self.lib.svt_hip_upload(C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p),
                                                C.c_size_t(arr.nbytes), C.c_void_p(stream)), "svt_hip_upload")
        check(self.lib, self.lib.svt_hip_stream_sync(C.c_void_p(stream)), "svt_hip_stream_sync")

    def download(self, dtype, shape, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.lib, self.lib.svt_hip_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                                  C.c_size_t(out.nbytes), C.c_void_p(stream)), "svt_hip_download")
        check(self.lib, self.lib.svt_hip_stream_sync(C.c_void_p(stream)), "svt_hip_stream_sync")
        return out

    def fill(self, value, stream=None):
        check(self.lib, self.lib.svt_hip_memset(C.c_void_p(self.ptr), int(value), C.c_size_t(self.nbytes),
                                                C.c_void_p(stream)), "svt_hip_memset")

    def free(self):
        if self.ptr:
            self.lib.svt_hip_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DevicePlane:
    """Device mirror of a HostPlane (same geometry)."""

    def __init__(self, lib, host_plane, upload=True):
        self.lib, self.h = lib, host_plane
        self.buf = DeviceBuffer(lib, host_plane.nbytes)
        if upload:
            self.buf.upload(host_plane.buf)

    def desc(self):
        return self.h.desc(self.buf.ptr)

    def download(self):
        return self.buf.download(np.uint8, self.h.buf.shape)


class DevicePyramid:
    def __init__(self, lib, host_pyr, upload=True):
        self.full = DevicePlane(lib, host_pyr.full, upload)
        self.quarter = DevicePlane(lib, host_pyr.quarter, upload)
        self.sixteenth = DevicePlane(lib, host_pyr.sixteenth, upload)

    def desc(self):
        return abi.Pyramid8(self.full.desc(), self.quarter.desc(), self.sixteenth.desc())


class DeviceMeOut:
    """Device output arrays of one picture, pre-filled with `fill` (as frames.alloc_me_out_host)."""

    def __init__(self, lib, prm, n_b64, fill=0xA5):
        self.lib, self.shapes, self.bufs = lib, frames.me_out_shapes(prm, n_b64), {}
        for name, (dt, shape) in self.shapes.items():
            b = DeviceBuffer(lib, int(np.prod(shape)) * np.dtype(dt).itemsize)
            b.fill(fill)
            self.bufs[name] = b

    def desc(self):
        return abi.MeFrameOut(**{k: v.ptr for k, v in self.bufs.items()})

    def download(self):
        return {k: self.bufs[k].download(dt, shape) for k, (dt, shape) in self.shapes.items()}


def me_frames(lib, jobs, stream=None, sync=True):
    arr = (abi.MeFrameJob * len(jobs))(*jobs)
    check(lib, lib.svt_hip_me_frames(arr, C.c_uint32(len(jobs)), C.c_void_p(stream)), "svt_hip_me_frames")
    if sync:
        check(lib, lib.svt_hip_stream_sync(C.c_void_p(stream)), "svt_hip_stream_sync")

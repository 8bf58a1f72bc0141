"""GPU (one device): the C entry points of the multi-GPU exchange step (include/svt_hip_shard.h) over RCCL with a
single-rank communicator — everything but the wire: librccl is found and bound, the communicator comes up, the broadcast is
ordered BEHIND the producer stream's kernels and the completion event orders a consumer stream behind it.  More than one rank
per GPU is refused by RCCL, so the N > 1 data movement itself is exercised only by `bench.py --gpus N` on a multi-GPU node
(torch.distributed transport) — DESIGN.md section 5."""
import ctypes as C

import numpy as np
import pytest

from svtav1_hip import abi, device

pytestmark = pytest.mark.gpu


def test_publish_reference_single_rank(hip):
    ident = (C.c_uint8 * 128)()
    device.check(hip, hip.svt_hip_comm_get_unique_id(ident), "comm_get_unique_id")
    comm = C.c_void_p()
    device.check(hip, hip.svt_hip_comm_create(ident, 1, 0, C.byref(comm)), "comm_create")
    try:
        nbytes = 27 * 1024 * 1024 + 12345              # about one padded 4K 10-bit 4:2:0 picture
        pic = device.DeviceBuffer(hip, nbytes)
        prod, side, cons = C.c_void_p(), C.c_void_p(), C.c_void_p()
        for s in (prod, side, cons):
            device.check(hip, hip.svt_hip_stream_create(C.byref(s)), "stream_create")
        done = C.c_void_p()
        out = device.DeviceBuffer(hip, nbytes)
        for round_, value in enumerate((0x5A, 0xC3)):
            # "in-loop filters" of the owner: fill the picture on the producer stream; publish right behind it
            device.check(hip, hip.svt_hip_memset(C.c_void_p(pic.ptr), value, C.c_size_t(nbytes), prod), "memset")
            device.check(hip, hip.svt_hip_publish_reference(C.c_void_p(pic.ptr), C.c_size_t(nbytes), 0, comm, prod, side, C.byref(done)), "publish")
            assert done.value
            # a consumer: waits for the publication, then copies the picture on ITS stream
            device.check(hip, hip.svt_hip_wait_reference(done, cons), "wait_reference")
            got = np.empty(nbytes, np.uint8)
            device.check(hip, hip.svt_hip_download(got.ctypes.data_as(C.c_void_p), C.c_void_p(pic.ptr), C.c_size_t(nbytes), cons), "download")
            device.check(hip, hip.svt_hip_stream_sync(cons), "sync")
            assert (got == value).all(), round_
        device.check(hip, hip.svt_hip_event_destroy(done), "event_destroy")
        for s in (prod, side, cons):
            device.check(hip, hip.svt_hip_stream_destroy(s), "stream_destroy")
    finally:
        device.check(hip, hip.svt_hip_comm_destroy(comm), "comm_destroy")


def test_publish_bad_arguments(hip):
    assert hip.svt_hip_publish_reference(None, C.c_size_t(10), 0, None, None, None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_comm_create(None, 1, 0, None) == abi.SVT_HIP_ERR_BAD_PARAMETER
    assert hip.svt_hip_wait_reference(None, None) == abi.SVT_HIP_ERR_BAD_PARAMETER

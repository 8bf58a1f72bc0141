"""CPU: the C-ABI shared library loads and exports every symbol the headers declare (no compute calls)."""
import ctypes as C
import glob
import os
import re

from svtav1_hip import abi


def declared_symbols():
    names = []
    for h in sorted(glob.glob(os.path.join(abi.REPO_ROOT, "include", "*.h"))):
        text = open(h).read()
        names += re.findall(r"SVT_HIP_API[^;(]*?\b(svt_\w+)\s*\(", text, flags=re.S)
    return names


def test_library_exports_every_declared_symbol():
    lib = abi.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported by libsvtav1_hip.so"


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (checked against the compiler through the oracle library)."""
    import pyorc
    orc = pyorc.oracle()
    assert orc.orc_sizeof_me_params() == C.sizeof(abi.MeParams)
    assert orc.orc_sizeof_me_job() == C.sizeof(abi.MeFrameJob)
    assert orc.orc_sizeof_plane() == C.sizeof(abi.Plane8)
    assert C.sizeof(abi.SadLoopDesc) == 48 and C.sizeof(abi.SadLoopResult) == 16
    assert C.sizeof(abi.MeSearchResult) == 16


def test_no_device_error_path():
    """Without a GPU the product path fails loudly instead of falling back to the CPU."""
    lib = abi.load()
    if lib.svt_hip_device_count() > 0:
        return
    assert lib.svt_hip_init(0) != 0
    assert b"no HIP device" in lib.svt_hip_last_error()

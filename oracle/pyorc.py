"""TEST INFRASTRUCTURE — ctypes access to the CPU restatement (oracle/libsvtoracle.so) and, when it has
been built in this container, to the real reference C path (oracle/_ref/libsvtref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(_HERE), "svt-av1-mod-by-patman_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from svtav1_hip import abi  # noqa: E402  (ABI structs only; no compute)

ORACLE_SO = os.path.join(_HERE, "libsvtoracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libsvtref.so")

_orc = None
_ref = None


def oracle():
    global _orc
    if _orc is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle oracle` (or __graft_entry__.build())")
        _orc = C.CDLL(ORACLE_SO)
        _orc.orc_nxm_sad.restype = C.c_uint32
        for n in ("orc_compute_sub_mean_8x8", "orc_compute_mean", "orc_compute_mean_squared_values"):
            getattr(_orc, n).restype = C.c_uint64
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The real reference functions (svt_*_c etc.) + our harness entry points (ref_*)."""
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError(f"{REF_SO} missing: `make -C oracle ref` needs /root/reference (this container)")
        _ref = C.CDLL(REF_SO)
        _ref.ref_init()
        _ref.svt_nxm_sad_kernel_helper_c.restype = C.c_uint32
        for n in ("svt_compute_sub_mean_8x8_c", "svt_compute_mean_c", "svt_compute_mean_squared_values_c"):
            getattr(_ref, n).restype = C.c_uint64
    return _ref

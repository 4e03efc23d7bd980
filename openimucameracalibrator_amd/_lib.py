"""Loader for the product library liboicc_hip.so (HIP kernels + C-ABI).

There is deliberately no fallback: if the shared object is missing or does not
export every symbol of include/oicc_hip.h this raises, and oicc_create itself
fails when no HIP device is usable.
"""
import ctypes
import os

from ._abi import Bound, BoundBa

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OICC_DEV_LIB") or os.path.join(_HERE, "csrc", "liboicc_hip.so")   # OICC_DEV_LIB: another BUILD of the same library (developer A/B timing, scripts/build_variant.sh)
_bound = None
_bound_ba = None


def load():
    global _bound
    if _bound is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liboicc_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C openimucameracalibrator_amd/csrc`; there is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _bound = Bound(lib, "oicc_", device=True)
    return _bound


def load_ba():
    """oicc_ba_* entry points (view bundle adjustment) of the same library."""
    global _bound_ba
    if _bound_ba is None:
        _bound_ba = BoundBa(load().lib, "oicc_ba_")
    return _bound_ba

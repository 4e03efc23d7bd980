"""Test-only binding of the CPU checker (oracle/liboicc_oracle.so).

The oracle exports the same C-ABI as include/oicc_hip.h under the prefix
``oicc_oracle_``; binding it with the product's ctypes table lets the tests drive
the oracle and the HIP library through the same SplineTrajectoryEstimator mirror.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
import ctypes
import os
import subprocess

from openimucameracalibrator_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboicc_oracle.so")
_bound = None


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oicc_oracle.cpp", "ba_oracle.cpp", "oicc_oracle_math.hpp", "Makefile")]
    if (not os.path.exists(ORACLE_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def load():
    global _bound
    if _bound is None:
        build()
        lib = ctypes.CDLL(ORACLE_LIB)
        _bound = _abi.Bound(lib, "oicc_oracle_", device=False)
        _bound.raw = lib
    return _bound


_bound_ba = None


def load_ba():
    """The oracle's view-bundle-adjustment entry points (oracle/ba_oracle.cpp)."""
    global _bound_ba
    if _bound_ba is None:
        b = load()
        _bound_ba = _abi.BoundBa(b.raw, "oicc_oracle_ba_")
        _bound_ba.raw = b.raw
    return _bound_ba

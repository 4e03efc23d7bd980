"""The branch-free sincos of the SO(3) spline (csrc/spline_math.h: fast_sincos -- two-piece pi/2 reduction, fdlibm kernel polynomials,
quadrant by selects; round 5) against libm in extended precision, on the host build of the same source (the analytic CPU path of
oracle/cpu_analytic.hpp compiles the product's item functions with OICC_HOST_MATH).  The kernels' results are held to the Jet oracle,
whose sines and cosines are libm's, by the GPU parity tests; this pins the function itself, argument range included, without a GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_backend


def _fast_sincos(x):
    raw = oracle_backend.load().raw
    raw.oicc_oracle_debug_fast_sincos.restype = None
    raw.oicc_oracle_debug_fast_sincos.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.empty_like(x); c = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    raw.oicc_oracle_debug_fast_sincos(x.ctypes.data_as(dp), x.size, s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    return s, c


@pytest.mark.parametrize("span", [1e-6, 0.78, 1.6, 3.2, 10.0, 100.0])
def test_fast_sincos_is_within_two_ulp_of_libm(span):
    rng = np.random.default_rng(int(span * 1000) + 1)
    x = rng.uniform(-span, span, 400000)
    s, c = _fast_sincos(x)
    xl = x.astype(np.longdouble)
    rs, rc = np.sin(xl), np.cos(xl)
    # error in units of the last place of the exact value (floored at the spacing near 1e-3: next to a zero of sin / cos the ABSOLUTE error is what the spline sees)
    for got, ref in ((s, rs), (c, rc)):
        ulp = np.spacing(np.maximum(np.abs(ref.astype(np.float64)), 1e-3))
        assert np.max(np.abs((got.astype(np.longdouble) - ref).astype(np.float64)) / ulp) < 2.0
    assert np.max(np.abs(s * s + c * c - 1.0)) < 1e-15


def test_fast_sincos_special_arguments():
    x = np.array([0.0, -0.0, np.pi / 4, -np.pi / 4, np.pi / 2, -np.pi / 2, np.pi, -np.pi, 3 * np.pi / 2, 2 * np.pi, 1e-300, 0.7853981633974484, 2.356194490192345])
    s, c = _fast_sincos(x)
    assert np.allclose(s, np.sin(x), rtol=0, atol=2.3e-16) and np.allclose(c, np.cos(x), rtol=0, atol=2.3e-16)
    assert s[0] == 0.0 and c[0] == 1.0 and s[10] == 1e-300

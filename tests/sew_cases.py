"""Seeded input signals of the spline-error-weighting tests: shared by the golden generator
(tests/golden/make_sew_golden.py, runs the REFERENCE python/sew.py in the build container) and
by tests/test_sew.py (oracle and HIP path)."""
import hashlib

import numpy as np

from openimucameracalibrator_amd import synthetic


def _handheld(n, rate, seed, dims=3):
    """Band-limited hand-held-like motion + white noise (odd length exercises the no-Nyquist case)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / rate
    sig = np.zeros((dims, n))
    for a in range(dims):
        for _ in range(6):
            f = rng.uniform(0.2, 6.0)
            sig[a] += rng.uniform(0.1, 1.0) / (1.0 + f) * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
        sig[a] += 0.02 * rng.standard_normal(n)
    return sig, t


def cases():
    """name -> (signal [dims, n], times [n], quality, min_dt, max_dt)."""
    out = {}
    for cfg in ("C1", "C2"):
        ds = synthetic.make_config(cfg)
        # exactly what get_sew_for_dataset.py:38-39 passes
        out[cfg + "_accel_r3"] = (ds.accel.T.copy(), ds.imu_t_s.copy(), 0.96, 0.01, 0.15)
        out[cfg + "_gyro_so3"] = (ds.gyro.T.copy(), ds.imu_t_s.copy(), 0.98, 0.01, 0.2)
    s, t = _handheld(4001, 200.0, 7)
    out["handheld_odd_defaults"] = (s, t, 0.97, None, None)          # default min/max dt (sew.py:153-157)
    s, t = _handheld(6000, 400.0, 11)
    out["handheld_endpoint"] = (s, t, 0.5, 0.005, 0.02)              # quality reached at max_dt: end-point return (sew.py:95-99)
    s, t = _handheld(3000, 100.0, 13, dims=1)
    out["single_axis_unreachable"] = (s[0], t, 0.999999, 0.05, 0.4)  # no dt satisfies: best dt of the back-off (sew.py:133-137)
    return out


def checksum(sig, t):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(np.round(np.atleast_2d(sig), 9)).tobytes()); h.update(np.ascontiguousarray(np.round(t, 9)).tobytes())
    return h.hexdigest()[:16]

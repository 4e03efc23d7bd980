"""ORACLE -- TEST INFRASTRUCTURE ONLY (only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this).

numpy restatement of the reference's spline-error-weighting pre-stage, python/sew.py
(knot_spacing_and_variance :199-235 and what it calls).  PINNED: tests/test_sew.py checks it
against golden outputs of the reference module itself (tests/golden/sew_golden.json, generated
in the build container by tests/golden/make_sew_golden.py which imports /root/reference/python/sew.py).
scipy.optimize.brentq [EXT, SciPy] is restated from its published algorithm (Brent's method as in
scipy/optimize/Zeros/brentq.c: xtol 2e-12, rtol 4*eps, 100 iterations).
"""
import math

import numpy as np


def interpolation_response(freq_hz, dt):
    """sew.py:35-57 + :60-76: cubic B-spline interpolation response, normalised by its DC value:
    H(f) = 3 sinc(f dt)^4 / (2 + cos(2 pi f dt)),  sinc(x) = sin(pi x)/(pi x)."""
    x = np.asarray(freq_hz, dtype=np.float64) * dt
    s = np.sinc(x)                      # numpy's sinc is the normalised one
    return 3.0 * s ** 4 / (2.0 + np.cos(2.0 * np.pi * x))


def energy(spectrum):
    """sew.py:79-80."""
    return float(np.sum(np.abs(spectrum) ** 2) / len(spectrum))


def reference_spectrum(signal):
    """sew.py:170-179: per-frequency norm over the axes, DC removed, scaled by sqrt(1/d)."""
    sig = np.atleast_2d(np.asarray(signal, dtype=np.float64))
    if sig.ndim != 2:
        raise ValueError("Signal must be at most 2D")
    spec = np.fft.fft(sig, axis=1)
    spec[:, 0] = 0.0
    return math.sqrt(1.0 / sig.shape[0]) * np.linalg.norm(spec, axis=0)


def brent_root(f, xa, xb, xtol=2e-12, rtol=8.881784197001252e-16, maxiter=100):
    """Brent's method with the bookkeeping of SciPy's brentq [EXT]."""
    xpre, xcur = xa, xb
    fpre, fcur = f(xpre), f(xcur)
    if fpre == 0.0:
        return xpre
    if fcur == 0.0:
        return xcur
    if (fpre < 0) == (fcur < 0):
        raise ValueError("f(a) and f(b) must have different signs")
    xblk = fblk = spre = scur = 0.0
    for _ in range(maxiter):
        if fpre != 0.0 and fcur != 0.0 and ((fpre < 0) != (fcur < 0)):
            xblk, fblk = xpre, fpre
            spre = scur = xcur - xpre
        if abs(fblk) < abs(fcur):
            xpre, xcur, xblk = xcur, xblk, xcur
            fpre, fcur, fblk = fcur, fblk, fcur
        delta = (xtol + rtol * abs(xcur)) / 2.0
        sbis = (xblk - xcur) / 2.0
        if fcur == 0.0 or abs(sbis) < delta:
            return xcur
        if abs(spre) > delta and abs(fcur) < abs(fpre):
            if xpre == xblk:
                stry = -fcur * (xcur - xpre) / (fcur - fpre)
            else:
                dpre = (fpre - fcur) / (xpre - xcur)
                dblk = (fblk - fcur) / (xblk - xcur)
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
            if 2.0 * abs(stry) < min(abs(spre), 3.0 * abs(sbis) - delta):
                spre, scur = scur, stry
            else:
                spre = scur = sbis
        else:
            spre = scur = sbis
        xpre, fpre = xcur, fcur
        if abs(scur) > delta:
            xcur += scur
        else:
            xcur += delta if sbis > 0 else -delta
        fcur = f(xcur)
    return xcur


def largest_dt_with_quality(quality_func, min_q, min_dt, max_dt):
    """sew.py:83-137: end point, halving back-off until the quality is reached, then Brent."""
    dt = max_dt
    if quality_func(dt) >= min_q:
        return dt
    step = max_dt * 0.5
    best_q, best_dt = 0.0, None
    while True:
        dt = max(dt - step, min_dt)
        q = quality_func(dt)
        if q > min_q:
            return brent_root(lambda x: quality_func(x) - min_q, dt, max_dt)
        step *= 0.5
        if q > best_q:
            best_q, best_dt = q, dt
        if dt <= min_dt:
            return best_dt


def knot_spacing_and_variance(signal, times, quality, min_dt=None, max_dt=None):
    """sew.py:199-235 (find_uniform_knot_spacing_spectrum :141-159, dt_to_variance_spectrum :192-195)."""
    xhat = reference_spectrum(signal)
    times = np.asarray(times, dtype=np.float64)
    rate = 1.0 / np.mean(np.diff(times))
    freqs = np.fft.fftfreq(len(times), d=1.0 / rate)
    max_remove = energy(xhat) * (1.0 - quality)

    def quality_func(dt):
        return max_remove / energy((1.0 - interpolation_response(freqs, dt)) * xhat)

    if min_dt is None:
        min_dt = 1.0 / rate
    if max_dt is None:
        max_dt = (len(times) / 4.0) / rate
    dt = largest_dt_with_quality(quality_func, 1.0, min_dt, max_dt)
    variance = energy((1.0 - interpolation_response(freqs, dt)) * xhat) / len(xhat)
    return float(dt), float(variance)

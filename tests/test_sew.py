"""Spline error weighting pre-stage (python/sew.py, get_sew_for_dataset.py).

* not gpu: the numpy oracle (oracle/sew_oracle.py) against golden outputs of the REFERENCE module
  (tests/golden/sew_golden.json, produced by tests/golden/make_sew_golden.py importing
  /root/reference/python/sew.py) -- this oracle is pinned by the reference itself;
* gpu: the HIP path (hipFFT + spectral reduction kernels behind oicc_sew_knot_spacing_and_variance)
  against the same goldens and against the oracle.
Tolerances: knot spacing 1e-9 relative (Brent's xtol is 2e-12), variance 1e-9 relative.
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sew_cases  # noqa: E402
import sew_oracle  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "sew_golden.json")))
CASES = sew_cases.cases()


def close(a, b, rel=1e-9):
    return abs(a - b) <= rel * max(abs(a), abs(b))


def test_cases_are_the_ones_the_goldens_were_made_from():
    assert set(CASES) == set(GOLD)
    for name, (sig, t, q, lo, hi) in CASES.items():
        assert sew_cases.checksum(sig, t) == GOLD[name]["input_sha"], name
        assert (q, lo, hi, len(t)) == (GOLD[name]["quality"], GOLD[name]["min_dt"], GOLD[name]["max_dt"], GOLD[name]["n"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_goldens(name):
    sig, t, q, lo, hi = CASES[name]
    dt, var = sew_oracle.knot_spacing_and_variance(sig, t, q, min_dt=lo, max_dt=hi)
    assert close(dt, GOLD[name]["dt"]) and close(var, GOLD[name]["variance"])


def test_oracle_response_properties():
    f = np.linspace(0.0, 100.0, 2001)
    for dt in (0.01, 0.05, 0.2):
        H = sew_oracle.interpolation_response(f, dt)
        assert H[0] == 1.0 and np.all(H <= 1.0 + 1e-15) and np.all(H >= 0.0)       # low-pass, unit DC gain (sew.py:75-76)
        assert np.allclose(H, sew_oracle.interpolation_response(-f, dt), rtol=0, atol=0)   # even in f: half spectrum suffices
    # a longer knot spacing removes more of any spectrum
    xhat = sew_oracle.reference_spectrum(CASES["C2_accel_r3"][0])
    fr = np.fft.fftfreq(len(xhat), d=1 / 200.0)
    rem = [sew_oracle.energy((1 - sew_oracle.interpolation_response(fr, dt)) * xhat) for dt in (0.01, 0.02, 0.05, 0.1)]
    assert all(a < b for a, b in zip(rem, rem[1:]))


def test_brent_restatement_against_known_roots():
    r = sew_oracle.brent_root(lambda x: np.cos(x) - x, 0.0, 1.0)
    assert abs(r - 0.7390851332151607) < 1e-11
    r = sew_oracle.brent_root(lambda x: x ** 3 - 2 * x - 5, 2.0, 3.0)
    assert abs(r - 2.0945514815423265) < 1e-11
    with pytest.raises(ValueError):
        sew_oracle.brent_root(lambda x: x * x + 1, 0.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_reference_goldens_and_oracle(name):
    from openimucameracalibrator_amd import sew
    sig, t, q, lo, hi = CASES[name]
    dt, var = sew.knot_spacing_and_variance(sig, t, q, min_dt=lo, max_dt=hi)
    assert close(dt, GOLD[name]["dt"]) and close(var, GOLD[name]["variance"])
    odt, ovar = sew_oracle.knot_spacing_and_variance(sig, t, q, min_dt=lo, max_dt=hi)
    assert close(dt, odt) and close(var, ovar)


@pytest.mark.gpu
def test_hip_spline_weighting_json_and_errors():
    from openimucameracalibrator_amd import sew, synthetic
    ds = synthetic.make_config("C2")
    tel = dict(accelerometer=ds.accel.tolist(), gyroscope=ds.gyro.tolist(), timestamps_ns=(ds.imu_t_s * 1e9).round().astype(np.int64).tolist(), camera_fps=0.0)
    sw = sew.spline_weighting_for_telemetry(tel)
    assert set(sw) == {"so3", "r3", "camera_fps"} and sw["camera_fps"] == 30.0       # get_sew_for_dataset.py:46-51
    assert close(sw["r3"]["knot_spacing"], GOLD["C2_accel_r3"]["dt"], 1e-6) and close(sw["so3"]["knot_spacing"], GOLD["C2_gyro_so3"]["dt"], 1e-6)
    assert close(sw["r3"]["weighting_factor"], np.sqrt(GOLD["C2_accel_r3"]["variance"]), 1e-6)
    with pytest.raises(RuntimeError):
        sew.knot_spacing_and_variance(np.zeros((3, 4)), np.arange(4.0), 0.9)        # n < 8
    with pytest.raises(ValueError):
        sew.knot_spacing_and_variance(np.zeros((2, 2, 16)), np.arange(16.0), 0.9)   # more than 2-D (sew.py:173-174)


@pytest.mark.gpu
def test_hip_large_series_linearity_property():
    """C5-size series (200 k samples): scaling the signal scales the variance by the square and
    leaves the knot spacing unchanged (size-independent property, no oracle needed)."""
    from openimucameracalibrator_amd import sew
    rng = np.random.default_rng(3)
    n = 200000
    t = np.arange(n) / 200.0
    sig = np.cumsum(rng.standard_normal((3, n)), axis=1) * 0.01 + 0.05 * rng.standard_normal((3, n))
    d1, v1 = sew.knot_spacing_and_variance(sig, t, 0.97, min_dt=0.01, max_dt=0.3)
    d2, v2 = sew.knot_spacing_and_variance(3.0 * sig, t, 0.97, min_dt=0.01, max_dt=0.3)
    assert close(d1, d2, 1e-9) and close(9.0 * v1, v2, 1e-9) and 0.01 <= d1 <= 0.3


@pytest.mark.skipif(not os.path.exists("/root/reference/python/sew.py"), reason="live comparison: only where the reference tree is mounted")
def test_oracle_matches_the_reference_module_on_many_random_signals():
    """beyond the seven committed goldens: 60 seeded signals (random walk + oscillations + noise; odd and even lengths;
    one or three axes; several quality targets and spacing ranges) through the reference's python/sew.py and through
    oracle/sew_oracle.py, live."""
    import subprocess
    import sys as _sys
    # the reference module is imported in a child interpreter that never writes bytecode into the read-only tree
    code = r'''
import sys, json, numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/python")
import sew
out = []
for seed in range(60):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(400, 3000)); axes = 1 if seed % 5 == 0 else 3
    rate = float(rng.choice([100.0, 200.0, 400.0]))
    t = np.arange(n) / rate
    sig = np.cumsum(rng.standard_normal((axes, n)), axis=1) * rng.uniform(0.002, 0.05) + rng.uniform(0.0, 0.3) * rng.standard_normal((axes, n))
    for k in range(3):
        sig += rng.uniform(0.1, 1.0) * np.sin(2 * np.pi * rng.uniform(0.2, 15.0) * t + rng.uniform(0, 6.28))
    q = float(rng.choice([0.9, 0.96, 0.99, 0.999])); lo = float(rng.choice([0.005, 0.01, 0.02])); hi = float(rng.choice([0.1, 0.15, 0.4]))
    dt, var = sew.knot_spacing_and_variance(sig, t, q, min_dt=lo, max_dt=hi)
    out.append(dict(seed=seed, dt=float(dt), var=float(var)))
print(json.dumps(out))
'''
    ref = json.loads(subprocess.check_output([_sys.executable, "-c", code], env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1")).decode().splitlines()[-1])
    for r in ref:
        seed = r["seed"]
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(400, 3000)); axes = 1 if seed % 5 == 0 else 3
        rate = float(rng.choice([100.0, 200.0, 400.0]))
        t = np.arange(n) / rate
        sig = np.cumsum(rng.standard_normal((axes, n)), axis=1) * rng.uniform(0.002, 0.05) + rng.uniform(0.0, 0.3) * rng.standard_normal((axes, n))
        for k in range(3):
            sig += rng.uniform(0.1, 1.0) * np.sin(2 * np.pi * rng.uniform(0.2, 15.0) * t + rng.uniform(0, 6.28))
        q = float(rng.choice([0.9, 0.96, 0.99, 0.999])); lo = float(rng.choice([0.005, 0.01, 0.02])); hi = float(rng.choice([0.1, 0.15, 0.4]))
        dt, var = sew_oracle.knot_spacing_and_variance(sig, t, q, min_dt=lo, max_dt=hi)
        assert abs(dt - r["dt"]) <= 1e-9 * r["dt"], (seed, dt, r["dt"])
        assert abs(var - r["var"]) <= 1e-8 * abs(r["var"]) + 1e-300, (seed, var, r["var"])


class _OracleSewBackend:
    """test-only: the numpy oracle behind the C-ABI call shape openimucameracalibrator_amd.sew binds (no GPU here)."""

    def sew_knot_spacing_and_variance(self, device, dims, n, sig_p, t_p, q, lo, hi, dt_ref, var_ref, ev_ref):
        sig = np.ctypeslib.as_array(sig_p, shape=(dims, n)); t = np.ctypeslib.as_array(t_p, shape=(n,))
        dt, var = sew_oracle.knot_spacing_and_variance(sig, t, q, min_dt=lo if lo > 0 else None, max_dt=hi if hi > 0 else None)
        dt_ref._obj.value = dt; var_ref._obj.value = var; ev_ref._obj.value = 0
        return 0


@pytest.mark.skipif(not os.path.exists("/root/reference/python/get_sew_for_dataset.py"), reason="live comparison: only where the reference tree is mounted")
def test_twin_application_writes_what_the_reference_script_writes(tmp_path):
    """python -m openimucameracalibrator_amd.sew is the twin of python/get_sew_for_dataset.py: the REFERENCE SCRIPT ITSELF is run
    on a telemetry file (child interpreter, bytecode writing off; its unused imports cv2 / matplotlib, absent here, are
    stubbed) and its output JSON is compared with the twin's (host logic of the twin + numpy oracle instead of the device)."""
    import subprocess
    import sys as _sys
    from openimucameracalibrator_amd import synthetic, io_files, sew as twin
    ds = synthetic.make_config("C1", camera="gopro9_division")
    files = io_files.write_dataset_files(ds, str(tmp_path))
    tel = json.load(open(files["telemetry_json"]))
    tel["camera_fps"] = 59.94                                   # read_generic_json requires the key (telemetry_converter.py:237)
    json.dump(tel, open(files["telemetry_json"], "w"))
    out_ref = str(tmp_path / "sew_ref.json")
    code = r'''
import sys, types
sys.dont_write_bytecode = True
for name in ("cv2", "matplotlib", "matplotlib.pyplot", "natsort"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
sys.path.insert(0, "/root/reference/python")
sys.argv = ["get_sew_for_dataset.py", "--input_json_path", sys.argv[1], "--output_path", sys.argv[2]]
import get_sew_for_dataset
get_sew_for_dataset.main()
'''
    subprocess.check_call([_sys.executable, "-c", code, files["telemetry_json"], out_ref], env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"),
                          stdout=subprocess.DEVNULL)
    ref = json.load(open(out_ref))
    mine = twin.spline_weighting_for_telemetry(tel, backend=_OracleSewBackend())
    assert set(mine) == set(ref) == {"so3", "r3", "camera_fps"} and mine["camera_fps"] == ref["camera_fps"] == 59.94
    for k in ("so3", "r3"):
        assert set(mine[k]) == set(ref[k]) == {"knot_spacing", "weighting_factor", "quality_factor"}
        assert mine[k]["quality_factor"] == ref[k]["quality_factor"]
        assert abs(mine[k]["knot_spacing"] - ref[k]["knot_spacing"]) <= 1e-9 * ref[k]["knot_spacing"]
        assert abs(mine[k]["weighting_factor"] - ref[k]["weighting_factor"]) <= 1e-8 * ref[k]["weighting_factor"]

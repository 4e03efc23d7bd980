"""Spline error weighting pre-stage: host-side mirror of the reference's python/sew.py
interface (knot_spacing_and_variance, :199-235) and of python/get_sew_for_dataset.py (:38-56)
over the C-ABI entry oicc_sew_knot_spacing_and_variance.  FFT and spectral reductions run on
the MI355X; there is no CPU fallback."""
import argparse
import ctypes as C
import json

import numpy as np

from . import _lib
from . import io_files


def knot_spacing_and_variance(signal, times, quality, min_dt=None, max_dt=None, verbose=False, device=0, backend=None):
    """Same arguments and return value as the reference function: (dt, variance).
    signal: (dims, n) or (n,) array, times: (n,) seconds, quality in (0, 1)."""
    b = backend if backend is not None else _lib.load()
    sig = np.ascontiguousarray(np.atleast_2d(np.asarray(signal, dtype=np.float64)))
    if sig.ndim != 2:
        raise ValueError("Signal must be at most 2D")          # sew.py:173-174
    t = np.ascontiguousarray(np.asarray(times, dtype=np.float64).ravel())
    dims, n = sig.shape
    if t.shape[0] != n:
        raise ValueError("times and signal lengths differ")
    dt, var, ev = C.c_double(), C.c_double(), C.c_int32()
    rc = b.sew_knot_spacing_and_variance(int(device), dims, n, sig.ctypes.data_as(C.POINTER(C.c_double)),
                                         t.ctypes.data_as(C.POINTER(C.c_double)), float(quality),
                                         float(min_dt) if min_dt is not None else 0.0, float(max_dt) if max_dt is not None else 0.0,
                                         C.byref(dt), C.byref(var), C.byref(ev))
    if rc != 0:
        raise RuntimeError("oicc_sew_knot_spacing_and_variance failed with status %d" % rc)
    if verbose:
        print("dt=%.6e variance=%.6e (%d spectral evaluations)" % (dt.value, var.value, ev.value))
    return dt.value, var.value


def spline_weighting_for_telemetry(telemetry, q_so3=0.98, q_r3=0.96, device=0, backend=None):
    """get_sew_for_dataset.py:33-51: the spline_weighting JSON object the calibration CLI reads
    (read_misc.cc:40-44).  telemetry: dict with accelerometer[[3]], gyroscope[[3]], timestamps_ns[]."""
    accl = np.asarray(telemetry["accelerometer"], dtype=np.float64)
    gyro = np.asarray(telemetry["gyroscope"], dtype=np.float64)
    t = np.asarray(telemetry["timestamps_ns"], dtype=np.float64).squeeze() * 1e-9
    r3_dt, r3_var = knot_spacing_and_variance(accl.T, t, q_r3, min_dt=0.01, max_dt=0.15, device=device, backend=backend)
    so3_dt, so3_var = knot_spacing_and_variance(gyro.T, t, q_so3, min_dt=0.01, max_dt=0.2, device=device, backend=backend)
    fps = float(telemetry.get("camera_fps", 0.0) or 0.0)
    return {"so3": {"knot_spacing": so3_dt, "weighting_factor": float(np.sqrt(so3_var)), "quality_factor": q_so3},
            "r3": {"knot_spacing": r3_dt, "weighting_factor": float(np.sqrt(r3_var)), "quality_factor": q_r3},
            "camera_fps": fps if fps != 0.0 else 30.0}


def main(argv=None):
    ap = argparse.ArgumentParser(description="spline error weighting for a telemetry JSON (get_sew_for_dataset.py)")
    ap.add_argument("--input_json_path", default="", help="path to metadata json")
    ap.add_argument("--output_path", help="output path")
    ap.add_argument("--q_so3", default=0.98, type=float, help="quality value for the rotational component (gyro)")
    ap.add_argument("--q_r3", default=0.96, type=float, help="quality value for the translational component (accelerometer)")
    ap.add_argument("--use_gopro_importer", default=0, help="raw GoPro telemetry (get_sew_for_dataset.py:24-31); not supported here: convert it to the generic telemetry JSON first")
    ap.add_argument("--device", default=0, type=int)
    args = io_files.parse_reference_flags(ap, argv)
    if str(args.use_gopro_importer) not in ("0", "", "False", "false"):
        raise SystemExit("--use_gopro_importer: the raw GoPro telemetry importer (python/telemetry_converter.py) is outside this path; "
                         "pass the generic telemetry JSON {accelerometer, gyroscope, timestamps_ns, camera_fps}")
    with open(args.input_json_path) as f:
        tel = json.load(f)
    sw = spline_weighting_for_telemetry(tel, args.q_so3, args.q_r3, device=args.device)
    print("Knot spacing SO3:               {:.3f} seconds at quality level q_so3={}".format(sw["so3"]["knot_spacing"], args.q_so3))
    print("Knot spacing  R3:               {:.3f} seconds at quality level q_r3={}".format(sw["r3"]["knot_spacing"], args.q_r3))
    print("Gyroscope weighting factor:     {:.3f} at quality level q_so3={}".format(1.0 / sw["so3"]["weighting_factor"], args.q_so3))
    print("Accelerometer weighting factor: {:.3f} at quality level q_r3={}".format(1.0 / sw["r3"]["weighting_factor"], args.q_r3))
    print("Writing result to: ", args.output_path)
    with open(args.output_path, "w") as f:
        json.dump(sw, f)


if __name__ == "__main__":
    main()

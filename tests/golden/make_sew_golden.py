"""Regenerates tests/golden/sew_golden.json with the REFERENCE implementation: imports
/root/reference/python/sew.py (possible only in the build container) and records
knot_spacing_and_variance for the seeded signals of tests/sew_cases.py.  These goldens pin the
oracle (oracle/sew_oracle.py) and, through it and directly, the HIP path.
Run:  python tests/golden/make_sew_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference/python")
sys.dont_write_bytecode = True   # never write a __pycache__ into the read-only reference tree
import sew as reference_sew  # noqa: E402  (the reference module)
import sew_cases  # noqa: E402


def main():
    out = {}
    for name, (sig, t, q, lo, hi) in sew_cases.cases().items():
        dt, var = reference_sew.knot_spacing_and_variance(sig, t, q, min_dt=lo, max_dt=hi)
        out[name] = dict(quality=q, min_dt=lo, max_dt=hi, n=int(len(t)), input_sha=sew_cases.checksum(sig, t), dt=float(dt), variance=float(var))
        print(name, out[name])
    with open(os.path.join(HERE, "sew_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

"""Regenerates tests/golden/tiny_problem.json: outputs of the CPU checker (oracle/)
on the seeded `tiny` dataset.  The reference itself cannot be run here (no Eigen /
Ceres / TheiaSfM), so these are oracle-generated fixtures that pin the oracle
against regressions and give the -m gpu tests a committed target; the only goldens
that come from the reference are the doc-comment matrices in test_oracle_golden.py.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_backend  # noqa: E402
from openimucameracalibrator_amd import synthetic, estimator as E  # noqa: E402

FLAGS = E.SPLINE | E.T_I_C | E.GRAVITY_DIR


def main():
    ds = synthetic.make_config("tiny")
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    tr = cal.trajectory_
    cost, H, g = tr.Evaluate(FLAGS)
    r0, _ = tr.EvaluateBlocks(FLAGS, 0, 2 * cal.num_corners, want_jac=False)
    out = dict(seed=synthetic.SEED, num_blocks=cal.num_blocks, num_corners=cal.num_corners, P=int(len(g)),
               initial_cost=cost, grad_norm=float(np.linalg.norm(g)), grad_head=g[:12].tolist(), H_trace=float(np.trace(H)),
               H_fro=float(np.linalg.norm(H)), first_view_residuals=r0[:8].tolist(), initial_reproj=tr.GetMeanReprojectionError())
    s = tr.Optimize(50, FLAGS)
    out.update(lm_iterations=s["num_iterations"], lm_costs=[i["cost"] for i in tr.GetIterations()], final_cost=s["final_cost"],
               final_T_i_c=tr.GetT_i_c().tolist(), final_gravity=tr.GetGravity().tolist(), final_reproj=tr.GetMeanReprojectionError(),
               message=s["message"])
    with open(os.path.join(HERE, "tiny_problem.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tiny_problem.json")


if __name__ == "__main__":
    main()

"""Writes tests/golden/ba_problem.json: a small view-bundle-adjustment problem (inputs included, 12 views x 30 corners,
division-undistortion camera, 10 % outliers so that the Huber loss is active) with the CPU oracle's answers
(oracle/ba_oracle.cpp): cost / gradient / trace of J^T J at the start, the LM run of the first calibration stage, the
per-view refinement and the board-point refinement.  Regression pin for both the oracle and the HIP path; it is NOT a
reference output (Theia / Ceres cannot run here).
    python tests/golden/make_ba_golden.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_backend  # noqa: E402
from openimucameracalibrator_amd import camera_calibrator as CC  # noqa: E402


def adjuster(d, backend):
    ba = CC.ViewBundleAdjuster(backend=backend)
    ba.SetCamera(d["model"], d["intrinsics_init"]); ba.SetScenePoints(d["points"])
    ba.SetViews(d["pose_init"], d["corner_offset"], d["uv"], d["point_ids"])
    return ba


def answers(d, backend):
    flags = CC.BA_POSITION | CC.BA_ORIENTATION
    mask = CC.intrinsics_mask(d["model"], CC.FOCAL_LENGTH | CC.RADIAL_DISTORTION)
    ba = adjuster(d, backend)
    cost, H, g = ba.Evaluate(flags, mask)
    out = dict(P=int(len(g)), initial_cost=float(cost), grad_norm=float(np.linalg.norm(g)), grad_tail=[float(x) for x in g[-2:]], H_trace=float(np.trace(H)))
    s = ba.Optimize(100, flags, mask)
    out.update(lm_iterations=int(s["num_iterations"]), message=s["message"], lm_costs=[float(i["cost"]) for i in ba.Iterations()],
               final_intrinsics=[float(x) for x in ba.GetCamera()], final_pose0=[float(x) for x in ba.GetPoses()[0]])
    ba = adjuster(d, backend)
    it, fc = ba.OptimizeViews(50)
    out.update(view_iterations=[int(x) for x in it], view_costs=[float(x) for x in fc], view_pose5=[float(x) for x in ba.GetPoses()[5]])
    ba = adjuster(dict(d, pose_init=d["pose_true"], intrinsics_init=d["intrinsics"]), backend)
    s = ba.Optimize(50, CC.BA_POINTS, 0)
    out.update(points_iterations=int(s["num_iterations"]), points_final_cost=float(s["final_cost"]), point7=[float(x) for x in ba.GetScenePoints()[7]])
    return out


def load_inputs(gold):
    d = {k: np.array(v) for k, v in gold["inputs"].items() if k != "model"}
    d["model"] = gold["inputs"]["model"]
    d["corner_offset"] = d["corner_offset"].astype(np.int64); d["point_ids"] = d["point_ids"].astype(np.int32)
    return d


if __name__ == "__main__":
    ds = CC.make_calibration_dataset("gopro9_division", num_views=12, corners_per_view=30, outlier_fraction=0.1, seed=7)
    pts = ds["points"].copy(); pts[:, 2] += 3e-4 * np.cos(np.arange(48))
    intr0 = ds["intrinsics"].copy(); intr0[0] *= 1.04
    inputs = dict(model=int(ds["model"]), intrinsics=ds["intrinsics"].tolist(), intrinsics_init=intr0.tolist(), points=pts.tolist(),
                  pose_init=ds["pose_init"].tolist(), pose_true=ds["pose_true"].tolist(), corner_offset=ds["corner_offset"].tolist(),
                  uv=ds["uv"].tolist(), point_ids=ds["point_ids"].tolist())
    gold = dict(inputs=inputs)
    gold["answers"] = answers(load_inputs(gold), oracle_backend.load_ba())
    json.dump(gold, open(os.path.join(HERE, "ba_problem.json"), "w"))
    print({k: v for k, v in gold["answers"].items() if not isinstance(v, list)})

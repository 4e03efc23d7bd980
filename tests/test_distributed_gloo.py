"""N > 1 path on CPU: world_size-2 gloo run of the residual-block sharding
(SURVEY.md 8e).  Each rank holds its time window of views / IMU samples and the
full parameter set; the packed {J^T J, J^T r, cost} are summed with an all-reduce
(RCCL on the GPUs, gloo here) and must equal the single-process normal equations.
The per-rank evaluator is the CPU checker (no GPU in this container); the
sharding, the remote-measurement layout declaration and the reduction are the
product's host logic.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = 64 | 2 | 16   # SPLINE | T_I_C | GRAVITY_DIR


def _worker(rank, world, port, out_path, FLAGS=FLAGS):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_backend
    from openimucameracalibrator_amd import synthetic, estimator as E
    ds = synthetic.make_config("tiny", num_views=24, duration=2.4)
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds, shard=(rank, world))
    cost, H, g = cal.trajectory_.Evaluate(FLAGS)
    packed = torch.from_numpy(np.concatenate([H.ravel(), g, [cost], [cal.num_blocks]]))
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    if rank == 0:
        whole = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
        c1, H1, g1 = whole.trajectory_.Evaluate(FLAGS)
        P = len(g1)
        Hs = packed[:P * P].numpy().reshape(P, P); gs = packed[P * P:P * P + P].numpy(); cs = float(packed[P * P + P]); nb = int(packed[-1])
        ok = (H.shape == H1.shape and np.abs(Hs - H1).max() <= 1e-10 * np.abs(H1).max()
              and np.abs(gs - g1).max() <= 1e-10 * np.abs(g1).max() and abs(cs - c1) <= 1e-12 * c1 and nb == whole.num_blocks)
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch %g %g %g %d %d" % (np.abs(Hs - H1).max(), np.abs(gs - g1).max(), abs(cs - c1), nb, whole.num_blocks))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flags", [FLAGS, FLAGS | 1])   # | SplineOptimFlags::POINTS: every board point a variable on every rank (the ranks agree on the layout)
def test_two_rank_sharded_normal_equations_sum_to_whole(tmp_path, flags):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, out, flags), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _ba_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_backend
    from openimucameracalibrator_amd import camera_calibrator as CC

    def adjuster(d):
        ba = CC.ViewBundleAdjuster(backend=oracle_backend.load_ba())
        ba.SetCamera(d["model"], d["intrinsics"]); ba.SetScenePoints(d["points"])
        ba.SetViews(d["pose_init"], d["corner_offset"], d["uv"], d["point_ids"])
        return ba
    ds = CC.make_calibration_dataset("gopro9_division", num_views=13, corners_per_view=30, outlier_fraction=0.05)
    mine = CC.shard_views(ds, rank, world)
    flags, mask = CC.BA_POSITION | CC.BA_ORIENTATION, CC.intrinsics_mask(ds["model"], CC.FOCAL_LENGTH | CC.RADIAL_DISTORTION)
    # joint bundle adjustment: packed normal equations summed over the ranks = the whole problem's
    cost, H, g = adjuster(mine).Evaluate(flags, mask)
    packed = torch.from_numpy(np.concatenate([H.ravel(), g, [cost]]))
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    # per-view refinement: no collective in the data path; the poses of the shards are gathered afterwards
    ba = adjuster(mine)
    lo, hi = mine["shard"]
    it, fc = ba.OptimizeViews(50)
    pose = torch.from_numpy(ba.GetPoses().copy())
    own = torch.zeros(len(pose), dtype=torch.float64); own[lo:hi] = 1.0
    merged = pose * own[:, None]
    dist.all_reduce(merged, op=dist.ReduceOp.SUM)
    if rank == 0:
        whole = adjuster(ds)
        c1, H1, g1 = whole.Evaluate(flags, mask)
        P = len(g1)
        Hs = packed[:P * P].numpy().reshape(P, P); gs = packed[P * P:P * P + P].numpy(); cs = float(packed[-1])
        whole.OptimizeViews(50)
        ok = (np.abs(Hs - H1).max() <= 1e-10 * np.abs(H1).max() and np.abs(gs - g1).max() <= 1e-10 * np.abs(g1).max()
              and abs(cs - c1) <= 1e-12 * c1 and np.abs(merged.numpy() - whole.GetPoses()).max() == 0.0
              and np.all(it[hi:] == -1) and np.all(it[lo:hi] > 0))       # views without observations are left alone
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_bundle_adjustment_shards(tmp_path):
    """View bundle adjustment over 2 ranks: joint normal equations sum to the whole; per-view refinement needs no collective."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result_ba.txt")
    mp.spawn(_ba_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"

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


def _worker(rank, world, port, out_path):
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


def test_two_rank_sharded_normal_equations_sum_to_whole(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"

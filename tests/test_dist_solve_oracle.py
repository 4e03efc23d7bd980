"""The distributed linear solve on the CPU: the decomposition the HIP library uses --
every rank eliminates the interior of its own block range from the rows it owns, ONE gather of the separator messages, the top system
solved by every rank, the interior back-substituted, ONE gather of the solutions -- restated with dense numpy algebra
(oracle/dist_solve_oracle.py) and checked against numpy's solve of the whole system, in one process and with world_size 2 / 3 gloo
ranks that exchange exactly those two messages.  (The HIP kernels themselves are held to one-process LM steps and to the residual of
the packed normal equations on the GPU: tests/test_gpu_parity.py::test_distributed_cyclic_reduction_*.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dist_solve_oracle as D   # noqa: E402


@pytest.mark.parametrize("nblk,bs,a,last,b0s", [
    (8, 16, 5, None, [0, 4, 8]),                 # two equal ranges
    (11, 12, 9, 7, [0, 3, 4, 9, 11]),            # unequal ranges, a rank with one block (no interior), a short last block
    (5, 8, 3, None, [0, 1, 2, 3, 4, 5]),         # one block per rank: the top system is the whole system
    (29, 8, 9, 5, [0, 4, 8, 11, 15, 18, 22, 26, 29]),   # BASELINE config 2's block count on eight ranks
    (6, 10, 0, None, [0, 3, 6]),                 # no arrow
])
def test_decomposition_solves_the_whole_system(nblk, bs, a, last, b0s):
    M, rhs, edges = D.random_system(nblk, bs, a, seed=nblk * 7 + a, last=last)
    x = D.solve_distributed(M, rhs, edges, a, b0s)
    ref = np.linalg.solve(M, rhs)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()
    assert np.abs(M @ x - rhs).max() <= 1e-9 * np.abs(rhs).max()


def test_a_rank_reads_only_the_rows_it_owns():
    """What a rank is handed (owned_rows) holds no entry of another rank's rows: changing those changes nothing in its message."""
    M, rhs, edges = D.random_system(9, 10, 4, seed=3)
    own = D.owned_rows(M, rhs, edges, 4, 3, 6)
    m0, _ = D.forward(own, 4)
    M2 = M.copy(); pb = int(edges[-1])
    lo, hi = int(edges[3]), int(edges[6])
    mask = np.ones(M.shape[0], bool); mask[lo:hi] = False
    M2[np.ix_(mask, mask)] += 1.0          # every entry outside the rank's rows (and their mirror images)
    own2 = D.owned_rows(M2, rhs, edges, 4, 3, 6)
    m1, _ = D.forward(own2, 4)
    for k in m0:
        assert np.array_equal(np.asarray(m0[k]), np.asarray(m1[k])), k
    assert pb == 90


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nblk, bs, a = 13, 12, 6
    M, rhs, edges = D.random_system(nblk, bs, a, seed=11, last=9)          # (every rank builds the same system, then keeps ITS rows only)
    b0s = [nblk * k // world for k in range(world + 1)]
    pb = int(edges[-1])
    own = D.owned_rows(M, rhs, edges, a, b0s[rank], b0s[rank + 1])
    corner = M[pb:, pb:].copy() if rank == 0 else None
    rhs_a = rhs[pb:].copy() if rank == 0 else None
    del M
    msg, keep = D.forward(own, a, corner=corner, rhs_arrow=rhs_a)
    msgs = [None] * world
    dist.all_gather_object(msgs, msg)                                       # gather 1: the separator messages
    xs, xa = D.top_solve(msgs, a)                                           # (replicated)
    mine = D.backward(keep, xs[rank], xs[rank + 1] if rank + 1 < world else np.zeros(0), xa)
    parts = [None] * world
    dist.all_gather_object(parts, mine)                                     # gather 2: the step
    x = np.concatenate(parts + [xa])
    if rank == 0:
        M, rhs, _ = D.random_system(nblk, bs, a, seed=11, last=9)
        ref = np.linalg.solve(M, rhs)
        ok = np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch %g" % np.abs(x - ref).max())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_solve_the_whole_system_with_two_gathers(tmp_path, world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"

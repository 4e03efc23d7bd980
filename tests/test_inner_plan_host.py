"""The plan of the inner iterations (Ceres' use_inner_iterations, reference impl.h:266) is host logic of the product library:
parameter blocks in the reference's creation order, the Hessian graph, Ceres' recursive independent-set ordering, the items of
every block.  The library builds it from INTERVALS of neighbouring knots (no cliques, no adjacency lists: O(knots)); the oracle
restates Ceres literally (a clique per residual block, sorted adjacency lists).  These tests build the library's plan WITHOUT a
device (oicc_debug_create_host_only / oicc_debug_host_inner_plan, debug exports outside include/oicc_hip.h) and compare:
same blocks, same sets, same order inside the sets -- and check the definition: no residual block depends on two blocks of a set.
CPU only; the sweeps themselves are compared on the GPU (tests/test_gpu_parity.py::test_inner_iterations_match_the_oracle)."""
import ctypes as C

import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import _abi, _lib, synthetic, estimator as E

FLAGS1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR


class _HostOnly(E.SplineTrajectoryEstimator):
    """The mirror on a problem object that has no device behind it: only set-up and Add* calls are valid."""

    def __init__(self):
        b = _lib.load()
        raw = b.lib
        raw.oicc_debug_create_host_only.restype = C.c_int
        raw.oicc_debug_create_host_only.argtypes = [C.POINTER(_abi.H)]
        raw.oicc_debug_destroy_host_only.restype = None
        raw.oicc_debug_destroy_host_only.argtypes = [_abi.H]
        raw.oicc_debug_host_inner_plan.restype = C.c_int
        raw.oicc_debug_host_inner_plan.argtypes = [_abi.H, C.c_int32, _abi.c_i32p, C.c_int32, _abi.c_i32p, _abi.c_i32p]
        self._b = b
        self._raw = raw
        h = _abi.H()
        assert raw.oicc_debug_create_host_only(C.byref(h)) == 0
        self._h = h
        self._views = []; self._points = None
        self._T_i_c = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
        self._keep = []

    def __del__(self):
        if getattr(self, "_h", None):
            self._raw.oicc_debug_destroy_host_only(self._h); self._h = None

    def plan_shape(self, n_blocks):
        """[sets on the wave kernel, large shared blocks, their parts, largest part count, control blocks, set-kernel workgroups], parts per block"""
        raw = self._raw
        raw.oicc_debug_host_inner_plan_shape.restype = C.c_int
        raw.oicc_debug_host_inner_plan_shape.argtypes = [_abi.H, _abi.c_i32p, _abi.c_i32p, C.c_int32]
        out = np.zeros(6, dtype=np.int32); parts = np.zeros(n_blocks, dtype=np.int32)
        n = raw.oicc_debug_host_inner_plan_shape(self._h, out.ctypes.data_as(_abi.c_i32p), parts.ctypes.data_as(_abi.c_i32p), n_blocks)
        assert n == n_blocks, n
        return out, parts

    def plan(self, flags):
        cap = 1 << 17
        out = np.zeros((cap, 8), dtype=np.int32); ns = C.c_int32(0); nw = C.c_int32(0)
        n = self._raw.oicc_debug_host_inner_plan(self._h, int(flags), out.ctypes.data_as(_abi.c_i32p), cap, C.byref(ns), C.byref(nw))
        assert n >= 0, n
        return out[:n].copy(), ns.value, nw.value


def _oracle_ordering(cal, flags):
    raw = oracle_backend.load().raw
    raw.oicc_oracle_inner_ordering.restype = C.c_int
    raw.oicc_oracle_inner_ordering.argtypes = [_abi.H, C.c_int32, _abi.c_i32p, C.c_int32, _abi.c_i32p]
    cap = 1 << 17
    out = np.zeros((cap, 4), dtype=np.int32); ns = C.c_int32(0)
    n = raw.oicc_oracle_inner_ordering(cal.trajectory_._h, int(flags), out.ctypes.data_as(_abi.c_i32p), cap, C.byref(ns))
    assert n >= 0, n
    return out[:n].copy(), ns.value


def _pair(cfg, **kw):
    ds = synthetic.make_config(cfg, **kw)
    host = E.ImuCameraCalibrator(trajectory=_HostOnly()).BatchInitSpline(ds)
    cpu = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    return ds, host, cpu


@pytest.mark.parametrize("cfg,flags", [("tiny", FLAGS1), ("tiny", FLAGS1 | E.IMU_BIASES | E.CAM_LINE_DELAY | E.IMU_INTRINSICS), ("C1", FLAGS1 | E.CAM_LINE_DELAY),
                                       ("C2", FLAGS1), ("C2", FLAGS1 | E.IMU_BIASES), ("C3", FLAGS1),
                                       # SplineOptimFlags::POINTS: the board points as blocks (listed edges next to the intervals)
                                       ("tiny", FLAGS1 | E.POINTS), ("tiny", E.T_I_C | E.POINTS), ("C1", FLAGS1 | E.POINTS | E.CAM_LINE_DELAY | E.IMU_BIASES)])
def test_library_plan_equals_the_oracles_ordering(cfg, flags):
    ds, host, cpu = _pair(cfg)
    blocks, n_sets, n_wgs = host.trajectory_.plan(flags)
    ord_, n_sets_o = _oracle_ordering(cpu, flags)
    assert n_sets == n_sets_o and len(blocks) == len(ord_)
    assert np.array_equal(blocks[:, :3], ord_[:, :3])          # set, kind, knot index -- block by block, in processing order
    assert n_wgs >= len(blocks)


@pytest.mark.parametrize("cfg", ["tiny", "C2"])
def test_sets_are_independent(cfg):
    """Definition check from the measurements themselves (CalcTimes arithmetic, impl.h:764-788): inside a set no two knots of one
    spline share a window, and no view touches an SO(3) knot and an R^3 knot of the same set."""
    ds, host, _ = _pair(cfg)
    blocks, n_sets, _ = host.trajectory_.plan(FLAGS1)
    S = 10 ** 9
    start_ns = int(float(ds.view_t_s.min()) * S)
    dts, dtr = int(ds.dt_so3 * S), int(ds.dt_r3 * S)
    t_ns = (np.asarray(ds.view_t_s) * S).astype(np.int64)
    s_so3 = (t_ns - start_ns) // dts; s_r3 = (t_ns - start_ns) // dtr
    seen = set()
    for g in range(n_sets):
        blk = blocks[blocks[:, 0] == g]
        so3 = np.sort(blk[blk[:, 1] == 0][:, 2]); r3 = np.sort(blk[blk[:, 1] == 1][:, 2])
        assert np.all(np.diff(so3) >= 6) and np.all(np.diff(r3) >= 6), (g, so3, r3)
        for v in range(len(t_ns)):
            hit = np.sum((so3 >= s_so3[v]) & (so3 < s_so3[v] + 6)) + np.sum((r3 >= s_r3[v]) & (r3 < s_r3[v] + 6))
            assert hit <= 1, (g, v)
        for b in blk:
            key = (int(b[1]), int(b[2])); assert key not in seen; seen.add(key)
    assert len(seen) == len(blocks)


def test_views_added_out_of_time_order_give_the_plan_of_the_sorted_problem():
    """The reference's application fills its reconstruction in the string order of the corner file's keys ("0", "100000", "1000000",
    "1100000", ..., "200000", ...) and the estimator walks an unordered map: the library must not depend on the order of the
    Add*Measurement calls.  It sorts what it is given by time (sync_groups); plan and creation order equal those of the time-ordered
    problem, and the oracle -- which walks the views in time order when it numbers the parameter blocks -- agrees."""
    ds = synthetic.make_config("C1")
    order = ds.file_key_order()
    assert not np.array_equal(order, np.arange(ds.num_views))
    d2 = ds.with_view_order(order)
    ref, ns_ref, _ = E.ImuCameraCalibrator(trajectory=_HostOnly()).BatchInitSpline(ds).trajectory_.plan(FLAGS1)
    host = E.ImuCameraCalibrator(trajectory=_HostOnly()).BatchInitSpline(d2)
    cpu = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(d2)
    blocks, ns, _ = host.trajectory_.plan(FLAGS1)
    ord_, ns_o = _oracle_ordering(cpu, FLAGS1)
    assert ns == ns_o == ns_ref and np.array_equal(blocks[:, :3], ord_[:, :3]) and np.array_equal(blocks[:, :5], ref[:, :5])


def test_plan_at_scale_marks_wave_sets_and_large_shared_blocks():
    """Round 5: at BASELINE config 5 (30 011 blocks) the plan sends the sets of knot blocks that fill the device (>= 4 x 256 blocks) to
    the one-wave-per-block kernel and the two blocks every view / sample depends on (T_i_c: 500 000 corners, gravity: 200 000
    accelerometer samples) to the sequence-of-launches path, split into parts (of at least 1024 item slots, few enough to be resident
    together) that cover the block exactly once;
    with both options off, and at config 2 with the defaults, everything stays on the set kernel (the shared blocks on resident
    workgroups with a control block each)."""
    ds = synthetic.make_config("C5")
    host = E.ImuCameraCalibrator(trajectory=_HostOnly()).BatchInitSpline(ds)
    tr = host.trajectory_
    blocks, n_sets, n_wgs = tr.plan(FLAGS1)
    shape, parts = tr.plan_shape(len(blocks))
    kinds = blocks[:, 1]
    big = np.nonzero(parts)[0]
    assert shape[0] >= 12 and shape[1] == 2 == len(big) and shape[4] == 2                       # the 6 + 6 knot sets at least; T_i_c, gravity
    assert sorted(kinds[big]) == sorted([2, 3])                                                   # InnerKind: IK_TIC, IK_G
    total = int(blocks[big, 4].sum())
    per_part = max(1024, -(-(-(-total // 512)) // 256) * 256)       # all parts resident at once: two workgroups per compute unit (256 of them)
    for b in big:
        slots = int(blocks[b, 4])
        assert slots >= 65536 and parts[b] == min(1024, -(-slots // per_part))
    assert parts.sum() <= 512
    assert shape[2] == parts.sum() and shape[3] == parts.max() and shape[5] == n_wgs
    assert blocks[big[0], 0] == blocks[big[1], 0]                                                 # one set: Ceres puts the two in the same independent set
    # every other block keeps exactly its workgroups of the set kernel (one each here: no resident sharing left)
    assert n_wgs == len(blocks) - 2
    tr.SetOption("inner_wave_blocks", 2); tr.SetOption("inner_shared_launch_slots", 0)
    blocks2, n_sets2, n_wgs2 = tr.plan(FLAGS1)
    shape2, parts2 = tr.plan_shape(len(blocks2))
    assert np.array_equal(blocks2, blocks) and n_sets2 == n_sets
    assert shape2[0] == 0 and shape2[1] == 0 and shape2[2] == 0 and shape2[4] == 2 and not parts2.any() and n_wgs2 > len(blocks)   # resident parts of the two shared blocks
    small = E.ImuCameraCalibrator(trajectory=_HostOnly()).BatchInitSpline(synthetic.make_config("C2"))
    b3, _, w3 = small.trajectory_.plan(FLAGS1)
    shape3, parts3 = small.trajectory_.plan_shape(len(b3))
    assert shape3[0] == 0 and shape3[1] == 0 and shape3[4] == 2 and w3 > len(b3)


@pytest.mark.parametrize("cfg,world", [("C1", 2), ("C1", 4), ("C2", 2), ("C2", 4), ("C2", 8), ("C3", 8), ("C5", 8)])
def test_owned_ranges_of_time_shards_are_cut_on_64_row_blocks(cfg, world):
    """Round 6 (distributed linear solve): every rank of a time-sharded problem derives the owner plan from its own
    measurements and the DECLARED timestamps of the others (oicc_set_shard, oicc_declare_remote_measurements_from) -- host logic, built
    here without a device for every rank in turn: all ranks derive the same cuts; every cut but the last lies on a multiple of 64
    rows (the blocks of the cyclic reduction: a rank eliminates the blocks of its own range), ascending; every rank owns at least one
    block (BASELINE config 5 on eight ranks: 175-177 of 1407); what rank a sends to rank b is what b expects from a."""
    ds = synthetic.make_config(cfg)
    cuts_all, send, pb = [], [], None
    for r in range(world):
        host = E.ImuCameraCalibrator(trajectory=_HostOnly()).BatchInitSpline(ds, shard=(r, world), owner_computes=True)
        raw = host.trajectory_._raw
        raw.oicc_debug_host_owner_cuts.restype = C.c_int
        raw.oicc_debug_host_owner_cuts.argtypes = [_abi.H, C.c_int32, _abi.c_i32p, _abi.c_i32p, C.c_int32]
        cuts = np.zeros(world + 1, dtype=np.int32); rows = np.zeros(world, dtype=np.int32)
        n = raw.oicc_debug_host_owner_cuts(host.trajectory_._h, FLAGS1, cuts.ctypes.data_as(_abi.c_i32p), rows.ctypes.data_as(_abi.c_i32p), world + 1)
        assert n > 0, n
        pb = n if pb is None else pb
        assert n == pb
        cuts_all.append(cuts.copy()); send.append(rows.copy())
    for c in cuts_all[1:]:
        assert np.array_equal(c, cuts_all[0])
    c = cuts_all[0]
    assert c[0] == 0 and c[-1] == pb and (np.diff(c) > 0).all() and (c[:-1] % 64 == 0).all()
    blocks = np.diff(np.concatenate([c[:-1] // 64, [(pb + 63) // 64]]))
    assert (blocks >= 1).all() and blocks.sum() == (pb + 63) // 64
    if cfg == "C5":
        assert blocks.min() >= 170 and blocks.max() <= 182
    # halo rows: only neighbours exchange rows, at most a few hundred each way
    for a in range(world):
        for b in range(world):
            if abs(a - b) > 1:
                assert send[a][b] == 0
            elif a < b:   # (a cut rounded to a block boundary may leave all shared rows on one side of it: then only one of the two sends)
                assert 0 < send[a][b] + send[b][a] and max(send[a][b], send[b][a]) <= 250, (a, b, send[a][b], send[b][a])

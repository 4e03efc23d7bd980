#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on the MI355X hot path.

metric : residual+Jacobian blocks/sec (whole job), with ms_per_step = wall-clock
         per LM iteration of the GoPro9 continuous-time calibration (config C2).
step   : the device work and host synchronisation of ONE successful Levenberg-
         Marquardt iteration over all residual blocks (one block per view, per
         accelerometer sample, per gyroscope sample): residual + analytic Jacobian
         + J^T J / J^T r assembly (at the candidate: cost, gradient and normal equations in
         one pass), [exchange / all-reduce when N > 1], damped band+arrow solve (block
         cyclic reduction), retraction, trust-region decision.  N = 1: the decision runs
         on the device (LmCtl), the host polls a pinned word one iteration behind; N > 1:
         host-driven with a separate candidate cost pass and one read-back
         (liboicc_hip: oicc_run_lm_iterations == loop body of oicc_optimize).
N = 1  : BASELINE config[1] = C2 (GoPro9 Division-Undistortion 960x540, 200 views x
         40 corners, 4000 IMU samples, dt_r3/so3 = 0.1/0.05 s), synthetic, seeded.
N > 1  : BASELINE config[4] = C5 (10 000 views x 50 corners + 200 000 IMU samples, 1000 s),
         STRONG scaling: rank r holds the r-th of N time shards of the same problem,
         every rank holds all knots; J^T J/J^T r/cost are all-reduced (fp64 sum) over
         RCCL each pass and every rank runs the (replicated) solve.

Launch:  python bench.py                      (N=1)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
                --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
         python bench.py --gpus N             (spawns the line above itself)
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# keep stdout to the single JSON line: RCCL writes its banner / warnings to fd 1, so everything
# except the final line is routed to stderr
_REAL_STDOUT = os.dup(1)
os.dup2(2, 1)

import numpy as np  # noqa: E402


def algorithmic_model(cal, ds, summary_dims):
    """algorithmic bytes / FLOPs per Jacobian+assembly pass."""
    nv = int(cal.views_accepted.sum()); nc = cal.num_corners
    na = int(cal.accl_accepted.sum()); ng = int(cal.gyro_accepted.sum())
    tr = cal.trajectory_
    n_so3, n_r3 = tr.GetNumSO3Knots(), tr.GetNumR3Knots()
    P, Pb, a, hb = summary_dims
    params = 32 * n_so3 + 24 * n_r3 + 32 * len(ds.points) + 250
    out = 8 * (Pb * (hb + 1) + Pb * a + a * a + P)
    bytes_ = dict(view=24 * nc + 12 * nv + params + out, accel=28 * na + params + out, gyro=28 * ng + 32 * n_so3 + out,
                  solve=2 * 8 * (Pb * (hb + 1) + (a + 1) * Pb + (a + 1) ** 2) + 8 * P)
    flops = dict(view=6.9e3 * nc, accel=7.6e3 * na, gyro=3.7e3 * ng, solve=float(Pb) * hb * hb + 2.0 * Pb * hb * (a + 1))
    bytes_["blocks"] = 24 * nc + 12 * nv + 28 * na + 28 * ng + params + out     # the fused launch: every input once, the output once
    flops["blocks"] = flops["view"] + flops["accel"] + flops["gyro"]
    return bytes_, flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C5-size single-GPU Jacobian-pass measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher the driver would have used -- one process per GPU through
        # torch.distributed.run on 127.0.0.1 (a free port), stdout (rank 0's JSON line) passed through
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
        os.write(_REAL_STDOUT, r.stdout)
        raise SystemExit(r.returncode)

    import torch
    import torch.distributed as dist
    from openimucameracalibrator_amd import synthetic, estimator as E

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # OICC_BENCH_FORCE_ALLREDUCE=1 exercises the RCCL hook with a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("OICC_BENCH_FORCE_ALLREDUCE") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
    # ---- workload: N = 1: C2 (the configuration the metric is quoted on); N > 1: C5, time-sharded (strong scaling) ----
    wl_name = "C2" if world == 1 else "C5"
    ds = synthetic.make_config(wl_name)
    cal = E.ImuCameraCalibrator(device=local_rank)
    tr = cal.trajectory_
    reduce_path = "none"
    native = use_dist and os.environ.get("OICC_BENCH_TORCH_ALLREDUCE") != "1"
    if use_dist and not native:   # torch staging path: share torch's stream so that the all-reduce is ordered with the kernels
        tr.SetStream(torch.cuda.current_stream().cuda_stream)
    for kv in filter(None, os.environ.get("OICC_BENCH_OPTS", "").split(",")):   # developer A/B: library options for the timed steps, e.g. OICC_BENCH_OPTS=device_lm=0
        tr.SetOption(kv.split("=")[0], float(kv.split("=")[1]))
    cal.BatchInitSpline(ds, shard=(rank, world) if world > 1 else None, owner_computes=world > 1)   # owner-computes exchange (round 4) where the native RCCL path is up; else the all-reduce of the whole buffer

    # Every step of the set-up is agreed on by all ranks (MIN all-reduce of a success flag), so that a rank that cannot bind RCCL,
    # build the communicator or run the exchange takes every other rank to the fallback with it instead of leaving them in a collective.
    def all_ok(ok):
        f = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return int(f[0]) == 1

    if native:
        # native path: the library calls ncclAllReduce (RCCL over xGMI) in place on its own stream; torch.distributed only
        # carries the 128-byte ncclUniqueId from rank 0 to the other ranks
        my_id = None
        try:
            my_id = tr.RcclUniqueId()          # probes the binding on every rank; only rank 0's id is used
        except Exception as e:
            sys.stderr.write("rank %d: native RCCL binding unavailable (%s)\n" % (rank, e))
        native = all_ok(my_id is not None)
        if native:
            idt = torch.tensor(list(my_id), dtype=torch.uint8, device="cuda")
            dist.broadcast(idt, src=0)
            ok = True
            try:
                tr.EnableRccl(world, rank, bytes(idt.cpu().tolist()))
            except Exception as e:
                ok = False
                sys.stderr.write("rank %d: ncclCommInitRank through the library failed (%s)\n" % (rank, e))
            native = all_ok(ok)
        if native:
            reduce_path = "rccl-native"
        else:
            sys.stderr.write("falling back to the torch.distributed staging hook\n")
            tr.SetStream(torch.cuda.current_stream().cuda_stream)
    if use_dist and not native:
        reduce_path = "torch-staging"
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        staging = {}

        def allreduce(ptr, count, strm):
            # fp64 SUM over ranks of the library's packed {J^T J band, arrow, J^T r, cost}
            # buffer: staged through a torch tensor so RCCL runs via torch.distributed.
            t = staging.get(count)
            if t is None:
                t = staging[count] = torch.empty(count, dtype=torch.float64, device="cuda")
            nbytes = count * 8
            assert hip.hipMemcpyAsync(t.data_ptr(), ptr, nbytes, 3, strm) == 0
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            assert hip.hipMemcpyAsync(ptr, t.data_ptr(), nbytes, 3, strm) == 0
        tr.SetAllReduce(allreduce)

    # The owner-computes exchange (ncclSend / ncclRecv of halo rows, ncclBroadcast per owner) has only ever run through the transport
    # hooks of the two-process tests: before the timed steps depend on it, one trial exchange that every rank must survive -- else all
    # ranks go back to the all-reduce of the whole packed buffer (oicc_set_shard(1, 0) switches the exchange off).
    assembly_path = "single"
    if world > 1:
        assembly_path = "allreduce"
        ok = reduce_path == "rccl-native"
        if ok:
            try:
                tr.TimeExchange(flags, repeats=-1)           # (a local question, nothing is sent)
            except Exception as e:
                ok = False
                sys.stderr.write("rank %d: owner-computes exchange not set up (%s)\n" % (rank, e))
        ok = all_ok(ok)
        if ok:
            try:
                tr.TimeExchange(flags, repeats=1)            # a collective: every rank runs it
            except Exception as e:
                ok = False
                sys.stderr.write("rank %d: trial exchange failed (%s)\n" % (rank, e))
            ok = all_ok(ok)
        if ok:
            assembly_path = "owner-computes exchange"
            # ... and one trial of the distributed linear solve (round 6: two all-gathers inside the solve) -- a rank that fails takes
            # every rank back to the gathered band + replicated solve (option distributed_solve = 0 is part of what the ranks agree on)
            ok2 = True
            try:
                tr.TimeLinearSolve(flags, repeats=1)
            except Exception as e:
                ok2 = False
                sys.stderr.write("rank %d: trial of the distributed solve failed (%s)\n" % (rank, e))
            if not all_ok(ok2):
                tr.SetOption("distributed_solve", 0)
        else:
            tr.SetShard(1, 0)

    n_blocks_local = cal.num_blocks
    blocks = torch.tensor([n_blocks_local, cal.num_corners], dtype=torch.int64, device="cuda")
    if use_dist:
        dist.all_reduce(blocks)
    n_blocks, n_corners = int(blocks[0]), int(blocks[1])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        tr.RunLmIterations(flags, args.warmup)
    barrier()
    t0 = time.perf_counter()
    tr.RunLmIterations(flags, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
    ms_per_step = 1e3 * dt / args.steps
    value = n_blocks * args.steps / dt

    # N > 1: the all-reduce of the packed system, timed by HIP events on the library's stream (a collective: every rank runs it)
    allreduce_ms, allreduce_bytes = None, None
    exchange_ms, exchange_bytes = None, None
    if use_dist:
        try:
            allreduce_ms, allreduce_bytes = tr.TimeAllReduce(flags, repeats=10)
        except Exception as e:
            sys.stderr.write("rank %d: all-reduce timing failed (%s)\n" % (rank, e))
        barrier()
        # the owner-computes exchange the steps above used, where it is set up (world > 1, native RCCL with send / recv): a collective,
        # so every rank first agrees on whether it can run it
        if assembly_path == "owner-computes exchange":
            try:
                exchange_ms, exchange_bytes = tr.TimeExchange(flags, repeats=10)
            except Exception as e:
                sys.stderr.write("rank %d: exchange timing failed (%s)\n" % (rank, e))
        barrier()

    # N > 1, round 6: the linear solve the timed steps ran -- on agreed shards the DISTRIBUTED cyclic reduction (every rank reduces its
    # own band range, gathered separators, replicated top system, gathered step) -- timed on every rank by HIP events (a collective)
    dist_solve = None
    if world > 1 and assembly_path == "owner-computes exchange":
        try:
            mine = tr.TimeLinearSolve(flags, repeats=10)
            info = tr.DistributedSolveInfo()
            v = torch.tensor([mine, float(info["blocks"]), float(info["ranks"])], dtype=torch.float64, device="cuda")
            allv = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(allv, v)
            dist_solve = dict(solve_ms_per_rank=[float(t[0]) for t in allv], blocks_per_rank=[int(t[1]) for t in allv], distributed=bool(info["ranks"] == world),
                              note="HIP events around 10 solves on every rank's stream, both gathers included; distributed = False: the band was gathered and every rank solved the whole system")
        except Exception as e:
            sys.stderr.write("rank %d: solve timing failed (%s)\n" % (rank, e))
        barrier()

    # N > 1: the SAME workload on ONE GPU, measured in the same run -- rank 0 builds the whole (unsharded) C5 problem and times the
    # same steps alone while the other ranks wait at the barrier below: the N = 1 point of this curve (the default N = 1 line is
    # C2, another workload, and must not be compared with the N > 1 lines)
    n1_same = None
    if world > 1 and rank == 0:
        try:
            c1 = E.ImuCameraCalibrator(device=local_rank).BatchInitSpline(ds)
            c1.trajectory_.RunLmIterations(flags, max(args.warmup, 1))
            torch.cuda.synchronize(); t1 = time.perf_counter(); c1.trajectory_.RunLmIterations(flags, args.steps); torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            n1_same = dict(n_gpus=1, ms_per_step=1e3 * d1 / args.steps, value=c1.num_blocks * args.steps / d1, unit="blocks/s", blocks=c1.num_blocks,
                           speedup_of_this_run=(d1 / args.steps) / (dt / args.steps),
                           note="the whole C5 problem on rank 0's GPU alone, same steps, timed after the sharded run while the other ranks wait")
            del c1
        except Exception as e:
            n1_same = {"error": str(e)[:200]}
    barrier()

    out = None
    if rank == 0:
        # ---- per-kernel HIP-event timings (library stream) and roofline ----------
        if use_dist:
            tr.SetAllReduce(None)
        pass_ms, kern_ms = tr.TimeJacobianPass(flags, repeats=20)
        solve_ms = tr.TimeLinearSolve(flags, repeats=20)
        lay = tr.GetTangentLayout(flags)
        P = lay["P"]; Pb = 3 * int((lay["so3"] >= 0).sum() + (lay["r3"] >= 0).sum()); a = P - Pb
        # full calibration wall clock (N = 1: the whole C2 problem), twice from the same start: with the solver configuration the
        # reference runs (impl.h:255-276: use_inner_iterations = true; bounds line search and projected gradient norm implicit in
        # Ceres) -- the configuration BASELINE's "GoPro9 full calib" is quoted on -- and with plain LM steps (no inner sweeps)
        summ = None
        if world == 1:
            def full_calibration(make, reference_options):
                c = make()
                if reference_options:
                    c.trajectory_.UseReferenceSolverOptions()
                t1 = time.perf_counter()
                s1 = c.trajectory_.Optimize(50, flags)                      # continuous_time_imu_to_camera_calibration.cc:215-216
                reproj = c.trajectory_.GetMeanReprojectionError()
                s2 = c.trajectory_.Optimize(10, E.CAM_LINE_DELAY)            # :217-221
                secs = time.perf_counter() - t1
                return dict(seconds=secs, stage1_iterations=s1["num_iterations"], stage1_seconds=s1["seconds_total"], stage2_iterations=s2["num_iterations"],
                            stage2_seconds=s2["seconds_total"], final_reproj_error_px=reproj, final_cost=s1["final_cost"],
                            inner_sweeps=s1["inner_sweeps"] + s2["inner_sweeps"], inner_lm_iterations=s1["inner_lm_iterations"] + s2["inner_lm_iterations"],
                            seconds_inner=s1["seconds_inner"] + s2["seconds_inner"], line_search_steps=s1["line_search_steps"] + s2["line_search_steps"],
                            seconds_jacobian=s1["seconds_jacobian"] + s2["seconds_jacobian"], seconds_residual=s1["seconds_residual"] + s2["seconds_residual"],
                            seconds_linear_solver=s1["seconds_linear_solver"] + s2["seconds_linear_solver"],
                            seconds_setup=s1["seconds_setup"] + s2["seconds_setup"]), s1, c   # seconds_setup: uploads, layout + buffers, tiles, inner-iteration plan (part of seconds)
            make_gpu = lambda: E.ImuCameraCalibrator(device=local_rank).BatchInitSpline(ds)
            full_calibration(make_gpu, True)                                   # warm-up: code objects of the inner-iteration kernels
            # (the median of three fresh calibrations each: one run is 5 ms of wall clock, a hiccup of the box is 10 % of it)
            def three(reference_options):   # (one problem alive at a time: the calibrator of a run is released before the next is built)
                runs = []
                for _ in range(3):
                    r_, s_, c_ = full_calibration(make_gpu, reference_options)
                    runs.append((r_, s_)); del c_
                mid = sorted(runs, key=lambda r: r[0]["seconds"])[1]
                return dict(mid[0], all_runs_seconds=[r[0]["seconds"] for r in runs]), mid[1]
            full_ref, summ = three(True)
            full_plain, _ = three(False)
            hb = summ["half_bandwidth"]
        else:   # no collective may run on rank 0 alone: half bandwidth from the tangent layout (span of the knots of one SO(3) window and the R^3 windows it overlaps)
            so3o, r3o = lay["so3"], lay["r3"]
            dts, dtr = ds.dt_so3, ds.dt_r3
            hb = 0
            for s_ in range(0, len(so3o) - 5, max(1, (len(so3o) - 5) // 2000)):
                r_ = int(s_ * dts / dtr)
                offs = [o for o in list(so3o[s_:s_ + 6]) + list(r3o[r_:r_ + 7]) if o >= 0]
                if offs:
                    hb = max(hb, max(offs) + 2 - min(offs))
        b_alg, f_alg = algorithmic_model(cal, ds, (P, Pb, a, hb))
        # Kernel groups of one LM iteration.  "blocks" is the fused residual+Jacobian+Gram launch the
        # metric is named after (ONE launch per pass: all_blocks_kernel<true>); view/accel/gyro are its
        # three residual families timed as stand-alone launches; "solve" is the 13-launch block cyclic
        # reduction through the pivot inverses (bcri_build_invert / invert / schur / backward), a dependent-latency chain, reported as a group.
        times = dict(blocks=pass_ms, view=kern_ms[0], accel=kern_ms[1], gyro=kern_ms[2], solve=solve_ms)
        names = dict(blocks="tile_kernel<true, false, 4, *> + slab_merge_kernel", view="tile_kernel<true, false, 4, *> (views only) + slab_merge_kernel",
                     accel="tile_kernel<true, false, 4, *> (accelerometer only) + slab_merge_kernel", gyro="tile_kernel<true, false, 4, *> (gyroscope only) + slab_merge_kernel",
                     solve="bcri_build_invert_kernel + bcri_invert_kernel + bcri_schur_kernel + bcri_backward_kernel")
        kernels = {k: dict(kernel=names[k], ms=times[k], alg_bytes=b_alg[k], alg_flops=f_alg[k],
                           hbm_GBps=b_alg[k] / (times[k] * 1e-3) / 1e9 if times[k] > 0 else 0.0,
                           fp64_TFLOPs=f_alg[k] / (times[k] * 1e-3) / 1e12 if times[k] > 0 else 0.0) for k in times}
        # The roofline object is for the dominant SINGLE kernel, the fused Jacobian/Gram launch.  SURVEY.md 8(d):
        # arithmetic intensity 150-300 FLOP/B, so the binding roof is fp64 (MFMA f64 = vector f64 = 78.6 TFLOP/s
        # dense on MI355X), not HBM; the HBM view of the same launch is given next to it.
        dom = "blocks"
        traffic = None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_C2.csv")))
        pmc = pmcs[-1] if pmcs else ""
        if world == 1 and pmc:     # per-pass FETCH_SIZE + WRITE_SIZE of the tile kernel and the slab merge from the committed rocprofv3 --pmc passes
            tot = 0.0
            for line in open(pmc):
                if "tile_kernel<true" in line or "slab_merge_kernel" in line or "all_blocks_kernel<true>" in line:
                    tot += float(line.rsplit(",", 1)[1]) * 1024.0
            traffic = tot or None
        roofline = dict(bound="fp64", kernel=names[dom], achieved=kernels[dom]["fp64_TFLOPs"], peak=78.6, unit="TFLOP/s",
                        frac=kernels[dom]["fp64_TFLOPs"] / 78.6, traffic=traffic,
                        traffic_source=("profiles/" + os.path.basename(pmc) + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel pair, not measured in this run)") if traffic else None,
                        hbm=dict(achieved=kernels[dom]["hbm_GBps"], peak=8000.0, unit="GB/s", frac=kernels[dom]["hbm_GBps"] / 8000.0),
                        binds="fp64 datapath of the SIMDs: on MI355X v_mfma_f64_16x16x4_f64 (64 cycles) runs on the vector fp64 lanes, so the Gram MFMAs and "
                              "the spline / Jacobian VALU work of a wave add up (scripts/micro/mfma_lds_rates.hip); peak = 78.6 TFLOP/s for either",
                        note="achieved = algorithmic FLOPs of SURVEY 8(d) (6.9 / 7.6 / 3.7 kFLOP per corner / accelerometer / gyroscope block) x the blocks of one pass "
                             "/ the pass time by HIP events (tile kernel + slab merge).  C2 is 8160 blocks in ~200 workgroups: one wave per SIMD at most, latency bound; "
                             "the same kernels on the C5-size problem (throughput bound) are in extra_c5_single_gpu.  traffic = FETCH_SIZE + WRITE_SIZE per pass from "
                             "the newest profiles/r*_pmc_hbm_C2.csv (%s)." % os.path.basename(pmc),
                        step_share=dict(blocks_ms=pass_ms, solve_ms=solve_ms, step_ms=ms_per_step,
                                        solve_group=dict(kernels=names["solve"], ms=solve_ms, fp64_frac=kernels["solve"]["fp64_TFLOPs"] / 78.6,
                                                         why="dependent chain: ceil(log2 n)+1 inversions of 64 x 64 pivot blocks (pivot recurrences at ~200 cycles per column), "
                                                             "each followed by a Schur launch, and a back substitution launch per tree level: three kernel boundaries per level")),
                        kernels=kernels)
        out = {
            "metric": "residual+Jacobian blocks/sec; wall-clock per LM iter, GoPro9 full calib",
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seed 20241115)",
            "config": {"workload": ("C2 GoPro9 Division-Undistortion 960x540, %d views, %d corners, %d IMU blocks, dt_r3/so3=0.1/0.05 s" if world == 1 else
                                    "C5 synthetic, %d views, %d corners, %d IMU blocks, dt_r3/so3=0.1/0.05 s, " + "%d time shards (strong scaling), %s" % (
                                        world, "owner-computes exchange of JtJ/Jtr (halo rows + gather per owner + corner all-reduce)" if assembly_path == "owner-computes exchange" else "all-reduce of JtJ/Jtr"))
                                   % (ds.num_views, n_corners, n_blocks - ds.num_views),
                       "blocks": n_blocks, "corners": n_corners, "tangent_dim": P, "flags": "SPLINE|T_I_C|GRAVITY_DIR", "allreduce": reduce_path, "assembly": assembly_path,
                       "step": ("one LM iteration under device-side control (LmCtl): block-cyclic-reduction solve, retraction, ONE Jacobian+assembly pass at the candidate "
                                "(its corner holds the candidate's cost, its merge the gradient norm), trust-region decision on the device; the host enqueues and polls a pinned word one iteration behind"
                                if world == 1 and os.environ.get("OICC_BENCH_OPTS", "").find("device_lm=0") < 0 else
                                "one LM iteration, host-driven: Jacobian+assembly, [reduction], block-cyclic-reduction solve, retraction, [broadcast], cost pass, one host read-back")},
            "corners_per_s": n_corners * args.steps / dt,
            "jacobian_pass_ms": pass_ms,
            "roofline": roofline,
        }
        if n1_same is not None:
            out["n1_same_workload"] = n1_same
            out["scaling_note"] = "no multi-GPU node was available while this was built: the N > 1 path has run on hardware only through the driver; the curve is unmeasured until SCALE_rNN.json exists"
        if use_dist:   # where a rank's iteration goes: its shard's Jacobian pass and the replicated solve alone (HIP events, no collective), the rest = all-reduce of the packed system + cost, broadcast, cost pass, retraction
            out["per_rank"] = dict(jacobian_pass_ms=pass_ms, solve_ms=(max(dist_solve["solve_ms_per_rank"]) if dist_solve else solve_ms), solve_ms_whole_system_on_one_rank=solve_ms, distributed_solve=dist_solve, step_ms=ms_per_step,
                                   allreduce_ms=allreduce_ms, allreduce_bytes=allreduce_bytes, allreduce_timing="HIP events around 10 all-reduces of the packed normal equations on the library's stream (%s)" % reduce_path,
                                   exchange_ms=exchange_ms, exchange_bytes=exchange_bytes,
                                   exchange=("owner-computes (what the timed steps ran): halo rows to their owners (ncclSend / ncclRecv), all-gather of the diagonal and the gradient (the band rows stay with their owners: distributed solve), all-reduce of the arrow corner" if (dist_solve and dist_solve["distributed"]) else "owner-computes (what the timed steps ran): halo rows to their owners (ncclSend / ncclRecv), gather of the owned band ranges, all-reduce of the arrow corner") if exchange_ms is not None else "all-reduce of the whole packed buffer (owner-computes exchange not available on this path)",
                                   roofline_per_rank=dict(fp64_frac=kernels["blocks"]["fp64_TFLOPs"] / 78.6, hbm_frac=kernels["blocks"]["hbm_GBps"] / 8000.0, note="this rank's shard: algorithmic FLOPs / bytes of its Jacobian pass over the pass time by HIP events"),
                                   rest_ms=max(0.0, ms_per_step - pass_ms - (max(dist_solve["solve_ms_per_rank"]) if dist_solve else solve_ms) - ((exchange_ms if exchange_ms is not None else allreduce_ms) or 0.0)))
        if summ is not None:
            out["full_calibration"] = dict(full_ref, solver_options="reference: inner iterations + bounds line search + projected gradient norm (impl.h:255-276)",
                                           plain_lm=dict(full_plain, solver_options="plain Levenberg-Marquardt steps (no inner sweeps)"))
        # ---- extra: C5-size Jacobian pass on one GPU (before the CPU baselines: with their 128 OpenMP threads in the process the same
        # set-up measured 12 ms instead of 6) --------------------------------
        if not args.no_extra and world == 1:
            try:
                ds5 = synthetic.make_config("C5")
                c5 = E.ImuCameraCalibrator(device=local_rank).BatchInitSpline(ds5)
                p5, k5 = c5.trajectory_.TimeJacobianPass(flags, repeats=5)
                s5 = c5.trajectory_.TimeLinearSolve(flags, repeats=5)
                # one full LM iteration and one inner sweep at C5 size (the reference's solver options: every sweep visits all 30 011 parameter blocks)
                c5.trajectory_.RunLmIterations(flags, 2)
                torch.cuda.synchronize(); t1 = time.perf_counter(); c5.trajectory_.RunLmIterations(flags, 5); torch.cuda.synchronize()
                lm5 = 1e3 * (time.perf_counter() - t1) / 5
                c5r = E.ImuCameraCalibrator(device=local_rank).BatchInitSpline(ds5)
                c5r.trajectory_.UseReferenceSolverOptions()
                s5r = c5r.trajectory_.Optimize(3, flags)
                c5p = E.ImuCameraCalibrator(device=local_rank).BatchInitSpline(ds5)
                s5p = c5p.trajectory_.Optimize(1, flags)                     # plain LM: set-up without the inner-iteration plan
                # the FULL C5 calibration with the reference's solver options (stage 1 + stage 2, as full_calibration above)
                # (three fresh problems one after the other: the first large uploads / allocations of a process cost the runtime several
                # times what later ones do -- round 6 measured 20 / 12 / 3 ms for the same 47 MB -- so the line carries the median run and all three)
                c5_blocks, c5_corners, c5_accl = c5.num_blocks, c5.num_corners, int(c5.accl_accepted.sum())
                del c5, c5r, c5p      # (one data set at a time, as a calibration run has it: with four C5 problems alive every new one maps ~0.6 GB of fresh device memory -- 13 ms of set-up against 6)
                runs5 = []
                for _ in range(3):
                    c5f = E.ImuCameraCalibrator(device=local_rank).BatchInitSpline(ds5)
                    c5f.trajectory_.UseReferenceSolverOptions()
                    if os.environ.get("OICC_BENCH_VERBOSE_C5") == "1": c5f.trajectory_.SetOption("verbose", 2)   # (developer: the library's own set-up timers)
                    t1 = time.perf_counter()
                    f1 = c5f.trajectory_.Optimize(50, flags); rp5 = c5f.trajectory_.GetMeanReprojectionError(); f2 = c5f.trajectory_.Optimize(10, E.CAM_LINE_DELAY)
                    runs5.append(dict(seconds=time.perf_counter() - t1, stage1_iterations=f1["num_iterations"], stage2_iterations=f2["num_iterations"], inner_sweeps=f1["inner_sweeps"] + f2["inner_sweeps"],
                                      seconds_inner=f1["seconds_inner"] + f2["seconds_inner"], seconds_setup=f1["seconds_setup"] + f2["seconds_setup"], final_cost=f1["final_cost"], final_reproj_error_px=rp5))
                    del c5f
                full5 = dict(sorted(runs5, key=lambda r: r["seconds"])[1], all_runs_seconds=[r["seconds"] for r in runs5], all_runs_seconds_setup=[r["seconds_setup"] for r in runs5],
                             note="median of three fresh problems in this process, in the order run: all_runs_*")
                setup_ref = sorted(r["seconds_setup"] for r in runs5)[1]
                plain_setups = []
                for _ in range(3):
                    c5q = E.ImuCameraCalibrator(device=local_rank).BatchInitSpline(ds5)
                    plain_setups.append(c5q.trajectory_.Optimize(1, flags)["seconds_setup"]); del c5q
                out["extra_c5_single_gpu"] = dict(blocks=c5_blocks, corners=c5_corners, jacobian_pass_ms=p5, linear_solve_ms=s5, kernel="tile_kernel<true, false, 4, true> (chains of 8 tiles) + slab_merge_kernel",
                                                  lm_step_ms=lm5, inner_sweep_ms=1e3 * full5["seconds_inner"] / max(full5["inner_sweeps"], 1), inner_sweeps_timed=full5["inner_sweeps"],
                                                  inner_sweep_ms_first_use=1e3 * s5r["seconds_inner"] / max(s5r["inner_sweeps"], 1),   # (the first reference-option C5 solve of the process: code objects of the wave / shared-block kernels are loaded on first use -- 3.8 to 8.9 ms per sweep seen)
                                                  setup_ms_reference_options=1e3 * setup_ref, setup_ms_plain_lm=1e3 * sorted(plain_setups)[1],
                                                  setup_note="seconds_setup of the summary (uploads, layout + buffers, tiles, inner-iteration plan; both stages for the reference options), median of three fresh problems with no other C5 problem alive; with the earlier problems of this process still alive: %.2f / %.2f ms" % (1e3 * s5r["seconds_setup"], 1e3 * s5p["seconds_setup"]),
                                                  full_calibration_reference_options=full5,
                                                  fp64_frac_of_78p6=(6.9e3 * c5_corners + 11.3e3 * c5_accl) / (p5 * 1e-3) / 78.6e12,
                                                  kernel_ms=dict(view=k5[0], accel=k5[1], gyro=k5[2]),
                                                  blocks_per_s_jacobian_pass=c5_blocks / (p5 * 1e-3),
                                                  fp64_TFLOPs=(6.9e3 * c5_corners + 11.3e3 * c5_accl) / (p5 * 1e-3) / 1e12)
            except Exception as e:  # the extra must never break the bench line
                out["extra_c5_single_gpu"] = {"error": str(e)[:200]}
        # ---- CPU baseline: the oracle (CPU restatement of the Ceres path) ----------
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_backend
            ob = oracle_backend.load()
            ob.raw.oicc_oracle_num_threads.restype = ctypes.c_int
            cores = int(ob.raw.oicc_oracle_num_threads())
            iters = 3
            scan = {}
            for nth in sorted({8, 16, 32, 64, cores}):          # the CPU assembly keeps one partial system per thread: more threads is not always faster
                if nth > cores:
                    continue
                ccal = E.ImuCameraCalibrator(backend=ob).BatchInitSpline(ds)
                ccal.trajectory_.SetOption("num_threads", nth)
                t1 = time.perf_counter()
                cs = ccal.trajectory_.Optimize(iters if nth == min(16, cores) else 1, flags)
                scan[nth] = (time.perf_counter() - t1) / max(cs["num_iterations"], 1)
            nth_best = min(scan, key=scan.get)
            ccal = E.ImuCameraCalibrator(backend=ob).BatchInitSpline(ds)
            ccal.trajectory_.SetOption("num_threads", nth_best)
            t1 = time.perf_counter()
            cs = ccal.trajectory_.Optimize(iters, flags)
            cdt = time.perf_counter() - t1
            out["cpu_baseline"] = dict(value=n_blocks * cs["num_iterations"] / cdt, unit="blocks/s", cores=nth_best, host_cores=os.cpu_count(), openmp_max_threads=cores, kind="port",
                                       sample="first %d LM iterations of the same C2 problem with the CPU restatement of the Ceres path (forward-mode Jet autodiff in "
                                              "strides of 4, OpenMP over residual blocks, band+arrow Cholesky), best of %s threads: %.2f s" % (
                                                  cs["num_iterations"], "/".join(str(k) for k in sorted(scan)), cdt),
                                       ms_per_lm_iteration=1e3 * cdt / max(cs["num_iterations"], 1),
                                       thread_scan_ms_per_iteration={str(k): 1e3 * v for k, v in sorted(scan.items())})
            # second CPU baseline (SURVEY 8d ii): the same CPU LM loop with the closed-form Jacobians of the device kernels
            # compiled for the host (oracle/cpu_analytic.hpp) instead of forward-mode Jets, i.e. a CPU code without autodiff
            best = None
            for nth in sorted({8, 16, 32, 64, cores}):
                if nth > cores:
                    continue
                acal = E.ImuCameraCalibrator(backend=ob).BatchInitSpline(ds)
                acal.trajectory_.SetOption("analytic_jacobians", 1); acal.trajectory_.SetOption("num_threads", nth)
                acal.trajectory_.Optimize(1, flags)            # warm-up (thread pool, page faults)
                acal = E.ImuCameraCalibrator(backend=ob).BatchInitSpline(ds)
                acal.trajectory_.SetOption("analytic_jacobians", 1); acal.trajectory_.SetOption("num_threads", nth)
                t1 = time.perf_counter()
                as_ = acal.trajectory_.Optimize(iters, flags)
                adt = time.perf_counter() - t1
                if best is None or adt / as_["num_iterations"] < best[0] / best[1]:
                    best = (adt, as_["num_iterations"], nth)
            adt, ait, nth = best
            out["cpu_baseline"]["analytic"] = dict(value=n_blocks * ait / adt, unit="blocks/s", cores=nth, kind="port",
                                                   sample="the same %d LM iterations with analytic Jacobians (formulas of the device kernels on the host, OpenMP, "
                                                          "best of 8/16/32/64/%d threads): %.3f s" % (ait, cores, adt),
                                                   ms_per_lm_iteration=1e3 * adt / max(ait, 1))
            # the FULL calibration on the host cores with the reference's solver options, same start as full_calibration above:
            # the CPU restatement with Jets (the reference's cost profile) and with the closed-form Jacobians
            def cpu_full(analytic, nth_):
                def make():
                    c_ = E.ImuCameraCalibrator(backend=ob).BatchInitSpline(ds)
                    c_.trajectory_.SetOption("num_threads", nth_); c_.trajectory_.SetOption("analytic_jacobians", 1 if analytic else 0)
                    return c_
                r_, _, _ = full_calibration(make, True)
                return r_
            cj = cpu_full(False, nth_best); ca = cpu_full(True, nth)
            out["cpu_baseline"]["full_calibration_reference_options"] = dict(
                jets=dict(seconds=cj["seconds"], cores=nth_best, stage1_iterations=cj["stage1_iterations"], inner_sweeps=cj["inner_sweeps"], final_cost=cj["final_cost"]),
                analytic=dict(seconds=ca["seconds"], cores=nth, stage1_iterations=ca["stage1_iterations"], inner_sweeps=ca["inner_sweeps"], final_cost=ca["final_cost"]),
                gpu_seconds=full_ref["seconds"], speedup_vs_jets=cj["seconds"] / full_ref["seconds"], speedup_vs_analytic=ca["seconds"] / full_ref["seconds"], kind="port")
        # ---- extra: spline-error-weighting pre-stage, device vs the numpy oracle ----
        if not args.no_extra and world == 1:
            try:
                from openimucameracalibrator_amd import sew
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import sew_oracle
                rng = np.random.default_rng(5)
                n5 = 200000
                big = np.cumsum(rng.standard_normal((3, n5)), axis=1) * 0.01 + 0.05 * rng.standard_normal((3, n5))
                res = {}
                for nm, sig, tt in (("C2_accel", ds.accel.T.copy(), ds.imu_t_s), ("C5_size_series", big, np.arange(n5) / 200.0)):
                    sew.knot_spacing_and_variance(sig, tt, 0.96, min_dt=0.01, max_dt=0.15)   # warm-up (hipFFT plan, code load)
                    t1 = time.perf_counter(); g = sew.knot_spacing_and_variance(sig, tt, 0.96, min_dt=0.01, max_dt=0.15); tg = time.perf_counter() - t1
                    t1 = time.perf_counter(); o = sew_oracle.knot_spacing_and_variance(sig, tt, 0.96, min_dt=0.01, max_dt=0.15); to = time.perf_counter() - t1
                    res[nm] = dict(samples=int(sig.shape[1]), device_ms=1e3 * tg, numpy_oracle_ms=1e3 * to, dt=g[0], dt_oracle=o[0])
                out["extra_sew_prestage"] = res
                # gyro-to-camera initialisation on the C2 telemetry (view orientations of the data set), device vs numpy oracle
                from openimucameracalibrator_amd import rotation_init as RI
                import rotation_init_oracle as RO
                q_cw = ds.view_q_wc * np.array([-1.0, -1.0, -1.0, 1.0])
                RI.estimate_camera_imu_rotation(ds.view_t_s, q_cw, ds.imu_t_s, ds.gyro, 0.005)
                t1 = time.perf_counter(); rg = RI.estimate_camera_imu_rotation(ds.view_t_s, q_cw, ds.imu_t_s, ds.gyro, 0.005); tg = time.perf_counter() - t1
                t1 = time.perf_counter(); ro = RO.estimate_imu_to_camera_rotation(ds.view_t_s, q_cw, ds.imu_t_s, ds.gyro, 0.005, True); to = time.perf_counter() - t1
                out["extra_rotation_init"] = dict(imu_samples=int(len(ds.imu_t_s)), device_ms=1e3 * tg, numpy_oracle_ms=1e3 * to,
                                                   time_offset=rg["time_offset"], time_offset_oracle=ro[1], iterations=rg["iterations"])
            except Exception as e:
                out["extra_sew_prestage"] = {"error": str(e)[:200]}
        # ---- extra: view bundle adjustment: BASELINE config 0 (pinhole intrinsics, 30 frames) through the
        # three RunCalibration stages, and the per-view pose refinement of a 2000-frame corner file in one launch; the oracle
        # (CPU restatement of Theia's bundle adjuster, Jets, one core) timed beside it
        if not args.no_extra and world == 1:
            try:
                from openimucameracalibrator_amd import camera_calibrator as CC
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_backend
                ob_ba = oracle_backend.load_ba()
                dsc = CC.make_calibration_dataset("pinhole", num_views=30, corners_per_view=40)

                def run_calibration(backend):
                    cal_ = CC.CameraCalibrator("PINHOLE", device=local_rank, backend=backend)
                    cal_.SetScenePoints(dsc["points"])
                    for v in range(len(dsc["pose_init"])):
                        vid = cal_.AddView(CC.angle_axis_to_rotation(dsc["pose_init"][v, 3:]), dsc["pose_init"][v, :3], dsc["intrinsics"][0] * 1.05, 0.0,
                                           dsc["width"], dsc["height"], 0.1 * v)
                        for c in range(dsc["corner_offset"][v], dsc["corner_offset"][v + 1]):
                            cal_.AddObservation(vid, dsc["point_ids"][c], dsc["uv"][c])
                    t1_ = time.perf_counter(); ok_ = cal_.RunCalibration(); return cal_, ok_, time.perf_counter() - t1_
                run_calibration(None)                                  # warm-up (code load)
                cg, okg, tg = run_calibration(None)
                cc_, okc, tc = run_calibration(ob_ba)
                dsp = CC.make_calibration_dataset("gopro6_fisheye", num_views=2000, corners_per_view=40, pose_noise=(0.01, 0.01))

                def refine(backend):
                    ba_ = CC.ViewBundleAdjuster(device=local_rank, backend=backend)
                    ba_.SetCamera(dsp["model"], dsp["intrinsics"]); ba_.SetScenePoints(dsp["points"])
                    ba_.SetViews(dsp["pose_init"], dsp["corner_offset"], dsp["uv"], dsp["point_ids"])
                    ba_.ViewReprojectionErrors()                       # uploads done before the timed call
                    t1_ = time.perf_counter(); it_, _ = ba_.OptimizeViews(50); return ba_.GetPoses(), it_, time.perf_counter() - t1_
                refine(None)
                pg, itg, tpg = refine(None)
                pc, itc, tpc = refine(ob_ba)
                out["extra_view_bundle_adjustment"] = dict(
                    config0_calibration=dict(views=30, observations=int(dsc["corner_offset"][-1]), device_ms=1e3 * tg, oracle_ms=1e3 * tc, oracle_cores=1,
                                             lm_iterations=[s_["num_iterations"] for s_ in cg.summaries],
                                             focal_length=float(cg.GetIntrinsics()[0]), focal_length_oracle=float(cc_.GetIntrinsics()[0]),
                                             focal_length_truth=float(dsc["intrinsics"][0])),
                    pose_refinement=dict(views=2000, observations=int(dsp["corner_offset"][-1]), device_ms=1e3 * tpg, oracle_ms=1e3 * tpc, oracle_cores=1,
                                         lm_iterations_total=int(itg.sum()), same_iteration_counts=bool(np.array_equal(itg, itc)),
                                         max_pose_difference=float(np.abs(pg - pc).max())))
            except Exception as e:
                out["extra_view_bundle_adjustment"] = {"error": str(e)[:200]}
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()

"""Worker of test_two_processes_on_one_gpu_reduce_through_the_hook: rank r of WORLD_SIZE processes, ALL on GPU 0, holds the
r-th time shard of a problem and runs oicc_optimize with an all-reduce hook that stages through host memory and gloo
(RCCL refuses two ranks on one device; the hook is the product path under test, the transport is not).
usage: python mp_shard_worker.py <cfg> <flags> <iterations> <bounds_line_search> <out.json> [inner_iterations] [owner_computes]   (RANK / WORLD_SIZE / MASTER_* from the env)
owner_computes = 1: the round-4 exchange (oicc_set_shard): halo rows to their owners, gather of the owned band ranges, all-reduce of
the arrow corner only -- through the transport hooks (oicc_set_exchange: gloo send / recv / broadcast staged through host memory).
With inner_iterations = 1 every rank also builds the WHOLE problem on the device and hands it to its shard as the source of the
inner-iteration sweeps (oicc_set_inner_iteration_source): the reference's solver configuration on time-sharded ranks."""
import ctypes, json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E


def main():
    cfg, flags, iters, ls, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    inner = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    owner = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    calls = {"n": 0, "doubles": 0, "max": 0}

    def allreduce(ptr, count, strm):
        assert hip.hipStreamSynchronize(strm) == 0
        buf = torch.empty(count, dtype=torch.float64)
        assert hip.hipMemcpy(buf.data_ptr(), ptr, count * 8, 2) == 0      # device -> host
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        assert hip.hipMemcpy(ptr, buf.data_ptr(), count * 8, 1) == 0      # host -> device
        calls["n"] += 1; calls["doubles"] += count; calls["max"] = max(calls["max"], count)

    xch = {"sendrecv": 0, "broadcast": 0, "doubles": 0, "max_broadcast": 0}

    def exchange(op, sp, sc, rp, rc, peer, strm):
        assert hip.hipStreamSynchronize(strm) == 0
        if op == 0:      # send sc doubles to `peer`, receive rc doubles from it: the lower rank of the pair sends first
            sb = torch.empty(max(sc, 1), dtype=torch.float64); rb = torch.empty(max(rc, 1), dtype=torch.float64)
            if sc: assert hip.hipMemcpy(sb.data_ptr(), sp, sc * 8, 2) == 0
            for turn in (0, 1):
                if (turn == 0) == (rank < peer):
                    if sc: dist.send(sb[:sc], dst=peer)
                elif rc: dist.recv(rb[:rc], src=peer)
            if rc: assert hip.hipMemcpy(rp, rb.data_ptr(), rc * 8, 1) == 0
            xch["sendrecv"] += 1; xch["doubles"] += sc + rc
        else:            # `peer` holds sc doubles at sp; everybody receives them in place
            b = torch.empty(sc, dtype=torch.float64)
            if rank == peer: assert hip.hipMemcpy(b.data_ptr(), sp, sc * 8, 2) == 0
            dist.broadcast(b, src=peer)
            if rank != peer: assert hip.hipMemcpy(rp, b.data_ptr(), sc * 8, 1) == 0
            xch["broadcast"] += 1; xch["doubles"] += sc if rank != peer else 0; xch["max_broadcast"] = max(xch["max_broadcast"], sc)

    ds = synthetic.make_config(cfg)
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds, shard=(rank, world) if world > 1 else None, owner_computes=bool(owner))
    tr = cal.trajectory_
    tr.SetOption("bounds_line_search", ls); tr.SetOption("inner_iterations", inner)
    if os.environ.get("OICC_TEST_RADIUS") is not None: tr.SetOption("initial_trust_region_radius", float(os.environ["OICC_TEST_RADIUS"]))
    if os.environ.get("OICC_TEST_DISTRIBUTED_SOLVE") is not None: tr.SetOption("distributed_solve", int(os.environ["OICC_TEST_DISTRIBUTED_SOLVE"]))
    shared_launch = os.environ.get("OICC_TEST_SHARED_LAUNCH_SLOTS")     # (tests: the shared blocks of the sweeps as a sequence of launches at any size)
    if shared_launch is not None: tr.SetOption("inner_shared_launch_slots", int(shared_launch))
    if world > 1:
        tr.SetAllReduce(allreduce)
        if owner: tr.SetExchange(exchange)
        if inner:
            whole = E.ImuCameraCalibrator().BatchInitSpline(ds)
            # (the plan of the sweeps belongs to the source problem; oicc_optimize forwards the shard's plan options to it: round 6)
            tr.SetInnerIterationSource(whole.trajectory_)
    s = tr.Optimize(iters, flags)
    it = tr.GetIterations()
    info = (ctypes.c_int64 * 4)()
    tr._b.lib.oicc_debug_dist_solve_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    assert tr._b.lib.oicc_debug_dist_solve_info(tr._h, info) == 0
    res = dict(rejected=int(s["num_unsuccessful_steps"]), dist_solves=int(info[0]), dist_first_block=int(info[1]), dist_blocks=int(info[2]), dist_ranks=int(info[3]), band_row_doubles=int(s["half_bandwidth"]) + 1 + int(s["arrow_dim"]) + 1, band_dim=int(s["band_dim"]), rank=rank, blocks=cal.num_blocks, iterations=[dict(cost=i["cost"], ok=i["step_is_successful"], gmax=i["gradient_max_norm"]) for i in it],
               final_cost=s["final_cost"], inner_sweeps=s["inner_sweeps"], inner_lm_iterations=s["inner_lm_iterations"], T_i_c=[float(v) for v in tr.GetT_i_c()], hook_calls=calls["n"], hook_doubles=calls["doubles"], hook_max_doubles=calls["max"], P=int(s["num_parameters_tangent"]), exchange=xch)
    json.dump(res, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/bin/bash
# cyclic reduction through the pivot inverses (solver_algorithm 4 = the automatic choice): solver tests, per-kernel trace, phase clocks
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "parallel_solvers or bias_and_intrinsics or wide_borders or deep_tree or cyclic_reduction_on_c3 or very_short" 2>&1 | tail -6
python scripts/time_solver.py C2 2,4 2>&1 | grep "solve ms"
python scripts/time_solver.py C3 2,4 2>&1 | grep "solve ms"
python scripts/time_solver.py C4 2,4 2>&1 | grep "solve ms"
python scripts/prof_solver.py 4 2>&1 | tail -1
bash scripts/trace_solver.sh C2 4 2>&1 | tail -24

#!/bin/bash
# round 5, third GPU call: POINTS + inner iterations, the C1 line-delay fork, C5 set-up breakdown, the full suite
TAG=${1:-r05c}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python scripts/dbg_fork.py 40 > $O/fork.log 2>&1; tail -30 $O/fork.log
timeout 300 python scripts/time_setup.py C5 > $O/setup_c5.log 2>&1; tail -12 $O/setup_c5.log

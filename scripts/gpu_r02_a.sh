#!/bin/bash
# first GPU contact of round 2: parity suite, block cycle probes, C2/C5 pass timings per assembly mode
R=$PWD; O=$R/gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python scripts/time_modes.py > $O/time_modes.log 2>&1
cat $O/time_modes.log

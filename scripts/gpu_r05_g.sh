#!/bin/bash
# round 5: the wave-per-block inner kernel: its tests, the inner / POINTS / sharded tests again (kernel modes changed), sweep timing C5 / C4 / C2
TAG=${1:-r05g}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 900 -k "wave or inner or points or processes or out_of_time" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
for c in C5 C4 C2; do timeout 300 python scripts/time_wave.py $c 2 2>&1 | tail -3 | tee -a $O/wave.log; done

#!/bin/bash
# round 5: wave-per-block kernel, second look: the corrected test, run-to-run reproducibility of the case that diverged, kernel stats of a C5 sweep
TAG=${1:-r05h}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 900 -k "wave" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python scripts/dbg_wave_repro.py 2>&1 | tee $O/repro.log | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o run -- python $R/scripts/time_wave.py C5 1 > $O/stats_c5.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -8 $O/stats_c5/run_kernel_stats.csv | cut -c1-100,120-220; tail -3 $O/stats_c5.log

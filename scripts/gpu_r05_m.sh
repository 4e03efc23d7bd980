#!/bin/bash
# round 5, closing measurement: the GPU suite, smoke, the default bench line, kernel stats of the reference-option solves (C2, C5: the inner sweeps)
TAG=${1:-r05m}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inner -o run -- python $R/scripts/time_inner.py C2 3 > $O/stats_inner.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inner5 -o run -- python $R/scripts/time_inner.py C5 2 > $O/stats_inner5.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -2 $O/stats_inner.log; tail -2 $O/stats_inner5.log
head -12 $(find $O/stats_inner5 -name "*kernel_stats.csv" | head -1) | cut -c1-200

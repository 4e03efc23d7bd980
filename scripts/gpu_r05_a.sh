#!/bin/bash
# round 5, first GPU call: the whole GPU suite, the bench line, A/B of the LM step against the round-4 library on this box
# (scratch_bin/liboicc_r04.so, scripts/build_variant.sh r04 b243f28) and against the host-driven loop, inner sweeps, kernel stats.
TAG=${1:-r05a}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -n 3 --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra"
for rep in 1 2; do
  echo "cur:      $($B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $O/ab_step.log
  echo "host lm:  $(OICC_BENCH_OPTS=device_lm=0 $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $O/ab_step.log
  echo "r04 lib:  $(OICC_DEV_LIB=$R/scratch_bin/liboicc_r04.so $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $O/ab_step.log
done
for rep in 1 2; do
  echo "inner cur: $(python scripts/time_inner.py C2 3 2>&1 | tail -1)" | tee -a $O/ab_inner.log
  echo "inner r04: $(OICC_DEV_LIB=$R/scratch_bin/liboicc_r04.so python scripts/time_inner.py C2 3 2>&1 | tail -1)" | tee -a $O/ab_inner.log
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inner -o run -- python $R/scripts/time_inner.py C2 3 > $O/stats_inner.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -12 $f | cut -c1-150; done

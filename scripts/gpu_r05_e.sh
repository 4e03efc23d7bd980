#!/bin/bash
# round 5: the decision folded into the next build kernel + the flattened merge table: full GPU suite, step A/B, kernel stats of the step
TAG=${1:-r05e}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra"
for rep in 1 2 3; do
  echo "device lm: $($B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["jacobian_pass_ms"])')" | tee -a $O/ab_step.log
  echo "host lm:   $(OICC_BENCH_OPTS=device_lm=0 $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["jacobian_pass_ms"])')" | tee -a $O/ab_step.log
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -16 $O/stats/run_kernel_stats.csv | cut -c1-110,150-230

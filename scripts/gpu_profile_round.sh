#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_profile_round.sh TAG
# bench line + rocprofv3 kernel stats + HBM counters (separate --pmc passes) for C2 (the bench step) and for C5-size
# Jacobian passes (full passes only), and the reference-option solve with its inner sweeps (C2), all under gpurun_out/TAG
TAG=${1:-r03x}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra"
PASS5="python $R/scripts/prof_pass.py C5 10 full"
INNER="python $R/scripts/time_inner.py C2 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $BENCH > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $BENCH > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $BENCH > $O/pmc_write.log 2>&1
(cd $R && $PASS5) > $O/pass_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -o run -- $PASS5 > $O/stats5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc5_fetch -o run -- $PASS5 > $O/pmc5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -o run -- $PASS5 > $O/pmc5_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inner -o run -- $INNER > $O/stats_inner.log 2>&1
cd $R
python scripts/summarize_pmc.py $O/pmc_hbm_C2.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $BENCH ; MI355X, C2" $O/pmc_fetch $O/pmc_write
python scripts/summarize_pmc.py $O/pmc_hbm_C5.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $PASS5 ; MI355X, C5-size Jacobian + assembly passes" $O/pmc5_fetch $O/pmc5_write
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
tail -c 400 $O/bench.json; cat $O/pass_c5.log

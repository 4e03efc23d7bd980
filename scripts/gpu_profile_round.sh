#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_profile_round.sh TAG
# bench line + rocprofv3 kernel stats + HBM counters (separate --pmc passes), all under gpurun_out/TAG
TAG=${1:-r01x}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $BENCH > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $BENCH > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $BENCH > $O/pmc_write.log 2>&1
cd $R
python scripts/summarize_pmc.py $O/pmc_hbm.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $BENCH ; MI355X, C2" $O/pmc_fetch $O/pmc_write
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
tail -c 600 $O/bench.json

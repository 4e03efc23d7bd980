#!/bin/bash
# Jacobian + assembly pass of C2 / C5 under rocprofv3: kernel stats and (separate passes) HBM counters.  bash scripts/gpu_pass_profile.sh TAG [CFG ...]
export TMPDIR=/tmp
TAG=${1:-r04x}; shift; CFGS=${@:-C2 C5}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
for CFG in $CFGS; do
  PASS="python $R/scripts/prof_pass.py $CFG 10 full"
  (cd $R && python scripts/prof_pass.py $CFG 20) > $O/pass_$CFG.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$CFG -o run -- $PASS > $O/stats_$CFG.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$CFG -o run -- $PASS > $O/pmc_fetch_$CFG.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$CFG -o run -- $PASS > $O/pmc_write_$CFG.log 2>&1
  (cd $R && python scripts/summarize_pmc.py $O/pmc_hbm_pass_$CFG.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $PASS ; MI355X, $CFG Jacobian + assembly passes" $O/pmc_fetch_$CFG $O/pmc_write_$CFG)
  F=$(find $O/stats_$CFG -name "*kernel_stats.csv" | head -1); cp $F $O/kernel_stats_pass_$CFG.csv
  tail -1 $O/pass_$CFG.log; head -4 $O/kernel_stats_pass_$CFG.csv | cut -c1-160; grep -v "^#" $O/pmc_hbm_pass_$CFG.csv | grep "tile_kernel<true\|slab_merge" | cut -c1-200
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete

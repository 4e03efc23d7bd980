#!/bin/bash
# round 5, the measurement call: default bench line, kernel stats + HBM counters of the C2 step / the C5 pass / the inner sweeps
# (gpu_profile_round.sh), SQ counters of the tile kernel, resources of every kernel, the step A/B against the host-driven loop
TAG=${1:-r05d}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
bash scripts/gpu_tile_sq.sh $TAG > $O/tile_sq.log 2>&1; tail -3 $O/tile_sq.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra"
for rep in 1 2 3; do
  echo "device lm: $($B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $O/ab_step.log
  echo "host lm:   $(OICC_BENCH_OPTS=device_lm=0 $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $O/ab_step.log
done
python scripts/time_inner.py C2 3 2>&1 | tail -1 | tee $O/inner_c2.log
timeout 300 python scripts/time_setup.py C2 2>&1 | grep "^run" | tee $O/setup_c2.log

#!/bin/bash
# round 5, second GPU call: the whole GPU suite (exchange rewrite, owner-computes sweeps, out-of-order measurements, C5 entries),
# step A/B after the decide-kernel change
TAG=${1:-r05b}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra"
for rep in 1 2; do
  echo "cur:      $($B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $O/ab_step.log
  echo "r04 lib:  $(OICC_DEV_LIB=$R/scratch_bin/liboicc_r04.so $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $O/ab_step.log
done
python scripts/time_inner.py C2 3 2>&1 | tail -2 | tee $O/inner.log

#!/bin/bash
# round 5: what the driver runs at round end, as it runs it: the GPU suite sequentially, smoke(), the default bench line;
# plus the one-rank native RCCL path of bench.py and bench.py --gpus 1 through its own argument handling
TAG=${1:-r05f}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_seq.log 2>&1; tail -6 $O/pytest_seq.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -5 $O/smoke.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; tail -3 $O/bench.err
OICC_BENCH_FORCE_ALLREDUCE=1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_force_allreduce.json 2> $O/bench_force_allreduce.err; python -c "
import json; d=json.load(open('$O/bench_force_allreduce.json')); print('one-rank RCCL:', d['ms_per_step'], d['value'], d['config']['allreduce'], d['config']['assembly'])"; tail -2 $O/bench_force_allreduce.err

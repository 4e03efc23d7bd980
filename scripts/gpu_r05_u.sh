#!/bin/bash
# round 5, last call: the GPU suite, smoke and the default bench line on the final state
TAG=${1:-r05u}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -q -n 3 --timeout 400 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json

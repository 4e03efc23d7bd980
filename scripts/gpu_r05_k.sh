#!/bin/bash
# round 5: advance kernel with parallel row sums; the general wave build at two waves per SIMD (OICC_WAVE_DENSE) against one
TAG=${1:-r05k}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 900 -k "wave or shared_blocks" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
OICC_WAVE_DENSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 900 -k "wave" > $O/pytest_dense.log 2>&1; echo "pytest rc $?" >> $O/pytest_dense.log; tail -3 $O/pytest_dense.log
timeout 600 python scripts/time_wave.py C5 2 > $O/wave.log 2>&1; cat $O/wave.log
OICC_WAVE_DENSE=1 timeout 600 python scripts/time_wave.py C5 2 > $O/wave_dense.log 2>&1; cat $O/wave_dense.log

#!/bin/bash
# round 5: wave kernel with ONE block per workgroup (default) against four / eight (OICC_WAVE_GROUPED=1)
TAG=${1:-r05t}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 600 -k "wave" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
echo "one block per workgroup:" | tee $O/wave.log
timeout 300 python scripts/time_wave.py C5 2 2>&1 | tail -1 | tee -a $O/wave.log
echo "four / eight blocks per workgroup:" | tee -a $O/wave.log
OICC_WAVE_GROUPED=1 timeout 300 python scripts/time_wave.py C5 2 2>&1 | tail -1 | tee -a $O/wave.log

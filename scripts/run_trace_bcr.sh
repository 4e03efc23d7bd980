export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_bcr -o run -- python $R/scripts/time_solver.py C2 2 > /dev/null 2>&1
cd $R; python scripts/trace_gaps.py gpurun_out/tr_bcr | head -${2:-24}
python scripts/prof_solver.py 2 | tail -1
python scripts/time_solver.py ${1:-C5} 2 2>&1 | tail -1

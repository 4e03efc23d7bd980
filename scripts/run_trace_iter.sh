export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_iter -o run -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $R/gpurun_out/tr_iter_bench.json 2>/dev/null
cd $R; python scripts/trace_iteration.py gpurun_out/tr_iter
python -c "
import json; d=json.load(open('gpurun_out/tr_iter_bench.json')); print(d['value'], d['ms_per_step'])"
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-profiler:', d['value'], d['ms_per_step'], d['full_calibration'])"

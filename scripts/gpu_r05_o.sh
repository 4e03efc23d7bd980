#!/bin/bash
# round 5: branch-free half-angle sincos in the SO(3) forward pass -- the GPU suite, the default bench line, C5 / C2 sweeps
TAG=${1:-r05o}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - "$O" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
x = d["extra_c5_single_gpu"]
print("C2 step %.4f ms, %.2f M blocks/s, roofline frac %.4f, blocks pass %.4f ms, solve %.4f ms" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"], d["roofline"]["step_share"]["blocks_ms"], d["roofline"]["step_share"]["solve_ms"]))
print("C2 full calibration %.3f ms (sweeps %.3f), plain %.3f ms" % (1e3 * d["full_calibration"]["seconds"], 1e3 * d["full_calibration"]["seconds_inner"], 1e3 * d["full_calibration"]["plain_lm"]["seconds"]))
print("C5 pass %.4f ms (frac %.4f), LM step %.3f ms, sweep %.3f ms, full calibration %.2f ms (set-up %.2f)" % (x["jacobian_pass_ms"], x["fp64_frac_of_78p6"], x["lm_step_ms"], x["inner_sweep_ms"], 1e3 * x["full_calibration_reference_options"]["seconds"], 1e3 * x["full_calibration_reference_options"]["seconds_setup"]))
PY
timeout 600 python scripts/time_wave.py C5 2 2>&1 | tail -1

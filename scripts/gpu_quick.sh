#!/bin/bash
# quick GPU check while tuning the tile kernel: assembly parity tests, tile phase profile, pass timings of C2 / C5
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "assembly_modes or tiles_ or blocks_match or normal_equations or interior_rows" 2>&1 | tail -4
python scripts/prof_tile.py C5 2>&1 | tail -2
python scripts/prof_tile.py C2 2>&1 | tail -2
python scripts/prof_pass.py C5 20 2>&1 | tail -1
python scripts/prof_pass.py C2 50 2>&1 | tail -1

#!/bin/bash
# A/B of the tile phase clocks (prof_tile.py) for library builds on ONE box: bash scripts/ab_tile.sh CFG NAME [NAME ...]
CFG=$1; shift
for n in "$@"; do
  if [ "$n" = cur ]; then unset OICC_DEV_LIB; else export OICC_DEV_LIB=$PWD/scratch_bin/liboicc_$n.so; fi
  echo "$n: $(python scripts/prof_tile.py $CFG 2>&1 | tail -1)"
done

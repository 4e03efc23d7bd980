#!/bin/bash
# A/B timing of library builds on ONE box: bash scripts/ab_pass.sh NAME [NAME ...]   (scratch_bin/liboicc_NAME.so; "cur" = the in-tree build)
for rep in 1 2; do for n in "$@"; do
  if [ "$n" = cur ]; then unset OICC_DEV_LIB; else export OICC_DEV_LIB=$PWD/scratch_bin/liboicc_$n.so; fi
  echo "$n: $(python scripts/prof_pass.py C5 20 2>&1 | tail -1) | $(python scripts/prof_pass.py C2 50 2>&1 | tail -1)"
done; done

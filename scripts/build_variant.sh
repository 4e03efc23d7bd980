#!/bin/bash
# Build a second copy of the product library for A/B timing on one GPU box: bash scripts/build_variant.sh NAME [GITREF]
#   no GITREF: the working tree's csrc; GITREF: that commit's.  Result: scratch_bin/liboicc_NAME.so (select with OICC_DEV_LIB=...)
set -e
NAME=$1; REF=$2; EXTRA=$3;   # EXTRA: appended to HIPFLAGS (e.g. -DOICC_TILE_BODY_ATTR=__forceinline__)
 R=$(cd $(dirname $0)/.. && pwd); D=/tmp/oicc_variant_$NAME
rm -rf $D; mkdir -p $D
if [ -n "$REF" ]; then (cd $R && git archive $REF openimucameracalibrator_amd/csrc include) | tar -x -C $D
else mkdir -p $D/openimucameracalibrator_amd && cp -r $R/openimucameracalibrator_amd/csrc $D/openimucameracalibrator_amd/ && cp -r $R/include $D/; fi
(cd $D/openimucameracalibrator_amd/csrc && rm -f *.o *.so && make -j8 liboicc_hip.so ${EXTRA:+HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result $EXTRA"} > $D/build.log 2>&1) || { tail -20 $D/build.log; exit 1; }
mkdir -p $R/scratch_bin && cp $D/openimucameracalibrator_amd/csrc/liboicc_hip.so $R/scratch_bin/liboicc_$NAME.so && echo built scratch_bin/liboicc_$NAME.so

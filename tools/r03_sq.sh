#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r03w}
mkdir -p $ROOT/gpurun_out/$TAG
for cfg in ${COUNTERS:-c2:exact c2:fast c3:exact c3:fast c4:exact c4:fast}; do
  w=${cfg%%:*}; m=${cfg##*:}
  TAG=$TAG/sq_${w}_$m WORKLOAD=$w MATH=$m bash $ROOT/tools/r03_counters.sh > $ROOT/gpurun_out/$TAG/${TAG}_sq_${w}_$m.txt 2>&1
  echo "== $cfg"; cat $ROOT/gpurun_out/$TAG/${TAG}_sq_${w}_$m.txt | cut -c1-140
done

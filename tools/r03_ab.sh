#!/bin/bash
# A/B of prebuilt libraries on one box: bench lines only, alternating
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for rep in 1 2; do
for v in ${VARIANTS:-default sleep1 sleep4}; do
  lib=$ROOT/build/libmppi_$v.so
  [ "$v" = default ] && lib=$ROOT/mppi_numba_amd/libmppi_hip.so
  MPPI_HIP_LIB=$lib timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step']*1e3,2), round(d['kernel_us_in_loop']['rollout'],2), round(d['kernel_us_in_loop']['update'],2))"
done
done

#!/bin/bash
# A/B of prebuilt library variants (build/libmppi_<name>.so [+ _stamps]): bench line + stamp timeline each
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03var}
mkdir -p $OUT
cd $ROOT
for v in ${VARIANTS:-default map2}; do
  lib=$ROOT/build/libmppi_$v.so; st=$ROOT/build/libmppi_${v}_stamps.so
  [ "$v" = default ] && lib=$ROOT/mppi_numba_amd/libmppi_hip.so && st=$ROOT/build/libmppi_stamps.so
  for rep in 1 2; do
  MPPI_HIP_LIB=$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --math ${MATH:-exact} > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"])
except Exception as e:
    print("$v no json", e); print(open("$OUT/bench_$v.err").read()[-800:])
PY
  done
  MPPI_HIP_LIB=$st timeout 300 python tools/scan_stamps.py --flags 0 --math ${MATH:-exact} > $OUT/stamps_$v.txt 2>&1
  head -19 $OUT/stamps_$v.txt
done

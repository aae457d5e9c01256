#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03q}
mkdir -p $OUT
cd $ROOT
for f in ${FLAGS:-0}; do
  timeout 300 python bench.py --math fast --steps 200 --warmup 20 --no-cpu-baseline --debug-flags $f > $OUT/bench_fast_$f.json 2> $OUT/bench_fast_$f.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_fast_$f.json").read().strip().splitlines()[-1])
    print("fast flags $f", d["ms_per_step"], d["kernel_us_in_loop"]["rollout"], d["kernel_us_in_loop"]["update"], d["config"]["rollout_kernel"])
except Exception as e:
    print("no json", e)
PY
done
make -C mppi_numba_amd/csrc stamps > $OUT/stamps_build.log 2>&1
MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so timeout 300 python tools/scan_stamps.py --flags 0 > $OUT/stamps_0.txt 2>&1
cat $OUT/stamps_0.txt

#!/bin/bash
# first GPU visit of round 3: the scan kernel's tests, bench lines of its variants, stamps
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG:-r03b}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_scan.py -q -s -p no:cacheprovider > $OUT/pytest_scan.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_scan.log
tail -5 $OUT/pytest_scan.log
for f in 0 256 64 128; do
  timeout 300 python bench.py --math fast --steps 200 --warmup 20 --no-cpu-baseline --debug-flags $f > $OUT/bench_fast_$f.json 2> $OUT/bench_fast_$f.err
  echo "flags $f rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_fast_$f.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["kernel_us_in_loop"], d["config"]["rollout_kernel"])
except Exception as e:
    print("no json", e)
PY
done
make -C mppi_numba_amd/csrc stamps > $OUT/stamps_build.log 2>&1
for f in 0 256; do
  MPPI_HIP_LIB=$ROOT/build/libmppi_stamps.so timeout 300 python tools/scan_stamps.py --flags $f > $OUT/stamps_$f.txt 2>&1
done
cat $OUT/stamps_0.txt
